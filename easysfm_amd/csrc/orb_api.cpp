// C-ABI entry point for ORB detection + description (include/esfm.h, SURVEY.md section 8 row f-2, ORB half): the replacement for
// cv::ORB::create(max_num)->detect + ->compute at reference cpp_code/src/feature_matching.cpp:14-41 (feature type 'O').
// Host side: the pyramid geometry, the per-level feature quota, the two retainBest selections (a sort of a few thousand
// candidates, as OpenCV's KeyPointsFilter does on the host), the integer tables (disc half-widths, Gaussian weights, the test
// point pairs) and cos / sin of the keypoint angles through the host's libm.  All pixel work runs in orb_kernels.hip.
// The test point pairs are NOT OpenCV's learned bit_pattern_31_ (it ships only inside OpenCV): orb_pattern() below is this
// repo's documented generator -- descriptors are ORB descriptors in kind, not bit-compatible with OpenCV's.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "orb_kernels.hpp"
#include "surf_kernels.hpp"   // launch_surf_gray: cvtColor BGR2GRAY

using esfm::OrbCand;
using esfm::OrbKp;
using esfm::OrbLevels;
using esfm::OrbTables;

namespace {

int cv_round_f(float v) { return (int)std::lrintf(v); }
int cv_round_d(double v) { return (int)std::lrint(v); }

// 256 test point pairs (x0, y0, x1, y1): sum of four uniforms scaled to sigma 6.2, rounded, clipped to [-13, 13]; a pair whose two
// points coincide is redrawn.  LCG x <- 1664525 x + 1013904223 (mod 2^32), seed 31.
void orb_pattern(int8_t *out /* 1024 */)
{
    uint32_t s = 31u;
    int n = 0;
    while (n < 256) {
        int v[4];
        for (int k = 0; k < 4; ++k) {
            double acc = 0.0;
            for (int u = 0; u < 4; ++u) { s = s * 1664525u + 1013904223u; acc += (double)(s >> 8) / 16777216.0 - 0.5; }
            int q = cv_round_d(acc * (6.2 / 0.57735026918962576));
            q = std::min(13, std::max(-13, q));
            v[k] = q;
        }
        if (v[0] == v[2] && v[1] == v[3]) continue;
        for (int k = 0; k < 4; ++k) out[4 * n + k] = (int8_t)v[k];
        ++n;
    }
}

void make_tables(OrbTables *T)
{
    memset(T, 0, sizeof(*T));
    const int hp = esfm::kOrbHalfPatch;
    const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) T->umax[v] = cv_round_d(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (T->umax[v0] == T->umax[v0 + 1]) ++v0;
        T->umax[v] = v0;
        ++v0;
    }
    double g[7], sum = 0;
    for (int i = 0; i < 7; ++i) { const double x = i - 3; g[i] = std::exp(-0.5 * x * x / 4.0); sum += g[i]; }
    int tot = 0;
    for (int i = 0; i < 7; ++i) { T->gauss[i] = cv_round_d(256.0 * g[i] / sum); tot += T->gauss[i]; }
    T->gauss[3] += 256 - tot;
    orb_pattern(T->pattern);
}

bool cand_before(const OrbCand &p, const OrbCand &q)
{
    if (p.resp != q.resp) return p.resp > q.resp;
    if (p.y != q.y) return p.y < q.y;
    return p.x < q.x;
}
// KeyPointsFilter::retainBest on a sorted list: the first n and every later one that ties with the n-th
size_t retain_best(const std::vector<OrbCand> &c, int n)
{
    if (n >= (int)c.size()) return c.size();
    if (n <= 0) return 0;
    const float amb = c[(size_t)n - 1].resp;
    size_t k = (size_t)n;
    while (k < c.size() && c[k].resp >= amb) ++k;
    return k;
}

}  // namespace

extern "C" {

int esfm_orb_detect_and_compute(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, int nfeatures, int max_keypoints,
                                float *keypoints, uint8_t *descriptors, int32_t *n_keypoints)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(image && n_keypoints, "NULL argument");
    ESFM_REQUIRE(rows > 0 && cols > 0 && (channels == 1 || channels == 3), "image must be rows x cols x {1, 3}");
    ESFM_REQUIRE(nfeatures >= 0 && max_keypoints >= 0 && (max_keypoints == 0 || (keypoints && descriptors)), "output buffers");
    ESFM_REQUIRE((int64_t)rows * cols < ((int64_t)1 << 28), "image too large");
    *n_keypoints = 0;
    if (int rc = esfm::set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    constexpr int NL = esfm::kOrbLevels;

    OrbLevels L;
    float scale[NL];
    int quota[NL];
    {
        int64_t off = 0;
        for (int l = 0; l < NL; ++l) {
            scale[l] = (float)std::pow(1.2, (double)l);
            L.cols[l] = std::max(1, cv_round_f(cols / scale[l])); L.rows[l] = std::max(1, cv_round_f(rows / scale[l]));
            L.offset[l] = off;
            off += (((int64_t)L.rows[l] * L.cols[l] + 63) / 64) * 64;
        }
        L.total = off;
        const float factor = 1.f / 1.2f;
        float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)NL));
        int sum = 0;
        for (int l = 0; l < NL - 1; ++l) { quota[l] = cv_round_f(nd); sum += quota[l]; nd *= factor; }
        quota[NL - 1] = std::max(nfeatures - sum, 0);
    }
    const size_t n_px = (size_t)rows * cols;
    const int cap = (int)std::min<int64_t>(L.total / 4 + 1024, 1 << 24);
    esfm::DevBuf &b_img = ctx->stage_a, &b_pyr = ctx->stage_b, &b_blur = ctx->stage_c, &b_score = ctx->stage_d, &b_misc = ctx->stage_e;
    if (int rc = b_img.reserve(n_px * 3 + 16)) return rc;
    if (int rc = b_pyr.reserve((size_t)L.total + 64)) return rc;
    if (int rc = b_blur.reserve((size_t)L.total + 64)) return rc;
    if (int rc = b_score.reserve((size_t)L.total + 64)) return rc;
    const size_t tab_bytes = ((sizeof(OrbTables) + 63) / 64) * 64;
    if (int rc = b_misc.reserve(tab_bytes + 64 + sizeof(OrbCand) * (size_t)cap)) return rc;
    uint8_t *d_pyr = b_pyr.as<uint8_t>(), *d_blur = b_blur.as<uint8_t>(), *d_score = b_score.as<uint8_t>();
    OrbTables *d_tab = b_misc.as<OrbTables>();
    int32_t *d_ncand = reinterpret_cast<int32_t *>(b_misc.as<uint8_t>() + tab_bytes);
    OrbCand *d_cand = reinterpret_cast<OrbCand *>(b_misc.as<uint8_t>() + tab_bytes + 64);

    OrbTables T;
    make_tables(&T);
    ESFM_HIP_TRY(esfm::copy_h2d(d_tab, &T, sizeof(T), st));
    ESFM_HIP_TRY(hipMemsetAsync(d_ncand, 0, 64, st));
    if (channels == 3) {
        ESFM_HIP_TRY(esfm::copy_h2d(b_img.ptr, image, n_px * 3, st));
        if (int rc = esfm::launch_surf_gray(st, b_img.as<uint8_t>(), (int)n_px, d_pyr)) return rc;
    } else {
        ESFM_HIP_TRY(esfm::copy_h2d(d_pyr, image, n_px, st));
    }
    for (int l = 1; l < NL; ++l)
        if (int rc = esfm::launch_orb_resize(st, d_pyr + L.offset[l - 1], L.rows[l - 1], L.cols[l - 1], d_pyr + L.offset[l], L.rows[l], L.cols[l])) return rc;
    for (int l = 0; l < NL; ++l)
        if (int rc = esfm::launch_orb_blur(st, d_tab, d_pyr + L.offset[l], L.rows[l], L.cols[l], d_blur + L.offset[l])) return rc;
    if (int rc = esfm::launch_orb_fast(st, L, d_pyr, d_score, ctx)) return rc;
    if (int rc = esfm::launch_orb_nms(st, L, d_score, d_cand, d_ncand, cap)) return rc;
    int32_t n_cand = 0;
    ESFM_HIP_TRY(esfm::copy_d2h(&n_cand, d_ncand, sizeof(int32_t), st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    if (n_cand > cap) { esfm::set_error("ORB candidate buffer overflow (%d > %d)", n_cand, cap); return ESFM_ERR_NUMERIC; }
    if (n_cand == 0 || max_keypoints == 0) return ESFM_OK;
    std::vector<OrbCand> all((size_t)n_cand);
    ESFM_HIP_TRY(esfm::copy_d2h(all.data(), d_cand, sizeof(OrbCand) * (size_t)n_cand, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));

    // first selection: per level, the best 2 n_l by FAST score (the device appended in no particular order)
    std::vector<OrbCand> lvl[NL], sel;
    for (const OrbCand &c : all) lvl[c.level].push_back(c);
    std::vector<int> count1(NL, 0);
    for (int l = 0; l < NL; ++l) {
        std::sort(lvl[l].begin(), lvl[l].end(), cand_before);
        lvl[l].resize(retain_best(lvl[l], 2 * quota[l]));
        count1[(size_t)l] = (int)lvl[l].size();
        sel.insert(sel.end(), lvl[l].begin(), lvl[l].end());
    }
    if (sel.empty()) return ESFM_OK;
    ESFM_HIP_TRY(esfm::copy_h2d(d_cand, sel.data(), sizeof(OrbCand) * sel.size(), st));
    if (int rc = esfm::launch_orb_harris(st, L, d_pyr, d_cand, (int)sel.size())) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(sel.data(), d_cand, sizeof(OrbCand) * sel.size(), st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    // second selection: per level, the best n_l by Harris response
    std::vector<OrbCand> fin;
    {
        size_t o = 0;
        for (int l = 0; l < NL; ++l) {
            std::vector<OrbCand> v(sel.begin() + (long)o, sel.begin() + (long)(o + (size_t)count1[(size_t)l]));
            o += (size_t)count1[(size_t)l];
            std::sort(v.begin(), v.end(), cand_before);
            v.resize(retain_best(v, quota[l]));
            fin.insert(fin.end(), v.begin(), v.end());
        }
    }
    if ((int)fin.size() > max_keypoints) fin.resize((size_t)max_keypoints);
    const int n_kp = (int)fin.size();
    if (n_kp == 0) return ESFM_OK;
    // orientation on the device, cos / sin through the host's libm (as the CPU restatement takes them), then the descriptors
    const size_t kp_bytes = sizeof(OrbKp) * (size_t)n_kp, ang_bytes = sizeof(float) * (size_t)n_kp;
    if (int rc = ctx->knn_dist.reserve(kp_bytes + ang_bytes + 32 * (size_t)n_kp + 256)) return rc;
    OrbKp *d_kp = ctx->knn_dist.as<OrbKp>();
    float *d_ang = reinterpret_cast<float *>(ctx->knn_dist.as<uint8_t>() + ((kp_bytes + 63) / 64) * 64);
    uint8_t *d_desc = reinterpret_cast<uint8_t *>(d_ang) + ((ang_bytes + 63) / 64) * 64;
    ESFM_HIP_TRY(esfm::copy_h2d(d_cand, fin.data(), sizeof(OrbCand) * (size_t)n_kp, st));
    if (int rc = esfm::launch_orb_angles(st, L, d_tab, d_pyr, d_cand, n_kp, d_ang)) return rc;
    std::vector<float> ang((size_t)n_kp);
    ESFM_HIP_TRY(esfm::copy_d2h(ang.data(), d_ang, ang_bytes, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    std::vector<OrbKp> kps((size_t)n_kp);
    for (int k = 0; k < n_kp; ++k) {
        const OrbCand &c = fin[(size_t)k];
        const float sc = scale[c.level], inv = 1.f / sc;
        const float px = c.x * sc, py = c.y * sc;
        float *ko = keypoints + 7 * (size_t)k;
        ko[0] = px; ko[1] = py; ko[2] = 31.f * sc; ko[3] = ang[(size_t)k]; ko[4] = c.resp; ko[5] = (float)c.level; ko[6] = -1.f;
        float a = ang[(size_t)k];
        a *= (float)(3.14159265358979323846 / 180.f);
        OrbKp &q = kps[(size_t)k];
        q.cx = cv_round_f(px * inv); q.cy = cv_round_f(py * inv); q.level = c.level; q.pad = 0;
        // the DOUBLE cosine / sine of the float angle, rounded to float, as the C restatement (and OpenCV's `(float)cos(angle)`) take them:
        // std::cos(float) is cosf, which differs from that in the last bit once in ~10^5 angles -- one descriptor bit in 1 760 random
        // images of tests/stress_pixels.py
        q.a = (float)std::cos((double)a); q.b = (float)std::sin((double)a);
    }
    ESFM_HIP_TRY(esfm::copy_h2d(d_kp, kps.data(), kp_bytes, st));
    if (int rc = esfm::launch_orb_describe(st, L, d_tab, d_blur, d_kp, n_kp, d_desc)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(descriptors, d_desc, 32 * (size_t)n_kp, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    *n_keypoints = n_kp;
    return ESFM_OK;
}

}  // extern "C"
