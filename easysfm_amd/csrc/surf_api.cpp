// C-ABI entry point for SURF detection + description (include/esfm.h, SURVEY.md section 8 row f-2, SURF half): the replacement
// for cv::xfeatures2d::SURF::create(minHessian)->detect + SURF::create()->compute at reference
// cpp_code/src/feature_matching.cpp:43-58.  Host side: the pyramid geometry and the scaled box-filter patterns (a few dozen
// integers per layer), the two Gaussian weight tables (host exp(), as OpenCV's getGaussianKernel), the sort of the detected
// maxima by OpenCV's KeypointGreater order, and the per-keypoint window scratch layout.  All pixel work runs in surf_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "surf_kernels.hpp"

using esfm::SurfDescTables;
using esfm::SurfHF;
using esfm::SurfKeypoint;
using esfm::SurfLayer;
using esfm::SurfParams;

namespace {

int cv_round_f(float v) { return (int)std::lrintf(v); }

// surf.cpp resizeHaarPattern
void resize_haar(const int src[][5], SurfHF *dst, int n, int old_size, int new_size, int width_step)
{
    const float ratio = (float)new_size / old_size;
    for (int k = 0; k < n; ++k) {
        const int dx1 = cv_round_f(ratio * src[k][0]), dy1 = cv_round_f(ratio * src[k][1]);
        const int dx2 = cv_round_f(ratio * src[k][2]), dy2 = cv_round_f(ratio * src[k][3]);
        dst[k].p0 = dy1 * width_step + dx1; dst[k].p1 = dy2 * width_step + dx1;
        dst[k].p2 = dy1 * width_step + dx2; dst[k].p3 = dy2 * width_step + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

// getGaussianKernel(n, sigma, CV_32F), sigma > 0
void gaussian_kernel(int n, double sigma, float *out)
{
    double sum = 0, tmp[32];
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; tmp[i] = std::exp(scale2x * x * x); sum += tmp[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) out[i] = (float)(tmp[i] * sum);
}

// cv::KeypointGreater as fastHessianDetector sorts with it
bool kp_greater(const SurfKeypoint &a, const SurfKeypoint &b)
{
    if (a.response != b.response) return a.response > b.response;
    if (a.size != b.size) return a.size > b.size;
    if (a.octave != b.octave) return a.octave > b.octave;
    if (a.y != b.y) return a.y > b.y;
    return a.x < b.x;
}

}  // namespace

extern "C" {

int esfm_surf_detect_and_compute(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, double hessian_threshold,
                                 int max_keypoints, float *keypoints, float *descriptors, int32_t *n_keypoints)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(image && n_keypoints, "NULL argument");
    ESFM_REQUIRE(rows > 0 && cols > 0 && (channels == 1 || channels == 3), "image must be rows x cols x {1, 3}");
    ESFM_REQUIRE(max_keypoints >= 0 && (max_keypoints == 0 || (keypoints && descriptors)), "output buffers");
    ESFM_REQUIRE((int64_t)(rows + 1) * (cols + 1) * 255 < ((int64_t)1 << 31), "image too large for the 32-bit integral image");
    *n_keypoints = 0;
    if (int rc = esfm::set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const int sr = rows + 1, sc = cols + 1;

    // ---- pyramid geometry (fastHessianDetector / calcLayerDetAndTrace)
    SurfParams P;
    memset(&P, 0, sizeof(P));
    P.rows = rows; P.cols = cols; P.hessian_threshold = (float)hessian_threshold;
    static const int dx_s[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
    static const int dy_s[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
    static const int dxy_s[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
    int64_t total = 0;
    {
        int step = 1, index = 0;
        for (int oct = 0; oct < esfm::kSurfOctaves; ++oct) {
            for (int layer = 0; layer < esfm::kSurfOctaveLayers + 2; ++layer, ++index) {
                SurfLayer &L = P.layer[index];
                L.size = (9 + 6 * layer) << oct; L.step = step;
                L.rows = (sr - 1) / step; L.cols = (sc - 1) / step;
                L.offset = (int32_t)total;
                total += (int64_t)std::max(L.rows, 1) * std::max(L.cols, 1);
                L.valid = !(L.size > sr - 1 || L.size > sc - 1);
                if (L.valid) {
                    resize_haar(dx_s, L.dx, 3, 9, L.size, sc); resize_haar(dy_s, L.dy, 3, 9, L.size, sc); resize_haar(dxy_s, L.dxy, 4, 9, L.size, sc);
                    L.samples_i = 1 + (sr - 1 - L.size) / step; L.samples_j = 1 + (sc - 1 - L.size) / step; L.margin = (L.size / 2) / step;
                }
            }
            step *= 2;
        }
    }
    P.max_candidates = rows * cols / 4 + 1024;   // a maximum needs a strict 3 x 3 neighbourhood: at most a quarter of the samples

    // ---- device buffers
    const size_t n_px = (size_t)rows * cols;
    esfm::DevBuf &b_img = ctx->stage_a, &b_sum = ctx->stage_b, &b_det = ctx->stage_c, &b_misc = ctx->stage_d, &b_win = ctx->stage_e;
    if (int rc = b_img.reserve(n_px * (channels == 3 ? 4 : 1) + 16)) return rc;
    if (int rc = b_sum.reserve(sizeof(int32_t) * (size_t)sr * sc)) return rc;
    if (int rc = b_det.reserve(sizeof(float) * 2 * (size_t)total)) return rc;
    const size_t cand_bytes = sizeof(SurfKeypoint) * (size_t)P.max_candidates;
    if (int rc = b_misc.reserve(sizeof(SurfParams) + sizeof(SurfDescTables) + 64 + cand_bytes)) return rc;
    uint8_t *d_gray = b_img.as<uint8_t>();
    uint8_t *d_bgr = d_gray + ((n_px + 15) / 16) * 16;
    int32_t *d_sum = b_sum.as<int32_t>();
    float *d_det = b_det.as<float>(), *d_trace = d_det + total;
    uint8_t *misc = b_misc.as<uint8_t>();
    SurfParams *d_P = reinterpret_cast<SurfParams *>(misc);
    SurfDescTables *d_T = reinterpret_cast<SurfDescTables *>(misc + sizeof(SurfParams));
    int32_t *d_ncand = reinterpret_cast<int32_t *>(misc + sizeof(SurfParams) + sizeof(SurfDescTables));
    SurfKeypoint *d_cand = reinterpret_cast<SurfKeypoint *>(misc + sizeof(SurfParams) + sizeof(SurfDescTables) + 64);

    if (channels == 3) {
        ESFM_HIP_TRY(esfm::copy_h2d(d_bgr, image, n_px * 3, st));
        if (int rc = esfm::launch_surf_gray(st, d_bgr, (int)n_px, d_gray)) return rc;
    } else {
        ESFM_HIP_TRY(esfm::copy_h2d(d_gray, image, n_px, st));
    }
    ESFM_HIP_TRY(esfm::copy_h2d(d_P, &P, sizeof(P), st));
    ESFM_HIP_TRY(hipMemsetAsync(d_det, 0, sizeof(float) * 2 * (size_t)total, st));
    ESFM_HIP_TRY(hipMemsetAsync(d_ncand, 0, 64, st));
    if (int rc = esfm::launch_surf_integral(st, d_gray, rows, cols, d_sum)) return rc;
    if (int rc = esfm::launch_surf_det_trace(st, d_P, P, d_sum, d_det, d_trace, ctx)) return rc;
    if (int rc = esfm::launch_surf_maxima(st, d_P, P, d_det, d_trace, d_cand, d_ncand)) return rc;
    // the candidate count and the first candidates in ONE transfer (the count's 64-byte slot sits right in front of the list): the
    // usual image needs no second round trip for the list
    constexpr int kFirstCand = 4096;
    static_assert(sizeof(SurfKeypoint) == 32, "candidate records are 32 bytes");
    const int first_cap = std::min(kFirstCand, P.max_candidates);
    std::vector<uint8_t> head(64 + sizeof(SurfKeypoint) * (size_t)first_cap);
    ESFM_HIP_TRY(esfm::copy_d2h(head.data(), d_ncand, head.size(), st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    int32_t n_cand = 0;
    memcpy(&n_cand, head.data(), sizeof(int32_t));
    if (n_cand > P.max_candidates) { esfm::set_error("SURF candidate buffer overflow (%d > %d)", n_cand, P.max_candidates); return ESFM_ERR_NUMERIC; }
    if (n_cand == 0 || max_keypoints == 0) return ESFM_OK;
    std::vector<SurfKeypoint> kps((size_t)n_cand);
    memcpy(kps.data(), head.data() + 64, sizeof(SurfKeypoint) * (size_t)std::min(n_cand, first_cap));
    if (n_cand > first_cap) {
        ESFM_HIP_TRY(esfm::copy_d2h(kps.data() + first_cap, d_cand + first_cap, sizeof(SurfKeypoint) * (size_t)(n_cand - first_cap), st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
    }
    std::sort(kps.begin(), kps.end(), kp_greater);   // the device appends in no particular order; OpenCV sorts too

    // ---- descriptor tables and window scratch
    SurfDescTables T;
    memset(&T, 0, sizeof(T));
    {
        float G_ori[2 * esfm::kSurfOriRadius + 1], G_desc[esfm::kSurfPatch];
        gaussian_kernel(2 * esfm::kSurfOriRadius + 1, 2.5, G_ori);
        gaussian_kernel(esfm::kSurfPatch, 3.3, G_desc);
        const int R = esfm::kSurfOriRadius;
        for (int i = -R; i <= R; ++i) for (int j = -R; j <= R; ++j)
            if (i * i + j * j <= R * R) { T.aptx[T.n_ori] = i; T.apty[T.n_ori] = j; T.aptw[T.n_ori++] = G_ori[i + R] * G_ori[j + R]; }
        for (int i = 0; i < esfm::kSurfPatch; ++i) for (int j = 0; j < esfm::kSurfPatch; ++j) T.DW[i * esfm::kSurfPatch + j] = G_desc[i] * G_desc[j];
    }
    const int n_kp = n_cand;
    std::vector<int64_t> win_off((size_t)n_kp + 1, 0);
    std::vector<int32_t> win_of((size_t)n_kp);
    for (int k = 0; k < n_kp; ++k) {
        const float s = kps[(size_t)k].size * 1.2f / 9.0f;
        const int64_t w = (int64_t)((esfm::kSurfPatch + 1) * s);
        ESFM_REQUIRE(w < 1024, "keypoint scale beyond the descriptor window the kernels are built for");
        win_of[(size_t)k] = (int32_t)w;
        win_off[(size_t)k + 1] = win_off[(size_t)k] + (int64_t)esfm::surf_scratch_bytes((int)w);
    }
    // block tables of the two per-sample stages (256 items of ONE keypoint per block), widest windows first: their chunks are the
    // longest items, so they start first
    std::vector<int32_t> order((size_t)n_kp);
    for (int k = 0; k < n_kp; ++k) order[(size_t)k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return win_of[(size_t)a] > win_of[(size_t)b]; });
    std::vector<esfm::SurfBlk> blk_win, blk_row;
    for (int32_t k : order) {
        const int w = win_of[(size_t)k];
        const int n_win = w * esfm::surf_window_chunks(w), n_row = w * (esfm::kSurfPatch + 1);
        for (int f = 0; f < n_win; f += 256) blk_win.push_back({k, f});
        for (int f = 0; f < n_row; f += 256) blk_row.push_back({k, f});
    }
    // device layout behind the window scratch: descriptors | tables (Gaussian tables, scratch offsets, the two block tables) -- the
    // tables are packed on the host in the same layout and go up in one transfer
    const size_t off_bytes = sizeof(int64_t) * ((size_t)n_kp + 1), desc_bytes = sizeof(float) * 64 * (size_t)n_kp;
    const size_t t_bytes = esfm::surf_align16(sizeof(SurfDescTables)), o_bytes = esfm::surf_align16(off_bytes);
    const size_t bw_bytes = esfm::surf_align16(sizeof(esfm::SurfBlk) * blk_win.size()), br_bytes = esfm::surf_align16(sizeof(esfm::SurfBlk) * blk_row.size());
    const size_t tab_bytes = t_bytes + o_bytes + bw_bytes + br_bytes;
    if (int rc = b_win.reserve((size_t)win_off[(size_t)n_kp] + desc_bytes + tab_bytes + 128)) return rc;
    uint8_t *d_win = b_win.as<uint8_t>();
    float *d_desc = reinterpret_cast<float *>(d_win + esfm::surf_align16((size_t)win_off[(size_t)n_kp]));
    uint8_t *d_tab = reinterpret_cast<uint8_t *>(d_desc) + esfm::surf_align16(desc_bytes);
    std::vector<uint8_t> tab(tab_bytes);
    memcpy(tab.data(), &T, sizeof(T));
    memcpy(tab.data() + t_bytes, win_off.data(), off_bytes);
    if (!blk_win.empty()) memcpy(tab.data() + t_bytes + o_bytes, blk_win.data(), sizeof(esfm::SurfBlk) * blk_win.size());
    if (!blk_row.empty()) memcpy(tab.data() + t_bytes + o_bytes + bw_bytes, blk_row.data(), sizeof(esfm::SurfBlk) * blk_row.size());
    d_T = reinterpret_cast<SurfDescTables *>(d_tab);
    const int64_t *d_off = reinterpret_cast<const int64_t *>(d_tab + t_bytes);
    const esfm::SurfBlk *d_bw = reinterpret_cast<const esfm::SurfBlk *>(d_tab + t_bytes + o_bytes);
    const esfm::SurfBlk *d_br = reinterpret_cast<const esfm::SurfBlk *>(d_tab + t_bytes + o_bytes + bw_bytes);
    ESFM_HIP_TRY(esfm::copy_h2d(d_tab, tab.data(), tab_bytes, st));
    ESFM_HIP_TRY(esfm::copy_h2d(d_cand, kps.data(), sizeof(SurfKeypoint) * (size_t)n_kp, st));
    if (int rc = esfm::launch_surf_describe(st, d_P, d_T, d_gray, d_sum, d_cand, n_kp, d_off, d_bw, (int)blk_win.size(), d_br, (int)blk_row.size(), d_win,
                                            d_desc, ctx)) return rc;
    std::vector<float> desc(64 * (size_t)n_kp);
    ESFM_HIP_TRY(esfm::copy_d2h(kps.data(), d_cand, sizeof(SurfKeypoint) * (size_t)n_kp, st));
    ESFM_HIP_TRY(esfm::copy_d2h(desc.data(), d_desc, desc_bytes, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    int n_out = 0;
    for (int k = 0; k < n_kp && n_out < max_keypoints; ++k) {
        const SurfKeypoint &kp = kps[(size_t)k];
        if (!kp.valid) continue;   // SURFInvoker marks these with size = -1 and detectAndCompute drops them
        float *ko = keypoints + 7 * (size_t)n_out;
        ko[0] = kp.x; ko[1] = kp.y; ko[2] = kp.size; ko[3] = kp.angle; ko[4] = kp.response; ko[5] = (float)kp.octave; ko[6] = (float)kp.class_id;
        memcpy(descriptors + 64 * (size_t)n_out, desc.data() + 64 * (size_t)k, sizeof(float) * 64);
        ++n_out;
    }
    *n_keypoints = n_out;
    return ESFM_OK;
}

}  // extern "C"
