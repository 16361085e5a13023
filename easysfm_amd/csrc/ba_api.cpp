// C-ABI entry points for bundle adjustment (include/esfm.h, rows a-4..a-8 of SURVEY.md section 8).
// The Levenberg-Marquardt control flow restates Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy
// with DENSE_SCHUR, which is what BundleAdjustment::solveBA configures (reference
// cpp_code/src/ba.cpp:146-151, :201-206); the oracle (oracle/ba_ref.c) documents the upstream rules.
// All arithmetic on the observations runs in ba_kernels.hip; this file only sequences kernels, reads
// back a handful of scalars per iteration and takes the accept/reject decision.
#include <algorithm>
#include <chrono>
#include <thread>
#include <climits>
#include <cstdlib>
#include <cmath>
#include <cfloat>
#include <vector>

#include "ba_kernels.hpp"
#include "ba_linesearch.hpp"
#include "ba_sparse_plan.hpp"
#include "ba_chol_sparse.hpp"

using esfm::BADev;

struct esfm_ba_problem {
    esfm_ctx *ctx = nullptr;
    BADev d;
    std::vector<esfm_ctx::BaChunk> allocs;   // the chunks dev_alloc carves the problem's arrays from
    char *arena_cur = nullptr;            // free space of the newest chunk
    size_t arena_left = 0;
    std::vector<double> cam_nobs_local;  // this rank's observation count per camera-side block
    bool params_swapped = false;
    // box bounds (reference ba.cpp:155-162 reference camera, ba.cpp:190-194 intrinsics); +-inf where there is none
    int ref_cam = -1;
    double ref_threshold = 0.0;
    double calib_center[4] = {0, 0, 0, 0}, calib_tol = 0.0;
    esfm::ScalParts parts{};    // per-workgroup scalar partials pending on the device (BADev::parts points here)
    double *h_scal = nullptr;   // pinned host copy of the scalar slots: the LM loop reads them back twice per iteration
    unsigned long long seq = 0; // sequence number of the last publication (the flag sits behind the scalars)
    // structure of the reduced camera system (ba_sparse_plan.hpp): this rank's co-visible camera pairs, from the observation list at
    // creation; the plan and its device tables are built by the first solve that can use them (several ranks: from the union of
    // the ranks' pairs) and kept
    std::vector<uint8_t> pair_flags;
    std::vector<int32_t> h_pt_start, h_obs_cam;
    esfm::SparseSolve *sparse = nullptr;
    int sparse_key = -1;        // what `sparse` was planned for: 0 one rank, 1 several ranks; -1 not planned yet
    int sparse_leaf_max = 0;
    bool sparse_worthwhile = false;
};

namespace {

// upper bound (seconds) on the host's wait for one scalar read-back; see Solver::fetch_scal
long readback_timeout_s()
{
    static const long v = [] {
        const char *e = getenv("ESFM_BA_READBACK_TIMEOUT_S");
        const long t = e ? atol(e) : 0;
        return t > 0 ? t : 600L;
    }();
    return v;
}

// A problem's ~45 device arrays are carved from a few chunks (256-byte aligned, 256 spare bytes behind each array) instead of one
// hipMalloc each: the small problems of an incremental reconstruction -- a BA call every ba_frequency frames, a dozen cameras and a
// few thousand observations -- are set up and torn down once per call (scratch/ba_small_time.py: set-up 0.3 - 0.9 ms, tear-down 0.4 -
// 0.9 ms with one or two chunks; 0.14 / 0.0 ms once the chunks come from and go back to the context, below; the call's 4 - 5 ms are
// its up to 50 LM iterations of 0.075 - 0.087 ms, a launch-latency chain).  An
// array that does not fit the current chunk's rest opens a chunk of its own size (at least kArenaChunk): the large arrays of BA-512
// still get one allocation each.
constexpr size_t kArenaChunk = size_t(4) << 20;
template <class T> int dev_alloc(esfm_ba_problem *p, T **out, size_t count)
{
    const size_t bytes = (sizeof(T) * std::max<size_t>(count, 1) + 255) / 256 * 256 + 256;
    if (bytes > p->arena_left) {
        void *ptr = nullptr;
        size_t chunk = std::max(bytes, kArenaChunk);
        // a chunk the context kept from a destroyed problem, the smallest that fits (esfm_ba_problem_destroy synchronised the stream
        // before it handed the chunk over)
        auto &pool = p->ctx->ba_chunks;
        int best = -1;
        for (int i = 0; i < (int)pool.size(); ++i)
            if (pool[(size_t)i].bytes >= bytes && (best < 0 || pool[(size_t)i].bytes < pool[(size_t)best].bytes)) best = i;
        if (best >= 0) {
            ptr = pool[(size_t)best].ptr; chunk = pool[(size_t)best].bytes;
            pool.erase(pool.begin() + best);
        } else {
            hipError_t e = hipMalloc(&ptr, chunk);
            if (e != hipSuccess) {
                esfm::set_error("hipMalloc(%zu) failed: %s", chunk, hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? ESFM_ERR_OOM : ESFM_ERR_HIP;
            }
        }
        p->allocs.push_back({ptr, chunk});
        p->arena_cur = static_cast<char *>(ptr);
        p->arena_left = chunk;
    }
    *out = reinterpret_cast<T *>(p->arena_cur);
    p->arena_cur += bytes;
    p->arena_left -= bytes;
    return ESFM_OK;
}

void options_default(esfm_ba_options *o)
{
    o->max_num_iterations = 50;  // ba.cpp:202
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->verbose = 0;
    o->cauchy_a = 0.5;  // ba.cpp:150
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
}

struct Solver {
    esfm_ba_problem *P;
    esfm_ba_options opt;
    esfm_allreduce_fn ar;
    void *ar_user;
    hipStream_t st;
    double *h = nullptr;        // P->h_scal

    int allreduce(double *buf, int64_t count, int op)
    {
        if (!ar || count <= 0) return ESFM_OK;
        if (ar(ar_user, buf, count, op, reinterpret_cast<void *>(st)) != 0) {
            esfm::set_error("all-reduce callback failed");
            return ESFM_ERR_COMM;
        }
        return ESFM_OK;
    }
    bool scal_zeroed = false;   // the read-back kernel (ba_publish_scalars) leaves d.scal zeroed: a reset right after a fetch needs no memset
    int zero_scal()
    {
        if (!scal_zeroed) ESFM_HIP_TRY(hipMemsetAsync(P->d.scal, 0, sizeof(double) * esfm::SC_COUNT, st));
        scal_zeroed = false;
        esfm::ba_scal_discard(P->d, 0, esfm::SC_SUM_COUNT);
        return ESFM_OK;
    }
    // SUM the partial-sum slots and MAX the gradient slot across ranks, then fetch all scalars.
    int fetch_scal()
    {
        if (ar) { if (int rc = esfm::ba_scal_reduce(st, P->d)) return rc; }   // this rank's partials -> d.scal before the exchange
        if (int rc = allreduce(P->d.scal, esfm::SC_SUM_COUNT, ESFM_REDUCE_SUM)) return rc;
        if (int rc = allreduce(P->d.scal + esfm::SC_GMAX, esfm::SC_MAX_COUNT, ESFM_REDUCE_MAX)) return rc;
        // The device publishes the slots into pinned memory and then a sequence number; the host spins on that instead of
        // paying a stream synchronisation (two read-backs per LM iteration of ~0.4 ms: the wake-up latency is a tenth of it).
        unsigned long long *flag = reinterpret_cast<unsigned long long *>(h + esfm::SC_COUNT);
        const unsigned long long seq = ++P->seq;
        if (int rc = esfm::ba_publish_scalars(st, P->d, h, flag, seq)) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
            if ((spins & 0xFFF) == 0xFFF) {
                if (hipStreamQuery(st) != hipErrorNotReady) {           // finished (or failed) without the store being seen: settle by sync
                    ESFM_HIP_TRY(hipStreamSynchronize(st));
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                    esfm::set_error("BA scalar publication was not observed");
                    return ESFM_ERR_HIP;
                }
                // A large reduced system (6 n_cam up to 46 000) legitimately keeps the stream busy for seconds; hipStreamQuery above
                // is what detects completion and failure.  After a while stop burning a core -- and past a generous bound (10 min,
                // ESFM_BA_READBACK_TIMEOUT_S overrides) give up with an error instead of spinning forever behind a wedged stream
                // (a dataflow kernel that lost a flag, a stuck collective of a sharded solve).
                const auto waited = std::chrono::steady_clock::now() - t0;
                if (waited > std::chrono::seconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(200));
                if (waited > std::chrono::seconds(readback_timeout_s())) {
                    esfm::set_error("BA scalar read-back timed out (stream never completed)");
                    return ESFM_ERR_HIP;
                }
            }
        }
        scal_zeroed = true;
        return ESFM_OK;
    }
    // residuals + Jacobian at x, per-camera sums, per-point blocks; leaves cost/gmax in h[].
    int linearize(bool use_scaling, double radius, double cost_bound = -1.0)
    {
        const BADev &d = P->d;
        int deferred = 0;      // one rank: the slab reduction of the sweep's per-camera sums rides in the per-point launch
        if (int rc = esfm::ba_linearize(st, d, P->ctx->num_cu, opt.cauchy_a, use_scaling, P->ctx, ar ? nullptr : &deferred, cost_bound)) return rc;
        if (int rc = allreduce(d.camacc, (int64_t)esfm::ba_camacc_doubles(d.n_cam), ESFM_REDUCE_SUM)) return rc;
        if (int rc = esfm::ba_point_prep(st, d, radius, opt.min_lm_diagonal, opt.max_lm_diagonal, true, deferred)) return rc;
        return ESFM_OK;
    }
};

// Structure of the reduced camera system for this solve (see ba_sparse_plan.hpp).  Camera blocks (a, b) of S are non-zero only where
// a and b observe a common point; when that leaves at most half of the dense factorisation's tiles -- or half its dependency chain
// -- the tiled solve visits only the tiles of the symbolic fill (ba_chol_sparse.hip).  Several ranks: every rank holds the
// observations of ITS points, so the ranks' pair sets are united first (one small all-reduce per solve; four 13-bit counters per
// double, exact for up to 8191 ranks), every rank plans from the same union and the plans are identical.  ESFM_BA_SOLVE=dense keeps the
// dense path, =sparse takes the plan even where it does not pay (tests); ESFM_BA_LEAF_MAX: cameras per undissected leaf.
int plan_reduced_structure(esfm_ba_problem *P, Solver &S, bool multi)
{
    BADev &d = P->d;
    d.sparse = nullptr;
    const char *mode = getenv("ESFM_BA_SOLVE");
    const bool force_dense = mode && mode[0] == 'd', force_sparse = mode && mode[0] == 's';
    if (force_dense || d.has_calib || !esfm::ba_solve_is_tiled(d.n_cam) || P->h_pt_start.empty()) return ESFM_OK;
    const char *lm = getenv("ESFM_BA_LEAF_MAX");
    const int leaf_max = lm && atoi(lm) > 0 ? atoi(lm) : 32;
    const int key = multi ? 1 : 0;
    if (P->sparse_key != key || P->sparse_leaf_max != leaf_max) {
        esfm::ba_sparse_destroy(P->sparse); P->sparse = nullptr;
        if (P->pair_flags.empty()) P->pair_flags = esfm::cam_pair_flags(d.n_real_cam, d.n_pt, P->h_pt_start.data(), P->h_obs_cam.data());
        std::vector<uint8_t> all;
        if (multi) {
            const size_t nf = P->pair_flags.size(), nd = (nf + 3) / 4;
            std::vector<double> pk(nd, 0.0);
            for (size_t k = 0; k < nf; ++k) if (P->pair_flags[k]) pk[k / 4] += (double)(1ull << (13 * (k % 4)));
            double *dev = nullptr;
            ESFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(double) * std::max<size_t>(nd, 1)));
            int rc = ESFM_OK;
            if (esfm::copy_h2d(dev, pk.data(), sizeof(double) * nd, S.st) != hipSuccess || hipStreamSynchronize(S.st) != hipSuccess) rc = ESFM_ERR_HIP;
            if (rc == ESFM_OK) rc = S.allreduce(dev, (int64_t)nd, ESFM_REDUCE_SUM);
            if (rc == ESFM_OK && (esfm::copy_d2h(pk.data(), dev, sizeof(double) * nd, S.st) != hipSuccess || hipStreamSynchronize(S.st) != hipSuccess)) rc = ESFM_ERR_HIP;
            (void)hipFree(dev);
            if (rc != ESFM_OK) { if (rc == ESFM_ERR_HIP) esfm::set_error("exchange of the camera co-visibility failed"); return rc; }
            all.assign(nf, 0);
            for (size_t k = 0; k < nf; ++k) all[k] = (((unsigned long long)pk[k / 4] >> (13 * (k % 4))) & 0x1FFFull) ? 1 : 0;
        }
        const esfm::CamGraph g = esfm::cam_graph_from_tracks(d.n_real_cam, d.n_pt, P->h_pt_start.data(), P->h_obs_cam.data(), multi ? &all : nullptr);
        const esfm::SparsePlan plan = esfm::make_sparse_plan(g, leaf_max);
        P->sparse_worthwhile = plan.worthwhile();
        if (int rc = esfm::ba_sparse_create(S.st, plan, g, &P->sparse)) return rc;
        P->sparse_key = key; P->sparse_leaf_max = leaf_max;
    }
    if (P->sparse && (P->sparse_worthwhile || force_sparse)) d.sparse = P->sparse;
    return ESFM_OK;
}

int fill_ones(hipStream_t st, double *dst, size_t n)
{
    std::vector<double> ones(n, 1.0);
    if (n == 0) return ESFM_OK;
    ESFM_HIP_TRY(esfm::copy_h2d(dst, ones.data(), sizeof(double) * n, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return ESFM_OK;
}

// calib == NULL: fixed per-camera intrinsics K4_per_cam; else the shared free block fx, cx, fy, cy, carried as one more
// 6-wide camera-side block behind the n_real cameras (ba_kernels.hpp).
int create_impl(esfm_ctx *ctx, int n_real, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                const float *obs_uv, const float *K4_per_cam, const double *calib, double calib_tol, const double *cams,
                const double *pts, esfm_ba_problem **out)
{
    if (!ctx || !out) { esfm::set_error("ctx/out is NULL"); return ESFM_ERR_INVALID_ARG; }
    *out = nullptr;
    ESFM_REQUIRE(n_real >= 0 && n_pt >= 0 && n_obs >= 0, "negative size");
    ESFM_REQUIRE(n_obs == 0 || (cam_idx && pt_idx && obs_uv), "observation arrays are NULL");
    ESFM_REQUIRE(n_real == 0 || ((K4_per_cam || calib) && cams), "camera arrays are NULL");
    ESFM_REQUIRE(n_pt == 0 || pts, "pts is NULL");
    // Ceres rejects a variable block whose lower bound is not below its upper bound (Program::IsFeasible)
    ESFM_REQUIRE(!calib || calib_tol > 0.0, "intrinsics tolerance must be positive");
    if (calib) for (int i = 0; i < 4; ++i) if (!std::isfinite(calib[i])) { esfm::set_error("non-finite intrinsics"); return ESFM_ERR_NUMERIC; }
    const int n_cam = n_real + (calib ? 1 : 0);   // 6-wide blocks of the reduced system
    ESFM_REQUIRE((int64_t)6 * n_cam < 46000, "reduced system too large for this build (6 n_cam < 46000)");
    for (int k = 0; k < n_obs; ++k)
        ESFM_REQUIRE(cam_idx[k] >= 0 && cam_idx[k] < n_real && pt_idx[k] >= 0 && pt_idx[k] < n_pt, "observation index out of range");
    for (size_t i = 0; i < (size_t)6 * n_real; ++i) if (!std::isfinite(cams[i])) { esfm::set_error("non-finite camera parameter"); return ESFM_ERR_NUMERIC; }
    for (size_t i = 0; i < (size_t)3 * n_pt; ++i) if (!std::isfinite(pts[i])) { esfm::set_error("non-finite point parameter"); return ESFM_ERR_NUMERIC; }
    if (int rc = esfm::set_device(ctx)) return rc;

    // group observations by point (counting sort, stable: keeps the caller's order inside a point)
    std::vector<int32_t> pt_start((size_t)n_pt + 1, 0), order((size_t)n_obs), s_cam((size_t)n_obs), s_pt((size_t)n_obs);
    std::vector<float> s_uv((size_t)2 * n_obs);
    for (int k = 0; k < n_obs; ++k) pt_start[(size_t)pt_idx[k] + 1]++;
    for (int p = 0; p < n_pt; ++p) pt_start[(size_t)p + 1] += pt_start[(size_t)p];
    {
        std::vector<int32_t> fill(pt_start.begin(), pt_start.end() - 1);
        for (int k = 0; k < n_obs; ++k) order[(size_t)fill[(size_t)pt_idx[k]]++] = k;
    }
    auto P = new esfm_ba_problem();
    P->ctx = ctx;
    P->cam_nobs_local.assign((size_t)n_cam, 0.0);
    if (calib) {
        P->cam_nobs_local[(size_t)n_real] = (double)n_obs;   // every observation involves the intrinsics block
        for (int i = 0; i < 4; ++i) P->calib_center[i] = calib[i];
        P->calib_tol = calib_tol;
    }
    for (int t = 0; t < n_obs; ++t) {
        const int k = order[(size_t)t];
        s_cam[(size_t)t] = cam_idx[k]; s_pt[(size_t)t] = pt_idx[k];
        s_uv[2 * (size_t)t] = obs_uv[2 * (size_t)k]; s_uv[2 * (size_t)t + 1] = obs_uv[2 * (size_t)k + 1];
        P->cam_nobs_local[(size_t)cam_idx[k]] += 1.0;
    }
    if (!calib && esfm::ba_solve_is_tiled(n_cam)) { P->h_pt_start = pt_start; P->h_obs_cam = s_cam; }   // (for the reduced system's structure)
    BADev &d = P->d;
    d.n_cam = n_cam; d.n_pt = n_pt; d.n_obs = n_obs;
    d.n_real_cam = n_real; d.has_calib = calib ? 1 : 0;
    const size_t no = (size_t)n_obs, nc6 = (size_t)6 * n_cam, np3 = (size_t)3 * n_pt;
    int rc = ESFM_OK;
    auto A = [&](auto **ptr, size_t count) { if (rc == ESFM_OK) rc = dev_alloc(P, ptr, count); };
    A(&d.obs_cam, no); A(&d.obs_pt, no); A(&d.obs_uv, no); A(&d.pt_start, (size_t)n_pt + 1); A(&d.K4, (size_t)n_real);
    if (calib) A(&d.Jk, 4 * no);
    A(&d.lo_c, nc6); A(&d.up_c, nc6); A(&d.delta_c, nc6); A(&d.delta_p, np3);
    A(&d.cam_nobs, (size_t)n_cam);
    A(&d.x_c, nc6); A(&d.x_p, np3); A(&d.cand_c, nc6); A(&d.cand_p, np3); A(&d.x0_p, np3);
    A(&d.Jc, 12 * no); A(&d.Jp, 6 * no); A(&d.res, 2 * no);
    A(&d.scale_c, nc6); A(&d.scale_p, np3);
    A(&d.EtE, (size_t)6 * n_pt); A(&d.Etr, np3); A(&d.Minv, (size_t)6 * n_pt); A(&d.Aig, np3);
    A(&d.camacc, esfm::ba_camacc_doubles(n_cam)); A(&d.red, esfm::ba_red_doubles(n_cam));
    A(&d.y_c, nc6); A(&d.scal, (size_t)esfm::SC_COUNT);
    A(&d.chol, std::max((nc6 + 1) * (nc6 + 2) / 2 + 2, esfm::ba_chol_large_doubles(n_cam)));
    d.slab_cap = esfm::ba_schur_slab_doubles(n_cam, ctx->num_cu);
    if (d.slab_cap) A(&d.slabs, d.slab_cap);
    if ((size_t)n_real * 304 + 144 <= 160 * 1024) {   // kLinLdsPerCam: the sweep keeps the camera sums in LDS
        d.lin_slab_cap = (size_t)n_cam * 27 * 2 * (size_t)ctx->num_cu;
        A(&d.lin_slabs, d.lin_slab_cap);
    }
    // camera CSR of the (point-sorted) observations, cut into chunks, for the atomic-free per-camera sums
    std::vector<int32_t> cam_obs((size_t)n_obs), cchunk_cam, cchunk_beg, cchunk_end, cam_chunk0((size_t)n_real + 1, 0);
    {
        std::vector<int32_t> cstart((size_t)n_real + 1, 0);
        for (int t = 0; t < n_obs; ++t) cstart[(size_t)s_cam[(size_t)t] + 1]++;
        for (int c = 0; c < n_real; ++c) cstart[(size_t)c + 1] += cstart[(size_t)c];
        std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
        for (int t = 0; t < n_obs; ++t) cam_obs[(size_t)fill[(size_t)s_cam[(size_t)t]]++] = t;
        for (int c = 0; c < n_real; ++c) {
            cam_chunk0[(size_t)c] = (int32_t)cchunk_cam.size();
            for (int b0 = cstart[(size_t)c]; b0 < cstart[(size_t)c + 1]; b0 += esfm::kCamChunk) {
                cchunk_cam.push_back(c); cchunk_beg.push_back(b0); cchunk_end.push_back(std::min(b0 + esfm::kCamChunk, cstart[(size_t)c + 1]));
            }
        }
        cam_chunk0[(size_t)n_real] = (int32_t)cchunk_cam.size();
    }
    d.n_cchunks = (int)cchunk_cam.size();
    A(&d.cam_obs, no); A(&d.cchunk_cam, cchunk_cam.size()); A(&d.cchunk_beg, cchunk_cam.size()); A(&d.cchunk_end, cchunk_cam.size());
    A(&d.cam_chunk0, (size_t)n_real + 1); A(&d.cam_part, cchunk_cam.size() * (size_t)esfm::kCamPart);
    // room for a few launches' partials per slot between two read-backs (a full slot is flushed by an extra reduce launch)
    d.scal_cap = std::max(1 << 12, 4 * (std::max((n_pt + 63) / 64, (n_obs + 255) / 256) + n_pt / 256 + 2));
    A(&d.scal_part, (size_t)esfm::SC_SUM_COUNT * (size_t)d.scal_cap);
    d.parts = &P->parts;
    A(&d.qexp, nc6);
    // Windowed Schur for large camera counts: order points by their lowest camera, cut the observation stream into
    // ~2 chunks per CU.  (Structure only; built once per problem.)  A second table, on camera indices rotated by half the
    // camera count, takes the tracks that are only narrow there (the seam of a closed camera loop); what is wide in both
    // index spaces goes to the plain kernel.
    std::vector<int32_t> slot_obs, chunk_slot, chunk_cam0, slot_obs_b, chunk_slot_b, chunk_cam0_b, wide_obs;
    std::vector<int32_t> mslot_obs[2], mslot_pc[2], mbatch_slot[2], mchunk_batch0[2], mchunk_cam0[2];
    if (d.slab_cap == 0 && n_obs > 0) {
        const int rot = n_real / 2;
        auto rotated = [&](int c) { const int r = c + rot; return r >= n_real ? r - n_real : r; };
        std::vector<int32_t> lo_a((size_t)n_pt, INT32_MAX), hi_a((size_t)n_pt, -1), lo_b((size_t)n_pt, INT32_MAX), hi_b((size_t)n_pt, -1);
        for (int t = 0; t < n_obs; ++t) {
            const size_t p = (size_t)s_pt[(size_t)t];
            const int c = s_cam[(size_t)t], cr = rotated(c);
            lo_a[p] = std::min(lo_a[p], c); hi_a[p] = std::max(hi_a[p], c);
            lo_b[p] = std::min(lo_b[p], cr); hi_b[p] = std::max(hi_b[p], cr);
        }
        // Narrow tracks -- at most kSchurMfCams camera indices wide, no camera twice -- take the matrix-core kernel: per point the
        // Schur update is the rank-3 product (W M^-1) W' over its cameras' rows, a small dense GEMM once points with the same
        // cameras are processed together.
        std::vector<int32_t> mperm[2];
        std::vector<char> taken((size_t)n_pt, 0);
        for (int p = 0; p < n_pt; ++p) {
            const int b = pt_start[(size_t)p], e = pt_start[(size_t)p + 1];
            if (e <= b || e - b > esfm::kSchurMfCams) continue;
            bool dup = false;
            for (int t = b; t < e && !dup; ++t) for (int u = b; u < t; ++u) if (s_cam[(size_t)t] == s_cam[(size_t)u]) { dup = true; break; }
            if (dup) continue;
            if (hi_a[(size_t)p] - lo_a[(size_t)p] < esfm::kSchurMfCams) { mperm[0].push_back(p); taken[(size_t)p] = 1; }
            else if (hi_b[(size_t)p] - lo_b[(size_t)p] < esfm::kSchurMfCams) { mperm[1].push_back(p); taken[(size_t)p] = 1; }
        }
        for (int tb = 0; tb < 2; ++tb) {
            const std::vector<int32_t> &lo = tb ? lo_b : lo_a;
            std::stable_sort(mperm[tb].begin(), mperm[tb].end(), [&](int a, int b) { return lo[(size_t)a] < lo[(size_t)b]; });
        }
        int64_t total_all = 0;
        for (int tb = 0; tb < 2; ++tb) for (int p : mperm[tb]) total_all += pt_start[(size_t)p + 1] - pt_start[(size_t)p];
        // chunks of `per` observations, both tables; returns the number of chunks (= workgroups of the one launch)
        auto build_chunks = [&](int64_t per) {
            for (int tb = 0; tb < 2; ++tb) {
                mslot_obs[tb].clear(); mbatch_slot[tb].clear(); mchunk_batch0[tb].clear(); mchunk_cam0[tb].clear();
                const std::vector<int32_t> &perm = mperm[tb];
                const std::vector<int32_t> &lo = tb ? lo_b : lo_a, &hi = tb ? hi_b : hi_a;
                int64_t in_chunk = 0; int cw = 0, in_batch = 0, pts_batch = 0;
                for (int p : perm) {
                    const int t = pt_start[(size_t)p + 1] - pt_start[(size_t)p];
                    const bool new_chunk = mchunk_cam0[tb].empty() || in_chunk >= per || hi[(size_t)p] - cw >= esfm::kSchurMfCams;
                    if (new_chunk) {
                        mchunk_batch0[tb].push_back((int32_t)mbatch_slot[tb].size());
                        mchunk_cam0[tb].push_back(lo[(size_t)p]); cw = lo[(size_t)p]; in_chunk = 0;
                    }
                    if (new_chunk || in_batch + t > 64 || pts_batch >= 16) { mbatch_slot[tb].push_back((int32_t)mslot_obs[tb].size()); in_batch = 0; pts_batch = 0; }
                    // a point's observations in ascending camera-slot order (the kernel finds "the observation with slot s" by counting
                    // the lower bits of the point's slot mask); in the seam's table that is the ROTATED index
                    const size_t at = mslot_obs[tb].size();
                    for (int k = pt_start[(size_t)p]; k < pt_start[(size_t)p + 1]; ++k) mslot_obs[tb].push_back(k);
                    std::sort(mslot_obs[tb].begin() + (std::ptrdiff_t)at, mslot_obs[tb].end(), [&](int32_t a, int32_t b) {
                        const int ca = tb ? rotated(s_cam[(size_t)a]) : s_cam[(size_t)a], cb = tb ? rotated(s_cam[(size_t)b]) : s_cam[(size_t)b];
                        return ca < cb;
                    });
                    in_batch += t; ++pts_batch; in_chunk += t;
                }
                mbatch_slot[tb].push_back((int32_t)mslot_obs[tb].size());
                mchunk_batch0[tb].push_back((int32_t)mbatch_slot[tb].size() - 1);
            }
            return (int64_t)mchunk_cam0[0].size() + (int64_t)mchunk_cam0[1].size();
        };
        // Both tables run in ONE launch of two workgroups per CU (ba_schur_mfma_kernel: 78 KB of LDS, 244 registers): every chunk should
        // be resident from the start -- a chunk dispatched behind the others adds its whole length to the launch (round 5: the seam's
        // table had its own, much smaller `per`: a hundred short chunks behind 512 long ones, 20 us of tail).  One `per` for both, raised
        // by 1 % until the chunks fit the slots; camera windows can force more chunks than that (wide, scattered tracks): then the
        // first `per` stands.
        {
            auto count_chunks = [&](int64_t per) {          // build_chunks' chunk boundaries without the lists
                int64_t n = 0;
                for (int tb = 0; tb < 2; ++tb) {
                    const std::vector<int32_t> &lo = tb ? lo_b : lo_a, &hi = tb ? hi_b : hi_a;
                    int64_t in_chunk = 0; int cw = 0; bool any = false;
                    for (int p : mperm[tb]) {
                        if (!any || in_chunk >= per || hi[(size_t)p] - cw >= esfm::kSchurMfCams) { ++n; cw = lo[(size_t)p]; in_chunk = 0; any = true; }
                        in_chunk += pt_start[(size_t)p + 1] - pt_start[(size_t)p];
                    }
                }
                return n;
            };
            const int64_t slots = 2 * (int64_t)std::max(1, ctx->num_cu);
            const int64_t per0 = std::max<int64_t>(512, (total_all + slots - 1) / slots);
            int64_t per = per0;
            bool fits = false;
            for (int it = 0; it < 64 && !fits; ++it) {
                fits = count_chunks(per) <= slots;
                if (!fits) per += std::max<int64_t>(1, per / 100);
            }
            build_chunks(fits ? per : per0);
        }
        for (int tb = 0; tb < 2; ++tb) {
            d.n_mchunks[tb] = (int)mchunk_cam0[tb].size();
            if (d.n_mchunks[tb]) {
                mslot_pc[tb].resize(2 * mslot_obs[tb].size());
                for (size_t k = 0; k < mslot_obs[tb].size(); ++k) {
                    mslot_pc[tb][2 * k] = s_pt[(size_t)mslot_obs[tb][k]]; mslot_pc[tb][2 * k + 1] = s_cam[(size_t)mslot_obs[tb][k]];
                }
                A(&d.mslot_pc[tb], mslot_pc[tb].size());
                A(&d.mslot_obs[tb], mslot_obs[tb].size()); A(&d.mbatch_slot[tb], mbatch_slot[tb].size());
                A(&d.mchunk_batch0[tb], mchunk_batch0[tb].size()); A(&d.mchunk_cam0[tb], mchunk_cam0[tb].size());
            }
        }
        std::vector<int32_t> perm_a, perm_b;
        for (int p = 0; p < n_pt; ++p) {
            if (pt_start[(size_t)p + 1] <= pt_start[(size_t)p] || taken[(size_t)p]) continue;
            if (hi_a[(size_t)p] - lo_a[(size_t)p] < esfm::kSchurWinCams) perm_a.push_back(p);
            else if (hi_b[(size_t)p] - lo_b[(size_t)p] < esfm::kSchurWinCams) perm_b.push_back(p);
            else for (int t = pt_start[(size_t)p]; t < pt_start[(size_t)p + 1]; ++t) wide_obs.push_back(t);
        }
        auto build = [&](std::vector<int32_t> &perm, const std::vector<int32_t> &lo, std::vector<int32_t> &slots, std::vector<int32_t> &cslot,
                         std::vector<int32_t> &ccam0) {
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return lo[(size_t)a] < lo[(size_t)b]; });
            int64_t total = 0;
            for (int p : perm) total += pt_start[(size_t)p + 1] - pt_start[(size_t)p];
            const int want_chunks = std::max(1, 2 * ctx->num_cu);     // (1 to 6 chunks per CU measured alike on BA-512: 0.85-0.89 ms)
            const int64_t per = std::max<int64_t>(1024, (total + want_chunks - 1) / want_chunks);
            slots.reserve((size_t)total);
            int64_t in_chunk = 0;
            for (int p : perm) {
                if (cslot.empty() || in_chunk >= per) { cslot.push_back((int32_t)slots.size()); ccam0.push_back(lo[(size_t)p]); in_chunk = 0; }
                for (int t = pt_start[(size_t)p]; t < pt_start[(size_t)p + 1]; ++t) slots.push_back(t);
                in_chunk += pt_start[(size_t)p + 1] - pt_start[(size_t)p];
            }
            cslot.push_back((int32_t)slots.size());
        };
        build(perm_a, lo_a, slot_obs, chunk_slot, chunk_cam0);
        build(perm_b, lo_b, slot_obs_b, chunk_slot_b, chunk_cam0_b);
        d.n_chunks = (int)chunk_cam0.size(); d.n_chunks_b = (int)chunk_cam0_b.size();
        d.n_wide_obs = (int)wide_obs.size();
        A(&d.slot_obs, slot_obs.size()); A(&d.chunk_slot, chunk_slot.size()); A(&d.chunk_cam0, chunk_cam0.size());
        A(&d.slot_obs_b, slot_obs_b.size()); A(&d.chunk_slot_b, chunk_slot_b.size()); A(&d.chunk_cam0_b, chunk_cam0_b.size());
        A(&d.wide_obs, wide_obs.size());
    }
    // point chunks: consecutive points with at most 256 observations (and 256 points) per chunk; a longer track stands alone
    std::vector<int32_t> pchunk_pt0;
    {
        int p = 0;
        while (p < n_pt) {
            pchunk_pt0.push_back(p);
            int obs = 0, pts_in = 0;
            while (p < n_pt && pts_in < 256) {
                const int t = pt_start[(size_t)p + 1] - pt_start[(size_t)p];
                if (pts_in > 0 && obs + t > 256) break;
                obs += t; ++pts_in; ++p;
                if (obs > 256) break;       // a single long track
            }
        }
        pchunk_pt0.push_back(n_pt);
        d.n_pchunks = (int)pchunk_pt0.size() - 1;
        A(&d.pchunk_pt0, pchunk_pt0.size());
        A(&d.pchunk_info, (size_t)std::max(d.n_pchunks, 1));
    }
    std::vector<int32_t> pchunk_info(4 * (size_t)std::max(d.n_pchunks, 1), 0);
    for (int c = 0; c < d.n_pchunks; ++c) {
        const int p0 = pchunk_pt0[(size_t)c], p1 = pchunk_pt0[(size_t)c + 1];
        pchunk_info[4 * (size_t)c] = p0; pchunk_info[4 * (size_t)c + 1] = p1;
        pchunk_info[4 * (size_t)c + 2] = pt_start[(size_t)p0]; pchunk_info[4 * (size_t)c + 3] = pt_start[(size_t)p1];
    }
    if (rc != ESFM_OK) { esfm_ba_problem_destroy(P); return rc; }
    hipStream_t st = ctx->stream;
    auto up = [&](void *dst, const void *src, size_t bytes) {
        if (rc == ESFM_OK && bytes) {
            hipError_t e = esfm::copy_h2d(dst, src, bytes, st);
            if (e != hipSuccess) { esfm::set_error("hipMemcpyAsync H2D failed: %s", hipGetErrorString(e)); rc = ESFM_ERR_HIP; }
        }
    };
    up(d.obs_cam, s_cam.data(), sizeof(int32_t) * no); up(d.obs_pt, s_pt.data(), sizeof(int32_t) * no);
    up(d.obs_uv, s_uv.data(), sizeof(float) * 2 * no); up(d.pt_start, pt_start.data(), sizeof(int32_t) * ((size_t)n_pt + 1));
    if (!calib) up(d.K4, K4_per_cam, sizeof(float) * 4 * (size_t)n_real);
    const double calib_block[6] = {calib ? calib[0] : 0.0, calib ? calib[1] : 0.0, calib ? calib[2] : 0.0, calib ? calib[3] : 0.0, 0.0, 0.0};
    if (calib) up(d.x_c + 6 * (size_t)n_real, calib_block, sizeof(calib_block));
    if (d.n_chunks) {
        up(d.slot_obs, slot_obs.data(), sizeof(int32_t) * slot_obs.size());
        up(d.chunk_slot, chunk_slot.data(), sizeof(int32_t) * chunk_slot.size());
        up(d.chunk_cam0, chunk_cam0.data(), sizeof(int32_t) * chunk_cam0.size());
    }
    if (d.n_wide_obs) up(d.wide_obs, wide_obs.data(), sizeof(int32_t) * wide_obs.size());
    for (int tb = 0; tb < 2; ++tb) {
        if (!d.n_mchunks[tb]) continue;
        up(d.mslot_obs[tb], mslot_obs[tb].data(), sizeof(int32_t) * mslot_obs[tb].size());
        up(d.mslot_pc[tb], mslot_pc[tb].data(), sizeof(int32_t) * mslot_pc[tb].size());
        up(d.mbatch_slot[tb], mbatch_slot[tb].data(), sizeof(int32_t) * mbatch_slot[tb].size());
        up(d.mchunk_batch0[tb], mchunk_batch0[tb].data(), sizeof(int32_t) * mchunk_batch0[tb].size());
        up(d.mchunk_cam0[tb], mchunk_cam0[tb].data(), sizeof(int32_t) * mchunk_cam0[tb].size());
    }
    if (d.n_chunks_b) {
        up(d.slot_obs_b, slot_obs_b.data(), sizeof(int32_t) * slot_obs_b.size());
        up(d.chunk_slot_b, chunk_slot_b.data(), sizeof(int32_t) * chunk_slot_b.size());
        up(d.chunk_cam0_b, chunk_cam0_b.data(), sizeof(int32_t) * chunk_cam0_b.size());
    }
    up(d.pchunk_pt0, pchunk_pt0.data(), sizeof(int32_t) * pchunk_pt0.size());
    up(d.pchunk_info, pchunk_info.data(), sizeof(int32_t) * pchunk_info.size());
    up(d.cam_obs, cam_obs.data(), sizeof(int32_t) * no);
    up(d.cchunk_cam, cchunk_cam.data(), sizeof(int32_t) * cchunk_cam.size()); up(d.cchunk_beg, cchunk_beg.data(), sizeof(int32_t) * cchunk_beg.size());
    up(d.cchunk_end, cchunk_end.data(), sizeof(int32_t) * cchunk_end.size()); up(d.cam_chunk0, cam_chunk0.data(), sizeof(int32_t) * cam_chunk0.size());
    up(d.x_c, cams, sizeof(double) * 6 * (size_t)n_real); up(d.x_p, pts, sizeof(double) * np3);
    // red: the entries no kernel ever writes (blocks above the diagonal) must read as zero
    if (rc == ESFM_OK && hipMemsetAsync(d.red, 0, sizeof(double) * esfm::ba_red_doubles(n_cam), st) != hipSuccess) { esfm::set_error("memset failed"); rc = ESFM_ERR_HIP; }
    if (rc == ESFM_OK && hipStreamSynchronize(st) != hipSuccess) { esfm::set_error("stream sync failed"); rc = ESFM_ERR_HIP; }
    if (rc != ESFM_OK) { esfm_ba_problem_destroy(P); return rc; }
    *out = P;
    return ESFM_OK;
}

}  // namespace

extern "C" {

void esfm_ba_options_default(esfm_ba_options *opt) { if (opt) options_default(opt); }

int esfm_ba_problem_create(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                           const float *obs_uv, const float *K4_per_cam, const double *cams, const double *pts,
                           esfm_ba_problem **out)
{
    if (n_cam > 0 && !K4_per_cam) { esfm::set_error("K4_per_cam is NULL"); return ESFM_ERR_INVALID_ARG; }
    return create_impl(ctx, n_cam, n_pt, n_obs, cam_idx, pt_idx, obs_uv, K4_per_cam, nullptr, 0.0, cams, pts, out);
}

int esfm_ba_problem_create_free_calib(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                                      const float *obs_uv, const double *calib4, double calib_tolerance, const double *cams,
                                      const double *pts, esfm_ba_problem **out)
{
    if (!calib4) { esfm::set_error("calib4 is NULL"); return ESFM_ERR_INVALID_ARG; }
    return create_impl(ctx, n_cam, n_pt, n_obs, cam_idx, pt_idx, obs_uv, nullptr, calib4, calib_tolerance, cams, pts, out);
}

int esfm_ba_problem_set_calib(esfm_ba_problem *P, const double *calib4, double calib_tolerance)
{
    if (!P || !calib4) { esfm::set_error("NULL argument"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(P->d.has_calib, "problem was created with fixed intrinsics");
    ESFM_REQUIRE(calib_tolerance > 0.0, "intrinsics tolerance must be positive");
    for (int i = 0; i < 4; ++i) if (!std::isfinite(calib4[i])) { esfm::set_error("non-finite intrinsics"); return ESFM_ERR_NUMERIC; }
    if (int rc = esfm::set_device(P->ctx)) return rc;
    for (int i = 0; i < 4; ++i) P->calib_center[i] = calib4[i];
    P->calib_tol = calib_tolerance;
    ESFM_HIP_TRY(esfm::copy_h2d(P->d.x_c + 6 * (size_t)P->d.n_real_cam, calib4, sizeof(double) * 4, P->ctx->stream));
    ESFM_HIP_TRY(hipStreamSynchronize(P->ctx->stream));
    return ESFM_OK;
}

int esfm_ba_problem_get_calib(esfm_ba_problem *P, double *calib4)
{
    if (!P || !calib4) { esfm::set_error("NULL argument"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(P->d.has_calib, "problem was created with fixed intrinsics");
    if (int rc = esfm::set_device(P->ctx)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(calib4, P->d.x_c + 6 * (size_t)P->d.n_real_cam, sizeof(double) * 4, P->ctx->stream));
    ESFM_HIP_TRY(hipStreamSynchronize(P->ctx->stream));
    return ESFM_OK;
}

int esfm_ba_problem_fix_camera(esfm_ba_problem *P, int cam, double threshold)
{
    if (!P) { esfm::set_error("problem is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(cam < P->d.n_real_cam, "camera index out of range");
    ESFM_REQUIRE(cam < 0 || threshold > 0.0, "threshold must be positive");
    P->ref_cam = cam < 0 ? -1 : cam;
    P->ref_threshold = cam < 0 ? 0.0 : threshold;
    return ESFM_OK;
}

double esfm_ba_line_search_next_step(double f0, double g0, double x_prev, double f_prev, double g_prev, int prev_valid,
                                     double x_cur, double f_cur, double g_cur, int cur_valid)
{
    namespace ls = esfm::linesearch;
    ls::Sample ini, prev, cur;
    ini.x = 0.0; ini.f = f0; ini.g = g0; ini.valid = true;
    prev.x = x_prev; prev.f = f_prev; prev.g = g_prev; prev.valid = prev_valid != 0;
    cur.x = x_cur; cur.f = f_cur; cur.g = g_cur; cur.valid = cur_valid != 0;
    return ls::next_step(ini, prev, cur);
}

int esfm_ba_problem_set_params(esfm_ba_problem *P, const double *cams, const double *pts)
{
    if (!P || (!cams && P->d.n_real_cam) || (!pts && P->d.n_pt)) { esfm::set_error("NULL argument"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(P->ctx)) return rc;
    hipStream_t st = P->ctx->stream;
    if (P->d.n_real_cam) ESFM_HIP_TRY(esfm::copy_h2d(P->d.x_c, cams, sizeof(double) * 6 * (size_t)P->d.n_real_cam, st));
    if (P->d.n_pt) ESFM_HIP_TRY(esfm::copy_h2d(P->d.x_p, pts, sizeof(double) * 3 * (size_t)P->d.n_pt, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return ESFM_OK;
}

int esfm_ba_problem_get_params(esfm_ba_problem *P, double *cams, double *pts)
{
    if (!P || (!cams && P->d.n_real_cam) || (!pts && P->d.n_pt)) { esfm::set_error("NULL argument"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(P->ctx)) return rc;
    hipStream_t st = P->ctx->stream;
    if (P->d.n_real_cam) ESFM_HIP_TRY(esfm::copy_d2h(cams, P->d.x_c, sizeof(double) * 6 * (size_t)P->d.n_real_cam, st));
    if (P->d.n_pt) ESFM_HIP_TRY(esfm::copy_d2h(pts, P->d.x_p, sizeof(double) * 3 * (size_t)P->d.n_pt, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return ESFM_OK;
}

int esfm_ba_problem_destroy(esfm_ba_problem *P)
{
    if (!P) return ESFM_OK;
    if (P->ctx) { (void)hipSetDevice(P->ctx->device); (void)hipStreamSynchronize(P->ctx->stream); }
    // small chunks and the mailbox stay with the context for its next problem (at most kKeepChunks chunks of at most kKeepBytes)
    constexpr size_t kKeepChunks = 6, kKeepBytes = size_t(64) << 20;
    for (const auto &c : P->allocs) {
        if (P->ctx && c.bytes <= kKeepBytes && P->ctx->ba_chunks.size() < kKeepChunks) P->ctx->ba_chunks.push_back(c);
        else (void)hipFree(c.ptr);
    }
    esfm::ba_sparse_destroy(P->sparse);
    if (P->h_scal) {
        if (P->ctx && P->ctx->ba_mailboxes.size() < 2) P->ctx->ba_mailboxes.push_back(P->h_scal);
        else (void)hipHostFree(P->h_scal);
    }
    delete P;
    return ESFM_OK;
}

int esfm_ba_problem_cost(esfm_ba_problem *P, double cauchy_a, double *cost)
{
    if (!P || !cost) { esfm::set_error("NULL argument"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(P->ctx)) return rc;
    hipStream_t st = P->ctx->stream;
    ESFM_HIP_TRY(hipMemsetAsync(P->d.scal, 0, sizeof(double) * esfm::SC_COUNT, st));
    esfm::ba_scal_discard(P->d, 0, esfm::SC_SUM_COUNT);
    if (int rc = esfm::ba_cost(st, P->d, P->ctx->num_cu, P->d.x_c, P->d.x_p, cauchy_a, esfm::SC_CAND_COST, esfm::SC_CAND_BAD)) return rc;
    if (int rc = esfm::ba_scal_reduce(st, P->d)) return rc;
    double h[esfm::SC_COUNT];
    ESFM_HIP_TRY(esfm::copy_d2h(h, P->d.scal, sizeof(h), st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    *cost = h[esfm::SC_CAND_BAD] > 0.0 ? DBL_MAX : h[esfm::SC_CAND_COST];
    return ESFM_OK;
}

int esfm_ba_problem_solve(esfm_ba_problem *P, const esfm_ba_options *options, esfm_allreduce_fn allreduce, void *allreduce_user,
                          esfm_ba_summary *sum)
{
    if (!P) { esfm::set_error("problem is NULL"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(P->ctx)) return rc;
    esfm_ba_summary local_sum;
    if (!sum) sum = &local_sum;
    memset(sum, 0, sizeof(*sum));
    Solver S;
    S.P = P; S.ar = allreduce; S.ar_user = allreduce_user; S.st = P->ctx->stream;
    if (!P->h_scal && !P->ctx->ba_mailboxes.empty()) {
        P->h_scal = static_cast<double *>(P->ctx->ba_mailboxes.back());
        P->ctx->ba_mailboxes.pop_back();
        memset(P->h_scal, 0, sizeof(double) * (esfm::SC_COUNT + 2));
    }
    if (!P->h_scal) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&P->h_scal), sizeof(double) * (esfm::SC_COUNT + 2), hipHostMallocCoherent);   // explicit: the host spins on a device-written flag
        if (e == hipSuccess) memset(P->h_scal, 0, sizeof(double) * (esfm::SC_COUNT + 2));
        if (e != hipSuccess) { esfm::set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return ESFM_ERR_HIP; }
    }
    S.h = P->h_scal;
    if (options) S.opt = *options; else options_default(&S.opt);
    const esfm_ba_options &opt = S.opt;
    ESFM_REQUIRE(opt.initial_trust_region_radius > 0.0 && opt.max_num_iterations >= 0, "bad options");
    BADev &d = P->d;
    hipStream_t st = S.st;
    const bool multi = allreduce != nullptr;
    P->parts.single_rank = !multi; P->parts.grad_done = false;
    double *h = S.h;
    if (multi && !d.red_packed) { if (int rc = dev_alloc(P, &d.red_packed, esfm::ba_red_packed_doubles(d.n_cam))) return rc; }

    // camera observation counts over all shards; Jacobi scaling starts at 1
    if (d.n_cam) ESFM_HIP_TRY(esfm::copy_h2d(d.cam_nobs, P->cam_nobs_local.data(), sizeof(double) * (size_t)d.n_cam, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    if (int rc = S.allreduce(d.cam_nobs, d.n_cam, ESFM_REDUCE_SUM)) return rc;
    if (int rc = plan_reduced_structure(P, S, multi)) return rc;
    if (int rc = fill_ones(st, d.scale_c, (size_t)6 * d.n_cam)) return rc;
    if (int rc = fill_ones(st, d.scale_p, (size_t)3 * d.n_pt)) return rc;
    if (multi && d.n_pt) ESFM_HIP_TRY(hipMemcpyAsync(d.x0_p, d.x_p, sizeof(double) * 3 * (size_t)d.n_pt, hipMemcpyDeviceToDevice, st));
    {
        std::vector<double> cn((size_t)d.n_cam);
        if (d.n_cam) ESFM_HIP_TRY(esfm::copy_d2h(cn.data(), d.cam_nobs, sizeof(double) * (size_t)d.n_cam, st));
        std::vector<int32_t> ps((size_t)d.n_pt + 1);
        ESFM_HIP_TRY(esfm::copy_d2h(ps.data(), d.pt_start, sizeof(int32_t) * ((size_t)d.n_pt + 1), st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
        for (int c = 0; c < d.n_real_cam; ++c) sum->num_active_cameras += cn[(size_t)c] > 0.0;
        for (int p = 0; p < d.n_pt; ++p) sum->num_active_points += ps[(size_t)p + 1] > ps[(size_t)p];
        // box bounds, only on blocks that take part in the problem (Ceres drops unused blocks with their bounds)
        std::vector<double> lo((size_t)6 * d.n_cam, -INFINITY), up((size_t)6 * d.n_cam, INFINITY);
        d.constrained = 0;
        if (P->ref_cam >= 0 && cn[(size_t)P->ref_cam] > 0.0) {
            for (int i = 0; i < 6; ++i) { lo[6 * (size_t)P->ref_cam + i] = -P->ref_threshold; up[6 * (size_t)P->ref_cam + i] = P->ref_threshold; }
            d.constrained = 1;
        }
        if (d.has_calib && cn[(size_t)d.n_real_cam] > 0.0) {
            for (int i = 0; i < 4; ++i) {
                lo[6 * (size_t)d.n_real_cam + i] = P->calib_center[i] - P->calib_tol;
                up[6 * (size_t)d.n_real_cam + i] = P->calib_center[i] + P->calib_tol;
            }
            d.constrained = 1;
        }
        if (d.n_cam) {
            ESFM_HIP_TRY(esfm::copy_h2d(d.lo_c, lo.data(), sizeof(double) * lo.size(), st));
            ESFM_HIP_TRY(esfm::copy_h2d(d.up_c, up.data(), sizeof(double) * up.size(), st));
            ESFM_HIP_TRY(hipStreamSynchronize(st));
        }
    }
    const bool constrained = d.constrained != 0;

    const auto t0 = std::chrono::steady_clock::now();
    auto finish = [&](int rc) {
        sum->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    };

    // ---- iteration 0 (TrustRegionMinimizer::IterationZero) ----
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    if (int rc = S.zero_scal()) return finish(rc);
    if (constrained) { if (int rc = esfm::ba_project_cameras(st, d)) return finish(rc); }   // x <- Plus(x, 0)
    if (int rc = esfm::ba_param_sqnorm(st, d)) return finish(rc);
    if (int rc = S.linearize(false, radius)) return finish(rc);
    if (opt.jacobi_scaling) {
        if (int rc = esfm::ba_jacobi_scaling(st, d)) return finish(rc);
        // keep |x|^2, restart the other accumulators, and linearise again with scaled columns
        ESFM_HIP_TRY(hipMemsetAsync(d.scal, 0, sizeof(double) * esfm::SC_XNORM_SQ_PT, st));
        ESFM_HIP_TRY(hipMemsetAsync(d.scal + esfm::SC_LIN_BAD, 0, sizeof(double) * (esfm::SC_SUM_COUNT - esfm::SC_LIN_BAD), st));
        esfm::ba_scal_discard(d, 0, esfm::SC_XNORM_SQ_PT); esfm::ba_scal_discard(d, esfm::SC_LIN_BAD, esfm::SC_SUM_COUNT);
        ESFM_HIP_TRY(hipMemsetAsync(d.scal + esfm::SC_GMAX, 0, sizeof(double), st));
        if (int rc = S.linearize(true, radius)) return finish(rc);
    }
    if (int rc = esfm::ba_camera_gradient(st, d)) return finish(rc);
    if (int rc = S.fetch_scal()) return finish(rc);
    if (h[esfm::SC_LIN_BAD] > 0.0) {
        esfm::set_error("non-finite residual or Jacobian at the initial point");
        sum->termination = ESFM_BA_FAILURE;
        return finish(ESFM_ERR_NUMERIC);
    }
    double x_cost = h[esfm::SC_COST];
    double gmax = h[esfm::SC_GMAX];
    double x_norm = std::sqrt(h[esfm::SC_XNORM_SQ_PT] + h[esfm::SC_XNORM_SQ_CAM]);
    double prep_radius = radius;  // radius the per-point inverses were built with
    bool prep_singular = h[esfm::SC_PT_SINGULAR] > 0.0;   // ... and whether one of them could not be inverted (any rank)
    bool reuse_diagonal = false;  // the LM diagonal is a function of J only; kept for parity with the strategy's state
    int n_invalid = 0;
    sum->initial_cost = x_cost;
    {
        esfm_ba_iteration &it = sum->iterations[0];
        it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost;
        it.gradient_max_norm = gmax; it.trust_region_radius = radius;
    }
    sum->num_iterations = 0; sum->num_successful_steps = 1;
    if (opt.verbose)
        printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e\n",
               0, x_cost, 0.0, gmax, 0.0, 0.0, radius);
    bool terminated = false;
    int rc_final = ESFM_OK;
    if (gmax <= opt.gradient_tolerance) { sum->termination = ESFM_BA_CONVERGENCE; terminated = true; }

    // ---- main loop (TrustRegionMinimizer::Minimize) ----
    int iter = 0;
    double last_gmax = gmax;
    // No bounds: the read-back that follows the re-linearisation of an accepted step is DEFERRED to the next step's --
    // the next Schur complement and solve are enqueued straight behind the sweep (one host round trip and one read-back launch
    // less per accepted step).  What that read-back delivers -- the cost, gradient norm and validity at the accepted point --
    // is only needed after the next step has been computed: the right-hand side's fixed-point exponent takes its bound from the
    // candidate's cost (the same function value, computed by the back-substitution launch), the gradient-tolerance test and the
    // log entry of the accepted iteration are completed one read-back later (a step computed past convergence is discarded).
    const bool may_defer = !constrained && !opt.verbose;     // (sharded solves too: the deferred scalars are all-reduced like the others, every rank decides alike)
    bool pending_lin = false;
    double pending_cost_bound = 0.0;
    // completes the accepted iteration `it_acc` from the scalars of its re-linearisation; false: the solve ends here
    auto resolve_pending = [&](int it_acc) -> bool {
        pending_lin = false;
        if (h[esfm::SC_LIN_BAD] > 0.0) {
            esfm::set_error("non-finite residual or Jacobian after an accepted step");
            sum->termination = ESFM_BA_FAILURE; rc_final = ESFM_ERR_NUMERIC; terminated = true;
        }
        x_cost = h[esfm::SC_COST];
        prep_singular = h[esfm::SC_PT_SINGULAR] > 0.0;
        gmax = h[esfm::SC_GMAX]; last_gmax = gmax;
        if (it_acc < ESFM_BA_MAX_LOG) { sum->iterations[it_acc].cost = x_cost; sum->iterations[it_acc].gradient_max_norm = gmax; }
        if (!terminated && gmax <= opt.gradient_tolerance) { sum->termination = ESFM_BA_CONVERGENCE; terminated = true; }
        return !terminated;
    };
    while (!terminated) {
        if (iter >= opt.max_num_iterations) { sum->termination = ESFM_BA_NO_CONVERGENCE; break; }
        if (radius <= opt.min_trust_region_radius) {
            if (pending_lin) { if (int rc = S.fetch_scal()) return finish(rc); if (!resolve_pending(iter)) break; }
            sum->termination = ESFM_BA_CONVERGENCE; break;
        }
        ++iter;
        esfm_ba_iteration cur;
        memset(&cur, 0, sizeof(cur));
        cur.iteration = iter; cur.gradient_max_norm = last_gmax;
        // LevenbergMarquardtStrategy::ComputeStep: D^2 = clamp(diag(J'J)) / radius, then the Schur solve
        // (a deferred read-back: the slots were reset by the last read-back and hold the sweep's sums -- nothing to reset, nothing to forget)
        if (!pending_lin) { if (int rc = S.zero_scal()) return finish(rc); }
        bool reprepped = false;
        if (prep_radius != radius) {
            if (int rc = esfm::ba_point_prep(st, d, radius, opt.min_lm_diagonal, opt.max_lm_diagonal, false)) return finish(rc);
            prep_radius = radius; reprepped = true;
        }
        {
            esfm::KernelTimer tm(P->ctx, ESFM_K_BA_SCHUR);
            // |robustified residual vector| over all ranks (deferred read-back: the candidate's cost bounds the cost at the same point)
            const double rhs_bound = std::sqrt(2.0 * std::max(pending_lin ? pending_cost_bound : x_cost, 0.0));
            if (int rc = esfm::ba_schur(st, d, P->ctx->num_cu, d.slabs, d.slab_cap, rhs_bound)) return finish(rc);
            if (int rc = esfm::ba_schur_calib(st, d, rhs_bound)) return finish(rc);
        }
        if (multi && d.sparse) {
            // one exchange per LM iteration, of the co-visible camera blocks and the right-hand side only (BA-512: 1.4 MB instead of 37.8)
            if (int rc = esfm::ba_sparse_pack(st, d, d.sparse, d.red_packed)) return finish(rc);
            if (int rc = S.allreduce(d.red_packed, (int64_t)esfm::ba_sparse_packed_doubles(d.sparse, d.n_cam), ESFM_REDUCE_SUM)) return finish(rc);
        } else if (multi) {
            // one exchange per LM iteration: the block-lower-triangular S and the right-hand side, packed
            if (int rc = esfm::ba_red_pack(st, d, d.red_packed, false)) return finish(rc);
            if (int rc = S.allreduce(d.red_packed, (int64_t)esfm::ba_red_packed_doubles(d.n_cam), ESFM_REDUCE_SUM)) return finish(rc);
            if (int rc = esfm::ba_red_pack(st, d, d.red_packed, true)) return finish(rc);
        }
        {
            esfm::KernelTimer tm(P->ctx, ESFM_K_BA_SOLVE);
            if (int rc = esfm::ba_solve_reduced(st, d, radius, opt.min_lm_diagonal, opt.max_lm_diagonal)) return finish(rc);
        }
        if (int rc = esfm::ba_camera_step(st, d)) return finish(rc);
        // back-substitution; the candidate's cost at full step comes out of the same launch -- bounded problems also need the slope
        // there for the line search, which is ba_cost's job
        // (fused only on small problems: it saves a launch gap, but the chunk kernel's occupancy is LDS-bound and the extra f64
        // work costs more than ba_cost's own pass from ~1M observations: BA-512 281 us fused against 140 + 42 us)
        const bool fuse_cost = !constrained && d.n_obs < (1 << 20);
        if (int rc = esfm::ba_backsub(st, d, fuse_cost, opt.cauchy_a)) return finish(rc);
        if (!fuse_cost) {
            if (int rc = esfm::ba_cost(st, d, P->ctx->num_cu, d.cand_c, d.cand_p, opt.cauchy_a, esfm::SC_CAND_COST, esfm::SC_CAND_BAD, constrained)) return finish(rc);
        }
        if (int rc = S.fetch_scal()) return finish(rc);
        if (pending_lin) {
            // the accepted iteration iter - 1 is completed first; past convergence (or on failure) the step just computed is dropped
            if (!resolve_pending(iter - 1)) { sum->num_iterations = iter - 1; break; }
            cur.gradient_max_norm = last_gmax;
        }
        reuse_diagonal = true;
        const double model_cost_change = h[esfm::SC_MODEL_CHANGE];
        double step_norm = std::sqrt(h[esfm::SC_STEP_SQ_PT] + h[esfm::SC_STEP_SQ_CAM]);
        double cand_norm = std::sqrt(h[esfm::SC_CAND_SQ_PT] + h[esfm::SC_CAND_SQ_CAM]);
        if (reprepped) prep_singular = h[esfm::SC_PT_SINGULAR] > 0.0;
        const bool lin_ok = h[esfm::SC_CHOL_FAIL] == 0.0 && !prep_singular && std::isfinite(model_cost_change) &&
                            std::isfinite(step_norm);
        cur.model_cost_change = model_cost_change;
        cur.step_is_valid = lin_ok && (model_cost_change > 0.0);
        if (!cur.step_is_valid) {
            // HandleInvalidStep + StepIsInvalid
            if (++n_invalid >= opt.max_num_consecutive_invalid_steps) { sum->termination = ESFM_BA_FAILURE; terminated = true; }
            radius *= 0.5; reuse_diagonal = true;
            cur.cost = x_cost; cur.trust_region_radius = radius;
            sum->num_unsuccessful_steps++;
            if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
            sum->num_iterations = iter;
            continue;
        }
        n_invalid = 0;
        if (constrained) {
            // TrustRegionMinimizer::DoLineSearch: Armijo search from step size 1 along delta; every trial is one
            // take-step + cost-with-slope pass over the observations and one scalar read-back.
            namespace ls = esfm::linesearch;
            const double g0 = h[esfm::SC_GDOTD], dmax = h[esfm::SC_DMAX];
            auto sample_from_h = [&](double x) {
                ls::Sample s;
                s.x = x; s.f = h[esfm::SC_CAND_COST]; s.g = h[esfm::SC_LS_GRAD];
                s.valid = h[esfm::SC_CAND_BAD] == 0.0 && std::isfinite(s.f) && std::isfinite(s.g);
                return s;
            };
            auto evaluate_at = [&](double t, bool slope) -> int {
                if (int rc = S.zero_scal()) return rc;
                if (int rc = esfm::ba_take_step(st, d, t)) return rc;
                if (int rc = esfm::ba_cost(st, d, P->ctx->num_cu, d.cand_c, d.cand_p, opt.cauchy_a, esfm::SC_CAND_COST, esfm::SC_CAND_BAD, slope)) return rc;
                return S.fetch_scal();
            };
            ls::Sample initial, previous, current = sample_from_h(1.0);
            initial.x = 0.0; initial.f = x_cost; initial.g = g0; initial.valid = true;
            int ls_it = 0;
            bool ls_ok = true;
            while (!current.valid || current.f > x_cost + ls::kSufficientDecrease * g0 * current.x) {
                if (++ls_it >= ls::kMaxIterations) { ls_ok = false; break; }
                const double t = ls::next_step(initial, previous, current);
                if (t * dmax < ls::kMinStepSize) { ls_ok = false; break; }
                previous = current;
                if (int rc = evaluate_at(t, true)) return finish(rc);
                current = sample_from_h(t);
            }
            cur.line_search_steps = ls_it;
            // a failed search leaves delta as it was: back to the full step
            if (!ls_ok && current.x != 1.0) { if (int rc = evaluate_at(1.0, false)) return finish(rc); }
            step_norm = std::sqrt(h[esfm::SC_STEP_SQ_PT] + h[esfm::SC_STEP_SQ_CAM]);
            cand_norm = std::sqrt(h[esfm::SC_CAND_SQ_PT] + h[esfm::SC_CAND_SQ_CAM]);
        }
        const double cand_cost = h[esfm::SC_CAND_BAD] > 0.0 ? DBL_MAX : h[esfm::SC_CAND_COST];
        cur.step_norm = step_norm;
        cur.cost_change = x_cost - cand_cost;
        auto log_and_stop = [&]() {
            sum->termination = ESFM_BA_CONVERGENCE; terminated = true;
            cur.cost = x_cost; cur.trust_region_radius = radius;
            if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
            sum->num_iterations = iter;
        };
        // ParameterToleranceReached / FunctionToleranceReached: tested before acceptance, step not applied
        if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { log_and_stop(); break; }
        if (std::fabs(cur.cost_change) <= opt.function_tolerance * x_cost) { log_and_stop(); break; }
        cur.relative_decrease = (x_cost - cand_cost) / model_cost_change;
        if (cur.relative_decrease > opt.min_relative_decrease) {
            // HandleSuccessfulStep: x <- candidate (pointer swap), re-linearise
            std::swap(d.x_c, d.cand_c); std::swap(d.x_p, d.cand_p);
            P->params_swapped = !P->params_swapped;
            x_norm = cand_norm;
            const double q = 2.0 * cur.relative_decrease - 1.0;
            radius = radius / std::max(1.0 / 3.0, 1.0 - q * q * q);
            radius = std::min(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false;
            if (int rc = S.zero_scal()) return finish(rc);
            if (int rc = S.linearize(opt.jacobi_scaling != 0, radius, cand_cost < DBL_MAX ? cand_cost * (1.0 + 1e-9) : -1.0)) return finish(rc);
            prep_radius = radius;
            if (int rc = esfm::ba_camera_gradient(st, d)) return finish(rc);
            if (may_defer && iter < opt.max_num_iterations && radius > opt.min_trust_region_radius) {
                // read-back deferred to the next step's (see may_defer); cost and gradient norm of this log entry follow then
                pending_lin = true;
                pending_cost_bound = cand_cost * (1.0 + 1e-9);
                cur.step_is_successful = 1; cur.cost = cand_cost; cur.gradient_max_norm = last_gmax;
                sum->num_successful_steps++;
                cur.trust_region_radius = radius;
                if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
                sum->num_iterations = iter;
                continue;
            }
            if (int rc = S.fetch_scal()) return finish(rc);
            if (h[esfm::SC_LIN_BAD] > 0.0) {
                esfm::set_error("non-finite residual or Jacobian after an accepted step");
                sum->termination = ESFM_BA_FAILURE; rc_final = ESFM_ERR_NUMERIC; terminated = true;
            }
            x_cost = h[esfm::SC_COST];
            prep_singular = h[esfm::SC_PT_SINGULAR] > 0.0;
            gmax = h[esfm::SC_GMAX]; last_gmax = gmax;
            cur.step_is_successful = 1; cur.cost = x_cost; cur.gradient_max_norm = gmax;
            sum->num_successful_steps++;
            if (gmax <= opt.gradient_tolerance) { sum->termination = ESFM_BA_CONVERGENCE; terminated = true; }
        } else {
            // HandleUnsuccessfulStep + StepRejected
            cur.step_is_successful = 0; cur.cost = cand_cost;
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            sum->num_unsuccessful_steps++;
        }
        cur.trust_region_radius = radius;
        if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
        sum->num_iterations = iter;
        if (opt.verbose)
            printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e\n", iter, cur.cost, cur.cost_change, cur.gradient_max_norm, cur.step_norm,
                   cur.relative_decrease, radius);
    }
    (void)reuse_diagonal;
    sum->final_cost = x_cost;
    if (multi && d.n_pt) {
        // every rank ends with the full point set: sum the owners' deltas
        if (int rc = esfm::ba_points_delta(st, d, true)) return finish(rc);
        if (int rc = S.allreduce(d.x_p, (int64_t)3 * d.n_pt, ESFM_REDUCE_SUM)) return finish(rc);
        if (int rc = esfm::ba_points_delta(st, d, false)) return finish(rc);
    }
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return finish(rc_final);
}

int esfm_ba_solve(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                  const float *K4_per_cam, double *cams, double *pts, const esfm_ba_options *options, esfm_allreduce_fn allreduce,
                  void *allreduce_user, esfm_ba_summary *summary)
{
    esfm_ba_problem *P = nullptr;
    if (int rc = esfm_ba_problem_create(ctx, n_cam, n_pt, n_obs, cam_idx, pt_idx, obs_uv, K4_per_cam, cams, pts, &P)) return rc;
    int rc = esfm_ba_problem_solve(P, options, allreduce, allreduce_user, summary);
    if (rc == ESFM_OK || rc == ESFM_ERR_NUMERIC) {
        const int rc2 = esfm_ba_problem_get_params(P, cams, pts);
        if (rc == ESFM_OK) rc = rc2;
    }
    esfm_ba_problem_destroy(P);
    return rc;
}

int esfm_ba_solve_ex(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                     const float *K4_per_cam, double *cams, double *pts, double *calib4, double calib_tolerance, int ref_cam,
                     double ref_threshold, const esfm_ba_options *options, esfm_allreduce_fn allreduce, void *allreduce_user,
                     esfm_ba_summary *summary)
{
    esfm_ba_problem *P = nullptr;
    if (int rc = create_impl(ctx, n_cam, n_pt, n_obs, cam_idx, pt_idx, obs_uv, K4_per_cam, calib4, calib_tolerance, cams, pts, &P)) return rc;
    int rc = esfm_ba_problem_fix_camera(P, ref_cam, ref_threshold);
    if (rc == ESFM_OK) rc = esfm_ba_problem_solve(P, options, allreduce, allreduce_user, summary);
    if (rc == ESFM_OK || rc == ESFM_ERR_NUMERIC) {
        int rc2 = esfm_ba_problem_get_params(P, cams, pts);
        if (rc2 == ESFM_OK && calib4) rc2 = esfm_ba_problem_get_calib(P, calib4);
        if (rc == ESFM_OK) rc = rc2;
    }
    esfm_ba_problem_destroy(P);
    return rc;
}

// Greedy balance of observation counts over shards, points in index order.
int esfm_ba_shard_points(int n_pt, int n_obs, const int32_t *pt_idx, int world, int32_t *shard_of_point)
{
    if (n_pt < 0 || n_obs < 0 || world < 1 || (n_obs && !pt_idx) || (n_pt && !shard_of_point)) {
        esfm::set_error("esfm_ba_shard_points: bad arguments");
        return ESFM_ERR_INVALID_ARG;
    }
    std::vector<int64_t> cnt((size_t)n_pt, 0);
    for (int k = 0; k < n_obs; ++k) {
        if (pt_idx[k] < 0 || pt_idx[k] >= n_pt) { esfm::set_error("point index out of range"); return ESFM_ERR_INVALID_ARG; }
        cnt[(size_t)pt_idx[k]]++;
    }
    // contiguous ranges with ~n_obs/world observations each: keeps a shard's points (and their
    // observations, which are sorted by point on the device) contiguous
    const double target = world > 0 ? (double)n_obs / world : 0.0;
    int shard = 0;
    int64_t acc = 0;
    for (int p = 0; p < n_pt; ++p) {
        if (shard < world - 1 && (double)acc >= target * (shard + 1)) ++shard;
        shard_of_point[p] = shard;
        acc += cnt[(size_t)p];
    }
    return ESFM_OK;
}

// Host-only: the structure-aware plan of the reduced camera system for this observation list (ba_sparse_plan.hpp) -- what
// esfm_ba_problem_solve builds for itself; exported for tests and tools.
int esfm_ba_reduced_plan(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, int leaf_max,
                         int32_t *col_src, int col_cap, int32_t *tiles, int tile_cap, int32_t *info)
{
    if (n_cam < 0 || n_pt < 0 || n_obs < 0 || (n_obs && (!cam_idx || !pt_idx)) || !info || col_cap < 0 || tile_cap < 0) {
        esfm::set_error("esfm_ba_reduced_plan: bad arguments");
        return ESFM_ERR_INVALID_ARG;
    }
    std::vector<int32_t> pt_start((size_t)n_pt + 1, 0), s_cam((size_t)n_obs);
    for (int k = 0; k < n_obs; ++k) {
        if (cam_idx[k] < 0 || cam_idx[k] >= n_cam || pt_idx[k] < 0 || pt_idx[k] >= n_pt) { esfm::set_error("observation index out of range"); return ESFM_ERR_INVALID_ARG; }
        pt_start[(size_t)pt_idx[k] + 1]++;
    }
    for (int p = 0; p < n_pt; ++p) pt_start[(size_t)p + 1] += pt_start[(size_t)p];
    {
        std::vector<int32_t> fill(pt_start.begin(), pt_start.end() - 1);
        for (int k = 0; k < n_obs; ++k) s_cam[(size_t)fill[(size_t)pt_idx[k]]++] = cam_idx[k];
    }
    const esfm::CamGraph g = esfm::cam_graph_from_tracks(n_cam, n_pt, pt_start.data(), s_cam.data());
    const esfm::SparsePlan pl = esfm::make_sparse_plan(g, leaf_max > 0 ? leaf_max : 32);
    info[0] = pl.nb; info[1] = (int32_t)pl.tiles.size(); info[2] = pl.chain; info[3] = pl.dense_nb; info[4] = pl.worthwhile() ? 1 : 0;
    info[5] = (int32_t)std::min<long long>(pl.update_steps, INT32_MAX); info[6] = (int32_t)pl.node_kind.size(); info[7] = (int32_t)pl.wgs.size();
    info[8] = (int32_t)(g.adj.size() / 2 + (size_t)n_cam);       // camera blocks (a, b <= a) that can be non-zero: what several ranks exchange
    info[9] = 0;
    if ((size_t)col_cap < pl.col_src.size() || (size_t)tile_cap < pl.tiles.size()) {
        if (col_src || tiles) { esfm::set_error("esfm_ba_reduced_plan: output arrays too small (need %zu columns, %zu tiles)", pl.col_src.size(), pl.tiles.size()); return ESFM_ERR_INVALID_ARG; }
        return ESFM_OK;          // sizing call
    }
    if (col_src) for (size_t k = 0; k < pl.col_src.size(); ++k) col_src[k] = pl.col_src[k];
    if (tiles) for (size_t k = 0; k < pl.tiles.size(); ++k) { tiles[2 * k] = pl.tiles[k].I; tiles[2 * k + 1] = pl.tiles[k].J; }
    return ESFM_OK;
}

}  // extern "C"
