// Launch interface between ba_api.cpp (LM control loop on the host) and ba_kernels.hip.
#pragma once

#include "common.hpp"

namespace esfm {

// Scalar accumulators in BADev::scal (doubles).
//   [0, SC_SUM_COUNT)   : partial sums over this rank's observations/points -> SUM all-reduce
//   [SC_REPL0, SC_GMAX) : quantities every rank computes identically (camera blocks) -> no reduce
//   SC_GMAX             : max |gradient| over this rank's points and all cameras  -> MAX all-reduce
enum BAScalar {
    SC_COST = 0, SC_CAND_COST = 1, SC_MODEL_CHANGE = 2, SC_STEP_SQ_PT = 3, SC_CAND_SQ_PT = 4, SC_XNORM_SQ_PT = 5,
    SC_LIN_BAD = 6, SC_CAND_BAD = 7, SC_PT_SINGULAR = 8,
    SC_GDOTD = 9,     // gradient . delta = sum_obs r.(J step): initial slope of the Armijo search (bounded problems)
    SC_LS_GRAD = 10,  // slope of the cost along delta at a line-search trial point
    SC_SUM_COUNT = 12,
    SC_REPL0 = 12, SC_STEP_SQ_CAM = 12, SC_CAND_SQ_CAM = 13, SC_XNORM_SQ_CAM = 14, SC_CHOL_FAIL = 15,
    SC_GMAX = 16,     // max |gradient| (projected gradient step with bounds)
    SC_DMAX = 17,     // max |delta| over this rank's points and all camera-side unknowns
    SC_MAX_COUNT = 2, SC_COUNT = 24
};

// Device-resident bundle-adjustment problem.  Observations are sorted by point (CSR), so a
// point's observations are contiguous: obs k in [pt_start[p], pt_start[p+1]) belongs to point p.
//
// Free shared intrinsics (ReprojectErrorTerm_updatecalib, reference ba.h:170-222): the block fx, cx, fy, cy rides as
// one more 6-wide camera-side block (two padding unknowns with zero Jacobian columns) behind the real cameras, so
// n_cam = n_real_cam + has_calib is the number of 6-wide blocks of the reduced system and every per-block array
// (x_c, scale_c, camacc, red, y_c ...) simply has one more block; observations only ever name real cameras.
struct ScalParts {                             // host-side bookkeeping of one problem
    int n[SC_SUM_COUNT];                       // per-workgroup scalar partials pending on the device, per slot
    bool single_rank = true;                   // no all-reduce callback in the running solve
    bool grad_done = false;                    // the camera part of max|gradient| came with the last linearisation
    // the Schur buffer `red` between ba_schur and the tiled solve (one rank, no free intrinsics): left in FIXED POINT for
    // chol_assemble_kernel, which converts while it reads and leaves zeros behind -- one conversion pass over 75 MB and one memset of
    // it less per LM iteration at 512 cameras
    bool red_fixed = false;                    // red holds fixed-point integers (scaled by qexp / red_rhs_exp), not doubles
    bool red_clean = false;                    // red is all zeros: the next ba_schur need not clear it
    int red_rhs_exp = 0;
};
struct SparseSolve;                            // the structure-aware reduced solve of one problem (ba_chol_sparse.hpp); host object
// does ba_solve_reduced take the tiled path (ba_solve_reduced_large) for this camera count?
bool ba_solve_is_tiled(int n_cam);
struct ScalCounts { int n[SC_SUM_COUNT]; };   // kernel argument: how many partials each slot has pending
struct ScalBase { int b[6]; };                // kernel argument: first partial index of this launch, per slot it commits

struct BADev {
    int n_cam = 0, n_pt = 0, n_obs = 0;
    int n_real_cam = 0, has_calib = 0;
    int constrained = 0;          // any box bound on a camera-side unknown (lo_c / up_c hold +-inf elsewhere)
    // structure (static per problem)
    int32_t *obs_cam = nullptr;   // [n_obs] camera of sorted observation k
    int32_t *obs_pt = nullptr;    // [n_obs] point of sorted observation k
    float2 *obs_uv = nullptr;     // [n_obs] observed pixel (float, reference points_2d_)
    int32_t *pt_start = nullptr;  // [n_pt+1]
    float4 *K4 = nullptr;         // [n_cam] fx, cx, fy, cy
    double *cam_nobs = nullptr;   // [n_cam] observation count over ALL shards (double: all-reduced)
    // parameters
    double *x_c = nullptr, *x_p = nullptr;        // current point   [6 n_cam], [3 n_pt]
    double *cand_c = nullptr, *cand_p = nullptr;  // candidate point
    double *x0_p = nullptr;                       // points at solve() entry (multi-GPU merge)
    // linearisation (loss-corrected, column-scaled), SoA: J[a * n_obs + k]
    double *Jc = nullptr;   // 12 arrays: row 0 cols 0..5, then row 1 cols 0..5
    double *Jp = nullptr;   // 6 arrays: row 0 cols 0..2, row 1 cols 0..2
    double *res = nullptr;  // 2 arrays
    double *Jk = nullptr;   // has_calib, 4 arrays: d r0/d fx, d r0/d cx, d r1/d fy, d r1/d cy (the other four are 0)
    double *lo_c = nullptr, *up_c = nullptr;          // box bounds [6 n_cam]
    double *delta_c = nullptr, *delta_p = nullptr;    // the LM step in parameter units, [6 n_cam], [3 n_pt]
    double *scale_c = nullptr, *scale_p = nullptr;  // Jacobi scaling [6 n_cam], [3 n_pt]
    double *EtE = nullptr;   // [6 n_pt] E'E upper (xx,xy,xz,yy,yz,zz); its diagonal = point column norms
    double *Etr = nullptr;   // [3 n_pt]
    double *Minv = nullptr;  // [6 n_pt] (E'E + D_p^2)^-1
    double *Aig = nullptr;   // [3 n_pt] Minv * Etr
    // per-camera normal-equation pieces, one contiguous SUM all-reduce buffer:
    //   camacc = FtF (36 per camera, full symmetric) | Ftr (6 per camera)
    double *camacc = nullptr;
    // Schur part of the reduced system, one contiguous SUM all-reduce buffer:
    //   red = S_schur (n*n, only blocks with row camera >= column camera are written) | rhs_corr (n)
    double *red = nullptr;
    double *red_packed = nullptr;   // [ba_red_packed_doubles] exchange buffer (allocated on the first sharded solve)
    double *y_c = nullptr;   // [6 n_cam] solution of the reduced system
    double *scal = nullptr;  // [SC_COUNT]
    double *chol = nullptr;  // [(n+1)(n+2)/2] packed-lower work matrix for large n
    double *slabs = nullptr; // per-workgroup Schur slabs (small n_cam only)
    size_t slab_cap = 0;
    // windowed Schur (n_cam too large for the LDS-resident matrix): points ordered by lowest camera, cut in chunks
    int n_chunks = 0;
    int32_t *slot_obs = nullptr;    // [n_obs] observation indices in chunk order
    int32_t *chunk_slot = nullptr;  // [n_chunks+1] first slot of each chunk
    int32_t *chunk_cam0 = nullptr;  // [n_chunks] lowest camera of the chunk = window base
    int32_t *slot_obs_b = nullptr, *chunk_slot_b = nullptr, *chunk_cam0_b = nullptr;   // the same for the pass on rotated camera indices
    int n_chunks_b = 0;
    int32_t *wide_obs = nullptr;    // [n_wide_obs] observations of the points whose cameras span kSchurWinCams or more (not in slot_obs)
    int n_wide_obs = 0;
    // matrix-core Schur (ba_schur_mfma_kernel): the points whose cameras span at most kSchurMfCams indices (plain or rotated by half
    // the camera count: tables [0] and [1]), ordered by lowest camera and cut into chunks with one window base; a chunk's
    // observations in batches of whole points, <= 64 observations and <= 16 points each
    int n_mchunks[2] = {0, 0};
    int32_t *mslot_obs[2] = {nullptr, nullptr};      // observation indices in chunk / batch order
    int32_t *mslot_pc[2] = {nullptr, nullptr};       // [2 x slots] the observation's point and camera (one dependent load less per batch)
    int32_t *mbatch_slot[2] = {nullptr, nullptr};    // [n_batches + 1] first slot of each batch
    int32_t *mchunk_batch0[2] = {nullptr, nullptr};  // [n_mchunks + 1] first batch of each chunk
    int32_t *mchunk_cam0[2] = {nullptr, nullptr};    // [n_mchunks] window base (rotated index for table 1)
    // per-camera sums F'F / F'r of LARGE camera counts (the small ones are summed inside the sweep, see ba_linearize_kernel): the observations of every camera in ascending order (cam_obs, a CSR over
    // cameras built once per problem), cut into chunks of kCamChunk; one wave sums a chunk in a fixed order, a second
    // launch adds a camera's chunk sums in order.  Bit-reproducible whatever the launch timing.
    double *lin_slabs = nullptr;        // per-workgroup F'F / F'r slabs of the Jacobian sweep (27 n_cam doubles each), small camera counts
    size_t lin_slab_cap = 0;
    int32_t *cam_obs = nullptr;         // [n_obs]
    int32_t *cchunk_cam = nullptr, *cchunk_beg = nullptr, *cchunk_end = nullptr;   // [n_cchunks]
    int32_t *cam_chunk0 = nullptr;      // [n_real_cam + 1] first chunk of each camera
    int n_cchunks = 0;
    double *cam_part = nullptr;         // [n_cchunks][kCamPart]: 27 camera sums + 10 intrinsics-block sums per chunk
    // scalar sums across workgroups: per-workgroup partials, added in index order at the next read-back (scal_commit)
    double *scal_part = nullptr;        // [SC_SUM_COUNT][scal_cap]
    int scal_cap = 0;
    ScalParts *parts = nullptr;         // HOST memory (esfm_ba_problem): partials pending per slot; never dereferenced on the device
    SparseSolve *sparse = nullptr;      // HOST memory: set while the running solve takes the structure-aware reduced solve (ba_chol_sparse.hip)
    // Exact, hence order-independent, accumulation of the Schur complement: d.red is accumulated as 64-bit FIXED-POINT integers
    // (LDS / global u64 atomics, slab sums: integer addition is associative) and converted to f64 once, before the solve.  Entry
    // (r, c) is scaled by 2^(60 - qexp[r] - qexp[c]) with sqrt(diag(F'F)_i) < 2^qexp[i]:  Cauchy-Schwarz with E M^-1 E' <= I bounds
    // every partial sum  |sum_p f_r' E M^-1 E' f_c| <= sqrt(diag(F'F)_r diag(F'F)_c) < 2^(qexp[r] + qexp[c]),  so the integers stay
    // below 2^60; the right-hand side uses |residual| <= sqrt(2 cost) < 2^rhs_exp in place of the column factor.  The grid is 2^-60
    // of the entry's natural scale -- 7 bits finer than f64 -- so nothing is lost against f64 atomics (a common f64 grid, i.e.
    // (v + 1.5 2^k) - 1.5 2^k, was tried first: it costs 3 digits of the parameters).  Refreshed by ba_linearize from
    // this rank's observations.
    int32_t *qexp = nullptr;            // [6 n_cam]
    // point chunks of the back-substitution (and of the per-point normal blocks): consecutive points, <= 256 observations each
    int32_t *pchunk_pt0 = nullptr;      // [n_pchunks + 1]
    int4 *pchunk_info = nullptr;        // [n_pchunks] {first point, end point, first observation, end observation}: what a chunk's workgroup needs before it can request anything else, in ONE load
    int n_pchunks = 0;
};
constexpr int kCamChunk = 256, kCamPart = 37;
constexpr int kPtChunkObs = 256;               // observations per point chunk (back-substitution, per-point normal blocks)
constexpr int kSchurWinCams = 28;              // cameras in the windowed Schur kernel's LDS window
constexpr int kSchurMfCams = 13;               // cameras in the matrix-core Schur kernel's window (80 rows = 5 MFMA block rows)

inline size_t ba_camacc_doubles(int n_cam) { return (size_t)42 * (size_t)n_cam; }
inline size_t ba_red_doubles(int n_cam) { const size_t n = 6 * (size_t)n_cam; return n * n + n; }
// what travels between GPUs: the block-lower-triangular part of S row by row, then the right-hand side
inline size_t ba_red_packed_doubles(int n_cam) { return (size_t)18 * (size_t)n_cam * ((size_t)n_cam + 1) + 6 * (size_t)n_cam; }
int ba_red_pack(hipStream_t st, const BADev &d, double *packed, bool unpack);

// deferred_slabs != NULL: on one rank with the walking per-point kernel next, the reduction of the sweep's per-camera slabs is NOT
// launched; *deferred_slabs (> 0) must then be handed to the ba_point_prep call that follows, which does both in one launch
// cost_bound > 0: an upper bound on the cost at the point being linearised (the accepted candidate's), which lets the sweep of a large
// problem use the previous linearisation's fixed-point exponents in ONE pass (ba_linearize_kernel)
int ba_linearize(hipStream_t st, const BADev &d, int num_cu, double cauchy_a, bool use_scaling, esfm_ctx *timing_ctx, int *deferred_slabs = nullptr,
                 double cost_bound = -1.0);
int ba_point_prep(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag, bool fresh_jacobian, int deferred_slabs = 0);
int ba_jacobi_scaling(hipStream_t st, const BADev &d);
int ba_camera_gradient(hipStream_t st, const BADev &d);
// slabs: scratch for the LDS-privatised variant (n_cam small), >= ba_schur_slab_doubles(n_cam, num_cu) doubles, or NULL
int ba_schur(hipStream_t st, const BADev &d, int num_cu, double *slabs, size_t slab_capacity_doubles, double rhs_bound);
inline size_t ba_schur_slab_doubles(int n_cam, int num_cu)
{
    const size_t per = (size_t)n_cam * (n_cam + 1) / 2 * 36 + 6 * (size_t)n_cam;
    return per * sizeof(double) <= 156 * 1024 ? per * (size_t)num_cu : 0;
}
int ba_solve_reduced(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag);
// multi-workgroup blocked Cholesky for systems beyond one workgroup's LDS (ba_chol_large.hip); d.chol must hold
// ba_chol_large_doubles(n_cam) doubles
int ba_solve_reduced_large(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag);
size_t ba_chol_large_doubles(int n_cam);
// one-workgroup MFMA solve for n = 6 n_cam <= 176 (ba_chol_large.hip)
bool ba_chol_small_fits(int n_cam);
int ba_solve_reduced_small(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag);
// intrinsics row/column block of the reduced system (has_calib): runs after ba_schur, adds into d.red
int ba_schur_calib(hipStream_t st, const BADev &d, double rhs_bound);
// the pending per-workgroup partials -> d.scal (ba_publish_scalars does it itself; needed before an all-reduce of d.scal)
int ba_scal_reduce(hipStream_t st, const BADev &d);
// forget the pending partials of slots [first, end): the companion of a memset of those d.scal slots
void ba_scal_discard(const BADev &d, int first_slot, int end_slot);
int ba_publish_scalars(hipStream_t st, const BADev &d, double *host, unsigned long long *flag, unsigned long long seq);
#ifdef __HIPCC__
constexpr int kFxBits = 60;   // |v| <= B < 2^e  ->  |v 2^(kFxBits - e)| < 2^60: three bits of head-room in an int64
__device__ __forceinline__ double fx64_to_double(unsigned long long q, int sh) { return ldexp((double)(long long)q, -sh); }
// Candidate cameras from the reduced solve's y (one workgroup of up to 1024 threads; lds: >= 48 doubles): x + (-y) .* scaling for cameras that
// have observations, projected onto the box when the problem is bounded; step / candidate norms and max |delta|.
__device__ inline void ba_camera_step_body(const BADev &d, const double *y, double *lds)
{
    double ssq = 0.0, csq = 0.0, dmax = 0.0;
    // four strides' loads in flight at once (512 cameras and 256 threads: twelve dependent round trips, 14 us, as a plain loop); a
    // thread still adds its entries in index order, so the sums keep their bits
    constexpr int U = 4;
    const int n6 = 6 * d.n_cam;
    for (int i0 = threadIdx.x; i0 < n6; i0 += U * blockDim.x) {
        double nobs[U], xs[U], ys[U], sc[U], lo[U], up[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * blockDim.x;
            const bool ok = i < n6;
            nobs[u] = ok ? d.cam_nobs[i / 6] : 0.0; xs[u] = ok ? d.x_c[i] : 0.0; ys[u] = ok ? y[i] : 0.0; sc[u] = ok ? d.scale_c[i] : 0.0;
            lo[u] = (ok && d.constrained) ? d.lo_c[i] : 0.0; up[u] = (ok && d.constrained) ? d.up_c[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i >= n6) break;
            const bool active = nobs[u] > 0.0;
            const double x = xs[u];
            const double dl = active ? (-ys[u]) * sc[u] : 0.0;
            double cnd = active ? x + dl : x;
            if (d.constrained) {
                // ParameterBlock::Plus projects onto the box (lower bound first) [upstream parameter_block.h]
                if (active) cnd = fmin(fmax(cnd, lo[u]), up[u]);
                d.delta_c[i] = dl;
                dmax = fmax(dmax, fabs(dl));
            }
            d.cand_c[i] = cnd;
            if (active) { const double df = x - cnd; ssq += df * df; csq += cnd * cnd; }
        }
    }
    // 6 n_cam entries only: every thread past them holds zeros, so the sums are those of the first ceil(6 n_cam / 64) waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ssq += __shfl_xor(ssq, o); csq += __shfl_xor(csq, o); dmax = fmax(dmax, __shfl_xor(dmax, o)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { lds[wave] = ssq; lds[16 + wave] = csq; lds[32 + wave] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, m = 0.0;
        for (int w = 0; w < nw; ++w) { a += lds[w]; b += lds[16 + w]; m = fmax(m, lds[32 + w]); }
        d.scal[SC_STEP_SQ_CAM] = a; d.scal[SC_CAND_SQ_CAM] = b;
        if (d.constrained && m > 0.0) atomicMax(reinterpret_cast<unsigned long long *>(&d.scal[SC_DMAX]), (unsigned long long)__double_as_longlong(m));
    }
}
#endif

int ba_camera_step(hipStream_t st, const BADev &d);
// with_cost: also 1/2 sum rho of the candidate into SC_CAND_COST / SC_CAND_BAD (what ba_cost(cand_c, cand_p) would add)
int ba_backsub(hipStream_t st, const BADev &d, bool with_cost, double cauchy_a);
// with_slope: also the derivative of the cost along (delta_c, delta_p) at (cams, pts) into SC_LS_GRAD
int ba_cost(hipStream_t st, const BADev &d, int num_cu, const double *cams, const double *pts, double cauchy_a, int slot, int bad_slot,
            bool with_slope = false);
// candidate = Plus(x, t * delta) (projected onto the box) and its norms
int ba_take_step(hipStream_t st, const BADev &d, double t);
// x_c <- projection of x_c onto the box
int ba_project_cameras(hipStream_t st, const BADev &d);
int ba_param_sqnorm(hipStream_t st, const BADev &d);
int ba_points_delta(hipStream_t st, const BADev &d, bool to_delta);

}  // namespace esfm
