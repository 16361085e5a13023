/*
 * esfm.h -- C ABI of the MI355X-native EasySFM hot path (libesfm_hip.so).
 *
 * The reference (YuePanEdward/EasySFM) has no plugin/FFI layer: its de-facto
 * boundary for this path is three C++ member functions called from
 * cpp_code/test/sfm.cpp (the pair loop :140-161 and the BA sites :262,:314,:325):
 *
 *   FeatureMatching::matchFeaturesORB   cpp_code/include/feature_matching.h:17-18
 *   FeatureMatching::matchFeaturesSURF  cpp_code/include/feature_matching.h:20-21
 *   BundleAdjustment::doSFMBA           cpp_code/include/ba.h:84
 *
 * Every entry point below is what a binding for one of those call sites would
 * bind (plain pointers and sizes, no C++/torch/OpenCV types).  INTEGRATION.md
 * shows the few lines a maintainer adds to feature_matching.cpp / ba.cpp.
 *
 * Conventions
 *   - return value: ESFM_OK (0) or a negative esfm_status; nothing throws.
 *   - esfm_last_error() returns a thread-local, human-readable message for
 *     the last failing call on this thread.
 *   - "host" pointers are ordinary CPU memory; "_dev" entry points take HIP
 *     device pointers that are already resident in HBM and enqueue all work on
 *     the context's stream without synchronising (the caller synchronises).
 *     Host buffers travel in 512-KiB pieces through the runtime's staging
 *     buffer (the runtime would otherwise pin buffers of 1 MiB and more in
 *     place, which stalls the process' queues when the heap around them
 *     changes); a buffer the caller has registered (hipHostRegister) or
 *     allocated with hipHostMalloc goes in one DMA transfer -- worth doing for
 *     images and descriptor banks that are handed over repeatedly.
 *   - one esfm_ctx per host thread and per GPU; a context is not thread-safe.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point fails with ESFM_ERR_NO_DEVICE.
 */
#ifndef ESFM_H_
#define ESFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESFM_VERSION_MAJOR 0
#define ESFM_VERSION_MINOR 1

typedef enum esfm_status {
    ESFM_OK = 0,
    ESFM_ERR_INVALID_ARG = -1,
    ESFM_ERR_NO_DEVICE = -2,   /* no HIP device / not gfx950 / HIP runtime error at init */
    ESFM_ERR_HIP = -3,         /* a HIP runtime call failed; see esfm_last_error()        */
    ESFM_ERR_OOM = -4,
    ESFM_ERR_UNSUPPORTED = -5, /* e.g. descriptor width the kernels are not built for     */
    ESFM_ERR_NUMERIC = -6,     /* BA: non-finite input or an unusable linear system       */
    ESFM_ERR_COMM = -7,        /* BA: the all-reduce callback reported failure             */
    ESFM_ERR_STALE_PREPARED = -8 /* matching: a prepared descriptor buffer no longer holds the rows it was prepared from
                                    (only detected under esfm_ctx_set_prepared_check)       */
} esfm_status;

typedef struct esfm_ctx esfm_ctx;

/* ---- library / context -------------------------------------------------- */

/* "major.minor" of the ABI above. */
const char *esfm_version(void);
/* Message for the last error raised on the calling thread ("" if none). */
const char *esfm_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int esfm_device_count(void);

/* Creates a context on HIP device `device`.  `hip_stream` is a hipStream_t the
 * caller owns (e.g. torch.cuda.current_stream().cuda_stream) or NULL, in which
 * case the context creates and owns a non-blocking stream. */
int esfm_ctx_create(int device, void *hip_stream, esfm_ctx **out);
int esfm_ctx_destroy(esfm_ctx *ctx);
/* Blocks until everything enqueued on the context's stream has finished. */
int esfm_ctx_synchronize(esfm_ctx *ctx);
/* The hipStream_t the context enqueues on (for HIP-event timing by the caller). */
void *esfm_ctx_stream(esfm_ctx *ctx);

/* Per-kernel device timing (measurement only; off by default).  When enabled,
 * the library brackets each launch of the kernels below with hipEvents on the
 * context's stream.  esfm_ctx_kernel_time() synchronises the stream, adds the
 * elapsed times of all launches since the last call for that kernel to
 * *total_ms / *launches (caller zero-initialises) and recycles the events. */
typedef enum esfm_kernel_id {
    ESFM_K_L2_KNN = 0,        /* l2_knn_bf16x1_kernel (dim 64) / l2_knn_mfma_kernel: MFMA distance pass + fused top-k + re-rank */
    ESFM_K_HAMMING_KNN = 1,   /* hamming_fp4_kernel (256 bit; + its expansion when the buffer is not prepared) / hamming_knn_mfma_kernel / hamming_knn_kernel */
    ESFM_K_BA_LINEARIZE = 2,  /* ba_linearize_kernel: the Jacobian sweep                        */
    ESFM_K_BA_SCHUR = 3,      /* ba_schur_kernel                                                */
    ESFM_K_BA_SOLVE = 4,      /* ba_chol_solve_kernel                                           */
    ESFM_K_L2_RESCAN = 5,     /* l2_exact_scan_kernel                                           */
    ESFM_K_SOR_KNN = 6,       /* sor_knn_mean_kernel: k-NN mean distances of the outlier filter   */
    ESFM_K_TRIANGULATE = 7,   /* triangulate_dlt_kernel                                         */
    ESFM_K_RANSAC = 8,        /* essential_setup + _roots + _score kernels (one chunk)         */
    ESFM_K_SURF_DET = 9,      /* surf_det_trace_kernel                                          */
    ESFM_K_SURF_DESC = 10,    /* the four descriptor launches: surf_orient / window / rowsum / vector */
    ESFM_K_UNDISTORT = 11,    /* undistort_remap_kernel                                         */
    ESFM_K_ORB_FAST = 12,     /* orb_fast_kernel: FAST-9/16 score of every pyramid pixel          */
    ESFM_K_L2_SECOND = 13,    /* l2_finish_kernel: everything behind the one-product pass (re-rank of the ratio screen's survivors, threshold filter, ratio test, compaction) */
    ESFM_K_COUNT = 14
} esfm_kernel_id;
int esfm_ctx_set_kernel_timing(esfm_ctx *ctx, int enable);
int esfm_ctx_kernel_time(esfm_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);

/* ---- pairwise matching ---------- */

/*
 * 2-NN search, the replacement for
 *   matcher.knnMatch(q.descriptors, t.descriptors, nn2, 2)
 * at cpp_code/src/feature_matching.cpp:125 (SURF, float L2; exact brute force
 * as in python_code/feature_match.py:33-34) and :80 (ORB, "BruteForce-Hamming").
 *
 * Output for query row i: idx[2*i+k], dist[2*i+k], k = 0 (nearest), 1 (second).
 * Ordering rule: ascending (distance, train index); ties go to the lower
 * train index.  L2 distance = sqrtf of the float squared distance summed in the
 * order documented in oracle/match_ref.c; Hamming distance is the exact bit
 * count as float.  Missing neighbours (nt < 2) are reported as idx -1,
 * dist FLT_MAX.  Host pointers.
 */
int esfm_knn2_l2_f32(esfm_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim,
                     int32_t *idx /*2*nq*/, float *dist /*2*nq*/);
int esfm_knn2_hamming(esfm_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                      int32_t *idx /*2*nq*/, float *dist /*2*nq*/);

/*
 * Whole body of FeatureMatching::matchFeaturesSURF / matchFeaturesORB
 * (cpp_code/src/feature_matching.cpp:115-142 / :71-97) minus printing and GUI:
 * 2-NN, then keep query i iff (double)d0 < ratio * (double)d1 (:133 / :88),
 * survivors in ascending query order.  Writes at most nq entries to each output
 * array (cv::DMatch::queryIdx, ::trainIdx, ::distance) and the count to *n_out.
 * With nt < 2 (undefined behaviour in the reference) nothing is emitted.
 * Host pointers.
 */
int esfm_match_l2_f32(esfm_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim, double ratio,
                      int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out);
int esfm_match_hamming(esfm_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                       double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out);

/*
 * Batched form of the pair loop cpp_code/test/sfm.cpp:140-161: all descriptor
 * sets live in ONE device buffer (`desc_dev`, rows concatenated, row stride =
 * dim floats or nbytes bytes); set s owns rows [set_row_offset[s],
 * set_row_offset[s+1]).  pairs[2*p] = query set, pairs[2*p+1] = train set
 * (the reference uses query = later frame i, train = earlier frame j < i).
 *
 * Outputs are device buffers: pair p owns the slice
 * [out_offset[p], out_offset[p] + nq_p) of query_idx/train_idx/distance, of
 * which the first n_out[p] entries are valid, query-ascending.  out_offset is
 * a HOST array of n_pairs+1 entries filled by the call (exclusive prefix sum
 * of nq_p), so the device arrays need sum(nq_p) entries.
 * set_row_offset and pairs are HOST arrays.  Work is enqueued on the context's
 * stream; the call does not synchronise.
 */
typedef enum esfm_metric { ESFM_L2_F32 = 0, ESFM_HAMMING = 1 } esfm_metric;

int esfm_match_pairs_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev,
                         const int32_t *set_row_offset, int n_sets, int width /*dim or nbytes*/,
                         const int32_t *pairs, int n_pairs, double ratio,
                         int32_t *query_idx_dev, int32_t *train_idx_dev, float *distance_dev,
                         int32_t *n_out_dev /*n_pairs*/, int64_t *out_offset /*host, n_pairs+1*/);

/* The same with HOST pointers in and out (what a C++ host without a device allocator of its own calls once for the whole pair loop,
 * sfm.cpp:140-161): `desc_host` holds the rows of all sets back to back; they are uploaded once, prepared (below) once, every pair is
 * matched in one launch sequence and the per-pair slices come back in one read-back.  query_idx / train_idx / distance need
 * sum(nq_p) entries, n_out n_pairs, out_offset n_pairs + 1 (all host).  Synchronises.
 * Only the first n_out[p] entries of a pair's range are written (sparse results are packed on the device before the read-back: the
 * transfer is proportional to the matches, not to the queries); the rest of the range is left as the caller passed it. */
int esfm_match_pairs(esfm_ctx *ctx, esfm_metric metric, const void *desc_host,
                     const int32_t *set_row_offset, int n_sets, int width /*dim or nbytes*/,
                     const int32_t *pairs, int n_pairs, double ratio,
                     int32_t *query_idx, int32_t *train_idx, float *distance,
                     int32_t *n_out /*n_pairs*/, int64_t *out_offset /*n_pairs+1*/);

/*
 * Optional, once per resident descriptor buffer: derive and keep what the matcher computes from the rows before it can start --
 * for 64-float L2 descriptors the bf16 operand images, |row|^2 and the rounding residual norms of every row (l2_split_bf16_kernel),
 * for 32-byte Hamming descriptors the nibble-per-bit images of the FP4 matrix-core form -- so that the esfm_match_pairs_dev / esfm_knn2_pairs_dev calls that follow
 * on the SAME (desc_dev, total rows, width) skip that launch.  The reference has no counterpart: it re-reads cv::Mat rows in every
 * knnMatch call (feature_matching.cpp:80,125); here the frames' descriptors are uploaded once for the whole pair loop
 * (sfm.cpp:140-161) and this is part of the upload.  The caller promises not to modify the rows while they are prepared;
 * esfm_match_release_prepared, another prepare, or a match call on a different buffer ends it.  Other widths: a no-op.
 */
int esfm_match_prepare_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, int64_t total_rows, int width);
int esfm_match_release_prepared(esfm_ctx *ctx);
/* The same, but only if `desc_dev` IS the buffer currently prepared on this context (a no-op otherwise): what an owner of one of
 * several descriptor buffers calls when ITS buffer goes away, without taking the prepared state from whoever holds it now. */
int esfm_match_release_prepared_buffer(esfm_ctx *ctx, const void *desc_dev);
/* Which buffer is prepared on this context right now (NULL: none). */
int esfm_match_prepared_buffer(esfm_ctx *ctx, const void **desc_dev_out);
/* The prepared state is keyed on (desc_dev, metric, total_rows, width) and there is ONE per context: a second prepare replaces the
 * first.  The library cannot see a hipFree or an in-place rewrite, and a same-size hipMalloc routinely returns the address just
 * freed: a prepared buffer MUST be released (esfm_match_release_prepared) or prepared again BEFORE it is freed, rewritten or
 * replaced -- otherwise the next match call on that address runs on the old rows' operand images and returns wrong matches
 * without an error.  esfm_ctx_set_prepared_check(ctx, 1) is the debugging aid for exactly that: prepare then also keeps a 64-bit
 * fingerprint of the buffer, and every match call that is about to rely on the prepared operands re-derives it first (one read of
 * the buffer and one host round trip per call) and fails with ESFM_ERR_STALE_PREPARED -- ending the prepared state -- when the rows
 * have changed.  Off by default; the environment variable ESFM_CHECK_PREPARED=1 switches it on for every new context. */
int esfm_ctx_set_prepared_check(esfm_ctx *ctx, int enable);

/* Same pass, but returns the raw 2-NN table instead of the filtered list:
 * knn_idx_dev / knn_dist_dev hold 2 entries per query row, pair p at
 * [2*out_offset[p], 2*(out_offset[p]+nq_p)). */
int esfm_knn2_pairs_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev,
                        const int32_t *set_row_offset, int n_sets, int width,
                        const int32_t *pairs, int n_pairs,
                        int32_t *knn_idx_dev, float *knn_dist_dev, int64_t *out_offset);

/* Audit of the Hamming matcher's ratio screen (tests only; metric must be ESFM_HAMMING -- the L2 screen is audited through
 * esfm_ctx_set_l2_audit mode 4): the raw table of a pass that screens with `ratio` as esfm_match_pairs_dev's does.  A query
 * the pass dropped as "cannot pass d0 < ratio d1" (feature_matching.cpp:88) carries train index -2 in both slots; every other
 * query its exact 2-NN.  Every dropped query must fail the reference's test on the unscreened table. */
int esfm_knn2_pairs_screened_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev,
                                 const int32_t *set_row_offset, int n_sets, int width,
                                 const int32_t *pairs, int n_pairs, double ratio,
                                 int32_t *knn_idx_dev, float *knn_dist_dev, int64_t *out_offset);

/* Counters of the last L2 batched call on this context (after a synchronise):
 * queries whose MFMA candidate list could not be certified and were re-scanned
 * exactly (see DESIGN.md "certified re-rank").  For tests and profiling. */
int esfm_match_last_stats(esfm_ctx *ctx, int64_t *n_queries, int64_t *n_rescanned);
/* 64-float descriptors: queries the one-product bf16 pass could not certify and handed to the threshold-filter
 * pass (inside l2_finish_kernel), of which n_rescanned went on to the exact re-scan.  0 for other widths.  Queries the
 * ratio screen dropped (esfm_match_*: provably d0 >= ratio d1) are in neither count. */
int esfm_match_last_second_pass(esfm_ctx *ctx, int64_t *n_second_pass);

/* Audit of the L2 certificate (tests only; the default mode 0 is the product path).
 *   mode 1: the MFMA pass runs but the exact re-scan of uncertified queries is SKIPPED, so the
 *           2-NN table holds the pass's own answer for every query;
 *   mode 2: every query is brute-forced by l2_exact_scan_kernel (no MFMA pass);
 *   mode 3: (64-float descriptors) the one-product bf16 pass ALONE: the table holds its answer for every query and
 *           esfm_match_last_flagged() lists what it could not certify (mode 1 audits the two MFMA passes together);
 *   mode 4: (64-float descriptors, esfm_match_pairs_dev) the one-product pass alone WITH its ratio screen:
 *           esfm_match_last_flagged() lists the queries it dropped as "cannot pass d0 < ratio d1" (the reference's
 *           test, feature_matching.cpp:133).  Every listed query must fail that test on the brute-force table
 *           (rejected_but_would_pass == 0).  The match lists of a mode-4 call are not final (no second pass).
 * Diffing the two tables row by row and removing the rows esfm_match_last_flagged() lists gives
 * the number of queries the certificate accepted with a wrong answer; it must be 0. */
int esfm_ctx_set_l2_audit(esfm_ctx *ctx, int mode);
/* The 16 device-side counters of the last L2 batched call ([0] re-scanned, [1] second pass, the rest: instrumented builds only). */
int esfm_match_debug_counters(esfm_ctx *ctx, int32_t *out16);
/* The (pair, query row) entries the last L2 batched call flagged as uncertified: writes
 * min(*n, cap) entries of 2 x int32 to `out` (host) and the count to *n.  Synchronises. */
int esfm_match_last_flagged(esfm_ctx *ctx, int32_t *out, int64_t cap, int64_t *n);

/* Host-only helper (no GPU needed): the (i, j<i) pair list of sfm.cpp:140-143
 * for n_frames frames, restricted to shard `rank` of `world` by a cost-balanced
 * partition (cost = nq*nt when rows_per_frame is given, else 1).  Writes pairs
 * (query=i, train=j) to pairs_out (capacity n_frames*(n_frames-1)/2 pairs) and
 * returns the number written, or a negative esfm_status. */
int esfm_shard_pair_list(int n_frames, const int32_t *rows_per_frame /*or NULL*/, int rank, int world,
                         int32_t *pairs_out);

/* ---- bundle adjustment (SURVEY.md section 8 rows a-4 .. a-8) ------------- */

/* Solver options.  Defaults (esfm_ba_options_default) are what
 * cpp_code/src/ba.cpp:146-151,201-204 sets plus Ceres' own defaults for
 * TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_SCHUR ([upstream]). */
typedef struct esfm_ba_options {
    int32_t max_num_iterations;            /* 50      ba.cpp:202                     */
    int32_t jacobi_scaling;                /* 1                                      */
    int32_t max_num_consecutive_invalid_steps; /* 5                                  */
    int32_t verbose;                       /* 1 = print Ceres-like progress lines     */
    double cauchy_a;                       /* 0.5     ba.cpp:150; <= 0: squared loss  */
    double initial_trust_region_radius;    /* 1e4                                    */
    double max_trust_region_radius;        /* 1e16                                   */
    double min_trust_region_radius;        /* 1e-32                                  */
    double min_relative_decrease;          /* 1e-3                                   */
    double min_lm_diagonal;                /* 1e-6                                   */
    double max_lm_diagonal;                /* 1e32                                   */
    double function_tolerance;             /* 1e-6                                   */
    double gradient_tolerance;             /* 1e-10                                  */
    double parameter_tolerance;            /* 1e-8                                   */
} esfm_ba_options;

typedef enum esfm_ba_termination {
    ESFM_BA_CONVERGENCE = 0,     /* a tolerance was reached                   */
    ESFM_BA_NO_CONVERGENCE = 1,  /* max_num_iterations reached                */
    ESFM_BA_FAILURE = 2          /* too many invalid steps / numeric failure  */
} esfm_ba_termination;

typedef struct esfm_ba_iteration {
    int32_t iteration;
    int32_t step_is_valid;
    int32_t step_is_successful;
    int32_t line_search_steps;   /* Armijo contractions this iteration (bounds-constrained problems only) */
    double cost;                 /* as Ceres logs it: candidate cost on a rejected step */
    double cost_change;
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;    /* "tr_ratio" */
    double trust_region_radius;  /* radius AFTER this iteration's update */
    double model_cost_change;
} esfm_ba_iteration;

#define ESFM_BA_MAX_LOG 256

typedef struct esfm_ba_summary {
    int32_t termination;         /* esfm_ba_termination */
    int32_t num_iterations;      /* entries in `iterations` minus 1 = LM iterations run */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t num_active_cameras;  /* cameras / points with at least one observation */
    int32_t num_active_points;
    double initial_cost;
    double final_cost;
    double solve_seconds;        /* wall time of the LM loop (inputs already resident) */
    esfm_ba_iteration iterations[ESFM_BA_MAX_LOG]; /* [0] is iteration 0 */
} esfm_ba_summary;

void esfm_ba_options_default(esfm_ba_options *opt);

/* In-place all-reduce over `count` doubles at device pointer `buf_dev`, ordered
 * on `hip_stream`; op = ESFM_REDUCE_SUM or ESFM_REDUCE_MAX.  Return 0 on success.
 * Used only when the caller shards observations over several GPUs (one rank per
 * GPU); NULL = single GPU.  A torch.distributed (RCCL) implementation is in
 * easysfm_amd/ba.py; the library's own RCCL one is esfm_comm_allreduce below.
 * Per LM iteration the solver issues one SUM over the reduced camera system,
 * packed block-lower-triangular (36 n_cam (n_cam + 1) / 2 + 6 n_cam doubles:
 * 11.8 k at 25 cameras, 4.73 M = 37.8 MB at 512), one SUM over the
 * per-camera F'F / F'r blocks (42 n_cam doubles, accepted steps only), and SUM /
 * MAX over a handful of scalars. */
#define ESFM_REDUCE_SUM 0
#define ESFM_REDUCE_MAX 1
typedef int (*esfm_allreduce_fn)(void *user, double *buf_dev, int64_t count, int op, void *hip_stream);

/* The library's own exchange: RCCL over xGMI, one communicator per rank (= per GPU, per esfm_ctx).
 * Rank 0 calls esfm_comm_get_unique_id and hands the ESFM_COMM_ID_BYTES bytes to the other ranks by whatever the
 * host program has (a file, MPI, a torch.distributed store); every rank then calls esfm_comm_create with the same
 * id (collective: returns once all `world` ranks have joined).  esfm_comm_allreduce IS an esfm_allreduce_fn whose
 * `user` is the esfm_comm*, so a sharded solve is
 *     esfm_ba_problem_solve(p, opt, esfm_comm_allreduce, comm, &summary);
 * with no callback into the host language.  librccl is bound at run time; without it these return ESFM_ERR_COMM. */
#define ESFM_COMM_ID_BYTES 128
typedef struct esfm_comm esfm_comm;
int esfm_comm_get_unique_id(void *id_out /*ESFM_COMM_ID_BYTES*/);
int esfm_comm_create(esfm_ctx *ctx, const void *id /*ESFM_COMM_ID_BYTES*/, int rank, int world, esfm_comm **out);
int esfm_comm_destroy(esfm_comm *comm);
int esfm_comm_rank(const esfm_comm *comm);
int esfm_comm_world(const esfm_comm *comm);
/* ranks of the communicator as RCCL reports them (ncclCommCount; -1 on failure): "did RCCL see N ranks" for logs and bench lines */
int esfm_comm_rccl_ranks(const esfm_comm *comm);
int esfm_comm_allreduce(void *comm, double *buf_dev, int64_t count, int op, void *hip_stream);

/*
 * The replacement for setBAProblem's parameter packing + solveBA's
 * ceres::Solve (cpp_code/src/ba.cpp:58-114, :132-212) with calibration fixed
 * (ReprojectErrorTerm_fixcalib, cpp_code/include/ba.h:108-164):
 *
 *   residual_k = uv_k - project(K[cam_k], AngleAxis(cams[cam_k][0..2]) * pts[pt_k] + cams[cam_k][3..5])
 *   cost = 1/2 * sum_k rho(|residual_k|^2),  rho = Cauchy(a)  (ba.cpp:150)
 *
 * minimised by Levenberg-Marquardt with point-block Schur elimination
 * (DENSE_SCHUR, ba.cpp:201).  cams (6 doubles per camera: angle-axis, then
 * translation) and pts (3 doubles per point) are updated in place; parameter
 * blocks with no observation are left bit-identical.  K4_per_cam holds
 * fx, cx, fy, cy per camera (the four entries ba.h:142-143 reads).
 * Observations may come in any order.  Host pointers.
 *
 * Multi-GPU: each rank passes its own shard of the observations (all cameras,
 * all points, n_obs = local count) and the same allreduce callback; every rank
 * ends with identical cams and the full pts.
 */
int esfm_ba_solve(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs,
                  const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv /*2*n_obs*/,
                  const float *K4_per_cam /*4*n_cam*/,
                  double *cams /*6*n_cam*/, double *pts /*3*n_pt*/,
                  const esfm_ba_options *options /*NULL = defaults*/,
                  esfm_allreduce_fn allreduce /*or NULL*/, void *allreduce_user,
                  esfm_ba_summary *summary /*or NULL*/);

/* Resident form: upload once, iterate many times (what bench.py times). */
typedef struct esfm_ba_problem esfm_ba_problem;

int esfm_ba_problem_create(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs,
                           const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                           const float *K4_per_cam, const double *cams, const double *pts,
                           esfm_ba_problem **out);
/* Re-upload the parameter vector (restart from a new initial guess). */
int esfm_ba_problem_set_params(esfm_ba_problem *p, const double *cams, const double *pts);
int esfm_ba_problem_solve(esfm_ba_problem *p, const esfm_ba_options *options,
                          esfm_allreduce_fn allreduce, void *allreduce_user, esfm_ba_summary *summary);
int esfm_ba_problem_get_params(esfm_ba_problem *p, double *cams, double *pts);
int esfm_ba_problem_destroy(esfm_ba_problem *p);

/* One evaluation of the robustified cost 1/2 sum rho(|r|^2) at the problem's
 * current parameters (tests, and the candidate-cost kernel in isolation). */
int esfm_ba_problem_cost(esfm_ba_problem *p, double cauchy_a, double *cost);

/* ---- free shared intrinsics and box bounds ----------------------------------------------------
 * BundleAdjustment::solveBA(fix_calib_tolerance_BA != 0) (ba.cpp:167-196): the cost functor becomes
 * ReprojectErrorTerm_updatecalib (ba.h:170-222) over ONE shared parameter block fx, cx, fy, cy
 * (ba.cpp:107-113 takes it from calibs_[0]; ba.h:199-202 fixes the order), bounded to its initial value
 * +- tolerance (ba.cpp:190-194).  Independently, the reference frame's six pose parameters are bounded to
 * [-1e-10, +1e-10] (ba.cpp:134, :155-162 / :181-188), i.e. held at the origin.  With any bound Ceres runs its
 * constrained trust-region loop (projection onto the box, projected gradient norm, Armijo line search along
 * the LM step); esfm_ba_iteration.line_search_steps reports the contractions per iteration.
 *
 * esfm_ba_problem_create_free_calib: as esfm_ba_problem_create, without per-camera K4 and with calib4[4] =
 * fx, cx, fy, cy (doubles, as parameters_ holds them) and calib_tolerance > 0 (Ceres rejects an empty box).
 * The multi-GPU rules are unchanged: every rank passes the same calib4; the intrinsics ride in the all-reduced
 * reduced system like one more camera. */
int esfm_ba_problem_create_free_calib(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs,
                                      const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                                      const double *calib4, double calib_tolerance,
                                      const double *cams, const double *pts, esfm_ba_problem **out);
/* new start value and box centre of the intrinsics (problems created with free intrinsics only) */
int esfm_ba_problem_set_calib(esfm_ba_problem *p, const double *calib4, double calib_tolerance);
int esfm_ba_problem_get_calib(esfm_ba_problem *p, double *calib4);
/* Bound all six parameters of camera `cam` to [-threshold, +threshold] for the following solves
 * (ba.cpp:155-162 with threshold = 1e-10); cam < 0 removes the bound. */
int esfm_ba_problem_fix_camera(esfm_ba_problem *p, int cam, double threshold);

/* One-shot solveBA with the optional pieces: calib4 NULL = fixed intrinsics K4_per_cam, else free shared
 * intrinsics (in/out, K4_per_cam ignored); ref_cam < 0 = no reference camera. */
int esfm_ba_solve_ex(esfm_ctx *ctx, int n_cam, int n_pt, int n_obs,
                     const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                     const float *K4_per_cam, double *cams, double *pts,
                     double *calib4, double calib_tolerance, int ref_cam, double ref_threshold,
                     const esfm_ba_options *options, esfm_allreduce_fn allreduce, void *allreduce_user,
                     esfm_ba_summary *summary);

/* The step-length rule of that line search alone (host arithmetic, no GPU): next trial step after the trial
 * (x_cur, f_cur, g_cur) failed the sufficient-decrease test, given the start point (0, f0, g0) and optionally
 * the trial before; *_valid = 0 marks a sample whose evaluation failed.  [upstream line_search.cc
 * InterpolatingPolynomialMinimizingStepSize, CUBIC] */
double esfm_ba_line_search_next_step(double f0, double g0, double x_prev, double f_prev, double g_prev, int prev_valid,
                                     double x_cur, double f_cur, double g_cur, int cur_valid);

/* Host-only helper (no GPU needed): assigns each point to one of `world`
 * shards so that observation counts balance (greedy over points in index
 * order), writing shard_of_point[n_pt].  Observations follow their point. */
int esfm_ba_shard_points(int n_pt, int n_obs, const int32_t *pt_idx, int world, int32_t *shard_of_point);

/* Host-only (no GPU): the structure-aware plan of the reduced camera system for an observation list -- what esfm_ba_problem_solve
 * builds for itself when the camera count takes the tiled solve.  Block (a, b) of the reduced system is structurally non-zero only
 * if cameras a and b observe a common point (the reference adds one residual block per observation, cpp_code/src/ba.cpp:140-151,
 * and lets DENSE_SCHUR, :201, ignore that).  The cameras are ordered by nested dissection of that co-visibility graph, every
 * supernode padded to whole 64-column tiles; the factorisation visits only the tiles of the symbolic fill.
 *   col_src[k]   original unknown 6 cam + a of permuted column k, -1 for identity padding (nb * 64 entries)
 *   tiles[2 t]   block row / column of tile t of the factor (lower triangle, fill included; block row nb = the right-hand side)
 *   info[0..9]   nb, tiles, longest dependency chain in tile columns, tile columns of the dense path, 1 if the solve would use
 *                the plan (at most half the dense path's tiles or half its dependency chain), 64^3 products, supernodes, workgroups,
 *                co-visible camera blocks (a, b <= a) = what several ranks exchange per LM iteration (36 doubles each), 0
 * col_src / tiles may be NULL (with capacity 0) to size the arrays from info first.  leaf_max <= 0: the library's default. */
int esfm_ba_reduced_plan(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, int leaf_max,
                         int32_t *col_src, int col_cap, int32_t *tiles, int tile_cap, int32_t *info /*10*/);

/* ---- sparse-cloud statistical outlier removal ------------------------
 * CProceesing::SORFilter (cpp_code/include/cloudprocessing.hpp:24-36, called on the final cloud at
 * cpp_code/test/sfm.cpp:333) = pcl::StatisticalOutlierRemoval with MeanK (50) and StddevMulThresh (2.0):
 * per point the mean distance to its mean_k nearest neighbours (exact search among the finite points, float
 * squared distances, double sum of float square roots in ascending order); then mean and standard deviation of
 * those N numbers; a point is removed iff its mean distance > mean + std_mul * stddev.
 *
 * points: n rows of stride_floats floats with x, y, z first (3 = packed xyz, 8 = pcl::PointXYZRGB as
 * rgb_pointcloud->points stores it).  keep[n]: 1 = the point survives (the order of survivors is the input
 * order, like pcl::Filter::filter); mean_dist[n] (or NULL) receives the per-point mean distances; *threshold the
 * cut.  mean_k in [1, 63].  Host pointers.  Fewer than mean_k + 1 finite points is undefined in PCL; here the
 * neighbours that exist are summed and the sum is still divided by mean_k. */
int esfm_sor_filter(esfm_ctx *ctx, const float *points, int n, int stride_floats, int mean_k, double std_mul,
                    float *mean_dist /*n or NULL*/, uint8_t *keep /*n*/, int32_t *n_keep, double *threshold /*or NULL*/);
/* The k-NN pass alone on device-resident points (what bench.py times); asynchronous on the context's stream. */
int esfm_sor_mean_distances_dev(esfm_ctx *ctx, const float *points_dev, int n, int stride_floats, int mean_k,
                                float *mean_dist_dev /*n*/);

/* ---- two-view triangulation (triangulation part) -----------------------
 * cv::triangulatePoints as MotionEstimator::getDepthFast (cpp_code/src/estimate_motion.cpp:263, once per image pair inside
 * the matching loop, test/sfm.cpp:166) and doTriangulation (:333) call it: proj1 / proj2 are the 3 x 4 CV_32F projection
 * matrices [R | t] (row-major, 12 floats), pts1 / pts2 the correspondences as normalised image points
 * ((u - cx) / fx, (v - cy) / fy: pixel2cam, include/estimate_motion.h:41-46; 2 floats per point), and the result is the
 * 4 x N CV_32F matrix of homogeneous points, stored here point-major: points4d[4 i + 0..3] = X, Y, Z, W.  Each point is
 * the right singular vector of the smallest singular value of the 4 x 4 DLT system (rows x P[2] - P[0], y P[2] - P[1]),
 * evaluated in double like cvTriangulatePoints; it is defined up to sign, which the callers' division by W removes
 * (estimate_motion.cpp:271, :341).  Host pointers. */
int esfm_triangulate_points(esfm_ctx *ctx, const float *proj1, const float *proj2, const float *pts1, const float *pts2, int n,
                            float *points4d /*4*n*/);
/* Batched form for the pair loop: pair p uses proj1[12 p..], proj2[12 p..] and the points
 * [point_offset[p], point_offset[p+1]) of pts1 / pts2 / points4d; one launch for all pairs. */
int esfm_triangulate_pairs(esfm_ctx *ctx, int n_pairs, const float *proj1, const float *proj2, const int32_t *point_offset /*n_pairs+1*/,
                           const float *pts1, const float *pts2, float *points4d);

/* ---- essential-matrix RANSAC and pose recovery --------------------------
 * MotionEstimator::estimate2D2D_E5P_RANSAC (cpp_code/src/estimate_motion.cpp:27-97, once per matched pair at
 * cpp_code/test/sfm.cpp:165) = cv::findEssentialMat(pts1, pts2, K, CV_RANSAC, prob, threshold, mask) (:49) followed by
 * cv::recoverPose(E, pts1, pts2, K, R, t, mask) (:67).
 *
 * esfm_find_essential_mat: pts1 / pts2 are the n matched pixel positions (2 floats each, cv::Point2f), K4 = fx, cx, fy,
 * cy of the float camera matrix (the reference passes frame 1's K for both images, :43-44).  Points are normalised in
 * double, threshold is divided by (fx + fy) / 2, and OpenCV's RANSAC runs with 5 model points, confidence `prob` and at
 * most 1000 iterations on the sample stream of cv::RNG((uint64)-1): each sample goes through the 5-point kernel (up to 10
 * models), each model is scored by the Sampson distance (float error <= (float)threshold^2), a model replaces the best
 * iff it has more inliers (and more than 4), and the iteration count adapts (RANSACUpdateNumIters).  E[9] row-major with
 * unit Frobenius norm, mask[n] = inliers of the winning model, *iterations = iterations the sequential loop runs.
 * Returns ESFM_ERR_NUMERIC when no model is found (n < 5, or no model with at least 5 inliers): OpenCV returns an empty
 * matrix there.  Every model gets a canonical sign (largest-magnitude entry positive) and the models of one sample are
 * tried in ascending order of E[0][0] (OpenCV: cv::solvePoly's root order and its SVD basis' sign, both artefacts); this
 * only decides ties between models of the same sample.
 *
 * The `_pairs` forms batch many image pairs (pair p owns points [point_offset[p], point_offset[p+1]) and K4_per_pair[4 p..]):
 * all hypotheses of a chunk of iterations of all pairs are solved and scored in one launch; status[p] = 1 iff pair p has
 * a model.  Results are those of the per-pair calls. */
int esfm_find_essential_mat(esfm_ctx *ctx, const float *pts1, const float *pts2, int n, const float *K4, double prob,
                            double threshold, double *E /*9*/, uint8_t *mask /*n*/, int32_t *iterations /*or NULL*/);
int esfm_find_essential_pairs(esfm_ctx *ctx, int n_pairs, const int32_t *point_offset /*n_pairs+1*/, const float *pts1,
                              const float *pts2, const float *K4_per_pair, double prob, double threshold,
                              double *E /*9 per pair*/, uint8_t *mask /*per point*/, int32_t *status /*per pair*/,
                              int32_t *iterations /*per pair or NULL*/);
/* cv::recoverPose with distanceThresh = 50: the four poses of decomposeEssentialMat (SVD, W = [0 1 0; -1 0 0; 0 0 1]) in
 * the order (R1, t), (R2, t), (R1, -t), (R2, -t); for each, every point is triangulated in double against [I | 0] and
 * kept iff Z W > 0 and Z / W < 50 in the first camera and 0 < Z < 50 in the second; with `mask` (in/out, or NULL) the
 * tests are AND-ed with it; the pose with the most points wins (first in order on ties).  R[9] row-major, t[3] unit
 * length, *good = its point count. */
int esfm_recover_pose(esfm_ctx *ctx, const double *E, const float *pts1, const float *pts2, int n, const float *K4,
                      double *R, double *t, uint8_t *mask /*n, in/out, or NULL*/, int32_t *good /*or NULL*/);
int esfm_recover_pose_pairs(esfm_ctx *ctx, int n_pairs, const int32_t *point_offset, const float *pts1, const float *pts2,
                            const float *K4_per_pair, const double *E /*9 per pair*/, uint8_t *mask /*in/out or NULL*/,
                            double *R /*9 per pair*/, double *t /*3 per pair*/, int32_t *good /*per pair or NULL*/);
/* cv::solvePnPRansac(pts3d, pts2d, K, dist = 0, rvec, tvec, false, iterationsCount, reprojectionError, confidence, inliers,
 * cv::SOLVEPNP_EPNP) as MotionEstimator::estimate2D3D_P3P_RANSAC calls it (cpp_code/src/estimate_motion.cpp:161-162, once per
 * newly registered frame, cpp_code/test/sfm.cpp:288).  pts3d: n x 3 floats (cv::Point3f), pts2d: n x 2 float pixels, K4 = fx,
 * cx, fy, cy.  RANSAC with 5 model points on the cv::RNG((uint64)-1) sample stream: each sample is solved by EPnP (control
 * points, M'M null space, three beta approximations + 5 Gauss-Newton steps each, absolute orientation, smallest mean
 * reprojection error), scored by the squared pixel distance of the float projection (<= (float)reprojectionError^2), and the
 * iteration count adapts from iterationsCount; then EPnP is run once more on all inliers of the best model.  rvec =
 * cv::Rodrigues(R), tvec; R (or NULL) receives the rotation matrix, inlier_mask[n] (or NULL) the inliers of the best RANSAC
 * model (OpenCV returns their indices), *n_inliers their number.  Fewer than 5 points is ESFM_ERR_UNSUPPORTED (OpenCV switches
 * to P3P at 4); no model with at least 5 inliers is ESFM_ERR_NUMERIC (OpenCV returns false). */
int esfm_solve_pnp_ransac(esfm_ctx *ctx, const float *pts3d, const float *pts2d, int n, const float *K4, int iterations_count,
                          double reprojection_error, double confidence, double *rvec /*3*/, double *tvec /*3*/, double *R /*9 or NULL*/,
                          uint8_t *inlier_mask /*n or NULL*/, int32_t *n_inliers /*or NULL*/, int32_t *iterations /*or NULL*/);
/* The 5-point kernel alone (EMEstimatorCallback::runKernel [upstream five-point.cpp], what cv::findEssentialMat at reference
 * cpp_code/src/estimate_motion.cpp:49-51 runs on every RANSAC sample): n_samples samples of five correspondences in NORMALISED
 * coordinates, q1, q2 [n_samples][5][2] doubles with x2' E x1 = 0.  E_out [n_samples][10][9]: a sample's models (row-major, unit
 * Frobenius norm, largest-magnitude entry positive, ascending E[0][0]); n_models[n_samples] their number (0..10).  stages (or
 * NULL) [n_samples][117]: the intermediate values for stage-by-stage tests -- null-space basis N[4][9], det[11] (lowest degree
 * first), P[3][4], Qp[3][4], R[3][5] of B(z), the monic coefficients cc[10], the root estimates re[10], im[10], the sweeps of
 * the root iteration (-1: no polynomial).  esfm_five_point_models runs essential_setup_kernel's and essential_roots_kernel's
 * code on the GPU; esfm_five_point_models_host runs the host build of the same routines (easysfm_amd/csrc/five_point_core.hpp),
 * no GPU: the two and the CPU restatement agree to the bit (tests/test_five_point_stages.py). */
int esfm_five_point_models(esfm_ctx *ctx, const double *q1, const double *q2, int n_samples, double *E_out, int32_t *n_models,
                           double *stages /*or NULL*/);
int esfm_five_point_models_host(const double *q1, const double *q2, int n_samples, double *E_out, int32_t *n_models,
                                double *stages /*or NULL*/);
/* Host-only (no GPU): the first n_samples 5-index samples RANSAC draws for `count` points (cv::RNG replay). */
int esfm_ransac_sample_stream(int count, int n_samples, int32_t *idx /*5 per sample*/);

/* ---- SURF detection + description (SURF half) ----------------------------
 * FeatureMatching::detectFeaturesSURF (cpp_code/src/feature_matching.cpp:43-58): cv::xfeatures2d::SURF::create(minHessian)
 * ->detect(image, keypoints) then SURF::create()->compute(image, keypoints, descriptors), OpenCV defaults (4 octaves, 3 layers
 * per octave, 64-float descriptors, rotation-invariant).  image: rows x cols x channels uint8, row-major; channels = 3 is BGR
 * as cv::imread delivers it (converted with cvtColor's fixed-point weights), channels = 1 is already gray.  Outputs, strongest
 * first (OpenCV's KeypointGreater order): keypoints[7 k + 0..6] = pt.x, pt.y, size, angle (degrees), response, octave, class_id
 * (sign of the Laplacian); descriptors[64 k ..] unit-length.  At most max_keypoints are returned (OpenCV returns all; pass
 * rows * cols / 4 to be sure).  The box-filter Hessian pyramid, the 3 x 3 x 3 maxima with quadratic refinement, the dominant
 * orientation (cv::fastAtan2 polynomial, 60-degree window in 5-degree steps) and the descriptor (rotated bilinear window,
 * area shrink to 21 x 21, Gaussian-weighted 2 x 2 gradients, 4 x 4 cells) follow OpenCV's surf.cpp operation by operation. */
int esfm_surf_detect_and_compute(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, double hessian_threshold,
                                 int max_keypoints, float *keypoints /*7 per*/, float *descriptors /*64 per*/, int32_t *n_keypoints);

/*
 * FeatureMatching::detectFeaturesORB (cpp_code/src/feature_matching.cpp:14-41; feature type 'O', sfm.cpp:116):
 * cv::ORB::create(nfeatures)->detect + ->compute with OpenCV's defaults (scale factor 1.2, 8 levels, edge threshold 31,
 * HARRIS_SCORE, patch 31, FAST threshold 20).  image: rows x cols x channels (1 = gray, 3 = BGR) bytes, host.
 * keypoints: 7 floats each -- x, y (level-0 pixels), size, angle (degrees), response (Harris), octave, class_id (-1);
 * descriptors: 32 bytes each (cv::Mat CV_8U, what esfm_match_hamming takes).  Level by level, ordered inside a level by
 * (response descending, y, x); at most max_keypoints are written.  Documented deviation: the 256 intensity tests use this
 * library's own seeded point pairs, not OpenCV's learned rBRIEF table (which ships only inside OpenCV).
 */
int esfm_orb_detect_and_compute(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, int nfeatures,
                                int max_keypoints, float *keypoints /*7 each*/, uint8_t *descriptors /*32 each*/,
                                int32_t *n_keypoints);

/* ---- Image undistortion (undistort part) -------------------------------
 * MotionEstimator::doUnDistort (cpp_code/src/estimate_motion.cpp:431-441): cv::undistort(rgb_image, out, K, distort_coeff), run
 * once per imported frame before feature detection (cpp_code/test/sfm.cpp:97-98).  image / out: rows x cols x channels uint8
 * (1 or 3 interleaved channels), distinct buffers; K4 = fx, cx, fy, cy (no skew; it is also the new camera matrix); dist4 =
 * k1, k2, p1, p2 as DOUBLES -- the values cv::undistort actually sees.  (The reference fills its CV_64F coefficient matrix
 * through at<float>, cpp_code/src/data_io.cpp:118-121, so a distortion file reaches OpenCV as two doubles whose bit patterns
 * are the float pairs (k1, k2) and (p1, p2); the host mirrors reproduce that, this entry point takes whatever doubles result.)
 * Follows OpenCV's algorithm operation by operation: stripes of min(max(1, 4096 / cols), rows) rows with their own new camera
 * matrix (cy - y0) and closed-form 3 x 3 inverse, the normalised x accumulated along the row, the distortion model in double,
 * source coordinates rounded to 1/32 pixel (CV_16SC2 maps), 8-bit bilinear taps with 15-bit fixed-point weights, zero outside
 * the image (INTER_LINEAR, BORDER_CONSTANT). */
int esfm_undistort(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, const double *K4, const double *dist4,
                   uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* ESFM_H_ */
