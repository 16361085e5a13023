// Shared host-side plumbing of libesfm_hip.so: context, error reporting, device scratch.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/esfm.h"

namespace esfm {

// Thread-local last-error text behind esfm_last_error().
void set_error(const char *fmt, ...);
const char *get_error();

#define ESFM_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            ::esfm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                              __LINE__);                                                        \
            return e__ == hipErrorOutOfMemory ? ESFM_ERR_OOM : ESFM_ERR_HIP;                    \
        }                                                                                       \
    } while (0)

#define ESFM_REQUIRE(cond, msg)                     \
    do {                                            \
        if (!(cond)) {                              \
            ::esfm::set_error("%s: %s", __func__, msg); \
            return ESFM_ERR_INVALID_ARG;            \
        }                                           \
    } while (0)

// Host <-> device copies of caller-owned memory (ordered on `st` like hipMemcpyAsync; pageable memory makes them synchronous).
// The HIP runtime PINS a pageable buffer of 1 MiB or more for the transfer -- a userptr registration with the kernel driver -- and
// the driver stops every queue of the process while it re-validates such a registration after ordinary heap activity in the same
// mapping (malloc / free trimming the heap): bin/sfm_native's 768 x 512 x 3 images (1.18 MB) stalled a third of the frames for
// 20 - 38 ms each, ten times what their detection takes (scratch/e2e_hiplog.sh: the time sits inside hipMemcpyAsync; with
// GPU_PINNED_MIN_XFER_SIZE raised the stalls are gone).  Pieces below that size go through the runtime's own staging buffer
// instead, whatever the process' environment says.
// A buffer the caller has registered or allocated through HIP (hipHostRegister / hipHostMalloc) is pinned already and goes in one
// asynchronous transfer.  Price of the pieces: a pageable transfer runs at the speed of the host's copy into the staging buffer,
// 16 GB/s, instead of the DMA engines' 49 GB/s (esfm_undistort of a 3072 x 2048 x 3 image, both directions: 0.8 -> 2.3 ms;
// from registered memory 0.77 ms; scratch/copy_paths.py) -- bounded and proportional, unlike the stalls.
constexpr size_t kCopyPiece = 512u << 10;
inline bool host_memory_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }      // (an ordinary malloc'ed pointer: "invalid value"; the sticky error is cleared)
    return a.type == hipMemoryTypeHost;
}
inline hipError_t copy_pieces(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const void *host_side, hipStream_t st)
{
    if (bytes <= kCopyPiece || host_memory_is_pinned(host_side)) return bytes ? hipMemcpyAsync(dst, src, bytes, kind, st) : hipSuccess;
    for (size_t off = 0; off < bytes; off += kCopyPiece) {
        const size_t n = bytes - off < kCopyPiece ? bytes - off : kCopyPiece;
        const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, n, kind, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
inline hipError_t copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st) { return copy_pieces(dst_dev, src_host, bytes, hipMemcpyHostToDevice, src_host, st); }
inline hipError_t copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st) { return copy_pieces(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, dst_host, st); }

// A grow-only device buffer (scratch reused across calls; 288 GB of HBM makes
// holding on to the high-water mark the right trade).
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

}  // namespace esfm

struct esfm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cu = 256;
    // matching scratch
    esfm::DevBuf bank;   // esfm_match_pairs (host-pointer batched form): the uploaded descriptor rows of all sets
    esfm::DevBuf norms, pair_tab, knn_idx, knn_dist, flagged, counters, stage_a, stage_b, stage_c, stage_d, stage_e;
    esfm::DevBuf pair_cnt, pair_list;   // uncertified queries of the L2 pass binned per pair (one counter per pair; the pair's slice of the query numbering)
    esfm::DevBuf pair_cnt2, pair_list2;   // the same for the second (three-product) pass over what the one-product pass left uncertified
    esfm::DevBuf pair_cnt2b;           // second phase of pair_cnt2: a call fills one phase, its one-product kernel zeroes the other for the next call
    int l2_phase = 0;                  // phase the NEXT one-product call fills
    int l2_phase_pairs[2] = {0, 0};    // entries of each phase that may be non-zero
    int32_t *counters_cur = nullptr;   // the 16 counters of the last L2 call (esfm_match_last_stats / _second_pass / _flagged)
    esfm::DevBuf fin_done;             // l2_finish_kernel: per-pair arrival counters
    esfm::DevBuf surv_cnt, surv_cntb, surv_list;   // the ratio screen's survivors: per-pair counts (two phases), 48-byte entries in the pair's slice of the query numbering
    // esfm_match_prepare_dev: the derived per-row operands (bf16 images, norms, residual norms; 0/1 byte image for Hamming) in l2_hi /
    // norms / hm_exp belong to this descriptor buffer and are not recomputed by the match calls
    const void *prep_desc = nullptr;
    int prep_metric = 0, prep_width = 0;
    bool prep_hm_fp4 = false;          // Hamming: hm_exp holds the FP4 form's nibble images (hamming_fp4_kernel), not the byte image of the i8 form
    int64_t prep_rows = 0;
    // esfm_ctx_set_prepared_check: the prepared buffer's fingerprint ([0] at prepare, [1] re-derived by every match call that relies on it)
    int prep_check = 0;
    bool prep_has_sum = false;
    esfm::DevBuf prep_sum;
    esfm::DevBuf knn_d2;   // exact second-best d^2 of the queries the one-product pass left uncertified (the refine pass's thresholds)
    esfm::DevBuf l2_hi;    // one-product pass: bf16(t) and bf16(-2 q) images (128 B per row each) and the two residual norms per row
    esfm::DevBuf hm_exp;   // expanded descriptor image: 0/1 bytes + start values (Hamming MFMA) or bf16 hi/lo halves (L2), 256 B per row
    // pinned host staging for small tables / counters
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    int64_t last_n_queries = 0;
    size_t last_pair_bytes = 0;
    int l2_audit = 0;   // esfm_ctx_set_l2_audit: 0 product path, 1 no re-scan, 2 exact scan of every query, 3 one-product pass alone, 4 one-product pass alone with the ratio screen's rejections listed
    int pin(size_t bytes);
    // a second pinned area for the per-round tables of the RANSAC loops (samples up, model counts and inlier counts down)
    void *pinned_rounds = nullptr;
    size_t pinned_rounds_cap = 0;
    int pin_rounds(size_t bytes);
    // device chunks and pinned scalar mailboxes of destroyed BA problems, kept for the next problem of this context (an incremental
    // reconstruction sets one up and tears it down every ba_frequency frames: ba_api.cpp dev_alloc / esfm_ba_problem_destroy)
    struct BaChunk { void *ptr; size_t bytes; };
    std::vector<BaChunk> ba_chunks;
    std::vector<void *> ba_mailboxes;
    // optional per-kernel hipEvent timing (esfm_ctx_set_kernel_timing)
    bool timing = false;
    struct TimedLaunch { int id; hipEvent_t a, b; };
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t take_event();
    void time_begin(int id);
    void time_end();
};

namespace esfm {
// RAII bracket: records events around a launch when timing is enabled, otherwise does nothing.
struct KernelTimer {
    esfm_ctx *c;
    KernelTimer(esfm_ctx *ctx, int id) : c(ctx && ctx->timing ? ctx : nullptr) { if (c) c->time_begin(id); }
    ~KernelTimer() { if (c) c->time_end(); }
};
inline int set_device(const esfm_ctx *ctx)
{
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { set_error("hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e)); return ESFM_ERR_HIP; }
    return ESFM_OK;
}
}  // namespace esfm
