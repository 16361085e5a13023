// Context / error plumbing of libesfm_hip.so (include/esfm.h, "library / context").
#include <cstdlib>
#include "common.hpp"

namespace esfm {

static thread_local char g_err[1024] = {0};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *get_error() { return g_err; }

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return ESFM_OK;
    // grow geometrically so alternating sizes do not thrash hipMalloc
    size_t want = bytes + bytes / 4 + 256;
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        ptr = nullptr;
        return e == hipErrorOutOfMemory ? ESFM_ERR_OOM : ESFM_ERR_HIP;
    }
    cap = want;
    return ESFM_OK;
}

void DevBuf::release()
{
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

}  // namespace esfm

int esfm_ctx::pin(size_t bytes)
{
    if (bytes <= pinned_cap) return ESFM_OK;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr; pinned_cap = 0;
    size_t want = bytes * 2 + 4096;
    hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) { esfm::set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); return ESFM_ERR_OOM; }
    pinned_cap = want;
    return ESFM_OK;
}

int esfm_ctx::pin_rounds(size_t bytes)
{
    if (bytes <= pinned_rounds_cap) return ESFM_OK;
    if (pinned_rounds) (void)hipHostFree(pinned_rounds);
    pinned_rounds = nullptr; pinned_rounds_cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipHostMalloc(&pinned_rounds, want, hipHostMallocDefault);
    if (e != hipSuccess) { esfm::set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); return ESFM_ERR_OOM; }
    pinned_rounds_cap = want;
    return ESFM_OK;
}

hipEvent_t esfm_ctx::take_event()
{
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void esfm_ctx::time_begin(int id)
{
    TimedLaunch t{id, take_event(), take_event()};
    (void)hipEventRecord(t.a, stream);
    timed.push_back(t);
}

void esfm_ctx::time_end()
{
    if (!timed.empty()) (void)hipEventRecord(timed.back().b, stream);
}

extern "C" {

int esfm_ctx_set_kernel_timing(esfm_ctx *ctx, int enable)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ctx->timing = enable != 0;
    return ESFM_OK;
}

int esfm_ctx_kernel_time(esfm_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches)
{
    if (!ctx || !total_ms || !launches || kernel_id < 0 || kernel_id >= ESFM_K_COUNT) { esfm::set_error("bad argument"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<esfm_ctx::TimedLaunch> keep;
    for (auto &t : ctx->timed) {
        if (t.id != kernel_id) { keep.push_back(t); continue; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { *total_ms += (double)ms; *launches += 1; }
        ctx->event_pool.push_back(t.a); ctx->event_pool.push_back(t.b);
    }
    ctx->timed.swap(keep);
    return ESFM_OK;
}

const char *esfm_version(void)
{
    static char v[32];
    snprintf(v, sizeof(v), "%d.%d", ESFM_VERSION_MAJOR, ESFM_VERSION_MINOR);
    return v;
}

const char *esfm_last_error(void) { return esfm::get_error(); }

int esfm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int esfm_ctx_create(int device, void *hip_stream, esfm_ctx **out)
{
    if (!out) { esfm::set_error("esfm_ctx_create: out is NULL"); return ESFM_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        esfm::set_error("no HIP device available (%s); libesfm_hip has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return ESFM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { esfm::set_error("device %d out of range [0,%d)", device, n); return ESFM_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { esfm::set_error("hipGetDeviceProperties failed"); return ESFM_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        esfm::set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return ESFM_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { esfm::set_error("hipSetDevice(%d) failed", device); return ESFM_ERR_NO_DEVICE; }
    esfm_ctx *c = new esfm_ctx();
    if (const char *e = getenv("ESFM_CHECK_PREPARED")) c->prep_check = (e[0] != '\0' && e[0] != '0') ? 1 : 0;
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
        c->owns_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            esfm::set_error("hipStreamCreate failed");
            delete c;
            return ESFM_ERR_HIP;
        }
        c->owns_stream = true;
    }
    *out = c;
    return ESFM_OK;
}

int esfm_ctx_destroy(esfm_ctx *ctx)
{
    if (!ctx) return ESFM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    esfm::DevBuf *bufs[] = {&ctx->norms, &ctx->pair_tab, &ctx->knn_idx, &ctx->knn_dist, &ctx->flagged, &ctx->counters,
                            &ctx->stage_a, &ctx->stage_b, &ctx->stage_c, &ctx->stage_d, &ctx->stage_e, &ctx->hm_exp, &ctx->pair_cnt, &ctx->pair_list,
                            &ctx->pair_cnt2, &ctx->pair_list2, &ctx->l2_hi, &ctx->knn_d2, &ctx->pair_cnt2b, &ctx->fin_done, &ctx->bank, &ctx->surv_cnt, &ctx->surv_cntb, &ctx->surv_list, &ctx->prep_sum};
    for (auto *b : bufs) b->release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_rounds) (void)hipHostFree(ctx->pinned_rounds);
    for (auto &c : ctx->ba_chunks) (void)hipFree(c.ptr);
    for (void *m : ctx->ba_mailboxes) (void)hipHostFree(m);
    for (auto &t : ctx->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ESFM_OK;
}

int esfm_ctx_synchronize(esfm_ctx *ctx)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ESFM_OK;
}

void *esfm_ctx_stream(esfm_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

}  // extern "C"
