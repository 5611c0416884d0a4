// Context, memory and timing entry points of libmi_lte.so (see include/mi_lte.h).
#include <cmath>

#include <chrono>
#include <cstdlib>

#include <algorithm>

#include "ctx.hpp"
#include "build_id.h"

#include <cstring>

#include "lte_tables.h"

extern "C" {

int mi_lte_version(void) { return MI_LTE_VERSION; }
const char *mi_lte_build_id(void) { return MI_LTE_BUILD_ID; }

int mi_lte_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi_lte_ctx_create(int device, mi_lte_ctx **out)
{
    if (!out) return MI_LTE_ERR_INVALID_ARG;
    *out  = nullptr;
    int n = mi_lte_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        fprintf(stderr, "mi_lte: no usable HIP device %d (found %d); this library has no CPU fallback\n", device, n);
        return MI_LTE_ERR_NO_DEVICE;
    }
    mi_lte_ctx *ctx = new mi_lte_ctx();
    ctx->device     = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete ctx;
        return MI_LTE_ERR_HIP;
    }
    ctx->dev_name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "mi_lte: device %d is %s, but the kernels are built for gfx950 only\n", device, prop.gcnArchName);
        delete ctx;
        return MI_LTE_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return MI_LTE_ERR_HIP;
    }
    *out = ctx;
    return MI_LTE_OK;
}

void mi_lte_ctx_destroy(mi_lte_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->host_cache_free) ctx->host_cache_free(ctx);
    for (void *p : ctx->owned) (void)hipFree(p);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    if (ctx->side_scratch) (void)hipFree(ctx->side_scratch);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    for (int k = 0; k < 2; k++)
        if (ctx->ev_bounce[k]) (void)hipEventDestroy(ctx->ev_bounce[k]);
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mi_lte_set_turbo_small_batch(mi_lte_ctx *ctx, uint32_t n_cb_max)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    ctx->siso_small_max = n_cb_max;
    return MI_LTE_OK;
}

int mi_lte_set_turbo_merged(mi_lte_ctx *ctx, uint32_t on)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    ctx->merged_decode = on != 0;
    return MI_LTE_OK;
}

const char *mi_lte_last_error(const mi_lte_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char *mi_lte_device_name(const mi_lte_ctx *ctx) { return ctx ? ctx->dev_name.c_str() : ""; }
void       *mi_lte_stream(const mi_lte_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
const char *mi_lte_last_kernels(const mi_lte_ctx *ctx) { return ctx ? ctx->last_kernels.c_str() : ""; }

int mi_lte_malloc(mi_lte_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MI_HIP_CHECK(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return MI_LTE_OK;
}
int mi_lte_free(mi_lte_ctx *ctx, void *d_ptr)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MI_HIP_CHECK(ctx, hipFree(d_ptr));
    return MI_LTE_OK;
}
int mi_lte_memset(mi_lte_ctx *ctx, void *d_ptr, int value, size_t bytes)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return MI_LTE_OK;
}
} // extern "C"
namespace {
// 16-byte copies, every wavefront access one contiguous KiB.  Three shapes, the best of which is reported (tools/ubench/nt_copy.hip measured
// more: on an MI355X one access per thread streams best, 6.2 TB/s, 6.5 with the non-temporal hint; four loads in flight per thread 5.7 /
// 6.3; a grid-stride loop 4.6-4.9): one 16-byte load and store per thread; the same with the non-temporal hint; a grid-stride loop
typedef uint32_t mi_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ void k_copy16(const mi_u32x4 *__restrict__ src, mi_u32x4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    else dst[i] = src[i];
}
__global__ void k_copy16_loop(const mi_u32x4 *__restrict__ src, mi_u32x4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
} // namespace
extern "C" {
int mi_lte_device_copy_rate(mi_lte_ctx *ctx, size_t bytes, uint32_t reps, double *gb_per_s)
{
    if (!ctx || !gb_per_s || bytes < 16 || (bytes & 15) || reps == 0) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mi_u32x4 *a = nullptr, *b = nullptr;
    auto      guard = on_fail([&] { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(a); (void)hipFree(b); });
    MI_HIP_CHECK(ctx, hipMalloc((void **)&a, bytes));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&b, bytes));
    MI_HIP_CHECK(ctx, hipMemsetAsync(a, 0x5A, bytes, ctx->stream));
    const size_t n = bytes / 16;
    if ((n + 255) / 256 > 0x7FFFFFFFull) return MI_LTE_ERR_INVALID_ARG;
    const dim3 block(256), grid_all((unsigned)((n + 255) / 256)), grid_loop((unsigned)std::min<size_t>((n + 255) / 256, 256 * 32));
    double     best = 0.0;
    for (int shape = 0; shape < 3; shape++) {
        auto launch = [&](const mi_u32x4 *s, mi_u32x4 *d) {
            if (shape == 0) hipLaunchKernelGGL(k_copy16<false>, grid_all, block, 0, ctx->stream, s, d, n);
            else if (shape == 1) hipLaunchKernelGGL(k_copy16<true>, grid_all, block, 0, ctx->stream, s, d, n);
            else hipLaunchKernelGGL(k_copy16_loop, grid_loop, block, 0, ctx->stream, s, d, n);
        };
        launch(a, b); // warm-up (page tables, clocks)
        MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (uint32_t r = 0; r < reps; r++) launch((r & 1) ? b : a, (r & 1) ? a : b);
        MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        MI_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
        MI_HIP_CHECK(ctx, hipGetLastError());
        float ms = 0.f;
        MI_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        const double r = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
        ctx->copy_rates[shape] = r;
        best = std::max(best, r);
    }
    *gb_per_s = best;
    guard.armed = false;
    (void)hipFree(a); (void)hipFree(b);
    return MI_LTE_OK;
}
int mi_lte_device_copy_rates(const mi_lte_ctx *ctx, double *out3)
{
    if (!ctx || !out3) return MI_LTE_ERR_INVALID_ARG;
    for (int i = 0; i < 3; i++) out3[i] = ctx->copy_rates[i];
    return MI_LTE_OK;
}
// Copies of 16 KB .. 64 MB between the device and host memory do not use the runtime's copy engines: the first use of EACH engine in a
// process costs 7-9 ms (its queue is made then), the runtime rotates through them, and a caller with a few dozen subframes in hand meets
// that cost inside its first batches -- 8.2 ms of the 8.8 ms a 20 MHz cell scan spent on its 70-subframe pass went into one 30 KB copy of
// transport blocks (profiles/r05_scan_batch_trace_100rb.txt; HSA_ENABLE_SDMA=0 from outside removes it, and so does this from inside).
// Instead a kernel moves the bytes to / from 4 MiB of pinned host memory mapped into the device, 4 MiB at a time, and the host copies
// between that block and the caller's buffer.  Smaller copies stay on the runtime's staging path (13-30 us), larger ones amortise an
// engine's start-up and run at the link rate.
__global__ __launch_bounds__(256) void k_copy_words(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16, uint8_t *dst_tail, const uint8_t *src_tail, uint32_t n_tail)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
static constexpr size_t BOUNCE_LO = 16u << 10, BOUNCE_BYTES = 4u << 20, BOUNCE_HALF = BOUNCE_BYTES / 2, BOUNCE_HI = 64u << 20;
static const bool runtime_copies = getenv("MI_LTE_RUNTIME_COPIES") != nullptr; // (A/B switch: every copy through hipMemcpyAsync, as before round 5)
static bool bounce_ready(mi_lte_ctx *ctx, const void *d_ptr, size_t bytes)
{
    if (runtime_copies || ctx->bounce_failed || bytes <= BOUNCE_LO || bytes > BOUNCE_HI || ((uintptr_t)d_ptr & 15u)) return false;
    if (!ctx->h_bounce) {
        // the block is mapped for the context's OWN device (a multi-device process: whatever device the calling thread had current otherwise);
        // a failure is remembered -- the runtime path serves from then on, without a retry per copy
        if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&ctx->h_bounce, BOUNCE_BYTES, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&ctx->d_bounce, ctx->h_bounce, 0) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_bounce[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_bounce[1], hipEventDisableTiming) != hipSuccess) {
            if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
            for (int k = 0; k < 2; k++)
                if (ctx->ev_bounce[k]) { (void)hipEventDestroy(ctx->ev_bounce[k]); ctx->ev_bounce[k] = nullptr; }
            ctx->h_bounce = ctx->d_bounce = nullptr;
            ctx->bounce_failed = true;
            (void)hipGetLastError();
        }
    }
    return ctx->h_bounce != nullptr;
}
static void launch_copy(mi_lte_ctx *ctx, void *dst, const void *src, size_t n, hipStream_t stream = nullptr)
{
    const size_t n16 = n / 16;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256 + 1, 2048);
    k_copy_words<<<grid, 256, 0, stream ? stream : ctx->stream>>>((uint4 *)dst, (const uint4 *)src, n16, (uint8_t *)dst + 16 * n16, (const uint8_t *)src + 16 * n16, (uint32_t)(n - 16 * n16));
}
} // extern "C"
// The asynchronous pair for the library's own MAPPED pinned blocks (hostapi.cc's staging buffer): the same kernel on the block's device
// alias, no wait; small copies and anything unaligned go to the runtime.  The caller waits for the stream before it touches the block.
hipError_t mi_pinned_to_device(mi_lte_ctx *ctx, void *d_dst, const void *h_pinned, size_t bytes, hipStream_t stream)
{
    void *alias = nullptr;
    if (!runtime_copies && bytes > BOUNCE_LO && !(((uintptr_t)d_dst | (uintptr_t)h_pinned) & 15u) && hipHostGetDevicePointer(&alias, const_cast<void *>(h_pinned), 0) == hipSuccess) {
        launch_copy(ctx, d_dst, alias, bytes, stream);
        return hipGetLastError();
    }
    (void)hipGetLastError();
    return hipMemcpyAsync(d_dst, h_pinned, bytes, hipMemcpyHostToDevice, stream ? stream : ctx->stream);
}
hipError_t mi_device_to_pinned(mi_lte_ctx *ctx, void *h_pinned, const void *d_src, size_t bytes)
{
    void *alias = nullptr;
    if (!runtime_copies && bytes > BOUNCE_LO && !(((uintptr_t)d_src | (uintptr_t)h_pinned) & 15u) && hipHostGetDevicePointer(&alias, h_pinned, 0) == hipSuccess) {
        launch_copy(ctx, alias, d_src, bytes);
        return hipGetLastError();
    }
    (void)hipGetLastError();
    return hipMemcpyAsync(h_pinned, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
}
// Up to three blocks of pinned host memory to the device in ONE launch on the given stream (the host pipeline's per-chunk arrays: subframe numbers
// and cells; a slice's descriptors, offsets and code-block map).  A copy COMMAND per array costs a copy engine's start-up each and makes the
// runtime rotate engines between the sample copies; a kernel per array costs a dispatch each.  64 workgroups: the arrays are a few MB at most,
// and a wide grid next to the lanes' kernels waits for wave slots it does not need.  Sizes are multiples of 4 bytes; blocks that are not
// 16-byte aligned, or memory the device cannot see, go to the runtime instead.
__global__ __launch_bounds__(256) void k_copy_segments(MiCopySeg a, MiCopySeg b, MiCopySeg c)
{
    const MiCopySeg s = blockIdx.y == 0 ? a : blockIdx.y == 1 ? b : c;
    const size_t    n16 = s.bytes / 16;
    uint4          *d = (uint4 *)s.dst;
    const uint4    *h = (const uint4 *)s.src;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = h[i];
    const uint32_t n4 = (uint32_t)(s.bytes - 16 * n16) / 4;
    if (blockIdx.x == 0 && threadIdx.x < n4) ((uint32_t *)(d + n16))[threadIdx.x] = ((const uint32_t *)(h + n16))[threadIdx.x];
}
hipError_t mi_pinned_segments_to_device(mi_lte_ctx *ctx, const MiCopySeg *segs, uint32_t n_seg, hipStream_t stream)
{
    MiCopySeg k[3] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
    bool      by_kernel = !runtime_copies && n_seg >= 1 && n_seg <= 3;
    for (uint32_t i = 0; by_kernel && i < n_seg; i++) {
        void *alias = nullptr;
        if ((((uintptr_t)segs[i].dst | (uintptr_t)segs[i].src) & 15u) || (segs[i].bytes & 3u) || hipHostGetDevicePointer(&alias, const_cast<void *>(segs[i].src), 0) != hipSuccess) by_kernel = false;
        k[i] = {segs[i].dst, alias, segs[i].bytes};
    }
    (void)hipGetLastError();
    if (!by_kernel) {
        static bool said = false;
        if (!said && getenv("MI_LTE_PIPELINE_TRACE")) { said = true; fprintf(stderr, "mi_lte: pinned segments copied by the runtime (unaligned or not mapped)\n"); }
        for (uint32_t i = 0; i < n_seg; i++) {
            const hipError_t e = hipMemcpyAsync(segs[i].dst, segs[i].src, segs[i].bytes, hipMemcpyHostToDevice, stream ? stream : ctx->stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    static const unsigned seg_grid = getenv("MI_LTE_SIDE_COPY_BLOCKS") ? (unsigned)std::max(1, atoi(getenv("MI_LTE_SIDE_COPY_BLOCKS"))) : 64u; // (tuning aid)
    k_copy_segments<<<dim3(seg_grid, n_seg), 256, 0, stream ? stream : ctx->stream>>>(k[0], k[1], k[2]);
    return hipGetLastError();
}
extern "C" {
int mi_lte_memcpy_h2d(mi_lte_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (bounce_ready(ctx, d_dst, bytes)) {
        MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        // the two halves of the block in turn: the host fills one while the kernel behind the other is still reading it across the link
        // (one 4 MiB buffer, filled, copied and waited for in series, ran a 64 MB call at about half the link rate)
        uint32_t k = 0;
        for (size_t off = 0; off < bytes; off += BOUNCE_HALF, k++) {
            const size_t n = std::min(BOUNCE_HALF, bytes - off);
            uint8_t     *hb = (uint8_t *)ctx->h_bounce + (k & 1) * BOUNCE_HALF;
            if (k >= 2) MI_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_bounce[k & 1])); // the kernel that read this half two chunks ago is done
            memcpy(hb, (const uint8_t *)h_src + off, n);
            launch_copy(ctx, (uint8_t *)d_dst + off, (uint8_t *)ctx->d_bounce + (k & 1) * BOUNCE_HALF, n);
            MI_HIP_CHECK(ctx, hipGetLastError());
            MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev_bounce[k & 1], ctx->stream));
        }
        MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return MI_LTE_OK;
    }
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}
int mi_lte_memcpy_d2h(mi_lte_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (bounce_ready(ctx, d_src, bytes)) {
        MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        // two halves in flight: while the host empties one, the kernel is filling the other
        const size_t n_chunks = (bytes + BOUNCE_HALF - 1) / BOUNCE_HALF;
        for (size_t k = 0; k <= n_chunks; k++) {
            if (k < n_chunks) { // (chunk k goes into the half chunk k - 2 was emptied from, one iteration ago)
                const size_t off = k * BOUNCE_HALF, n = std::min(BOUNCE_HALF, bytes - off);
                launch_copy(ctx, (uint8_t *)ctx->d_bounce + (k & 1) * BOUNCE_HALF, (const uint8_t *)d_src + off, n);
                MI_HIP_CHECK(ctx, hipGetLastError());
                MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev_bounce[k & 1], ctx->stream));
            }
            if (k >= 1) {
                const size_t off = (k - 1) * BOUNCE_HALF, n = std::min(BOUNCE_HALF, bytes - off);
                MI_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_bounce[(k - 1) & 1])); // (also makes the kernel's writes to the host block visible)
                memcpy((uint8_t *)h_dst + off, (const uint8_t *)ctx->h_bounce + ((k - 1) & 1) * BOUNCE_HALF, n);
            }
        }
        return MI_LTE_OK;
    }
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}
int mi_lte_sync(mi_lte_ctx *ctx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}
int mi_lte_timer_start(mi_lte_ctx *ctx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MI_LTE_OK;
}
int mi_lte_timer_stop(mi_lte_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    MI_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    MI_HIP_CHECK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return MI_LTE_OK;
}

int mi_lte_profile_enable(mi_lte_ctx *ctx, int on)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    ctx->prof_on = on != 0;
    return MI_LTE_OK;
}
int mi_lte_profile_reset(mi_lte_ctx *ctx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_used = 0;
    ctx->prof_recs.clear();
    return MI_LTE_OK;
}
// "name:launches:total_ms;..." for every kernel bracketed since the last reset
const char *mi_lte_profile_report(mi_lte_ctx *ctx)
{
    if (!ctx) return "";
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return "";
    std::map<std::string, std::pair<int, double>> agg;
    for (auto &r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_pool[r.second], ctx->prof_pool[r.second + 1]) != hipSuccess) continue;
        auto &a = agg[r.first];
        a.first += 1;
        a.second += ms;
    }
    ctx->prof_report.clear();
    char buf[256];
    for (auto &kv : agg) {
        snprintf(buf, sizeof(buf), "%s:%d:%.6f;", kv.first.c_str(), kv.second.first, kv.second.second);
        ctx->prof_report += buf;
    }
    return ctx->prof_report.c_str();
}

} // extern "C"

void mi_prof_begin(mi_lte_ctx *ctx, const char *name)
{
    ctx->prof_armed = false;
    if (!ctx->prof_on) return;
    while (ctx->prof_pool.size() < ctx->prof_used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return; // this launch goes unbracketed
        ctx->prof_pool.push_back(e);
    }
    ctx->prof_recs.push_back({name, ctx->prof_used});
    (void)hipEventRecord(ctx->prof_pool[ctx->prof_used], ctx->stream);
    ctx->prof_armed = true;
}
void mi_prof_end(mi_lte_ctx *ctx)
{
    if (!ctx->prof_armed) return; // profiling off, or the begin side could not get its pair of events
    ctx->prof_armed = false;
    (void)hipEventRecord(ctx->prof_pool[ctx->prof_used + 1], ctx->stream);
    ctx->prof_used += 2;
}

// The wait of the per-call host forms (hostapi.cc).  Their caller is blocked on a result that is tens of microseconds of GPU work away, three
// times per subframe; hipStreamSynchronize sleeps on the completion interrupt, and the wake-up alone costs ~35 us per wait on the MI355X box
// (the scanner's per-subframe loop: 255 us with it, 134 us without; HSA_ENABLE_INTERRUPT=0 shows the same from outside).  So the host
// spins (mi_stream_wait_polling below), and gives the core back to the blocking wait only when the work turns out to be long (MI_POLL_US: a
// first call that builds tables, a big batch).
// The batch entry points wait with this: a call on a handful of units is a per-call caller's (spin), anything larger sleeps on the
// interrupt (when the spin was on hipStreamQuery, a batch that had been queried finished later: uplink workload 6.9 against 6.15 ms per
// step with as few as 32 queries at the start of the wait -- bisected; a batch has nothing to gain from spinning anyway).
hipError_t mi_stream_wait(mi_lte_ctx *ctx, size_t n_units) { return n_units <= 8 ? mi_stream_wait_polling(ctx) : hipStreamSynchronize(ctx->stream); }

#ifndef MI_POLL_US
#define MI_POLL_US 200 // the work of a per-call form is done within this; a batch call goes to the blocking wait
#endif
// The wait itself: a one-thread kernel behind the call's work stores a sequence number into a word of coherent host memory and the host
// spins on that word -- no runtime call in the loop.  (Spinning on hipStreamQuery instead cost 6-10 % more per call: scan loop 58 -> 51 us
// per subframe at 1.4 MHz, 103 -> 95 at 20 MHz, the uplink subframe 355 -> 325 us; tools/ab/flag_wait.sh.)  Everything a per-call form hands
// back is in coherent host memory written by kernels that precede the flag kernel in the stream, so when the word has changed the results
// are there; anything the runtime itself must know to be complete (frees, re-allocations) still goes through its own synchronisation.
// (Letting the call's last kernel store the word itself -- every workgroup fences its results out to the system and counts itself in, the
// last one stores the number -- was measured too: 48.5 -> 52 us per subframe at 1.4 MHz, 92 -> 103 at 20 MHz.  A system-scope fence in
// each of a hundred workgroups that have just written across the link costs more than one more launch.)
__global__ void k_done_flag(volatile uint32_t *flag, uint32_t seq) { *flag = seq; }

hipError_t mi_stream_wait_polling(mi_lte_ctx *ctx)
{
    static const bool never_poll = getenv("MI_LTE_BLOCKING_WAIT") != nullptr; // a deployment that would rather have the core than the ~35 us per wait
    if (never_poll) return hipStreamSynchronize(ctx->stream);
    if (!ctx->h_flag) {
        if (hipHostMalloc((void **)&ctx->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return hipStreamSynchronize(ctx->stream);
        (void)hipHostGetDevicePointer((void **)&ctx->d_flag, ctx->h_flag, 0);
        *ctx->h_flag = 0;
    }
    const uint32_t want = ++ctx->flag_seq;
    k_done_flag<<<1, 1, 0, ctx->stream>>>(ctx->d_flag, want);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned n = 1;; n++) {
        if (*(volatile uint32_t *)ctx->h_flag == want) return hipSuccess;
        if ((n & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(MI_POLL_US)) return hipStreamSynchronize(ctx->stream);
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

int mi_ctx_reserve_scratch(mi_lte_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) MI_HIP_CHECK(ctx, hipFree(ctx->scratch));
    ctx->scratch       = nullptr;
    ctx->scratch_bytes = 0;
    MI_HIP_CHECK(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return MI_LTE_OK;
}

int mi_ctx_small_results(mi_lte_ctx *ctx, size_t bytes, void **h, void **d)
{
    if (bytes > MI_SMALL_BYTES) return MI_LTE_ERR_INVALID_ARG;
    if (!ctx->h_small) {
        MI_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_small, MI_SMALL_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
        MI_HIP_CHECK(ctx, hipHostGetDevicePointer(&ctx->d_small, ctx->h_small, 0));
    }
    *h = ctx->h_small;
    *d = ctx->d_small;
    return MI_LTE_OK;
}

// QPP interleaver tables.  spec == 0 reproduces the reference's uint32 arithmetic
// (liblte_phy.cc:10954-10958: idx = (f1*i + f2*i*i) % K, which wraps for 20 block sizes and is then
// not a permutation); spec != 0 is the exact 3GPP TS 36.212 5.1.3.2.3 index.
int mi_ctx_turbo_tables(mi_lte_ctx *ctx, uint32_t K, int spec, TurboTables *out)
{
    const uint64_t key = (uint64_t)K | ((uint64_t)(spec ? 1 : 0) << 32);
    auto           it  = ctx->turbo_tables.find(key);
    if (it != ctx->turbo_tables.end()) {
        *out = it->second;
        return MI_LTE_OK;
    }
    uint32_t f1 = 0, f2 = 0;
    bool     found = false;
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (LTE_QPP_ROWS[r].K == K) { f1 = LTE_QPP_ROWS[r].f1; f2 = LTE_QPP_ROWS[r].f2; found = true; }
    if (!found) {
        ctx->err = "K is not an LTE turbo block size";
        return MI_LTE_ERR_INVALID_ARG;
    }
    std::vector<uint16_t> pi(K), inv(K, 0xFFFF);
    for (uint32_t i = 0; i < K; i++) {
        uint32_t idx;
        if (spec) idx = (uint32_t)(((uint64_t)f1 * i + (uint64_t)f2 * i * i) % K);
        else      idx = (f1 * i + f2 * i * i) % K; // wraps in uint32 exactly like the reference
        pi[i]    = (uint16_t)idx;
        inv[idx] = (uint16_t)i; // ascending i: the last (largest) writer wins, as in the reference's loop
    }
    const uint32_t K16 = (K + 15u) & ~15u, zero_slot = 2u * ((K + 63u) & ~63u);
    std::vector<uint16_t> inv2(K16, (uint16_t)zero_slot);
    for (uint32_t j = 0; j < K; j++)
        if (inv[j] != 0xFFFF) inv2[j] = (uint16_t)(2u * inv[j]);
    const uint32_t Kp64 = (K + 63u) & ~63u;
    std::vector<uint32_t> pi_row(Kp64, Kp64), inv_row(Kp64, Kp64);
    for (uint32_t j = 0; j < K; j++) {
        pi_row[j] = pi[j];
        if (inv[j] != 0xFFFF) inv_row[j] = inv[j];
    }
    TurboTables t;
    MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_pi_row, sizeof(uint32_t) * Kp64));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_inv_row, sizeof(uint32_t) * Kp64));
    ctx->owned.push_back(t.d_pi_row);
    ctx->owned.push_back(t.d_inv_row);
    MI_H2D(ctx, t.d_pi_row, pi_row.data(), sizeof(uint32_t) * Kp64);
    MI_H2D(ctx, t.d_inv_row, inv_row.data(), sizeof(uint32_t) * Kp64);
    MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_pi, sizeof(uint16_t) * K));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_inv, sizeof(uint16_t) * K));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_inv2, sizeof(uint16_t) * K16));
    ctx->owned.push_back(t.d_pi);
    ctx->owned.push_back(t.d_inv);
    ctx->owned.push_back(t.d_inv2);
    MI_H2D(ctx, t.d_inv2, inv2.data(), sizeof(uint16_t) * K16);
    MI_H2D(ctx, t.d_pi, pi.data(), sizeof(uint16_t) * K);
    MI_H2D(ctx, t.d_inv, inv.data(), sizeof(uint16_t) * K);
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->turbo_tables[key] = t;
    *out                   = t;
    return MI_LTE_OK;
}

// Gold sequence c(n) = x1(n+1600) ^ x2(n+1600) (3GPP TS 36.211 7.2; generate_prs_c,
// liblte_phy.cc:9669-9704).  x1 does not depend on the seed; x2 is a linear function of the 31 seed
// bits.  Table layout: x1[w] and x2b[j][w] hold output bits 32w..32w+31 (bit b of the word = c index
// 32w+b) of the x1 sequence and of the x2 sequence seeded with 1<<j.  A kernel gets any word of any
// sequence as x1[w] ^ XOR_{j in seed} x2b[j][w].
int mi_ctx_gold_tables(mi_lte_ctx *ctx)
{
    if (ctx->d_gold_x1) return MI_LTE_OK;
    const uint32_t W = 4096; // 131072 bits >= the largest PDSCH allocation (100 PRB x 64QAM ~ 86400 bits)
    std::vector<uint32_t> x1w(W, 0), x2w((size_t)31 * W, 0);
    {
        uint32_t x1 = 0x54D21B24u; // x1 after the 1600-31 step warm-up (same constant as the reference)
        for (uint32_t n = 0; n < 32 * W; n++) {
            uint32_t nb = ((x1 >> 3) ^ x1) & 1u;
            x1          = (x1 >> 1) | (nb << 30);
            x1w[n >> 5] |= nb << (n & 31);
        }
    }
    for (int j = 0; j < 31; j++) {
        uint32_t x2 = 1u << j;
        for (uint32_t n = 0; n < 1600 - 31; n++) {
            uint32_t nb = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
            x2          = (x2 >> 1) | (nb << 30);
        }
        for (uint32_t n = 0; n < 32 * W; n++) {
            uint32_t nb = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
            x2          = (x2 >> 1) | (nb << 30);
            x2w[(size_t)j * W + (n >> 5)] |= nb << (n & 31);
        }
    }
    MI_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_gold_x1, sizeof(uint32_t) * W));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_gold_x2b, sizeof(uint32_t) * 31 * W));
    ctx->owned.push_back(ctx->d_gold_x1);
    ctx->owned.push_back(ctx->d_gold_x2b);
    MI_H2D(ctx, ctx->d_gold_x1, x1w.data(), sizeof(uint32_t) * W);
    MI_H2D(ctx, ctx->d_gold_x2b, x2w.data(), sizeof(uint32_t) * 31 * W);
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gold_words = W;
    return MI_LTE_OK;
}

// Forward-DFT twiddles exp(-2*pi*i*k/4096), evaluated in double and rounded once (the odd entries serve the uplink's half-sub-carrier
// rotation), followed by the per-pass tables the Stockham passes read (MI_FFT_TWC_* in ctx.hpp): for butterfly k of a radix-R pass over
// sub-transforms of length Ns the entries w, w^2 (R >= 4), w^4 (R = 8) with w = exp(-2*pi*i*k/(Ns*R)), side by side -- the same values the
// strided reads of the big table gave, but one or two 16-byte loads per butterfly out of a few cache lines per wavefront instead of three
// 8-byte loads that touched up to 64 lines each (the front end's FFT spent a third of its time on those).
int mi_ctx_fft_twiddles(mi_lte_ctx *ctx)
{
    if (ctx->d_fft_tw) return MI_LTE_OK;
    const uint32_t N = 4096;
    std::vector<float2> tw(N + MI_FFT_TWC_TOTAL, make_float2(0.0f, 0.0f));
    for (uint32_t k = 0; k < N; k++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)N;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    auto fill = [&](uint32_t off, uint32_t Ns, uint32_t R) { // entry k: tw[k * 4096 / (Ns * R)] and its 2nd / 4th power, R / 2 (at least 1) slots wide
        const uint32_t st = N / (Ns * R), wd = R >= 8 ? 4 : R == 4 ? 2 : 1;
        for (uint32_t k = 0; k < Ns; k++) {
            float2 *e = &tw[N + off + (size_t)k * wd];
            e[0] = tw[k * st];
            if (R >= 4) e[1] = tw[2 * k * st];
            if (R >= 8) e[2] = tw[4 * k * st];
            if (R == 16) e[3] = tw[8 * k * st];
        }
    };
    fill(MI_FFT_TWC_P2, 8, 8);
    fill(MI_FFT_TWC_P3, 64, 8);
    fill(MI_FFT_TWC_L2048, 512, 4);
    fill(MI_FFT_TWC_L1024, 512, 2);
    fill(MI_FFT_TWC_L256, 64, 4);
    fill(MI_FFT_TWC_L128, 64, 2);
    fill(MI_FFT_TWC_X2, 8, 16);
    fill(MI_FFT_TWC_X3, 128, 16);
    MI_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_fft_tw, sizeof(float2) * tw.size()));
    ctx->owned.push_back(ctx->d_fft_tw);
    MI_H2D(ctx, ctx->d_fft_tw, tw.data(), sizeof(float2) * tw.size());
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}

// CRC24A (36.212 5.1.1, g = 0x1864CFB; reference calc_crc, liblte_phy.cc:9713-9743) is linear over
// GF(2): a(x)*x^24 mod g == p(x) exactly when the XOR of x^e mod g over the set bits of the whole
// block a|p (e = distance of the bit from the end of the block) is zero.
// tab[MI_CRC_TAB_BIAS + e] = (x^e mod g) << 8 for e = -8 .. 6143: the remainder sits in the upper 24 bits, so that "times x" is a shift,
// the sign bit and one conditional XOR (k_turbo_vote derives a unit's sixteen weights from one entry that way); the eight negative
// exponents let the short last unit of a K % 16 == 8 block start from the same relative position as a full one.
int mi_ctx_crc_table(mi_lte_ctx *ctx)
{
    if (ctx->d_crc_tab) return MI_LTE_OK;
    const uint32_t N = 6144 + MI_CRC_TAB_BIAS;
    std::vector<uint32_t> tab(N);
    uint32_t r = 1; // x^0
    for (uint32_t e = 0; e < 6144; e++) {
        tab[MI_CRC_TAB_BIAS + e] = r << 8;
        r <<= 1;
        if (r & 0x1000000u) r ^= 0x1864CFBu;
    }
    r = 1;
    for (uint32_t e = 1; e <= MI_CRC_TAB_BIAS; e++) { // x^-e: divide by x (g's constant term is 1, so x is invertible)
        if (r & 1u) r ^= 0x1864CFBu;
        r >>= 1;
        tab[MI_CRC_TAB_BIAS - e] = r << 8;
    }
    MI_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_crc_tab, sizeof(uint32_t) * N));
    ctx->owned.push_back(ctx->d_crc_tab);
    MI_H2D(ctx, ctx->d_crc_tab, tab.data(), sizeof(uint32_t) * N);
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}
