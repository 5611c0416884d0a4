// tools/ubench/nt_copy.hip -- which half of a streaming kernel the non-temporal hint pays for on this device: a 16-bytes-per-lane copy of 1 GiB
// with the hint on the loads, on the stores, on both, on neither; read-only (sum) and write-only (fill) kernels as well.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/nt_copy.hip -o gpurun_out/nt_copy && gpurun_out/nt_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NTL, bool NTS, int U> __global__ void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    u32x4        v[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
        const size_t i = base + (size_t)j * blockDim.x;
        if (i < n) v[j] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
        const size_t i = base + (size_t)j * blockDim.x;
        if (i < n) { if (NTS) __builtin_nontemporal_store(v[j], dst + i); else dst[i] = v[j]; }
    }
}
template <bool NT> __global__ void k_read(const u32x4 *__restrict__ src, uint32_t *__restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
    uint32_t     acc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t i = base + (size_t)j * blockDim.x;
        if (i < n) { u32x4 v = NT ? __builtin_nontemporal_load(src + i) : src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT> __global__ void k_fill(u32x4 *__restrict__ dst, size_t n)
{
    const size_t base = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t i = base + (size_t)j * blockDim.x;
        u32x4        v = {(uint32_t)i, 1u, 2u, 3u};
        if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
    }
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    u32x4 *a, *b; uint32_t *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0x5A, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s\n", name, moved * 10 / (ms * 1e-3) / 1e9);
    };
    const dim3 blk(256), g4((unsigned)((n + 1023) / 1024)), g1((unsigned)((n + 255) / 256)), g8((unsigned)((n + 2047) / 2048));
    time("copy  U=4  loads plain, stores plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<false, false, 4>), g4, blk, 0, 0, a, b, n); });
    time("copy  U=4  loads nt,    stores plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<true, false, 4>), g4, blk, 0, 0, a, b, n); });
    time("copy  U=4  loads plain, stores nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<false, true, 4>), g4, blk, 0, 0, a, b, n); });
    time("copy  U=4  loads nt,    stores nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<true, true, 4>), g4, blk, 0, 0, a, b, n); });
    time("copy  U=1  loads nt,    stores nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<true, true, 1>), g1, blk, 0, 0, a, b, n); });
    time("copy  U=1  loads plain, stores plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<false, false, 1>), g1, blk, 0, 0, a, b, n); });
    time("copy  U=8  loads nt,    stores nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<true, true, 8>), g8, blk, 0, 0, a, b, n); });
    time("read  U=4  plain", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_read<false>), g4, blk, 0, 0, a, o, n); });
    time("read  U=4  nt", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_read<true>), g4, blk, 0, 0, a, o, n); });
    time("fill  U=4  plain", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fill<false>), g4, blk, 0, 0, b, n); });
    time("fill  U=4  nt", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fill<true>), g4, blk, 0, 0, b, n); });
    return 0;
}
