// tools/ubench/store_rate.hip -- what a write-heavy kernel can reach on the device, by store width.
// The symbol rows of k_dl_fft (and the soft bits of k_pdsch_demod, the tiles of k_turbo_prep) were written 4 bytes per lane: a wave-wide
// store covers 256 bytes.  This program writes (and, for comparison, reads) the same buffer with 2 / 4 / 8 / 16 bytes per lane, rows of
// 4800 bytes like a symbol row, to see whether the store width -- not the byte count -- is what those kernels wait for.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V> __global__ __launch_bounds__(256) void k_fill(V *p, size_t n_per_block, V val)
{
    V *b = p + (size_t)blockIdx.x * n_per_block;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256) b[i] = val;
}
template <typename V> __global__ __launch_bounds__(256) void k_fill_nt(V *p, size_t n_per_block, V val)
{
    V *b = p + (size_t)blockIdx.x * n_per_block;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256) __builtin_nontemporal_store(val, b + i);
}
template <typename V> __global__ __launch_bounds__(256) void k_read(const V *p, size_t n_per_block, uint32_t *sink)
{
    const V *b = p + (size_t)blockIdx.x * n_per_block;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256) { V v = b[i]; uint16_t lo; __builtin_memcpy(&lo, &v, 2); acc ^= lo; }
    if (acc == 0x1234u) *sink = acc; // (a value the XOR of 16-bit words can take: the loads stay)
}
// like k_dl_fft2k's store phase: 128 threads, ten 4-byte stores per thread and plane at a 512-byte stride, two planes
__global__ __launch_bounds__(128) void k_rows4(float *p, size_t row_floats)
{
    float *re = p + (size_t)blockIdx.x * 2 * row_floats, *im = re + row_floats;
    for (int r = 0; r < 9; r++) { re[threadIdx.x + 128 * r] = 1.0f; im[threadIdx.x + 128 * r] = 2.0f; }
    if (threadIdx.x < 48) { re[threadIdx.x + 1152] = 1.0f; im[threadIdx.x + 1152] = 2.0f; }
}
__global__ __launch_bounds__(128) void k_rows16(float4 *p, size_t row_vec)
{
    float4 *re = p + (size_t)blockIdx.x * 2 * row_vec, *im = re + row_vec;
    const float4 a = make_float4(1, 1, 1, 1), b = make_float4(2, 2, 2, 2);
    for (int r = 0; r < 2; r++) { re[threadIdx.x + 128 * r] = a; im[threadIdx.x + 128 * r] = b; }
    if (threadIdx.x < 44) { re[threadIdx.x + 256] = a; im[threadIdx.x + 256] = b; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main()
{
    const size_t bytes = (size_t)4 << 30, blocks = 65536, per_block = bytes / blocks; // 64 KiB per workgroup
    void *p; uint32_t *sink;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&sink, 4));
    printf("buffer %zu MiB, %zu workgroups of 256 threads, %zu bytes each\n", bytes >> 20, blocks, per_block);
#define RUN(name, expr) do { const double ms = time_ms([&] { expr; }, 10); printf("%-40s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms * 1e-9); } while (0)
    RUN("store  2 B per lane", (k_fill<uint16_t><<<blocks, 256>>>((uint16_t *)p, per_block / 2, (uint16_t)7)));
    RUN("store  4 B per lane", (k_fill<float><<<blocks, 256>>>((float *)p, per_block / 4, 1.0f)));
    RUN("store  8 B per lane", (k_fill<float2><<<blocks, 256>>>((float2 *)p, per_block / 8, make_float2(1, 2))));
    RUN("store 16 B per lane", (k_fill<float4><<<blocks, 256>>>((float4 *)p, per_block / 16, make_float4(1, 2, 3, 4))));
    RUN("store  4 B per lane, non-temporal", (k_fill_nt<float><<<blocks, 256>>>((float *)p, per_block / 4, 1.0f)));
    RUN("store 16 B per lane, non-temporal", (k_fill_nt<f4><<<blocks, 256>>>((f4 *)p, per_block / 16, f4{1, 2, 3, 4})));
    RUN("load   2 B per lane", (k_read<uint16_t><<<blocks, 256>>>((const uint16_t *)p, per_block / 2, sink)));
    RUN("load   4 B per lane", (k_read<float><<<blocks, 256>>>((const float *)p, per_block / 4, sink)));
    RUN("load  16 B per lane", (k_read<float4><<<blocks, 256>>>((const float4 *)p, per_block / 16, sink)));
    {
        const size_t rows = bytes / 9600;
        const double ms4 = time_ms([&] { k_rows4<<<rows, 128>>>((float *)p, 1200); }, 10);
        printf("%-40s %7.3f ms  %6.2f TB/s\n", "symbol rows, 4 B per lane (as k_dl_fft)", ms4, rows * 9600.0 / ms4 * 1e-9);
        const double ms16 = time_ms([&] { k_rows16<<<rows, 128>>>((float4 *)p, 300); }, 10);
        printf("%-40s %7.3f ms  %6.2f TB/s\n", "symbol rows, 16 B per lane", ms16, rows * 9600.0 / ms16 * 1e-9);
    }
    return 0;
}
