// Micro-benchmark (round 4): what does a sub-dword LDS read cost on gfx950?  k_turbo_prep / k_turbo_perm / k_turbo_vote gather bytes
// (ds_read_u8 / ds_read_i8 / ds_read_u16) out of a staged block -- ~80 per thread in k_turbo_prep, whose LDS pipe is its busiest
// resource (SQ_ACTIVE_INST_LDS: 72 % of the kernel).  MI355X_MICROARCH.md lists the dword-and-wider reads only.
// Each kernel issues ITER x 16 independent reads per thread; 256-thread workgroups, 8 per CU (8 wavefronts per SIMD), 16 KB of LDS each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_rate tools/ubench/lds_rate.hip && tools/ubench/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 512;

// pattern 0: lane-linear (address = lane * width); 1: the rate-un-matching gather (even lanes one column, odd lanes another, row = lane / 2);
// 2: pseudo-random bytes; 3: all lanes one address
__device__ __forceinline__ uint32_t addr_of(int pat, uint32_t lane, uint32_t i, uint32_t width)
{
    switch (pat) {
    case 0: return (lane * width + i * 64 * width) & 16383u;
    case 1: return (((lane & 1) ? 5000u : 0u) + (lane >> 1) + 103u * i) & 16383u & ~(width - 1);
    case 2: return ((lane * 2654435761u + i * 40503u) >> 7) & 16383u & ~(width - 1);
    default: return (i * 64u) & 16383u;
    }
}
#define KERN(name, ASMSTR, WIDTH, NREG)                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t *out, int pat)                         \
    {                                                                                           \
        __shared__ __attribute__((aligned(16))) uint8_t lds[16384];                             \
        for (uint32_t i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u; \
        __syncthreads();                                                                        \
        uint32_t a[16];                                                                         \
        for (int i = 0; i < 16; i++) a[i] = addr_of(pat, threadIdx.x & 63, i, WIDTH) + (uint32_t)(uintptr_t)lds; \
        uint32_t acc = 0;                                                                       \
        for (int it = 0; it < ITER; it++) {                                                     \
            uint32_t v[16][NREG];                                                               \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                    \
                if (NREG == 1) asm volatile(ASMSTR : "=v"(v[i][0]) : "v"(a[i]));                \
                else if (NREG == 2) asm volatile(ASMSTR : "=v"(*reinterpret_cast<uint2 *>(v[i])) : "v"(a[i])); \
                else asm volatile(ASMSTR : "=v"(*reinterpret_cast<uint4 *>(v[i])) : "v"(a[i])); \
            }                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
            _Pragma("unroll") for (int i = 0; i < 16; i++) acc += v[i][0];                      \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                              \
    }
KERN(k_u8, "ds_read_u8 %0, %1", 1, 1)
KERN(k_i8, "ds_read_i8 %0, %1", 1, 1)
KERN(k_u16, "ds_read_u16 %0, %1", 2, 1)
KERN(k_b32, "ds_read_b32 %0, %1", 4, 1)
KERN(k_b64, "ds_read_b64 %0, %1", 8, 2)
KERN(k_b128, "ds_read_b128 %0, %1", 16, 4)
KERN(k_u8_d16, "ds_read_u8_d16 %0, %1", 1, 1)

template <typename K> void run(const char *name, K kern)
{
    uint32_t *d;
    const int blocks = 256 * 8;
    CHECK(hipMalloc(&d, sizeof(uint32_t) * blocks * 256));
    printf("%-16s", name);
    for (int pat = 0; pat < 4; pat++) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, pat);
        CHECK(hipDeviceSynchronize());
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, pat);
        (void)hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr_per_cu = (double)blocks * 4 * ITER * 16 / 256; // wave-instructions through one CU's LDS pipe
        printf("  %7.2f", ms * 1e-3 * 2.4e9 / winstr_per_cu);
    }
    printf("\n");
    (void)hipFree(d);
}
int main()
{
    printf("LDS cycles (at 2.4 GHz) per wave64 read instruction per CU, 8 wavefronts per SIMD\n%-16s  %7s  %7s  %7s  %7s\n", "", "linear", "rm-gath", "random", "same");
    run("ds_read_u8", k_u8); run("ds_read_i8", k_i8); run("ds_read_u8_d16", k_u8_d16); run("ds_read_u16", k_u16); run("ds_read_b32", k_b32); run("ds_read_b64", k_b64); run("ds_read_b128", k_b128);
    return 0;
}
