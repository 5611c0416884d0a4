// Is v_cmp -> v_cndmask slow because of the VCC round trip, or inherently?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
// A: compiler-chosen (select in C++)
__global__ __launch_bounds__(256) void k_c(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (v[i] < a + i) ? v[(i + 1) & 7] : b;
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// B: 8 compares into 8 SGPR pairs, then 8 cndmasks
__global__ __launch_bounds__(256) void k_asm(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
        unsigned long long m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(v[i]), "v"(a + i));
        int n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(n[i]) : "v"(b), "v"(v[(i + 1) & 7]), "s"(m[i]));
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = n[i];
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// C: cmp only (result consumed via s_or into a scalar accumulate)   D: cndmask only with fixed mask
__global__ __launch_bounds__(256) void k_cmp_only(int *out, int a, int b)
{
    int v[8]; unsigned long long acc = 0;
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { unsigned long long m; asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(v[i]), "v"(a + it)); acc ^= m; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)acc + v[0];
}
__global__ __launch_bounds__(256) void k_cnd_only(int *out, int a, int b)
{
    int v[8]; unsigned long long m = __ballot(threadIdx.x & 1);
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]), "s"(m));
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// E: max3/min via v_max_i32 dependent on loop variable to defeat hoisting
__global__ __launch_bounds__(256) void k_max(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_max_i32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 3) & 7]));
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_add(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 3) & 7]));
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bfi(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 3) & 7]), "v"(v[(i + 5) & 7]));
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ashr(int *out, int a, int b)
{
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(v[i]) : "v"(v[(i + 3) & 7]));
    }
    int s = 0; for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> void run(const char *name, K kern, double instr_per_iter, int blocks_per_cu)
{
    int *d; hipMalloc(&d, 4 * 256 * 8 * 256);
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3, 5); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3, 5); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITER * instr_per_iter, rate = winstr / (ms * 1e-3) / 1024;
    printf("%-28s waves/SIMD=%d  %7.3f ms  %.2f cycles per wave64 instr @2.4GHz\n", name, blocks_per_cu, ms, 2.4e9 / rate);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 8}) {
        run("C++ select (cmp+cnd)", k_c, 16, w);
        run("asm 8xcmp_e64 then 8xcnd_e64", k_asm, 16, w);
        run("v_cmp_e64 only (+s_xor)", k_cmp_only, 8, w);
        run("v_cndmask_e64 only", k_cnd_only, 8, w);
        run("v_max_i32", k_max, 8, w);
        run("v_add_u32", k_add, 8, w);
        run("v_bfi_b32", k_bfi, 8, w);
        run("v_ashrrev_i32", k_ashr, 8, w);
    }
}
