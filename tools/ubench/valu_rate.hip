// Micro-benchmark: per-instruction VALU throughput on gfx950 for the ops the trellis kernels use.
// Each kernel runs ITER iterations of 16 independent chains of one op; 8 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;

#define KERNEL(name, TYPE, INIT, BODY)                                                      \
    __global__ __launch_bounds__(256) void name(TYPE *out, TYPE a, TYPE b)                    \
    {                                                                                       \
        TYPE v[16];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) v[i] = INIT;                         \
        for (int it = 0; it < ITER; it++) {                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; i++) { BODY; }                        \
        }                                                                                   \
        TYPE s = 0;                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += v[i];                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                     \
    }

KERNEL(k_add_u32, int, (int)threadIdx.x + i, v[i] = v[i] + a)
KERNEL(k_sub_f32, float, (float)threadIdx.x + i, v[i] = v[i] - a)
KERNEL(k_fma_f32, float, (float)threadIdx.x + i, v[i] = fmaf(v[i], a, b))
KERNEL(k_mad_i24, int, (int)threadIdx.x + i, v[i] = __mul24(v[i], a) + b)
KERNEL(k_cmp_cnd_i32, int, (int)threadIdx.x + i, v[i] = (v[i] < a) ? v[(i + 1) & 15] : b)
KERNEL(k_cmp_cnd_f32, float, (float)threadIdx.x + i, v[i] = (v[i] < a) ? v[(i + 1) & 15] : b)
KERNEL(k_alignbit, int, (int)threadIdx.x + i, v[i] = (int)__builtin_amdgcn_alignbit((unsigned)v[i], (unsigned)a, 31))
KERNEL(k_max_i32, int, (int)threadIdx.x + i, v[i] = max(v[i], a + i))
KERNEL(k_max_f32, float, (float)threadIdx.x + i, v[i] = fmaxf(v[i], a + i))
KERNEL(k_min3_i32, int, (int)threadIdx.x + i, v[i] = min(min(v[i], a + i), b))
KERNEL(k_and_or, int, (int)threadIdx.x + i, v[i] = (v[i] & a) | b)
KERNEL(k_ashr, int, (int)threadIdx.x + i, v[i] = (v[i] >> 1) + 1)

typedef float float2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_add_f32(float *out, float a, float b)
{
    float2_t v[16], c = {a, b};
    for (int i = 0; i < 16; i++) v[i] = float2_t{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] + c;
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, typename K> void run(const char *name, K kern, double ops_per_body)
{
    T *d;
    const int blocks = 256 * 8, threads = 256; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    CHECK(hipMalloc(&d, sizeof(T) * blocks * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, (T)3, (T)5);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, (T)3, (T)5);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * (threads / 64) * ITER * 16 * ops_per_body; // wave-instructions
    double per_simd_per_s = winstr / (ms * 1e-3) / (256 * 4);
    printf("%-16s %8.3f ms  %.2f G wave-instr/s/SIMD  -> %.2f cycles per wave64 instr @2.4GHz (body = %.0f instr)\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, ops_per_body);
    hipFree(d);
}

int main()
{
    run<int>("v_add_u32", k_add_u32, 1);
    run<float>("v_sub_f32", k_sub_f32, 1);
    run<float>("v_fma_f32", k_fma_f32, 1);
    run<int>("v_mad_i32_i24", k_mad_i24, 1);
    run<int>("cmp+cndmask i32", k_cmp_cnd_i32, 2);
    run<float>("cmp+cndmask f32", k_cmp_cnd_f32, 2);
    run<int>("v_alignbit", k_alignbit, 1);
    run<int>("v_max_i32", k_max_i32, 1);
    run<float>("v_max_f32", k_max_f32, 1);
    run<int>("v_min3_i32", k_min3_i32, 1);
    run<int>("v_and_or_b32", k_and_or, 1);
    run<int>("ashr+add", k_ashr, 2);
    run<float>("v_pk_add_f32", k_pk_add_f32, 1);
    return 0;
}
