// Micro-benchmark: issue rate of the packed-16-bit / bit-manipulation VALU ops a packed trellis
// kernel would use on gfx950 (inline asm so the opcode is exactly the one named).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;

#define ASMK(name, ASMSTR)                                                                   \
    __global__ __launch_bounds__(256) void name(int *out, int a, int b)                       \
    {                                                                                        \
        int v[16];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; i++) v[i] = (int)threadIdx.x * 65537 + i;  \
        int va = a + (int)threadIdx.x, vb = b;                                               \
        for (int it = 0; it < ITER; it++) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                   \
                asm volatile(ASMSTR : "+v"(v[i]) : "v"(va), "v"(vb));                        \
        }                                                                                    \
        int s = 0;                                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += v[i];                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }

ASMK(k_add_u32, "v_add_u32 %0, %0, %1")
ASMK(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
ASMK(k_pk_add_u16_opsel, "v_pk_add_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]")
ASMK(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
ASMK(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0")
ASMK(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
ASMK(k_pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
ASMK(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
ASMK(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
ASMK(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
ASMK(k_bfe_i32, "v_bfe_i32 %0, %0, 8, 8")
ASMK(k_mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
ASMK(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
ASMK(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
ASMK(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
ASMK(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x30")
ASMK(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
ASMK(k_add3, "v_add3_u32 %0, %0, %1, %2")
ASMK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
ASMK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
ASMK(k_cmp_lt, "v_cmp_lt_i32 vcc, %0, %1")
ASMK(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0")
ASMK(k_dpp_add, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

template <typename K> void run(const char *name, K kern)
{
    int *d;
    const int blocks = 256 * 8, threads = 256; // 8 waves per SIMD
    CHECK(hipMalloc(&d, sizeof(int) * blocks * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * (threads / 64) * ITER * 16;
    double per_simd_per_s = winstr / (ms * 1e-3) / (256 * 4);
    printf("%-22s %8.3f ms  %.3f G wave-instr/s/SIMD  -> %.2f cycles per wave64 instr @2.4GHz\n", name, ms, per_simd_per_s / 1e9,
           2.4e9 / per_simd_per_s);
    hipFree(d);
}
// one wave per SIMD, dependent vs independent issue
template <typename K> void run1(const char *name, K kern)
{
    int *d;
    const int blocks = 256 * 4, threads = 64;
    CHECK(hipMalloc(&d, sizeof(int) * blocks * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * ITER * 16;
    double per_simd_per_s = winstr / (ms * 1e-3) / (256 * 4);
    printf("%-22s %8.3f ms  (1 wave/SIMD) -> %.2f cycles per wave64 instr @2.4GHz\n", name, ms, 2.4e9 / per_simd_per_s);
    hipFree(d);
}
#define R(k) run(#k, k)
int main()
{
    R(k_add_u32); R(k_pk_add_u16); R(k_pk_add_u16_opsel); R(k_pk_sub_i16); R(k_pk_ashr_i16); R(k_pk_mul_lo_u16); R(k_pk_mad_i16);
    R(k_pk_min_i16); R(k_bfi_b32); R(k_perm_b32); R(k_bfe_i32); R(k_mul_i32_i24); R(k_mad_i32_i24); R(k_mul_lo_u32); R(k_alignbit);
    R(k_bitop3); R(k_lshl_add); R(k_add3); R(k_and_or); R(k_cndmask); R(k_cmp_lt); R(k_sdwa_add); R(k_dpp_add);
    run1("k_add_u32", k_add_u32); run1("k_pk_add_u16", k_pk_add_u16); run1("k_bfi_b32", k_bfi_b32); run1("k_cndmask", k_cndmask);
    return 0;
}
