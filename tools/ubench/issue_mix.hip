// Micro-benchmark (round 4): WHEN does a gfx950 SIMD reach the 2-cycles-per-wave64-instruction rate that issue_rate.hip measures for
// v_add_u32 / v_sub_u32 / v_and / v_or / v_xor / v_mov / v_lshrrev / v_ashrrev / v_bitop3 / v_add|mul|fma_f32 in isolation?
// Hand-allocated registers (one asm block per loop) so that operand banks (VGPR number mod 4), SGPR / literal / inline operands,
// opcode alternation and dependency distance are what the line says they are.  Times are wall x measured clock / instructions, per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/issue_mix tools/ubench/issue_mix.hip && tools/ubench/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Stamp { long long cyc, real; };
constexpr int ITER = 2048; // loop trips; every body below is 16 VALU instructions

#define CLOB "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "s20", "s21", "s22", "s23", "scc", "vcc"
#define KERN(name, BODY)                                                                     \
    __global__ __launch_bounds__(256) void name(int *out, Stamp *st, int a, int b)           \
    {                                                                                        \
        __syncthreads();                                                                     \
        const long long r0 = wall_clock64(), t0 = clock64();                                 \
        asm volatile("s_movk_i32 s20, 0x800\n\ts_mov_b32 s21, 0x00040004\n\ts_mov_b32 s22, 0x80008000\n"  \
                     "1:\n" BODY "\ts_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n" ::: CLOB); \
        const long long t1 = clock64(), r1 = wall_clock64();                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a;                                      \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0}; \
    }
#define R4(A, B, C, D) A B C D
#define X16(S) S S S S S S S S S S S S S S S S

// one opcode, operand banks controlled: dst/src0 in bank 0, second source in bank 1 | bank 0
KERN(add_nobank, "v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v12, v12, v25\n v_add_u32 v16, v16, v29\n v_add_u32 v20, v20, v33\n v_add_u32 v24, v24, v37\n v_add_u32 v28, v28, v17\n v_add_u32 v32, v32, v21\n"
                 "v_add_u32 v36, v36, v25\n v_add_u32 v40, v40, v29\n v_add_u32 v4, v4, v33\n v_add_u32 v8, v8, v37\n v_add_u32 v12, v12, v17\n v_add_u32 v16, v16, v21\n v_add_u32 v20, v20, v25\n v_add_u32 v24, v24, v29\n")
KERN(add_bank, "v_add_u32 v4, v4, v16\n v_add_u32 v8, v8, v20\n v_add_u32 v12, v12, v24\n v_add_u32 v16, v16, v28\n v_add_u32 v20, v20, v32\n v_add_u32 v24, v24, v36\n v_add_u32 v28, v28, v40\n v_add_u32 v32, v32, v4\n"
               "v_add_u32 v36, v36, v8\n v_add_u32 v40, v40, v12\n v_add_u32 v4, v4, v16\n v_add_u32 v8, v8, v20\n v_add_u32 v12, v12, v24\n v_add_u32 v16, v16, v28\n v_add_u32 v20, v20, v32\n v_add_u32 v24, v24, v36\n")
// distinct destination (three-address): dst bank 2, sources banks 0 and 1
KERN(add_3addr, "v_add_u32 v2, v4, v17\n v_add_u32 v6, v8, v21\n v_add_u32 v10, v12, v25\n v_add_u32 v14, v16, v29\n v_add_u32 v18, v20, v33\n v_add_u32 v22, v24, v37\n v_add_u32 v26, v28, v17\n v_add_u32 v30, v32, v21\n"
                "v_add_u32 v34, v36, v25\n v_add_u32 v38, v40, v29\n v_add_u32 v2, v4, v33\n v_add_u32 v6, v8, v37\n v_add_u32 v10, v12, v17\n v_add_u32 v14, v16, v21\n v_add_u32 v18, v20, v25\n v_add_u32 v22, v24, v29\n")
// second operand an SGPR / a 32-bit literal / an inline constant
KERN(add_sgpr, X16("v_add_u32 v4, s21, v4\n") )
KERN(add_sgpr16, "v_add_u32 v1, s21, v1\n v_add_u32 v2, s21, v2\n v_add_u32 v3, s21, v3\n v_add_u32 v4, s21, v4\n v_add_u32 v5, s21, v5\n v_add_u32 v6, s21, v6\n v_add_u32 v7, s21, v7\n v_add_u32 v8, s21, v8\n"
                 "v_add_u32 v9, s21, v9\n v_add_u32 v10, s21, v10\n v_add_u32 v11, s21, v11\n v_add_u32 v12, s21, v12\n v_add_u32 v13, s21, v13\n v_add_u32 v14, s21, v14\n v_add_u32 v15, s21, v15\n v_add_u32 v16, s21, v16\n")
KERN(add_lit16, "v_add_u32 v1, 0x40004, v1\n v_add_u32 v2, 0x40004, v2\n v_add_u32 v3, 0x40004, v3\n v_add_u32 v4, 0x40004, v4\n v_add_u32 v5, 0x40004, v5\n v_add_u32 v6, 0x40004, v6\n v_add_u32 v7, 0x40004, v7\n v_add_u32 v8, 0x40004, v8\n"
                "v_add_u32 v9, 0x40004, v9\n v_add_u32 v10, 0x40004, v10\n v_add_u32 v11, 0x40004, v11\n v_add_u32 v12, 0x40004, v12\n v_add_u32 v13, 0x40004, v13\n v_add_u32 v14, 0x40004, v14\n v_add_u32 v15, 0x40004, v15\n v_add_u32 v16, 0x40004, v16\n")
KERN(add_inl16, "v_add_u32 v1, 4, v1\n v_add_u32 v2, 4, v2\n v_add_u32 v3, 4, v3\n v_add_u32 v4, 4, v4\n v_add_u32 v5, 4, v5\n v_add_u32 v6, 4, v6\n v_add_u32 v7, 4, v7\n v_add_u32 v8, 4, v8\n"
                "v_add_u32 v9, 4, v9\n v_add_u32 v10, 4, v10\n v_add_u32 v11, 4, v11\n v_add_u32 v12, 4, v12\n v_add_u32 v13, 4, v13\n v_add_u32 v14, 4, v14\n v_add_u32 v15, 4, v15\n v_add_u32 v16, 4, v16\n")
// bitop3: three VGPRs in three banks | all in one bank | third operand an SGPR
KERN(bitop3_nobank, "v_bitop3_b32 v4, v4, v17, v18 bitop3:0xe4\n v_bitop3_b32 v8, v8, v21, v22 bitop3:0xe4\n v_bitop3_b32 v12, v12, v25, v26 bitop3:0xe4\n v_bitop3_b32 v16, v16, v29, v30 bitop3:0xe4\n"
                    "v_bitop3_b32 v20, v20, v33, v34 bitop3:0xe4\n v_bitop3_b32 v24, v24, v37, v38 bitop3:0xe4\n v_bitop3_b32 v28, v28, v17, v18 bitop3:0xe4\n v_bitop3_b32 v32, v32, v21, v22 bitop3:0xe4\n"
                    "v_bitop3_b32 v36, v36, v25, v26 bitop3:0xe4\n v_bitop3_b32 v40, v40, v29, v30 bitop3:0xe4\n v_bitop3_b32 v4, v4, v33, v34 bitop3:0xe4\n v_bitop3_b32 v8, v8, v37, v38 bitop3:0xe4\n"
                    "v_bitop3_b32 v12, v12, v17, v18 bitop3:0xe4\n v_bitop3_b32 v16, v16, v21, v22 bitop3:0xe4\n v_bitop3_b32 v20, v20, v25, v26 bitop3:0xe4\n v_bitop3_b32 v24, v24, v29, v30 bitop3:0xe4\n")
KERN(bitop3_bank, "v_bitop3_b32 v4, v4, v16, v20 bitop3:0xe4\n v_bitop3_b32 v8, v8, v20, v24 bitop3:0xe4\n v_bitop3_b32 v12, v12, v24, v28 bitop3:0xe4\n v_bitop3_b32 v16, v16, v28, v32 bitop3:0xe4\n"
                  "v_bitop3_b32 v20, v20, v32, v36 bitop3:0xe4\n v_bitop3_b32 v24, v24, v36, v40 bitop3:0xe4\n v_bitop3_b32 v28, v28, v40, v4 bitop3:0xe4\n v_bitop3_b32 v32, v32, v4, v8 bitop3:0xe4\n"
                  "v_bitop3_b32 v36, v36, v8, v12 bitop3:0xe4\n v_bitop3_b32 v40, v40, v12, v16 bitop3:0xe4\n v_bitop3_b32 v4, v4, v16, v20 bitop3:0xe4\n v_bitop3_b32 v8, v8, v20, v24 bitop3:0xe4\n"
                  "v_bitop3_b32 v12, v12, v24, v28 bitop3:0xe4\n v_bitop3_b32 v16, v16, v28, v32 bitop3:0xe4\n v_bitop3_b32 v20, v20, v32, v36 bitop3:0xe4\n v_bitop3_b32 v24, v24, v36, v40 bitop3:0xe4\n")
KERN(bitop3_sgpr, "v_bitop3_b32 v4, v4, v17, s22 bitop3:0xe4\n v_bitop3_b32 v8, v8, v21, s22 bitop3:0xe4\n v_bitop3_b32 v12, v12, v25, s22 bitop3:0xe4\n v_bitop3_b32 v16, v16, v29, s22 bitop3:0xe4\n"
                  "v_bitop3_b32 v20, v20, v33, s22 bitop3:0xe4\n v_bitop3_b32 v24, v24, v37, s22 bitop3:0xe4\n v_bitop3_b32 v28, v28, v17, s22 bitop3:0xe4\n v_bitop3_b32 v32, v32, v21, s22 bitop3:0xe4\n"
                  "v_bitop3_b32 v36, v36, v25, s22 bitop3:0xe4\n v_bitop3_b32 v40, v40, v29, s22 bitop3:0xe4\n v_bitop3_b32 v4, v4, v33, s22 bitop3:0xe4\n v_bitop3_b32 v8, v8, v37, s22 bitop3:0xe4\n"
                  "v_bitop3_b32 v12, v12, v17, s22 bitop3:0xe4\n v_bitop3_b32 v16, v16, v21, s22 bitop3:0xe4\n v_bitop3_b32 v20, v20, v25, s22 bitop3:0xe4\n v_bitop3_b32 v24, v24, v29, s22 bitop3:0xe4\n")
// alternation of opcodes, no bank conflicts, all independent
#define ALT8(A, B) A " v4, v4, v17\n" B " v8, v8, v21\n" A " v12, v12, v25\n" B " v16, v16, v29\n" A " v20, v20, v33\n" B " v24, v24, v37\n" A " v28, v28, v17\n" B " v32, v32, v21\n" \
                   A " v36, v36, v25\n" B " v40, v40, v29\n" A " v4, v4, v33\n" B " v8, v8, v37\n" A " v12, v12, v17\n" B " v16, v16, v21\n" A " v20, v20, v25\n" B " v24, v24, v29\n"
KERN(alt_add_sub, ALT8("v_add_u32", "v_sub_u32"))
KERN(alt_add_xor, ALT8("v_add_u32", "v_xor_b32"))
KERN(alt_add_pkadd, ALT8("v_add_u32", "v_pk_add_u16"))
KERN(alt_pkadd_pksub, ALT8("v_pk_add_u16", "v_pk_sub_i16"))
KERN(alt_add_lshr, ALT8("v_add_u32", "v_lshrrev_b32"))
KERN(alt_add_addf32, ALT8("v_add_u32", "v_add_f32"))
// 3 full-rate then 1 half-rate (the SWAR mix is about 3.4 : 1)
#define MIX31(A, B) A " v4, v4, v17\n" A " v8, v8, v21\n" A " v12, v12, v25\n" B " v16, v16, v29\n" A " v20, v20, v33\n" A " v24, v24, v37\n" A " v28, v28, v17\n" B " v32, v32, v21\n" \
                    A " v36, v36, v25\n" A " v40, v40, v29\n" A " v4, v4, v33\n" B " v8, v8, v37\n" A " v12, v12, v17\n" A " v16, v16, v21\n" A " v20, v20, v25\n" B " v24, v24, v29\n"
KERN(mix31_add_pkadd, MIX31("v_add_u32", "v_pk_add_u16"))
// add feeding a bitop3 feeding an add ... : dependency distance 1 (one chain), 2 (two chains), 4
KERN(dep1, X16("v_add_u32 v4, v4, v17\n"))
KERN(dep2, "v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n"
           "v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n")
KERN(dep4, "v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v12, v12, v25\n v_add_u32 v16, v16, v29\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v12, v12, v25\n v_add_u32 v16, v16, v29\n"
           "v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v12, v12, v25\n v_add_u32 v16, v16, v29\n v_add_u32 v4, v4, v17\n v_add_u32 v8, v8, v21\n v_add_u32 v12, v12, v25\n v_add_u32 v16, v16, v29\n")
// a scalar instruction between every two vector ones (what a clobber of vcc in an inline asm makes the compiler insert)
KERN(add_snop, X16("v_add_u32 v4, v4, v17\n s_nop 0\n"))
KERN(add_salu, "v_add_u32 v4, v4, v17\n s_add_u32 s23, s23, 1\n v_add_u32 v8, v8, v21\n s_add_u32 s23, s23, 1\n v_add_u32 v12, v12, v25\n s_add_u32 s23, s23, 1\n v_add_u32 v16, v16, v29\n s_add_u32 s23, s23, 1\n"
               "v_add_u32 v20, v20, v33\n s_add_u32 s23, s23, 1\n v_add_u32 v24, v24, v37\n s_add_u32 s23, s23, 1\n v_add_u32 v28, v28, v17\n s_add_u32 s23, s23, 1\n v_add_u32 v32, v32, v21\n s_add_u32 s23, s23, 1\n"
               "v_add_u32 v36, v36, v25\n s_add_u32 s23, s23, 1\n v_add_u32 v40, v40, v29\n s_add_u32 s23, s23, 1\n v_add_u32 v4, v4, v33\n s_add_u32 s23, s23, 1\n v_add_u32 v8, v8, v37\n s_add_u32 s23, s23, 1\n"
               "v_add_u32 v12, v12, v17\n s_add_u32 s23, s23, 1\n v_add_u32 v16, v16, v21\n s_add_u32 s23, s23, 1\n v_add_u32 v20, v20, v25\n s_add_u32 s23, s23, 1\n v_add_u32 v24, v24, v29\n s_add_u32 s23, s23, 1\n")

struct Result { double mhz, ms; };
template <typename K> Result launch(K kern, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256;
    int   *d;
    Stamp *st;
    CHECK(hipMalloc(&d, sizeof(int) * blocks * threads));
    CHECK(hipMalloc(&st, sizeof(Stamp) * blocks * 4));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, st, 3, 5);
    CHECK(hipDeviceSynchronize());
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, st, 3, 5);
    (void)hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
    double cs = 0, rs = 0;
    for (auto &x : h) { cs += x.cyc; rs += x.real; }
    (void)hipFree(d); (void)hipFree(st);
    return Result{cs / rs * 100.0, ms};
}
template <typename K> void row(const char *name, K k)
{
    printf("%-18s", name);
    for (int w : {1, 2, 4, 8}) {
        Result r = launch(k, w);
        printf(" %6.2f", r.ms * 1e-3 * r.mhz * 1e6 / ((double)ITER * 16 * w));
    }
    printf("\n");
}
#define ROW(k) row(#k, k)
int main()
{
    printf("cycles per wave64 VALU instruction per SIMD (wall x clock / (instructions per wavefront x wavefronts per SIMD))\n%-18s %6s %6s %6s %6s\n", "", "w=1", "w=2", "w=4", "w=8");
    ROW(add_nobank); ROW(add_bank); ROW(add_3addr); ROW(add_sgpr); ROW(add_sgpr16); ROW(add_lit16); ROW(add_inl16);
    ROW(bitop3_nobank); ROW(bitop3_bank); ROW(bitop3_sgpr);
    ROW(alt_add_sub); ROW(alt_add_xor); ROW(alt_add_pkadd); ROW(alt_pkadd_pksub); ROW(alt_add_lshr); ROW(alt_add_addf32); ROW(mix31_add_pkadd);
    ROW(dep1); ROW(dep2); ROW(dep4); ROW(add_snop); ROW(add_salu);
    return 0;
}
