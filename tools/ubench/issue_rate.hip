// Micro-benchmark (round 4, review item 1): how many cycles does one wave64 VALU instruction occupy a gfx950 SIMD for?
//
// DESIGN.md priced every instruction floor at 4 cycles per wave64 instruction; MI355X_MICROARCH.md says "SIMD-32 ... 2 cycles".
// This program measures it, per opcode of the trellis loops (k_turbo_siso, k_bcjr_half), at 1 / 2 / 4 / 8 resident wavefronts per SIMD,
// with independent chains (throughput) and with one dependent chain (latency), in shader cycles (s_memtime) and in wall time
// (s_memrealtime, 100 MHz), so the clock the SIMDs really ran at is part of the output.  The last section times the forward
// add-compare-select step of k_turbo_siso (turbo.hip: acs_step2, restated here with register-resident inputs: no memory, no LDS)
// so that the kernel's measured cycles per trellis step can be set against the same arithmetic with nothing to wait for.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/issue_rate tools/ubench/issue_rate.hip && tools/ubench/issue_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2048; // x 16 instructions

struct Stamp { long long cyc, real; };

#define PROLOGUE                                                                             \
    int v[16];                                                                               \
    _Pragma("unroll") for (int i = 0; i < 16; i++) v[i] = (int)threadIdx.x * 65537 + i * 257; \
    int va = a + (int)threadIdx.x * 3, vb = b ^ (int)threadIdx.x;                            \
    asm volatile("" : "+v"(va), "+v"(vb));                                                   \
    __syncthreads();                                                                         \
    const long long r0 = wall_clock64(), t0 = clock64();
#define EPILOGUE                                                                             \
    const long long t1 = clock64(), r1 = wall_clock64();                                     \
    int s = 0;                                                                               \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s += v[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};

// 16 independent chains of one opcode
#define INDEP(name, ASMSTR)                                                                  \
    __global__ __launch_bounds__(256) void name(int *out, Stamp *st, int a, int b)           \
    {                                                                                        \
        PROLOGUE                                                                             \
        for (int it = 0; it < ITER; it++) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASMSTR : "+v"(v[i]) : "v"(va), "v"(vb) : "vcc"); \
        }                                                                                    \
        EPILOGUE                                                                             \
    }
// one dependent chain of the same opcode
#define DEP(name, ASMSTR)                                                                    \
    __global__ __launch_bounds__(256) void name(int *out, Stamp *st, int a, int b)           \
    {                                                                                        \
        PROLOGUE                                                                             \
        for (int it = 0; it < ITER; it++) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASMSTR : "+v"(v[0]) : "v"(va), "v"(vb) : "vcc"); \
        }                                                                                    \
        EPILOGUE                                                                             \
    }
#define BOTH(stem, ASMSTR) INDEP(i_##stem, ASMSTR) DEP(d_##stem, ASMSTR)

BOTH(add_u32, "v_add_u32 %0, %0, %1")
BOTH(and_b32, "v_and_b32 %0, %0, %1")
BOTH(xor_b32, "v_xor_b32 %0, %0, %1")
BOTH(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
BOTH(mov_b32, "v_mov_b32 %0, %1")
BOTH(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
BOTH(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
BOTH(pk_ashrrev_i16, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]")
BOTH(pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
BOTH(pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]")
BOTH(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
BOTH(pk_min_i16, "v_pk_min_i16 %0, %0, %1")
BOTH(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
BOTH(bitop3_b32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca")
BOTH(bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
BOTH(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
BOTH(or3_b32, "v_or3_b32 %0, %0, %1, %2")
BOTH(lshl_or_b32, "v_lshl_or_b32 %0, %0, 1, %1")
BOTH(lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
BOTH(add3_u32, "v_add3_u32 %0, %0, %1, %2")
BOTH(perm_b32, "v_perm_b32 %0, %0, %1, %2")
BOTH(bfe_u32, "v_bfe_u32 %0, %0, 3, 8")
BOTH(bfe_i32, "v_bfe_i32 %0, %0, 8, 8")
BOTH(alignbit_b32, "v_alignbit_b32 %0, %0, %1, 31")
BOTH(max_i32, "v_max_i32 %0, %0, %1")
BOTH(max3_i32, "v_max3_i32 %0, %0, %1, %2")
BOTH(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
BOTH(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
BOTH(add_f32, "v_add_f32 %0, %0, %1")
BOTH(mul_f32, "v_mul_f32 %0, %0, %1")
BOTH(fma_f32, "v_fma_f32 %0, %0, %1, %2")
BOTH(max_f32, "v_max_f32 %0, %0, %1")
BOTH(cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
BOTH(add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0")
BOTH(add_u32_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
BOTH(sad_u8, "v_sad_u8 %0, %0, %1, %2")
BOTH(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
BOTH(rcp_f32, "v_rcp_f32 %0, %0")
BOTH(nop, "s_nop 0")
BOTH(sub_u32, "v_sub_u32 %0, %0, %1")
BOTH(subrev_u32, "v_subrev_u32 %0, %0, %1")
BOTH(or_b32, "v_or_b32 %0, %0, %1")
BOTH(not_b32, "v_not_b32 %0, %0")
BOTH(xnor_b32, "v_xnor_b32 %0, %0, %1")
BOTH(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
BOTH(ashrrev_i32, "v_ashrrev_i32 %0, 1, %0")
BOTH(min_u32, "v_min_u32 %0, %0, %1")
BOTH(med3_i32, "v_med3_i32 %0, %0, %1, %2")
BOTH(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
BOTH(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
BOTH(sub_f32, "v_sub_f32 %0, %0, %1")
BOTH(fmac_f32, "v_fmac_f32 %0, %1, %2")
BOTH(min_f32, "v_min_f32 %0, %0, %1")
BOTH(add_f16, "v_add_f16 %0, %0, %1")
BOTH(fma_f16, "v_fma_f16 %0, %0, %1, %2")
BOTH(max_f16, "v_max_f16 %0, %0, %1")
BOTH(pk_add_f16, "v_pk_add_f16 %0, %0, %1")
BOTH(pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
BOTH(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
BOTH(pk_max_f16, "v_pk_max_f16 %0, %0, %1")
BOTH(add_u16, "v_add_u16 %0, %0, %1")
BOTH(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
BOTH(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
BOTH(cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
BOTH(dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
BOTH(dot4_i32_i8, "v_dot4_i32_i8 %0, %0, %1, %2")
BOTH(cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %1")
BOTH(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1")
BOTH(cmp_cnd, "v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
BOTH(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
BOTH(mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BOTH(readlane_free, "v_add_co_u32 %0, vcc, %0, %1")
BOTH(addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")

// 64-bit-pair opcodes (two registers per operand): v_pk_fma_f32 / v_pk_add_f32 / v_lshl_add_u64
#define INDEP64(name, ASMSTR)                                                                \
    __global__ __launch_bounds__(256) void name(int *out, Stamp *st, int a, int b)           \
    {                                                                                        \
        PROLOGUE                                                                             \
        double w[8]; double wa = (double)va, wb = (double)vb;                                \
        _Pragma("unroll") for (int i = 0; i < 8; i++) w[i] = __hiloint2double(v[2 * i + 1], v[2 * i]); \
        asm volatile("" : "+v"(wa), "+v"(wb));                                               \
        for (int it = 0; it < ITER; it++) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASMSTR : "+v"(w[i & 7]) : "v"(wa), "v"(wb)); \
        }                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; i++) v[i] = __double2loint(w[i]) + __double2hiint(w[i]); \
        EPILOGUE                                                                             \
    }
INDEP64(i_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
INDEP64(i_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
INDEP64(i_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
INDEP64(i_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
INDEP64(i_fma_f64, "v_fma_f64 %0, %0, %1, %2")
INDEP64(i_add_f64, "v_add_f64 %0, %0, %1")

// ---- the forward ACS step of k_turbo_siso (turbo.hip), inputs from registers
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s      as_v2s(uint32_t w) { return __builtin_bit_cast(v2s, w); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t neg_mask(v2s d)
{
    uint32_t m;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(as_u32(d)));
    return m;
}
__device__ __forceinline__ v2s bit_select(uint32_t m, v2s yes, v2s no) { return as_v2s((as_u32(yes) & m) | (as_u32(no) & ~m)); }
__device__ __forceinline__ uint32_t sign_bits(v2s n) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2u, n) >> 15); }
__device__ __forceinline__ void acs_step2(v2s (&pm)[8], v2s x, v2s y, uint32_t &acc)
{
    const v2s m0 = x >> 15, mx = m0 ^ (y >> 15), nmx = ~mx;
    const v2s uP = ((x + y) << 1) & nmx;
    const v2s uQ = ((x - y) << 1) & mx;
    const v2s c4 = (m0 & (v2s)(8)) - (v2s)(4);
    const v2s P2 = c4 & nmx, Q2 = c4 & mx;
    const v2s n0 = pm[1] - pm[0], n1 = pm[3] - pm[2], n2 = pm[5] - pm[4], n3 = pm[7] - pm[6];
    acc = (acc << 1) | sign_bits(n0);
    acc = (acc << 1) | sign_bits(n1);
    acc = (acc << 1) | sign_bits(n2);
    acc = (acc << 1) | sign_bits(n3);
    auto sel = [](v2s d, v2s yes, v2s no) { return bit_select(neg_mask(d), yes, no); };
    v2s nw[8];
    nw[0] = sel(n0 - P2, pm[1] + uP, pm[0] - uP);
    nw[4] = sel(n0 + P2, pm[1] - uP, pm[0] + uP);
    nw[1] = sel(n1 - Q2, pm[3] + uQ, pm[2] - uQ);
    nw[5] = sel(n1 + Q2, pm[3] - uQ, pm[2] + uQ);
    nw[2] = sel(n2 + Q2, pm[5] - uQ, pm[4] + uQ);
    nw[6] = sel(n2 - Q2, pm[5] + uQ, pm[4] - uQ);
    nw[3] = sel(n3 + P2, pm[7] - uP, pm[6] + uP);
    nw[7] = sel(n3 - P2, pm[7] + uP, pm[6] - uP);
#pragma unroll
    for (int s = 0; s < 8; s++) pm[s] = nw[s];
}
template <int R> __device__ __forceinline__ v2s byte_pair(uint32_t w0, uint32_t w1)
{
    return as_v2s(__builtin_amdgcn_perm(w1, w0, (uint32_t)(4 + R) << 24 | 0x0C0000u | (uint32_t)R << 8 | 0x0Cu)) >> 8;
}
constexpr int ACS_GROUPS = 4096; // groups of 8 trellis steps (two trellises per lane)

// ---- the same step in "SWAR" form: both trellises in the 16-bit fields of ONE 32-bit register, but walked with 32-bit adds and
// subtracts (2 cycles per wave64 instruction, see the table above) instead of v_pk_*_i16 (4 cycles).  A 32-bit add of two packed
// words is exact per field as long as (a) signed per-field operands are built arithmetically (hi * 65536 + lo as a 32-bit number,
// which is what a 32-bit subtraction of two packed NON-NEGATIVE words yields) and (b) no field leaves [0, 2^16).  (b) holds because
// only differences of path metrics matter and those stay below 3*512 + 3*508: every 16 steps all eight metrics are moved by the
// same amount so that state 0 sits at 2^14, the others then lie within +-3060 of it and drift by at most 508 per step.
// What stays at the 4-cycle rate: the two byte extractions, the sign masks (v_pk_ashrrev_i16) and one shift per decision bit.
__device__ __forceinline__ uint32_t pk_sra15(uint32_t d)
{
    uint32_t m;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(d));
    return m;
}
__device__ __forceinline__ uint32_t pk_srl7(uint32_t d) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2u, d) >> 7); }
// v_bitop3_b32 by hand: left to itself the compiler picks v_bfi_b32 / v_and_or_b32 for these, which issue at half the rate
__device__ __forceinline__ uint32_t bsel(uint32_t m, uint32_t yes, uint32_t no) { return __builtin_amdgcn_bitop3_b32(yes, no, m, 0xE4); }
__device__ __forceinline__ uint32_t push_bit15(uint32_t acc, uint32_t n, uint32_t G) { return __builtin_amdgcn_bitop3_b32(acc >> 1, n, G, 0x72); } // (acc >> 1) & ~G | ~n & G
// zx, zy: the step's two soft inputs, biased (x + 128), in the HIGH byte of each field, low bytes 0.  acc: decision bits, newest at bit 15
__device__ __forceinline__ void acs_step_swar(uint32_t (&v)[8], uint32_t zx, uint32_t zy, uint32_t &acc)
{
    constexpr uint32_t G = 0x80008000u, K4 = 0x00040004u, K8 = 0x00080008u, K512 = 0x02000200u;
    const uint32_t px = pk_sra15(zx);      // 0xFFFF where x >= 0
    const uint32_t mx = pk_sra15(zx ^ zy); // 0xFFFF where the signs differ
    const uint32_t X = pk_srl7(zx), Y = pk_srl7(zy); // 2x + 256, 2y + 256
    const uint32_t E  = X + bsel(mx, K512 - Y, Y);   // 512 + 2(x + y) or 512 + 2(x - y)
    const uint32_t uP = bsel(mx, K512, E) - K512;    // arithmetic form of -w*P
    const uint32_t uQ = bsel(mx, E, K512) - K512;    //                    -w*Q
    const uint32_t F  = __builtin_amdgcn_bitop3_b32(px, K8, K8, 0x0C);                    // x < 0 ? 8 : 0
    const uint32_t P2 = bsel(mx, K4, F) - K4, Q2 = bsel(mx, F, K4) - K4;
    const uint32_t n0 = (v[1] | G) - v[0], n1 = (v[3] | G) - v[2], n2 = (v[5] | G) - v[4], n3 = (v[7] | G) - v[6]; // 2^15 + difference
    acc = push_bit15(acc, n0, G); // decision = difference < 0 = bit 15 clear
    acc = push_bit15(acc, n1, G);
    acc = push_bit15(acc, n2, G);
    acc = push_bit15(acc, n3, G);
    auto sel = [](uint32_t d, uint32_t yes, uint32_t no) { return bsel(pk_sra15(d), no, yes); }; // bit 15 clear (d < 2^15) ? yes : no
    uint32_t nw[8];
    nw[0] = sel(n0 - P2, v[1] + uP, v[0] - uP);
    nw[4] = sel(n0 + P2, v[1] - uP, v[0] + uP);
    nw[1] = sel(n1 - Q2, v[3] + uQ, v[2] - uQ);
    nw[5] = sel(n1 + Q2, v[3] - uQ, v[2] + uQ);
    nw[2] = sel(n2 + Q2, v[5] - uQ, v[4] + uQ);
    nw[6] = sel(n2 - Q2, v[5] + uQ, v[4] - uQ);
    nw[3] = sel(n3 + P2, v[7] - uP, v[6] + uP);
    nw[7] = sel(n3 - P2, v[7] + uP, v[6] - uP);
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = nw[s];
}
template <int R> __device__ __forceinline__ uint32_t byte_hi(uint32_t w0, uint32_t w1) // (w0.byte R << 8, w1.byte R << 8)
{
    return __builtin_amdgcn_perm(w1, w0, (uint32_t)(4 + R) << 24 | 0x0C0000u | (uint32_t)R << 8 | 0x0Cu);
}

// IMPL 0: v_pk_*_i16 (turbo.hip as of round 3), 1: SWAR.  CHECK: fold every decision word (in the packed form's bit order) and the
// final metric differences into the output so that the two implementations can be compared bit for bit.
template <int IMPL, bool CHECK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_acs(int *out, Stamp *st, int a, int b)
{
    v2s      pm[8];
    uint32_t v[8];
#pragma unroll
    for (int s = 0; s < 8; s++) { pm[s] = (v2s)(0); v[s] = 0x40004000u; }
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)(a + i) * 2654435761u ^ ((blockIdx.x * 256 + threadIdx.x) * 0x9E3779B9u + b);
    uint32_t sink = 0;
    __syncthreads();
    const long long r0 = wall_clock64(), t0 = clock64();
    for (int g = 0; g < ACS_GROUPS; g++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w[i] = CHECK ? w[i] * 1664525u + 1013904223u + (uint32_t)g : w[i] + (uint32_t)g; // the check wants every byte value; the timing run one add per word
            asm volatile("" : "+v"(w[i]));
        }
        uint32_t acc_lo = 0, acc_hi = 0;
        if (IMPL == 0) {
            acs_step2(pm, byte_pair<0>(w[0], w[1]), byte_pair<0>(w[2], w[3]), acc_lo);
            acs_step2(pm, byte_pair<1>(w[0], w[1]), byte_pair<1>(w[2], w[3]), acc_lo);
            acs_step2(pm, byte_pair<2>(w[0], w[1]), byte_pair<2>(w[2], w[3]), acc_lo);
            acs_step2(pm, byte_pair<3>(w[0], w[1]), byte_pair<3>(w[2], w[3]), acc_lo);
            acs_step2(pm, byte_pair<0>(w[4], w[5]), byte_pair<0>(w[6], w[7]), acc_hi);
            acs_step2(pm, byte_pair<1>(w[4], w[5]), byte_pair<1>(w[6], w[7]), acc_hi);
            acs_step2(pm, byte_pair<2>(w[4], w[5]), byte_pair<2>(w[6], w[7]), acc_hi);
            acs_step2(pm, byte_pair<3>(w[4], w[5]), byte_pair<3>(w[6], w[7]), acc_hi);
        } else {
            uint32_t z[8];
#pragma unroll
            for (int i = 0; i < 8; i++) z[i] = w[i] ^ 0x80808080u; // bias: one xor per word and four steps
            acs_step_swar(v, byte_hi<0>(z[0], z[1]), byte_hi<0>(z[2], z[3]), acc_lo);
            acs_step_swar(v, byte_hi<1>(z[0], z[1]), byte_hi<1>(z[2], z[3]), acc_lo);
            acs_step_swar(v, byte_hi<2>(z[0], z[1]), byte_hi<2>(z[2], z[3]), acc_lo);
            acs_step_swar(v, byte_hi<3>(z[0], z[1]), byte_hi<3>(z[2], z[3]), acc_lo);
            acs_step_swar(v, byte_hi<0>(z[4], z[5]), byte_hi<0>(z[6], z[7]), acc_hi);
            acs_step_swar(v, byte_hi<1>(z[4], z[5]), byte_hi<1>(z[6], z[7]), acc_hi);
            acs_step_swar(v, byte_hi<2>(z[4], z[5]), byte_hi<2>(z[6], z[7]), acc_hi);
            acs_step_swar(v, byte_hi<3>(z[4], z[5]), byte_hi<3>(z[6], z[7]), acc_hi);
            if (g & 1) { // every 16 steps: state 0 back to 2^14 in both fields, everybody else by the same amount
                const uint32_t c = v[0] - 0x40004000u;
#pragma unroll
                for (int s = 1; s < 8; s++) v[s] -= c;
                v[0] = 0x40004000u;
            }
            if (CHECK) { // the packed form's bit order: first decision in bit 15 of its field
                acc_lo = __builtin_amdgcn_alignbit(__brev(acc_lo), __brev(acc_lo), 16);
                acc_hi = __builtin_amdgcn_alignbit(__brev(acc_hi), __brev(acc_hi), 16);
            }
        }
        const uint32_t d0 = __builtin_amdgcn_perm(acc_lo, acc_hi, 0x05040100u), d1 = __builtin_amdgcn_perm(acc_lo, acc_hi, 0x07060302u);
        sink = CHECK ? (sink * 31u + d0) * 31u + d1 : sink ^ (d0 + d1);
    }
    const long long t1 = clock64(), r1 = wall_clock64();
    uint32_t s = sink;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const uint32_t d = IMPL == 0 ? as_u32(pm[i] - pm[0]) : ((v[i] | 0x80008000u) - v[0]) ^ 0x80008000u; // per-field difference mod 2^16
        s = CHECK ? s * 131u + d : s + d;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};
}

struct Result { double cyc_med, cyc_max, mhz, ms; };
template <typename K> Result launch(K kern, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256; // a 256-thread workgroup puts one wavefront on each SIMD of its CU
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
    std::vector<double> c(h.size());
    double cs = 0, rs = 0;
    for (size_t i = 0; i < h.size(); i++) { c[i] = (double)h[i].cyc; cs += h[i].cyc; rs += h[i].real; }
    std::sort(c.begin(), c.end());
    (void)hipFree(d); (void)hipFree(st);
    return Result{c[c.size() / 2], c.back(), cs / rs * 100.0, ms}; // s_memrealtime ticks at 100 MHz
}

// cycles per instruction as the SIMD sees them: kernel wall time x measured clock / (instructions per wavefront x wavefronts per SIMD)
static double per_instr(const Result &r, int w, double n_instr) { return r.ms * 1e-3 * r.mhz * 1e6 / (n_instr * w); }
template <typename K> void row(const char *name, K ki, K kd)
{
    printf("%-20s", name);
    Result r8{};
    for (int w : {1, 2, 4, 8}) {
        r8 = launch(ki, w);
        printf(" %6.2f", per_instr(r8, w, (double)ITER * 16));
    }
    Result d1 = launch(kd, 1), d8 = launch(kd, 8);
    printf("   | dep: %6.2f %6.2f | clk %5.0f MHz, wall %7.3f ms at w=8\n", per_instr(d1, 1, (double)ITER * 16), per_instr(d8, 8, (double)ITER * 16), r8.mhz, r8.ms);
}
template <typename K> void row1(const char *name, K ki)
{
    printf("%-20s", name);
    Result r8{};
    for (int w : {1, 2, 4, 8}) {
        r8 = launch(ki, w);
        printf(" %6.2f", per_instr(r8, w, (double)ITER * 16));
    }
    printf("   |                      | clk %5.0f MHz, wall %7.3f ms at w=8\n", r8.mhz, r8.ms);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    printf("cycles a SIMD spends per wave64 instruction = kernel wall time x clock (s_memtime / s_memrealtime of the same run) / (instructions per wavefront x wavefronts per SIMD)\n");
    printf("%-20s %6s %6s %6s %6s   | dep:   w=1    w=8 |\n", "16 indep. chains", "w=1", "w=2", "w=4", "w=8");
#define ROW(stem) row(#stem, i_##stem, d_##stem)
    ROW(add_u32); ROW(and_b32); ROW(xor_b32); ROW(lshlrev_b32); ROW(mov_b32);
    ROW(pk_add_u16); ROW(pk_sub_i16); ROW(pk_ashrrev_i16); ROW(pk_lshlrev_b16); ROW(pk_lshrrev_b16); ROW(pk_max_i16); ROW(pk_min_i16); ROW(pk_mad_i16);
    ROW(bitop3_b32); ROW(bfi_b32); ROW(and_or_b32); ROW(or3_b32); ROW(lshl_or_b32); ROW(lshl_add_u32); ROW(add3_u32); ROW(perm_b32);
    ROW(bfe_u32); ROW(bfe_i32); ROW(alignbit_b32); ROW(max_i32); ROW(max3_i32); ROW(mad_i32_i24); ROW(mul_lo_u32);
    ROW(add_f32); ROW(mul_f32); ROW(fma_f32); ROW(max_f32); ROW(cndmask_b32); ROW(add_u32_sdwa); ROW(add_u32_dpp); ROW(sad_u8);
    ROW(cvt_f32_i32); ROW(rcp_f32); ROW(nop);
    ROW(sub_u32); ROW(subrev_u32); ROW(or_b32); ROW(not_b32); ROW(xnor_b32); ROW(lshrrev_b32); ROW(ashrrev_i32); ROW(min_u32); ROW(med3_i32);
    ROW(mul_u32_u24); ROW(mad_u32_u24); ROW(sub_f32); ROW(fmac_f32); ROW(min_f32); ROW(add_f16); ROW(fma_f16); ROW(max_f16); ROW(pk_add_f16); ROW(pk_mul_f16);
    ROW(pk_fma_f16); ROW(pk_max_f16); ROW(add_u16); ROW(cvt_f32_ubyte0); ROW(cvt_i32_f32); ROW(cvt_pk_i16_i32); ROW(dot2_i32_i16); ROW(dot4_i32_i8);
    ROW(cmp_lt_i32); ROW(cmp_lt_f32); ROW(cmp_cnd); ROW(cndmask_sgpr); ROW(mov_dpp); ROW(readlane_free); ROW(addc_co);
    row1("pk_fma_f32", i_pk_fma_f32); row1("pk_add_f32", i_pk_add_f32); row1("pk_mul_f32", i_pk_mul_f32); row1("lshl_add_u64", i_lshl_add_u64);
    row1("fma_f64", i_fma_f64); row1("add_f64", i_add_f64);

    printf("\nforward ACS of k_turbo_siso, inputs in registers (8 trellis steps x 2 trellises per group); cycles per SIMD per trellis step and resident wavefront = wall x clock / (steps x w):\n");
    for (int impl = 0; impl < 2; impl++)
        for (int w : {1, 2, 4, 8}) {
            Result r = impl ? launch(k_acs<1, false>, w) : launch(k_acs<0, false>, w);
            printf("  %-14s w=%d: %7.1f cycles per step  (clk %.0f MHz; wall %.3f ms)\n", impl ? "SWAR (32-bit)" : "v_pk_*_i16", w, per_instr(r, w, (double)ACS_GROUPS * 8), r.mhz, r.ms);
        }
    { // bit-for-bit comparison of the two forms: every decision word of 4096 x 8 steps and the final metric differences, 2 x 65536 trellises
        const int blocks = 256, threads = 256;
        int  *d0, *d1;
        Stamp *st;
        CHECK(hipMalloc(&d0, sizeof(int) * blocks * threads)); CHECK(hipMalloc(&d1, sizeof(int) * blocks * threads)); CHECK(hipMalloc(&st, sizeof(Stamp) * blocks * 4));
        hipLaunchKernelGGL((k_acs<0, true>), dim3(blocks), dim3(threads), 0, 0, d0, st, 3, 5);
        hipLaunchKernelGGL((k_acs<1, true>), dim3(blocks), dim3(threads), 0, 0, d1, st, 3, 5);
        CHECK(hipDeviceSynchronize());
        std::vector<int> h0(blocks * threads), h1(blocks * threads);
        CHECK(hipMemcpy(h0.data(), d0, sizeof(int) * h0.size(), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h1.data(), d1, sizeof(int) * h1.size(), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); i++) bad += h0[i] != h1[i];
        printf("  self-check: %zu of %zu lanes differ between the two forms (decision words of %d steps + final metric differences, hashed)\n", bad, h0.size(), ACS_GROUPS * 8);
    }
    return 0;
}
