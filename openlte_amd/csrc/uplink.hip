// PUSCH receive chain on gfx950 (BASELINE config 5): restates liblte_phy_pusch_channel_decode
// (liblte/src/liblte_phy.cc:2801-2935) -> ulsch_channel_decode (:12363-12501) for a batch of allocations
// (one per scheduled UE) inside the envelope the reference itself handles: single antenna, one codeword,
// no RI/ACK/CQI multiplexing, one code block per transport block.
//
//   k_pusch_demod : one workgroup per allocation.
//       DMRS least-squares estimate on symbols 3 and 10 and magnitude/phase interpolation over the 12
//       data symbols (get_ulsch_ce :13718-13790), one-tap equaliser (pre_decoder_and_matched_filter_ul
//       :6708-6736), transform pre-decoding = 12 unnormalised backward DFTs of size M = 12*N_prb times
//       sqrt(M) (:6627-6660; the reference's scaling is reproduced as is), modulation de-mapping, descrambling
//       (:2893-2898) and the channel de-interleaver, which without control bits is the transpose
//       g[(k*12 + s)*Q_m + q] = h[(s*M + k)*Q_m + q] (:12108-12225).  M is any size FFTW would plan
//       (N_prb divisible by 2, 3 or 5 -> radices 4, 2, 3, 5 and whatever prime is left), done as a
//       mixed-radix Stockham transform through LDS.
//   then the downlink's turbo stage (turbo.hip) with the UL-SCH rate-matching rule (N_cb = K_w).
//
// The DFT sizes are not powers of two and FFTW's operation order is unspecified, so like the downlink FFT
// this stage is tolerance-checked; everything from the int8 soft bits on is integer-exact.
#include <algorithm>
#include <map>
#include <tuple>

#include <type_traits>

#include <cstring>
#include "ctx.hpp"
#include "phy_dev.hpp"
#include "lte_tables.h"

namespace {

constexpr int N_SC_MAX = 1200;

constexpr uint32_t EST_ROWS = 9; // k_pusch_demod: LDS floats kept per subcarrier of the allocation

struct PuschDesc {
    uint32_t subfr, cell;   // of the allocation's unit
    uint32_t dmrs_off;      // float offset of dmrs_0_re | dmrs_0_im | dmrs_1_re | dmrs_1_im (M each) in the DMRS pool
};

// The transform pre-decoding of an allocation of N_prb resource blocks, M = 12 N_prb: the radices of its Stockham passes in the order they
// run (9, 3, 5, 8, 4, 2, then the primes that are left) with each pass's sizes, sqrt(M) as the reference forms it (liblte_phy.cc:6644: integer
// argument -> double sqrt, stored to float) and the reciprocals the index arithmetic divides by.  One row per N_prb, made once per context on
// the host (pusch_shapes below): forming them in the kernel -- a double-precision square root, a dozen 32-bit divisions of workgroup-uniform
// values that have no scalar instruction and so run on the vector unit, the radix search -- was 11 % of its vector instructions at 6 PRB.
constexpr uint32_t MAX_PASSES = 6;
struct DftPass {
    uint32_t R, Ns, nb, tstride; // radix, product of the radices before it, M / R, M / (Ns R)
    float    r_nb, r_ns;         // reciprocals for quot() below
};
struct PuschShape {
    float    sqrt_M, r_M;
    uint32_t n_pass, pad;
    DftPass  pass[MAX_PASSES];
};
static_assert(sizeof(PuschShape) == 160, "one row = 40 words");
constexpr uint32_t N_SHAPES = 111; // N_prb 0 .. 110

// floor(n / d) for n < 2^20 as the truncated float product n * r, r = the float next above or equal to 1 / d (recip_up on the host):
// n r >= n / d, so a multiple of d never lands below its quotient (the quotient is representable and rounding is monotonic), and the excess
// n / d * 2^-22 stays under the 1 / d that separates n / d from the next integer while n < 2^22.  Three full-rate instructions where
// v_mul_hi_u32 issues at a quarter of the rate.
__device__ __forceinline__ uint32_t quot(uint32_t n, float r) { return (uint32_t)((float)n * r); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmuli(float2 a) { return make_float2(-a.y, a.x); } // a * i

// R-point backward DFTs in registers: v[q] <- sum_r v[r] exp(+2*pi*i * r*q/R).
__device__ __forceinline__ void dft3(float2 &a, float2 &b, float2 &c)
{
    const float  h3 = 0.86602540378443864676f; // sin(2 pi / 3)
    const float2 s = cadd(b, c), d = csub(b, c);
    const float2 m = make_float2(a.x - 0.5f * s.x, a.y - 0.5f * s.y), n = make_float2(-h3 * d.y, h3 * d.x);
    a = cadd(a, s);
    b = cadd(m, n);
    c = csub(m, n);
}
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 s02 = cadd(a, c), d02 = csub(a, c), s13 = cadd(b, d), d13 = cmuli(csub(b, d));
    a = cadd(s02, s13);
    b = cadd(d02, d13);
    c = csub(s02, s13);
    d = csub(d02, d13);
}
template <uint32_t R> __device__ __forceinline__ void butterfly(float2 (&v)[R]);
template <> __device__ __forceinline__ void butterfly<2>(float2 (&v)[2])
{
    const float2 a = v[0];
    v[0] = cadd(a, v[1]);
    v[1] = csub(a, v[1]);
}
template <> __device__ __forceinline__ void butterfly<3>(float2 (&v)[3]) { dft3(v[0], v[1], v[2]); }
template <> __device__ __forceinline__ void butterfly<4>(float2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void butterfly<5>(float2 (&v)[5])
{
    const float  c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f; // cos(2 pi / 5), cos(4 pi / 5)
    const float  s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;  // sin(2 pi / 5), sin(4 pi / 5)
    const float2 a1 = cadd(v[1], v[4]), d1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), d2 = csub(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 n1 = cmuli(make_float2(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y));
    const float2 n2 = cmuli(make_float2(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y));
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
    v[4] = csub(m1, n1);
}
template <> __device__ __forceinline__ void butterfly<8>(float2 (&v)[8])
{
    // even outputs from v[r] + v[r+4], odd ones from (v[r] - v[r+4]) * exp(+2*pi*i * r/8), a 4-point transform of each
    const float r2 = 0.70710678118654752440f;
    float2      a[4], b[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) { a[r] = cadd(v[r], v[r + 4]); b[r] = csub(v[r], v[r + 4]); }
    b[1] = make_float2(r2 * (b[1].x - b[1].y), r2 * (b[1].x + b[1].y));
    b[2] = cmuli(b[2]);
    b[3] = make_float2(-r2 * (b[3].x + b[3].y), r2 * (b[3].x - b[3].y));
    dft4(a[0], a[1], a[2], a[3]);
    dft4(b[0], b[1], b[2], b[3]);
#pragma unroll
    for (uint32_t p = 0; p < 4; p++) { v[2 * p] = a[p]; v[2 * p + 1] = b[p]; }
}
template <> __device__ __forceinline__ void butterfly<9>(float2 (&v)[9])
{
    // r = r1 + 3 r2, q = 3 q1 + q2: 3-point transforms over r2, the twiddles exp(+2*pi*i * r1*q2/9), 3-point transforms over r1
    const float2 w1 = make_float2(0.76604444311897803520f, 0.64278760968653932632f);  // exp(2 pi i / 9)
    const float2 w2 = make_float2(0.17364817766693034885f, 0.98480775301220805937f);  // exp(4 pi i / 9)
    const float2 w4 = make_float2(-0.93969262078590838405f, 0.34202014332566873304f); // exp(8 pi i / 9)
    dft3(v[0], v[3], v[6]); // r1 = 0: results indexed by q2 in v[0], v[3], v[6]
    dft3(v[1], v[4], v[7]); // r1 = 1
    dft3(v[2], v[5], v[8]); // r1 = 2
    v[4] = cmul(v[4], w1); v[7] = cmul(v[7], w2); // r1 = 1: q2 = 1, 2
    v[5] = cmul(v[5], w2); v[8] = cmul(v[8], w4); // r1 = 2: q2 = 1, 2
    dft3(v[0], v[1], v[2]); // q2 = 0: outputs q = 0, 3, 6
    dft3(v[3], v[4], v[5]); // q2 = 1: outputs q = 1, 4, 7
    dft3(v[6], v[7], v[8]); // q2 = 2: outputs q = 2, 5, 8
    const float2 t1 = v[1], t2 = v[2], t3 = v[3], t5 = v[5], t6 = v[6], t7 = v[7];
    v[1] = t3; v[2] = t6; v[3] = t1; v[5] = t7; v[6] = t2; v[7] = t5; // v[3*q2 + q1] -> v[3*q1 + q2]
}

// One Stockham pass of radix R in {2, 3, 4, 5, 8, 9} over S symbols of M points each (symbol s at in + s*M_max), sign +1 (backward
// transform), one thread per butterfly:
//   out[(j-k)*R + k + q*Ns] = sum_r in[j + r*M/R] * w^(r*k) * exp(+2*pi*i * r*q/R),  k = j mod Ns,  w = exp(+2*pi*i / (Ns*R)),
// the twiddle w^(r*k) being entry r*k*M/(Ns*R) (< M) of the allocation's table tw[t] = exp(+2*pi*i*t/M).
template <uint32_t R, uint32_t THREADS>
__device__ __forceinline__ void dft_pass_r(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw,
                                           uint32_t S, uint32_t M_max, const DftPass ps)
{
    const uint32_t nb = ps.nb, Ns = ps.Ns, tstride = ps.tstride;
    for (uint32_t o = threadIdx.x; o < S * nb; o += THREADS) {
        const uint32_t sy = quot(o, ps.r_nb), j = o - __umul24(sy, nb), k = Ns > 1 ? j - __umul24(quot(j, ps.r_ns), Ns) : 0;
        const float2  *x = in + __umul24(sy, M_max) + j;
        float2         v[R];
#pragma unroll
        for (uint32_t r = 0; r < R; r++) v[r] = x[r * nb];
        if (Ns > 1) { // uniform; the first pass has k = 0 throughout
            const uint32_t t = __umul24(k, tstride);
#pragma unroll
            for (uint32_t r = 1; r < R; r++) v[r] = cmul(v[r], tw[r * t]);
        }
        butterfly<R>(v);
        float2 *y = out + __umul24(sy, M_max) + __umul24(j - k, R) + k;
#pragma unroll
        for (uint32_t q = 0; q < R; q++) y[q * Ns] = v[q];
    }
}

// The same pass for any other radix (the prime FFTW would be left with when N_prb has a factor >= 7), one thread per output:
// step = (k + q*Ns) * M/(Ns*R) < M, twiddle index r*step mod M.
template <uint32_t THREADS>
__device__ __forceinline__ void dft_pass(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw,
                                         uint32_t S, uint32_t M, float r_M, uint32_t M_max, const DftPass ps)
{
    const uint32_t R = ps.R, nb = ps.nb, Ns = ps.Ns, tstride = ps.tstride;
    for (uint32_t o = threadIdx.x; o < S * M; o += THREADS) {
        const uint32_t sy = quot(o, r_M), oo = o - sy * M;
        const uint32_t q = quot(oo, ps.r_nb), j = oo - q * nb, k = j - quot(j, ps.r_ns) * Ns, step = (k + q * Ns) * tstride;
        const float2  *x = in + sy * M_max + j;
        float    ar = 0.0f, ai = 0.0f;
        uint32_t t  = 0; // r*step mod M
        for (uint32_t r = 0; r < R; r++) {
            const float2 w = tw[t], v = x[r * nb];
            ar += v.x * w.x - v.y * w.y;
            ai += v.x * w.y + v.y * w.x;
            t += step;
            if (t >= M) t -= M;
        }
        out[sy * M_max + (j - k) * R + k + q * Ns] = make_float2(ar, ai);
    }
}

#ifndef PUSCH_THREADS
#define PUSCH_THREADS 256
#endif
#ifndef WPE
#define WPE 6
#endif
// THREADS = the workgroup's width as a compile-time constant (64 / 128 / 192 / 256): every phase is a loop `o += THREADS` -- with blockDim.x the
// stride was re-read from the dispatch packet in every iteration of the de-mapper's loop (a store in the body could alias it as far as the
// compiler knows) -- and the index products are 24-bit multiplies (v_mul_lo_u32 / v_mad_u64_u32 issue at a quarter of the rate)
template <uint32_t THREADS>
__attribute__((amdgpu_waves_per_eu(WPE, 8)))
__global__ __launch_bounds__(THREADS) void k_pusch_demod(const float *__restrict__ subframes, uint32_t sf_stride,
                                                     const mi_lte_pdsch_alloc *__restrict__ allocs, const PuschDesc *__restrict__ desc,
                                                     const float *__restrict__ dmrs_pool, GoldTables gt, int8_t *__restrict__ e_base,
                                                     const uint32_t *__restrict__ e_off, uint32_t *__restrict__ e_len, uint32_t M_max,
                                                     uint32_t S_par, float r_S_par, const PuschShape *__restrict__ shapes)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // LDS: est[9][M_max] (mag0 mag1 dmag | unit vectors of ang0, ang1, dang) | tw[M_max] | buf A[S_par][M_max] | buf B[S_par][M_max] (float2) | scrambling words
    float    *est  = sm;
    float2   *tw   = reinterpret_cast<float2 *>(sm + EST_ROWS * (size_t)M_max);
    float2   *bufA = tw + M_max, *bufB = bufA + (size_t)S_par * M_max;
    uint32_t *cw   = reinterpret_cast<uint32_t *>(bufB + (size_t)S_par * M_max);

    const uint32_t a_idx = blockIdx.x;
    const mi_lte_pdsch_alloc &al = allocs[a_idx];
    const PuschDesc           ds = desc[a_idx];
    const uint32_t N_prb = al.N_prb, M = 12 * N_prb;
    const PuschShape &sh = shapes[N_prb]; // (uniform: scalar loads)
    const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
    const uint32_t N_bits = 12 * M * Qm;
    const float   *rx_re = subframes + (size_t)al.unit * sf_stride, *rx_im = rx_re + 16 * N_SC_MAX;
    if (threadIdx.x == 0) e_len[a_idx] = N_bits;

    // scrambling sequence (c_init per liblte_phy.cc:2893), one word of slack for the 2-word window below
    const uint32_t c_init = (al.rnti << 14) | (0u << 13) | (ds.subfr << 9) | ds.cell, n_words = (N_bits + 31) / 32;
    for (uint32_t w = threadIdx.x; w <= n_words; w += THREADS) cw[w] = gold_word(gt, c_init, w);

    // ---- DMRS estimates and their interpolation slopes (get_ulsch_ce, liblte_phy.cc:13745-13768); DFT twiddles.
    // The reference interpolates in polar form, h(s) = (mag_b + n f_mag) exp(i (ang_b + n f_ang)), n = -3..3 around DMRS symbol b:
    // what is kept per subcarrier is exp(i ang_b) = t_b / |t_b| and exp(i f_ang), so the 12 data symbols cost complex products,
    // not 12 sin/cos pairs.  Work is spread as (subcarrier, task): task 0/1 = DMRS symbol 0/1, task 2 = the twiddle.
    const float *d_pool = dmrs_pool + ds.dmrs_off; // dmrs_0_re | dmrs_0_im | dmrs_1_re | dmrs_1_im (M each)
    for (uint32_t o = threadIdx.x; o < 3 * M; o += THREADS) {
        const uint32_t task = o >= 2 * M ? 2 : o >= M ? 1 : 0, i = o - task * M;
        if (task == 2) {
            float sn, cs;
            sincospif(2.0f * (float)i / (float)M, &sn, &cs);
            tw[i] = make_float2(cs, sn);
        } else {
            const uint32_t rb = __umul24(i, 10923u) >> 17; // i / 12 for i < 1536
            const uint32_t sc = al.prb[task][rb] * 12 + (i - 12 * rb), L = task ? 10 : 3;
            const float    cr = rx_re[L * N_SC_MAX + sc], ci = rx_im[L * N_SC_MAX + sc];
            const float    dr = d_pool[2 * task * M + i], di = d_pool[(2 * task + 1) * M + i];
            const float    t_re = cr * dr + ci * di, t_im = ci * dr - cr * di;
            const float    mag = sqrtf(t_re * t_re + t_im * t_im);
            est[task * M_max + i] = mag;
            est[(3 + 2 * task) * M_max + i] = mag > 0.0f ? t_re / mag : 1.0f; // atan2f(0, 0) = 0
            est[(4 + 2 * task) * M_max + i] = mag > 0.0f ? t_im / mag : 0.0f;
            if (task == 0) est[8 * M_max + i] = atan2f(t_im, t_re);
            else           est[2 * M_max + i] = atan2f(t_im, t_re); // parked in the rows the second step overwrites
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < M; i += THREADS) {
        const float f_mag = (est[M_max + i] - est[i]) / 7;
        float       f_ang = est[2 * M_max + i] - est[8 * M_max + i];
        if ((double)f_ang >= M_PI) f_ang = (float)((double)f_ang - 2 * M_PI); // float compared / corrected in double (:13758-13764)
        else if ((double)f_ang <= -M_PI) f_ang = (float)((double)f_ang + 2 * M_PI);
        f_ang /= 7;
        float sn, cs;
        sincosf(f_ang, &sn, &cs);
        est[2 * M_max + i] = f_mag; est[7 * M_max + i] = cs; est[8 * M_max + i] = sn;
    }
    __syncthreads();

    const float sqrt_M = sh.sqrt_M; // liblte_phy.cc:6644 (integer argument -> double sqrt, stored to float: formed on the host)
    int8_t     *e      = e_base + (size_t)e_off[a_idx] * 64; // 64-byte units

    for (uint32_t s0 = 0; s0 < 12; s0 += S_par) { // S_par data symbols at a time (all 12 when they fit in LDS)
        const uint32_t S = S_par; // (12, 6, 3, 2 or 1: always a divisor of 12)
        // ---- channel estimate of each symbol and the one-tap equaliser.  One item per (subcarrier, slot): the slot's DMRS
        // estimate, its unit vector and the powers of exp(i f_ang) are fetched / formed once and serve the slot's (up to) six data
        // symbols; S is 12 (both slots) or divides 6 (part of one slot).
        const uint32_t slot0 = s0 >= 6, n_slots = S == 12 ? 2 : 1, sp0 = s0 - 6 * slot0, sp1 = sp0 + (S == 12 ? 6 : S);
        for (uint32_t o = threadIdx.x; o < n_slots * M; o += THREADS) {
            const uint32_t b = slot0 + (o >= M), i = o - (o >= M ? M : 0);
            const float    mag = est[b * M_max + i], f_mag = est[2 * M_max + i];
            const float2   u = make_float2(est[(3 + 2 * b) * M_max + i], est[(4 + 2 * b) * M_max + i]);
            const float2   f1 = make_float2(est[7 * M_max + i], est[8 * M_max + i]), f2 = cmul(f1, f1), f3 = cmul(f2, f1);
            const uint32_t rb = __umul24(i, 10923u) >> 17; // i / 12 for i < 1536
            const uint32_t sc = al.prb[b][rb] * 12 + (i - 12 * rb);
            const float   *z_re = rx_re + 7 * b * N_SC_MAX + sc, *z_im = rx_im + 7 * b * N_SC_MAX + sc;
            float2        *dst = bufA + ((int)(6 * b) - (int)s0) * (int)M_max + (int)i; // symbol s = 6 b + sp goes to row s - s0
#pragma unroll
            for (uint32_t sp = 0; sp < 6; sp++) {
                if (sp < sp0 || sp >= sp1) continue; // uniform
                // liblte_phy.cc:13770-13780: symbols 0-2 / 3-5 of a slot lie -3..-1 / +1..+3 steps from its DMRS symbol (symbol 3 of 7)
                const int    n = sp < 3 ? (int)sp - 3 : (int)sp - 2;
                const uint32_t l = sp < 3 ? sp : sp + 1; // position in the slot, skipping the DMRS symbol
                const float2 fa = n == 1 || n == -1 ? f1 : n == 2 || n == -2 ? f2 : f3;
                const float2 f = make_float2(fa.x, n < 0 ? -fa.y : fa.y);
                const float  cm = mag + (float)n * f_mag;
                const float2 ph = cmul(u, f);
                const float  h_re = cm * ph.x, h_im = cm * ph.y;
                const float  zr = z_re[l * N_SC_MAX], zi = z_im[l * N_SC_MAX];
                const float  hn = 1.0f / (h_re * h_re + h_im * h_im);
                dst[sp * M_max] = make_float2((zr * h_re + zi * h_im) * hn, (zi * h_re - zr * h_im) * hn);
            }
        }
        __syncthreads();
        // ---- transform pre-decoding: M-point backward DFTs, radices 9, 3, 5, 8, 4, 2, then the remaining prime.  Odd radices go
        // first: the first pass writes its R outputs R float2 apart across lanes, which is conflict-free in LDS only for odd R
        // (M = 12 N_prb always has a factor 3 to start with).
        float2 *src = bufA, *dst = bufB;
        for (uint32_t p = 0; p < sh.n_pass; p++) { // uniform over the workgroup
            const DftPass ps = sh.pass[p];
            switch (ps.R) {
            case 9:  dft_pass_r<9, THREADS>(src, dst, tw, S, M_max, ps); break;
            case 3:  dft_pass_r<3, THREADS>(src, dst, tw, S, M_max, ps); break;
            case 5:  dft_pass_r<5, THREADS>(src, dst, tw, S, M_max, ps); break;
            case 8:  dft_pass_r<8, THREADS>(src, dst, tw, S, M_max, ps); break;
            case 4:  dft_pass_r<4, THREADS>(src, dst, tw, S, M_max, ps); break;
            case 2:  dft_pass_r<2, THREADS>(src, dst, tw, S, M_max, ps); break;
            default: dft_pass<THREADS>(src, dst, tw, S, M, sh.r_M, M_max, ps); break;
            }
            __syncthreads();
            float2 *t = src; src = dst; dst = t;
        }
        // ---- de-map, descramble, de-interleave (transpose): soft bit q of symbol k goes to (k*12 + s)*Q_m + q.  Instantiated per modulation
        // (uniform over the workgroup): no modulation branches and no re-read of the allocation descriptor per element
        auto demap_all = [&](auto modc) {
        constexpr uint32_t MOD = decltype(modc)::value, QM = MOD == 3 ? 6 : MOD == 2 ? 4 : MOD == 1 ? 2 : 1;
        for (uint32_t o = threadIdx.x; o < S * M; o += THREADS) {
            // (all twelve symbols side by side is the common case: o / 12 for o < 2^15 as one 24-bit multiply and a shift)
            const uint32_t k = S == 12 ? __umul24(o, 43691u) >> 19 : quot(o, r_S_par), sy = o - __umul24(k, S), s = s0 + sy; // neighbouring threads write neighbouring bytes of e
            const float2   x = src[__umul24(sy, M_max) + k];
            int8_t         b[6] = {0, 0, 0, 0, 0, 0};
            demap_symbol(sqrt_M * x.x, sqrt_M * x.y, MOD, b);
            const uint32_t n0 = (__umul24(s, M) + k) * QM, w = n0 >> 5, sh = n0 & 31;
            const uint32_t c  = __builtin_amdgcn_alignbit(cw[w + 1], cw[w], sh);
            int8_t        *ob = e + (__umul24(k, 12u) + s) * QM; // (32-bit offset: an allocation's soft bits are at most 12 * 1320 * 6 bytes)
            // descrambled soft bits q, q + 1 as one 16-bit word (the Q_m bytes of a symbol start on an even address)
            auto pair = [&](uint32_t q) -> uint32_t {
                const int lo = ((c >> q) & 1u) ? -b[q] : b[q], hi = ((c >> (q + 1)) & 1u) ? -b[q + 1] : b[q + 1];
                return (uint32_t)(uint8_t)lo | (uint32_t)(uint8_t)hi << 8;
            };
            if (QM == 2) *reinterpret_cast<uint16_t *>(ob) = (uint16_t)pair(0);
            else if (QM == 4) *reinterpret_cast<uint32_t *>(ob) = pair(0) | pair(2) << 16;
            else if (QM == 6) {
                *reinterpret_cast<uint16_t *>(ob)     = (uint16_t)pair(0);
                *reinterpret_cast<uint16_t *>(ob + 2) = (uint16_t)pair(2);
                *reinterpret_cast<uint16_t *>(ob + 4) = (uint16_t)pair(4);
            } else ob[0] = (c & 1u) ? (int8_t)-b[0] : b[0];
        }
        };
        switch (al.mod_type) {
        case 0:  demap_all(std::integral_constant<uint32_t, 0>{}); break;
        case 1:  demap_all(std::integral_constant<uint32_t, 1>{}); break;
        case 2:  demap_all(std::integral_constant<uint32_t, 2>{}); break;
        default: demap_all(std::integral_constant<uint32_t, 3>{}); break;
        }
        __syncthreads();
    }
}

// ---- PUCCH formats 1 / 1a / 1b (liblte_phy_pucch_format_1_1a_1b_channel_decode, liblte_phy.cc:2961-3146) -----------------------
// One wavefront per PUCCH resource: the resource's PRB pair (N_1_p in slot 0, N_rb_ul - N_1_p - 1 in slot 1), a channel estimate
// per slot from its three reference symbols (get_ulcch_ce :13799-13868: mean magnitude, the FIRST symbol's phase), one-tap
// equalisation, and the coherent sum over 2 x 4 x 12 elements of z / (s_ns w(i) r_u,v(n)) -- summed in the reference's order by one
// lane, divided by 95 as the reference does (its loop index after the last element, not the count) -- then the reference's decision.
// The sequences are the caller's (what liblte_phy_ul_init left in LIBLTE_PHY_STRUCT): per resource 352 floats, see mi_lte.h.
struct PucchRes { uint32_t unit, format, N_1_p; };
constexpr uint32_t PUCCH_TAB_FLOATS = 4 * 36 + 2 * 96 + 16;

__device__ __forceinline__ float pucch_soft(float rx_re, float rx_im, float exp_re, float exp_im)
{
    const float d_re = rx_re - exp_re, d_im = rx_im - exp_im;
    float dist = sqrtf(d_re * d_re + d_im * d_im); // (C++ overload resolution in the reference: sqrt(float) is the float function)
    const float cap = 1.0f - (1.0f / 120);
    if (dist >= cap) dist = cap;
    return 1.0f - dist;
}

__global__ __launch_bounds__(64) void k_pucch_decode(const float *__restrict__ subframes, uint32_t sf_stride, uint32_t N_rb_ul, uint32_t N_ant,
                                                     const PucchRes *__restrict__ res, const float *__restrict__ tabs, uint8_t *__restrict__ out /*[n][4]*/)
{
    __shared__ float c_re[2][12], c_im[2][12], t_re[96], t_im[96];
    const uint32_t r = blockIdx.x, ln = threadIdx.x;
    const PucchRes pr = res[r];
    const float *base = subframes + (size_t)pr.unit * sf_stride, *y_re = base, *y_im = base + 16 * N_SC_MAX;
    const float *tb = tabs + (size_t)r * PUCCH_TAB_FLOATS;
    const float *dm_re[2] = {tb, tb + 72}, *dm_im[2] = {tb + 36, tb + 108};
    const float *ruv_re = tb + 144, *ruv_im = tb + 240, *sw_re = tb + 336, *sw_im = tb + 344;
    const uint32_t prb[2] = {pr.N_1_p, N_rb_ul - pr.N_1_p - 1};
    if (ln < 24) { // (slot, sub-carrier): channel estimate from the three reference symbols 2, 3, 4 (9, 10, 11)
        const uint32_t s = ln / 12, j = ln % 12;
        float ave_mag = 0, ang0 = 0;
        for (uint32_t i = 0; i < 3; i++) {
            const uint32_t L = 7 * s + 2 + i, k = prb[s] * 12 + j, idx = i * 12 + j;
            const float cr = y_re[L * N_SC_MAX + k], ci = y_im[L * N_SC_MAX + k], dr = dm_re[s][idx], di = dm_im[s][idx];
            const float denom = dr * dr + di * di;
            const float tr = (1 / denom) * (cr * dr + ci * di), ti = (1 / denom) * (ci * dr - cr * di);
            ave_mag += sqrtf(tr * tr + ti * ti) / 3; // (:13842; float sqrt / int)
            if (i == 0) ang0 = atan2f(ti, tr);
        }
        c_re[s][j] = ave_mag * cosf(ang0);
        c_im[s][j] = ave_mag * sinf(ang0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t idx = ln; idx < 96; idx += 64) { // element (slot m', data symbol i, sub-carrier j)
        const uint32_t mp = idx / 48, i = (idx % 48) / 12, j = idx % 12, L = 7 * mp + (i < 2 ? i : i + 3), k = prb[mp] * 12 + j;
        const float zr = y_re[L * N_SC_MAX + k], zi = y_im[L * N_SC_MAX + k], hr = c_re[mp][j], hi = c_im[mp][j];
        const float hn = hr * hr + hi * hi;
        const float xr = (zr * hr + zi * hi) / hn, xi = (zi * hr - zr * hi) / hn; // pre_decoder_and_matched_filter_ul (:6727-6732)
        const float swr = sw_re[mp * 4 + i], swi = sw_im[mp * 4 + i], rr = ruv_re[idx], ri = ruv_im[idx];
        const float pr_ = swr * rr - swi * ri, pi_ = swr * ri + swi * rr, denom = pr_ * pr_ + pi_ * pi_;
        t_re[idx] = (1 / denom) * (xr * pr_ + xi * pi_);
        t_im[idx] = (1 / denom) * (-xr * pi_ + xi * pr_);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (ln == 0) {
        float d_re = 0, d_im = 0;
        for (uint32_t idx = 0; idx < 96; idx++) { d_re += t_re[idx]; d_im += t_im[idx]; }
        d_re /= 95u; // the reference divides by its last index (:3097-3098)
        d_im /= 95u;
        d_re *= sqrt((double)N_ant);
        d_im *= sqrt((double)N_ant);
        uint8_t b0 = 0, b1 = 0, nb = 1;
        float   sd;
        if (pr.format <= 1) {
            if (d_re < 0) { sd = pucch_soft(d_re, d_im, -1, 0); b0 = 1; }
            else          { sd = pucch_soft(d_re, d_im, 1, 0); b0 = 0; }
        } else {
            const float ang = ref_atan2f(d_im, d_re); // (decisions on the angle: the host libm's rounding, phy_dev.hpp)
            nb = 2;
            if (ang >= M_PI / 4 && ang < 3 * M_PI / 4)        { sd = pucch_soft(d_re, d_im, 0, 1); b0 = 1; b1 = 0; }
            else if (ang >= -M_PI / 4 && ang < M_PI / 4)      { sd = pucch_soft(d_re, d_im, 1, 0); b0 = 0; b1 = 0; }
            else if (ang >= -3 * M_PI / 4 && ang < -M_PI / 4) { sd = pucch_soft(d_re, d_im, 0, -1); b0 = 0; b1 = 1; }
            else                                              { sd = pucch_soft(d_re, d_im, -1, 0); b0 = 1; b1 = 1; }
        }
        out[4 * r] = b0; out[4 * r + 1] = b1; out[4 * r + 2] = nb; out[4 * r + 3] = sd > 0.5f ? 0 : 1; // LIBLTE_SUCCESS / LIBLTE_ERROR_INVALID_INPUTS
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host side: plans

struct mi_lte_pusch_plan {
    mi_lte_dl_cfg cfg;
    uint32_t      n_alloc = 0, out_stride = 0, M_max = 0, words_max = 0;
    size_t        e_bytes = 0;
    mi_lte_pdsch_alloc *d_allocs = nullptr;
    PuschDesc          *d_desc   = nullptr;
    float              *d_dmrs   = nullptr;
    uint32_t *d_e_off = nullptr, *d_e_len = nullptr, *d_cb_alloc = nullptr;
    int8_t   *d_e = nullptr;
    struct Group { uint32_t K, n_cb, cb_base, e_max; };
    std::vector<Group>    groups;
    std::vector<uint32_t> h_e_off, h_e_len;
    MiMultiCache          multi; // the merged decode's device tables for `groups` (turbo.hip: mi_turbo_ref_multi)
};

// the float next above or equal to 1 / d (see quot)
static float recip_up(uint32_t d)
{
    const double rd = 1.0 / (double)d;
    float        r  = (float)rd;
    if ((double)r < rd) r = nextafterf(r, 2.0f);
    return r;
}

// One PuschShape per N_prb, in HBM for the life of the context
static int pusch_shapes(mi_lte_ctx *ctx)
{
    if (ctx->d_pusch_shapes) return MI_LTE_OK;
    std::vector<PuschShape> tab(N_SHAPES);
    memset(tab.data(), 0, sizeof(PuschShape) * N_SHAPES);
    for (uint32_t n = 1; n < N_SHAPES; n++) {
        PuschShape    &sh = tab[n];
        const uint32_t M  = 12 * n;
        sh.sqrt_M = (float)sqrt((double)M);
        sh.r_M    = recip_up(M);
        uint32_t rem = M, Ns = 1;
        while (rem > 1) {
            uint32_t R;
            if (rem % 9 == 0) R = 9;
            else if (rem % 3 == 0) R = 3;
            else if (rem % 5 == 0) R = 5;
            else if (rem % 8 == 0) R = 8;
            else if (rem % 4 == 0) R = 4;
            else if (rem % 2 == 0) R = 2;
            else { R = 7; while (rem % R) R += 2; }
            if (sh.n_pass == MAX_PASSES) { ctx->err = "transform size with more passes than the table has room for"; return MI_LTE_ERR_UNSUPPORTED; }
            sh.pass[sh.n_pass++] = {R, Ns, M / R, M / (Ns * R), recip_up(M / R), recip_up(Ns)};
            Ns *= R;
            rem /= R;
        }
    }
    void *d = nullptr;
    MI_HIP_CHECK(ctx, hipMalloc(&d, sizeof(PuschShape) * N_SHAPES));
    ctx->owned.push_back(d);
    MI_H2D(ctx, d, tab.data(), sizeof(PuschShape) * N_SHAPES);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    ctx->d_pusch_shapes = d;
    return MI_LTE_OK;
}

static uint32_t ul_qpp_size_at_least(uint32_t B)
{
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (LTE_QPP_ROWS[r].K >= B) return LTE_QPP_ROWS[r].K;
    return 0;
}

// h_dmrs (optional): caller-supplied reference signals, 4 x 12*N_prb floats per allocation back to back -- the
// per-call host form passes the arrays liblte_phy_ul_init left in the caller's LIBLTE_PHY_STRUCT
extern "C" void mi_lte_pusch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pusch_plan *pl);
int mi_pusch_plan_create_impl(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, const uint32_t *h_unit_subfr_num,
                              const uint32_t *h_unit_n_id_cell, uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                              const float *h_dmrs, mi_lte_pusch_plan **out)
{
    if (!ctx || !cfg || (!ul && !h_dmrs) || !h_unit_subfr_num || !h_unit_n_id_cell || !h_allocs || !out || n_alloc == 0 || n_units == 0)
        return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl    = new mi_lte_pusch_plan();
    auto  guard = on_fail([&] { (void)hipStreamSynchronize(ctx->stream); mi_lte_pusch_plan_destroy(nullptr, pl); });
    pl->cfg     = *cfg;
    pl->n_alloc = n_alloc;
    std::map<uint32_t, std::vector<uint32_t>> byK;
    std::map<uint32_t, uint32_t>              emaxK;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> dmrs_at; // (cell, subframe, N_prb) -> float offset
    std::vector<float>     dmrs;
    std::vector<PuschDesc> desc(n_alloc);
    uint32_t max_tbs = 0;
    size_t   off = 0;
    pl->h_e_off.resize(n_alloc);
    pl->h_e_len.resize(n_alloc);
    for (uint32_t a = 0; a < n_alloc; a++) {
        const mi_lte_pdsch_alloc &al = h_allocs[a];
        const uint32_t B = al.tbs + 24, K = (B <= 6144) ? ul_qpp_size_at_least(B) : 0;
        // N_prb the reference has a transform pre-decoding plan for (liblte_phy.cc:2360-2377)
        const bool planned = al.N_prb > 0 && al.N_prb < cfg->N_rb_dl && (al.N_prb % 2 == 0 || al.N_prb % 3 == 0 || al.N_prb % 5 == 0);
        if (K == 0 || !planned || al.mod_type > 3 || al.unit >= n_units) {
            ctx->err = "PUSCH allocation outside the envelope (one code block; N_prb < N_rb_ul and divisible by 2, 3 or 5) or malformed";
            return MI_LTE_ERR_UNSUPPORTED;
        }
        for (uint32_t sl = 0; sl < 2; sl++) // a resource block past the carrier would be read out of the neighbouring symbol row
            for (uint32_t i = 0; i < al.N_prb; i++)
                if (al.prb[sl][i] >= cfg->N_rb_dl) {
                    ctx->err = "PUSCH allocation names a resource block outside the carrier";
                    return MI_LTE_ERR_INVALID_ARG;
                }
        const uint32_t sf = h_unit_subfr_num[al.unit] % 10, cell = h_unit_n_id_cell[al.unit], M = 12 * al.N_prb;
        if (h_dmrs) {
            const uint32_t at = (uint32_t)dmrs.size();
            dmrs.insert(dmrs.end(), h_dmrs, h_dmrs + 4 * (size_t)M);
            h_dmrs += 4 * (size_t)M;
            desc[a] = {sf, cell, at};
        } else {
            auto key = std::make_tuple(cell, sf, al.N_prb);
            auto it  = dmrs_at.find(key);
            if (it == dmrs_at.end()) {
                const uint32_t at = (uint32_t)dmrs.size();
                dmrs.resize(dmrs.size() + 4 * (size_t)M);
                int rc = mi_lte_ul_dmrs_pusch(ul, cell, sf, al.N_prb, &dmrs[at], &dmrs[at + M], &dmrs[at + 2 * M], &dmrs[at + 3 * M]);
                if (rc != MI_LTE_OK) return rc;
                it = dmrs_at.emplace(key, at).first;
            }
            desc[a] = {sf, cell, it->second};
        }
        byK[K].push_back(a);
        max_tbs = std::max(max_tbs, al.tbs);
        const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1, E = 12 * M * Qm;
        pl->M_max     = std::max(pl->M_max, M);
        pl->words_max = std::max(pl->words_max, (E + 31) / 32 + 1);
        emaxK[K]       = std::max(emaxK[K], E);
        pl->h_e_off[a] = (uint32_t)(off >> 6); // in 64-byte units
        pl->h_e_len[a] = E;
        off += (E + 63) & ~63u;
    }
    if (pl->words_max > 4096) { ctx->err = "allocation larger than the scrambling table"; return MI_LTE_ERR_UNSUPPORTED; }
    pl->e_bytes    = off;
    pl->out_stride = (max_tbs + 63) & ~63u;
    std::vector<uint32_t> cb_alloc;
    for (auto &kv : byK) {
        pl->groups.push_back({kv.first, (uint32_t)kv.second.size(), (uint32_t)cb_alloc.size(), emaxK[kv.first]});
        cb_alloc.insert(cb_alloc.end(), kv.second.begin(), kv.second.end());
    }
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_desc, sizeof(PuschDesc) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_dmrs, sizeof(float) * std::max<size_t>(dmrs.size(), 1)));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_off, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_len, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_cb_alloc, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e, pl->e_bytes ? pl->e_bytes : 64));
    MI_H2D(ctx, pl->d_allocs, h_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc);
    MI_H2D(ctx, pl->d_desc, desc.data(), sizeof(PuschDesc) * n_alloc);
    MI_H2D(ctx, pl->d_dmrs, dmrs.data(), sizeof(float) * dmrs.size());
    MI_H2D(ctx, pl->d_e_off, pl->h_e_off.data(), sizeof(uint32_t) * n_alloc);
    MI_H2D(ctx, pl->d_cb_alloc, cb_alloc.data(), sizeof(uint32_t) * n_alloc);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    guard.armed = false;
    *out = pl;
    return MI_LTE_OK;
}

extern "C" {

int mi_lte_pusch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, const uint32_t *h_unit_subfr_num,
                             const uint32_t *h_unit_n_id_cell, uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                             mi_lte_pusch_plan **out)
{
    if (!ul) return MI_LTE_ERR_INVALID_ARG;
    return mi_pusch_plan_create_impl(ctx, cfg, ul, h_unit_subfr_num, h_unit_n_id_cell, n_units, h_allocs, n_alloc, nullptr, out);
}

void mi_lte_pusch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pusch_plan *pl)
{
    if (!pl) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(pl->d_allocs);
    (void)hipFree(pl->d_desc);
    (void)hipFree(pl->d_dmrs);
    (void)hipFree(pl->d_e_off);
    (void)hipFree(pl->d_e_len);
    (void)hipFree(pl->d_cb_alloc);
    (void)hipFree(pl->d_e);
    mi_multi_cache_free(&pl->multi);
    delete pl;
}

uint32_t mi_lte_pusch_plan_out_stride(const mi_lte_pusch_plan *pl) { return pl ? pl->out_stride : 0; }

int mi_lte_pusch_plan_soft_bits(const mi_lte_pusch_plan *pl, uint32_t alloc, const int8_t **d_e, uint32_t *n_bits)
{
    if (!pl || alloc >= pl->n_alloc || !d_e || !n_bits) return MI_LTE_ERR_INVALID_ARG;
    *d_e    = pl->d_e + (size_t)pl->h_e_off[alloc] * 64;
    *n_bits = pl->h_e_len[alloc];
    return MI_LTE_OK;
}

int mi_lte_pusch_decode_run(mi_lte_ctx *ctx, mi_lte_pusch_plan *pl, const float *d_subframes, uint8_t *d_out_bits, int32_t *d_status)
{
    if (!ctx || !pl || !d_subframes || !d_out_bits || !d_status) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    if ((rc = pusch_shapes(ctx)) != MI_LTE_OK) return rc;
    GoldTables   gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    // as many of the 12 data symbols side by side as fit in ~40 KiB of ping-pong buffers (all 12 up to 17 PRB)
    uint32_t S_par = 12;
    while (S_par > 1 && (size_t)S_par * pl->M_max * 2 * sizeof(float2) > 40 * 1024) S_par = S_par == 12 ? 6 : S_par == 6 ? 3 : S_par - 1; // 12, or a divisor of 6
    const size_t lds = sizeof(float) * EST_ROWS * (size_t)pl->M_max + sizeof(float2) * (size_t)pl->M_max * (1 + 2 * S_par) +
                       sizeof(uint32_t) * (pl->words_max + 1);
    // Workgroup size: the kernel's phases hold 3 M, M, 2 M, 12 M / R and 12 M items (M = 12 N_prb sub-carriers) between barriers, and the
    // heavy ones (atan2f, sincosf, the divisions) are the short ones -- a workgroup much wider than 3 M leaves whole wavefronts waiting at
    // every barrier for the one that has the transcendental work (SQ_WAIT_ANY: 67 % of the wave time at 256 threads and M = 72).  Measured on
    // W5 (16 UEs x 6 PRB): 64 / 128 / 192 / 256 threads = see profiles/r04_variants_pusch_threads.txt; larger allocations keep 256.
    uint32_t threads = pl->M_max <= 96 ? 192u : PUSCH_THREADS;
    if (const char *ev = getenv("MI_LTE_PUSCH_THREADS")) { // (tuning aid)
        const int t = atoi(ev);
        if (t >= 64 && t <= (int)PUSCH_THREADS && t % 64 == 0) threads = (uint32_t)t;
    }
#define MI_PUSCH_LAUNCH(T) MI_LAUNCH(ctx, "k_pusch_demod", k_pusch_demod<T>, dim3(pl->n_alloc), dim3(T), lds, d_subframes, (uint32_t)mi_lte_ul_subframe_floats(), \
                                    pl->d_allocs, pl->d_desc, pl->d_dmrs, gt, pl->d_e, pl->d_e_off, pl->d_e_len, pl->M_max, S_par, recip_up(S_par), \
                                    static_cast<const PuschShape *>(ctx->d_pusch_shapes))
    if (threads == 64) MI_PUSCH_LAUNCH(64); else if (threads == 128) MI_PUSCH_LAUNCH(128); else if (threads == 192) MI_PUSCH_LAUNCH(192); else MI_PUSCH_LAUNCH(256);
#undef MI_PUSCH_LAUNCH
    MI_HIP_CHECK(ctx, hipGetLastError());
    // several block sizes (the UEs of a subframe rarely share one): one launch set over all of them (turbo.hip: KSeg), as in the PDSCH chain
    std::vector<MiKGroup> take;
    if (ctx->merged_decode && pl->groups.size() >= 2)
        for (auto &gr : pl->groups)
            if (mi_turbo_ref_multi_takes(gr.K, gr.e_max)) take.push_back(MiKGroup{gr.K, gr.n_cb, gr.cb_base, gr.e_max});
    if (take.size() >= 2) {
        rc = mi_turbo_ref_multi(ctx, take.data(), (uint32_t)take.size(), pl->d_allocs, pl->d_cb_alloc, pl->d_e, pl->d_e_off, pl->d_e_len, d_out_bits, pl->out_stride, d_status,
                                /*ul=*/true, false, &pl->multi);
        if (rc != MI_LTE_OK) return rc;
    }
    for (auto &gr : pl->groups) {
        if (take.size() >= 2 && mi_turbo_ref_multi_takes(gr.K, gr.e_max)) continue;
        rc = mi_turbo_ref_group(ctx, gr.K, gr.n_cb, pl->d_allocs, pl->d_cb_alloc + gr.cb_base, pl->d_e, pl->d_e_off, pl->d_e_len,
                                d_out_bits, pl->out_stride, d_status, gr.e_max, /*ul=*/true);
        if (rc != MI_LTE_OK) return rc;
    }
    ctx->last_kernels = take.size() >= 2 ? "k_pusch_demod:1,k_cb_desc:1,k_turbo_prep,k_turbo_siso:2,k_turbo_perm,k_turbo_vote per workgroup width over all block sizes"
                                         : "k_pusch_demod:1,k_turbo_prep,k_turbo_siso,k_turbo_perm,k_turbo_vote per block size";
    return MI_LTE_OK;
}

} // extern "C"

// The same decode in three steps for a caller that builds ONE launch chain per subframe (hostapi.cc: mi_lte_ul_subframe_decode_host): the
// descriptors and tables go into the context's mapped pinned block while the stream is idle (the caller has waited), the kernel is queued
// behind the caller's front end, the results are read after the caller's single wait.  At most MI_PUCCH_STAGED_MAX resources.
int mi_pucch_stage(mi_lte_ctx *ctx, uint32_t N_rb_ul, const mi_lte_pucch_res *h_res, const float *h_tables, uint32_t n_res, MiPucchStaged *st)
{
    if (!ctx || !h_res || !h_tables || !st || n_res == 0 || n_res > MI_PUCCH_STAGED_MAX) return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t r = 0; r < n_res; r++)
        if (h_res[r].format > 2 || h_res[r].N_1_p_pucch >= N_rb_ul) return MI_LTE_ERR_INVALID_ARG;
    const size_t b_res = sizeof(PucchRes) * (size_t)n_res, b_tab = sizeof(float) * PUCCH_TAB_FLOATS * (size_t)n_res, b_out = 4 * (size_t)n_res;
    st->n_res = n_res;
    st->o_tab = (b_res + 255) & ~(size_t)255;
    st->o_out = (st->o_tab + b_tab + 255) & ~(size_t)255;
    int rc = mi_ctx_small_results(ctx, st->o_out + b_out, (void **)&st->h_base, (void **)&st->d_base);
    if (rc != MI_LTE_OK) return rc;
    PucchRes *res = (PucchRes *)st->h_base;
    for (uint32_t r = 0; r < n_res; r++) res[r] = PucchRes{h_res[r].unit, h_res[r].format, h_res[r].N_1_p_pucch};
    memcpy(st->h_base + st->o_tab, h_tables, b_tab);
    return MI_LTE_OK;
}
int mi_pucch_launch(mi_lte_ctx *ctx, const MiPucchStaged *st, uint32_t N_rb_ul, const float *d_subframes)
{
    MI_LAUNCH(ctx, "k_pucch_decode", k_pucch_decode, dim3(st->n_res), dim3(64), 0, d_subframes, (uint32_t)mi_lte_ul_subframe_floats(), N_rb_ul, 1u,
              (const PucchRes *)st->d_base, (const float *)(st->d_base + st->o_tab), (uint8_t *)(st->d_base + st->o_out));
    MI_HIP_CHECK(ctx, hipGetLastError());
    return MI_LTE_OK;
}
void mi_pucch_collect(const MiPucchStaged *st, uint8_t *h_bits, uint32_t *h_n_bits, uint32_t *h_rc)
{
    const uint8_t *o = (const uint8_t *)st->h_base + st->o_out;
    for (uint32_t r = 0; r < st->n_res; r++) { h_bits[2 * r] = o[4 * r]; h_bits[2 * r + 1] = o[4 * r + 1]; h_n_bits[r] = o[4 * r + 2]; h_rc[r] = o[4 * r + 3]; }
}

extern "C" {
// liblte_phy_pucch_format_1_1a_1b_channel_decode for a batch of PUCCH resources over UL device subframes
int mi_lte_pucch_decode_run(mi_lte_ctx *ctx, uint32_t N_rb_ul, uint32_t N_ant, const float *d_subframes, const mi_lte_pucch_res *h_res,
                            const float *h_tables, uint32_t n_res, uint8_t *h_bits /*[n_res][2]*/, uint32_t *h_n_bits, uint32_t *h_rc)
{
    if (!ctx || !d_subframes || !h_res || !h_tables || n_res == 0 || !h_bits || !h_n_bits || !h_rc || N_rb_ul < 6 || N_rb_ul > 100 || N_ant != 1)
        return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t r = 0; r < n_res; r++)
        if (h_res[r].format > 2 || h_res[r].N_1_p_pucch >= N_rb_ul) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t b_res = sizeof(PucchRes) * (size_t)n_res, b_tab = sizeof(float) * PUCCH_TAB_FLOATS * (size_t)n_res, b_out = 4 * (size_t)n_res;
    const size_t o_tab = (b_res + 255) & ~(size_t)255, o_out = (o_tab + b_tab + 255) & ~(size_t)255;
    // a few resources (a per-call caller has one): descriptors, reference tables and results live in pinned host memory that the kernel reads
    // and writes itself -- three copy commands less; a batch goes through scratch
    char *base, *h_base = nullptr;
    int   rc = MI_LTE_OK;
    if (mi_ctx_small_results(ctx, o_out + b_out, (void **)&h_base, (void **)&base) != MI_LTE_OK) {
        h_base = nullptr;
        rc = mi_ctx_reserve_scratch(ctx, o_out + b_out);
        if (rc != MI_LTE_OK) return rc;
        base = (char *)ctx->scratch;
    }
    std::vector<PucchRes> res(n_res);
    for (uint32_t r = 0; r < n_res; r++) res[r] = PucchRes{h_res[r].unit, h_res[r].format, h_res[r].N_1_p_pucch};
    if (h_base) {
        MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx)); // (a kernel of an earlier call may still be reading the block)
        memcpy(h_base, res.data(), b_res);
        memcpy(h_base + o_tab, h_tables, b_tab);
    } else {
        MI_H2D(ctx, base, res.data(), b_res);
        MI_H2D(ctx, base + o_tab, h_tables, b_tab);
    }
    MI_LAUNCH(ctx, "k_pucch_decode", k_pucch_decode, dim3(n_res), dim3(64), 0, d_subframes, (uint32_t)mi_lte_ul_subframe_floats(), N_rb_ul, N_ant,
              (const PucchRes *)base, (const float *)(base + o_tab), (uint8_t *)(base + o_out));
    MI_HIP_CHECK(ctx, hipGetLastError());
    std::vector<uint8_t> o_copy;
    if (!h_base) {
        o_copy.resize(b_out);
        MI_D2H(ctx, o_copy.data(), base + o_out, b_out);
    }
    MI_HIP_CHECK(ctx, h_base ? mi_stream_wait(ctx, n_res) : hipStreamSynchronize(ctx->stream)); // (a copy into pageable memory is done when the runtime says so)
    const uint8_t *o = h_base ? (const uint8_t *)h_base + o_out : o_copy.data();
    for (uint32_t r = 0; r < n_res; r++) { h_bits[2 * r] = o[4 * r]; h_bits[2 * r + 1] = o[4 * r + 1]; h_n_bits[r] = o[4 * r + 2]; h_rc[r] = o[4 * r + 3]; }
    ctx->last_kernels = "k_pucch_decode:1";
    return MI_LTE_OK;
}

} // extern "C"
