// Device helpers shared by the downlink (chain.hip) and uplink (uplink.hip) demodulators.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace {

struct GoldTables { const uint32_t *x1; const uint32_t *x2b; uint32_t words; };

// word w of the Gold sequence for c_init: x1's word XOR the basis words of x2 for every set bit of c_init (the sequence is linear in
// c_init).  The table reads are issued eight at a time so that they are in flight together -- a plain loop over the set bits
// makes every L2 round trip wait for the previous one, which dominated the prologue of the per-allocation kernels.
__device__ __forceinline__ uint32_t gold_word(const GoldTables &gt, uint32_t c_init, uint32_t w)
{
    uint32_t v = gt.x1[w], m = c_init;
    while (m) {
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool     on = m != 0;
            const uint32_t b  = on ? (uint32_t)__builtin_ctz(m) : 0u;
            t[k] = gt.x2b[__umul24(b, gt.words) + w]; // unconditional (row 0 when the bits are used up): no branch between the loads; b < 31, words < 2^13
            t[k] = on ? t[k] : 0u;
            m &= m - 1; // 0 stays 0
        }
        v ^= (t[0] ^ t[1]) ^ (t[2] ^ t[3]) ^ (t[4] ^ t[5]) ^ (t[6] ^ t[7]);
    }
    return v;
}

// wrap_phase (liblte_phy.cc:14105-14116): float difference compared against the double constant,
// the +-2*pi correction is evaluated in double and rounded back to float.
// The comparison needs no double: (float)M_PI = 3.14159274... is the smallest float that is >= the double constant (its predecessor
// 3.14159250... is below it), so for a float d, (double)d >= M_PI exactly when d >= (float)M_PI, and likewise on the negative side.
__device__ __forceinline__ float wrap_phase(float p1, float p2)
{
    const float pi_up = 3.14159274101257324f;
    while (p1 - p2 >= pi_up) p1 = (float)((double)p1 - 2 * M_PI);
    while (p1 - p2 <= -pi_up) p1 = (float)((double)p1 + 2 * M_PI);
    return p1;
}

// The same with the common case first: the wrap loops run only when |p1 - p2| >= pi on entry (never for NaN), which one comparison decides.
__device__ __forceinline__ float wrap_phase_rare(float p1, float p2)
{
    if (__builtin_expect(fabsf(p1 - p2) >= 3.14159274101257324f, 0)) p1 = wrap_phase(p1, p2);
    return p1;
}
// x / 3.0f (the second and fourth segments of the time interpolation divide by their three symbols)
__device__ __forceinline__ float div3(float x) { return x / 3.0f; }

// sin / cos of an unwrapped phase for the time interpolation: two-constant reduction to [-pi, pi] (the phases are a few turns at
// most), then the hardware's v_sin_f32 / v_cos_f32, which take revolutions.  The estimate is a float-tolerance stage
// (TOL_CE = 1e-4 in tests/test_frontend_gpu.py; measured against the CPU restatement of the reference the estimate's relative L2 error is 2.4e-7 this way and
// 2.2e-7 with libm's sincosf, which spent two thirds of the estimator's instructions here).
__device__ __forceinline__ void ce_sincos(float x, float &sn, float &cs)
{
    const float k = rintf(x * 0.15915494309189533577f);
    float       r = fmaf(-k, 6.28318548202514648438f, x); // 2 pi rounded to float ...
    r             = fmaf(-k, -1.74845553146951715e-07f, r); // ... and the rest of it
    r *= 0.15915494309189533577f;
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}

// Time interpolation of the channel estimate for antenna ports 0 and 1 (liblte_phy.cc:6119-6190): magnitude m[z] and phase a[z] at the
// 14 symbols of a subframe from the values M[i], A[i] at the CRS symbols 0, 4, 7, 11 and 14 (the next subframe's first).  The
// reference's order is kept: symbols 0, 4, 7, 11 take the CRS-symbol values as they are; then segment by segment the later phase is
// wrapped against the (already wrapped) earlier one, the slope is (hi - lo) / d with the phase difference wrapped against 0, and the
// symbols in between are reached by repeated subtraction from the later end.  The estimate itself is m * (cos a, sin a).
__device__ __forceinline__ void ce_time_interp5(const float (&M)[5], float (&A)[5], float (&m)[14], float (&a)[14])
{
    m[0] = M[0]; a[0] = A[0]; m[4] = M[1]; a[4] = A[1]; m[7] = M[2]; a[7] = A[2]; m[11] = M[3]; a[11] = A[3];
    float fm, fa, cm, ca;
#define MI_CE_SLOPE(hi, lo, dv) do { fm = (M[hi] - M[lo]) / (dv); A[hi] = wrap_phase(A[hi], A[lo]); fa = A[hi] - A[lo]; \
                                     fa = wrap_phase(fa, 0.0f); fa /= (dv); cm = M[hi]; ca = A[hi]; } while (0)
    MI_CE_SLOPE(1, 0, 4);
#pragma unroll
    for (int z = 3; z > 0; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
    MI_CE_SLOPE(2, 1, 3);
#pragma unroll
    for (int z = 6; z > 4; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
    MI_CE_SLOPE(3, 2, 4);
#pragma unroll
    for (int z = 10; z > 7; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
    MI_CE_SLOPE(4, 3, 3);
#pragma unroll
    for (int z = 13; z > 11; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
#undef MI_CE_SLOPE
}
// The same interpolation for ONE slot (symbols 7*slot .. 7*slot + 6 -> m[0..6], a[0..6]): what the PDSCH demodulator's threads need, each
// working on one slot of one sub-carrier.  The second slot's segments still depend on the phases wrapped in the first slot's (A1 against A0,
// A2 against the wrapped A1), so those two wraps are made, but not the first slot's slopes and chains.
__device__ __forceinline__ void ce_time_interp5_slot(const float (&M)[5], float (&A)[5], uint32_t slot, float (&m)[7], float (&a)[7])
{
    float fm, fa, cm, ca;
#define MI_CE_SLOPE(hi, lo, dv) do { fm = (M[hi] - M[lo]) / (dv); A[hi] = wrap_phase(A[hi], A[lo]); fa = A[hi] - A[lo]; \
                                     fa = wrap_phase(fa, 0.0f); fa /= (dv); cm = M[hi]; ca = A[hi]; } while (0)
    if (slot == 0) {
        m[0] = M[0]; a[0] = A[0]; m[4] = M[1]; a[4] = A[1]; // symbols 0 and 4 as estimated (before A1 is wrapped)
        MI_CE_SLOPE(1, 0, 4);
#pragma unroll
        for (int z = 3; z > 0; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
        MI_CE_SLOPE(2, 1, 3);
#pragma unroll
        for (int z = 6; z > 4; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
    } else {
        m[0] = M[2]; a[0] = A[2]; m[4] = M[3]; a[4] = A[3]; // symbols 7 and 11 as estimated
        A[1] = wrap_phase(A[1], A[0]);
        A[2] = wrap_phase(A[2], A[1]);
        MI_CE_SLOPE(3, 2, 4);
#pragma unroll
        for (int z = 3; z > 0; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; } // symbols 10, 9, 8
        MI_CE_SLOPE(4, 3, 3);
#pragma unroll
        for (int z = 6; z > 4; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; } // symbols 13, 12
    }
#undef MI_CE_SLOPE
}

// atan2f as the reference's host computes it -- glibc 2.35 (the image's libm: the fdlibm single-precision algorithm with glibc's 2^25
// large-argument threshold), restated operation by operation in float arithmetic (this file is compiled with -ffp-contract=off), so that
// the DECISIONS the reference takes on an angle come out the same where the angle sits within an ulp of 0, +-pi/4, +-pi/2, +-3pi/4 or pi.
// The device library's atan2f is good to a few ulp, which is not enough there: the differential soak found one QPSK symbol in 7.7e8
// (re = -1.07, im = +4.0e-8: libm 0x40490fda < pi, device atan2f 0x40490fdb > pi, the other quadrant, tools/repro_seed17.py).
// Bit-identical to libm's atan2f on 1.2e8 argument pairs incl. NaN / infinities / zeros (tests/test_oracle.py through
// mi_lte_model_atan2f).  A host with another libm may round differently: SURVEY 8c files that under the libm tolerance.
__host__ __device__ inline uint32_t ref_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
__host__ __device__ inline float    ref_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ inline float ref_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)ref_f2u(x), ix = hx & 0x7fffffff;
    int           id;
    if (ix >= 0x4c000000) { // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) { // |x| < 0.4375
        if (ix < 0x39800000) return x; // |x| < 2^-12
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
__host__ __device__ inline float ref_atan2f(float y, float x)
{
    const float   tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)ref_f2u(x), hy = (int32_t)ref_f2u(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return ref_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2 * sign(x) + sign(y)
    if (iy == 0) return m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float         z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = ref_atanf(fabsf(y / x));
    return m == 0 ? z : m == 1 ? ref_u2f(ref_f2u(z) ^ 0x80000000u) : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

// The same out of line: it is reached by one symbol in tens of thousands, and where the call site sits in a branch the benchmark's
// modulation never takes (k_pdsch_demod: QPSK next to the 64QAM loop) inlining it changed that loop's register allocation -- 4.42 against
// 4.72 ms per 65 536 subframes.  Where the QPSK path IS the hot loop (k_pusch_demod, k_pdcch_decode) a call inside it costs more than the
// code (W5 2.65 -> 2.54 M subframes/s), so demap_symbol takes the choice as a template parameter (profiles/r03t_*, r03u_*).
__host__ __device__ __attribute__((noinline)) float ref_atan2f_call(float y, float x) { return ref_atan2f(y, x); }

// (int)(127 * get_soft_decision(...)) (liblte_phy.cc:13880-13900 with max_dist = 1, scaled and truncated as modulation_demapper does with it).
// The reference's sqrtf is the correctly rounded one, which on this hardware is v_sqrt_f32 plus a dozen instructions of scaling and fix-up; the
// integer it ends in depends on that last bit only when 127 (1 - dist) lies next to an integer.  So: the hardware's root (1 ulp), the same
// float operations after it, and the exact route whenever the result is within 2^-13 of an integer (or the squared distance is down where
// v_sqrt_f32 no longer sees its argument).  Two roots that differ by up to 2 ulp (<= 2^-23) move 1 - dist by at most 3 * 2^-24 and the product by
// at most 127 * 3 * 2^-24 + 2^-17 < 2^-15, a quarter of the guard: outside it both routes truncate to the same integer; inside it the exact
// route is the one that runs -- for the whole wavefront when any of its lanes asks (one symbol in 4 000, so one wavefront in 60).  NaN falls
// through both routes alike.
__device__ __forceinline__ int soft_decision_127(float rx_re, float rx_im, float exp_re, float exp_im)
{
    const float d_re = rx_re - exp_re, d_im = rx_im - exp_im, d2 = d_re * d_re + d_im * d_im;
    const float cap = 1.0f - (1.0f / 120);
    const float s = __builtin_amdgcn_sqrtf(d2);
    float       v = 127 * (1.0f - (s >= cap ? cap : s));
    const float f = __builtin_amdgcn_fractf(v);
    // (a wave-uniform branch around something the compiler may not hoist: left to itself it turns the `if` into selects and computes both
    // routes for everybody)
    if (__builtin_amdgcn_ballot_w64(!(f > 0x1p-13f && f < 1.0f - 0x1p-13f && d2 >= 0x1p-96f)) != 0) {
        asm volatile("; exact route" ::: "memory");
        float dist = sqrtf(d2);
        if (dist >= cap) dist = cap;
        v = 127 * (1.0f - dist);
    }
    return (int)v;
}

// modulation_demapper (liblte_phy.cc:9502-9660) for one symbol; writes Q_m int8 values
template <bool ATAN_OUT_OF_LINE = false> __device__ __forceinline__ void demap_symbol(float re, float im, uint32_t mod, int8_t *b)
{
    const float r2 = (float)(1 / sqrt(2.0)), t10 = (float)(2 / sqrt(10.0)), t42 = (float)(2 / sqrt(42.0)),
                f42 = (float)(4 / sqrt(42.0)), s42 = (float)(6 / sqrt(42.0));
    if (mod == 3) {
        const float ar = fabsf(re), ai = fabsf(im);
        b[0] = (re > 0) ? 127 : -127;
        b[1] = (im > 0) ? 127 : -127;
        if (ar < f42) { b[2] = 127;  b[4] = (ar > t42) ? 127 : -127; }
        else          { b[2] = -127; b[4] = (ar < s42) ? 127 : -127; }
        if (ai < f42) { b[3] = 127;  b[5] = (ai > t42) ? 127 : -127; }
        else          { b[3] = -127; b[5] = (ai < s42) ? 127 : -127; }
    } else if (mod == 2) {
        b[0] = (re > 0) ? 127 : -127;
        b[1] = (im > 0) ? 127 : -127;
        b[2] = (fabsf(re) < t10) ? 127 : -127;
        b[3] = (fabsf(im) < t10) ? 127 : -127;
    } else if (mod == 1) {
        float er, ei;
        // The reference picks the quadrant from atan2f(im, re) compared with 0, +-pi/2 and pi in double.  Away from the axes that
        // is the pair of signs (the margin below is two orders wider than the rounding of any atan2f); on or next to an axis --
        // zeros, signed zeros, angles that round to float(pi/2) or float(pi), NaN -- the comparisons themselves are evaluated, on the
        // host libm's atan2f (ref_atan2f above)
        const float ar = fabsf(re), ai = fabsf(im);
        if (ar > 1e-5f * ai && ai > 1e-5f * ar) { // false for NaN, zeros and infinities as well
            er = re > 0 ? r2 : -r2;
            ei = im > 0 ? r2 : -r2;
        } else {
            const float ang = ATAN_OUT_OF_LINE ? ref_atan2f_call(im, re) : ref_atan2f(im, re);
            if (((double)ang >= 0) && ((double)ang < M_PI / 2))         { er = r2;  ei = r2; }
            else if (((double)ang >= -M_PI / 2) && ((double)ang < 0))   { er = r2;  ei = -r2; }
            else if (((double)ang >= M_PI / 2) && ((double)ang < M_PI)) { er = -r2; ei = r2; }
            else                                                        { er = -r2; ei = -r2; }
        }
        const int m = soft_decision_127(re, im, er, ei);
        b[0] = (int8_t)((er > 0) ? m : -m);
        b[1] = (int8_t)((ei > 0) ? m : -m);
    } else {
        const float ang = ATAN_OUT_OF_LINE ? ref_atan2f_call(im, re) : ref_atan2f(im, re);
        if (((double)ang > -M_PI / 4) && ((double)ang < 3 * M_PI / 4)) b[0] = (int8_t)soft_decision_127(re, im, r2, r2);
        else                                                          b[0] = (int8_t)(-soft_decision_127(re, im, -r2, -r2));
    }
}


// The hard decisions of modulation_demapper for 16QAM / 64QAM (liblte_phy.cc:9560-9660) as a bit mask: bit k set <=> soft bit k is
// -127 (all of them are +-127 for these two modulations).  Same comparisons as demap_symbol above, negated where the reference's
// branch yields -127, so NaNs fall the same way.
__device__ __forceinline__ uint32_t qam_neg_bits(float re, float im, uint32_t mod)
{
    const float t10 = (float)(2 / sqrt(10.0)), t42 = (float)(2 / sqrt(42.0)), f42 = (float)(4 / sqrt(42.0)), s42 = (float)(6 / sqrt(42.0));
    const float ar = fabsf(re), ai = fabsf(im);
    uint32_t    t = (!(re > 0) ? 1u : 0u) | (!(im > 0) ? 2u : 0u);
    if (mod == 3) {
        const bool in_r = ar < f42, in_i = ai < f42;
        t |= (!in_r ? 4u : 0u) | (!in_i ? 8u : 0u);
        t |= ((in_r ? !(ar > t42) : !(ar < s42)) ? 16u : 0u) | ((in_i ? !(ai > t42) : !(ai < s42)) ? 32u : 0u);
    } else {
        t |= (!(ar < t10) ? 4u : 0u) | (!(ai < t10) ? 8u : 0u);
    }
    return t;
}
// The same mask for a symbol x = (nr, ni) / den WITHOUT the two divisions.  The reference divides (liblte_phy.cc:7680-7690, 7694-7765)
// and then only compares the quotients with 0 and with k * 2/sqrt(42) (k = 1, 2, 3; 2/sqrt(10) for 16QAM) -- liblte_phy.cc:9573-9659.
// With u = |n| * (rcp(den) * sqrt(42)/2) those thresholds sit at the integers: floor(u) picks the amplitude ring, the sign bit of n the
// half plane.  u carries a relative error below 2^-21 against the reference's rounded quotient measured in threshold units (v_rcp_f32 is
// 1 ulp, three roundings, the two constants' own roundings, the quotient's rounding: 2.5 + 1 + 0.5 units of 2^-23, times u <= 3 at the
// last threshold -> 1.5e-6), so a decision can only differ when u lies within that of an integer.  Such symbols -- anything within
// 2^-16 of an integer, which also catches zeros, underflowing quotients, a zero / subnormal / infinite / NaN denominator (u becomes 0,
// inf or NaN: the test is written so that NaN fails it) and |u| >= 2^23 -- take the reference's own route: the IEEE divisions and
// qam_neg_bits above.  So the mask is the reference's by construction; the guarded route is taken by about one symbol in 16 000.
// (tests/test_chain_gpu.py::test_qam_decisions_next_to_the_thresholds places symbols 0-3 ulp around every threshold.)
template <uint32_t MOD> __device__ __forceinline__ uint32_t qam_neg_bits_nodiv(float nr, float ni, float den)
{
    static_assert(MOD == 2 || MOD == 3, "16QAM / 64QAM");
    constexpr float    C      = MOD == 3 ? 3.24037034920393f : 1.58113883008419f; // sqrt(42)/2, sqrt(10)/2
    constexpr uint32_t K_TOP  = MOD == 3 ? 3u : 1u;                               // rings past the last threshold are the last ring
    // ring k -> (bit 2, bit 4) of the real part; the imaginary part's bits are one position higher.  64QAM, rings 0..3: bit 2 (outer half)
    // 0 0 1 1, bit 4 (-127 in the innermost and outermost ring) 1 0 0 1.  16QAM, rings 0..1: bit 2 = k.
    constexpr uint32_t RINGS  = MOD == 3 ? (16u | (0u << 8) | (4u << 16) | (20u << 24)) : (0u | (4u << 8));
    const float rc = __builtin_amdgcn_rcpf(den) * C;
    const float ur = nr * rc, ui = ni * rc;
    const float fr = ur - rintf(ur), fi = ui - rintf(ui);
    if (!(fabsf(fr) >= 0x1p-16f) || !(fabsf(fi) >= 0x1p-16f)) return qam_neg_bits(nr / den, ni / den, MOD);
    const uint32_t kr = min((uint32_t)fabsf(ur), K_TOP), ki = min((uint32_t)fabsf(ui), K_TOP); // (v_cvt_u32_f32 saturates)
    const uint32_t s  = (ref_f2u(ur) >> 31) | ((ref_f2u(ui) >> 30) & 2u);
    return s | ((RINGS >> (8 * kr)) & 0xFFu) | (((RINGS >> (8 * ki)) & 0xFFu) << 1);
}
// entry t of the table that turns such a mask into soft bits: byte k = (t >> k) & 1 ? -127 : 127, k = 0..5
__device__ __forceinline__ uint2 qam_lut_entry(uint32_t t)
{
    uint32_t w[2] = {0, 0};
    for (uint32_t k = 0; k < 6; k++) w[k >> 2] |= (((t >> k) & 1u) ? 0x81u : 0x7Fu) << (8 * (k & 3));
    return make_uint2(w[0], w[1]);
}

} // namespace
