// Device helpers shared by the downlink (chain.hip) and uplink (uplink.hip) demodulators.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

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
            t[k] = gt.x2b[b * gt.words + w]; // unconditional (row 0 when the bits are used up): no branch between the loads
            t[k] = on ? t[k] : 0u;
            m &= m - 1; // 0 stays 0
        }
        v ^= (t[0] ^ t[1]) ^ (t[2] ^ t[3]) ^ (t[4] ^ t[5]) ^ (t[6] ^ t[7]);
    }
    return v;
}

// get_soft_decision (liblte_phy.cc:13880-13900) with max_dist = 1
__device__ __forceinline__ float soft_decision(float rx_re, float rx_im, float exp_re, float exp_im)
{
    const float d_re = rx_re - exp_re, d_im = rx_im - exp_im;
    float dist = sqrtf(d_re * d_re + d_im * d_im);
    const float cap = 1.0f - (1.0f / 120);
    if (dist >= cap) dist = cap;
    return 1.0f - dist;
}

// modulation_demapper (liblte_phy.cc:9502-9660) for one symbol; writes Q_m int8 values
__device__ __forceinline__ void demap_symbol(float re, float im, uint32_t mod, int8_t *b)
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
        // is the pair of signs (atan2f is good to a few ulp, the margin below is five orders wider); on or next to an axis --
        // zeros, signed zeros, angles that round to float(pi/2) or float(pi), NaN -- the comparisons themselves are evaluated.
        const float ar = fabsf(re), ai = fabsf(im);
        if (ar > 1e-5f * ai && ai > 1e-5f * ar) { // false for NaN, zeros and infinities as well
            er = re > 0 ? r2 : -r2;
            ei = im > 0 ? r2 : -r2;
        } else {
            const float ang = atan2f(im, re);
            if (((double)ang >= 0) && ((double)ang < M_PI / 2))         { er = r2;  ei = r2; }
            else if (((double)ang >= -M_PI / 2) && ((double)ang < 0))   { er = r2;  ei = -r2; }
            else if (((double)ang >= M_PI / 2) && ((double)ang < M_PI)) { er = -r2; ei = r2; }
            else                                                        { er = -r2; ei = -r2; }
        }
        const int m = (int)(127 * soft_decision(re, im, er, ei));
        b[0] = (int8_t)((er > 0) ? m : -m);
        b[1] = (int8_t)((ei > 0) ? m : -m);
    } else {
        const float ang = atan2f(im, re);
        if (((double)ang > -M_PI / 4) && ((double)ang < 3 * M_PI / 4)) b[0] = (int8_t)(int)(127 * soft_decision(re, im, r2, r2));
        else                                                          b[0] = (int8_t)(-(int)(127 * soft_decision(re, im, -r2, -r2)));
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
// entry t of the table that turns such a mask into soft bits: byte k = (t >> k) & 1 ? -127 : 127, k = 0..5
__device__ __forceinline__ uint2 qam_lut_entry(uint32_t t)
{
    uint32_t w[2] = {0, 0};
    for (uint32_t k = 0; k < 6; k++) w[k >> 2] |= (((t >> k) & 1u) ? 0x81u : 0x7Fu) << (8 * (k & 3));
    return make_uint2(w[0], w[1]);
}

} // namespace
