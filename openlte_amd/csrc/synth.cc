// Host-side synthesis of benchmark / test inputs (the role LTE_fdd_dl_file_gen plays for the
// reference, LTE_fdd_dl_file_gen/src/LTE_fdd_dl_fg_samp_buf.cc:269-668): a minimal LTE downlink
// transmitter -- CRC24A, turbo encode, rate match, scramble, QAM map, CRS, OFDM modulate, a one-tap
// channel and int8 quantisation -- written to the 3GPP sections the reference's TX side follows
// (cited per function).  Pure host C++, no GPU work; it only produces the captures the receive
// kernels are fed with, so there is still no CPU path for anything the library *decodes*.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mi_lte.h"
#include "lte_tables.h"
#include "synth.hpp"

namespace synth {

// splitmix64: small, seedable, reproducible across platforms
uint64_t Rng::next()
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double Rng::uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
double Rng::normal()
{
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}

// 3GPP TS 36.212 5.1.1, gCRC24A (reference: calc_crc, liblte_phy.cc:9713-9743)
void crc24a(const uint8_t *bits, uint32_t n, uint8_t p[24])
{
    uint32_t rem = 0;
    for (uint32_t i = 0; i < n + 24; i++) {
        rem = (rem << 1) | (i < n ? bits[i] : 0u);
        if (rem & 0x1000000u) rem ^= 0x1864CFBu;
    }
    for (int i = 0; i < 24; i++) p[i] = (uint8_t)((rem >> (23 - i)) & 1u);
}

bool qpp_params(uint32_t K, uint32_t *f1, uint32_t *f2)
{
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (LTE_QPP_ROWS[r].K == K) { *f1 = LTE_QPP_ROWS[r].f1; *f2 = LTE_QPP_ROWS[r].f2; return true; }
    return false;
}

// QPP interleaver, 36.212 5.1.3.2.3.  ref_wrap reproduces the reference's uint32 evaluation
// (liblte_phy.cc:10954-10958) so that captures made here loop back through the REF decoder the way
// captures made by the reference's own generator do.
void qpp_map(uint32_t K, bool ref_wrap, std::vector<uint16_t> &pi)
{
    uint32_t f1 = 0, f2 = 0;
    qpp_params(K, &f1, &f2);
    pi.resize(K);
    for (uint32_t i = 0; i < K; i++)
        pi[i] = ref_wrap ? (uint16_t)((f1 * i + f2 * i * i) % K) : (uint16_t)(((uint64_t)f1 * i + (uint64_t)f2 * i * i) % K);
}

// 36.212 5.1.3.2.1 constituent encoder g0 = 1+D^2+D^3 (feedback), g1 = 1+D+D^3, with the 3+1 step
// termination of 5.1.3.2.2 (reference: turbo_constituent_encoder, liblte_phy.cc:10855-10924)
static void rsc(const uint8_t *in, uint32_t K, uint8_t *z, uint8_t *x_tail)
{
    int s1 = 0, s2 = 0, s3 = 0;
    for (uint32_t i = 0; i < K + 4; i++) {
        int fb = s2 ^ s3;
        int s0 = (i < K) ? (fb ^ in[i]) : 0;
        z[i]      = (uint8_t)(s0 ^ s1 ^ s3);
        x_tail[i] = (uint8_t)fb;
        s3 = s2; s2 = s1; s1 = s0;
    }
}

// 36.212 5.1.3.2: d planar d0[D] d1[D] d2[D], D = K+4 (reference: turbo_encode, liblte_phy.cc:10541-10589)
void turbo_encode(const uint8_t *c, uint32_t K, bool ref_wrap, uint8_t *d)
{
    const uint32_t        D = K + 4;
    std::vector<uint16_t> pi;
    std::vector<uint8_t>  z(D), x(D), zp(D), xp(D), cp(K);
    qpp_map(K, ref_wrap, pi);
    rsc(c, K, z.data(), x.data());
    for (uint32_t i = 0; i < K; i++) cp[i] = c[pi[i]];
    rsc(cp.data(), K, zp.data(), xp.data());
    uint8_t *d0 = d, *d1 = d + D, *d2 = d + 2 * D;
    for (uint32_t i = 0; i < K; i++) { d0[i] = c[i]; d1[i] = z[i]; d2[i] = zp[i]; }
    d0[K] = x[K];      d1[K] = z[K];       d2[K] = x[K + 1];
    d0[K + 1] = z[K + 1]; d1[K + 1] = x[K + 2]; d2[K + 1] = z[K + 2];
    d0[K + 2] = xp[K];    d1[K + 2] = zp[K];    d2[K + 2] = xp[K + 1];
    d0[K + 3] = zp[K + 1]; d1[K + 3] = xp[K + 2]; d2[K + 3] = zp[K + 2];
}

} // namespace synth

extern "C" {

// n code blocks of size K: random information bits -> turbo encode -> soft values in the reference's
// interleaved layout d[i*3+x], value +amp for bit 0 / -amp for bit 1, each sign flipped with
// probability flip ("64QAM-like" hard +-127 soft bits, SURVEY 8d W3 distribution (i)).
int mi_lte_synth_turbo_soft_i8(uint32_t K, uint32_t n, double flip, int amp, uint64_t seed, int ref_wrap,
                               int8_t *h_soft /* n*3*(K+4) */, uint8_t *h_tx_bits /* n*K, may be NULL */)
{
    uint32_t f1, f2;
    if (!h_soft || !synth::qpp_params(K, &f1, &f2) || amp < 1 || amp > 127) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t       D = K + 4;
    synth::Rng           rng(seed);
    std::vector<uint8_t> c(K), d(3 * D);
    for (uint32_t b = 0; b < n; b++) {
        for (uint32_t i = 0; i < K; i++) c[i] = (uint8_t)(rng.next() & 1u);
        synth::turbo_encode(c.data(), K, ref_wrap != 0, d.data());
        int8_t *o = h_soft + (size_t)b * 3 * D;
        for (uint32_t i = 0; i < D; i++)
            for (int x = 0; x < 3; x++) {
                int v = d[x * D + i] ? -amp : amp;
                if (rng.uniform() < flip) v = -v;
                o[i * 3 + x] = (int8_t)v;
            }
        if (h_tx_bits) memcpy(h_tx_bits + (size_t)b * K, c.data(), K);
    }
    return MI_LTE_OK;
}

// same blocks through a BPSK/AWGN channel: float soft values 1-2b + sigma*n
int mi_lte_synth_turbo_soft_f32(uint32_t K, uint32_t n, double sigma, uint64_t seed, int ref_wrap, float *h_soft,
                                uint8_t *h_tx_bits)
{
    uint32_t f1, f2;
    if (!h_soft || !synth::qpp_params(K, &f1, &f2)) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t       D = K + 4;
    synth::Rng           rng(seed);
    std::vector<uint8_t> c(K), d(3 * D);
    for (uint32_t b = 0; b < n; b++) {
        for (uint32_t i = 0; i < K; i++) c[i] = (uint8_t)(rng.next() & 1u);
        synth::turbo_encode(c.data(), K, ref_wrap != 0, d.data());
        float *o = h_soft + (size_t)b * 3 * D;
        for (uint32_t i = 0; i < D; i++)
            for (int x = 0; x < 3; x++) o[i * 3 + x] = (float)((d[x * D + i] ? -1.0 : 1.0) + sigma * rng.normal());
        if (h_tx_bits) memcpy(h_tx_bits + (size_t)b * K, c.data(), K);
    }
    return MI_LTE_OK;
}

} // extern "C"
