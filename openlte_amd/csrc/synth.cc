// Host-side synthesis of benchmark / test inputs (the role LTE_fdd_dl_file_gen plays for the
// reference, LTE_fdd_dl_file_gen/src/LTE_fdd_dl_fg_samp_buf.cc:269-668): a minimal LTE downlink
// transmitter -- CRC24A, turbo encode, rate match, scramble, QAM map, CRS, OFDM modulate, a one-tap
// channel and int8 quantisation -- written to the 3GPP sections the reference's TX side follows
// (cited per function).  Pure host C++, no GPU work; it only produces the captures the receive
// kernels are fed with, so there is still no CPU path for anything the library *decodes*.
#include <algorithm>
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


// Gold sequence, 36.211 7.2 (reference: generate_prs_c, liblte_phy.cc:9669-9704)
void gold(uint32_t c_init, uint32_t len, uint8_t *c)
{
    uint32_t x1 = 1, x2 = c_init;
    for (uint32_t n = 0; n < 1600; n++) { // Nc = 1600: advance both registers past the discarded prefix
        uint32_t n1 = ((x1 >> 3) ^ x1) & 1u, n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
        x1 = (x1 >> 1) | (n1 << 30);
        x2 = (x2 >> 1) | (n2 << 30);
    }
    for (uint32_t n = 0; n < len; n++) {
        c[n]        = (uint8_t)((x1 ^ x2) & 1u);
        uint32_t n1 = ((x1 >> 3) ^ x1) & 1u, n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
        x1 = (x1 >> 1) | (n1 << 30);
        x2 = (x2 >> 1) | (n2 << 30);
    }
}

// sub-block interleaver + circular buffer read-out, 36.212 5.1.4.1 (reference:
// liblte_phy_rate_match_turbo, liblte_phy.cc:11081-11237).  d planar, F = 0 only.
void rate_match(const uint8_t *d, uint32_t D, uint32_t N_cb_limit, uint32_t rv, uint32_t E, uint8_t *e)
{
    uint32_t R = (D + 31) / 32, K_pi = 32 * R, N_d = K_pi - D, K_w = 3 * K_pi;
    std::vector<int16_t> w(K_w);
    auto y = [&](int x, uint32_t n) -> int16_t { return n < N_d ? (int16_t)-1 : (int16_t)d[x * D + (n - N_d)]; };
    for (uint32_t c = 0; c < 32; c++)
        for (uint32_t r = 0; r < R; r++) {
            uint32_t i = c * R + r, n = 32 * r + LTE_SUBBLOCK_COL_PERM[c];
            w[i]                = y(0, n);
            w[K_pi + 2 * i]     = y(1, n);
            w[K_pi + 2 * i + 1] = y(2, (LTE_SUBBLOCK_COL_PERM[c] + 32 * r + 1) % K_pi);
        }
    uint32_t N_cb = N_cb_limit < K_w ? N_cb_limit : K_w;
    uint32_t k0   = R * (2 * ((N_cb + 8 * R - 1) / (8 * R)) * rv + 2);
    for (uint32_t k = 0, j = 0; k < E; j++) {
        int16_t v = w[(k0 + j) % N_cb];
        if (v >= 0) e[k++] = (uint8_t)v;
    }
}

// 36.211 7.1.2-7.1.4 (reference: modulation_mapper, liblte_phy.cc:8704-9490)
void modulate(const uint8_t *b, uint32_t n_sym, uint32_t mod, float *re, float *im)
{
    const float r2 = (float)(1 / std::sqrt(2.0)), r10 = (float)(1 / std::sqrt(10.0)), r42 = (float)(1 / std::sqrt(42.0));
    for (uint32_t i = 0; i < n_sym; i++) {
        if (mod == 1) {
            re[i] = r2 * (1 - 2 * b[2 * i]);
            im[i] = r2 * (1 - 2 * b[2 * i + 1]);
        } else if (mod == 2) {
            const uint8_t *q = b + 4 * i;
            re[i] = r10 * (float)((1 - 2 * q[0]) * (2 - (1 - 2 * q[2])));
            im[i] = r10 * (float)((1 - 2 * q[1]) * (2 - (1 - 2 * q[3])));
        } else if (mod == 3) {
            const uint8_t *q = b + 6 * i;
            re[i] = r42 * (float)((1 - 2 * q[0]) * (4 - (1 - 2 * q[2]) * (2 - (1 - 2 * q[4]))));
            im[i] = r42 * (float)((1 - 2 * q[1]) * (4 - (1 - 2 * q[3]) * (2 - (1 - 2 * q[5]))));
        } else {
            re[i] = im[i] = r2 * (1 - 2 * b[i]);
        }
    }
}

// PBCH / PSS / SSS sub-carrier window and the PDSCH RE exclusion rule (36.211 6.3.5; reference:
// liblte_phy.cc:3722-3789 on RX, :3501-3540 and :3621-3657 on TX)
void sync_window(uint32_t N_rb_dl, uint32_t *first_sc, uint32_t *last_sc)
{
    switch (N_rb_dl) {
    case 6:  *first_sc = 0;           *last_sc = 71;          break;
    case 15: *first_sc = 4 * 12 + 6;  *last_sc = 11 * 12 - 7; break;
    case 25: *first_sc = 9 * 12 + 6;  *last_sc = 16 * 12 - 7; break;
    case 50: *first_sc = 22 * 12;     *last_sc = 28 * 12 - 1; break;
    case 75: *first_sc = 34 * 12 + 6; *last_sc = 41 * 12 - 7; break;
    default: *first_sc = 47 * 12;     *last_sc = 53 * 12 - 1; break;
    }
}
bool pdsch_re_excluded(uint32_t N_ant, uint32_t cell, uint32_t sf, uint32_t L, uint32_t j, uint32_t sc, uint32_t first_sc,
                       uint32_t last_sc)
{
    if (N_ant == 1 && (L % 7) == 0 && (cell % 6) == (j % 6)) return true;
    if (N_ant == 1 && (L % 7) == 4 && ((cell + 3) % 6) == (j % 6)) return true;
    if (N_ant >= 2 && ((L % 7) == 0 || (L % 7) == 4) && (cell % 3) == (j % 3)) return true;
    if (N_ant == 4 && (L % 7) == 1 && (cell % 3) == (j % 3)) return true;
    const bool in_win = sc >= first_sc && sc <= last_sc;
    if (sf == 0 && in_win && L >= 7 && L <= 10) return true;
    if ((sf == 0 || sf == 5) && in_win && (L == 5 || L == 6)) return true;
    return false;
}

// unnormalised inverse DFT (the reference's FFTW_BACKWARD plan, liblte_phy.cc:2311-2315), radix-2
void idft(std::vector<double> &xr, std::vector<double> &xi)
{
    const size_t n = xr.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(xr[i], xr[j]); std::swap(xi[i], xi[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = 2.0 * M_PI / (double)len;
        for (size_t j = 0; j < len / 2; j++) {
            const double wr = std::cos(ang * (double)j), wi = std::sin(ang * (double)j);
            for (size_t b = 0; b < n; b += len) {
                const size_t p = b + j, q = p + len / 2;
                const double tr = xr[q] * wr - xi[q] * wi, ti = xr[q] * wi + xi[q] * wr;
                xr[q] = xr[p] - tr; xi[q] = xi[p] - ti;
                xr[p] += tr;        xi[p] += ti;
            }
        }
    }
}

} // namespace synth

extern "C" {

size_t mi_lte_synth_unit_len(uint32_t fft_size)
{
    const size_t s = 2048 / (fft_size ? fft_size : 2048);
    return ((30720 + 4400) / s + 15) / 16 * 16;
}

int mi_lte_synth_dl_units_i8(const mi_lte_dl_cfg *cfg, uint32_t n_units, const uint32_t *h_subfr_num,
                             const uint32_t *h_n_id_cell, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_allocs,
                             uint32_t n_alloc, const mi_lte_synth_channel *chan, int8_t *h_iq, uint8_t *h_tx_bits,
                             uint32_t tbs_stride)
{
    if (!cfg || !h_subfr_num || !h_n_id_cell || !chan || !h_iq || cfg->N_ant != 1 || (n_alloc && !h_allocs)) return MI_LTE_ERR_INVALID_ARG;
    if (!synth::valid_grid(cfg->fft_size, cfg->N_rb_dl) || N_pdcch_symbs < 1 || N_pdcch_symbs > 4) return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t u = 0; u < n_units; u++) {
        if (h_n_id_cell[u] > 503) return MI_LTE_ERR_INVALID_ARG;
        for (uint32_t a = 0; a < n_alloc; a++)
            if (!synth::valid_alloc(h_allocs[(size_t)u * n_alloc + a], cfg->N_rb_dl) || (h_tx_bits && h_allocs[(size_t)u * n_alloc + a].tbs > tbs_stride))
                return MI_LTE_ERR_INVALID_ARG;
    }
    const uint32_t N = cfg->fft_size, sc = 2048 / N, cp0 = 160 / sc, cpe = 144 / sc, N_rb = cfg->N_rb_dl, half = 6 * N_rb, N_sc = 12 * N_rb;
    const size_t   unit_len = mi_lte_synth_unit_len(N);
    uint32_t       first_sc, last_sc;
    synth::sync_window(N_rb, &first_sc, &last_sc);
    synth::Rng rng(chan->seed);
    std::vector<float>  g_re(16 * N_sc), g_im(16 * N_sc);
    std::vector<double> xr(N), xi(N), t_re(unit_len + 64), t_im(unit_len + 64);
    const float         r2 = (float)(1 / std::sqrt(2.0));
    double              scale_all = 1.0;

    for (uint32_t u = 0; u < n_units; u++) {
        const uint32_t sf = h_subfr_num[u] % 10, cell = h_n_id_cell[u];
        std::fill(g_re.begin(), g_re.end(), 0.f);
        std::fill(g_im.begin(), g_im.end(), 0.f);
        // CRS for port 0 on symbols 0,4,7,11 of this subframe and symbol 0 of the next (36.211 6.10.1;
        // reference: liblte_phy_map_crs, liblte_phy.cc:5144-5262)
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t l = s % 7;
            if (!(l == 0 || l == 4)) continue;
            const uint32_t ns = (2 * sf + s / 7) % 20, v = (l == 0) ? 0 : 3;
            uint8_t        c[440];
            synth::gold(1024 * (7 * (ns + 1) + l + 1) * (2 * cell + 1) + 2 * cell + 1, 440, c);
            for (uint32_t j = 0; j < 2 * N_rb; j++) {
                const uint32_t k = 6 * j + (v + cell % 6) % 6, mp = j + 110 - N_rb;
                g_re[s * N_sc + k] = r2 * (1 - 2 * (float)c[2 * mp]);
                g_im[s * N_sc + k] = r2 * (1 - 2 * (float)c[2 * mp + 1]);
            }
        }
        // PDSCH allocations
        for (uint32_t a = 0; a < n_alloc; a++) {
            const mi_lte_pdsch_alloc &al = h_allocs[(size_t)u * n_alloc + a];
            const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
            // RE list in mapping order
            std::vector<uint32_t> res;
            for (uint32_t L = N_pdcch_symbs; L < 14; L++)
                for (uint32_t pi = 0; pi < al.N_prb; pi++)
                    for (uint32_t j = 0; j < 12; j++) {
                        const uint32_t scx = al.prb[L / 7][pi] * 12 + j;
                        if (!synth::pdsch_re_excluded(1, cell, sf, L, j, scx, first_sc, last_sc)) res.push_back(L * N_sc + scx);
                    }
            const uint32_t G = (uint32_t)res.size() * Qm, E = G / (2 * Qm) * (2 * Qm); // dlsch_channel_encode with N_l = 2 (:3572-3584)
            const uint32_t B = al.tbs + 24;
            uint32_t       K = 0, f1, f2;
            for (int r = 0; r < LTE_QPP_N_SIZES; r++)
                if (LTE_QPP_ROWS[r].K >= B) { K = LTE_QPP_ROWS[r].K; break; }
            if (B > 6144 || K != B || !synth::qpp_params(K, &f1, &f2)) return MI_LTE_ERR_UNSUPPORTED; // single block, F = 0 only
            std::vector<uint8_t> b(K), d(3 * (K + 4)), e(E), c(E);
            for (uint32_t i = 0; i < al.tbs; i++) b[i] = (uint8_t)(rng.next() & 1u);
            synth::crc24a(b.data(), al.tbs, b.data() + al.tbs);
            if (h_tx_bits) memcpy(h_tx_bits + ((size_t)u * n_alloc + a) * tbs_stride, b.data(), al.tbs);
            synth::turbo_encode(b.data(), K, true, d.data());
            const uint32_t K_mimo = (al.tx_mode == 3 || al.tx_mode == 4 || al.tx_mode == 8 || al.tx_mode == 9) ? 2 : 1;
            synth::rate_match(d.data(), K + 4, 250368 / (K_mimo * 8), al.rv_idx, E, e.data());
            synth::gold((al.rnti << 14) | (sf << 9) | cell, E, c.data());
            for (uint32_t i = 0; i < E; i++) e[i] ^= c[i];
            const uint32_t     M = E / Qm;
            std::vector<float> m_re(M), m_im(M);
            synth::modulate(e.data(), M, al.mod_type, m_re.data(), m_im.data());
            for (uint32_t i = 0; i < M; i++) { g_re[res[i]] = m_re[i]; g_im[res[i]] = m_im[i]; }
        }
        // OFDM modulation of 14 + 2 symbols (36.211 6.12; reference: symbols_to_samples_dl, liblte_phy.cc:8484-8530)
        std::fill(t_re.begin(), t_re.end(), 0.0);
        std::fill(t_im.begin(), t_im.end(), 0.0);
        size_t pos = 0;
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t cp = (s % 7 == 0) ? cp0 : cpe;
            std::fill(xr.begin(), xr.end(), 0.0);
            std::fill(xi.begin(), xi.end(), 0.0);
            for (uint32_t i = 0; i < half; i++) {
                xr[i + 1]     = g_re[s * N_sc + half + i];     xi[i + 1]     = g_im[s * N_sc + half + i];
                xr[N - 1 - i] = g_re[s * N_sc + half - 1 - i]; xi[N - 1 - i] = g_im[s * N_sc + half - 1 - i];
            }
            synth::idft(xr, xi);
            for (uint32_t i = 0; i < cp && pos + i < unit_len; i++) { t_re[pos + i] = xr[N - cp + i]; t_im[pos + i] = xi[N - cp + i]; }
            for (uint32_t i = 0; i < N && pos + cp + i < unit_len; i++) { t_re[pos + cp + i] = xr[i]; t_im[pos + cp + i] = xi[i]; }
            pos += cp + N;
            if (pos >= unit_len) break;
        }
        // one-tap channel, integer delay, AWGN, int8 quantisation
        // max_delay < 0: a static channel -- no random phase, no delay, one int8 scale for every unit -- so that consecutive units can be
        // laid end to end as ONE capture (a unit's look-ahead symbols are then what the next unit starts with)
        const bool   fixed = chan->max_delay < 0;
        const double gain = chan->gain_min + (chan->gain_max - chan->gain_min) * rng.uniform();
        const double ph   = fixed ? 0.0 * rng.uniform() : 2.0 * M_PI * rng.uniform() - M_PI;
        const uint32_t dly = fixed ? (uint32_t)(0.0 * rng.uniform()) : (uint32_t)(rng.uniform() * (chan->max_delay + 0.999));
        double p_sig = 0, peak = 0;
        for (size_t i = 0; i < unit_len; i++) {
            p_sig += t_re[i] * t_re[i] + t_im[i] * t_im[i];
            peak = std::max(peak, std::max(std::fabs(t_re[i]), std::fabs(t_im[i])));
        }
        p_sig /= (double)unit_len;
        if (!fixed) scale_all = (peak > 0 ? chan->peak / peak : 1.0);
        else if (u == 0) {
            // static: ONE scale whatever the units carry (and whichever call made them): the RMS of a fully loaded symbol of unit-power
            // resource elements lands at peak / 3.6 -- OFDM's crest factor stays inside int8
            std::fill(xr.begin(), xr.end(), 0.0);
            std::fill(xi.begin(), xi.end(), 0.0);
            for (uint32_t i = 0; i < half; i++) { xr[i + 1] = (i & 1) ? 1.0 : -1.0; xr[N - 1 - i] = (i % 3) ? 1.0 : -1.0; }
            synth::idft(xr, xi);
            double pr = 0;
            for (uint32_t i = 0; i < N; i++) pr += xr[i] * xr[i] + xi[i] * xi[i];
            scale_all = chan->peak / (3.6 * std::sqrt(pr / N));
        }
        const double scale = scale_all;
        const double sigma = chan->snr_db >= 200 ? 0.0 : std::sqrt(p_sig / std::pow(10.0, chan->snr_db / 10.0) / 2.0);
        const double hr = gain * std::cos(ph), hi = gain * std::sin(ph);
        int8_t *o = h_iq + (size_t)u * unit_len * 2;
        for (size_t i = 0; i < unit_len; i++) {
            double sr = 0, si = 0;
            if (i >= dly) { sr = t_re[i - dly]; si = t_im[i - dly]; }
            double yr = hr * sr - hi * si + sigma * rng.normal(), yi = hr * si + hi * sr + sigma * rng.normal();
            long qr = std::lround(yr * scale / std::max(1.0, chan->gain_max)), qi = std::lround(yi * scale / std::max(1.0, chan->gain_max));
            o[2 * i]     = (int8_t)std::max(-127L, std::min(127L, qr));
            o[2 * i + 1] = (int8_t)std::max(-127L, std::min(127L, qi));
        }
    }
    return MI_LTE_OK;
}


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
