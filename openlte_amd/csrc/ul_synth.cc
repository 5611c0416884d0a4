// Host-side synthesis of uplink test / benchmark captures: a minimal PUSCH transmitter written as the inverse
// of the reference's RECEIVER (liblte_phy_pusch_channel_decode, liblte/src/liblte_phy.cc:2801-2935), following
// 36.212 5.2.2 (UL-SCH coding) and 36.211 5.3-5.6 (scrambling, modulation, transform precoding, DMRS, SC-FDMA).
// The reference's own uplink transmit helpers cannot serve as the model: its channel interleaver walks the
// bit matrix per bit instead of per modulation symbol (liblte_phy.cc:12023-12042) and its SC-FDMA modulator
// multiplies where it should add an offset (:8565-8570) -- the eNodeB never transmits uplink, so neither is
// exercised.  Pure host C++; nothing here is part of the receive path.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mi_lte.h"
#include "lte_tables.h"
#include "prach_sets.hpp"
#include "synth.hpp"

extern "C" {

size_t mi_lte_synth_ul_unit_len(uint32_t fft_size)
{
    const size_t s = 2048 / (fft_size ? fft_size : 2048);
    return (30720 / s + 16 + 15) / 16 * 16; // one subframe plus slack for the channel delay
}

int mi_lte_synth_ul_units_i8(const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, uint32_t n_units, const uint32_t *h_subfr_num,
                             const uint32_t *h_n_id_cell, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                             const mi_lte_synth_channel *chan, int8_t *h_iq, uint8_t *h_tx_bits, uint32_t tbs_stride)
{
    if (!cfg || !ul || !h_subfr_num || !h_n_id_cell || !h_allocs || !chan || !h_iq) return MI_LTE_ERR_INVALID_ARG;
    if (!synth::valid_grid(cfg->fft_size, cfg->N_rb_dl)) return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t u = 0; u < n_units; u++) {
        if (h_n_id_cell[u] > 503) return MI_LTE_ERR_INVALID_ARG;
        for (uint32_t a = 0; a < n_alloc; a++)
            if (!synth::valid_alloc(h_allocs[(size_t)u * n_alloc + a], cfg->N_rb_dl) || (h_tx_bits && h_allocs[(size_t)u * n_alloc + a].tbs > tbs_stride))
                return MI_LTE_ERR_INVALID_ARG;
    }
    const uint32_t N = cfg->fft_size, sc = 2048 / N, cp0 = 160 / sc, cpe = 144 / sc, N_rb = cfg->N_rb_dl, half = 6 * N_rb, N_sc = 12 * N_rb;
    const size_t unit_len = mi_lte_synth_ul_unit_len(N);
    synth::Rng   rng(chan->seed);
    std::vector<float>  g_re(14 * (size_t)N_sc), g_im(14 * (size_t)N_sc);
    std::vector<double> xr(N), xi(N), t_re(unit_len + 64), t_im(unit_len + 64);

    for (uint32_t u = 0; u < n_units; u++) {
        const uint32_t sf = h_subfr_num[u] % 10, cell = h_n_id_cell[u];
        std::fill(g_re.begin(), g_re.end(), 0.f);
        std::fill(g_im.begin(), g_im.end(), 0.f);
        for (uint32_t a = 0; a < n_alloc; a++) {
            const mi_lte_pdsch_alloc &al = h_allocs[(size_t)u * n_alloc + a];
            const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
            const uint32_t M = 12 * al.N_prb, G = 12 * M * Qm, B = al.tbs + 24;
            uint32_t       K = 0, f1, f2;
            for (int r = 0; r < LTE_QPP_N_SIZES; r++)
                if (LTE_QPP_ROWS[r].K >= B) { K = LTE_QPP_ROWS[r].K; break; }
            if (al.N_prb == 0 || al.N_prb > N_rb || B > 6144 || K != B || !synth::qpp_params(K, &f1, &f2)) return MI_LTE_ERR_UNSUPPORTED;
            // UL-SCH: CRC24A, turbo code, rate matching with N_cb = K_w (36.212 5.2.2.1-5.2.2.5)
            std::vector<uint8_t> b(K), d(3 * (K + 4)), g(G), h(G), c(G);
            for (uint32_t i = 0; i < al.tbs; i++) b[i] = (uint8_t)(rng.next() & 1u);
            synth::crc24a(b.data(), al.tbs, b.data() + al.tbs);
            if (h_tx_bits) memcpy(h_tx_bits + ((size_t)u * n_alloc + a) * tbs_stride, b.data(), al.tbs);
            synth::turbo_encode(b.data(), K, true, d.data());
            synth::rate_match(d.data(), K + 4, 0xFFFFFFFFu, al.rv_idx, G, g.data());
            // channel interleaver without control information (36.212 5.2.2.8): the R' x 12 matrix of Q_m-bit symbols is
            // written row by row and read column by column -> h[(s*M + k)*Q + q] = g[(k*12 + s)*Q + q]
            for (uint32_t s = 0; s < 12; s++)
                for (uint32_t k = 0; k < M; k++)
                    for (uint32_t q = 0; q < Qm; q++) h[((size_t)s * M + k) * Qm + q] = g[((size_t)k * 12 + s) * Qm + q];
            synth::gold((al.rnti << 14) | (sf << 9) | cell, G, c.data());
            for (uint32_t i = 0; i < G; i++) h[i] ^= c[i];
            std::vector<float> m_re(12 * (size_t)M), m_im(12 * (size_t)M);
            synth::modulate(h.data(), 12 * M, al.mod_type, m_re.data(), m_im.data());
            // transform precoding (36.211 5.3.3): forward DFT of each symbol's M points scaled by 1/sqrt(M), what a real
            // UE sends (data and DMRS at the same power).  The reference's receiver then recovers M times the
            // constellation point (unnormalised backward DFT times sqrt(M), liblte_phy.cc:6644-6657): QPSK decisions
            // survive (every soft bit saturates at +-1), the inner bits of 16/64QAM do not -- reproduced, not repaired.
            std::vector<float> d0r(M), d0i(M), d1r(M), d1i(M);
            int rc = mi_lte_ul_dmrs_pusch(ul, cell, sf, al.N_prb, d0r.data(), d0i.data(), d1r.data(), d1i.data());
            if (rc != MI_LTE_OK) return rc;
            const double scale = 1.0 / std::sqrt((double)M);
            std::vector<double> cs(M), sn(M);
            for (uint32_t t = 0; t < M; t++) { cs[t] = std::cos(-2.0 * M_PI * t / M); sn[t] = std::sin(-2.0 * M_PI * t / M); }
            for (uint32_t s = 0; s < 12; s++) {
                const uint32_t L = s < 3 ? s : s < 9 ? s + 1 : s + 2;
                for (uint32_t k = 0; k < M; k++) {
                    double ar = 0, ai = 0;
                    for (uint32_t n = 0; n < M; n++) {
                        const uint32_t t = (uint32_t)(((uint64_t)k * n) % M);
                        ar += m_re[s * M + n] * cs[t] - m_im[s * M + n] * sn[t];
                        ai += m_re[s * M + n] * sn[t] + m_im[s * M + n] * cs[t];
                    }
                    const uint32_t scx = al.prb[L / 7][k / 12] * 12 + k % 12;
                    g_re[L * N_sc + scx] = (float)(ar * scale);
                    g_im[L * N_sc + scx] = (float)(ai * scale);
                }
            }
            for (uint32_t k = 0; k < M; k++) { // DMRS on symbols 3 and 10 (36.211 5.5.2.1.2)
                const uint32_t s0 = al.prb[0][k / 12] * 12 + k % 12, s1 = al.prb[1][k / 12] * 12 + k % 12;
                g_re[3 * N_sc + s0]  = d0r[k]; g_im[3 * N_sc + s0]  = d0i[k];
                g_re[10 * N_sc + s1] = d1r[k]; g_im[10 * N_sc + s1] = d1i[k];
            }
        }
        // SC-FDMA modulation (36.211 5.6): grid column i sits on bin (N - half + i) mod N of an N-point inverse DFT whose
        // output is rotated by exp(+i*pi*n/N) (the half-sub-carrier shift the receiver undoes, liblte_phy.cc:8685-8690)
        std::fill(t_re.begin(), t_re.end(), 0.0);
        std::fill(t_im.begin(), t_im.end(), 0.0);
        size_t pos = 0;
        for (uint32_t s = 0; s < 14; s++) {
            const uint32_t cp = (s % 7 == 0) ? cp0 : cpe;
            std::fill(xr.begin(), xr.end(), 0.0);
            std::fill(xi.begin(), xi.end(), 0.0);
            for (uint32_t i = 0; i < N_sc; i++) {
                const uint32_t bin = (N - half + i) % N;
                xr[bin] = g_re[s * N_sc + i];
                xi[bin] = g_im[s * N_sc + i];
            }
            synth::idft(xr, xi);
            for (uint32_t n = 0; n < N; n++) {
                const double cr = std::cos(M_PI * n / N), ci = std::sin(M_PI * n / N), a = xr[n], b2 = xi[n];
                xr[n] = (a * cr - b2 * ci) / N;
                xi[n] = (a * ci + b2 * cr) / N;
            }
            // cyclic prefix: the rotated symbol continued backwards, x(n - N) = -x(n) because of the half-carrier shift
            for (uint32_t i = 0; i < cp && pos + i < unit_len; i++) { t_re[pos + i] = -xr[N - cp + i]; t_im[pos + i] = -xi[N - cp + i]; }
            for (uint32_t i = 0; i < N && pos + cp + i < unit_len; i++) { t_re[pos + cp + i] = xr[i]; t_im[pos + cp + i] = xi[i]; }
            pos += cp + N;
        }
        // flat channel, integer delay, AWGN, int8 quantisation
        const double   gain = chan->gain_min + (chan->gain_max - chan->gain_min) * rng.uniform();
        const double   ph   = 2.0 * M_PI * rng.uniform() - M_PI;
        const uint32_t dly  = (uint32_t)(rng.uniform() * (chan->max_delay + 0.999));
        double p_sig = 0, peak = 0;
        for (size_t i = 0; i < unit_len; i++) {
            p_sig += t_re[i] * t_re[i] + t_im[i] * t_im[i];
            peak = std::max(peak, std::max(std::fabs(t_re[i]), std::fabs(t_im[i])));
        }
        p_sig /= (double)unit_len;
        const double scale = (peak > 0 ? chan->peak / peak : 1.0);
        const double sigma = chan->snr_db >= 200 ? 0.0 : std::sqrt(p_sig / std::pow(10.0, chan->snr_db / 10.0) / 2.0);
        const double hr = gain * std::cos(ph), hi = gain * std::sin(ph);
        int8_t *o = h_iq + (size_t)u * unit_len * 2;
        for (size_t i = 0; i < unit_len; i++) {
            double sr = 0, si = 0;
            if (i >= dly) { sr = t_re[i - dly]; si = t_im[i - dly]; }
            const double yr = hr * sr - hi * si + sigma * rng.normal(), yi = hr * si + hi * sr + sigma * rng.normal();
            const long qr = std::lround(yr * scale / std::max(1.0, chan->gain_max)), qi = std::lround(yi * scale / std::max(1.0, chan->gain_max));
            o[2 * i]     = (int8_t)std::max(-127L, std::min(127L, qr));
            o[2 * i + 1] = (int8_t)std::max(-127L, std::min(127L, qi));
        }
    }
    return MI_LTE_OK;
}


// ---- PRACH (36.211 5.7.2-5.7.3): preamble v of root u is x_u((n + C_v) mod 839); its 839-point DFT sits on the PRACH
// sub-carriers (1.25 kHz spacing) phi + K(k0 + 1/2) + k in natural order; cyclic prefix + sequence (twice for formats 2, 3)
size_t mi_lte_synth_prach_len(uint32_t fft_size, uint32_t preamble_format)
{
    const PrachGeom pg = prach_geom(preamble_format > 4 ? 0 : preamble_format);
    const uint32_t  sc = 2048 / (fft_size ? fft_size : 2048);
    return ((pg.T_cp_30 + pg.T_fft_30 * pg.reps + 1024) / sc + 15) / 16 * 16;
}

// the physical roots u of the cell's 64 preambles in the order they are enumerated (prach_sets.hpp: prach_root_set; host arithmetic only)
int mi_lte_prach_root_set(const mi_lte_prach_cfg *pc, uint32_t *h_u /* [64] */, uint32_t *n_roots)
{
    if (!pc || !h_u || !n_roots) return MI_LTE_ERR_INVALID_ARG;
    const PrachRootSet rs = prach_root_set(pc->preamble_format, pc->root_seq_idx, pc->zczc, pc->hs_flag != 0);
    if (!rs.ok) return pc->preamble_format > 4 || pc->root_seq_idx >= prach_geom(pc->preamble_format > 4 ? 0 : pc->preamble_format).n_root_idx ? MI_LTE_ERR_INVALID_ARG : MI_LTE_ERR_UNSUPPORTED;
    for (uint32_t r = 0; r < rs.n_roots; r++) h_u[r] = rs.u[r];
    *n_roots = rs.n_roots;
    return MI_LTE_OK;
}

int mi_lte_synth_prach_i8(const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *pc, uint32_t n_occ, const uint32_t *h_preamble_idx,
                          const uint32_t *h_delay, const mi_lte_synth_channel *chan, int8_t *h_iq)
{
    if (!cfg || !pc || !h_preamble_idx || !h_delay || !chan || !h_iq || pc->preamble_format > 4 || pc->root_seq_idx >= prach_geom(pc->preamble_format).n_root_idx)
        return MI_LTE_ERR_INVALID_ARG;
    // the six PRACH resource blocks lie on the grid (k_0 below is unsigned arithmetic), and a delay keeps the preamble inside the occasion
    if (!synth::valid_grid(cfg->fft_size, cfg->N_rb_dl) || pc->freq_offset + 6 > cfg->N_rb_dl) return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t o = 0; o < n_occ; o++)
        if (h_delay[o] > 1024 / (2048 / cfg->fft_size)) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t  fmt = pc->preamble_format;
    const PrachGeom pg = prach_geom(fmt); // format 4: N_zc = 139 on 7.5 kHz sub-carriers, 4 096 samples, one repetition (36.211 table 5.7.1-1)
    const uint32_t N_ZC = pg.n_zc;
    const uint32_t N = cfg->fft_size, sc = 2048 / N, T = pg.T_fft_30 / sc, T_cp = pg.T_cp_30 / sc;
    const uint32_t reps = pg.reps;
    const size_t   len = mi_lte_synth_prach_len(N, fmt);
    const uint32_t k_0 = pc->freq_offset * 12 - cfg->N_rb_dl * 12 / 2 + N / 2, start = pg.phi + pg.K * k_0 + pg.K / 2;
    synth::Rng rng(chan->seed);
    std::vector<double> xr(N_ZC), xi(N_ZC), Xr(N_ZC), Xi(N_ZC), cz(N_ZC), sz(N_ZC), ct(T), st(T), sr(T), si(T), t_re(len), t_im(len);
    for (uint32_t t = 0; t < N_ZC; t++) { cz[t] = std::cos(-2.0 * M_PI * t / N_ZC); sz[t] = std::sin(-2.0 * M_PI * t / N_ZC); }
    for (uint32_t t = 0; t < T; t++) { ct[t] = std::cos(2.0 * M_PI * t / T); st[t] = std::sin(2.0 * M_PI * t / T); }
    for (uint32_t o = 0; o < n_occ; o++) {
        // which root and cyclic shift carry preamble index p (prach_preamble_seq_gen's enumeration, liblte_phy.cc:7155-7290)
        uint32_t p = h_preamble_idx[o] % 64, r = 0, u = 0, C_v = 0;
        for (;; r++) {
            if (r >= 64) return MI_LTE_ERR_INVALID_ARG;
            u = prach_root(fmt, (pc->root_seq_idx + r) % pg.n_root_idx); // the logical root order is cyclic (36.211 5.7.2; prach.hip)
            const PrachSets ps = prach_sets(u, pc->zczc, pc->hs_flag != 0, fmt);
            if (!ps.ok) return MI_LTE_ERR_UNSUPPORTED; // (the reference's own generator divides by zero / reads past its table there)
            if (p <= ps.v_max) {
                C_v = pc->hs_flag ? ps.d_start * (p / ps.N_RA_shift) + (p % ps.N_RA_shift) * ps.N_cs : p * ps.N_cs;
                break;
            }
            p -= ps.v_max + 1;
        }
        for (uint32_t n = 0; n < N_ZC; n++) {
            const uint32_t m = (n + C_v) % N_ZC;
            const double   ph = -M_PI * u * m * (m + 1) / N_ZC;
            xr[n] = std::cos(ph); xi[n] = std::sin(ph);
        }
        for (uint32_t k = 0; k < N_ZC; k++) {
            double ar = 0, ai = 0;
            uint32_t t = 0;
            for (uint32_t n = 0; n < N_ZC; n++) {
                ar += xr[n] * cz[t] - xi[n] * sz[t];
                ai += xr[n] * sz[t] + xi[n] * cz[t];
                t += k; if (t >= N_ZC) t -= N_ZC;
            }
            Xr[k] = ar; Xi[k] = ai;
        }
        for (uint32_t n = 0; n < T; n++) { // s(n) = sum_k X(k) exp(+2*pi*i*idx_k*n/T)
            double ar = 0, ai = 0;
            for (uint32_t k = 0; k < N_ZC; k++) {
                const uint32_t idx = (k + start + T / 2) % T, t = (uint32_t)(((uint64_t)idx * n) % T);
                ar += Xr[k] * ct[t] - Xi[k] * st[t];
                ai += Xr[k] * st[t] + Xi[k] * ct[t];
            }
            sr[n] = ar; si[n] = ai;
        }
        std::fill(t_re.begin(), t_re.end(), 0.0);
        std::fill(t_im.begin(), t_im.end(), 0.0);
        const uint32_t dly = h_delay[o];
        for (uint32_t i = 0; i < T_cp + reps * T; i++) {
            const uint32_t n = (i + T - (T_cp % T)) % T; // cyclic prefix = tail of the sequence
            if (dly + i < len) { t_re[dly + i] = sr[n]; t_im[dly + i] = si[n]; }
        }
        const double gain = chan->gain_min + (chan->gain_max - chan->gain_min) * rng.uniform(), ph = 2.0 * M_PI * rng.uniform() - M_PI;
        double p_sig = 0, peak = 0;
        for (size_t i = 0; i < len; i++) { p_sig += t_re[i] * t_re[i] + t_im[i] * t_im[i]; peak = std::max(peak, std::max(std::fabs(t_re[i]), std::fabs(t_im[i]))); }
        p_sig /= (double)len;
        const double scale = peak > 0 ? chan->peak / peak : 1.0;
        const double sigma = chan->snr_db >= 200 ? 0.0 : std::sqrt(p_sig / std::pow(10.0, chan->snr_db / 10.0) / 2.0);
        const double hr = gain * std::cos(ph), hi = gain * std::sin(ph);
        int8_t *ob = h_iq + (size_t)o * len * 2;
        for (size_t i = 0; i < len; i++) {
            const double yr = hr * t_re[i] - hi * t_im[i] + sigma * rng.normal(), yi = hr * t_im[i] + hi * t_re[i] + sigma * rng.normal();
            const long qr = std::lround(yr * scale / std::max(1.0, chan->gain_max)), qi = std::lround(yi * scale / std::max(1.0, chan->gain_max));
            ob[2 * i]     = (int8_t)std::max(-127L, std::min(127L, qr));
            ob[2 * i + 1] = (int8_t)std::max(-127L, std::min(127L, qi));
        }
    }
    return MI_LTE_OK;
}

} // extern "C"
