// The uplink transmit functions of liblte_phy on the host (see tx.cc for the why): liblte_phy_pusch_channel_encode (liblte_phy.cc:2664-2799) and
// liblte_phy_generate_prach (:3219-3297).  The eNodeB never calls either -- they are the UE side of the reference's own loop-back tests -- and
// the PUSCH one is not a 36.212 transmitter: its channel interleaver steps through the multiplexed bits one BIT per modulation symbol
// (:12049-12061) and its resource mapping starts at sub-carrier 0 whatever the PRBs are (:2766-2790), which is why the library's test captures
// come from ul_synth.cc instead.  Restated as it is: what the reference's function leaves in the grid is what this one leaves.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mi_lte.h"
#include "prach_sets.hpp"
#include "synth.hpp"
#include "tx_host.h"

using namespace tx;

namespace {
// X[k] = sum_j x[j] w^(j k), w = exp(sign * 2 pi i / n), in float64 by the definition (the transform sizes here are 12 N_prb and 839 / 139: no
// radix-2), the terms added in rising j.  nz lists the inputs that are not zero (a zero term adds nothing to the sum).
struct Dft {
    uint32_t            n;
    std::vector<double> wr, wi;
    Dft(uint32_t n_, int sign) : n(n_), wr(n_), wi(n_)
    {
        for (uint32_t k = 0; k < n; k++) {
            const double a = 2.0 * M_PI * (double)k / (double)n;
            wr[k] = cos(a), wi[k] = (double)sign * sin(a);
        }
    }
    void run(const double *xr, const double *xi, const uint32_t *nz, uint32_t n_nz, float *out_re, float *out_im) const
    {
        for (uint32_t k = 0; k < n; k++) {
            double sr = 0.0, si = 0.0;
            for (uint32_t t = 0; t < n_nz; t++) {
                const uint32_t j = nz[t], idx = (uint32_t)(((uint64_t)j * k) % n);
                sr += xr[j] * wr[idx] - xi[j] * wi[idx];
                si += xr[j] * wi[idx] + xi[j] * wr[idx];
            }
            out_re[k] = (float)sr, out_im[k] = (float)si;
        }
    }
};
} // namespace

struct mi_lte_tx_ul {
    uint8_t c[MI_LTE_TX_MAX_CODE_BLOCKS][6176], e[MI_LTE_TX_MAX_CODE_BLOCKS][18432]; // ulsch_c_bits / ulsch_tx_e_bits: see dlsch_encode in tx.cc
    // pusch_z: the reference codes as many BITS as the allocation has resource elements (:2700-2706), so it has 144 N_prb / Q_m symbols for 144 N_prb
    // elements; its pre-coder copies that many transformed symbols (:6690-6696) and the mapping reads all 144 N_prb -- above BPSK the rest is what
    // earlier calls left in this array
    float z_re[14400], z_im[14400];
};

extern "C" {

int mi_lte_tx_ul_create(mi_lte_tx_ul **out)
{
    if (!out) return MI_LTE_ERR_ARG;
    mi_lte_tx_ul *t = new (std::nothrow) mi_lte_tx_ul;
    if (!t) return MI_LTE_ERR_NOMEM;
    memset(t, 0, sizeof(*t));
    *out = t;
    return MI_LTE_OK;
}
void mi_lte_tx_ul_destroy(mi_lte_tx_ul *t) { delete t; }

int mi_lte_pusch_channel_encode(mi_lte_tx_ul *t, const mi_lte_ul_cfg *ul, uint32_t ul_cell, uint32_t N_rb_ul, uint32_t N_sc_rb_ul, const mi_lte_tx_alloc *al,
                                uint32_t N_id_cell, uint32_t N_ant, uint32_t subfr_num, float *tx_re, float *tx_im)
{
    if (!t || !ul || !al || !tx_re || !tx_im) return 1;
    // one port, one layer, one codeword: the reference's layer mapper and pre-coder do nothing otherwise; a transform-precoder plan exists for
    // N_prb below N_rb_ul with a factor 2, 3 or 5 (liblte_phy.cc:2360-2378)
    const uint32_t N_prb = al->N_prb;
    if (N_ant != 1 || al->N_layers != 1 || al->N_codewords != 1 || subfr_num > 9) return 1;
    if (N_prb == 0 || N_prb >= N_rb_ul || !(N_prb % 2 == 0 || N_prb % 3 == 0 || N_prb % 5 == 0) || N_sc_rb_ul != 12 || 12 * N_prb > MI_LTE_TX_GRID_SC) return 1;
    if (!al->msg[0] && al->msg_bits[0]) return 1;
    const uint32_t Q_m = al->mod_type == MI_LTE_MOD_BPSK ? 1 : al->mod_type == MI_LTE_MOD_QPSK ? 2 : al->mod_type == MI_LTE_MOD_16QAM ? 4 : 6;
    const uint32_t M_sc = 12 * N_prb, G = M_sc * 12, tbs = al->tbs;
    if ((tbs + 24 + 6119) / 6120 > MI_LTE_TX_MAX_CODE_BLOCKS) return 1;

    // UL-SCH coding (ulsch_channel_encode, liblte_phy.cc:12236-12350; 36.212 5.2.2) without control information
    std::vector<uint8_t> b(tbs + 24, 0);
    for (uint32_t i = 0; i < al->msg_bits[0] && i < tbs; i++) b[i] = al->msg[0][i];
    crc_bits(b.data(), tbs, 0x1864CFB, 24, b.data() + tbs);
    uint32_t C = 0, F = 0, N_c[MI_LTE_TX_MAX_CODE_BLOCKS + 1] = {0};
    mi_lte_code_block_segmentation(b.data(), tbs + 24, &C, &F, &t->c[0][0], 6176, N_c);
    std::vector<uint8_t>  d(3 * (6176 + 4)), f(G + 64), g(G + 64);
    std::vector<uint32_t> N_e(C);
    const uint32_t        G_prime = G / Q_m, lambda = G_prime % C;
    for (uint32_t cb = 0; cb < C; cb++) {
        turbo_encode(&t->c[cb][0], N_c[cb], d.data());
        N_e[cb] = cb <= C - lambda - 1 ? Q_m * (G_prime / C) : Q_m * (uint32_t)ceilf((float)G_prime / (float)C);
        rate_match_turbo(d.data(), 3 * (N_c[cb] + 4), C, al->tx_mode, 1, 1, MI_LTE_CHAN_ULSCH, al->rv_idx, N_e[cb], t->e[0]);
    }
    uint32_t N_f = 0;
    for (uint32_t r = 0; r < C; r++)
        for (uint32_t j = 0; j < N_e[r]; j++) f[N_f++] = t->e[r][j];
    // data / control multiplexing with no control bits: the data in groups of Q_m, H' = the number of groups (:11901-11935)
    uint32_t H = 0;
    for (uint32_t i = 0; i < N_f; i += Q_m, H++)
        for (uint32_t n = 0; n < Q_m; n++) g[H * Q_m + n] = f[i + n];
    // channel interleaver (:11987-12099): 12 columns, R' = H' / 12 rows of symbols, written row by row -- symbol s takes the Q_m bits that START
    // AT BIT s of g, the reference's step -- and read column by column
    const uint32_t       R = (H * Q_m / 12) / Q_m;
    std::vector<uint8_t> h((size_t)12 * R * Q_m + 8), scr(h.size());
    uint32_t             N_h = 0;
    for (uint32_t col = 0; col < 12; col++)
        for (uint32_t row = 0; row < R; row++) {
            const uint32_t s = row * 12 + col;
            for (uint32_t k = 0; k < Q_m; k++) h[N_h++] = s < H ? g[s + k] : 0;
        }
    std::vector<uint8_t> c(N_h);
    synth::gold((al->rnti << 14) | (subfr_num << 9) | N_id_cell, N_h, c.data());
    for (uint32_t i = 0; i < N_h; i++) scr[i] = h[i] ^ c[i];

    // 36.211 5.3.2-5.3.3: modulation, then a DFT of M_sc points per SC-FDMA symbol scaled by (float)(1 / sqrt(M_sc))
    std::vector<float> d_re(N_h + 8), d_im(N_h + 8);
    uint32_t           M_symb = 0;
    modulate(scr.data(), N_h, al->mod_type, d_re.data(), d_im.data(), &M_symb);
    const float scale = 1 / sqrt(M_sc);
    {
        const Dft             dft(M_sc, -1);
        std::vector<double>   xr(M_sc), xi(M_sc);
        std::vector<uint32_t> all(M_sc);
        std::vector<float>    o_re(M_sc), o_im(M_sc);
        for (uint32_t j = 0; j < M_sc; j++) all[j] = j;
        for (uint32_t i = 0; i < M_symb / M_sc; i++) { // 12 / Q_m whole SC-FDMA symbols
            for (uint32_t j = 0; j < M_sc; j++) xr[j] = d_re[i * M_sc + j], xi[j] = d_im[i * M_sc + j];
            dft.run(xr.data(), xi.data(), all.data(), M_sc, o_re.data(), o_im.data());
            for (uint32_t j = 0; j < M_sc; j++) t->z_re[i * M_sc + j] = scale * o_re[j], t->z_im[i * M_sc + j] = scale * o_im[j];
        }
    }
    // resource mapping (:2754-2793): symbols 3 and 10 carry the reference signal of (subframe, N_prb), the others the data, from sub-carrier 0
    std::vector<float> rs(4 * (size_t)M_sc);
    if (mi_lte_ul_dmrs_pusch(ul, ul_cell, subfr_num, N_prb, &rs[0], &rs[M_sc], &rs[2 * M_sc], &rs[3 * M_sc]) != MI_LTE_OK) return 1;
    uint32_t idx = 0;
    for (uint32_t L = 0; L < 14; L++)
        for (uint32_t j = 0; j < M_sc; j++) {
            if (L == 3) tx_re[MI_LTE_TX_GRID_AT(0, L, j)] = rs[j], tx_im[MI_LTE_TX_GRID_AT(0, L, j)] = rs[M_sc + j];
            else if (L == 10) tx_re[MI_LTE_TX_GRID_AT(0, L, j)] = rs[2 * M_sc + j], tx_im[MI_LTE_TX_GRID_AT(0, L, j)] = rs[3 * M_sc + j];
            else tx_re[MI_LTE_TX_GRID_AT(0, L, j)] = t->z_re[idx], tx_im[MI_LTE_TX_GRID_AT(0, L, j)] = t->z_im[idx], idx++;
        }
    return 0;
}

// 36.211 5.7.2-5.7.3 as liblte_phy_generate_prach builds it (:3219-3297): preamble preamble_idx of the configuration's 64 (prach_preamble_seq_gen
// :7136-7291: root after root, the phase of x_u in double from the integer root, the values stored as float), its N_zc-point DFT placed on the PRACH
// sub-carriers of a T_fft-point inverse transform, the sequence (twice for formats 2 / 3) behind its cyclic prefix.  Writes T_cp + T_seq samples.
size_t mi_lte_generate_prach_len(uint32_t fft_size, uint32_t preamble_format)
{
    const PrachGeom pg = prach_geom(preamble_format);
    return ((size_t)pg.T_cp_30 + (size_t)pg.T_fft_30 * pg.reps) * fft_size / 2048;
}
int mi_lte_generate_prach(const mi_lte_prach_cfg *prach, uint32_t fft_size, uint32_t N_rb_ul, uint32_t N_sc_rb_ul, uint32_t preamble_idx, uint32_t freq_offset,
                          float *samps_re, float *samps_im)
{
    if (!prach || !samps_re || !samps_im || preamble_idx > 63) return 1;
    if (fft_size != 128 && fft_size != 256 && fft_size != 512 && fft_size != 1024 && fft_size != 2048) return 1;
    const uint32_t     fmt = prach->preamble_format > 4 ? 4 : prach->preamble_format;
    const PrachGeom    pg  = prach_geom(fmt);
    const PrachRootSet rs  = prach_root_set(fmt, prach->root_seq_idx, prach->zczc, prach->hs_flag != 0);
    if (!rs.ok) return 1;
    // which root and which of its cyclic shifts
    uint32_t u = 0, C_v = 0, n_gen = 0;
    bool     found = false;
    for (uint32_t r = 0; r < rs.n_roots && !found; r++) {
        const PrachSets ps = prach_sets(rs.u[r], prach->zczc, prach->hs_flag != 0, fmt);
        for (uint32_t v = 0; v <= ps.v_max; v++, n_gen++)
            if (n_gen == preamble_idx) {
                u = rs.u[r], found = true;
                C_v = prach->hs_flag ? ps.d_start * (uint32_t)floor(v / ps.N_RA_shift) + (v % ps.N_RA_shift) * ps.N_cs : v * ps.N_cs;
                break;
            }
    }
    if (!found) return 1;
    const uint32_t N_zc = pg.n_zc, T_fft = pg.T_fft_30 * fft_size / 2048, T_cp = pg.T_cp_30 * fft_size / 2048, T_seq = T_fft * pg.reps;
    std::vector<double>   xr(N_zc), xi(N_zc);
    std::vector<uint32_t> all(N_zc);
    for (uint32_t i = 0; i < N_zc; i++) {
        const uint32_t n     = (i + C_v) % N_zc;
        const double   phase = -M_PI * u * n * (n + 1) / N_zc;
        xr[i] = (float)cos(phase), xi[i] = (float)sin(phase), all[i] = i;
    }
    std::vector<float> X_re(N_zc), X_im(N_zc);
    Dft(N_zc, -1).run(xr.data(), xi.data(), all.data(), N_zc, X_re.data(), X_im.data());
    // the spectrum turned by half its length onto the T_fft grid: first PRACH sub-carrier phi + K (k_0 + 1/2), k_0 from the PRB offset
    const uint32_t K = pg.K, k_0 = freq_offset * N_sc_rb_ul - N_rb_ul * N_sc_rb_ul / 2 + fft_size / 2, start = pg.phi + K * k_0 + K / 2;
    std::vector<double>   yr(T_fft, 0.0), yi(T_fft, 0.0);
    std::vector<uint32_t> nz(N_zc);
    for (uint32_t i = 0; i < N_zc; i++) {
        const uint32_t idx = (i + start + T_fft / 2) % T_fft;
        yr[idx] = X_re[(i + N_zc / 2) % N_zc], yi[idx] = X_im[(i + N_zc / 2) % N_zc], nz[i] = idx;
    }
    // (rising index order for the sum)
    std::vector<uint32_t> order(nz);
    for (uint32_t a = 1; a < N_zc; a++) // the run of indices wraps at most once: a rotation sorts it
        if (order[a] < order[a - 1]) { std::vector<uint32_t> rot(order.begin() + a, order.end()); rot.insert(rot.end(), order.begin(), order.begin() + a); order.swap(rot); break; }
    std::vector<float> s_re(T_fft), s_im(T_fft);
    Dft(T_fft, +1).run(yr.data(), yi.data(), order.data(), N_zc, s_re.data(), s_im.data());
    for (uint32_t rep = 0; rep < pg.reps; rep++)
        memcpy(samps_re + T_cp + (size_t)rep * T_fft, s_re.data(), T_fft * sizeof(float)), memcpy(samps_im + T_cp + (size_t)rep * T_fft, s_im.data(), T_fft * sizeof(float));
    for (uint32_t i = 0; i < T_cp; i++) samps_re[i] = samps_re[T_seq + i], samps_im[i] = samps_im[T_seq + i];
    return 0;
}

} // extern "C"
