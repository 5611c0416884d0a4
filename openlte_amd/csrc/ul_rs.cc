// Uplink reference signals on the host: the PUSCH demodulation reference signal (DMRS) that the receive
// chain correlates against.  The reference builds these once per cell in liblte_phy_ul_init
// (liblte/src/liblte_phy.cc:2381-2400) by calling generate_dmrs_pusch (:6868-6990) -> generate_ul_rs
// (:6745-6860) for every (subframe, N_prb); here they are produced on demand per (cell, subframe, N_prb)
// when a PUSCH plan is created, with the reference's arithmetic kept operation for operation (float or double libm
// calls exactly where the reference's C++ overload resolution puts them, results rounded to float on store), so that on the same host
// libm the sequences agree with the reference bit for bit.  36.211 v10.1.0 sections 5.5.1 and 5.5.2.1.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mi_lte.h"
#include "lte_tables.h"
#include "synth.hpp"

namespace {

// largest prime below m (the Zadoff-Chu length N_zc^RS, 36.211 5.5.1.1); the reference scans a table of the
// primes below 2048 from the top and never looks at its first entry (liblte_phy.cc:6772-6779)
uint32_t largest_prime_below(uint32_t m)
{
    for (uint32_t c = m ? m - 1 : 0; c >= 3; c--) {
        bool prime = true;
        for (uint32_t d = 2; d * d <= c; d++)
            if (c % d == 0) { prime = false; break; }
        if (prime) return c;
    }
    return 0;
}

uint32_t bits_to_u8(const uint8_t *c) // 8 sequence bits, LSB first (liblte_phy.cc:6796-6799, :6920-6924)
{
    uint32_t v = 0;
    for (uint32_t i = 0; i < 8; i++) v += (uint32_t)c[i] << i;
    return v;
}

// r_{u,v}^{(alpha)}(n), n < 12*N_prb, for one slot (generate_ul_rs; control = chan_type ULCCH: the sequence-shift pattern without
// delta_ss, liblte_phy.cc:6782-6788)
void ul_rs_slot(const mi_lte_ul_cfg &ul, uint32_t N_slot, uint32_t N_id_cell, uint32_t N_prb, float alpha, float *rs_re, float *rs_im, bool control = false)
{
    const uint32_t M_sc = 12 * N_prb, N_zc = largest_prime_below(M_sc);
    const uint32_t f_ss = control ? N_id_cell % 30 : ((N_id_cell % 30) + ul.group_assignment_pusch) % 30;
    uint32_t       u, v = 0;
    if (ul.group_hopping_enabled) { // group hopping pattern f_gh(ns), 36.211 5.5.1.3
        uint8_t c[160];
        synth::gold(N_id_cell / 30, 160, c);
        u = (bits_to_u8(c + 8 * N_slot) % 30 + f_ss) % 30;
    } else
        u = f_ss % 30;
    if (M_sc >= 72 && !ul.group_hopping_enabled && ul.sequence_hopping_enabled) { // sequence hopping, 36.211 5.5.1.4
        uint8_t c[20];
        synth::gold(((N_id_cell / 30) << 5) + f_ss, 20, c);
        v = c[N_slot];
    }
    std::vector<float> base_re(M_sc), base_im(M_sc);
    if (M_sc >= 36) { // Zadoff-Chu base sequence, cyclically extended (36.211 5.5.1.1)
        const float q_bar = (float)N_zc * (float)(u + 1) / (float)31;
        int32_t     q;
        if ((((uint32_t)(2 * q_bar)) % 2) == 0) q = (int32_t)((uint32_t)(q_bar + 0.5) + v);
        else                                    q = (int32_t)((uint32_t)(q_bar + 0.5) - v);
        std::vector<float> zc_re(N_zc), zc_im(N_zc);
        for (uint32_t i = 0; i < N_zc; i++) {
            const double arg = -M_PI * q * i * (i + 1) / N_zc; // double throughout, (i+1) formed as an unsigned integer
            zc_re[i] = (float)std::cos(arg);
            zc_im[i] = (float)std::sin(arg);
        }
        for (uint32_t i = 0; i < M_sc; i++) { base_re[i] = zc_re[i % N_zc]; base_im[i] = zc_im[i % N_zc]; }
    } else { // computer-generated QPSK sequences for one and two resource blocks (36.211 tables 5.5.1.2-1, -2)
        for (uint32_t i = 0; i < M_sc; i++) {
            const int32_t phi = (M_sc == 12) ? LTE_UL_RS_PHI_12[u][i] : LTE_UL_RS_PHI_24[u][i % 24];
            base_re[i] = (float)std::cos(phi * M_PI / 4);
            base_im[i] = (float)std::sin(phi * M_PI / 4);
        }
    }
    for (uint32_t i = 0; i < M_sc; i++) { // cyclic shift alpha: all in float -- in C++ cos(float) is the float overload
        const float ai = alpha * i, cs = std::cos(ai), sn = std::sin(ai); // (the compiled reference calls cosf/sinf here)
        rs_re[i] = cs * base_re[i] - sn * base_im[i];
        rs_im[i] = sn * base_re[i] + cs * base_im[i];
    }
}

} // namespace

extern "C" int mi_lte_ul_dmrs_pusch(const mi_lte_ul_cfg *ul, uint32_t N_id_cell, uint32_t N_subfr, uint32_t N_prb, float *d0_re,
                                    float *d0_im, float *d1_re, float *d1_im)
{
    if (!ul || !d0_re || !d0_im || !d1_re || !d1_im || N_prb == 0 || N_prb > 110 || N_subfr > 9 || N_id_cell > 503 ||
        ul->cyclic_shift > 7 || ul->cyclic_shift_dci > 7 || ul->group_assignment_pusch > 29)
        return MI_LTE_ERR_INVALID_ARG;
    const uint32_t N_slot = 2 * N_subfr, M_sc = 12 * N_prb;
    const uint32_t f_ss   = ((N_id_cell % 30) + ul->group_assignment_pusch) % 30;
    // n_PN(ns): 8 bits of the cell's pseudo-random sequence per slot (36.211 5.5.2.1.1)
    std::vector<uint8_t> c(8 * 7 * 20);
    synth::gold(((N_id_cell / 30) << 5) + f_ss, 8 * 7 * 20, c.data());
    const uint32_t n_pn[2] = {bits_to_u8(c.data() + 8 * 7 * N_slot), bits_to_u8(c.data() + 8 * 7 * (N_slot + 1))};
    const uint32_t n1 = LTE_N1_DMRS[ul->cyclic_shift], n2 = LTE_N2_DMRS_LAMBDA[ul->cyclic_shift_dci][0];
    float         *out_re[2] = {d0_re, d1_re}, *out_im[2] = {d0_im, d1_im};
    const int32_t  w[2]      = {1, LTE_W_DMRS_LAMBDA[ul->cyclic_shift_dci][0]};
    for (int s = 0; s < 2; s++) {
        const uint32_t n_cs  = (n1 + n2 + n_pn[s]) % 12;
        const float    alpha = 2 * M_PI * n_cs / 12; // double expression rounded to float, as the reference stores it
        ul_rs_slot(*ul, N_slot + s, N_id_cell, N_prb, alpha, out_re[s], out_im[s]);
        for (uint32_t i = 0; i < M_sc; i++) { // orthogonal cover w(m)
            out_re[s][i] = w[s] * out_re[s][i];
            out_im[s][i] = w[s] * out_im[s][i];
        }
    }
    return MI_LTE_OK;
}


// The PUCCH format 1 / 1a / 1b sequences of one (subframe, resource) in the layout mi_lte_pucch_decode_run takes (MI_LTE_PUCCH_TAB_FLOATS): what
// liblte_phy_ul_init leaves in LIBLTE_PHY_STRUCT through generate_dmrs_pucch (liblte_phy.cc:2401-2421, :6986-7129; 36.211 v10.1.0 sections
// 5.4.1 and 5.5.2.2) and what the decoder derives from it (s(n_s) and the orthogonal cover, :3058-3083).  Integer steps as the reference
// takes them (its unsigned arithmetic included), floating point where its overloads put it.
extern "C" int mi_lte_ul_pucch_tables(const mi_lte_ul_cfg *ul, uint32_t N_id_cell, uint32_t N_subfr, uint32_t N_1_p_pucch, uint32_t N_cs_an,
                                      uint32_t delta_pucch_shift, uint32_t N_ant, float *t)
{
    if (!ul || !t || N_subfr > 9 || N_id_cell > 503 || N_1_p_pucch > 255 || N_cs_an > 7 || delta_pucch_shift < 1 || delta_pucch_shift > 12 ||
        !(N_ant == 1 || N_ant == 2 || N_ant == 4) || ul->group_assignment_pusch > 29)
        return MI_LTE_ERR_INVALID_ARG;
    const uint32_t N_slot = 2 * N_subfr, n1 = N_1_p_pucch & 0xFFu, dps = delta_pucch_shift, N_cs_1 = N_cs_an;
    const bool     mixed = n1 < 3 * N_cs_1 / dps; // a resource in the mixed resource block
    const uint32_t N_prime = mixed ? N_cs_1 : 12u;
    uint32_t       n_prime[2], n_oc[2];
    if (mixed) {
        n_prime[0] = n1;
        const uint32_t h_p = (n_prime[0] + 2) % (3 * N_prime / dps);
        n_prime[1] = h_p / 3 + (h_p % 3) * N_prime / dps;
    } else {
        n_prime[0] = (n1 - 3 * N_cs_1 / dps) % (3 * 12 / dps);
        n_prime[1] = ((3 * (n_prime[0] + 1)) % ((3 * 12 / dps) + 1)) - 1; // (unsigned: a remainder of 0 wraps, as in the reference)
    }
    for (int i = 0; i < 2; i++) n_oc[i] = n_prime[i] * dps / N_prime;
    // n_cs^cell(n_s, l): 8 bits of the cell's sequence per symbol (36.211 5.4)
    std::vector<uint8_t> c(8 * 7 * 20);
    synth::gold(N_id_cell, 8 * 7 * 20, c.data());
    float r_re[2][7][12], r_im[2][7][12];
    for (uint32_t i = 0; i < 2; i++)
        for (uint32_t j = 0; j < 7; j++) {
            const uint32_t n_cs_cell = bits_to_u8(c.data() + 8 * 7 * (N_slot + i) + 8 * j);
            const uint32_t n_cs_p    = (n_cs_cell + ((n_prime[i] * dps + (n_oc[i] % dps)) % N_prime)) % 12;
            const float    alpha     = 2 * M_PI * n_cs_p / 12; // double expression rounded to float, as the reference stores it
            ul_rs_slot(*ul, N_slot + i, N_id_cell, 1, alpha, r_re[i][j], r_im[i][j], true);
        }
    // DMRS: symbols 2, 3, 4 of each slot with the cover w(m) = exp(i phase) of 36.211 table 5.5.2.2.1-2, z(m) = 1
    static const float W_PHASE[3][3] = {{0, 0, 0}, {0, (float)(2 * M_PI / 3), (float)(4 * M_PI / 3)}, {0, (float)(4 * M_PI / 3), (float)(2 * M_PI / 3)}};
    for (uint32_t i = 0; i < 2; i++) {
        float *d_re = t + 72 * i, *d_im = d_re + 36;
        for (uint32_t j = 0; j < 3; j++) {
            const float w_re = std::cos(W_PHASE[n_oc[i] % 3][j]), w_im = std::sin(W_PHASE[n_oc[i] % 3][j]); // (float overloads, like the reference's calls)
            for (uint32_t k = 0; k < 12; k++) {
                const float a_re = r_re[i][j + 2][k], a_im = r_im[i][j + 2][k];
                d_re[12 * j + k] = (float)((1 / std::sqrt((double)N_ant)) * (w_re * a_re - w_im * a_im));
                d_im[12 * j + k] = (float)((1 / std::sqrt((double)N_ant)) * (w_re * a_im + w_im * a_re));
            }
        }
    }
    // data symbols 0, 1, 5, 6 of each slot: r_u_v and s(n_s) w(i)
    static const uint32_t symb[4] = {0, 1, 5, 6};
    static const int32_t  W4[3][4] = {{1, 1, 1, 1}, {1, -1, 1, -1}, {1, -1, -1, 1}}; // 36.211 table 5.4.1-2
    for (uint32_t m = 0; m < 2; m++) {
        float s_re, s_im;
        if ((n_prime[m] % 2) == 0) { s_re = 1; s_im = 0; }
        else { s_re = (float)std::cos(M_PI / 2); s_im = (float)std::sin(M_PI / 2); }
        for (uint32_t i = 0; i < 4; i++) {
            for (uint32_t k = 0; k < 12; k++) {
                t[144 + (m * 4 + i) * 12 + k] = r_re[m][symb[i]][k];
                t[240 + (m * 4 + i) * 12 + k] = r_im[m][symb[i]][k];
            }
            t[336 + m * 4 + i] = s_re * W4[n_oc[m] % 3][i];
            t[344 + m * 4 + i] = s_im * W4[n_oc[m] % 3][i];
        }
    }
    return MI_LTE_OK;
}
