// liblte_phy_pdcch_channel_encode on the host (liblte_phy.cc:4113-4517; see tx.cc for the why): PCFICH (pcfich_channel_map :7805-7878 over
// cfi_channel_encode :13616-13639), PHICH (phich_channel_map :8076-8215) and the DCIs of the subframe's allocations -- format 1A for downlink,
// format 0 for uplink grants (dci_1a_pack :13114-13247, dci_0_pack :13054-13122), each at aggregation level 4 in the first free candidate of the
// common search space (the reference's user search space is commented out) -- CRC with the RNTI, tail-biting convolutional code, rate matching
// (dci_channel_encode :12882-12935), scrambling, QPSK, transmit diversity, the REG interleaver with the cell's cyclic shift and the mapping of
// 36.211 6.8.5.  36.211 v10.1.0 6.7-6.9, 36.212 v10.1.0 5.3.3-5.3.4.
//
// What the caller's structs get back is part of the interface and is written here as the reference writes it: pcfich->N_reg / k / n,
// phich->N_reg / k / z, pdcch->N_symbs, and the transport block size of every downlink allocation (36.213 table 7.1.7.2.1-1 at the DCI's MCS).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mi_lte.h"
#include "lte_tables.h"
#include "synth.hpp"
#include "tx_host.h"

using namespace tx;

namespace {
inline void put_bits(uint8_t *&at, uint32_t value, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) *at++ = (value >> (n - 1 - i)) & 1u;
}
// resource indication value of a contiguous run (36.213 7.1.6.3 / 8.1.1) and the width of its field, the logarithms in float as the reference takes them
inline uint32_t riv_bits(uint32_t N_rb) { return (uint32_t)ceilf(logf(N_rb * (N_rb + 1) / 2) / logf(2)); }
inline uint32_t riv(uint32_t N_rb, uint32_t N_prb, uint32_t first) { return (N_prb - 1) <= N_rb / 2 ? N_rb * (N_prb - 1) + first : N_rb * (N_rb - N_prb + 1) + (N_rb - 1 - first); }
// 36.212 5.3.3.1.2: payload sizes that are ambiguous get one more bit
inline bool ambiguous_size(uint32_t n) { return n == 12 || n == 14 || n == 16 || n == 20 || n == 24 || n == 26 || n == 32 || n == 40 || n == 44 || n == 56; }

// 36.212 5.3.3.1.3, localised allocation; an SI / paging / random-access RNTI carries N_PRB^1A = 3 in the TPC field's place and no new-data bit.
// Returns the payload size and the transport block size the DCI stands for.
uint32_t pack_1a(const mi_lte_tx_alloc &al, uint32_t N_rb_dl, uint8_t *out, uint32_t *tbs)
{
    uint8_t   *at        = out;
    const bool broadcast = al.rnti == 0xFFFF || al.rnti == 0xFFFE || (al.rnti >= 0x0001 && al.rnti <= 0x003C);
    put_bits(at, 1, 1); // format 1A
    put_bits(at, 0, 1); // localised
    put_bits(at, riv(N_rb_dl, al.N_prb, al.prb[0][0]), riv_bits(N_rb_dl));
    put_bits(at, al.mcs, 5);
    put_bits(at, 0, 3);                         // HARQ process
    put_bits(at, broadcast ? 0 : al.ndi, 1);
    put_bits(at, al.rv_idx, 2);
    put_bits(at, broadcast ? 1 : al.tpc, 2);
    *tbs = 8u * LTE_TBS_DIV8[al.mcs][(broadcast ? 3 : al.N_prb) - 1];
    if (ambiguous_size((uint32_t)(at - out))) put_bits(at, 0, 1);
    return (uint32_t)(at - out);
}
// 36.212 5.3.3.1.1, no hopping, one cluster
uint32_t pack_0(const mi_lte_tx_alloc &al, uint32_t N_rb_ul, uint8_t *out)
{
    uint8_t *at = out;
    put_bits(at, 0, 1); // format 0
    put_bits(at, 0, 1); // no hopping
    put_bits(at, riv(N_rb_ul, al.N_prb, al.prb[0][0]), riv_bits(N_rb_ul));
    put_bits(at, al.mcs, 5);
    put_bits(at, al.ndi, 1);
    put_bits(at, al.tpc, 2);
    put_bits(at, 0, 3); // cyclic shift
    put_bits(at, 0, 1); // CSI request
    put_bits(at, 0, 1); // padding
    if (ambiguous_size((uint32_t)(at - out))) put_bits(at, 0, 1);
    return (uint32_t)(at - out);
}

// REG order of 36.212 5.1.4.2.1 for n REGs (pdcch_permute_pre_calc, liblte_phy.cc:7970-8067): the sub-block interleaver of the convolutional
// rate matching applied to the REG numbers
void reg_permutation(uint32_t n, std::vector<uint32_t> &map)
{
    const uint32_t R = (n + 31) / 32, K_pi = 32 * R, N_dummy = K_pi - n;
    map.clear();
    for (uint32_t k = 0; k < K_pi && map.size() < n; k++) {
        const uint32_t col = k / R, row = k % R;
        uint32_t       j   = (col + 16) % 32; // 36.212 table 5.1.4-2 = the bit-reversed order turned by 16
        j                  = ((j & 1) << 4) | ((j & 2) << 2) | (j & 4) | ((j & 8) >> 2) | ((j & 16) >> 4);
        const uint32_t t   = 32 * row + j;
        if (t >= N_dummy) map.push_back(t - N_dummy);
    }
}
} // namespace

extern "C" int mi_lte_pdcch_channel_encode(mi_lte_tx *t, uint32_t N_rb_dl, uint32_t N_rb_ul, uint32_t N_sc_rb_dl, uint32_t N_group_phich, uint32_t N_sf_phich,
                                           mi_lte_pcfich *pcfich, mi_lte_phich *phich, mi_lte_tx_alloc *allocs, uint32_t N_alloc, uint32_t *N_pdcch_symbs, uint32_t N_id_cell,
                                           uint32_t N_ant, uint32_t phich_dur, uint32_t subfr_num, float *tx_re, float *tx_im)
{
    if (!t || !pcfich || !phich || (!allocs && N_alloc) || !N_pdcch_symbs || N_id_cell > 503 || !tx_re || !tx_im) return 1;
    if ((N_ant != 1 && N_ant != 2 && N_ant != 4) || N_sc_rb_dl != 12 || N_rb_dl < 6 || N_rb_dl > 100 || N_group_phich > 25 || N_sf_phich != 4 || N_alloc > 6) return 1;
    float         *y_re = t->ctl + mi_lte_tx::CTL_Y_RE, *y_im = t->ctl + mi_lte_tx::CTL_Y_IM, *cce_re = t->ctl + mi_lte_tx::CTL_CCE_RE, *cce_im = t->ctl + mi_lte_tx::CTL_CCE_IM;
    float          d_re[288], d_im[288], x_re[288 + 8], x_im[288 + 8];
    uint8_t        c[1152], scr[288];
    uint32_t       M_symb, M_layer, M_ap;
    const uint32_t N_sc = N_rb_dl * N_sc_rb_dl;
    auto           grid = [&](uint32_t p, uint32_t L, uint32_t k, float re, float im) { tx_re[MI_LTE_TX_GRID_AT(p, L, k)] = re, tx_im[MI_LTE_TX_GRID_AT(p, L, k)] = im; };

    // ---- PCFICH: the 32-bit code word of the CFI (36.212 table 5.3.4-1; anything but 1..3 is the reserved all-zero word), four REGs a quarter
    // of the band apart in symbol 0
    {
        const uint32_t cfi = pcfich->cfi;
        uint8_t        w[32];
        for (uint32_t i = 0; i < 32; i++) w[i] = cfi >= 1 && cfi <= 3 ? (uint8_t)(i % 3 != cfi - 1) : 0; // <0,1,1,...>, <1,0,1,...>, <1,1,0,...>
        synth::gold((((subfr_num + 1) * (2 * N_id_cell + 1)) << 9) + N_id_cell, 32, c);
        for (uint32_t i = 0; i < 32; i++) scr[i] = w[i] ^ c[i];
        modulate(scr, 32, MI_LTE_MOD_QPSK, d_re, d_im, &M_symb);
        layer_map_dl(d_re, d_im, M_symb, N_ant, 1, x_re, x_im, &M_layer);
        pre_code_dl(x_re, x_im, M_layer, N_ant, y_re, y_im, 576, &M_ap);
        pcfich->N_reg = 4;
        const uint32_t k_hat = (N_sc_rb_dl / 2) * (N_id_cell % (2 * N_rb_dl));
        for (uint32_t i = 0; i < 4; i++) {
            pcfich->k[i] = (k_hat + (i * N_rb_dl / 2) * N_sc_rb_dl / 2) % N_sc;
            pcfich->n[i] = (pcfich->k[i] / 6) - 0.5;
            for (uint32_t p = 0; p < N_ant; p++)
                for (uint32_t j = 0, idx = 0; j < 6; j++)
                    if (N_id_cell % 3 != j % 3) {
                        if (pcfich->k[i] + j >= MI_LTE_TX_GRID_SC) return 1;
                        grid(p, 0, pcfich->k[i] + j, y_re[p * 288 + idx + i * 4], y_im[p * 288 + idx + i * 4]);
                        idx++;
                    }
        }
    }

    // ---- PHICH (normal duration, normal prefix): per group the sum of its up to eight acknowledgements, each BPSK symbol repeated three times under
    // an orthogonal sequence (36.211 table 6.9.1-2: four real, four imaginary Walsh rows) and the cell's scrambling, in three REGs a third apart
    {
        static const int8_t walsh[4][4] = {{1, 1, 1, 1}, {1, -1, 1, -1}, {1, 1, -1, -1}, {1, -1, -1, 1}};
        phich->N_reg = N_group_phich * 3;
        synth::gold((((subfr_num + 1) * (2 * N_id_cell + 1)) << 9) + N_id_cell, 12, c);
        for (uint32_t m = 0, idx = 0; m < N_group_phich; m++, idx += 3) {
            for (uint32_t i = 0; i < 12; i++) d_re[i] = d_im[i] = 0;
            for (uint32_t seq = 0; seq < 8; seq++) {
                if (!phich->present[m][seq]) continue;
                const uint8_t hi = phich->b[m][seq] ? 1 : 0, bits[3] = {hi, hi, hi}; // 36.212 5.3.5: the indicator three times
                uint32_t      three;
                modulate(bits, 3, MI_LTE_MOD_BPSK, phich->z_re, phich->z_im, &three);
                for (uint32_t i = 0; i < 12; i++) {
                    const float w_re = seq < 4 ? (float)walsh[seq][i % 4] : 0.0f, w_im = seq < 4 ? 0.0f : (float)walsh[seq - 4][i % 4];
                    const float z_re = c[i] == 1 ? -phich->z_re[i / 4] : phich->z_re[i / 4], z_im = c[i] == 1 ? -phich->z_im[i / 4] : phich->z_im[i / 4];
                    d_re[i] += w_re * z_re - w_im * z_im;
                    d_im[i] += w_re * z_im + w_im * z_re;
                }
            }
            layer_map_dl(d_re, d_im, 12, N_ant, 1, x_re, x_im, &M_layer);
            pre_code_dl(x_re, x_im, M_layer, N_ant, y_re, y_im, 576, &M_ap);
            if (phich_dur != 0) continue; // (the extended duration is not mapped by the reference either)
            const uint32_t n_l = N_rb_dl * 2 - pcfich->N_reg;
            uint32_t       n_hat[3];
            for (uint32_t i = 0; i < 3; i++) n_hat[i] = (N_id_cell + m + i * n_l / 3) % n_l;
            for (uint32_t i = 0; i < pcfich->N_reg; i++) // REG numbers count the REGs the PCFICH left
                for (uint32_t j = 0; j < 3; j++)
                    if (n_hat[j] > pcfich->n[i]) n_hat[j]++;
            // (one element counter for all three REGs AND all ports: port p's values are taken from where the counter stands, liblte_phy.cc:8196-8211)
            for (uint32_t i = 0, y_idx = 0; i < 3; i++) {
                phich->k[idx + i] = n_hat[i] * 6;
                for (uint32_t p = 0; p < N_ant; p++)
                    for (uint32_t j = 0; j < 6; j++)
                        if (N_id_cell % 3 != j % 3) {
                            if (phich->k[idx + i] + j >= MI_LTE_TX_GRID_SC) return 1;
                            grid(p, 0, phich->k[idx + i] + j, y_re[p * 288 + y_idx], y_im[p * 288 + y_idx]);
                            y_idx++;
                        }
            }
        }
    }

    // ---- PDCCH
    if (N_alloc == 0) return 0;
    const uint32_t N_symbs = pcfich->cfi + (N_rb_dl <= 10 ? 1 : 0);
    *N_pdcch_symbs         = N_symbs;
    uint32_t N_reg         = N_symbs * (N_rb_dl * 3) - N_rb_dl - pcfich->N_reg - phich->N_reg;
    if (N_ant == 4) N_reg -= N_rb_dl;
    if (N_symbs < 1 || N_symbs > 4 || N_reg == 0 || N_reg > 787) return 1;
    const uint32_t N_cce = N_reg / 9;
    for (uint32_t p = 0; p < N_ant; p++)
        for (uint32_t i = 0; i < N_cce; i++) {
            for (uint32_t j = 0; j < 36; j++) cce_re[(p * 87 + i) * 36 + j] = 0, cce_im[(p * 87 + i) * 36 + j] = 0;
            t->pdcch_cce_used[i] = 0;
        }
    synth::gold((subfr_num << 9) + N_id_cell, 1152, c);
    for (uint32_t a = 0; a < N_alloc; a++) {
        mi_lte_tx_alloc &al = allocs[a];
        uint8_t          dci[64 + 16], d3[3 * (64 + 16)], e[288];
        uint32_t         n_dci;
        if (al.mcs > 26 && al.chan_type == MI_LTE_CHAN_DLSCH) return 1; // (the reference reads past its TBS table)
        if (al.chan_type == MI_LTE_CHAN_DLSCH) {
            if (al.N_prb == 0 || al.N_prb > 110) return 1;
            n_dci = pack_1a(al, N_rb_dl, dci, &al.tbs);
        } else
            n_dci = pack_0(al, N_rb_ul, dci);
        crc_bits(dci, n_dci, 0x11021, 16, dci + n_dci);
        for (uint32_t i = 0; i < 16; i++) dci[n_dci + i] ^= (al.rnti >> (15 - i)) & 1u;
        conv_encode_tb(dci, n_dci + 16, d3);
        rate_match_conv(d3, 3 * (n_dci + 16), 288, e);
        for (uint32_t css = 0; css < 4; css++) {
            if (t->pdcch_cce_used[4 * css] || t->pdcch_cce_used[4 * css + 1] || t->pdcch_cce_used[4 * css + 2] || t->pdcch_cce_used[4 * css + 3]) continue;
            for (uint32_t i = 0; i < 288; i++) scr[i] = e[i] ^ c[288 * css + i];
            modulate(scr, 288, MI_LTE_MOD_QPSK, d_re, d_im, &M_symb);
            layer_map_dl(d_re, d_im, M_symb, N_ant, 1, x_re, x_im, &M_layer);
            pre_code_dl(x_re, x_im, M_layer, N_ant, y_re, y_im, 576, &M_ap);
            for (uint32_t p = 0; p < N_ant; p++)
                for (uint32_t i = 0, idx = 0; i < 4; i++) {
                    for (uint32_t j = 0; j < 36; j++, idx++) cce_re[(p * 87 + 4 * css + i) * 36 + j] = y_re[p * 288 + idx], cce_im[(p * 87 + 4 * css + i) * 36 + j] = y_im[p * 288 + idx];
                    t->pdcch_cce_used[4 * css + i] = 1;
                }
            break;
        }
    }
    // REGs of the CCEs, interleaved (36.212 5.1.4.2.1), shifted by the cell identity, mapped symbol by symbol up the band.  The reference holds the
    // interleaver's order in a table with one row per REG count, filled for regions of one, two and three symbols of a one- or two-port cell
    // (liblte_phy.cc:7989-7992); any other count -- four ports, a fourth symbol at 1.4 MHz -- finds a row nobody filled: REG 0 everywhere.
    for (uint32_t p = 0; p < N_ant; p++)
        for (uint32_t i = 0; i < 9 * N_cce; i++)
            for (uint32_t k = 0; k < 4; k++) t->pdcch_reg_re[p][i][k] = cce_re[(p * 87 + i / 9) * 36 + (i % 9) * 4 + k], t->pdcch_reg_im[p][i][k] = cce_im[(p * 87 + i / 9) * 36 + (i % 9) * 4 + k];
    std::vector<uint32_t> map;
    bool                  known = false;
    for (uint32_t n = 1; n <= 3; n++) known = known || N_reg == n * (N_rb_dl * 3) - N_rb_dl - 4 - N_group_phich * 3;
    if (known) reg_permutation(N_reg, map);
    else map.assign(N_reg, 0);
    if (N_reg == 787 && N_ant >= 2) {
        // 787 REGs (20 MHz, three symbols, N_g = 1/6) is one more than the table has rows: that row IS the two uint16 work vectors behind the table
        // (liblte_phy.h:423-425), and a cell initialised with more than one port copies the order into it once per port (liblte_phy.cc:8059-8065) --
        // the second copy reads half-words of the first: every even entry 0, every odd entry i the order's entry (787 + i) / 2
        std::vector<uint32_t> twice(787, 0);
        for (uint32_t i = 1; i < 787; i += 2) twice[i] = map[(787 + i) / 2];
        map.swap(twice);
    }
    std::vector<float> s_re((size_t)4 * N_reg * 4), s_im(s_re.size());
    for (uint32_t p = 0; p < N_ant; p++)
        for (uint32_t i = 0; i < N_reg; i++) {
            const uint32_t src = map[(i + N_id_cell) % N_reg];
            for (uint32_t k = 0; k < 4; k++) s_re[((size_t)p * N_reg + i) * 4 + k] = t->pdcch_reg_re[p][src][k], s_im[((size_t)p * N_reg + i) * 4 + k] = t->pdcch_reg_im[p][src][k];
        }
    uint32_t m = 0;
    for (uint32_t k = 0; k < N_sc; k++)
        for (uint32_t l = 0; l < N_symbs; l++) {
            const bool six = l == 0 || (l == 1 && N_ant == 4); // symbols with reference signals: REGs of six elements, four of them free
            if (k % (six ? 6 : 4) != 0 || m >= N_reg) continue;
            if (l == 0) {
                bool taken = false;
                for (uint32_t i = 0; i < pcfich->N_reg; i++) taken = taken || k == pcfich->k[i];
                for (uint32_t i = 0; i < phich->N_reg && i < 75; i++) taken = taken || k == phich->k[i];
                if (taken) continue;
            }
            for (uint32_t i = 0, idx = 0; i < (six ? 6u : 4u); i++) {
                if (six && N_id_cell % 3 == i % 3) continue;
                for (uint32_t p = 0; p < N_ant; p++) grid(p, l, k + i, s_re[((size_t)p * N_reg + m) * 4 + idx], s_im[((size_t)p * N_reg + m) * 4 + idx]);
                idx++;
            }
            m++;
        }
    return 0;
}
