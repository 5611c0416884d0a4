// tx_check.cc -- `lifecycle_check tx` (CPU only): the transmit functions the own-lifecycle build of the shim defines (openlte_amd/csrc/tx*.cc behind
// liblte_phy_shim.cc) against the reference's, linked into this TEST binary under the names *_cpu (shim/Makefile: phy_renamed_all.o).  Both sides
// get the same calls in the same order on structs of their own -- the reference's transmit functions read scratch that earlier calls left in the
// struct, so the ORDER is part of the input -- and must leave the same grids (compared bit for bit as 32-bit words), bits and return codes.  The
// two transforms (OFDM modulation, PRACH) are compared to float rounding: the reference's run on this box's FFTW stand-in (oracle/ref/fftw_shim.c).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "liblte_phy.h"
#include "liblte_rrc.h"

LIBLTE_ERROR_ENUM liblte_phy_init_cpu(LIBLTE_PHY_STRUCT **phy_struct, LIBLTE_PHY_FS_ENUM fs, uint16 N_id_cell, uint8 N_ant, uint32 N_rb_dl, uint32 N_sc_rb_dl, float phich_res);
LIBLTE_ERROR_ENUM liblte_phy_cleanup_cpu(LIBLTE_PHY_STRUCT *phy_struct);
void              liblte_phy_rate_match_turbo_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint8 *d_bits, uint32 N_d_bits, uint32 N_codeblocks, uint32 tx_mode, uint32 N_soft, uint32 M_dl_harq,
                                                  LIBLTE_PHY_CHAN_TYPE_ENUM chan_type, uint32 rv_idx, uint32 N_e_bits, uint8 *e_bits);
LIBLTE_ERROR_ENUM liblte_phy_pdsch_channel_encode_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_PDCCH_STRUCT *pdcch, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe);
LIBLTE_ERROR_ENUM liblte_phy_bch_channel_encode_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint8 *in_bits, uint32 N_in_bits, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe,
                                                    uint32 sfn);
LIBLTE_ERROR_ENUM liblte_phy_map_crs_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_cell, uint8 N_ant);
LIBLTE_ERROR_ENUM liblte_phy_map_pss_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_2, uint8 N_ant);
LIBLTE_ERROR_ENUM liblte_phy_map_sss_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint32 N_id_1, uint32 N_id_2, uint8 N_ant);
LIBLTE_ERROR_ENUM liblte_phy_create_dl_subframe_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, uint8 ant, float *i_samps, float *q_samps);
LIBLTE_ERROR_ENUM liblte_phy_pdcch_channel_encode_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_PCFICH_STRUCT *pcfich, LIBLTE_PHY_PHICH_STRUCT *phich, LIBLTE_PHY_PDCCH_STRUCT *pdcch,
                                                      uint32 N_id_cell, uint8 N_ant, float phich_res, LIBLTE_RRC_PHICH_DURATION_ENUM phich_dur, LIBLTE_PHY_SUBFRAME_STRUCT *subframe);
LIBLTE_ERROR_ENUM liblte_phy_ul_init_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint16 N_id_cell, uint32 prach_root_seq_idx, uint32 prach_preamble_format, uint32 prach_zczc,
                                         bool prach_hs_flag, uint8 group_assignment_pusch, bool group_hopping_enabled, bool sequence_hopping_enabled, uint8 cyclic_shift,
                                         uint8 cyclic_shift_dci, uint8 N_cs_an, uint8 delta_pucch_shift);
LIBLTE_ERROR_ENUM liblte_phy_pusch_channel_encode_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_ALLOCATION_STRUCT *alloc, uint32 N_id_cell, uint8 N_ant, LIBLTE_PHY_SUBFRAME_STRUCT *subframe);
LIBLTE_ERROR_ENUM liblte_phy_generate_prach_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint32 preamble_idx, uint32 freq_offset, float *samps_re, float *samps_im);
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_and_n_prb_for_dl_cpu(uint32 N_bits, uint32 N_rb_dl, uint8 mcs, uint32 *tbs, uint32 *N_prb);

namespace {
uint32 g_x = 2463534242u;
uint32 rnd() { g_x ^= g_x << 13; g_x ^= g_x >> 17; g_x ^= g_x << 5; return g_x; }
uint32 rnd(uint32 n) { return rnd() % n; }

long g_n = 0, g_bad = 0;
#define CHECK(cond, ...) do { g_n++; if (!(cond)) { if (g_bad < 30) { printf("  " __VA_ARGS__); printf("\n"); } g_bad++; } } while (0)

LIBLTE_PHY_SUBFRAME_STRUCT g_s1, g_s2;
bool grids_equal(uint32 *first_p, uint32 *first_l, uint32 *first_k)
{
    if (!memcmp(g_s1.tx_symb_re, g_s2.tx_symb_re, sizeof g_s1.tx_symb_re) && !memcmp(g_s1.tx_symb_im, g_s2.tx_symb_im, sizeof g_s1.tx_symb_im)) return true;
    for (uint32 p = 0; p < 4; p++)
        for (uint32 l = 0; l < 16; l++)
            for (uint32 k = 0; k < 1200; k++)
                if (memcmp(&g_s1.tx_symb_re[p][l][k], &g_s2.tx_symb_re[p][l][k], 4) || memcmp(&g_s1.tx_symb_im[p][l][k], &g_s2.tx_symb_im[p][l][k], 4)) {
                    *first_p = p, *first_l = l, *first_k = k;
                    return false;
                }
    return false;
}
void fill_grids(uint32 sf)
{
    // the same arbitrary contents on both sides: what a function leaves alone is compared too
    for (uint32 p = 0; p < 4; p++)
        for (uint32 l = 0; l < 16; l++)
            for (uint32 k = 0; k < 1200; k++) {
                const float v = (float)((p * 16 + l) * 1200 + k) * 1e-3f;
                g_s1.tx_symb_re[p][l][k] = g_s2.tx_symb_re[p][l][k] = v;
                g_s1.tx_symb_im[p][l][k] = g_s2.tx_symb_im[p][l][k] = -v;
            }
    g_s1.num = g_s2.num = sf;
}

struct Pair {
    LIBLTE_PHY_STRUCT *own = NULL, *ref = NULL;
    bool open(LIBLTE_PHY_FS_ENUM fs, uint16 cell, uint8 n_ant, uint32 n_rb)
    {
        if (liblte_phy_init(&own, fs, cell, n_ant, n_rb, 12, 1.0f) != LIBLTE_SUCCESS || liblte_phy_init_cpu(&ref, fs, cell, n_ant, n_rb, 12, 1.0f) != LIBLTE_SUCCESS) return false;
        // scratch that a transmit call can read before it has written it: from zeros on both sides (a fresh struct is zero pages in practice)
        memset(ref->pdsch_d_re, 0, sizeof ref->pdsch_d_re), memset(ref->pdsch_d_im, 0, sizeof ref->pdsch_d_im);
        memset(ref->pdsch_x_re, 0, sizeof ref->pdsch_x_re), memset(ref->pdsch_x_im, 0, sizeof ref->pdsch_x_im);
        memset(ref->pdsch_y_re, 0, sizeof ref->pdsch_y_re), memset(ref->pdsch_y_im, 0, sizeof ref->pdsch_y_im);
        memset(ref->dlsch_tx_e_bits, 0, sizeof ref->dlsch_tx_e_bits), memset(ref->dlsch_c_bits, 0, sizeof ref->dlsch_c_bits);
        ref->bch_N_bits = 0;
        return true;
    }
    void close() { liblte_phy_cleanup(own), liblte_phy_cleanup_cpu(ref); }
};
const struct { LIBLTE_PHY_FS_ENUM fs; uint32 n_rb; } BW[6] = {{LIBLTE_PHY_FS_1_92MHZ, 6},   {LIBLTE_PHY_FS_3_84MHZ, 15},  {LIBLTE_PHY_FS_7_68MHZ, 25},
                                                             {LIBLTE_PHY_FS_15_36MHZ, 50}, {LIBLTE_PHY_FS_30_72MHZ, 75}, {LIBLTE_PHY_FS_30_72MHZ, 100}};

void check_rate_match()
{
    Pair P;
    if (!P.open(LIBLTE_PHY_FS_30_72MHZ, 1, 1, 100)) { g_bad++; return; }
    static const uint32 Ks[14] = {40, 48, 104, 512, 520, 1008, 1056, 2048, 2112, 3264, 4160, 5056, 6080, 6144};
    static uint8 d[3 * 6148], e1[40000], e2[40000];
    for (int k = 0; k < 14; k++)
        for (uint32 trial = 0; trial < 24; trial++) {
            const uint32 D = Ks[k] + 4, F = trial % 3 == 2 ? rnd(Ks[k] < 64 ? 8 : 56) : 0;
            for (uint32 i = 0; i < 3 * D; i++) d[i] = (uint8)(rnd() & 1);
            for (uint32 i = 0; i < F; i++) d[i] = 100; // filler <NULL>s in front of the systematic stream
            const uint32 rv = trial % 4, C = 1 + rnd(3), tx_mode = 1 + rnd(9), harq = 1 + rnd(10);
            const uint32 N_soft = trial % 5 == 4 ? 3 * D * C * (2 + rnd(3)) / 3 + 64 : 250368; // a circular buffer shorter than the code word now and then
            const LIBLTE_PHY_CHAN_TYPE_ENUM ct = (LIBLTE_PHY_CHAN_TYPE_ENUM)(trial % 7 == 6 ? 2 : trial % 2);
            const uint32 E = 1 + rnd(trial % 2 ? 3 * D * 2 : 3 * D);
            memset(e1, 9, sizeof e1), memset(e2, 9, sizeof e2);
            liblte_phy_rate_match_turbo(P.own, d, 3 * D, C, tx_mode, N_soft, harq, ct, rv, E, e1);
            liblte_phy_rate_match_turbo_cpu(P.ref, d, 3 * D, C, tx_mode, N_soft, harq, ct, rv, E, e2);
            CHECK(!memcmp(e1, e2, sizeof e1), "rate_match_turbo(K %u, F %u, C %u, mode %u, N_soft %u, harq %u, chan %d, rv %u, E %u)", Ks[k], F, C, tx_mode, N_soft, harq, (int)ct, rv, E);
        }
    P.close();
}

void check_signals()
{
    for (int b = 0; b < 6; b++)
        for (uint32 n_ant = 1; n_ant <= 4; n_ant *= 2) {
            Pair P;
            if (!P.open(BW[b].fs, (uint16)(3 * b + 1), (uint8)n_ant, BW[b].n_rb)) { g_bad++; return; }
            uint32 p = 0, l = 0, k = 0;
            for (uint32 cell = 0; cell < 504; cell += (b == 5 ? 1 : 7))
                for (uint32 sf = 0; sf < 10; sf++) {
                    fill_grids(sf);
                    const LIBLTE_ERROR_ENUM a1 = liblte_phy_map_crs(P.own, &g_s1, cell, (uint8)n_ant), a2 = liblte_phy_map_crs_cpu(P.ref, &g_s2, cell, (uint8)n_ant);
                    const LIBLTE_ERROR_ENUM b1 = liblte_phy_map_pss(P.own, &g_s1, cell % 3, (uint8)n_ant), b2 = liblte_phy_map_pss_cpu(P.ref, &g_s2, cell % 3, (uint8)n_ant);
                    const LIBLTE_ERROR_ENUM c1 = liblte_phy_map_sss(P.own, &g_s1, cell / 3, cell % 3, (uint8)n_ant), c2 = liblte_phy_map_sss_cpu(P.ref, &g_s2, cell / 3, cell % 3, (uint8)n_ant);
                    CHECK(a1 == a2 && b1 == b2 && c1 == c2 && grids_equal(&p, &l, &k), "map_crs / _pss / _sss(N_rb_dl %u, %u ports, cell %u, subframe %u): codes %d %d %d vs %d %d %d, port %u symbol %u sub-carrier %u",
                          BW[b].n_rb, n_ant, cell, sf, a1, b1, c1, a2, b2, c2, p, l, k);
                }
            // the cell the struct was made for goes through the reference's stored sequences, the others through its generator: both seen above
            CHECK(liblte_phy_map_crs(P.own, NULL, 1, 1) == liblte_phy_map_crs_cpu(P.ref, NULL, 1, 1) && liblte_phy_map_crs(P.own, &g_s1, 504, 1) == liblte_phy_map_crs_cpu(P.ref, &g_s2, 504, 1) &&
                      liblte_phy_map_pss(NULL, &g_s1, 0, 1) == liblte_phy_map_pss_cpu(NULL, &g_s2, 0, 1) && liblte_phy_map_sss(P.own, NULL, 0, 0, 1) == liblte_phy_map_sss_cpu(P.ref, NULL, 0, 0, 1),
                  "map_*: argument checks");
            P.close();
        }
}

void check_bch()
{
    for (int b = 0; b < 6; b += 1)
        for (uint32 n_ant = 1; n_ant <= 4; n_ant *= 2) {
            Pair P;
            if (!P.open(BW[b].fs, 5, (uint8)n_ant, BW[b].n_rb)) { g_bad++; return; }
            uint32 p = 0, l = 0, k = 0;
            for (uint32 cell = b; cell < 504; cell += 41) {
                uint8 mib[24];
                // a caller that starts in the middle of a 40 ms period, runs over three of them and changes the MIB with every frame
                for (uint32 sfn = 2 + cell % 3; sfn < 14; sfn++) {
                    for (int i = 0; i < 24; i++) mib[i] = (uint8)(rnd() & 1);
                    fill_grids(0);
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_bch_channel_encode(P.own, mib, 24, cell, (uint8)n_ant, &g_s1, sfn);
                    const LIBLTE_ERROR_ENUM e2 = liblte_phy_bch_channel_encode_cpu(P.ref, mib, 24, cell, (uint8)n_ant, &g_s2, sfn);
                    CHECK(e1 == e2 && grids_equal(&p, &l, &k), "bch_channel_encode(N_rb_dl %u, %u ports, cell %u, sfn %u): %d vs %d, port %u symbol %u sub-carrier %u", BW[b].n_rb, n_ant, cell, sfn,
                          e1, e2, p, l, k);
                }
                // (leave the period closed on both sides before the cell changes: the reference keeps the old cell's scrambling otherwise -- so does the handle)
            }
            P.close();
        }
}

void check_pdsch()
{
    static LIBLTE_PHY_PDCCH_STRUCT pd;
    for (int b = 0; b < 6; b++)
        for (uint32 n_ant = 1; n_ant <= 4; n_ant *= 2) {
            Pair P;
            if (!P.open(BW[b].fs, 9, (uint8)n_ant, BW[b].n_rb)) { g_bad++; return; }
            uint32 p = 0, l = 0, k = 0;
            const uint32 n_rb = BW[b].n_rb;
            for (uint32 trial = 0; trial < 160; trial++) {
                memset(&pd, 0, sizeof pd);
                const uint32 sf = trial % 4 == 0 ? 0 : trial % 4 == 1 ? 5 : rnd(10), cell = rnd(504);
                pd.N_symbs = 1 + rnd(n_rb == 6 ? 4 : 3);
                pd.N_alloc = 1 + rnd(3);
                uint32 next_prb = rnd(3);
                for (uint32 a = 0; a < pd.N_alloc; a++) {
                    LIBLTE_PHY_ALLOCATION_STRUCT &al = pd.alloc[a];
                    al.mod_type       = (LIBLTE_PHY_MODULATION_TYPE_ENUM)(trial % 9 == 8 ? 0 : 1 + rnd(3));
                    al.pre_coder_type = LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY;
                    al.chan_type      = trial % 13 == 12 && a == 1 ? LIBLTE_PHY_CHAN_TYPE_ULSCH : LIBLTE_PHY_CHAN_TYPE_DLSCH; // (not a downlink allocation: skipped)
                    al.N_codewords    = trial % 11 == 10 ? 2 : 1;
                    al.tx_mode        = 1 + rnd(4);
                    al.rv_idx         = rnd(4);
                    al.rnti           = (uint16)(1 + rnd(65000));
                    // a run of PRBs, the second slot's list its own now and then (the mapping reads it; the pricing reads slot 0's)
                    const uint32 q = al.mod_type == LIBLTE_PHY_MODULATION_TYPE_64QAM ? 6 : al.mod_type == LIBLTE_PHY_MODULATION_TYPE_16QAM ? 4 : 2;
                    uint32       want = 1 + rnd(n_rb == 6 ? 3 : 9) * (al.N_codewords == 2 ? 1 : 1);
                    if (want * 150 * q * al.N_codewords > 9000) want = 9000 / (150 * q * al.N_codewords);
                    if (n_ant == 4 && al.mod_type == LIBLTE_PHY_MODULATION_TYPE_BPSK && want > 4) want = 4; // (4 ports x 5000 symbols: the reference's arrays)
                    if (want == 0) want = 1;
                    if (next_prb + want > n_rb) { pd.N_alloc = a; break; }
                    al.N_prb = want;
                    for (uint32 i = 0; i < want; i++) al.prb[0][i] = next_prb + i, al.prb[1][i] = trial % 5 == 4 ? n_rb - 1 - (next_prb + i) : next_prb + i;
                    next_prb += want + rnd(2);
                    // a transport block: from the table (one code block), an arbitrary size with filler bits, or -- now and then -- several code blocks
                    uint32 tbs = 0, n_prb_out = 0;
                    liblte_phy_get_tbs_and_n_prb_for_dl_cpu(16 + rnd(q * 100 * want), n_rb, (uint8)rnd(27), &tbs, &n_prb_out);
                    if (tbs == 0 || tbs > 6120) tbs = 16 + 8 * rnd(700);
                    if (trial % 7 == 3) tbs = 17 + rnd(3000);
                    if (trial % 17 == 16) tbs = 6200 + rnd(9000);
                    al.tbs = tbs;
                    for (uint32 cw = 0; cw < al.N_codewords; cw++) {
                        al.msg[cw].N_bits = trial % 6 == 5 ? tbs - rnd(tbs < 16 ? 1 : 16) : tbs; // (shorter than the block: zero-padded)
                        if (al.msg[cw].N_bits > LIBLTE_MAX_MSG_SIZE) al.msg[cw].N_bits = LIBLTE_MAX_MSG_SIZE;
                        for (uint32 i = 0; i < al.msg[cw].N_bits; i++) al.msg[cw].msg[i] = (uint8)(rnd() & 1);
                    }
                }
                if (pd.N_alloc == 0) continue;
                fill_grids(sf);
                if (getenv("TX_TRACE")) printf("pdsch %u %u %u: sf %u allocs %u mod %d cw %u tbs %u nprb %u\n", n_rb, n_ant, trial, sf, pd.N_alloc, (int)pd.alloc[0].mod_type, pd.alloc[0].N_codewords, pd.alloc[0].tbs, pd.alloc[0].N_prb);
                const LIBLTE_ERROR_ENUM e1 = liblte_phy_pdsch_channel_encode(P.own, &pd, cell, (uint8)n_ant, &g_s1);
                if (getenv("TX_TRACE")) printf("  own done\n");
                const LIBLTE_ERROR_ENUM e2 = liblte_phy_pdsch_channel_encode_cpu(P.ref, &pd, cell, (uint8)n_ant, &g_s2);
                CHECK(e1 == e2 && grids_equal(&p, &l, &k), "pdsch_channel_encode(N_rb_dl %u, %u ports, trial %u: subframe %u, cell %u, %u allocations, first: mod %d N_prb %u tbs %u cw %u): %d vs %d, port %u symbol %u sub-carrier %u",
                      n_rb, n_ant, trial, sf, cell, pd.N_alloc, (int)pd.alloc[0].mod_type, pd.alloc[0].N_prb, pd.alloc[0].tbs, pd.alloc[0].N_codewords, e1, e2, p, l, k);
            }
            CHECK(liblte_phy_pdsch_channel_encode(P.own, NULL, 1, 1, &g_s1) == liblte_phy_pdsch_channel_encode_cpu(P.ref, NULL, 1, 1, &g_s2) &&
                      liblte_phy_pdsch_channel_encode(P.own, &pd, 504, 1, &g_s1) == liblte_phy_pdsch_channel_encode_cpu(P.ref, &pd, 504, 1, &g_s2) &&
                      liblte_phy_pdsch_channel_encode(P.own, &pd, 1, 1, NULL) == liblte_phy_pdsch_channel_encode_cpu(P.ref, &pd, 1, 1, NULL),
                  "pdsch_channel_encode: argument checks");
            P.close();
        }
}

void check_ofdm()
{
    static float i1[30720], q1[30720], i2[30720], q2[30720];
    double worst = 0;
    for (int b = 0; b < 6; b++) {
        Pair P;
        if (!P.open(BW[b].fs, 9, 2, BW[b].n_rb)) { g_bad++; return; }
        fill_grids(3);
        for (uint32 pp = 0; pp < 2; pp++)
            for (uint32 l = 0; l < 14; l++)
                for (uint32 k = 0; k < 12 * BW[b].n_rb; k++) {
                    g_s1.tx_symb_re[pp][l][k] = g_s2.tx_symb_re[pp][l][k] = (float)((int)rnd(15) - 7) * 0.154303f;
                    g_s1.tx_symb_im[pp][l][k] = g_s2.tx_symb_im[pp][l][k] = (float)((int)rnd(15) - 7) * 0.154303f;
                }
        for (uint8 ant = 0; ant < 2; ant++) {
            memset(i1, 0, sizeof i1), memset(q1, 0, sizeof q1), memset(i2, 0, sizeof i2), memset(q2, 0, sizeof q2);
            const LIBLTE_ERROR_ENUM e1 = liblte_phy_create_dl_subframe(P.own, &g_s1, ant, i1, q1), e2 = liblte_phy_create_dl_subframe_cpu(P.ref, &g_s2, ant, i2, q2);
            double num = 0, den = 0;
            for (uint32 n = 0; n < P.ref->N_samps_per_subfr; n++) {
                num += (double)(i1[n] - i2[n]) * (i1[n] - i2[n]) + (double)(q1[n] - q2[n]) * (q1[n] - q2[n]);
                den += (double)i2[n] * i2[n] + (double)q2[n] * q2[n];
            }
            const double rel = sqrt(num / (den > 0 ? den : 1));
            if (rel > worst) worst = rel;
            CHECK(e1 == e2 && den > 0 && rel < 2e-7, "create_dl_subframe(N_rb_dl %u, port %u): %d vs %d, relative L2 of the difference %.3g", BW[b].n_rb, ant, e1, e2, rel);
        }
        CHECK(liblte_phy_create_dl_subframe(P.own, &g_s1, 0, NULL, q1) == liblte_phy_create_dl_subframe_cpu(P.ref, &g_s2, 0, NULL, q2), "create_dl_subframe: argument checks");
        P.close();
    }
    printf("  create_dl_subframe: worst relative L2 against the reference on the float64 transform stand-in %.3g\n", worst);
}
void check_pdcch()
{
    static LIBLTE_PHY_PDCCH_STRUCT  pd1, pd2;
    static LIBLTE_PHY_PCFICH_STRUCT pc1, pc2;
    static LIBLTE_PHY_PHICH_STRUCT  ph1, ph2;
    const float res[4] = {1.0f / 6, 0.5f, 1.0f, 2.0f};
    for (int b = 0; b < 6; b++)
        for (uint32 n_ant = 1; n_ant <= 4; n_ant *= 2)
            for (int g = 0; g < 4; g++) {
                if (BW[b].n_rb == 6 && g == 3) continue; // (the reference's own init crashes there: lifecycle_check.cc)
                LIBLTE_PHY_STRUCT *own = NULL, *ref = NULL;
                if (liblte_phy_init(&own, BW[b].fs, 7, (uint8)n_ant, BW[b].n_rb, 12, res[g]) != LIBLTE_SUCCESS || liblte_phy_init_cpu(&ref, BW[b].fs, 7, (uint8)n_ant, BW[b].n_rb, 12, res[g]) != LIBLTE_SUCCESS) { g_bad++; return; }
                // (the control region's scratch from zeros on both sides, see Pair::open)
                memset(ref->pdcch_y_re, 0, sizeof ref->pdcch_y_re), memset(ref->pdcch_y_im, 0, sizeof ref->pdcch_y_im), memset(ref->pdcch_cce_re, 0, sizeof ref->pdcch_cce_re);
                memset(ref->pdcch_cce_im, 0, sizeof ref->pdcch_cce_im), memset(ref->pdcch_reg_re, 0, sizeof ref->pdcch_reg_re), memset(ref->pdcch_reg_im, 0, sizeof ref->pdcch_reg_im);
                memset(ref->pdcch_cce_used, 0, sizeof ref->pdcch_cce_used);
                uint32 p = 0, l = 0, k = 0;
                for (uint32 trial = 0; trial < 40; trial++) {
                    memset(&pd1, 0, sizeof pd1), memset(&pc1, 0, sizeof pc1), memset(&ph1, 0, sizeof ph1);
                    const uint32 sf = rnd(10), cell = rnd(504);
                    pc1.cfi = 1 + rnd(3);
                    // (a region without a single REG left for the PDCCH -- four ports, one symbol, many PHICH groups -- sends the reference's unsigned count round: not called)
                    auto regs = [&](uint32 cfi) { return (int)((cfi + (BW[b].n_rb <= 10 ? 1 : 0)) * 3 * BW[b].n_rb) - (int)BW[b].n_rb - 4 - 3 * (int)own->N_group_phich - (n_ant == 4 ? (int)BW[b].n_rb : 0); };
                    while (pc1.cfi < 3 && regs(pc1.cfi) <= 0) pc1.cfi++;
                    if (regs(pc1.cfi) <= 0) continue;
                    for (uint32 m = 0; m < own->N_group_phich && m < 25; m++)
                        for (uint32 q = 0; q < 8; q++) ph1.present[m][q] = rnd(3) == 0, ph1.b[m][q] = (uint8)(rnd() & 1);
                    pd1.N_alloc = trial % 8 == 7 ? 0 : 1 + rnd(trial % 5 == 4 ? 6 : 3); // (more than four: the common search space runs out, the rest is dropped)
                    pd1.N_symbs = 77;
                    for (uint32 a = 0; a < pd1.N_alloc; a++) {
                        LIBLTE_PHY_ALLOCATION_STRUCT &al = pd1.alloc[a];
                        static const uint16 special[4] = {0xFFFF, 0xFFFE, 0x0001, 0x003C};
                        al.chan_type = rnd(4) == 0 ? LIBLTE_PHY_CHAN_TYPE_ULSCH : LIBLTE_PHY_CHAN_TYPE_DLSCH;
                        al.rnti      = rnd(3) == 0 ? special[rnd(4)] : (uint16)(0x3D + rnd(60000));
                        al.N_prb     = 1 + rnd(BW[b].n_rb);
                        al.prb[0][0] = rnd(BW[b].n_rb - al.N_prb + 1);
                        al.mcs = (uint8)rnd(27), al.ndi = rnd() & 1, al.tpc = (uint8)rnd(4), al.rv_idx = rnd(4), al.tbs = 4242;
                    }
                    pd2 = pd1, pc2 = pc1, ph2 = ph1;
                    fill_grids(sf);
                    if (getenv("TX_TRACE")) printf("pdcch %u %u %d trial %u: cfi %u allocs %u groups %u\n", BW[b].n_rb, n_ant, g, trial, pc1.cfi, pd1.N_alloc, own->N_group_phich);
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_pdcch_channel_encode(own, &pc1, &ph1, &pd1, cell, (uint8)n_ant, res[g], LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s1);
                    if (getenv("TX_TRACE")) printf("  own done\n");
                    const LIBLTE_ERROR_ENUM e2 = liblte_phy_pdcch_channel_encode_cpu(ref, &pc2, &ph2, &pd2, cell, (uint8)n_ant, res[g], LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s2);
                    const bool structs = !memcmp(&pc1, &pc2, sizeof pc1) && !memcmp(&ph1, &ph2, sizeof ph1) && !memcmp(&pd1, &pd2, sizeof pd1);
                    CHECK(e1 == e2 && structs && grids_equal(&p, &l, &k), "pdcch_channel_encode(N_rb_dl %u, %u ports, phich %.3f, trial %u: subframe %u, cell %u, cfi %u, %u allocations): %d vs %d, structs %s (N_symbs %u / %u, tbs %u / %u), port %u symbol %u sub-carrier %u",
                          BW[b].n_rb, n_ant, res[g], trial, sf, cell, pc1.cfi, pd1.N_alloc, e1, e2, structs ? "equal" : "DIFFER", pd1.N_symbs, pd2.N_symbs, pd1.alloc[0].tbs, pd2.alloc[0].tbs, p, l, k);
                }
                CHECK(liblte_phy_pdcch_channel_encode(own, NULL, &ph1, &pd1, 1, 1, 1, LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s1) == liblte_phy_pdcch_channel_encode_cpu(ref, NULL, &ph2, &pd2, 1, 1, 1, LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s2) &&
                          liblte_phy_pdcch_channel_encode(own, &pc1, &ph1, &pd1, 504, 1, 1, LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s1) == liblte_phy_pdcch_channel_encode_cpu(ref, &pc2, &ph2, &pd2, 504, 1, 1, LIBLTE_RRC_PHICH_DURATION_NORMAL, &g_s2),
                      "pdcch_channel_encode: argument checks");
                liblte_phy_cleanup(own), liblte_phy_cleanup_cpu(ref);
            }
}

// relative L2 distance of two grids / sample vectors (the transforms on the reference's side run on the FFTW stand-in)
double rel_l2(const float *a_re, const float *a_im, const float *b_re, const float *b_im, size_t n)
{
    double num = 0, den = 0;
    for (size_t i = 0; i < n; i++) {
        num += (double)(a_re[i] - b_re[i]) * (a_re[i] - b_re[i]) + (double)(a_im[i] - b_im[i]) * (a_im[i] - b_im[i]);
        den += (double)b_re[i] * b_re[i] + (double)b_im[i] * b_im[i];
    }
    return den > 0 ? sqrt(num / den) : (num > 0 ? 1.0 : 0.0);
}

void check_pusch()
{
    static LIBLTE_PHY_ALLOCATION_STRUCT al;
    double worst = 0;
    long   exact = 0, total = 0;
    const struct { uint32 delta_ss; bool group_hop, seq_hop; uint8 cs, cs_dci; } ulc[3] = {{0, false, false, 0, 0}, {7, true, false, 3, 5}, {13, false, true, 6, 1}};
    for (int b = 1; b < 6; b++) {
        Pair P;
        const uint32 cell = 40 * b + 3;
        if (!P.open(BW[b].fs, (uint16)cell, 1, BW[b].n_rb)) { g_bad++; return; }
        const int u = b % 3;
        if (liblte_phy_ul_init(P.own, (uint16)cell, 0, 0, 1, false, (uint8)ulc[u].delta_ss, ulc[u].group_hop, ulc[u].seq_hop, ulc[u].cs, ulc[u].cs_dci, 0, 1) != LIBLTE_SUCCESS ||
            liblte_phy_ul_init_cpu(P.ref, (uint16)cell, 0, 0, 1, false, (uint8)ulc[u].delta_ss, ulc[u].group_hop, ulc[u].seq_hop, ulc[u].cs, ulc[u].cs_dci, 0, 1) != LIBLTE_SUCCESS) { g_bad++; return; }
        memset(P.ref->pusch_z_re, 0, sizeof P.ref->pusch_z_re), memset(P.ref->pusch_z_im, 0, sizeof P.ref->pusch_z_im); // (see Pair::open)
        memset(P.ref->ulsch_c_bits, 0, sizeof P.ref->ulsch_c_bits), memset(P.ref->ulsch_tx_e_bits, 0, sizeof P.ref->ulsch_tx_e_bits);
        for (uint32 trial = 0; trial < 28; trial++) {
            memset(&al, 0, sizeof al);
            static const uint32 sizes[12] = {2, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 20}; // (sizes the reference has a transform plan for: a factor 2, 3 or 5 -- it calls a null plan otherwise)
            uint32 n_prb = sizes[rnd(12)];
            if (n_prb >= BW[b].n_rb) n_prb = 2;
            al.N_prb = n_prb, al.N_layers = 1, al.N_codewords = 1, al.tx_mode = 1, al.rv_idx = rnd(4), al.rnti = (uint16)(1 + rnd(65000));
            al.mod_type  = (LIBLTE_PHY_MODULATION_TYPE_ENUM)(trial % 5 == 4 ? 0 : 1 + rnd(3));
            al.chan_type = LIBLTE_PHY_CHAN_TYPE_ULSCH;
            for (uint32 i = 0; i < n_prb; i++) al.prb[0][i] = al.prb[1][i] = 3 + i;
            al.tbs = trial % 9 == 8 ? 6200 + rnd(7000) : trial % 4 == 3 ? 17 + rnd(2000) : 8 * (2 + rnd(180)) ;
            al.msg[0].N_bits = al.tbs > LIBLTE_MAX_MSG_SIZE ? LIBLTE_MAX_MSG_SIZE : al.tbs;
            for (uint32 i = 0; i < al.msg[0].N_bits; i++) al.msg[0].msg[i] = (uint8)(rnd() & 1);
            const uint32 sf = rnd(10);
            fill_grids(sf);
            if (getenv("TX_TRACE")) printf("pusch %u trial %u: N_prb %u mod %d tbs %u\n", BW[b].n_rb, trial, n_prb, (int)al.mod_type, al.tbs);
            const LIBLTE_ERROR_ENUM e1 = liblte_phy_pusch_channel_encode(P.own, &al, cell, 1, &g_s1);
            if (getenv("TX_TRACE")) printf("  own done\n");
            const LIBLTE_ERROR_ENUM e2 = liblte_phy_pusch_channel_encode_cpu(P.ref, &al, cell, 1, &g_s2);
            uint32 p = 0, l = 0, k = 0;
            const bool same = grids_equal(&p, &l, &k);
            const double rel = rel_l2(&g_s1.tx_symb_re[0][0][0], &g_s1.tx_symb_im[0][0][0], &g_s2.tx_symb_re[0][0][0], &g_s2.tx_symb_im[0][0][0], 14 * 1200);
            // the two reference-signal symbols and everything the call leaves alone are not transformed: bit for bit
            const bool rs_same = !memcmp(g_s1.tx_symb_re[0][3], g_s2.tx_symb_re[0][3], sizeof g_s1.tx_symb_re[0][3]) && !memcmp(g_s1.tx_symb_im[0][10], g_s2.tx_symb_im[0][10], sizeof g_s1.tx_symb_im[0][10]) &&
                                 !memcmp(g_s1.tx_symb_re[1], g_s2.tx_symb_re[1], sizeof g_s1.tx_symb_re[1]) && !memcmp(&g_s1.tx_symb_re[0][14], &g_s2.tx_symb_re[0][14], 2 * sizeof g_s1.tx_symb_re[0][0]);
            total++, exact += same;
            if (rel > worst) worst = rel;
            CHECK(e1 == e2 && rs_same && rel < 2e-7, "pusch_channel_encode(N_rb_ul %u, trial %u: N_prb %u, mod %d, tbs %u, rv %u, subframe %u): %d vs %d, relative L2 %.3g, reference signals %s, first at symbol %u sub-carrier %u",
                  BW[b].n_rb, trial, n_prb, (int)al.mod_type, al.tbs, al.rv_idx, sf, e1, e2, rel, rs_same ? "equal" : "DIFFER", l, k);
        }
        CHECK(liblte_phy_pusch_channel_encode(P.own, NULL, cell, 1, &g_s1) == LIBLTE_ERROR_INVALID_INPUTS && liblte_phy_pusch_channel_encode(P.own, &al, cell, 1, NULL) == LIBLTE_ERROR_INVALID_INPUTS,
              "pusch_channel_encode: argument checks"); // (the reference dereferences alloc before it tests it: not called with NULL)
        P.close();
    }
    printf("  pusch_channel_encode: %ld of %ld grids bit for bit, worst relative L2 %.3g\n", exact, total, worst);
}

void check_prach()
{
    static float r1[60000], i1[60000], r2[60000], i2[60000];
    double worst = 0;
    const struct { int bw; uint32 fmt, root, zczc; bool hs; uint32 pre, off; } tc[9] = {{0, 0, 22, 1, false, 0, 0},   {0, 0, 700, 12, false, 63, 0}, {1, 1, 5, 6, false, 17, 3},
                                                                                       {2, 2, 128, 9, false, 40, 10}, {1, 3, 837, 4, false, 9, 2},   {0, 0, 300, 5, true, 33, 0},
                                                                                       {2, 4, 7, 3, false, 21, 4},    {3, 0, 410, 14, false, 50, 20}, {5, 0, 0, 0, false, 0, 50}};
    for (int c = 0; c < 9; c++) {
        Pair P;
        if (!P.open(BW[tc[c].bw].fs, 11, 1, BW[tc[c].bw].n_rb)) { g_bad++; return; }
        if (liblte_phy_ul_init(P.own, 11, tc[c].root, tc[c].fmt, tc[c].zczc, tc[c].hs, 0, false, false, 0, 0, 0, 1) != LIBLTE_SUCCESS ||
            liblte_phy_ul_init_cpu(P.ref, 11, tc[c].root, tc[c].fmt, tc[c].zczc, tc[c].hs, 0, false, false, 0, 0, 0, 1) != LIBLTE_SUCCESS) { g_bad++; return; }
        const size_t n = P.ref->prach_T_cp + P.ref->prach_T_seq;
        memset(r1, 0, sizeof r1), memset(i1, 0, sizeof i1), memset(r2, 0, sizeof r2), memset(i2, 0, sizeof i2);
        const LIBLTE_ERROR_ENUM e1 = liblte_phy_generate_prach(P.own, tc[c].pre, tc[c].off, r1, i1), e2 = liblte_phy_generate_prach_cpu(P.ref, tc[c].pre, tc[c].off, r2, i2);
        const double rel = rel_l2(r1, i1, r2, i2, n + 16);
        if (rel > worst) worst = rel;
        CHECK(e1 == e2 && n <= 59000 && rel < 2e-7, "generate_prach(N_rb_ul %u, format %u, root %u, zczc %u, hs %d, preamble %u, offset %u): %d vs %d, %zu samples, relative L2 %.3g", BW[tc[c].bw].n_rb,
              tc[c].fmt, tc[c].root, tc[c].zczc, (int)tc[c].hs, tc[c].pre, tc[c].off, e1, e2, n, rel);
        CHECK(liblte_phy_generate_prach(P.own, 0, 0, NULL, i1) == liblte_phy_generate_prach_cpu(P.ref, 0, 0, NULL, i2), "generate_prach: argument checks");
        P.close();
    }
    printf("  generate_prach: worst relative L2 %.3g\n", worst);
}
} // namespace

int tx_check()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    if (getenv("TX_SEED")) { g_x ^= 2654435761u * (uint32)atoi(getenv("TX_SEED")); if (!g_x) g_x = 1; printf("  seed %s\n", getenv("TX_SEED")); } // further draws of every random case (soak)
    check_rate_match();
    printf("  rate_match: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_signals();
    printf("  signals: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_bch();
    printf("  bch: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_pdsch();
    printf("  pdsch: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_pdcch();
    printf("  pdcch: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_ofdm();
    printf("  ofdm: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_pusch();
    printf("  pusch: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    check_prach();
    printf("  prach: %ld comparisons so far, %ld differ\n", g_n, g_bad);
    printf("lifecycle_check tx: %ld comparisons %s\n", g_n, g_bad ? "DIFFER" : "equal");
    return g_bad ? 1 : 0;
}
