// lifecycle_check.cc -- the shim's OWN liblte_phy_init / liblte_phy_update_n_rb_dl / liblte_phy_cleanup (liblte_phy_shim.cc built with
// -DMI_LTE_SHIM_OWN_LIFECYCLE) against the reference's, which this TEST binary links under the names *_cpu (shim/Makefile: phy_renamed_all.o).
// Every sampling rate x bandwidth it carries x PHICH resource x prefix through init (with a pair it does not carry the reference's init
// runs its PDCCH pre-calculation on an unset N_rb_dl), then every bandwidth -- valid or not -- through update: the return codes and every field a caller or a replaced entry
// point reads must be equal (liblte_phy.cc:2210-2335, :2592-2647).  Exit code 0 and "lifecycle_check: N combinations equal" on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>

#include "liblte_phy.h"

LIBLTE_ERROR_ENUM liblte_phy_init_cpu(LIBLTE_PHY_STRUCT **phy_struct, LIBLTE_PHY_FS_ENUM fs, uint16 N_id_cell, uint8 N_ant, uint32 N_rb_dl, uint32 N_sc_rb_dl,
                                      float phich_res);
LIBLTE_ERROR_ENUM liblte_phy_cleanup_cpu(LIBLTE_PHY_STRUCT *phy_struct);
LIBLTE_ERROR_ENUM liblte_phy_update_n_rb_dl_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint32 N_rb_dl);
LIBLTE_ERROR_ENUM liblte_phy_ul_init_cpu(LIBLTE_PHY_STRUCT *phy_struct, uint16 N_id_cell, uint32 prach_root_seq_idx, uint32 prach_preamble_format, uint32 prach_zczc,
                                         bool prach_hs_flag, uint8 group_assignment_pusch, bool group_hopping_enabled, bool sequence_hopping_enabled, uint8 cyclic_shift,
                                         uint8 cyclic_shift_dci, uint8 N_cs_an, uint8 delta_pucch_shift);
LIBLTE_ERROR_ENUM liblte_phy_pucch_format_1_1a_1b_channel_decode_cpu(LIBLTE_PHY_STRUCT *phy_struct, LIBLTE_PHY_SUBFRAME_STRUCT *subframe, LIBLTE_PHY_PUCCH_FORMAT_ENUM format,
                                                                     uint32 N_id_cell, uint8 N_ant, uint32 N_1_p_pucch, uint8 *out_bits, uint32 *N_out_bits);
extern int32 W_5_4_1_2_cpu[3][4];

// `lifecycle_check pucch` (needs the GPU): PUCCH format 1 / 1a / 1b resources built from the tables the REFERENCE's liblte_phy_ul_init computed
// (36.211 5.4.1, through a per-slot complex gain), decoded by the reference on its struct and by the shim's own-lifecycle build on a struct of
// its own -- whose decoder asks the library's generator (mi_lte_ul_pucch_tables) instead of reading tables out of the struct.  Error code,
// bit count and bits must be equal for every case.
static int pucch_check()
{
    static LIBLTE_PHY_SUBFRAME_STRUCT ps;
    int bad = 0, n = 0;
    const struct { uint32 cell, delta_ss; bool hop; uint8 n_cs_an, shift; } cfgs[3] = {{17, 3, false, 0, 1}, {301, 0, true, 0, 2}, {44, 7, false, 2, 0}};
    for (int c = 0; c < 3; c++) {
        LIBLTE_PHY_STRUCT *own = NULL, *ref = NULL;
        if (liblte_phy_init(&own, LIBLTE_PHY_FS_30_72MHZ, cfgs[c].cell, 1, 100, 12, 1.0f) != LIBLTE_SUCCESS || liblte_phy_init_cpu(&ref, LIBLTE_PHY_FS_30_72MHZ, cfgs[c].cell, 1, 100, 12, 1.0f) != LIBLTE_SUCCESS) return 1;
        if (liblte_phy_ul_init(own, cfgs[c].cell, 0, 0, 1, false, cfgs[c].delta_ss, cfgs[c].hop, false, 0, 0, cfgs[c].n_cs_an, cfgs[c].shift) != LIBLTE_SUCCESS ||
            liblte_phy_ul_init_cpu(ref, cfgs[c].cell, 0, 0, 1, false, cfgs[c].delta_ss, cfgs[c].hop, false, 0, 0, cfgs[c].n_cs_an, cfgs[c].shift) != LIBLTE_SUCCESS) return 1;
        const struct { LIBLTE_PHY_PUCCH_FORMAT_ENUM fmt; float d_re, d_im; } tc[5] = {{LIBLTE_PHY_PUCCH_FORMAT_1, 1, 0}, {LIBLTE_PHY_PUCCH_FORMAT_1A, -1, 0}, {LIBLTE_PHY_PUCCH_FORMAT_1A, 1, 0},
                                                                                      {LIBLTE_PHY_PUCCH_FORMAT_1B, 0, 1}, {LIBLTE_PHY_PUCCH_FORMAT_1B, -1, 0}};
        for (uint32 sf = 0; sf < 10; sf += 3)
            for (uint32 n1 = 0; n1 < 12; n1 += (n1 < 3 ? 1 : 4))
                for (int t = 0; t < 5; t++) {
                    memset(&ps, 0, sizeof(ps));
                    ps.num = sf;
                    const uint32 symb[4] = {0, 1, 5, 6};
                    for (uint32 m = 0; m < 2; m++) {
                        const uint32 prb = m == 0 ? n1 : ref->N_rb_ul - n1 - 1;
                        const float  hr = m == 0 ? 0.8f : -0.3f, hi = m == 0 ? 0.5f : 1.1f; // per-slot channel
                        const bool   even = (ref->pucch_n_prime_p[sf][n1][m] % 2) == 0;
                        const float  s_re = even ? 1.0f : 0.0f, s_im = even ? 0.0f : 1.0f;
                        for (uint32 j = 0; j < 12; j++) {
                            for (uint32 i = 0; i < 4; i++) {
                                const float w = (float)W_5_4_1_2_cpu[ref->pucch_n_oc_p[sf][n1][m]][i];
                                const float rr = ref->pucch_r_u_v_alpha_p_re[sf][n1][m][symb[i]][j], ri = ref->pucch_r_u_v_alpha_p_im[sf][n1][m][symb[i]][j];
                                const float ar = tc[t].d_re * s_re - tc[t].d_im * s_im, ai = tc[t].d_re * s_im + tc[t].d_im * s_re;
                                const float xr = w * (ar * rr - ai * ri), xi = w * (ar * ri + ai * rr);
                                ps.rx_symb_re[7 * m + symb[i]][prb * 12 + j] = hr * xr - hi * xi;
                                ps.rx_symb_im[7 * m + symb[i]][prb * 12 + j] = hr * xi + hi * xr;
                            }
                            for (uint32 i = 0; i < 3; i++) {
                                const float dr = m == 0 ? ref->pucch_dmrs_0_re[sf][n1][i * 12 + j] : ref->pucch_dmrs_1_re[sf][n1][i * 12 + j];
                                const float di = m == 0 ? ref->pucch_dmrs_0_im[sf][n1][i * 12 + j] : ref->pucch_dmrs_1_im[sf][n1][i * 12 + j];
                                ps.rx_symb_re[7 * m + 2 + i][prb * 12 + j] = hr * dr - hi * di;
                                ps.rx_symb_im[7 * m + 2 + i][prb * 12 + j] = hr * di + hi * dr;
                            }
                        }
                    }
                    uint8  b1[4] = {0, 0, 0, 0}, b2[4] = {0, 0, 0, 0};
                    uint32 nb1 = 0, nb2 = 0;
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_pucch_format_1_1a_1b_channel_decode(own, &ps, tc[t].fmt, cfgs[c].cell, 1, n1, b1, &nb1);
                    const LIBLTE_ERROR_ENUM e2 = liblte_phy_pucch_format_1_1a_1b_channel_decode_cpu(ref, &ps, tc[t].fmt, cfgs[c].cell, 1, n1, b2, &nb2);
                    n++;
                    if (e1 != e2 || nb1 != nb2 || b1[0] != b2[0] || (nb2 == 2 && b1[1] != b2[1])) {
                        printf("  cfg %d subframe %u resource %u case %d: own err=%d n=%u bits=%u%u, reference err=%d n=%u bits=%u%u\n", c, sf, n1, t, (int)e1, nb1, b1[0], b1[1], (int)e2, nb2, b2[0], b2[1]);
                        bad++;
                    }
                }
        liblte_phy_cleanup(own);
        liblte_phy_cleanup_cpu(ref);
    }
    printf("lifecycle_check pucch: %d decodes %s\n", n, bad ? "DIFFER" : "equal");
    return bad ? 1 : 0;
}

// `lifecycle_check helpers` (CPU only): the seven scheduler-side helpers the own-lifecycle build defines (sched.cc behind the shim) against the
// reference's, linked here under the names *_cpu -- over their whole argument ranges, the outputs pre-set to the same sentinel on both sides
// so that "left untouched" is compared too.
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_mcs_and_n_prb_for_dl_cpu(uint32 N_bits, uint32 N_subframe, uint32 N_rb_dl, uint16 rnti, uint32 *tbs, uint8 *mcs, uint32 *N_prb);
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_and_n_prb_for_dl_cpu(uint32 N_bits, uint32 N_rb_dl, uint8 mcs, uint32 *tbs, uint32 *N_prb);
LIBLTE_ERROR_ENUM liblte_phy_get_tbs_mcs_and_n_prb_for_ul_cpu(uint32 N_bits, uint32 N_rb_ul, uint32 *tbs, uint8 *mcs, uint32 *N_prb);
LIBLTE_ERROR_ENUM liblte_phy_get_n_cce_cpu(LIBLTE_PHY_STRUCT *phy_struct, float phich_res, uint32 N_pdcch_symbs, uint8 N_ant, uint32 *N_cce);
void liblte_phy_pucch_map_sr_config_idx_cpu(uint32 i_sr, uint32 *sr_periodicity, uint32 *N_offset_sr);
void liblte_phy_code_block_segmentation_cpu(uint8 *b_bits, uint32 N_b_bits, uint32 *N_codeblocks, uint32 *N_filler_bits, uint8 *c_bits, uint32 N_c_bits_max, uint32 *N_c_bits);
void liblte_phy_code_block_desegmentation_cpu(uint8 *c_bits, uint32 *N_c_bits, uint32 N_c_bits_max, uint32 tbs, uint8 *b_bits, uint32 N_b_bits);

#include <unistd.h>
int tx_check(); // tx_check.cc: the transmit functions
static int helpers_check()
{
    static const uint32 bws[6] = {6, 15, 25, 50, 75, 100};
    long n = 0, bad = 0;
#define CHECK(cond, ...) do { n++; if (!(cond)) { if (bad < 20) { printf("  " __VA_ARGS__); printf("\n"); } bad++; } } while (0)
    // DL grant sizes: every message size up to past the table's largest entry for user RNTIs; the broadcast RNTIs (their search reads the
    // N_PRB = 3 column only, largest entry 2216) for every subframe number
    static const uint16 user[3] = {0x003D, 0x1234, 0xFFF3}, bcast[4] = {0xFFFF, 0xFFFE, 0x0001, 0x003C};
    for (int b = 0; b < 6; b++) {
        for (int r = 0; r < 3; r++)
            for (uint32 bits = 0; bits <= 76000; bits += (bits < 9000 ? 1 : 7)) {
                uint32 t1 = 0xAAAA, t2 = 0xAAAA, p1 = 0xBBBB, p2 = 0xBBBB; uint8 m1 = 0xCC, m2 = 0xCC;
                const LIBLTE_ERROR_ENUM e1 = liblte_phy_get_tbs_mcs_and_n_prb_for_dl(bits, 3, bws[b], user[r], &t1, &m1, &p1);
                const LIBLTE_ERROR_ENUM e2 = liblte_phy_get_tbs_mcs_and_n_prb_for_dl_cpu(bits, 3, bws[b], user[r], &t2, &m2, &p2);
                CHECK(e1 == e2 && t1 == t2 && m1 == m2 && p1 == p2, "tbs_mcs_n_prb_for_dl(%u, N_rb_dl %u, rnti %x): %d %u %u %u vs %d %u %u %u", bits, bws[b], user[r], e1, t1, m1, p1, e2, t2, m2, p2);
            }
        for (int r = 0; r < 4; r++)
            for (uint32 sf = 0; sf < 10; sf++)
                for (uint32 bits = 0; bits <= 2400; bits++) {
                    uint32 t1 = 300 + bits % 7, t2 = t1, p1 = 0xBBBB, p2 = 0xBBBB; uint8 m1 = 0xCC, m2 = 0xCC; // (a message past the column's end prices the caller's own *tbs)
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_get_tbs_mcs_and_n_prb_for_dl(bits, sf, bws[b], bcast[r], &t1, &m1, &p1);
                    const LIBLTE_ERROR_ENUM e2 = liblte_phy_get_tbs_mcs_and_n_prb_for_dl_cpu(bits, sf, bws[b], bcast[r], &t2, &m2, &p2);
                    CHECK(e1 == e2 && t1 == t2 && m1 == m2 && p1 == p2, "tbs_mcs_n_prb_for_dl(%u, sf %u, N_rb_dl %u, rnti %x): %d %u %u %u vs %d %u %u %u", bits, sf, bws[b], bcast[r], e1, t1, m1, p1, e2, t2, m2, p2);
                }
        for (uint32 mcs = 0; mcs < 32; mcs++)
            for (uint32 bits = 0; bits <= 76000; bits += (bits < 9000 ? 1 : 7)) {
                uint32 t1 = 0xAAAA, t2 = 0xAAAA, p1 = 0xBBBB, p2 = 0xBBBB;
                const LIBLTE_ERROR_ENUM e1 = liblte_phy_get_tbs_and_n_prb_for_dl(bits, bws[b], (uint8)mcs, &t1, &p1);
                const LIBLTE_ERROR_ENUM e2 = liblte_phy_get_tbs_and_n_prb_for_dl_cpu(bits, bws[b], (uint8)mcs, &t2, &p2);
                CHECK(e1 == e2 && t1 == t2 && p1 == p2, "tbs_and_n_prb_for_dl(%u, N_rb_dl %u, mcs %u): %d %u %u vs %d %u %u", bits, bws[b], mcs, e1, t1, p1, e2, t2, p2);
            }
        for (uint32 bits = 0; bits <= 9000; bits++) {
            uint32 t1 = 0xAAAA, t2 = 0xAAAA, p1 = 0xBBBB, p2 = 0xBBBB; uint8 m1 = 0xCC, m2 = 0xCC;
            const LIBLTE_ERROR_ENUM e1 = liblte_phy_get_tbs_mcs_and_n_prb_for_ul(bits, bws[b], &t1, &m1, &p1);
            const LIBLTE_ERROR_ENUM e2 = liblte_phy_get_tbs_mcs_and_n_prb_for_ul_cpu(bits, bws[b], &t2, &m2, &p2);
            CHECK(e1 == e2 && t1 == t2 && m1 == m2 && p1 == p2, "tbs_mcs_n_prb_for_ul(%u): %d %u %u %u vs %d %u %u %u", bits, e1, t1, m1, p1, e2, t2, m2, p2);
        }
        // control channel elements: every PHICH group count a cell of this bandwidth can have (N_g = 1/6 .. 2 -> ceil(N_g * N_rb / 8)), and one it cannot
        static LIBLTE_PHY_STRUCT ps; // (46 MB: static)
        ps.N_rb_dl = bws[b];
        for (uint32 g = 0; g <= 26; g++)
            for (uint32 cfi = 1; cfi <= 4; cfi++)
                for (uint8 ant = 1; ant <= 4; ant <<= 1) {
                    ps.N_group_phich = g;
                    uint32 c1 = 0xAAAA, c2 = 0xAAAA;
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_get_n_cce(&ps, 1.0f, cfi, ant, &c1), e2 = liblte_phy_get_n_cce_cpu(&ps, 1.0f, cfi, ant, &c2);
                    CHECK(e1 == e2 && c1 == c2, "get_n_cce(N_rb_dl %u, groups %u, cfi %u, ports %u): %u vs %u", bws[b], g, cfi, ant, c1, c2);
                }
    }
    for (uint32 i = 0; i < 400; i++) {
        uint32 a1 = 1, a2 = 1, o1 = 2, o2 = 2;
        liblte_phy_pucch_map_sr_config_idx(i, &a1, &o1);
        liblte_phy_pucch_map_sr_config_idx_cpu(i, &a2, &o2);
        CHECK(a1 == a2 && o1 == o2, "pucch_map_sr_config_idx(%u): %u %u vs %u %u", i, a1, o1, a2, o2);
    }
    // segmentation / desegmentation: every transport block length up to one code block, a spread of lengths that need two to five
    // (there the reference computes each block's CRC over what the caller's buffer holds where the parity goes: the buffers start out equal)
    {
        const uint32 MAXC = 6, STRIDE = 6176;
        static uint8 b[40000], c1[6 * 6176], c2[6 * 6176], d1[40000], d2[40000];
        uint32 x = 12345;
        for (uint32 i = 0; i < sizeof b; i++) { x = x * 1103515245u + 12345u; b[i] = (uint8)((x >> 16) & 1u); }
        fflush(stdout);
        const int keep = dup(1), nul = open("/dev/null", 1); // the reference's desegmentation prints a line per code block when there are several
        for (uint32 B = 1; B <= 30000; B += (B <= 6200 ? 1 : 131)) {
            for (uint32 i = 0; i < MAXC * STRIDE; i++) { x = x * 1103515245u + 12345u; c1[i] = c2[i] = (uint8)((x >> 16) & 1u); }
            uint32 nc1 = 77, nc2 = 77, f1 = 88, f2 = 88, l1[MAXC], l2[MAXC];
            for (uint32 i = 0; i < MAXC; i++) l1[i] = l2[i] = 99;
            liblte_phy_code_block_segmentation(b, B, &nc1, &f1, c1, STRIDE, l1);
            liblte_phy_code_block_segmentation_cpu(b, B, &nc2, &f2, c2, STRIDE, l2);
            CHECK(nc1 == nc2 && f1 == f2 && !memcmp(l1, l2, sizeof l1) && !memcmp(c1, c2, sizeof c1), "code_block_segmentation(%u): %u blocks %u filler vs %u %u", B, nc1, f1, nc2, f2);
            if (B >= 25) { // and back: tbs = B - 24
                memset(d1, 7, sizeof d1); memset(d2, 7, sizeof d2);
                dup2(nul, 1);
                liblte_phy_code_block_desegmentation_cpu(c2, l2, STRIDE, B - 24, d2, B);
                fflush(stdout);
                dup2(keep, 1);
                liblte_phy_code_block_desegmentation(c1, l1, STRIDE, B - 24, d1, B);
                CHECK(!memcmp(d1, d2, sizeof d1), "code_block_desegmentation(tbs %u)", B - 24);
            }
        }
        close(nul); close(keep);
    }
#undef CHECK
    printf("lifecycle_check helpers: %ld comparisons %s\n", n, bad ? "DIFFER" : "equal");
    return bad ? 1 : 0;
}

static int diff(const char *what, const LIBLTE_PHY_STRUCT *a, const LIBLTE_PHY_STRUCT *b, bool with_bw)
{
    int n = 0;
#define F(f) do { if (a->f != b->f) { printf("  %s: %s differs: %u (own) vs %u (reference)\n", what, #f, (unsigned)a->f, (unsigned)b->f); n++; } } while (0)
    F(fs); F(N_samps_per_symb); F(N_samps_cp_l_0); F(N_samps_cp_l_else); F(N_samps_per_slot); F(N_samps_per_subfr); F(N_samps_per_frame);
    F(N_sc_rb_dl); F(N_sc_rb_ul); F(N_ant); F(ul_init); F(N_sf_phich);
    if (with_bw) { F(N_rb_dl); F(N_rb_ul); F(FFT_size); F(FFT_pad_size); F(N_group_phich); }
#undef F
    return n;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "pucch")) return pucch_check();
    if (argc > 1 && !strcmp(argv[1], "helpers")) return helpers_check();
    if (argc > 1 && !strcmp(argv[1], "tx")) return tx_check();
    static const LIBLTE_PHY_FS_ENUM fss[5] = {LIBLTE_PHY_FS_1_92MHZ, LIBLTE_PHY_FS_3_84MHZ, LIBLTE_PHY_FS_7_68MHZ, LIBLTE_PHY_FS_15_36MHZ, LIBLTE_PHY_FS_30_72MHZ};
    static const uint32 rbs[8] = {6, 15, 25, 50, 75, 100, 7, 110};
    static const float  res[4] = {1.0f / 6, 0.5f, 1.0f, 2.0f};
    static const int    n_bw[5] = {1, 2, 3, 4, 6}; // bandwidths a sampling rate carries (liblte_phy.cc:2603-2638)
    int bad = 0, n = 0;
    setvbuf(stdout, NULL, _IONBF, 0);
    for (int f = 0; f < 5; f++)
        for (int r = 0; r < n_bw[f]; r++)
            for (int g = 0; g < 4; g++)
                for (uint32 sc = 12; sc <= 24; sc += 12) {
                    if (rbs[r] == 6 && g == 3) continue; // (the reference's own init crashes in its PDCCH pre-calculation for N_g = 2 at 6 RB)
                    LIBLTE_PHY_STRUCT *own = NULL, *ref = NULL;
                    const uint8 n_ant = (uint8)(1u << (g % 3));
                    const LIBLTE_ERROR_ENUM e1 = liblte_phy_init(&own, fss[f], (uint16)(17 * f + r), n_ant, rbs[r], sc, res[g]);
                    const LIBLTE_ERROR_ENUM e2 = liblte_phy_init_cpu(&ref, fss[f], (uint16)(17 * f + r), n_ant, rbs[r], sc, res[g]);
                    char what[96];
                    snprintf(what, sizeof what, "fs %d N_rb_dl %u phich %.3f N_sc %u", f, rbs[r], res[g], sc);
                    if (getenv("LC_TRACE")) printf("%s\n", what);
                    if (e1 != e2 || !own || !ref) { printf("  %s: init returns %d vs %d\n", what, (int)e1, (int)e2); bad++; continue; }
                    bad += diff(what, own, ref, true);
                    // every bandwidth through update on both structs
                    for (int r2 = 0; r2 < 8; r2++) {
                        const LIBLTE_ERROR_ENUM u1 = liblte_phy_update_n_rb_dl(own, rbs[r2]), u2 = liblte_phy_update_n_rb_dl_cpu(ref, rbs[r2]);
                        if (u1 != u2) { printf("  %s: update to %u returns %d vs %d\n", what, rbs[r2], (int)u1, (int)u2); bad++; }
                        if (u1 == LIBLTE_SUCCESS && u2 == LIBLTE_SUCCESS) {
                            own->N_group_phich = ref->N_group_phich; // (update does not touch it)
                            bad += diff(what, own, ref, true);
                        }
                        n++;
                    }
                    if (liblte_phy_update_n_rb_dl(NULL, 6) != liblte_phy_update_n_rb_dl_cpu(NULL, 6)) bad++;
                    if (liblte_phy_cleanup(own) != LIBLTE_SUCCESS || liblte_phy_cleanup_cpu(ref) != LIBLTE_SUCCESS) bad++;
                }
    if (liblte_phy_init(NULL, LIBLTE_PHY_FS_1_92MHZ, 0, 1, 6, 12, 1) != liblte_phy_init_cpu(NULL, LIBLTE_PHY_FS_1_92MHZ, 0, 1, 6, 12, 1)) bad++;
    if (liblte_phy_cleanup(NULL) != liblte_phy_cleanup_cpu(NULL)) bad++;
    printf("lifecycle_check: %d combinations %s\n", n, bad ? "DIFFER" : "equal");
    return bad ? 1 : 0;
}
