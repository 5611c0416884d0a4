// lifecycle_check.cc -- the shim's OWN liblte_phy_init / liblte_phy_update_n_rb_dl / liblte_phy_cleanup (liblte_phy_shim.cc built with
// -DMI_LTE_SHIM_OWN_LIFECYCLE) against the reference's, which this TEST binary links under the names *_cpu (shim/Makefile: phy_renamed_all.o).
// Every sampling rate x bandwidth it carries x PHICH resource x prefix through init (with a pair it does not carry the reference's init
// runs its PDCCH pre-calculation on an unset N_rb_dl), then every bandwidth -- valid or not -- through update: the return codes and every field a caller or a replaced entry
// point reads must be equal (liblte_phy.cc:2210-2335, :2592-2647).  Exit code 0 and "lifecycle_check: N combinations equal" on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
