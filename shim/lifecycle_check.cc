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

int main()
{
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
