// dropin_demo.cc -- a caller written purely against the reference's liblte_phy API, in the order
// LTE_fdd_dl_file_scan uses it (LTE_fdd_dl_fs_samp_buf.cc:445-470): build one 20 MHz subframe with the
// reference's TX side, then liblte_phy_get_dl_subframe_and_ce -> liblte_phy_pdsch_channel_decode.
// Linked twice by shim/Makefile: against the unmodified reference objects (CPU), and against the
// reference objects + shim + libmi_lte.so (the three hot-path symbols now run on the GPU).
// Prints a line per allocation; the two builds must print the same lines.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "liblte_phy.h"

int main(int argc, char **argv)
{
    const uint32 N_id_cell = 17, sf_num = 1, cfi = 2;
    LIBLTE_PHY_STRUCT *phy = NULL;
    if (LIBLTE_SUCCESS != liblte_phy_init(&phy, LIBLTE_PHY_FS_30_72MHZ, N_id_cell, 1, 100, 12, 1.0f)) return 2;
    LIBLTE_PHY_SUBFRAME_STRUCT *tx = (LIBLTE_PHY_SUBFRAME_STRUCT *)calloc(1, sizeof(*tx));
    LIBLTE_PHY_SUBFRAME_STRUCT *rx = (LIBLTE_PHY_SUBFRAME_STRUCT *)calloc(1, sizeof(*rx));
    LIBLTE_PHY_PDCCH_STRUCT    *pd = (LIBLTE_PHY_PDCCH_STRUCT *)calloc(1, sizeof(*pd));
    float *i_s = (float *)calloc(3 * 30720 + 64, sizeof(float)), *q_s = (float *)calloc(3 * 30720 + 64, sizeof(float));
    const uint32 tbs[3] = {3240, 2024, 680}, nprb[3] = {12, 8, 4};
    const LIBLTE_PHY_MODULATION_TYPE_ENUM mod[3] = {LIBLTE_PHY_MODULATION_TYPE_64QAM, LIBLTE_PHY_MODULATION_TYPE_64QAM, LIBLTE_PHY_MODULATION_TYPE_16QAM};
    srand(7);
    pd->N_symbs = cfi;
    pd->N_alloc = 3;
    uint32 first = 0;
    for (int a = 0; a < 3; a++) {
        LIBLTE_PHY_ALLOCATION_STRUCT *al = &pd->alloc[a];
        al->pre_coder_type = LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY;
        al->mod_type = mod[a]; al->chan_type = LIBLTE_PHY_CHAN_TYPE_DLSCH; al->tbs = tbs[a]; al->rv_idx = 0; al->N_prb = nprb[a];
        for (uint32 i = 0; i < nprb[a]; i++) al->prb[0][i] = al->prb[1][i] = first + i;
        first += nprb[a];
        al->N_codewords = 1; al->N_layers = 1; al->tx_mode = 1; al->rnti = 0x100 + a;
        al->msg[0].N_bits = tbs[a];
        for (uint32 i = 0; i < tbs[a]; i++) al->msg[0].msg[i] = rand() & 1;
    }
    for (uint32 s = sf_num; s < sf_num + 2; s++) { // the subframe + a CRS-only successor for the look-ahead symbols
        memset(tx->tx_symb_re, 0, sizeof(tx->tx_symb_re));
        memset(tx->tx_symb_im, 0, sizeof(tx->tx_symb_im));
        tx->num = s;
        liblte_phy_map_crs(phy, tx, N_id_cell, 1);
        if (s == sf_num) liblte_phy_pdsch_channel_encode(phy, pd, N_id_cell, 1, tx);
        liblte_phy_create_dl_subframe(phy, tx, 0, &i_s[s * 30720], &q_s[s * 30720]);
    }
    if (LIBLTE_SUCCESS != liblte_phy_get_dl_subframe_and_ce(phy, i_s, q_s, 0, sf_num, N_id_cell, 1, rx)) { printf("front end failed\n"); return 3; }
    double acc = 0;
    for (int l = 0; l < 14; l++) for (int k = 0; k < 1200; k++) acc += rx->rx_ce_re[0][l][k] * rx->rx_ce_re[0][l][k] + rx->rx_ce_im[0][l][k] * rx->rx_ce_im[0][l][k];
    printf("mean |h|^2 = %.3f\n", acc / (14 * 1200));
    int bad = 0;
    for (int a = 0; a < 3; a++) {
        uint8 out[LIBLTE_MAX_MSG_SIZE]; uint32 n = 0;
        LIBLTE_ERROR_ENUM e = liblte_phy_pdsch_channel_decode(phy, rx, &pd->alloc[a], cfi, N_id_cell, 1, out, &n);
        uint32 diff = 0;
        for (uint32 i = 0; i < n; i++) diff += out[i] != pd->alloc[a].msg[0].msg[i];
        printf("alloc %d: err=%d N_out_bits=%u bit_errors=%u\n", a, (int)e, n, diff);
        bad += (e != LIBLTE_SUCCESS) || diff;
    }
    liblte_phy_cleanup(phy);
    return bad ? 1 : 0;
}
