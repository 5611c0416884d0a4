// dropin_ul_demo.cc -- an uplink caller written purely against the reference's liblte_phy API, in the order
// LTE_fdd_enodeb's radio thread uses it (LTE_fdd_enb_phy.cc:832-917): liblte_phy_init + liblte_phy_ul_init, then per
// subframe liblte_phy_get_ul_subframe and one liblte_phy_pusch_channel_decode per scheduled UE.  The capture is an
// int8 I,Q file of one subframe (tests write it with the library's host transmitter; the reference cannot transmit
// uplink).  Linked twice by shim/Makefile, CPU-only and with the two entry points on the GPU; same lines expected.
//
//   dropin_ul_* <capture.bin> <N_rb_ul> <N_id_cell> <subframe> <delta_ss> <group_hop> <seq_hop> <cs> <cs_dci>
//               then per UE: <mod> <tbs> <rnti> <first_prb> <N_prb>
//   with PRACH_CAPTURE=<file> PRACH_CFG="root,fmt,zczc,hs,freq_offset" in the environment it also runs liblte_phy_detect_prach
//   over that capture (one occasion starting at the file's first sample); with PUCCH_DEMO=1 it also decodes four PUCCH format 1/1a/1b
//   resources it builds itself from the sequences in the struct (liblte_phy_pucch_format_1_1a_1b_channel_decode); with ENB_DL_TX=1 it also builds and
//   modulates the downlink subframe of the same TTI the way LTE_fdd_enb_phy.cc:557-770 does
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <cmath>

#include "liblte_phy.h"
#include "liblte_rrc.h"
#ifdef MI_LTE_HAVE_UL_SUBFRAME_DECODE
#include "liblte_phy_ext.h"
#endif

int main(int argc, char **argv)
{
    if (argc < 15 || (argc - 10) % 5) { fprintf(stderr, "usage: see the file header\n"); return 2; }
    const uint32 N_rb = atoi(argv[2]), cell = atoi(argv[3]), sf_num = atoi(argv[4]);
    const LIBLTE_PHY_FS_ENUM fs = N_rb <= 6 ? LIBLTE_PHY_FS_1_92MHZ : N_rb <= 15 ? LIBLTE_PHY_FS_3_84MHZ : N_rb <= 25 ? LIBLTE_PHY_FS_7_68MHZ
                                : N_rb <= 50 ? LIBLTE_PHY_FS_15_36MHZ : LIBLTE_PHY_FS_30_72MHZ;
    LIBLTE_PHY_STRUCT *phy = NULL;
    if (LIBLTE_SUCCESS != liblte_phy_init(&phy, fs, cell, 1, N_rb, 12, 1.0f)) return 3;
    uint32 pr[5] = {0, 0, 1, 0, 0};
    if (getenv("PRACH_CFG")) sscanf(getenv("PRACH_CFG"), "%u,%u,%u,%u,%u", &pr[0], &pr[1], &pr[2], &pr[3], &pr[4]);
    if (LIBLTE_SUCCESS != liblte_phy_ul_init(phy, cell, pr[0], pr[1], pr[2], pr[3] != 0, atoi(argv[5]), atoi(argv[6]) != 0, atoi(argv[7]) != 0, atoi(argv[8]),
                                             atoi(argv[9]), 0, 1))
        return 3;
    const uint32 n = phy->N_samps_per_subfr;
    float *i_s = (float *)calloc(n + 64, sizeof(float)), *q_s = (float *)calloc(n + 64, sizeof(float));
    FILE  *f   = fopen(argv[1], "rb");
    if (!f) return 4;
    for (uint32 k = 0; k < n; k++) {
        signed char v[2];
        if (fread(v, 1, 2, f) != 2) break;
        i_s[k] = v[0];
        q_s[k] = v[1];
    }
    fclose(f);
    LIBLTE_PHY_SUBFRAME_STRUCT *rx = (LIBLTE_PHY_SUBFRAME_STRUCT *)calloc(1, sizeof(*rx));
    rx->num = sf_num;
    if (LIBLTE_SUCCESS != liblte_phy_get_ul_subframe(phy, i_s, q_s, rx)) { printf("get_ul_subframe failed\n"); return 5; }
    double acc = 0;
    for (int l = 0; l < 14; l++) for (uint32 k = 0; k < 12 * N_rb; k++) acc += rx->rx_symb_re[l][k] * rx->rx_symb_re[l][k] + rx->rx_symb_im[l][k] * rx->rx_symb_im[l][k];
    printf("grid energy = %.4e\n", acc);
    LIBLTE_PHY_ALLOCATION_STRUCT *al = (LIBLTE_PHY_ALLOCATION_STRUCT *)calloc(1, sizeof(*al));
    for (int a = 10; a < argc; a += 5) {
        memset(al, 0, sizeof(*al));
        al->mod_type  = (LIBLTE_PHY_MODULATION_TYPE_ENUM)atoi(argv[a]);
        al->chan_type = LIBLTE_PHY_CHAN_TYPE_ULSCH;
        al->tbs = atoi(argv[a + 1]); al->rnti = atoi(argv[a + 2]); al->N_prb = atoi(argv[a + 4]);
        for (uint32 i = 0; i < al->N_prb; i++) al->prb[0][i] = al->prb[1][i] = atoi(argv[a + 3]) + i;
        al->N_codewords = 1; al->N_layers = 1; al->tx_mode = 1; al->rv_idx = 0;
        uint8 out[LIBLTE_MAX_MSG_SIZE]; uint32 nb = 0;
        LIBLTE_ERROR_ENUM e = liblte_phy_pusch_channel_decode(phy, rx, al, cell, 1, out, &nb);
        uint32 h = 2166136261u; // FNV-1a over the decoded bits
        for (uint32 i = 0; i < nb; i++) h = (h ^ out[i]) * 16777619u;
        printf("rnti 0x%x: err=%d N_out_bits=%u hash=%08x\n", (unsigned)al->rnti, (int)e, nb, h);
    }
    if (getenv("UL_DEMO_REPEAT")) { // what the eNodeB's radio thread pays per subframe: get_ul_subframe + one pusch_channel_decode per UE (stderr)
        const int reps = atoi(getenv("UL_DEMO_REPEAT"));
        const auto t0 = std::chrono::steady_clock::now();
        uint32 ok = 0;
        for (int r = 0; r < reps; r++) {
            liblte_phy_get_ul_subframe(phy, i_s, q_s, rx);
            for (int a = 10; a < argc; a += 5) {
                memset(al, 0, sizeof(*al));
                al->mod_type  = (LIBLTE_PHY_MODULATION_TYPE_ENUM)atoi(argv[a]);
                al->chan_type = LIBLTE_PHY_CHAN_TYPE_ULSCH;
                al->tbs = atoi(argv[a + 1]); al->rnti = atoi(argv[a + 2]); al->N_prb = atoi(argv[a + 4]);
                for (uint32 i = 0; i < al->N_prb; i++) al->prb[0][i] = al->prb[1][i] = atoi(argv[a + 3]) + i;
                al->N_codewords = 1; al->N_layers = 1; al->tx_mode = 1; al->rv_idx = 0;
                uint8 out[LIBLTE_MAX_MSG_SIZE]; uint32 nb = 0;
                ok += LIBLTE_SUCCESS == liblte_phy_pusch_channel_decode(phy, rx, al, cell, 1, out, &nb);
            }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "timing: %d x (get_ul_subframe + %d pusch_channel_decode): %.1f us per subframe, %u decodes ok\n", reps, (argc - 10) / 5, us / reps, ok);
    }
#ifdef MI_LTE_HAVE_UL_SUBFRAME_DECODE
    if (getenv("UL_DEMO_ONE_CALL")) { // the same subframe through the shim's one-call form (liblte_phy_ext.h): same lines as above, prefixed
        const uint32 n_ue = (argc - 10) / 5;
        LIBLTE_PHY_ALLOCATION_STRUCT *als = (LIBLTE_PHY_ALLOCATION_STRUCT *)calloc(n_ue, sizeof(*als));
        for (uint32 u = 0; u < n_ue; u++) {
            const int a = 10 + 5 * u;
            als[u].mod_type  = (LIBLTE_PHY_MODULATION_TYPE_ENUM)atoi(argv[a]);
            als[u].chan_type = LIBLTE_PHY_CHAN_TYPE_ULSCH;
            als[u].tbs = atoi(argv[a + 1]); als[u].rnti = atoi(argv[a + 2]); als[u].N_prb = atoi(argv[a + 4]);
            for (uint32 i = 0; i < als[u].N_prb; i++) als[u].prb[0][i] = als[u].prb[1][i] = atoi(argv[a + 3]) + i;
            als[u].N_codewords = 1; als[u].N_layers = 1; als[u].tx_mode = 1; als[u].rv_idx = 0;
        }
        uint8 *ob = (uint8 *)calloc(n_ue, LIBLTE_MAX_MSG_SIZE);
        uint32 *onb = (uint32 *)calloc(n_ue, sizeof(uint32));
        LIBLTE_ERROR_ENUM *ost = (LIBLTE_ERROR_ENUM *)calloc(n_ue, sizeof(LIBLTE_ERROR_ENUM));
        LIBLTE_ERROR_ENUM e = liblte_phy_ul_subframe_decode(phy, i_s, q_s, sf_num, cell, als, n_ue, ob, onb, ost, NULL, NULL, 0, NULL, NULL, NULL);
        if (e != LIBLTE_SUCCESS) printf("one call: failed (%d)\n", (int)e);
        for (uint32 u = 0; u < n_ue && e == LIBLTE_SUCCESS; u++) {
            uint32 h = 2166136261u;
            for (uint32 i = 0; i < onb[u]; i++) h = (h ^ ob[(size_t)u * LIBLTE_MAX_MSG_SIZE + i]) * 16777619u;
            printf("one call: rnti 0x%x: err=%d N_out_bits=%u hash=%08x\n", (unsigned)als[u].rnti, (int)ost[u], onb[u], h);
        }
        if (getenv("UL_DEMO_REPEAT")) {
            const int reps = atoi(getenv("UL_DEMO_REPEAT"));
            const auto t0 = std::chrono::steady_clock::now();
            uint32 ok = 0;
            for (int r = 0; r < reps; r++) {
                liblte_phy_ul_subframe_decode(phy, i_s, q_s, sf_num, cell, als, n_ue, ob, onb, ost, NULL, NULL, 0, NULL, NULL, NULL);
                for (uint32 u = 0; u < n_ue; u++) ok += ost[u] == LIBLTE_SUCCESS;
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "timing: %d x (ul_subframe_decode, %u UEs in one call): %.1f us per subframe, %u decodes ok\n", reps, n_ue, us / reps, ok);
        }
    }
#endif
    if (getenv("PRACH_CAPTURE")) {
        FILE *pf = fopen(getenv("PRACH_CAPTURE"), "rb");
        if (!pf) return 6;
        const uint32 np = phy->prach_T_cp + 2 * phy->prach_T_fft;
        float *pi_s = (float *)calloc(np + 64, sizeof(float)), *pq_s = (float *)calloc(np + 64, sizeof(float));
        for (uint32 k = 0; k < np; k++) {
            signed char v[2];
            if (fread(v, 1, 2, pf) != 2) break;
            pi_s[k] = v[0];
            pq_s[k] = v[1];
        }
        fclose(pf);
        uint32 nd = 0, dp = 0, ta = 0;
        LIBLTE_ERROR_ENUM e = liblte_phy_detect_prach(phy, pi_s, pq_s, pr[4], &nd, &dp, &ta);
        printf("prach: err=%d N_det_pre=%u det_pre=%u det_ta=%u\n", (int)e, nd, nd ? dp : 0, nd ? ta : 0);
    }
    if (getenv("PUCCH_DEMO")) {
        // PUCCH formats 1 / 1a / 1b: the reference cannot transmit them either, so a resource is built here from the sequences
        // liblte_phy_ul_init left in the struct (36.211 5.4.1), through a fixed complex gain, and handed to the decoder
        static LIBLTE_PHY_SUBFRAME_STRUCT ps;
        extern int32 W_5_4_1_2[3][4];
        const struct { LIBLTE_PHY_PUCCH_FORMAT_ENUM fmt; uint32 n1; float d_re, d_im; } tc[4] = {
            {LIBLTE_PHY_PUCCH_FORMAT_1, 0, 1, 0}, {LIBLTE_PHY_PUCCH_FORMAT_1A, 1, -1, 0}, {LIBLTE_PHY_PUCCH_FORMAT_1B, 2, 0, 1}, {LIBLTE_PHY_PUCCH_FORMAT_1B, 1, -1, 0}};
        for (int t = 0; t < 4; t++) {
            memset(&ps, 0, sizeof(ps));
            ps.num = sf_num;
            const uint32 symb[4] = {0, 1, 5, 6}, n1 = tc[t].n1;
            for (uint32 m = 0; m < 2; m++) {
                const uint32 prb = m == 0 ? n1 : phy->N_rb_ul - n1 - 1;
                const float  hr = m == 0 ? 0.8f : -0.3f, hi = m == 0 ? 0.5f : 1.1f; // per-slot channel
                const float  s_re = (phy->pucch_n_prime_p[sf_num][n1][m] % 2) == 0 ? 1.0f : 0.0f, s_im = (phy->pucch_n_prime_p[sf_num][n1][m] % 2) == 0 ? 0.0f : 1.0f;
                for (uint32 j = 0; j < 12; j++) {
                    for (uint32 i = 0; i < 4; i++) {
                        const float w = (float)W_5_4_1_2[phy->pucch_n_oc_p[sf_num][n1][m]][i];
                        const float rr = phy->pucch_r_u_v_alpha_p_re[sf_num][n1][m][symb[i]][j], ri = phy->pucch_r_u_v_alpha_p_im[sf_num][n1][m][symb[i]][j];
                        // x = d * s * w * r ;  y = h * x
                        const float ar = tc[t].d_re * s_re - tc[t].d_im * s_im, ai = tc[t].d_re * s_im + tc[t].d_im * s_re;
                        const float xr = w * (ar * rr - ai * ri), xi = w * (ar * ri + ai * rr);
                        ps.rx_symb_re[7 * m + symb[i]][prb * 12 + j] = hr * xr - hi * xi;
                        ps.rx_symb_im[7 * m + symb[i]][prb * 12 + j] = hr * xi + hi * xr;
                    }
                    for (uint32 i = 0; i < 3; i++) {
                        const float dr = m == 0 ? phy->pucch_dmrs_0_re[sf_num][n1][i * 12 + j] : phy->pucch_dmrs_1_re[sf_num][n1][i * 12 + j];
                        const float di = m == 0 ? phy->pucch_dmrs_0_im[sf_num][n1][i * 12 + j] : phy->pucch_dmrs_1_im[sf_num][n1][i * 12 + j];
                        ps.rx_symb_re[7 * m + 2 + i][prb * 12 + j] = hr * dr - hi * di;
                        ps.rx_symb_im[7 * m + 2 + i][prb * 12 + j] = hr * di + hi * dr;
                    }
                }
            }
            uint8  b[4] = {0, 0, 0, 0};
            uint32 nb = 0;
            LIBLTE_ERROR_ENUM e = liblte_phy_pucch_format_1_1a_1b_channel_decode(phy, &ps, tc[t].fmt, cell, 1, n1, b, &nb);
            printf("pucch format %s resource %u: err=%d N_out_bits=%u bits=%u%u\n", liblte_phy_pucch_format_text[tc[t].fmt], n1, (int)e, nb, b[0], nb == 2 ? b[1] : 0);
        }
    }
    if (getenv("ENB_DL_TX")) {
        // ... and the other half of the eNodeB's TTI (LTE_fdd_enb_phy.cc:557-770): the downlink subframe it transmits while it receives this one --
        // synchronisation and reference signals, the MIB in subframe 0, a control region with HARQ acknowledgements, a downlink assignment and
        // an uplink grant sized by the scheduler-side searches, the PDSCH, OFDM modulation.  With the receive calls above that is every
        // liblte_phy function LTE_fdd_enodeb calls; the lines must not depend on which build runs them.
        static LIBLTE_PHY_SUBFRAME_STRUCT tx;
        static LIBLTE_PHY_PDCCH_STRUCT    pd;
        static LIBLTE_PHY_PCFICH_STRUCT   pcfich;
        static LIBLTE_PHY_PHICH_STRUCT    phich;
        memset(&tx, 0, sizeof tx), memset(&pd, 0, sizeof pd), memset(&pcfich, 0, sizeof pcfich), memset(&phich, 0, sizeof phich);
        uint32 x = 2463534242u + cell;
        for (uint32 sfn = 0; sfn < 2; sfn++) {
            tx.num = sf_num;
            for (uint32 p = 0; p < 1; p++)
                for (uint32 l = 0; l < 16; l++) memset(tx.tx_symb_re[p][l], 0, sizeof tx.tx_symb_re[p][l]), memset(tx.tx_symb_im[p][l], 0, sizeof tx.tx_symb_im[p][l]);
            if (sf_num == 0 || sf_num == 5) { liblte_phy_map_pss(phy, &tx, cell % 3, 1); liblte_phy_map_sss(phy, &tx, cell / 3, cell % 3, 1); }
            liblte_phy_map_crs(phy, &tx, cell, 1);
            if (sf_num == 0) {
                uint8 mib[24];
                for (int i = 0; i < 24; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; mib[i] = x & 1; }
                liblte_phy_bch_channel_encode(phy, mib, 24, cell, 1, &tx, sfn);
            }
            memset(&pd, 0, sizeof pd), memset(&phich, 0, sizeof phich);
            pcfich.cfi = 2;
            phich.present[0][1] = true, phich.b[0][1] = 1, phich.present[1 % phy->N_group_phich][4] = true;
            LIBLTE_PHY_ALLOCATION_STRUCT &dl = pd.alloc[0], &ul = pd.alloc[1];
            dl.msg[0].N_bits = 300 + 40 * sfn;
            for (uint32 i = 0; i < dl.msg[0].N_bits; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; dl.msg[0].msg[i] = x & 1; }
            dl.rnti = 0x0042, dl.chan_type = LIBLTE_PHY_CHAN_TYPE_DLSCH, dl.pre_coder_type = LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY, dl.mod_type = LIBLTE_PHY_MODULATION_TYPE_QPSK;
            dl.N_codewords = 1, dl.tx_mode = 1, dl.rv_idx = 0, dl.ndi = sfn & 1, dl.tpc = 1;
            const LIBLTE_ERROR_ENUM es = liblte_phy_get_tbs_mcs_and_n_prb_for_dl(dl.msg[0].N_bits, sf_num, phy->N_rb_dl, dl.rnti, &dl.tbs, &dl.mcs, &dl.N_prb);
            for (uint32 i = 0; i < dl.N_prb; i++) dl.prb[0][i] = dl.prb[1][i] = 1 + i;
            ul.rnti = 0x0042, ul.chan_type = LIBLTE_PHY_CHAN_TYPE_ULSCH, ul.ndi = 1, ul.tpc = 2;
            const LIBLTE_ERROR_ENUM eu = liblte_phy_get_tbs_mcs_and_n_prb_for_ul(500, phy->N_rb_ul, &ul.tbs, &ul.mcs, &ul.N_prb);
            ul.prb[0][0] = ul.prb[1][0] = 2;
            pd.N_alloc = 2;
            uint32 n_cce = 0, sr_per = 0, sr_off = 0;
            liblte_phy_get_n_cce(phy, 1.0f, pcfich.cfi, 1, &n_cce);
            liblte_phy_pucch_map_sr_config_idx(17 + sf_num, &sr_per, &sr_off);
            const LIBLTE_ERROR_ENUM e1 = liblte_phy_pdcch_channel_encode(phy, &pcfich, &phich, &pd, cell, 1, 1.0f, LIBLTE_RRC_PHICH_DURATION_NORMAL, &tx);
            const LIBLTE_ERROR_ENUM e2 = liblte_phy_pdsch_channel_encode(phy, &pd, cell, 1, &tx);
            float *ti = (float *)calloc(n + 64, sizeof(float)), *tq = (float *)calloc(n + 64, sizeof(float));
            const LIBLTE_ERROR_ENUM e3 = liblte_phy_create_dl_subframe(phy, &tx, 0, ti, tq);
            uint32 h = 2166136261u; // FNV-1a over the samples as a 12-bit converter would send them
            double pw = 0;
            for (uint32 k = 0; k < n; k++) {
                const int a = (int)lrintf(ti[k] * 16.0f), b = (int)lrintf(tq[k] * 16.0f);
                h = (h ^ (uint32)(a & 0xFFFF)) * 16777619u, h = (h ^ (uint32)(b & 0xFFFF)) * 16777619u;
                pw += (double)ti[k] * ti[k] + (double)tq[k] * tq[k];
            }
            printf("dl tx frame %u subframe %u: searches %d %d (tbs %u mcs %u N_prb %u | tbs %u mcs %u N_prb %u) n_cce %u sr %u/%u encode %d %d %d N_symbs %u power %.5e samples %08x\n", sfn, sf_num,
                   (int)es, (int)eu, dl.tbs, (unsigned)dl.mcs, dl.N_prb, ul.tbs, (unsigned)ul.mcs, ul.N_prb, n_cce, sr_per, sr_off, (int)e1, (int)e2, (int)e3, pd.N_symbs, pw, h);
            free(ti), free(tq);
        }
    }
    liblte_phy_cleanup(phy);
    return 0;
}
