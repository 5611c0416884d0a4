// capture_gen.cc -- writes an int8 I,Q downlink capture (the file format LTE_fdd_dl_file_scan reads,
// LTE_fdd_dl_fs_samp_buf.cc:657-694) of a small synthetic cell: PSS/SSS/CRS, PBCH with the MIB, SIB1 in subframe 5
// of even frames and SIB2 in subframe 3 of every 8th frame, each with its PCFICH/PDCCH.  It plays the role of
// LTE_fdd_dl_file_gen for the drop-in scan test (BASELINE config 1 / SURVEY 8d W1) and, like that tool, is nothing
// but a caller of the reference's liblte_phy / liblte_rrc TX API -- linked against the unmodified reference objects.
//
//   capture_gen <out.bin> <N_rb_dl: 6|15|25|50|75|100> <N_id_cell> <N_frames> [carrier offset in Hz] [leading samples]
// (a carrier offset rotates sample k of the file by exp(+j*2*pi*f*k/fs) before the int8 conversion: what the scanner's coarse timing measures
// and its freq_shift takes out again, LTE_fdd_dl_fs_samp_buf.cc:239, :696-713; leading samples: the file starts that many samples before a frame)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "liblte_mac.h"
#include "liblte_phy.h"
#include "liblte_rrc.h"

struct Cell {
    LIBLTE_PHY_STRUCT *phy;
    uint32             N_rb_dl, N_id_cell, N_ant;
    float              phich_res;
    LIBLTE_RRC_MIB_STRUCT mib;
};

// one SI message (already packed) -> allocation a of the subframe's PDCCH list
static bool add_si(Cell &c, LIBLTE_PHY_PDCCH_STRUCT *pd, LIBLTE_RRC_BCCH_DLSCH_MSG_STRUCT *msg, uint32 sf, uint32 rv)
{
    LIBLTE_PHY_ALLOCATION_STRUCT *al = &pd->alloc[pd->N_alloc];
    liblte_rrc_pack_bcch_dlsch_msg(msg, &al->msg[0]);
    if (LIBLTE_SUCCESS != liblte_phy_get_tbs_mcs_and_n_prb_for_dl(al->msg[0].N_bits, sf, c.N_rb_dl, LIBLTE_MAC_SI_RNTI, &al->tbs, &al->mcs, &al->N_prb))
    {
        fprintf(stderr, "SI message of %u bits does not fit subframe %u at %u RB\n", al->msg[0].N_bits, sf, c.N_rb_dl);
        return false;
    }
    al->pre_coder_type = LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY;
    al->mod_type       = LIBLTE_PHY_MODULATION_TYPE_QPSK;
    al->rv_idx         = rv;
    al->N_codewords    = 1;
    al->rnti           = LIBLTE_MAC_SI_RNTI;
    al->tx_mode        = 1;
    pd->N_alloc++;
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: capture_gen <out.bin> <N_rb_dl> <N_id_cell> <N_frames>\n"); return 2; }
    Cell c;
    c.N_rb_dl = atoi(argv[2]); c.N_id_cell = atoi(argv[3]); c.N_ant = 1;
    const uint32 n_frames = atoi(argv[4]);
    const double cfo_hz   = argc > 5 ? atof(argv[5]) : 0.0;
    const uint32 lead     = argc > 6 ? atoi(argv[6]) : 0;
    LIBLTE_PHY_FS_ENUM       fs;
    LIBLTE_RRC_DL_BANDWIDTH_ENUM bw;
    switch (c.N_rb_dl) {
    case 6:  fs = LIBLTE_PHY_FS_1_92MHZ;  bw = LIBLTE_RRC_DL_BANDWIDTH_6;  break;
    case 15: fs = LIBLTE_PHY_FS_3_84MHZ;  bw = LIBLTE_RRC_DL_BANDWIDTH_15; break;
    case 25: fs = LIBLTE_PHY_FS_7_68MHZ;  bw = LIBLTE_RRC_DL_BANDWIDTH_25; break;
    case 50: fs = LIBLTE_PHY_FS_15_36MHZ; bw = LIBLTE_RRC_DL_BANDWIDTH_50; break;
    case 75: fs = LIBLTE_PHY_FS_30_72MHZ; bw = LIBLTE_RRC_DL_BANDWIDTH_75; break;
    case 100: fs = LIBLTE_PHY_FS_30_72MHZ; bw = LIBLTE_RRC_DL_BANDWIDTH_100; break;
    default: fprintf(stderr, "bad N_rb_dl\n"); return 2;
    }
    c.mib.dl_bw = bw; c.mib.phich_config.dur = LIBLTE_RRC_PHICH_DURATION_NORMAL; c.mib.phich_config.res = LIBLTE_RRC_PHICH_RESOURCE_1;
    c.phich_res = liblte_rrc_phich_resource_num[c.mib.phich_config.res];
    if (LIBLTE_SUCCESS != liblte_phy_init(&c.phy, fs, c.N_id_cell, c.N_ant, c.N_rb_dl, LIBLTE_PHY_N_SC_RB_DL_NORMAL_CP, c.phich_res)) return 3;

    // system information of the synthetic cell
    static LIBLTE_RRC_BCCH_DLSCH_MSG_STRUCT si;
    static LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1_STRUCT sib1;
    static LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2_STRUCT sib2;
    memset(&sib1, 0, sizeof(sib1));
    memset(&sib2, 0, sizeof(sib2));
    sib1.N_plmn_ids = 1; sib1.plmn_id[0].id.mcc = 0xF310; sib1.plmn_id[0].id.mnc = 0xFF26; sib1.plmn_id[0].resv_for_oper = LIBLTE_RRC_NOT_RESV_FOR_OPER;
    sib1.N_sched_info = 1; sib1.sched_info[0].N_sib_mapping_info = 0; sib1.sched_info[0].si_periodicity = LIBLTE_RRC_SI_PERIODICITY_RF8;
    sib1.cell_barred = LIBLTE_RRC_CELL_NOT_BARRED; sib1.intra_freq_reselection = LIBLTE_RRC_INTRA_FREQ_RESELECTION_ALLOWED;
    sib1.si_window_length = LIBLTE_RRC_SI_WINDOW_LENGTH_MS2; sib1.cell_id = 0x00ABCDE; sib1.tracking_area_code = 0x0107;
    sib1.q_rx_lev_min = -124; sib1.q_rx_lev_min_offset = 1; sib1.freq_band_indicator = 7; sib1.system_info_value_tag = 3;
    sib1.p_max_present = true; sib1.p_max = 23; sib1.tdd = false;
    sib2.rr_config_common_sib.rach_cnfg.num_ra_preambles = LIBLTE_RRC_NUMBER_OF_RA_PREAMBLES_N52;
    sib2.rr_config_common_sib.rach_cnfg.max_harq_msg3_tx = 4;
    sib2.rr_config_common_sib.prach_cnfg.root_sequence_index = 22;
    sib2.rr_config_common_sib.prach_cnfg.prach_cnfg_info.prach_config_index = 3;
    sib2.rr_config_common_sib.prach_cnfg.prach_cnfg_info.zero_correlation_zone_config = 11;
    sib2.rr_config_common_sib.prach_cnfg.prach_cnfg_info.prach_freq_offset = 2;
    sib2.rr_config_common_sib.pdsch_cnfg.rs_power = 15; sib2.rr_config_common_sib.pdsch_cnfg.p_b = 1;
    sib2.rr_config_common_sib.pusch_cnfg.n_sb = 1; sib2.rr_config_common_sib.pusch_cnfg.enable_64_qam = false;
    sib2.rr_config_common_sib.pusch_cnfg.ul_rs.group_assignment_pusch = 3; sib2.rr_config_common_sib.pusch_cnfg.ul_rs.cyclic_shift = 2;
    sib2.rr_config_common_sib.pucch_cnfg.n1_pucch_an = 36;
    sib2.rr_config_common_sib.ul_pwr_ctrl.p0_nominal_pusch = -85; sib2.rr_config_common_sib.ul_pwr_ctrl.p0_nominal_pucch = -110;
    sib2.rr_config_common_sib.ul_pwr_ctrl.delta_preamble_msg3 = 4;
    sib2.additional_spectrum_emission = 1; sib2.time_alignment_timer = LIBLTE_RRC_TIME_ALIGNMENT_TIMER_SF1920;

    static LIBLTE_PHY_SUBFRAME_STRUCT sfm;
    static LIBLTE_PHY_PCFICH_STRUCT   pcfich;
    static LIBLTE_PHY_PHICH_STRUCT    phich;
    static LIBLTE_PHY_PDCCH_STRUCT    pd;
    static LIBLTE_BIT_MSG_STRUCT      bch;
    const uint32 n_sf = c.phy->N_samps_per_subfr;
    float *ti = (float *)calloc(n_sf + 64, sizeof(float)), *tq = (float *)calloc(n_sf + 64, sizeof(float));
    FILE  *f  = fopen(argv[1], "wb");
    if (!f) return 4;
    for (uint32 k = 0; k < lead; k++) { signed char z[2] = {0, 0}; fwrite(z, 1, 2, f); }
    unsigned long long k_abs = lead;
    const double       w     = 2.0 * M_PI * cfo_hz / (double)c.phy->fs;
    memset(&phich, 0, sizeof(phich)); // no HARQ indicators
    memset(&pcfich, 0, sizeof(pcfich));
    pcfich.cfi = 2;                   // the caller chooses the control format indicator (liblte_phy.cc:4171)
    for (uint32 sfn = 0; sfn < n_frames; sfn++)
        for (uint32 sf = 0; sf < 10; sf++) {
            memset(sfm.tx_symb_re, 0, sizeof(sfm.tx_symb_re));
            memset(sfm.tx_symb_im, 0, sizeof(sfm.tx_symb_im));
            sfm.num = sf;
            if (sf == 0 || sf == 5) {
                liblte_phy_map_pss(c.phy, &sfm, c.N_id_cell % 3, c.N_ant);
                liblte_phy_map_sss(c.phy, &sfm, c.N_id_cell / 3, c.N_id_cell % 3, c.N_ant);
            }
            liblte_phy_map_crs(c.phy, &sfm, c.N_id_cell, c.N_ant);
            if (sf == 0) {
                c.mib.sfn_div_4 = sfn / 4;
                liblte_rrc_pack_bcch_bch_msg(&c.mib, &bch);
                liblte_phy_bch_channel_encode(c.phy, bch.msg, bch.N_bits, c.N_id_cell, c.N_ant, &sfm, sfn);
            }
            pd.N_alloc = 0;
            if (sf == 5 && (sfn % 2) == 0) { // SIB1: subframe 5 of even frames, rv per 36.321 5.3.1
                si.N_sibs = 0; si.sibs[0].sib_type = LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1;
                memcpy(&si.sibs[0].sib, &sib1, sizeof(sib1));
                add_si(c, &pd, &si, sf, (uint32)ceilf(1.5f * ((sfn / 2) % 4)) % 4);
            } else if (sf == 3 && (sfn % 8) == 0) { // SIB2 every 8th frame (a subframe in which the reference decodes its own PDCCH for this size)
                si.N_sibs = 1; si.sibs[0].sib_type = LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2;
                memcpy(&si.sibs[0].sib, &sib2, sizeof(sib2));
                add_si(c, &pd, &si, sf, 0);
            }
            if (pd.N_alloc) {
                uint32 next = 0;
                for (uint32 a = 0; a < pd.N_alloc; a++)
                    for (uint32 j = 0; j < pd.alloc[a].N_prb; j++) pd.alloc[a].prb[0][j] = pd.alloc[a].prb[1][j] = next++;
                const int e1 = liblte_phy_pdcch_channel_encode(c.phy, &pcfich, &phich, &pd, c.N_id_cell, c.N_ant, c.phich_res, c.mib.phich_config.dur, &sfm);
                const int e2 = liblte_phy_pdsch_channel_encode(c.phy, &pd, c.N_id_cell, c.N_ant, &sfm);
                if (e1 || e2 || getenv("CAPTURE_GEN_VERBOSE"))
                    fprintf(stderr, "sfn %u sf %u: %u bits tbs %u N_prb %u -> pdcch %d pdsch %d\n", sfn, sf, pd.alloc[0].msg[0].N_bits, pd.alloc[0].tbs, pd.alloc[0].N_prb, e1, e2);
            }
            liblte_phy_create_dl_subframe(c.phy, &sfm, 0, ti, tq);
            for (uint32 k = 0; k < n_sf; k++, k_abs++) {
                float a = ti[k], b = tq[k];
                if (cfo_hz != 0.0) {
                    const double cr = cos(w * (double)k_abs), ci = sin(w * (double)k_abs);
                    a = (float)(ti[k] * cr - tq[k] * ci);
                    b = (float)(tq[k] * cr + ti[k] * ci);
                }
                signed char v[2] = {(signed char)a, (signed char)b};
                fwrite(v, 1, 2, f);
            }
        }
    fclose(f);
    liblte_phy_cleanup(c.phy);
    return 0;
}
