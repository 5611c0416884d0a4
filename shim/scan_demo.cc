// scan_demo.cc -- a GNU-Radio-free cell scan over an int8 I,Q capture, in the order LTE_fdd_dl_file_scan walks
// the liblte_phy API (LTE_fdd_dl_fs_samp_buf.cc:277-600): coarse timing -> PSS + fine timing -> SSS -> PBCH/MIB ->
// PCFICH/PDCCH + PDSCH for SIB1 -> every subframe of the following frames for the other SI messages.  It is written
// purely against the reference's public headers and linked twice by shim/Makefile: against the unmodified reference
// objects (scan_cpu), and against the same objects with liblte_phy_get_dl_subframe_and_ce / liblte_phy_pdsch_channel_decode
// (and the other shimmed symbols) coming from liblte_phy_shim.cc -> libmi_lte.so (scan_gpu).  Both must print the same report
// (BASELINE config 1 / SURVEY 8d W1: "plumbing only").
//
//   scan_* <capture.bin> <fs: 1.92|3.84|7.68|15.36|30.72> [max SI frames]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "liblte_mac.h"
#include "liblte_phy.h"
#include "liblte_rrc.h"

static double now_s()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Scan {
    LIBLTE_PHY_STRUCT *phy;
    float             *i, *q;
    uint32             n;
};

// the scanner's own frequency correction (LTE_fdd_dl_fs_samp_buf.cc:696-713), applied to the whole buffer
static void derotate(Scan &s, float f_off)
{
    for (uint32 k = 0; k < s.n; k++) {
        const float cr = cosf((k + 1) * f_off * 2 * M_PI / s.phy->fs), ci = sinf((k + 1) * f_off * 2 * M_PI / s.phy->fs);
        const float a = s.i[k], b = s.q[k];
        s.i[k] = a * cr + b * ci;
        s.q[k] = b * cr - a * ci;
    }
}

static void report_sib1(const LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1_STRUCT *b)
{
    printf("SIB1: plmn=%03x-%02x tac=0x%04x cell_identity=0x%07x barred=%d q_rx_lev_min=%d band=%u value_tag=%u si_window=%d",
           b->plmn_id[0].id.mcc & 0xFFF, b->plmn_id[0].id.mnc & 0xFF, b->tracking_area_code, b->cell_id, (int)b->cell_barred,
           (int)b->q_rx_lev_min, b->freq_band_indicator, b->system_info_value_tag, (int)b->si_window_length);
    if (b->p_max_present) printf(" p_max=%d", (int)b->p_max);
    printf(" n_sched=%u\n", b->N_sched_info);
}
static void report_sib2(const LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2_STRUCT *b)
{
    const LIBLTE_RRC_RR_CONFIG_COMMON_SIB_STRUCT *r = &b->rr_config_common_sib;
    printf("SIB2: ra_preambles=%d msg3_harq=%u prach_root=%u prach_cfg=%u zczc=%u prach_freq_offset=%u rs_power=%d p_b=%u "
           "pusch_group_assignment=%u ul_cyclic_shift=%u n1_pucch_an=%u p0_pusch=%d p0_pucch=%d delta_msg3=%d ta_timer=%d\n",
           (int)r->rach_cnfg.num_ra_preambles, r->rach_cnfg.max_harq_msg3_tx, r->prach_cnfg.root_sequence_index,
           r->prach_cnfg.prach_cnfg_info.prach_config_index, r->prach_cnfg.prach_cnfg_info.zero_correlation_zone_config,
           r->prach_cnfg.prach_cnfg_info.prach_freq_offset, (int)r->pdsch_cnfg.rs_power, r->pdsch_cnfg.p_b,
           r->pusch_cnfg.ul_rs.group_assignment_pusch, r->pusch_cnfg.ul_rs.cyclic_shift, r->pucch_cnfg.n1_pucch_an,
           (int)r->ul_pwr_ctrl.p0_nominal_pusch, (int)r->ul_pwr_ctrl.p0_nominal_pucch, (int)r->ul_pwr_ctrl.delta_preamble_msg3,
           (int)b->time_alignment_timer);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: scan <capture.bin> <fs MHz> [max SI frames] [int8 | gr_complex]\n"); return 2; }
    // the two formats LTE_fdd_dl_file_scan reads (LTE_fdd_dl_fs_samp_buf.cc:657-694): int8 I,Q pairs (default) or gr_complex = float32 pairs
    const bool gr_complex = argc > 4 && strcmp(argv[4], "gr_complex") == 0;
    const double fs_mhz = atof(argv[2]);
    const LIBLTE_PHY_FS_ENUM fs = fs_mhz < 2 ? LIBLTE_PHY_FS_1_92MHZ : fs_mhz < 4 ? LIBLTE_PHY_FS_3_84MHZ : fs_mhz < 8 ? LIBLTE_PHY_FS_7_68MHZ
                                : fs_mhz < 16 ? LIBLTE_PHY_FS_15_36MHZ : LIBLTE_PHY_FS_30_72MHZ;
    const uint32 si_frames = argc > 3 ? atoi(argv[3]) : 9;
    Scan s;
    if (LIBLTE_SUCCESS != liblte_phy_init(&s.phy, fs, LIBLTE_PHY_INIT_N_ID_CELL_UNKNOWN, 4, LIBLTE_PHY_N_RB_DL_1_4MHZ, LIBLTE_PHY_N_SC_RB_DL_NORMAL_CP,
                                          liblte_rrc_phich_resource_num[LIBLTE_RRC_PHICH_RESOURCE_1]))
        return 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 4;
    fseek(f, 0, SEEK_END);
    s.n = (uint32)(ftell(f) / (gr_complex ? 8 : 2));
    fseek(f, 0, SEEK_SET);
    const uint32 pad = 2 * s.phy->N_samps_per_frame;
    s.i = (float *)calloc(s.n + pad, sizeof(float));
    s.q = (float *)calloc(s.n + pad, sizeof(float));
    for (uint32 k = 0; k < s.n; k++) {
        if (gr_complex) {
            float v[2];
            if (fread(v, 4, 2, f) != 2) break;
            s.i[k] = v[0];
            s.q[k] = v[1];
        } else {
            signed char v[2];
            if (fread(v, 1, 2, f) != 2) break;
            s.i[k] = v[0];
            s.q[k] = v[1];
        }
    }
    fclose(f);
    const uint32 n_frame = s.phy->N_samps_per_frame, n_subfr = s.phy->N_samps_per_subfr;
    printf("capture: %u samples (%.1f frames) at %s Hz\n", s.n, (double)s.n / n_frame, liblte_phy_fs_text[fs]);

    // wall-clock of the phases goes to stderr (the report on stdout is what the two builds are compared on): the first liblte_phy call
    // of the GPU build carries the HIP start-up (context, code objects), the per-subframe loops do not
    const double t_start = now_s();
    double       t_subframes = 0;
    uint32       n_subframes = 0;
    static LIBLTE_PHY_COARSE_TIMING_STRUCT timing;
    if (LIBLTE_SUCCESS != liblte_phy_dl_find_coarse_timing_and_freq_offset(s.phy, s.i, s.q, 160, &timing)) { printf("no coarse timing\n"); return 1; }
    const double t_coarse = now_s() - t_start;
    printf("coarse timing: %u correlation peak(s)\n", timing.n_corr_peaks);
    static LIBLTE_PHY_SUBFRAME_STRUCT       sf;
    static LIBLTE_PHY_PCFICH_STRUCT         pcfich;
    static LIBLTE_PHY_PHICH_STRUCT          phich;
    static LIBLTE_PHY_PDCCH_STRUCT          pdcch;
    static LIBLTE_BIT_MSG_STRUCT            msg;
    static LIBLTE_RRC_MIB_STRUCT            mib;
    static LIBLTE_RRC_BCCH_DLSCH_MSG_STRUCT si;
    int    cells = 0;
    uint32 seen[8];
    float  applied = 0;
    for (uint32 p = 0; p < timing.n_corr_peaks; p++) {
        derotate(s, timing.freq_offset[p] - applied);
        applied = timing.freq_offset[p];
        uint32 N_id_2, N_id_1, pss_symb, frame_start;
        float  pss_thresh, f_off;
        if (LIBLTE_SUCCESS != liblte_phy_find_pss_and_fine_timing(s.phy, s.i, s.q, timing.symb_starts[p], &N_id_2, &pss_symb, &pss_thresh, &f_off)) continue;
        if (fabs(f_off) > 100) { derotate(s, f_off); applied += f_off; }
        if (LIBLTE_SUCCESS != liblte_phy_find_sss(s.phy, s.i, s.q, N_id_2, timing.symb_starts[p], pss_thresh, &N_id_1, &frame_start)) continue;
        const uint32 N_id_cell = 3 * N_id_1 + N_id_2;
        bool         dup = false;
        for (int k = 0; k < cells; k++) dup |= seen[k] == N_id_cell; // another correlation peak of a cell already reported
        if (dup || cells == 8) continue;
        uint8 N_ant, sfn_off;
        if (!(LIBLTE_SUCCESS == liblte_phy_get_dl_subframe_and_ce(s.phy, s.i, s.q, frame_start, 0, N_id_cell, 4, &sf) &&
              LIBLTE_SUCCESS == liblte_phy_bch_channel_decode(s.phy, &sf, N_id_cell, &N_ant, msg.msg, &msg.N_bits, &sfn_off) &&
              LIBLTE_SUCCESS == liblte_rrc_unpack_bcch_bch_msg(&msg, &mib)))
            continue;
        static const uint32 rb_of_bw[6] = {6, 15, 25, 50, 75, 100};
        const uint32 N_rb_dl = rb_of_bw[mib.dl_bw];
        liblte_phy_update_n_rb_dl(s.phy, N_rb_dl);
        uint32      sfn       = (mib.sfn_div_4 << 2) + sfn_off;
        const float phich_res = liblte_rrc_phich_resource_num[mib.phich_config.res];
        printf("cell %u: frame start %u, %u antenna port(s), MIB: N_rb_dl=%u phich_dur=%d phich_res=%d sfn=%u\n", N_id_cell,
               frame_start % n_frame, (unsigned)N_ant, N_rb_dl, (int)mib.phich_config.dur, (int)mib.phich_config.res, sfn);
        seen[cells++] = N_id_cell;
        // SIB1 sits in subframe 5 of even frames
        uint32 r = frame_start;
        if (sfn % 2) { r += n_frame; sfn++; }
        bool got_sib1 = false;
        for (; !got_sib1 && r + 2 * n_frame < s.n; r += 2 * n_frame, sfn += 2) {
            if (LIBLTE_SUCCESS == liblte_phy_get_dl_subframe_and_ce(s.phy, s.i, s.q, r, 5, N_id_cell, N_ant, &sf) &&
                LIBLTE_SUCCESS == liblte_phy_pdcch_channel_decode(s.phy, &sf, N_id_cell, N_ant, phich_res, mib.phich_config.dur, &pcfich, &phich, &pdcch) &&
                LIBLTE_SUCCESS == liblte_phy_pdsch_channel_decode(s.phy, &sf, &pdcch.alloc[0], pdcch.N_symbs, N_id_cell, N_ant, msg.msg, &msg.N_bits) &&
                LIBLTE_SUCCESS == liblte_rrc_unpack_bcch_dlsch_msg(&msg, &si) && si.N_sibs == 1 && si.sibs[0].sib_type == LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1) {
                printf("sfn %u subframe 5: CFI=%u tbs=%u N_prb=%u rv=%u -> ", sfn, pdcch.N_symbs, pdcch.alloc[0].tbs, pdcch.alloc[0].N_prb, pdcch.alloc[0].rv_idx);
                report_sib1((LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1_STRUCT *)&si.sibs[0].sib);
                got_sib1 = true;
            }
        }
        if (!got_sib1) { printf("cell %u: SIB1 not found\n", N_id_cell); continue; }
        // every subframe of the following frames: any other system information
        uint32 n_pdsch = 0, n_fail = 0;
        bool   got_sib2 = false;
        const double t_loop = now_s();
        for (uint32 fr = 0; fr < si_frames && r + n_frame + n_subfr < s.n; fr++, r += n_frame, sfn++)
            for (uint32 n = 0; n < 10; n++) {
                n_subframes++;
                if (LIBLTE_SUCCESS != liblte_phy_get_dl_subframe_and_ce(s.phy, s.i, s.q, r, n, N_id_cell, N_ant, &sf)) continue;
                if (LIBLTE_SUCCESS != liblte_phy_pdcch_channel_decode(s.phy, &sf, N_id_cell, N_ant, phich_res, mib.phich_config.dur, &pcfich, &phich, &pdcch)) continue;
                if (LIBLTE_SUCCESS != liblte_phy_pdsch_channel_decode(s.phy, &sf, &pdcch.alloc[0], pdcch.N_symbs, N_id_cell, N_ant, msg.msg, &msg.N_bits)) { n_fail++; continue; }
                n_pdsch++;
                if (LIBLTE_MAC_SI_RNTI != pdcch.alloc[0].rnti || LIBLTE_SUCCESS != liblte_rrc_unpack_bcch_dlsch_msg(&msg, &si)) continue;
                for (uint32 k = 0; k < si.N_sibs; k++)
                    if (si.sibs[k].sib_type == LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2 && !got_sib2) {
                        printf("sfn %u subframe %u: CFI=%u tbs=%u N_prb=%u -> ", sfn, n, pdcch.N_symbs, pdcch.alloc[0].tbs, pdcch.alloc[0].N_prb);
                        report_sib2((LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2_STRUCT *)&si.sibs[k].sib);
                        got_sib2 = true;
                    }
            }
        t_subframes += now_s() - t_loop;
        printf("cell %u: %u PDSCH transport blocks decoded after SIB1, %u with a PDCCH but a failed CRC\n", N_id_cell, n_pdsch, n_fail);
    }
    printf("%d cell(s) found\n", cells);
    fprintf(stderr, "timing: total %.4f s; first call (coarse timing, with any start-up) %.4f s; per-subframe loop (get_dl_subframe_and_ce + pdcch + pdsch) "
                    "%.4f s for %u subframes = %.1f us per subframe\n",
            now_s() - t_start, t_coarse, t_subframes, n_subframes, n_subframes ? 1e6 * t_subframes / n_subframes : 0.0);
    liblte_phy_cleanup(s.phy);
    return cells ? 0 : 1;
}
