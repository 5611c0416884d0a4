// scan_batch.cc -- the cell scan of scan_demo.cc (LTE_fdd_dl_file_scan's state machine, LTE_fdd_dl_fs_samp_buf.cc:277-600) written
// the way the BATCH entry points of include/mi_lte.h are meant to be used: the capture goes to HBM once and stays there; every loop
// over subframes of the per-call scanner is one launch per stage --
//
//     int8 -> planar float (:657-694), coarse timing, [frequency correction (:696-713) on the device], PSS + fine timing, SSS
//     PBCH                 : front end (4-port estimates) of subframe 0 at the frame start -> mi_lte_pbch_decode_run
//     SIB1 search          : subframe 5 of EVERY even frame at once: front end -> PDCCH -> one dynamic PDSCH plan over the DCIs found
//     other SI, all frames : every subframe of the following frames at once, the same three stages, one wait per stage
//
// -- and only CFIs, DCIs, verdicts and transport blocks come back.  No liblte_phy symbol is linked: the PHY is libmi_lte.so alone; the
// reference contributes its RRC unpackers (liblte_rrc.cc) for the report, exactly as in scan_demo.cc.  The report on stdout must equal
// scan_cpu's (tests/test_dropin_gpu.py): the decisions are taken in the per-call scanner's order on the batch's results.
//
//   scan_batch <capture.bin> <fs: 1.92|3.84|7.68|15.36|30.72> [max SI frames]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <initializer_list>
#include <vector>

#include "liblte_mac.h"
#include "liblte_rrc.h"
#include "mi_lte.h"

static double now_s()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define CK(call)                                                                                                     \
    do {                                                                                                             \
        const int rc_ = (call);                                                                                      \
        if (rc_ != MI_LTE_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, mi_lte_last_error(ctx)); exit(5); } \
    } while (0)

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

static void probe(mi_lte_ctx *ctx, const char *where)
{
    if (!getenv("MI_LTE_SCAN_PROBE")) return;
    void *d; std::vector<uint8_t> h(30720);
    mi_lte_malloc(ctx, 30720, &d); mi_lte_memset(ctx, d, 1, 30720); mi_lte_sync(ctx);
    double t0 = now_s(); mi_lte_memcpy_d2h(ctx, h.data(), d, 30720);
    fprintf(stderr, "  probe %-22s: 30 KB to the host in %.0f us\n", where, 1e6 * (now_s() - t0));
    mi_lte_free(ctx, d);
}
// One batch of subframes of one cell through front end -> PCFICH / PDCCH -> PDSCH of the first allocation of every subframe whose
// control region decoded (what the per-call scanner hands to liblte_phy_pdsch_channel_decode: pdcch.alloc[0], pdcch.N_symbs).
struct Batch {
    std::vector<uint64_t>         start;  // sample index of each unit's subframe
    std::vector<uint32_t>         sf;     // its subframe number
    std::vector<uint32_t>         rc, cfi, n_symbs, n_dci; // PDCCH results per unit
    std::vector<mi_lte_pdcch_dci> dci;
    std::vector<int32_t>          status; // PDSCH verdict per unit (-1: no PDSCH decode was attempted)
    std::vector<uint8_t>          bits;   // [unit][stride] one bit per byte
    uint32_t                      stride = 0;
};

// a device array that is kept between batches and only ever grows (hipFree costs 0.2 ms a call, a plan's set-up ten of them)
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
    void *get(mi_lte_ctx *ctx, size_t bytes)
    {
        if (bytes > cap) {
            if (p) mi_lte_free(ctx, p);
            cap = bytes + bytes / 2;
            if (mi_lte_malloc(ctx, cap, &p) != MI_LTE_OK) { fprintf(stderr, "mi_lte_malloc failed: %s\n", mi_lte_last_error(ctx)); exit(5); }
        }
        return p;
    }
    void release(mi_lte_ctx *ctx) { if (p) mi_lte_free(ctx, p); p = nullptr; cap = 0; }
};

struct Scanner {
    mi_lte_ctx *ctx = nullptr;
    float      *d_i = nullptr, *d_q = nullptr;
    uint32_t    fft = 0, fs_hz = 0, n = 0;
    double      t_stage[3] = {0, 0, 0};
    // kept from batch to batch: the unit arrays, the subframes, the outputs, and the two plans (re-made when the cell's geometry or a
    // capacity changes -- i.e. once per cell here)
    DevBuf             b_start, b_sf, b_cell, b_sub, b_out, b_st;
    mi_lte_pdcch_plan *cp = nullptr;
    mi_lte_pdsch_plan *pp = nullptr;
    mi_lte_dl_cfg      cp_cfg = {0, 0, 0, 0}, pp_cfg = {0, 0, 0, 0};
    uint32_t           cp_cell = ~0u, pp_alloc = 0;
    float              cp_res = -1;
    size_t             pp_soft = 0;

    static bool same(const mi_lte_dl_cfg &a, const mi_lte_dl_cfg &b) { return a.fft_size == b.fft_size && a.N_rb_dl == b.N_rb_dl && a.N_ant == b.N_ant && a.sample_format == b.sample_format; }
    void release()
    {
        if (cp) mi_lte_pdcch_plan_destroy(ctx, cp);
        if (pp) mi_lte_pdsch_plan_destroy(ctx, pp);
        cp = nullptr; pp = nullptr;
        for (DevBuf *d : {&b_start, &b_sf, &b_cell, &b_sub, &b_out, &b_st}) d->release(ctx);
    }

    void run(Batch &b, uint32_t N_rb_dl, uint32_t N_ant, uint32_t cell, float phich_res)
    {
        const uint32_t nu = (uint32_t)b.start.size();
        b.rc.assign(nu, 1); b.cfi.assign(nu, 0); b.n_symbs.assign(nu, 0); b.n_dci.assign(nu, 0);
        b.dci.assign((size_t)nu * MI_LTE_PDCCH_MAX_DCI, mi_lte_pdcch_dci());
        b.status.assign(nu, -1);
        if (nu == 0) return;
        mi_lte_dl_cfg cfg = {fft, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
        std::vector<uint32_t> cells(nu, cell);
        uint64_t *d_start = (uint64_t *)b_start.get(ctx, sizeof(uint64_t) * nu);
        uint32_t *d_sf = (uint32_t *)b_sf.get(ctx, sizeof(uint32_t) * nu), *d_cell = (uint32_t *)b_cell.get(ctx, sizeof(uint32_t) * nu);
        float    *d_sub = (float *)b_sub.get(ctx, sizeof(float) * mi_lte_subframe_floats(N_ant) * nu);
        CK(mi_lte_memcpy_h2d(ctx, d_start, b.start.data(), sizeof(uint64_t) * nu));
        CK(mi_lte_memcpy_h2d(ctx, d_sf, b.sf.data(), sizeof(uint32_t) * nu));
        CK(mi_lte_memcpy_h2d(ctx, d_cell, cells.data(), sizeof(uint32_t) * nu));
        double t0 = now_s();
        CK(mi_lte_dl_frontend_batch(ctx, &cfg, d_i, d_q, d_start, d_sf, d_cell, nu, d_sub));
        CK(mi_lte_sync(ctx));
        probe(ctx, "after front end");
        t_stage[0] += now_s() - t0; t0 = now_s();
        if (!cp || !same(cfg, cp_cfg) || cell != cp_cell || phich_res != cp_res) {
            if (cp) mi_lte_pdcch_plan_destroy(ctx, cp);
            CK(mi_lte_pdcch_plan_create(ctx, &cfg, phich_res, 0, 0, &cell, 1, &cp));
            cp_cfg = cfg; cp_cell = cell; cp_res = phich_res;
        }
        CK(mi_lte_pdcch_decode_run(ctx, cp, d_sub, d_sf, d_cell, nu, b.rc.data(), b.cfi.data(), b.n_symbs.data(), b.n_dci.data(), b.dci.data()));
        probe(ctx, "after pdcch");
        t_stage[1] += now_s() - t0; t0 = now_s();
        // the first allocation of every subframe whose control region decoded; transport blocks of more than one code block are outside
        // the envelope (the reference's own path for them is broken) and count as failed decodes
        std::vector<mi_lte_pdsch_alloc> al;
        std::vector<uint32_t>           unit_of;
        size_t                          soft = 0;
        for (uint32_t u = 0; u < nu; u++) {
            if (b.rc[u] != 0 || b.n_dci[u] == 0) continue;
            const mi_lte_pdsch_alloc &a = b.dci[(size_t)u * MI_LTE_PDCCH_MAX_DCI].alloc;
            b.status[u] = MI_LTE_DECODE_FAIL;
            if (!mi_lte_pdsch_alloc_decodable(&cfg, &a, 2)) continue; // (a chance-CRC DCI must not end the scan: scan_cpu goes on, too)
            al.push_back(a);
            al.back().unit = u;
            unit_of.push_back(u);
            const uint32_t qm = a.mod_type == 3 ? 6 : a.mod_type == 2 ? 4 : a.mod_type == 1 ? 2 : 1;
            soft += ((size_t)(14 - a.n_pdcch_symbs) * a.N_prb * 12 * qm + 63) & ~(size_t)63;
        }
        if (!al.empty()) {
            if (!pp || !same(cfg, pp_cfg) || al.size() > pp_alloc || soft > pp_soft) { // room for every subframe of a batch of this size to carry a full-band QPSK block
                if (pp) mi_lte_pdsch_plan_destroy(ctx, pp);
                pp_alloc = std::max<uint32_t>((uint32_t)al.size(), nu);
                pp_soft  = std::max(soft, (size_t)pp_alloc * (((size_t)13 * N_rb_dl * 12 * 2 + 63) & ~(size_t)63));
                CK(mi_lte_pdsch_plan_create_dynamic(ctx, &cfg, pp_alloc, pp_soft, &pp));
                pp_cfg = cfg;
            }
            const double ta = now_s();
            CK(mi_lte_pdsch_plan_assign(ctx, pp, 2, al.data(), (uint32_t)al.size()));
            b.stride = mi_lte_pdsch_plan_out_stride(pp);
            uint8_t *d_out = (uint8_t *)b_out.get(ctx, (size_t)al.size() * b.stride);
            int32_t *d_st = (int32_t *)b_st.get(ctx, sizeof(int32_t) * al.size());
            const double tb = now_s();
            CK(mi_lte_pdsch_decode_run(ctx, pp, d_sub, d_sf, d_cell, d_out, d_st));
            const double tc = now_s();
            std::vector<int32_t> st(al.size());
            std::vector<uint8_t> ob((size_t)al.size() * b.stride);
            CK(mi_lte_memcpy_d2h(ctx, st.data(), d_st, sizeof(int32_t) * al.size()));
            const double td = now_s();
            CK(mi_lte_memcpy_d2h(ctx, ob.data(), d_out, ob.size()));
            if (getenv("MI_LTE_SCAN_TRACE"))
                fprintf(stderr, "  pdsch stage of %u units, %zu allocations: plan %.0f us, assign %.0f us, decode_run (launches) %.0f us, wait + verdicts %.0f us, %zu bytes of transport blocks %.0f us\n",
                        nu, al.size(), 1e6 * (ta - t0), 1e6 * (tb - ta), 1e6 * (tc - tb), 1e6 * (td - tc), ob.size(), 1e6 * (now_s() - td));
            b.bits.assign((size_t)nu * b.stride, 0);
            for (size_t k = 0; k < al.size(); k++) {
                b.status[unit_of[k]] = st[k];
                memcpy(&b.bits[(size_t)unit_of[k] * b.stride], &ob[k * b.stride], b.stride);
            }
        }
        t_stage[2] += now_s() - t0;
    }
    // unit u's decoded transport block as the message the RRC unpackers take
    static void to_msg(const Batch &b, uint32_t u, LIBLTE_BIT_MSG_STRUCT *msg)
    {
        const mi_lte_pdsch_alloc &a = b.dci[(size_t)u * MI_LTE_PDCCH_MAX_DCI].alloc;
        msg->N_bits = a.tbs;
        memcpy(msg->msg, &b.bits[(size_t)u * b.stride], a.tbs);
    }
};

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: scan_batch <capture.bin> <fs MHz> [max SI frames] [int8 | gr_complex]\n"); return 2; }
    // the two formats LTE_fdd_dl_file_scan reads (LTE_fdd_dl_fs_samp_buf.cc:657-694): int8 I,Q pairs (default) or gr_complex = float32 pairs
    const bool gr_complex = argc > 4 && strcmp(argv[4], "gr_complex") == 0;
    const size_t samp_bytes = gr_complex ? 8 : 2;
    const double fs_mhz = atof(argv[2]);
    const int    fsi = fs_mhz < 2 ? 0 : fs_mhz < 4 ? 1 : fs_mhz < 8 ? 2 : fs_mhz < 16 ? 3 : 4;
    static const char    *fs_text[5] = {"1.92", "3.84", "7.68", "15.36", "30.72"}; // liblte_phy_fs_text (liblte_phy.h:205)
    static const uint32_t fs_num[5]  = {1920000, 3840000, 7680000, 15360000, 30720000};
    const uint32_t si_frames = argc > 3 ? atoi(argv[3]) : 9;
    Scanner s;
    s.fft = 128u << fsi; s.fs_hz = fs_num[fsi];
    const uint32_t sc = 2048 / s.fft, n_subfr = 30720 / sc, n_frame = 10 * n_subfr;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 4;
    fseek(f, 0, SEEK_END);
    s.n = (uint32_t)(ftell(f) / samp_bytes);
    fseek(f, 0, SEEK_SET);
    std::vector<int8_t> raw((size_t)s.n * samp_bytes);
    if (fread(raw.data(), samp_bytes, s.n, f) != s.n) return 4;
    fclose(f);
    printf("capture: %u samples (%.1f frames) at %s Hz\n", s.n, (double)s.n / n_frame, fs_text[fsi]);

    const double t_start = now_s();
    if (mi_lte_ctx_create(0, &s.ctx) != MI_LTE_OK) { fprintf(stderr, "no usable gfx950 device (the library has no CPU path)\n"); return 3; }
    mi_lte_ctx *ctx = s.ctx;
    probe(ctx, "after ctx_create");
    // the capture in HBM: int8 pairs once, then planar float with two frames of zeros behind it (the per-call scanner's calloc'ed pad)
    const uint64_t pad = 2ull * n_frame, n_buf = (uint64_t)s.n + pad;
    int8_t *d_raw;
    CK(mi_lte_malloc(ctx, (size_t)s.n * samp_bytes, (void **)&d_raw));
    CK(mi_lte_malloc(ctx, sizeof(float) * n_buf, (void **)&s.d_i));
    CK(mi_lte_malloc(ctx, sizeof(float) * n_buf, (void **)&s.d_q));
    CK(mi_lte_memset(ctx, s.d_i, 0, sizeof(float) * n_buf));
    CK(mi_lte_memset(ctx, s.d_q, 0, sizeof(float) * n_buf));
    CK(mi_lte_memcpy_h2d(ctx, d_raw, raw.data(), (size_t)s.n * samp_bytes));
    if (gr_complex) CK(mi_lte_iq_f32_pairs_to_planar(ctx, reinterpret_cast<const float *>(d_raw), s.n, s.d_i, s.d_q));
    else            CK(mi_lte_iq_i8_to_planar(ctx, d_raw, s.n, s.d_i, s.d_q));
    mi_lte_free(ctx, d_raw);

    mi_lte_dl_cfg cfg6 = {s.fft, 6, 1, MI_LTE_IQ_F32_PLANAR}; // before the MIB only the centre six resource blocks are known to exist
    if ((uint64_t)s.n + pad < mi_lte_coarse_timing_samples(s.fft, 160)) { printf("no coarse timing\n"); return 1; }
    mi_lte_coarse_timing timing;
    CK(mi_lte_coarse_timing_run(ctx, &cfg6, s.d_i, s.d_q, 0, 160, &timing));
    const double t_coarse = now_s() - t_start;
    probe(ctx, "after coarse timing");
    printf("coarse timing: %u correlation peak(s)\n", timing.n_corr_peaks);

    static LIBLTE_BIT_MSG_STRUCT            msg;
    static LIBLTE_RRC_MIB_STRUCT            mib;
    static LIBLTE_RRC_BCCH_DLSCH_MSG_STRUCT si;
    int      cells = 0;
    uint32_t seen[8], n_subframes = 0;
    float    applied = 0;
    double   t_subframes = 0;
    for (uint32_t p = 0; p < timing.n_corr_peaks; p++) {
        CK(mi_lte_freq_shift_run(ctx, s.d_i, s.d_q, 0, s.n, timing.freq_offset[p] - applied, s.fs_hz));
        applied = timing.freq_offset[p];
        uint32_t N_id_2, N_id_1, pss_symb, frame_start, found = 0;
        float    pss_thresh, f_off;
        if (MI_LTE_OK != mi_lte_find_pss_run(ctx, &cfg6, s.d_i, s.d_q, 0, timing.symb_starts[p], &N_id_2, &pss_symb, &pss_thresh, &f_off)) continue;
        if (fabs(f_off) > 100) { CK(mi_lte_freq_shift_run(ctx, s.d_i, s.d_q, 0, s.n, f_off, s.fs_hz)); applied += f_off; }
        if (MI_LTE_OK != mi_lte_find_sss_run(ctx, &cfg6, s.d_i, s.d_q, 0, N_id_2, timing.symb_starts[p], pss_thresh, &N_id_1, &frame_start, &found) || !found) continue;
        const uint32_t N_id_cell = 3 * N_id_1 + N_id_2;
        bool           dup = false;
        for (int k = 0; k < cells; k++) dup |= seen[k] == N_id_cell; // another correlation peak of a cell already reported
        if (dup || cells == 8) continue;
        // ---- PBCH: subframe 0 at the frame start, estimated for four ports
        uint32_t N_ant = 0, sfn_off = 0, mib_bits = 0;
        {
            mi_lte_dl_cfg cfg4 = {s.fft, 6, 4, MI_LTE_IQ_F32_PLANAR};
            uint64_t      st = frame_start, *d_st;
            uint32_t      zero = 0, *d_sf, *d_cell;
            float        *d_sub;
            CK(mi_lte_malloc(ctx, 8, (void **)&d_st)); CK(mi_lte_malloc(ctx, 4, (void **)&d_sf)); CK(mi_lte_malloc(ctx, 4, (void **)&d_cell));
            CK(mi_lte_malloc(ctx, sizeof(float) * mi_lte_subframe_floats(4), (void **)&d_sub));
            CK(mi_lte_memcpy_h2d(ctx, d_st, &st, 8)); CK(mi_lte_memcpy_h2d(ctx, d_sf, &zero, 4)); CK(mi_lte_memcpy_h2d(ctx, d_cell, &N_id_cell, 4));
            CK(mi_lte_dl_frontend_batch(ctx, &cfg4, s.d_i, s.d_q, d_st, d_sf, d_cell, 1, d_sub));
            CK(mi_lte_pbch_decode_run(ctx, &cfg4, d_sub, d_cell, 1, &N_ant, &sfn_off, &mib_bits));
            mi_lte_free(ctx, d_st); mi_lte_free(ctx, d_sf); mi_lte_free(ctx, d_cell); mi_lte_free(ctx, d_sub);
        }
        probe(ctx, "after pbch");
        if (N_ant == 0) continue;
        msg.N_bits = 24;
        for (uint32_t k = 0; k < 24; k++) msg.msg[k] = (mib_bits >> (23 - k)) & 1u;
        if (LIBLTE_SUCCESS != liblte_rrc_unpack_bcch_bch_msg(&msg, &mib)) continue;
        static const uint32_t rb_of_bw[6] = {6, 15, 25, 50, 75, 100};
        const uint32_t N_rb_dl = rb_of_bw[mib.dl_bw];
        uint32_t       sfn       = (mib.sfn_div_4 << 2) + sfn_off;
        const float    phich_res = liblte_rrc_phich_resource_num[mib.phich_config.res];
        printf("cell %u: frame start %u, %u antenna port(s), MIB: N_rb_dl=%u phich_dur=%d phich_res=%d sfn=%u\n", N_id_cell,
               frame_start % n_frame, (unsigned)N_ant, N_rb_dl, (int)mib.phich_config.dur, (int)mib.phich_config.res, sfn);
        seen[cells++] = N_id_cell;
        // ---- SIB1: subframe 5 of every even frame, all of them in one batch; the first that decodes as SIB1 is the per-call scanner's
        uint32_t r = frame_start;
        if (sfn % 2) { r += n_frame; sfn++; }
        Batch b1;
        for (uint32_t rr = r; rr + 2 * n_frame < s.n; rr += 2 * n_frame) { b1.start.push_back((uint64_t)rr + 5ull * n_subfr); b1.sf.push_back(5); }
        s.run(b1, N_rb_dl, N_ant, N_id_cell, phich_res);
        bool got_sib1 = false;
        for (uint32_t k = 0; k < b1.start.size(); k++, r += 2 * n_frame, sfn += 2) {
            if (b1.rc[k] != 0 || b1.status[k] != 0) continue;
            Scanner::to_msg(b1, k, &msg);
            if (LIBLTE_SUCCESS == liblte_rrc_unpack_bcch_dlsch_msg(&msg, &si) && si.N_sibs == 1 && si.sibs[0].sib_type == LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1) {
                const mi_lte_pdsch_alloc &a = b1.dci[(size_t)k * MI_LTE_PDCCH_MAX_DCI].alloc;
                printf("sfn %u subframe 5: CFI=%u tbs=%u N_prb=%u rv=%u -> ", sfn, b1.n_symbs[k], a.tbs, a.N_prb, a.rv_idx);
                report_sib1((LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_1_STRUCT *)&si.sibs[0].sib);
                got_sib1 = true;
                r += 2 * n_frame; sfn += 2; // (the per-call loop's increment runs once more before its condition ends it)
                break;
            }
        }
        if (!got_sib1) { printf("cell %u: SIB1 not found\n", N_id_cell); continue; }
        // ---- every subframe of the following frames: any other system information
        Batch bs;
        std::vector<uint32_t> sfn_of;
        {
            uint32_t rr = r, sf_n = sfn;
            for (uint32_t fr = 0; fr < si_frames && rr + n_frame + n_subfr < s.n; fr++, rr += n_frame, sf_n++)
                for (uint32_t nn = 0; nn < 10; nn++) { bs.start.push_back((uint64_t)rr + (uint64_t)nn * n_subfr); bs.sf.push_back(nn); sfn_of.push_back(sf_n); }
        }
        const double t_loop = now_s();
        s.run(bs, N_rb_dl, N_ant, N_id_cell, phich_res);
        t_subframes += now_s() - t_loop;
        n_subframes += (uint32_t)bs.start.size();
        uint32_t n_pdsch = 0, n_fail = 0;
        bool     got_sib2 = false;
        for (uint32_t k = 0; k < bs.start.size(); k++) {
            if (bs.rc[k] != 0) continue;
            if (bs.status[k] != 0) { n_fail++; continue; }
            n_pdsch++;
            const mi_lte_pdsch_alloc &a = bs.dci[(size_t)k * MI_LTE_PDCCH_MAX_DCI].alloc;
            Scanner::to_msg(bs, k, &msg);
            if (LIBLTE_MAC_SI_RNTI != a.rnti || LIBLTE_SUCCESS != liblte_rrc_unpack_bcch_dlsch_msg(&msg, &si)) continue;
            for (uint32_t j = 0; j < si.N_sibs; j++)
                if (si.sibs[j].sib_type == LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2 && !got_sib2) {
                    printf("sfn %u subframe %u: CFI=%u tbs=%u N_prb=%u -> ", sfn_of[k], bs.sf[k], bs.n_symbs[k], a.tbs, a.N_prb);
                    report_sib2((LIBLTE_RRC_SYS_INFO_BLOCK_TYPE_2_STRUCT *)&si.sibs[j].sib);
                    got_sib2 = true;
                }
        }
        printf("cell %u: %u PDSCH transport blocks decoded after SIB1, %u with a PDCCH but a failed CRC\n", N_id_cell, n_pdsch, n_fail);
    }
    printf("%d cell(s) found\n", cells);
    fprintf(stderr, "timing: total %.4f s; up to the coarse timing (HIP start-up, upload) %.4f s; all-subframes batch (front end + pdcch + pdsch) "
                    "%.4f s for %u subframes = %.1f us per subframe; stages: front end %.4f s, pdcch %.4f s, pdsch %.4f s\n",
            now_s() - t_start, t_coarse, t_subframes, n_subframes, n_subframes ? 1e6 * t_subframes / n_subframes : 0.0, s.t_stage[0], s.t_stage[1], s.t_stage[2]);
    s.release();
    mi_lte_free(ctx, s.d_i); mi_lte_free(ctx, s.d_q);
    mi_lte_ctx_destroy(ctx);
    return cells ? 0 : 1;
}
