// Per-call, host-pointer forms of the three reference entry points on the hot path (see
// include/mi_lte.h).  They exist for drop-in use through shim/liblte_phy_shim.cc: stage through HBM,
// run the same batch kernels with a batch of one, copy back.  No arithmetic happens on the host.
#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? 0 : -1; }
};
// the LTE transform lengths and a grid that fits inside them: checked before any size arithmetic is done with fft_size
bool valid_fft(uint32_t fft_size, uint32_t N_rb)
{
    return (fft_size == 128 || fft_size == 256 || fft_size == 512 || fft_size == 1024 || fft_size == 2048) && N_rb >= 6 && N_rb * 12 < fft_size;
}
} // namespace

extern "C" {

// liblte_phy_get_dl_subframe_and_ce (liblte_phy.cc:5905-6200), argument checks as at :5937-5943
int mi_lte_get_dl_subframe_and_ce_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q,
                                       uint32_t frame_start_idx, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant,
                                       float *h_symb_re, float *h_symb_im, float *h_ce_re, float *h_ce_im)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !(N_ant == 1 || N_ant == 2 || N_ant == 4) || !h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || !valid_fft(fft_size, N_rb_dl))
        return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t sc = 2048 / fft_size;
    const size_t   per_sf = 30720 / sc, need = per_sf + 2 * fft_size + 160 / sc + 144 / sc - 1; // last sample symbol 15 reads, +1
    const size_t   start = (size_t)frame_start_idx + (size_t)subfr_num * per_sf;
    DevBuf d_i, d_q, d_par, d_sub;
    const size_t nf = mi_lte_subframe_floats(N_ant);
    if (d_i.alloc(need * 4) || d_q.alloc(need * 4) || d_par.alloc(32) || d_sub.alloc(nf * 4)) return MI_LTE_ERR_NOMEM;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_i.p, h_i + start, need * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_q.p, h_q + start, need * 4, hipMemcpyHostToDevice, ctx->stream));
    struct { uint64_t start; uint32_t sf, cell; } par = {0, subfr_num, N_id_cell};
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_par.p, &par, sizeof(par), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemsetAsync(d_sub.p, 0, nf * 4, ctx->stream));
    mi_lte_dl_cfg cfg = {fft_size, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR | MI_LTE_IQ_ALL_ROWS}; // every row the reference's struct holds
    int rc = mi_lte_dl_frontend_batch(ctx, &cfg, d_i.p, d_q.p, (const uint64_t *)d_par.p, (const uint32_t *)((char *)d_par.p + 8),
                                      (const uint32_t *)((char *)d_par.p + 12), 1, (float *)d_sub.p);
    if (rc != MI_LTE_OK) return rc;
    const size_t row = 16 * 1200 * sizeof(float);
    float       *s   = (float *)d_sub.p;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_symb_re, s, row, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_symb_im, s + 16 * 1200, row, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_ce_re, s + 2 * 16 * 1200, row * N_ant, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_ce_im, s + (2 + N_ant) * 16 * 1200, row * N_ant, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// liblte_phy_pdsch_channel_decode (liblte_phy.cc:3690-3853)
int mi_lte_pdsch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im,
                                     const float *h_ce_re, const float *h_ce_im, uint32_t subfr_num, const mi_lte_pdsch_alloc *alloc,
                                     uint32_t N_pdcch_symbs, uint32_t N_id_cell, uint32_t N_ant, uint8_t *h_out_bits,
                                     uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || !alloc || N_id_cell > 503 || !h_out_bits || !N_out_bits) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t fft = N_rb_dl <= 6 ? 128 : N_rb_dl <= 15 ? 256 : N_rb_dl <= 25 ? 512 : N_rb_dl <= 50 ? 1024 : 2048;
    mi_lte_dl_cfg       cfg = {fft, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
    mi_lte_pdsch_alloc  a   = *alloc;
    a.unit                  = 0;
    mi_lte_pdsch_plan *plan = nullptr;
    int                rc   = mi_lte_pdsch_plan_create(ctx, &cfg, N_pdcch_symbs, &a, 1, &plan);
    if (rc == MI_LTE_ERR_UNSUPPORTED) return 2; // outside the envelope the reference itself decodes: report a decode failure
    if (rc != MI_LTE_OK) return rc;
    const size_t nf = mi_lte_subframe_floats(N_ant), row = 16 * 1200;
    const uint32_t stride = mi_lte_pdsch_plan_out_stride(plan);
    DevBuf d_sub, d_par, d_out, d_st;
    if (d_sub.alloc(nf * 4) || d_par.alloc(16) || d_out.alloc(stride) || d_st.alloc(4)) { mi_lte_pdsch_plan_destroy(ctx, plan); return MI_LTE_ERR_NOMEM; }
    float *s = (float *)d_sub.p;
    uint32_t par[2] = {subfr_num, N_id_cell};
    hipError_t e = hipMemcpyAsync(s, h_symb_re, row * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + row, h_symb_im, row * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + 2 * row, h_ce_re, row * 4 * N_ant, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + (2 + N_ant) * row, h_ce_im, row * 4 * N_ant, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_par.p, par, 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { mi_lte_pdsch_plan_destroy(ctx, plan); ctx->err = hipGetErrorString(e); return MI_LTE_ERR_HIP; }
    rc = mi_lte_pdsch_decode_run(ctx, plan, s, (const uint32_t *)d_par.p, (const uint32_t *)d_par.p + 1, (uint8_t *)d_out.p, (int32_t *)d_st.p);
    int32_t st = 2;
    if (rc == MI_LTE_OK) {
        e = hipMemcpyAsync(&st, d_st.p, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && st == 0) { // the reference copies the bits out only when the CRC matched (:12861-12869)
            e = hipMemcpy(h_out_bits, d_out.p, a.tbs, hipMemcpyDeviceToHost);
            *N_out_bits = a.tbs;
        }
        if (e != hipSuccess) { rc = MI_LTE_ERR_HIP; ctx->err = hipGetErrorString(e); }
    }
    mi_lte_pdsch_plan_destroy(ctx, plan);
    return rc != MI_LTE_OK ? rc : (int)st;
}

// liblte_phy_pdcch_channel_decode (liblte_phy.cc:4519-5135): PCFICH + common-search-space DCIs of one subframe
int mi_lte_pdcch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im, const float *h_ce_re,
                                     const float *h_ce_im, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float phich_res,
                                     uint32_t phich_dur_extended, uint32_t flags, uint32_t *cfi, uint32_t *N_symbs, uint32_t *N_dci,
                                     mi_lte_pdcch_dci *dci /*[MI_LTE_PDCCH_MAX_DCI]*/)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || N_id_cell > 503 || !cfi || !N_symbs || !N_dci || !dci) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t fft = N_rb_dl <= 6 ? 128 : N_rb_dl <= 15 ? 256 : N_rb_dl <= 25 ? 512 : N_rb_dl <= 50 ? 1024 : 2048;
    mi_lte_dl_cfg      cfg  = {fft, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
    mi_lte_pdcch_plan *plan = nullptr;
    int rc = mi_lte_pdcch_plan_create(ctx, &cfg, phich_res, phich_dur_extended, flags, &N_id_cell, 1, &plan);
    if (rc != MI_LTE_OK) return rc;
    const size_t nf = mi_lte_subframe_floats(N_ant), row = 16 * 1200;
    DevBuf d_sub, d_par;
    if (d_sub.alloc(nf * 4) || d_par.alloc(16)) { mi_lte_pdcch_plan_destroy(ctx, plan); return MI_LTE_ERR_NOMEM; }
    float   *s      = (float *)d_sub.p;
    uint32_t par[2] = {subfr_num, N_id_cell}, h_rc = 1;
    // only the control region is read: symbols 0..3
    const size_t ctl = 4 * 1200 * sizeof(float);
    hipError_t   e   = hipMemcpyAsync(s, h_symb_re, ctl, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + row, h_symb_im, ctl, hipMemcpyHostToDevice, ctx->stream);
    for (uint32_t p = 0; p < N_ant && e == hipSuccess; p++) {
        e = hipMemcpyAsync(s + (2 + p) * row, h_ce_re + p * row, ctl, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s + (2 + N_ant + p) * row, h_ce_im + p * row, ctl, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_par.p, par, 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { mi_lte_pdcch_plan_destroy(ctx, plan); ctx->err = hipGetErrorString(e); return MI_LTE_ERR_HIP; }
    rc = mi_lte_pdcch_decode_run(ctx, plan, s, (const uint32_t *)d_par.p, (const uint32_t *)d_par.p + 1, 1, &h_rc, cfi, N_symbs, N_dci, dci);
    mi_lte_pdcch_plan_destroy(ctx, plan);
    return rc != MI_LTE_OK ? rc : (int)h_rc;
}

// liblte_phy_bch_channel_decode (liblte_phy.cc:3968-4105)
int mi_lte_bch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im, const float *h_ce_re,
                                   const float *h_ce_im, uint32_t N_id_cell, uint8_t *N_ant, uint8_t *h_out_bits, uint32_t *N_out_bits, uint8_t *offset)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || N_id_cell > 503 || !N_ant || !h_out_bits || !N_out_bits || !offset) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t fft = N_rb_dl <= 6 ? 128 : N_rb_dl <= 15 ? 256 : N_rb_dl <= 25 ? 512 : N_rb_dl <= 50 ? 1024 : 2048;
    mi_lte_dl_cfg cfg = {fft, N_rb_dl, 4, MI_LTE_IQ_F32_PLANAR};
    const size_t  nf = mi_lte_subframe_floats(4), row = 16 * 1200;
    DevBuf d_sub, d_par;
    if (d_sub.alloc(nf * 4) || d_par.alloc(16)) return MI_LTE_ERR_NOMEM;
    float *s = (float *)d_sub.p;
    // only symbols 7..10 are read
    const size_t o = 7 * 1200, len = 4 * 1200 * sizeof(float);
    hipError_t   e = hipMemcpyAsync(s + o, h_symb_re + o, len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + row + o, h_symb_im + o, len, hipMemcpyHostToDevice, ctx->stream);
    for (uint32_t p = 0; p < 4 && e == hipSuccess; p++) {
        e = hipMemcpyAsync(s + (2 + p) * row + o, h_ce_re + p * row + o, len, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s + (6 + p) * row + o, h_ce_im + p * row + o, len, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_par.p, &N_id_cell, 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return MI_LTE_ERR_HIP; }
    uint32_t n_ant = 0, off = 0, mib = 0;
    int rc = mi_lte_pbch_decode_run(ctx, &cfg, s, (const uint32_t *)d_par.p, 1, &n_ant, &off, &mib);
    if (rc != MI_LTE_OK) return rc;
    *N_ant = (uint8_t)n_ant; // the reference zeroes it before trying (:4029)
    if (n_ant == 0) return 2;
    for (uint32_t i = 0; i < 24; i++) h_out_bits[i] = (uint8_t)((mib >> (23 - i)) & 1u);
    *N_out_bits = 24;
    *offset     = (uint8_t)off;
    return 0;
}

// liblte_phy_pucch_format_1_1a_1b_channel_decode (liblte_phy.cc:2961-3146); the sequences are the caller's (mi_lte.h)
int mi_lte_pucch_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_symb_re, const float *h_symb_im, uint32_t format, uint32_t N_ant,
                             uint32_t N_1_p_pucch, const float *h_tables, uint8_t *h_out_bits, uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || format > 2 || !h_tables || !h_out_bits || !N_out_bits) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf d_sub;
    const size_t row = 16 * 1200;
    if (d_sub.alloc(mi_lte_ul_subframe_floats() * 4)) return MI_LTE_ERR_NOMEM;
    float *s = (float *)d_sub.p;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(s, h_symb_re, 14 * 1200 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(s + row, h_symb_im, 14 * 1200 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    mi_lte_pucch_res r = {0, format, N_1_p_pucch};
    uint8_t  bits[2] = {0, 0};
    uint32_t nb = 0, rc2 = 1;
    int rc = mi_lte_pucch_decode_run(ctx, N_rb_ul, N_ant, s, &r, h_tables, 1, bits, &nb, &rc2);
    if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_INVALID_ARG ? 1 : rc;
    h_out_bits[0] = bits[0];
    if (nb == 2) h_out_bits[1] = bits[1];
    *N_out_bits = nb;
    return (int)rc2;
}

// ---- initial synchronisation (liblte_phy.cc:5697-5852, :5306-5510, :5578-5687): stage n samples of each array, run the device search
namespace {
int stage_iq(mi_lte_ctx *ctx, const float *h_i, const float *h_q, size_t n, DevBuf &d_i, DevBuf &d_q)
{
    if (d_i.alloc(n * 4) || d_q.alloc(n * 4)) return MI_LTE_ERR_NOMEM;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_i.p, h_i, n * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_q.p, h_q, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return MI_LTE_OK;
}
} // namespace

int mi_lte_dl_find_coarse_timing_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t N_slots,
                                      mi_lte_coarse_timing *out)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !out || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mi_lte_dl_cfg cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    DevBuf d_i, d_q;
    int    rc = stage_iq(ctx, h_i, h_q, mi_lte_coarse_timing_samples(fft_size, N_slots), d_i, d_q);
    if (rc != MI_LTE_OK) return rc;
    return mi_lte_coarse_timing_run(ctx, &cfg, d_i.p, d_q.p, 0, N_slots, out);
}

int mi_lte_find_pss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t *symb_starts,
                         uint32_t *N_id_2, uint32_t *pss_symb, float *pss_thresh, float *freq_offset)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !symb_starts || !N_id_2 || !pss_symb || !pss_thresh || !freq_offset || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mi_lte_dl_cfg  cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    const uint32_t sc = 2048 / fft_size;
    uint32_t       last = 0;
    for (int j = 0; j < 7; j++) last = symb_starts[j] > last ? symb_starts[j] : last;
    const size_t n = (size_t)last + 11 * (15360 / sc) + 160 / sc + fft_size + 38; // last window of the 84, or of the +39 fine-timing trial
    DevBuf d_i, d_q;
    int    rc = stage_iq(ctx, h_i, h_q, n, d_i, d_q);
    if (rc != MI_LTE_OK) return rc;
    return mi_lte_find_pss_run(ctx, &cfg, d_i.p, d_q.p, 0, symb_starts, N_id_2, pss_symb, pss_thresh, freq_offset);
}

int mi_lte_find_sss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t N_id_2, uint32_t *symb_starts,
                         float pss_thresh, uint32_t *N_id_1, uint32_t *frame_start_idx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !symb_starts || !N_id_1 || !frame_start_idx || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mi_lte_dl_cfg  cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    const uint32_t sc = 2048 / fft_size;
    DevBuf d_i, d_q;
    int    rc = stage_iq(ctx, h_i, h_q, (size_t)symb_starts[5] + 160 / sc - 1 + fft_size, d_i, d_q);
    if (rc != MI_LTE_OK) return rc;
    uint32_t found = 0;
    rc = mi_lte_find_sss_run(ctx, &cfg, d_i.p, d_q.p, 0, N_id_2, symb_starts, pss_thresh, N_id_1, frame_start_idx, &found);
    return rc != MI_LTE_OK ? rc : (found ? 0 : 1);
}

// liblte_phy_get_ul_subframe (liblte_phy.cc:6209-6236)
int mi_lte_get_ul_subframe_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const float *h_i, const float *h_q, float *h_symb_re,
                                float *h_symb_im)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !h_symb_re || !h_symb_im || !valid_fft(fft_size, N_rb_ul)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t sc = 2048 / fft_size;
    const size_t   need = 30720 / sc; // the last symbol's window ends one sample before the subframe does
    DevBuf d_i, d_q, d_par, d_sub;
    const size_t nf = mi_lte_ul_subframe_floats();
    if (d_i.alloc(need * 4) || d_q.alloc(need * 4) || d_par.alloc(8) || d_sub.alloc(nf * 4)) return MI_LTE_ERR_NOMEM;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_i.p, h_i, need * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_q.p, h_q, need * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemsetAsync(d_par.p, 0, 8, ctx->stream));
    mi_lte_dl_cfg cfg = {fft_size, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    int rc = mi_lte_ul_frontend_batch(ctx, &cfg, d_i.p, d_q.p, (const uint64_t *)d_par.p, 1, (float *)d_sub.p);
    if (rc != MI_LTE_OK) return rc;
    const size_t row = 14 * 1200 * sizeof(float); // rows 14, 15 of the caller's struct are left alone, as the reference leaves them
    float       *s   = (float *)d_sub.p;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_symb_re, s, row, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_symb_im, s + 16 * 1200, row, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// liblte_phy_pusch_channel_decode (liblte_phy.cc:2801-2935).  The reference signals are the caller's: the arrays
// liblte_phy_ul_init stored in LIBLTE_PHY_STRUCT (pusch_dmrs_0/1_re/im[subframe][N_prb]).  A failed CRC is reported
// as 1 = LIBLTE_ERROR_INVALID_INPUTS, which is what the reference returns in that case (:2809, :2929).
int mi_lte_pusch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_symb_re, const float *h_symb_im, uint32_t subfr_num,
                                     const mi_lte_pdsch_alloc *alloc, uint32_t N_id_cell, uint32_t N_ant, const float *h_dmrs_0_re,
                                     const float *h_dmrs_0_im, const float *h_dmrs_1_re, const float *h_dmrs_1_im, uint8_t *h_out_bits,
                                     uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !alloc || !h_out_bits || !N_out_bits || !h_dmrs_0_re || !h_dmrs_0_im || !h_dmrs_1_re || !h_dmrs_1_im) return 1;
    (void)N_ant;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t fft = N_rb_ul <= 6 ? 128 : N_rb_ul <= 15 ? 256 : N_rb_ul <= 25 ? 512 : N_rb_ul <= 50 ? 1024 : 2048;
    mi_lte_dl_cfg      cfg = {fft, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    mi_lte_pdsch_alloc a   = *alloc;
    a.unit                 = 0;
    const size_t       M   = 12 * (size_t)a.N_prb;
    std::vector<float> dm(4 * M);
    memcpy(&dm[0], h_dmrs_0_re, M * 4); memcpy(&dm[M], h_dmrs_0_im, M * 4);
    memcpy(&dm[2 * M], h_dmrs_1_re, M * 4); memcpy(&dm[3 * M], h_dmrs_1_im, M * 4);
    mi_lte_pusch_plan *plan = nullptr;
    int rc = mi_pusch_plan_create_impl(ctx, &cfg, nullptr, &subfr_num, &N_id_cell, 1, &a, 1, dm.data(), &plan);
    if (rc == MI_LTE_ERR_UNSUPPORTED) return 1;
    if (rc != MI_LTE_OK) return rc;
    const size_t   nf = mi_lte_ul_subframe_floats(), row = 16 * 1200;
    const uint32_t stride = mi_lte_pusch_plan_out_stride(plan);
    DevBuf d_sub, d_out, d_st;
    if (d_sub.alloc(nf * 4) || d_out.alloc(stride) || d_st.alloc(4)) { mi_lte_pusch_plan_destroy(ctx, plan); return MI_LTE_ERR_NOMEM; }
    float     *s = (float *)d_sub.p;
    hipError_t e = hipMemcpyAsync(s, h_symb_re, row * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s + row, h_symb_im, row * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { mi_lte_pusch_plan_destroy(ctx, plan); ctx->err = hipGetErrorString(e); return MI_LTE_ERR_HIP; }
    rc = mi_lte_pusch_decode_run(ctx, plan, s, (uint8_t *)d_out.p, (int32_t *)d_st.p);
    int32_t st = 2;
    if (rc == MI_LTE_OK) {
        e = hipMemcpyAsync(&st, d_st.p, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && st == 0) {
            e = hipMemcpy(h_out_bits, d_out.p, a.tbs, hipMemcpyDeviceToHost);
            *N_out_bits = a.tbs;
        }
        if (e != hipSuccess) { rc = MI_LTE_ERR_HIP; ctx->err = hipGetErrorString(e); }
    }
    mi_lte_pusch_plan_destroy(ctx, plan);
    return rc != MI_LTE_OK ? rc : (st == 0 ? 0 : 1);
}

// liblte_phy_detect_prach (liblte_phy.cc:3299-3479): h_re / h_im point at the occasion's first cyclic-prefix sample; the root
// spectra are the caller's (LIBLTE_PHY_STRUCT::prach_x_u_fft_re/im, filled by liblte_phy_ul_init).  N_det_pre is always
// written; det_pre / det_ta only when a preamble was found, like the reference.
int mi_lte_detect_prach_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const mi_lte_prach_cfg *prach, const float *h_x_u_fft_re,
                             const float *h_x_u_fft_im, uint32_t n_roots, const float *h_re, const float *h_im, uint32_t *N_det_pre,
                             uint32_t *det_pre, uint32_t *det_ta)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!prach || !h_x_u_fft_re || !h_x_u_fft_im || !h_re || !h_im || !N_det_pre || !det_pre || !det_ta || !valid_fft(fft_size, N_rb_ul)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mi_lte_dl_cfg      cfg = {fft_size, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    mi_lte_prach_plan *plan = nullptr;
    int rc = mi_lte_prach_plan_create_roots(ctx, &cfg, prach, h_x_u_fft_re, h_x_u_fft_im, n_roots, &plan);
    if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_UNSUPPORTED ? 1 : rc;
    const size_t need = mi_lte_prach_occasion_samples(plan);
    DevBuf d_i, d_q, d_s;
    if (d_i.alloc(need * 4) || d_q.alloc(need * 4) || d_s.alloc(8)) { mi_lte_prach_plan_destroy(ctx, plan); return MI_LTE_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(d_i.p, h_re, need * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_q.p, h_im, need * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_s.p, 0, 8, ctx->stream);
    if (e != hipSuccess) { mi_lte_prach_plan_destroy(ctx, plan); ctx->err = hipGetErrorString(e); return MI_LTE_ERR_HIP; }
    uint32_t n = 0, p = 0, ta = 0;
    rc = mi_lte_prach_detect_run(ctx, plan, d_i.p, d_q.p, (const uint64_t *)d_s.p, 1, &n, &p, &ta);
    mi_lte_prach_plan_destroy(ctx, plan);
    if (rc != MI_LTE_OK) return rc;
    *N_det_pre = n;
    if (n) { *det_pre = p; *det_ta = ta; }
    return 0;
}

// liblte_phy_rate_unmatch_turbo (liblte_phy.cc:11246-11490)
int mi_lte_rate_unmatch_turbo_host(mi_lte_ctx *ctx, const float *h_e, uint32_t N_e, uint32_t N_dummy_bits, uint32_t C, uint32_t tx_mode,
                                   uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type, uint32_t rv_idx, float *h_d, uint32_t *N_d)
{
    if (!ctx || !h_e || !h_d || !N_d) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf d_e, d_d;
    if (d_e.alloc((size_t)N_e * 4) || d_d.alloc((size_t)3 * N_dummy_bits * 4)) return MI_LTE_ERR_NOMEM;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(d_e.p, h_e, (size_t)N_e * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = mi_lte_rate_unmatch_turbo_batch(ctx, (const float *)d_e.p, N_e, N_dummy_bits, C, tx_mode, N_soft, M_dl_harq, chan_type, rv_idx, 1,
                                             (float *)d_d.p);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipMemcpyAsync(h_d, d_d.p, (size_t)3 * N_dummy_bits * 4, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *N_d = 3 * N_dummy_bits;
    return 0;
}

} // extern "C"
