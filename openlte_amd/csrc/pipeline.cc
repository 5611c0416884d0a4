// Whole-chain batches from HOST buffers: captures live in host memory (LTE_fdd_dl_fs_samp_buf.cc:657-694), so a caller that
// wants more than one subframe at a time hands over int8 I,Q units and gets transport blocks back.  The batch is cut into chunks that flow
//     H2D (int8 IQ)  ->  front end  ->  PDSCH chain  ->  D2H (packed bits + verdicts)
// on LANES: each lane is a context of its own (stream, scratch, plan, device buffers), chunks go to the lanes round robin, and because
// a lane's stream orders its own three phases while the lanes are independent of each other, one lane's copies run under another
// lane's kernels.  Nothing is allocated per run.  With the 70 KB of samples per subframe the PCIe link is the limit (a x16 Gen5 link
// moves about 50 GB/s: ~0.7 M subframes/s per GPU), which is the point: the device-resident rate is three times that.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {
struct Lane {
    mi_lte_ctx        *ctx  = nullptr;
    mi_lte_pdsch_plan *plan = nullptr;
    int8_t   *d_iq = nullptr;
    uint64_t *d_start = nullptr;
    uint32_t *d_sf = nullptr, *d_cell = nullptr;
    float    *d_sub = nullptr;
    uint8_t  *d_out = nullptr;
    int32_t  *d_st = nullptr;
    hipEvent_t done = nullptr;
};
} // namespace

struct mi_lte_dl_pipeline {
    mi_lte_dl_cfg cfg;
    uint32_t      cfi = 0, n_alloc = 0, chunk = 0, out_stride = 0;
    size_t        unit_samples = 0;
    int           device = 0;
    std::vector<Lane> lanes;
    std::string   err;
};

extern "C" {

void *mi_lte_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void mi_lte_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

void mi_lte_dl_pipeline_destroy(mi_lte_dl_pipeline *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (Lane &l : p->lanes) {
        if (l.ctx) (void)mi_lte_sync(l.ctx);
        if (l.plan) mi_lte_pdsch_plan_destroy(l.ctx, l.plan);
        (void)hipFree(l.d_iq); (void)hipFree(l.d_start); (void)hipFree(l.d_sf); (void)hipFree(l.d_cell);
        (void)hipFree(l.d_sub); (void)hipFree(l.d_out); (void)hipFree(l.d_st);
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.ctx) mi_lte_ctx_destroy(l.ctx);
    }
    delete p;
}

int mi_lte_dl_pipeline_create(int device, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_unit_allocs, uint32_t n_alloc_per_unit,
                              uint32_t chunk_units, uint32_t n_lanes, mi_lte_dl_pipeline **out)
{
    if (!cfg || !h_unit_allocs || !out || n_alloc_per_unit == 0 || chunk_units == 0 || n_lanes == 0 || n_lanes > 8) return MI_LTE_ERR_INVALID_ARG;
    if ((cfg->sample_format & 0xFFu) != MI_LTE_IQ_I8) return MI_LTE_ERR_INVALID_ARG; // host batches are int8 captures
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048)) return MI_LTE_ERR_INVALID_ARG;
    auto *p = new mi_lte_dl_pipeline();
    auto  guard = on_fail([&] { mi_lte_dl_pipeline_destroy(p); });
    p->cfg = *cfg; p->cfi = N_pdcch_symbs; p->n_alloc = n_alloc_per_unit; p->chunk = chunk_units; p->device = device;
    const uint32_t sc = 2048 / N;
    p->unit_samples = ((30720 + 4400) / sc + 15) / 16 * 16; // one subframe + the two look-ahead symbols, a multiple of 16 samples (= mi_lte_synth_unit_len)
    std::vector<mi_lte_pdsch_alloc> allocs((size_t)chunk_units * n_alloc_per_unit);
    for (uint32_t u = 0; u < chunk_units; u++)
        for (uint32_t a = 0; a < n_alloc_per_unit; a++) {
            allocs[(size_t)u * n_alloc_per_unit + a]      = h_unit_allocs[a];
            allocs[(size_t)u * n_alloc_per_unit + a].unit = u;
        }
    std::vector<uint64_t> starts(chunk_units);
    for (uint32_t u = 0; u < chunk_units; u++) starts[u] = (uint64_t)u * p->unit_samples;
    p->lanes.resize(n_lanes);
    for (Lane &l : p->lanes) {
        int rc = mi_lte_ctx_create(device, &l.ctx);
        if (rc != MI_LTE_OK) return rc;
        rc = mi_lte_pdsch_plan_create(l.ctx, cfg, N_pdcch_symbs, allocs.data(), (uint32_t)allocs.size(), &l.plan);
        if (rc != MI_LTE_OK) { p->err = mi_lte_last_error(l.ctx); return rc; }
        mi_lte_pdsch_plan_set_output(l.plan, 1);
        p->out_stride = mi_lte_pdsch_plan_out_stride(l.plan);
        const size_t n_al = allocs.size();
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_iq, (size_t)chunk_units * p->unit_samples * 2 + 64));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_start, sizeof(uint64_t) * chunk_units));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_sf, sizeof(uint32_t) * chunk_units));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_cell, sizeof(uint32_t) * chunk_units));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_sub, sizeof(float) * mi_lte_subframe_floats(cfg->N_ant) * chunk_units));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_out, n_al * p->out_stride));
        MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_st, n_al * sizeof(int32_t)));
        MI_HIP_CHECK(l.ctx, hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        MI_HIP_CHECK(l.ctx, hipMemcpy(l.d_start, starts.data(), sizeof(uint64_t) * chunk_units, hipMemcpyHostToDevice));
        MI_HIP_CHECK(l.ctx, hipMemset(l.d_sf, 0, sizeof(uint32_t) * chunk_units));   // a ragged last chunk decodes the units past its end too
        MI_HIP_CHECK(l.ctx, hipMemset(l.d_cell, 0, sizeof(uint32_t) * chunk_units)); // (results dropped): they must at least be valid numbers
        MI_HIP_CHECK(l.ctx, hipMemset(l.d_sub, 0, sizeof(float) * mi_lte_subframe_floats(cfg->N_ant) * chunk_units));
        MI_HIP_CHECK(l.ctx, hipMemset(l.d_iq, 0, (size_t)chunk_units * p->unit_samples * 2 + 64));
    }
    guard.armed = false;
    *out = p;
    return MI_LTE_OK;
}

uint32_t    mi_lte_dl_pipeline_out_stride(const mi_lte_dl_pipeline *p) { return p ? p->out_stride : 0; }
size_t      mi_lte_dl_pipeline_unit_samples(const mi_lte_dl_pipeline *p) { return p ? p->unit_samples : 0; }
const char *mi_lte_dl_pipeline_last_error(const mi_lte_dl_pipeline *p)
{
    if (!p) return "null pipeline";
    if (!p->err.empty()) return p->err.c_str();
    for (const Lane &l : p->lanes)
        if (l.ctx && *mi_lte_last_error(l.ctx)) return mi_lte_last_error(l.ctx);
    return "";
}

// h_iq: n_units units of unit_samples int8 I,Q pairs back to back; h_out_packed: [n_units * n_alloc][out_stride]; h_status likewise.
// The three host arrays should be pinned (mi_lte_host_alloc): with pageable memory the copies fall back to the driver's staged path and
// stop overlapping.  Returns when every result is in host memory.
int mi_lte_dl_pipeline_run(mi_lte_dl_pipeline *p, const int8_t *h_iq, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell, uint32_t n_units,
                           uint8_t *h_out_packed, int32_t *h_status)
{
    if (!p || !h_iq || !h_subfr_num || !h_n_id_cell || !h_out_packed || !h_status || n_units == 0) return MI_LTE_ERR_INVALID_ARG;
    if (hipSetDevice(p->device) != hipSuccess) return MI_LTE_ERR_HIP;
    const size_t unit_bytes = p->unit_samples * 2, al_per_chunk = (size_t)p->chunk * p->n_alloc;
    uint32_t c = 0;
    for (uint32_t u0 = 0; u0 < n_units; u0 += p->chunk, c++) {
        Lane          &l = p->lanes[c % p->lanes.size()];
        const uint32_t n = std::min(p->chunk, n_units - u0);
        hipStream_t    st = (hipStream_t)mi_lte_stream(l.ctx);
        MI_HIP_CHECK(l.ctx, hipMemcpyAsync(l.d_iq, h_iq + (size_t)u0 * unit_bytes, (size_t)n * unit_bytes, hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(l.ctx, hipMemcpyAsync(l.d_sf, h_subfr_num + u0, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(l.ctx, hipMemcpyAsync(l.d_cell, h_n_id_cell + u0, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
        int rc = mi_lte_dl_frontend_batch(l.ctx, &p->cfg, l.d_iq, nullptr, l.d_start, l.d_sf, l.d_cell, n, l.d_sub);
        // a ragged last chunk: the plan covers a whole chunk, the units past the end decode whatever an earlier chunk left in the lane's
        // buffers and their results are not copied back
        if (rc == MI_LTE_OK) rc = mi_lte_pdsch_decode_run(l.ctx, l.plan, l.d_sub, l.d_sf, l.d_cell, l.d_out, l.d_st);
        if (rc != MI_LTE_OK) { p->err = mi_lte_last_error(l.ctx); return rc; }
        const size_t n_al = (size_t)n * p->n_alloc;
        MI_HIP_CHECK(l.ctx, hipMemcpyAsync(h_out_packed + (size_t)c * al_per_chunk * p->out_stride, l.d_out, n_al * p->out_stride, hipMemcpyDeviceToHost, st));
        MI_HIP_CHECK(l.ctx, hipMemcpyAsync(h_status + (size_t)c * al_per_chunk, l.d_st, n_al * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    for (Lane &l : p->lanes) {
        int rc = mi_lte_sync(l.ctx);
        if (rc != MI_LTE_OK) return rc;
    }
    return MI_LTE_OK;
}

} // extern "C"
