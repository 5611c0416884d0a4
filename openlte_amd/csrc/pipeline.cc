// Whole-chain batches from HOST buffers: captures live in host memory (LTE_fdd_dl_fs_samp_buf.cc:657-694), so a caller that
// wants more than one subframe at a time hands over int8 I,Q and gets transport blocks back.  The batch is cut into chunks that flow
//     H2D (int8 IQ)  ->  front end  ->  PDSCH chain  ->  D2H (packed bits + verdicts)
// on LANES: each lane is a context of its own (stream, scratch, plans, device buffers), and because a lane's stream orders its own three
// phases while the lanes are independent of each other, one lane's copies run under another lane's kernels.  Nothing is allocated per
// run.  With the 70 KB of samples per subframe the PCIe link is the limit (a x16 Gen5 link moves about 50 GB/s: ~0.7 M subframes/s per
// GPU), which is the point: the device-resident rate is three times that.
//
// Several devices ("each GPU owns a host thread ... static round-robin of subframes over 1/2/4/8 devices"): the chunks go to the
// devices block-cyclically (chunk c -> device c mod G), every device is driven by its own host thread for the duration of a run, and its
// lanes by that thread alone -- no collective, no peer access, results land in the caller's arrays at the allocation's own index.
//
// Three shapes of input:
//   units + one allocation template   mi_lte_dl_pipeline_run         (a semi-static grant pattern: planned once per lane)
//   units + per-unit allocation lists mi_lte_dl_pipeline_run_units   (a capture after its PDCCH pass: each chunk re-plans a dynamic plan)
//   one contiguous capture            mi_lte_dl_pipeline_run_capture (split on subframe boundaries, each chunk copied with the look-ahead
//                                                                      samples behind its last subframe -- the halo of SURVEY 8e)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "ctx.hpp"

namespace {
struct Lane {
    mi_lte_ctx        *ctx  = nullptr;
    mi_lte_pdsch_plan *plan = nullptr; // the template's plan over a whole chunk (template mode)
    mi_lte_pdsch_plan *dyn  = nullptr; // re-assigned per chunk (per-unit lists, and the ragged last chunk of template mode)
    int8_t   *d_iq = nullptr;
    uint64_t *d_start_units = nullptr, *d_start_capture = nullptr;
    uint32_t *d_sf = nullptr, *d_cell = nullptr; // two halves of one block (d_sf owns it)
    uint32_t *h_meta = nullptr;                  // pinned staging of a chunk's subframe numbers and cells, laid out like that block
    float    *d_sub = nullptr;
    uint8_t  *d_out = nullptr;
    int32_t  *d_st = nullptr;
    hipEvent_t in_free = nullptr, out_free = nullptr; // the front end has read d_iq / the results have left d_out (copy-stream mode)
    hipEvent_t meta_out = nullptr;                    // h_meta has been copied
    bool       used = false;
};
struct Dev {
    int               device = 0;
    std::vector<Lane> lanes;
    std::string       err;
    int               rc = MI_LTE_OK;
    int               node = -1;              // NUMA node of the device's PCIe slot (-1: the system does not say)
    std::vector<int>  node_cpus;              // the CPUs of that node the process may run on
    hipStream_t       h2d = nullptr, d2h = nullptr; // one stream per copy direction (see device_worker)
    std::vector<hipEvent_t> ev;               // four per chunk of the last run: before H2D, after H2D, after the kernels, after D2H
    mi_lte_pipeline_dev_stats stats = {};     // the last run, as this device saw it
};
struct Job { // one run, as the device threads see it
    const int8_t  *h_iq = nullptr;        // units back to back (unit_bytes each), or the capture from its first subframe on
    bool           capture = false;
    const uint32_t *h_sf = nullptr, *h_cell = nullptr;
    uint32_t        n_units = 0;
    const mi_lte_pdsch_alloc *h_allocs = nullptr; // per-unit lists: CSR over units (h_first[u] .. h_first[u+1]); template mode: nullptr
    const uint32_t *h_first = nullptr;
    uint32_t        cfi = 2;
    uint8_t        *h_out = nullptr;
    int32_t        *h_status = nullptr;
};
} // namespace

struct mi_lte_dl_pipeline {
    mi_lte_dl_cfg cfg;
    uint32_t      cfi = 0, n_alloc = 0, chunk = 0, out_stride = 0, max_alloc_per_unit = 0;
    size_t        unit_samples = 0, sf_samples = 0, look_samples = 0, max_soft_per_unit = 0;
    std::vector<mi_lte_pdsch_alloc> tmpl; // the allocation template (template mode)
    std::vector<Dev> devs;
    std::string   err;
};

// ---- host placement: which NUMA node a device hangs off, its CPUs, memory bound to it
static int sysfs_int(const std::string &path, int dflt)
{
    FILE *f = fopen(path.c_str(), "r");
    int   v = dflt;
    if (f) { if (fscanf(f, "%d", &v) != 1) v = dflt; fclose(f); }
    return v;
}
static std::vector<int> parse_cpulist(const std::string &path) // "0-63,128-191"
{
    std::vector<int> cpus;
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return cpus;
    char buf[4096];
    if (fgets(buf, sizeof buf, f)) {
        for (char *t = strtok(buf, ",\n"); t; t = strtok(nullptr, ",\n")) {
            int a, b;
            if (sscanf(t, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; c++) cpus.push_back(c); }
            else if (sscanf(t, "%d", &a) == 1) cpus.push_back(a);
        }
    }
    fclose(f);
    return cpus;
}
static int device_numa_node(int device)
{
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return -1;
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    return sysfs_int(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node", -1);
}
// the CPUs of a node that this process is allowed on (a container's cpuset may be a subset)
static std::vector<int> node_cpus_allowed(int node)
{
    std::vector<int> out;
    if (node < 0) return out;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    for (int c : parse_cpulist("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"))
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) out.push_back(c);
    return out;
}
// pin the calling thread; returns how many CPUs the mask holds (0: not pinned)
static uint32_t pin_this_thread(const std::vector<int> &cpus)
{
    if (cpus.empty()) return 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0 ? (uint32_t)cpus.size() : 0u;
}
namespace {
struct HostBlock { size_t bytes; int node; }; // blocks of mi_lte_host_alloc_on (mapped here, registered with the runtime)
std::mutex                      g_host_mu;
std::map<const void *, HostBlock> g_host_blocks;
} // namespace

static void free_lane(Lane &l)
{
    if (l.ctx) (void)mi_lte_sync(l.ctx);
    if (l.plan) mi_lte_pdsch_plan_destroy(l.ctx, l.plan);
    if (l.dyn) mi_lte_pdsch_plan_destroy(l.ctx, l.dyn);
    (void)hipFree(l.d_iq); (void)hipFree(l.d_start_units); (void)hipFree(l.d_start_capture); (void)hipFree(l.d_sf); if (l.h_meta) (void)hipHostFree(l.h_meta);
    (void)hipFree(l.d_sub); (void)hipFree(l.d_out); (void)hipFree(l.d_st);
    if (l.in_free) (void)hipEventDestroy(l.in_free);
    if (l.out_free) (void)hipEventDestroy(l.out_free);
    if (l.meta_out) (void)hipEventDestroy(l.meta_out);
    if (l.ctx) mi_lte_ctx_destroy(l.ctx);
    l = Lane();
}

// soft bits of an allocation with the largest control region excluded... the smallest one (1 symbol): the upper bound a dynamic plan is sized for
static size_t soft_bound(const mi_lte_pdsch_alloc &a)
{
    const uint32_t qm = a.mod_type == 3 ? 6 : a.mod_type == 2 ? 4 : a.mod_type == 1 ? 2 : 1;
    return ((size_t)13 * a.N_prb * 12 * qm + 63) & ~(size_t)63;
}

static int make_lane(mi_lte_dl_pipeline *p, Dev &d, Lane &l)
{
    int rc = mi_lte_ctx_create(d.device, &l.ctx);
    if (rc != MI_LTE_OK) return rc;
    const uint32_t chunk = p->chunk;
    const size_t   cap_alloc = (size_t)chunk * p->max_alloc_per_unit;
    if (!p->tmpl.empty()) {
        std::vector<mi_lte_pdsch_alloc> allocs((size_t)chunk * p->n_alloc);
        for (uint32_t u = 0; u < chunk; u++)
            for (uint32_t a = 0; a < p->n_alloc; a++) {
                allocs[(size_t)u * p->n_alloc + a]      = p->tmpl[a];
                allocs[(size_t)u * p->n_alloc + a].unit = u;
            }
        rc = mi_lte_pdsch_plan_create(l.ctx, &p->cfg, p->cfi, allocs.data(), (uint32_t)allocs.size(), &l.plan);
        if (rc != MI_LTE_OK) { d.err = mi_lte_last_error(l.ctx); return rc; }
        mi_lte_pdsch_plan_set_output(l.plan, 1);
        mi_pdsch_plan_wide_stride(l.plan); // the same output layout as the dynamic plan's
    }
    rc = mi_lte_pdsch_plan_create_dynamic(l.ctx, &p->cfg, (uint32_t)cap_alloc, (size_t)chunk * p->max_soft_per_unit, &l.dyn);
    if (rc != MI_LTE_OK) { d.err = mi_lte_last_error(l.ctx); return rc; }
    mi_lte_pdsch_plan_set_output(l.dyn, 1);
    p->out_stride = mi_lte_pdsch_plan_out_stride(l.dyn); // the largest single-code-block stride: template and per-unit runs share the output layout
    std::vector<uint64_t> su(chunk), sc(chunk);
    for (uint32_t u = 0; u < chunk; u++) { su[u] = (uint64_t)u * p->unit_samples; sc[u] = (uint64_t)u * p->sf_samples; }
    const size_t iq_bytes = (size_t)chunk * p->unit_samples * 2 + 64;
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_iq, iq_bytes));
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_start_units, sizeof(uint64_t) * chunk));
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_start_capture, sizeof(uint64_t) * chunk));
    const uint32_t chunk_al = (chunk + 3) & ~3u; // (16-byte aligned halves)
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_sf, sizeof(uint32_t) * 2 * chunk_al));
    l.d_cell = l.d_sf + chunk_al;
    MI_HIP_CHECK(l.ctx, hipHostMalloc((void **)&l.h_meta, sizeof(uint32_t) * 2 * chunk_al, hipHostMallocMapped));
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_sub, sizeof(float) * mi_lte_subframe_floats(p->cfg.N_ant) * chunk));
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_out, cap_alloc * p->out_stride));
    MI_HIP_CHECK(l.ctx, hipMalloc((void **)&l.d_st, cap_alloc * sizeof(int32_t)));
    MI_HIP_CHECK(l.ctx, hipMemcpy(l.d_start_units, su.data(), sizeof(uint64_t) * chunk, hipMemcpyHostToDevice));
    MI_HIP_CHECK(l.ctx, hipMemcpy(l.d_start_capture, sc.data(), sizeof(uint64_t) * chunk, hipMemcpyHostToDevice));
    MI_HIP_CHECK(l.ctx, hipMemset(l.d_sf, 0, sizeof(uint32_t) * 2 * chunk_al));
    MI_HIP_CHECK(l.ctx, hipMemset(l.d_sub, 0, sizeof(float) * mi_lte_subframe_floats(p->cfg.N_ant) * chunk));
    MI_HIP_CHECK(l.ctx, hipMemset(l.d_iq, 0, iq_bytes));
    MI_HIP_CHECK(l.ctx, hipEventCreateWithFlags(&l.in_free, hipEventDisableTiming));
    MI_HIP_CHECK(l.ctx, hipEventCreateWithFlags(&l.out_free, hipEventDisableTiming));
    MI_HIP_CHECK(l.ctx, hipEventCreateWithFlags(&l.meta_out, hipEventDisableTiming));
    return MI_LTE_OK;
}

// The chunks of one device, in order, on its lanes round robin.  Runs on the device's own host thread.
static void device_worker(mi_lte_dl_pipeline *p, uint32_t di, const Job *job)
{
    Dev           &d = p->devs[di];
    const uint32_t G = (uint32_t)p->devs.size(), n_chunks = (job->n_units + p->chunk - 1) / p->chunk;
    d.rc = MI_LTE_OK;
    d.err.clear();
    d.stats = {};
    d.stats.device = (uint32_t)d.device; d.stats.numa_node = -1;
    const auto t_start = std::chrono::steady_clock::now();
    if (hipSetDevice(d.device) != hipSuccess) { d.rc = MI_LTE_ERR_HIP; d.err = "hipSetDevice failed"; return; }
    if (G > 1) { // a thread of the pipeline's own: keep it (and the staging it does) next to its device
        d.stats.n_cpus = pin_this_thread(d.node_cpus);
        if (d.stats.n_cpus) d.stats.numa_node = d.node;
    }
    const size_t unit_bytes = p->unit_samples * 2;
    uint32_t     k = 0; // chunks this device has taken
    const uint32_t my_chunks = di < n_chunks ? (n_chunks - di + G - 1) / G : 0;
    while (d.ev.size() < (size_t)4 * my_chunks) { // (created once, reused by later runs)
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { d.rc = MI_LTE_ERR_HIP; d.err = "hipEventCreate failed"; return; }
        d.ev.push_back(e);
    }
    std::vector<mi_lte_pdsch_alloc> local;
    std::vector<size_t>             refused; // allocations outside the decodable envelope: status 2 at their own index
    auto fail = [&](Lane &l, int rc) { d.rc = rc; d.err = mi_lte_last_error(l.ctx); };
    // Copies on streams of their own (default): ONE stream carries every host-to-device copy of the device in chunk order, one every
    // device-to-host copy, the lanes' streams carry kernels only and meet the copies through events.  A lane's stream that also carried its
    // copies (rounds 2-4, MI_LTE_PIPELINE_LANE_COPIES=1) put up to `lanes` input copies on the link at once, each behind its own lane's
    // previous results copy: 43 GB/s of the 57.6 GB/s one copy alone reaches.  In order on one stream they run back to back at that rate.
    static const bool lane_copies = getenv("MI_LTE_PIPELINE_LANE_COPIES") != nullptr;
    for (Lane &l : d.lanes) l.used = false;
    for (uint32_t c = di; c < n_chunks; c += G, k++) {
        Lane          &l  = d.lanes[k % d.lanes.size()];
        const uint32_t u0 = c * p->chunk, n = std::min(p->chunk, job->n_units - u0);
        hipStream_t    st = (hipStream_t)mi_lte_stream(l.ctx), s_in = lane_copies ? st : d.h2d, s_out = lane_copies ? st : d.d2h;
        hipError_t     e;
        hipEvent_t    *ev = &d.ev[(size_t)4 * k];
        if (!lane_copies && l.used) (void)hipStreamWaitEvent(s_in, l.in_free, 0); // the lane's previous chunk has been read out of d_iq
        (void)hipEventRecord(ev[0], s_in);
        const size_t   in_bytes = job->capture ? ((size_t)n * p->sf_samples + p->look_samples) * 2 : (size_t)n * unit_bytes;
        d.stats.chunks++; d.stats.units += n; d.stats.h2d_bytes += in_bytes + 8 * (size_t)n;
        if (job->capture) // the chunk's subframes and the look-ahead samples behind the last of them, as they lie in the capture
            e = hipMemcpyAsync(l.d_iq, job->h_iq + (size_t)u0 * p->sf_samples * 2, ((size_t)n * p->sf_samples + p->look_samples) * 2, hipMemcpyHostToDevice, s_in);
        else
            e = hipMemcpyAsync(l.d_iq, job->h_iq + (size_t)u0 * unit_bytes, (size_t)n * unit_bytes, hipMemcpyHostToDevice, s_in);
        // subframe numbers and cells: through the lane's pinned block and ONE copy kernel (two more copy commands per chunk made the runtime
        // rotate its engines between the sample copies, and each costs an engine's start-up on the input stream)
        static const char *meta_env = getenv("MI_LTE_PIPELINE_META_KERNEL"); // (A/B switch: "0" / "1")
        const bool meta_by_kernel = !lane_copies && (meta_env ? meta_env[0] == '1' : job->h_allocs != nullptr);
        if (e == hipSuccess && !meta_by_kernel) {
            e = hipMemcpyAsync(l.d_sf, job->h_sf + u0, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s_in);
            if (e == hipSuccess) e = hipMemcpyAsync(l.d_cell, job->h_cell + u0, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s_in);
        } else if (e == hipSuccess) {
            if (l.used) (void)hipEventSynchronize(l.meta_out); // the lane's previous chunk has left the staging block (long ago)
            const uint32_t half = (uint32_t)(l.d_cell - l.d_sf);
            memcpy(l.h_meta, job->h_sf + u0, sizeof(uint32_t) * n);
            memcpy(l.h_meta + half, job->h_cell + u0, sizeof(uint32_t) * n);
            // Only the n refreshed entries of each half are moved (two segments of one launch).  ORDER: this kernel writes l.d_sf / l.d_cell, which
            // the lane's PREVIOUS chunk's kernels read -- the only thing that keeps it behind them is the wait on l.in_free issued on s_in at the
            // top of this iteration (copy-stream mode only, which is the only mode that takes this branch: meta_by_kernel implies !lane_copies).
            // Anything that writes l.d_sf earlier than that wait breaks the previous chunk.
            const MiCopySeg seg[2] = {{l.d_sf, l.h_meta, sizeof(uint32_t) * (size_t)((n + 3u) & ~3u)}, {l.d_cell, l.h_meta + half, sizeof(uint32_t) * (size_t)((n + 3u) & ~3u)}};
            e = mi_pinned_segments_to_device(l.ctx, seg, 2, s_in);
            if (e == hipSuccess) e = hipEventRecord(l.meta_out, s_in);
        }
        if (e != hipSuccess) { d.rc = MI_LTE_ERR_HIP; d.err = hipGetErrorString(e); break; }
        // per-unit lists: the chunk's slice of the caller's list goes into the lane's dynamic plan now, its descriptor copies behind the samples
        // on the input stream (the host's part of it, ~1 ms, runs while the samples are in flight)
        int  rc_assign = MI_LTE_OK;
        bool assigned = false;
        if (job->h_allocs && !lane_copies && job->h_first[u0 + n] > job->h_first[u0]) {
            std::vector<uint32_t> bad;
            const size_t a0s = job->h_first[u0];
            static const bool slice_on_input = getenv("MI_LTE_SLICE_ON_LANE_STREAM") == nullptr; // (A/B switch)
            rc_assign = mi_pdsch_plan_assign_slice(l.ctx, l.dyn, job->cfi, job->h_allocs + a0s, (uint32_t)(job->h_first[u0 + n] - a0s), u0, &bad, slice_on_input ? s_in : nullptr);
            for (uint32_t i : bad) refused.push_back(a0s + i);
            assigned = true;
        }
        (void)hipEventRecord(ev[1], s_in);
        if (rc_assign != MI_LTE_OK) { fail(l, rc_assign); break; }
        if (!lane_copies) {
            (void)hipStreamWaitEvent(st, ev[1], 0);                      // the chunk's samples are on the device
            if (l.used) (void)hipStreamWaitEvent(st, l.out_free, 0);    // the lane's previous results have left d_out
        }
        int rc = mi_lte_dl_frontend_batch(l.ctx, &p->cfg, l.d_iq, nullptr, job->capture ? l.d_start_capture : l.d_start_units, l.d_sf, l.d_cell, n, l.d_sub);
        if (rc != MI_LTE_OK) { fail(l, rc); break; }
        // which plan decodes the chunk, and where its results go
        mi_lte_pdsch_plan *plan;
        size_t             a0, n_al;
        if (job->h_allocs) { // per-unit lists: the chunk's slice of the caller's list, unit numbers made chunk-local
            a0   = job->h_first[u0];
            n_al = job->h_first[u0 + n] - a0;
            if (n_al == 0) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); (void)hipEventRecord(l.in_free, st); (void)hipEventRecord(l.out_free, st); l.used = true; continue; }
            if (!assigned) {
                std::vector<uint32_t> bad;
                rc = mi_pdsch_plan_assign_slice(l.ctx, l.dyn, job->cfi, job->h_allocs + a0, (uint32_t)n_al, u0, &bad);
                for (uint32_t i : bad) refused.push_back(a0 + i);
            }
            plan = l.dyn;
        } else if (n == p->chunk) {
            a0 = (size_t)u0 * p->n_alloc; n_al = (size_t)n * p->n_alloc;
            plan = l.plan;
        } else { // the ragged last chunk of a template run: the template over n units only, in the dynamic plan
            a0 = (size_t)u0 * p->n_alloc; n_al = (size_t)n * p->n_alloc;
            local.resize(n_al);
            for (uint32_t u = 0; u < n; u++)
                for (uint32_t a = 0; a < p->n_alloc; a++) { local[(size_t)u * p->n_alloc + a] = p->tmpl[a]; local[(size_t)u * p->n_alloc + a].unit = u; }
            rc = mi_lte_pdsch_plan_assign(l.ctx, l.dyn, p->cfi, local.data(), (uint32_t)n_al);
            plan = l.dyn;
        }
        if (rc == MI_LTE_OK) rc = mi_lte_pdsch_decode_run(l.ctx, plan, l.d_sub, l.d_sf, l.d_cell, l.d_out, l.d_st);
        if (rc != MI_LTE_OK) { fail(l, rc); break; }
        (void)hipEventRecord(ev[2], st);
        (void)hipEventRecord(l.in_free, st); // the kernels have read d_iq, d_sf, d_cell: the lane's next input may come
        if (!lane_copies) (void)hipStreamWaitEvent(s_out, ev[2], 0);
        e = hipMemcpyAsync(job->h_out + a0 * p->out_stride, l.d_out, n_al * p->out_stride, hipMemcpyDeviceToHost, s_out);
        if (e == hipSuccess) e = hipMemcpyAsync(job->h_status + a0, l.d_st, n_al * sizeof(int32_t), hipMemcpyDeviceToHost, s_out);
        if (e != hipSuccess) { d.rc = MI_LTE_ERR_HIP; d.err = hipGetErrorString(e); break; }
        (void)hipEventRecord(ev[3], s_out);
        (void)hipEventRecord(l.out_free, s_out);
        l.used = true;
        d.stats.d2h_bytes += n_al * ((size_t)p->out_stride + sizeof(int32_t));
    }
    const uint32_t k_done = k; // chunks whose four events were all recorded
    // (a failed chunk leaves the loop, not the function: copies of earlier chunks into the caller's arrays may still be in flight)
    for (Lane &l : d.lanes) {
        const int rc = mi_lte_sync(l.ctx);
        if (rc != MI_LTE_OK && d.rc == MI_LTE_OK) fail(l, rc);
    }
    if ((hipStreamSynchronize(d.h2d) != hipSuccess || hipStreamSynchronize(d.d2h) != hipSuccess) && d.rc == MI_LTE_OK) { d.rc = MI_LTE_ERR_HIP; d.err = "copy stream failed"; }
    for (size_t i : refused) { // LIBLTE_ERROR_DECODE_FAIL, and an all-zero row instead of the stand-in's bits (after the copies of the stand-ins' results have landed)
        job->h_status[i] = 2;
        memset(job->h_out + i * p->out_stride, 0, p->out_stride);
    }
    if (d.rc == MI_LTE_OK)
        for (uint32_t c = 0; c < k_done; c++) { // every lane has been waited for: the events are complete
            float ms[3] = {0, 0, 0};
            for (int ph = 0; ph < 3; ph++) (void)hipEventElapsedTime(&ms[ph], d.ev[(size_t)4 * c + ph], d.ev[(size_t)4 * c + ph + 1]);
            d.stats.h2d_s += ms[0] * 1e-3; d.stats.kernel_s += ms[1] * 1e-3; d.stats.d2h_s += ms[2] * 1e-3;
            if (getenv("MI_LTE_PIPELINE_TRACE")) { // the chunk's four events on the clock of the run's first
                float t0 = 0;
                (void)hipEventElapsedTime(&t0, d.ev[0], d.ev[(size_t)4 * c]);
                fprintf(stderr, "  device %d chunk %2u lane %u: H2D %7.3f .. %7.3f ms, kernels .. %7.3f, D2H .. %7.3f\n", d.device, c, (unsigned)(c % d.lanes.size()), t0, t0 + ms[0],
                        t0 + ms[0] + ms[1], t0 + ms[0] + ms[1] + ms[2]);
            }
        }
    d.stats.wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
}

static int run_job(mi_lte_dl_pipeline *p, const Job &job)
{
    p->err.clear();
    if (p->devs.size() == 1) device_worker(p, 0, &job); // one device: the caller's thread drives it
    else {
        std::vector<std::thread> th;
        for (uint32_t di = 0; di < p->devs.size(); di++) th.emplace_back(device_worker, p, di, &job);
        for (auto &t : th) t.join();
    }
    for (Dev &d : p->devs)
        if (d.rc != MI_LTE_OK) { p->err = d.err; return d.rc; }
    return MI_LTE_OK;
}

extern "C" {

void *mi_lte_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) == hipSuccess ? p : nullptr; // portable: pinned for every device of the process
}
void mi_lte_host_free(void *p)
{
    if (!p) return;
    HostBlock b{0, -1};
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_blocks.find(p);
        if (it != g_host_blocks.end()) { b = it->second; g_host_blocks.erase(it); }
    }
    if (b.bytes) { (void)hipHostUnregister(p); (void)munmap(p, b.bytes); }
    else         (void)hipHostFree(p);
}

int mi_lte_device_numa_node(int device) { return device_numa_node(device); }

void *mi_lte_host_alloc_on(int device, size_t bytes)
{
    // device < 0: one block that every device reads a share of (a contiguous capture) -- its pages interleaved over all memory nodes
    std::vector<int> nodes = device < 0 ? parse_cpulist("/sys/devices/system/node/online") : std::vector<int>();
    const bool interleave = device < 0 && nodes.size() > 1;
    const int  node = device < 0 ? (interleave ? -2 : -1) : device_numa_node(device);
    if ((node >= 0 && node < 1024) || interleave) {
        const size_t len = ((bytes ? bytes : 1) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            unsigned long mask[16] = {0};
            if (!interleave) nodes.assign(1, node);
            for (int nd : nodes)
                if (nd >= 0 && nd < 1024) mask[nd / (8 * sizeof(unsigned long))] |= 1ul << (nd % (8 * sizeof(unsigned long)));
#ifdef SYS_mbind
            const long rc = syscall(SYS_mbind, m, len, interleave ? 3 /* MPOL_INTERLEAVE */ : 2 /* MPOL_BIND */, mask, (unsigned long)(8 * sizeof mask + 1), 0u); // maxnode counts bits + 1 (the kernel reads maxnode - 1 of them)
#else
            const long rc = -1;
#endif
            if (rc == 0) {
                (void)madvise(m, len, MADV_HUGEPAGE);
                memset(m, 0, len); // first touch under the policy: the pages exist, on that node, before the runtime pins them
                if (hipHostRegister(m, len, hipHostRegisterPortable) == hipSuccess) {
                    std::lock_guard<std::mutex> lk(g_host_mu);
                    g_host_blocks[m] = HostBlock{len, node};
                    return m;
                }
            }
            (void)munmap(m, len);
        }
    }
    return mi_lte_host_alloc(bytes); // node unknown, binding refused (a container without CAP_SYS_NICE), or registration failed
}

int mi_lte_host_alloc_node(const void *p)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_blocks.find(p);
    return it == g_host_blocks.end() ? -1 : it->second.node;
}

int mi_lte_dl_pipeline_device_stats(const mi_lte_dl_pipeline *p, uint32_t index, mi_lte_pipeline_dev_stats *out)
{
    if (!p || !out || index >= p->devs.size()) return MI_LTE_ERR_INVALID_ARG;
    *out = p->devs[index].stats;
    return MI_LTE_OK;
}

void mi_lte_dl_pipeline_destroy(mi_lte_dl_pipeline *p)
{
    if (!p) return;
    for (Dev &d : p->devs) {
        (void)hipSetDevice(d.device);
        if (d.h2d) (void)hipStreamSynchronize(d.h2d);
        if (d.d2h) (void)hipStreamSynchronize(d.d2h);
        for (Lane &l : d.lanes) free_lane(l);
        for (hipEvent_t e : d.ev) (void)hipEventDestroy(e);
        if (d.h2d) (void)hipStreamDestroy(d.h2d);
        if (d.d2h) (void)hipStreamDestroy(d.d2h);
    }
    delete p;
}

// h_unit_allocs != NULL: template mode (every unit carries these n_alloc_per_unit allocations; per-unit runs are possible too, with at most
// that many allocations per unit).  h_unit_allocs == NULL: per-unit lists only, at most n_alloc_per_unit allocations and max_soft_bytes_per_unit
// soft bits per unit ON AVERAGE over a chunk (0: a full-band 64QAM allocation per unit).
int mi_lte_dl_pipeline_create_multi(const int *devices, uint32_t n_devices, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs,
                                    const mi_lte_pdsch_alloc *h_unit_allocs, uint32_t n_alloc_per_unit, size_t max_soft_bytes_per_unit, uint32_t chunk_units,
                                    uint32_t n_lanes, mi_lte_dl_pipeline **out)
{
    if (!devices || n_devices == 0 || n_devices > 64 || !cfg || !out || n_alloc_per_unit == 0 || chunk_units == 0 || n_lanes == 0 || n_lanes > 8)
        return MI_LTE_ERR_INVALID_ARG;
    if ((cfg->sample_format & 0xFFu) != MI_LTE_IQ_I8) return MI_LTE_ERR_INVALID_ARG; // host batches are int8 captures
    if (N_pdcch_symbs < 1 || N_pdcch_symbs > 4) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048)) return MI_LTE_ERR_INVALID_ARG;
    auto *p = new mi_lte_dl_pipeline();
    auto  guard = on_fail([&] { mi_lte_dl_pipeline_destroy(p); });
    p->cfg = *cfg; p->cfi = N_pdcch_symbs; p->n_alloc = n_alloc_per_unit; p->max_alloc_per_unit = n_alloc_per_unit; p->chunk = chunk_units;
    const uint32_t sc = 2048 / N;
    p->sf_samples   = 30720 / sc;
    p->look_samples = 4400 / sc + (4400 % sc ? 1 : 0);
    p->unit_samples = ((30720 + 4400) / sc + 15) / 16 * 16; // one subframe + the two look-ahead symbols, a multiple of 16 samples (= mi_lte_synth_unit_len)
    if (h_unit_allocs) {
        p->tmpl.assign(h_unit_allocs, h_unit_allocs + n_alloc_per_unit);
        size_t s = 0;
        for (auto &a : p->tmpl) s += soft_bound(a);
        p->max_soft_per_unit = std::max(s, max_soft_bytes_per_unit);
    } else
        p->max_soft_per_unit = max_soft_bytes_per_unit ? ((max_soft_bytes_per_unit + 63) & ~(size_t)63) * 1 + 64 * (size_t)n_alloc_per_unit
                                                       : (size_t)13 * cfg->N_rb_dl * 12 * 6 + 64 * (size_t)n_alloc_per_unit;
    p->devs.resize(n_devices);
    for (uint32_t di = 0; di < n_devices; di++) {
        Dev &d   = p->devs[di];
        d.device = devices[di];
        d.lanes.resize(n_lanes);
        if (hipSetDevice(d.device) != hipSuccess) return MI_LTE_ERR_NO_DEVICE;
        d.node      = device_numa_node(d.device);
        d.node_cpus = node_cpus_allowed(d.node);
        // (a high-priority input stream was tried for the sake of its copy kernels: 0.75 -> 0.65 M subframes/s, profiles/r05_host_pipeline_profile.txt section 7)
        if (hipStreamCreateWithFlags(&d.h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&d.d2h, hipStreamNonBlocking) != hipSuccess) return MI_LTE_ERR_HIP;
        for (Lane &l : d.lanes) {
            const int rc = make_lane(p, d, l);
            if (rc != MI_LTE_OK) { p->err = d.err; return rc; }
        }
    }
    guard.armed = false;
    *out = p;
    return MI_LTE_OK;
}

int mi_lte_dl_pipeline_create(int device, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_unit_allocs, uint32_t n_alloc_per_unit,
                              uint32_t chunk_units, uint32_t n_lanes, mi_lte_dl_pipeline **out)
{
    if (!h_unit_allocs) return MI_LTE_ERR_INVALID_ARG;
    return mi_lte_dl_pipeline_create_multi(&device, 1, cfg, N_pdcch_symbs, h_unit_allocs, n_alloc_per_unit, 0, chunk_units, n_lanes, out);
}

uint32_t    mi_lte_dl_pipeline_out_stride(const mi_lte_dl_pipeline *p) { return p ? p->out_stride : 0; }
size_t      mi_lte_dl_pipeline_unit_samples(const mi_lte_dl_pipeline *p) { return p ? p->unit_samples : 0; }
uint32_t    mi_lte_dl_pipeline_n_devices(const mi_lte_dl_pipeline *p) { return p ? (uint32_t)p->devs.size() : 0; }
const char *mi_lte_dl_pipeline_last_error(const mi_lte_dl_pipeline *p)
{
    if (!p) return "null pipeline";
    return p->err.c_str();
}

// h_iq: n_units units of unit_samples int8 I,Q pairs back to back; h_out_packed: [n_units * n_alloc][out_stride]; h_status likewise.
// The host arrays should be pinned (mi_lte_host_alloc): with pageable memory the copies fall back to the driver's staged path and
// stop overlapping.  Returns when every result is in host memory.
int mi_lte_dl_pipeline_run(mi_lte_dl_pipeline *p, const int8_t *h_iq, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell, uint32_t n_units,
                           uint8_t *h_out_packed, int32_t *h_status)
{
    if (!p || !h_iq || !h_subfr_num || !h_n_id_cell || !h_out_packed || !h_status || n_units == 0) return MI_LTE_ERR_INVALID_ARG;
    if (p->tmpl.empty()) { p->err = "the pipeline was created without an allocation template: use mi_lte_dl_pipeline_run_units"; return MI_LTE_ERR_INVALID_ARG; }
    Job job;
    job.h_iq = h_iq; job.h_sf = h_subfr_num; job.h_cell = h_n_id_cell; job.n_units = n_units; job.h_out = h_out_packed; job.h_status = h_status;
    return run_job(p, job);
}

static int check_lists(mi_lte_dl_pipeline *p, const mi_lte_pdsch_alloc *h_allocs, const uint32_t *h_first, uint32_t n_units)
{
    for (uint32_t u = 0; u < n_units; u++) {
        if (h_first[u + 1] < h_first[u] || h_first[u + 1] - h_first[u] > p->max_alloc_per_unit) { p->err = "allocation list: not sorted by unit, or more allocations in a unit than the pipeline was created for"; return MI_LTE_ERR_INVALID_ARG; }
        for (uint32_t a = h_first[u]; a < h_first[u + 1]; a++)
            if (h_allocs[a].unit != u) { p->err = "allocation list: allocs[h_first[u] .. h_first[u+1]) must carry unit == u"; return MI_LTE_ERR_INVALID_ARG; }
    }
    return MI_LTE_OK;
}

int mi_lte_dl_pipeline_run_units(mi_lte_dl_pipeline *p, const int8_t *h_iq, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell, uint32_t n_units,
                                 const mi_lte_pdsch_alloc *h_allocs, const uint32_t *h_first, uint32_t N_pdcch_symbs, uint8_t *h_out_packed, int32_t *h_status)
{
    if (!p || !h_iq || !h_subfr_num || !h_n_id_cell || !h_allocs || !h_first || !h_out_packed || !h_status || n_units == 0 || N_pdcch_symbs < 1 || N_pdcch_symbs > 4)
        return MI_LTE_ERR_INVALID_ARG;
    const int rc = check_lists(p, h_allocs, h_first, n_units);
    if (rc != MI_LTE_OK) return rc;
    Job job;
    job.h_iq = h_iq; job.h_sf = h_subfr_num; job.h_cell = h_n_id_cell; job.n_units = n_units; job.h_allocs = h_allocs; job.h_first = h_first; job.cfi = N_pdcch_symbs;
    job.h_out = h_out_packed; job.h_status = h_status;
    return run_job(p, job);
}

int mi_lte_dl_pipeline_run_capture(mi_lte_dl_pipeline *p, const int8_t *h_capture, uint64_t n_samples, uint64_t first_subframe_start, uint32_t n_subframes,
                                   uint32_t first_subfr_num, uint32_t N_id_cell, const mi_lte_pdsch_alloc *h_allocs, const uint32_t *h_first, uint32_t N_pdcch_symbs,
                                   uint8_t *h_out_packed, int32_t *h_status)
{
    if (!p || !h_capture || !h_allocs || !h_first || !h_out_packed || !h_status || n_subframes == 0 || N_pdcch_symbs < 1 || N_pdcch_symbs > 4 || N_id_cell > 503)
        return MI_LTE_ERR_INVALID_ARG;
    if (first_subframe_start + (uint64_t)n_subframes * p->sf_samples + p->look_samples > n_samples) {
        p->err = "capture too short: the last subframe needs its two look-ahead symbols";
        return MI_LTE_ERR_INVALID_ARG;
    }
    const int rc = check_lists(p, h_allocs, h_first, n_subframes);
    if (rc != MI_LTE_OK) return rc;
    std::vector<uint32_t> sf(n_subframes), cell(n_subframes, N_id_cell);
    for (uint32_t u = 0; u < n_subframes; u++) sf[u] = (first_subfr_num + u) % 10;
    Job job;
    job.h_iq = h_capture + first_subframe_start * 2; job.capture = true; job.h_sf = sf.data(); job.h_cell = cell.data(); job.n_units = n_subframes;
    job.h_allocs = h_allocs; job.h_first = h_first; job.cfi = N_pdcch_symbs; job.h_out = h_out_packed; job.h_status = h_status;
    return run_job(p, job);
}

} // extern "C"
