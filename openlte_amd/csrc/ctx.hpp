// Internal context of libmi_lte.so: one GPU, one stream, grow-only scratch, per-K table cache.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/mi_lte.h"

struct TurboTables {       // device-resident per-K tables
    uint16_t *d_pi  = nullptr; // interleaver index pi[i]
    uint16_t *d_inv = nullptr; // inv[j] = largest i with pi[i] == j, 0xFFFF if none ("hole")
    uint16_t *d_inv2 = nullptr; // k_turbo_vote's form: 2 * inv[j] (a byte offset into its int16 array), holes and the entries from K up to the
                                // next multiple of 16 = 2 * kpad64(K), the offset of the array's zero slot
    uint32_t *d_pi_row = nullptr, *d_inv_row = nullptr; // the BCJR kernels' form (32-bit entries: read with scalar loads), kpad64(K) entries each: pi[i] resp. inv[j] as a row index, with
                                                        // holes and the entries from K on = kpad64(K), the index of the all-zero row
};

struct RmTables { // per-K rank tables of the fused turbo rate un-matching (see turbo.hip)
    uint16_t *d_tabs = nullptr;
    uint32_t *d_nnn  = nullptr;
};

// Per-pass twiddle tables behind the 4096-entry table (offsets in float2 from d_fft_tw + 4096; see mi_ctx_fft_twiddles): second pass
// (sub-transform length 8, radix 8: 4 slots per butterfly), third pass (64, radix 8; also the last pass of the 512-point transform),
// and the last pass of the 2048 / 1024 / 256 / 128-point transforms (radix 4: 2 slots, radix 2: 1 slot)
enum : uint32_t { MI_FFT_TWC_P2 = 0, MI_FFT_TWC_P3 = 32, MI_FFT_TWC_L2048 = 288, MI_FFT_TWC_L1024 = 1312, MI_FFT_TWC_L256 = 1824, MI_FFT_TWC_L128 = 1952,
                  MI_FFT_TWC_X2 = 2016, MI_FFT_TWC_X3 = 2048, // the radix-16 passes of the 2048-point transform: w, w^2, w^4, w^8 per butterfly
                  MI_FFT_TWC_TOTAL = 2560 };

constexpr uint32_t MI_CRC_TAB_BIAS = 8;

struct mi_lte_ctx {
    int                device = -1;
    hipStream_t        stream = nullptr;
    hipEvent_t         ev0 = nullptr, ev1 = nullptr;
    std::string        err;
    std::string        dev_name;
    std::string        last_kernels;
    double             copy_rates[3] = {0, 0, 0}; // mi_lte_device_copy_rate's three kernel shapes, GB/s of the last call
    void              *scratch       = nullptr;
    size_t             scratch_bytes = 0;
    // a second stream with its own scratch: the smaller block-size groups of a PDSCH decode run there next to the largest one (chain.hip)
    hipStream_t        side_stream = nullptr;
    hipEvent_t         ev_fork = nullptr, ev_join = nullptr;
    void              *side_scratch       = nullptr;
    size_t             side_scratch_bytes = 0;
    uint32_t          *h_flag = nullptr, *d_flag = nullptr; // the completion word of the per-call waits (mi_stream_wait_polling) and its sequence number
    uint32_t           flag_seq = 0;
    uint32_t *bcjr_early_buf = nullptr; size_t bcjr_early_cap = 0; // ctx-owned copy of the change words (the scratch they are counted in is shared with every other kernel family)
    struct { uint32_t *chg; uint32_t n_pairs, n_iter, n_cb; } bcjr_early = {nullptr, 0, 0, 0}; // the last early-termination decode's change words (bcjr.hip)
    bool               bcjr_block_lds_set = false; // hipFuncSetAttribute(k_bcjr_block, max dynamic LDS) made on this context's device
    uint32_t           siso_small_max = 4096; // code blocks per decode up to which k_turbo_siso_small runs (mi_lte_set_turbo_small_batch)
    bool               merged_decode = true;  // several block sizes in one decode: one launch set over all of them (mi_lte_set_turbo_merged; turbo.hip: KSeg)
    void              *h_small = nullptr, *d_small = nullptr; // MI_SMALL_BYTES of pinned host memory the kernels can write (mi_ctx_small_results)
    void              *h_bounce = nullptr, *d_bounce = nullptr; // 4 MiB of pinned host memory mapped into the device: mi_lte_memcpy_* move mid-size copies through it with a kernel
    hipEvent_t         ev_bounce[2] = {nullptr, nullptr};       // ... in two halves: one event behind the copy kernel of each
    bool               bounce_failed = false;                   // the block could not be made: the runtime's copies from then on
    std::map<uint64_t, TurboTables> turbo_tables; // key = K | (spec << 32)
    std::map<uint32_t, RmTables>    rm_tables;    // key = K
    std::vector<void *> owned;                    // allocations released at destroy

    // Gold-sequence tables (see mi_ctx_gold_tables)
    uint32_t *d_gold_x1 = nullptr, *d_gold_x2b = nullptr;
    uint32_t  gold_words = 0;
    void     *d_pusch_shapes = nullptr; // uplink.hip: one PuschShape per N_prb (transform pre-decoding passes), made on first use

    float2   *d_fft_tw  = nullptr; // exp(-2*pi*i*k/4096), k = 0..4095, then the per-pass tables at 4096 + MI_FFT_TWC_* (mi_ctx_fft_twiddles)
    uint32_t *d_crc_tab = nullptr; // (x^e mod gCRC24A) << 8 at index MI_CRC_TAB_BIAS + e, e = -8..6143 (mi_ctx_crc_table)
    float2   *d_prach_tab = nullptr; // chirp | filter spectrum | twiddles of the 839-point chirp-z transform (prach.hip)
    float2   *d_prach_tab4 = nullptr; // the same for the 139-point sequences of preamble format 4

    // optional per-launch HIP-event bracketing (mi_lte_profile_*): pairs are resolved at report time
    bool                                   prof_on = false, prof_armed = false;
    std::vector<hipEvent_t>                prof_pool;
    size_t                                 prof_used = 0;
    std::vector<std::pair<const char *, size_t>> prof_recs; // (kernel name, index of start event)
    std::string                            prof_report;

    // persistent state of the per-call host forms (hostapi.cc): staging buffers, the device subframe, plan caches
    void *host_cache = nullptr;
    void (*host_cache_free)(mi_lte_ctx *) = nullptr;
};

void mi_prof_begin(mi_lte_ctx *ctx, const char *name);
void mi_prof_end(mi_lte_ctx *ctx);

// launch on the context's stream, bracketed by events when profiling is enabled
#define MI_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                         \
    do {                                                                                             \
        mi_prof_begin((ctx), (name));                                                                \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);                  \
        mi_prof_end((ctx));                                                                          \
    } while (0)

#define MI_HIP_CHECK(ctx, call)                                                                      \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return MI_LTE_ERR_HIP;                                                                   \
        }                                                                                            \
    } while (0)

// host <-> device copy AND wait through mi_lte_memcpy_* (ctx.cc): mid-size copies keep off the runtime's copy engines, whose first
// use costs 7-9 ms each; for the tables, plan arrays and results the library itself moves
#define MI_H2D(ctx, d, h, n) do { const int rc_ = mi_lte_memcpy_h2d((ctx), (d), (h), (n)); if (rc_ != MI_LTE_OK) return rc_; } while (0)
#define MI_D2H(ctx, h, d, n) do { const int rc_ = mi_lte_memcpy_d2h((ctx), (h), (d), (n)); if (rc_ != MI_LTE_OK) return rc_; } while (0)

// runs f when the enclosing function leaves without having disarmed it: the error paths of the plan constructors (every MI_HIP_CHECK
// is a return) wait for the stream -- copies from host vectors that are about to go out of scope may still be queued -- and release
// what was built so far
template <typename F> struct OnFail {
    F    f;
    bool armed = true;
    ~OnFail() { if (armed) f(); }
};
template <typename F> OnFail<F> on_fail(F f) { return OnFail<F>{f}; }

int   mi_ctx_reserve_scratch(mi_lte_ctx *ctx, size_t bytes);
hipError_t mi_pinned_to_device(mi_lte_ctx *ctx, void *d_dst, const void *h_pinned, size_t bytes, hipStream_t stream = nullptr); // asynchronous, from / to a MAPPED pinned block (ctx.cc); stream: the context's unless given
hipError_t mi_device_to_pinned(mi_lte_ctx *ctx, void *h_pinned, const void *d_src, size_t bytes);
struct MiCopySeg { void *dst; const void *src; size_t bytes; };
hipError_t mi_pinned_segments_to_device(mi_lte_ctx *ctx, const MiCopySeg *segs, uint32_t n_seg, hipStream_t stream); // 1..3 pinned blocks, one launch (ctx.cc)
// Where a kernel puts a few KB of results that the host reads right after the wait: pinned host memory mapped into the device (no copy
// command, which costs more API time than a small call's kernel runs for).  *h and *d are the host's and the device's pointer to the same
// bytes; MI_LTE_ERR_INVALID_ARG when bytes > MI_SMALL_BYTES (the caller then takes its scratch + copy route).
constexpr size_t MI_SMALL_BYTES = 64 * 1024;
int   mi_ctx_small_results(mi_lte_ctx *ctx, size_t bytes, void **h, void **d);
hipError_t mi_stream_wait(mi_lte_ctx *ctx, size_t n_units); // polls for a call on a handful of units, sleeps on the interrupt for a batch
hipError_t mi_stream_wait_polling(mi_lte_ctx *ctx); // the per-call forms' wait: polls the stream instead of sleeping on an interrupt (ctx.cc)
int   mi_ctx_gold_tables(mi_lte_ctx *ctx);
int   mi_ctx_crc_table(mi_lte_ctx *ctx);
int   mi_ctx_fft_twiddles(mi_lte_ctx *ctx);
int   mi_turbo_ref_group(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, const mi_lte_pdsch_alloc *d_allocs,
                         const uint32_t *d_cb_alloc, const int8_t *d_e, const uint32_t *d_e_off, const uint32_t *d_e_len,
                         uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status, uint32_t e_max_bytes, bool ul = false, bool packed = false);
// the merged REF decode of a batch with many code-block sizes (turbo.hip: KSeg, mi_turbo_ref_multi)
struct MiKGroup { uint32_t K, n_cb, cb_base, e_max; }; // a block size's code blocks: slots cb_base .. cb_base + n_cb of the batch's code-block order; e_max: its longest allocation's soft bits
struct MiMultiGeom { // launch geometry derived from the groups; classes = workgroup widths 64 (c + 1)
    uint64_t arr_bytes = 0;                           // bytes of one scratch array over all sizes
    uint32_t n_slots = 0, n_wv1 = 0, n_wv23 = 0;
    uint32_t grid_cb[6] = {0}, grid_perm[6] = {0}, lds_prep[6] = {0}, kp_max[6] = {0};
    uint32_t map_cb[6] = {0}, map_perm[6] = {0}, map_wv1 = 0, map_wv23 = 0; // where each launch's map starts (entries)
    uint32_t ord_wv1 = 0, ord_wv23 = 0, n_ord1 = 0, n_ord23 = 0;             // the trellis kernel's launch order: launched wavefront -> wavefront of the map (entries; 0 = none)
    uint32_t siso_pad1 = 0, siso_pad23 = 0;                                  // dynamic LDS the trellis kernel is launched with: it holds nothing, it limits the resident workgroups per compute unit
    int      one_size[6] = {-1, -1, -1, -1, -1, -1};   // the index of a width's ONLY size (its prep launch then takes the per-size kernel), -1 otherwise
    uint64_t off_one[6] = {0}; uint32_t e_cap_one[6] = {0};
    uint32_t map_ws1 = 0, map_ws23 = 0, n_ws1 = 0, n_ws23 = 0, gpw1 = 1, gpw23 = 1, kp_all = 0; // the state-parallel trellis kernel's launches (a handful of blocks)
    size_t   map_off = 0;                             // bytes from the table's start to the maps
};
struct MiMultiCache { void *d_tab = nullptr; size_t cap = 0; std::vector<MiKGroup> built_for; MiMultiGeom geom; };
bool  mi_turbo_ref_multi_takes(uint32_t K, uint32_t e_max_bytes);
int   mi_turbo_ref_multi(mi_lte_ctx *ctx, const MiKGroup *groups, uint32_t n_groups, const mi_lte_pdsch_alloc *d_allocs, const uint32_t *d_cb_alloc,
                         const int8_t *d_e, const uint32_t *d_e_off, const uint32_t *d_e_len, uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status,
                         bool ul, bool packed, MiMultiCache *cache);
void  mi_multi_cache_free(MiMultiCache *cache);
int   mi_turbo_bcjr_group(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, const mi_lte_pdsch_alloc *d_allocs, const uint32_t *d_cb_alloc, const int8_t *d_e,
                          const uint32_t *d_e_off, const uint32_t *d_e_len, uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status, bool ul,
                          int8_t *d_soft, uint8_t *d_c_bits, uint32_t n_iter, int qpp_spec, bool packed = false, uint32_t e_max_bytes = 0, bool block_mode = false, bool early = false);
int   mi_ctx_turbo_tables(mi_lte_ctx *ctx, uint32_t K, int spec, TurboTables *out);
struct mi_lte_pdsch_plan;
int   mi_pdsch_plan_create_mapped(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t max_alloc, size_t max_soft_bytes, mi_lte_pdsch_plan **out); // chain.hip
int   mi_pdsch_plan_assign_slice(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_src, uint32_t n_alloc, uint32_t unit0,
                                 std::vector<uint32_t> *refused, hipStream_t copy_stream = nullptr); // chain.hip: the host pipeline's one-pass assignment
void  mi_pdsch_plan_wide_stride(mi_lte_pdsch_plan *pl); // one output stride whatever the plan holds: that of the largest single-code-block transport block (pipeline.cc)
constexpr uint32_t MI_PUCCH_STAGED_MAX = 32; // (352 floats of tables each: inside the 64 KB mapped block)
struct MiPucchStaged { char *h_base = nullptr, *d_base = nullptr; size_t o_tab = 0, o_out = 0; uint32_t n_res = 0; };
int   mi_pucch_stage(mi_lte_ctx *ctx, uint32_t N_rb_ul, const mi_lte_pucch_res *h_res, const float *h_tables, uint32_t n_res, MiPucchStaged *st); // uplink.hip
int   mi_pucch_launch(mi_lte_ctx *ctx, const MiPucchStaged *st, uint32_t N_rb_ul, const float *d_subframes);
void  mi_pucch_collect(const MiPucchStaged *st, uint8_t *h_bits, uint32_t *h_n_bits, uint32_t *h_rc);
struct mi_lte_pusch_plan;
int   mi_pusch_plan_create_impl(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, const uint32_t *h_unit_subfr_num,
                                const uint32_t *h_unit_n_id_cell, uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                                const float *h_dmrs, mi_lte_pusch_plan **out);
int   mi_fft_rows(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, const uint64_t *d_win_start,
                  uint32_t n_rows, float *d_rows);
int   mi_turbo_bcjr_batch(mi_lte_ctx *ctx, const int8_t *d_soft, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits, bool early = false);
struct MiBcjrBufs { int8_t *S1, *P1, *S2, *P2, *tail; void *aux; }; // what a prep kernel fills (layouts: bcjr.hip); aux: 32 bytes per code block of its own
int   mi_turbo_bcjr_begin(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, MiBcjrBufs *out);
int   mi_turbo_bcjr_iterate(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits, bool early = false);
int   mi_turbo_bcjr_block_batch(mi_lte_ctx *ctx, const int8_t *d_soft, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits);
