/*
 * mi_lte.h -- C-ABI of the MI355X-native LTE receive chains (libmi_lte.so): the downlink hot path (front end, PDSCH,
 * turbo), and -- widened one row at a time -- the eNodeB uplink (SC-FDMA front end, PUSCH, PRACH), the downlink control
 * region (PCFICH, PDCCH, PBCH) and initial synchronisation (coarse timing, PSS, SSS).
 *
 * Drop-in boundary for the hot path of mgp25/OpenLTE's liblte_phy (reference paths are relative to
 * the reference root).  The reference has no plugin/FFI mechanism: its apps link liblte statically
 * and call C++-mangled liblte_phy_* functions over caller-visible structs (liblte/hdr/liblte_phy.h).
 * The binding a maintainer adds is therefore a small C++ translation unit that includes the
 * reference's own liblte_phy.h, defines the liblte_phy_* symbols of this path and forwards to the
 * entry points below (shim/liblte_phy_shim.cc in this repo).
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types.  Pointers named d_* are DEVICE pointers
 *    (HIP memory on the context's GPU, from mi_lte_malloc or any other HIP allocator);
 *    pointers named h_* are HOST pointers.
 *  - every batch function returns MI_LTE_OK (0) or a negative mi_lte_status; decode verdicts
 *    (LIBLTE_ERROR_ENUM values, liblte/hdr/liblte_common.h:59-65) are reported per block in
 *    output arrays, never through the return code.  The per-call *_host forms at the end mirror the
 *    reference's own functions instead: negative on infrastructure errors, else the LIBLTE_ERROR_ENUM value.
 *  - one mi_lte_ctx per GPU and per host thread (the reference's LIBLTE_PHY_STRUCT is likewise not
 *    re-entrant: all its scratch lives in the struct).  All work is issued on the context's stream.
 *  - there is NO CPU fallback: if no gfx950 device is usable every entry point fails loudly.
 */
#ifndef MI_LTE_H
#define MI_LTE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_LTE_VERSION 100

typedef enum {
    MI_LTE_OK               = 0,
    MI_LTE_ERR_INVALID_ARG  = -1, /* NULL / out-of-range argument (reference: LIBLTE_ERROR_INVALID_INPUTS) */
    MI_LTE_ERR_NO_DEVICE    = -2, /* no usable gfx950 GPU: the product path has no CPU fallback          */
    MI_LTE_ERR_HIP          = -3, /* a HIP runtime call failed; see mi_lte_last_error                    */
    MI_LTE_ERR_UNSUPPORTED  = -4, /* outside the envelope (e.g. multi-code-block transport blocks)       */
    MI_LTE_ERR_NOMEM        = -5
} mi_lte_status;

/* per-block verdicts, numerically identical to LIBLTE_ERROR_ENUM (liblte_common.h:59-65) */
enum { MI_LTE_DECODE_SUCCESS = 0, MI_LTE_DECODE_INVALID_INPUTS = 1, MI_LTE_DECODE_FAIL = 2, MI_LTE_DECODE_INVALID_CRC = 3,
       MI_LTE_DECODE_INVALID_CONTENTS = 4 };

typedef struct mi_lte_ctx mi_lte_ctx; /* opaque */

/* ---------------------------------------------------------------- context, memory, timing */
int         mi_lte_version(void);
/* identity of the device code this library was built from: "file.hip:hhhhhhhhHHHHHHHH;" per kernel source (sha1 of the file, sha1 of the
 * shared headers, eight hex digits each).  The committed profiler tables (profiles/pmc_traffic_*.json, sq_counters_*.json) carry the id of
 * the build they were measured on; bench.py reports their numbers only for kernels whose file has not changed since. */
const char *mi_lte_build_id(void);
int         mi_lte_device_count(void);
int         mi_lte_ctx_create(int device, mi_lte_ctx **out);
void        mi_lte_ctx_destroy(mi_lte_ctx *ctx);
const char *mi_lte_last_error(const mi_lte_ctx *ctx);
const char *mi_lte_device_name(const mi_lte_ctx *ctx);
void       *mi_lte_stream(const mi_lte_ctx *ctx); /* the hipStream_t all work is issued on */

int mi_lte_malloc(mi_lte_ctx *ctx, size_t bytes, void **d_ptr);
int mi_lte_free(mi_lte_ctx *ctx, void *d_ptr);
int mi_lte_memset(mi_lte_ctx *ctx, void *d_ptr, int value, size_t bytes);
int mi_lte_memcpy_h2d(mi_lte_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int mi_lte_memcpy_d2h(mi_lte_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int mi_lte_sync(mi_lte_ctx *ctx);

/* HIP-event stopwatch on the context's stream (what bench.py times kernels with) */
int mi_lte_timer_start(mi_lte_ctx *ctx);
int mi_lte_timer_stop(mi_lte_ctx *ctx, float *elapsed_ms); /* records, synchronises, returns ms */

/* What a streaming kernel reaches on this device ("use the measured copy bandwidth as a second denominator"): a 16-bytes-per-lane
 * copy of `bytes` (a multiple of 16) from one scratch buffer to another, `reps` launches between two events on the context's stream, in
 * three kernel shapes; *gb_per_s = the best shape's 2 * bytes * reps / time (bytes read + bytes written). */
int mi_lte_device_copy_rate(mi_lte_ctx *ctx, size_t bytes, uint32_t reps, double *gb_per_s);
/* the last call's three kernel shapes (one 16-byte access per thread; the same with the non-temporal hint; a grid-stride loop): *gb_per_s
 * was their best */
int mi_lte_device_copy_rates(const mi_lte_ctx *ctx, double *out3);

/* Per-kernel timing: when enabled every kernel launch the library issues is bracketed by a pair
 * of HIP events on the context's stream.  The report is "kernel:launches:total_ms;..." since the
 * last reset (this is how bench.py measures the dominant kernel's average duration live). */
int         mi_lte_profile_enable(mi_lte_ctx *ctx, int on);
int         mi_lte_profile_reset(mi_lte_ctx *ctx);
const char *mi_lte_profile_report(mi_lte_ctx *ctx);

/* ---------------------------------------------------------------- DL front end
 * Replaces liblte_phy_get_dl_subframe_and_ce() (liblte/hdr/liblte_phy.h:1170-1177, implementation
 * liblte/src/liblte_phy.cc:5905-6200) for a batch of independent subframe "units": 14 OFDM symbol
 * FFTs + the look-ahead symbols the interpolation needs (symbol 14 = the next subframe's first symbol;
 * with N_ant = 4 also symbol 15, which only ports 2 and 3 read -- row 15 is not produced for N_ant <= 2;
 * the window starts one sample early, liblte_phy.cc:8621) and CRS channel estimation / interpolation for N_ant ports.
 *
 * Samples: either interleaved int8 I,Q (the capture file format the reference's callers convert
 * from, LTE_fdd_dl_file_scan/src/LTE_fdd_dl_fs_samp_buf.cc:657-694) in d_samples_a, or planar fp32
 * i_samps / q_samps (the reference API's own arguments) in d_samples_a / d_samples_b.
 * d_unit_start[u] is the sample index of unit u's subframe start (the reference's
 * frame_start_idx + subfr_num*N_samps_per_subfr); a unit reads up to start + 30720 + 4400 samples at
 * 30.72 MHz (scaled for smaller FFT sizes).
 *
 * Output: one "device subframe" per unit, mi_lte_subframe_floats(N_ant) floats, laid out like the
 * receive half of LIBLTE_PHY_SUBFRAME_STRUCT (liblte_phy.h:226-239) with row stride 1200:
 *     rx_symb_re[16][1200] rx_symb_im[16][1200] rx_ce_re[N_ant][16][1200] rx_ce_im[N_ant][16][1200]
 * (channel-estimate rows 14,15 are never written, as in the reference). */
typedef enum {
    MI_LTE_IQ_I8 = 0, MI_LTE_IQ_F32_PLANAR = 1,
    MI_LTE_IQ_ALL_ROWS = 0x100, /* OR into sample_format for mi_lte_dl_frontend_batch: produce symbol row 15 for N_ant <= 2 as well, as the
                                  reference's struct holds it (the per-call host form and the shim set it) */
    MI_LTE_CE_COMPACT = 0x200  /* OR into sample_format of BOTH mi_lte_dl_frontend_batch and mi_lte_pdsch_plan_create (single-port cells): the
                                  estimator stops after the frequency direction and leaves, instead of the 14 estimate rows, the magnitude and
                                  phase rows at the five CRS symbols (rows 0-4 of rx_ce_re = magnitude, rows 0-4 of rx_ce_im = phase); the PDSCH
                                  demodulator then runs the reference's time interpolation (liblte_phy.cc:6119-6190) itself, for its own
                                  resource elements only.  Same arithmetic in the same order, so soft bits and decoded blocks are identical
                                  to the full form; 172 KB per subframe less HBM traffic.  Device subframes in this form are for the PDSCH
                                  chain only (the control-channel decoders and the host forms want the estimate rows). */
} mi_lte_iq_format;
typedef struct {
    uint32_t fft_size;      /* 128, 256, 512, 1024, 2048 = N_samps_per_symb (liblte_phy.cc:2226-2274) */
    uint32_t N_rb_dl;       /* 6..100 */
    uint32_t N_ant;         /* 1, 2, 4 */
    uint32_t sample_format; /* mi_lte_iq_format */
} mi_lte_dl_cfg;

size_t mi_lte_subframe_floats(uint32_t N_ant);
int    mi_lte_dl_frontend_batch(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a,
                                const void *d_samples_b, const uint64_t *d_unit_start, const uint32_t *d_subfr_num,
                                const uint32_t *d_n_id_cell, uint32_t n_units, float *d_subframes);

/* ---------------------------------------------------------------- PDSCH allocations
 * Compact form of LIBLTE_PHY_ALLOCATION_STRUCT (liblte/hdr/liblte_phy.h:684-702): the fields
 * liblte_phy_pdsch_channel_decode reads, plus the index of the subframe unit the allocation lives in. */
typedef struct {
    uint32_t unit;           /* index into the batch of device subframes                         */
    uint32_t mod_type;       /* LIBLTE_PHY_MODULATION_TYPE_ENUM: 0 BPSK 1 QPSK 2 16QAM 3 64QAM     */
    uint32_t tbs;            /* transport block size in bits                                     */
    uint32_t rv_idx;
    uint32_t tx_mode;
    uint32_t rnti;
    uint32_t N_prb;
    uint32_t n_pdcch_symbs;  /* control-region size of the allocation's subframe (the reference's N_pdcch_symbs argument, 1..4); 0 = the
                                plan's.  mi_lte_pdcch_decode_run fills it in the allocations it returns, so that the DCIs of a capture's
                                subframes -- each with its own CFI -- go into ONE plan */
    uint8_t  prb[2][112];    /* PRB indices per slot (alloc->prb[L/7][...]), first N_prb valid    */
} mi_lte_pdsch_alloc;

/* ---------------------------------------------------------------- PDSCH decode
 * Replaces liblte_phy_pdsch_channel_decode() (liblte/hdr/liblte_phy.h:906-913, implementation
 * liblte/src/liblte_phy.cc:3690-3853) and, under it, dlsch_channel_decode() (:12762-12872) for a
 * batch of allocations over a batch of device subframes produced by mi_lte_dl_frontend_batch:
 * RE extraction, pre-decoding (single port, or the reference's transmit-diversity combiner),
 * layer de-mapping, modulation de-mapping, descrambling, turbo rate un-matching
 * (liblte_phy_rate_unmatch_turbo, liblte_phy.h:1311-1323), REF-mode turbo decoding, filler removal
 * and the CRC24A check.  M_dl_harq = 8 and N_soft = 250368 are fixed as in the reference (:3843-3844).
 *
 * Envelope: one code block per transport block (tbs + 24 <= 6144); larger ones make plan_create
 * return MI_LTE_ERR_UNSUPPORTED (the reference's own multi-block path is broken).
 *
 * A plan holds the device copy of the allocation list and its grouping by code-block size, so a
 * repeated schedule (the benchmark, or a semi-static grant pattern) pays for planning once.
 * Outputs of run():  d_out_bits[a*out_stride + i], i < tbs: decoded transport block of allocation a,
 * one bit per byte (the reference's out_bits), meaningful when d_status[a] == 0;
 * d_status[a]: 0 = LIBLTE_SUCCESS, 2 = LIBLTE_ERROR_DECODE_FAIL (CRC mismatch). */
typedef struct mi_lte_pdsch_plan mi_lte_pdsch_plan;
int      mi_lte_pdsch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs,
                                  const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc, mi_lte_pdsch_plan **out);
void     mi_lte_pdsch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pdsch_plan *plan);
/* A plan whose allocation list changes from run to run -- a capture, where every subframe brings its own DCIs
 * (LTE_fdd_dl_file_scan/src/LTE_fdd_dl_fs_samp_buf.cc:445-515 decodes what liblte_phy_pdcch_channel_decode just found): the device arrays
 * and a pinned staging block are sized once (at most max_alloc allocations with at most max_soft_bytes of soft bits between them,
 * every allocation's share rounded up to 64 bytes), and mi_lte_pdsch_plan_assign re-plans inside them -- host grouping by code-block
 * size plus three asynchronous copies on the context's stream; no allocation, no wait.  The allocations' own n_pdcch_symbs (as
 * mi_lte_pdcch_decode_run returns them) let subframes with different control-region sizes share the plan.  The output stride of a
 * dynamic plan is that of the largest single-code-block transport block (6144 bytes, 768 packed) whatever is assigned. */
int      mi_lte_pdsch_plan_create_dynamic(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t max_alloc, size_t max_soft_bytes,
                                          mi_lte_pdsch_plan **out);
int      mi_lte_pdsch_plan_assign(mi_lte_ctx *ctx, mi_lte_pdsch_plan *plan, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_allocs,
                                  uint32_t n_alloc);
uint32_t mi_lte_pdsch_plan_n_alloc(const mi_lte_pdsch_plan *plan);
/* 1 when the plans decode this allocation, 0 when they refuse it: more than one code block (tbs + 24 > 6144, the reference's own C > 1
 * path is broken), N_prb == 0 or > N_rb_dl, a modulation past 64QAM, a control region outside 1..4 symbols, or a resource block
 * outside the carrier (an invalid RIV: a DCI whose 16-bit CRC matched on noise).  liblte_phy_pdsch_channel_decode (liblte_phy.cc:3690-3853)
 * fails such an allocation on its own -- the callers that plan many allocations at once (mi_lte_dl_pipeline_run_units / _run_capture,
 * mi_lte_dl_subframe_decode_host, shim/scan_batch.cc) use this test to report status 2 (LIBLTE_ERROR_DECODE_FAIL) at the allocation's
 * own index instead of failing the whole list; mi_lte_pdsch_plan_create / _assign themselves keep returning an error for it. */
int      mi_lte_pdsch_alloc_decodable(const mi_lte_dl_cfg *cfg, const mi_lte_pdsch_alloc *alloc, uint32_t N_pdcch_symbs);
/* Decoder of the plan's transport blocks: MI_LTE_TURBO_REF (default; the reference's decoder, bit-exact) or MI_LTE_TURBO_BCJR with
 * n_iter iterations (the max-log-MAP decoder of mi_lte_turbo_decode_batch; the reference has no such mode).  In BCJR mode the soft
 * bits are rate-un-matched to int8 channel values (sums of repeats saturated to +-127) and the decoded block is finished like
 * dlsch_channel_decode does (filler removed, CRC24A, one bit per byte).  qpp_spec != 0: the exact 3GPP interleaver (what a
 * standard eNodeB transmits); 0: the reference transmitter's uint32-wrapped one (they differ for 20 block sizes). */
int      mi_lte_pdsch_plan_set_decoder(mi_lte_pdsch_plan *plan, uint32_t mode /* mi_lte_turbo_mode */, uint32_t n_iter, int qpp_spec);
/* Output form: packed = 0 (default) one bit per byte, the reference's out_bits (liblte_common.h:53); packed != 0 eight bits per byte, the
 * first bit of the transport block in the most significant position of byte 0 (what liblte_bits_2_value would assemble, and SURVEY 8d's
 * K/8-byte accounting): 8x less to bring back over PCIe.  Changes mi_lte_pdsch_plan_out_stride. */
int      mi_lte_pdsch_plan_set_output(mi_lte_pdsch_plan *plan, uint32_t packed);
uint32_t mi_lte_pdsch_plan_out_stride(const mi_lte_pdsch_plan *plan);
int      mi_lte_pdsch_decode_run(mi_lte_ctx *ctx, mi_lte_pdsch_plan *plan, const float *d_subframes,
                                 const uint32_t *d_subfr_num, const uint32_t *d_n_id_cell, uint8_t *d_out_bits,
                                 int32_t *d_status);
/* stage tap: device pointers to the descrambled soft bits (int8) of one allocation and to their count */
int      mi_lte_pdsch_plan_soft_bits(const mi_lte_pdsch_plan *plan, uint32_t alloc, const int8_t **d_e,
                                     const uint32_t **d_len);

/* ---------------------------------------------------------------- turbo decode
 * Replaces turbo_decode() (liblte/src/liblte_phy.cc:10620-10845) for a batch of code blocks of one
 * size K.  Input layout is the reference's: per block 3*(K+4) soft values INTERLEAVED d[i*3+x]
 * (x = 0 systematic, 1 parity-1, 2 parity-2), positive = bit 0, the value 10000 (RX_NULL_BIT,
 * liblte_phy.cc:1620) marks a punctured position.  Blocks are contiguous: block b starts at element
 * b*3*(K+4).
 *
 *   MI_LTE_TURBO_REF   bit-exact restatement of the reference's Steps 0-14 (three hard-metric SISO
 *                      Viterbi passes + soft re-encodes + 4-way vote), including its uint32 QPP
 *                      wrap-around; de-interleaver holes read as 0.  n_iter is ignored.
 *   MI_LTE_TURBO_BCJR  fixed-point max-log-MAP (extrinsic scaled by 3/4), n_iter full iterations, int8 LLRs
 *   MI_LTE_TURBO_BCJR_BLOCK  the same decoder laid out for a HANDFUL of blocks (the per-call forms): one code block per wavefront, the
 *                      64 lanes walk 64 segments of the block, every array of the decode in LDS, one launch for all iterations.  Its
 *                      alpha recursion restarts (from the previous iteration's value) every 32-96 steps instead of every K/8, so its
 *                      output is specified by its own model (lo_turbo_decode_bcjr_block) and can differ from MI_LTE_TURBO_BCJR's on
 *                      blocks near the decoding threshold; ~4x lower latency for a single K = 6144 block.
 *                      (MI_LTE_SOFT_I8) only; qpp_spec != 0 selects the exact 3GPP interleaver instead of the
 *                      reference's wrapped one.  Not a behaviour of the reference (its decoder is REF): specified
 *                      by oracle/lte_oracle.c lo_turbo_decode_bcjr, which the kernels match bit for bit.
 *
 *   MI_LTE_TURBO_BCJR_EARLY  MI_LTE_TURBO_BCJR with hard-decision-aided early termination: from the second iteration on, a tile pair (128
 *                      code blocks that share a wavefront) stops once an iteration changes none of its hard decisions; n_iter is the most
 *                      iterations any pair runs.  A block's output is MI_LTE_TURBO_BCJR's with n_iter = the iterations its pair ran
 *                      (mi_lte_turbo_early_exit_iterations reports them).  A throughput mode of its own: never what the 8-iteration
 *                      numbers are quoted on.
 *
 * Output: d_c_bits, one decoded bit per byte, K bytes per block (the reference's c_bits). */
typedef enum { MI_LTE_TURBO_REF = 0, MI_LTE_TURBO_BCJR = 1, MI_LTE_TURBO_BCJR_BLOCK = 2, MI_LTE_TURBO_BCJR_EARLY = 3 } mi_lte_turbo_mode;
typedef enum { MI_LTE_SOFT_F32 = 0, MI_LTE_SOFT_I8 = 1, MI_LTE_SOFT_I16 = 2 } mi_lte_soft_type;

int mi_lte_turbo_decode_batch(mi_lte_ctx *ctx, const void *d_soft, mi_lte_soft_type soft_type, uint32_t K,
                              uint32_t n_cb, mi_lte_turbo_mode mode, uint32_t n_iter, int qpp_spec,
                              uint8_t *d_c_bits);

/* iterations every tile pair (code blocks 128p .. 128p + 127) of the context's last MI_LTE_TURBO_BCJR_EARLY decode ran: h_pair_iters[p] in
 * 2..n_iter; *n_pairs pairs (MI_LTE_ERR_INVALID_ARG with *n_pairs set when max_pairs is too small) */
int mi_lte_turbo_early_exit_iterations(mi_lte_ctx *ctx, uint32_t *h_pair_iters, uint32_t max_pairs, uint32_t *n_pairs, uint32_t *n_iter);

/* The reference-faithful decoder's trellis kernel comes in two shapes with identical results: code blocks on the lanes (lock-step tiles of
 * 64: the throughput shape, 64k blocks per launch) and, for a decode of at most n_cb_max code blocks, STATES on the lanes (four lanes per
 * trellis, the traceback as a parallel composition of state maps): a third of the latency when one transport block is all there is, which
 * is what a per-call caller (the shim) has.  Default 4096; 0 = always the lock-step kernel.  A tuning / test knob, not a semantic one. */
int mi_lte_set_turbo_small_batch(mi_lte_ctx *ctx, uint32_t n_cb_max);
/* A PDSCH / PUSCH decode whose allocations have several code-block sizes launches each of its kernels ONCE over the blocks of all sizes
 * (the per-code-block kernels once per workgroup width) instead of size by size; on = 0 keeps the per-size launches.  Identical results
 * (tests/test_mixed_gpu.py); a test / tuning knob like the one above.  Default on. */
int mi_lte_set_turbo_merged(mi_lte_ctx *ctx, uint32_t on);

/* ---------------------------------------------------------------- turbo rate un-matching
 * Replaces liblte_phy_rate_unmatch_turbo() (liblte/hdr/liblte_phy.h:1311-1323, implementation
 * liblte/src/liblte_phy.cc:11246-11490) with its float interface, for n_cb code blocks that share
 * one parameter set: e bits in (N_e_bits per block), interleaved d[i*3+x] out (3*D per block, D = the
 * reference's N_dummy_bits = K+4), RX_NULL_BIT = 10000.0f where nothing was received.  chan_type uses
 * LIBLTE_PHY_CHAN_TYPE_ENUM values (0 DLSCH, 1 PCH limit the soft buffer; 2 ULSCH, 3 ULCCH do not).
 * (Inside mi_lte_pdsch_decode_run the same gather is fused into the decoder and never touches HBM.) */
int mi_lte_rate_unmatch_turbo_batch(mi_lte_ctx *ctx, const float *d_e_bits, uint32_t N_e_bits, uint32_t D,
                                    uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq,
                                    uint32_t chan_type, uint32_t rv_idx, uint32_t n_cb, float *d_d_bits);

/* bytes of device scratch the decoder holds for (K, n_cb); grows on demand, reported for sizing */
size_t mi_lte_turbo_scratch_bytes(uint32_t K, uint32_t n_cb);
size_t mi_lte_turbo_bcjr_scratch_bytes(uint32_t K, uint32_t n_cb);

/* name and launch count of the kernels the last batch call issued (for bench.py / profiles) */
const char *mi_lte_last_kernels(const mi_lte_ctx *ctx);

/* ---------------------------------------------------------------- uplink (eNodeB receive side)
 * SURVEY 8f N1 / BASELINE config 5: the PUSCH receive chain of LTE_fdd_enodeb
 * (LTE_fdd_enodeb/src/LTE_fdd_enb_phy.cc:832-917 calls liblte_phy_get_ul_subframe and
 * liblte_phy_pusch_channel_decode once per scheduled UE).
 *
 * mi_lte_ul_frontend_batch replaces liblte_phy_get_ul_subframe() (liblte/hdr/liblte_phy.h:1190-1193,
 * implementation liblte/src/liblte_phy.cc:6209-6236 over samples_to_symbols_ul :8654-8692): 14 SC-FDMA
 * symbols per subframe unit; the reference's 2N-point FFT of the zero-padded N samples, of which it keeps
 * the odd bins (the half-sub-carrier shift), is computed as an N-point FFT of the samples rotated by
 * exp(-i*pi*n/N); the window starts one sample early (:8680) like the downlink one.
 * Output per unit: mi_lte_ul_subframe_floats() floats, rx_symb_re[16][1200] rx_symb_im[16][1200]
 * (rows 14, 15 unused), the receive half of LIBLTE_PHY_SUBFRAME_STRUCT. */
size_t mi_lte_ul_subframe_floats(void);
int    mi_lte_ul_frontend_batch(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg /* N_rb_dl = N_rb_ul, N_ant ignored */,
                                const void *d_samples_a, const void *d_samples_b, const uint64_t *d_unit_start,
                                uint32_t n_units, float *d_subframes);

/* Cell-level uplink configuration: the arguments of liblte_phy_ul_init() (liblte_phy.h:613-625) that the
 * PUSCH demodulation reference signals depend on (generate_dmrs_pusch, liblte_phy.cc:6868-6990); liblte_phy.h:613-625. */
typedef struct {
    uint32_t group_assignment_pusch;   /* delta_ss */
    uint32_t group_hopping_enabled;
    uint32_t sequence_hopping_enabled;
    uint32_t cyclic_shift;             /* n1_DMRS index, 0..7 */
    uint32_t cyclic_shift_dci;         /* n2_DMRS index, 0..7 */
} mi_lte_ul_cfg;

/* PUSCH demodulation reference signal of one (subframe, N_prb): 4 x 12*N_prb floats
 * (dmrs_0_re, dmrs_0_im for symbol 3; dmrs_1_re, dmrs_1_im for symbol 10).  Host function; restates
 * generate_dmrs_pusch / generate_ul_rs (liblte_phy.cc:6745-6990) with the reference's own arithmetic
 * (double-precision libm calls on float-rounded arguments), layer 0. */
int mi_lte_ul_dmrs_pusch(const mi_lte_ul_cfg *ul, uint32_t N_id_cell, uint32_t N_subfr, uint32_t N_prb, float *h_dmrs_0_re,
                         float *h_dmrs_0_im, float *h_dmrs_1_re, float *h_dmrs_1_im);

/* mi_lte_pusch_plan_* / mi_lte_pusch_decode_run replace liblte_phy_pusch_channel_decode()
 * (liblte_phy.h:722-728, implementation liblte_phy.cc:2801-2935) and ulsch_channel_decode (:12363-12501)
 * for a batch of allocations (one per scheduled UE) over a batch of uplink device subframes: RE extraction,
 * DMRS least-squares estimate + magnitude/phase interpolation (get_ulsch_ce :13718-13790), one-tap
 * equaliser (:6708-6736), transform pre-decoding (12 unnormalised backward DFTs of size 12*N_prb scaled by
 * sqrt(12*N_prb), :6627-6660 -- the reference's scaling, reproduced), de-mapping, descrambling, channel
 * de-interleaving (:12108-12225; with no RI/ACK/CQI bits it is a 12-column transpose), turbo rate
 * un-matching with the UL-SCH soft-buffer rule (N_cb = K_w), REF turbo decoding, CRC24A.
 * alloc.unit selects the subframe unit; the unit's subframe number and cell id are host arrays here because
 * the reference signals are generated on the host per (cell, subframe, N_prb).  N_prb must be one the
 * reference has an FFTW plan for: N_prb < N_rb_ul and divisible by 2, 3 or 5 (liblte_phy.cc:2360-2377);
 * single antenna, one code block per transport block. */
typedef struct mi_lte_pusch_plan mi_lte_pusch_plan;
int      mi_lte_pusch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul,
                                  const uint32_t *h_unit_subfr_num, const uint32_t *h_unit_n_id_cell, uint32_t n_units,
                                  const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc, mi_lte_pusch_plan **out);
void     mi_lte_pusch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pusch_plan *plan);
uint32_t mi_lte_pusch_plan_out_stride(const mi_lte_pusch_plan *plan);
int      mi_lte_pusch_decode_run(mi_lte_ctx *ctx, mi_lte_pusch_plan *plan, const float *d_subframes, uint8_t *d_out_bits,
                                 int32_t *d_status);
/* stage tap: device pointer to the de-interleaved, descrambled soft bits (int8, 12*12*N_prb*Q_m of them) */
int      mi_lte_pusch_plan_soft_bits(const mi_lte_pusch_plan *plan, uint32_t alloc, const int8_t **d_e, uint32_t *n_bits);

/* PRACH detection: replaces liblte_phy_detect_prach() (liblte_phy.h:862-868, implementation liblte_phy.cc:3299-3479)
 * for a batch of PRACH occasions (d_occ_start[o] = sample index of the occasion's first cyclic-prefix sample; an
 * occasion spans mi_lte_prach_occasion_samples() samples), preamble formats 0-4 (format 4, the TDD UpPTS preamble of
 * liblte_phy.cc:2462-2468: N_zc = 139 on 7.5 kHz sub-carriers, T_fft = 4 096, root table 5.7.2-5, N_cs table 5.7.2-3 -- the same
 * kernels with the sequence length as an argument).  The 839 (139) PRACH sub-carriers are
 * computed directly from the T_fft samples behind the prefix, correlated with every root sequence the cell's 64
 * preambles use, and the reference's verdict -- one preamble if the peak reaches 50 x the averaged correlation
 * power (:3460-3474) -- is evaluated on the host: h_N_det_pre[o] in {0,1}, h_det_pre[o] = preamble index,
 * h_det_ta[o] = timing advance, exactly the reference's three outputs.  The plan generates the root sequences'
 * spectra itself (prach_preamble_seq_gen :7130-7290 + the 839-point DFTs of liblte_phy_ul_init :2496-2508);
 * mi_lte_prach_plan_create_roots takes them from the caller instead (LIBLTE_PHY_STRUCT::prach_x_u_fft_re/im).
 * A 64-preamble set that starts near the end of the logical root table continues at index 0 (36.211 5.7.2: the order is cyclic);
 * the reference indexes past its table there (liblte_phy.cc:7168-7171), so only the caller's-roots form reproduces it in that case. */
typedef struct {
    uint32_t root_seq_idx;     /* logical root sequence index, 0..837 (format 4: 0..137) */
    uint32_t preamble_format;  /* 0..4 */
    uint32_t zczc;             /* zeroCorrelationZoneConfig */
    uint32_t hs_flag;          /* restricted set */
    uint32_t freq_offset;      /* n_PRBoffset^RA */
} mi_lte_prach_cfg;
typedef struct mi_lte_prach_plan mi_lte_prach_plan;
int      mi_lte_prach_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg /* N_rb_dl = N_rb_ul */, const mi_lte_prach_cfg *prach,
                                  mi_lte_prach_plan **out);
int      mi_lte_prach_plan_create_roots(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *prach,
                                        const float *h_x_u_fft_re /*[n_roots][839], the first N_zc of a row used*/, const float *h_x_u_fft_im,
                                        uint32_t n_roots, mi_lte_prach_plan **out);
void     mi_lte_prach_plan_destroy(mi_lte_ctx *ctx, mi_lte_prach_plan *plan);
uint32_t mi_lte_prach_plan_n_roots(const mi_lte_prach_plan *plan);
/* Host arithmetic only (no device): the physical root sequence numbers u of the configuration's 64 preambles in the order
 * prach_preamble_seq_gen enumerates them (liblte_phy.cc:7155-7290; cyclic logical order, see above).  MI_LTE_ERR_UNSUPPORTED where the
 * reference itself cannot process the configuration (a restricted-set root without a cyclic shift, zeroCorrelationZoneConfig past the table). */
int      mi_lte_prach_root_set(const mi_lte_prach_cfg *prach, uint32_t *h_u /* [64] */, uint32_t *n_roots);
uint32_t mi_lte_prach_occasion_samples(const mi_lte_prach_plan *plan);
int      mi_lte_prach_detect_run(mi_lte_ctx *ctx, mi_lte_prach_plan *plan, const void *d_samples_a, const void *d_samples_b,
                                 const uint64_t *d_occ_start, uint32_t n_occ, uint32_t *h_N_det_pre, uint32_t *h_det_pre,
                                 uint32_t *h_det_ta);
/* The same detection (liblte_phy_detect_prach, liblte/hdr/liblte_phy.h:727-733, liblte_phy.cc:3293-3480) in two halves for a caller that keeps
 * the stream busy -- a receiver that hands the device the next subframes while the random-access verdicts of these are on their way:
 * _launch queues the kernels and the copy of the per-root maxima (pinned memory of the plan's own) and returns at once; _fetch waits for
 * the stream and forms the verdicts of the launch before it (max_occ = room in the three arrays).  One launch in flight per plan. */
int      mi_lte_prach_detect_launch(mi_lte_ctx *ctx, mi_lte_prach_plan *plan, const void *d_samples_a, const void *d_samples_b,
                                    const uint64_t *d_occ_start, uint32_t n_occ);
int      mi_lte_prach_detect_fetch(mi_lte_ctx *ctx, mi_lte_prach_plan *plan, uint32_t *h_N_det_pre, uint32_t *h_det_pre,
                                   uint32_t *h_det_ta, uint32_t max_occ);

/* ---------------------------------------------------------------- PUCCH formats 1 / 1a / 1b
 * mi_lte_pucch_decode_run replaces liblte_phy_pucch_format_1_1a_1b_channel_decode() (liblte/hdr/liblte_phy.h:775-782,
 * implementation liblte/src/liblte_phy.cc:2961-3146 with get_ulcch_ce :13799-13868) for a batch of PUCCH resources over UL device
 * subframes (mi_lte_ul_frontend_batch) -- the fourth receive call of the eNodeB's radio thread (LTE_fdd_enb_phy.cc:867).
 * The sequences are the caller's, i.e. what liblte_phy_ul_init left in LIBLTE_PHY_STRUCT for (subframe N, resource n = N_1_p_pucch);
 * per resource h_tables holds MI_LTE_PUCCH_TAB_FLOATS floats:
 *     pucch_dmrs_0_re[N][n][36] pucch_dmrs_0_im[..][36] pucch_dmrs_1_re[..][36] pucch_dmrs_1_im[..][36]
 *     r_re[2][4][12] r_im[2][4][12]   = pucch_r_u_v_alpha_p_re/_im[N][n][m'][symbol 0, 1, 5, 6][12]
 *     sw_re[2][4]    sw_im[2][4]      = s_ns(m') * W_5_4_1_2[pucch_n_oc_p[N][n][m']][i]   (s_ns = 1 or (cos(pi/2), sin(pi/2)), :3058-3068)
 * Outputs per resource: h_bits[2r], h_bits[2r+1], h_n_bits (1 for formats 1 / 1a, 2 for 1b), h_rc = 0 (LIBLTE_SUCCESS) or 1
 * (LIBLTE_ERROR_INVALID_INPUTS: soft decision <= 0.5, the reference's failure value).  N_ant = 1 like the reference. */
#define MI_LTE_PUCCH_TAB_FLOATS 352
typedef struct {
    uint32_t unit;         /* index into the batch of UL device subframes */
    uint32_t format;       /* 0 = format 1, 1 = 1a, 2 = 1b (LIBLTE_PHY_PUCCH_FORMAT_ENUM) */
    uint32_t N_1_p_pucch;
} mi_lte_pucch_res;
/* The tables of one (subframe, resource) from the cell's configuration, for callers that have no LIBLTE_PHY_STRUCT: restates what
 * liblte_phy_ul_init computes through generate_dmrs_pucch (liblte_phy.cc:2401-2421, :6986-7129) and what the decoder derives from it
 * (:3058-3083).  delta_pucch_shift = deltaPUCCH-Shift (1..3 in 36.211; i.e. the value liblte_phy_ul_init is given + 1, :2414); host function. */
int mi_lte_ul_pucch_tables(const mi_lte_ul_cfg *ul, uint32_t N_id_cell, uint32_t N_subfr, uint32_t N_1_p_pucch, uint32_t N_cs_an,
                           uint32_t delta_pucch_shift, uint32_t N_ant, float *h_tables /* [MI_LTE_PUCCH_TAB_FLOATS] */);
int mi_lte_pucch_decode_run(mi_lte_ctx *ctx, uint32_t N_rb_ul, uint32_t N_ant, const float *d_subframes, const mi_lte_pucch_res *h_res,
                            const float *h_tables, uint32_t n_res, uint8_t *h_bits, uint32_t *h_n_bits, uint32_t *h_rc);

/* ---------------------------------------------------------------- PCFICH + PDCCH (common search space)
 * mi_lte_pdcch_plan_* / mi_lte_pdcch_decode_run replace liblte_phy_pdcch_channel_decode()
 * (liblte/hdr/liblte_phy.h:1012-1020, implementation liblte/src/liblte_phy.cc:4519-5135) for a batch of device
 * subframes -- the call that sits between liblte_phy_get_dl_subframe_and_ce and liblte_phy_pdsch_channel_decode in the
 * reference's receiver (LTE_fdd_dl_fs_samp_buf.cc:445-470).  Per subframe: the control-format indicator from the
 * PCFICH, N_symbs = cfi (+1 for N_rb_dl <= 10), and the DCIs found for SI-, P- and RA-RNTI in the reference's six
 * common-search-space candidates (four at aggregation level 4, two at level 8; formats 1A and 1C each), in the
 * reference's order and capped at LIBLTE_PHY_PDCCH_MAX_ALLOC = 6.  Each DCI comes raw (payload bits) and unpacked into
 * the allocation liblte_phy_pdsch_channel_decode needs (dci_1a_unpack :13273-13378, dci_1c_unpack :13400-13611);
 * a DCI that does not unpack is dropped like the reference does.  h_rc[u] is the reference's return value for the
 * subframe: 3 (LIBLTE_ERROR_INVALID_CRC, h_cfi[u] = 0) when the PCFICH did not decode, else what its last DCI unpacker
 * call returned (0 or 4), or 1 (LIBLTE_ERROR_INVALID_INPUTS, its initial value) when no DCI was found.
 * Candidates reaching past the subframe's last CCE are decoded with the missing CCEs as erasures -- the reference reads
 * stale scratch there, zeros on a fresh LIBLTE_PHY_STRUCT.
 *
 * The plan holds the resource-element index tables of the listed cells (a cell the plan was not built for decodes
 * as cfi = 0).  PHICH: only the positions are needed (its REGs are excluded from the PDCCH), normal duration only --
 * the reference does not handle the extended duration either (:8280-8283).
 *
 * Transmit diversity: the reference passes its per-port estimate array [4][288] to the combiner with a port stride of
 * 576 (:4881, :7935), so with two ports the second port's estimate is read as zero and with four ports the rows are
 * scrambled and nothing decodes.  By default the plan reproduces exactly that arithmetic (parity with the reference,
 * rows it never writes taken as the zeros of a fresh LIBLTE_PHY_STRUCT); MI_LTE_PDCCH_PER_PORT_ESTIMATES gives every
 * port its own estimate instead -- the decoder the reference meant, which does decode 4-port cells. */
#define MI_LTE_PDCCH_MAX_DCI 6
#define MI_LTE_PDCCH_PER_PORT_ESTIMATES 1u
typedef struct {
    uint32_t rnti;         /* 0xFFFF SI, 0xFFFE P, 0x0001..0x003C RA                               */
    uint32_t format;       /* 0 = DCI 1A, 1 = DCI 1C                                              */
    uint32_t candidate;    /* 0..3 aggregation level 4 (CCE 4*c), 4..5 aggregation level 8        */
    uint32_t n_bits;       /* DCI size for this bandwidth (:4835-4857)                            */
    uint32_t payload;      /* the DCI, first bit in bit n_bits-1                                  */
    uint32_t mcs;
    uint32_t alloc_valid;  /* the DCI unpacked into alloc                                         */
    uint32_t reserved;
    mi_lte_pdsch_alloc alloc; /* unit, mod_type, tbs, rv_idx, tx_mode, rnti, N_prb, prb[2][N_prb]   */
} mi_lte_pdcch_dci;
typedef struct mi_lte_pdcch_plan mi_lte_pdcch_plan;
int  mi_lte_pdcch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, float phich_res /* N_g: 1/6, 1/2, 1, 2 */,
                              uint32_t phich_dur_extended, uint32_t flags, const uint32_t *h_cells, uint32_t n_cells,
                              mi_lte_pdcch_plan **out);
void mi_lte_pdcch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pdcch_plan *plan);
int  mi_lte_pdcch_decode_run(mi_lte_ctx *ctx, mi_lte_pdcch_plan *plan, const float *d_subframes, const uint32_t *d_subfr_num,
                             const uint32_t *d_n_id_cell, uint32_t n_units, uint32_t *h_rc, uint32_t *h_cfi, uint32_t *h_n_symbs,
                             uint32_t *h_n_dci, mi_lte_pdcch_dci *h_dci /* [n_units][MI_LTE_PDCCH_MAX_DCI] */);
/* the plan's index tables on their own (host arithmetic): grid index l*1200 + k of the 16 PCFICH resource elements and of
 * the six candidates' resource elements in decoding order (0xFFFFFFFF: CCE past the subframe's last one) */
int  mi_lte_pdcch_re_tables(uint32_t N_rb_dl, uint32_t N_ant, uint32_t N_id_cell, float phich_res, uint32_t N_symbs,
                            uint32_t *pcfich /* [16] */, uint32_t *cand /* [6][288] */);
/* what the reference leaves in LIBLTE_PHY_PCFICH_STRUCT::k, n and LIBLTE_PHY_PHICH_STRUCT::N_reg, k (:7903-7910, :8243-8278) */
int  mi_lte_ctrl_reg_positions(uint32_t N_rb_dl, uint32_t N_id_cell, float phich_res, uint32_t *pcfich_k /* [4] */,
                               float *pcfich_n /* [4] */, uint32_t *phich_N_reg, uint32_t *phich_k /* [75] */);
/* atan2f as the de-mappers evaluate it where the reference DECIDES on an angle (QPSK / BPSK quadrants, PUCCH 1b regions): the algorithm of
 * the reference host's libm (glibc 2.35) restated in float arithmetic, here on the host (tests pin it to libm bit for bit) */
int  mi_lte_model_atan2f(const float *h_y, const float *h_x, float *h_out, size_t n);
/* the two DCI unpackers on their own (host arithmetic; what the shim and tests compare with the reference's) */
int  mi_lte_dci_1a_unpack(uint32_t payload, uint32_t n_bits, uint32_t rnti, uint32_t N_rb_dl, uint32_t N_ant, mi_lte_pdcch_dci *out);
int  mi_lte_dci_1c_unpack(uint32_t payload, uint32_t n_bits, uint32_t rnti, uint32_t N_rb_dl, uint32_t N_ant, mi_lte_pdcch_dci *out);

/* ---------------------------------------------------------------- PBCH
 * mi_lte_pbch_decode_run replaces liblte_phy_bch_channel_decode() (liblte/hdr/liblte_phy.h:947-953, implementation
 * liblte/src/liblte_phy.cc:3968-4105 with bch_channel_decode :12581-12650) for a batch of device subframes holding subframe 0
 * of a frame with channel estimates for all four ports (cfg->N_ant = 4: the reference tries 1, 2 and 4 ports against the
 * same struct, LTE_fdd_dl_fs_samp_buf.cc:395-410).  Twelve hypotheses per subframe -- {1, 2, 4} ports x {0..3} position
 * in the 40 ms BCH period -- in the reference's order, first success wins: h_N_ant[u] (0 = LIBLTE_ERROR_DECODE_FAIL),
 * h_offset[u] (the reference's *offset: SFN mod 4), h_mib[u] = the 24 BCH bits, first bit in bit 23. */
int mi_lte_pbch_decode_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const float *d_subframes, const uint32_t *d_n_id_cell,
                           uint32_t n_units, uint32_t *h_N_ant, uint32_t *h_offset, uint32_t *h_mib);

/* ---------------------------------------------------------------- initial synchronisation
 * The three searches that precede the first subframe (LTE_fdd_dl_fs_samp_buf.cc:277-395), over samples resident in HBM
 * (format as for mi_lte_dl_frontend_batch; `start` = index of the capture's first sample in the buffer):
 *   mi_lte_coarse_timing_run  replaces liblte_phy_dl_find_coarse_timing_and_freq_offset (liblte_phy.h:1134-1138, impl.
 *       liblte_phy.cc:5697-5852); reads mi_lte_coarse_timing_samples(fft_size, N_slots) samples.  mi_lte_coarse_timing is
 *       LIBLTE_PHY_COARSE_TIMING_STRUCT (liblte_phy.h:1128-1132).
 *   mi_lte_find_pss_run       replaces liblte_phy_find_pss_and_fine_timing (liblte_phy.h:1068-1075, impl. :5306-5510);
 *       symb_starts[7] is read and rewritten like the reference's; reads up to symb_starts[6] + 12 slots + one symbol + 40.
 *   mi_lte_find_sss_run       replaces liblte_phy_find_sss (liblte_phy.h:1106-1113, impl. :5578-5687); *found = 0 is the
 *       reference's LIBLTE_ERROR_INVALID_INPUTS return (no SSS above 0.9 x pss_thresh).
 * The device computes every correlation sum in the reference's summation order; the decisions on them are evaluated on the
 * host with the reference's expressions.  Coarse timing is bit-identical for integer-valued samples (all sums are exact);
 * the PSS / SSS stages sit behind an FFT and agree to its tolerance (decisions identical unless two candidates tie). */
typedef struct {
    float    freq_offset[5];
    uint32_t symb_starts[5][7];
    uint32_t n_corr_peaks;
} mi_lte_coarse_timing;
size_t mi_lte_coarse_timing_samples(uint32_t fft_size, uint32_t N_slots);
int    mi_lte_coarse_timing_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b,
                                uint64_t start, uint32_t N_slots, mi_lte_coarse_timing *out);
int    mi_lte_find_pss_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, uint64_t start,
                           uint32_t *symb_starts /* [7] in/out */, uint32_t *N_id_2, uint32_t *pss_symb, float *pss_thresh,
                           float *freq_offset);
int    mi_lte_find_sss_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, uint64_t start,
                           uint32_t N_id_2, uint32_t *symb_starts /* [7] in/out */, float pss_thresh, uint32_t *N_id_1,
                           uint32_t *frame_start_idx, uint32_t *found);

/* The scanner's caller-side steps between the searches, for a capture that stays in HBM:
 *   mi_lte_iq_i8_to_planar  int8 I,Q pairs -> planar fp32 i_samps / q_samps (LTE_fdd_dl_fs_samp_buf.cc:657-694);
 *   mi_lte_freq_shift_run   LTE_fdd_dl_fs_samp_buf::freq_shift (:696-713) in place: sample i (buffer index first_index + k for
 *       k < n_samples, the reference's loop variable) is rotated by exp(-j*(i+1)*freq_offset*2*pi/fs), the argument evaluated in
 *       the mixed float / double precision of the reference's expression; cosf / sinf are the device library's (<= 2 ulp), so the
 *       result agrees with the host's to float rounding, like the FFT behind it. */
int mi_lte_iq_i8_to_planar(mi_lte_ctx *ctx, const int8_t *d_iq, uint64_t n_samples, float *d_i_samps, float *d_q_samps);
/* the scanner's other input format, gr_complex = interleaved float32 pairs (LTE_fdd_dl_fs_samp_buf.cc:686-692) -> planar fp32 */
int mi_lte_iq_f32_pairs_to_planar(mi_lte_ctx *ctx, const float *d_iq, uint64_t n_samples, float *d_i_samps, float *d_q_samps);
int mi_lte_freq_shift_run(mi_lte_ctx *ctx, float *d_i_samps, float *d_q_samps, uint64_t first_index, uint64_t n_samples, float freq_offset,
                          uint32_t fs);

/* ---------------------------------------------------------------- per-call host-pointer forms
 * The bodies of the reference's three entry points on this path, for callers that hold host
 * buffers exactly as the reference's callers do (LTE_fdd_dl_fs_samp_buf.cc:378-515,
 * LTE_fdd_enb_phy.cc).  Each stages its arguments through HBM, runs the batch kernels with a batch
 * of one, and copies the results back; shim/liblte_phy_shim.cc forwards the liblte_phy_* symbols
 * here.  Return value: MI_LTE_* on infrastructure errors (< 0), else the LIBLTE_ERROR_ENUM value the
 * reference would return (0 success, 1 invalid inputs, 2 decode fail). */
/* Nothing in these forms allocates per call: a context keeps pinned staging, device buffers and ONE device subframe tagged with the host
 * arrays it mirrors (rx_symb_re's address + a 64-bit hash of the whole contents, rows 0-13 of every plane), plus plans cached by
 * everything they were built from.  The decoders called on a subframe this context produced with mi_lte_get_dl_subframe_and_ce_host /
 * _get_ul_subframe_host -- or uploaded for an earlier call -- read the copy that is already in HBM; a subframe whose arrays changed in
 * any element hashes differently and is uploaded again (mi_lte_host_cache_invalidate forces that).  One call at a time per context, like the reference's
 * "one call at a time per LIBLTE_PHY_STRUCT" (SURVEY 8b). */
int mi_lte_host_cache_stats(mi_lte_ctx *ctx, uint64_t *n_subframe_reuse, uint64_t *n_subframe_upload, uint32_t *n_plans);
int mi_lte_host_cache_invalidate(mi_lte_ctx *ctx);
/* The explicit contract instead of the hash: in MI_LTE_HOST_CACHE_EXPLICIT mode the copy in HBM counts as current for as long as the
 * decoders are handed the same arrays (rx_symb_re's address, same port count) -- a caller that changes a subframe's contents between
 * decodes says so with mi_lte_host_cache_invalidate.  What LTE_fdd_dl_fs_samp_buf.cc does (it never touches a subframe between
 * get_dl_subframe_and_ce and the decodes) qualifies; the default, MI_LTE_HOST_CACHE_FINGERPRINT, is safe for any caller. */
enum { MI_LTE_HOST_CACHE_FINGERPRINT = 0, MI_LTE_HOST_CACHE_EXPLICIT = 1 };
int mi_lte_host_cache_set_mode(mi_lte_ctx *ctx, uint32_t mode);
/* One subframe in one call: the work of liblte_phy_get_dl_subframe_and_ce + liblte_phy_pdcch_channel_decode +
 * liblte_phy_pdsch_channel_decode for every DCI found (the loop at LTE_fdd_dl_fs_samp_buf.cc:445-515), arguments as those calls take
 * them.  The subframe stays in HBM (no LIBLTE_PHY_SUBFRAME_STRUCT comes back), the host waits twice.  Returns what
 * liblte_phy_pdcch_channel_decode would (0, 1, 2) or MI_LTE_* < 0; on 0, dci[k] (k < *N_dci) is decoded into
 * h_out_bits + k * out_stride (out_stride >= 6120) with status[k] = 0 and N_out_bits[k] = tbs, or status[k] = 2 and nothing
 * written (CRC mismatch, or a transport block outside the single-code-block envelope). */
int mi_lte_dl_subframe_decode_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i_samps, const float *h_q_samps,
                                   uint32_t frame_start_idx, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float phich_res,
                                   uint32_t phich_dur_extended, uint32_t flags, uint32_t *cfi, uint32_t *N_symbs, uint32_t *N_dci,
                                   mi_lte_pdcch_dci *dci /* [MI_LTE_PDCCH_MAX_DCI] */, uint8_t *h_out_bits, uint32_t out_stride,
                                   uint32_t *N_out_bits /* [MI_LTE_PDCCH_MAX_DCI] */, int32_t *status /* [MI_LTE_PDCCH_MAX_DCI] */);
/* One UPLINK subframe in one call: the work of liblte_phy_get_ul_subframe (liblte_phy.h:1190-1193, liblte_phy.cc:6209-6236) +
 * liblte_phy_pucch_format_1_1a_1b_channel_decode per resource (liblte_phy.h:775-782, liblte_phy.cc:2961-3146) +
 * liblte_phy_pusch_channel_decode per scheduled UE (liblte_phy.h:722-728, liblte_phy.cc:2801-2935) -- the eNodeB radio thread's receive
 * half of a TTI (LTE_fdd_enb_phy.cc:832-917) -- as one launch chain with one wait; the received grid stays in HBM.  h_i / h_q: the
 * subframe's 30720 / (2048 / fft_size) samples; ul: what liblte_phy_ul_init was given (the reference signals are generated from it);
 * allocs[k] (k < n_alloc <= MI_LTE_UL_SUBFRAME_MAX_ALLOC) is decoded into h_out_bits + k * out_stride (out_stride >= 6120) with
 * status[k] = 0 and N_out_bits[k] = tbs, or status[k] = 1 (the reference's failure value on this path: CRC mismatch, or an allocation it
 * has no transform plan for) and nothing written; pucch[r] (unit 0) with h_pucch_tables as for mi_lte_pucch_decode_run.  Returns 0,
 * 1 (invalid arguments, the reference's value) or MI_LTE_* < 0. */
#define MI_LTE_UL_SUBFRAME_MAX_ALLOC 16
int mi_lte_ul_subframe_decode_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const float *h_i_samps, const float *h_q_samps,
                                   uint32_t subfr_num, uint32_t N_id_cell, const mi_lte_ul_cfg *ul, const mi_lte_pdsch_alloc *allocs,
                                   uint32_t n_alloc, uint8_t *h_out_bits, uint32_t out_stride, uint32_t *N_out_bits, int32_t *status,
                                   const mi_lte_pucch_res *pucch, const float *h_pucch_tables, uint32_t n_pucch, uint8_t *h_pucch_bits /*[n_pucch][2]*/,
                                   uint32_t *h_pucch_n_bits, uint32_t *h_pucch_rc);
int mi_lte_get_dl_subframe_and_ce_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i_samps,
                                       const float *h_q_samps, uint32_t frame_start_idx, uint32_t subfr_num,
                                       uint32_t N_id_cell, uint32_t N_ant, float *h_rx_symb_re /*[16][1200]*/,
                                       float *h_rx_symb_im, float *h_rx_ce_re /*[4][16][1200]*/, float *h_rx_ce_im);
int mi_lte_pdsch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_rx_symb_re, const float *h_rx_symb_im,
                                     const float *h_rx_ce_re, const float *h_rx_ce_im, uint32_t subfr_num,
                                     const mi_lte_pdsch_alloc *alloc, uint32_t N_pdcch_symbs, uint32_t N_id_cell,
                                     uint32_t N_ant, uint8_t *h_out_bits, uint32_t *N_out_bits);
/* liblte_phy_pdcch_channel_decode: returns the reference's value (0; 1 no DCI found; 3 PCFICH failed; 4 last DCI did
 * not unpack) with *cfi, *N_symbs, *N_dci and dci[] as mi_lte_pdcch_decode_run writes them for one subframe */
int mi_lte_pdcch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_rx_symb_re, const float *h_rx_symb_im,
                                     const float *h_rx_ce_re, const float *h_rx_ce_im, uint32_t subfr_num, uint32_t N_id_cell,
                                     uint32_t N_ant, float phich_res, uint32_t phich_dur_extended, uint32_t flags, uint32_t *cfi,
                                     uint32_t *N_symbs, uint32_t *N_dci, mi_lte_pdcch_dci *dci /* [MI_LTE_PDCCH_MAX_DCI] */);
/* liblte_phy_bch_channel_decode: 0 with *N_ant, h_out_bits[24], *N_out_bits = 24 and *offset written, or 2
 * (LIBLTE_ERROR_DECODE_FAIL) with *N_ant = 0 and the rest untouched; h_rx_ce_* hold all four ports */
int mi_lte_bch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_rx_symb_re, const float *h_rx_symb_im,
                                   const float *h_rx_ce_re /*[4][16][1200]*/, const float *h_rx_ce_im, uint32_t N_id_cell, uint8_t *N_ant,
                                   uint8_t *h_out_bits, uint32_t *N_out_bits, uint8_t *offset);
/* liblte_phy_pucch_format_1_1a_1b_channel_decode: 0 or 1 (the reference's failure value) with the bit(s) and their count written
 * in both cases, as the reference does; h_tables as for mi_lte_pucch_decode_run */
int mi_lte_pucch_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_rx_symb_re, const float *h_rx_symb_im, uint32_t format,
                             uint32_t N_ant, uint32_t N_1_p_pucch, const float *h_tables, uint8_t *h_out_bits, uint32_t *N_out_bits);
/* the three synchronisation searches on host sample arrays (they read exactly as far as the reference does); find_sss
 * returns 1 (LIBLTE_ERROR_INVALID_INPUTS, the reference's value) when no SSS clears the threshold */
int mi_lte_dl_find_coarse_timing_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i_samps, const float *h_q_samps,
                                      uint32_t N_slots, mi_lte_coarse_timing *out);
int mi_lte_find_pss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i_samps, const float *h_q_samps,
                         uint32_t *symb_starts, uint32_t *N_id_2, uint32_t *pss_symb, float *pss_thresh, float *freq_offset);
int mi_lte_find_sss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i_samps, const float *h_q_samps, uint32_t N_id_2,
                         uint32_t *symb_starts, float pss_thresh, uint32_t *N_id_1, uint32_t *frame_start_idx);
/* uplink: liblte_phy_get_ul_subframe (h_i / h_q point at the subframe's first sample; 14 rows of 1200 floats are
 * written) and liblte_phy_pusch_channel_decode (the DMRS arrays are the caller's, i.e. what liblte_phy_ul_init
 * stored in LIBLTE_PHY_STRUCT::pusch_dmrs_{0,1}_{re,im}[subframe][N_prb]; a CRC failure returns 1 =
 * LIBLTE_ERROR_INVALID_INPUTS like the reference, liblte_phy.cc:2809, :2929) */
int mi_lte_get_ul_subframe_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const float *h_i_samps, const float *h_q_samps,
                                float *h_rx_symb_re /*[16][1200]*/, float *h_rx_symb_im);
int mi_lte_pusch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_rx_symb_re, const float *h_rx_symb_im,
                                     uint32_t subfr_num, const mi_lte_pdsch_alloc *alloc, uint32_t N_id_cell, uint32_t N_ant,
                                     const float *h_dmrs_0_re, const float *h_dmrs_0_im, const float *h_dmrs_1_re,
                                     const float *h_dmrs_1_im, uint8_t *h_out_bits, uint32_t *N_out_bits);
/* liblte_phy_detect_prach: h_re / h_im point at the occasion's first cyclic-prefix sample; root spectra from the caller's struct
 * (LIBLTE_PHY_STRUCT::prach_x_u_fft_*, what liblte_phy_ul_init left there), or both pointers NULL: the library generates the cell's
 * root set itself (mi_lte_prach_plan_create; n_roots is ignored) */
int mi_lte_detect_prach_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const mi_lte_prach_cfg *prach,
                             const float *h_x_u_fft_re /*[n_roots][839]*/, const float *h_x_u_fft_im, uint32_t n_roots, const float *h_re,
                             const float *h_im, uint32_t *N_det_pre, uint32_t *det_pre, uint32_t *det_ta);
int mi_lte_rate_unmatch_turbo_host(mi_lte_ctx *ctx, const float *h_e_bits, uint32_t N_e_bits, uint32_t N_dummy_bits,
                                   uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq,
                                   uint32_t chan_type, uint32_t rv_idx, float *h_d_bits, uint32_t *N_d_bits);

/* ---------------------------------------------------------------- scheduler-side helpers ("CPU pass-through restatements")
 * Pure host functions of a few small integers that LTE_fdd_enodeb calls every TTI next to the receive chains; no context, no device.
 * Each returns what the reference's function returns (a LIBLTE_ERROR_ENUM value where that one does) and leaves its outputs untouched
 * exactly where the reference does (a search that finds nothing).  sched.cc; compared with the compiled reference over their whole
 * argument ranges by shim/lifecycle_check.
 *   mi_lte_tbs                              36.213 table 7.1.7.2.1-1, I_TBS 0..26, N_PRB 1..110 (TBS_71721, liblte_phy.cc:477-746); 0 outside
 *   mi_lte_get_tbs_mcs_and_n_prb_for_dl     liblte_phy_get_tbs_mcs_and_n_prb_for_dl   liblte_phy.h:1210   liblte_phy.cc:6251-6357
 *   mi_lte_get_tbs_and_n_prb_for_dl         liblte_phy_get_tbs_and_n_prb_for_dl       liblte_phy.h:1234   liblte_phy.cc:6359-6407
 *   mi_lte_get_tbs_mcs_and_n_prb_for_ul     liblte_phy_get_tbs_mcs_and_n_prb_for_ul   liblte_phy.h:1253   liblte_phy.cc:6409-6475
 *   mi_lte_get_n_cce                        liblte_phy_get_n_cce                      liblte_phy.h:1271   liblte_phy.cc:6477-6505 (the struct's
 *                                           N_rb_dl and N_group_phich as arguments; phich_res is not read by the reference either)
 *   mi_lte_pucch_map_sr_config_idx          liblte_phy_pucch_map_sr_config_idx        liblte_phy.h:830    liblte_phy.cc:3183-3217
 *   mi_lte_code_block_segmentation          liblte_phy_code_block_segmentation        liblte_phy.h:1337   liblte_phy.cc:9753-9865
 *   mi_lte_code_block_desegmentation        liblte_phy_code_block_desegmentation      liblte_phy.h:1357   liblte_phy.cc:9875-9987 */
uint32_t mi_lte_tbs(uint32_t I_tbs, uint32_t N_prb);
int      mi_lte_get_tbs_mcs_and_n_prb_for_dl(uint32_t N_bits, uint32_t N_subframe, uint32_t N_rb_dl, uint16_t rnti, uint32_t *tbs, uint8_t *mcs,
                                             uint32_t *N_prb);
int      mi_lte_get_tbs_and_n_prb_for_dl(uint32_t N_bits, uint32_t N_rb_dl, uint8_t mcs, uint32_t *tbs, uint32_t *N_prb);
int      mi_lte_get_tbs_mcs_and_n_prb_for_ul(uint32_t N_bits, uint32_t N_rb_ul, uint32_t *tbs, uint8_t *mcs, uint32_t *N_prb);
uint32_t mi_lte_get_n_cce(uint32_t N_rb_dl, uint32_t N_group_phich, uint32_t N_pdcch_symbs, uint32_t N_ant);
void     mi_lte_pucch_map_sr_config_idx(uint32_t i_sr, uint32_t *sr_periodicity, uint32_t *N_offset_sr);
void     mi_lte_code_block_segmentation(const uint8_t *b_bits, uint32_t N_b_bits, uint32_t *N_codeblocks, uint32_t *N_filler_bits, uint8_t *c_bits,
                                        uint32_t N_c_bits_max, uint32_t *N_c_bits);
void     mi_lte_code_block_desegmentation(const uint8_t *c_bits, const uint32_t *N_c_bits, uint32_t N_c_bits_max, uint32_t tbs, uint8_t *b_bits,
                                          uint32_t N_b_bits);

/* ---------------------------------------------------------------- whole-chain batches from host buffers
 * Captures live in host memory, so the batch form for callers that hold host buffers: int8 I,Q in, transport blocks back packed eight
 * bits per byte ([allocation][mi_lte_dl_pipeline_out_stride()], first bit most significant) with one verdict per allocation.  The batch
 * is cut into chunks of chunk_units subframes that flow H2D -> front end -> PDSCH chain -> D2H on n_lanes independent lanes per device
 * (a context, a stream, plans and device buffers each), so that one lane's copies run under another lane's kernels; nothing is allocated
 * per run.  Host arrays should come from mi_lte_host_alloc (pinned, for every device of the process): with pageable memory the copies are
 * staged by the driver and do not overlap.  cfg->sample_format: MI_LTE_IQ_I8, optionally | MI_LTE_CE_COMPACT.  The link is the limit by
 * design: 70 KB of samples per 20 MHz subframe.
 *
 * Several devices: mi_lte_dl_pipeline_create_multi takes the device ordinals; chunk c of a run goes to device c mod n_devices,
 * every device is driven by its own host thread for the duration of the run, results land at the allocation's own index -- no collective,
 * no peer access.  The same ordinal may be listed more than once (the tests do: G "devices" on one GPU must reproduce the single-device
 * result bit for bit).
 *
 * Three shapes of input:
 *   mi_lte_dl_pipeline_run          units (each mi_lte_dl_pipeline_unit_samples() pairs: one subframe + the two look-ahead symbols, as for
 *                                   mi_lte_dl_frontend_batch with d_unit_start[u] = u * that) that all carry the allocation template the
 *                                   pipeline was created with (a semi-static grant pattern; the template's `unit` field is ignored);
 *                                   results at [unit * n_alloc_per_unit + a].
 *   mi_lte_dl_pipeline_run_units    the same units with PER-UNIT allocation lists -- what a capture yields once its PDCCHs are decoded
 *                                   (LTE_fdd_dl_fs_samp_buf.cc:445-515): h_allocs sorted by unit, h_first[u] .. h_first[u+1] the slice of
 *                                   unit u (n_units + 1 entries, at most n_alloc_per_unit per unit), each allocation's own n_pdcch_symbs or,
 *                                   where that is 0, N_pdcch_symbs; results at the allocation's index in h_allocs.
 *   mi_lte_dl_pipeline_run_capture  one CONTIGUOUS capture: n_subframes subframes from sample first_subframe_start on, subframe numbers
 *                                   first_subfr_num, +1, ... mod 10, one cell.  The capture is split on subframe boundaries and every chunk is
 *                                   copied with the 4 400 look-ahead samples (at 30.72 MHz) behind its last subframe -- the halo -- so that
 *                                   the split is invisible: results equal the unsplit run's. */
typedef struct mi_lte_dl_pipeline mi_lte_dl_pipeline;
void       *mi_lte_host_alloc(size_t bytes);
void        mi_lte_host_free(void *p);
/* Host placement ("each GPU owns a host thread"; eight devices at ~45 GB/s each read ~350 GB/s of host memory, which one
 * NUMA node does not serve).  mi_lte_device_numa_node: the node of the device's PCIe slot (/sys/bus/pci/devices/<bdf>/numa_node), -1
 * where the system does not say.  mi_lte_host_alloc_on: pinned memory for every device of the process like mi_lte_host_alloc, with its
 * pages bound to that node (anonymous mapping + mbind + hipHostRegister); falls back to mi_lte_host_alloc's placement where the node is
 * unknown or binding is not permitted -- mi_lte_host_alloc_node tells which happened (the node the block was bound to, -1 if none).
 * device < 0: a block that every device reads a share of (one contiguous capture for mi_lte_dl_pipeline_run_capture): its pages are
 * interleaved over all memory nodes (mi_lte_host_alloc_node: -2).
 * Release with mi_lte_host_free.  A multi-device pipeline pins every device's host thread to the CPUs of the device's node for the
 * duration of a run (the caller's own thread, which drives a single-device pipeline, is left alone). */
int         mi_lte_device_numa_node(int device);
void       *mi_lte_host_alloc_on(int device, size_t bytes);
int         mi_lte_host_alloc_node(const void *p);
int         mi_lte_dl_pipeline_create(int device, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_unit_allocs,
                                      uint32_t n_alloc_per_unit, uint32_t chunk_units, uint32_t n_lanes, mi_lte_dl_pipeline **out);
/* h_unit_allocs may be NULL (per-unit lists only): then n_alloc_per_unit is the most allocations a unit may carry and
 * max_soft_bytes_per_unit the soft bits of a unit's allocations, averaged over a chunk (0: one full-band 64QAM allocation's worth) */
int         mi_lte_dl_pipeline_create_multi(const int *devices, uint32_t n_devices, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs,
                                            const mi_lte_pdsch_alloc *h_unit_allocs, uint32_t n_alloc_per_unit, size_t max_soft_bytes_per_unit,
                                            uint32_t chunk_units, uint32_t n_lanes, mi_lte_dl_pipeline **out);
void        mi_lte_dl_pipeline_destroy(mi_lte_dl_pipeline *p);
uint32_t    mi_lte_dl_pipeline_out_stride(const mi_lte_dl_pipeline *p);
size_t      mi_lte_dl_pipeline_unit_samples(const mi_lte_dl_pipeline *p);
uint32_t    mi_lte_dl_pipeline_n_devices(const mi_lte_dl_pipeline *p);
const char *mi_lte_dl_pipeline_last_error(const mi_lte_dl_pipeline *p);
/* What device slot `index` (0 .. n_devices-1) did in the pipeline's last run, so that a scaling run explains its own bottleneck: chunks
 * and units it took, bytes over its link each way, the time its stream spent in each phase summed over its chunks (HIP events on the
 * lanes' streams; phases of different lanes overlap, so the sums may exceed wall_s), the wall time of its host thread, and where that
 * thread ran (numa_node / n_cpus of the affinity mask it was given; -1 / 0: not pinned). */
typedef struct {
    uint32_t device, chunks, units, n_cpus;
    int32_t  numa_node, reserved;
    uint64_t h2d_bytes, d2h_bytes;
    double   wall_s, h2d_s, kernel_s, d2h_s;
} mi_lte_pipeline_dev_stats;
int         mi_lte_dl_pipeline_device_stats(const mi_lte_dl_pipeline *p, uint32_t index, mi_lte_pipeline_dev_stats *out);
int         mi_lte_dl_pipeline_run(mi_lte_dl_pipeline *p, const int8_t *h_iq, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell, uint32_t n_units,
                                   uint8_t *h_out_packed, int32_t *h_status);
int         mi_lte_dl_pipeline_run_units(mi_lte_dl_pipeline *p, const int8_t *h_iq, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell,
                                         uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, const uint32_t *h_first, uint32_t N_pdcch_symbs,
                                         uint8_t *h_out_packed, int32_t *h_status);
int         mi_lte_dl_pipeline_run_capture(mi_lte_dl_pipeline *p, const int8_t *h_capture, uint64_t n_samples, uint64_t first_subframe_start,
                                           uint32_t n_subframes, uint32_t first_subfr_num, uint32_t N_id_cell, const mi_lte_pdsch_alloc *h_allocs,
                                           const uint32_t *h_first, uint32_t N_pdcch_symbs, uint8_t *h_out_packed, int32_t *h_status);

/* ---------------------------------------------------------------- the transmit side (host code, no device work)
 * The ten transmit functions of liblte_phy that LTE_fdd_enodeb's PHY and LTE_fdd_dl_file_gen call (2: "CPU pass-through") restated on
 * the host so that a link against this library and the shim needs no object of the reference's PHY.  Bit outputs are the reference's bit for
 * bit; float outputs come from the same float operations in the same order; the transforms (the reference: FFTW3f) run in float64 rounded to
 * float.  Return values are LIBLTE_ERROR_ENUM's: 0 success, 1 invalid inputs -- also for what the reference has no defined behaviour for
 * (three ports, a pre-coder type it leaves without output, more than 10 000 coded bits per allocation: its own arrays end there).
 * shim/lifecycle_check.cc (`tx`) compares every one with the compiled reference.
 *
 * Grids are LIBLTE_PHY_SUBFRAME_STRUCT::tx_symb_re / tx_symb_im (liblte_phy.h:234-235): float [4 ports][16 symbols][1200 sub-carriers].
 *
 *   mi_lte_rate_match_turbo       liblte_phy_rate_match_turbo       liblte_phy.h:1288  liblte_phy.cc:11081-11237 (d planar, as the reference takes it)
 *   mi_lte_pdsch_channel_encode   liblte_phy_pdsch_channel_encode   liblte_phy.h:888   liblte_phy.cc:3489-3688
 *   mi_lte_bch_channel_encode     liblte_phy_bch_channel_encode     liblte_phy.h:927   liblte_phy.cc:3863-3966
 *   mi_lte_map_crs / _pss / _sss  liblte_phy_map_crs / _pss / _sss  liblte_phy.h:1034 / 1051 / 1089   liblte_phy.cc:5144 / 5265 / 5520
 *   mi_lte_create_dl_subframe     liblte_phy_create_dl_subframe     liblte_phy.h:1152  liblte_phy.cc:5862-5903
 * The handle is the scratch of LIBLTE_PHY_STRUCT that outlives a call (the PBCH's 40 ms block; what a later call can read of an earlier one). */
#define MI_LTE_TX_GRID_SC 1200u
#define MI_LTE_TX_GRID_AT(p, L, k) (((size_t)(p) * 16u + (L)) * MI_LTE_TX_GRID_SC + (k))
#define MI_LTE_TX_GRID_FLOATS (4u * 16u * MI_LTE_TX_GRID_SC)
enum { MI_LTE_MOD_BPSK = 0, MI_LTE_MOD_QPSK = 1, MI_LTE_MOD_16QAM = 2, MI_LTE_MOD_64QAM = 3 };        /* LIBLTE_PHY_MODULATION_TYPE_ENUM */
enum { MI_LTE_CHAN_DLSCH = 0, MI_LTE_CHAN_PCH = 1, MI_LTE_CHAN_ULSCH = 2, MI_LTE_CHAN_ULCCH = 3 };     /* LIBLTE_PHY_CHAN_TYPE_ENUM       */
enum { MI_LTE_PRECODER_TX_DIVERSITY = 0, MI_LTE_PRECODER_SPATIAL_MULTIPLEXING = 1 };                   /* LIBLTE_PHY_PRE_CODER_TYPE_ENUM  */
/* LIBLTE_PHY_ALLOCATION_STRUCT (liblte_phy.h:684-702) as the transmit functions read it */
typedef struct {
    const uint8_t *msg[2];      /* transport-block bits of codeword 0 / 1, one per byte                 */
    uint32_t       msg_bits[2]; /* how many of them are given; the block is zero-padded to tbs          */
    uint32_t       pre_coder_type, mod_type, chan_type, tbs, rv_idx, N_prb;
    uint32_t       prb[2][110]; /* per slot                                                             */
    uint32_t       N_codewords, N_layers, tx_mode, rnti;
    uint32_t       mcs, tpc, ndi, dl_alloc; /* what the DCI of the allocation carries (PDCCH encode)    */
} mi_lte_tx_alloc;
typedef struct mi_lte_tx mi_lte_tx;
int  mi_lte_tx_create(mi_lte_tx **out);
void mi_lte_tx_destroy(mi_lte_tx *tx);
void mi_lte_rate_match_turbo(const uint8_t *d_bits, uint32_t N_d_bits, uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq,
                             uint32_t chan_type, uint32_t rv_idx, uint32_t N_e_bits, uint8_t *e_bits);
int  mi_lte_pdsch_channel_encode(mi_lte_tx *tx, uint32_t N_rb_dl, uint32_t N_sc_rb_dl, const mi_lte_tx_alloc *allocs, uint32_t N_alloc, uint32_t N_pdcch_symbs,
                                 uint32_t N_id_cell, uint32_t N_ant, uint32_t subfr_num, float *tx_re, float *tx_im);
int  mi_lte_bch_channel_encode(mi_lte_tx *tx, uint32_t N_rb_dl, uint32_t N_sc_rb_dl, const uint8_t *in_bits, uint32_t N_in_bits, uint32_t N_id_cell, uint32_t N_ant,
                               uint32_t sfn, float *tx_re, float *tx_im);
int  mi_lte_map_crs(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float *tx_re, float *tx_im);
int  mi_lte_map_pss(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t N_id_2, uint32_t N_ant, float *tx_re, float *tx_im);
int  mi_lte_map_sss(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t subfr_num, uint32_t N_id_1, uint32_t N_id_2, uint32_t N_ant, float *tx_re, float *tx_im);
/* the control region (tx_ctrl.cc): liblte_phy_pdcch_channel_encode (liblte_phy.h:988, liblte_phy.cc:4113-4517) -- PCFICH, PHICH and one DCI per
 * allocation (format 1A for downlink, format 0 for uplink grants) at aggregation level 4 in the common search space.  mi_lte_pcfich / mi_lte_phich
 * have the layout of LIBLTE_PHY_PCFICH_STRUCT / LIBLTE_PHY_PHICH_STRUCT (liblte_phy.h:972-986); like the reference the call writes the REG
 * positions into them, the region's size into *N_pdcch_symbs and each downlink allocation's transport block size into allocs[].tbs. */
typedef struct { float n[4]; uint32_t k[4]; uint32_t cfi; uint32_t N_reg; } mi_lte_pcfich;
typedef struct { float z_re[3], z_im[3]; uint32_t k[75]; uint32_t N_reg; uint8_t b[25][8]; uint8_t present[25][8]; } mi_lte_phich;
int mi_lte_pdcch_channel_encode(mi_lte_tx *tx, uint32_t N_rb_dl, uint32_t N_rb_ul, uint32_t N_sc_rb_dl, uint32_t N_group_phich, uint32_t N_sf_phich,
                                mi_lte_pcfich *pcfich, mi_lte_phich *phich, mi_lte_tx_alloc *allocs, uint32_t N_alloc, uint32_t *N_pdcch_symbs, uint32_t N_id_cell,
                                uint32_t N_ant, uint32_t phich_dur, uint32_t subfr_num, float *tx_re, float *tx_im);
/* the uplink pair (tx_ul.cc): liblte_phy_pusch_channel_encode (liblte_phy.h:704, liblte_phy.cc:2664-2799; one port, one layer, one codeword, an N_prb the
 * reference has a transform-precoder plan for) and liblte_phy_generate_prach (liblte_phy.h:845, liblte_phy.cc:3219-3297; T_cp + T_seq samples).  ul /
 * ul_cell are what liblte_phy_ul_init was given (the reference signal of symbols 3 and 10 comes from mi_lte_ul_dmrs_pusch). */
typedef struct mi_lte_tx_ul mi_lte_tx_ul;
int    mi_lte_tx_ul_create(mi_lte_tx_ul **out);
void   mi_lte_tx_ul_destroy(mi_lte_tx_ul *tx);
int    mi_lte_pusch_channel_encode(mi_lte_tx_ul *tx, const mi_lte_ul_cfg *ul, uint32_t ul_cell, uint32_t N_rb_ul, uint32_t N_sc_rb_ul, const mi_lte_tx_alloc *alloc,
                                   uint32_t N_id_cell, uint32_t N_ant, uint32_t subfr_num, float *tx_re, float *tx_im);
size_t mi_lte_generate_prach_len(uint32_t fft_size, uint32_t preamble_format);
int    mi_lte_generate_prach(const mi_lte_prach_cfg *prach, uint32_t fft_size, uint32_t N_rb_ul, uint32_t N_sc_rb_ul, uint32_t preamble_idx, uint32_t freq_offset,
                             float *samps_re, float *samps_im);
int  mi_lte_create_dl_subframe(uint32_t N_samps_per_symb, uint32_t N_used_sc, uint32_t N_samps_cp_l_0, uint32_t N_samps_cp_l_else, const float *tx_re,
                               const float *tx_im, uint32_t ant, float *i_samps, float *q_samps);

/* ---------------------------------------------------------------- input synthesis (host side)
 * A minimal LTE downlink transmitter for benchmark / test captures, the role LTE_fdd_dl_file_gen
 * plays for the reference (LTE_fdd_dl_file_gen/src/LTE_fdd_dl_fg_samp_buf.cc:269-668).  Host code
 * only; these functions never touch the GPU and nothing the library decodes runs on the CPU. */
int mi_lte_synth_turbo_soft_i8(uint32_t K, uint32_t n, double flip, int amp, uint64_t seed, int ref_wrap,
                               int8_t *h_soft, uint8_t *h_tx_bits);
int mi_lte_synth_turbo_soft_f32(uint32_t K, uint32_t n, double sigma, uint64_t seed, int ref_wrap, float *h_soft,
                                uint8_t *h_tx_bits);

typedef struct {
    double   gain_min, gain_max; /* |h| drawn uniformly per unit                                  */
    double   max_delay;          /* integer timing offset drawn from 0..max_delay samples; < 0: a STATIC channel (no random phase, no
                                    delay, one int8 scale for all units): consecutive units laid end to end are one capture */
    double   snr_db;             /* AWGN relative to the mean signal power; >= 200 disables noise   */
    double   peak;               /* int8 full-scale target for the signal peak (e.g. 100)           */
    uint64_t seed;
} mi_lte_synth_channel;

/* n_units single-port (N_ant = 1) subframes, each with its own cell id / subframe number and
 * n_alloc PDSCH allocations (allocs[u*n_alloc + a], .unit ignored), CFI = N_pdcch_symbs, random
 * transport blocks.  Writes unit_len = 30720*s + 4400*s (s = fft_size/2048, rounded up to a
 * multiple of 16) complex int8 samples per unit to h_iq, and the transport-block bits (one per byte,
 * tbs_stride bytes per allocation) to h_tx_bits. */
size_t mi_lte_synth_unit_len(uint32_t fft_size);
int    mi_lte_synth_dl_units_i8(const mi_lte_dl_cfg *cfg, uint32_t n_units, const uint32_t *h_subfr_num,
                                const uint32_t *h_n_id_cell, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_allocs,
                                uint32_t n_alloc, const mi_lte_synth_channel *chan, int8_t *h_iq, uint8_t *h_tx_bits,
                                uint32_t tbs_stride);

/* control regions: PCFICH + n_dci format-1A DCIs (rnti = 0: slot unused) at aggregation level 4 in candidates 0..n_dci-1,
 * standard transmit diversity on cfg->N_ant ports, through a smooth random channel per port (gain_min..gain_max) and
 * AWGN (snr_db), written directly as device-subframe grids (mi_lte_subframe_floats(N_ant) floats per unit, symbols 0-3
 * filled) with noisy channel estimates.  Input of mi_lte_pdcch_decode_run for the benchmark and the tests. */
typedef struct {
    uint32_t rnti, mcs, N_prb, rb_start, rv_idx;
} mi_lte_synth_dci;
int mi_lte_synth_ctrl_grids(const mi_lte_dl_cfg *cfg, float phich_res, uint32_t n_units, const uint32_t *h_subfr_num,
                            const uint32_t *h_n_id_cell, const uint32_t *h_cfi, const mi_lte_synth_dci *h_dci, uint32_t n_dci,
                            const mi_lte_synth_channel *chan, float *h_grids);

/* n_units uplink subframes for tests / benchmarks: every unit carries n_alloc PUSCH transmissions
 * (allocs[u*n_alloc + a], .unit ignored) with random transport blocks, through one flat channel per unit.
 * The transmit chain is the inverse of the reference's RECEIVER (36.212 5.2.2 / 36.211 5.3-5.6); the
 * reference's own uplink transmit helpers are not usable as a model (its channel interleaver and SC-FDMA
 * modulator index wrongly, liblte_phy.cc:12023-12042, :8565-8570 -- the eNodeB never transmits uplink).
 * The transform precoder scales by 1/sqrt(12*N_prb) as a real UE does; the reference's receiver then sees
 * 12*N_prb times the constellation point (its transform pre-decoding scaling, liblte_phy.cc:6644-6657), so QPSK
 * decodes with every soft bit at +-1 while 16/64QAM fails -- in the reference and, identically, here.
 * Writes mi_lte_synth_ul_unit_len(fft_size) complex int8 samples per unit. */
size_t mi_lte_synth_ul_unit_len(uint32_t fft_size);
int    mi_lte_synth_ul_units_i8(const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, uint32_t n_units,
                                const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell, const mi_lte_pdsch_alloc *h_allocs,
                                uint32_t n_alloc, const mi_lte_synth_channel *chan, int8_t *h_iq, uint8_t *h_tx_bits,
                                uint32_t tbs_stride);

/* n_occ PRACH occasions (format 0-3 preambles per 36.211 5.7.2-5.7.3): preamble h_preamble_idx[o] of the cell's 64,
 * delayed by h_delay[o] samples, through a flat channel + AWGN; mi_lte_synth_prach_len() complex int8 samples each. */
size_t mi_lte_synth_prach_len(uint32_t fft_size, uint32_t preamble_format);
int    mi_lte_synth_prach_i8(const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *prach, uint32_t n_occ, const uint32_t *h_preamble_idx,
                             const uint32_t *h_delay, const mi_lte_synth_channel *chan, int8_t *h_iq);

#ifdef __cplusplus
}
#endif
#endif
