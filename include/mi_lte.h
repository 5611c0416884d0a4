/*
 * mi_lte.h -- C-ABI of the MI355X-native LTE downlink receive chain (libmi_lte.so).
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
 *  - every function returns MI_LTE_OK (0) or a negative mi_lte_status; decode verdicts
 *    (LIBLTE_ERROR_ENUM values, liblte/hdr/liblte_common.h:59-65) are reported per block in
 *    output arrays, never through the return code.
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
enum { MI_LTE_DECODE_SUCCESS = 0, MI_LTE_DECODE_INVALID_INPUTS = 1, MI_LTE_DECODE_FAIL = 3, MI_LTE_DECODE_INVALID_CRC = 4 };

typedef struct mi_lte_ctx mi_lte_ctx; /* opaque */

/* ---------------------------------------------------------------- context, memory, timing */
int         mi_lte_version(void);
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

/* Per-kernel timing: when enabled every kernel launch the library issues is bracketed by a pair
 * of HIP events on the context's stream.  The report is "kernel:launches:total_ms;..." since the
 * last reset (this is how bench.py measures the dominant kernel's average duration live). */
int         mi_lte_profile_enable(mi_lte_ctx *ctx, int on);
int         mi_lte_profile_reset(mi_lte_ctx *ctx);
const char *mi_lte_profile_report(mi_lte_ctx *ctx);

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
 *   MI_LTE_TURBO_BCJR  max-log-MAP, n_iter full iterations; qpp_spec != 0 selects the exact 3GPP
 *                      interleaver instead of the reference's wrapped one.
 *
 * Output: d_c_bits, one decoded bit per byte, K bytes per block (the reference's c_bits). */
typedef enum { MI_LTE_TURBO_REF = 0, MI_LTE_TURBO_BCJR = 1 } mi_lte_turbo_mode;
typedef enum { MI_LTE_SOFT_F32 = 0, MI_LTE_SOFT_I8 = 1, MI_LTE_SOFT_I16 = 2 } mi_lte_soft_type;

int mi_lte_turbo_decode_batch(mi_lte_ctx *ctx, const void *d_soft, mi_lte_soft_type soft_type, uint32_t K,
                              uint32_t n_cb, mi_lte_turbo_mode mode, uint32_t n_iter, int qpp_spec,
                              uint8_t *d_c_bits);

/* bytes of device scratch the decoder holds for (K, n_cb); grows on demand, reported for sizing */
size_t mi_lte_turbo_scratch_bytes(uint32_t K, uint32_t n_cb);

/* name and launch count of the kernels the last batch call issued (for bench.py / profiles) */
const char *mi_lte_last_kernels(const mi_lte_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
