/*
 * lte_oracle.h -- CPU restatement of the reference's DL receive chain.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C, single-threaded, written for clarity.  Every function cites the reference lines it
 * restates (paths relative to the reference root).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product library (openlte_amd/csrc) never does.
 *
 * Pinning (see oracle/README.md): the reference ships no golden vectors for this path, so the
 * restatement is pinned against the reference ITSELF, compiled in place into oracle/_ref by
 * oracle/ref/Makefile (tests/test_oracle_vs_ref.py, run wherever /root/reference exists), and
 * against fixtures generated from that build and committed under tests/golden/.
 */
#ifndef LTE_ORACLE_H
#define LTE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LO_RX_NULL 10000.0f /* RX_NULL_BIT / RX_NULL_SYMB, liblte_phy.cc:1023,1620 */
#define LO_N_SC_MAX 1200
#define LO_MAX_K 6144

enum { LO_MOD_BPSK = 0, LO_MOD_QPSK = 1, LO_MOD_16QAM = 2, LO_MOD_64QAM = 3 }; /* liblte_phy.h:212-217 */
enum { LO_CHAN_DLSCH = 0, LO_CHAN_PCH = 1, LO_CHAN_ULSCH = 2, LO_CHAN_ULCCH = 3 }; /* liblte_phy.h:219-224 */
enum { LO_SUCCESS = 0, LO_ERR_INVALID_INPUTS = 1, LO_ERR_DECODE_FAIL = 2, LO_ERR_INVALID_CRC = 3,
       LO_ERR_INVALID_CONTENTS = 4 }; /* liblte_common.h:59-65 */

/* Receive half of LIBLTE_PHY_SUBFRAME_STRUCT (liblte_phy.h:226-239). */
typedef struct {
    float    rx_symb_re[16][LO_N_SC_MAX];
    float    rx_symb_im[16][LO_N_SC_MAX];
    float    rx_ce_re[4][16][LO_N_SC_MAX];
    float    rx_ce_im[4][16][LO_N_SC_MAX];
    uint32_t num;
} lo_subframe_t;

/* The numerology fields of LIBLTE_PHY_STRUCT the path reads (liblte_phy.cc:2226-2277, 2592-2656). */
typedef struct {
    uint32_t N_samps_per_symb, N_samps_cp_l_0, N_samps_cp_l_else, N_samps_per_slot, N_samps_per_subfr;
    uint32_t N_rb_dl, N_sc_rb_dl, FFT_size, FFT_pad_size;
} lo_cfg_t;

/* Per-allocation descriptor: the fields of LIBLTE_PHY_ALLOCATION_STRUCT (liblte_phy.h:684-702)
 * that liblte_phy_pdsch_channel_decode reads. */
typedef struct {
    uint32_t mod_type, tbs, rv_idx, N_prb, tx_mode, rnti, pre_coder_type, N_codewords;
    uint32_t prb[110];
} lo_alloc_t;

int  lo_cfg_init(lo_cfg_t *cfg, uint32_t fft_size, uint32_t N_rb_dl);

/* ---- tables / sequences ---- */
int  lo_qpp_params(uint32_t K, uint32_t *f1, uint32_t *f2);
void lo_qpp_map_ref(uint32_t K, uint16_t *pi);  /* uint32-wrapping index, as the reference computes it */
void lo_qpp_map_spec(uint32_t K, uint16_t *pi); /* exact 3GPP index (64-bit arithmetic)               */
void lo_prs_c(uint32_t c_init, uint32_t len, uint8_t *c);
void lo_generate_crs(uint32_t N_s, uint32_t L, uint32_t N_id_cell, float *crs_re, float *crs_im);
void lo_crc24a(const uint8_t *bits, uint32_t n, uint8_t p[24]);

/* ---- turbo, reference-faithful ("REF") decoder ---- */
void lo_viterbi_siso(const int8_t *in, uint32_t K, int8_t *out);
void lo_fb_soft(const int8_t *x, uint32_t K, int8_t *fb);
void lo_turbo_decode_ref(const float *d_interleaved, uint32_t K, uint8_t *c_bits);
/* stage taps for debugging / stage parity: 0 q(d0) 1 A1 2 C1 3 I0 4 I1 5 B1 6 B2 7 D1 8 D2 9 C2 10 C3 */
void lo_turbo_decode_ref_taps(const float *d_interleaved, uint32_t K, uint8_t *c_bits, int8_t *taps /*[11][K]*/);

/* ---- turbo, max-log-MAP ("BCJR") decoder: restatement of OUR kernel, not of the reference ---- */
/* Fixed-point max-log-MAP, defined jointly with the kernel (openlte_amd/csrc/bcjr.hip); NOT a behaviour of the
 * reference -- parity for this mode is oracle <-> kernel bit-exactness plus decoding performance.
 * soft: 3(K+4) values interleaved d[i*3+x], positive = bit 0, clipped to +-127.  qpp_spec != 0: exact 3GPP
 * interleaver, else the reference's uint32-wrapped one (gather semantics; de-interleaving takes the last writer,
 * holes read 0). */
uint32_t lo_bcjr_n_seg(uint32_t K); /* 4, 2 or 1 independently decoded segments per block (see lte_oracle.c) */
void lo_turbo_decode_bcjr(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits);
uint32_t lo_bcjr_block_seg_len(uint32_t K); /* steps per alpha segment in the one-block-per-wavefront mode */
void lo_turbo_decode_bcjr_block(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits);

/* ---- encoder side (input synthesis for tests) ---- */
void lo_turbo_encode(const uint8_t *c_bits, uint32_t K, uint8_t *d_planar /* 3(K+4) */);

/* ---- rate (un)matching ---- */
uint32_t lo_rate_unmatch_turbo(const float *e_bits, uint32_t N_e_bits, uint32_t D, uint32_t N_codeblocks,
                               uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type,
                               uint32_t rv_idx, float *d_bits);
void lo_rate_match_turbo(const uint8_t *d_planar, uint32_t N_d_bits, uint32_t N_codeblocks, uint32_t tx_mode,
                         uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type, uint32_t rv_idx,
                         uint32_t N_e_bits, uint8_t *e_bits);

/* ---- PDSCH demodulation ---- */
uint32_t lo_modulation_demapper(const float *d_re, const float *d_im, uint32_t M_symb, uint32_t mod_type,
                                int8_t *bits);
uint32_t lo_pre_decoder_dl(const float *y_re, const float *y_im, const float *h_re, const float *h_im,
                           uint32_t h_len, uint32_t M_ap_symb, uint32_t N_ant, float *x_re, float *x_im);
uint32_t lo_layer_demapper_dl(const float *x_re, const float *x_im, uint32_t M_layer_symb, uint32_t N_ant,
                              float *d_re, float *d_im);
uint32_t lo_pdsch_extract(const lo_cfg_t *cfg, const lo_subframe_t *sf, const lo_alloc_t *alloc,
                          uint32_t N_pdcch_symbs, uint32_t N_id_cell, uint32_t N_ant, float *y_re, float *y_im,
                          float *c_re /*[N_ant][cap]*/, float *c_im, uint32_t cap);

/* ---- DL-SCH ---- */
int lo_segmentation_params(uint32_t B, uint32_t *C, uint32_t *F, uint32_t *K_plus, uint32_t *K_minus,
                           uint32_t *C_minus);
int lo_dlsch_channel_decode(const float *in_bits, uint32_t N_in_bits, uint32_t tbs, uint32_t tx_mode,
                            uint32_t rv_idx, uint32_t M_dl_harq, uint32_t N_soft, uint8_t *out_bits,
                            uint32_t *N_out_bits, uint8_t *c_bits_tap /* may be NULL; K bytes */);

/* ---- front end ---- */
void lo_samples_to_symbols_dl(const lo_cfg_t *cfg, const float *samps_re, const float *samps_im,
                              uint32_t slot_start_idx, uint32_t symbol_offset, float *symb_re, float *symb_im);
int  lo_get_dl_subframe_and_ce(const lo_cfg_t *cfg, const float *i_samps, const float *q_samps,
                               uint32_t frame_start_idx, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant,
                               lo_subframe_t *sf);
int  lo_pdsch_channel_decode(const lo_cfg_t *cfg, const lo_subframe_t *sf, const lo_alloc_t *alloc,
                             uint32_t N_pdcch_symbs, uint32_t N_id_cell, uint32_t N_ant, uint8_t *out_bits,
                             uint32_t *N_out_bits, int8_t *soft_bits_tap /* may be NULL */,
                             uint32_t *N_soft_bits_tap);

/* ---- timing helper for bench.py's cpu_baseline leg ---- */
double lo_time_turbo_decode_ref(const float *d_interleaved, uint32_t K, uint32_t n_cb, uint8_t *c_bits);

/* ---- the host libm's atan2f over arrays (tests pin mi_lte_model_atan2f to it) ---- */
void lo_libm_atan2f(const float *y, const float *x, float *out, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
