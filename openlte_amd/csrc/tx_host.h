// Shared between the host-side transmit files (tx.cc, tx_ctrl.cc, tx_ul.cc): the bit-level and symbol-level steps more than one channel uses.
#pragma once
#include <cstdint>

#define MI_LTE_TX_MAX_CODE_BLOCKS 5 // the reference's DL-SCH scratch holds five (liblte_phy.h:577-579)
#define MI_LTE_ERR_ARG MI_LTE_ERR_INVALID_ARG

namespace tx {
constexpr uint8_t TX_NULL = 100; // <NULL> of 36.212 as the reference encodes it among bits (TX_NULL_BIT, liblte_phy.cc:1621) and symbols (:1024)
void crc_bits(const uint8_t *a, uint32_t n, uint32_t poly, uint32_t L, uint8_t *p);
void turbo_encode(const uint8_t *c, uint32_t K, uint8_t *d_planar);
void rate_match_turbo(const uint8_t *d, uint32_t N_d_bits, uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type,
                      uint32_t rv_idx, uint32_t N_e_bits, uint8_t *e);
void conv_encode_tb(const uint8_t *c, uint32_t n, uint8_t *d3);
void rate_match_conv(const uint8_t *d3, uint32_t N_d_bits, uint32_t N_e_bits, uint8_t *e);
void layer_map_dl(const float *d_re, const float *d_im, uint32_t M_symb, uint32_t N_ant, uint32_t N_codewords, float *x_re, float *x_im, uint32_t *M_layer);
void pre_code_dl(const float *x_re, const float *x_im, uint32_t M_layer, uint32_t N_ant, float *y_re, float *y_im, uint32_t y_len, uint32_t *M_ap);
void modulate(const uint8_t *bits, uint32_t N_bits, uint32_t mod, float *re, float *im, uint32_t *M_symb);
} // namespace tx

// The scratch of LIBLTE_PHY_STRUCT that outlives a call and that a later call can see (tx.cc's file comment), zeroed at creation.
struct mi_lte_tx {
    float   pdsch_d_re[10000 + 4], pdsch_d_im[10000 + 4], pdsch_x_re[10000 + 8], pdsch_x_im[10000 + 8];
    float   pdsch_y_re[4 * 5000], pdsch_y_im[4 * 5000];
    uint8_t dlsch_e[MI_LTE_TX_MAX_CODE_BLOCKS][18432], dlsch_c[MI_LTE_TX_MAX_CODE_BLOCKS][6176];
    // PBCH: the 1920 bits of a 40 ms period, coded in the first call of the period, and the cell's scrambling sequence
    uint32_t bch_N_bits;
    uint8_t  bch_encode_bits[1920], bch_c[1920];
    float    bch_d_re[480], bch_d_im[480], bch_x_re[480], bch_x_im[480], bch_y_re[4 * 240], bch_y_im[4 * 240];
    // Control region (tx_ctrl.cc).  The reference's pre-coder output rows are 288 symbols long (liblte_phy.h:409-410) but it is handed 576 as the row
    // length (liblte_phy.cc:4261-4269, :7842-7850, :8159-8167) while the mapping reads rows of 288: port p is WRITTEN at p * 576 and READ at p * 288
    // of one block of memory that runs on into the CCE arrays.  ctl is that block -- pdcch_y_re, pdcch_y_im, pdcch_cce_re, pdcch_cce_im in the
    // struct's order -- so that every port count leaves what the reference leaves.
    static constexpr uint32_t CTL_Y_RE = 0, CTL_Y_IM = 4 * 288, CTL_CCE_RE = 8 * 288, CTL_CCE_N = 4 * 87 * 36, CTL_CCE_IM = CTL_CCE_RE + CTL_CCE_N;
    float   ctl[8 * 288 + 2 * 4 * 87 * 36 + 2048];
    float   pdcch_reg_re[4][787][4], pdcch_reg_im[4][787][4]; // REGs behind the last whole CCE keep what an earlier call put there
    uint8_t pdcch_cce_used[87];                               // (re-set for the CCEs of the call's region only)
};
