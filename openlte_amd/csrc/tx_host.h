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
void modulate(const uint8_t *bits, uint32_t N_bits, uint32_t mod, float *re, float *im, uint32_t *M_symb);
} // namespace tx
