// liblte_phy_ext.h -- what the GPU-backed shim offers BEYOND the reference's API (shim/liblte_phy_shim.cc; not a reference header).
//
// liblte_phy_ul_subframe_decode: the receive half of an eNodeB TTI (LTE_fdd_enodeb/src/LTE_fdd_enb_phy.cc:832-917: liblte_phy_get_ul_subframe,
// then liblte_phy_pucch_format_1_1a_1b_channel_decode per PUCCH resource and liblte_phy_pusch_channel_decode per scheduled UE) in ONE call:
// on the GPU that is one launch chain and one wait instead of 1 + N_pucch + N_allocs of each, and the received grid stays in HBM.
// Same arguments as those calls take them; allocation k's bits go to out_bits + k * LIBLTE_MAX_MSG_SIZE with N_out_bits[k] and status[k] as
// liblte_phy_pusch_channel_decode would have returned them, PUCCH resource r's to pucch_bits + 2 r with N_pucch_bits[r] / pucch_status[r].
#ifndef LIBLTE_PHY_EXT_H
#define LIBLTE_PHY_EXT_H
#include "liblte_phy.h"

#define LIBLTE_PHY_UL_SUBFRAME_MAX_ALLOC 16
LIBLTE_ERROR_ENUM liblte_phy_ul_subframe_decode(LIBLTE_PHY_STRUCT *phy_struct, float *i_samps, float *q_samps, uint8 subfr_num, uint32 N_id_cell,
                                                LIBLTE_PHY_ALLOCATION_STRUCT *allocs, uint32 N_allocs, uint8 *out_bits, uint32 *N_out_bits,
                                                LIBLTE_ERROR_ENUM *status, LIBLTE_PHY_PUCCH_FORMAT_ENUM *pucch_format, uint32 *N_1_p_pucch,
                                                uint32 N_pucch, uint8 *pucch_bits, uint32 *N_pucch_bits, LIBLTE_ERROR_ENUM *pucch_status);
#endif
