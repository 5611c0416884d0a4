// Cyclic-shift sets of one PRACH root sequence (36.211 5.7.2), computed exactly as the reference derives them in
// prach_preamble_seq_gen / liblte_phy_detect_prach (liblte/src/liblte_phy.cc:3336-3413, :7183-7252).  Host only.
#pragma once
#include <cstdint>

#include "lte_tables.h"

struct PrachSets { uint32_t N_cs, v_max, N_RA_shift, d_start; bool ok; };

// ok = false: a configuration the reference itself cannot process -- zeroCorrelationZoneConfig past the table (15 with the restricted set:
// the reference reads past PRACH_5_7_2_2_RS, liblte_phy.cc:295, :7190) or a restricted-set root without a single cyclic shift
// (N_RA_shift = 0: its C_v expression divides by it, :7258).  The callers refuse those instead of dividing by zero.

inline PrachSets prach_sets(uint32_t u, uint32_t zczc, bool hs)
{
    constexpr uint32_t N_ZC = 839;
    if (zczc > 15 || (hs && zczc > 14)) return PrachSets{0, 0, 0, 0, false};
    PrachSets s{hs ? (uint32_t)LTE_PRACH_NCS_RESTRICTED[zczc] : (uint32_t)LTE_PRACH_NCS_UNRESTRICTED[zczc], 0, 0, 0, true};
    if (hs) {
        uint32_t p;
        for (p = 1; p <= N_ZC; p++)
            if (((p * u) % N_ZC) == 1) break;
        const uint32_t d_u = (p < N_ZC / 2) ? p : N_ZC - p;
        uint32_t       N_RA_group;
        int32_t        N_neg;
        if (d_u >= s.N_cs && d_u < N_ZC / 3) {
            s.N_RA_shift = d_u / s.N_cs;
            s.d_start    = 2 * d_u + s.N_RA_shift * s.N_cs;
            N_RA_group   = N_ZC / s.d_start;
            N_neg        = (int32_t)((N_ZC - 2 * d_u - N_RA_group * s.d_start) / s.N_cs);
            if (N_neg < 0) N_neg = 0;
        } else {
            s.N_RA_shift = (N_ZC - 2 * d_u) / s.N_cs;
            s.d_start    = N_ZC - 2 * d_u + s.N_RA_shift * s.N_cs;
            N_RA_group   = d_u / s.d_start;
            N_neg        = (int32_t)((d_u - N_RA_group * s.d_start) / s.N_cs);
            if (N_neg < 0) N_neg = 0;
            if (N_neg > (int32_t)s.N_RA_shift) N_neg = (int32_t)s.N_RA_shift;
        }
        s.v_max = s.N_RA_shift * N_RA_group + (uint32_t)N_neg - 1; // (uint32 like the reference's: no shift at all wraps to "every remaining preamble")
        if (s.N_RA_shift == 0) s.ok = false;
    } else
        s.v_max = s.N_cs == 0 ? 0 : N_ZC / s.N_cs - 1;
    return s;
}
