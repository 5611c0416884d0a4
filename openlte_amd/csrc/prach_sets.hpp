// Cyclic-shift sets of one PRACH root sequence (36.211 5.7.2), computed exactly as the reference derives them in
// prach_preamble_seq_gen / liblte_phy_detect_prach (liblte/src/liblte_phy.cc:3336-3413, :7183-7252).  Host only.
#pragma once
#include <cstdint>

#include "lte_tables.h"

struct PrachSets { uint32_t N_cs, v_max, N_RA_shift, d_start; bool ok; };

// ok = false: a configuration the reference itself cannot process -- zeroCorrelationZoneConfig past the table (15 with the restricted set:
// the reference reads past PRACH_5_7_2_2_RS, liblte_phy.cc:295, :7190) or a restricted-set root without a single cyclic shift
// (N_RA_shift = 0: its C_v expression divides by it, :7258).  The callers refuse those instead of dividing by zero.

// Preamble format 4 (the TDD UpPTS preamble): N_zc = 139, 138 root indices (36.211 table 5.7.2-5: u = 1, 138, 2, 137, ... -- the
// reference's PRACH_5_7_2_5, liblte_phy.cc:373), N_cs from table 5.7.2-3 whatever the high-speed flag says (:7189-7192, :3347-3350); the
// restricted-set arithmetic below is then run with 139 in the place of 839, as the reference does when the flag is set.
struct PrachGeom { uint32_t n_zc, T_fft_30, T_cp_30, K, phi, phases, reps, n_root_idx; }; // T_* at 30.72 MHz; phases: T_fft = phases x a power of two
inline PrachGeom prach_geom(uint32_t fmt)
{
    static const uint32_t cp_of_fmt[4] = {3168, 21024, 6240, 21024}; // liblte_phy.cc:2430-2470
    if (fmt >= 4) return PrachGeom{139, 4096, 448, 2, 2, 4, 1, 138};
    return PrachGeom{839, 24576, cp_of_fmt[fmt], 12, 7, 24, fmt >= 2 ? 2u : 1u, 838};
}
inline uint32_t prach_root(uint32_t fmt, uint32_t idx) // physical root u of logical index idx (idx < prach_geom(fmt).n_root_idx)
{
    if (fmt >= 4) return (idx & 1u) ? 139u - (idx + 1u) / 2u : idx / 2u + 1u;
    return LTE_PRACH_ROOT_ORDER[idx];
}

inline PrachSets prach_sets(uint32_t u, uint32_t zczc, bool hs, uint32_t fmt = 0)
{
    const uint32_t N_ZC = fmt >= 4 ? 139u : 839u;
    static const uint16_t ncs_fmt4[7] = {2, 4, 6, 8, 10, 12, 15}; // 36.211 table 5.7.2-3
    if (fmt >= 4 ? zczc > 6 : (zczc > 15 || (hs && zczc > 14))) return PrachSets{0, 0, 0, 0, false};
    PrachSets s{fmt >= 4 ? (uint32_t)ncs_fmt4[zczc] : hs ? (uint32_t)LTE_PRACH_NCS_RESTRICTED[zczc] : (uint32_t)LTE_PRACH_NCS_UNRESTRICTED[zczc], 0, 0, 0, true};
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

// The physical roots of a cell's 64 preambles, in the order prach_preamble_seq_gen enumerates them (liblte_phy.cc:7155-7290): root after
// root in the logical order, v_max + 1 cyclic shifts from each, until 64 are there.  The logical order is cyclic (36.211 5.7.2: index 0
// follows the last one); the reference does not wrap -- it indexes its table with root_seq_idx + n and reads whatever lies behind it
// (:7168-7171) -- so for a set that runs past the table only the first (table size - root_seq_idx) roots are the reference's.
// ok = false: a configuration the reference itself cannot process (PrachSets::ok of one of the roots), or indices out of range.
struct PrachRootSet { uint32_t n_roots; uint32_t u[64]; bool ok; };
inline PrachRootSet prach_root_set(uint32_t fmt, uint32_t root_seq_idx, uint32_t zczc, bool hs)
{
    PrachRootSet rs{0, {0}, true};
    if (fmt > 4) { rs.ok = false; return rs; }
    const PrachGeom pg = prach_geom(fmt);
    if (root_seq_idx >= pg.n_root_idx) { rs.ok = false; return rs; }
    uint32_t n_gen = 0;
    while (n_gen < 64 && rs.n_roots < 64) {
        const uint32_t  u  = prach_root(fmt, (root_seq_idx + rs.n_roots) % pg.n_root_idx);
        const PrachSets pr = prach_sets(u, zczc, hs, fmt);
        if (!pr.ok) { rs.ok = false; return rs; }
        rs.u[rs.n_roots++] = u;
        n_gen += pr.v_max + 1 ? pr.v_max + 1 : 64; // (a wrapped v_max: the reference takes every remaining preamble from this root)
    }
    return rs;
}
