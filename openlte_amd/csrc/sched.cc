// The scheduler-side helpers of liblte_phy: pure functions of a few small integers that LTE_fdd_enodeb's MAC / PHY call every TTI
// (LTE_fdd_enodeb/src/LTE_fdd_enb_phy.cc, LTE_fdd_enb_mac.cc) next to the receive chains.  Host code, no device work: they are here
// so that a link against libmi_lte.so + the shim needs no object of the reference's PHY for them (shim/liblte_phy_shim.cc,
// -DMI_LTE_SHIM_OWN_LIFECYCLE).  Each function restates the reference's arithmetic INCLUDING what it leaves untouched when a search
// finds nothing (callers see their own previous values then); shim/lifecycle_check.cc compares every one of them with the compiled
// reference over its whole argument range.
#include <cstring>

#include "../../include/mi_lte.h"
#include "lte_tables.h"

namespace {

// 36.213 table 7.1.7.2.1-1 by (I_TBS, N_PRB - 1)
inline uint32_t tbs_at(uint32_t i_tbs, uint32_t col) { return 8u * LTE_TBS_DIV8[i_tbs][col]; }

// QPSK bits one PRB in the middle of the band carries on a two-port cell with three control symbols (what the reference prices
// SI / paging / random-access-response grants with: get_num_bits_in_prb(N_subframe, 3, N_rb_dl / 2, N_rb_dl, 2, QPSK),
// liblte_phy.cc:13936-14086).  152 resource elements minus the control region's 32; the middle PRB lies wholly inside the
// PBCH / PSS / SSS window of every standard bandwidth (and of the "anything else is 100 PRB" default for 94..105), which takes
// 68 elements in subframe 0 and 24 in subframe 5.
uint32_t centre_prb_qpsk_bits(uint32_t N_subframe, uint32_t N_rb_dl)
{
    const uint32_t prb = N_rb_dl / 2;
    uint32_t       lo, hi;
    switch (N_rb_dl) {
    case 6:  lo = 0;  hi = 5;  break;
    case 15: lo = 4;  hi = 10; break;
    case 25: lo = 9;  hi = 15; break;
    case 50: lo = 22; hi = 27; break;
    case 75: lo = 34; hi = 40; break;
    default: lo = 47; hi = 52; break;
    }
    uint32_t n_re = 120;
    if (prb >= lo && prb <= hi) {
        // 15 / 25 / 75 PRB: the window's first and last PRB are only half covered; N_rb_dl / 2 is never one of those two
        if (N_subframe == 0) n_re -= 68;
        else if (N_subframe == 5) n_re -= 24;
    }
    return 2 * n_re;
}

// turbo block sizes around B' / C (36.212 5.1.2): K+ = the smallest size with C * K >= B', K- = the next smaller size (0 if none)
void k_plus_minus(uint32_t C, uint32_t B_prime, uint32_t *K_plus, uint32_t *K_minus)
{
    *K_plus = *K_minus = 0;
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (C * (uint32_t)LTE_QPP_ROWS[r].K >= B_prime) {
            *K_plus = LTE_QPP_ROWS[r].K;
            if (r > 0) *K_minus = LTE_QPP_ROWS[r - 1].K;
            break;
        }
}

struct SegPlan { uint32_t L, C, F, K_plus, K_minus, C_minus; };

// liblte_phy.cc:9779-9830 and, word for word the same, :9902-9946
SegPlan seg_plan(uint32_t B)
{
    SegPlan p{};
    uint32_t B_prime;
    if (B <= 6144) {
        p.L = 0; p.C = 1; B_prime = B;
        uint32_t km;
        k_plus_minus(1, B_prime, &p.K_plus, &km);
        p.K_minus = 0; p.C_minus = 0;
    } else {
        p.L = 24;
        // the reference rounds B / (Z - L) up in float arithmetic: ceilf((float)B / (float)6120)
        const float q = (float)B / (float)(6144 - 24);
        p.C = (uint32_t)q;
        if ((float)p.C < q) p.C++;
        B_prime = B + p.C * p.L;
        k_plus_minus(p.C, B_prime, &p.K_plus, &p.K_minus);
        // (K+ = 40 leaves K- = 0: cannot happen for C > 1, every B' / C is above 3060)
        p.C_minus = (p.C * p.K_plus - B_prime) / (p.K_plus - p.K_minus);
    }
    p.F = (p.C - p.C_minus) * p.K_plus + p.C_minus * p.K_minus - B_prime;
    return p;
}

// calc_crc with the CRC24B generator (liblte_phy.cc:9713-9743, CRC24B 0x1800063): remainder of a(x) * x^24, most significant bit first
void crc24b_bits(const uint8_t *a, uint32_t n, uint8_t *p)
{
    uint32_t rem = 0;
    for (uint32_t i = 0; i < n + 24; i++) {
        rem <<= 1;
        if (i < n) rem |= a[i];
        if (rem & 0x1000000u) rem ^= 0x1800063u;
    }
    for (uint32_t i = 0; i < 24; i++) p[i] = (uint8_t)((rem >> (23 - i)) & 1u);
}

} // namespace

extern "C" {

uint32_t mi_lte_tbs(uint32_t I_tbs, uint32_t N_prb)
{
    return (I_tbs < 27 && N_prb >= 1 && N_prb <= 110) ? tbs_at(I_tbs, N_prb - 1) : 0u;
}

int mi_lte_get_tbs_mcs_and_n_prb_for_dl(uint32_t N_bits, uint32_t N_subframe, uint32_t N_rb_dl, uint16_t rnti, uint32_t *tbs, uint8_t *mcs, uint32_t *N_prb)
{
    if (!tbs || !mcs || !N_prb) return MI_LTE_DECODE_INVALID_INPUTS;
    const bool broadcast = rnti == 0xFFFFu /* SI-RNTI */ || rnti == 0xFFFEu /* P-RNTI */ || (rnti >= 0x0001u && rnti <= 0x003Cu) /* RA-RNTI */;
    if (broadcast) {
        // DCI 1A for these RNTIs addresses the table's N_PRB = 2 / 3 columns only; the reference searches the wider one (its index 2).
        // A message larger than that column's last entry leaves *tbs / *mcs as the caller passed them, and the search below then
        // prices whatever *tbs holds -- kept, so that the two implementations agree on every input
        for (uint32_t i = 0; i < 27; i++)
            if (N_bits <= tbs_at(i, 2)) { *tbs = tbs_at(i, 2); *mcs = (uint8_t)i; break; }
        // as many PRBs as bring the code rate to 1/4, else 1/3: the first n with tbs * rate < bits(n)
        const uint32_t per_prb = centre_prb_qpsk_bits(N_subframe, N_rb_dl);
        *N_prb = 0;
        for (uint32_t rate = 4; rate > 2 && *N_prb == 0; rate--)
            for (uint32_t n = 1; n <= N_rb_dl; n++)
                if (*tbs * rate < per_prb * n) { *N_prb = n; break; }
        return *N_prb ? MI_LTE_DECODE_SUCCESS : MI_LTE_DECODE_INVALID_INPUTS;
    }
    // user traffic: the lowest I_TBS row, and in it the fewest PRBs, that hold the message; I_TBS -> MCS skips the two indices at
    // which the modulation order changes (36.213 table 7.1.7.1-1: MCS 10 and 17 repeat I_TBS 9 and 15)
    *N_prb = 0;
    const uint32_t cols = N_rb_dl < 110 ? N_rb_dl : 110;
    for (uint32_t i = 0; i < 27 && *N_prb == 0; i++)
        for (uint32_t j = 0; j < cols; j++)
            if (N_bits <= tbs_at(i, j)) {
                *tbs   = tbs_at(i, j);
                *N_prb = j + 1;
                *mcs   = (uint8_t)(i <= 9 ? i : i <= 15 ? i + 1 : i + 2);
                break;
            }
    return *N_prb ? MI_LTE_DECODE_SUCCESS : MI_LTE_DECODE_INVALID_INPUTS;
}

int mi_lte_get_tbs_and_n_prb_for_dl(uint32_t N_bits, uint32_t N_rb_dl, uint8_t mcs, uint32_t *tbs, uint32_t *N_prb)
{
    if (!tbs || !N_prb || mcs > 28) return MI_LTE_DECODE_INVALID_INPUTS;
    const uint32_t i_tbs = mcs <= 9 ? mcs : mcs <= 16 ? mcs - 1u : mcs - 2u;
    const uint32_t cols = N_rb_dl < 110 ? N_rb_dl : 110;
    for (uint32_t j = 0; j < cols; j++) // (nothing fits: the outputs stay as they were, and the verdict is still SUCCESS -- the reference's)
        if (N_bits <= tbs_at(i_tbs, j)) { *tbs = tbs_at(i_tbs, j); *N_prb = j + 1; break; }
    return MI_LTE_DECODE_SUCCESS;
}

int mi_lte_get_tbs_mcs_and_n_prb_for_ul(uint32_t N_bits, uint32_t N_rb_ul, uint32_t *tbs, uint8_t *mcs, uint32_t *N_prb)
{
    (void)N_rb_ul; // the reference searches N_PRB 1..11 whatever the bandwidth ("keeps processing reasonable", liblte_phy.cc:6428)
    if (!tbs || !mcs || !N_prb) return MI_LTE_DECODE_INVALID_INPUTS;
    for (uint32_t i = 0; i < 27; i++)
        for (uint32_t j = 0; j < 11; j++) {
            const uint32_t n = j + 1; // SC-FDMA widths have the factors 2, 3, 5 only (36.211 5.3.3); the reference's test is "a multiple of one of them"
            if (N_bits <= tbs_at(i, j) && (n % 2 == 0 || n % 3 == 0 || n % 5 == 0)) {
                *tbs   = tbs_at(i, j);
                *N_prb = n;
                *mcs   = (uint8_t)(i <= 10 ? i : i <= 19 ? i + 1 : i + 2); // 36.213 table 8.6.1-1: MCS 11 and 21 repeat I_TBS 10 and 19
                return MI_LTE_DECODE_SUCCESS;
            }
        }
    return MI_LTE_DECODE_INVALID_INPUTS;
}

uint32_t mi_lte_get_n_cce(uint32_t N_rb_dl, uint32_t N_group_phich, uint32_t N_pdcch_symbs, uint32_t N_ant)
{
    // resource-element groups of the control region (three per PRB and symbol, the CRS symbol's third one missing -- on four ports also the
    // second symbol's) minus PCFICH (4) and PHICH (3 per group), nine to a control channel element; unsigned arithmetic as in the reference
    uint32_t n_reg = N_pdcch_symbs * (N_rb_dl * 3u) - N_rb_dl - 4u - N_group_phich * 3u;
    if (N_ant == 4) n_reg -= N_rb_dl;
    return n_reg / 9u;
}

void mi_lte_pucch_map_sr_config_idx(uint32_t i_sr, uint32_t *sr_periodicity, uint32_t *N_offset_sr)
{
    if (!sr_periodicity || !N_offset_sr) return;
    // 36.213 table 10.1.5-1: index ranges of width 5, 10, 20, 40, 80 (period = width), then 2 and 1 (the reference sends everything
    // from 157 up to period 1)
    static const struct { uint32_t first, period; } row[] = {{0, 5}, {5, 10}, {15, 20}, {35, 40}, {75, 80}, {155, 2}, {157, 1}};
    int r = 6;
    while (r > 0 && i_sr < row[r].first) r--;
    *sr_periodicity = row[r].period;
    *N_offset_sr    = i_sr - row[r].first;
}

void mi_lte_code_block_segmentation(const uint8_t *b_bits, uint32_t N_b_bits, uint32_t *N_codeblocks, uint32_t *N_filler_bits, uint8_t *c_bits,
                                    uint32_t N_c_bits_max, uint32_t *N_c_bits)
{
    if (!b_bits || !N_codeblocks || !N_filler_bits || !c_bits || !N_c_bits) return;
    const SegPlan p = seg_plan(N_b_bits);
    *N_codeblocks  = p.C;
    *N_filler_bits = p.F;
    memset(c_bits, 100 /* TX_NULL_SYMB */, p.F); // filler positions at the head of the first block
    uint32_t s = 0;
    for (uint32_t r = 0; r < p.C; r++) {
        uint8_t       *c   = c_bits + (size_t)r * N_c_bits_max;
        const uint32_t K_r = r < p.C_minus ? p.K_minus : p.K_plus, first = r == 0 ? p.F : 0u;
        for (uint32_t k = first; k < K_r - p.L; k++) c[k] = b_bits[s++];
        N_c_bits[r] = K_r;
        if (p.C > 1) {
            // The reference takes the block CRC over K_r positions -- the K_r - 24 payload bits AND the 24 positions the parity is about
            // to occupy, as the caller's buffer holds them -- and reports K_r + 24 as the block's length (liblte_phy.cc:9847-9856).
            // Restated as it is: this is the multi-code-block path whose receiver side cannot decode
            uint8_t par[24];
            crc24b_bits(c, K_r, par);
            N_c_bits[r] = K_r + 24;
            memcpy(c + K_r - 24, par, 24);
        }
    }
}

void mi_lte_code_block_desegmentation(const uint8_t *c_bits, const uint32_t *N_c_bits, uint32_t N_c_bits_max, uint32_t tbs, uint8_t *b_bits, uint32_t N_b_bits)
{
    (void)N_b_bits; // (the reference does not read it either)
    if (!c_bits || !N_c_bits || !b_bits) return;
    const SegPlan p = seg_plan(tbs + 24);
    uint32_t s = 0;
    for (uint32_t r = 0; r < p.C; r++) {
        const uint8_t *c   = c_bits + (size_t)r * N_c_bits_max;
        const uint32_t K_r = r < p.C_minus ? p.K_minus : p.K_plus, first = r == 0 ? p.F : 0u;
        // (the reference checks each block's CRC24B here when C > 1 and only prints the outcome, liblte_phy.cc:9962-9976: no effect on the output)
        for (uint32_t k = first; k < K_r - p.L; k++) b_bits[s++] = c[k];
    }
}

} // extern "C"
