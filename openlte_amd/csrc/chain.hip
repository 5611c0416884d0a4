// PDSCH receive chain on gfx950: RE extraction + pre-decoding + demapping + descrambling in one kernel,
// then DL-SCH decoding (rate un-matching fused into the turbo decoder's prep kernel, CRC fused into
// its vote kernel; see turbo.hip).  Restates liblte_phy_pdsch_channel_decode
// (liblte/src/liblte_phy.cc:3690-3853) -> dlsch_channel_decode (:12762-12872) for a batch of
// allocations, inside the envelope the reference itself can decode (one code block per allocation,
// SURVEY F4).
#include <algorithm>
#include <cstring>
#include <map>
#include <type_traits>

#include "ctx.hpp"
#include "phy_dev.hpp"

namespace {

constexpr int N_SC_MAX = 1200;

struct DemodGeom { uint32_t N_rb_dl, N_ant, cfi, sf_stride; };
// PBCH / PSS / SSS window (liblte_phy.cc:3722-3742)
__device__ __forceinline__ void sync_window(uint32_t N_rb_dl, uint32_t &first_sc, uint32_t &last_sc)
{
    switch (N_rb_dl) {
    case 6:  first_sc = 0;           last_sc = 71;          break;
    case 15: first_sc = 4 * 12 + 6;  last_sc = 11 * 12 - 7; break;
    case 25: first_sc = 9 * 12 + 6;  last_sc = 16 * 12 - 7; break;
    case 50: first_sc = 22 * 12;     last_sc = 28 * 12 - 1; break;
    case 75: first_sc = 34 * 12 + 6; last_sc = 41 * 12 - 7; break;
    default: first_sc = 47 * 12;     last_sc = 53 * 12 - 1; break;
    }
}
// 12-bit mask of the sub-carriers of (symbol L, PRB prb) that carry PDSCH (liblte_phy.cc:3753-3789): per sub-carrier the
// reference first tests the CRS positions of the port count, and only where none matches the PBCH / PSS / SSS window of
// subframes 0 and 5 -- i.e. the union of two masks, both of which are closed forms of (cell, L, prb).
__device__ __forceinline__ uint32_t pdsch_mask(uint32_t N_ant, uint32_t cell, uint32_t sf, uint32_t L, uint32_t prb,
                                               uint32_t first_sc, uint32_t last_sc)
{
    const uint32_t l7 = L >= 7 ? L - 7 : L, c6 = cell % 6, c3 = cell % 3;
    uint32_t skip = 0;
    if (N_ant == 1) {
        if (l7 == 0) skip = 0x041u << c6;                     // j % 6 == cell % 6
        else if (l7 == 4) skip = 0x041u << ((c6 + 3) % 6);    // j % 6 == (cell + 3) % 6
    } else if (l7 == 0 || l7 == 4 || (N_ant == 4 && l7 == 1))
        skip = (0x249u << c3) & 0xFFFu;                       // j % 3 == cell % 3
    const bool win = (sf == 0 && L >= 7 && L <= 10) || ((sf == 0 || sf == 5) && (L == 5 || L == 6));
    const uint32_t s0 = prb * 12;
    if (win && last_sc >= s0 && first_sc <= s0 + 11) {
        const uint32_t lo = first_sc > s0 ? first_sc - s0 : 0u, hi = last_sc - s0 < 11u ? last_sc - s0 : 11u;
        skip |= ((2u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    return ~skip & 0xFFFu;
}

// The kernel is latency-bound (a chain of dependent table reads, then scattered plane reads per resource element), so it lives on
// occupancy: 8 waves per SIMD with a few spilled registers beat 4 without (2.80 -> 2.18 ms per 32k subframes); the single-port
// case is its own instantiation so that the 2/4-port combiners do not set its register count.
template <bool ONE_PORT, bool COMPACT = false>
#ifndef DEMOD_WPE
#define DEMOD_WPE 8
#endif
#ifndef DEMOD_WPE_C
#define DEMOD_WPE_C 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COMPACT ? DEMOD_WPE_C : DEMOD_WPE, 8))) void k_pdsch_demod(const float *__restrict__ subframes, DemodGeom g,
                                                     const mi_lte_pdsch_alloc *__restrict__ allocs,
                                                     const uint32_t *__restrict__ subfr_num, const uint32_t *__restrict__ n_id_cell,
                                                     GoldTables gt, int8_t *__restrict__ e_base, const uint32_t *__restrict__ e_off,
                                                     uint32_t *__restrict__ e_len, uint32_t max_pairs, uint32_t max_words,
                                                     uint32_t e_lds_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smu[]; // tab[max_pairs+1] (offset, mask | position) | cw[...] | soft bits
    __shared__ uint2 qam_lut[64]; // hard-decision mask -> six soft bits (16QAM / 64QAM)
    const uint32_t a_idx = blockIdx.x;
    const mi_lte_pdsch_alloc &al = allocs[a_idx];
    const uint32_t unit = al.unit, sf = subfr_num[unit], cell = n_id_cell[unit], N_ant = ONE_PORT ? 1u : g.N_ant, N_prb = al.N_prb;
    const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
    uint2    *tab = reinterpret_cast<uint2 *>(smu);
    uint32_t *cw  = smu + ((2 * (max_pairs + 1) + 3u) & ~3u); // cw and e_lds stay 16-byte aligned
    uint32_t first_sc, last_sc;
    sync_window(g.N_rb_dl, first_sc, last_sc);

    // ---- phases 1 and 2 side by side.  Wave 0: which REs, in the reference's loop order L -> PRB -> sub-carrier
    // (liblte_phy.cc:3744-3802) -- per (symbol, PRB) pair, for ALL 14 symbols, a 12-bit mask (0 in the control region) with the pair's
    // position L * 1200 + first sub-carrier above it and, by a scan inside the wave, the running count of REs before the pair.
    // Waves 1-3: the scrambling sequence words (c_init per :3831) for the upper bound of the bit count, which does not wait for the scan.
    // One barrier for both.
    const uint32_t cfi = al.n_pdcch_symbs ? al.n_pdcch_symbs : g.cfi; // the allocation's own control-region size, or the plan's
    const uint32_t n_pairs = 14 * N_prb, pair0 = cfi * N_prb;          // pairs [0, pair0) are the control region: empty masks
    const uint32_t c_init = ((al.rnti << 14) | (0u << 13) | (sf << 9) | cell) & 0x7FFFFFFFu; // 31 bits: the Gold basis has 31 rows
    if (threadIdx.x < 64) {
        const uint32_t ln = threadIdx.x, per = (n_pairs + 63) / 64, q0 = ln * per, q1 = min(q0 + per, n_pairs);
        // q / N_prb as a truncated float product (q < 1540): the hardware reciprocal is within 1 ulp of 1 / N_prb, three ulp on top make it an
        // upper bound, and the excess -- under 5 * 2^-23 of q / N_prb -- stays far below the 1 / N_prb that separates a quotient from the next
        // integer.  The exact magic number (0xFFFFFFFF / N_prb + 1) was a 32-bit division of a workgroup-uniform value on the vector unit.
        const float r_prb = __builtin_amdgcn_rcpf((float)N_prb) * (1.0f + 0x1.8p-22f);
        uint32_t local = 0;
        for (uint32_t q = q0; q < q1; q++) {
            const uint32_t L = (uint32_t)((float)q * r_prb), prb = al.prb[L >= 7 ? 1 : 0][q - __umul24(L, N_prb)];
            const uint32_t m = L < cfi ? 0u : pdsch_mask(N_ant, cell, sf, L, prb, first_sc, last_sc);
            tab[q].y = m | ((L * N_SC_MAX + prb * 12) << 12); // low 12 bits: RE mask, high 20 bits: L*1200 + first sub-carrier
            local += __popc(m);
        }
        uint32_t incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t n = __shfl_up(incl, o);
            if ((int)ln >= o) incl += n;
        }
        uint32_t run = incl - local;
        for (uint32_t q = q0; q < q1; q++) {
            tab[q].x = run;
            run += __popc(tab[q].y & 0xFFFu);
        }
        if (ln == 63) tab[n_pairs] = make_uint2(incl, 0u); // total
    } else {
        const uint32_t n_words_ub = ((n_pairs - pair0) * 12 * Qm + 31) / 32; // <= max_words; one word of slack for the 2-word window in put_bits
        if (threadIdx.x < 128) qam_lut[threadIdx.x - 64] = qam_lut_entry(threadIdx.x - 64);
        for (uint32_t w = threadIdx.x - 64; w <= n_words_ub; w += blockDim.x - 64) cw[w] = gold_word(gt, c_init, w);
    }
    __syncthreads();
    const uint32_t M_ap = tab[n_pairs].x;
    // pre-decoder / layer de-mapper symbol counts (liblte_phy.cc:7683, 7693, 7720, 7497).  M_ap is a multiple of N_ant for every
    // allocation the extraction above can produce: each (slot, PRB) contributes 12, 8 or -- next to the PBCH / PSS / SSS window of the
    // odd bandwidths -- 6 + 6, 4 + 4 + 6 + 6 resource elements (enumerated over every bandwidth, cell, subframe and control-region size
    // in tests/test_fuzz_cpu.py), so the reference's `M_ap_symb % 4 != 0` branch (:7766-7795, with the mis-strided layer de-mapper
    // it would feed, :7473-7514) is dead code on this path and the division is exact.
    const uint32_t n_grp = M_ap >> (N_ant >> 1), M_symb = n_grp * N_ant, N_bits = M_symb * Qm; // (N_ant is 1, 2 or 4)
    if (threadIdx.x == 0) e_len[a_idx] = N_bits;

    // ---- phase 3: per group of N_ant REs: gather, pre-decode, de-map, descramble.  The modulation is uniform over the workgroup: the whole
    // phase is instantiated per modulation (`run` below), so that the per-element code holds no modulation branch, no re-read of the
    // allocation descriptor and compile-time bit counts -- as one generic routine it was a dozen scalar branches and a scalar load per element
    const float *base = subframes + (size_t)unit * g.sf_stride;
    const float *y_re_p = base, *y_im_p = base + 16 * N_SC_MAX;
    const float *h_re_p = base + 2 * 16 * N_SC_MAX, *h_im_p = h_re_p + (size_t)N_ant * 16 * N_SC_MAX;
    int8_t *e = e_base + (size_t)e_off[a_idx] * 64; // offsets are kept in 64-byte units: a batch may hold more than 4 GiB of soft bits
    auto locate = [&](uint32_t idx) -> uint32_t { // RE index -> L * 1200 + sub-carrier
        uint32_t lo = pair0, hi = n_pairs; // tab[lo].x <= idx < tab[hi].x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tab[mid].x <= idx) lo = mid; else hi = mid;
        }
        const uint2 t = tab[lo];
        uint32_t r = idx - t.x, m = t.y & 0xFFFu;
        for (; r; r--) m &= m - 1;
        return (t.y >> 12) + (uint32_t)__builtin_ctz(m);
    };
    // soft bits are assembled in LDS when they fit and leave with 16-byte stores
    int8_t    *e_lds  = reinterpret_cast<int8_t *>(cw + max_words);
    const bool via_lds = N_bits <= e_lds_cap;
    auto run = [&](auto modc) {
    constexpr uint32_t MOD = decltype(modc)::value, QM = MOD == 3 ? 6 : MOD == 2 ? 4 : MOD == 1 ? 2 : 1;
    constexpr bool     QAM = MOD >= 2;
    // descramble + store the Q_m soft bits of symbol idx: c bit set -> negate (liblte_phy.cc:3833-3836).
    // Q_m*idx is even for Q_m >= 2, so pairs of bytes leave as one 16-bit store.
    auto put_bits = [&](auto *dst, uint32_t idx, const int8_t (&b)[6]) {
        const uint32_t n0 = idx * QM, w = n0 >> 5, sh = n0 & 31;
        const uint32_t c = __builtin_amdgcn_alignbit(cw[w + 1], cw[w], sh); // bits n0.. of the scrambling sequence (cw is padded by one word)
        if (QM == 1) dst[n0] = (c & 1u) ? (int8_t)-b[0] : b[0];
        else {
#pragma unroll
            for (uint32_t k = 0; k < 6; k += 2)
                if (k < QM) {
                    const int lo = ((c >> k) & 1u) ? -b[k] : b[k], hi = ((c >> (k + 1)) & 1u) ? -b[k + 1] : b[k + 1];
                    *reinterpret_cast<uint16_t *>(dst + n0 + k) = (uint16_t)((lo & 0xFF) | ((hi & 0xFF) << 8));
                }
        }
    };
    // 16QAM / 64QAM: every soft bit is +-127, so a symbol is its mask of negative bits; XOR with the scrambling bits and one table
    // read give the Q_m bytes (4: one store; 6: three 16-bit stores, the symbol starts on an even address)
    auto put_qam = [&](auto *dst, uint32_t idx, uint32_t neg) {
        const uint32_t n0 = QM == 4 ? idx << 2 : __umul24(idx, 6u), w = n0 >> 5, sh = n0 & 31;
        const uint32_t c  = __builtin_amdgcn_alignbit(cw[w + 1], cw[w], sh);
        const uint2    v  = qam_lut[(neg ^ c) & (QM == 4 ? 15u : 63u)];
        if (QM == 4) *reinterpret_cast<uint32_t *>(dst + n0) = v.x;
        else {
            *reinterpret_cast<uint16_t *>(dst + n0)     = (uint16_t)v.x;
            *reinterpret_cast<uint16_t *>(dst + n0 + 2) = (uint16_t)(v.x >> 16);
            *reinterpret_cast<uint16_t *>(dst + n0 + 4) = (uint16_t)v.y;
        }
    };
    // equalised symbol (nr, ni) / den -> soft bits of RE idx.  16QAM / 64QAM: the decisions without the divisions (qam_neg_bits_nodiv:
    // the reference's own divisions under a guard next to the thresholds); BPSK / QPSK grade the quotient, so they divide.
    auto put_symbol = [&](auto *dst, uint32_t idx, float nr, float ni, float den) {
        if constexpr (QAM) put_qam(dst, idx, qam_neg_bits_nodiv<MOD>(nr, ni, den));
        else {
            int8_t b[6] = {0, 0, 0, 0, 0, 0};
            demap_symbol<true>(nr / den, ni / den, MOD, b);
            put_bits(dst, idx, b);
        }
    };
    if (ONE_PORT && COMPACT) {
        // MI_LTE_CE_COMPACT: the estimate arrives as magnitude / phase rows at the five CRS symbols (mag in the real-part plane, phase in
        // the imaginary-part plane, rows 0-4).  One thread per (slot, PRB, sub-carrier): it loads the six values its slot's two segments
        // hang on (and, in the second slot, the two phases its first row is wrapped through), runs the reference's time interpolation
        // (the later segments depend on the phases wrapped in the earlier ones), and equalises its slot's resource elements with
        // m * (cos a, sin a) -- the values k_dl_ce would have written, bit for bit.  Rows are addressed as a uniform row pointer plus one
        // 32-bit byte offset per thread (global_load ... saddr), the pair table gives the PRB and the first RE in one LDS read.
        constexpr uint32_t ROW = N_SC_MAX * 4, Y_IM = 16 * ROW, H_M = 32 * ROW, H_A = 48 * ROW; // byte offsets of the planes in a unit
        const uint32_t per_slot = N_prb * 12, n_items = 2 * per_slot;
        // one buffer descriptor for the unit: a load is (descriptor, 32-bit byte offset per thread, row offset in a scalar register)
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(g.sf_stride * 4u), 0x00020000);
        auto ld = [&](uint32_t row, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, (int)row, 0)); };
        auto body = [&](auto *dst) {
            for (uint32_t item = threadIdx.x; item < n_items; item += blockDim.x) {
                const uint32_t s = item >= per_slot ? 1u : 0u, r = item - s * per_slot, i = __umul24(r, 43691u) >> 19, j = r - ((i << 3) + (i << 2)); // r / 12 for r < 2^15; 12 i as shifts (the product form became a 64-bit multiply-add at a quarter of the rate)
                const uint32_t bit = 1u << j, below = bit - 1u;
                const uint2   *tp = tab + (s ? 7 * N_prb : 0u) + i; // the pair of the slot's first symbol; the next symbols' are N_prb apart
                const uint2    t0 = *tp;
                const uint32_t vy = ((t0.y >> 12) + j) << 2;                  // byte offset of (symbol 7s, sub-carrier k) in a plane
                const uint32_t vk = vy - s * (7 * ROW), vh = vk + s * (2 * ROW); // (row 0, k) and (row 2s, k)
                const float M0 = ld(H_M, vh), M1 = ld(H_M + ROW, vh), M2 = ld(H_M + 2 * ROW, vh);
                const float P0 = ld(H_A, vh), P1 = ld(H_A + ROW, vh), P2 = ld(H_A + 2 * ROW, vh);
                const float Q0 = ld(H_A, vk), Q1 = ld(H_A + ROW, vk); // the first slot's phases (second slot only; the same rows again in the first)
                // the slot's symbols: requested before the interpolation needs its inputs
                float yr[7], yi[7];
#pragma unroll
                for (int t = 0; t < 7; t++) { yr[t] = ld(t * ROW, vy); yi[t] = ld(Y_IM + t * ROW, vy); }
                // liblte_phy.cc:6119-6190 for symbols 7s .. 7s+6 (ce_time_interp5 above is the whole subframe): symbols 7s and 7s+4 take
                // the CRS-symbol values as estimated; the slopes see the phases wrapped against their predecessors, from symbol 0 on
                float m[7], a[7];
                m[0] = M0; a[0] = P0; m[4] = M1; a[4] = P1;
                float W0 = P0;
                if (s) W0 = wrap_phase_rare(P0, wrap_phase_rare(Q1, Q0));
                {
                    const float fm = (M1 - M0) / 4, W1 = wrap_phase_rare(P1, W0);
                    const float fa = wrap_phase_rare(W1 - W0, 0.0f) / 4;
                    float cm = M1, ca = W1;
#pragma unroll
                    for (int z = 3; z > 0; z--) { cm -= fm; ca -= fa; m[z] = cm; a[z] = ca; }
                    const float gm = div3(M2 - M1), W2 = wrap_phase_rare(P2, W1);
                    const float ga = div3(wrap_phase_rare(W2 - W1, 0.0f));
                    cm = M2; ca = W2;
#pragma unroll
                    for (int z = 6; z > 4; z--) { cm -= gm; ca -= ga; m[z] = cm; a[z] = ca; }
                }
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const uint2 pr = t ? tp[t * N_prb] : t0;
                    if (!(pr.y & bit)) continue;
                    const uint32_t idx = pr.x + __popc(pr.y & below);
                    float sn, cs;
                    ce_sincos(a[t], sn, cs);
                    const float mm = m[t], hr = mm * cs, hi = mm * sn;
                    const float hn = hr * hr + hi * hi;
                    put_symbol(dst, idx, yr[t] * hr + yi[t] * hi, yi[t] * hr - yr[t] * hi, hn);
                }
            }
        };
        if (via_lds) body(e_lds); else body(e);
    } else if (ONE_PORT) { // one RE = one symbol (liblte_phy.cc:7684-7690): thread per (PRB-symbol pair, sub-carrier), no search
        constexpr int UNR = 4; // loads of UNR independent REs in flight per thread (the kernel is latency-bound otherwise)
        // 21 pairs x 12 sub-carriers per sweep of the workgroup: a thread keeps its sub-carrier, so nothing is divided in the loop
        const uint32_t q_thr = threadIdx.x / 12, j = threadIdx.x - 12 * q_thr, below = (1u << j) - 1u;
        const bool     lane_on = threadIdx.x < 252;
        auto body = [&](auto *dst) {
            for (uint32_t qb = pair0; qb < n_pairs; qb += 21 * UNR) {
                float    yr[UNR], yi[UNR], hr[UNR], hi[UNR];
                uint32_t idx[UNR];
                bool     on[UNR];
#pragma unroll
                for (int r = 0; r < UNR; r++) {
                    const uint32_t q = qb + 21 * r + q_thr;
                    const bool     in = lane_on && q < n_pairs;
                    const uint2    pr = tab[in ? q : 0u];
                    on[r]  = in && ((pr.y >> j) & 1u);
                    idx[r] = pr.x + __popc(pr.y & below);
                    const uint32_t ob = on[r] ? ((pr.y >> 12) + j) << 2 : 0u; // byte offset in 32 bits: one register serves the four planes
                    yr[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(y_re_p) + ob);
                    yi[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(y_im_p) + ob);
                    hr[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(h_re_p) + ob);
                    hi[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(h_im_p) + ob);
                }
#pragma unroll
                for (int r = 0; r < UNR; r++) {
                    if (!on[r]) continue;
                    const float hn = hr[r] * hr[r] + hi[r] * hi[r];
                    put_symbol(dst, idx[r], yr[r] * hr[r] + yi[r] * hi[r], yi[r] * hr[r] - yr[r] * hi[r], hn);
                }
            }
        };
        if (via_lds) body(e_lds); else body(e);
    } else
    for (uint32_t i = threadIdx.x; i < n_grp; i += blockDim.x) {
        float n_re[4], n_im[4], den[4]; // x_p = (n_re, n_im) / den
        if (N_ant == 2) { // Alamouti combiner with the reference's normaliser (liblte_phy.cc:7694-7717)
            const uint32_t p0 = locate(2 * i), p1 = locate(2 * i + 1);
            const float y0r = y_re_p[p0], y0i = y_im_p[p0], y1r = y_re_p[p1], y1i = y_im_p[p1];
            const float h0r = h_re_p[p0], h0i = h_im_p[p0], h1r = h_re_p[16 * N_SC_MAX + p0], h1i = h_im_p[16 * N_SC_MAX + p0];
            const float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i;
            const float hn = sqrtf(a0 * a0 + a1 * a1);
            den[0] = den[1] = hn;
            n_re[0] = h0r * y0r + h0i * y0i + h1r * y1r + h1i * y1i;
            n_im[0] = h0r * y0i - h0i * y0r - h1r * y1i + h1i * y1r;
            n_re[1] = -h1r * y0r - h1i * y0i + h0r * y1r + h0i * y1i;
            n_im[1] = h1r * y0i - h1i * y0r + h0r * y1i - h0i * y1r;
        } else { // N_ant == 4 (liblte_phy.cc:7721-7765); M_ap % 4 == 0 always, see the note at n_grp
            const uint32_t p0 = locate(4 * i), p1 = locate(4 * i + 1), p2 = locate(4 * i + 2), p3 = locate(4 * i + 3);
            const size_t   ps = 16 * N_SC_MAX;
            const float y0r = y_re_p[p0], y0i = y_im_p[p0], y1r = y_re_p[p1], y1i = y_im_p[p1];
            const float y2r = y_re_p[p2], y2i = y_im_p[p2], y3r = y_re_p[p3], y3i = y_im_p[p3];
            const float h0r = h_re_p[p0], h0i = h_im_p[p0], h2r = h_re_p[2 * ps + p0], h2i = h_im_p[2 * ps + p0];
            const float h1r = h_re_p[ps + p2], h1i = h_im_p[ps + p2], h3r = h_re_p[3 * ps + p2], h3i = h_im_p[3 * ps + p2];
            const float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i, a2 = h2r * h2r + h2i * h2i, a3 = h3r * h3r + h3i * h3i;
            const float n02 = sqrtf(a0 * a0 + a2 * a2), n13 = sqrtf(a1 * a1 + a3 * a3);
            den[0] = den[1] = n02; den[2] = den[3] = n13;
            n_re[0] = h0r * y0r + h0i * y0i + h2r * y1r + h2i * y1i;
            n_im[0] = h0r * y0i - h0i * y0r - h2r * y1i + h2i * y1r;
            n_re[1] = -h2r * y0r - h2i * y0i + h0r * y1r + h0i * y1i;
            n_im[1] = -(-h2r * y0i + h2i * y0r - h0r * y1i + h0i * y1r);
            n_re[2] = h1r * y2r + h1i * y2i + h3r * y3r + h3i * y3i;
            n_im[2] = h1r * y2i - h1i * y2r - h3r * y3i + h3i * y3r;
            n_re[3] = -h3r * y2r - h3i * y2i + h1r * y3r + h1i * y3i;
            n_im[3] = -(-h3r * y2i + h3i * y2r - h1r * y3i + h1i * y3r);
        }
        // layer de-mapping d[i*N_ant + p] = x_p[i] (liblte_phy.cc:7506-7513), de-map, descramble (:3833-3836)
        for (uint32_t p = 0; p < N_ant; p++) {
            if (via_lds) put_symbol(e_lds, i * N_ant + p, n_re[p], n_im[p], den[p]);
            else         put_symbol(e, i * N_ant + p, n_re[p], n_im[p], den[p]);
        }
    }
    };
    switch (al.mod_type) {
    case 0:  run(std::integral_constant<uint32_t, 0>{}); break;
    case 1:  run(std::integral_constant<uint32_t, 1>{}); break;
    case 2:  run(std::integral_constant<uint32_t, 2>{}); break;
    default: run(std::integral_constant<uint32_t, 3>{}); break;
    }
    if (via_lds) {
        __syncthreads();
        const uint32_t nq = (N_bits + 15) >> 4; // the allocation's slot in e_base is padded to 64 bytes
        for (uint32_t w = threadIdx.x; w < nq; w += blockDim.x) reinterpret_cast<uint4 *>(e)[w] = reinterpret_cast<const uint4 *>(e_lds)[w];
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host side: plans

struct mi_lte_pdsch_plan {
    mi_lte_dl_cfg cfg;
    uint32_t      decoder = MI_LTE_TURBO_REF, n_iter = 8; // MI_LTE_TURBO_BCJR: mi_lte_pdsch_plan_set_decoder
    int           qpp_spec = 0;
    int8_t       *d_bcjr_soft = nullptr;                  // [max n_cb][3(K+4)] int8 channel values of the group being decoded
    uint8_t      *d_bcjr_bits = nullptr;                  // [max n_cb][K] its hard decisions
    size_t        bcjr_soft_cap = 0, bcjr_bits_cap = 0;
    uint32_t      cfi = 0, n_alloc = 0, out_stride = 0, max_pairs = 0, max_words = 0, max_tbs = 0, packed = 0;
    size_t        e_bytes = 0;
    // capacity of the device arrays (a dynamic plan is re-assigned within it; a static plan's capacity is its first assignment)
    uint32_t      cap_alloc = 0;
    size_t        cap_e_bytes = 0;
    bool          dynamic = false, wide = false; // wide: the output stride of the largest single-code-block transport block, whatever is held
    bool          mapped = false; // the descriptor arrays ARE the pinned staging block, mapped into the device (mi_pdsch_plan_create_mapped)
    mi_lte_pdsch_alloc *d_allocs = nullptr;
    uint32_t *d_e_off = nullptr, *d_e_len = nullptr, *d_cb_alloc = nullptr;
    int8_t   *d_e = nullptr;
    // pinned staging for re-assignments: allocs | e_off | cb_alloc, copied with one command each on the context's stream
    void       *h_stage = nullptr;
    hipEvent_t  staged = nullptr;
    using Group = MiKGroup; // { K, n_cb, cb_base, e_max }
    std::vector<Group>    groups;
    std::vector<uint32_t> h_e_off;
    MiMultiCache          multi; // the merged decode's device tables for `groups` (turbo.hip: mi_turbo_ref_multi)
};

static uint32_t qpp_size_at_least(uint32_t B);
#include "lte_tables.h"
static uint32_t qpp_size_at_least(uint32_t B)
{
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (LTE_QPP_ROWS[r].K >= B) return LTE_QPP_ROWS[r].K;
    return 0;
}

// Host side of a plan: group the allocations by code-block size, lay their soft bits out.  Fills everything but the device arrays;
// cb_alloc receives the allocation index of every code-block slot, group after group.
static int plan_layout(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                       std::vector<uint32_t> &cb_alloc, bool prbs_checked = false)
{
    const mi_lte_dl_cfg *cfg = &pl->cfg;
    pl->cfi       = N_pdcch_symbs;
    pl->n_alloc   = n_alloc;
    pl->max_pairs = pl->max_words = 0;
    pl->groups.clear();
    // two passes and a counting sort by code-block size (188 sizes): a capture's chunk re-plans tens of thousands of allocations per call
    static_assert(LTE_QPP_N_SIZES <= 256, "size index fits a byte");
    uint32_t cnt[LTE_QPP_N_SIZES] = {0}, emax[LTE_QPP_N_SIZES] = {0};
    std::vector<uint8_t> kidx(n_alloc);
    uint32_t max_tbs = 0;
    size_t   off = 0;
    pl->h_e_off.resize(n_alloc);
    for (uint32_t a = 0; a < n_alloc; a++) {
        const mi_lte_pdsch_alloc &al = h_allocs[a];
        const uint32_t B = al.tbs + 24;
        int            r = -1;
        if (B <= 6144) { // first size >= B: the sizes step by 8, 16, 32, 64 (36.212 table 5.1.3-3)
            r = B <= 40 ? 0 : B <= 512 ? (int)((B - 40 + 7) / 8) : B <= 1024 ? 59 + (int)((B - 512 + 15) / 16) : B <= 2048 ? 91 + (int)((B - 1024 + 31) / 32)
                                                                                                                              : 123 + (int)((B - 2048 + 63) / 64);
            if (r >= LTE_QPP_N_SIZES || LTE_QPP_ROWS[r].K < B || (r > 0 && LTE_QPP_ROWS[r - 1].K >= B)) r = -2; // (table and closed form disagree: fall back)
            if (r == -2)
                for (r = 0; r < LTE_QPP_N_SIZES && LTE_QPP_ROWS[r].K < B; r++) {}
        }
        const uint32_t cfi = al.n_pdcch_symbs ? al.n_pdcch_symbs : N_pdcch_symbs;
        if (r < 0 || r >= LTE_QPP_N_SIZES || al.N_prb == 0 || al.N_prb > cfg->N_rb_dl || al.mod_type > 3 || cfi < 1 || cfi > 4) {
            // multi-code-block transport blocks: the reference's own C > 1 path is broken
            ctx->err = "allocation outside the single-code-block envelope (tbs + 24 > 6144) or malformed";
            return MI_LTE_ERR_UNSUPPORTED;
        }
        for (uint32_t s = 0; s < 2 && !prbs_checked; s++) // a resource block past the carrier would be read out of the neighbouring symbol row
            for (uint32_t i = 0; i < al.N_prb; i++)
                if (al.prb[s][i] >= cfg->N_rb_dl) {
                    ctx->err = "allocation names a resource block outside the carrier";
                    return MI_LTE_ERR_INVALID_ARG;
                }
        kidx[a] = (uint8_t)r;
        cnt[r]++;
        max_tbs = std::max(max_tbs, al.tbs);
        const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
        const uint32_t pairs = (14 - cfi) * al.N_prb, e_max = pairs * 12 * Qm;
        pl->max_pairs = std::max(pl->max_pairs, 14 * al.N_prb); // the demodulator's pair table spans all 14 symbols
        pl->max_words = std::max(pl->max_words, (e_max + 31) / 32);
        emax[r]        = std::max(emax[r], e_max);
        pl->h_e_off[a] = (uint32_t)(off >> 6); // in 64-byte units
        off += (e_max + 63) & ~63u;
    }
    if (pl->max_words > 4095) { // the demodulator reads one word past the allocation's last scrambling word
        ctx->err = "allocation larger than the scrambling table";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    pl->e_bytes    = off;
    pl->max_tbs    = max_tbs;
    const uint32_t st_tbs = (pl->dynamic || pl->wide) ? 6120u : max_tbs; // a dynamic plan keeps ONE output stride over its assignments: the largest single-code-block size
    pl->out_stride = pl->packed ? (((st_tbs + 7) / 8 + 63) & ~63u) : ((st_tbs + 63) & ~63u);
    cb_alloc.resize(n_alloc);
    uint32_t base[LTE_QPP_N_SIZES], run = 0;
    for (int r = 0; r < LTE_QPP_N_SIZES; r++) { // groups in ascending block size, allocations inside a group in their own order
        base[r] = run;
        if (cnt[r]) pl->groups.push_back({LTE_QPP_ROWS[r].K, cnt[r], run, emax[r]});
        run += cnt[r];
    }
    for (uint32_t a = 0; a < n_alloc; a++) cb_alloc[base[kidx[a]]++] = a;
    return MI_LTE_OK;
}

static int plan_device_arrays(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, uint32_t cap_alloc, size_t cap_e_bytes)
{
    pl->cap_alloc   = cap_alloc;
    pl->cap_e_bytes = cap_e_bytes ? cap_e_bytes : 64;
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_allocs, sizeof(mi_lte_pdsch_alloc) * cap_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_off, sizeof(uint32_t) * cap_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_len, sizeof(uint32_t) * cap_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_cb_alloc, sizeof(uint32_t) * cap_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e, pl->cap_e_bytes));
    return MI_LTE_OK;
}

extern "C" {

void mi_lte_pdsch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl);

int mi_lte_pdsch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t N_pdcch_symbs,
                             const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc, mi_lte_pdsch_plan **out)
{
    if (!ctx || !cfg || !h_allocs || !out || n_alloc == 0 || N_pdcch_symbs < 1 || N_pdcch_symbs > 4) return MI_LTE_ERR_INVALID_ARG;
    if (!(cfg->N_ant == 1 || cfg->N_ant == 2 || cfg->N_ant == 4)) return MI_LTE_ERR_INVALID_ARG;
    if ((cfg->sample_format & MI_LTE_CE_COMPACT) && cfg->N_ant != 1) { ctx->err = "MI_LTE_CE_COMPACT: single-port cells only"; return MI_LTE_ERR_UNSUPPORTED; }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl      = new mi_lte_pdsch_plan();
    auto  guard   = on_fail([&] { (void)hipStreamSynchronize(ctx->stream); mi_lte_pdsch_plan_destroy(nullptr, pl); });
    pl->cfg       = *cfg;
    std::vector<uint32_t> cb_alloc;
    int rc = plan_layout(ctx, pl, N_pdcch_symbs, h_allocs, n_alloc, cb_alloc);
    if (rc != MI_LTE_OK) return rc;
    rc = plan_device_arrays(ctx, pl, n_alloc, pl->e_bytes);
    if (rc != MI_LTE_OK) return rc;
    MI_H2D(ctx, pl->d_allocs, h_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc);
    MI_H2D(ctx, pl->d_e_off, pl->h_e_off.data(), sizeof(uint32_t) * n_alloc);
    MI_H2D(ctx, pl->d_cb_alloc, cb_alloc.data(), sizeof(uint32_t) * n_alloc);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    guard.armed = false;
    *out = pl;
    return MI_LTE_OK;
}

// A plan whose allocation list changes from run to run (a capture: every subframe has its own DCIs, LTE_fdd_dl_fs_samp_buf.cc:445-515):
// device arrays and pinned staging sized once, mi_lte_pdsch_plan_assign re-plans inside them with three asynchronous copies and no
// allocation, no wait.
int mi_lte_pdsch_plan_create_dynamic(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t max_alloc, size_t max_soft_bytes, mi_lte_pdsch_plan **out)
{
    if (!ctx || !cfg || !out || max_alloc == 0) return MI_LTE_ERR_INVALID_ARG;
    if (!(cfg->N_ant == 1 || cfg->N_ant == 2 || cfg->N_ant == 4)) return MI_LTE_ERR_INVALID_ARG;
    if ((cfg->sample_format & MI_LTE_CE_COMPACT) && cfg->N_ant != 1) { ctx->err = "MI_LTE_CE_COMPACT: single-port cells only"; return MI_LTE_ERR_UNSUPPORTED; }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl    = new mi_lte_pdsch_plan();
    auto  guard = on_fail([&] { mi_lte_pdsch_plan_destroy(nullptr, pl); });
    pl->cfg     = *cfg;
    pl->dynamic = true;
    // the capacity a multiple of four entries: the staging block's three arrays (260-byte structs, then two uint32 arrays) then all start on 16
    // bytes whatever the caller asked for, which is what the copy kernel of mi_pdsch_plan_assign_slice needs (otherwise three copy commands)
    max_alloc = (max_alloc + 3u) & ~3u;
    int rc = plan_device_arrays(ctx, pl, max_alloc, (max_soft_bytes + 63) & ~(size_t)63);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipHostMalloc(&pl->h_stage, (sizeof(mi_lte_pdsch_alloc) + 2 * sizeof(uint32_t)) * (size_t)max_alloc, hipHostMallocDefault));
    MI_HIP_CHECK(ctx, hipEventCreateWithFlags(&pl->staged, hipEventDisableTiming));
    guard.armed = false;
    *out = pl;
    return MI_LTE_OK;
}
} // extern "C"

// The per-call forms' variant of a dynamic plan (hostapi.cc: a handful of allocations per subframe): the three descriptor arrays are not
// copied at all -- they live in pinned host memory mapped into the device, mi_lte_pdsch_plan_assign stores into it and the kernels read
// the few hundred bytes over the link (a copy command costs more API time than that).  Contract: the caller waits for the stream between a
// run of the plan and its next assignment (every per-call form ends with a wait).
int mi_pdsch_plan_create_mapped(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, uint32_t max_alloc, size_t max_soft_bytes, mi_lte_pdsch_plan **out)
{
    if (!ctx || !cfg || !out || max_alloc == 0 || max_alloc > 64) return MI_LTE_ERR_INVALID_ARG;
    if (!(cfg->N_ant == 1 || cfg->N_ant == 2 || cfg->N_ant == 4) || (cfg->sample_format & MI_LTE_CE_COMPACT)) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl    = new mi_lte_pdsch_plan();
    auto  guard = on_fail([&] { mi_lte_pdsch_plan_destroy(nullptr, pl); });
    pl->cfg     = *cfg;
    pl->dynamic = pl->mapped = true;
    pl->cap_alloc   = max_alloc;
    pl->cap_e_bytes = std::max<size_t>((max_soft_bytes + 63) & ~(size_t)63, 64);
    MI_HIP_CHECK(ctx, hipHostMalloc(&pl->h_stage, (sizeof(mi_lte_pdsch_alloc) + 2 * sizeof(uint32_t)) * (size_t)max_alloc, hipHostMallocMapped | hipHostMallocCoherent));
    void *d_stage = nullptr;
    MI_HIP_CHECK(ctx, hipHostGetDevicePointer(&d_stage, pl->h_stage, 0));
    pl->d_allocs   = (mi_lte_pdsch_alloc *)d_stage;
    pl->d_e_off    = (uint32_t *)(pl->d_allocs + max_alloc);
    pl->d_cb_alloc = pl->d_e_off + max_alloc;
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_len, sizeof(uint32_t) * max_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e, pl->cap_e_bytes));
    guard.armed = false;
    *out = pl;
    return MI_LTE_OK;
}

extern "C" {

int mi_lte_pdsch_alloc_decodable(const mi_lte_dl_cfg *cfg, const mi_lte_pdsch_alloc *al, uint32_t N_pdcch_symbs)
{ // the per-allocation conditions of plan_layout, for callers that must not lose a whole list to one chance-CRC DCI
    if (!cfg || !al) return 0;
    const uint32_t cfi = al->n_pdcch_symbs ? al->n_pdcch_symbs : N_pdcch_symbs;
    if (al->tbs + 24 > 6144 || al->tbs + 24 < al->tbs || al->N_prb == 0 || al->N_prb > cfg->N_rb_dl || al->N_prb > 110 || al->mod_type > 3 || cfi < 1 || cfi > 4) return 0;
    if ((14 - cfi) * al->N_prb * 12 * (al->mod_type == 3 ? 6u : al->mod_type == 2 ? 4u : al->mod_type == 1 ? 2u : 1u) > 4095u * 32u) return 0; // the scrambling table
    for (uint32_t s = 0; s < 2; s++)
        for (uint32_t i = 0; i < al->N_prb; i++)
            if (al->prb[s][i] >= cfg->N_rb_dl) return 0;
    return 1;
}

int mi_lte_pdsch_plan_assign(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc)
{
    if (!ctx || !pl || !pl->dynamic || !h_allocs || n_alloc == 0 || N_pdcch_symbs < 1 || N_pdcch_symbs > 4) return MI_LTE_ERR_INVALID_ARG;
    if (n_alloc > pl->cap_alloc) { ctx->err = "more allocations than the dynamic plan was created for"; return MI_LTE_ERR_INVALID_ARG; }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> cb_alloc;
    int rc = plan_layout(ctx, pl, N_pdcch_symbs, h_allocs, n_alloc, cb_alloc);
    if (rc != MI_LTE_OK) { pl->n_alloc = 0; return rc; }
    if (pl->e_bytes > pl->cap_e_bytes) { pl->n_alloc = 0; ctx->err = "more soft bits than the dynamic plan was created for"; return MI_LTE_ERR_INVALID_ARG; }
    if (!pl->mapped) MI_HIP_CHECK(ctx, hipEventSynchronize(pl->staged)); // the previous assignment's copies have left the staging block
    auto *sa = (mi_lte_pdsch_alloc *)pl->h_stage;
    auto *so = (uint32_t *)(sa + pl->cap_alloc), *sc = so + pl->cap_alloc;
    memcpy(sa, h_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc);
    memcpy(so, pl->h_e_off.data(), sizeof(uint32_t) * n_alloc);
    memcpy(sc, cb_alloc.data(), sizeof(uint32_t) * n_alloc);
    if (pl->mapped) return MI_LTE_OK; // the kernels read the block itself (mi_pdsch_plan_create_mapped: the caller has waited for the last run)
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_allocs, sa, sizeof(mi_lte_pdsch_alloc) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_e_off, so, sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_cb_alloc, sc, sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipEventRecord(pl->staged, ctx->stream));
    return MI_LTE_OK;
}

uint32_t mi_lte_pdsch_plan_n_alloc(const mi_lte_pdsch_plan *pl) { return pl ? pl->n_alloc : 0; }
} // extern "C"

// The host pipeline's assignment (pipeline.cc: one per chunk of a capture, tens of thousands of allocations each, on the thread that also
// feeds the copy engines): the chunk's slice of the caller's list goes into the plan's staging block in ONE pass -- unit numbers made
// chunk-local, every allocation tested once (mi_lte_pdsch_alloc_decodable's conditions), one outside the envelope replaced by a one-block
// QPSK stand-in and its index reported -- instead of a copy, a test pass and the layout's own test pass.
// copy_stream: where the three descriptor copies go (nullptr: the context's stream).  The pipeline passes its input-copy stream: a third
// engine moving 4.8 MB per chunk next to the sample and result copies slowed both (profiles/r05_host_pipeline_profile.txt, section 5).
int mi_pdsch_plan_assign_slice(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, uint32_t N_pdcch_symbs, const mi_lte_pdsch_alloc *h_src, uint32_t n_alloc, uint32_t unit0,
                               std::vector<uint32_t> *refused, hipStream_t copy_stream)
{
    if (!ctx || !pl || !pl->dynamic || pl->mapped || !h_src || n_alloc == 0 || N_pdcch_symbs < 1 || N_pdcch_symbs > 4) return MI_LTE_ERR_INVALID_ARG;
    const hipStream_t cs = copy_stream ? copy_stream : ctx->stream;
    if (n_alloc > pl->cap_alloc) { ctx->err = "more allocations than the dynamic plan was created for"; return MI_LTE_ERR_INVALID_ARG; }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MI_HIP_CHECK(ctx, hipEventSynchronize(pl->staged)); // the previous assignment's copies have left the staging block
    auto *sa = (mi_lte_pdsch_alloc *)pl->h_stage;
    auto *so = (uint32_t *)(sa + pl->cap_alloc), *sc = so + pl->cap_alloc;
    for (uint32_t i = 0; i < n_alloc; i++) {
        mi_lte_pdsch_alloc &a = sa[i];
        a = h_src[i];
        a.unit -= unit0;
        if (!mi_lte_pdsch_alloc_decodable(&pl->cfg, &a, N_pdcch_symbs)) {
            // a DCI that passed its CRC by chance: the reference fails that one allocation (liblte_phy.cc:3690-3853); the slot keeps its place
            a.N_prb = 1; a.prb[0][0] = a.prb[1][0] = 0; a.mod_type = 1; a.tbs = 16; a.rv_idx = 0; a.tx_mode = pl->cfg.N_ant == 1 ? 1 : 2;
            if (a.n_pdcch_symbs > 4) a.n_pdcch_symbs = 0;
            if (refused) refused->push_back(i);
        }
    }
    std::vector<uint32_t> cb_alloc;
    int rc = plan_layout(ctx, pl, N_pdcch_symbs, sa, n_alloc, cb_alloc, true);
    if (rc != MI_LTE_OK) { pl->n_alloc = 0; return rc; }
    if (pl->e_bytes > pl->cap_e_bytes) { pl->n_alloc = 0; ctx->err = "more soft bits than the dynamic plan was created for"; return MI_LTE_ERR_INVALID_ARG; }
    memcpy(so, pl->h_e_off.data(), sizeof(uint32_t) * n_alloc);
    memcpy(sc, cb_alloc.data(), sizeof(uint32_t) * n_alloc);
    // The three arrays leave the (pinned) staging block with a copy KERNEL on the same stream, not with copy commands: the runtime rotates its
    // copy engines from command to command, and three small commands per chunk between the sample copies put every third chunk's samples on
    // the engine the result copy of the chunk before was using -- a 3.1 ms + 1.0 ms pair where 2.4 + 0.3 is the rule
    // (profiles/r05_host_pipeline_profile.txt section 6, and section 7 for this)
    static const bool slice_by_engine = getenv("MI_LTE_SLICE_COPY_COMMANDS") != nullptr; // (A/B switch)
    if (slice_by_engine) {
        MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_allocs, sa, sizeof(mi_lte_pdsch_alloc) * n_alloc, hipMemcpyHostToDevice, cs));
        MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_e_off, so, sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, cs));
        MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_cb_alloc, sc, sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, cs));
    } else {
        const MiCopySeg segs[3] = {{pl->d_allocs, sa, sizeof(mi_lte_pdsch_alloc) * n_alloc}, {pl->d_e_off, so, sizeof(uint32_t) * n_alloc}, {pl->d_cb_alloc, sc, sizeof(uint32_t) * n_alloc}};
        MI_HIP_CHECK(ctx, mi_pinned_segments_to_device(ctx, segs, 3, cs));
    }
    MI_HIP_CHECK(ctx, hipEventRecord(pl->staged, cs));
    return MI_LTE_OK;
}
void mi_pdsch_plan_wide_stride(mi_lte_pdsch_plan *pl) { pl->wide = true; (void)mi_lte_pdsch_plan_set_output(pl, pl->packed); }
extern "C" {

void mi_lte_pdsch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl)
{
    if (!pl) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (!pl->mapped) { // (mapped: the three are views of h_stage)
        (void)hipFree(pl->d_allocs);
        (void)hipFree(pl->d_e_off);
        (void)hipFree(pl->d_cb_alloc);
    }
    (void)hipFree(pl->d_e_len);
    (void)hipFree(pl->d_e);
    if (pl->d_bcjr_soft) (void)hipFree(pl->d_bcjr_soft);
    if (pl->d_bcjr_bits) (void)hipFree(pl->d_bcjr_bits);
    if (pl->h_stage) (void)hipHostFree(pl->h_stage);
    if (pl->staged) (void)hipEventDestroy(pl->staged);
    mi_multi_cache_free(&pl->multi);
    delete pl;
}

uint32_t mi_lte_pdsch_plan_out_stride(const mi_lte_pdsch_plan *pl) { return pl ? pl->out_stride : 0; }

// the de-mappers' atan2f (phy_dev.hpp: the host libm's algorithm restated) evaluated on the host, for the test that pins it to libm
int mi_lte_model_atan2f(const float *h_y, const float *h_x, float *h_out, size_t n)
{
    if (!h_y || !h_x || !h_out) return MI_LTE_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; i++) h_out[i] = ref_atan2f(h_y[i], h_x[i]);
    return MI_LTE_OK;
}

int mi_lte_pdsch_plan_soft_bits(const mi_lte_pdsch_plan *pl, uint32_t alloc, const int8_t **d_e, const uint32_t **d_len)
{
    if (!pl || alloc >= pl->n_alloc || !d_e || !d_len) return MI_LTE_ERR_INVALID_ARG;
    *d_e   = pl->d_e + (size_t)pl->h_e_off[alloc] * 64;
    *d_len = pl->d_e_len + alloc;
    return MI_LTE_OK;
}

int mi_lte_pdsch_plan_set_decoder(mi_lte_pdsch_plan *pl, uint32_t mode, uint32_t n_iter, int qpp_spec)
{
    const bool bcjr = mode == MI_LTE_TURBO_BCJR || mode == MI_LTE_TURBO_BCJR_BLOCK || mode == MI_LTE_TURBO_BCJR_EARLY;
    if (!pl || !(mode == MI_LTE_TURBO_REF || bcjr) || (bcjr && (n_iter == 0 || n_iter > 64))) return MI_LTE_ERR_INVALID_ARG;
    pl->decoder = mode; pl->n_iter = n_iter; pl->qpp_spec = qpp_spec;
    return MI_LTE_OK;
}

int mi_lte_pdsch_plan_set_output(mi_lte_pdsch_plan *pl, uint32_t packed)
{
    if (!pl) return MI_LTE_ERR_INVALID_ARG;
    pl->packed     = packed ? 1u : 0u;
    const uint32_t st_tbs = (pl->dynamic || pl->wide) ? 6120u : pl->max_tbs;
    pl->out_stride = packed ? (((st_tbs + 7) / 8 + 63) & ~63u) : ((st_tbs + 63) & ~63u);
    return MI_LTE_OK;
}

int mi_lte_pdsch_decode_run(mi_lte_ctx *ctx, mi_lte_pdsch_plan *pl, const float *d_subframes, const uint32_t *d_subfr_num,
                            const uint32_t *d_n_id_cell, uint8_t *d_out_bits, int32_t *d_status)
{
    if (!ctx || !pl || !d_subframes || !d_subfr_num || !d_n_id_cell || !d_out_bits || !d_status) return MI_LTE_ERR_INVALID_ARG;
    if (pl->n_alloc == 0) { ctx->err = "the dynamic plan holds no allocations (mi_lte_pdsch_plan_assign)"; return MI_LTE_ERR_INVALID_ARG; }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    DemodGeom  g{pl->cfg.N_rb_dl, pl->cfg.N_ant, pl->cfi, (uint32_t)mi_lte_subframe_floats(pl->cfg.N_ant)};
    GoldTables gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    // LDS: pair table | scrambling words | (when it fits in 32 KiB) the allocation's soft bits
    const uint32_t words_al = (pl->max_words + 1u + 3u) & ~3u, e_bytes = (pl->max_words * 32 + 63u) & ~63u;
    const uint32_t e_cap = (e_bytes <= 32 * 1024) ? e_bytes : 0;
    const uint32_t pairs_al = ((2 * (pl->max_pairs + 1) + 3u) & ~3u);
    const size_t lds = sizeof(uint32_t) * ((size_t)pairs_al + words_al) + e_cap;
    uint32_t threads = 256;
    if (const char *ev = getenv("MI_LTE_PDSCH_THREADS")) { // (tuning aid)
        const int t = atoi(ev);
        if (t >= 64 && t <= 256 && t % 64 == 0) threads = (uint32_t)t;
    }
    if (g.N_ant == 1 && (pl->cfg.sample_format & MI_LTE_CE_COMPACT))
        MI_LAUNCH(ctx, "k_pdsch_demod", (k_pdsch_demod<true, true>), dim3(pl->n_alloc), dim3(threads), lds, d_subframes, g, pl->d_allocs, d_subfr_num,
                  d_n_id_cell, gt, pl->d_e, pl->d_e_off, pl->d_e_len, pl->max_pairs, words_al, e_cap);
    else if (g.N_ant == 1)
        MI_LAUNCH(ctx, "k_pdsch_demod", k_pdsch_demod<true>, dim3(pl->n_alloc), dim3(threads), lds, d_subframes, g, pl->d_allocs, d_subfr_num,
                  d_n_id_cell, gt, pl->d_e, pl->d_e_off, pl->d_e_len, pl->max_pairs, words_al, e_cap);
    else
        MI_LAUNCH(ctx, "k_pdsch_demod", k_pdsch_demod<false>, dim3(pl->n_alloc), dim3(threads), lds, d_subframes, g, pl->d_allocs, d_subfr_num,
                  d_n_id_cell, gt, pl->d_e, pl->d_e_off, pl->d_e_len, pl->max_pairs, words_al, e_cap);
    MI_HIP_CHECK(ctx, hipGetLastError());
    const bool bcjr = pl->decoder == MI_LTE_TURBO_BCJR || pl->decoder == MI_LTE_TURBO_BCJR_BLOCK || pl->decoder == MI_LTE_TURBO_BCJR_EARLY;
    if (bcjr) {
        size_t soft = 0, bits = 0;
        for (auto &gr : pl->groups) { soft = std::max(soft, (size_t)gr.n_cb * 3 * (gr.K + 4)); bits = std::max(bits, (size_t)gr.n_cb * gr.K); }
        if (soft > pl->bcjr_soft_cap || bits > pl->bcjr_bits_cap) { // first run, or a re-assigned plan that needs more
            MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (pl->d_bcjr_soft) (void)hipFree(pl->d_bcjr_soft);
            if (pl->d_bcjr_bits) (void)hipFree(pl->d_bcjr_bits);
            pl->d_bcjr_soft = nullptr; pl->d_bcjr_bits = nullptr;
            MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_bcjr_soft, soft));
            MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_bcjr_bits, bits));
            pl->bcjr_soft_cap = soft; pl->bcjr_bits_cap = bits;
        }
    }
    auto run_group = [&](const mi_lte_pdsch_plan::Group &gr) -> int {
        if (bcjr)
            return mi_turbo_bcjr_group(ctx, gr.K, gr.n_cb, pl->d_allocs, pl->d_cb_alloc + gr.cb_base, pl->d_e, pl->d_e_off, pl->d_e_len, d_out_bits,
                                       pl->out_stride, d_status, false, pl->d_bcjr_soft, pl->d_bcjr_bits, pl->n_iter, pl->qpp_spec, pl->packed != 0, gr.e_max,
                                       pl->decoder == MI_LTE_TURBO_BCJR_BLOCK, pl->decoder == MI_LTE_TURBO_BCJR_EARLY);
        return mi_turbo_ref_group(ctx, gr.K, gr.n_cb, pl->d_allocs, pl->d_cb_alloc + gr.cb_base, pl->d_e, pl->d_e_off, pl->d_e_len, d_out_bits,
                                  pl->out_stride, d_status, gr.e_max, false, pl->packed != 0);
    };
    // The block-size groups of a decode are independent of each other (own code blocks, own status words and output rows).  In a batch the
    // groups differ widely in size -- W4: 65 536 blocks of K = 1088 next to 524 288 of K = 3264 -- and the small ones do not fill the device
    // (the K = 1088 trellis walk is one or two wavefronts per SIMD: 0.46 ms of latency for 3 % of the work), so with the reference-faithful
    // decoder every group but the largest can run on a side stream, in a scratch block of its own, next to the largest one.
    size_t big = 0, total_cb = 0;
    for (size_t i = 0; i < pl->groups.size(); i++) {
        total_cb += pl->groups[i].n_cb;
        if ((size_t)pl->groups[i].n_cb * pl->groups[i].K > (size_t)pl->groups[big].n_cb * pl->groups[big].K) big = i;
    }
    // Opt-in (MI_LTE_GROUP_STREAMS=1): worth 1 % of the W4 step (24.77 -> 24.55 ms), and it stretches the launches that overlap -- the
    // per-launch times the roofline accounting of bench.py and the rocprof summaries are built on stop describing a kernel alone.
    static const bool side_ok = [] { const char *e = getenv("MI_LTE_GROUP_STREAMS"); return e && atoi(e) != 0; }();
    // A decode with several block sizes -- a batch, or a per-call caller's subframe with a handful of transport blocks: ONE launch set over all sizes
    // (turbo.hip: KSeg).  A cell's TTIs hold dozens of the 188 sizes; size by size that is ~7 launches per size in series, each a sliver
    // of the device -- 65 536 mixed subframes took 72.6 ms that way, 48.8 of them in the decoder (profiles/r06_chain_mixed_per_size.json).
    // MI_LTE_NO_MERGED_DECODE=1 keeps the per-size launches (A/B).
    static const bool merged_off = [] { const char *e = getenv("MI_LTE_NO_MERGED_DECODE"); return e && atoi(e) != 0; }();
    if (!bcjr && !merged_off && ctx->merged_decode && !side_ok && pl->groups.size() >= 2) {
        // (MI_LTE_MERGE_MAX_TILES=n, tuning aid: a size with more than n tiles of 64 blocks keeps its own launches.  No limit by default: W4's two
        // sizes run 23.6-23.9 ms merged -- the trellis kernel's two launches instead of four, 6.5 against 6.8 ms -- and 24.0-24.3 size by size,
        // profiles/r06_variants_merged_prep.txt)
        static const uint32_t max_tiles = [] { const char *e = getenv("MI_LTE_MERGE_MAX_TILES"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 0xFFFFFFFFu; }();
        auto merged = [&](const MiKGroup &gr) { return mi_turbo_ref_multi_takes(gr.K, gr.e_max) && (gr.n_cb + 63) / 64 <= max_tiles; };
        std::vector<MiKGroup> take;
        for (auto &gr : pl->groups)
            if (merged(gr)) take.push_back(gr);
        if (take.size() >= 2) {
            rc = mi_turbo_ref_multi(ctx, take.data(), (uint32_t)take.size(), pl->d_allocs, pl->d_cb_alloc, pl->d_e, pl->d_e_off, pl->d_e_len, d_out_bits, pl->out_stride, d_status,
                                    false, pl->packed != 0, &pl->multi);
            if (rc != MI_LTE_OK) return rc;
            for (auto &gr : pl->groups) // (an allocation repeated over more than 258 laps of its circular buffer: 32-bit sums, the per-size path)
                if (!merged(gr) && (rc = run_group(gr)) != MI_LTE_OK) return rc;
            ctx->last_kernels = take.size() == pl->groups.size() ? "k_pdsch_demod:1,k_cb_desc:1,k_turbo_prep,k_turbo_siso:2,k_turbo_perm,k_turbo_vote per workgroup width over all block sizes"
                                                                  : "k_pdsch_demod:1,k_cb_desc:1,k_turbo_prep,k_turbo_siso:2,k_turbo_perm,k_turbo_vote per workgroup width over all block sizes but the per-size ones";
            return MI_LTE_OK;
        }
    }
    if (!bcjr && side_ok && pl->groups.size() >= 2 && total_cb >= 16384) {
        if (!ctx->side_stream) {
            MI_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            MI_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            MI_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        MI_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream)); // the soft bits are there, the previous run's side work was joined
        MI_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        auto swap_side = [&] { std::swap(ctx->stream, ctx->side_stream); std::swap(ctx->scratch, ctx->side_scratch); std::swap(ctx->scratch_bytes, ctx->side_scratch_bytes); };
        swap_side(); // the launch helpers take the stream and the scratch from the context
        for (size_t i = 0; i < pl->groups.size() && rc == MI_LTE_OK; i++)
            if (i != big) rc = run_group(pl->groups[i]);
        swap_side();
        // whatever happened on the side stream is joined before this function returns, on every path: groups already launched there
        // still write d_out_bits / d_status / the side scratch, and a later mi_lte_sync only waits for ctx->stream
        const hipError_t e_rec = hipEventRecord(ctx->ev_join, ctx->side_stream);
        if (rc == MI_LTE_OK && e_rec == hipSuccess) rc = run_group(pl->groups[big]);
        const hipError_t e_join = e_rec == hipSuccess ? hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) : hipStreamSynchronize(ctx->side_stream);
        if (rc != MI_LTE_OK) return rc;
        MI_HIP_CHECK(ctx, e_rec);
        MI_HIP_CHECK(ctx, e_join);
    } else {
        for (auto &gr : pl->groups) {
            rc = run_group(gr);
            if (rc != MI_LTE_OK) return rc;
        }
    }
    ctx->last_kernels = bcjr ? "k_pdsch_demod:1,k_rm_to_i8,k_bcjr_*,k_crc_finish per block size"
                                                         : "k_pdsch_demod:1,k_turbo_prep,k_turbo_siso,k_turbo_perm,k_turbo_vote per block size";
    return MI_LTE_OK;
}

} // extern "C"
