// PCFICH + PDCCH common-search-space decoding on gfx950: restates liblte_phy_pdcch_channel_decode
// (liblte/src/liblte_phy.cc:4519-5135) for a batch of device subframes -- the step between the front end and
// liblte_phy_pdsch_channel_decode in every real subframe (LTE_fdd_dl_fs_samp_buf.cc:445-470).
//
// Everything that only depends on the cell and on the control-format indicator is index arithmetic and is done once
// per plan on the host, following the reference step by step: PCFICH resource elements (pcfich_channel_demap :7887-7925),
// PHICH resource-element groups (phich_channel_demap :8224-8290), the PDCCH REG walk with its PCFICH/PHICH/CRS
// exclusions (:4590-4700), the cyclic-shift and sub-block-interleaver undo (:4702-4824), the CCEs and the six
// common-search-space candidates (4 x aggregation 4, 2 x aggregation 8), and the convolutional rate-unmatching index map
// (rate_unmatch_conv :11597-11755).  The device does the arithmetic:
//
//   k_pdcch_decode : one workgroup per subframe, one wavefront per candidate.  PCFICH: gather, transmit-diversity
//       combiner, QPSK soft de-map, descramble, CFI by minimum distance (cfi_channel_decode :13648-13706).  Per candidate:
//       gather / combine / de-map / descramble (:4862-4905), then for DCI formats 1A and 1C: rate un-matching as an
//       LDS scatter-add, the reference's K = 7 Viterbi decoder (viterbi_decode :10161-10332: all states start at 0 -- it
//       is not tail-biting aware --, survivor chosen on the Hamming metric while a weighted metric is accumulated,
//       traceback by re-comparing stored metrics) with ONE TRELLIS STATE PER LANE (64 states = one wavefront; the two
//       predecessor metrics arrive by lane shuffle, the traceback compares are one ballot per step), CRC16 and the RNTI
//       test for SI-, P- and RA-RNTI (dci_channel_decode :12952-13040).
// The soft values are integers from the de-mapper on, so the decoder is exact; the de-mapper itself (atan2f / sqrtf) is
// float.  Candidate CCEs past the last CCE of the subframe read stale scratch in the reference; here they count as erasures
// (what a fresh LIBLTE_PHY_STRUCT gives).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "ctx.hpp"
#include "phy_dev.hpp"
#include "lte_tables.h"
#include "synth.hpp"

namespace {

constexpr int      N_SC_MAX = 1200;
constexpr uint32_t N_CAND = 6, RE_MAX = 288, NO_RE = 0xFFFFFFFFu;

struct PdcchDev {
    uint32_t N_rb_dl, N_ant, n_cells, dci_size[2];
    uint32_t per_port; // 1: every port's own estimate (MI_LTE_PDCCH_PER_PORT_ESTIMATES); 0: the rows the reference really reads
    const uint32_t *cells;    // [n_cells] cell ids the tables were built for
    const uint32_t *pcfich;   // [n_cells][16] RE index (l*1200 + k) of the PCFICH
    const uint32_t *cand;     // [n_cells][4 (N_symbs - 1)][N_CAND][RE_MAX] RE index, NO_RE = candidate absent
    const uint16_t *rm_map;   // [2 formats][2 (E = 288, 576)][576] position in d[3*N_c] of every received bit
};

struct PdcchResult { // per unit
    uint32_t cfi;            // 0: PCFICH did not decode
    uint32_t n_symbs;
    uint32_t rnti[12];       // slot = candidate*2 + format (0 = 1A, 1 = 1C); 0 = nothing found
    uint32_t payload[12];    // DCI bits, first bit in the MSB of the n-bit field
};

// transmit-diversity combiner of one group of N_ant resource elements (pre_decoder_and_matched_filter_dl,
// liblte_phy.cc:7645-7796); h[p][e]: estimate of port p at element e of the group
template <uint32_t N_ant>
__device__ __forceinline__ void combine(const float (&yr)[4], const float (&yi)[4], const float (&hr)[4][4], const float (&hi)[4][4], float (&xr)[4],
                                        float (&xi)[4])
{
    if (N_ant == 1) {
        const float hn = hr[0][0] * hr[0][0] + hi[0][0] * hi[0][0];
        xr[0] = (yr[0] * hr[0][0] + yi[0] * hi[0][0]) / hn;
        xi[0] = (yi[0] * hr[0][0] - yr[0] * hi[0][0]) / hn;
    } else if (N_ant == 2) {
        const float h0r = hr[0][0], h0i = hi[0][0], h1r = hr[1][0], h1i = hi[1][0];
        const float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i, hn = sqrtf(a0 * a0 + a1 * a1);
        xr[0] = (h0r * yr[0] + h0i * yi[0] + h1r * yr[1] + h1i * yi[1]) / hn;
        xi[0] = (h0r * yi[0] - h0i * yr[0] - h1r * yi[1] + h1i * yr[1]) / hn;
        xr[1] = (-h1r * yr[0] - h1i * yi[0] + h0r * yr[1] + h0i * yi[1]) / hn;
        xi[1] = (h1r * yi[0] - h1i * yr[0] + h0r * yi[1] - h0i * yr[1]) / hn;
    } else {
        const float h0r = hr[0][0], h0i = hi[0][0], h2r = hr[2][0], h2i = hi[2][0];
        const float h1r = hr[1][2], h1i = hi[1][2], h3r = hr[3][2], h3i = hi[3][2];
        const float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i, a2 = h2r * h2r + h2i * h2i, a3 = h3r * h3r + h3i * h3i;
        const float n02 = sqrtf(a0 * a0 + a2 * a2), n13 = sqrtf(a1 * a1 + a3 * a3);
        xr[0] = (h0r * yr[0] + h0i * yi[0] + h2r * yr[1] + h2i * yi[1]) / n02;
        xi[0] = (h0r * yi[0] - h0i * yr[0] - h2r * yi[1] + h2i * yr[1]) / n02;
        xr[1] = (-h2r * yr[0] - h2i * yi[0] + h0r * yr[1] + h0i * yi[1]) / n02;
        xi[1] = -(-h2r * yi[0] + h2i * yr[0] - h0r * yi[1] + h0i * yr[1]) / n02;
        xr[2] = (h1r * yr[2] + h1i * yi[2] + h3r * yr[3] + h3i * yi[3]) / n13;
        xi[2] = (h1r * yi[2] - h1i * yr[2] - h3r * yi[3] + h3i * yr[3]) / n13;
        xr[3] = (-h3r * yr[2] - h3i * yi[2] + h1r * yr[3] + h1i * yi[3]) / n13;
        xi[3] = -(-h3r * yi[2] + h3i * yr[2] - h1r * yi[3] + h1i * yr[3]) / n13;
    }
}

// gather -> combine -> QPSK soft de-map -> descramble for n_re resource elements listed in re[]; soft[2*n_re] integer soft bits
//
// Which estimate the combiner gets for port p.  The reference hands pre_decoder_and_matched_filter_dl the array
// pdcch_c_est_re[4][288] with a port stride of 576 (liblte_phy.cc:4881, :7935), so "port 1" is row 2, and "ports 2, 3"
// run off the end into pdcch_c_est_im rows 0, 2 (real part) and the transmitter's scratch (imaginary part, zeros in a
// receiver).  With two ports row 2 is never written: port 1's estimate is 0 and the second antenna goes unsuppressed; with
// four ports nothing decodes.  per_port = 0 reproduces exactly that (rows never written read as the zeros a fresh
// LIBLTE_PHY_STRUCT holds); per_port = 1 is the decoder the reference meant.
// (n_planes = number of channel-estimate planes the device subframe was laid out with; N_ant = ports the combiner assumes)
template <uint32_t N_ant>
__device__ __forceinline__ void demod_res_n(const float *__restrict__ base, uint32_t n_planes, uint32_t per_port, const uint32_t *__restrict__ re,
                                            uint32_t n_re, const GoldTables &gt, uint32_t c_init, uint32_t c_off, int *soft, uint32_t ln)
{
    const float *y_re_p = base, *y_im_p = base + 16 * N_SC_MAX, *h_re_p = base + 2 * 16 * N_SC_MAX;
    const float *h_im_p = h_re_p + (size_t)n_planes * 16 * N_SC_MAX;
    for (uint32_t g = ln; g < n_re / N_ant; g += 64) {
        float yr[4], yi[4], hr[4][4], hi[4][4], xr[4], xi[4];
        const bool absent = re[g * N_ant] == NO_RE;
        for (uint32_t e = 0; e < N_ant; e++) {
            const uint32_t p = absent ? 0u : re[g * N_ant + e];
            yr[e] = y_re_p[p]; yi[e] = y_im_p[p];
            for (uint32_t a = 0; a < N_ant; a++) { hr[a][e] = h_re_p[(size_t)a * 16 * N_SC_MAX + p]; hi[a][e] = h_im_p[(size_t)a * 16 * N_SC_MAX + p]; }
        }
        if (!per_port && N_ant == 2) {
            for (uint32_t e = 0; e < 2; e++) hr[1][e] = hi[1][e] = 0.0f;
        } else if (!per_port && N_ant == 4) {
            for (uint32_t e = 0; e < 4; e++) {
                const float re0 = hr[0][e], im0 = hi[0][e], re2 = hr[2][e], im2 = hi[2][e];
                (void)re0;
                hr[1][e] = re2; hi[1][e] = im2;
                hr[2][e] = im0; hi[2][e] = 0.0f;
                hr[3][e] = im2; hi[3][e] = 0.0f;
            }
        }
        combine<N_ant>(yr, yi, hr, hi, xr, xi);
        for (uint32_t e = 0; e < N_ant; e++) { // layer de-mapping d[g*N_ant + e] = x_e[g] (layer_demapper_dl, :7473-7514)
            int8_t b[6] = {0, 0, 0, 0, 0, 0};
            demap_symbol(xr[e], xi[e], 1 /* QPSK */, b);
            const uint32_t n = 2 * (g * N_ant + e), cn = c_off + n;
            const uint32_t w = gold_word(gt, c_init, cn >> 5), w2 = gold_word(gt, c_init, (cn + 1) >> 5);
            soft[n]     = absent ? 0 : ((w >> (cn & 31)) & 1u) ? -(int)b[0] : (int)b[0];
            soft[n + 1] = absent ? 0 : ((w2 >> ((cn + 1) & 31)) & 1u) ? -(int)b[1] : (int)b[1];
        }
    }
}

__device__ __forceinline__ void demod_res(const float *__restrict__ base, uint32_t n_planes, uint32_t N_ant, uint32_t per_port,
                                          const uint32_t *__restrict__ re, uint32_t n_re, const GoldTables &gt, uint32_t c_init, uint32_t c_off,
                                          int *soft, uint32_t ln)
{
    // the port count is a template parameter so that the small per-group arrays stay in registers
    if (N_ant == 1) demod_res_n<1>(base, n_planes, per_port, re, n_re, gt, c_init, c_off, soft, ln);
    else if (N_ant == 2) demod_res_n<2>(base, n_planes, per_port, re, n_re, gt, c_init, c_off, soft, ln);
    else demod_res_n<4>(base, n_planes, per_port, re, n_re, gt, c_init, c_off, soft, ln);
}

// The reference's K = 7, rate-1/3 Viterbi decoder (viterbi_decode, liblte_phy.cc:10161-10332) over N trellis steps, one state per
// lane: all states start at 0 (it is not tail-biting aware), the survivor is chosen on the Hamming branch metric while the
// weighted metric p + w*br is what is stored, the end state is the first strict minimum, and the traceback re-compares the
// stored metrics of each predecessor pair (one ballot per step, kept in dec[]).  d: 3N integer soft bits in LDS.  The decoded
// bits come back on lane 0, bit t at position t of (lo, hi).
__device__ __forceinline__ void viterbi_k7(const int *d, uint32_t N, uint64_t *dec, uint32_t ln, uint32_t &bits_lo, uint32_t &bits_hi)
{
    // trellis labels of this lane's state (:10197-10222): register = input bit | predecessor state
    const uint32_t G[3] = {0133, 0171, 0165};
    uint32_t lab[2] = {0, 0};
    for (uint32_t k = 0; k < 2; k++) {
        const uint32_t prev = (2 * ln + k) & 63u, reg = ((ln >> 5) << 6) | prev;
        for (uint32_t o = 0; o < 3; o++) lab[k] |= ((uint32_t)__popc(reg & G[o]) & 1u) << o;
    }
    int pm = 0;
    for (uint32_t i = 0; i < N; i++) {
        // which of each predecessor pair has the larger stored metric (what the traceback re-compares, :10300-10308)
        const int      up = __shfl_down(pm, 1);
        const uint64_t gt_mask = __ballot(pm > up); // bit 2m: pm[2m] > pm[2m+1]
        if (ln == 0) dec[i] = gt_mask;
        const int d0 = d[3 * i], d1 = d[3 * i + 1], d2 = d[3 * i + 2];
        const uint32_t in = (d0 < 0 ? 1u : 0u) | (d1 < 0 ? 2u : 0u) | (d2 < 0 ? 4u : 0u);
        const int      w  = abs(d0) + abs(d1) + abs(d2);
        const int      p0 = __shfl(pm, (2 * ln) & 63), p1 = __shfl(pm, (2 * ln + 1) & 63);
        const int      b0 = __popc(lab[0] ^ in), b1 = __popc(lab[1] ^ in);
        pm = (b0 + p0 > b1 + p1) ? p1 + w * b1 : p0 + w * b0; // select on the Hamming metric, accumulate the weighted one
    }
    // end state: first strict minimum (:10281-10292)
    int      best = pm;
    uint32_t st   = ln;
    for (int o = 32; o > 0; o >>= 1) {
        const int      ob = __shfl_xor(best, o);
        const uint32_t os = __shfl_xor(st, o);
        if (ob < best || (ob == best && os < st)) { best = ob; st = os; }
    }
    bits_lo = bits_hi = 0;
    if (ln == 0) {
        // traceback (:10294-10309) and bit read-out (:10313-10331): states s_N .. s_0, bit t from (s_{t+1}, s_t)
        uint32_t cur = st;
        for (int i = (int)N - 1; i >= 0; i--) {
            const uint32_t p0 = (2 * cur) & 63u;
            const uint32_t prev = ((dec[i] >> p0) & 1ull) ? p0 + 1 : p0;
            const uint32_t bit  = (cur < prev) ? 0u : (cur > prev) ? 1u : (cur == 0 ? 0u : 1u);
            if (i < 32) bits_lo |= bit << i; else bits_hi |= bit << (i - 32);
            cur = prev;
        }
    }
}

// CRC16 (polynomial 0x11021, calc_crc :9713-9743) of bits 0..n_info-1 XOR the 16 parity bits that follow them; bit t of the block at
// position t of (lo, hi).  Also returns the information bits, first bit in the MSB of an n_info-bit field.
__device__ __forceinline__ uint32_t crc16_syndrome(uint32_t bits_lo, uint32_t bits_hi, uint32_t n_info, uint32_t &payload)
{
    auto bit_at = [&](uint32_t t) { return t < 32 ? (bits_lo >> t) & 1u : (bits_hi >> (t - 32)) & 1u; };
    uint32_t rem = 0, par = 0;
    payload = 0;
    for (uint32_t t = 0; t < n_info + 16; t++) {
        rem = (rem << 1) | (t < n_info ? bit_at(t) : 0u);
        if (rem & 0x10000u) rem ^= 0x11021u;
    }
    for (uint32_t t = 0; t < n_info; t++) payload = (payload << 1) | bit_at(t);
    for (uint32_t t = 0; t < 16; t++) par = (par << 1) | bit_at(n_info + t);
    return (par ^ rem) & 0xFFFFu;
}

__global__ __launch_bounds__(384) void k_pdcch_decode(const float *__restrict__ subframes, uint32_t sf_stride,
                                                      const uint32_t *__restrict__ subfr_num, const uint32_t *__restrict__ n_id_cell,
                                                      PdcchDev P, GoldTables gt, PdcchResult *__restrict__ out)
{
    __shared__ int      soft[N_CAND][2 * RE_MAX];
    __shared__ int      dbits[N_CAND][3 * 48];
    __shared__ uint64_t dec[N_CAND][48];
    __shared__ int      pc_soft[32];
    __shared__ uint32_t s_cfi, s_cell_idx;
    const uint32_t unit = blockIdx.x, wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const uint32_t sf = subfr_num[unit], cell = n_id_cell[unit];
    const float   *base = subframes + (size_t)unit * sf_stride;
    PdcchResult   *res  = out + unit;

    if (threadIdx.x == 0) {
        uint32_t ci = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < P.n_cells; k++)
            if (P.cells[k] == cell) { ci = k; break; }
        s_cell_idx = ci;
        s_cfi      = 0;
    }
    if (threadIdx.x < 12) { res->rnti[threadIdx.x] = 0; res->payload[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t ci = s_cell_idx;
    if (ci == 0xFFFFFFFFu) { // a cell the plan was not built for
        if (threadIdx.x == 0) { res->cfi = 0; res->n_symbs = 0; }
        return;
    }
    // ---- PCFICH (pcfich_channel_demap + cfi_channel_decode)
    if (wave == 0) {
        demod_res(base, P.N_ant, P.N_ant, P.per_port, P.pcfich + (size_t)ci * 16, 16, gt, (((sf + 1) * (2 * cell + 1)) << 9) + cell, 0, pc_soft, ln);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint32_t bit = 0, m[4] = {0, 0, 0, 0};
        if (ln < 32) {
            bit = pc_soft[ln] >= 0 ? 0u : 1u;
            const uint32_t r = ln % 3; // CFI code words (36.212 table 5.3.4-1): 011 011.. | 101 101.. | 110 110.. | all zero
            m[0] = ((r == 0 ? 0u : 1u) != bit); m[1] = ((r == 1 ? 0u : 1u) != bit); m[2] = ((r == 2 ? 0u : 1u) != bit); m[3] = (0u != bit);
        }
        uint32_t ber[4];
        for (int k = 0; k < 4; k++) ber[k] = (uint32_t)__popcll(__ballot(ln < 32 && m[k]));
        if (ln == 0) {
            uint32_t min_ber = 32, cfi = 0;
            for (uint32_t k = 0; k < 4; k++)
                if (ber[k] < min_ber) { min_ber = ber[k]; cfi = k + 1; }
            s_cfi = (min_ber < 4) ? cfi : 0u; // CFI_N_ACCEPTABLE_BERS (:2092)
        }
    }
    __syncthreads();
    const uint32_t cfi = s_cfi, n_symbs = cfi + (P.N_rb_dl <= 10 ? 1u : 0u);
    if (threadIdx.x == 0) { res->cfi = cfi; res->n_symbs = cfi ? n_symbs : 0; }
    if (cfi == 0 || n_symbs > 4 || wave >= N_CAND) return;

    // ---- one candidate per wave
    const uint32_t  c = wave, n_re = c < 4 ? 144u : 288u, E = 2 * n_re;
    const uint32_t *re = P.cand + (((size_t)ci * 4 + (n_symbs - 1)) * N_CAND + c) * RE_MAX;
    if (re[0] == NO_RE) return; // the candidate reaches past the last CCE
    const uint32_t c_off = c < 4 ? c * 288u : (c - 4) * 576u; // offset into the subframe's scrambling sequence (:4908, :5035)
    demod_res(base, P.N_ant, P.N_ant, P.per_port, re, n_re, gt, (sf << 9) + cell, c_off, soft[c], ln);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    for (uint32_t f = 0; f < 2; f++) {
        const uint32_t n_out = P.dci_size[f], N = n_out + 16; // information + CRC bits = trellis steps
        // rate un-matching with soft combining (rate_unmatch_conv): every received bit adds onto its d position
        for (uint32_t k = ln; k < 3 * N; k += 64) dbits[c][k] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint16_t *map = P.rm_map + ((size_t)f * 2 + (c < 4 ? 0 : 1)) * 576;
        for (uint32_t k = ln; k < E; k += 64) atomicAdd(&dbits[c][map[k]], soft[c][k]);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint32_t bits_lo, bits_hi;
        viterbi_k7(dbits[c], N, dec[c], ln, bits_lo, bits_hi);
        if (ln == 0) {
            uint32_t       payload;
            const uint32_t x = crc16_syndrome(bits_lo, bits_hi, n_out, payload); // = RNTI when the CRC matches (UE antenna mask 0)
            if (x == 0xFFFFu || x == 0xFFFEu || (x >= 1 && x <= 0x3Cu)) { // SI-, P-, RA-RNTI (:4915-4947)
                res->rnti[c * 2 + f]    = x;
                res->payload[c * 2 + f] = payload;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- PBCH (liblte_phy_bch_channel_decode, liblte_phy.cc:3968-4105; bch_channel_decode :12581-12650) ----------------------
// One workgroup per subframe 0, twelve wavefronts = {1, 2, 4 antenna ports} x {4 positions in the 40 ms BCH period}: each gathers
// the 240 PBCH resource elements (symbols 7-10, centre 72 sub-carriers, CRS positions of symbols 7, 8 left out) with the
// estimates of the ports its hypothesis needs, combines / de-maps / descrambles with the quarter of the 1920-bit sequence its
// position selects, rate-un-matches (every d position gets 4 of the 480 bits), runs the same Viterbi decoder over 40 steps and
// checks CRC16 against its port-count mask.  The reference tries the hypotheses in this order and stops at the first success:
// the lowest successful wave index wins.
struct PbchLap { uint8_t pos[120]; }; // d position (3i + x) of the k-th bit of one lap of the circular buffer (40 bits per stream)
struct PbchResult { uint32_t N_ant, offset, mib; };

__global__ __launch_bounds__(768) void k_pbch_decode(const float *__restrict__ subframes, uint32_t sf_stride, uint32_t N_rb_dl,
                                                     const uint32_t *__restrict__ n_id_cell, PbchLap lap, GoldTables gt,
                                                     PbchResult *__restrict__ out)
{
    __shared__ uint32_t re[240];
    __shared__ int      soft[12][480];
    __shared__ int      dbits[12][120];
    __shared__ uint64_t dec[12][40];
    __shared__ uint32_t s_mib[12], s_win;
    const uint32_t unit = blockIdx.x, wave = threadIdx.x >> 6, ln = threadIdx.x & 63, cell = n_id_cell[unit];
    const float   *base = subframes + (size_t)unit * sf_stride;
    if (threadIdx.x < 240) {
        const uint32_t t = threadIdx.x, k0 = 6 * N_rb_dl - 36, r = cell % 3;
        uint32_t sym, i;
        if (t < 96) { // symbols 7, 8: the two non-CRS residues of every group of three sub-carriers
            const uint32_t j = t % 48, lo = r == 0 ? 1u : 0u, hi = r == 2 ? 1u : 2u;
            sym = 7 + t / 48;
            i   = 3 * (j / 2) + ((j & 1) ? hi : lo);
        } else {
            sym = 9 + (t - 96) / 72;
            i   = (t - 96) % 72;
        }
        re[t] = sym * N_SC_MAX + k0 + i;
    }
    if (threadIdx.x == 0) s_win = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t p = wave < 4 ? 1u : wave < 8 ? 2u : 4u, off = wave & 3u;
    demod_res(base, 4, p, 1, re, 240, gt, cell, off * 480, soft[wave], ln); // the PBCH estimates are strided correctly (:4030)
    for (uint32_t k = ln; k < 120; k += 64) dbits[wave][k] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = ln; k < 480; k += 64) atomicAdd(&dbits[wave][lap.pos[k % 120]], soft[wave][k]);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    uint32_t bits_lo, bits_hi;
    viterbi_k7(dbits[wave], 40, dec[wave], ln, bits_lo, bits_hi);
    if (ln == 0) {
        uint32_t       mib;
        const uint32_t x = crc16_syndrome(bits_lo, bits_hi, 24, mib), mask = p == 1 ? 0u : p == 2 ? 0xFFFFu : 0x5555u; // 36.212 table 5.3.1.1-1
        s_mib[wave] = mib;
        if (x == mask) atomicMin(&s_win, wave);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t w = s_win;
        out[unit] = w == 0xFFFFFFFFu ? PbchResult{0, 0, 0} : PbchResult{w < 4 ? 1u : w < 8 ? 2u : 4u, w & 3u, s_mib[w]};
    }
}

// ---- host: index tables ------------------------------------------------------------------------

// positions of the PCFICH / PHICH resource-element groups in symbol 0 (pcfich_channel_demap, phich_channel_demap)
struct CtrlRegs { uint32_t pcfich_k[4]; float pcfich_n[4]; std::vector<uint32_t> phich_k; };

CtrlRegs ctrl_regs(uint32_t N_rb_dl, uint32_t cell, float phich_res)
{
    CtrlRegs r;
    const uint32_t k_hat = 6 * (cell % (2 * N_rb_dl));
    for (uint32_t i = 0; i < 4; i++) {
        r.pcfich_k[i] = (k_hat + (i * N_rb_dl / 2) * 12 / 2) % (N_rb_dl * 12);
        r.pcfich_n[i] = (r.pcfich_k[i] / 6) - 0.5;
    }
    const uint32_t N_group = (uint32_t)ceilf((float)phich_res * ((float)N_rb_dl / (float)8));
    const uint32_t n_l = N_rb_dl * 2 - 4;
    for (uint32_t m = 0; m < N_group; m++) {
        uint32_t n_hat[3];
        for (uint32_t i = 0; i < 3; i++) n_hat[i] = (cell + m + i * n_l / 3) % n_l;
        for (uint32_t i = 0; i < 4; i++)
            for (uint32_t j = 0; j < 3; j++)
                if (n_hat[j] > r.pcfich_n[i]) n_hat[j]++; // sequential, against the float REG number, as the reference does
        for (uint32_t i = 0; i < 3; i++) r.phich_k.push_back(n_hat[i] * 6);
    }
    return r;
}

// natural position -> rank in the sub-block interleaver's read-out order, dummies skipped (36.212 5.1.4.2.1; the reference
// rebuilds this per call, liblte_phy.cc:4717-4812)
std::vector<uint32_t> cc_interleaver_rank(uint32_t n)
{
    uint32_t R = 0;
    while (n > 32 * R) R++;
    const uint32_t K_pi = 32 * R, N_dummy = K_pi - n;
    std::vector<uint32_t> rank(K_pi, 0);
    uint32_t k = 0;
    for (uint32_t j = 0; j < 32; j++)     // read column by column ...
        for (uint32_t i = 0; i < R; i++) { // ... of the column-permuted matrix
            const uint32_t nat = i * 32 + LTE_SUBBLOCK_COL_PERM_CC[j];
            if (nat >= N_dummy) rank[nat] = k++;
        }
    std::vector<uint32_t> out(n);
    for (uint32_t i = 0; i < n; i++) out[i] = rank[N_dummy + i];
    return out;
}

// RE lists of the six common-search-space candidates for one (cell, N_symbs)
void candidate_res(uint32_t N_rb_dl, uint32_t N_ant, uint32_t cell, float phich_res, uint32_t N_symbs, uint32_t *out /*[N_CAND][RE_MAX]*/)
{
    std::fill(out, out + N_CAND * RE_MAX, NO_RE);
    const CtrlRegs cr = ctrl_regs(N_rb_dl, cell, phich_res);
    int64_t n_reg = (int64_t)N_symbs * (N_rb_dl * 3) - N_rb_dl - 4 - (int64_t)cr.phich_k.size();
    if (N_ant == 4) n_reg -= N_rb_dl;
    if (n_reg <= 0) return;
    const uint32_t N_reg = (uint32_t)n_reg, N_cce = N_reg / 9;
    std::vector<uint32_t> reg(4 * (size_t)N_reg, NO_RE); // REG m: its 4 RE indices (l*1200 + k)
    uint32_t m = 0;
    for (uint32_t k = 0; k < N_rb_dl * 12; k++)            // Steps 1-10 of 36.211 6.8.5 as the reference walks them
        for (uint32_t l = 0; l < N_symbs; l++) {
            if (m >= N_reg) continue;
            if (l == 0 || (l == 1 && N_ant == 4)) {
                bool valid = (k % 6) == 0;
                if (l == 0) {
                    for (uint32_t i = 0; i < 4; i++) valid &= k != cr.pcfich_k[i];
                    for (uint32_t kk : cr.phich_k) valid &= k != kk;
                }
                if (!valid) continue;
                uint32_t idx = 0;
                for (uint32_t i = 0; i < 6; i++)
                    if ((cell % 3) != (i % 3)) reg[4 * m + idx++] = l * N_SC_MAX + k + i; // skip the CRS positions
                m++;
            } else if ((k % 4) == 0) {
                for (uint32_t i = 0; i < 4; i++) reg[4 * m + i] = l * N_SC_MAX + k + i;
                m++;
            }
        }
    // undo the cell-specific cyclic shift, then the sub-block interleaver
    std::vector<uint32_t> shifted(4 * (size_t)N_reg), perm(4 * (size_t)N_reg);
    for (uint32_t i = 0; i < N_reg; i++) std::copy(&reg[4 * i], &reg[4 * i] + 4, &shifted[4 * ((i + cell) % N_reg)]);
    const std::vector<uint32_t> rank = cc_interleaver_rank(N_reg);
    for (uint32_t i = 0; i < N_reg; i++) std::copy(&shifted[4 * rank[i]], &shifted[4 * rank[i]] + 4, &perm[4 * i]);
    for (uint32_t c = 0; c < N_CAND; c++) {
        // the reference always tries all six candidates; CCEs past the last one hold whatever its scratch held before --
        // zeros on a fresh LIBLTE_PHY_STRUCT, which de-map to soft value 0.  Here those elements are marked absent and
        // contribute 0 (the only state-dependent corner of the reference's decoder)
        const uint32_t L = c < 4 ? 4 : 8, first = (c < 4 ? c : c - 4) * L;
        for (uint32_t j = 0; j < L * 36; j++)
            if (first + j / 36 < N_cce) out[c * RE_MAX + j] = perm[(size_t)(first + j / 36) * 36 + j % 36];
    }
}

// where received bit k of E lands in d[3*N_c] (rate_unmatch_conv :11597-11755: three sub-block-interleaved streams back to
// back, circular read from position 0, dummies skipped)
void conv_rm_map(uint32_t N_c, uint32_t E, uint16_t *map)
{
    uint32_t R = 0;
    while (N_c > 32 * R) R++;
    const uint32_t K_pi = 32 * R, K_w = 3 * K_pi, N_dummy = K_pi - N_c;
    for (uint32_t k = 0, j = 0; k < E; j++) {
        const uint32_t w = j % K_w, x = w / K_pi, pos = w % K_pi, col = pos / R, row = pos % R;
        const uint32_t nat = row * 32 + LTE_SUBBLOCK_COL_PERM_CC[col];
        if (nat >= N_dummy) map[k++] = (uint16_t)((nat - N_dummy) * 3 + x);
    }
}

void dci_sizes(uint32_t N_rb_dl, uint32_t *s1a, uint32_t *s1c) // liblte_phy.cc:4835-4857
{
    switch (N_rb_dl) {
    case 6:  *s1a = 21; *s1c = 9;  break;
    case 15: *s1a = 22; *s1c = 11; break;
    case 25: *s1a = 25; *s1c = 13; break;
    case 50: *s1a = 27; *s1c = 13; break;
    case 75: *s1a = 27; *s1c = 14; break;
    default: *s1a = 28; *s1c = 15; break;
    }
}

} // namespace

struct mi_lte_pdcch_plan {
    mi_lte_dl_cfg cfg;
    PdcchDev      dev{};
    std::vector<void *> owned;
};

extern "C" {

// the six LTE bandwidths and a PHICH resource N_g in (0, 2] (36.211 6.9: 1/6, 1/2, 1, 2): what bounds the REG tables below (a negative or
// not-a-number N_g would size them by a wrapped group count)
static bool standard_ctrl_cfg(uint32_t nrb, float phich_res)
{
    return (nrb == 6 || nrb == 15 || nrb == 25 || nrb == 50 || nrb == 75 || nrb == 100) && phich_res > 0.0f && phich_res <= 2.0f;
}

int mi_lte_pdcch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, float phich_res, uint32_t phich_dur_extended, uint32_t flags,
                             const uint32_t *h_cells, uint32_t n_cells, mi_lte_pdcch_plan **out)
{
    if (!ctx || !cfg || !h_cells || n_cells == 0 || !out) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t nrb = cfg->N_rb_dl;
    if (!standard_ctrl_cfg(nrb, phich_res) || !(cfg->N_ant == 1 || cfg->N_ant == 2 || cfg->N_ant == 4) ||
        phich_dur_extended) { // the reference does not handle the extended PHICH duration either (:8280-8283)
        ctx->err = "PDCCH plan: standard bandwidths, 1/2/4 ports, PHICH resource in (0, 2], normal PHICH duration";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl = new mi_lte_pdcch_plan();
    auto  guard = on_fail([&] { (void)hipStreamSynchronize(ctx->stream); mi_lte_pdcch_plan_destroy(nullptr, pl); });
    pl->cfg  = *cfg;
    std::vector<uint32_t> cells(h_cells, h_cells + n_cells), pcf((size_t)n_cells * 16), cand((size_t)n_cells * 4 * N_CAND * RE_MAX);
    for (uint32_t c = 0; c < n_cells; c++) {
        if (cells[c] > 503) return MI_LTE_ERR_INVALID_ARG;
        for (uint32_t ns = 1; ns <= 4; ns++)
            mi_lte_pdcch_re_tables(nrb, cfg->N_ant, cells[c], phich_res, ns, &pcf[(size_t)c * 16], &cand[((size_t)c * 4 + ns - 1) * N_CAND * RE_MAX]);
    }
    uint32_t sz[2];
    dci_sizes(nrb, &sz[0], &sz[1]);
    std::vector<uint16_t> rm((size_t)2 * 2 * 576, 0);
    for (uint32_t f = 0; f < 2; f++)
        for (uint32_t e = 0; e < 2; e++) conv_rm_map(sz[f] + 16, e ? 576 : 288, &rm[((size_t)f * 2 + e) * 576]);
    auto up = [&](const void *h, size_t bytes, void **d) -> int {
        if (hipMalloc(d, bytes ? bytes : 4) != hipSuccess) return -1;
        pl->owned.push_back(*d);
        return mi_lte_memcpy_h2d(ctx, *d, h, bytes) == MI_LTE_OK ? 0 : -1;
    };
    void *d_cells, *d_pcf, *d_cand, *d_rm;
    if (up(cells.data(), cells.size() * 4, &d_cells) || up(pcf.data(), pcf.size() * 4, &d_pcf) || up(cand.data(), cand.size() * 4, &d_cand) ||
        up(rm.data(), rm.size() * 2, &d_rm)) {
        ctx->err = "PDCCH plan: device allocation failed";
        return MI_LTE_ERR_NOMEM;
    }
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    pl->dev = PdcchDev{nrb, cfg->N_ant, n_cells, {sz[0], sz[1]}, (flags & MI_LTE_PDCCH_PER_PORT_ESTIMATES) ? 1u : 0u, (const uint32_t *)d_cells, (const uint32_t *)d_pcf, (const uint32_t *)d_cand,
                       (const uint16_t *)d_rm};
    guard.armed = false;
    *out = pl;
    return MI_LTE_OK;
}

// the index tables on their own (host arithmetic): where the PCFICH and the six candidates live in the subframe grid
int mi_lte_pdcch_re_tables(uint32_t N_rb_dl, uint32_t N_ant, uint32_t N_id_cell, float phich_res, uint32_t N_symbs, uint32_t *pcfich /*[16]*/,
                           uint32_t *cand /*[6][288]*/)
{
    if (!pcfich || !cand || !standard_ctrl_cfg(N_rb_dl, phich_res) || !(N_ant == 1 || N_ant == 2 || N_ant == 4) || N_id_cell > 503 || N_symbs < 1 || N_symbs > 4)
        return MI_LTE_ERR_INVALID_ARG;
    const CtrlRegs cr = ctrl_regs(N_rb_dl, N_id_cell, phich_res);
    for (uint32_t i = 0; i < 4; i++) {
        uint32_t idx = 0;
        for (uint32_t j = 0; j < 6; j++)
            if ((N_id_cell % 3) != (j % 3)) pcfich[i * 4 + idx++] = cr.pcfich_k[i] + j;
    }
    candidate_res(N_rb_dl, N_ant, N_id_cell, phich_res, N_symbs, cand);
    return MI_LTE_OK;
}

// PCFICH / PHICH resource-element-group positions as the reference reports them in LIBLTE_PHY_PCFICH_STRUCT::k, n and
// LIBLTE_PHY_PHICH_STRUCT::k, N_reg (pcfich_channel_demap :7903-7910, phich_channel_demap :8243-8278)
int mi_lte_ctrl_reg_positions(uint32_t N_rb_dl, uint32_t N_id_cell, float phich_res, uint32_t *pcfich_k /*[4]*/, float *pcfich_n /*[4]*/,
                              uint32_t *phich_N_reg, uint32_t *phich_k /*[75]*/)
{
    if (!pcfich_k || !pcfich_n || !phich_N_reg || !phich_k || !standard_ctrl_cfg(N_rb_dl, phich_res) || N_id_cell > 503) return MI_LTE_ERR_INVALID_ARG;
    const CtrlRegs cr = ctrl_regs(N_rb_dl, N_id_cell, phich_res);
    if (cr.phich_k.size() > 75) return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < 4; i++) { pcfich_k[i] = cr.pcfich_k[i]; pcfich_n[i] = cr.pcfich_n[i]; }
    *phich_N_reg = (uint32_t)cr.phich_k.size();
    std::copy(cr.phich_k.begin(), cr.phich_k.end(), phich_k);
    return MI_LTE_OK;
}

void mi_lte_pdcch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pdcch_plan *pl)
{
    if (!pl) return;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    for (void *p : pl->owned) (void)hipFree(p);
    delete pl;
}

static bool common_rnti(uint32_t rnti) { return rnti == 0xFFFFu || rnti == 0xFFFEu || (rnti >= 1 && rnti <= 0x3Cu); }

// DCI format 1A for SI-/P-/RA-RNTI -> allocation (dci_1a_unpack, liblte_phy.cc:13273-13378; localized VRBs only, like the
// reference: a distributed assignment leaves prb[][] untouched).  Returns 0, or 4 = LIBLTE_ERROR_INVALID_CONTENTS.
int mi_lte_dci_1a_unpack(uint32_t payload, uint32_t n_bits, uint32_t rnti, uint32_t N_rb_dl, uint32_t N_ant, mi_lte_pdcch_dci *o)
{
    if (!o || n_bits > 32 || N_rb_dl == 0 || N_rb_dl > 110) return MI_LTE_ERR_INVALID_ARG; // (110: LIBLTE_PHY_N_RB_DL_MAX)
    uint32_t pos = n_bits;
    auto take = [&](uint32_t n) { pos = pos >= n ? pos - n : 0; return (payload >> pos) & ((1u << n) - 1u); };
    if (take(1) == 0) return 4;                   // flagged as DCI format 0
    if (!common_rnti(rnti)) return 0;             // the reference leaves the allocation alone and still reports success
    const uint32_t distributed = take(1);
    const uint32_t riv_len = (uint32_t)ceilf(logf(N_rb_dl * (N_rb_dl + 1) / 2) / logf(2));
    const uint32_t riv = take(riv_len);
    o->alloc.N_prb = riv / N_rb_dl + 1;
    const uint32_t rb_start = riv % N_rb_dl;
    o->mcs = take(5);
    take(3);                                      // HARQ process
    take(1);                                      // new-data indicator
    o->alloc.rv_idx = take(2);
    const uint32_t tpc = take(2), col = (tpc % 2) == 0 ? 0 : 1; // N_PRB^1A = 2 or 3
    if (!distributed)
        for (uint32_t i = 0; i < o->alloc.N_prb && i < 112; i++) o->alloc.prb[0][i] = o->alloc.prb[1][i] = (uint8_t)(rb_start + i);
    o->alloc.mod_type = 1;                        // QPSK
    o->alloc.tx_mode  = N_ant == 1 ? 1 : 2;
    o->alloc.rnti     = rnti;
    if (o->mcs >= 27) return 4;
    o->alloc.tbs = LTE_TBS_NPRB_2_3[o->mcs][col];
    return 0;
}

// DCI format 1C -> allocation (dci_1c_unpack, liblte_phy.cc:13400-13611): distributed VRBs of 36.211 6.2.3.2 and the
// format-1C transport-block sizes.  A RIV that matches no (length, start) pair leaves N_prb as the caller set it.
int mi_lte_dci_1c_unpack(uint32_t payload, uint32_t n_bits, uint32_t rnti, uint32_t N_rb_dl, uint32_t N_ant, mi_lte_pdcch_dci *o)
{
    if (!o || n_bits > 32 || N_rb_dl < 6 || N_rb_dl > 110) return MI_LTE_ERR_INVALID_ARG;
    uint32_t pos = n_bits;
    auto take = [&](uint32_t n) { pos = pos >= n ? pos - n : 0; return (payload >> pos) & ((1u << n) - 1u); };
    const uint32_t gap2 = N_rb_dl < 50 ? 0u : take(1);
    uint32_t N_gap;
    if (N_rb_dl <= 10) N_gap = (uint32_t)ceilf(N_rb_dl / 2.0);
    else if (N_rb_dl == 11) N_gap = 4;
    else if (N_rb_dl <= 19) N_gap = 8;
    else if (N_rb_dl <= 26) N_gap = 12;
    else if (N_rb_dl <= 44) N_gap = 18;
    else if (N_rb_dl <= 49) N_gap = 27;
    else if (N_rb_dl <= 63) N_gap = gap2 ? 9 : 27;
    else if (N_rb_dl <= 70) N_gap = gap2 ? 16 : 32;
    else N_gap = gap2 ? 16 : 48;
    const uint32_t n_vrb_gap1 = 2 * std::min(N_gap, N_rb_dl - N_gap), n_vrb_gap2 = (N_rb_dl / (2 * N_gap)) * 2 * N_gap;
    const uint32_t n_vrb = gap2 ? n_vrb_gap2 : n_vrb_gap1, step = N_rb_dl <= 49 ? 2 : 4, n_tilde = gap2 ? 2 * N_gap : n_vrb;
    if (!common_rnti(rnti)) return 4;
    const uint32_t q = n_vrb_gap1 / step, riv_len = (uint32_t)ceilf(logf(q * (q + 1) / 2.0) / logf(2));
    const uint32_t riv = take(riv_len), nd = n_vrb / step;
    uint32_t rb_start = 0;
    for (uint32_t len = 1; len <= nd; len++)      // every (length, start) pair is tried and the last match stands
        for (uint32_t st = 0; st + 1 <= nd; st++) {
            const uint32_t code = (len - 1) <= nd / 2 ? nd * (len - 1) + st : nd * (nd - len + 1) + (nd - 1 - st);
            if (riv == code) { o->alloc.N_prb = len * step; rb_start = st * step; }
        }
    o->mcs = take(5);
    const uint32_t P = N_rb_dl <= 10 ? 1 : N_rb_dl <= 26 ? 2 : N_rb_dl <= 63 ? 3 : 4;
    const uint32_t n_row = (uint32_t)(ceilf(n_tilde / (4.0 * P)) * P), n_null = 4 * n_row - n_tilde;
    for (uint32_t i = 0, v = rb_start; i < o->alloc.N_prb && i < 112; i++, v++) {
        const uint32_t vt = v % n_tilde, blk = n_tilde * (v / n_tilde);
        const uint32_t two_col = 2 * n_row * (vt % 2) + vt / 2 + blk, four_col = n_row * (vt % 4) + vt / 4 + blk;
        uint32_t even;
        if (n_null != 0 && vt >= n_tilde - n_null && (vt % 2) == 1) even = two_col - n_row;
        else if (n_null != 0 && vt >= n_tilde - n_null && (vt % 2) == 0) even = two_col - n_row + n_null / 2;
        else if (n_null != 0 && vt < n_tilde - n_null && (vt % 4) >= 2) even = four_col - n_null / 2;
        else even = four_col;
        const uint32_t odd = (even + n_tilde / 2) % n_tilde + blk;
        o->alloc.prb[0][i] = (uint8_t)(even < n_tilde / 2 ? even : even + N_gap - n_tilde / 2);
        o->alloc.prb[1][i] = (uint8_t)(odd < n_tilde / 2 ? odd : odd + N_gap - n_tilde / 2);
    }
    o->alloc.mod_type = 1;
    o->alloc.tx_mode  = N_ant == 1 ? 1 : 2;
    o->alloc.rnti     = rnti;
    o->alloc.tbs      = LTE_TBS_DCI_1C[o->mcs & 31]; // a 5-bit field: always in range
    return 0;
}

int mi_lte_pdcch_decode_run(mi_lte_ctx *ctx, mi_lte_pdcch_plan *pl, const float *d_subframes, const uint32_t *d_subfr_num,
                            const uint32_t *d_n_id_cell, uint32_t n_units, uint32_t *h_rc, uint32_t *h_cfi, uint32_t *h_n_symbs, uint32_t *h_n_dci,
                            mi_lte_pdcch_dci *h_dci /*[n_units][MI_LTE_PDCCH_MAX_DCI]*/)
{
    if (!ctx || !pl || !d_subframes || !d_subfr_num || !d_n_id_cell || n_units == 0 || !h_rc || !h_cfi || !h_n_symbs || !h_n_dci || !h_dci)
        return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    // a few units' results go straight into pinned host memory; a batch's through scratch and one copy
    const size_t res_bytes = sizeof(PdcchResult) * (size_t)n_units;
    PdcchResult *d_res, *h_res = nullptr;
    if (mi_ctx_small_results(ctx, res_bytes, (void **)&h_res, (void **)&d_res) != MI_LTE_OK) {
        h_res = nullptr;
        rc = mi_ctx_reserve_scratch(ctx, res_bytes);
        if (rc != MI_LTE_OK) return rc;
        d_res = (PdcchResult *)ctx->scratch;
    }
    GoldTables   gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    MI_LAUNCH(ctx, "k_pdcch_decode", k_pdcch_decode, dim3(n_units), dim3(384), 0, d_subframes, (uint32_t)mi_lte_subframe_floats(pl->cfg.N_ant),
              d_subfr_num, d_n_id_cell, pl->dev, gt, d_res);
    MI_HIP_CHECK(ctx, hipGetLastError());
    std::vector<PdcchResult> res_copy;
    if (!h_res) {
        res_copy.resize(n_units);
        MI_D2H(ctx, res_copy.data(), d_res, res_bytes);
    }
    MI_HIP_CHECK(ctx, h_res ? mi_stream_wait(ctx, n_units) : hipStreamSynchronize(ctx->stream)); // (a copy into pageable memory is done when the runtime says so)
    const PdcchResult *res = h_res ? h_res : res_copy.data();
    for (uint32_t u = 0; u < n_units; u++) {
        h_cfi[u]     = res[u].cfi;
        h_n_symbs[u] = res[u].n_symbs;
        uint32_t n = 0, rc = res[u].cfi ? 1u : 3u; // INVALID_INPUTS until an unpacker ran; INVALID_CRC: no CFI (:4519-4571)
        mi_lte_pdcch_dci *o = h_dci + (size_t)u * MI_LTE_PDCCH_MAX_DCI;
        // the reference's order: aggregation-4 candidates 0..3 (1A then 1C each), then aggregation-8 candidates 0, 1; at most
        // LIBLTE_PHY_PDCCH_MAX_ALLOC = 6 allocations; a 1A that does not unpack is not counted (:4948-4960)
        for (uint32_t slot = 0; slot < 12 && n < MI_LTE_PDCCH_MAX_DCI; slot++) {
            if (res[u].rnti[slot] == 0) continue;
            mi_lte_pdcch_dci d;
            memset(&d, 0, sizeof(d));
            d.rnti = res[u].rnti[slot]; d.format = slot & 1; d.candidate = slot >> 1; d.n_bits = pl->dev.dci_size[slot & 1];
            d.payload = res[u].payload[slot];
            d.alloc.unit = u;
            const int urc = d.format == 0 ? mi_lte_dci_1a_unpack(d.payload, d.n_bits, d.rnti, pl->dev.N_rb_dl, pl->dev.N_ant, &d)
                                          : mi_lte_dci_1c_unpack(d.payload, d.n_bits, d.rnti, pl->dev.N_rb_dl, pl->dev.N_ant, &d);
            rc = urc == 0 ? 0u : 4u;  // the reference returns whatever its last unpacker call returned
            if (urc != 0) continue; // not counted (:4948-4960, :4989-5000)
            d.alloc_valid = 1;
            d.alloc.n_pdcch_symbs = res[u].n_symbs; // the allocation can go into a PDSCH plan as it is
            o[n++] = d;
        }
        h_n_dci[u] = n;
        h_rc[u]    = rc;
    }
    ctx->last_kernels = "k_pdcch_decode:1";
    return MI_LTE_OK;
}


// liblte_phy_bch_channel_decode for a batch of device subframes (subframe 0 of a frame, estimated for 4 ports as the reference's
// callers do, LTE_fdd_dl_fs_samp_buf.cc:395-410): h_N_ant[u] = 0 is LIBLTE_ERROR_DECODE_FAIL
int mi_lte_pbch_decode_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const float *d_subframes, const uint32_t *d_n_id_cell, uint32_t n_units,
                           uint32_t *h_N_ant, uint32_t *h_offset, uint32_t *h_mib)
{
    if (!ctx || !cfg || !d_subframes || !d_n_id_cell || n_units == 0 || !h_N_ant || !h_offset || !h_mib) return MI_LTE_ERR_INVALID_ARG;
    if (cfg->N_ant != 4 || cfg->N_rb_dl < 6 || cfg->N_rb_dl > 100) {
        ctx->err = "PBCH decode reads the estimates of all four ports: the device subframes must be laid out with N_ant = 4";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    const size_t res_bytes = sizeof(PbchResult) * (size_t)n_units;
    PbchResult  *d_res, *h_res = nullptr;
    if (mi_ctx_small_results(ctx, res_bytes, (void **)&h_res, (void **)&d_res) != MI_LTE_OK) {
        h_res = nullptr;
        rc = mi_ctx_reserve_scratch(ctx, res_bytes);
        if (rc != MI_LTE_OK) return rc;
        d_res = (PbchResult *)ctx->scratch;
    }
    PbchLap  lap;
    uint16_t map[120];
    conv_rm_map(40, 120, map);
    for (uint32_t k = 0; k < 120; k++) lap.pos[k] = (uint8_t)map[k];
    GoldTables  gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    MI_LAUNCH(ctx, "k_pbch_decode", k_pbch_decode, dim3(n_units), dim3(768), 0, d_subframes, (uint32_t)mi_lte_subframe_floats(4), cfg->N_rb_dl, d_n_id_cell,
              lap, gt, d_res);
    MI_HIP_CHECK(ctx, hipGetLastError());
    std::vector<PbchResult> res_copy;
    if (!h_res) {
        res_copy.resize(n_units);
        MI_D2H(ctx, res_copy.data(), d_res, res_bytes);
    }
    MI_HIP_CHECK(ctx, h_res ? mi_stream_wait(ctx, n_units) : hipStreamSynchronize(ctx->stream));
    const PbchResult *res = h_res ? h_res : res_copy.data();
    for (uint32_t u = 0; u < n_units; u++) { h_N_ant[u] = res[u].N_ant; h_offset[u] = res[u].offset; h_mib[u] = res[u].mib; }
    ctx->last_kernels = "k_pbch_decode:1";
    return MI_LTE_OK;
}

// ---- input synthesis (benchmark and tests): control regions as 36.211 / 36.212 transmit them --------------------------
// PCFICH + up to four format-1A DCIs at aggregation level 4 in the common search space (what the reference's own
// transmitter sends, liblte_phy.cc:4113-4330), through a smooth random channel per port + AWGN, directly as device-subframe
// grids with noisy channel estimates.  Transmit diversity follows 36.211 6.3.4.3 on all ports.
int mi_lte_synth_ctrl_grids(const mi_lte_dl_cfg *cfg, float phich_res, uint32_t n_units, const uint32_t *h_subfr_num, const uint32_t *h_n_id_cell,
                            const uint32_t *h_cfi, const mi_lte_synth_dci *h_dci /*[n_units][n_dci]*/, uint32_t n_dci,
                            const mi_lte_synth_channel *chan, float *h_grids)
{
    if (!cfg || !h_subfr_num || !h_n_id_cell || !h_cfi || (n_dci && !h_dci) || n_dci > 4 || !chan || !h_grids) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t nrb = cfg->N_rb_dl, n_ant = cfg->N_ant;
    if (!(n_ant == 1 || n_ant == 2 || n_ant == 4)) return MI_LTE_ERR_INVALID_ARG;
    uint32_t sz[2];
    dci_sizes(nrb, &sz[0], &sz[1]);
    const uint32_t K = sz[0] + 16;
    std::vector<uint16_t> map(576);
    conv_rm_map(K, 288, map.data());
    const size_t plane = 16 * (size_t)N_SC_MAX, nf = (2 + 2 * (size_t)n_ant) * plane;
    const double r2 = 1.0 / std::sqrt(2.0);
    synth::Rng   rng(chan->seed);
    std::vector<double> tr(n_ant * plane), ti(n_ant * plane);
    std::vector<uint32_t> cand(N_CAND * RE_MAX);
    uint32_t pcf[16];
    for (uint32_t u = 0; u < n_units; u++) {
        const uint32_t sf = h_subfr_num[u], cell = h_n_id_cell[u], cfi = h_cfi[u], n_symbs = cfi + (nrb <= 10 ? 1 : 0);
        if (cfi < 1 || cfi > 4 || n_symbs > 4 || cell > 503 || sf > 9) return MI_LTE_ERR_INVALID_ARG;
        int rc = mi_lte_pdcch_re_tables(nrb, n_ant, cell, phich_res, n_symbs, pcf, cand.data());
        if (rc != MI_LTE_OK) return rc;
        std::fill(tr.begin(), tr.end(), 0.0);
        std::fill(ti.begin(), ti.end(), 0.0);
        // d[0..n): QPSK symbols of scrambled bits b -> transmit-diversity pre-coding onto the elements pos[]
        auto place = [&](const uint32_t *pos, const uint8_t *b, uint32_t n) {
            std::vector<double> dr(n), di(n);
            for (uint32_t i = 0; i < n; i++) { dr[i] = (1 - 2 * (int)b[2 * i]) * r2; di[i] = (1 - 2 * (int)b[2 * i + 1]) * r2; }
            auto put = [&](uint32_t port, uint32_t p, double re, double im) { tr[port * plane + p] = re; ti[port * plane + p] = im; };
            if (n_ant == 1) {
                for (uint32_t i = 0; i < n; i++) put(0, pos[i], dr[i], di[i]);
            } else {
                for (uint32_t i = 0; i + 1 < n; i += 2) { // pairs (x0, x1): port a sends x0, x1; port b sends -x1*, x0*
                    const uint32_t a = n_ant == 2 ? 0 : ((i & 2) ? 1 : 0), bb = n_ant == 2 ? 1 : ((i & 2) ? 3 : 2);
                    put(a, pos[i], r2 * dr[i], r2 * di[i]);
                    put(a, pos[i + 1], r2 * dr[i + 1], r2 * di[i + 1]);
                    put(bb, pos[i], -r2 * dr[i + 1], r2 * di[i + 1]);
                    put(bb, pos[i + 1], r2 * dr[i], -r2 * di[i]);
                }
            }
        };
        { // PCFICH (36.212 5.3.4, 36.211 6.7)
            uint8_t b[32], c[32];
            synth::gold((((sf + 1) * (2 * cell + 1)) << 9) + cell, 32, c);
            for (uint32_t i = 0; i < 32; i++) b[i] = (uint8_t)((cfi == 4 ? 0u : ((i % 3) == cfi - 1 ? 0u : 1u)) ^ c[i]);
            place(pcf, b, 16);
        }
        std::vector<uint8_t> c(4 * 288);
        synth::gold((sf << 9) + cell, 4 * 288, c.data());
        for (uint32_t a = 0; a < n_dci; a++) {
            const mi_lte_synth_dci &d = h_dci[(size_t)u * n_dci + a];
            if (d.rnti == 0) continue; // unused slot
            if (cand[a * RE_MAX + 143] == NO_RE) continue; // this candidate's CCEs do not all exist: nothing is sent (as the reference)
            // DCI format 1A for SI-/P-/RA-RNTI (36.212 5.3.3.1.3; the reference's dci_1a_pack :13138-13215)
            std::vector<uint8_t> bits(K, 0);
            uint32_t             pos = 0;
            auto push = [&](uint32_t v, uint32_t n) { for (uint32_t i = 0; i < n; i++) bits[pos++] = (uint8_t)((v >> (n - 1 - i)) & 1u); };
            const uint32_t riv_len = (uint32_t)ceilf(logf(nrb * (nrb + 1) / 2) / logf(2));
            if (d.N_prb == 0 || d.N_prb - 1 > nrb / 2 || d.rb_start + d.N_prb > nrb || d.mcs > 26) return MI_LTE_ERR_INVALID_ARG;
            push(1, 1); push(0, 1); push(nrb * (d.N_prb - 1) + d.rb_start, riv_len); push(d.mcs, 5); push(0, 3); push(0, 1); push(d.rv_idx, 2); push(1, 2);
            // CRC16 masked with the RNTI (36.212 5.3.3.2)
            uint32_t rem = 0;
            for (uint32_t t = 0; t < sz[0] + 16; t++) {
                rem = (rem << 1) | (t < sz[0] ? bits[t] : 0u);
                if (rem & 0x10000u) rem ^= 0x11021u;
            }
            rem ^= d.rnti & 0xFFFFu;
            for (uint32_t i = 0; i < 16; i++) bits[sz[0] + i] = (uint8_t)((rem >> (15 - i)) & 1u);
            // tail-biting convolutional code, K = 7, rate 1/3 (36.212 5.1.3.1)
            std::vector<uint8_t> dd(3 * K);
            uint32_t             state = 0;
            for (uint32_t i = 0; i < 6; i++) state |= (uint32_t)bits[K - 1 - i] << (5 - i); // most recent input in bit 5
            const uint32_t G[3] = {0133, 0171, 0165};
            for (uint32_t t = 0; t < K; t++) {
                const uint32_t reg = ((uint32_t)bits[t] << 6) | state;
                for (uint32_t o = 0; o < 3; o++) dd[3 * t + o] = (uint8_t)(__builtin_popcount(reg & G[o]) & 1);
                state = reg >> 1;
            }
            // rate matching to E = 288 (aggregation level 4) and scrambling at the candidate's offset
            uint8_t e[288];
            for (uint32_t k = 0; k < 288; k++) e[k] = dd[map[k]] ^ c[a * 288 + k];
            place(&cand[a * RE_MAX], e, 144);
        }
        // channel + noise -> rx grid and estimates (symbols 0..3 only: the control region)
        float *g = h_grids + (size_t)u * nf;
        std::fill(g, g + nf, 0.0f);
        const double sig = chan->snr_db >= 200 ? 0.0 : std::pow(10.0, -chan->snr_db / 20.0) * r2;
        std::vector<double> rr(4 * N_SC_MAX, 0.0), ri(4 * N_SC_MAX, 0.0);
        for (uint32_t p = 0; p < n_ant; p++) {
            const double amp = chan->gain_min + (chan->gain_max - chan->gain_min) * rng.uniform();
            const double tau = (2 * rng.uniform() - 1) * 2e-3, ph = (2 * rng.uniform() - 1) * M_PI;
            for (uint32_t l = 0; l < 4; l++)
                for (uint32_t k = 0; k < 12 * nrb; k++) {
                    const double hr = amp * std::cos(ph + 2 * M_PI * tau * k), hi = amp * std::sin(ph + 2 * M_PI * tau * k);
                    const size_t q = l * N_SC_MAX + k;
                    rr[q] += hr * tr[p * plane + q] - hi * ti[p * plane + q];
                    ri[q] += hr * ti[p * plane + q] + hi * tr[p * plane + q];
                    g[(2 + p) * plane + q]         = (float)(hr + 0.2 * sig * rng.normal());
                    g[(2 + n_ant + p) * plane + q] = (float)(hi + 0.2 * sig * rng.normal());
                }
        }
        for (uint32_t l = 0; l < 4; l++)
            for (uint32_t k = 0; k < 12 * nrb; k++) {
                const size_t q = l * N_SC_MAX + k;
                g[q]         = (float)(rr[q] + sig * rng.normal());
                g[plane + q] = (float)(ri[q] + sig * rng.normal());
            }
    }
    return MI_LTE_OK;
}

} // extern "C"
