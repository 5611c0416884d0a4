// The TRANSMIT side of liblte_phy on the host: the ten functions LTE_fdd_enodeb's PHY and LTE_fdd_dl_file_gen call next to the receive
// chains (LTE_fdd_enodeb/src/LTE_fdd_enb_phy.cc:557-770, LTE_fdd_dl_file_gen/src/LTE_fdd_dl_fg_samp_buf.cc:269-668) --
//
//   liblte_phy_rate_match_turbo       liblte_phy.cc:11081-11237      liblte_phy_map_crs            :5144-5263
//   liblte_phy_pdsch_channel_encode   :3489-3688                     liblte_phy_map_pss            :5265-5304
//   liblte_phy_bch_channel_encode     :3863-3966                     liblte_phy_map_sss            :5520-5576
//   liblte_phy_pdcch_channel_encode   :4113-4517                     liblte_phy_create_dl_subframe :5862-5903
//   liblte_phy_pusch_channel_encode   :2664-2799                     liblte_phy_generate_prach     :3219-3297
//
// SURVEY 2 lists them as CPU pass-through: a transmitter runs once per TTI on a few kilobits, there is nothing to batch, and nothing here
// touches the device.  They exist so that a link against libmi_lte.so + the shim (-DMI_LTE_SHIM_OWN_LIFECYCLE) needs NO object of the
// reference's PHY (shim/liblte_phy_shim.cc).  Written from 36.211 / 36.212 v10.1.0 and the reference's observable behaviour: bit outputs are
// the reference's bit for bit, float outputs are produced by the same float operations in the same order (constants rounded to float where the
// reference rounds them, libm calls on the argument types its expressions have), the transforms (no FFTW here) in float64 rounded to float.
// Where the reference reads scratch that an earlier call left behind (a resource count that disagrees with the mapping loop, the two symbols
// past an odd quadruple on four ports, code blocks after the first) the handle keeps the same scratch with the same lifetime, starting from
// zeros as a freshly mapped LIBLTE_PHY_STRUCT does.  shim/lifecycle_check.cc (`tx`) compares every function with the compiled reference.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mi_lte.h"
#include "synth.hpp"
#include "tx_host.h"

namespace tx {

// ---------------------------------------------------------------------------------------------------------------- bit-level coding
// 36.212 5.1.1: remainder of a(x) * x^L by the generator, MSB first (calc_crc, liblte_phy.cc:9713-9743); poly carries its x^L term
void crc_bits(const uint8_t *a, uint32_t n, uint32_t poly, uint32_t L, uint8_t *p)
{
    uint32_t rem = 0;
    for (uint32_t i = 0; i < n + L; i++) {
        rem = (rem << 1) | (i < n ? a[i] : 0);
        if (rem >> L & 1) rem ^= poly;
    }
    for (uint32_t i = 0; i < L; i++) p[i] = (rem >> (L - 1 - i)) & 1;
}

// one constituent encoder of 36.212 5.1.3.2.1 with its termination: z = parity, fb = the feedback bit of every step (the termination's
// systematic outputs are feedback bits); an input that is not 0 / 1 -- the <NULL> filler, 100 -- counts by its parity (turbo_constituent_encoder,
// liblte_phy.cc:10855-10924)
static void rsc_encode(const uint8_t *in, uint32_t K, uint8_t *z, uint8_t *fb)
{
    uint32_t r1 = 0, r2 = 0, r3 = 0;
    for (uint32_t i = 0; i < K + 4; i++) {
        const uint32_t f = r2 ^ r3, s = i < K ? ((f + in[i]) & 1u) : 0u;
        fb[i] = (uint8_t)f;
        z[i]  = (uint8_t)(s ^ r1 ^ r3);
        r3 = r2, r2 = r1, r1 = s;
    }
}

// 36.212 5.1.3.2: d planar, three streams of K + 4 (turbo_encode, liblte_phy.cc:10541-10589; the interleaver index in uint32 arithmetic as
// :10954-10958 evaluates it).  Filler bits stay <NULL> in the systematic stream only -- the reference does not mark the parity of a filler.
void turbo_encode(const uint8_t *c, uint32_t K, uint8_t *d)
{
    std::vector<uint8_t> z(K + 4), f(K + 4), cp(K), zp(K + 4), fp(K + 4);
    uint32_t             f1 = 0, f2 = 0;
    synth::qpp_params(K, &f1, &f2);
    rsc_encode(c, K, z.data(), f.data());
    for (uint32_t i = 0; i < K; i++) cp[i] = c[(f1 * i + f2 * i * i) % K];
    rsc_encode(cp.data(), K, zp.data(), fp.data());
    const uint32_t D = K + 4;
    uint8_t       *d0 = d, *d1 = d + D, *d2 = d + 2 * D;
    for (uint32_t i = 0; i < K; i++) d0[i] = c[i], d1[i] = z[i], d2[i] = zp[i];
    // 36.212 5.1.3.2.2: x_K z_K x_K+1 z_K+1 x_K+2 z_K+2 x'_K z'_K x'_K+1 z'_K+1 x'_K+2 z'_K+2 dealt round-robin over the three streams
    const uint8_t tail[12] = {f[K], z[K], f[K + 1], z[K + 1], f[K + 2], z[K + 2], fp[K], zp[K], fp[K + 1], zp[K + 1], fp[K + 2], zp[K + 2]};
    for (uint32_t t = 0; t < 4; t++) d0[K + t] = tail[3 * t], d1[K + t] = tail[3 * t + 1], d2[K + t] = tail[3 * t + 2];
}

static inline uint32_t bitrev5(uint32_t j) { return ((j & 1) << 4) | ((j & 2) << 2) | (j & 4) | ((j & 8) >> 2) | ((j & 16) >> 4); }

// 36.212 5.1.4.1 (liblte_phy_rate_match_turbo, liblte_phy.cc:11081-11237): d planar in three streams of N_d_bits / 3
void rate_match_turbo(const uint8_t *d, uint32_t N_d_bits, uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type,
                      uint32_t rv_idx, uint32_t N_e_bits, uint8_t *e)
{
    const uint32_t D = N_d_bits / 3, R = (D + 31) / 32, K_pi = 32 * R, N_dummy = K_pi - D;
    std::vector<uint8_t> w(3 * (size_t)K_pi + 2);
    // a stream behind its dummies, read as a 32-column matrix: element t of the padded stream
    auto at = [&](uint32_t x, uint32_t t) -> uint8_t { return t < N_dummy ? TX_NULL : d[(size_t)D * x + (t - N_dummy)]; };
    for (uint32_t k = 0; k < K_pi; k++) {
        const uint32_t col = k / R, row = k % R;
        w[k]                = at(0, 32 * row + bitrev5(col));                      // v(0): columns in bit-reversed order, read down the columns
        w[K_pi + 2 * k]     = at(1, 32 * row + bitrev5(col));                      // v(1) on the even places behind it
        w[K_pi + 2 * k + 1] = at(2, (bitrev5(col) + 32 * row + 1) % K_pi);         // v(2), shifted by one, on the odd ones
    }
    const uint32_t K_w = 3 * K_pi, K_mimo = (tx_mode == 3 || tx_mode == 4 || tx_mode == 8 || tx_mode == 9) ? 2 : 1;
    if (D == 0 || M_dl_harq == 0 || N_codeblocks == 0) return; // (the reference divides by these: nothing is written instead)
    const uint32_t N_ir = N_soft / (K_mimo * (M_dl_harq < 8 ? M_dl_harq : 8));
    uint32_t       N_cb = K_w;
    if ((chan_type == MI_LTE_CHAN_DLSCH || chan_type == MI_LTE_CHAN_PCH) && N_ir / N_codeblocks < K_w) N_cb = N_ir / N_codeblocks;
    if (N_cb == 0) return; // (a soft buffer without room for a single bit: the reference takes a remainder by zero)
    const uint32_t k_0 = R * (2 * (uint32_t)ceilf((float)N_cb / (float)(8 * R)) * rv_idx + 2);
    bool any = false; // (a window of the circular buffer that holds nothing but <NULL>s would be walked for ever)
    for (uint32_t j = 0; j < N_cb && !any; j++) any = w[j] != TX_NULL;
    if (!any) return;
    for (uint32_t k = 0, j = 0; k < N_e_bits; j++) {
        const uint8_t b = w[(k_0 + j) % N_cb];
        if (b != TX_NULL) e[k++] = b;
    }
}

// 36.212 5.1.3.1, constraint length 7, rate 1/3, tail biting: d[3 i + x] (conv_encode with tail_bit, liblte_phy.cc:9997-10059; generators
// 133 / 171 / 165 octal as every caller passes them)
void conv_encode_tb(const uint8_t *c, uint32_t n, uint8_t *d)
{
    static const uint32_t g[3] = {0133, 0171, 0165};
    uint32_t              reg  = 0; // bit 6 = newest
    for (uint32_t i = 0; i < 7; i++) reg |= (uint32_t)(c[n - 1 - i] & 1u) << (6 - i);
    for (uint32_t i = 0; i < n; i++) {
        reg = (reg >> 1) | ((uint32_t)(c[i] & 1u) << 6);
        for (uint32_t x = 0; x < 3; x++) d[3 * i + x] = (uint8_t)(__builtin_popcount(reg & g[x]) & 1);
    }
}

// 36.212 5.1.4.2 (rate_match_conv, liblte_phy.cc:11499-11588): d interleaved d[3 i + x]; the column order is 5.1.4-2's = the turbo one turned by 16
void rate_match_conv(const uint8_t *d, uint32_t N_d_bits, uint32_t N_e_bits, uint8_t *e)
{
    const uint32_t D = N_d_bits / 3, R = (D + 31) / 32, K_pi = 32 * R, N_dummy = K_pi - D;
    std::vector<uint8_t> w(3 * (size_t)K_pi);
    for (uint32_t x = 0; x < 3; x++)
        for (uint32_t k = 0; k < K_pi; k++) {
            const uint32_t t = 32 * (k % R) + bitrev5((k / R + 16) % 32);
            w[(size_t)x * K_pi + k] = t < N_dummy ? TX_NULL : d[3 * (t - N_dummy) + x];
        }
    for (uint32_t k = 0, j = 0; k < N_e_bits; j++) {
        const uint8_t b = w[j % (3 * K_pi)];
        if (b != TX_NULL) e[k++] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------- symbols
// 36.211 7.1.1-7.1.4 (modulation_mapper, liblte_phy.cc:8704-9490): the tables are the specification's, the amplitudes (float)(1 / sqrt(n)) times a
// small integer in float; a last, incomplete group is filled with zero bits
void modulate(const uint8_t *bits, uint32_t N_bits, uint32_t mod, float *re, float *im, uint32_t *M_symb)
{
    const float a2 = 1 / sqrt(2), a10 = 1 / sqrt(10), a42 = 1 / sqrt(42);
    if (mod == MI_LTE_MOD_BPSK) {
        for (uint32_t i = 0; i < N_bits; i++) re[i] = im[i] = bits[i] == 0 ? a2 : -a2;
        *M_symb = N_bits;
        return;
    }
    const uint32_t Q = mod == MI_LTE_MOD_QPSK ? 2 : mod == MI_LTE_MOD_16QAM ? 4 : 6, M = (N_bits + Q - 1) / Q;
    for (uint32_t i = 0; i < M; i++) {
        uint32_t b[6];
        for (uint32_t q = 0; q < Q; q++) b[q] = i * Q + q < N_bits ? bits[i * Q + q] : 0;
        int cr, ci;
        if (Q == 2) {
            re[i] = b[0] ? -a2 : a2, im[i] = b[1] ? -a2 : a2;
            continue;
        } else if (Q == 4) {
            cr = (1 - 2 * (int)b[0]) * (2 - (1 - 2 * (int)b[2])), ci = (1 - 2 * (int)b[1]) * (2 - (1 - 2 * (int)b[3]));
            re[i] = cr * a10, im[i] = ci * a10;
        } else {
            cr = (1 - 2 * (int)b[0]) * (4 - (1 - 2 * (int)b[2]) * (2 - (1 - 2 * (int)b[4])));
            ci = (1 - 2 * (int)b[1]) * (4 - (1 - 2 * (int)b[3]) * (2 - (1 - 2 * (int)b[5])));
            re[i] = cr * a42, im[i] = ci * a42;
        }
    }
    *M_symb = M;
}

// 36.211 6.3.3.1 / 6.3.3.3 (layer_mapper_dl, liblte_phy.cc:7293-7463) for the cases dl_layers_supported() admits: x holds the layers one after
// another, M_layer_symb each.  On four ports a symbol count that is 2 modulo 4 reads the two symbols behind the block -- whatever the previous
// call left in d, as in the reference (the specification appends two nulls there).
void layer_map_dl(const float *d_re, const float *d_im, uint32_t M_symb, uint32_t N_ant, uint32_t N_codewords, float *x_re, float *x_im, uint32_t *M_layer)
{
    const uint32_t v = (N_ant == 1 && N_codewords == 1) ? 1 : N_ant == 2 ? 2 : 4;
    const uint32_t M = v == 4 ? (M_symb % 4 == 0 ? M_symb / 4 : (M_symb + 2) / 4) : M_symb / v;
    for (uint32_t i = 0; i < M; i++)
        for (uint32_t l = 0; l < v; l++) x_re[l * M + i] = d_re[v * i + l], x_im[l * M + i] = d_im[v * i + l];
    *M_layer = M;
}

// 36.211 6.3.4.1 / 6.3.4.3 (pre_coder_dl, liblte_phy.cc:7526-7636): y[p * y_len + n]; every product is (float)(1 / sqrt 2) or its negative
// times the layer's value
void pre_code_dl(const float *x_re, const float *x_im, uint32_t M, uint32_t N_ant, float *y_re, float *y_im, uint32_t y_len, uint32_t *M_ap)
{
    const float a = 1 / sqrt(2);
    if (N_ant == 1) {
        memcpy(y_re, x_re, M * sizeof(float)), memcpy(y_im, x_im, M * sizeof(float));
        *M_ap = M;
        return;
    }
    // one Alamouti pair of layers (l0, l1) on ports (p0, p1), output places n0 / n0 + 1 of every group
    auto pair = [&](uint32_t l0, uint32_t l1, uint32_t p0, uint32_t p1, uint32_t group, uint32_t n0, uint32_t i) {
        float *r0 = y_re + (size_t)p0 * y_len, *i0 = y_im + (size_t)p0 * y_len, *r1 = y_re + (size_t)p1 * y_len, *i1 = y_im + (size_t)p1 * y_len;
        const uint32_t n = group * i + n0;
        r0[n]     = +a * x_re[l0 * M + i], i0[n]     = +a * x_im[l0 * M + i];
        r1[n]     = -a * x_re[l1 * M + i], i1[n]     = +a * x_im[l1 * M + i];
        r0[n + 1] = +a * x_re[l1 * M + i], i0[n + 1] = +a * x_im[l1 * M + i];
        r1[n + 1] = +a * x_re[l0 * M + i], i1[n + 1] = -a * x_im[l0 * M + i];
    };
    if (N_ant == 2) {
        for (uint32_t i = 0; i < M; i++) pair(0, 1, 0, 1, 2, 0, i);
        *M_ap = 2 * M;
        return;
    }
    const bool nulls = M > 0 && x_re[2 * M + M - 1] == (float)TX_NULL && x_im[2 * M + M - 1] == (float)TX_NULL && x_re[3 * M + M - 1] == (float)TX_NULL &&
                       x_im[3 * M + M - 1] == (float)TX_NULL;
    // four ports: layers (0, 1) as an Alamouti pair on ports (0, 2) in places 0 / 1 of every group of four, layers (2, 3) on ports (1, 3) in places
    // 2 / 3, the other ports resting.  Stored place by place, port by port, real part first -- the control channels hand this function rows that
    // overlap (tx_ctrl.cc), where the order of the stores decides what is left
    for (uint32_t i = 0; i < M; i++)
        for (uint32_t n = 0; n < 4; n++) {
            const uint32_t la = n < 2 ? 0 : 2, lb = la + 1, pa = n < 2 ? 0 : 1, pb = pa + 2; // the pair's layers and ports
            const uint32_t first = (n & 1) ? lb : la, second = (n & 1) ? la : lb;             // place 0: (a, conj-ish b); place 1: (b, a)
            for (uint32_t p = 0; p < 4; p++) {
                float re = 0, im = 0;
                if (p == pa) re = +a * x_re[first * M + i], im = +a * x_im[first * M + i];
                else if (p == pb) {
                    if (n & 1) re = +a * x_re[second * M + i], im = -a * x_im[second * M + i];
                    else re = -a * x_re[second * M + i], im = +a * x_im[second * M + i];
                }
                y_re[(size_t)p * y_len + 4 * i + n] = re;
                y_im[(size_t)p * y_len + 4 * i + n] = im;
            }
        }
    *M_ap = nulls ? 4 * M - 2 : 4 * M;
}

// resource elements of one PRB that carry PDSCH, as the reference PRICES them (get_num_bits_in_prb, liblte_phy.cc:13936-14086) -- an estimate by
// PRB, not the mapping loop's count: the two differ where the PBCH / PSS / SSS window cuts a PRB in a way the estimate does not model
static uint32_t bits_in_prb(uint32_t sf, uint32_t n_ctrl, uint32_t prb, uint32_t N_rb_dl, uint32_t N_ant, uint32_t mod)
{
    uint32_t lo, hi;
    bool     half_edges;
    switch (N_rb_dl) {
    case 6:  half_edges = false, lo = 0,  hi = 5;  break;
    case 15: half_edges = true,  lo = 4,  hi = 10; break;
    case 25: half_edges = true,  lo = 9,  hi = 15; break;
    case 50: half_edges = false, lo = 22, hi = 27; break;
    case 75: half_edges = true,  lo = 34, hi = 40; break;
    default: half_edges = false, lo = 47, hi = 52; break;
    }
    // per port count: all elements less the reference signals, less the control region, less what the window takes in subframe 0 from a whole
    // PRB / from a half-covered one
    uint32_t n_re, sf0_whole, sf0_half;
    if (N_ant == 1) n_re = 160 - ((n_ctrl - 1) * 12 + 10), sf0_whole = 70, sf0_half = 35;
    else if (N_ant == 2) n_re = 152 - ((n_ctrl - 1) * 12 + 8), sf0_whole = 68, sf0_half = 34;
    else n_re = 144 - (n_ctrl == 1 ? 8 : (n_ctrl - 2) * 12 + 16), sf0_whole = 64, sf0_half = 32;
    if (prb >= lo && prb <= hi) {
        const bool half = half_edges && (prb == lo || prb == hi);
        if (sf == 0) n_re -= half ? sf0_half : sf0_whole;
        else if (sf == 5) n_re -= half ? 12 : 24;
    }
    return mod == MI_LTE_MOD_BPSK ? n_re : mod == MI_LTE_MOD_QPSK ? 2 * n_re : mod == MI_LTE_MOD_16QAM ? 4 * n_re : mod == MI_LTE_MOD_64QAM ? 6 * n_re : 0;
}

// first / last sub-carrier of the PBCH / PSS / SSS window as the PDSCH mapping sees it (liblte_phy.cc:3521-3540)
static void sync_window(uint32_t N_rb_dl, uint32_t N_sc, uint32_t *first, uint32_t *last)
{
    switch (N_rb_dl) {
    case 6:  *first = 0,             *last = 6 * N_sc - 1;  break;
    case 15: *first = 4 * N_sc + 6,  *last = 11 * N_sc - 7; break;
    case 25: *first = 9 * N_sc + 6,  *last = 16 * N_sc - 7; break;
    case 50: *first = 22 * N_sc,     *last = 28 * N_sc - 1; break;
    case 75: *first = 34 * N_sc + 6, *last = 41 * N_sc - 7; break;
    default: *first = 47 * N_sc,     *last = 53 * N_sc - 1; break;
    }
}

// does symbol L, sub-carrier j of its PRB / sc of the grid belong to something else than the PDSCH (36.211 6.3.5; liblte_phy.cc:3640-3676)
static bool pdsch_re_taken(uint32_t N_ant, uint32_t cell, uint32_t sf, uint32_t L, uint32_t j, uint32_t sc, uint32_t first, uint32_t last)
{
    const uint32_t l = L % 7;
    if (N_ant == 1 && ((l == 0 && cell % 6 == j % 6) || (l == 4 && (cell + 3) % 6 == j % 6))) return true;
    if ((N_ant == 2 || N_ant == 4) && (l == 0 || l == 4) && cell % 3 == j % 3) return true;
    if (N_ant == 4 && l == 1 && cell % 3 == j % 3) return true;
    if (sc < first || sc > last) return false;
    if (sf == 0 && L >= 7 && L <= 10) return true;               // PBCH
    return (sf == 0 || sf == 5) && (L == 5 || L == 6);           // SSS, PSS
}

static bool dl_layers_supported(uint32_t N_ant, uint32_t N_codewords, uint32_t pre_coder_type)
{
    if (N_ant != 1 && N_ant != 2 && N_ant != 4) return false; // three ports index past the reference's own layer pointers
    if (N_codewords != 1 && N_codewords != 2) return false;
    if (N_ant == 1 && N_codewords == 1) return true;
    return pre_coder_type == MI_LTE_PRECODER_TX_DIVERSITY;    // anything else leaves the reference's pre-coder without an output
}

} // namespace tx

using namespace tx;

extern "C" {

int mi_lte_tx_create(mi_lte_tx **out)
{
    if (!out) return MI_LTE_ERR_ARG;
    mi_lte_tx *t = new (std::nothrow) mi_lte_tx;
    if (!t) return MI_LTE_ERR_ARG;
    memset(t, 0, sizeof(*t));
    *out = t;
    return MI_LTE_OK;
}
void mi_lte_tx_destroy(mi_lte_tx *tx) { delete tx; }

void mi_lte_rate_match_turbo(const uint8_t *d_bits, uint32_t N_d_bits, uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq,
                             uint32_t chan_type, uint32_t rv_idx, uint32_t N_e_bits, uint8_t *e_bits)
{
    rate_match_turbo(d_bits, N_d_bits, N_codeblocks, tx_mode, N_soft, M_dl_harq, chan_type, rv_idx, N_e_bits, e_bits);
}

// DL-SCH coding of one codeword (dlsch_channel_encode, liblte_phy.cc:12664-12757; 36.212 5.3.2): out = the G' * N_l * Q_m bits of all code
// blocks.  Every block is rate-matched into row 0 of the e-bit scratch and the concatenation then reads row r for block r, as the reference
// does: a transport block of one code block (everything the reference's receiver accepts) is exact, one of several gets the last block's
// bits first and then what the rows held before.
static int dlsch_encode(mi_lte_tx *t, const uint8_t *msg, uint32_t N_msg_bits, uint32_t tbs, uint32_t tx_mode, uint32_t rv_idx, uint32_t G, uint32_t N_l, uint32_t Q_m,
                        uint32_t M_dl_harq, uint32_t N_soft, uint8_t *out, uint32_t *N_out)
{
    std::vector<uint8_t> b(tbs + 24, 0);
    for (uint32_t i = 0; i < N_msg_bits && i < tbs; i++) b[i] = msg[i];
    crc_bits(b.data(), tbs, 0x1864CFB, 24, b.data() + tbs);
    uint32_t C = 0, F = 0;
    std::vector<uint32_t> N_c(MI_LTE_TX_MAX_CODE_BLOCKS + 1, 0);
    if ((tbs + 24 + 6119) / 6120 > MI_LTE_TX_MAX_CODE_BLOCKS) return MI_LTE_ERR_ARG;
    // (the code-block rows live in the handle: with several code blocks the reference computes each block's CRC over what its row held before,
    // liblte_phy.cc:9833-9850 -- sched.cc restates that -- so the rows' history is part of the input)
    std::vector<uint8_t> d(3 * (6176 + 4)); // (with several code blocks the segmentation reports K + 24 as a block's length, and that is what gets encoded: sched.cc)
    uint8_t             *c = &t->dlsch_c[0][0];
    mi_lte_code_block_segmentation(b.data(), tbs + 24, &C, &F, c, 6176, N_c.data());
    std::vector<uint32_t> N_e(C);
    const uint32_t        G_prime = G / (N_l * Q_m), lambda = G_prime % C;
    for (uint32_t cb = 0; cb < C; cb++) {
        turbo_encode(c + (size_t)cb * 6176, N_c[cb], d.data());
        N_e[cb] = cb <= C - lambda - 1 ? N_l * Q_m * (G_prime / C) : N_l * Q_m * (uint32_t)ceilf((float)G_prime / (float)C);
        if (N_e[cb] > 18432) return MI_LTE_ERR_ARG;
        rate_match_turbo(d.data(), 3 * (N_c[cb] + 4), C, tx_mode, N_soft, M_dl_harq, MI_LTE_CHAN_DLSCH, rv_idx, N_e[cb], t->dlsch_e[0]);
    }
    uint32_t k = 0;
    for (uint32_t r = 0; r < C; r++)
        for (uint32_t j = 0; j < N_e[r]; j++) out[k++] = t->dlsch_e[r][j];
    *N_out = k;
    return MI_LTE_OK;
}

int mi_lte_pdsch_channel_encode(mi_lte_tx *t, uint32_t N_rb_dl, uint32_t N_sc_rb_dl, const mi_lte_tx_alloc *allocs, uint32_t N_alloc, uint32_t N_pdcch_symbs,
                                uint32_t N_id_cell, uint32_t N_ant, uint32_t subfr_num, float *tx_re, float *tx_im)
{
    if (!t || (!allocs && N_alloc) || N_id_cell > 503 || !tx_re || !tx_im) return 1;
    uint32_t first_sc, last_sc;
    sync_window(N_rb_dl, N_sc_rb_dl, &first_sc, &last_sc);
    std::vector<uint8_t> enc(10000), scr(2 * 10000);
    for (uint32_t a = 0; a < N_alloc; a++) {
        const mi_lte_tx_alloc &al = allocs[a];
        if (al.chan_type != MI_LTE_CHAN_DLSCH) continue;
        if (al.N_prb > 110 || !dl_layers_supported(N_ant, al.N_codewords, al.pre_coder_type)) return 1;
        uint32_t G = 0;
        for (uint32_t i = 0; i < al.N_prb; i++) G += bits_in_prb(subfr_num, N_pdcch_symbs, al.prb[0][i], N_rb_dl, N_ant, al.mod_type);
        const uint32_t Q_m = al.mod_type == MI_LTE_MOD_BPSK ? 1 : al.mod_type == MI_LTE_MOD_QPSK ? 2 : al.mod_type == MI_LTE_MOD_16QAM ? 4 : 6;
        if (G == 0 || G > 10000 || G / (2 * Q_m) == 0) return 1; // beyond 10 000 bits the reference runs over its own arrays (liblte_phy.h:355-362)
        uint32_t N_bits = 0, n_scr = 0;
        for (uint32_t cw = 0; cw < al.N_codewords; cw++) {
            if (!al.msg[cw] && al.msg_bits[cw]) return 1;
            if (dlsch_encode(t, al.msg[cw], al.msg_bits[cw], al.tbs, al.tx_mode, al.rv_idx, G, 2, Q_m, 8, 250368, enc.data(), &N_bits) != MI_LTE_OK) return 1;
            std::vector<uint8_t> c(N_bits);
            synth::gold((al.rnti << 14) | (cw << 13) | (subfr_num << 9) | N_id_cell, N_bits, c.data());
            for (uint32_t j = 0; j < N_bits; j++) scr[n_scr++] = enc[j] ^ c[j];
        }
        // (the bit count of the LAST codeword goes to the mapper, whatever the number of codewords: liblte_phy.cc:3591-3596)
        uint32_t M_symb, M_layer, M_ap;
        modulate(scr.data(), N_bits, al.mod_type, t->pdsch_d_re, t->pdsch_d_im, &M_symb);
        if (N_ant == 4 && M_symb + 2 > 5000) return 1; // the fourth port's row of the reference's pre-coder output ends there (liblte_phy.h:353-354)
        layer_map_dl(t->pdsch_d_re, t->pdsch_d_im, M_symb, N_ant, al.N_codewords, t->pdsch_x_re, t->pdsch_x_im, &M_layer);
        pre_code_dl(t->pdsch_x_re, t->pdsch_x_im, M_layer, N_ant, t->pdsch_y_re, t->pdsch_y_im, 5000, &M_ap);
        for (uint32_t p = 0; p < N_ant; p++) {
            uint32_t idx = 0;
            for (uint32_t L = N_pdcch_symbs; L < 14; L++)
                for (uint32_t n = 0; n < al.N_prb; n++) {
                    const uint32_t prb = al.prb[L / 7][n];
                    if (prb >= 100) return 1;
                    for (uint32_t j = 0; j < N_sc_rb_dl; j++) {
                        const uint32_t sc = prb * N_sc_rb_dl + j;
                        if (pdsch_re_taken(N_ant, N_id_cell, subfr_num, L, j, sc, first_sc, last_sc)) continue;
                        if ((size_t)p * 5000 + idx >= 4 * 5000 || sc >= MI_LTE_TX_GRID_SC) return 1;
                        tx_re[MI_LTE_TX_GRID_AT(p, L, sc)] = t->pdsch_y_re[(size_t)p * 5000 + idx];
                        tx_im[MI_LTE_TX_GRID_AT(p, L, sc)] = t->pdsch_y_im[(size_t)p * 5000 + idx];
                        idx++;
                    }
                }
        }
    }
    return 0;
}

// 36.212 5.3.1 + 36.211 6.6 (liblte_phy_bch_channel_encode :3863-3966 over bch_channel_encode :12510-12573): the MIB's 24 bits are coded into the
// 1920 bits of a 40 ms period when the handle holds none (first call, or the call after a frame with sfn % 4 == 3), every call sends the quarter
// of its frame -- so a caller that starts in the middle of a period sends, like the reference's, quarters of a block coded from the MIB of that moment
int mi_lte_bch_channel_encode(mi_lte_tx *t, uint32_t N_rb_dl, uint32_t N_sc_rb_dl, const uint8_t *in_bits, uint32_t N_in_bits, uint32_t N_id_cell, uint32_t N_ant,
                              uint32_t sfn, float *tx_re, float *tx_im)
{
    (void)N_in_bits; // the reference reads 24 bits whatever it is told
    if (!t || !in_bits || N_id_cell > 503 || !tx_re || !tx_im) return 1;
    if (N_ant != 1 && N_ant != 2 && N_ant != 4) return 1;
    if (t->bch_N_bits == 0) {
        uint8_t c[40], d[120];
        memcpy(c, in_bits, 24);
        crc_bits(in_bits, 24, 0x11021, 16, c + 24);
        for (uint32_t i = 0; i < 16; i++) c[24 + i] ^= N_ant == 1 ? 0 : N_ant == 2 ? 1 : (i & 1); // 36.212 table 5.3.1.1-1
        conv_encode_tb(c, 40, d);
        rate_match_conv(d, 120, 1920, t->bch_encode_bits);
        t->bch_N_bits = 1920;
        synth::gold(N_id_cell, 1920, t->bch_c);
    }
    uint8_t        scr[480];
    const uint32_t off = (sfn % 4) * 480;
    for (uint32_t i = 0; i < 480; i++) scr[i] = t->bch_encode_bits[off + i] ^ t->bch_c[off + i];
    if (sfn % 4 == 3) t->bch_N_bits = 0;
    uint32_t M_symb, M_layer, M_ap;
    modulate(scr, 480, MI_LTE_MOD_QPSK, t->bch_d_re, t->bch_d_im, &M_symb);
    layer_map_dl(t->bch_d_re, t->bch_d_im, M_symb, N_ant, 1, t->bch_x_re, t->bch_x_im, &M_layer);
    pre_code_dl(t->bch_x_re, t->bch_x_im, M_layer, N_ant, t->bch_y_re, t->bch_y_im, 240, &M_ap);
    // symbols 7 and 8 skip the reference-signal places of ALL port counts, 9 and 10 take all 72 (36.211 6.6.4); the element counter runs on
    // across the ports as the reference's does (liblte_phy.cc:3934-3953): port p starts 48 elements into its own row
    if ((N_rb_dl * N_sc_rb_dl) / 2 < 36 || (N_rb_dl * N_sc_rb_dl) / 2 + 36 > MI_LTE_TX_GRID_SC) return 1;
    uint32_t idx = 0;
    for (uint32_t p = 0; p < N_ant; p++)
        for (uint32_t i = 0; i < 72; i++) {
            const uint32_t k = (N_rb_dl * N_sc_rb_dl) / 2 - 36 + i;
            const float   *yr = t->bch_y_re + (size_t)p * 240, *yi = t->bch_y_im + (size_t)p * 240;
            if (N_id_cell % 3 != i % 3) {
                if ((size_t)p * 240 + idx + 48 >= 4 * 240) return 1;
                tx_re[MI_LTE_TX_GRID_AT(p, 7, k)] = yr[idx], tx_im[MI_LTE_TX_GRID_AT(p, 7, k)] = yi[idx];
                tx_re[MI_LTE_TX_GRID_AT(p, 8, k)] = yr[idx + 48], tx_im[MI_LTE_TX_GRID_AT(p, 8, k)] = yi[idx + 48];
                idx++;
            }
            tx_re[MI_LTE_TX_GRID_AT(p, 9, k)] = yr[i + 96], tx_im[MI_LTE_TX_GRID_AT(p, 9, k)] = yi[i + 96];
            tx_re[MI_LTE_TX_GRID_AT(p, 10, k)] = yr[i + 168], tx_im[MI_LTE_TX_GRID_AT(p, 10, k)] = yi[i + 168];
        }
    return 0;
}

// 36.211 6.10.1 (liblte_phy_map_crs :5144-5263 over generate_crs :8300-8333): r(m) = (float)(1 / sqrt 2) * (1 - 2 c(2m)) + j ..., m' = m + 110 - N_rb_dl,
// ports 0 / 1 in symbols 0, 4, 7, 11 with offsets 0 / 3, ports 2 / 3 in symbols 1, 8.  The reference caches the sequences of one cell in its struct
// (crs_*_storage) and computes them otherwise: the same values either way.
int mi_lte_map_crs(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float *tx_re, float *tx_im)
{
    if (!tx_re || !tx_im || N_id_cell > 503 || N_rb_dl > 100 || N_ant > 4) return 1;
    const float    a = 1 / sqrt(2);
    const uint32_t N_cp = N_sc_rb_dl == 12 ? 1 : 0;
    uint8_t        c[440];
    for (uint32_t p = 0; p < N_ant; p++) {
        const uint32_t n_sym = p < 2 ? 4 : 2;
        for (uint32_t s = 0; s < n_sym; s++) {
            const uint32_t sym = p < 2 ? (s % 2 ? 4 : 0) + 7 * (s / 2) : 1 + 7 * s;
            // (port 3's second offset is 6 in the reference's table, 36.211's 3 + 3 (n_s mod 2): the same place modulo 6)
            const uint32_t v  = p == 0 ? (s % 2 ? 3 : 0) : p == 1 ? (s % 2 ? 0 : 3) : p == 2 ? (s ? 3 : 0) : (s ? 6 : 3);
            const uint32_t ns = 2 * subfr_num + sym / 7, l = sym % 7;
            synth::gold(1024 * (7 * (ns + 1) + l + 1) * (2 * N_id_cell + 1) + 2 * N_id_cell + N_cp, 440, c);
            for (uint32_t j = 0; j < 2 * N_rb_dl; j++) {
                const uint32_t k = 6 * j + (v + N_id_cell % 6) % 6, m = j + 110 - N_rb_dl;
                tx_re[MI_LTE_TX_GRID_AT(p, sym, k)] = a * (1 - 2 * (float)c[2 * m]);
                tx_im[MI_LTE_TX_GRID_AT(p, sym, k)] = a * (1 - 2 * (float)c[2 * m + 1]);
            }
        }
    }
    return 0;
}

// 36.211 6.11.1 (liblte_phy_map_pss :5265-5304 over generate_pss :8342-8368): Zadoff-Chu roots 25 / 29 / 34 (anything but 0 and 1 is 34), the
// phase evaluated in double from a float root and rounded to float for cosf / sinf, 62 values around the carrier in symbol 6
int mi_lte_map_pss(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t N_id_2, uint32_t N_ant, float *tx_re, float *tx_im)
{
    if (!tx_re || !tx_im || N_ant > 4 || (N_rb_dl * N_sc_rb_dl) / 2 < 31 || (N_rb_dl * N_sc_rb_dl) / 2 + 31 > MI_LTE_TX_GRID_SC) return 1;
    const float root = N_id_2 == 0 ? 25 : N_id_2 == 1 ? 29 : 34;
    for (uint32_t i = 0; i < 62; i++) {
        const uint32_t n  = i < 31 ? i : i + 1;
        const float    re = cosf(-M_PI * root * n * (n + 1) / 63), im = sinf(-M_PI * root * n * (n + 1) / 63);
        for (uint32_t p = 0; p < N_ant; p++) {
            const uint32_t k = i - 31 + (N_rb_dl * N_sc_rb_dl) / 2;
            tx_re[MI_LTE_TX_GRID_AT(p, 6, k)] = re, tx_im[MI_LTE_TX_GRID_AT(p, 6, k)] = im;
        }
    }
    return 0;
}

// 36.211 6.11.2 (liblte_phy_map_sss :5520-5576 over generate_sss :8377-8475): the interleaved m-sequences of subframe 0 / 5 in symbol 5; other
// subframes are left alone
int mi_lte_map_sss(uint32_t N_rb_dl, uint32_t N_sc_rb_dl, uint32_t subfr_num, uint32_t N_id_1, uint32_t N_id_2, uint32_t N_ant, float *tx_re, float *tx_im)
{
    if (!tx_re || !tx_im || N_ant > 4 || (N_rb_dl * N_sc_rb_dl) / 2 < 31 || (N_rb_dl * N_sc_rb_dl) / 2 + 31 > MI_LTE_TX_GRID_SC) return 1;
    if (subfr_num != 0 && subfr_num != 5) return 0;
    const uint32_t q_prime = N_id_1 / 30, q = (N_id_1 + q_prime * (q_prime + 1) / 2) / 30, m_prime = N_id_1 + q * (q + 1) / 2;
    const uint32_t m0 = m_prime % 31, m1 = (m0 + m_prime / 31 + 1) % 31;
    int            s[31], c[31], z[31];
    uint8_t        xs[31] = {0, 0, 0, 0, 1}, xc[31] = {0, 0, 0, 0, 1}, xz[31] = {0, 0, 0, 0, 1};
    for (uint32_t i = 0; i < 26; i++) {
        xs[i + 5] = (xs[i + 2] + xs[i]) % 2;
        xc[i + 5] = (xc[i + 3] + xc[i]) % 2;
        xz[i + 5] = (xz[i + 4] + xz[i + 2] + xz[i + 1] + xz[i]) % 2;
    }
    for (uint32_t i = 0; i < 31; i++) s[i] = 1 - 2 * xs[i], c[i] = 1 - 2 * xc[i], z[i] = 1 - 2 * xz[i];
    for (uint32_t i = 0; i < 31; i++) {
        const int s0 = s[(i + m0) % 31], s1 = s[(i + m1) % 31], c0 = c[(i + N_id_2) % 31], c1 = c[(i + N_id_2 + 3) % 31];
        const int z0 = z[(i + m0 % 8) % 31], z1 = z[(i + m1 % 8) % 31];
        const float even = subfr_num == 0 ? s0 * c0 : s1 * c0, odd = subfr_num == 0 ? s1 * c1 * z0 : s0 * c1 * z1;
        for (uint32_t p = 0; p < N_ant; p++) {
            const uint32_t k = 2 * i - 31 + (N_rb_dl * N_sc_rb_dl) / 2;
            tx_re[MI_LTE_TX_GRID_AT(p, 5, k)] = even, tx_im[MI_LTE_TX_GRID_AT(p, 5, k)] = 0;
            tx_re[MI_LTE_TX_GRID_AT(p, 5, k + 1)] = odd, tx_im[MI_LTE_TX_GRID_AT(p, 5, k + 1)] = 0;
        }
    }
    return 0;
}

// 36.211 6.12 (liblte_phy_create_dl_subframe :5862-5895 over symbols_to_samples_dl :8484-8531): the grid's sub-carriers around DC (DC itself
// empty), an unnormalised inverse transform of N_samps_per_symb points, the cyclic prefix in front.  The transform runs in float64 and is rounded
// to float once (the reference's is FFTW's single-precision plan: agreement to float rounding, not bit for bit).
int mi_lte_create_dl_subframe(uint32_t N_samps_per_symb, uint32_t N_used_sc, uint32_t N_samps_cp_l_0, uint32_t N_samps_cp_l_else, const float *tx_re, const float *tx_im,
                              uint32_t ant, float *i_samps, float *q_samps)
{
    if (!tx_re || !tx_im || !i_samps || !q_samps || ant > 3) return 1;
    if (N_samps_per_symb < 128 || (N_samps_per_symb & (N_samps_per_symb - 1)) || N_used_sc >= N_samps_per_symb || N_used_sc > MI_LTE_TX_GRID_SC || N_used_sc % 2) return 1;
    std::vector<double> xr(N_samps_per_symb), xi(N_samps_per_symb);
    const uint32_t      half = N_used_sc / 2;
    size_t              at   = 0;
    for (uint32_t L = 0; L < 14; L++) {
        const uint32_t cp = L % 7 == 0 ? N_samps_cp_l_0 : N_samps_cp_l_else;
        std::fill(xr.begin(), xr.end(), 0.0), std::fill(xi.begin(), xi.end(), 0.0);
        const float *sr = tx_re + MI_LTE_TX_GRID_AT(ant, L, 0), *si = tx_im + MI_LTE_TX_GRID_AT(ant, L, 0);
        for (uint32_t i = 0; i < half; i++) {
            xr[i + 1] = sr[half + i], xi[i + 1] = si[half + i];                                               // upper half above DC
            xr[N_samps_per_symb - 1 - i] = sr[half - 1 - i], xi[N_samps_per_symb - 1 - i] = si[half - 1 - i]; // lower half below it
        }
        synth::idft(xr, xi);
        float *o_re = i_samps + at, *o_im = q_samps + at;
        for (uint32_t n = 0; n < N_samps_per_symb; n++) o_re[cp + n] = (float)xr[n], o_im[cp + n] = (float)xi[n];
        for (uint32_t n = 0; n < cp; n++) o_re[n] = o_re[N_samps_per_symb + n], o_im[n] = o_im[N_samps_per_symb + n];
        at += N_samps_per_symb + cp;
    }
    return 0;
}

} // extern "C"
