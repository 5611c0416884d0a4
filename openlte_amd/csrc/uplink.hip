// PUSCH receive chain on gfx950 (BASELINE config 5): restates liblte_phy_pusch_channel_decode
// (liblte/src/liblte_phy.cc:2801-2935) -> ulsch_channel_decode (:12363-12501) for a batch of allocations
// (one per scheduled UE) inside the envelope the reference itself handles: single antenna, one codeword,
// no RI/ACK/CQI multiplexing, one code block per transport block.
//
//   k_pusch_demod : one workgroup per allocation.
//       DMRS least-squares estimate on symbols 3 and 10 and magnitude/phase interpolation over the 12
//       data symbols (get_ulsch_ce :13718-13790), one-tap equaliser (pre_decoder_and_matched_filter_ul
//       :6708-6736), transform pre-decoding = 12 unnormalised backward DFTs of size M = 12*N_prb times
//       sqrt(M) (:6627-6660; the reference's scaling is reproduced as is), modulation de-mapping, descrambling
//       (:2893-2898) and the channel de-interleaver, which without control bits is the transpose
//       g[(k*12 + s)*Q_m + q] = h[(s*M + k)*Q_m + q] (:12108-12225).  M is any size FFTW would plan
//       (N_prb divisible by 2, 3 or 5 -> radices 4, 2, 3, 5 and whatever prime is left), done as a
//       mixed-radix Stockham transform through LDS.
//   then the downlink's turbo stage (turbo.hip) with the UL-SCH rate-matching rule (N_cb = K_w).
//
// The DFT sizes are not powers of two and FFTW's operation order is unspecified, so like the downlink FFT
// this stage is tolerance-checked; everything from the int8 soft bits on is integer-exact.
#include <algorithm>
#include <map>
#include <tuple>

#include "ctx.hpp"
#include "phy_dev.hpp"
#include "lte_tables.h"

namespace {

constexpr int N_SC_MAX = 1200;

struct PuschDesc {
    uint32_t subfr, cell;   // of the allocation's unit
    uint32_t dmrs_off;      // float offset of dmrs_0_re | dmrs_0_im | dmrs_1_re | dmrs_1_im (M each) in the DMRS pool
};

// One Stockham pass of radix R (any R >= 2) over S symbols of M points each (symbol s at in + s*M_max), sign +1
// (backward transform):
//   out[(j-k)*R + k + q*Ns] = sum_r in[j + r*M/R] * exp(+2*pi*i * r*(k + q*Ns)/(Ns*R)),  k = j mod Ns.
// One thread per output; Ns*R divides M, so the twiddle is entry (r*(k + q*Ns) mod Ns*R) * M/(Ns*R) of the
// allocation's table tw[t] = exp(+2*pi*i*t/M).
__device__ __forceinline__ void dft_pass(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw,
                                         uint32_t S, uint32_t M, uint32_t M_max, uint32_t R, uint32_t Ns)
{
    const uint32_t nb = M / R, period = Ns * R, tstride = M / period;
    for (uint32_t o = threadIdx.x; o < S * M; o += blockDim.x) {
        const uint32_t sy = o / M, oo = o - sy * M;
        const uint32_t q = oo / nb, j = oo - q * nb, k = j % Ns, step = (k + q * Ns) * tstride; // step < M
        const float2  *x = in + sy * M_max + j;
        float    ar = 0.0f, ai = 0.0f;
        uint32_t t  = 0; // r*step mod M
        for (uint32_t r = 0; r < R; r++) {
            const float2 w = tw[t], v = x[r * nb];
            ar += v.x * w.x - v.y * w.y;
            ai += v.x * w.y + v.y * w.x;
            t += step;
            if (t >= M) t -= M;
        }
        out[sy * M_max + (j - k) * R + k + q * Ns] = make_float2(ar, ai);
    }
}

__global__ __launch_bounds__(256) void k_pusch_demod(const float *__restrict__ subframes, uint32_t sf_stride,
                                                     const mi_lte_pdsch_alloc *__restrict__ allocs, const PuschDesc *__restrict__ desc,
                                                     const float *__restrict__ dmrs_pool, GoldTables gt, int8_t *__restrict__ e_base,
                                                     const uint32_t *__restrict__ e_off, uint32_t *__restrict__ e_len, uint32_t M_max,
                                                     uint32_t S_par)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // LDS: est[6][M_max] (mag0 ang0 mag1 ang1 dmag dang) | tw[M_max] | buf A[S_par][M_max] | buf B[S_par][M_max] (float2) | scrambling words
    float    *est  = sm;
    float2   *tw   = reinterpret_cast<float2 *>(sm + 6 * (size_t)M_max);
    float2   *bufA = tw + M_max, *bufB = bufA + (size_t)S_par * M_max;
    uint32_t *cw   = reinterpret_cast<uint32_t *>(bufB + (size_t)S_par * M_max);

    const uint32_t a_idx = blockIdx.x;
    const mi_lte_pdsch_alloc &al = allocs[a_idx];
    const PuschDesc           ds = desc[a_idx];
    const uint32_t N_prb = al.N_prb, M = 12 * N_prb;
    const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1;
    const uint32_t N_bits = 12 * M * Qm;
    const float   *rx_re = subframes + (size_t)al.unit * sf_stride, *rx_im = rx_re + 16 * N_SC_MAX;
    if (threadIdx.x == 0) e_len[a_idx] = N_bits;

    // scrambling sequence (c_init per liblte_phy.cc:2893), one word of slack for the 2-word window below
    const uint32_t c_init = (al.rnti << 14) | (0u << 13) | (ds.subfr << 9) | ds.cell, n_words = (N_bits + 31) / 32;
    for (uint32_t w = threadIdx.x; w <= n_words; w += blockDim.x) cw[w] = gold_word(gt, c_init, w);

    // ---- DMRS estimates and their interpolation slopes (get_ulsch_ce, liblte_phy.cc:13745-13768); DFT twiddles
    const float *d0_re = dmrs_pool + ds.dmrs_off, *d0_im = d0_re + M, *d1_re = d0_im + M, *d1_im = d1_re + M;
    for (uint32_t i = threadIdx.x; i < M; i += blockDim.x) {
        const uint32_t sc0 = al.prb[0][i / 12] * 12 + i % 12, sc1 = al.prb[1][i / 12] * 12 + i % 12;
        const float c0r = rx_re[3 * N_SC_MAX + sc0], c0i = rx_im[3 * N_SC_MAX + sc0];
        const float c1r = rx_re[10 * N_SC_MAX + sc1], c1i = rx_im[10 * N_SC_MAX + sc1];
        float t_re = c0r * d0_re[i] + c0i * d0_im[i], t_im = c0i * d0_re[i] - c0r * d0_im[i];
        const float mag_0 = sqrtf(t_re * t_re + t_im * t_im), ang_0 = atan2f(t_im, t_re);
        t_re = c1r * d1_re[i] + c1i * d1_im[i];
        t_im = c1i * d1_re[i] - c1r * d1_im[i];
        const float mag_1 = sqrtf(t_re * t_re + t_im * t_im), ang_1 = atan2f(t_im, t_re);
        const float f_mag = (mag_1 - mag_0) / 7;
        float       f_ang = ang_1 - ang_0;
        if ((double)f_ang >= M_PI) f_ang = (float)((double)f_ang - 2 * M_PI); // float compared / corrected in double (:13758-13764)
        else if ((double)f_ang <= -M_PI) f_ang = (float)((double)f_ang + 2 * M_PI);
        f_ang /= 7;
        est[0 * M_max + i] = mag_0; est[1 * M_max + i] = ang_0; est[2 * M_max + i] = mag_1;
        est[3 * M_max + i] = ang_1; est[4 * M_max + i] = f_mag; est[5 * M_max + i] = f_ang;
        float sn, cs;
        sincospif(2.0f * (float)i / (float)M, &sn, &cs);
        tw[i] = make_float2(cs, sn);
    }
    __syncthreads();

    const float sqrt_M = (float)sqrt((double)M); // liblte_phy.cc:6644 (integer argument -> double sqrt, stored to float)
    int8_t     *e      = e_base + (size_t)e_off[a_idx] * 64; // 64-byte units

    for (uint32_t s0 = 0; s0 < 12; s0 += S_par) { // S_par data symbols at a time (all 12 when they fit in LDS)
        const uint32_t S = min(S_par, 12u - s0);
        // ---- channel estimate of each symbol and the one-tap equaliser
        for (uint32_t o = threadIdx.x; o < S * M; o += blockDim.x) {
            const uint32_t sy = o / M, i = o - sy * M, s = s0 + sy;
            const uint32_t L = s < 3 ? s : s < 9 ? s + 1 : s + 2; // data symbols in time order, skipping DMRS symbols 3 and 10
            const float mag_0 = est[i], ang_0 = est[M_max + i], mag_1 = est[2 * M_max + i], ang_1 = est[3 * M_max + i];
            const float f_mag = est[4 * M_max + i], f_ang = est[5 * M_max + i];
            float cm, ca; // liblte_phy.cc:13770-13780
            if (s < 3)      { cm = mag_0 - (float)(3 - s) * f_mag;       ca = ang_0 - (float)(3 - s) * f_ang; }
            else if (s < 6) { cm = mag_0 + (float)(1 + (s - 3)) * f_mag; ca = ang_0 + (float)(1 + (s - 3)) * f_ang; }
            else if (s < 9) { cm = mag_1 - (float)(3 - (s - 6)) * f_mag; ca = ang_1 - (float)(3 - (s - 6)) * f_ang; }
            else            { cm = mag_1 + (float)(1 + (s - 9)) * f_mag; ca = ang_1 + (float)(1 + (s - 9)) * f_ang; }
            float sn, cs;
            sincosf(ca, &sn, &cs); // the compiled reference resolves cos(float) to cosf (C++ overload)
            const float    h_re = cm * cs, h_im = cm * sn;
            const uint32_t sc = al.prb[L / 7][i / 12] * 12 + i % 12;
            const float    z_re = rx_re[L * N_SC_MAX + sc], z_im = rx_im[L * N_SC_MAX + sc];
            const float    hn = h_re * h_re + h_im * h_im;
            bufA[sy * M_max + i] = make_float2((z_re * h_re + z_im * h_im) / hn, (z_im * h_re - z_re * h_im) / hn);
        }
        __syncthreads();
        // ---- transform pre-decoding: M-point backward DFTs, radices 4, 2, 3, 5, then the remaining prime
        float2  *src = bufA, *dst = bufB;
        uint32_t rem = M, Ns = 1;
        while (rem > 1) { // uniform over the workgroup
            uint32_t R;
            if (rem % 4 == 0) R = 4;
            else if (rem % 2 == 0) R = 2;
            else if (rem % 3 == 0) R = 3;
            else if (rem % 5 == 0) R = 5;
            else {
                R = 7;
                while (rem % R) R += 2;
            }
            dft_pass(src, dst, tw, S, M, M_max, R, Ns);
            __syncthreads();
            Ns *= R;
            rem /= R;
            float2 *t = src; src = dst; dst = t;
        }
        // ---- de-map, descramble, de-interleave (transpose): soft bit q of symbol k goes to (k*12 + s)*Q_m + q
        for (uint32_t o = threadIdx.x; o < S * M; o += blockDim.x) {
            const uint32_t sy = o % S, k = o / S, s = s0 + sy; // neighbouring threads write neighbouring bytes of e
            const float2   x = src[sy * M_max + k];
            int8_t         b[6] = {0, 0, 0, 0, 0, 0};
            demap_symbol(sqrt_M * x.x, sqrt_M * x.y, al.mod_type, b);
            const uint32_t n0 = (s * M + k) * Qm, w = n0 >> 5, sh = n0 & 31;
            const uint32_t c  = __builtin_amdgcn_alignbit(cw[w + 1], cw[w], sh);
            int8_t        *ob = e + (size_t)(k * 12 + s) * Qm;
            for (uint32_t q = 0; q < Qm; q++) ob[q] = ((c >> q) & 1u) ? (int8_t)-b[q] : b[q];
        }
        __syncthreads();
    }
}

// ---- PUCCH formats 1 / 1a / 1b (liblte_phy_pucch_format_1_1a_1b_channel_decode, liblte_phy.cc:2961-3146) -----------------------
// One wavefront per PUCCH resource: the resource's PRB pair (N_1_p in slot 0, N_rb_ul - N_1_p - 1 in slot 1), a channel estimate
// per slot from its three reference symbols (get_ulcch_ce :13799-13868: mean magnitude, the FIRST symbol's phase), one-tap
// equalisation, and the coherent sum over 2 x 4 x 12 elements of z / (s_ns w(i) r_u,v(n)) -- summed in the reference's order by one
// lane, divided by 95 as the reference does (its loop index after the last element, not the count) -- then the reference's decision.
// The sequences are the caller's (what liblte_phy_ul_init left in LIBLTE_PHY_STRUCT): per resource 352 floats, see mi_lte.h.
struct PucchRes { uint32_t unit, format, N_1_p; };
constexpr uint32_t PUCCH_TAB_FLOATS = 4 * 36 + 2 * 96 + 16;

__device__ __forceinline__ float pucch_soft(float rx_re, float rx_im, float exp_re, float exp_im)
{
    const float d_re = rx_re - exp_re, d_im = rx_im - exp_im;
    float dist = sqrtf(d_re * d_re + d_im * d_im); // (C++ overload resolution in the reference: sqrt(float) is the float function)
    const float cap = 1.0f - (1.0f / 120);
    if (dist >= cap) dist = cap;
    return 1.0f - dist;
}

__global__ __launch_bounds__(64) void k_pucch_decode(const float *__restrict__ subframes, uint32_t sf_stride, uint32_t N_rb_ul, uint32_t N_ant,
                                                     const PucchRes *__restrict__ res, const float *__restrict__ tabs, uint8_t *__restrict__ out /*[n][4]*/)
{
    __shared__ float c_re[2][12], c_im[2][12], t_re[96], t_im[96];
    const uint32_t r = blockIdx.x, ln = threadIdx.x;
    const PucchRes pr = res[r];
    const float *base = subframes + (size_t)pr.unit * sf_stride, *y_re = base, *y_im = base + 16 * N_SC_MAX;
    const float *tb = tabs + (size_t)r * PUCCH_TAB_FLOATS;
    const float *dm_re[2] = {tb, tb + 72}, *dm_im[2] = {tb + 36, tb + 108};
    const float *ruv_re = tb + 144, *ruv_im = tb + 240, *sw_re = tb + 336, *sw_im = tb + 344;
    const uint32_t prb[2] = {pr.N_1_p, N_rb_ul - pr.N_1_p - 1};
    if (ln < 24) { // (slot, sub-carrier): channel estimate from the three reference symbols 2, 3, 4 (9, 10, 11)
        const uint32_t s = ln / 12, j = ln % 12;
        float ave_mag = 0, ang0 = 0;
        for (uint32_t i = 0; i < 3; i++) {
            const uint32_t L = 7 * s + 2 + i, k = prb[s] * 12 + j, idx = i * 12 + j;
            const float cr = y_re[L * N_SC_MAX + k], ci = y_im[L * N_SC_MAX + k], dr = dm_re[s][idx], di = dm_im[s][idx];
            const float denom = dr * dr + di * di;
            const float tr = (1 / denom) * (cr * dr + ci * di), ti = (1 / denom) * (ci * dr - cr * di);
            ave_mag += sqrtf(tr * tr + ti * ti) / 3; // (:13842; float sqrt / int)
            if (i == 0) ang0 = atan2f(ti, tr);
        }
        c_re[s][j] = ave_mag * cosf(ang0);
        c_im[s][j] = ave_mag * sinf(ang0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t idx = ln; idx < 96; idx += 64) { // element (slot m', data symbol i, sub-carrier j)
        const uint32_t mp = idx / 48, i = (idx % 48) / 12, j = idx % 12, L = 7 * mp + (i < 2 ? i : i + 3), k = prb[mp] * 12 + j;
        const float zr = y_re[L * N_SC_MAX + k], zi = y_im[L * N_SC_MAX + k], hr = c_re[mp][j], hi = c_im[mp][j];
        const float hn = hr * hr + hi * hi;
        const float xr = (zr * hr + zi * hi) / hn, xi = (zi * hr - zr * hi) / hn; // pre_decoder_and_matched_filter_ul (:6727-6732)
        const float swr = sw_re[mp * 4 + i], swi = sw_im[mp * 4 + i], rr = ruv_re[idx], ri = ruv_im[idx];
        const float pr_ = swr * rr - swi * ri, pi_ = swr * ri + swi * rr, denom = pr_ * pr_ + pi_ * pi_;
        t_re[idx] = (1 / denom) * (xr * pr_ + xi * pi_);
        t_im[idx] = (1 / denom) * (-xr * pi_ + xi * pr_);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (ln == 0) {
        float d_re = 0, d_im = 0;
        for (uint32_t idx = 0; idx < 96; idx++) { d_re += t_re[idx]; d_im += t_im[idx]; }
        d_re /= 95u; // the reference divides by its last index (:3097-3098)
        d_im /= 95u;
        d_re *= sqrt((double)N_ant);
        d_im *= sqrt((double)N_ant);
        uint8_t b0 = 0, b1 = 0, nb = 1;
        float   sd;
        if (pr.format <= 1) {
            if (d_re < 0) { sd = pucch_soft(d_re, d_im, -1, 0); b0 = 1; }
            else          { sd = pucch_soft(d_re, d_im, 1, 0); b0 = 0; }
        } else {
            const float ang = atan2f(d_im, d_re);
            nb = 2;
            if (ang >= M_PI / 4 && ang < 3 * M_PI / 4)        { sd = pucch_soft(d_re, d_im, 0, 1); b0 = 1; b1 = 0; }
            else if (ang >= -M_PI / 4 && ang < M_PI / 4)      { sd = pucch_soft(d_re, d_im, 1, 0); b0 = 0; b1 = 0; }
            else if (ang >= -3 * M_PI / 4 && ang < -M_PI / 4) { sd = pucch_soft(d_re, d_im, 0, -1); b0 = 0; b1 = 1; }
            else                                              { sd = pucch_soft(d_re, d_im, -1, 0); b0 = 1; b1 = 1; }
        }
        out[4 * r] = b0; out[4 * r + 1] = b1; out[4 * r + 2] = nb; out[4 * r + 3] = sd > 0.5f ? 0 : 1; // LIBLTE_SUCCESS / LIBLTE_ERROR_INVALID_INPUTS
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host side: plans

struct mi_lte_pusch_plan {
    mi_lte_dl_cfg cfg;
    uint32_t      n_alloc = 0, out_stride = 0, M_max = 0, words_max = 0;
    size_t        e_bytes = 0;
    mi_lte_pdsch_alloc *d_allocs = nullptr;
    PuschDesc          *d_desc   = nullptr;
    float              *d_dmrs   = nullptr;
    uint32_t *d_e_off = nullptr, *d_e_len = nullptr, *d_cb_alloc = nullptr;
    int8_t   *d_e = nullptr;
    struct Group { uint32_t K, n_cb, cb_base, e_max; };
    std::vector<Group>    groups;
    std::vector<uint32_t> h_e_off, h_e_len;
};

static uint32_t ul_qpp_size_at_least(uint32_t B)
{
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (LTE_QPP_ROWS[r].K >= B) return LTE_QPP_ROWS[r].K;
    return 0;
}

// h_dmrs (optional): caller-supplied reference signals, 4 x 12*N_prb floats per allocation back to back -- the
// per-call host form passes the arrays liblte_phy_ul_init left in the caller's LIBLTE_PHY_STRUCT
int mi_pusch_plan_create_impl(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, const uint32_t *h_unit_subfr_num,
                              const uint32_t *h_unit_n_id_cell, uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                              const float *h_dmrs, mi_lte_pusch_plan **out)
{
    if (!ctx || !cfg || (!ul && !h_dmrs) || !h_unit_subfr_num || !h_unit_n_id_cell || !h_allocs || !out || n_alloc == 0 || n_units == 0)
        return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl    = new mi_lte_pusch_plan();
    pl->cfg     = *cfg;
    pl->n_alloc = n_alloc;
    std::map<uint32_t, std::vector<uint32_t>> byK;
    std::map<uint32_t, uint32_t>              emaxK;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> dmrs_at; // (cell, subframe, N_prb) -> float offset
    std::vector<float>     dmrs;
    std::vector<PuschDesc> desc(n_alloc);
    uint32_t max_tbs = 0;
    size_t   off = 0;
    pl->h_e_off.resize(n_alloc);
    pl->h_e_len.resize(n_alloc);
    for (uint32_t a = 0; a < n_alloc; a++) {
        const mi_lte_pdsch_alloc &al = h_allocs[a];
        const uint32_t B = al.tbs + 24, K = (B <= 6144) ? ul_qpp_size_at_least(B) : 0;
        // N_prb the reference has a transform pre-decoding plan for (liblte_phy.cc:2360-2377)
        const bool planned = al.N_prb > 0 && al.N_prb < cfg->N_rb_dl && (al.N_prb % 2 == 0 || al.N_prb % 3 == 0 || al.N_prb % 5 == 0);
        if (K == 0 || !planned || al.mod_type > 3 || al.unit >= n_units) {
            ctx->err = "PUSCH allocation outside the envelope (one code block; N_prb < N_rb_ul and divisible by 2, 3 or 5) or malformed";
            delete pl;
            return MI_LTE_ERR_UNSUPPORTED;
        }
        const uint32_t sf = h_unit_subfr_num[al.unit] % 10, cell = h_unit_n_id_cell[al.unit], M = 12 * al.N_prb;
        if (h_dmrs) {
            const uint32_t at = (uint32_t)dmrs.size();
            dmrs.insert(dmrs.end(), h_dmrs, h_dmrs + 4 * (size_t)M);
            h_dmrs += 4 * (size_t)M;
            desc[a] = {sf, cell, at};
        } else {
            auto key = std::make_tuple(cell, sf, al.N_prb);
            auto it  = dmrs_at.find(key);
            if (it == dmrs_at.end()) {
                const uint32_t at = (uint32_t)dmrs.size();
                dmrs.resize(dmrs.size() + 4 * (size_t)M);
                int rc = mi_lte_ul_dmrs_pusch(ul, cell, sf, al.N_prb, &dmrs[at], &dmrs[at + M], &dmrs[at + 2 * M], &dmrs[at + 3 * M]);
                if (rc != MI_LTE_OK) { delete pl; return rc; }
                it = dmrs_at.emplace(key, at).first;
            }
            desc[a] = {sf, cell, it->second};
        }
        byK[K].push_back(a);
        max_tbs = std::max(max_tbs, al.tbs);
        const uint32_t Qm = al.mod_type == 3 ? 6 : al.mod_type == 2 ? 4 : al.mod_type == 1 ? 2 : 1, E = 12 * M * Qm;
        pl->M_max     = std::max(pl->M_max, M);
        pl->words_max = std::max(pl->words_max, (E + 31) / 32 + 1);
        emaxK[K]       = std::max(emaxK[K], E);
        pl->h_e_off[a] = (uint32_t)(off >> 6); // in 64-byte units
        pl->h_e_len[a] = E;
        off += (E + 63) & ~63u;
    }
    if (pl->words_max > 4096) { ctx->err = "allocation larger than the scrambling table"; delete pl; return MI_LTE_ERR_UNSUPPORTED; }
    pl->e_bytes    = off;
    pl->out_stride = (max_tbs + 63) & ~63u;
    std::vector<uint32_t> cb_alloc;
    for (auto &kv : byK) {
        pl->groups.push_back({kv.first, (uint32_t)kv.second.size(), (uint32_t)cb_alloc.size(), emaxK[kv.first]});
        cb_alloc.insert(cb_alloc.end(), kv.second.begin(), kv.second.end());
    }
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_desc, sizeof(PuschDesc) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_dmrs, sizeof(float) * std::max<size_t>(dmrs.size(), 1)));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_off, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e_len, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_cb_alloc, sizeof(uint32_t) * n_alloc));
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_e, pl->e_bytes ? pl->e_bytes : 64));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_allocs, h_allocs, sizeof(mi_lte_pdsch_alloc) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_desc, desc.data(), sizeof(PuschDesc) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_dmrs, dmrs.data(), sizeof(float) * dmrs.size(), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_e_off, pl->h_e_off.data(), sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_cb_alloc, cb_alloc.data(), sizeof(uint32_t) * n_alloc, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = pl;
    return MI_LTE_OK;
}

extern "C" {

int mi_lte_pusch_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_ul_cfg *ul, const uint32_t *h_unit_subfr_num,
                             const uint32_t *h_unit_n_id_cell, uint32_t n_units, const mi_lte_pdsch_alloc *h_allocs, uint32_t n_alloc,
                             mi_lte_pusch_plan **out)
{
    if (!ul) return MI_LTE_ERR_INVALID_ARG;
    return mi_pusch_plan_create_impl(ctx, cfg, ul, h_unit_subfr_num, h_unit_n_id_cell, n_units, h_allocs, n_alloc, nullptr, out);
}

void mi_lte_pusch_plan_destroy(mi_lte_ctx *ctx, mi_lte_pusch_plan *pl)
{
    if (!pl) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(pl->d_allocs);
    (void)hipFree(pl->d_desc);
    (void)hipFree(pl->d_dmrs);
    (void)hipFree(pl->d_e_off);
    (void)hipFree(pl->d_e_len);
    (void)hipFree(pl->d_cb_alloc);
    (void)hipFree(pl->d_e);
    delete pl;
}

uint32_t mi_lte_pusch_plan_out_stride(const mi_lte_pusch_plan *pl) { return pl ? pl->out_stride : 0; }

int mi_lte_pusch_plan_soft_bits(const mi_lte_pusch_plan *pl, uint32_t alloc, const int8_t **d_e, uint32_t *n_bits)
{
    if (!pl || alloc >= pl->n_alloc || !d_e || !n_bits) return MI_LTE_ERR_INVALID_ARG;
    *d_e    = pl->d_e + (size_t)pl->h_e_off[alloc] * 64;
    *n_bits = pl->h_e_len[alloc];
    return MI_LTE_OK;
}

int mi_lte_pusch_decode_run(mi_lte_ctx *ctx, mi_lte_pusch_plan *pl, const float *d_subframes, uint8_t *d_out_bits, int32_t *d_status)
{
    if (!ctx || !pl || !d_subframes || !d_out_bits || !d_status) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    GoldTables   gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    // as many of the 12 data symbols side by side as fit in ~40 KiB of ping-pong buffers (all 12 up to 17 PRB)
    uint32_t S_par = 12;
    while (S_par > 1 && (size_t)S_par * pl->M_max * 2 * sizeof(float2) > 40 * 1024) S_par = S_par == 12 ? 6 : S_par == 6 ? 4 : S_par == 4 ? 3 : S_par - 1;
    const size_t lds = sizeof(float) * 6 * (size_t)pl->M_max + sizeof(float2) * (size_t)pl->M_max * (1 + 2 * S_par) +
                       sizeof(uint32_t) * (pl->words_max + 1);
    MI_LAUNCH(ctx, "k_pusch_demod", k_pusch_demod, dim3(pl->n_alloc), dim3(256), lds, d_subframes, (uint32_t)mi_lte_ul_subframe_floats(),
              pl->d_allocs, pl->d_desc, pl->d_dmrs, gt, pl->d_e, pl->d_e_off, pl->d_e_len, pl->M_max, S_par);
    MI_HIP_CHECK(ctx, hipGetLastError());
    for (auto &gr : pl->groups) {
        rc = mi_turbo_ref_group(ctx, gr.K, gr.n_cb, pl->d_allocs, pl->d_cb_alloc + gr.cb_base, pl->d_e, pl->d_e_off, pl->d_e_len,
                                d_out_bits, pl->out_stride, d_status, gr.e_max, /*ul=*/true);
        if (rc != MI_LTE_OK) return rc;
    }
    ctx->last_kernels = "k_pusch_demod:1,k_turbo_prep,k_turbo_siso,k_turbo_perm,k_turbo_vote per block size";
    return MI_LTE_OK;
}

// liblte_phy_pucch_format_1_1a_1b_channel_decode for a batch of PUCCH resources over UL device subframes
int mi_lte_pucch_decode_run(mi_lte_ctx *ctx, uint32_t N_rb_ul, uint32_t N_ant, const float *d_subframes, const mi_lte_pucch_res *h_res,
                            const float *h_tables, uint32_t n_res, uint8_t *h_bits /*[n_res][2]*/, uint32_t *h_n_bits, uint32_t *h_rc)
{
    if (!ctx || !d_subframes || !h_res || !h_tables || n_res == 0 || !h_bits || !h_n_bits || !h_rc || N_rb_ul < 6 || N_rb_ul > 100 || N_ant != 1)
        return MI_LTE_ERR_INVALID_ARG;
    for (uint32_t r = 0; r < n_res; r++)
        if (h_res[r].format > 2 || h_res[r].N_1_p_pucch >= N_rb_ul) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t b_res = sizeof(PucchRes) * (size_t)n_res, b_tab = sizeof(float) * PUCCH_TAB_FLOATS * (size_t)n_res, b_out = 4 * (size_t)n_res;
    const size_t o_tab = (b_res + 255) & ~(size_t)255, o_out = (o_tab + b_tab + 255) & ~(size_t)255;
    int rc = mi_ctx_reserve_scratch(ctx, o_out + b_out);
    if (rc != MI_LTE_OK) return rc;
    char *base = (char *)ctx->scratch;
    std::vector<PucchRes> res(n_res);
    for (uint32_t r = 0; r < n_res; r++) res[r] = PucchRes{h_res[r].unit, h_res[r].format, h_res[r].N_1_p_pucch};
    MI_HIP_CHECK(ctx, hipMemcpyAsync(base, res.data(), b_res, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(base + o_tab, h_tables, b_tab, hipMemcpyHostToDevice, ctx->stream));
    MI_LAUNCH(ctx, "k_pucch_decode", k_pucch_decode, dim3(n_res), dim3(64), 0, d_subframes, (uint32_t)mi_lte_ul_subframe_floats(), N_rb_ul, N_ant,
              (const PucchRes *)base, (const float *)(base + o_tab), (uint8_t *)(base + o_out));
    MI_HIP_CHECK(ctx, hipGetLastError());
    std::vector<uint8_t> o(b_out);
    MI_HIP_CHECK(ctx, hipMemcpyAsync(o.data(), base + o_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t r = 0; r < n_res; r++) { h_bits[2 * r] = o[4 * r]; h_bits[2 * r + 1] = o[4 * r + 1]; h_n_bits[r] = o[4 * r + 2]; h_rc[r] = o[4 * r + 3]; }
    ctx->last_kernels = "k_pucch_decode:1";
    return MI_LTE_OK;
}

} // extern "C"
