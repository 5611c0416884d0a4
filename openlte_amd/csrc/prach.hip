// PRACH detection on gfx950 (the rest of SURVEY 8f N1 / BASELINE config 5): restates liblte_phy_detect_prach
// (liblte/src/liblte_phy.cc:3299-3479) for preamble formats 0-3 and a batch of PRACH occasions.
//
// The reference transforms the T_fft = 24 576 samples behind the cyclic prefix with one FFT and keeps N_zc = 839 bins
// (:3421-3433), multiplies them with the conjugate spectrum of each root sequence and goes back with an 839-point
// inverse DFT per root (:3438-3458), then looks for one peak >= 50 x the (recursively averaged) correlation power.
//
//   k_prach_bins : only those 839 bins are needed, so they are computed directly: one thread per (bin, eighth of the
//                  samples), rotating a phasor that is re-seeded from an exactly reduced integer angle every 64 samples.
//                  24 576 x 839 complex MACs per occasion; no 24 576-point FFT, no 196 KB of LDS.
//   k_prach_corr : one workgroup per (occasion, root): spectrum product, 839-point inverse DFT with a twiddle table in
//                  LDS (839 is prime), per-root sum / maximum / arg-maximum of the correlation power.
// The verdict (threshold, preamble index, timing advance) is the reference's scalar arithmetic on the host (:3460-3474).
// FFTW's rounding is unspecified, so the correlation powers are tolerance-level; the outputs are integers.
#include <cmath>
#include <vector>

#include "ctx.hpp"
#include "lte_tables.h"
#include "prach_sets.hpp"

namespace {

constexpr uint32_t N_ZC = 839;

template <typename T> struct Samp;
template <> struct Samp<int8_t> {
    const int8_t *p;
    __device__ __forceinline__ float2 at(size_t n) const
    {
        const char2 v = *reinterpret_cast<const char2 *>(p + 2 * n);
        return make_float2((float)v.x, (float)v.y);
    }
};
template <> struct Samp<float> {
    const float *i, *q;
    __device__ __forceinline__ float2 at(size_t n) const { return make_float2(i[n], q[n]); }
};

// x_hat[occ][b] = sum_n x[n] exp(-2*pi*i*n*idx_b/T), idx_b = (b + start + T/2) mod T        (liblte_phy.cc:3421-3433)
template <typename T>
__global__ __launch_bounds__(256) void k_prach_bins(Samp<T> src, const uint64_t *__restrict__ occ_start, uint32_t T_cp, uint32_t T_fft,
                                                    uint32_t start, float2 *__restrict__ x_hat)
{
    __shared__ float2 part[8][32];
    const uint32_t occ = blockIdx.y, bl = threadIdx.x & 31, sg = threadIdx.x >> 5, b = blockIdx.x * 32 + bl;
    const uint32_t idx = (min(b, N_ZC - 1) + start + T_fft / 2) % T_fft, seg = T_fft / 8;
    const size_t   first = occ_start[occ] + T_cp;
    float ar = 0.f, ai = 0.f;
    for (uint32_t n0 = sg * seg; n0 < (sg + 1) * seg; n0 += 64) {
        // exp(-2*pi*i*n0*idx/T) from the exactly reduced angle, then a 64-step rotation
        const uint32_t t = (uint32_t)(((uint64_t)n0 * idx) % T_fft);
        float ws, wc, rs, rc;
        sincospif(-2.0f * (float)t / (float)T_fft, &ws, &wc);
        sincospif(-2.0f * (float)idx / (float)T_fft, &rs, &rc);
        for (uint32_t k = 0; k < 64; k++) {
            const float2 x = src.at(first + n0 + k);
            ar += x.x * wc - x.y * ws;
            ai += x.x * ws + x.y * wc;
            const float nc = wc * rc - ws * rs, ns = wc * rs + ws * rc;
            wc = nc; ws = ns;
        }
    }
    part[sg][bl] = make_float2(ar, ai);
    __syncthreads();
    if (sg == 0 && b < N_ZC) {
        float2 s = part[0][bl];
        for (int k = 1; k < 8; k++) { s.x += part[k][bl].x; s.y += part[k][bl].y; }
        x_hat[(size_t)occ * N_ZC + b] = s;
    }
}

struct CorrOut { float sum, max_val; uint32_t max_off, pad; };

// per (occasion, root): in[j] = X_u[j] * conj(x_hat[j]); out[m] = sum_j in[j] exp(+2*pi*i*j*m/839); power statistics
__global__ __launch_bounds__(256) void k_prach_corr(const float2 *__restrict__ x_hat, const float2 *__restrict__ xu_fft, uint32_t n_roots,
                                                    CorrOut *__restrict__ out)
{
    __shared__ float2   tw[N_ZC], in[N_ZC];
    __shared__ float    r_sum[4], r_max[4];
    __shared__ uint32_t r_off[4];
    const uint32_t occ = blockIdx.y, root = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < N_ZC; j += blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)j / (float)N_ZC, &s, &c);
        tw[j] = make_float2(c, s);
        const float2 u = xu_fft[(size_t)root * N_ZC + j], h = x_hat[(size_t)occ * N_ZC + j];
        in[j] = make_float2(u.x * h.x + u.y * h.y, u.y * h.x - u.x * h.y); // liblte_phy.cc:3443-3444
    }
    __syncthreads();
    float    sum = 0.f, mx = -1.f;
    uint32_t off = 0;
    for (uint32_t m = threadIdx.x; m < N_ZC; m += blockDim.x) { // ascending m per thread: the first maximum wins
        float    ar = 0.f, ai = 0.f;
        uint32_t t = 0;
        for (uint32_t j = 0; j < N_ZC; j++) {
            const float2 w = tw[t], v = in[j];
            ar += v.x * w.x - v.y * w.y;
            ai += v.x * w.y + v.y * w.x;
            t += m;
            if (t >= N_ZC) t -= N_ZC;
        }
        const float p = ar * ar + ai * ai;
        sum += p;
        if (p > mx) { mx = p; off = m; }
    }
    // wave reduction: sum; maximum with the smallest offset among equals (the reference scans offsets in ascending order)
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        const float    om = __shfl_xor(mx, o);
        const uint32_t oo = __shfl_xor(off, o);
        if (om > mx || (om == mx && oo < off)) { mx = om; off = oo; }
    }
    if ((threadIdx.x & 63) == 0) { r_sum[threadIdx.x >> 6] = sum; r_max[threadIdx.x >> 6] = mx; r_off[threadIdx.x >> 6] = off; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            sum += r_sum[w];
            if (r_max[w] > mx || (r_max[w] == mx && r_off[w] < off)) { mx = r_max[w]; off = r_off[w]; }
        }
        out[(size_t)occ * n_roots + root] = {sum, mx, off, 0};
    }
}

} // namespace

struct mi_lte_prach_plan {
    mi_lte_dl_cfg cfg;
    uint32_t      T_fft = 0, T_cp = 0, start = 0, N_cs = 0, v_max = 0, n_roots = 0;
    float2       *d_xu_fft = nullptr;
};

static int prach_plan_common(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *pc, const float *h_xu_fft_re,
                             const float *h_xu_fft_im, uint32_t n_roots_given, mi_lte_prach_plan **out)
{
    if (!ctx || !cfg || !pc || !out) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048) || cfg->N_rb_dl * 12 >= N) return MI_LTE_ERR_INVALID_ARG;
    if (pc->preamble_format > 3 || pc->zczc > (pc->hs_flag ? 14u : 15u) || pc->root_seq_idx > 837) {
        ctx->err = "PRACH: preamble formats 0-3 only (format 4 is TDD), zczc / root index out of range";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl = new mi_lte_prach_plan();
    pl->cfg  = *cfg;
    const uint32_t sc = 2048 / N; // 30.72 MHz / fs
    static const uint32_t cp_of_fmt[4] = {3168, 21024, 6240, 21024};
    pl->T_fft = 24576 / sc;                          // liblte_phy.cc:2417-2448
    pl->T_cp  = cp_of_fmt[pc->preamble_format] / sc;
    const uint32_t k_0 = pc->freq_offset * 12 - cfg->N_rb_dl * 12 / 2 + N / 2, K = 12; // :3416-3418 (uint32 arithmetic)
    pl->start = 7 + K * k_0 + K / 2;                 // :3426
    {   // N_cs and v_max of the FIRST root, as liblte_phy_detect_prach uses them for every root (:3336-3413)
        const PrachSets ps = prach_sets(LTE_PRACH_ROOT_ORDER[pc->root_seq_idx], pc->zczc, pc->hs_flag != 0);
        pl->N_cs = ps.N_cs; pl->v_max = ps.v_max;
    }
    std::vector<float2> xu;
    if (h_xu_fft_re && h_xu_fft_im) { // the caller's spectra (what liblte_phy_ul_init left in LIBLTE_PHY_STRUCT)
        pl->n_roots = n_roots_given;
        xu.resize((size_t)pl->n_roots * N_ZC);
        for (size_t k = 0; k < xu.size(); k++) xu[k] = make_float2(h_xu_fft_re[k], h_xu_fft_im[k]);
    } else { // roots needed for 64 preambles (prach_preamble_seq_gen, :7130-7290), their forward DFTs in double
        uint32_t n_gen = 0;
        while (n_gen < 64 && pc->root_seq_idx + pl->n_roots < 838) {
            n_gen += prach_sets(LTE_PRACH_ROOT_ORDER[pc->root_seq_idx + pl->n_roots], pc->zczc, pc->hs_flag != 0).v_max + 1;
            pl->n_roots++;
        }
        xu.resize((size_t)pl->n_roots * N_ZC);
        std::vector<double> xr(N_ZC), xi(N_ZC), cs(N_ZC), sn(N_ZC);
        for (uint32_t t = 0; t < N_ZC; t++) { cs[t] = std::cos(-2.0 * M_PI * t / N_ZC); sn[t] = std::sin(-2.0 * M_PI * t / N_ZC); }
        for (uint32_t r = 0; r < pl->n_roots; r++) {
            const uint32_t u = LTE_PRACH_ROOT_ORDER[pc->root_seq_idx + r];
            for (uint32_t i = 0; i < N_ZC; i++) { // x_u(n), rounded to float like the reference's arrays (:7167-7172)
                const double ph = -M_PI * u * i * (i + 1) / N_ZC;
                xr[i] = (double)(float)std::cos(ph);
                xi[i] = (double)(float)std::sin(ph);
            }
            for (uint32_t k = 0; k < N_ZC; k++) {
                double   ar = 0, ai = 0;
                uint32_t t = 0;
                for (uint32_t n = 0; n < N_ZC; n++) {
                    ar += xr[n] * cs[t] - xi[n] * sn[t];
                    ai += xr[n] * sn[t] + xi[n] * cs[t];
                    t += k;
                    if (t >= N_ZC) t -= N_ZC;
                }
                xu[(size_t)r * N_ZC + k] = make_float2((float)ar, (float)ai);
            }
        }
    }
    MI_HIP_CHECK(ctx, hipMalloc((void **)&pl->d_xu_fft, sizeof(float2) * xu.size()));
    MI_HIP_CHECK(ctx, hipMemcpyAsync(pl->d_xu_fft, xu.data(), sizeof(float2) * xu.size(), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = pl;
    return MI_LTE_OK;
}

extern "C" {

int mi_lte_prach_plan_create(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *pc, mi_lte_prach_plan **out)
{
    return prach_plan_common(ctx, cfg, pc, nullptr, nullptr, 0, out);
}
int mi_lte_prach_plan_create_roots(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *pc, const float *h_xu_fft_re,
                                   const float *h_xu_fft_im, uint32_t n_roots, mi_lte_prach_plan **out)
{
    if (!h_xu_fft_re || !h_xu_fft_im || n_roots == 0) return MI_LTE_ERR_INVALID_ARG;
    return prach_plan_common(ctx, cfg, pc, h_xu_fft_re, h_xu_fft_im, n_roots, out);
}
void mi_lte_prach_plan_destroy(mi_lte_ctx *ctx, mi_lte_prach_plan *pl)
{
    if (!pl) return;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    (void)hipFree(pl->d_xu_fft);
    delete pl;
}
uint32_t mi_lte_prach_plan_n_roots(const mi_lte_prach_plan *pl) { return pl ? pl->n_roots : 0; }
uint32_t mi_lte_prach_occasion_samples(const mi_lte_prach_plan *pl) { return pl ? pl->T_cp + pl->T_fft : 0; }

int mi_lte_prach_detect_run(mi_lte_ctx *ctx, mi_lte_prach_plan *pl, const void *d_samples_a, const void *d_samples_b,
                            const uint64_t *d_occ_start, uint32_t n_occ, uint32_t *h_N_det_pre, uint32_t *h_det_pre, uint32_t *h_det_ta)
{
    if (!ctx || !pl || !d_samples_a || !d_occ_start || n_occ == 0 || !h_N_det_pre || !h_det_pre || !h_det_ta) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t xh_bytes = sizeof(float2) * (size_t)n_occ * N_ZC, co_bytes = sizeof(CorrOut) * (size_t)n_occ * pl->n_roots;
    int rc = mi_ctx_reserve_scratch(ctx, xh_bytes + co_bytes + 64);
    if (rc != MI_LTE_OK) return rc;
    float2  *d_xh = (float2 *)ctx->scratch;
    CorrOut *d_co = (CorrOut *)((char *)ctx->scratch + ((xh_bytes + 63) & ~(size_t)63));
    const dim3 g1((N_ZC + 31) / 32, n_occ);
    if (pl->cfg.sample_format == MI_LTE_IQ_I8) {
        Samp<int8_t> s{(const int8_t *)d_samples_a};
        MI_LAUNCH(ctx, "k_prach_bins", (k_prach_bins<int8_t>), g1, dim3(256), 0, s, d_occ_start, pl->T_cp, pl->T_fft, pl->start, d_xh);
    } else {
        if (!d_samples_b) return MI_LTE_ERR_INVALID_ARG;
        Samp<float> s{(const float *)d_samples_a, (const float *)d_samples_b};
        MI_LAUNCH(ctx, "k_prach_bins", (k_prach_bins<float>), g1, dim3(256), 0, s, d_occ_start, pl->T_cp, pl->T_fft, pl->start, d_xh);
    }
    MI_LAUNCH(ctx, "k_prach_corr", k_prach_corr, dim3(pl->n_roots, n_occ), dim3(256), 0, d_xh, pl->d_xu_fft, pl->n_roots, d_co);
    MI_HIP_CHECK(ctx, hipGetLastError());
    std::vector<CorrOut> co((size_t)n_occ * pl->n_roots);
    MI_HIP_CHECK(ctx, hipMemcpyAsync(co.data(), d_co, co_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t o = 0; o < n_occ; o++) { // the reference's scalar verdict (liblte_phy.cc:3436-3474)
        float    ave_val = 0, max_val = 0;
        uint32_t max_root = 0, max_offset = 0;
        for (uint32_t r = 0; r < pl->n_roots; r++) {
            const CorrOut &c = co[(size_t)o * pl->n_roots + r];
            if (c.max_val > max_val) { max_val = c.max_val; max_root = r; max_offset = c.max_off; }
            ave_val += c.sum;   // the reference adds the 839 powers one by one onto the running average ...
            ave_val /= N_ZC;    // ... and divides after every root (:3455-3457)
        }
        if (max_val >= 50 * ave_val && max_val != 0) {
            h_N_det_pre[o] = 1;
            if (pl->N_cs == 0) {
                h_det_pre[o] = max_root * (pl->v_max + 1);
                h_det_ta[o]  = (uint32_t)(int64_t)(((max_offset % N_ZC) * 29.155 / 16) - 1);
            } else {
                h_det_pre[o] = max_root * (pl->v_max + 1) + ((max_offset + pl->N_cs) % N_ZC) / pl->N_cs;
                h_det_ta[o]  = (uint32_t)(int64_t)((((pl->N_cs - ((max_offset + pl->N_cs) % N_ZC)) % pl->N_cs) * 29.155 / 16) - 1);
            }
        } else {
            h_N_det_pre[o] = 0;
            h_det_pre[o] = h_det_ta[o] = 0;
        }
    }
    ctx->last_kernels = "k_prach_bins:1,k_prach_corr:1";
    return MI_LTE_OK;
}

} // extern "C"
