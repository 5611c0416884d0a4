// PRACH detection on gfx950 (the rest of SURVEY 8f N1 / BASELINE config 5): restates liblte_phy_detect_prach
// (liblte/src/liblte_phy.cc:3299-3479) for preamble formats 0-4 and a batch of PRACH occasions.  (Format 4 -- the TDD UpPTS
// preamble: N_zc = 139, T_fft = 4 096, 7.5 kHz sub-carriers, :2462-2468 -- runs through the same three kernels with 4 decimation phases
// instead of 24 and its own chirp tables; the text below describes formats 0-3.)
//
// The reference transforms the T_fft = 24 576 samples behind the cyclic prefix with one FFT and keeps N_zc = 839 bins
// (:3421-3433), multiplies them with the conjugate spectrum of each root sequence and goes back with an 839-point
// inverse DFT per root (:3438-3458), then looks for one peak >= 50 x the (recursively averaged) correlation power.
//
//   k_prach_fft + k_prach_bins : only those 839 bins are needed.  T_fft = 24 x N2 (N2 = 64 ... 1024), so the samples are split
//                  into their 24 decimation phases, each gets an N2-point FFT in LDS, and every kept bin is a 24-term
//                  combination of one output of each (exact decimation-in-time identity; no 24 576-point FFT).
//   k_prach_corr : one workgroup per (occasion, root): spectrum product, 839-point inverse DFT (839 is prime: chirp-z transform
//                  with two 2048-point FFTs in LDS), per-root sum / maximum / arg-maximum of the correlation power.
// The verdict (threshold, preamble index, timing advance) is the reference's scalar arithmetic on the host (:3460-3474).
// FFTW's rounding is unspecified, so the correlation powers are tolerance-level; the outputs are integers.
#include <cmath>
#include <vector>

#include "ctx.hpp"
#include "lte_tables.h"
#include "prach_sets.hpp"

namespace {

constexpr uint32_t N_ZC_MAX = 839; // formats 0-3; format 4: 139 (every kernel takes the length as an argument)

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

// Forward FFT of N = 2^n points (N <= 2048) in LDS, unnormalised: radix-4 Stockham passes between two buffers (and one radix-2 pass when n is
// odd), twiddles from tw[k] = exp(-2*pi*i*k/2048), k < 1024 (in LDS: w and w^2 are read, w^3 is their product).  Returns the buffer that
// holds the result.  (Until late in round 4 these were radix-2 passes -- eleven LDS round trips for 2048 points -- and k_prach_fft
// evaluated sincospif per butterfly: 40 of its 55 instructions.)
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 *lds_fft_pow2(float2 *in, float2 *out, uint32_t N, const float2 *tw)
{
    uint32_t Ns = 1;
    for (; N / Ns >= 4; Ns <<= 2) {
        const uint32_t nb = N >> 2, st = 2048u / (Ns << 2);
        for (uint32_t j = threadIdx.x; j < nb; j += blockDim.x) {
            const uint32_t k = j & (Ns - 1);
            float2 a = in[j], b = in[j + nb], c = in[j + 2 * nb], d = in[j + 3 * nb];
            if (Ns > 1) {
                const float2 w1 = tw[k * st], w2 = tw[2 * k * st];
                b = cmulf(b, w1); c = cmulf(c, w2); d = cmulf(d, cmulf(w1, w2));
            }
            const float2 s02 = make_float2(a.x + c.x, a.y + c.y), d02 = make_float2(a.x - c.x, a.y - c.y);
            const float2 s13 = make_float2(b.x + d.x, b.y + d.y), d13 = make_float2(b.y - d.y, d.x - b.x); // -i (b - d)
            const uint32_t j0 = ((j - k) << 2) + k;
            out[j0]          = make_float2(s02.x + s13.x, s02.y + s13.y);
            out[j0 + Ns]     = make_float2(d02.x + d13.x, d02.y + d13.y);
            out[j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
            out[j0 + 3 * Ns] = make_float2(d02.x - d13.x, d02.y - d13.y);
        }
        __syncthreads();
        float2 *t = in; in = out; out = t;
    }
    if (Ns < N) { // one radix-2 pass left (Ns = N / 2)
        const uint32_t st = 2048u / (Ns << 1);
        for (uint32_t j = threadIdx.x; j < N / 2; j += blockDim.x) {
            const uint32_t k = j & (Ns - 1);
            const float2   u = in[j], v = cmulf(in[j + N / 2], tw[k * st]);
            const uint32_t j0 = ((j - k) << 1) + k;
            out[j0]      = make_float2(u.x + v.x, u.y + v.y);
            out[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2 *t = in; in = out; out = t;
    }
    return in;
}

// x_hat[occ][b] = sum_n x[n] exp(-2*pi*i*n*idx_b/T), idx_b = (b + start + T/2) mod T        (liblte_phy.cc:3421-3433)
// T = 24 * N2 with N2 a power of two (64 ... 1024), so the 839 bins come from the decimation-in-time split n = 24 m + r:
//     X[k] = sum_{r < 24} exp(-2*pi*i*r*k/T) * F_r[k mod N2],   F_r = N2-point DFT of x[24 m + r]
// k_prach_fft: one workgroup per (occasion, r), radix-2 Stockham passes in LDS; k_prach_bins: the 24-term combination for the
// 839 bins that are kept.  (The first version summed every bin directly: 24 576 x 839 complex MACs per occasion, 8.5 us at 20 MHz.)
template <typename T>
__global__ __launch_bounds__(256) void k_prach_fft(Samp<T> src, const uint64_t *__restrict__ occ_start, uint32_t T_cp, uint32_t N2, uint32_t P,
                                                   const float2 *__restrict__ tw_g, float2 *__restrict__ F)
{
    extern __shared__ float2 lds[]; // two buffers of N2 | tw[1024]
    const uint32_t occ = blockIdx.y, r = blockIdx.x;
    const size_t   first = occ_start[occ] + T_cp;
    float2 *in = lds, *out = lds + N2, *tw = lds + 2 * N2;
    for (uint32_t k = threadIdx.x; k < 1024; k += blockDim.x) tw[k] = tw_g[k];
    for (uint32_t i = threadIdx.x; i < N2; i += blockDim.x) in[i] = src.at(first + (size_t)P * i + r);
    __syncthreads();
    const float2 *res = lds_fft_pow2(in, out, N2, tw);
    float2 *dst = F + ((size_t)occ * P + r) * N2;
    for (uint32_t i = threadIdx.x; i < N2; i += blockDim.x) dst[i] = res[i];
}

__global__ __launch_bounds__(256) void k_prach_bins(const float2 *__restrict__ F, uint32_t N2, uint32_t P, uint32_t T_fft, uint32_t start, uint32_t N_ZC,
                                                    float2 *__restrict__ x_hat)
{
    const uint32_t occ = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
    if (b >= N_ZC) return;
    const uint32_t idx = (b + start + T_fft / 2) % T_fft, km = idx & (N2 - 1);
    const float2  *f = F + (size_t)occ * P * N2 + km;
    float ar = 0.f, ai = 0.f;
    for (uint32_t r = 0; r < P; r++) {
        const uint32_t t = (r * idx) % T_fft; // r * idx < 24 * 24576: no overflow
        float ws, wc;
        sincospif(-2.0f * (float)t / (float)T_fft, &ws, &wc);
        const float2 v = f[(size_t)r * N2];
        ar += v.x * wc - v.y * ws;
        ai += v.x * ws + v.y * wc;
    }
    x_hat[(size_t)occ * N_ZC + b] = make_float2(ar, ai);
}

struct CorrOut { float sum, max_val; uint32_t max_off, pad; };

// per (occasion, root): in[j] = X_u[j] * conj(x_hat[j]); out[m] = sum_j in[j] exp(+2*pi*i*j*m/839); power statistics.
// 839 is prime, so the inverse DFT is done as a chirp-z (Bluestein) convolution with two 2048-point FFTs in LDS:
//     j*m = (j^2 + m^2 - (m-j)^2)/2   =>   out[m] = c[m] * sum_j (in[j] c[j]) * conj(c)[m-j],   c[n] = exp(+pi*i*n^2/839)
// the convolution with the fixed sequence conj(c) is a product with its precomputed 2048-point spectrum (plan: d_bspec), the chirp
// comes from a table with exactly reduced angles (plan: d_chirp).  16x fewer operations than the 839 x 839 direct sums it replaces
// (2.8 -> 0.3 ms for 1638 occasions x 8 roots).
constexpr uint32_t BL = 2048;

__global__ __launch_bounds__(256) void k_prach_corr(const float2 *__restrict__ x_hat, const float2 *__restrict__ xu_fft, uint32_t n_roots, uint32_t N_ZC,
                                                    const float2 *__restrict__ chirp, const float2 *__restrict__ bspec, const float2 *__restrict__ tw_g,
                                                    CorrOut *__restrict__ out)
{
    extern __shared__ float2 lds[]; // a[BL] | b[BL] | tw[BL/2]
    __shared__ float    r_sum[4], r_max[4];
    __shared__ uint32_t r_off[4];
    float2 *A = lds, *B = lds + BL, *tw = lds + 2 * BL;
    const uint32_t occ = blockIdx.y, root = blockIdx.x;
    for (uint32_t k = threadIdx.x; k < BL / 2; k += blockDim.x) tw[k] = tw_g[k];
    for (uint32_t j = threadIdx.x; j < BL; j += blockDim.x) {
        float2 v = make_float2(0.f, 0.f);
        if (j < N_ZC) {
            const float2 u = xu_fft[(size_t)root * N_ZC + j], h = x_hat[(size_t)occ * N_ZC + j], c = chirp[j];
            const float2 in = make_float2(u.x * h.x + u.y * h.y, u.y * h.x - u.x * h.y); // liblte_phy.cc:3443-3444
            v = make_float2(in.x * c.x - in.y * c.y, in.x * c.y + in.y * c.x);
        }
        A[j] = v;
    }
    __syncthreads();
    float2 *X = lds_fft_pow2(A, B, BL, tw), *Y = (X == A) ? B : A;
    // product with the chirp filter's spectrum, conjugated so that the same forward FFT performs the inverse transform
    for (uint32_t k = threadIdx.x; k < BL; k += blockDim.x) {
        const float2 x = X[k], g = bspec[k];
        Y[k] = make_float2(x.x * g.x - x.y * g.y, -(x.x * g.y + x.y * g.x));
    }
    __syncthreads();
    float2 *Z = lds_fft_pow2(Y, X, BL, tw); // = conj(BL * convolution)
    float    sum = 0.f, mx = -1.f;
    uint32_t off = 0;
    for (uint32_t m = threadIdx.x; m < N_ZC; m += blockDim.x) { // ascending m per thread: the first maximum wins
        const float2 z = Z[m], c = chirp[m];
        const float  zr = z.x * (1.0f / BL), zi = -z.y * (1.0f / BL);
        const float  ar = zr * c.x - zi * c.y, ai = zr * c.y + zi * c.x;
        const float  p = ar * ar + ai * ai;
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
    uint32_t      T_fft = 0, T_cp = 0, start = 0, N_cs = 0, v_max = 0, n_roots = 0, n_zc = 839, phases = 24;
    float2       *d_xu_fft = nullptr;
    float2       *d_chirp = nullptr, *d_bspec = nullptr, *d_tw = nullptr; // Bluestein tables of the 839-point inverse DFT (owned by the context)
    // mi_lte_prach_detect_launch / _fetch: the per-root maxima of the launch that is in flight, in pinned host memory of the plan's own
    void      *h_pending = nullptr;
    size_t     pending_cap = 0;
    uint32_t   pending_occ = 0;
    hipEvent_t pending_done = nullptr;
};

// Bluestein tables of the 839-point inverse DFT, built once per context (they depend on nothing but 839 and 2048):
// chirp c[n] = exp(+pi*i*n^2/839) with n^2 reduced mod 2*839 in integers; the 2048-point spectrum of the wrapped filter conj(c)
// (double-precision radix-2 FFT on the host); the FFT twiddles.
static int prach_bluestein_tables(mi_lte_ctx *ctx, uint32_t N_ZC, float2 **d_chirp, float2 **d_bspec, float2 **d_tw)
{
    float2 *&cached = N_ZC == 839 ? ctx->d_prach_tab : ctx->d_prach_tab4; // one set per sequence length (839: formats 0-3, 139: format 4)
    if (cached) {
        *d_chirp = cached; *d_bspec = cached + 1024; *d_tw = cached + 1024 + BL;
        return MI_LTE_OK;
    }
    const double PI = 3.14159265358979323846;
    std::vector<float2> tab(1024 + BL + BL / 2);
    std::vector<double> br(BL, 0.0), bi(BL, 0.0);
    for (uint32_t n = 0; n < N_ZC; n++) {
        const double ang = PI * (double)((uint64_t)n * n % (2 * N_ZC)) / N_ZC;
        tab[n] = make_float2((float)cos(ang), (float)sin(ang));
        br[n] = cos(ang); bi[n] = -sin(ang);
        if (n) { br[BL - n] = br[n]; bi[BL - n] = bi[n]; }
    }
    // in-place decimation-in-time FFT: bit-reversal permutation, then 11 butterfly stages
    for (uint32_t i = 0, j = 0; i < BL; i++) {
        if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
        uint32_t m = BL >> 1;
        while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
        j |= m;
    }
    for (uint32_t len = 2; len <= BL; len <<= 1)
        for (uint32_t i = 0; i < BL; i += len)
            for (uint32_t k = 0; k < len / 2; k++) {
                const double a = -2 * PI * k / len, wr = cos(a), wi = sin(a);
                const double xr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi, xi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
                br[i + k + len / 2] = br[i + k] - xr; bi[i + k + len / 2] = bi[i + k] - xi;
                br[i + k] += xr; bi[i + k] += xi;
            }
    for (uint32_t k = 0; k < BL; k++) tab[1024 + k] = make_float2((float)br[k], (float)bi[k]);
    for (uint32_t k = 0; k < BL / 2; k++) tab[1024 + BL + k] = make_float2((float)cos(-2 * PI * k / BL), (float)sin(-2 * PI * k / BL));
    float2 *d = nullptr;
    MI_HIP_CHECK(ctx, hipMalloc((void **)&d, sizeof(float2) * tab.size()));
    ctx->owned.push_back(d);
    MI_H2D(ctx, d, tab.data(), sizeof(float2) * tab.size());
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    cached = d;
    *d_chirp = d; *d_bspec = d + 1024; *d_tw = d + 1024 + BL;
    return MI_LTE_OK;
}

extern "C" void mi_lte_prach_plan_destroy(mi_lte_ctx *ctx, mi_lte_prach_plan *pl);
static int prach_plan_common(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const mi_lte_prach_cfg *pc, const float *h_xu_fft_re,
                             const float *h_xu_fft_im, uint32_t n_roots_given, mi_lte_prach_plan **out)
{
    if (!ctx || !cfg || !pc || !out) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048) || cfg->N_rb_dl * 12 >= N) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t   fmt = pc->preamble_format;
    const PrachGeom  pg  = prach_geom(fmt);
    const uint32_t   N_ZC = pg.n_zc;
    if (fmt > 4 || (fmt == 4 ? pc->zczc > 6u : pc->zczc > (pc->hs_flag ? 14u : 15u)) || pc->root_seq_idx >= pg.n_root_idx) {
        ctx->err = "PRACH: preamble format, zeroCorrelationZoneConfig or root index out of range";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto *pl = new mi_lte_prach_plan();
    auto  guard = on_fail([&] { (void)hipStreamSynchronize(ctx->stream); mi_lte_prach_plan_destroy(nullptr, pl); });
    pl->cfg  = *cfg;
    const uint32_t sc = 2048 / N; // 30.72 MHz / fs
    pl->n_zc = N_ZC; pl->phases = pg.phases;
    pl->T_fft = pg.T_fft_30 / sc;                    // liblte_phy.cc:2430-2470
    pl->T_cp  = pg.T_cp_30 / sc;
    const uint32_t k_0 = pc->freq_offset * 12 - cfg->N_rb_dl * 12 / 2 + N / 2, K = pg.K; // :3416-3418 (uint32 arithmetic); K = 15000 / delta_f_RA
    pl->start = pg.phi + K * k_0 + K / 2;            // :3426
    {   // N_cs and v_max of the FIRST root, as liblte_phy_detect_prach uses them for every root (:3336-3413)
        const PrachSets ps = prach_sets(prach_root(fmt, pc->root_seq_idx), pc->zczc, pc->hs_flag != 0, fmt);
        if (!ps.ok) {
            ctx->err = "PRACH: zeroCorrelationZoneConfig / root outside what the reference can process (restricted set: config 15, or a root without a cyclic shift)";
            return MI_LTE_ERR_UNSUPPORTED;
        }
        pl->N_cs = ps.N_cs; pl->v_max = ps.v_max;
    }
    std::vector<float2> xu;
    if (h_xu_fft_re && h_xu_fft_im) { // the caller's spectra (what liblte_phy_ul_init left in LIBLTE_PHY_STRUCT: rows of 839, the first N_zc used)
        pl->n_roots = n_roots_given;
        xu.resize((size_t)pl->n_roots * N_ZC);
        for (uint32_t r = 0; r < pl->n_roots; r++)
            for (uint32_t k = 0; k < N_ZC; k++) xu[(size_t)r * N_ZC + k] = make_float2(h_xu_fft_re[(size_t)r * N_ZC_MAX + k], h_xu_fft_im[(size_t)r * N_ZC_MAX + k]);
    } else { // roots needed for 64 preambles (prach_preamble_seq_gen, :7130-7290), their forward DFTs in double
        // (prach_sets.hpp: the logical root order is cyclic, a set that starts near the end of the table wraps; the reference reads past its table)
        const PrachRootSet rs = prach_root_set(fmt, pc->root_seq_idx, pc->zczc, pc->hs_flag != 0);
        if (!rs.ok) {
            ctx->err = "PRACH: a root of the 64-preamble set has no cyclic shift in the restricted set (the reference divides by zero there)";
            return MI_LTE_ERR_UNSUPPORTED;
        }
        pl->n_roots = rs.n_roots;
        xu.resize((size_t)pl->n_roots * N_ZC);
        std::vector<double> xr(N_ZC), xi(N_ZC), cs(N_ZC), sn(N_ZC);
        for (uint32_t t = 0; t < N_ZC; t++) { cs[t] = std::cos(-2.0 * M_PI * t / N_ZC); sn[t] = std::sin(-2.0 * M_PI * t / N_ZC); }
        for (uint32_t r = 0; r < pl->n_roots; r++) {
            const uint32_t u = rs.u[r];
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
    MI_H2D(ctx, pl->d_xu_fft, xu.data(), sizeof(float2) * xu.size());
    int rcb = prach_bluestein_tables(ctx, N_ZC, &pl->d_chirp, &pl->d_bspec, &pl->d_tw);
    if (rcb != MI_LTE_OK) return rcb;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    guard.armed = false;
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
    if (pl->h_pending) (void)hipHostFree(pl->h_pending);
    if (pl->pending_done) (void)hipEventDestroy(pl->pending_done);
    delete pl;
}
uint32_t mi_lte_prach_plan_n_roots(const mi_lte_prach_plan *pl) { return pl ? pl->n_roots : 0; }
uint32_t mi_lte_prach_occasion_samples(const mi_lte_prach_plan *pl) { return pl ? pl->T_cp + pl->T_fft : 0; }

// The kernels of one batch of occasions and where their per-root maxima go: *h_co = pinned host memory the kernel wrote itself (a few
// occasions), or d_co in the context's scratch for the caller to copy
static int prach_launch(mi_lte_ctx *ctx, mi_lte_prach_plan *pl, const void *d_samples_a, const void *d_samples_b, const uint64_t *d_occ_start, uint32_t n_occ,
                        bool small_results, CorrOut **h_co_out, CorrOut **d_co_out, size_t *co_bytes_out)
{
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t N_ZC = pl->n_zc, P = pl->phases;
    const size_t xh_bytes = sizeof(float2) * (size_t)n_occ * N_ZC, co_bytes = sizeof(CorrOut) * (size_t)n_occ * pl->n_roots;
    const size_t f_off = (xh_bytes + co_bytes + 64 + 255) & ~(size_t)255, f_bytes = sizeof(float2) * (size_t)n_occ * pl->T_fft; // P x N2 per occasion
    int rc = mi_ctx_reserve_scratch(ctx, f_off + f_bytes);
    if (rc != MI_LTE_OK) return rc;
    float2  *d_xh = (float2 *)ctx->scratch;
    // the per-root maxima of a few occasions go straight into pinned host memory (no copy command for a per-call caller); a batch's through scratch
    CorrOut *d_co = (CorrOut *)((char *)ctx->scratch + ((xh_bytes + 63) & ~(size_t)63)), *h_co = nullptr;
    if (!small_results || mi_ctx_small_results(ctx, co_bytes, (void **)&h_co, (void **)&d_co) != MI_LTE_OK) {
        h_co = nullptr;
        d_co = (CorrOut *)((char *)ctx->scratch + ((xh_bytes + 63) & ~(size_t)63));
    }
    const uint32_t N2 = pl->T_fft / P;
    if (pl->T_fft != P * N2 || (N2 & (N2 - 1)) || N2 < 2 || N2 > 1024) { ctx->err = "PRACH: T_fft must be 24 (format 4: 4) x a power of two"; return MI_LTE_ERR_UNSUPPORTED; }
    float2 *d_F = (float2 *)((char *)ctx->scratch + f_off);
    if (pl->cfg.sample_format == MI_LTE_IQ_I8) {
        Samp<int8_t> s{(const int8_t *)d_samples_a};
        MI_LAUNCH(ctx, "k_prach_fft", (k_prach_fft<int8_t>), dim3(P, n_occ), dim3(256), (2 * N2 + 1024) * sizeof(float2), s, d_occ_start, pl->T_cp, N2, P, (const float2 *)pl->d_tw, d_F);
    } else {
        if (!d_samples_b) return MI_LTE_ERR_INVALID_ARG;
        Samp<float> s{(const float *)d_samples_a, (const float *)d_samples_b};
        MI_LAUNCH(ctx, "k_prach_fft", (k_prach_fft<float>), dim3(P, n_occ), dim3(256), (2 * N2 + 1024) * sizeof(float2), s, d_occ_start, pl->T_cp, N2, P, (const float2 *)pl->d_tw, d_F);
    }
    MI_LAUNCH(ctx, "k_prach_bins", k_prach_bins, dim3((N_ZC + 255) / 256, n_occ), dim3(256), 0, (const float2 *)d_F, N2, P, pl->T_fft, pl->start, N_ZC, d_xh);
    MI_LAUNCH(ctx, "k_prach_corr", k_prach_corr, dim3(pl->n_roots, n_occ), dim3(256), sizeof(float2) * (2 * BL + BL / 2), d_xh, pl->d_xu_fft, pl->n_roots, N_ZC,
              (const float2 *)pl->d_chirp, (const float2 *)pl->d_bspec, (const float2 *)pl->d_tw, d_co);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_prach_bins:1,k_prach_corr:1";
    *h_co_out = h_co; *d_co_out = d_co; *co_bytes_out = co_bytes;
    return MI_LTE_OK;
}

// the reference's scalar verdict per occasion (liblte_phy.cc:3436-3474) from the per-root maxima
static void prach_verdicts(const mi_lte_prach_plan *pl, const CorrOut *co, uint32_t n_occ, uint32_t *h_N_det_pre, uint32_t *h_det_pre, uint32_t *h_det_ta)
{
    const uint32_t N_ZC = pl->n_zc;
    for (uint32_t o = 0; o < n_occ; o++) {
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
}

int mi_lte_prach_detect_run(mi_lte_ctx *ctx, mi_lte_prach_plan *pl, const void *d_samples_a, const void *d_samples_b,
                            const uint64_t *d_occ_start, uint32_t n_occ, uint32_t *h_N_det_pre, uint32_t *h_det_pre, uint32_t *h_det_ta)
{
    if (!ctx || !pl || !d_samples_a || !d_occ_start || n_occ == 0 || !h_N_det_pre || !h_det_pre || !h_det_ta) return MI_LTE_ERR_INVALID_ARG;
    CorrOut *h_co = nullptr, *d_co = nullptr;
    size_t   co_bytes = 0;
    int rc = prach_launch(ctx, pl, d_samples_a, d_samples_b, d_occ_start, n_occ, true, &h_co, &d_co, &co_bytes);
    if (rc != MI_LTE_OK) return rc;
    std::vector<CorrOut> co_copy;
    if (!h_co) {
        co_copy.resize((size_t)n_occ * pl->n_roots);
        MI_D2H(ctx, co_copy.data(), d_co, co_bytes);
    }
    MI_HIP_CHECK(ctx, h_co ? mi_stream_wait(ctx, n_occ) : hipStreamSynchronize(ctx->stream)); // (a copy into pageable memory is done when the runtime says so)
    prach_verdicts(pl, h_co ? h_co : co_copy.data(), n_occ, h_N_det_pre, h_det_pre, h_det_ta);
    return MI_LTE_OK;
}

// The same in two halves for a caller that keeps the stream busy: _launch queues the kernels and the copy of the per-root maxima into pinned
// memory of the plan's own and returns; _fetch waits for THAT copy (an event behind it, not the stream: whatever the caller queued since keeps
// running) and forms the verdicts.  One launch in flight per plan (a second one before the fetch would overwrite the first's results:
// MI_LTE_ERR_INVALID_ARG).
int mi_lte_prach_detect_launch(mi_lte_ctx *ctx, mi_lte_prach_plan *pl, const void *d_samples_a, const void *d_samples_b, const uint64_t *d_occ_start,
                               uint32_t n_occ)
{
    if (!ctx || !pl || !d_samples_a || !d_occ_start || n_occ == 0 || pl->pending_occ) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t need = sizeof(CorrOut) * (size_t)n_occ * pl->n_roots;
    if (need > pl->pending_cap) {
        if (pl->h_pending) (void)hipHostFree(pl->h_pending);
        pl->h_pending = nullptr; pl->pending_cap = 0;
        MI_HIP_CHECK(ctx, hipHostMalloc(&pl->h_pending, need, hipHostMallocMapped));
        pl->pending_cap = need;
    }
    CorrOut *h_co = nullptr, *d_co = nullptr;
    size_t   co_bytes = 0;
    int rc = prach_launch(ctx, pl, d_samples_a, d_samples_b, d_occ_start, n_occ, false, &h_co, &d_co, &co_bytes);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_device_to_pinned(ctx, pl->h_pending, d_co, co_bytes));
    if (!pl->pending_done) MI_HIP_CHECK(ctx, hipEventCreateWithFlags(&pl->pending_done, hipEventDisableTiming));
    MI_HIP_CHECK(ctx, hipEventRecord(pl->pending_done, ctx->stream));
    pl->pending_occ = n_occ;
    return MI_LTE_OK;
}

int mi_lte_prach_detect_fetch(mi_lte_ctx *ctx, mi_lte_prach_plan *pl, uint32_t *h_N_det_pre, uint32_t *h_det_pre, uint32_t *h_det_ta, uint32_t max_occ)
{
    if (!ctx || !pl || !h_N_det_pre || !h_det_pre || !h_det_ta || pl->pending_occ == 0 || max_occ < pl->pending_occ) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MI_HIP_CHECK(ctx, hipEventSynchronize(pl->pending_done));
    prach_verdicts(pl, (const CorrOut *)pl->h_pending, pl->pending_occ, h_N_det_pre, h_det_pre, h_det_ta);
    pl->pending_occ = 0;
    return MI_LTE_OK;
}

} // extern "C"
