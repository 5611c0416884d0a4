// Initial synchronisation on gfx950: the three searches a receiver runs before it can read a single subframe
// (LTE_fdd_dl_fs_samp_buf.cc:277-395):
//
//   liblte_phy_dl_find_coarse_timing_and_freq_offset   liblte_phy.cc:5697-5852   cyclic-prefix autocorrelation over N_slots slots
//   liblte_phy_find_pss_and_fine_timing                liblte_phy.cc:5306-5510   84 symbols x 3 PSS x 3 frequency shifts, then 80 timings
//   liblte_phy_find_sss                                liblte_phy.cc:5578-5687   1 symbol x 168 x 2 SSS
//
// The first is the dominant cost of scanning a capture (N_slots x 15360 x 144 complex MACs = 354 M for the scanner's 160 slots);
// all three are sliding correlations with no dependence between positions.  Division of labour: the device computes the
// correlation SUMS (and the FFTs under them), each in the reference's own summation order -- a serial float accumulation per
// output, -ffp-contract=off -- and the host evaluates the reference's decisions (mean gate, product of first and fourth symbol,
// peak picking with blanking, atan2f for the frequency offset, arg-max scans with strict '>' in loop order, thresholds) on
// those few thousand numbers with the reference's own expressions.  For integer-valued samples (the int8 capture format) the
// coarse-timing sums are exact integers below 2^24, so that stage is bit-identical to the reference; the PSS/SSS stages sit
// behind an FFT and agree to its tolerance, with identical decisions unless two candidates tie to 1e-6.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int N_SC_MAX = 1200;

template <typename T> struct Samples;
template <> struct Samples<int8_t> {
    const int8_t *p;
    __device__ __forceinline__ void at(size_t n, float &re, float &im) const
    {
        const char2 v = *reinterpret_cast<const char2 *>(p + 2 * n);
        re = (float)v.x; im = (float)v.y;
    }
};
template <> struct Samples<float> {
    const float *i, *q;
    __device__ __forceinline__ void at(size_t n, float &re, float &im) const { re = i[n]; im = q[n]; }
};

// corr[slot][i] = sum_{j < cp} x[b + j] * conj(x[b + j + N]), b = start + slot*n_slot + i  (liblte_phy.cc:5731-5742)
// A thread owns FOUR neighbouring windows: window i + 1 uses the samples of window i shifted by one, so a step of the (serial, reference-
// order) sum loads one new sample per side and serves four accumulators -- a quarter of the LDS reads the one-window-per-thread form made,
// which were all the kernel did (two 8-byte reads for four multiply-adds).  Element k of the staged samples sits at k + k / 4: the lanes
// of a wavefront then read at a stride of five 8-byte slots, which spreads over all banks (a stride of four would hit eight of them).
constexpr uint32_t CP_WIN = 4, CP_BLK = 256 * CP_WIN;
template <typename S>
__global__ __launch_bounds__(256) void k_cp_corr(S src, uint64_t start, uint32_t n_slot, uint32_t N, uint32_t cp, float2 *__restrict__ corr)
{
    // the 1024 windows of a block overlap: stage the 1024 + cp samples at both ends of the symbol once (cp <= 144)
    constexpr uint32_t NS = CP_BLK + 144;
    __shared__ float2 xa[NS + NS / 4 + 1], xb[NS + NS / 4 + 1];
    const uint32_t i0 = blockIdx.x * CP_BLK, slot = blockIdx.y, t = threadIdx.x;
    const size_t   b0 = start + (size_t)slot * n_slot + i0;
    const uint32_t n_win = min(CP_BLK, n_slot - i0); // windows of this block; nothing past the last one's samples is read (the capture may end there)
    for (uint32_t k = t; k < n_win + cp; k += 256) {
        float re, im;
        src.at(b0 + k, re, im);
        xa[k + (k >> 2)] = make_float2(re, im);
        src.at(b0 + k + N, re, im);
        xb[k + (k >> 2)] = make_float2(re, im);
    }
    __syncthreads();
    const uint32_t w0 = CP_WIN * t; // first window of this thread inside the block
    if (i0 + w0 >= n_slot) return;
    // (re, im) as one packed pair: (a.x b.x, a.x b.y) + (a.y b.y, -(a.y b.x)), then the sum added to the accumulator -- the same four products,
    // the same two additions per component in the same order as the scalar form (x - y and x + (-y) are the same IEEE operation), in four
    // packed instructions instead of eight: the kernel is bound by instruction issue (88 us per search at four cycles per instruction)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f    acc[CP_WIN] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    float2 a[CP_WIN], b[CP_WIN]; // samples w0 + j .. w0 + j + 3
#pragma unroll
    for (uint32_t p = 0; p < CP_WIN - 1; p++) { a[p + 1] = xa[(w0 + p) + ((w0 + p) >> 2)]; b[p + 1] = xb[(w0 + p) + ((w0 + p) >> 2)]; }
    for (uint32_t j = 0; j < cp; j++) { // serial, in the reference's order, for each of the four windows
#pragma unroll
        for (uint32_t p = 0; p < CP_WIN - 1; p++) { a[p] = a[p + 1]; b[p] = b[p + 1]; }
        const uint32_t k = w0 + j + CP_WIN - 1;
        a[CP_WIN - 1] = xa[k + (k >> 2)];
        b[CP_WIN - 1] = xb[k + (k >> 2)];
#pragma unroll
        for (uint32_t p = 0; p < CP_WIN; p++) {
            // (the operand selectors do the broadcasts, the swap and the sign: written as vector code the compiler builds (b.y, -b.x) in registers)
            const v2f av = {a[p].x, a[p].y}, bv = {b[p].x, b[p].y};
            v2f t1, t2;
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t1) : "v"(av), "v"(bv));                            // (a.x b.x,  a.x b.y)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t2) : "v"(av), "v"(bv)); // (a.y b.y, -a.y b.x)
            acc[p] += t1 + t2;
        }
    }
#pragma unroll
    for (uint32_t p = 0; p < CP_WIN; p++)
        if (i0 + w0 + p < n_slot) corr[(size_t)slot * n_slot + i0 + w0 + p] = make_float2(acc[p].x, acc[p].y);
}

// acc[i] = sum over slots, in slot order, of |corr|^2 (:5743)
__global__ __launch_bounds__(256) void k_cp_accum(const float2 *__restrict__ corr, uint32_t n_slot, uint32_t N_slots, float *__restrict__ acc)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slot) return;
    float a = 0;
    for (uint32_t s = 0; s < N_slots; s++) {
        const float2 c = corr[(size_t)s * n_slot + i];
        a += c.x * c.x + c.y * c.y;
    }
    acc[i] = a;
}

// out[p][slot] = corr[slot][pos[p]]
__global__ void k_cp_gather(const float2 *__restrict__ corr, uint32_t n_slot, uint32_t N_slots, const uint32_t *__restrict__ pos, float2 *__restrict__ out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (s < N_slots) out[(size_t)p * N_slots + s] = corr[(size_t)s * n_slot + pos[p]];
}

// out[r][q] = sum_z rows[r][z0_q + t] * conj(seq_q[t]), t = 0..61 in order (the reference sums over all sub-carriers; the others are 0)
__global__ __launch_bounds__(64) void k_seq_corr(const float *__restrict__ rows, uint32_t n_rows, const float2 *__restrict__ seq, const uint32_t *__restrict__ z0,
                                                 uint32_t n_seq, float2 *__restrict__ out)
{
    const uint32_t q = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
    if (q >= n_seq) return;
    const float  *re = rows + (size_t)r * 2 * N_SC_MAX + z0[q], *im = re + N_SC_MAX;
    const float2 *sq = seq + (size_t)q * 62;
    float cr = 0, ci = 0;
    for (uint32_t t = 0; t < 62; t++) {
        const float2 p = sq[t];
        cr += re[t] * p.x + im[t] * p.y;
        ci += re[t] * p.y - im[t] * p.x;
    }
    out[(size_t)r * n_seq + q] = make_float2(cr, ci);
}

struct Geom { uint32_t N, cp0, cpe, n_slot, n_frame, n_sc; };
int geom(const mi_lte_dl_cfg *cfg, Geom *g)
{
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048) || cfg->N_rb_dl < 6 || cfg->N_rb_dl > 100 || cfg->N_rb_dl * 12 >= N)
        return MI_LTE_ERR_INVALID_ARG;
    const uint32_t sc = 2048 / N;
    *g = Geom{N, 160 / sc, 144 / sc, 15360 / sc, 307200 / sc, cfg->N_rb_dl * 12};
    return MI_LTE_OK;
}

struct Dev { // RAII device scratch for the per-call tables
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    int get(size_t n) { return hipMalloc(&p, n ? n : 4) == hipSuccess ? 0 : -1; }
};

// generate_pss (liblte_phy.cc:8342-8368), same expressions (double argument, cosf/sinf of its float rounding)
void gen_pss(uint32_t N_id_2, float2 *pss)
{
    const float root_idx = N_id_2 == 0 ? 25 : N_id_2 == 1 ? 29 : 34;
    for (uint32_t i = 0; i < 31; i++) pss[i] = make_float2(cosf(-M_PI * root_idx * i * (i + 1) / 63), sinf(-M_PI * root_idx * i * (i + 1) / 63));
    for (uint32_t i = 31; i < 62; i++)
        pss[i] = make_float2(cosf(-M_PI * root_idx * (i + 1) * (i + 2) / 63), sinf(-M_PI * root_idx * (i + 1) * (i + 2) / 63));
}

// generate_sss (liblte_phy.cc:8377-8474): the two length-62 sequences of (N_id_1, N_id_2) for subframes 0 and 5
void gen_sss(uint32_t N_id_1, uint32_t N_id_2, float *s0, float *s5)
{
    const uint32_t q_prime = N_id_1 / 30, q = (N_id_1 + q_prime * (q_prime + 1) / 2) / 30, m_prime = N_id_1 + q * (q + 1) / 2;
    const uint32_t m0 = m_prime % 31, m1 = (m0 + m_prime / 31 + 1) % 31;
    int xs[31] = {0}, xc[31] = {0}, xz[31] = {0};
    xs[4] = xc[4] = xz[4] = 1;
    for (uint32_t i = 0; i < 26; i++) {
        xs[i + 5] = (xs[i + 2] + xs[i]) % 2;
        xc[i + 5] = (xc[i + 3] + xc[i]) % 2;
        xz[i + 5] = (xz[i + 4] + xz[i + 2] + xz[i + 1] + xz[i]) % 2;
    }
    auto st = [&](uint32_t i) { return 1 - 2 * xs[i % 31]; };
    auto ct = [&](uint32_t i) { return 1 - 2 * xc[i % 31]; };
    auto zt = [&](uint32_t i) { return 1 - 2 * xz[i % 31]; };
    for (uint32_t i = 0; i < 31; i++) {
        const int s0m0 = st(i + m0), s1m1 = st(i + m1), c0 = ct(i + N_id_2), c1 = ct(i + N_id_2 + 3), z1m0 = zt(i + (m0 % 8)), z1m1 = zt(i + (m1 % 8));
        s0[2 * i] = (float)(s0m0 * c0); s0[2 * i + 1] = (float)(s1m1 * c1 * z1m0);
        s5[2 * i] = (float)(s1m1 * c0); s5[2 * i + 1] = (float)(s0m0 * c1 * z1m1);
    }
}

template <typename F> int with_samples(const mi_lte_dl_cfg *cfg, const void *a, const void *b, F f)
{
    if (cfg->sample_format == MI_LTE_IQ_I8) return f(Samples<int8_t>{(const int8_t *)a});
    if (cfg->sample_format == MI_LTE_IQ_F32_PLANAR && b) return f(Samples<float>{(const float *)a, (const float *)b});
    return MI_LTE_ERR_INVALID_ARG;
}

// FFT rows at the given window starts, then correlations with n_seq sequences: h_out[r][q] (float2)
int fft_and_corr(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *a, const void *b, const std::vector<uint64_t> &win, const std::vector<float2> &seq,
                 const std::vector<uint32_t> &z0, std::vector<float2> &h_out)
{
    const uint32_t n_rows = (uint32_t)win.size(), n_seq = (uint32_t)z0.size();
    // carve the per-call tables out of the context's scratch (no allocation on the search path)
    struct Ptr { void *p; } d_win, d_rows, d_seq, d_z0, d_out;
    const size_t sz[5] = {8 * (size_t)n_rows, (size_t)n_rows * 2 * N_SC_MAX * 4, seq.size() * 8, 4 * (size_t)n_seq, (size_t)n_rows * n_seq * 8};
    size_t       off[5], tot = 0;
    for (int i = 0; i < 5; i++) { off[i] = tot; tot += (sz[i] + 255) & ~(size_t)255; }
    int rc0 = mi_ctx_reserve_scratch(ctx, tot);
    if (rc0 != MI_LTE_OK) return rc0;
    char *base = (char *)ctx->scratch;
    d_win.p = base + off[0]; d_rows.p = base + off[1]; d_seq.p = base + off[2]; d_z0.p = base + off[3]; d_out.p = base + off[4];
    MI_H2D(ctx, d_win.p, win.data(), 8 * n_rows);
    MI_H2D(ctx, d_seq.p, seq.data(), seq.size() * 8);
    MI_H2D(ctx, d_z0.p, z0.data(), 4 * n_seq);
    int rc = mi_fft_rows(ctx, cfg, a, b, (const uint64_t *)d_win.p, n_rows, (float *)d_rows.p);
    if (rc != MI_LTE_OK) return rc;
    MI_LAUNCH(ctx, "k_seq_corr", k_seq_corr, dim3((n_seq + 63) / 64, n_rows), dim3(64), 0, (const float *)d_rows.p, n_rows, (const float2 *)d_seq.p,
              (const uint32_t *)d_z0.p, n_seq, (float2 *)d_out.p);
    MI_HIP_CHECK(ctx, hipGetLastError());
    h_out.resize((size_t)n_rows * n_seq);
    MI_D2H(ctx, h_out.data(), d_out.p, h_out.size() * 8);
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MI_LTE_OK;
}

// The scanner's carrier-frequency correction (LTE_fdd_dl_file_scan/src/LTE_fdd_dl_fs_samp_buf.cc:696-713), in place on planar float
// samples: sample i is multiplied by exp(-j * arg), arg = (i+1)*f*2*M_PI/fs evaluated the way the C++ expression is typed -- (i+1)
// converted to float, float product with f, float x 2, then double x M_PI / fs, rounded to float for cosf / sinf.
__global__ __launch_bounds__(256) void k_freq_shift(float *__restrict__ si, float *__restrict__ sq, uint64_t first, uint64_t n, float f_off, uint32_t fs)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i   = (uint32_t)(first + k);
        const float    t   = ((float)(i + 1u) * f_off) * 2.0f;
        const float    arg = (float)((double)t * M_PI / (double)fs);
        const float    cr = cosf(arg), ci = sinf(arg);
        const float    a = si[k], b = sq[k];
        si[k] = a * cr + b * ci;
        sq[k] = b * cr - a * ci;
    }
}
// int8 I,Q interleaved (the capture file format) -> planar float, what the reference's callers do on the host (:657-694)
__global__ __launch_bounds__(256) void k_i8_to_planar(const int8_t *__restrict__ iq, uint64_t n, float *__restrict__ si, float *__restrict__ sq)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const char2 v = reinterpret_cast<const char2 *>(iq)[k];
        si[k] = (float)v.x;
        sq[k] = (float)v.y;
    }
}

// gr_complex (interleaved float32 pairs, the reference scanner's other input format, :686-692) -> planar float
__global__ __launch_bounds__(256) void k_f32_pairs_to_planar(const float2 *__restrict__ iq, uint64_t n, float *__restrict__ si, float *__restrict__ sq)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const float2 v = iq[k];
        si[k] = v.x;
        sq[k] = v.y;
    }
}

inline float mag(float2 c) { return (float)sqrt(c.x * c.x + c.y * c.y); } // abs_corr = sqrt(re*re + im*im): float expression, double sqrt

} // namespace

extern "C" {

size_t mi_lte_coarse_timing_samples(uint32_t fft_size, uint32_t N_slots)
{
    const uint32_t sc = 2048 / (fft_size ? fft_size : 2048);
    return (size_t)(N_slots + 1) * (15360 / sc) + 144 / sc + fft_size; // the last window ends at N_slots*n_slot + n_slot - 1 + cp - 1 + N
}

int mi_lte_coarse_timing_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, uint64_t start,
                             uint32_t N_slots, mi_lte_coarse_timing *out)
{
    if (!ctx || !cfg || !d_samples_a || !out || N_slots == 0) return MI_LTE_ERR_INVALID_ARG;
    Geom g;
    int  rc = geom(cfg, &g);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n_corr = (size_t)N_slots * g.n_slot;
    rc = mi_ctx_reserve_scratch(ctx, n_corr * 8 + (size_t)g.n_slot * 4 + 8 * 4 + (size_t)5 * N_slots * 8);
    if (rc != MI_LTE_OK) return rc;
    float2   *d_corr = (float2 *)ctx->scratch;
    float    *d_acc  = (float *)(d_corr + n_corr);
    uint32_t *d_pos  = (uint32_t *)(d_acc + g.n_slot);
    float2   *d_pk   = (float2 *)(d_pos + 8);
    rc = with_samples(cfg, d_samples_a, d_samples_b, [&](auto src) {
        MI_LAUNCH(ctx, "k_cp_corr", (k_cp_corr<decltype(src)>), dim3((g.n_slot + CP_BLK - 1) / CP_BLK, N_slots), dim3(256), 0, src, start, g.n_slot, g.N, g.cpe, d_corr);
        return MI_LTE_OK;
    });
    if (rc != MI_LTE_OK) return rc;
    MI_LAUNCH(ctx, "k_cp_accum", k_cp_accum, dim3((g.n_slot + 255) / 256), dim3(256), 0, (const float2 *)d_corr, g.n_slot, N_slots, d_acc);
    MI_HIP_CHECK(ctx, hipGetLastError());
    // ---- the reference's decisions on the n_slot accumulated values (:5746-5805)
    std::vector<float> c(2 * 15360 + 1, 0.0f); // dl_timing_abs_corr[LIBLTE_PHY_N_SAMPS_PER_SLOT_30_72MHZ*2] (liblte_phy.h:475)
    MI_D2H(ctx, c.data(), d_acc, (size_t)g.n_slot * 4);
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t ns = g.n_slot, sym_else = g.N + g.cpe, n_blank = sym_else / 10;
    float corr_mean = 0;
    for (uint32_t i = 0; i < ns; i++) { corr_mean += c[i]; c[i + ns] = c[i]; }
    corr_mean /= ns;
    for (uint32_t i = 0; i < ns; i++)
        if (c[i] <= corr_mean) c[i] = c[i + ns] = 0;
    const uint32_t fourth = g.N + g.cp0 + sym_else * 3; // first symbol x fourth symbol: the two CRS symbols of a slot
    for (uint32_t i = 0; i < ns; i++) c[i] *= c[fourth + i];
    int32_t  peak[5] = {0, 0, 0, 0, 0};
    uint32_t n_peaks = 5;
    for (uint32_t i = 0; i < 5; i++) {
        float best = 0;
        peak[i]    = 0;
        for (uint32_t j = 0; j < ns; j++)
            if (c[j] > best) { best = c[j]; peak[i] = (int32_t)j; }
        if (best == 0) { n_peaks = i; break; }
        int32_t t = peak[i]; // blank the peak and its six companions one symbol apart (:5783-5800)
        while (t > 0) t -= (int32_t)sym_else;
        for (uint32_t j = 0; j < 7; j++) {
            t += (int32_t)sym_else;
            for (uint32_t k = 0; k < n_blank; k++) {
                const uint32_t idx = (uint32_t)t - n_blank / 2 + k; // unsigned, as in the reference: a negative position wraps and is skipped
                if (idx < 2 * 15360) c[idx] = 0;                    // (the reference's '<=' also touches one float past its array)
            }
        }
    }
    // ---- frequency offset from the phase of the same correlations at the peaks (:5807-5828)
    memset(out, 0, sizeof(*out));
    out->n_corr_peaks = n_peaks;
    if (n_peaks) {
        uint32_t pos[5];
        for (uint32_t i = 0; i < n_peaks; i++) pos[i] = (uint32_t)peak[i];
        std::vector<float2> pk((size_t)n_peaks * N_slots);
        MI_H2D(ctx, d_pos, pos, 4 * n_peaks);
        MI_LAUNCH(ctx, "k_cp_gather", k_cp_gather, dim3((N_slots + 63) / 64, n_peaks), dim3(64), 0, (const float2 *)d_corr, g.n_slot, N_slots,
                  (const uint32_t *)d_pos, d_pk);
        MI_D2H(ctx, pk.data(), d_pk, pk.size() * 8);
        MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        float freq_err[5] = {0, 0, 0, 0, 0};
        for (uint32_t s = 0; s < N_slots; s++)
            for (uint32_t i = 0; i < n_peaks; i++) {
                const float2 v = pk[(size_t)i * N_slots + s];
                freq_err[i] += atan2f(v.y, v.x) / (g.N * 2 * M_PI * (0.0005 / g.n_slot));
            }
        for (uint32_t i = 0; i < n_peaks; i++) out->freq_offset[i] = freq_err[i] / N_slots;
    }
    for (uint32_t i = 0; i < n_peaks; i++) { // symbol start locations (:5830-5841)
        int32_t t = peak[i];
        while (t > 0) t -= (int32_t)sym_else;
        for (uint32_t j = 0; j < 7; j++) out->symb_starts[i][j] = (uint32_t)t + (j + 1) * sym_else;
    }
    ctx->last_kernels = "k_cp_corr:1,k_cp_accum:1,k_cp_gather:1";
    return MI_LTE_OK;
}

int mi_lte_find_pss_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, uint64_t start,
                        uint32_t *symb_starts /*[7], in/out*/, uint32_t *N_id_2, uint32_t *pss_symb, float *pss_thresh, float *freq_offset)
{
    if (!ctx || !cfg || !d_samples_a || !symb_starts || !N_id_2 || !pss_symb || !pss_thresh || !freq_offset) return MI_LTE_ERR_INVALID_ARG;
    Geom g;
    int  rc = geom(cfg, &g);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // nine sequences: PSS k at sub-carrier shift -1, 0, +1 (:5347-5366)
    std::vector<float2>   seq(9 * 62);
    std::vector<uint32_t> z0(9);
    for (uint32_t k = 0; k < 3; k++) {
        float2 pss[62];
        gen_pss(k, pss);
        for (uint32_t sft = 0; sft < 3; sft++) {
            std::copy(pss, pss + 62, &seq[(k * 3 + sft) * 62]);
            z0[k * 3 + sft] = g.n_sc / 2 - 31 + sft - 1;
        }
    }
    std::vector<uint64_t> win(84);
    for (uint32_t i = 0; i < 12; i++)
        for (uint32_t j = 0; j < 7; j++) win[i * 7 + j] = start + symb_starts[j] + (uint64_t)g.n_slot * i + g.cp0 - 1;
    std::vector<float2> c;
    rc = fft_and_corr(ctx, cfg, d_samples_a, d_samples_b, win, seq, z0, c);
    if (rc != MI_LTE_OK) return rc;
    float   corr_max = 0;
    int32_t shift    = 0;
    for (uint32_t r = 0; r < 84; r++)
        for (uint32_t k = 0; k < 3; k++)
            for (uint32_t sft = 0; sft < 3; sft++) { // the reference tests -1, 0, +1 in this order with a strict '>' (:5401-5421)
                const float a = mag(c[(size_t)r * 9 + k * 3 + sft]);
                if (a > corr_max) { corr_max = a; shift = (int32_t)sft - 1; *pss_symb = r; *N_id_2 = k; }
            }
    *freq_offset = 15000.0f * shift;
    // fine timing: 80 offsets around the PSS symbol (:5441-5480)
    const uint32_t N_s = *pss_symb / 7, N_symb = *pss_symb % 7;
    std::vector<uint64_t> win2(80);
    for (int32_t i = -40; i < 40; i++) {
        int32_t idx = (int32_t)(symb_starts[N_symb] + g.n_slot * N_s);
        if (i < 0) { if (idx >= -i) idx += i; } else idx += i;
        win2[i + 40] = start + (uint64_t)(uint32_t)idx + g.cp0 - 1;
    }
    std::vector<float2>   seq1(seq.begin() + (*N_id_2 * 3 + (shift + 1)) * 62, seq.begin() + (*N_id_2 * 3 + (shift + 1) + 1) * 62);
    std::vector<uint32_t> z1(1, z0[*N_id_2 * 3 + (shift + 1)]);
    rc = fft_and_corr(ctx, cfg, d_samples_a, d_samples_b, win2, seq1, z1, c);
    if (rc != MI_LTE_OK) return rc;
    corr_max      = 0;
    int8_t timing = 0;
    for (int32_t i = -40; i < 40; i++) {
        const float a = mag(c[i + 40]);
        if (a > corr_max) { corr_max = a; timing = (int8_t)i; }
    }
    *pss_thresh = corr_max;
    uint32_t t = symb_starts[N_symb] + g.n_slot * N_s + timing;
    while (t + g.N + g.cpe < g.n_slot) t += g.n_frame;
    symb_starts[0] = t + (g.N + g.cpe) * 1 - g.n_slot;
    for (uint32_t i = 1; i < 7; i++) symb_starts[i] = t + (g.N + g.cpe) * i + g.N + g.cp0 - g.n_slot;
    ctx->last_kernels = "k_sync_fft:2,k_seq_corr:2";
    return MI_LTE_OK;
}

int mi_lte_find_sss_run(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, uint64_t start,
                        uint32_t N_id_2, uint32_t *symb_starts /*[7]; [5] may be advanced by whole frames*/, float pss_thresh, uint32_t *N_id_1,
                        uint32_t *frame_start_idx, uint32_t *found)
{
    if (!ctx || !cfg || !d_samples_a || !symb_starts || !N_id_1 || !frame_start_idx || !found || N_id_2 > 2) return MI_LTE_ERR_INVALID_ARG;
    Geom g;
    int  rc = geom(cfg, &g);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<float2>   seq((size_t)336 * 62);
    std::vector<uint32_t> z0(336, g.n_sc / 2 - 31);
    for (uint32_t i = 0; i < 168; i++) {
        float s0[62], s5[62];
        gen_sss(i, N_id_2, s0, s5);
        for (uint32_t t = 0; t < 62; t++) { seq[(size_t)(2 * i) * 62 + t] = make_float2(s0[t], 0.0f); seq[(size_t)(2 * i + 1) * 62 + t] = make_float2(s5[t], 0.0f); }
    }
    std::vector<uint64_t> win(1, start + symb_starts[5] + g.cp0 - 1);
    std::vector<float2>   c;
    rc = fft_and_corr(ctx, cfg, d_samples_a, d_samples_b, win, seq, z0, c);
    if (rc != MI_LTE_OK) return rc;
    const float    sss_thresh = pss_thresh * 0.9;
    const uint32_t to_sss     = (g.N + g.cpe) * 4 + g.N + g.cp0; // from the frame (or half-frame) start to the SSS symbol
    *found = 0;
    for (uint32_t i = 0; i < 168 && !*found; i++)
        for (uint32_t h = 0; h < 2; h++) // subframe 0's sequence first, then subframe 5's (:5633-5676)
            if (mag(c[2 * i + h]) > sss_thresh) {
                const uint32_t back = to_sss + (h ? g.n_slot * 10 : 0);
                while (symb_starts[5] < back) symb_starts[5] += g.n_frame;
                *N_id_1          = i;
                *frame_start_idx = symb_starts[5] - back;
                *found           = 1;
                break;
            }
    ctx->last_kernels = "k_sync_fft:1,k_seq_corr:1";
    return MI_LTE_OK;
}

int mi_lte_freq_shift_run(mi_lte_ctx *ctx, float *d_i_samps, float *d_q_samps, uint64_t first_index, uint64_t n_samples, float freq_offset, uint32_t fs)
{
    if (!ctx || !d_i_samps || !d_q_samps || fs == 0) return MI_LTE_ERR_INVALID_ARG;
    if (n_samples == 0) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_samples + 255) / 256, 256 * 32);
    MI_LAUNCH(ctx, "k_freq_shift", k_freq_shift, dim3(grid), dim3(256), 0, d_i_samps, d_q_samps, first_index, n_samples, freq_offset, fs);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_freq_shift:1";
    return MI_LTE_OK;
}

int mi_lte_iq_i8_to_planar(mi_lte_ctx *ctx, const int8_t *d_iq, uint64_t n_samples, float *d_i_samps, float *d_q_samps)
{
    if (!ctx || !d_iq || !d_i_samps || !d_q_samps) return MI_LTE_ERR_INVALID_ARG;
    if (n_samples == 0) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_samples + 255) / 256, 256 * 32);
    MI_LAUNCH(ctx, "k_i8_to_planar", k_i8_to_planar, dim3(grid), dim3(256), 0, d_iq, n_samples, d_i_samps, d_q_samps);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_i8_to_planar:1";
    return MI_LTE_OK;
}

int mi_lte_iq_f32_pairs_to_planar(mi_lte_ctx *ctx, const float *d_iq, uint64_t n_samples, float *d_i_samps, float *d_q_samps)
{
    if (!ctx || !d_iq || !d_i_samps || !d_q_samps) return MI_LTE_ERR_INVALID_ARG;
    if (n_samples == 0) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_samples + 255) / 256, 256 * 32);
    MI_LAUNCH(ctx, "k_f32_pairs_to_planar", k_f32_pairs_to_planar, dim3(grid), dim3(256), 0, reinterpret_cast<const float2 *>(d_iq), n_samples, d_i_samps, d_q_samps);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_f32_pairs_to_planar:1";
    return MI_LTE_OK;
}

} // extern "C"
