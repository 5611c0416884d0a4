// DL front end on gfx950: OFDM demodulation (15 or 16 FFTs per subframe) + CRS channel estimation.
// Restates liblte_phy_get_dl_subframe_and_ce (liblte/src/liblte_phy.cc:5905-6200) for a batch of
// independent subframe units.  HBM-bound by design: int8 IQ in, fp32 symbols + estimates out.
//
//   k_dl_fft  : one workgroup per (unit, OFDM symbol).  N-point Stockham FFT through LDS, radix-8
//               passes in registers, int8->fp32 conversion fused into the first pass's loads and the
//               guard-band drop / spectrum un-shift fused into the last pass's stores
//               (samples_to_symbols_dl, liblte_phy.cc:8593-8644, scale = 0).  N = 2048 (20 MHz) runs
//               k_dl_fft2k: 8 x 16 x 16 with 128 threads per symbol, two LDS exchanges (round 4).
//   k_dl_ce   : one workgroup per (unit, antenna port): pilot LS estimates, the reference's
//               sequential phase unwrap, frequency interpolation, then time interpolation and
//               mag/phase -> re/im for all 14 symbols (liblte_phy.cc:5959-6194).
#include <type_traits>

#include "ctx.hpp"
#include "phy_dev.hpp"

namespace {

constexpr int N_SC_MAX = 1200; // row stride of every subframe array (LIBLTE_PHY_N_RB_DL_20MHZ * 12)

struct DlGeom {
    uint32_t N;        // FFT size = samples per symbol
    uint32_t cp0, cpe; // cyclic prefix of symbol 0 / other symbols
    uint32_t n_slot;   // samples per slot
    uint32_t half;     // used sub-carriers per side = 6 * N_rb_dl
    uint32_t N_rb_dl;
    uint32_t N_ant;
    uint32_t sf_stride; // floats per device subframe
    uint32_t ul;        // 1: uplink SC-FDMA demodulation (samples_to_symbols_ul, liblte_phy.cc:8654-8692)
    uint32_t ce_compact; // 1: the estimator stops after the frequency direction and writes magnitude / phase rows (MI_LTE_CE_COMPACT)
    float    r_n_pil, r_nq; // the floats next above 1 / (2 N_rb_dl) and 1 / (3 N_rb_dl): k_dl_ce's row / column splits (quot_f below)
};

// floor(n / d) for n < 2^20 as the truncated float product n * r, r = the float next above or equal to 1 / d: n r >= n / d, so a multiple of d
// never lands below its quotient, and the excess (n / d) 2^-22 stays under the 1 / d that separates n / d from the next integer.  Three
// full-rate instructions; the compiler's own sequence for a division by a run-time value is ~25 with quarter-rate multiplies in it.
__device__ __forceinline__ uint32_t quot_f(uint32_t n, float r) { return (uint32_t)((float)n * r); }
static float recip_up(uint32_t d)
{
    const double rd = 1.0 / (double)d;
    float        r  = (float)rd;
    if ((double)r < rd) r = nextafterf(r, 2.0f);
    return r;
}
__device__ __forceinline__ uint32_t mod6_lt12(uint32_t x) { return x >= 6 ? x - 6 : x; } // x % 6 for x < 12

// The FFT has no bit-exact reference (FFTW's operation order is unspecified; parity is to tolerance), so its
// complex multiplies may use fused multiply-adds; the rest of the library is built with -ffp-contract=off.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f    to_v(float2 a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ float2 to_f2(v2f a) { return __builtin_bit_cast(float2, a); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x) as one packed multiply and one packed fused multiply-add, the swap and the sign taken by the
    // second instruction's own operand selectors: written as vector code the compiler spends a v_xor and a v_mov per product on building
    // (-b.y, b.x) -- 60 of the 2048-point transform's 390 vector instructions, in a kernel whose time is three quarters instruction issue
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(to_v(a)), "v"(to_v(b)));                                              // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(to_v(a)), "v"(to_v(b)), "v"(t)); // (-a.y b.y, a.y b.x) + t
    return to_f2(r);
}
// the same product with a wave-uniform b (the fixed twiddles inside a butterfly): b may sit in a scalar register pair, so that nobody
// has to move a constant into vector registers first (a packed instruction issues in 4 cycles with or without a scalar operand)
__device__ __forceinline__ float2 cmul_u(float2 a, float2 b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(to_v(a)), "s"(to_v(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(to_v(a)), "s"(to_v(b)), "v"(t));
    return to_f2(r);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); } // a * (-i)

// forward DFTs of size 2/4/8 on registers, natural-order output
__device__ __forceinline__ void dft2(float2 &a, float2 &b) { float2 t = a; a = cadd(t, b); b = csub(t, b); }
__device__ __forceinline__ void dft4(float2 *v)
{
    float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(a, c); v[1] = cadd(b, d); v[2] = csub(a, c); v[3] = csub(b, d);
}
__device__ __forceinline__ void dft8(float2 *v)
{
    const float h = 0.70710678118654752440f;
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft4(e);
    dft4(o);
    o[1] = cmul(o[1], make_float2(h, -h)); // W8^1
    o[2] = mul_mi(o[2]);                   // W8^2
    o[3] = cmul(o[3], make_float2(-h, -h)); // W8^3
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
}
template <int R> __device__ __forceinline__ void dftR(float2 *v)
{
    if (R == 8) dft8(v);
    else if (R == 4) dft4(v);
    else dft2(v[0], v[1]);
}

// Twiddles w^1 .. w^(R-1) of butterfly k of a radix-R pass, w = exp(-2*pi*i*k/(Ns*R)): w, w^2, w^4 come from the pass's own table
// (ctx.hpp MI_FFT_TWC_*: R / 2 float2 slots per butterfly, side by side), the rest are products.
template <int R> __device__ __forceinline__ void twiddles(const float2 *__restrict__ tc, uint32_t k, float2 (&w)[8])
{
    // (tc is uniform over the workgroup: scalar base + the thread's 32-bit byte offset, not a 64-bit sum per read)
    const char *tb = reinterpret_cast<const char *>(tc);
    if (R == 2) w[1] = *reinterpret_cast<const float2 *>(tb + (size_t)(k * 8u));
    else {
        const float4 a = *reinterpret_cast<const float4 *>(tb + (size_t)(k * (R / 2) * 8u));
        w[1] = make_float2(a.x, a.y);
        w[2] = make_float2(a.z, a.w);
        w[3] = cmul(w[1], w[2]);
        if (R == 8) {
            w[4] = *reinterpret_cast<const float2 *>(tb + 16 + (size_t)(k * 32u));
            w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
        }
    }
}

// LDS index padding: one extra float2 slot every 32 breaks the power-of-two strides of the passes
__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + (i >> 5); }

// One Stockham pass of radix R over buf (N points, sub-transform length Ns so far).
// SRC: 0 = LDS, 1 = gather from global int8/float samples.  DST: 0 = LDS, 1 = scatter to the symbol row.
// tc: the pass's twiddle table (unused when Ns = 1)
template <int R, typename LoadF, typename StoreF>
__device__ __forceinline__ void fft_pass(const float2 *__restrict__ tc, uint32_t N, uint32_t Ns, LoadF load, StoreF store)
{
    const uint32_t nb = N / R;
    for (uint32_t j = threadIdx.x; j < nb; j += blockDim.x) {
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = load(j, r, nb);
        const uint32_t k = j & (Ns - 1);
        if (Ns > 1) {
            float2 w[8];
            twiddles<R>(tc, k, w);
#pragma unroll
            for (int r = 1; r < R; r++) v[r] = cmul(v[r], w[r]);
        }
        dftR<R>(v);
        const uint32_t j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; r++) store(j0 + r * Ns, v[r]);
    }
}

template <typename T> struct SampleSrc;
template <> struct SampleSrc<int8_t> { // interleaved I,Q int8 (capture file format, LTE_fdd_dl_fs_samp_buf.cc:657-694)
    const int8_t *p;
    typedef uint32_t raw_t; // a fetched sample (I in byte 0, Q in byte 1) as it is held until its transform starts
    __device__ __forceinline__ raw_t raw(size_t n) const { return *reinterpret_cast<const uint16_t *>(p + 2 * n); }
    // sample f + o + c with f uniform over the workgroup, o the thread's own 32-bit offset and c a constant: a scalar base, one offset
    // register shared by all of a thread's loads and an immediate -- the 64-bit form costs two vector adds per load
    __device__ __forceinline__ raw_t raw_at(size_t f, uint32_t o, uint32_t c) const { return *reinterpret_cast<const uint16_t *>((p + 2 * (f + c)) + (size_t)(2u * o)); }
    static __device__ __forceinline__ float2 cvt(raw_t v)
    {
        float x, y; // one conversion per component, straight from its byte (the compiler's own choice shifts Q down first)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(x) : "v"(v));
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(y) : "v"(v));
        return make_float2(x, y);
    }
    __device__ __forceinline__ float2 at(size_t n) const
    {
        const char2 v = *reinterpret_cast<const char2 *>(p + 2 * n);
        return make_float2((float)v.x, (float)v.y);
    }
};
template <> struct SampleSrc<float> { // planar i_samps / q_samps as the reference API takes them
    const float *i, *q;
    typedef float2 raw_t;
    __device__ __forceinline__ raw_t raw(size_t n) const { return make_float2(i[n], q[n]); }
    __device__ __forceinline__ raw_t raw_at(size_t f, uint32_t o, uint32_t c) const
    {
        const char *bi = reinterpret_cast<const char *>(i + f + c), *bq = reinterpret_cast<const char *>(q + f + c);
        return make_float2(*reinterpret_cast<const float *>(bi + (size_t)(4u * o)), *reinterpret_cast<const float *>(bq + (size_t)(4u * o)));
    }
    static __device__ __forceinline__ float2 cvt(raw_t v) { return v; }
    __device__ __forceinline__ float2 at(size_t n) const { return make_float2(i[n], q[n]); }
};

// One workgroup per (symbol, unit).  (A persistent variant -- one workgroup walking all symbols of its unit with the next symbol's
// samples prefetched -- was measured slower on the MI355X, 9.5 vs 6.5 ms per 64k subframes: the compiler hoists the twiddles
// out of the symbol loop and the register count collapses the occupancy; it is not kept.)
// The transform length is a template parameter: with N fixed every LDS address of the passes is one per-thread base plus an
// immediate offset (pad() of a sum splits into constants), which takes a fifth of the kernel's vector instructions away.
template <typename T, bool RAW, uint32_t NT>
__global__ __launch_bounds__(256) void k_dl_fft(SampleSrc<T> src, const uint64_t *__restrict__ unit_start, DlGeom g,
                                                const float2 *__restrict__ tw, float *__restrict__ subframes)
{
    extern __shared__ __attribute__((aligned(16))) float2 buf[]; // pad(N) entries
    constexpr uint32_t N = NT, nb = N / 8;
    const uint32_t unit = blockIdx.y, j = threadIdx.x, half = g.half;
    const bool     act = j < nb; // one radix-8 butterfly per thread (N <= 2048)
    const size_t   ustart = unit_start[unit];
    // window start: slot start + symbol offset + CP - 1  (one sample early: liblte_phy.cc:8621)
    auto win = [&](uint32_t sym) -> size_t {
        const uint32_t so = sym % 7;
        return RAW ? ustart // (one symbol per unit, the window start given directly: sync.hip)
                   : ustart + (size_t)(sym / 7) * g.n_slot + (size_t)(N + g.cpe) * so + (so ? g.cp0 - g.cpe : 0) + (so == 0 ? g.cp0 : g.cpe) - 1;
    };
    typedef typename SampleSrc<T>::raw_t raw_t;
    auto fetch = [&](uint32_t sym, raw_t (&v)[8]) {
        const size_t f = win(sym);
        if (act) {
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = src.raw_at(f, j, r * nb);
        }
    };
    // element base + r * stride of the padded buffer: when the stride is a multiple of the padding period the pad of the sum splits,
    // so the R accesses of a butterfly are one address and R immediate offsets
    auto at = [&](uint32_t base, uint32_t r, uint32_t stride) -> float2 & {
        return (stride % 32 == 0) ? buf[pad(base) + r * (stride + stride / 32)] : buf[pad(base + r * stride)];
    };
    auto ld_s = [&](uint32_t base, uint32_t r, uint32_t stride) { return at(base, r, stride); };
    const uint32_t dc = g.ul ? 0u : 1u; // downlink skips the DC bin, the half-shifted uplink grid has none

    const float2 *__restrict__ twc = tw + 4096; // the per-pass tables
    const uint32_t sym = blockIdx.x;
    raw_t cur[8];
    fetch(sym, cur);
    {
        float *row_re = subframes + (size_t)unit * g.sf_stride + (size_t)sym * N_SC_MAX;
        float *row_im = row_re + (RAW ? N_SC_MAX : 16 * N_SC_MAX);
        auto st_g = [&](uint32_t o, float2 v) { // keep bins dc..half-1+dc and N-half..N-1 (liblte_phy.cc:8625-8634, :8685-8690)
            const bool     pos = o >= dc && o < half + dc, neg = o >= N - half;
            const uint32_t b   = 4u * (pos ? half + o - dc : o - (N - half)); // byte offset in the row: scalar base + 32-bit offset addressing
            if (pos || neg) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(row_re) + (size_t)b) = v.x;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(row_im) + (size_t)b) = v.y;
            }
        };
        // radix plan: 8,8,8,4 (2048) | 8,8,8,2 (1024) | 8,8,8 (512) | 8,8,4 (256) | 8,8,2 (128)
        // first radix-8 pass (sub-transform length 1: no twiddles) straight from the fetched samples.
        // uplink: the reference takes the ODD bins of a 2N-point FFT of the N samples zero-padded to 2N
        // (liblte_phy.cc:8676-8690), i.e. the N-point FFT of x[n]*exp(-i*pi*n/N) -- the rotation is applied here
        if (act) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float2 x = SampleSrc<T>::cvt(cur[r]);
                v[r] = g.ul ? cmul(x, tw[(j + r * nb) * (2048u / N)]) : x;
            }
            dft8(v);
#pragma unroll
            for (int r = 0; r < 8; r++) buf[pad(j * 8 + r)] = v[r];
        }
        uint32_t Ns = 8;
        __syncthreads();
        {   // second radix-8 pass: read all, then write (in place)
            float2 v[8];
            if (act) {
#pragma unroll
                for (int r = 0; r < 8; r++) v[r] = at(j, r, nb);
            }
            __syncthreads();
            if (act) {
                const uint32_t k = j & (Ns - 1);
                float2 w[8];
                twiddles<8>(twc + MI_FFT_TWC_P2, k, w);
#pragma unroll
                for (int r = 1; r < 8; r++) v[r] = cmul(v[r], w[r]);
                dft8(v);
                // j0 = 8 (j - k) + k with k < 8 and j - k a multiple of 8: (j0 + 8 r) >> 5 = (j - k) / 4 + (r >> 2)
                const uint32_t j0 = (j - k) * 8 + k, pj0 = j0 + ((j - k) >> 2);
#pragma unroll
                for (int r = 0; r < 8; r++) buf[pj0 + 8 * r + (r >> 2)] = v[r];
            }
            Ns *= 8;
        }
        __syncthreads();
        if (N == 128) fft_pass<2>(twc + MI_FFT_TWC_L128, N, Ns, ld_s, st_g);
        else if (N == 256) fft_pass<4>(twc + MI_FFT_TWC_L256, N, Ns, ld_s, st_g);
        else if (N == 512) fft_pass<8>(twc + MI_FFT_TWC_P3, N, Ns, ld_s, st_g);
        else {
            {   // third radix-8 pass in place
                float2 v[8];
                if (act) {
#pragma unroll
                    for (int r = 0; r < 8; r++) v[r] = at(j, r, nb);
                }
                __syncthreads();
                if (act) {
                    const uint32_t k = j & (Ns - 1);
                    float2 w[8];
                    twiddles<8>(twc + MI_FFT_TWC_P3, k, w);
#pragma unroll
                    for (int r = 1; r < 8; r++) v[r] = cmul(v[r], w[r]);
                    dft8(v);
                    const uint32_t j0 = (j - k) * 8 + k;
#pragma unroll
                    for (int r = 0; r < 8; r++) at(j0, r, 64) = v[r]; // Ns = 64 here
                }
                Ns *= 8;
            }
            __syncthreads();
            if (N == 1024) fft_pass<2>(twc + MI_FFT_TWC_L1024, N, Ns, ld_s, st_g);
            else           fft_pass<4>(twc + MI_FFT_TWC_L2048, N, Ns, ld_s, st_g);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The 2048-point transform as 8 x 16 x 16 (round 4): 128 threads per symbol, two LDS exchanges and three barriers instead of three and
// five (k_dl_fft above keeps every other length).  The 8 x 8 x 8 x 4 plan wrote 48 KB and read 48 KB of LDS per symbol and the LDS
// pipe was as busy as the vector ALU; this one moves 32 + 32 KB, multiplies by 15/16 + 15/16 of its points' twiddles
// instead of 7/8 + 7/8 + 3/4, and its last pass knows at compile time that outputs 5..10 of every radix-16 butterfly are guard band
// (bins 640..1407: LTE uses at most 600 sub-carriers per side), so six of its sixteen stores do not exist.
__device__ __forceinline__ void dft16(float2 *v)
{
    // 16 = 4 x 4: X[b + 4c] = sum_a W4^(ac) W16^(ab) sum_d W4^(bd) x[a + 4d]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    float2 t[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        t[a][0] = v[a]; t[a][1] = v[a + 4]; t[a][2] = v[a + 8]; t[a][3] = v[a + 12];
        dft4(t[a]);
    }
    t[1][1] = cmul_u(t[1][1], make_float2(c1, -s1));  // W16^1
    t[1][2] = cmul_u(t[1][2], make_float2(h, -h));    // W16^2
    t[1][3] = cmul_u(t[1][3], make_float2(s1, -c1));  // W16^3
    t[2][1] = cmul_u(t[2][1], make_float2(h, -h));    // W16^2
    t[2][2] = mul_mi(t[2][2]);                      // W16^4
    t[2][3] = cmul_u(t[2][3], make_float2(-h, -h));   // W16^6
    t[3][1] = cmul_u(t[3][1], make_float2(s1, -c1));  // W16^3
    t[3][2] = cmul_u(t[3][2], make_float2(-h, -h));   // W16^6
    t[3][3] = cmul_u(t[3][3], make_float2(-c1, s1));  // W16^9
#pragma unroll
    for (int b = 0; b < 4; b++) {
        float2 u[4] = {t[0][b], t[1][b], t[2][b], t[3][b]};
        dft4(u);
#pragma unroll
        for (int c = 0; c < 4; c++) v[b + 4 * c] = u[c];
    }
}

// w^1 .. w^15 of butterfly k: w, w^2, w^4, w^8 side by side in the pass's table (32 bytes per butterfly), the rest as products
__device__ __forceinline__ void twiddles16(const float2 *__restrict__ tc, uint32_t k, float2 (&w)[16])
{
    const char  *tb = reinterpret_cast<const char *>(tc);
    const float4 a = *reinterpret_cast<const float4 *>(tb + (size_t)(k * 32u)), b = *reinterpret_cast<const float4 *>(tb + 16 + (size_t)(k * 32u));
    w[1] = make_float2(a.x, a.y); w[2] = make_float2(a.z, a.w); w[4] = make_float2(b.x, b.y); w[8] = make_float2(b.z, b.w);
    w[3] = cmul(w[1], w[2]);
    w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
#pragma unroll
    for (int r = 1; r < 8; r++) w[8 + r] = cmul(w[8], w[r]);
}

// LDS padding of the 2048-point buffer: two float2 slots (16 bytes, so that 16-byte accesses stay aligned) after every 32
__device__ __forceinline__ uint32_t pad2(uint32_t i) { return i + ((i >> 5) << 1); }

template <typename T, bool RAW>
__global__ __launch_bounds__(128) void k_dl_fft2k(SampleSrc<T> src, const uint64_t *__restrict__ unit_start, DlGeom g,
                                                  const float2 *__restrict__ tw, float *__restrict__ subframes)
{
    __shared__ __attribute__((aligned(16))) float2 buf[2048 + 2 * 64];
    constexpr uint32_t N = 2048;
    const uint32_t unit = blockIdx.y, j = threadIdx.x, half = g.half, sym = blockIdx.x;
    const size_t   ustart = unit_start[unit];
    const uint32_t so = sym % 7;
    const size_t   f = RAW ? ustart : ustart + (size_t)(sym / 7) * g.n_slot + (size_t)(N + g.cpe) * so + (so ? g.cp0 - g.cpe : 0) + (so == 0 ? g.cp0 : g.cpe) - 1;
    typedef typename SampleSrc<T>::raw_t raw_t;
    raw_t cur[16];
#pragma unroll
    for (int m = 0; m < 16; m++) cur[m] = src.raw_at(f, j, m * 128);
    const float2 *__restrict__ twc = tw + 4096;
    // pass 1, radix 8, no twiddles: butterflies j and j + 128 (inputs x[b + 256 r]) straight from the fetched samples, outputs at 8 b + r
    auto pass1 = [&](auto ul) {
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float2 x = SampleSrc<T>::cvt(cur[hb + 2 * r]);
                v[r] = decltype(ul)::value ? cmul(x, tw[j + 128 * (hb + 2 * r)]) : x; // (uplink: the half-sub-carrier rotation, as in k_dl_fft)
            }
            dft8(v);
            const uint32_t b = j + 128 * hb; // pad2(8 b + r) = 8 b + r + 2 (b >> 2)
            float4 *dst = reinterpret_cast<float4 *>(&buf[8 * b + 2 * (b >> 2)]);
#pragma unroll
            for (int r = 0; r < 4; r++) dst[r] = make_float4(v[2 * r].x, v[2 * r].y, v[2 * r + 1].x, v[2 * r + 1].y);
        }
    };
    if (g.ul) pass1(std::true_type{});
    else      pass1(std::false_type{});
    __syncthreads();
    const uint32_t rd = j + 2 * (j >> 5); // pad2(j + 128 r) = rd + 136 r
    {   // pass 2, radix 16 over sub-transforms of length 8, in place
        float2 v[16], w[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = buf[rd + 136 * r];
        __syncthreads();
        const uint32_t k = j & 7;
        twiddles16(twc + MI_FFT_TWC_X2, k, w);
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = cmul(v[r], w[r]);
        dft16(v);
        const uint32_t wr = 136 * (j >> 3) + k; // output 128 (j >> 3) + k + 8 r: pad2 adds 2 (4 (j >> 3) + (r >> 2))
#pragma unroll
        for (int r = 0; r < 16; r++) buf[wr + 8 * r + 2 * (r >> 2)] = v[r];
    }
    __syncthreads();
    {   // pass 3, radix 16 over sub-transforms of length 128: butterfly j, bins j + 128 r
        float2 v[16], w[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = buf[rd + 136 * r];
        twiddles16(twc + MI_FFT_TWC_X3, j, w);
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = cmul(v[r], w[r]);
        dft16(v);
        float         *row_re = subframes + (size_t)unit * g.sf_stride + (size_t)sym * N_SC_MAX;
        float         *row_im = row_re + (RAW ? N_SC_MAX : 16 * N_SC_MAX);
        const uint32_t dc = g.ul ? 0u : 1u;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r >= 5 && r <= 10) continue; // bins 640 .. 1407: guard band whatever the bandwidth (half <= 600)
            const uint32_t o = j + 128 * r;
            const bool     pos = o >= dc && o < half + dc, neg = o >= N - half;
            const uint32_t b   = 4u * (pos ? half + o - dc : o - (N - half));
            if (pos || neg) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(row_re) + (size_t)b) = v[r].x;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(row_im) + (size_t)b) = v[r].y;
            }
        }
    }
}

// one instantiation per LTE transform length
#ifdef MI_FFT_OLD2K // (tools/ab: the 8 x 8 x 8 x 4 plan for the 2048-point transform, to time the two against each other)
#define MI_FFT2K_LAUNCH(name, T, RAW, grid, ...) MI_LAUNCH(ctx, name, (k_dl_fft<T, RAW, 2048>), grid, dim3(256), lds_fft, __VA_ARGS__)
#else
#define MI_FFT2K_LAUNCH(name, T, RAW, grid, ...) MI_LAUNCH(ctx, name, (k_dl_fft2k<T, RAW>), grid, dim3(128), 0, __VA_ARGS__)
#endif
#define FFT_LAUNCH(name, T, RAW, grid, ...)                                                                                        \
    switch (g.N) {                                                                                                                 \
    case 2048: MI_FFT2K_LAUNCH(name, T, RAW, grid, __VA_ARGS__); break;                                                            \
    case 1024: MI_LAUNCH(ctx, name, (k_dl_fft<T, RAW, 1024>), grid, dim3(256), lds_fft, __VA_ARGS__); break;                       \
    case 512:  MI_LAUNCH(ctx, name, (k_dl_fft<T, RAW, 512>), grid, dim3(256), lds_fft, __VA_ARGS__); break;                        \
    case 256:  MI_LAUNCH(ctx, name, (k_dl_fft<T, RAW, 256>), grid, dim3(256), lds_fft, __VA_ARGS__); break;                        \
    case 128:  MI_LAUNCH(ctx, name, (k_dl_fft<T, RAW, 128>), grid, dim3(256), lds_fft, __VA_ARGS__); break;                        \
    default: return MI_LTE_ERR_INVALID_ARG;                                                                                        \
    }

// ------------------------------------------------------------------------------------------------
// channel estimation

__global__ __launch_bounds__(256) void k_dl_ce(const uint32_t *__restrict__ subfr_num, const uint32_t *__restrict__ n_id_cell,
                                               DlGeom g, GoldTables gt, float *__restrict__ subframes)
{
    extern __shared__ __attribute__((aligned(16))) float sm[]; // mag[n_i][N_sc] ang[n_i][N_sc]
    __shared__ uint32_t crs_bits[5][14];
    const uint32_t unit = blockIdx.x, p = blockIdx.y, N_sc = 12 * g.N_rb_dl, n_pil = 2 * g.N_rb_dl;
    const uint32_t sf = subfr_num[unit], cell = n_id_cell[unit], v_shift = cell % 6;
    float *base   = subframes + (size_t)unit * g.sf_stride;
    const float *sym_re = base, *sym_im = base + 16 * N_SC_MAX;
    float *ce_re = base + 2 * 16 * N_SC_MAX + (size_t)p * 16 * N_SC_MAX;
    float *ce_im = ce_re + (size_t)g.N_ant * 16 * N_SC_MAX;

    // CRS symbols / frequency offsets of this port (liblte_phy.cc:5973-6014), as select chains so
    // that nothing is indexed dynamically out of registers
    const uint32_t N_sym = (p < 2) ? 5 : 3;
    // The CRS symbols are independent until the time interpolation.  The compact form stops before it, so there a workgroup is ONE
    // wavefront working on ONE CRS symbol (blockIdx.z): a fifth of the LDS (five times the resident workgroups), no wavefront waiting
    // at a barrier while another walks the unwrap, and the serial part of a subframe's estimate spread over five workgroups.
    const uint32_t i_lo = g.ce_compact ? blockIdx.z : 0u, i_hi = g.ce_compact ? i_lo + 1 : N_sym, n_i = i_hi - i_lo;
    float *mag = sm, *ang = sm + n_i * N_sc; // row r = CRS symbol i_lo + r
    auto sym_of = [p](uint32_t i) -> uint32_t {
        return (p < 2) ? (i == 0 ? 0u : i == 1 ? 4u : i == 2 ? 7u : i == 3 ? 11u : 14u) : (i == 0 ? 1u : i == 1 ? 8u : 15u);
    };
    auto voff_of = [p](uint32_t i) -> uint32_t {
        return (p < 2) ? ((((i & 1) ^ (p & 1)) != 0) ? 3u : 0u) : (p == 2 ? ((i & 1) ? 3u : 0u) : ((i & 1) ? 6u : 3u));
    };

    // CRS bits (generate_crs, liblte_phy.cc:8300-8333; slots per :5960-5967)
    if (threadIdx.x < n_i * 14) {
        const uint32_t ri = __umul24(threadIdx.x, 4682u) >> 16, i = i_lo + ri, w = threadIdx.x - 14 * ri; // / 14 and % 14 for < 5461
        const uint32_t sy = sym_of(i), slot = sy >= 14 ? 2u : sy >= 7 ? 1u : 0u, l = sy - 7 * slot; // (sy <= 15)
        const uint32_t ns2 = (sf * 2) % 20 + slot, ns = ns2 >= 20 ? ns2 - 20 : ns2;                   // (sf is uniform: its part is scalar arithmetic)
        const uint32_t c_init = (__umul24(7 * (ns + 1) + l + 1, 2 * cell + 1) << 10) + 2 * cell + 1;
        crs_bits[i - i_lo][w] = gold_word(gt, c_init, w);
    }
    __syncthreads();

    // least-squares estimate at every pilot (liblte_phy.cc:6023-6030)
    const float r2 = (float)(1.0 / sqrt(2.0));
    for (uint32_t t = threadIdx.x; t < n_i * n_pil; t += blockDim.x) {
        const uint32_t ri = quot_f(t, g.r_n_pil), i = i_lo + ri, j = t - __umul24(ri, n_pil);
        const uint32_t k = 6 * j + mod6_lt12(voff_of(i) + v_shift), mp = j + 110 - g.N_rb_dl;
        const uint32_t b0 = (crs_bits[ri][(2 * mp) >> 5] >> ((2 * mp) & 31)) & 1u, b1 = (crs_bits[ri][(2 * mp + 1) >> 5] >> ((2 * mp + 1) & 31)) & 1u;
        const float rs_re = r2 * (1 - 2 * (float)b0), rs_im = r2 * (1 - 2 * (float)b1);
        const float s_re = sym_re[__umul24(sym_of(i), (uint32_t)N_SC_MAX) + k], s_im = sym_im[__umul24(sym_of(i), (uint32_t)N_SC_MAX) + k];
        const float t_re = s_re * rs_re + s_im * rs_im, t_im = s_im * rs_re - s_re * rs_im;
        mag[__umul24(ri, N_sc) + k] = sqrtf(t_re * t_re + t_im * t_im);
        ang[__umul24(ri, N_sc) + k] = atan2f(t_im, t_re);
    }
    __syncthreads();
    // unwrap along frequency (liblte_phy.cc:6033-6035): u_0 = r_0, u_j = wrap_phase(r_j, u_{j-1}).  The chain is
    // serial in the reference; here one wave per CRS symbol guesses the wrap count c_j of every pilot from
    // the RAW neighbours (delta_j = steps wrap_phase(r_j, r_{j-1}) takes, c_j = prefix sum), applies the c_j
    // rounded +-2*pi steps to r_j, and then VERIFIES every link with the reference's own test
    // (wrap_phase(r_j, u_{j-1}) == u_j bit for bit).  If every link verifies, u is exactly the serial result
    // (induction over j); otherwise (a phase step within rounding of +-pi) lane 0 redoes the symbol serially.
    {
        const uint32_t wave = threadIdx.x >> 6, ln = threadIdx.x & 63, n_wave = blockDim.x >> 6;
        const uint32_t C = (n_pil + 63) / 64; // pilots per lane (<= 4 for 100 RB)
        for (uint32_t i = i_lo + wave; i < i_hi; i += n_wave) {
            const uint32_t off = mod6_lt12(voff_of(i) + v_shift);
            float *a = ang + __umul24(i - i_lo, N_sc) + off;
            float  r[4], uu[4];
            int    c[4], run = 0;
            const uint32_t j0 = __umul24(ln, C); // the lane's first pilot
            float  rprev = (ln > 0 && (j0 - 1) < n_pil) ? a[6 * (j0 - 1)] : 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t j = j0 + k;
                r[k] = (k < C && j < n_pil) ? a[6 * j] : 0.0f;
                int d = 0;
                if (k < C && j < n_pil && j > 0) {
                    float p1 = r[k];
                    while ((double)(p1 - rprev) >= M_PI) { p1 = (float)((double)p1 - 2 * M_PI); d--; }
                    while ((double)(p1 - rprev) <= -M_PI) { p1 = (float)((double)p1 + 2 * M_PI); d++; }
                }
                run += d;
                c[k]  = run; // inclusive within the lane
                rprev = r[k];
            }
            int incl = run; // wave-inclusive scan of the per-lane totals
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(incl, o);
                if ((int)ln >= o) incl += n;
            }
            const int base = incl - run;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                int   cc = c[k] + base;
                float p1 = r[k];
                for (; cc < 0; cc++) p1 = (float)((double)p1 - 2 * M_PI);
                for (; cc > 0; cc--) p1 = (float)((double)p1 + 2 * M_PI);
                uu[k] = p1;
            }
            // verify every link against the reference's own rule
            const uint32_t last_k = C - 1;
            float uprev = __shfl_up(last_k == 0 ? uu[0] : last_k == 1 ? uu[1] : last_k == 2 ? uu[2] : uu[3], 1);
            bool  bad = false;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t j = j0 + k;
                if (k < C && j < n_pil) {
                    if (j > 0) bad |= (__float_as_uint(wrap_phase(r[k], uprev)) != __float_as_uint(uu[k]));
                    uprev = uu[k];
                }
            }
            if (__any(bad)) {
                if (ln == 0) {
                    float prev = a[0];
                    for (uint32_t j = 1; j < n_pil; j++) {
                        prev     = wrap_phase(a[6 * j], prev);
                        a[6 * j] = prev;
                    }
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t j = j0 + k;
                    if (k < C && j < n_pil) a[6 * j] = uu[k];
                }
            }
        }
    }
    __syncthreads();
    // frequency interpolation between pilots, edges continue the first / last slope
    // (liblte_phy.cc:6037-6063; repeated subtraction kept to reproduce the rounding)
    for (uint32_t t = threadIdx.x; t < n_i * n_pil; t += blockDim.x) {
        const uint32_t ri = quot_f(t, g.r_n_pil), i = i_lo + ri, j = t - __umul24(ri, n_pil);
        if (j == 0) continue;
        const uint32_t off = mod6_lt12(voff_of(i) + v_shift), k = 6 * j + off;
        float *m = mag + __umul24(ri, N_sc), *a = ang + __umul24(ri, N_sc);
        const float fm = (m[k] - m[k - 6]) / 6, fa = (a[k] - a[k - 6]) / 6;
        float cm = m[k], ca = a[k];
        for (uint32_t z = 1; z < 6; z++) { cm -= fm; ca -= fa; m[k - z] = cm; a[k - z] = ca; }
        if (j == 1) {
            cm = m[k - 6]; ca = a[k - 6];
            for (uint32_t z = 1; z < off + 1; z++) { cm -= fm; ca -= fa; m[k - 6 - z] = cm; a[k - 6 - z] = ca; }
        }
        if (j == n_pil - 1) {
            cm = m[k]; ca = a[k];
            for (uint32_t z = 1; z < (5 - off) + 1; z++) { cm -= fm; ca -= fa; m[k + z] = cm; a[k + z] = ca; }
        }
    }
    __syncthreads();

    if (g.ce_compact) {
        // compact form for the PDSCH chain (MI_LTE_CE_COMPACT): the demodulator interpolates along time itself (ce_time_interp5, the same
        // code as below), so what leaves this kernel is the magnitude / phase rows at the CRS symbols -- rows 0..N_sym-1 of the port's
        // real-part plane hold mag, the same rows of its imaginary-part plane hold ang
        const uint32_t nq = N_sc >> 2; // N_sc is a multiple of 12
        for (uint32_t t = threadIdx.x; t < n_i * nq; t += blockDim.x) {
            const uint32_t ri = quot_f(t, g.r_nq), i = i_lo + ri, c = t - __umul24(ri, nq);
            reinterpret_cast<float4 *>(ce_re + __umul24(i, (uint32_t)N_SC_MAX))[c] = reinterpret_cast<const float4 *>(mag + __umul24(ri, N_sc))[c];
            reinterpret_cast<float4 *>(ce_im + __umul24(i, (uint32_t)N_SC_MAX))[c] = reinterpret_cast<const float4 *>(ang + __umul24(ri, N_sc))[c];
        }
        return;
    }
    // time interpolation per sub-carrier (liblte_phy.cc:6066-6193).  A call on a handful of units spreads this part -- 14 sin/cos pairs per
    // sub-carrier, two thirds of the kernel's time -- over gridDim.z workgroups, each of which has made the (cheap) frequency-direction part
    // for itself: 23 -> 11 us for the one subframe of a per-call caller at 20 MHz
    const uint32_t j_per = (N_sc + gridDim.z - 1) / gridDim.z, j_end = min(N_sc, (blockIdx.z + 1) * j_per);
    for (uint32_t j = blockIdx.z * j_per + threadIdx.x; j < j_end; j += blockDim.x) {
        float M[5], A[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { M[i] = mag[i * N_sc + j]; A[i] = ang[i * N_sc + j]; }
#define EMIT(z, m, a) do { float sn_, cs_; ce_sincos((a), sn_, cs_); ce_re[(z) * N_SC_MAX + j] = (m) * cs_; ce_im[(z) * N_SC_MAX + j] = (m) * sn_; } while (0)
        if (N_sym == 3) {
            float fm, fa, cm, ca;
#define SLOPE(hi, lo, dv) do { fm = (M[hi] - M[lo]) / (dv); A[hi] = wrap_phase(A[hi], A[lo]); fa = A[hi] - A[lo]; \
                               fa = wrap_phase(fa, 0.0f); fa /= (dv); } while (0)
            EMIT(1, M[0], A[0]);
            EMIT(8, M[1], A[1]);
            SLOPE(1, 0, 7);
            cm = M[1]; ca = A[1];
            for (int z = 7; z > 1; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
            cm = M[0] - fm; ca = A[0] - fa; // symbol 0 extrapolated with the 1->8 slope (:6093-6098)
            EMIT(0, cm, ca);
            SLOPE(2, 1, 7);
            cm = M[2] - fm; ca = A[2] - fa;
            for (int z = 13; z > 8; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
#undef SLOPE
        } else {
            float m[14], a[14];
            ce_time_interp5(M, A, m, a);
#pragma unroll
            for (int z = 0; z < 14; z++) EMIT(z, m[z], a[z]);
        }
#undef EMIT
    }
}

int make_geom(const mi_lte_dl_cfg *cfg, DlGeom *g)
{
    const uint32_t N = cfg->fft_size;
    if (!(N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048)) return MI_LTE_ERR_INVALID_ARG;
    if (!(cfg->N_ant == 1 || cfg->N_ant == 2 || cfg->N_ant == 4)) return MI_LTE_ERR_INVALID_ARG;
    if (cfg->N_rb_dl < 6 || cfg->N_rb_dl > 100 || cfg->N_rb_dl * 12 >= N) return MI_LTE_ERR_INVALID_ARG;
    const uint32_t sc = 2048 / N;
    g->N = N; g->cp0 = 160 / sc; g->cpe = 144 / sc; g->n_slot = 15360 / sc;
    g->half = 6 * cfg->N_rb_dl; g->N_rb_dl = cfg->N_rb_dl; g->N_ant = cfg->N_ant;
    g->sf_stride = (uint32_t)mi_lte_subframe_floats(cfg->N_ant);
    g->ul = 0;
    g->ce_compact = 0;
    g->r_n_pil = recip_up(2 * cfg->N_rb_dl);
    g->r_nq    = recip_up(3 * cfg->N_rb_dl);
    return MI_LTE_OK;
}

} // namespace

extern "C" size_t mi_lte_subframe_floats(uint32_t N_ant) { return (size_t)(2 + 2 * N_ant) * 16 * N_SC_MAX; }

extern "C" int mi_lte_dl_frontend_batch(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a,
                                        const void *d_samples_b, const uint64_t *d_unit_start,
                                        const uint32_t *d_subfr_num, const uint32_t *d_n_id_cell, uint32_t n_units,
                                        float *d_subframes)
{
    if (!ctx || !cfg || !d_samples_a || !d_unit_start || !d_subfr_num || !d_n_id_cell || !d_subframes || n_units == 0)
        return MI_LTE_ERR_INVALID_ARG;
    DlGeom g;
    int    rc = make_geom(cfg, &g);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = mi_ctx_gold_tables(ctx);
    if (rc != MI_LTE_OK) return rc;
    rc = mi_ctx_fft_twiddles(ctx);
    if (rc != MI_LTE_OK) return rc;
    const size_t lds_fft = sizeof(float2) * (g.N + g.N / 32 + 1);
    // 14 symbols + the look-ahead symbols of the next subframe that the CRS interpolation reads: symbol 14 (ports 0 and 1) and, only
    // with four ports, symbol 15 (ports 2 and 3 carry their CRS in the second symbol of a slot)
    const uint32_t fmt = cfg->sample_format & ~(uint32_t)(MI_LTE_IQ_ALL_ROWS | MI_LTE_CE_COMPACT);
    if (cfg->sample_format & MI_LTE_CE_COMPACT) {
        if (cfg->N_ant != 1) { ctx->err = "MI_LTE_CE_COMPACT: single-port cells only"; return MI_LTE_ERR_UNSUPPORTED; }
        g.ce_compact = 1;
    }
    const uint32_t n_sym = (cfg->N_ant > 2 || (cfg->sample_format & MI_LTE_IQ_ALL_ROWS)) ? 16 : 15;
    if (fmt == MI_LTE_IQ_I8) {
        SampleSrc<int8_t> s{(const int8_t *)d_samples_a};
        FFT_LAUNCH("k_dl_fft", int8_t, false, dim3(n_sym, n_units), s, d_unit_start, g, ctx->d_fft_tw, d_subframes);
    } else if (fmt == MI_LTE_IQ_F32_PLANAR) {
        if (!d_samples_b) return MI_LTE_ERR_INVALID_ARG;
        SampleSrc<float> s{(const float *)d_samples_a, (const float *)d_samples_b};
        FFT_LAUNCH("k_dl_fft", float, false, dim3(n_sym, n_units), s, d_unit_start, g, ctx->d_fft_tw, d_subframes);
    } else
        return MI_LTE_ERR_INVALID_ARG;
    GoldTables gt{ctx->d_gold_x1, ctx->d_gold_x2b, ctx->gold_words};
    if (g.ce_compact) { // one wavefront per (unit, CRS symbol); single-port cells only, so five symbols
        const size_t lds_ce = sizeof(float) * 2 * 12 * g.N_rb_dl;
        MI_LAUNCH(ctx, "k_dl_ce", k_dl_ce, dim3(n_units, g.N_ant, 5), dim3(64), lds_ce, d_subfr_num, d_n_id_cell, g, gt, d_subframes);
    } else {
        const size_t lds_ce = sizeof(float) * 10 * 12 * g.N_rb_dl;
        const uint32_t n_split = n_units * g.N_ant >= 256 ? 1u : std::max(1u, 12 * g.N_rb_dl / 240); // a batch fills the device by itself
        MI_LAUNCH(ctx, "k_dl_ce", k_dl_ce, dim3(n_units, g.N_ant, n_split), dim3(256), lds_ce, d_subfr_num, d_n_id_cell, g, gt, d_subframes);
    }
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_dl_fft:1,k_dl_ce:1";
    return MI_LTE_OK;
}

// n_rows OFDM symbols at arbitrary window starts (samples_to_symbols_dl with scale 0, liblte_phy.cc:8592-8634; the caller adds
// CP - 1): d_rows[r] = 1200 real parts then 1200 imaginary parts.  Used by the synchronisation searches (sync.hip).
int mi_fft_rows(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b, const uint64_t *d_win_start,
                uint32_t n_rows, float *d_rows)
{
    mi_lte_dl_cfg c1 = *cfg;
    c1.N_ant         = 1;
    DlGeom g;
    int    rc = make_geom(&c1, &g);
    if (rc != MI_LTE_OK) return rc;
    g.sf_stride = 2 * N_SC_MAX; // one row of real parts, one of imaginary parts per window
    rc = mi_ctx_fft_twiddles(ctx);
    if (rc != MI_LTE_OK) return rc;
    const size_t lds_fft = sizeof(float2) * (g.N + g.N / 32 + 1);
    if (cfg->sample_format == MI_LTE_IQ_I8) {
        SampleSrc<int8_t> s{(const int8_t *)d_samples_a};
        FFT_LAUNCH("k_sync_fft", int8_t, true, dim3(1, n_rows), s, d_win_start, g, ctx->d_fft_tw, d_rows);
    } else {
        if (!d_samples_b) return MI_LTE_ERR_INVALID_ARG;
        SampleSrc<float> s{(const float *)d_samples_a, (const float *)d_samples_b};
        FFT_LAUNCH("k_sync_fft", float, true, dim3(1, n_rows), s, d_win_start, g, ctx->d_fft_tw, d_rows);
    }
    MI_HIP_CHECK(ctx, hipGetLastError());
    return MI_LTE_OK;
}

// ------------------------------------------------------------------------------------------------
// uplink front end: 14 SC-FDMA symbols per unit, no channel estimation here (the PUSCH estimate is per
// allocation, see uplink.hip)
extern "C" size_t mi_lte_ul_subframe_floats(void) { return (size_t)2 * 16 * N_SC_MAX; }

extern "C" int mi_lte_ul_frontend_batch(mi_lte_ctx *ctx, const mi_lte_dl_cfg *cfg, const void *d_samples_a, const void *d_samples_b,
                                        const uint64_t *d_unit_start, uint32_t n_units, float *d_subframes)
{
    if (!ctx || !cfg || !d_samples_a || !d_unit_start || !d_subframes || n_units == 0) return MI_LTE_ERR_INVALID_ARG;
    mi_lte_dl_cfg c1 = *cfg;
    c1.N_ant         = 1;
    DlGeom g;
    int    rc = make_geom(&c1, &g);
    if (rc != MI_LTE_OK) return rc;
    g.ul        = 1;
    g.sf_stride = (uint32_t)mi_lte_ul_subframe_floats();
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = mi_ctx_fft_twiddles(ctx);
    if (rc != MI_LTE_OK) return rc;
    const size_t lds_fft = sizeof(float2) * (g.N + g.N / 32 + 1);
    if (cfg->sample_format == MI_LTE_IQ_I8) {
        SampleSrc<int8_t> s{(const int8_t *)d_samples_a};
        FFT_LAUNCH("k_ul_fft", int8_t, false, dim3(14, n_units), s, d_unit_start, g, ctx->d_fft_tw, d_subframes);
    } else if (cfg->sample_format == MI_LTE_IQ_F32_PLANAR) {
        if (!d_samples_b) return MI_LTE_ERR_INVALID_ARG;
        SampleSrc<float> s{(const float *)d_samples_a, (const float *)d_samples_b};
        FFT_LAUNCH("k_ul_fft", float, false, dim3(14, n_units), s, d_unit_start, g, ctx->d_fft_tw, d_subframes);
    } else
        return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_ul_fft:1";
    return MI_LTE_OK;
}
