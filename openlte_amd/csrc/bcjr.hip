// Max-log-MAP ("BCJR") turbo decoding on gfx950 -- the decoder BASELINE.json's north star sketches, offered as
// MI_LTE_TURBO_BCJR next to the reference-faithful REF mode.  The reference has no such decoder, so
// this mode is specified by a plain-C fixed-point model in the test tree (lo_turbo_decode_bcjr): every operation
// in the same order, and the kernels below must match it bit for bit.
//
// Mapping: lock-step tiles like the REF decoder, TWO code blocks per lane: lane l of a wavefront walks the trellises of
// code block l of tile 2p and of tile 2p+1 in the two halves of its registers (v_pk_add_i16 / v_pk_max_i16; the model's number
// ranges keep every intermediate inside int16, so nothing saturates).  The 8 alpha / beta metrics of a trellis live in 8 VGPRs,
// max* is a plain packed max, nothing crosses lanes.
//
// One launch per half-iteration (k_bcjr_half): a wavefront takes a segment of its tile pair and works through it in blocks of 32
// steps -- forward (alpha, normalised and kept every 8 steps), then backward over the same 32 steps (alpha re-run inside each 8-step
// window from its checkpoint, LLR, extrinsic, beta), beta starting from what the following block reached in the previous iteration
// ("next iteration initialisation", part of the specification).  Nothing per trellis step is stored: per step and code block the
// kernel reads 3 bytes (systematic, parity, a-priori) and writes 1 (extrinsic), plus 2 bytes of boundary state -- against 18 bytes
// for the forward / backward / permute kernels this replaces.
//
// The exchange between the two constituent decoders needs no kernel of its own: extrinsics are stored one ROW per trellis step
// (128 bytes: the 64 lanes of both tiles of a pair), so "A[t] = E[pi[t]]" is a row index taken from a table that is the same for
// every lane -- a scalar load and a fully coalesced 128-byte access per step.  Row Kp of every pair is zero: holes of the
// de-interleaver and the padding past K point there.
//
// Trellis (36.212 5.1.3.2.1, feedback 1+D^2+D^3, parity 1+D+D^3; state = 4 r1 + 2 r2 + r3): the predecessors of
// state n are 2(n&3) and 2(n&3)+1 with complementary (u,z) labels -- see the table next to lo_turbo_decode_bcjr.  Branch
// metric g(u,z) = [u==0](Ls+La) + [z==0]Lp, positive LLR = bit 0.
#include "ctx.hpp"

namespace {

constexpr int BCJR_NEG = -6000, BCJR_X_MAX = 340, BCJR_LE_MAX = 254;

__host__ __device__ inline uint32_t kpad64(uint32_t K) { return (K + 63u) & ~63u; }
__device__ __forceinline__ int sb(uint32_t w, int k) { return (int)__builtin_amdgcn_sbfe(w, 8 * k, 8); }

__device__ __forceinline__ size_t g8(uint32_t tile, uint32_t Kp, uint32_t lane, uint32_t gran) { return (size_t)tile * Kp * 64 + (size_t)gran * 1024 + lane * 16; }   // byte offset, int8 granule arrays
// XCD-aware block -> code block mapping of the per-code-block kernels (see turbo.hip)
__host__ __device__ inline uint32_t xcd_chunk(uint32_t n_cb) { return ((((n_cb + 63u) >> 6) + 7u) >> 3) << 6; }
__device__ __forceinline__ uint32_t xcd_cb(uint32_t b, uint32_t n_cb) { return (b & 7u) * xcd_chunk(n_cb) + (b >> 3); }

__host__ __device__ inline uint32_t bcjr_n_seg(uint32_t K)
{
    const uint32_t nblk = (K + 63) / 64;
    for (uint32_t n = 8; n > 1; n >>= 1)
        if (nblk % n == 0 && (nblk / n) * 64 >= 512) return n;
    return 1;
}

// pairs of int16 in one register: low half = the code block of tile 2p, high half = tile 2p+1
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s      as_v2s(uint32_t w) { return __builtin_bit_cast(v2s, w); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s      vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2s      vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s      splat(int v) { return (v2s)((short)v); }
// byte R of the two words, sign-extended, as the pair (lo: w0, hi: w1)
template <int R> __device__ __forceinline__ v2s byte_pair(uint32_t w0, uint32_t w1)
{
    return as_v2s(__builtin_amdgcn_perm(w1, w0, (uint32_t)(4 + R) << 24 | 0x0C0000u | (uint32_t)R << 8 | 0x0Cu)) >> 8;
}
__device__ __forceinline__ v2s byte_pair_rt(uint32_t w0, uint32_t w1, int r) // r = 0..3, resolved at compile time after unrolling
{
    return r == 0 ? byte_pair<0>(w0, w1) : r == 1 ? byte_pair<1>(w0, w1) : r == 2 ? byte_pair<2>(w0, w1) : byte_pair<3>(w0, w1);
}
// Global accesses at "wavefront-uniform base + the lane's 32-bit offset": the base is pinned to scalar registers (an empty asm the
// optimiser cannot look through -- left alone it re-associates "(base + row) + lane" into "(base + lane) + row" and carries the first sum
// as a 64-bit vector pair, one v_lshl_add_u64 per access) and the access goes through an explicit global-address-space pointer (the asm
// hides where the pointer came from; without the qualifier the access would become a flat one).  The offset's zero-extension has to be
// formed in the basic block of the access -- hoisted out of a loop it reaches instruction selection as a 64-bit pair and the scalar-base
// form is not recognised -- hence local_off() once per loop body.
__device__ __forceinline__ uint32_t local_off(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const char gconst_char_t;
typedef __attribute__((address_space(1))) char       gchar_t;
template <typename T> __device__ __forceinline__ T ld_sbase(const void *ubase, uint32_t voff)
{
    gconst_char_t *b = reinterpret_cast<gconst_char_t *>(reinterpret_cast<uintptr_t>(ubase));
    asm volatile("" : "+s"(b));
    return *reinterpret_cast<__attribute__((address_space(1))) const T *>(b + voff);
}
template <typename T> __device__ __forceinline__ void st_sbase(void *ubase, uint32_t voff, T v)
{
    gchar_t *b = reinterpret_cast<gchar_t *>(reinterpret_cast<uintptr_t>(ubase));
    asm volatile("" : "+s"(b));
    *reinterpret_cast<__attribute__((address_space(1))) T *>(b + voff) = v;
}
__device__ __forceinline__ uint32_t word_of(const uint4 &q, int k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; }

__device__ __forceinline__ void norm8(v2s (&v)[8]) // subtract the maximum, floor at BCJR_NEG (per half)
{
    const v2s m = vmax(vmax(vmax(v[0], v[1]), vmax(v[2], v[3])), vmax(vmax(v[4], v[5]), vmax(v[6], v[7])));
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = vmax(v[s] - m, splat(BCJR_NEG));
}
__device__ __forceinline__ void alpha_step(const v2s (&a)[8], v2s g00, v2s g01, v2s g10, v2s (&o)[8])
{
    o[0] = vmax(a[0] + g00, a[1]);       o[4] = vmax(a[0], a[1] + g00);
    o[1] = vmax(a[2] + g10, a[3] + g01); o[5] = vmax(a[2] + g01, a[3] + g10);
    o[2] = vmax(a[4] + g01, a[5] + g10); o[6] = vmax(a[4] + g10, a[5] + g01);
    o[3] = vmax(a[6], a[7] + g00);       o[7] = vmax(a[6] + g00, a[7]);
}
__device__ __forceinline__ void beta_step(v2s (&b)[8], v2s g00, v2s g01, v2s g10)
{
    v2s o[8];
    o[0] = vmax(b[0] + g00, b[4]);       o[1] = vmax(b[0], b[4] + g00);
    o[2] = vmax(b[1] + g10, b[5] + g01); o[3] = vmax(b[1] + g01, b[5] + g10);
    o[4] = vmax(b[2] + g01, b[6] + g10); o[5] = vmax(b[2] + g10, b[6] + g01);
    o[6] = vmax(b[3], b[7] + g00);       o[7] = vmax(b[3] + g00, b[7]);
#pragma unroll
    for (int s = 0; s < 8; s++) b[s] = o[s];
}

// ---- layouts
//   S1 P1 S2 P2   int8, per tile, "16-byte granules": granule g (16 steps) of lane l at g*1024 + l*16 -- a wave-wide access is one KiB
//   E1 E2 HD      int8, per tile PAIR, one 128-byte row per step: byte (t, lane, h) at (pair*(Kp+1) + t)*128 + lane*2 + h, h = tile & 1;
//                 E1 = extrinsic halves of decoder 1 in natural order, E2 = those of decoder 2 in ITS (interleaved) order, HD = hard
//                 decisions of the last half-iteration in decoder 2's order; row Kp is zero
//   tail          per block: t1s[3] t1p[3] t2s[3] t2p[3] pad[4]
//   boundaries    alpha: [seg][pair][lane][8 x v2s]; beta: [blk32][pair][lane][8 x v2s]; two buffers each (read: previous iteration)
struct BcjrBufs {
    int8_t *S1, *P1, *S2, *P2;
    int8_t *tail;
};
__device__ __forceinline__ size_t ex_row(uint32_t pair, uint32_t Kp, uint32_t t) { return ((size_t)pair * (Kp + 1) + t) * 128; }

// ------------------------------------------------------------------------------------------------
// prep: split the interleaved d[i*3+x] input into tile granules, interleave the systematic stream
__global__ __launch_bounds__(384) void k_bcjr_prep(const int8_t *__restrict__ soft, uint32_t K, uint32_t n_cb,
                                                   const uint16_t *__restrict__ pi, BcjrBufs B)
{
    extern __shared__ __attribute__((aligned(16))) int8_t sm[]; // S1[Kp]
    const uint32_t cb = xcd_cb(blockIdx.x, n_cb), tile = cb >> 6, lane = cb & 63, Kp = kpad64(K), n_units = Kp >> 4, u = threadIdx.x;
    if (cb >= n_cb) return;
    const int8_t *d = soft + (size_t)cb * 3 * (K + 4);
    const size_t  o8 = g8(tile, Kp, lane, u);
    const int     nv = (u < n_units) ? min(16, max(0, (int)K - 16 * (int)u)) : -1;
    uint32_t s1[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0};
    if (nv > 0) {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(d + (size_t)u * 48); // 48 (24) bytes, 4-byte aligned
#pragma unroll
        for (int w = 0; w < 12; w++) {
            const uint32_t x = (w < 6 || nv > 8) ? g[w] : 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = 4 * w + k, i = e / 3, st = e - 3 * i; // element e = 3*i + stream
                const int v = max(sb(x, k), -127);                  // clip to +-127
                const uint32_t byte = ((uint32_t)v & 0xFFu) << (8 * (i & 3));
                if (st == 0) s1[i >> 2] |= byte; else if (st == 1) p1[i >> 2] |= byte; else p2[i >> 2] |= byte;
            }
        }
    }
    if (nv >= 0) {
        *reinterpret_cast<uint4 *>(B.S1 + o8) = make_uint4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<uint4 *>(B.P1 + o8) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4 *>(B.P2 + o8) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        *reinterpret_cast<uint4 *>(sm + 16 * u) = make_uint4(s1[0], s1[1], s1[2], s1[3]);
    }
    if (u == 0) { // termination bits: x[3r + stream] = d_stream[K + r]  (36.212 5.1.3.2.2)
        const int8_t *x = d + 3 * (size_t)K;
        int8_t       *t = B.tail + ((size_t)cb << 4);
        auto c = [&](int i) { return (int8_t)max((int)x[i], -127); };
        t[0] = c(0); t[1] = c(2); t[2] = c(4);   t[3] = c(1); t[4] = c(3);  t[5] = c(5);  // decoder 1: x_K x_K+1 x_K+2 | z_K z_K+1 z_K+2
        t[6] = c(6); t[7] = c(8); t[8] = c(10);  t[9] = c(7); t[10] = c(9); t[11] = c(11); // decoder 2
    }
    __syncthreads();
    if (nv >= 0) { // S2[i] = S1[pi[i]]
        uint32_t s2[4] = {0, 0, 0, 0};
        if (nv > 0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(pi + 16 * (size_t)u);
            const uint4  lo = p[0], hi = (nv > 8) ? p[1] : make_uint4(0, 0, 0, 0);
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t idx = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t v   = (uint8_t)sm[idx];
                s2[k >> 2] |= ((k < nv) ? v : 0u) << (8 * (k & 3));
            }
        }
        *reinterpret_cast<uint4 *>(B.S2 + o8) = make_uint4(s2[0], s2[1], s2[2], s2[3]);
    }
}

// The same for EIGHT neighbouring code blocks per workgroup (lanes 8 m .. 8 m + 7 of one tile): a granule is 16 bytes per lane, so a
// workgroup that owns one code block writes 16 bytes into each 128-byte line of its four arrays and seven other workgroups the rest --
// the PMC pass of round 4 counted 3.3 GB written for 1.6 GB of granules (profiles/r04_turbo_bcjr_summary.md).  Here a wavefront's
// store covers 8 units x 8 blocks = eight whole lines.  Item = (unit u, block b), b fastest.
__global__ __launch_bounds__(512) void k_bcjr_prep8(const int8_t *__restrict__ soft, uint32_t K, uint32_t n_cb,
                                                    const uint16_t *__restrict__ pi, BcjrBufs B)
{
    extern __shared__ __attribute__((aligned(16))) int8_t sm[]; // S1[8][Kp]
    const uint32_t grp = (blockIdx.x & 7u) * (xcd_chunk(n_cb) >> 3) + (blockIdx.x >> 3), cb0 = grp * 8, tile = cb0 >> 6, lane0 = cb0 & 63;
    const uint32_t Kp = kpad64(K), n_units = Kp >> 4;
    if (cb0 >= n_cb) return;
    for (uint32_t item = threadIdx.x; item < 8 * n_units; item += blockDim.x) {
        const uint32_t b = item & 7u, u = item >> 3, cb = cb0 + b;
        if (cb >= n_cb) continue;
        const int8_t *d = soft + (size_t)cb * 3 * (K + 4);
        const size_t  o8 = g8(tile, Kp, lane0 + b, u);
        const int     nv = min(16, max(0, (int)K - 16 * (int)u));
        uint32_t s1[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0};
        if (nv > 0) {
            const uint32_t *g = reinterpret_cast<const uint32_t *>(d + (size_t)u * 48); // 48 (24) bytes, 4-byte aligned
#pragma unroll
            for (int w = 0; w < 12; w++) {
                const uint32_t x = (w < 6 || nv > 8) ? g[w] : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = 4 * w + k, i = e / 3, st = e - 3 * i; // element e = 3*i + stream
                    const int v = max(sb(x, k), -127);                  // clip to +-127
                    const uint32_t byte = ((uint32_t)v & 0xFFu) << (8 * (i & 3));
                    if (st == 0) s1[i >> 2] |= byte; else if (st == 1) p1[i >> 2] |= byte; else p2[i >> 2] |= byte;
                }
            }
        }
        *reinterpret_cast<uint4 *>(B.S1 + o8) = make_uint4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<uint4 *>(B.P1 + o8) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4 *>(B.P2 + o8) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        *reinterpret_cast<uint4 *>(sm + b * Kp + 16 * u) = make_uint4(s1[0], s1[1], s1[2], s1[3]);
    }
    if (threadIdx.x < 8 && cb0 + threadIdx.x < n_cb) { // termination bits: x[3r + stream] = d_stream[K + r]  (36.212 5.1.3.2.2)
        const uint32_t cb = cb0 + threadIdx.x;
        const int8_t  *x = soft + (size_t)cb * 3 * (K + 4) + 3 * (size_t)K;
        int8_t        *t = B.tail + ((size_t)cb << 4);
        auto c = [&](int i) { return (int8_t)max((int)x[i], -127); };
        t[0] = c(0); t[1] = c(2); t[2] = c(4);   t[3] = c(1); t[4] = c(3);  t[5] = c(5);  // decoder 1: x_K x_K+1 x_K+2 | z_K z_K+1 z_K+2
        t[6] = c(6); t[7] = c(8); t[8] = c(10);  t[9] = c(7); t[10] = c(9); t[11] = c(11); // decoder 2
    }
    __syncthreads();
    for (uint32_t item = threadIdx.x; item < 8 * n_units; item += blockDim.x) { // S2[i] = S1[pi[i]]
        const uint32_t b = item & 7u, u = item >> 3, cb = cb0 + b;
        if (cb >= n_cb) continue;
        const int nv = min(16, max(0, (int)K - 16 * (int)u));
        uint32_t  s2[4] = {0, 0, 0, 0};
        if (nv > 0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(pi + 16 * (size_t)u);
            const uint4  lo = p[0], hi = (nv > 8) ? p[1] : make_uint4(0, 0, 0, 0);
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int8_t  *s1b = sm + b * Kp;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t idx = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t v   = (uint8_t)s1b[idx];
                s2[k >> 2] |= ((k < nv) ? v : 0u) << (8 * (k & 3));
            }
        }
        *reinterpret_cast<uint4 *>(B.S2 + g8(tile, Kp, lane0 + b, u)) = make_uint4(s2[0], s2[1], s2[2], s2[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// one half-iteration of one constituent decoder over a segment of a tile pair
struct HalfArgs {
    const int8_t   *S, *P;      // this decoder's systematic / parity granules
    const int8_t   *A;          // a-priori halves: the other decoder's extrinsic rows ...
    const uint32_t *row;        // ... through this table: A[t] = row A_rows[row[t]] (pi towards decoder 2, the inverse map back; Kp = the zero row)
    int8_t         *E;          // extrinsic halves out, row t
    uint8_t        *HD;         // LAST: hard decisions out, row t
    const int8_t   *tail;
    uint32_t        tail_off;   // 0 / 6: the decoder's three termination pairs inside a block's tail record
    const uint4    *a_rd, *b_rd;
    uint4          *a_wr, *b_wr;
    uint32_t        K, n_cb, n_tiles, n_pairs;
    // early termination (MI_LTE_TURBO_BCJR_EARLY): chg[it * n_pairs + pair] != 0 <=> some hard decision of the pair's 128 code blocks changed
    // in iteration `it` against iteration it - 1 (always set for it = 0); a pair whose iteration it - 1 changed nothing has stopped
    uint32_t       *chg;
    uint32_t        it;
    uint32_t        noap_rt;    // EARLY instantiations: 1 = this launch is the decode's first half-iteration (see NOAP)
};

#ifndef BCJR_WPE
#define BCJR_WPE 4
#endif
// LAST: the decoder-2 half of the final iteration (writes the hard decisions).  EARLY: hard-decision-aided stopping per tile pair -- every
// decoder-2 half writes the decisions and notes whether any differs from the previous iteration's; both halves of a later iteration
// return at once for a pair that has stopped (its extrinsics, boundary states and decisions stay what its last iteration left).
// NOAP: the first half-iteration of a decode -- the a-priori values are zero and are not read (nor is their array filled any more: 400 MB
// per 65 536-block decode).  A template parameter for the fixed-iteration kernels: as a run-time test the uniform branch around the eight
// row reads of a window cost their other fifteen half-iterations 3 %.  The EARLY instantiations take the run-time flag (HalfArgs::noap_rt)
// instead -- with them the same branch came out 5 % FASTER per half-iteration than the branch-free code (W3 early termination, same box:
// 6.53 ms with the template form, 6.19-6.32 with the flag; profiles/r04_variants_bcjr_first_half.txt), so each mode keeps what measured best.
template <bool LAST, bool EARLY = false, bool NOAP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BCJR_WPE, 8))) void k_bcjr_half(HalfArgs g)
{
    if (EARLY && g.it >= 2 && g.chg[(size_t)(g.it - 1) * g.n_pairs + blockIdx.x] == 0) return; // the pair stopped (uniform)
    uint32_t hd_diff = 0;
    // alpha checkpoints of the current 32-step block: [window][first | second four states][lane].  The windows are walked by real
    // loops (a fully unrolled block is 67 KB of code, more than the instruction cache holds), so what is indexed by the window number
    // lives here rather than in registers
    __shared__ uint4 chk[4][2][64];
    const uint32_t pair = blockIdx.x, seg = blockIdx.y, n_seg = gridDim.y, lane = threadIdx.x, K = g.K, Kp = kpad64(K);
    const uint32_t seg_len = ((Kp >> 6) / n_seg) * 64, t_lo = seg * seg_len, t_hi = min(t_lo + seg_len, K), n_blk = (K + 31) >> 5;
    const uint32_t tile0 = 2 * pair, tile1 = min(2 * pair + 1, g.n_tiles - 1); // an odd tile count: the last tile twice (its results land in the spare half)
    // Every global address of the loops below is a wavefront-uniform base (scalar registers, scalar arithmetic) plus the lane's own 32-bit
    // offset: written as per-lane 64-bit pointers they cost a v_lshl_add_u64 (and often a 64-bit shift) per access -- a tenth of the
    // loops' vector instructions in a kernel that is bound by exactly those
    const char *s0 = reinterpret_cast<const char *>(g.S) + g8(tile0, Kp, 0, 0), *s1 = reinterpret_cast<const char *>(g.S) + g8(tile1, Kp, 0, 0);
    const char *p0 = reinterpret_cast<const char *>(g.P) + g8(tile0, Kp, 0, 0), *p1 = reinterpret_cast<const char *>(g.P) + g8(tile1, Kp, 0, 0);
    const char *arow = reinterpret_cast<const char *>(g.A) + ex_row(pair, Kp, 0);
    char       *erow = reinterpret_cast<char *>(g.E) + ex_row(pair, Kp, 0);
    char       *hrow = LAST ? reinterpret_cast<char *>(g.HD) + ex_row(pair, Kp, 0) : nullptr;
    const size_t bnd_lane = ((size_t)pair * 64 + lane) * 2, bnd_stride = (size_t)g.n_pairs * 64 * 2; // in uint4 units: 8 x v2s = two uint4

    // the eight steps of the window that starts at t0 (a multiple of 8): Ls + La and Lp of both trellises
    auto load_window = [&](uint32_t t0, v2s (&lsa)[8], v2s (&lp)[8]) {
        const size_t   go = (size_t)(t0 >> 4) * 1024 + (t0 & 8u);
        const uint32_t lane16 = local_off(lane * 16u), lane2 = local_off(lane * 2u);
        const u32x2  S0 = ld_sbase<u32x2>(s0 + go, lane16), S1 = ld_sbase<u32x2>(s1 + go, lane16);
        const u32x2  P0 = ld_sbase<u32x2>(p0 + go, lane16), P1 = ld_sbase<u32x2>(p1 + go, lane16);
        // uniform, and read through the constant address space: only then does the compiler dare scalar loads (the kernel stores to global
        // memory, and a plain pointer inside an argument struct carries no promise that the table is not what it stores to) -- as vector
        // loads the eight row numbers of a window sat in vector registers and every row address was a 64-bit vector shift and add
        typedef __attribute__((address_space(4))) const uint32_t const_u32_t;
        const_u32_t *rp = reinterpret_cast<const_u32_t *>(reinterpret_cast<uintptr_t>(g.row + t0));
        const uint32_t rw[8] = {rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7]};
        uint32_t aq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!(NOAP || (EARLY && g.noap_rt))) {
#pragma unroll
            for (int r = 0; r < 8; r++) aq[r] = ld_sbase<uint16_t>(arow + (size_t)rw[r] * 128, lane2); // (q of tile 2p) | (q of tile 2p+1) << 8
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const v2s la = as_v2s(__builtin_amdgcn_perm(0u, aq[r], 0x010C000Cu)) >> 7; // 2q per half: the byte in the half's upper byte, arithmetic shift by 7
            lsa[r] = byte_pair_rt(r < 4 ? S0.x : S0.y, r < 4 ? S1.x : S1.y, r & 3) + la;
            lp[r]  = byte_pair_rt(r < 4 ? P0.x : P0.y, r < 4 ? P1.x : P1.y, r & 3);
        }
    };

    v2s a[8];
    a[0] = splat(0);
#pragma unroll
    for (int s = 1; s < 8; s++) a[s] = splat(BCJR_NEG);
    if (seg > 0) { // alpha the previous segment reached in the previous iteration
        const uint4 c0 = g.a_rd[seg * bnd_stride + bnd_lane], c1 = g.a_rd[seg * bnd_stride + bnd_lane + 1];
        a[0] = as_v2s(c0.x); a[1] = as_v2s(c0.y); a[2] = as_v2s(c0.z); a[3] = as_v2s(c0.w);
        a[4] = as_v2s(c1.x); a[5] = as_v2s(c1.y); a[6] = as_v2s(c1.z); a[7] = as_v2s(c1.w);
    }
    for (uint32_t b0 = t_lo; b0 < t_hi; b0 += 32) {
        const uint32_t blk = b0 >> 5, n_w = min(32u, t_hi - b0) >> 3; // 1..4 windows of 8 steps (uniform)
        // ---- forward: alpha, normalised and kept every 8 steps
#pragma unroll 1
        for (uint32_t w = 0; w < n_w; w++) {
            v2s lsa[8], lp[8];
            load_window(b0 + 8 * w, lsa, lp);
            norm8(a);
            chk[w][0][lane] = make_uint4(as_u32(a[0]), as_u32(a[1]), as_u32(a[2]), as_u32(a[3]));
            chk[w][1][lane] = make_uint4(as_u32(a[4]), as_u32(a[5]), as_u32(a[6]), as_u32(a[7]));
#pragma unroll
            for (int r = 0; r < 8; r++) {
                v2s o[8];
                alpha_step(a, lsa[r] + lp[r], lsa[r], lp[r], o);
#pragma unroll
                for (int s = 0; s < 8; s++) a[s] = o[s];
            }
        }

        // ---- beta at the block's end
        v2s b[8];
        if (blk + 1 == n_blk) { // termination: only the a = 0 edges exist, state 2j + r3 continues to state j
            const uint32_t cb0 = min(tile0 * 64 + lane, g.n_cb - 1), cb1 = min(tile1 * 64 + lane, g.n_cb - 1);
            const int8_t  *ta = g.tail + ((size_t)cb0 << 4) + g.tail_off, *tb = g.tail + ((size_t)cb1 << 4) + g.tail_off;
            b[0] = splat(0);
#pragma unroll
            for (int s = 1; s < 8; s++) b[s] = splat(BCJR_NEG);
#pragma unroll
            for (int k = 2; k >= 0; k--) {
                v2s ls, lp;
                ls.x = ta[k]; ls.y = tb[k]; lp.x = ta[3 + k]; lp.y = tb[3 + k];
                const v2s g00 = ls + lp, g01 = ls, g10 = lp;
                v2s       o[8];
                o[0] = b[0] + g00; o[1] = b[0];       o[2] = b[1] + g10; o[3] = b[1] + g01;
                o[4] = b[2] + g01; o[5] = b[2] + g10; o[6] = b[3];       o[7] = b[3] + g00;
#pragma unroll
                for (int s = 0; s < 8; s++) b[s] = o[s];
            }
            norm8(b);
        } else { // what block blk + 1 reached at its start in the previous iteration
            const uint4 c0 = g.b_rd[blk * bnd_stride + bnd_lane], c1 = g.b_rd[blk * bnd_stride + bnd_lane + 1];
            b[0] = as_v2s(c0.x); b[1] = as_v2s(c0.y); b[2] = as_v2s(c0.z); b[3] = as_v2s(c0.w);
            b[4] = as_v2s(c1.x); b[5] = as_v2s(c1.y); b[6] = as_v2s(c1.z); b[7] = as_v2s(c1.w);
        }

        // ---- backward, window by window: alpha re-run from the checkpoint, LLR / extrinsic per step, beta step
#pragma unroll 1
        for (int w = (int)n_w - 1; w >= 0; w--) {
            v2s lsa[8], lp[8];
            load_window(b0 + 8 * w, lsa, lp); // the same 8 steps again (they are one L2 hit away): cheaper than holding 32 steps in registers
            v2s al[8][8];
            {
                const uint4 c0 = chk[w][0][lane], c1 = chk[w][1][lane];
                al[0][0] = as_v2s(c0.x); al[0][1] = as_v2s(c0.y); al[0][2] = as_v2s(c0.z); al[0][3] = as_v2s(c0.w);
                al[0][4] = as_v2s(c1.x); al[0][5] = as_v2s(c1.y); al[0][6] = as_v2s(c1.z); al[0][7] = as_v2s(c1.w);
            }
#pragma unroll
            for (int r = 1; r < 8; r++) alpha_step(al[r - 1], lsa[r - 1] + lp[r - 1], lsa[r - 1], lp[r - 1], al[r]);
            const uint32_t lane2 = local_off(lane * 2u);
#pragma unroll
            for (int r = 7; r >= 0; r--) {
                const v2s g00 = lsa[r] + lp[r], g01 = lsa[r], g10 = lp[r];
                const v2s(&x)[8] = al[r];
                // The twelve sums "branch metric + beta of the edge's end state" serve twice: their pairwise maxima are the beta step, and
                // added to alpha of the edge's start state they are the LLR's candidates (integer + and max: the model's grouping
                // max(m00 + g00, ..) gives the same values).  u0[s] / u1[s]: the edge that leaves state s with u = 0 / u = 1.
                const v2s u0[8] = {b[0] + g00, b[4] + g00, b[5] + g01, b[1] + g01, b[2] + g01, b[6] + g01, b[7] + g00, b[3] + g00};
                const v2s u1[8] = {b[4], b[0], b[1] + g10, b[5] + g10, b[6] + g10, b[2] + g10, b[3], b[7]};
                v2s l0 = x[0] + u0[0], l1 = x[0] + u1[0];
#pragma unroll
                for (int s = 1; s < 8; s++) { l0 = vmax(l0, x[s] + u0[s]); l1 = vmax(l1, x[s] + u1[s]); }
                const v2s llr = l0 - l1;
                v2s       e   = vmin(vmax(llr - lsa[r], splat(-BCJR_X_MAX)), splat(BCJR_X_MAX));
                e             = (e * (v2s)(3)) >> 2;              // the 3/4 scaling (arithmetic shift): within +-255
                e             = vmax(e, splat(-BCJR_LE_MAX)) >> 1; // the stored half (the upper clamp would not change it: 255 >> 1 = 254 >> 1)
                const size_t ro = (size_t)(b0 + 8 * w + r) * 128;
                st_sbase<uint16_t>(erow + ro, lane2, (uint16_t)__builtin_amdgcn_perm(0u, as_u32(e), 0x0C0C0200u)); // the two low bytes
                if (LAST) {
                    const uint32_t neg = as_u32(llr) >> 15; // bit 0: low half negative, bit 16: high half negative
                    const uint16_t hd  = (uint16_t)((neg & 1u) | ((neg >> 8) & 0x100u));
                    if (EARLY) hd_diff |= (uint32_t)(ld_sbase<uint16_t>(hrow + ro, lane2) ^ hd);
                    st_sbase<uint16_t>(hrow + ro, lane2, hd);
                }
#pragma unroll
                for (int s = 0; s < 8; s++) b[s] = vmax(u0[s], u1[s]); // the beta step
            }
            norm8(b);
        }
        if (blk > 0) { // beta at this block's start = the end of block blk - 1, for its next iteration (already normalised)
            g.b_wr[(blk - 1) * bnd_stride + bnd_lane]     = make_uint4(as_u32(b[0]), as_u32(b[1]), as_u32(b[2]), as_u32(b[3]));
            g.b_wr[(blk - 1) * bnd_stride + bnd_lane + 1] = make_uint4(as_u32(b[4]), as_u32(b[5]), as_u32(b[6]), as_u32(b[7]));
        }
    }
    if (seg + 1 < n_seg) { // hand the end state to the next segment's next iteration
        norm8(a);
        g.a_wr[(seg + 1) * bnd_stride + bnd_lane]     = make_uint4(as_u32(a[0]), as_u32(a[1]), as_u32(a[2]), as_u32(a[3]));
        g.a_wr[(seg + 1) * bnd_stride + bnd_lane + 1] = make_uint4(as_u32(a[4]), as_u32(a[5]), as_u32(a[6]), as_u32(a[7]));
    }
    if (EARLY && LAST) { // lanes past the batch end (n_cb % 128 != 0) walk zeros: their decisions never change
        const bool any = g.it == 0 || __any((int)(hd_diff != 0));
        if (lane == 0 && any) atomicOr(&g.chg[(size_t)g.it * g.n_pairs + pair], 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// hard decisions in natural order: c[j] = HD[inv[j]] (decoder 2's last a-posteriori sign), a hole of the de-interleaver falls back on
// the sign of S1[j] (its a-priori value is the zero a hole reads).  One workgroup per (tile pair, 64 bit positions): 64 rows of 128
// bytes come in coalesced, go through LDS, and leave as 64 contiguous bytes per code block.
// (Until late in round 4 the 64 rows were fetched one after the other -- a scalar load of the row number, a byte load per thread, a wait,
// an LDS store, 64 times: 0.62 ms for 65 536 blocks of K = 6144, a tenth of an early-termination decode.  Now the row numbers are read once,
// the rows come in as sixteen independent 4-byte loads per thread, and a code block's 64 bytes leave as four 16-byte stores.)
__global__ __launch_bounds__(128) void k_bcjr_final(const uint8_t *__restrict__ HD, const uint32_t *__restrict__ inv_row, const int8_t *__restrict__ S1,
                                                    uint32_t K, uint32_t n_cb, uint32_t n_tiles, uint8_t *__restrict__ c_bits)
{
    __shared__ uint32_t sm32[64][33]; // 64 rows of 128 bytes (+ 4: the column reads below then fall on different banks)
    __shared__ uint32_t rows[64];
    __shared__ uint64_t hole_rows;
    const uint32_t pair = blockIdx.x, j0 = blockIdx.y * 64, Kp = kpad64(K), th = threadIdx.x;
    const uint8_t *base = HD + ex_row(pair, Kp, 0);
    if (th < 64) {
        const uint32_t j = j0 + th, r = j < K ? inv_row[j] : Kp;
        rows[th] = r;
        const uint64_t holes = __ballot(r == Kp && j < K);
        if (th == 0) hole_rows = holes;
    }
    __syncthreads();
    {
        const uint32_t sub = th >> 5, dw = th & 31u; // four rows per pass, 32 dwords each
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const uint32_t *>(base + (size_t)rows[4 * i + sub] * 128 + 4 * dw);
#pragma unroll
        for (int i = 0; i < 16; i++) sm32[4 * i + sub][dw] = v[i];
    }
    __syncthreads();
    uint8_t (*sm)[132] = reinterpret_cast<uint8_t (*)[132]>(sm32);
    for (uint64_t hm = hole_rows; hm; hm &= hm - 1) { // a hole: the sign of the systematic value of that code block at position j (rare: the wrapped sizes)
        const uint32_t jj = (uint32_t)__builtin_ctzll(hm), jh = j0 + jj;
        const uint32_t tile = min(2 * pair + (th & 1u), n_tiles - 1), lane = th >> 1; // byte th = lane * 2 + h
        sm[jj][th] = (uint8_t)(S1[g8(tile, Kp, lane, jh >> 4) + (jh & 15u)] < 0 ? 1 : 0);
    }
    if (hole_rows) __syncthreads(); // uniform
    const uint32_t h = th >> 6, lane = th & 63u, tile = 2 * pair + h, cb = tile * 64 + lane; // thread -> code block
    if (tile >= n_tiles || cb >= n_cb) return;
    uint8_t *o = c_bits + (size_t)cb * K + j0;
    const uint32_t n = min(64u, K > j0 ? K - j0 : 0u), col = lane * 2 + h; // n: a multiple of 8
#pragma unroll
    for (uint32_t q = 0; q < 64; q += 16) {
        if (q >= n) break; // uniform
        uint32_t w[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            w[k] = (uint32_t)sm[q + 4 * k][col] | (uint32_t)sm[q + 4 * k + 1][col] << 8 | (uint32_t)sm[q + 4 * k + 2][col] << 16 | (uint32_t)sm[q + 4 * k + 3][col] << 24;
        // (cb * K + j0 + q is a multiple of 8, of 16 only for even cb * K / 8: two 8-byte stores)
        *reinterpret_cast<uint2 *>(o + q) = make_uint2(w[0], w[1]);
        if (q + 8 < n) *reinterpret_cast<uint2 *>(o + q + 8) = make_uint2(w[2], w[3]);
    }
}


// ------------------------------------------------------------------------------------------------
// MI_LTE_TURBO_BCJR_BLOCK: ONE code block per wavefront, the whole decode in one launch (the form BASELINE.json's north star sketches; meant
// for the handful of blocks a per-call caller has, where the batch kernels above would leave 63 of 64 lanes idle).  The 64 lanes are 64
// alpha SEGMENTS of the block (lo_bcjr_block_seg_len: every lane a whole number of the 32-step beta blocks, 96 steps at K = 6144), each lane
// with its trellis' eight metrics in its own registers -- the same arithmetic as k_bcjr_half, one trellis per lane, plain 32-bit integers.
// Everything the iterations touch lives in LDS: the four channel-value arrays, both extrinsic arrays, the interleaver tables (as LDS slots), the
// boundary states of both constituent decoders, the alpha checkpoints.  The exchange between the decoders is a byte gather out of LDS.
// Segment g's steps sit at g * (L + 8) + t: the 8 bytes of padding put the 64 lanes' 8-byte window reads on 64 different banks.
// Specification: lo_turbo_decode_bcjr_block (the batch model with alpha restarting every L steps instead of every K / n_seg).
constexpr uint32_t BLK_PAD = 8;
__host__ __device__ inline uint32_t bcjr_block_seg_len(uint32_t K) { return 32u * ((((K + 31u) >> 5) + 63u) >> 6); }
__host__ __device__ inline uint32_t bcjr_block_arr_bytes(uint32_t K) // one int8 array in the padded layout + a zero slot, rounded to 16
{
    const uint32_t L = bcjr_block_seg_len(K), n_seg = (K + L - 1) / L;
    return (n_seg * (L + BLK_PAD) + 16u + 15u) & ~15u;
}
struct BlockLds { uint32_t arr, S1, P1, S2, P2, E1, E2, HD, tabP, tabI, aB, bB, chk, total; };
__host__ __device__ inline BlockLds bcjr_block_lds(uint32_t K)
{
    BlockLds l;
    l.arr = bcjr_block_arr_bytes(K);
    const uint32_t n_blk = (K + 31) >> 5;
    l.S1 = 0; l.P1 = l.arr; l.S2 = 2 * l.arr; l.P2 = 3 * l.arr; l.E1 = 4 * l.arr; l.E2 = 5 * l.arr; l.HD = 6 * l.arr;
    l.tabP = 7 * l.arr;                 // uint16 [arr]: LDS slot of E1 that decoder 2 reads at its step t (slot layout as the arrays)
    l.tabI = l.tabP + 2 * l.arr;        // uint16 [arr]: LDS slot of E2 that decoder 1 reads at its step t (a hole: the zero slot)
    l.aB   = l.tabI + 2 * l.arr;        // int16 [2 decoders][2 buffers][64 segments][8]
    l.bB   = l.aB + 2 * 2 * 64 * 8 * 2; // int16 [2][2][n_blk][8]
    l.chk  = l.bB + 2 * 2 * n_blk * 8 * 2; // int32 [4 windows][8 states][64 lanes]
    l.total = l.chk + 4 * 8 * 64 * 4;
    return l;
}

__device__ __forceinline__ void norm8i(int (&v)[8])
{
    const int m = max(max(max(v[0], v[1]), max(v[2], v[3])), max(max(v[4], v[5]), max(v[6], v[7])));
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = max(v[s] - m, BCJR_NEG);
}
__device__ __forceinline__ void alpha_step_i(const int (&a)[8], int g00, int g01, int g10, int (&o)[8])
{
    o[0] = max(a[0] + g00, a[1]);       o[4] = max(a[0], a[1] + g00);
    o[1] = max(a[2] + g10, a[3] + g01); o[5] = max(a[2] + g01, a[3] + g10);
    o[2] = max(a[4] + g01, a[5] + g10); o[6] = max(a[4] + g10, a[5] + g01);
    o[3] = max(a[6], a[7] + g00);       o[7] = max(a[6] + g00, a[7]);
}

__global__ __launch_bounds__(64) void k_bcjr_block(const int8_t *__restrict__ soft, uint32_t K, uint32_t n_cb, uint32_t n_iter,
                                                   const uint32_t *__restrict__ pi_row, const uint32_t *__restrict__ inv_row, uint8_t *__restrict__ c_bits)
{
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    const uint32_t cb = blockIdx.x, g = threadIdx.x, Kp = kpad64(K), L = bcjr_block_seg_len(K), n_seg = (K + L - 1) / L, n_blk = (K + 31) >> 5;
    const BlockLds lay = bcjr_block_lds(K);
    const uint32_t zero_slot = n_seg * (L + BLK_PAD); // holds 0 in E1, E2: what a hole or the padding past K reads
    const uint32_t inv_L = 0xFFFFFFFFu / L + 1u;      // t / L = mulhi(t, inv_L) for t < 2^16 (L is 32, 64 or 96)
    auto slot = [&](uint32_t t) { const uint32_t q = __umulhi(t, inv_L); return q * (L + BLK_PAD) + (t - q * L); };
    int8_t   *S1 = sm + lay.S1, *P1 = sm + lay.P1, *S2 = sm + lay.S2, *P2 = sm + lay.P2, *E1 = sm + lay.E1, *E2 = sm + lay.E2;
    uint8_t  *HD = reinterpret_cast<uint8_t *>(sm + lay.HD);
    uint16_t *tabP = reinterpret_cast<uint16_t *>(sm + lay.tabP), *tabI = reinterpret_cast<uint16_t *>(sm + lay.tabI);
    int16_t  *aB = reinterpret_cast<int16_t *>(sm + lay.aB), *bB = reinterpret_cast<int16_t *>(sm + lay.bB);
    int      *chk = reinterpret_cast<int *>(sm + lay.chk);
    const int8_t *d = soft + (size_t)cb * 3 * (K + 4);

    // ---- prologue: zero what must read as zero (extrinsics, boundary states = "uniform", the padding), split the input, build the tables
    for (uint32_t w = g; w < (lay.chk >> 2); w += 64) reinterpret_cast<uint32_t *>(sm)[w] = 0u;
    __syncthreads();
    auto clip = [](int v) { return (int8_t)max(v, -127); };
    for (uint32_t i = g; i < K; i += 64) {
        const uint32_t p = slot(i);
        S1[p] = clip(d[3 * i]); P1[p] = clip(d[3 * i + 1]); P2[p] = clip(d[3 * i + 2]);
    }
    for (uint32_t t = g; t < Kp; t += 64) { // the tables as LDS slots; entries from K on (and holes) point at the zero slot
        const uint32_t pr = pi_row[t], ir = inv_row[t], p = t < K ? slot(t) : zero_slot;
        if (t < K) { tabP[p] = (uint16_t)(pr < K ? slot(pr) : zero_slot); tabI[p] = (uint16_t)(ir < K ? slot(ir) : zero_slot); }
    }
    __syncthreads();
    for (uint32_t i = g; i < K; i += 64) S2[slot(i)] = S1[tabP[slot(i)]]; // S2[i] = S1[pi[i]]
    // termination bits: x[3r + stream] = d_stream[K + r]  (36.212 5.1.3.2.2), as in k_bcjr_prep
    int tl[12];
    {
        const int8_t *x = d + 3 * (size_t)K;
        const int idx[12] = {0, 2, 4, 1, 3, 5, 6, 8, 10, 7, 9, 11};
#pragma unroll
        for (int k = 0; k < 12; k++) tl[k] = max((int)x[idx[k]], -127);
    }
    __syncthreads();

    const uint32_t t_lo = g * L, t_hi = min(t_lo + L, K), base = g * (L + BLK_PAD); // this lane's segment, and its first slot
    const bool     on = g < n_seg;

    // one half-iteration of constituent decoder `dec` (0 / 1) in iteration `it`; LAST: also the decisions (decoder 2's order)
    auto half = [&](uint32_t dec, uint32_t it, bool last) {
        const int8_t   *S = dec ? S2 : S1, *P = dec ? P2 : P1, *A = dec ? E1 : E2;
        int8_t         *E = dec ? E2 : E1;
        const uint16_t *row = dec ? tabP : tabI;
        const uint32_t  rd = it & 1u, wr = rd ^ 1u;
        int16_t        *a_rd = aB + ((dec * 2 + rd) * 64) * 8, *a_wr = aB + ((dec * 2 + wr) * 64) * 8;
        int16_t        *b_rd = bB + ((dec * 2 + rd) * n_blk) * 8, *b_wr = bB + ((dec * 2 + wr) * n_blk) * 8;
        if (!on) return;
        auto load_window = [&](uint32_t p0, int (&lsa)[8], int (&lp)[8]) { // the eight steps whose first slot is p0 (a multiple of 8)
            const uint2 sv = *reinterpret_cast<const uint2 *>(S + p0), pv = *reinterpret_cast<const uint2 *>(P + p0);
            const uint4 rw = *reinterpret_cast<const uint4 *>(row + p0);
            const uint32_t r[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int a = A[(r[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
                lsa[k] = sb(k < 4 ? sv.x : sv.y, k & 3) + 2 * a;
                lp[k]  = sb(k < 4 ? pv.x : pv.y, k & 3);
            }
        };
        int a[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
        if (g > 0) {
#pragma unroll
            for (int s = 0; s < 8; s++) a[s] = a_rd[g * 8 + s];
        }
        for (uint32_t b0 = t_lo; b0 < t_hi; b0 += 32) {
            const uint32_t blk = b0 >> 5, n_w = min(32u, t_hi - b0) >> 3, p_blk = base + (b0 - t_lo);
#pragma unroll 1
            for (uint32_t w = 0; w < n_w; w++) {
                int lsa[8], lp[8];
                load_window(p_blk + 8 * w, lsa, lp);
                norm8i(a);
#pragma unroll
                for (int s = 0; s < 8; s++) chk[(w * 8 + s) * 64 + g] = a[s];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    int o[8];
                    alpha_step_i(a, lsa[r] + lp[r], lsa[r], lp[r], o);
#pragma unroll
                    for (int s = 0; s < 8; s++) a[s] = o[s];
                }
            }
            int b[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
            if (blk + 1 == n_blk) { // termination: only the a = 0 edges exist, state 2j + r3 continues to state j
#pragma unroll
                for (int k = 2; k >= 0; k--) {
                    const int ls = tl[6 * dec + k], lp = tl[6 * dec + 3 + k], g00 = ls + lp, g01 = ls, g10 = lp;
                    int       o[8];
                    o[0] = b[0] + g00; o[1] = b[0];       o[2] = b[1] + g10; o[3] = b[1] + g01;
                    o[4] = b[2] + g01; o[5] = b[2] + g10; o[6] = b[3];       o[7] = b[3] + g00;
#pragma unroll
                    for (int s = 0; s < 8; s++) b[s] = o[s];
                }
                norm8i(b);
            } else {
#pragma unroll
                for (int s = 0; s < 8; s++) b[s] = b_rd[blk * 8 + s];
            }
#pragma unroll 1
            for (int w = (int)n_w - 1; w >= 0; w--) {
                int lsa[8], lp[8];
                load_window(p_blk + 8 * w, lsa, lp);
                int al[8][8];
#pragma unroll
                for (int s = 0; s < 8; s++) al[0][s] = chk[(w * 8 + s) * 64 + g];
#pragma unroll
                for (int r = 1; r < 8; r++) alpha_step_i(al[r - 1], lsa[r - 1] + lp[r - 1], lsa[r - 1], lp[r - 1], al[r]);
#pragma unroll
                for (int r = 7; r >= 0; r--) {
                    const int g00 = lsa[r] + lp[r], g01 = lsa[r], g10 = lp[r];
                    const int(&x)[8] = al[r];
                    const int u0[8] = {b[0] + g00, b[4] + g00, b[5] + g01, b[1] + g01, b[2] + g01, b[6] + g01, b[7] + g00, b[3] + g00};
                    const int u1[8] = {b[4], b[0], b[1] + g10, b[5] + g10, b[6] + g10, b[2] + g10, b[3], b[7]};
                    int l0 = x[0] + u0[0], l1 = x[0] + u1[0];
#pragma unroll
                    for (int s = 1; s < 8; s++) { l0 = max(l0, x[s] + u0[s]); l1 = max(l1, x[s] + u1[s]); }
                    const int llr = l0 - l1;
                    int       e   = min(max(llr - lsa[r], -BCJR_X_MAX), BCJR_X_MAX);
                    e             = (e * 3) >> 2;
                    e             = min(max(e, -BCJR_LE_MAX), BCJR_LE_MAX) >> 1;
                    const uint32_t p = p_blk + 8 * w + r;
                    E[p] = (int8_t)e;
                    if (last) HD[p] = llr < 0 ? 1 : 0;
#pragma unroll
                    for (int s = 0; s < 8; s++) b[s] = max(u0[s], u1[s]);
                }
                norm8i(b);
            }
            if (blk > 0) {
#pragma unroll
                for (int s = 0; s < 8; s++) b_wr[(blk - 1) * 8 + s] = (int16_t)b[s];
            }
        }
        if (g + 1 < n_seg) {
            norm8i(a);
#pragma unroll
            for (int s = 0; s < 8; s++) a_wr[(g + 1) * 8 + s] = (int16_t)a[s];
        }
    };
    for (uint32_t it = 0; it < n_iter; it++) {
        half(0, it, false);
        __syncthreads();
        half(1, it, it + 1 == n_iter);
        __syncthreads();
    }
    // ---- decisions in natural order: c[j] = HD[inv[j]]; a hole of the de-interleaver falls back on the sign of S1[j]
    uint8_t *o = c_bits + (size_t)cb * K;
    for (uint32_t j = g; j < K; j += 64) {
        const uint32_t p = slot(j), r = tabI[p];
        o[j] = r != zero_slot ? HD[r] : (uint8_t)(S1[p] < 0 ? 1 : 0);
    }
}

} // namespace

extern "C" size_t mi_lte_turbo_bcjr_scratch_bytes(uint32_t K, uint32_t n_cb)
{
    const size_t n_tiles = (n_cb + 63) / 64, n_pairs = (n_tiles + 1) / 2, Kp = kpad64(K), n_blk = (Kp + 31) / 32;
    return n_tiles * (Kp * 64 * 4 + 64 * 16)                 // S1 P1 S2 P2, tails
           + n_pairs * ((Kp + 1) * 128 * 3)                  // E1 E2 HD rows
           + n_pairs * 64 * 32 * (2 * 2 * 8 + 2 * 2 * n_blk) // boundary states [decoder][buffer][segment | block]
           + n_tiles * 64 * 32                                // 32 bytes per code block for the caller's prep kernel (MiBcjrBufs::aux)
           + n_pairs * 64 * 4                                 // early termination: one word per (iteration <= 64, pair)
           + 4096;
}

// The decode in three steps, so that a caller with its own first kernel (the PDSCH chain rate-un-matches straight into the granule arrays:
// k_rm_bcjr_prep, turbo.hip) can replace the middle one:
//   mi_turbo_bcjr_begin    scratch laid out and zeroed where the first half-iteration must read zero; returns the arrays the prep kernel fills
//   (prep)                 S1 P1 S2 P2 granules + termination records
//   mi_turbo_bcjr_iterate  n_iter full iterations + the decisions in natural order
struct BcjrLayout { size_t n_tiles, n_pairs, Kp, a8, ex, n_blk, one, per_buf; uint8_t *base; int8_t *E1, *E2; uint8_t *HD; uint4 *bnd0; uint32_t *chg; };
static BcjrLayout bcjr_layout(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb)
{
    BcjrLayout l;
    l.n_tiles = (n_cb + 63) / 64; l.n_pairs = (l.n_tiles + 1) / 2; l.Kp = kpad64(K); l.a8 = l.n_tiles * l.Kp * 64; l.ex = l.n_pairs * (l.Kp + 1) * 128;
    l.n_blk = (l.Kp + 31) / 32;
    l.base  = (uint8_t *)ctx->scratch;
    l.E1    = (int8_t *)(l.base + 4 * l.a8 + l.n_tiles * 64 * 16); l.E2 = l.E1 + l.ex;
    l.HD    = (uint8_t *)(l.E2 + l.ex);
    // boundary states: per decoder and buffer, alpha [8 segments] then beta [n_blk blocks], each [pair][lane][8 x v2s = 2 x uint4]; uniform (0) before the first iteration
    l.one     = l.n_pairs * 64 * 2; // uint4 per [segment | block]
    l.bnd0    = (uint4 *)(((uintptr_t)(l.HD + l.ex) + 255) & ~(uintptr_t)255);
    l.per_buf = (8 + l.n_blk) * l.one;
    const uintptr_t aux = ((uintptr_t)(l.bnd0 + 4 * l.per_buf) + 255) & ~(uintptr_t)255; // MiBcjrBufs::aux, 32 bytes per code block
    l.chg = (uint32_t *)((aux + l.n_tiles * 64 * 32 + 255) & ~(uintptr_t)255);
    return l;
}

int mi_turbo_bcjr_begin(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, MiBcjrBufs *out)
{
    int rc = mi_ctx_reserve_scratch(ctx, mi_lte_turbo_bcjr_scratch_bytes(K, n_cb));
    if (rc != MI_LTE_OK) return rc;
    const BcjrLayout l = bcjr_layout(ctx, K, n_cb);
    out->S1 = (int8_t *)l.base; out->P1 = out->S1 + l.a8; out->S2 = out->P1 + l.a8; out->P2 = out->S2 + l.a8;
    out->tail = (int8_t *)(l.base + 4 * l.a8);
    out->aux  = (void *)(((uintptr_t)(l.bnd0 + 4 * l.per_buf) + 255) & ~(uintptr_t)255);
    // boundary states "uniform" before the first iteration: the buffer iteration 0 reads (index 0) of either decoder; iteration 0 writes the other one
    MI_HIP_CHECK(ctx, hipMemsetAsync(l.bnd0, 0, l.per_buf * sizeof(uint4), ctx->stream));
    MI_HIP_CHECK(ctx, hipMemsetAsync(l.bnd0 + 2 * l.per_buf, 0, l.per_buf * sizeof(uint4), ctx->stream));
    // the zero row of every pair in E1 and E2 (what a hole of the interleaver and the padding past K read); the a-priori values of the
    // first half-iteration are zero too, but that launch does not read them (k_bcjr_half's NOAP) -- filling E2 for it was 400 MB per decode
    MI_HIP_CHECK(ctx, hipMemset2DAsync(l.E1 + l.Kp * 128, (l.Kp + 1) * 128, 0, 128, l.n_pairs, ctx->stream));
    MI_HIP_CHECK(ctx, hipMemset2DAsync(l.E2 + l.Kp * 128, (l.Kp + 1) * 128, 0, 128, l.n_pairs, ctx->stream));
    if (n_cb % 64 || (l.n_tiles & 1)) MI_HIP_CHECK(ctx, hipMemsetAsync(l.base, 0, 4 * l.a8 + l.n_tiles * 64 * 16, ctx->stream)); // lanes past the batch end stay defined
    return MI_LTE_OK;
}

int mi_turbo_bcjr_iterate(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits, bool early)
{
    if (n_iter == 0 || n_iter > 64) return MI_LTE_ERR_INVALID_ARG;
    TurboTables tb;
    int         rc = mi_ctx_turbo_tables(ctx, K, qpp_spec ? 1 : 0, &tb);
    if (rc != MI_LTE_OK) return rc;
    const BcjrLayout l = bcjr_layout(ctx, K, n_cb);
    const size_t     n_pairs = l.n_pairs, Kp = l.Kp, one = l.one, per_buf = l.per_buf;
    BcjrBufs B;
    B.S1 = (int8_t *)l.base; B.P1 = B.S1 + l.a8; B.S2 = B.P1 + l.a8; B.P2 = B.S2 + l.a8;
    B.tail = (int8_t *)(l.base + 4 * l.a8);
    int8_t  *E1 = l.E1, *E2 = l.E2;
    uint8_t *HD = l.HD;
    uint4   *bnd0 = l.bnd0;
    const size_t n_tiles = l.n_tiles;
    const uint32_t n_seg = bcjr_n_seg(K);
    if (early) // the change words start at zero (the decisions need no initial value: the first iteration always counts as a change)
        MI_HIP_CHECK(ctx, hipMemsetAsync(l.chg, 0, (size_t)n_iter * n_pairs * sizeof(uint32_t), ctx->stream));
    ctx->bcjr_early = {nullptr, (uint32_t)n_pairs, n_iter, n_cb}; // set once the change words sit in the context's own buffer (below)
    for (uint32_t it = 0; it < n_iter; it++) {
        const bool last = it + 1 == n_iter;
        const uint32_t rd = it & 1u, wr = rd ^ 1u;
        auto args = [&](int dec) {
            uint4 *d = bnd0 + (size_t)dec * 2 * per_buf;
            HalfArgs h;
            h.S = dec ? B.S2 : B.S1; h.P = dec ? B.P2 : B.P1;
            h.A = dec ? E1 : E2; h.row = dec ? tb.d_pi_row : tb.d_inv_row; h.E = dec ? E2 : E1; h.HD = HD;
            h.tail = B.tail; h.tail_off = dec ? 6u : 0u;
            h.a_rd = d + rd * per_buf; h.b_rd = d + rd * per_buf + 8 * one; h.a_wr = d + wr * per_buf; h.b_wr = d + wr * per_buf + 8 * one;
            h.K = K; h.n_cb = n_cb; h.n_tiles = (uint32_t)n_tiles; h.n_pairs = (uint32_t)n_pairs;
            h.chg = l.chg; h.it = it; h.noap_rt = (early && it == 0 && dec == 0) ? 1u : 0u;
            return h;
        };
        if (early) {
            MI_LAUNCH(ctx, "k_bcjr_half", (k_bcjr_half<false, true>), dim3(n_pairs, n_seg), dim3(64), 0, args(0));
            MI_LAUNCH(ctx, "k_bcjr_half", (k_bcjr_half<true, true>), dim3(n_pairs, n_seg), dim3(64), 0, args(1));
            continue;
        }
        if (it == 0) MI_LAUNCH(ctx, "k_bcjr_half", (k_bcjr_half<false, false, true>), dim3(n_pairs, n_seg), dim3(64), 0, args(0)); // (its a-priori array is not filled)
        else         MI_LAUNCH(ctx, "k_bcjr_half", k_bcjr_half<false>, dim3(n_pairs, n_seg), dim3(64), 0, args(0));
        if (!last) MI_LAUNCH(ctx, "k_bcjr_half", k_bcjr_half<false>, dim3(n_pairs, n_seg), dim3(64), 0, args(1));
        else       MI_LAUNCH(ctx, "k_bcjr_half", k_bcjr_half<true>, dim3(n_pairs, n_seg), dim3(64), 0, args(1));
    }
    MI_LAUNCH(ctx, "k_bcjr_final", k_bcjr_final, dim3(n_pairs, Kp / 64), dim3(128), 0, (const uint8_t *)HD, (const uint32_t *)tb.d_inv_row, (const int8_t *)B.S1, K, n_cb,
              (uint32_t)n_tiles, d_c_bits);
    MI_HIP_CHECK(ctx, hipGetLastError());
    if (early) { // mi_lte_turbo_early_exit_iterations may be asked after other work has reused (or re-allocated) the scratch: keep a copy
        const size_t n_words = (size_t)n_iter * n_pairs;
        if (n_words > ctx->bcjr_early_cap) {
            uint32_t *nb = nullptr;
            MI_HIP_CHECK(ctx, hipMalloc((void **)&nb, n_words * sizeof(uint32_t)));
            ctx->owned.push_back(nb); // (the smaller one it replaces stays owned until the context goes)
            ctx->bcjr_early_buf = nb; ctx->bcjr_early_cap = n_words;
        }
        MI_HIP_CHECK(ctx, hipMemcpyAsync(ctx->bcjr_early_buf, l.chg, n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->bcjr_early.chg = ctx->bcjr_early_buf;
    }
    ctx->last_kernels = "k_bcjr_prep:1,k_bcjr_half: 2 per iteration,k_bcjr_final:1";
    return MI_LTE_OK;
}

// n_iter full iterations of max-log-MAP over n_cb code blocks of size K; int8 soft input in the reference's layout
int mi_turbo_bcjr_batch(mi_lte_ctx *ctx, const int8_t *d_soft, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits, bool early)
{
    if (n_iter == 0 || n_iter > 64) return MI_LTE_ERR_INVALID_ARG;
    TurboTables tb;
    int         rc = mi_ctx_turbo_tables(ctx, K, qpp_spec ? 1 : 0, &tb);
    if (rc != MI_LTE_OK) return rc;
    MiBcjrBufs mb;
    rc = mi_turbo_bcjr_begin(ctx, K, n_cb, &mb);
    if (rc != MI_LTE_OK) return rc;
    BcjrBufs B{mb.S1, mb.P1, mb.S2, mb.P2, mb.tail};
    const size_t   Kp = kpad64(K);
    const uint32_t cb_threads = (uint32_t)(((Kp >> 4) + 63) & ~(size_t)63);
    static const bool one_per_wg = [] { const char *e = getenv("MI_LTE_BCJR_PREP1"); return e && atoi(e) != 0; }(); // (A/B: the one-block-per-workgroup kernel)
    if (one_per_wg || 8 * Kp > 64 * 1024) MI_LAUNCH(ctx, "k_bcjr_prep", k_bcjr_prep, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), Kp, d_soft, K, n_cb, tb.d_pi, B);
    else MI_LAUNCH(ctx, "k_bcjr_prep", k_bcjr_prep8, dim3(xcd_chunk(n_cb)), dim3(512), 8 * Kp, d_soft, K, n_cb, tb.d_pi, B);
    (void)cb_threads;
    return mi_turbo_bcjr_iterate(ctx, K, n_cb, n_iter, qpp_spec, d_c_bits, early);
}

// MI_LTE_TURBO_BCJR_BLOCK: one wavefront per code block, one launch for the whole decode (k_bcjr_block)
int mi_turbo_bcjr_block_batch(mi_lte_ctx *ctx, const int8_t *d_soft, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits)
{
    if (n_iter == 0 || n_iter > 64) return MI_LTE_ERR_INVALID_ARG;
    TurboTables tb;
    int         rc = mi_ctx_turbo_tables(ctx, K, qpp_spec ? 1 : 0, &tb);
    if (rc != MI_LTE_OK) return rc;
    const BlockLds lay = bcjr_block_lds(K);
    if (!ctx->bcjr_block_lds_set) { // more than 64 KB of dynamic LDS has to be asked for: a per-DEVICE attribute, so once per context
        MI_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bcjr_block, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->bcjr_block_lds_set = true;
    }
    MI_LAUNCH(ctx, "k_bcjr_block", k_bcjr_block, dim3(n_cb), dim3(64), lay.total, d_soft, K, n_cb, n_iter, (const uint32_t *)tb.d_pi_row,
              (const uint32_t *)tb.d_inv_row, d_c_bits);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_bcjr_block:1";
    return MI_LTE_OK;
}

// iterations each tile pair of the last MI_LTE_TURBO_BCJR_EARLY decode on this context ran (host side: reads the change words back)
extern "C" int mi_lte_turbo_early_exit_iterations(mi_lte_ctx *ctx, uint32_t *h_pair_iters, uint32_t max_pairs, uint32_t *n_pairs, uint32_t *n_iter)
{
    if (!ctx || !h_pair_iters || !n_pairs || !n_iter) return MI_LTE_ERR_INVALID_ARG;
    if (!ctx->bcjr_early.chg) { ctx->err = "no early-termination decode has run on this context"; return MI_LTE_ERR_INVALID_ARG; }
    const uint32_t np = ctx->bcjr_early.n_pairs, ni = ctx->bcjr_early.n_iter;
    *n_pairs = np; *n_iter = ni;
    if (max_pairs < np) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> chg((size_t)ni * np);
    MI_D2H(ctx, chg.data(), ctx->bcjr_early.chg, chg.size() * sizeof(uint32_t));
    MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t p = 0; p < np; p++) {
        uint32_t done = ni;
        for (uint32_t it = 2; it < ni; it++)
            if (chg[(size_t)(it - 1) * np + p] == 0) { done = it; break; }
        h_pair_iters[p] = done;
    }
    return MI_LTE_OK;
}
