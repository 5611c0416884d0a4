// Max-log-MAP ("BCJR") turbo decoding on gfx950 -- the decoder BASELINE.json's north star sketches, offered as
// MI_LTE_TURBO_BCJR next to the reference-faithful REF mode.  The reference has no such decoder, so
// this mode is specified by a plain-C fixed-point model in the test tree (lo_turbo_decode_bcjr): every operation
// in the same order, and the kernels below must match it bit for bit.
//
// Mapping: the same lock-step tiles as the REF decoder (lane = code block, 64 trellises per wavefront, all
// per-step arrays "line per block" in HBM) -- the 8 alpha / beta metrics of a block live in its lane's VGPRs,
// max* is a plain v_max, nothing crosses lanes.  A forward pass stores the normalised alpha vector every 8
// steps (16 B per block per 8 steps); the backward pass re-runs alpha inside each 8-step window from that
// checkpoint (56 values in registers) while beta walks down, so no per-step state ever goes to memory.
// The extrinsic permutation between the two constituent decoders is a per-code-block LDS gather.
// Occupancy: a 64k-block batch is only 1024 tiles, one wave per SIMD; blocks of >= 1024 steps are therefore cut
// into 4 (2) segments decoded by separate waves, whose boundary alpha / beta come from the neighbouring segment's
// previous iteration ("next iteration initialisation", double-buffered) -- part of the mode's specification.
//
// Trellis (36.212 5.1.3.2.1, feedback 1+D^2+D^3, parity 1+D+D^3; state = 4 r1 + 2 r2 + r3): the predecessors of
// state n are 2(n&3) and 2(n&3)+1 with complementary (u,z) labels -- see the table next to lo_turbo_decode_bcjr.  Branch
// metric g(u,z) = [u==0](Ls+La) + [z==0]Lp, positive LLR = bit 0.
#include "ctx.hpp"

namespace {

constexpr int BCJR_NEG = -32000, BCJR_LE_MAX = 1023;

__host__ __device__ inline uint32_t kpad64(uint32_t K) { return (K + 63u) & ~63u; }
__device__ __forceinline__ int sb(uint32_t w, int k) { return (int)__builtin_amdgcn_sbfe(w, 8 * k, 8); }
__device__ __forceinline__ int sh(uint32_t w, int k) { return (int)__builtin_amdgcn_sbfe(w, 16 * k, 16); }
__device__ __forceinline__ uint32_t pk16(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

__device__ __forceinline__ size_t g8(uint32_t tile, uint32_t Kp, uint32_t lane, uint32_t gran) { return (size_t)tile * Kp * 64 + (size_t)gran * 1024 + lane * 16; }   // byte offset, int8 arrays
__device__ __forceinline__ size_t g16(uint32_t tile, uint32_t Kp, uint32_t lane, uint32_t gran) { return (size_t)tile * Kp * 128 + (size_t)gran * 1024 + lane * 16; } // byte offset, int16 arrays
// XCD-aware block -> code block mapping of the per-code-block kernels (see turbo.hip)
__host__ __device__ inline uint32_t xcd_chunk(uint32_t n_cb) { return ((((n_cb + 63u) >> 6) + 7u) >> 3) << 6; }
__device__ __forceinline__ uint32_t xcd_cb(uint32_t b, uint32_t n_cb) { return (b & 7u) * xcd_chunk(n_cb) + (b >> 3); }

__host__ __device__ inline uint32_t bcjr_n_seg(uint32_t K)
{
    const uint32_t nblk = (K + 63) / 64;
    for (uint32_t n = 4; n > 1; n >>= 1)
        if (nblk % n == 0 && (nblk / n) * 64 >= 512) return n;
    return 1;
}
struct BcjrBnd { int16_t *a_rd, *a_wr, *b_rd, *b_wr; uint32_t n_tiles; }; // [segment][tile][lane][8] each

__device__ __forceinline__ void norm8(int (&v)[8]) // subtract the maximum, floor at BCJR_NEG
{
    const int m = max(max(max(v[0], v[1]), max(v[2], v[3])), max(max(v[4], v[5]), max(v[6], v[7])));
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = max(v[s] - m, BCJR_NEG);
}
__device__ __forceinline__ void alpha_step(const int (&a)[8], int g00, int g01, int g10, int (&o)[8])
{
    o[0] = max(a[0] + g00, a[1]);       o[4] = max(a[0], a[1] + g00);
    o[1] = max(a[2] + g10, a[3] + g01); o[5] = max(a[2] + g01, a[3] + g10);
    o[2] = max(a[4] + g01, a[5] + g10); o[6] = max(a[4] + g10, a[5] + g01);
    o[3] = max(a[6], a[7] + g00);       o[7] = max(a[6] + g00, a[7]);
}
__device__ __forceinline__ void beta_step(int (&b)[8], int g00, int g01, int g10)
{
    int o[8];
    o[0] = max(b[0] + g00, b[4]);       o[1] = max(b[0], b[4] + g00);
    o[2] = max(b[1] + g10, b[5] + g01); o[3] = max(b[1] + g01, b[5] + g10);
    o[4] = max(b[2] + g01, b[6] + g10); o[5] = max(b[2] + g10, b[6] + g01);
    o[6] = max(b[3], b[7] + g00);       o[7] = max(b[3] + g00, b[7]);
#pragma unroll
    for (int s = 0; s < 8; s++) b[s] = o[s];
}

// ---- layouts of one tile (64 code blocks), "16-byte granules": the lock-step kernels consume 16 B per lane at a time,
//      so granule g of lane l sits at g*1024 + l*16 -- every wave-wide access is one contiguous KiB.
//      int8 arrays: granule = 16 steps (step t in granule t/16, byte t%16); int16 arrays: granule = 8 steps;
//      checkpoints: [window][lane][8 x int16] (the same shape); tails: [lane][16 B]
struct BcjrBufs {
    int8_t  *S1, *P1, *S2, *P2; // systematic / parity of the two constituent decoders (S2 = interleaved S1)
    int16_t *A, *E, *post;      // a-priori in, extrinsic out, a-posteriori of the last half-iteration
    int16_t *chk;
    int8_t  *tail;              // per block: t1s[3] t1p[3] t2s[3] t2p[3] pad[4]
};

// ------------------------------------------------------------------------------------------------
// prep: split the interleaved d[i*3+x] input into tile lines, interleave the systematic stream, clear the a-priori
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
    if (nv >= 0) { // a-priori = 0 (two uint4 of int16 per unit)
        char *ab = reinterpret_cast<char *>(B.A);
        *reinterpret_cast<uint4 *>(ab + g16(tile, Kp, lane, 2 * u))     = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(ab + g16(tile, Kp, lane, 2 * u + 1)) = make_uint4(0, 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------
// forward pass of one constituent decoder: alpha, normalised and checkpointed every 8 steps
__global__ __launch_bounds__(64) void k_bcjr_fwd(const int8_t *__restrict__ S, const int8_t *__restrict__ P,
                                                 const int16_t *__restrict__ A, int16_t *__restrict__ chk, uint32_t K, BcjrBnd bnd)
{
    const uint32_t tile = blockIdx.x, seg = blockIdx.y, n_seg = gridDim.y, lane = threadIdx.x, Kp = kpad64(K), nblk = Kp >> 6, n_win = Kp >> 3;
    const uint32_t seg_blk = nblk / n_seg;
    const char *ps = reinterpret_cast<const char *>(S), *pp = reinterpret_cast<const char *>(P), *pa = reinterpret_cast<const char *>(A);
    uint4      *pc = reinterpret_cast<uint4 *>(chk + ((size_t)tile * n_win * 64 + lane) * 8);
    (void)nblk;
    int a[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
    if (seg > 0) { // alpha the previous segment reached in the previous iteration
        const uint4 c4 = *reinterpret_cast<const uint4 *>(bnd.a_rd + (((size_t)seg * bnd.n_tiles + tile) * 64 + lane) * 8);
        a[0] = sh(c4.x, 0); a[1] = sh(c4.x, 1); a[2] = sh(c4.y, 0); a[3] = sh(c4.y, 1);
        a[4] = sh(c4.z, 0); a[5] = sh(c4.z, 1); a[6] = sh(c4.w, 0); a[7] = sh(c4.w, 1);
    }
    for (uint32_t blk = seg * seg_blk; blk < (seg + 1) * seg_blk; blk++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { // 16 steps = 2 windows per quarter line
            const uint32_t t0 = blk * 64 + q * 16;
            if (t0 >= K) break; // uniform
            const uint4 s4 = *reinterpret_cast<const uint4 *>(ps + g8(tile, Kp, lane, t0 >> 4));
            const uint4 p4 = *reinterpret_cast<const uint4 *>(pp + g8(tile, Kp, lane, t0 >> 4));
            const uint4 a0 = *reinterpret_cast<const uint4 *>(pa + g16(tile, Kp, lane, t0 >> 3));
            const uint4 a1 = *reinterpret_cast<const uint4 *>(pa + g16(tile, Kp, lane, (t0 >> 3) + 1));
            const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
            const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int w = 0; w < 2; w++) {
                if (t0 + 8 * w >= K) break; // uniform (K % 8 == 0)
                norm8(a);
                pc[(size_t)((t0 >> 3) + w) * 64] = make_uint4(pk16(a[0], a[1]), pk16(a[2], a[3]), pk16(a[4], a[5]), pk16(a[6], a[7]));
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = 8 * w + r, lsa = sb(sw[i >> 2], i & 3) + sh(aw[i >> 1], i & 1), lp = sb(pw[i >> 2], i & 3);
                    int o[8];
                    alpha_step(a, lsa + lp, lsa, lp, o);
#pragma unroll
                    for (int s = 0; s < 8; s++) a[s] = o[s];
                }
            }
        }
    }
    if (seg + 1 < n_seg) { // hand the end state to the next segment's next iteration
        norm8(a);
        *reinterpret_cast<uint4 *>(bnd.a_wr + (((size_t)(seg + 1) * bnd.n_tiles + tile) * 64 + lane) * 8) =
            make_uint4(pk16(a[0], a[1]), pk16(a[2], a[3]), pk16(a[4], a[5]), pk16(a[6], a[7]));
    }
}

// ------------------------------------------------------------------------------------------------
// backward pass: beta from the termination, then window by window: alpha re-run from the checkpoint,
// LLR / extrinsic per step, beta step; beta normalised after every window
template <bool POST>
__global__ __launch_bounds__(64) void k_bcjr_bwd(const int8_t *__restrict__ S, const int8_t *__restrict__ P,
                                                 const int16_t *__restrict__ A, const int16_t *__restrict__ chk,
                                                 const int8_t *__restrict__ tail, uint32_t tail_off, int16_t *__restrict__ E,
                                                 int16_t *__restrict__ post, uint32_t K, uint32_t n_cb, BcjrBnd bnd)
{
    const uint32_t tile = blockIdx.x, seg = blockIdx.y, n_seg = gridDim.y, lane = threadIdx.x, Kp = kpad64(K), nblk = Kp >> 6, n_win = Kp >> 3;
    const uint32_t seg_blk = nblk / n_seg;
    const char  *ps = reinterpret_cast<const char *>(S), *pp = reinterpret_cast<const char *>(P), *pa = reinterpret_cast<const char *>(A);
    char        *pe = reinterpret_cast<char *>(E), *po = POST ? reinterpret_cast<char *>(post) : nullptr;
    const uint4 *pc = reinterpret_cast<const uint4 *>(chk + ((size_t)tile * n_win * 64 + lane) * 8);
    (void)nblk;

    int b[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
    if (seg + 1 < n_seg) { // beta the next segment reached in the previous iteration
        const uint4 c4 = *reinterpret_cast<const uint4 *>(bnd.b_rd + (((size_t)seg * bnd.n_tiles + tile) * 64 + lane) * 8);
        b[0] = sh(c4.x, 0); b[1] = sh(c4.x, 1); b[2] = sh(c4.y, 0); b[3] = sh(c4.y, 1);
        b[4] = sh(c4.z, 0); b[5] = sh(c4.z, 1); b[6] = sh(c4.w, 0); b[7] = sh(c4.w, 1);
    } else {   // termination: only the a = 0 edges exist, state 2j + r3 continues to state j
        const uint32_t cb = min(tile * 64 + lane, n_cb - 1);
        const int8_t  *t  = tail + ((size_t)cb << 4) + tail_off;
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            const int ls = t[k], lp = t[3 + k], g00 = ls + lp, g01 = ls, g10 = lp;
            int o[8];
            o[0] = b[0] + g00; o[1] = b[0];       o[2] = b[1] + g10; o[3] = b[1] + g01;
            o[4] = b[2] + g01; o[5] = b[2] + g10; o[6] = b[3];       o[7] = b[3] + g00;
#pragma unroll
            for (int s = 0; s < 8; s++) b[s] = o[s];
        }
        norm8(b);
    }
    for (int blk = (int)((seg + 1) * seg_blk) - 1; blk >= (int)(seg * seg_blk); blk--) {
#pragma unroll
        for (int q = 3; q >= 0; q--) {
            const uint32_t t0 = (uint32_t)blk * 64 + q * 16;
            if (t0 >= K) continue; // uniform
            const uint4 s4 = *reinterpret_cast<const uint4 *>(ps + g8(tile, Kp, lane, t0 >> 4));
            const uint4 p4 = *reinterpret_cast<const uint4 *>(pp + g8(tile, Kp, lane, t0 >> 4));
            const uint4 a0 = *reinterpret_cast<const uint4 *>(pa + g16(tile, Kp, lane, t0 >> 3));
            const uint4 a1 = *reinterpret_cast<const uint4 *>(pa + g16(tile, Kp, lane, (t0 >> 3) + 1));
            const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
            const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int w = 1; w >= 0; w--) {
                if (t0 + 8 * w >= K) continue; // uniform
                const uint4 c4 = pc[(size_t)((t0 >> 3) + w) * 64];
                int al[8][8];
                al[0][0] = sh(c4.x, 0); al[0][1] = sh(c4.x, 1); al[0][2] = sh(c4.y, 0); al[0][3] = sh(c4.y, 1);
                al[0][4] = sh(c4.z, 0); al[0][5] = sh(c4.z, 1); al[0][6] = sh(c4.w, 0); al[0][7] = sh(c4.w, 1);
                int lsa[8], lp[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = 8 * w + r;
                    lsa[r] = sb(sw[i >> 2], i & 3) + sh(aw[i >> 1], i & 1);
                    lp[r]  = sb(pw[i >> 2], i & 3);
                }
#pragma unroll
                for (int r = 1; r < 8; r++) alpha_step(al[r - 1], lsa[r - 1] + lp[r - 1], lsa[r - 1], lp[r - 1], al[r]);
                int ev[8], pv[8];
#pragma unroll
                for (int r = 7; r >= 0; r--) {
                    const int g00 = lsa[r] + lp[r], g01 = lsa[r], g10 = lp[r];
                    const int(&x)[8] = al[r];
                    const int m00 = max(max(x[0] + b[0], x[1] + b[4]), max(x[7] + b[3], x[6] + b[7]));
                    const int m01 = max(max(x[3] + b[1], x[2] + b[5]), max(x[4] + b[2], x[5] + b[6]));
                    const int m10 = max(max(x[2] + b[1], x[3] + b[5]), max(x[5] + b[2], x[4] + b[6]));
                    const int m11 = max(max(x[1] + b[0], x[0] + b[4]), max(x[6] + b[3], x[7] + b[7]));
                    const int llr = max(m00 + g00, m01 + g01) - max(m10 + g10, m11);
                    const int e   = ((llr - lsa[r]) * 3) >> 2;
                    ev[r] = min(max(e, -BCJR_LE_MAX), BCJR_LE_MAX);
                    pv[r] = min(max(llr, -32767), 32767);
                    beta_step(b, g00, g01, g10);
                }
                norm8(b);
                const size_t off = g16(tile, Kp, lane, (t0 >> 3) + w);
                *reinterpret_cast<uint4 *>(pe + off) = make_uint4(pk16(ev[0], ev[1]), pk16(ev[2], ev[3]), pk16(ev[4], ev[5]), pk16(ev[6], ev[7]));
                if (POST)
                    *reinterpret_cast<uint4 *>(po + off) = make_uint4(pk16(pv[0], pv[1]), pk16(pv[2], pv[3]), pk16(pv[4], pv[5]), pk16(pv[6], pv[7]));
            }
        }
    }
    if (seg > 0) // beta at this segment's start = the previous segment's end, for its next iteration (already normalised)
        *reinterpret_cast<uint4 *>(bnd.b_wr + (((size_t)(seg - 1) * bnd.n_tiles + tile) * 64 + lane) * 8) =
            make_uint4(pk16(b[0], b[1]), pk16(b[2], b[3]), pk16(b[4], b[5]), pk16(b[6], b[7]));
}

// ------------------------------------------------------------------------------------------------
// extrinsic exchange: A[i] = E[tab[i]] (tab = pi towards decoder 2, inv -- with 0xFFFF holes reading 0 -- back
// towards decoder 1).  FINAL: instead of A, the hard decisions c[j] = post[inv[j]] < 0 (a hole falls back on S1[j]).
template <bool FINAL>
__global__ __launch_bounds__(384) void k_bcjr_perm(const int16_t *__restrict__ src, const uint16_t *__restrict__ tab, uint32_t K,
                                                   uint32_t n_cb, int16_t *__restrict__ A, const int8_t *__restrict__ S1,
                                                   uint8_t *__restrict__ c_bits)
{
    extern __shared__ __attribute__((aligned(16))) int16_t sm16[]; // src[Kp]
    const uint32_t cb = xcd_cb(blockIdx.x, n_cb), tile = cb >> 6, lane = cb & 63, Kp = kpad64(K), n_units = Kp >> 4, u = threadIdx.x;
    if (cb >= n_cb) return;
    const size_t off0 = g16(tile, Kp, lane, 2 * u), off1 = g16(tile, Kp, lane, 2 * u + 1); // byte offsets of the unit's two granules
    const int    nv   = (u < n_units) ? min(16, max(0, (int)K - 16 * (int)u)) : -1;
    if (nv >= 0) {
        const char *sb8 = reinterpret_cast<const char *>(src);
        reinterpret_cast<uint4 *>(sm16 + 16 * u)[0] = *reinterpret_cast<const uint4 *>(sb8 + off0);
        reinterpret_cast<uint4 *>(sm16 + 16 * u)[1] = *reinterpret_cast<const uint4 *>(sb8 + off1);
    }
    __syncthreads();
    if (nv <= 0) {
        if (nv == 0 && !FINAL) {
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(A) + off0) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(A) + off1) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const uint4 *p = reinterpret_cast<const uint4 *>(tab + 16 * (size_t)u);
    const uint4  lo = p[0], hi = (nv > 8) ? p[1] : make_uint4(~0u, ~0u, ~0u, ~0u);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    int v[16];
    bool hole[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t idx = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        hole[k] = idx == 0xFFFFu;
        const int t = sm16[hole[k] ? 0u : idx]; // unconditional read, masked
        v[k] = hole[k] ? 0 : t;
    }
    if (!FINAL) {
        char *ab = reinterpret_cast<char *>(A);
        *reinterpret_cast<uint4 *>(ab + off0) = make_uint4(pk16(v[0], v[1]), pk16(v[2], v[3]), pk16(v[4], v[5]), pk16(v[6], v[7]));
        *reinterpret_cast<uint4 *>(ab + off1) = make_uint4(pk16(v[8], v[9]), pk16(v[10], v[11]), pk16(v[12], v[13]), pk16(v[14], v[15]));
    } else {
        const uint4    s4 = *reinterpret_cast<const uint4 *>(S1 + g8(tile, Kp, lane, u));
        const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w};
        uint32_t ob[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int l = hole[k] ? sb(sw[k >> 2], k & 3) : v[k];
            ob[k >> 2] |= (l < 0 ? 1u : 0u) << (8 * (k & 3));
        }
        uint2 *o = reinterpret_cast<uint2 *>(c_bits + (size_t)cb * K + 16 * u); // 8-byte aligned (K % 8 == 0)
        o[0] = make_uint2(ob[0], ob[1]);
        if (nv > 8) o[1] = make_uint2(ob[2], ob[3]);
    }
}

} // namespace

extern "C" size_t mi_lte_turbo_bcjr_scratch_bytes(uint32_t K, uint32_t n_cb)
{
    const size_t n_tiles = (n_cb + 63) / 64, Kp = kpad64(K);
    return n_tiles * (Kp * 64 * 4 + Kp * 128 * 3 + Kp * 128 + 64 * 16 + 2 * 2 * 2 * 4 * 64 * 16); // + boundary states [dec][a|b][buf][seg]
}

// n_iter full iterations of max-log-MAP over n_cb code blocks of size K; int8 soft input in the reference's layout
int mi_turbo_bcjr_batch(mi_lte_ctx *ctx, const int8_t *d_soft, uint32_t K, uint32_t n_cb, uint32_t n_iter, int qpp_spec, uint8_t *d_c_bits)
{
    if (n_iter == 0 || n_iter > 64) return MI_LTE_ERR_INVALID_ARG;
    TurboTables tb;
    int         rc = mi_ctx_turbo_tables(ctx, K, qpp_spec ? 1 : 0, &tb);
    if (rc != MI_LTE_OK) return rc;
    const size_t n_tiles = (n_cb + 63) / 64, Kp = kpad64(K), a8 = n_tiles * Kp * 64, a16 = n_tiles * Kp * 128;
    rc = mi_ctx_reserve_scratch(ctx, mi_lte_turbo_bcjr_scratch_bytes(K, n_cb));
    if (rc != MI_LTE_OK) return rc;
    uint8_t *base = (uint8_t *)ctx->scratch;
    BcjrBufs B;
    B.S1 = (int8_t *)base; B.P1 = B.S1 + a8; B.S2 = B.P1 + a8; B.P2 = B.S2 + a8;
    B.A  = (int16_t *)(base + 4 * a8); B.E = (int16_t *)(base + 4 * a8 + a16); B.post = (int16_t *)(base + 4 * a8 + 2 * a16);
    B.chk  = (int16_t *)(base + 4 * a8 + 3 * a16);
    B.tail = (int8_t *)(base + 4 * a8 + 4 * a16);
    // segment boundary states: [decoder][alpha|beta][buffer][segment][tile][lane][8 x int16], uniform (0) before the first iteration
    const uint32_t n_seg   = bcjr_n_seg(K);
    const size_t   bnd_one = (size_t)4 * n_tiles * 64 * 8; // int16 elements of one [segment][tile][lane][8] array
    int16_t       *bnd0    = (int16_t *)(base + 4 * a8 + 4 * a16 + n_tiles * 64 * 16);
    MI_HIP_CHECK(ctx, hipMemsetAsync(bnd0, 0, 8 * bnd_one * sizeof(int16_t), ctx->stream));
    auto bnd = [&](int dec, uint32_t it) {
        const uint32_t rd = it & 1u, wr = rd ^ 1u;
        int16_t *d = bnd0 + (size_t)dec * 4 * bnd_one;
        return BcjrBnd{d + rd * bnd_one, d + wr * bnd_one, d + (2 + rd) * bnd_one, d + (2 + wr) * bnd_one, (uint32_t)n_tiles};
    };
    if (n_cb % 64) MI_HIP_CHECK(ctx, hipMemsetAsync(base, 0, 4 * a8 + 3 * a16, ctx->stream)); // lanes past the batch end stay defined
    const uint32_t cb_threads = (uint32_t)(((Kp >> 4) + 63) & ~(size_t)63);
    MI_LAUNCH(ctx, "k_bcjr_prep", k_bcjr_prep, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), Kp, d_soft, K, n_cb, tb.d_pi, B);
    for (uint32_t it = 0; it < n_iter; it++) {
        const bool last = it + 1 == n_iter;
        MI_LAUNCH(ctx, "k_bcjr_fwd", k_bcjr_fwd, dim3(n_tiles, n_seg), dim3(64), 0, B.S1, B.P1, B.A, B.chk, K, bnd(0, it));
        MI_LAUNCH(ctx, "k_bcjr_bwd", k_bcjr_bwd<false>, dim3(n_tiles, n_seg), dim3(64), 0, B.S1, B.P1, B.A, B.chk, B.tail, 0u, B.E, B.post, K, n_cb, bnd(0, it));
        MI_LAUNCH(ctx, "k_bcjr_perm", k_bcjr_perm<false>, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), 2 * Kp, B.E, tb.d_pi, K, n_cb, B.A, B.S1, d_c_bits);
        MI_LAUNCH(ctx, "k_bcjr_fwd", k_bcjr_fwd, dim3(n_tiles, n_seg), dim3(64), 0, B.S2, B.P2, B.A, B.chk, K, bnd(1, it));
        if (!last) {
            MI_LAUNCH(ctx, "k_bcjr_bwd", k_bcjr_bwd<false>, dim3(n_tiles, n_seg), dim3(64), 0, B.S2, B.P2, B.A, B.chk, B.tail, 6u, B.E, B.post, K, n_cb, bnd(1, it));
            MI_LAUNCH(ctx, "k_bcjr_perm", k_bcjr_perm<false>, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), 2 * Kp, B.E, tb.d_inv, K, n_cb, B.A, B.S1, d_c_bits);
        } else {
            MI_LAUNCH(ctx, "k_bcjr_bwd", k_bcjr_bwd<true>, dim3(n_tiles, n_seg), dim3(64), 0, B.S2, B.P2, B.A, B.chk, B.tail, 6u, B.E, B.post, K, n_cb, bnd(1, it));
            MI_LAUNCH(ctx, "k_bcjr_perm", k_bcjr_perm<true>, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), 2 * Kp, B.post, tb.d_inv, K, n_cb, B.A, B.S1, d_c_bits);
        }
    }
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_bcjr_prep:1,k_bcjr_fwd,k_bcjr_bwd,k_bcjr_perm: 2 each per iteration";
    return MI_LTE_OK;
}
