// Turbo decoding on gfx950 (MI355X) -- hand-written HIP, no MFMA (nothing here is a dense contraction).
//
// REF mode restates the reference's turbo_decode() Steps 0-14 (liblte/src/liblte_phy.cc:10620-10845)
// bit-exactly.  The algorithm is a strictly serial 8-state trellis walk per code block (the survivor
// is chosen on a hard metric while a weighted metric is accumulated, so there is no associative
// form to scan), therefore the parallel axes are: code blocks, and the elementwise / gather steps.
//
// Mapping ("lock-step tiles"):
//   * 64 code blocks of equal K form a TILE; one wavefront walks the 64 trellises in lock-step,
//     lane = code block, the 8 path metrics of a block live in that lane's VGPRs (no cross-lane
//     traffic, every data-dependent branch of the reference becomes a select).
//   * all per-step arrays of a tile are stored "line per block": element (lane, step t) sits at
//     byte (t/64)*4096 + lane*64 + (t%64), so a lane streams its own 64-byte cache line per 64
//     trellis steps (4 x global_load_dwordx4) and a wave-wide access is one contiguous 4 KiB burst.
//   * traceback needs, per step, which of each predecessor pair had the larger stored metric:
//     4 bits per step per block, packed 8 steps to a dword, streamed to HBM (3 KiB/pass at K=6144).
//   * the steps that are parallel over the trellis index (quantise, interleave, soft re-encode,
//     vote) run as one workgroup per code block with the block staged in LDS.
//
// Kernel sequence per batch:  prep -> siso(pass 1) -> perm -> siso(pass 2 | pass 3) -> vote
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr uint32_t RX_NULL_AS_INT = 10000; // RX_NULL_BIT, liblte_phy.cc:1620

__host__ __device__ inline uint32_t kpad64(uint32_t K) { return (K + 63u) & ~63u; }

// ------------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ int sbyte(uint32_t w, int k) { return (int)__builtin_amdgcn_sbfe(w, 8 * k, 8); }

// Reads at absolute LDS byte addresses.  The compiler adds the start of the dynamic LDS block to every access as a late-resolved
// constant (a v_add per gathered byte); the per-code-block kernels declare no static LDS, so their dynamic block starts at address 0
// (checked on the host before the first launch, no_static_lds) and "block + constant + index" is the index register plus an
// immediate offset.
typedef __attribute__((address_space(3))) const uint8_t  lds_u8_t;
typedef __attribute__((address_space(3))) const int8_t   lds_i8_t;
typedef __attribute__((address_space(3))) const uint16_t lds_u16_t;
__device__ __forceinline__ uint32_t lds_u8(uint32_t addr) { return *(lds_u8_t *)(uintptr_t)addr; }
__device__ __forceinline__ int      lds_i8(uint32_t addr) { return *(lds_i8_t *)(uintptr_t)addr; }
__device__ __forceinline__ uint32_t lds_u16(uint32_t addr) { return *(lds_u16_t *)(uintptr_t)addr; }

// pairs of int16 in one register (v_pk_*_i16), used by the per-code-block kernels and the SISO kernel below
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s      as_v2s(uint32_t w) { return __builtin_bit_cast(v2s, w); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s      even2(uint32_t w) { return (as_v2s(w) << 8) >> 8; }
__device__ __forceinline__ v2s      odd2(uint32_t w) { return as_v2s(w) >> 8; }
__device__ __forceinline__ uint32_t merge_bytes(v2s e, v2s o) { return __builtin_amdgcn_perm(as_u32(o), as_u32(e), 0x06020400u); } // low bytes of (e.lo, o.lo, e.hi, o.hi)
__device__ __forceinline__ v2s      abs2(v2s a) { return __builtin_elementwise_max(a, (v2s)(0) - a); }
// 0xFFFF per negative half.  Opaque on purpose: from a visible "x >> 15" feeding an and/or select the compiler rebuilds per-half
// compares and selects, which costs twice the instructions of the mask + one bit-select it replaces.
__device__ __forceinline__ uint32_t neg_mask(v2s d)
{
    uint32_t m;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(as_u32(d))); // the inline constant has no upper half
    return m;
}
__device__ __forceinline__ v2s bit_select(uint32_t m, v2s yes, v2s no) { return as_v2s((as_u32(yes) & m) | (as_u32(no) & ~m)); }
// (a + b) >> 1 per half, for sums that fit their 16 bits (magnitudes <= 127 here).  The sum is made opaque: the compiler otherwise
// recognises the "average" idiom and, having no such instruction, expands it into the overflow-safe (a & b) + ((a ^ b) >> 1) -- four
// instructions (some of them split per half) where these two do.
__device__ __forceinline__ v2s half_sum(v2s a, v2s b)
{
    uint32_t t = as_u32(a + b);
    asm("" : "+v"(t));
    return as_v2s(t) >> 1;
}

// sign * ((|a|+|b|) >> 1), sign negative iff exactly one operand is negative (0 counts as positive).
// This one form covers the four branches of Step 3 (liblte_phy.cc:10688-10707) and the g=03 soft
// re-encoder conv_encode_soft (liblte_phy.cc:10123-10147).
__device__ __forceinline__ int soft_xor(int a, int b)
{
    int mag = (abs(a) + abs(b)) >> 1;
    return ((a < 0) != (b < 0)) ? -mag : mag;
}


// Wavefront-wide reductions on the vector ALU's own cross-lane paths (DPP within a row of 16 lanes, v_readlane across the four rows) --
// __shfl_xor is a ds_bpermute, i.e. an instruction of the LDS pipe, and that pipe is the busiest unit of the per-code-block kernels
// (72 % of k_turbo_prep's time, SQ_ACTIVE_INST_LDS): three block-wide maxima there were eighteen of its ~110 LDS instructions per wavefront.
// After the four steps every lane of a row holds the row's result; the four rows are combined in scalar registers (the result is uniform).
template <typename Op> __device__ __forceinline__ int wave_reduce_i(int v, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]: lane ^ 1
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]: lane ^ 2
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false)); // row_half_mirror: the other quad of the eight
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false)); // row_mirror: the other eight of the row
    return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int      wave_max_i(int v) { return wave_reduce_i(v, [](int a, int b) { return max(a, b); }); }
__device__ __forceinline__ uint32_t wave_xor_u(uint32_t v) { return (uint32_t)wave_reduce_i((int)v, [](int a, int b) { return a ^ b; }); }
__device__ __forceinline__ float    wave_max_f(float v) // (non-negative inputs: |x| maxima -- the integer order of their bit patterns is the float order)
{
    return __int_as_float(wave_max_i(__float_as_int(v)));
}

// (the same precondition: non-negative inputs -- maxima of fabsf values; a negative value would compare as a large unsigned pattern)
__device__ __forceinline__ float block_max_abs_f(float v, float *red /* >= 8 floats */)
{
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (uint32_t w = 1; w < (blockDim.x >> 6); w++) r = fmaxf(r, red[w]);
    return r;
}
__device__ __forceinline__ int block_max_i(int v, int *red)
{
    v = wave_max_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = red[0];
    for (uint32_t w = 1; w < (blockDim.x >> 6); w++) r = max(r, red[w]);
    return r;
}

// two maxima in one round (red: >= 16 ints)
__device__ __forceinline__ void block_max_i2(int &a, int &b, int *red)
{
    a = wave_max_i(a); b = wave_max_i(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[8 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = red[0]; b = red[8];
    for (uint32_t w = 1; w < (blockDim.x >> 6); w++) { a = max(a, red[w]); b = max(b, red[8 + w]); }
}

struct PrepOut { uint8_t *arr[6]; }; // X0 X1 X2 I0 M1 M2

// XCD-aware block -> code block mapping for the one-workgroup-per-code-block kernels.  Workgroup b runs
// on XCD b % 8 (observed dispatch order; used for speed only), and a tile line is 64 bytes per code
// block inside a 4 KiB burst shared by 64 code blocks: giving each XCD a contiguous range of tiles keeps
// all writers (readers) of a burst behind one L2, so lines leave for HBM whole instead of in halves.
__host__ __device__ inline uint32_t xcd_chunk(uint32_t n_cb) { return ((((n_cb + 63u) >> 6) + 7u) >> 3) << 6; } // code blocks per XCD
__device__ __forceinline__ uint32_t xcd_cb(uint32_t b, uint32_t n_cb) { return (b & 7u) * xcd_chunk(n_cb) + (b >> 3); }

// ------------------------------------------------------------------------------------------------
// prep: Step 0 (NULL -> 0), Step 1 scaling to int8, Step 4 (interleave d0), SISO output magnitudes
// for passes 1 and 2.  One workgroup per code block.
//   q(x) = (int8)(x*127/max|x|)            liblte_phy.cc:10645-10664 (max over the K triples only)
//   M[t] = (int8)(127*(w_t/W)), w_t = |in[2t]|+|in[2t+1]|, W = max_t w_t   liblte_phy.cc:10449,10498-10524
//   (the branch weight is the same for every state, so the reference's path-dependent max_weight
//    reduces to max_t w_t; the sign is applied by the traceback)
// ---- "unit owner" mapping of the per-code-block kernels (prep, perm, vote):
// a unit is 16 consecutive trellis indices = one 16-byte quarter of a tile line; thread t owns units
// t, t+256 (NSLOT = 2 only for K > 4096), keeps them in registers across the phases and exchanges
// only what is really permuted (q(d0), C1, D1/D2) through LDS.  Little LDS -> many resident waves.
__device__ __forceinline__ uint4 pack16(const int (&q)[16])
{
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        w[k] = (uint32_t)(q[4 * k] & 0xFF) | ((uint32_t)(q[4 * k + 1] & 0xFF) << 8) | ((uint32_t)(q[4 * k + 2] & 0xFF) << 16) |
               ((uint32_t)(q[4 * k + 3] & 0xFF) << 24);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ size_t unit_off(size_t tile_off, uint32_t lane, uint32_t u)
{
    return tile_off + (size_t)(u >> 2) * 4096 + lane * 64 + (u & 3) * 16;
}
// 16 uint16 table entries starting at index 16u (nvalid = 16 or 8); the entries past nvalid read `pad`
// (callers pass the index of a slot that holds 0, so that nothing downstream needs a per-element predicate
// -- predicated LDS reads compile to one branch + one waitcnt each)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const u32x4_t const_u32x4_t; // sixteen bytes of a table that no kernel writes, read through the constant address space
__device__ __forceinline__ void load_idx16_at(const_u32x4_t *p /* the unit's first sixteen bytes */, int nvalid, uint32_t (&idx)[16], uint32_t pad)
{
    const uint32_t pp = pad | (pad << 16);
    const u32x4_t  lo = p[0], hi = (nvalid > 8) ? p[1] : (u32x4_t)(pp);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int k = 0; k < 16; k++) idx[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
}
__device__ __forceinline__ void load_idx16(const uint16_t *tab, uint32_t u, int nvalid, uint32_t (&idx)[16], uint32_t pad = 0)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(tab + 16 * (size_t)u);
    const uint32_t pp = pad | (pad << 16);
    const uint4  lo = p[0], hi = (nvalid > 8) ? p[1] : make_uint4(pp, pp, pp, pp);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int k = 0; k < 16; k++) idx[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
}

// sixteen table entries as fetched, for kernels that request them long before they unpack them
struct IdxRaw { uint4 lo, hi; };
__device__ __forceinline__ void unpack_idx16(const IdxRaw &r, uint32_t (&idx)[16])
{
    const uint32_t w[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
#pragma unroll
    for (int k = 0; k < 16; k++) idx[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
}

struct KSeg; // (a block size's row of a merged decode's table, below)
typedef __attribute__((address_space(4))) const KSeg const_seg_fwd_t;
// ---- where the soft values of a code block come from
// (a) directly from the caller, in the reference's interleaved d[i*3+x] layout
template <typename T> struct SrcDirect {
    static constexpr bool kIntegerValued = sizeof(T) != 4; // int8 / int16 soft values
    static constexpr bool kIntPath = false;
    static constexpr bool kPacked = false;
    uint32_t e_cap;
    const T *soft;
    const T *d;
    __device__ __forceinline__ void init(uint32_t cb, uint32_t K) { d = soft + (size_t)cb * 3 * (K + 4); }
    __device__ __forceinline__ void seg(const_seg_fwd_t &) {} // (a merged decode always comes from rate-matched soft bits)
    __device__ __forceinline__ bool hard_inputs() const { return false; }
    __device__ __forceinline__ bool stage_e(int8_t *) { return false; }
    // v[Src::kPacked ? 0 : x][k] = d[(16u+k)*3 + x] for k < nvalid (16 or 8), with Step 0 (RX_NULL_BIT -> 0, liblte_phy.cc:10636-10642)
    __device__ __forceinline__ void load16(uint32_t u, int nvalid, float (&v)[3][16], uint32_t, bool) const
    {
        const T *p = d + (size_t)u * 48;
        auto put = [&](int e, float t) { // element e = 3*k + x of the unit
            const int k = e / 3, x = e - 3 * k;
            v[x][k]     = (k < nvalid && t != (float)RX_NULL_AS_INT) ? t : 0.0f;
        };
        if (sizeof(T) == 1) { // 48 (24) bytes, 4-byte aligned
            const uint32_t *g = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
            for (int w = 0; w < 12; w++) {
                const uint32_t x = (w < 6 || nvalid > 8) ? g[w] : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) put(4 * w + k, (float)sbyte(x, k));
            }
        } else if (sizeof(T) == 2) { // 96 (48) bytes, 16-byte aligned
            const uint4 *g = reinterpret_cast<const uint4 *>(p);
#pragma unroll
            for (int w = 0; w < 6; w++) {
                const uint4    x = (w < 3 || nvalid > 8) ? g[w] : make_uint4(0, 0, 0, 0);
                const uint32_t c[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int k = 0; k < 8; k++) put(8 * w + k, (float)(int)__builtin_amdgcn_sbfe(c[k >> 1], 16 * (k & 1), 16));
            }
        } else { // 192 (96) bytes, 16-byte aligned
            const float4 *g = reinterpret_cast<const float4 *>(p);
#pragma unroll
            for (int w = 0; w < 12; w++) {
                const float4 x = (w < 6 || nvalid > 8) ? g[w] : make_float4(0, 0, 0, 0);
                put(4 * w, x.x); put(4 * w + 1, x.y); put(4 * w + 2, x.z); put(4 * w + 3, x.w);
            }
        }
    }
};

// (b) from rate-matched, descrambled soft bits e[0..E) of a PDSCH allocation: turbo rate
// un-matching (liblte_phy_rate_unmatch_turbo, liblte_phy.cc:11246-11490) fused in as a gather.
// The circular buffer w[0..3*K_pi) holds, column-major, the sub-block-interleaved streams; the only
// NULL positions the reference's receiver honours are the N_d = 32R - D head-padding slots of each
// stream.  Walking w from k0 and consuming one e per non-NULL position means position p
// receives e[rank(p) + t*Nnn], t = 0,1,.. (repeats are soft-combined by addition, :11402-11416), with
// rank(p) = number of non-NULL positions between k0 and p in walk order.  Head padding sits in row 0,
// so "NULLs below p" is a popcount over the 32 columns.
struct RmGeom {
    uint32_t R, K_pi, N_d, N_cb, k0m, Nnn, cnt_k0, mask0, mask2;
    __device__ __forceinline__ static uint32_t lowmask(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }
    __device__ __forceinline__ uint32_t nulls_below(uint32_t p) const
    {
        if (p <= K_pi) return __popc(mask0 & lowmask((p + R - 1) / R));
        const uint32_t q = p - K_pi, a = (q + 1) >> 1, b = q >> 1;
        return __popc(mask0) + __popc(mask0 & lowmask((a + R - 1) / R)) + __popc(mask2 & lowmask((b + R - 1) / R)) +
               ((b > K_pi - 1) ? 1u : 0u);
    }
    __device__ __forceinline__ uint32_t cnt(uint32_t p) const { return p - nulls_below(p); }
    // geometry per liblte_phy.cc:11283-11287 (R), :11371-11399 (K_w, N_ir, N_cb, k0) with the constants
    // liblte_phy_pdsch_channel_decode passes (M_dl_harq = 8, N_soft = 250368, :3843-3844), C = 1
    __device__ __forceinline__ void init(uint32_t D, uint32_t tx_mode, uint32_t rv, uint32_t N_soft = 250368,
                                         uint32_t M_dl_harq = 8, uint32_t C = 1, bool limited = true)
    {
        R    = (D + 31) / 32;
        K_pi = 32 * R;
        N_d  = K_pi - D;
        const uint32_t K_w = 3 * K_pi, K_mimo = (tx_mode == 3 || tx_mode == 4 || tx_mode == 8 || tx_mode == 9) ? 2 : 1;
        const uint32_t N_ir = N_soft / (K_mimo * (M_dl_harq < 8 ? M_dl_harq : 8));
        N_cb = (limited && N_ir / C < K_w) ? N_ir / C : K_w; // DLSCH/PCH only (liblte_phy.cc:11387-11398)
        const uint32_t k0 = R * (2 * ((N_cb + 8 * R - 1) / (8 * R)) * rv + 2);
        k0m  = k0 % N_cb;
        mask0 = mask2 = 0;
        for (uint32_t c = 0; c < 32; c++) {
            const uint32_t P = __brev(c) >> 27; // inter-column permutation = 5-bit reversal (36.212 table 5.1.4-1)
            if (P < N_d) mask0 |= 1u << c;
            if (P + 1 < N_d) mask2 |= 1u << c;
        }
        Nnn    = cnt(N_cb);
        cnt_k0 = cnt(k0m);
    }
    // circular-buffer position of d[i*3+x]
    __device__ __forceinline__ uint32_t pos(uint32_t i, int x) const
    {
        const uint32_t n = i + N_d;
        if (x == 2) {
            const uint32_t m = n - 1, ii = (__brev(m & 31) >> 27) * R + (m >> 5);
            return K_pi + 2 * ii + 1;
        }
        const uint32_t ii = (__brev(n & 31) >> 27) * R + (n >> 5);
        return x == 0 ? ii : K_pi + 2 * ii;
    }
    // the same position together with its non-NULL count below it, without integer division: the
    // element sits in column c, row r of its stream's R x 32 matrix, and only row 0 holds NULLs, so
    // "NULL slots below" is a popcount over the columns up to c (+1 if r > 0)
    __device__ __forceinline__ void pos_cnt(uint32_t i, int x, uint32_t &p, uint32_t &cn) const
    {
        const uint32_t n = i + N_d - (x == 2 ? 1u : 0u), c = __brev(n & 31) >> 27, r = n >> 5, ii = c * R + r;
        const uint32_t cc = c + (r > 0 ? 1u : 0u);
        uint32_t nulls;
        if (x == 0) {
            p     = ii;
            nulls = __popc(mask0 & lowmask(cc));
        } else if (x == 1) {
            p     = K_pi + 2 * ii;
            nulls = __popc(mask0) + __popc(mask0 & lowmask(cc)) + __popc(mask2 & lowmask(cc));
        } else {
            p     = K_pi + 2 * ii + 1;
            nulls = __popc(mask0) + __popc(mask0 & lowmask(c + 1)) + __popc(mask2 & lowmask(cc));
        }
        cn = p - nulls;
    }
};

// Everything the per-code-block kernels need to know about their block, gathered once per decode by k_cb_desc: read through the
// allocation tables it is a chain of three dependent L2 round trips at the start of every workgroup (slot -> allocation -> its fields ->
// its soft bits), which was a third of a k_turbo_prep workgroup's lifetime
struct CbDesc { uint32_t alloc, e_off, E, combo, Nnn, hard, tbs, pad; }; // 32 bytes

struct GroupDesc {                 // one launch = the code blocks of one size K out of a PDSCH batch
    const mi_lte_pdsch_alloc *allocs;
    const uint32_t *cb_alloc;      // [n_cb] allocation index of each code-block slot
    const int8_t   *e_base;        // descrambled soft bits of all allocations
    const uint32_t *e_off;         // [n_alloc] offset of an allocation's soft bits in 64-byte units
    const uint32_t *e_len;         // [n_alloc] E (written by the demodulation kernel)
    uint8_t        *out_bits;      // [n_alloc][out_stride] decoded transport block, one bit per byte
    uint32_t        out_stride;
    int32_t        *status;        // [n_alloc] LIBLTE_ERROR_ENUM value
    const uint32_t *crc_tab;       // (x^e mod gCRC24A) << 8 at index MI_CRC_TAB_BIAS + e (ctx.cc)
    uint32_t        ul;            // 1: UL-SCH soft-buffer rule (N_cb = K_w: chan_type ULSCH, liblte_phy.cc:11387-11398, :12437-12449)
    uint32_t        packed;        // 1: out_bits holds eight bits per byte, first bit in the most significant position (liblte_value_2_bits order)
    const CbDesc   *desc;          // [n_cb] filled by k_cb_desc before the first kernel of the group (REF decoder)
};

// One block size of a MERGED decode (mi_turbo_ref_multi: a PDSCH batch whose allocations have many code-block sizes -- a cell's TTIs
// have dozens of the 188).  Launched size by size such a batch is ~7 launches per size in series, each far too small for the device (a
// trellis walk is as long for one tile as for a thousand); merged, every kernel below is launched ONCE over the tiles / code blocks of
// all sizes (the per-code-block kernels once per workgroup width, 64 .. 384 threads), and what the per-size launches pass as kernel
// arguments -- K, the block count, the interleaver and rank tables, where the size's tiles lie in the scratch arrays -- is read from
// this table by the workgroup (wavefront) at its start: one scalar load of its index in `map`, then scalar loads of the row.
struct KSeg {
    uint32_t        K, n_cb, cb_base, n_tiles; // cb_base: the size's first slot in the batch's code-block order (cb_alloc, desc)
    uint32_t        wg_cb;                     // first workgroup of the size in its prep / vote launch (a multiple of 8: blockIdx % 8 = XCD)
    uint32_t        wg_perm, perm_grid;        // the same for perm, and the size's own grid there (a workgroup takes PERM_NB blocks grid apart)
    uint32_t        e_cap;                     // LDS bytes prep stages an allocation's soft bits in (0: gathers from global memory)
    uint32_t        wv1, wv23;                 // first wavefront of the size in SISO pass 1 / passes 2 + 3
    uint64_t        arr_off;                   // where the size's tiles start in each of the eleven byte arrays (traceback words: half of it)
    // mi_ctx_turbo_tables / rm_rank_tables of K.  Typed as GLOBAL pointers: a pointer that comes out of memory is otherwise of unknown address
    // space and every load through it a flat_load, which counts against the LDS counter too -- k_turbo_prep's LDS gathers then waited for its
    // table loads (W4 as a merged decode: 4.88 ms, 4.53 without them)
    __attribute__((address_space(1))) const uint16_t *pi, *inv2, *tabs;
    __attribute__((address_space(1))) const uint32_t *nnn;
    uint32_t        ws1, ws23;                 // first workgroup of the size in the state-parallel trellis kernel's launches (a handful of blocks: k_turbo_siso_small)
    uint64_t        pad;
};
static_assert(sizeof(KSeg) == 96, "KSeg is read with scalar loads: keep it a multiple of 16 bytes");
struct MultiArgs { const KSeg *segs; const uint32_t *map; }; // map: per 512 workgroups (prep, vote: a size's grid is a multiple of that), per 128 (perm), per wavefront (siso) the index of its size
// Both reads go through the CONSTANT address space: only then are they scalar loads (the kernels store to global memory, and a plain pointer
// carries no promise that the table is not what they store to); the scalar cache serves the row every workgroup of a CU reads.
// What remains: on W4 (two sizes of 8192 and 1024 tiles, each filling the device by itself) the merged k_turbo_prep takes 4.3-4.5 ms
// against 4.1 for the per-size kernel ON THE SAME merged layout (profiles/r06_variants_merged_prep.txt: the per-size kernel launched through
// this path, everything else merged, is the fastest W4 of all, 23.6-23.8 ms) with identical registers, LDS, grid and all but identical listings:
// what is left is the workgroup's start -- two more dependent scalar loads (map entry, row) before its first own load can go out, in a
// kernel whose workgroups live ~14 us.  The maps are coarse (one entry per 512 / 128 workgroups: a few KB, scalar-cache resident) for that
// reason.  A size that fills the device by itself keeps its own launches (chain.hip); the merged path is for many sizes of a few tiles each.
typedef __attribute__((address_space(4))) const KSeg     const_seg_t;
typedef __attribute__((address_space(4))) const uint32_t const_map_t;
__device__ __forceinline__ const_seg_t &multi_seg(const MultiArgs &ma, uint32_t slot)
{
    const uint32_t idx = reinterpret_cast<const_map_t *>(reinterpret_cast<uintptr_t>(ma.map))[__builtin_amdgcn_readfirstlane(slot)];
    return reinterpret_cast<const_seg_t *>(reinterpret_cast<uintptr_t>(ma.segs))[idx];
}


struct SrcRateUnmatch {
    static constexpr bool kIntegerValued = true; // sums of int8 soft bits
    uint32_t e_cap; // bytes of LDS available for staging e (0 = gather from global)
    GroupDesc       g;
    const uint16_t *tabs; // [12][3K] rank of every d element in the order e is consumed: DL per (rv, K_mimo), then UL per rv
    const uint32_t *nnn;  // [12]     non-NULL slots per lap of the circular buffer
    const uint16_t *tab;
    const int8_t   *e;
    uint32_t        E, Nnn, K_;
    __device__ __forceinline__ void seg(const_seg_fwd_t &sg); // merged launch: the size's tables and descriptors (defined below KSeg)
    __device__ __forceinline__ void init(uint32_t cb, uint32_t K)
    {
        // (through the constant address space: written by the launch before this one, uniform -> scalar loads, also where the pointer itself
        // came out of a merged decode's table and the compiler no longer sees a kernel argument behind it)
        const_u32x4_t *dp = reinterpret_cast<const_u32x4_t *>(reinterpret_cast<uintptr_t>(g.desc + cb));
        const u32x4_t  d0 = dp[0], d1 = dp[1]; // alloc e_off E combo | Nnn hard tbs -
        tab = tabs + (size_t)d0.w * 3 * K;
        Nnn = d1.x;
        K_  = K;
        e   = g.e_base + (size_t)d0.y * 64;
        E   = d0.z;
        hard = d1.y != 0;
    }
    bool hard;
    __device__ __forceinline__ bool hard_inputs() const { return hard; }
    // copy the allocation's soft bits into LDS with wide loads (the gather is otherwise a chain of
    // dependent byte loads from L2), followed by zeros up to the next 16-byte boundary and at least at index E: the
    // gather below reads "nothing" (a NULL slot, a rank the allocation does not reach) as e[min(rank, E)] = 0.
    // Returns whether the copy was made.
    __device__ __forceinline__ bool stage_e(int8_t *lds)
    {
        if (E + 16 > e_cap || E >= 0xFFFFu) return false;
        const uint32_t nq = E >> 4, tail = E & 15u; // e_off is 64-byte aligned and padded
        const uint4   *gp = reinterpret_cast<const uint4 *>(e);
        uint4         *l  = reinterpret_cast<uint4 *>(lds);
#pragma unroll 4
        for (uint32_t w = threadIdx.x; w < nq; w += blockDim.x) l[w] = gp[w];
        if (threadIdx.x == (nq & 63u)) { // the quarter line that holds index E
            uint4 t = make_uint4(0, 0, 0, 0);
            if (tail) {
                t = gp[nq];
                uint32_t c[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const int keep = (int)tail - 4 * (int)k; // bytes of word k below index E
                    c[k] = keep >= 4 ? c[k] : keep <= 0 ? 0u : (c[k] & ((1u << (8 * keep)) - 1u));
                }
                t = make_uint4(c[0], c[1], c[2], c[3]);
            }
            l[nq] = t;
        }
        __syncthreads();
        return true;
    }
    // the gather from the staged copy: v[Src::kPacked ? 0 : x][k] = sum over laps t of e[min(rank + t*Nnn, E)], ranks 0xFFFF (NULL, or past the
    // block end) included -- three operations per element and lap, no predicate
    // (emit(x, acc) receives the 16 sums of stream x; it is called stream by stream so that a caller that packs them keeps only one
    // stream unpacked at a time)
    // the rank words of a unit's three streams (16 ranks of 16 bits each; 0xFFFF = NULL or past the block end)
    struct RankWords { uint4 raw[3][2]; };
    __device__ __forceinline__ void load_ranks(uint32_t u, int nvalid, RankWords &rw) const
    {
#pragma unroll
        for (int x = 0; x < 3; x++) {
            rw.raw[x][0] = rw.raw[x][1] = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (nvalid > 0) {
                const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)x * K_ + 16 * (size_t)u);
                rw.raw[x][0] = p[0];
                if (nvalid > 8) rw.raw[x][1] = p[1];
            }
        }
    }
    template <typename Emit> __device__ __forceinline__ void gather16_staged(uint32_t el /* LDS address of the staged copy */, uint32_t u, int nvalid, Emit emit) const
    {
        RankWords rw;
        load_ranks(u, nvalid, rw);
        gather16_staged(el, rw, emit);
    }
    template <typename Emit> __device__ __forceinline__ void gather16_staged(uint32_t el, const RankWords &rw, Emit emit) const
    {
        const uint4 (&raw)[3][2] = rw.raw;
#pragma unroll
        for (int x = 0; x < 3; x++) {
            const uint32_t w[8] = {raw[x][0].x, raw[x][0].y, raw[x][0].z, raw[x][0].w, raw[x][1].x, raw[x][1].y, raw[x][1].z, raw[x][1].w};
            uint32_t       r[16];
            int            acc[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                r[k]   = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu);
                acc[k] = lds_i8(el + min(r[k], E));
            }
            if (Nnn < E) { // more than one lap of the circular buffer: repeats are added (soft combining)
                // A lap past the first only reaches the ranks below E - base -- with E just above one lap (the W4 allocations: 9936 soft
                // bits for 9804 positions) that is 1 % of the positions, all in one stream -- so a stream none of whose ranks in this
                // wavefront is reached skips its sixteen reads of the zero slot (wave-uniform test on the smallest rank)
                uint32_t rmin = r[0];
#pragma unroll
                for (int k = 1; k < 16; k++) rmin = min(rmin, r[k]);
                for (uint32_t base = Nnn; base < E; base += Nnn) {
                    if (!__any(rmin + base < E)) break; // the ranks only grow with the lap
#pragma unroll
                    for (int k = 0; k < 16; k++) acc[k] += lds_i8(el + min(r[k] + base, E));
                }
            }
            emit(x, acc);
        }
    }
    // d[(16u+k)*3+x] = sum over laps t of e[rank + t*Nnn] (first visit stores, repeats add, :11402-11416).
    // All first-lap loads of a unit are independent; later laps (uniform trip count) are masked.
    template <typename P, typename Emit> __device__ __forceinline__ void gather16(P ep, uint32_t u, int nvalid, Emit emit) const
    {
        // the rank words of all three streams are requested before the first is used: one L2 round trip instead of three
        RankWords rw;
        load_ranks(u, nvalid, rw); // (a missing upper half reads 0xFFFF: never filled, like the positions past nvalid)
        const uint4 (&raw)[3][2] = rw.raw;
#pragma unroll
        for (int x = 0; x < 3; x++) {
            uint32_t r[16];
            int      acc[16];
            {
                const uint32_t w[8] = {raw[x][0].x, raw[x][0].y, raw[x][0].z, raw[x][0].w, raw[x][1].x, raw[x][1].y, raw[x][1].z, raw[x][1].w};
#pragma unroll
                for (int k = 0; k < 16; k++) r[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) { // 0xFFFF: never filled -> RX_NULL_BIT -> 0 in Step 0
                const bool ok = k < nvalid && r[k] != 0xFFFFu && r[k] < E;
                const int  t  = (int)ep[ok ? r[k] : 0u]; // unconditional load, clamped index: no branch, loads stay in flight together
                acc[k]        = ok ? t : 0;
                if (!ok) r[k] = 0xFFFFFFFFu - 65536u * 4u; // keeps r + t*Nnn >= E without overflowing
            }
            for (uint32_t base = Nnn; base < E; base += Nnn) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t q = r[k] + base;
                    const int      t = (int)ep[q < E ? q : 0u];
                    acc[k] += (q < E) ? t : 0;
                }
            }
            emit(x, acc);
        }
    }
    static constexpr bool kPacked = false;
    static constexpr bool kIntPath = true; // the sums are small integers: the quantiser below never leaves integer arithmetic
    __device__ __forceinline__ void load16(uint32_t u, int nvalid, int (&v)[3][16], uint32_t e_lds, bool in_lds) const
    {
        auto emit = [&](int x, const int (&acc)[16]) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[x][k] = acc[k];
        };
        if (in_lds) gather16_staged(e_lds, u, nvalid, emit);
        else        gather16(e, u, nvalid, emit);
    }
    // the same with the sums kept two per register as int16 (vp[x][j] = values 2j, 2j+1 of stream x): 24 registers instead of 48 across
    // the block-wide maximum, which is what lets the kernel run 8 waves per SIMD.  Valid while the sums fit (SrcRateUnmatchPk below).
    __device__ __forceinline__ void load16_pk(uint32_t u, int nvalid, uint32_t (&vp)[3][8], uint32_t e_lds, bool in_lds) const
    {
        auto emit = [&](int x, const int (&acc)[16]) {
#pragma unroll
            for (int j = 0; j < 8; j++) vp[x][j] = __builtin_amdgcn_perm((uint32_t)acc[2 * j + 1], (uint32_t)acc[2 * j], 0x05040100u);
        };
        if (in_lds) gather16_staged(e_lds, u, nvalid, emit);
        else        gather16(e, u, nvalid, emit);
    }
    // An allocation longer than the LDS block (repetition: several laps of the circular buffer, and a merged decode sizes the block for
    // one lap and a quarter, not for a size's longest allocation -- k_turbo_prep's occupancy is what its LDS leaves): ONE LAP AT A TIME.
    // Lap t's soft bits e[t Nnn .. t Nnn + len) are staged (from the 16-byte line they start in: `shift` bytes of the lap before come along
    // and are never read), rank r of the lap reads index min(r, len) -- a zero at len serves the NULL slots and the ranks the last lap does
    // not reach -- and the sums accumulate as int16 pairs.  Called by EVERY thread of the workgroup (barriers inside).
    __device__ __forceinline__ bool window_ok() const { return Nnn + 48 <= e_cap; }
    __device__ __forceinline__ void gather_windowed_pk(uint32_t u, int nvalid, uint32_t (&vp)[3][8], int8_t *lds, uint32_t el) const
    {
        uint4 *l = reinterpret_cast<uint4 *>(lds);
        for (uint32_t base = 0; base < E; base += Nnn) {
            const uint32_t len = min(Nnn, E - base), shift = base & 15u, end = shift + len, nq = end >> 4, tail = end & 15u;
            const uint4   *gp = reinterpret_cast<const uint4 *>(e + (base - shift)); // (the allocation starts on a 64-byte line and is padded to one)
            RankWords rw;
            load_ranks(u, nvalid, rw); // (per lap: a few L2 hits under the copy, instead of 24 registers across the barriers)
            if (base) __syncthreads(); // the lap before has been read
#pragma unroll 4
            for (uint32_t w = threadIdx.x; w < nq; w += blockDim.x) l[w] = gp[w];
            if (threadIdx.x == (nq & 63u)) { // the quarter line that holds index `end`: zero from there on
                uint4 t = make_uint4(0, 0, 0, 0);
                if (tail) {
                    t = gp[nq];
                    uint32_t c[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const int keep = (int)tail - 4 * (int)k;
                        c[k] = keep >= 4 ? c[k] : keep <= 0 ? 0u : (c[k] & ((1u << (8 * keep)) - 1u));
                    }
                    t = make_uint4(c[0], c[1], c[2], c[3]);
                }
                l[nq] = t;
            }
            __syncthreads();
            if (nvalid > 0) {
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    const uint32_t w[8] = {rw.raw[x][0].x, rw.raw[x][0].y, rw.raw[x][0].z, rw.raw[x][0].w, rw.raw[x][1].x, rw.raw[x][1].y, rw.raw[x][1].z, rw.raw[x][1].w};
                    int acc[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const uint32_t r = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu);
                        acc[k] = lds_i8(el + shift + min(r, len));
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        vp[x][j] = as_u32(as_v2s(vp[x][j]) + as_v2s(__builtin_amdgcn_perm((uint32_t)acc[2 * j + 1], (uint32_t)acc[2 * j], 0x05040100u)));
                }
            }
        }
    }
};
// chosen by the host when no sum can leave int16: ceil(E / Nnn) laps of |e| <= 127 each
__device__ __forceinline__ void SrcRateUnmatch::seg(const_seg_fwd_t &sg)
{
    tabs = (const uint16_t *)sg.tabs; nnn = (const uint32_t *)sg.nnn; g.desc += sg.cb_base;
}
struct SrcRateUnmatchPk : SrcRateUnmatch { static constexpr bool kPacked = true; };

// LDS tables that replace the per-element IEEE divisions: the quantiser and the SISO output magnitude
// are functions of one small integer and a per-block constant, so each distinct value is divided once
// (with exactly the reference's float expression) and every element looks its result up.
constexpr uint32_t QTAB_N = 4096; // q(x) for |x| <= 2047 (signed index x + max on the integer path); larger maxima divide per element
constexpr uint32_t QTAB_HALF = 2048;
constexpr uint32_t MTAB_N = 256;  // w = |a|+|b| <= 254
// k_turbo_prep's LDS: mtab1 | mtab2 | reduction scratch (64 B) | { staged e [e_cap]  OVER  qtab [QTAB_N] | q(d0) [Kp] }.  The soft bits are dead once
// every wavefront has summed its own (the first block-wide maximum is the fence), the quantiser table and q(d0) are written after it: they
// share the bytes.  Before round 6 the four lay side by side -- 9.7 KB for a 64-thread workgroup, 16.7 for a 128-thread one, which held those
// widths at 4 and 4.5 wavefronts per SIMD where the registers allow 6.
constexpr uint32_t PREP_RED_AT = 2 * MTAB_N, PREP_E_AT = PREP_RED_AT + 64, PREP_QTAB_AT = PREP_E_AT, PREP_Q0_AT = PREP_QTAB_AT + QTAB_N;
__host__ __device__ constexpr uint32_t prep_lds_bytes(uint32_t Kp, uint32_t e_cap) { return PREP_E_AT + (e_cap > QTAB_N + Kp ? e_cap : QTAB_N + Kp); }

// byte-parallel helpers for the per-code-block kernels.  Soft values travel as four int8 per register; |a| of one of them is
// the masked byte SAD of the biased word (x + 128 per byte, i.e. word ^ 0x80808080) against 0x80 in that byte alone
// (v_msad_u8 skips the bytes whose reference byte is 0), so |a_k| + |b_k| is two instructions on the packed words.
__device__ __forceinline__ uint32_t pack4u(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) { return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24); } // bytes < 256
__device__ __forceinline__ uint32_t bias4(uint32_t w) { return w ^ 0x80808080u; }
template <int K4> __device__ __forceinline__ uint32_t abs_sum2(uint32_t wa_biased, uint32_t wb_biased) // |a[K4]| + |b[K4]|
{
    return __builtin_amdgcn_msad_u8(wb_biased, 0x80u << (8 * K4), __builtin_amdgcn_msad_u8(wa_biased, 0x80u << (8 * K4), 0u));
}
// w[k] = |a[k]| + |b[k]| for the 16 values of a unit, and their maximum folded into wmax
__device__ __forceinline__ void abs_sum16(const uint4 &A, const uint4 &B, uint32_t (&w)[16], uint32_t &wmax)
{
    const uint32_t a[4] = {bias4(A.x), bias4(A.y), bias4(A.z), bias4(A.w)}, b[4] = {bias4(B.x), bias4(B.y), bias4(B.z), bias4(B.w)};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        w[4 * j]     = abs_sum2<0>(a[j], b[j]);
        w[4 * j + 1] = abs_sum2<1>(a[j], b[j]);
        w[4 * j + 2] = abs_sum2<2>(a[j], b[j]);
        w[4 * j + 3] = abs_sum2<3>(a[j], b[j]);
        wmax = max(max(wmax, w[4 * j]), max(w[4 * j + 1], max(w[4 * j + 2], w[4 * j + 3])));
    }
}
// m[k] = byte at LDS address base + w[k], as sixteen packed bytes (table entries are 0..127)
__device__ __forceinline__ uint4 lookup16(uint32_t base, const uint32_t (&w)[16])
{
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = pack4u(lds_u8(base + w[4 * j]), lds_u8(base + w[4 * j + 1]), lds_u8(base + w[4 * j + 2]), lds_u8(base + w[4 * j + 3]));
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// g[k] = byte at LDS address base + idx[k], as sixteen packed bytes
__device__ __forceinline__ uint4 gather16_bytes(uint32_t base, const uint32_t (&idx)[16]) { return lookup16(base, idx); }

template <typename Src, int NSLOT, bool MULTI = false>
#ifndef PREP_WPE
#define PREP_WPE 6
#endif
#ifndef PREP_WPE_PK
#define PREP_WPE_PK 6 // the int16-pair variant needs 79 VGPRs unspilled (7 waves: 5.3 ms, 8 waves: 8.9 ms, 6: 4.65 ms per 64k subframes)
#endif
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(Src::kPacked ? PREP_WPE_PK : PREP_WPE, 8))) void k_turbo_prep(Src src, uint32_t K_arg, uint32_t n_cb_arg,
                                                    const uint16_t *__restrict__ pi_arg, PrepOut out, MultiArgs ma)
{
    static_assert(NSLOT == 1, "one unit per thread (K <= 6144 -> at most 384 units)");
    uint32_t K = K_arg, n_cb = n_cb_arg, bidx = blockIdx.x;
    size_t   seg_off = 0;
    const uint16_t *__restrict__ pi = pi_arg;
    if constexpr (MULTI) { // the workgroup's block size out of a merged launch (KSeg)
        const_seg_t &sg = multi_seg(ma, blockIdx.x >> 9);
        K = sg.K; n_cb = sg.n_cb; pi = (const uint16_t *)sg.pi; bidx = blockIdx.x - sg.wg_cb; seg_off = sg.arr_off;
        src.e_cap = sg.e_cap;
        src.seg(sg);
    }
    // the layout above.  No static LDS next to it: the dynamic block then starts at LDS address 0 and the table / gather reads below are "index
    // register + immediate offset", with no base to add
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    const uint32_t cb = xcd_cb(bidx, n_cb), tile = cb >> 6, lane = cb & 63, Kp = kpad64(K), n_units = Kp >> 4;
    const size_t   tile_off = seg_off + (size_t)tile * Kp * 64;
    if (cb >= n_cb) { // uniform
        // the lanes behind the last code block of a size's last tile: the trellis kernel walks them like any lane (a merged launch does not
        // clear the scratch first), so they get zeros to walk
        if (MULTI && cb < ((n_cb + 63u) & ~63u) && threadIdx.x < n_units)
#pragma unroll
            for (int a = 0; a < 6; a++) *reinterpret_cast<uint4 *>(out.arr[a] + unit_off(tile_off, lane, threadIdx.x)) = make_uint4(0, 0, 0, 0);
        return;
    }
    int8_t        *qtab = sm + PREP_QTAB_AT, *qc = qtab + QTAB_HALF, *mtab1 = sm, *mtab2 = mtab1 + MTAB_N, *e_lds = sm + PREP_E_AT;
    int8_t        *q0_lds = sm + PREP_Q0_AT;
    float         *red_f  = reinterpret_cast<float *>(sm + PREP_RED_AT);
    int           *red_i  = reinterpret_cast<int *>(sm + PREP_RED_AT);
    src.init(cb, K);
    const uint32_t u  = threadIdx.x;
    const int      nv = (u < n_units) ? min(16, max(0, (int)K - 16 * (int)u)) : -1; // 16, 8 (last unit of a K % 16 == 8 block), 0 (padding), -1
    // (Tried in round 6: the rank words of the gather requested HERE, before the soft bits are staged, so that the table's round trip runs
    // under the copy's.  The 24 registers they occupy across the copy do not fit next to the gather's own peak: 100 bytes of scratch per
    // lane, and the kernel went from 4.15 to 7.84 ms on W4 -- gpurun_out/hoist1.log.  The kernel is at its register limit for 6 waves.)
    const bool     e_in_lds = src.stage_e(e_lds);

    typedef typename std::conditional<Src::kIntPath, int, float>::type val_t;
    val_t    v[Src::kPacked ? 1 : 3][16]; // the sums of the three streams ...
    uint32_t vp[3][8];                    // ... or, two per register as int16, when the source says they fit
    float    mx;
    int      mxi = 0;
    if constexpr (Src::kPacked) {
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int j = 0; j < 8; j++) vp[x][j] = 0;
        if (!e_in_lds && src.window_ok()) src.gather_windowed_pk(u, nv, vp, e_lds, PREP_E_AT); // (uniform; vp starts at zero)
        else if (nv > 0) src.load16_pk(u, nv, vp, PREP_E_AT, e_in_lds);
        v2s hi = (v2s)(0), lo = (v2s)(0);
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int j = 0; j < 8; j++) { hi = __builtin_elementwise_max(hi, as_v2s(vp[x][j])); lo = __builtin_elementwise_min(lo, as_v2s(vp[x][j])); }
        mxi = block_max_i(max(max((int)hi.x, (int)hi.y), -min((int)lo.x, (int)lo.y)), red_i);
        mx  = (float)mxi;
    } else {
        if (nv > 0) src.load16(u, nv, v, PREP_E_AT, e_in_lds);
        else {
#pragma unroll
            for (int x = 0; x < 3; x++)
#pragma unroll
                for (int k = 0; k < 16; k++) v[Src::kPacked ? 0 : x][k] = 0;
        }
        if constexpr (Src::kIntPath) {
            int hi = 0, lo = 0;
#pragma unroll
            for (int x = 0; x < 3; x++)
#pragma unroll
                for (int k = 0; k < 16; k += 2) { hi = max(max(hi, v[Src::kPacked ? 0 : x][k]), v[x][k + 1]); lo = min(min(lo, v[Src::kPacked ? 0 : x][k]), v[x][k + 1]); }
            mxi = block_max_i(max(hi, -lo), red_i);
            mx  = (float)mxi;
        } else {
            float mxv = 0;
#pragma unroll
            for (int x = 0; x < 3; x++)
#pragma unroll
                for (int k = 0; k < 16; k++) mxv = fmaxf(mxv, fabsf(v[Src::kPacked ? 0 : x][k]));
            mx = block_max_abs_f(mxv, red_f);
        }
    }
    const bool use_qtab = Src::kIntegerValued && mx < (float)QTAB_HALF; // uniform over the workgroup
    // max = 127 (any saturated soft bit and no repetition: every 16QAM / 64QAM allocation): q(d) = (int)(d * 127.0f / 127.0f) = d,
    // the product and the quotient being exact -- no table, the bytes are the values themselves
    const bool identity = Src::kIntPath && mxi == 127;
    // max = 254 (two laps of saturated soft bits, the W4 allocations): q(d) = (int)(d * 127.0f / 254.0f) is d / 2 truncated towards zero for
    // every integer |d| <= 254 (checked exhaustively in float arithmetic, tests/test_oracle.py) -- two packed instructions instead of a table read
    const bool halving = Src::kPacked && mxi == 254;
    if (use_qtab && !identity && !halving) {
        if constexpr (Src::kIntPath) { // signed table centred on a fixed slot: entry QTAB_HALF + d holds q(d) -- one read per element
            for (int d = (int)threadIdx.x - mxi; d <= mxi; d += (int)blockDim.x) qc[d] = (int8_t)(int)((float)d * 127.0f / mx);
        } else {
            for (uint32_t a = threadIdx.x; a <= (uint32_t)mx; a += blockDim.x) qtab[a] = (int8_t)(int)((float)a * 127.0f / mx);
        }
        __syncthreads();
    }

    // quantise stream by stream and keep only the packed bytes (4 VGPRs per stream) across the barriers
    uint4 Q[3];
#pragma unroll
    for (int x = 0; x < 3; x++) {
        if constexpr (Src::kPacked) {
            uint32_t o[4];
            if (identity) { // the low bytes of the four int16 values of two registers
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = __builtin_amdgcn_perm(vp[x][2 * j + 1], vp[x][2 * j], 0x06040200u);
            } else if (halving) { // (d - (d >> 15)) >> 1 = d / 2 towards zero, per int16 half
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const v2s a = as_v2s(vp[x][2 * j]), b = as_v2s(vp[x][2 * j + 1]);
                    o[j] = __builtin_amdgcn_perm(as_u32((b - (b >> 15)) >> 1), as_u32((a - (a >> 15)) >> 1), 0x06040200u);
                }
            } else {
                int d[16];
#pragma unroll
                for (int j = 0; j < 8; j++) { d[2 * j] = (int)__builtin_amdgcn_sbfe(vp[x][j], 0, 16); d[2 * j + 1] = (int)vp[x][j] >> 16; }
                if (use_qtab) {
#pragma unroll
                    for (int j = 0; j < 4; j++) // 0 past the block end -> q(0) = 0
                        o[j] = pack4u(lds_u8(PREP_QTAB_AT + QTAB_HALF + d[4 * j]), lds_u8(PREP_QTAB_AT + QTAB_HALF + d[4 * j + 1]), lds_u8(PREP_QTAB_AT + QTAB_HALF + d[4 * j + 2]), lds_u8(PREP_QTAB_AT + QTAB_HALF + d[4 * j + 3]));
                } else {
                    int q[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) q[k] = (int)((float)d[k] * 127.0f / mx);
                    const uint4 t = pack16(q);
                    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
                }
            }
            Q[x] = make_uint4(o[0], o[1], o[2], o[3]);
        } else if (Src::kIntPath && identity) {
            int q[16];
#pragma unroll
            for (int k = 0; k < 16; k++) q[k] = (int)v[Src::kPacked ? 0 : x][k];
            Q[x] = pack16(q);
        } else if (Src::kIntPath && use_qtab) {
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) // 0 past the block end -> q(0) = 0
                o[j] = pack4u(lds_u8(PREP_QTAB_AT + QTAB_HALF + v[Src::kPacked ? 0 : x][4 * j]), lds_u8(PREP_QTAB_AT + QTAB_HALF + v[Src::kPacked ? 0 : x][4 * j + 1]), lds_u8(PREP_QTAB_AT + QTAB_HALF + v[Src::kPacked ? 0 : x][4 * j + 2]), lds_u8(PREP_QTAB_AT + QTAB_HALF + v[Src::kPacked ? 0 : x][4 * j + 3]));
            Q[x] = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            int q[16];
            if (use_qtab) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float f = v[Src::kPacked ? 0 : x][k]; // 0 past the block end -> q = 0; the lookup is unconditional (no per-element branch)
                    const int   t = qtab[(int)fabsf(f)]; // (int)(-a*127/mx) == -(int)(a*127/mx): IEEE division and truncation are odd
                    q[k]          = f < 0.0f ? -t : t;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) q[k] = (int)((float)v[Src::kPacked ? 0 : x][k] * 127.0f / mx); // v = 0 past the block end
            }
            Q[x] = pack16(q);
        }
        if (nv >= 0) *reinterpret_cast<uint4 *>(out.arr[x] + unit_off(tile_off, lane, u)) = Q[x];
    }
    if (nv >= 0) *reinterpret_cast<uint4 *>(q0_lds + 16 * u) = Q[0];
    __syncthreads();

    uint4 I0 = make_uint4(0, 0, 0, 0);
    if (nv > 0) {
        uint32_t idx[16];
        // past the block end: slot K, which holds q = 0 whenever K % 16 == 8.
        // Merged launch: the table's address came out of memory, not out of a __restrict__ kernel argument, and with that the promise that
        // nothing the kernel stores to is the table; read through the constant address space the promise is back (no measurable effect:
        // the per-size kernel requests the indices behind the barrier too).
        if constexpr (MULTI)
            load_idx16_at(reinterpret_cast<const_u32x4_t *>(reinterpret_cast<uintptr_t>(pi + 16 * (size_t)u)), nv, idx, K);
        else
            load_idx16(pi, u, nv, idx, K);
        I0 = gather16_bytes(PREP_Q0_AT, idx);
    }
    if (nv >= 0) *reinterpret_cast<uint4 *>(out.arr[3] + unit_off(tile_off, lane, u)) = I0;
    if (identity && src.hard_inputs()) {
        // Every soft value of the block is 0 or +-127 (hard-decision 16QAM / 64QAM soft bits, one lap of the circular buffer, identity
        // quantiser): a branch weight w = |a| + |b| is 127 x (how many of the two are non-zero), and that count is the sum of the
        // values' lowest bits (0x7F and 0x81 are odd, 0 is even).  The SISO output magnitude (int8)(127 * (w / W)) is then 127 where
        // w == W, 63 where w is half of W = 254, 0 where w == 0 -- four elements per instruction, no table, no second barrier.
        auto cnt4 = [](uint32_t a, uint32_t b) { return (a & 0x01010101u) + (b & 0x01010101u); };
        const uint32_t c1[4] = {cnt4(Q[1].x, Q[0].x), cnt4(Q[1].y, Q[0].y), cnt4(Q[1].z, Q[0].z), cnt4(Q[1].w, Q[0].w)};
        const uint32_t c2[4] = {cnt4(Q[2].x, I0.x), cnt4(Q[2].y, I0.y), cnt4(Q[2].z, I0.z), cnt4(Q[2].w, I0.w)};
        const uint32_t o1 = c1[0] | c1[1] | c1[2] | c1[3], o2 = c2[0] | c2[1] | c2[2] | c2[3];
        int w1max = (o1 & 0x02020202u) ? 254 : (o1 ? 127 : 0), w2max = (o2 & 0x02020202u) ? 254 : (o2 ? 127 : 0);
        block_max_i2(w1max, w2max, red_i);
        if (w1max != 0 && w2max != 0) { // (an all-zero stream pair divides by zero in the reference: the table path below reproduces what it stores)
            auto mag4 = [](uint32_t cnt, int W) { return W == 254 ? (cnt << 6) - ((cnt | (cnt >> 1)) & 0x01010101u) : (cnt << 7) - cnt; };
            if (nv >= 0) {
                *reinterpret_cast<uint4 *>(out.arr[4] + unit_off(tile_off, lane, u)) = make_uint4(mag4(c1[0], w1max), mag4(c1[1], w1max), mag4(c1[2], w1max), mag4(c1[3], w1max));
                *reinterpret_cast<uint4 *>(out.arr[5] + unit_off(tile_off, lane, u)) = make_uint4(mag4(c2[0], w2max), mag4(c2[1], w2max), mag4(c2[2], w2max), mag4(c2[3], w2max));
            }
            return;
        }
    }
    uint32_t w1[16], w2[16], w1m = 0, w2m = 0; // pass-1 / pass-2 branch weights |q(d1)| + |q(d0)|, |q(d2)| + |I0|: 0 past the block end
    abs_sum16(Q[1], Q[0], w1, w1m);
    abs_sum16(Q[2], I0, w2, w2m);
    int w1max = (int)w1m, w2max = (int)w2m;
    block_max_i2(w1max, w2max, red_i);
    // (int8)(127 * (w / W)) has closed forms for the two maxima that saturated soft values produce: W = 254 -> w >> 1, W = 127 -> w, for
    // every w <= W (checked exhaustively in float arithmetic, tests/test_oracle.py); any other maximum goes through the per-block table
    auto closed = [](int W) { return W == 254 || W == 127; };
    auto closed16 = [](const uint32_t (&w)[16], int W) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t p = pack4u(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
            o[j] = W == 254 ? (p >> 1) & 0x7F7F7F7Fu : p;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    if (!(closed(w1max) && closed(w2max))) {
        const float W1 = (float)w1max, W2 = (float)w2max;
        for (uint32_t t = threadIdx.x; t < MTAB_N; t += blockDim.x) { // w <= 254 always; entries past W are never read
            const float w = (float)t;
            mtab1[t] = (int8_t)(int)(127.0f * (w / W1));
            mtab2[t] = (int8_t)(int)(127.0f * (w / W2));
        }
        __syncthreads();
    }
    if (nv >= 0) {
        *reinterpret_cast<uint4 *>(out.arr[4] + unit_off(tile_off, lane, u)) = closed(w1max) ? closed16(w1, w1max) : lookup16(0, w1); // past the end: entry 0 = 0
        *reinterpret_cast<uint4 *>(out.arr[5] + unit_off(tile_off, lane, u)) = closed(w2max) ? closed16(w2, w2max) : lookup16(MTAB_N, w2);
    }
}

// ------------------------------------------------------------------------------------------------
// siso: viterbi_decode_siso (liblte_phy.cc:10341-10529) for constraint length 4, g = {015,013},
// 64 code blocks in lock-step.  blockIdx.y selects the pass (argument set).
//
// Trellis facts used (derived from liblte_phy.cc:10379-10408):
//   predecessors of s are 2(s&3) and 2(s&3)+1; expected outputs for predecessor k=1 are the
//   complement of those for k=0, so br(s,1) = -br(s,0) =: -beta_s with
//   beta = {P, Q, -Q, -P, -P, -Q, Q, P}[s],  P = e0+e1, Q = e0-e1,  e = +1 if the hard input bit is 1.
//   ACS (liblte_phy.cc:10454-10463): take k=1 iff beta + PM[p0] > -beta + PM[p1]
//                                    <=>  PM[p1] - PM[p0] < 2*beta; new PM = PM[pk] +- w*beta.
//   Traceback (liblte_phy.cc:10484-10497) re-compares the STORED metrics: bit_j = PM[2j] > PM[2j+1].
//
// Two trellises per lane.  The comparisons of the reference -- PM[p1] - PM[p0] against 2*beta in the ACS, PM[2j] > PM[2j+1] in the
// traceback, the minimum over the end states -- only ever look at DIFFERENCES of path metrics, and those stay small: the ACS takes
// the predecessor with the lower stored metric up to a slack of |2*beta| <= 4 and moves it by |w*beta| <= 508, and every state is
// reached from every state in three steps, so at any time max PM - min PM <= 3*512 + 3*508 < 2^15.  The metrics themselves grow
// without bound in the reference's ints, but kept modulo 2^16 every difference of two of them is still exact.  That lets one lane
// walk two independent trellises in the halves of its registers (v_pk_*_i16): two tiles of the first pass, or passes 2 and 3 of the
// same tile (which share their first input, q(d2)).

struct SisoPass {
    const uint8_t *in_a; // first soft value of each pair  (in[2t])
    const uint8_t *in_b; // second soft value of each pair (in[2t+1])
    const uint8_t *mag;  // |output| per step
    uint8_t       *out;  // signed SISO output
    uint32_t      *dec;  // traceback bits: [tile][blk][lane][8 words]
};
struct SisoArgs { SisoPass p[2]; };

typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sign_bits(v2s n) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2u, n) >> 15); } // 1 per negative half

// one trellis step of both halves.  x2, y2: TWICE the step's two soft inputs as int16 pairs (the byte extraction shifts by 7 instead of 8:
// the doubling the branch terms need comes for free); acc: the traceback bits of the current four steps, 16 bits per half
// (bit = PM[2j] > PM[2j+1] = sign of n_j; step S of the four, pair j -> bit 15 - (4S + j)).
// Every instruction of this loop costs four cycles of its SIMD, also the ones that issue in two on their own (v_and, v_xor, v_bitop3):
// profiles/r04_ubench_issue_mix.txt -- a SIMD that alternates between the two kinds runs both at the slow rate -- so what counts is
// the number of instructions, 66 per step and pair of trellises: 7 for the two inputs (2 byte permutes, 2 shifts, 2 sign masks and
// an exclusive-or), 10 branch terms, 4 metric differences, 8 decision bits (a sign mask and one and-or with the bit's own
// constant each: collecting the signs by shifts took 10), 8 threshold tests + 8 masks + 16 candidates + 8 selects.
template <int S> __device__ __forceinline__ void acs_step2(v2s (&pm)[8], v2s x2, v2s y2, uint32_t &acc)
{
    // With w = |x|+|y| and e = +1 for a negative soft value:  w*P = -2(x+y) if the two signs agree, else 0;
    // w*Q = -2(x-y) if they differ, else 0;  2P, 2Q = +-4 (sign of x) under the same conditions.
    const v2s mx = as_v2s(neg_mask(x2 ^ y2)), nmx = ~mx; // mx = -1 iff the signs differ (opaque: see neg_mask)
    const v2s uP = (x2 + y2) & nmx;                          // -w*P
    const v2s uQ = (x2 - y2) & mx;                           // -w*Q
    // 2P, 2Q = (x < 0 ? 4 : -4) where the signs agree / differ, else 0: 0xFFFC ^ (sign mask & 0xFFF8) is 4 or -4, and the masking by
    // mx is the same instruction (v_bitop3) -- four instructions for the pair, the add-and-mask form took five
    const uint32_t m0p = neg_mask(x2) & 0xFFF8FFF8u; // (the opaque mask: from a visible shift the compiler builds compares and selects)
    const v2s P2 = as_v2s(__builtin_amdgcn_bitop3_b32(as_u32(mx), m0p, 0xFFFCFFFCu, 0x06)); // ~mx & (0xFFFC ^ m0p)
    const v2s Q2 = as_v2s(__builtin_amdgcn_bitop3_b32(as_u32(mx), m0p, 0xFFFCFFFCu, 0x60)); //  mx & (0xFFFC ^ m0p)
    const v2s n0 = pm[1] - pm[0], n1 = pm[3] - pm[2], n2 = pm[5] - pm[4], n3 = pm[7] - pm[6];
    constexpr uint32_t top = 0x00010001u << (15 - 4 * S); // the step's first bit in both halves
    // acc | (mask & bit) as ONE instruction each (the compiler's own choice is an and per bit and an or3 per two)
    acc = __builtin_amdgcn_bitop3_b32(acc, neg_mask(n0), top, 0xF8);
    acc = __builtin_amdgcn_bitop3_b32(acc, neg_mask(n1), top >> 1, 0xF8);
    acc = __builtin_amdgcn_bitop3_b32(acc, neg_mask(n2), top >> 2, 0xF8);
    acc = __builtin_amdgcn_bitop3_b32(acc, neg_mask(n3), top >> 3, 0xF8);
    auto sel = [](v2s d, v2s yes, v2s no) { return bit_select(neg_mask(d), yes, no); }; // d < 0 ? yes : no per half
    v2s nw[8];
    nw[0] = sel(n0 - P2, pm[1] + uP, pm[0] - uP); // beta =  P: (n0 <  P2) ? ..
    nw[4] = sel(n0 + P2, pm[1] - uP, pm[0] + uP); // beta = -P: (n0 < -P2) ? ..
    nw[1] = sel(n1 - Q2, pm[3] + uQ, pm[2] - uQ); // beta =  Q
    nw[5] = sel(n1 + Q2, pm[3] - uQ, pm[2] + uQ); // beta = -Q
    nw[2] = sel(n2 + Q2, pm[5] - uQ, pm[4] + uQ); // beta = -Q
    nw[6] = sel(n2 - Q2, pm[5] + uQ, pm[4] - uQ); // beta =  Q
    nw[3] = sel(n3 + P2, pm[7] - uP, pm[6] + uP); // beta = -P
    nw[7] = sel(n3 - P2, pm[7] + uP, pm[6] - uP); // beta =  P
#pragma unroll
    for (int s = 0; s < 8; s++) pm[s] = nw[s];
}

// twice byte r of the two words, sign-extended, as the pair (lo: word0, hi: word1)
template <int R> __device__ __forceinline__ v2s byte_pair2(uint32_t w0, uint32_t w1)
{
    return as_v2s(__builtin_amdgcn_perm(w1, w0, (uint32_t)(4 + R) << 24 | 0x0C0000u | (uint32_t)R << 8 | 0x0Cu)) >> 7;
}

// Traceback two steps at a time.  One step of the reference's traceback (liblte_phy.cc:10483-10527) maps (state at t+1, the four stored
// compare bits of time t) to (state at t, sign of the output); two of them are a function of 3 + 8 bits, kept as a 2048-entry LDS
// table per workgroup: entry = state after both steps | byte masks (0xFF = negative) of the two outputs in bits 8-23, the first
// step's in the upper byte.  The outputs are then the magnitude words with the masked bytes negated (carry-free byte arithmetic).
__device__ __forceinline__ uint32_t traceback_entry(uint32_t byte, uint32_t cur)
{
    uint32_t mask = 0;
    for (int k = 0; k < 2; k++) {
        const uint32_t nib = k ? byte >> 4 : byte & 15u; // the step processed first sits in the low nibble
        const uint32_t j = cur & 3u, bit = (nib >> (3 - j)) & 1u, st = 2 * j + bit; // pair j is bit (3-j) of the nibble
        const bool     pos = (cur < st) || (cur == st && cur == 0); // "+" when the step moved to a lower state, or stayed in state 0
        if (!pos) mask |= k ? 0x00FFu : 0xFF00u;
        cur = st;
    }
    return cur | mask << 8;
}
__device__ __forceinline__ uint32_t negate_bytes(uint32_t w, uint32_t m) // -b in the bytes where m is 0xFF, b elsewhere
{
    const uint32_t x = w ^ m, y = m & 0x01010101u;
    return ((x & 0x7F7F7F7Fu) + y) ^ (x & 0x80808080u);
}

// Workgroups of four independent wavefronts (they only share the traceback table).  mode 0: trellis h of wavefront b is tile 2b + h of
// pass p[0] (first pass, two tiles per lane); mode 1: trellis h is tile b of pass p[h] (passes 2 and 3 of one tile)
#ifndef SISO_WPE
#define SISO_WPE 4
#endif
// MULTI: a merged launch over the tiles of many block sizes (KSeg): n_tiles_arg is then the launch's wavefront count and K, the tile range and
// the arrays' offsets come from the wavefront's row of the table (the wavefronts are ordered by falling K: the long walks start first)
template <bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SISO_WPE, 8))) void k_turbo_siso(SisoArgs args, uint32_t K_arg, uint32_t n_tiles_arg, uint32_t mode, MultiArgs ma,
                                                                                                         const uint32_t *__restrict__ order)
{
    __shared__ uint32_t tb_lut[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) tb_lut[i] = traceback_entry(i >> 3, i & 7u);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t       K = K_arg, n_tiles = n_tiles_arg, wv = blockIdx.x * 4 + (threadIdx.x >> 6), n_wv = mode ? n_tiles : (n_tiles + 1) / 2;
    size_t         seg_off = 0;
    if constexpr (MULTI) {
        // the launch order is not the map's order (mi_turbo_ref_multi: walks of different lengths dealt out so that every compute unit gets its share)
        if (order) wv = __builtin_amdgcn_readfirstlane(order[wv]);
        if (wv >= n_tiles_arg) return; // wavefront-uniform
        const_seg_t &sg = multi_seg(ma, wv);
        K = sg.K; n_tiles = sg.n_tiles; seg_off = sg.arr_off;
        wv -= mode ? sg.wv23 : sg.wv1;
        n_wv = mode ? n_tiles : (n_tiles + 1) / 2;
    }
    const uint32_t Kp = kpad64(K), nblk = Kp >> 6;
    if (wv >= n_wv) return; // wavefront-uniform
    const uint8_t *pa[2], *pb[2], *pmag[2];
    uint8_t       *pout[2];
    uint32_t      *dec[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const SisoPass &ps   = args.p[mode ? h : 0];
        const uint32_t  tile = mode ? wv : min(2 * wv + h, n_tiles - 1); // an odd tile count: the last one twice
        const size_t    off  = seg_off + (size_t)tile * Kp * 64 + lane * 64;
        pa[h] = ps.in_a + off; pb[h] = ps.in_b + off; pmag[h] = ps.mag + off; pout[h] = ps.out + off;
        dec[h] = ps.dec + (seg_off >> 3) + ((size_t)tile * nblk * 64 + lane) * 8; // (traceback: 32 bytes per step and tile = half an array's 64)
    }

    v2s pm[8];
#pragma unroll
    for (int s = 0; s < 8; s++) pm[s] = (v2s)(0); // all path metrics start at 0 (liblte_phy.cc:10411-10418)

    // ---- forward add-compare-select.  A block is 64 steps = one 64-byte line per input and trellis, held as four quarters whose
    // registers are refilled with the next block's data while the current block's last quarter(s) are walked
    uint4 A[2][4], B[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            A[h][q] = reinterpret_cast<const uint4 *>(pa[h])[q];
            B[h][q] = reinterpret_cast<const uint4 *>(pb[h])[q];
        }
    for (uint32_t blk = 0; blk < nblk; blk++) {
        const uint32_t nxt = (blk + 1 < nblk) ? blk + 1 : blk;
        uint32_t       dw[2][8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int g2 = 0; g2 < 2; g2++) { // 2 groups of 8 steps = one decision word per trellis each
                const int g = 2 * q + g2;
                uint32_t  acc_lo = 0, acc_hi = 0; // steps 0-3 / 4-7 of the group, 16 bits per trellis
                if (blk * 64 + g * 8 < K) {       // uniform: K is a multiple of 8
                    const uint32_t a0l = g2 ? A[0][q].z : A[0][q].x, a0h = g2 ? A[0][q].w : A[0][q].y;
                    const uint32_t a1l = g2 ? A[1][q].z : A[1][q].x, a1h = g2 ? A[1][q].w : A[1][q].y;
                    const uint32_t b0l = g2 ? B[0][q].z : B[0][q].x, b0h = g2 ? B[0][q].w : B[0][q].y;
                    const uint32_t b1l = g2 ? B[1][q].z : B[1][q].x, b1h = g2 ? B[1][q].w : B[1][q].y;
                    acs_step2<0>(pm, byte_pair2<0>(a0l, a1l), byte_pair2<0>(b0l, b1l), acc_lo);
                    acs_step2<1>(pm, byte_pair2<1>(a0l, a1l), byte_pair2<1>(b0l, b1l), acc_lo);
                    acs_step2<2>(pm, byte_pair2<2>(a0l, a1l), byte_pair2<2>(b0l, b1l), acc_lo);
                    acs_step2<3>(pm, byte_pair2<3>(a0l, a1l), byte_pair2<3>(b0l, b1l), acc_lo);
                    acs_step2<0>(pm, byte_pair2<0>(a0h, a1h), byte_pair2<0>(b0h, b1h), acc_hi);
                    acs_step2<1>(pm, byte_pair2<1>(a0h, a1h), byte_pair2<1>(b0h, b1h), acc_hi);
                    acs_step2<2>(pm, byte_pair2<2>(a0h, a1h), byte_pair2<2>(b0h, b1h), acc_hi);
                    acs_step2<3>(pm, byte_pair2<3>(a0h, a1h), byte_pair2<3>(b0h, b1h), acc_hi);
                }
                dw[0][g] = __builtin_amdgcn_perm(acc_lo, acc_hi, 0x05040100u); // steps 0-3 in the upper half, 4-7 in the lower
                dw[1][g] = __builtin_amdgcn_perm(acc_lo, acc_hi, 0x07060302u);
            }
            // refill for the next block: quarters 0-2 together once the third quarter is done, the last quarter after itself.  The three
            // loads of a lane hit the same 64-byte line back to back (one L2 request, two L1 hits); refilling each quarter as soon as it
            // was free spread the four loads of a line over a block's worth of steps, and each of them went to the L2 on its own
            if (q >= 2) {
#pragma unroll
                for (int qq = (q == 2 ? 0 : 3); qq <= (q == 2 ? 2 : 3); qq++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        A[h][qq] = reinterpret_cast<const uint4 *>(pa[h] + (size_t)nxt * 4096)[qq];
                        B[h][qq] = reinterpret_cast<const uint4 *>(pb[h] + (size_t)nxt * 4096)[qq];
                    }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint4 *dp = reinterpret_cast<uint4 *>(dec[h] + (size_t)blk * 64 * 8);
            dp[0] = make_uint4(dw[h][0], dw[h][1], dw[h][2], dw[h][3]);
            dp[1] = make_uint4(dw[h][4], dw[h][5], dw[h][6], dw[h][7]);
        }
    }

    // ---- end state: first strict minimum (liblte_phy.cc:10467-10481), on the metrics relative to state 0
    int cur[2] = {0, 0};
    {
        int best[2] = {0, 0};
#pragma unroll
        for (int s = 1; s < 8; s++) {
            const uint32_t d  = as_u32(pm[s] - pm[0]);
            const int      dh[2] = {(int)__builtin_amdgcn_sbfe(d, 0, 16), (int)d >> 16};
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (dh[h] < best[h]) { best[h] = dh[h]; cur[h] = s; }
        }
    }

    // ---- traceback + signed soft output (liblte_phy.cc:10483-10527), the two trellises interleaved
    uint4 M[2][4], Mn[2][4], D[2][2], Dn[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint4 *dp = reinterpret_cast<const uint4 *>(dec[h] + (size_t)(nblk - 1) * 64 * 8);
        D[h][0] = dp[0];
        D[h][1] = dp[1];
#pragma unroll
        for (int q = 0; q < 4; q++) M[h][q] = reinterpret_cast<const uint4 *>(pmag[h] + (size_t)(nblk - 1) * 4096)[q];
    }
    for (int blk = (int)nblk - 1; blk >= 0; blk--) {
        const int prv = blk > 0 ? blk - 1 : 0; // prefetch the block below while this one is traced back
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 *dpn = reinterpret_cast<const uint4 *>(dec[h] + (size_t)prv * 64 * 8);
            Dn[h][0] = dpn[0];
            Dn[h][1] = dpn[1];
#pragma unroll
            for (int q = 0; q < 4; q++) Mn[h][q] = reinterpret_cast<const uint4 *>(pmag[h] + (size_t)prv * 4096)[q];
        }
#pragma unroll
        for (int q = 3; q >= 0; q--) {
            uint32_t ow[2][4];
#pragma unroll
            for (int g2 = 1; g2 >= 0; g2--) {
                const int g = 2 * q + g2;
                uint32_t  o_hi[2] = {0, 0}, o_lo[2] = {0, 0};
                if ((uint32_t)blk * 64 + g * 8 < K) {
                    uint32_t word[2], mlo[2], mhi[2], msk[2][4];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint4 dq = D[h][g >> 2];
                        word[h] = (g & 3) == 0 ? dq.x : (g & 3) == 1 ? dq.y : (g & 3) == 2 ? dq.z : dq.w;
                        mlo[h]  = g2 ? M[h][q].z : M[h][q].x;
                        mhi[h]  = g2 ? M[h][q].w : M[h][q].y;
                    }
                    // the compare bits of step r sit in nibble (7-r) of the word: byte b holds steps 7-2b (low nibble, traced first) and 6-2b
#pragma unroll
                    for (int b = 0; b < 4; b++) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const uint32_t e = tb_lut[((word[h] >> (8 * b)) & 0xFFu) << 3 | (uint32_t)cur[h]];
                            cur[h]    = (int)(e & 7u);
                            msk[h][b] = e >> 8;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++) { // steps 7..4 -> bytes 3..0 of o_hi, steps 3..0 -> o_lo
                        o_hi[h] = negate_bytes(mhi[h], msk[h][0] << 16 | msk[h][1]);
                        o_lo[h] = negate_bytes(mlo[h], msk[h][2] << 16 | msk[h][3]);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; h++) { ow[h][2 * g2] = o_lo[h]; ow[h][2 * g2 + 1] = o_hi[h]; }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                reinterpret_cast<uint4 *>(pout[h] + (size_t)blk * 4096)[q] = make_uint4(ow[h][0], ow[h][1], ow[h][2], ow[h][3]);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            D[h][0] = Dn[h][0];
            D[h][1] = Dn[h][1];
#pragma unroll
            for (int q = 0; q < 4; q++) M[h][q] = Mn[h][q];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// siso for a handful of code blocks (the per-call forms: one transport block per call).  k_turbo_siso puts code blocks on lanes, so one
// block is one lane issuing ~70 instructions per trellis step with the other 63 idle: 112 us per launch at K = 928, which is all of a
// per-call PUSCH decode's time.  Here the lanes are the STATES: four lanes per trellis, lane j holding the metrics of states j and j + 4
// (X, Y).  Lane j computes the new states j and j + 4 -- both come from the pair (PM[2j], PM[2j+1]), with opposite branch signs -- and
// keeps the two as the halves of one register (metrics modulo 2^16 as in k_turbo_siso), so the pair arrives with one quad_perm DPP move and one
// byte permute per operand (no LDS) and both new states are one packed compare-select: 11 instructions per step instead of 70.  The branch
// terms of a step depend on the inputs only, so a pre-pass with the lanes as 64 STEPS computes them for a chunk at a time into LDS.
// The traceback is not walked at all: one step of it is a map of the 8 states onto themselves that depends on that step's four compare
// bits only (state s came from 2(s&3) + bit[s&3]); maps compose associatively, so a suffix scan over the 64 steps of a chunk (lanes as
// steps again, a map = 8 x 3 bits in a register) gives every step's state at once, and with it the sign of every output.
// Same arithmetic as k_turbo_siso, same arrays in and out.
constexpr uint32_t SMALL_G = 8; // trellises per wavefront at most (4 lanes each)
// sel(n - c, b + u, a - u) for the lane's lower state, the upper state takes (-c, -u): both at once on int16 pairs, cw = (c, -c), uw = (u, -u)
struct SmallPar { uint32_t cw, uw; };

// a map of the 8 states onto themselves as 8 bytes (x: states 0-3, y: states 4-7); (F o G)[s] = F[G[s]] is two byte permutes
__device__ __forceinline__ uint2 map_compose(uint2 F, uint2 G)
{
    return make_uint2(__builtin_amdgcn_perm(F.y, F.x, G.x), __builtin_amdgcn_perm(F.y, F.x, G.y));
}
__device__ __forceinline__ uint32_t map_at(uint2 F, uint32_t s) { return ((s & 4u ? F.y : F.x) >> (8 * (s & 3u))) & 7u; }
// lane i <- lane i + N of its row of 16; lanes whose source is past the row end keep `keep`
template <int N> __device__ __forceinline__ uint2 map_row_shl(uint2 v, uint2 keep)
{
    return make_uint2((uint32_t)__builtin_amdgcn_update_dpp((int)keep.x, (int)v.x, 0x100 + N, 0xF, 0xF, false),
                      (uint32_t)__builtin_amdgcn_update_dpp((int)keep.y, (int)v.y, 0x100 + N, 0xF, 0xF, false));
}

// gpw trellises per wavefront (1, 2, 4 or 8): the pre-pass and the traceback take the wavefront's trellises one after the other, so the
// host asks for as few per wavefront as still leaves the device a wavefront or two per SIMD (siso_small_gpw)
// MULTI: a merged launch over the code blocks of several sizes (KSeg): a wavefront's trellises are all of one size, which it reads from its row
template <bool MULTI>
__global__ __launch_bounds__(64) void k_turbo_siso_small(SisoArgs args, uint32_t K_arg, uint32_t n_cb_arg, uint32_t mode, uint32_t gpw, MultiArgs ma)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_small[];
    SmallPar (*par)[64][4] = reinterpret_cast<SmallPar(*)[64][4]>(sm_small);                    // [gpw][64 steps][4 lanes]
    uint32_t K = K_arg, n_cb = n_cb_arg, bidx = blockIdx.x;
    size_t   seg_off = 0;
    if constexpr (MULTI) {
        const_seg_t &sg = multi_seg(ma, blockIdx.x);
        K = sg.K; n_cb = sg.n_cb; seg_off = sg.arr_off; bidx = blockIdx.x - (mode ? sg.ws23 : sg.ws1);
    }
    const uint32_t Kp = kpad64(K), n_w32 = Kp >> 5;
    uint32_t      *decw = sm_small + gpw * 64 * 4 * 2;                                           // [gpw][n_w32][4]: lane j's compare bits, 32 steps per word, first step in bit 31
    const uint32_t lane = threadIdx.x, gi = lane >> 2, j = lane & 3u;
    // trellis T = gpw * b + g of the launch: mode 0: code block T, pass p[0]; mode 1: code block T / 2, pass p[T & 1]
    const uint32_t T0 = bidx * gpw, n_tr = n_cb * (mode ? 2u : 1u);
    const uint32_t n_g = min(gpw, n_tr - T0); // trellises of this wavefront (uniform)
    auto pass_of = [&](uint32_t g) -> const SisoPass & { return args.p[mode ? (T0 + g) & 1u : 0u]; };
    auto off_of  = [&](uint32_t g) -> size_t { // the code block's lane of its tile: element t at (t / 64) * 4096 + t % 64 from here
        const uint32_t cb = mode ? (T0 + g) >> 1 : T0 + g;
        return seg_off + (size_t)(cb >> 6) * Kp * 64 + (cb & 63u) * 64;
    };
    const uint32_t n_chunk = (K + 63) >> 6;

    // ---- forward add-compare-select
    uint32_t XY = 0;   // (PM[j], PM[j + 4]) as int16 halves; all path metrics start at 0 (liblte_phy.cc:10411-10418)
    uint32_t accv = 0;
    const uint32_t half_sel = j < 2 ? 0x01000100u : 0x03020302u; // v_perm selector: this lane's pair comes from the quad's lower / upper halves
#ifdef SMALL_PAD
    uint32_t padv[4] = {lane, lane + 1, lane + 2, lane + 3}; // A/B builds: independent instructions in the step loop (is it issue or latency?)
#endif
    // the inputs of a chunk are requested a chunk ahead (lanes = steps): a wavefront on its own has nothing else to hide that round trip behind
    int xs[SMALL_G], ys[SMALL_G];
    auto fetch = [&](uint32_t ch) {
#pragma unroll
        for (uint32_t g = 0; g < SMALL_G; g++)
            if (g < n_g) {
                const size_t e = off_of(g) + (size_t)ch * 4096 + lane;
                xs[g] = (int8_t)pass_of(g).in_a[e];
                ys[g] = (int8_t)pass_of(g).in_b[e];
            }
    };
    fetch(0);
    for (uint32_t ch = 0; ch < n_chunk; ch++) {
        // branch terms of the chunk's 64 steps, one trellis after the other, lanes = steps (acs_step2's formulas)
#pragma unroll
        for (uint32_t g = 0; g < SMALL_G; g++) {
            if (g >= n_g) break;
            const int      x  = xs[g], y = ys[g];
            const int      m0 = x >> 31, mx = m0 ^ (y >> 31), nmx = ~mx; // mx = -1 iff the signs differ
            const int      uP = ((x + y) << 1) & nmx, uQ = ((x - y) << 1) & mx;
            const int      c4 = (m0 & 8) - 4, P2 = c4 & nmx, Q2 = c4 & mx;
            auto pm16 = [](int v) { return ((uint32_t)v & 0xFFFFu) | ((uint32_t)(-v) << 16); }; // (v, -v)
            const uint32_t cP = pm16(P2), uPw = pm16(uP), cQ = pm16(Q2), uQw = pm16(uQ);
            auto swp = [](uint32_t w) { return (w >> 16) | (w << 16); };                        // (-v, v)
            uint4 *dst = reinterpret_cast<uint4 *>(&par[g][lane][0]);
            dst[0] = make_uint4(cP, uPw, cQ, uQw);                       // states 0 (4): (P2, uP); 1 (5): (Q2, uQ)
            dst[1] = make_uint4(swp(cQ), swp(uQw), swp(cP), swp(uPw));   // states 2 (6): (-Q2, -uQ); 3 (7): (-P2, -uP)
        }
        __syncthreads(); // (one wavefront: orders the LDS accesses for the compiler)
        if (ch + 1 < n_chunk) fetch(ch + 1);
        const uint32_t n_t = min(64u, K - ch * 64);
        if (gi < n_g) {
            // eight steps with their terms requested together (K is a multiple of 8)
            auto steps8 = [&](uint32_t t8) {
                SmallPar pr[8];
#pragma unroll
                for (int k = 0; k < 8; k++) pr[k] = par[gi][t8 + k][j];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // (a, b) = (PM[2j], PM[2j+1]), each in both halves: states 0..3 are the quad's lower halves, states 4..7 its upper halves
                    const uint32_t t0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0x88, 0xF, 0xF, false); // quad_perm [0,2,0,2]
                    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0xDD, 0xF, 0xF, false); // quad_perm [1,3,1,3]
                    const v2s a = as_v2s(__builtin_amdgcn_perm(t0, t0, half_sel)), b = as_v2s(__builtin_amdgcn_perm(t1, t1, half_sel));
                    const v2s n = b - a;
                    accv = __builtin_amdgcn_alignbit(accv, as_u32(n), 31); // (accv << 1) | (PM[2j] > PM[2j+1])
                    const v2s uw = as_v2s(pr[k].uw);
                    XY = as_u32(bit_select(neg_mask(n - as_v2s(pr[k].cw)), b + uw, a - uw)); // lower state: n - c < 0 ? b + u : a - u; upper: (-c, -u)
#ifdef SMALL_PAD
#pragma unroll
                    for (int z = 0; z < SMALL_PAD; z++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(padv[z & 3]) : "v"(lane));
#endif
                }
            };
            // the step loop is bound by instruction issue (a wavefront on its own issues one instruction per ~4.9 cycles: 8 more per step cost
            // 16 ns, tools/ab/small_pad.sh), so whole 32-step words run without a test in between
            uint32_t t = 0;
            for (; t + 32 <= n_t; t += 32) {
                steps8(t); steps8(t + 8); steps8(t + 16); steps8(t + 24);
                decw[(gi * n_w32 + ((ch * 64 + t) >> 5)) * 4 + j] = accv;
            }
            if (t < n_t) { // the block's last word when it is not full: its first step still goes to bit 31
                const uint32_t w = (ch * 64 + t) >> 5, left = n_t - t;
                for (; t < n_t; t += 8) steps8(t);
                decw[(gi * n_w32 + w) * 4 + j] = accv << (32u - left);
            }
        }
        __syncthreads();
    }

#ifdef SMALL_PAD
    if ((padv[0] ^ padv[1] ^ padv[2] ^ padv[3]) == 0x12345678u) decw[0] = 0; // (keeps the padding alive)
#endif
    // ---- end state: first strict minimum (liblte_phy.cc:10467-10481), on the metrics relative to state 0
    uint32_t cur = 0;
    {
        // quad_perm [k,k,k,k]: every lane of the quad sees all eight metrics (differences on 16 bits, as they are kept)
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0x00, 0xF, 0xF, false), q1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0x55, 0xF, 0xF, false);
        const uint32_t q2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0xAA, 0xF, 0xF, false), q3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)XY, 0xFF, 0xF, 0xF, false);
        const short pm[8] = {(short)q0, (short)q1, (short)q2, (short)q3, (short)(q0 >> 16), (short)(q1 >> 16), (short)(q2 >> 16), (short)(q3 >> 16)};
        int best = 0;
#pragma unroll
        for (int st = 1; st < 8; st++) {
            const int d = (short)(pm[st] - pm[0]);
            if (d < best) { best = d; cur = st; }
        }
    }

    // ---- traceback + signed soft output (liblte_phy.cc:10483-10527) by function composition, one trellis after the other, lanes = steps
    const uint2 MAP_ID = make_uint2(0x03020100u, 0x07060504u);
    for (uint32_t g = 0; g < n_g; g++) {
        uint32_t       end = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(g * 4)); // state after the last step
        const uint8_t *mag = pass_of(g).mag + off_of(g);
        uint8_t       *out = pass_of(g).out + off_of(g);
        uint8_t m_nxt = mag[(size_t)(n_chunk - 1) * 4096 + lane]; // the magnitudes too are requested a chunk ahead
        for (int ch = (int)n_chunk - 1; ch >= 0; ch--) {
            const uint32_t t = (uint32_t)ch * 64 + lane;
            const uint8_t  m = m_nxt;
            if (ch > 0) m_nxt = mag[(size_t)(ch - 1) * 4096 + lane];
            const uint4    w = *reinterpret_cast<const uint4 *>(&decw[(g * n_w32 + (t >> 5)) * 4]);
            const uint32_t sh = 31u - (t & 31u);
            uint2          G = MAP_ID; // a step past the block end changes nothing
            if (t < K) {               // state s came from 2 (s & 3) + (compare bit of pair s & 3)
                const uint32_t f = 0x06040200u + (((w.x >> sh) & 1u) | ((w.y >> sh) & 1u) << 8 | ((w.z >> sh) & 1u) << 16 | ((w.w >> sh) & 1u) << 24);
                G = make_uint2(f, f);
            }
            // suffix composition G_t = f_t o f_{t+1} o ... o f_63 (the later steps are applied first): inside each row of 16 lanes by four
            // DPP shifts, then the rows behind are applied whole (their first lane holds their composition)
            G = map_compose(G, map_row_shl<1>(G, MAP_ID));
            G = map_compose(G, map_row_shl<2>(G, MAP_ID));
            G = map_compose(G, map_row_shl<4>(G, MAP_ID));
            G = map_compose(G, map_row_shl<8>(G, MAP_ID));
            const uint32_t row = lane >> 4;
#pragma unroll
            for (int r = 1; r < 4; r++) {
                const uint2 R = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)G.x, 16 * r), (uint32_t)__builtin_amdgcn_readlane((int)G.y, 16 * r));
                if (row < (uint32_t)r) G = map_compose(G, R); // rows r = row + 1 .. 3 in this order: nearest first
            }
            const uint2    Gn  = make_uint2((uint32_t)__shfl_down((int)G.x, 1u), (uint32_t)__shfl_down((int)G.y, 1u));
            const uint32_t nxt = lane == 63 ? end : map_at(Gn, end); // state at t + 1
            const uint32_t st  = map_at(G, end);                      // state at t
            if (t < K) {
                const bool    pos = (nxt < st) || (nxt == st && nxt == 0); // "+" when the step moved to a lower state, or stayed in state 0
                out[(size_t)ch * 4096 + lane] = pos ? m : (uint8_t)(0u - m);
            }
            end = (uint32_t)__builtin_amdgcn_readlane((int)st, 0);
        }
    }
}

// A unit of 16 values x[0..15] plus its three-value halo x[-3..-1], as pairs: E[j] = (x[4j-4], x[4j-2]), O[j] = (x[4j-3], x[4j-1]),
// j = 0 (halo word) .. 4, each pair split into magnitudes and sign masks (0 / 0xFFFF): soft_xor of two such values is
// ((m_a + m_b) >> 1, s_a ^ s_b), three instructions per pair.  The delayed sequences the soft re-encoder needs are
//   x[k-2]: even pairs (E[j].hi, E[j+1].lo), odd pairs (O[j].hi, O[j+1].lo)  -- one v_alignbit per component
//   x[k-3]: even pairs O[j], odd pairs = the even pairs of x[k-2]              -- free
// with x[<0] = +127 in the first unit (conv_encode_soft register preset, liblte_phy.cc:10097-10100).
struct SM { v2s m; uint32_t s; }; // |x| and the sign mask of a pair
__device__ __forceinline__ SM   to_sm(v2s x) { return SM{abs2(x), neg_mask(x)}; }
__device__ __forceinline__ v2s  to_tc(const SM &a) { return as_v2s(as_u32(a.m) ^ a.s) - as_v2s(a.s); } // back to two's complement
__device__ __forceinline__ SM   sxor_sm(const SM &a, const SM &b) { return SM{half_sum(a.m, b.m), a.s ^ b.s}; }
struct UnitWords { uint32_t w[5]; }; // halo word, then the unit's four words
__device__ __forceinline__ UnitWords load_unit_words(const uint8_t *arr, size_t tile_off, uint32_t lane, uint32_t u)
{
    const uint4 c    = *reinterpret_cast<const uint4 *>(arr + unit_off(tile_off, lane, u));
    uint32_t    prev = 0x7F7F7F7Fu;
    if (u > 0) prev = *reinterpret_cast<const uint32_t *>(arr + unit_off(tile_off, lane, u - 1) + 12);
    return UnitWords{{prev, c.x, c.y, c.z, c.w}};
}
// the pairs of one word, and of the word before it: converted as the walk over the unit reaches them, so that only two words of
// an array are live in sign-magnitude form at a time
struct WordPairs { v2s et, ot; SM e, o; }; // two's complement and sign-magnitude
__device__ __forceinline__ WordPairs word_pairs(uint32_t w)
{
    const v2s e = even2(w), o = odd2(w);
    return WordPairs{e, o, to_sm(e), to_sm(o)};
}
__device__ __forceinline__ SM delay_sm(const SM &cur, const SM &prev) // (prev.hi, cur.lo)
{
    return SM{as_v2s(__builtin_amdgcn_alignbit(as_u32(cur.m), as_u32(prev.m), 16)), __builtin_amdgcn_alignbit(cur.s, prev.s, 16)};
}
// a soft_xor result in all three forms its consumers want: two's complement, magnitude, and the sign mask OF THE VALUE -- a result
// of magnitude 0 is +0 to whatever reads it next (soft_xor counts 0 as positive), whatever the signs of its operands were
struct SX { v2s tc, m; uint32_t s; };
__device__ __forceinline__ SX sxor_full(const SM &a, const SM &b)
{
    const SM  r  = sxor_sm(a, b);
    const v2s tc = to_tc(r);
    return SX{tc, r.m, neg_mask(tc)};
}
// fb = soft_xor(x[k-2], x[k-3]) for the values of word `cur`, `prev` being the word before it: even and odd pairs
__device__ __forceinline__ void feedback_sm(const WordPairs &cur, const WordPairs &prev, SX &fe, SX &fo)
{
    const SM d2e = delay_sm(cur.e, prev.e), d2o = delay_sm(cur.o, prev.o);
    fe = sxor_full(d2e, prev.o);
    fo = sxor_full(d2o, d2e);
}
// soft_xor(a, f) in two's complement
__device__ __forceinline__ v2s sxor_tc(const SM &a, const SX &f) { return to_tc(SM{half_sum(a.m, f.m), a.s ^ f.s}); }

// ------------------------------------------------------------------------------------------------
// perm: Steps 2, 3, 5 and the pass-3 output magnitudes.  One workgroup per code block.
//   C1 = soft_xor(A1, fb(A1)); I1[i] = C1[pi[i]]; M3 from pairs (q(d2), I1)
struct PermArgs { const uint8_t *A1; const uint8_t *X2; uint8_t *out[2]; /* I1, M3 */ };

// A workgroup handles PERM_NB code blocks one after the other (workgroup b + i * gridDim.x, i.e. the same XCD's chunk each time): thread u's
// sixteen interleaver indices are the same for every block, so they are read once and stay in registers -- read per block, the table (the
// same 6.5 KB for every workgroup of the launch) was 15 % of the kernel's time.
#ifndef PERM_NB
#define PERM_NB 4
#endif
static_assert(PERM_NB == 4, "a size's perm grid is a multiple of 128 workgroups (the merged decode's map granularity) only for four blocks per workgroup");
template <int NSLOT, bool MULTI = false>
__global__ __launch_bounds__(384) void k_turbo_perm(PermArgs a, uint32_t K_arg, uint32_t n_cb_arg, const uint16_t *__restrict__ pi_arg, MultiArgs ma)
{
    static_assert(NSLOT == 1, "one unit per thread");
    uint32_t K = K_arg, n_cb = n_cb_arg, bidx = blockIdx.x, grid = gridDim.x;
    size_t   seg_off = 0;
    const uint16_t *__restrict__ pi = pi_arg;
    if constexpr (MULTI) { // the workgroup's block size out of a merged launch (KSeg)
        const_seg_t &sg = multi_seg(ma, blockIdx.x >> 7);
        K = sg.K; n_cb = sg.n_cb; pi = (const uint16_t *)sg.pi; bidx = blockIdx.x - sg.wg_perm; grid = sg.perm_grid;
        seg_off = sg.arr_off;
    }
    extern __shared__ __attribute__((aligned(16))) int8_t smp[]; // mtab[256] | C1[Kp] | reduction scratch (32 B); no static LDS (see k_turbo_prep)
    int8_t *mtab = smp, *sm = smp + MTAB_N;
    const uint32_t Kp = kpad64(K), n_units = Kp >> 4;
    int           *red_i = reinterpret_cast<int *>(sm + Kp);
    const uint32_t u0 = threadIdx.x;
    const int      nv = (u0 < n_units) ? min(16, max(0, (int)K - 16 * (int)u0)) : -1;
    IdxRaw         praw; // the unit's interleaver indices; past the block end: slot K (C1 = 0 there)
    {
        const uint4 *ip = reinterpret_cast<const uint4 *>(pi + 16 * (size_t)(nv > 0 ? u0 : 0));
        const uint32_t kk = K | K << 16;
        praw.lo = ip[0];
        praw.hi = (nv > 8) ? ip[1] : make_uint4(kk, kk, kk, kk);
    }
#pragma unroll 1
    for (uint32_t it = 0; it < PERM_NB; it++) {
        const uint32_t cb = xcd_cb(bidx + it * grid, n_cb), tile = cb >> 6, lane = cb & 63;
        const size_t   tile_off = seg_off + (size_t)tile * Kp * 64;
        if (cb >= n_cb) { // uniform
            if (MULTI && cb < ((n_cb + 63u) & ~63u) && nv >= 0) { // the idle lanes of a size's last tile get zeros to walk (see k_turbo_prep)
                *reinterpret_cast<uint4 *>(a.out[0] + unit_off(tile_off, lane, u0)) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(a.out[1] + unit_off(tile_off, lane, u0)) = make_uint4(0, 0, 0, 0);
            }
            continue;
        }
        // keep the loop body what the one-block kernel was: with the thread index and the packed indices opaque per iteration nothing derived
        // from them (addresses, the sixteen unpacked indices) is hoisted out of the loop, which costs 33 registers and two waves per SIMD
        uint32_t u = u0;
        asm volatile("" : "+v"(u), "+v"(praw.lo.x), "+v"(praw.lo.y), "+v"(praw.lo.z), "+v"(praw.lo.w), "+v"(praw.hi.x), "+v"(praw.hi.y), "+v"(praw.hi.z), "+v"(praw.hi.w));
        uint4          X2 = make_uint4(0, 0, 0, 0);
        if (nv >= 0) {
            const UnitWords wa = load_unit_words(a.A1, tile_off, lane, u);
            WordPairs       pa = word_pairs(wa.w[0]);
            X2 = *reinterpret_cast<const uint4 *>(a.X2 + unit_off(tile_off, lane, u));
            uint32_t c1[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { // Steps 2-3; 0 past the block end
                const WordPairs ca = word_pairs(wa.w[j + 1]);
                SX              fe, fo;
                feedback_sm(ca, pa, fe, fo);
                c1[j] = (4 * j < nv) ? merge_bytes(sxor_tc(ca.e, fe), sxor_tc(ca.o, fo)) : 0u;
                pa    = ca;
            }
            *reinterpret_cast<uint4 *>(sm + 16 * u) = make_uint4(c1[0], c1[1], c1[2], c1[3]);
        }
        __syncthreads();
        uint4 I1 = make_uint4(0, 0, 0, 0);
        if (nv > 0) {
            uint32_t idx[16];
            unpack_idx16(praw, idx);
            I1 = gather16_bytes(MTAB_N, idx); // Step 5
        }
        if (nv >= 0) *reinterpret_cast<uint4 *>(a.out[0] + unit_off(tile_off, lane, u)) = I1;
        uint32_t w[16], wm = 0; // |q(d2)| + |I1|; both are 0 past the block end
        if (nv <= 0) X2 = make_uint4(0, 0, 0, 0);
        abs_sum16(X2, I1, w, wm);
        const int   wmax = block_max_i((int)wm, red_i); // (its two barriers also fence this block's C1 reads from the next block's writes)
        if (wmax == 254 || wmax == 127) { // closed forms of (int8)(127 * (w / W)), as in k_turbo_prep
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pack4u(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
                o[j] = wmax == 254 ? (p >> 1) & 0x7F7F7F7Fu : p;
            }
            if (nv >= 0) *reinterpret_cast<uint4 *>(a.out[1] + unit_off(tile_off, lane, u)) = make_uint4(o[0], o[1], o[2], o[3]);
            continue;
        }
        const float W = (float)wmax;
        for (uint32_t t = threadIdx.x; t < MTAB_N; t += blockDim.x) mtab[t] = (int8_t)(int)(127.0f * ((float)t / W)); // one division per distinct w
        __syncthreads();
        if (nv >= 0) *reinterpret_cast<uint4 *>(a.out[1] + unit_off(tile_off, lane, u)) = lookup16(0, w);
        __syncthreads(); // the table is rebuilt for the next block
    }
}

// ------------------------------------------------------------------------------------------------
// vote: Steps 2-3 (again, C1 is cheap to recompute), 8-14.  One workgroup per code block.
struct VoteArgs { const uint8_t *X0, *A1, *B1, *B2; };

// GROUP = false: write the K hard bits of block cb to c_bits (turbo_decode's own output).
// GROUP = true : finish dlsch_channel_decode (liblte_phy.cc:12840-12869): drop the F filler positions
//                (liblte_phy_code_block_desegmentation, :9948-9986), check CRC24A (calc_crc :9713-9743)
//                and report LIBLTE_SUCCESS / LIBLTE_ERROR_DECODE_FAIL like liblte_phy_pdsch_channel_decode.
template <bool GROUP, int NSLOT, bool MULTI = false>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_turbo_vote(VoteArgs a, uint32_t K_arg, uint32_t n_cb_arg, const uint16_t *__restrict__ inv_arg,
                                                    uint8_t *__restrict__ c_bits, GroupDesc g, MultiArgs ma)
{
    static_assert(NSLOT == 1, "one unit per thread");
    uint32_t K = K_arg, n_cb = n_cb_arg, bidx = blockIdx.x;
    size_t   seg_off = 0;
    const uint16_t *__restrict__ inv = inv_arg;
    if constexpr (MULTI) { // the workgroup's block size out of a merged launch (KSeg)
        const_seg_t &sg = multi_seg(ma, blockIdx.x >> 9);
        K = sg.K; n_cb = sg.n_cb; inv = (const uint16_t *)sg.inv2; bidx = blockIdx.x - sg.wg_cb; seg_off = sg.arr_off;
        g.desc += sg.cb_base;
    }
    extern __shared__ __attribute__((aligned(16))) int8_t sm[]; // D12[Kp + 16] (int16: D1 + D2, a zero slot at Kp) | bits[Kp] | reduction scratch (32 B); no static LDS
    const uint32_t cb = xcd_cb(bidx, n_cb), tile = cb >> 6, lane = cb & 63, Kp = kpad64(K), n_units = Kp >> 4;
    if (cb >= n_cb) return;
    const size_t   tile_off = seg_off + (size_t)tile * Kp * 64;
    int8_t        *d12 = sm, *bits = sm + 2 * Kp + 32;
    uint32_t      *red_u = reinterpret_cast<uint32_t *>(bits + Kp);
    const uint32_t u  = threadIdx.x;
    const int      nv = (u < n_units) ? min(16, max(0, (int)K - 16 * (int)u)) : -1;
    // the de-interleaver's byte offsets into D12 (ctx.hpp, d_inv2: holes and the positions past the block end point at the zero slot), 16 per
    // unit; needed after the barrier, requested now
    IdxRaw vraw;
    {
        const uint4 *ip = reinterpret_cast<const uint4 *>(inv + 16 * (size_t)(nv > 0 ? u : 0));
        vraw.lo = ip[0];
        vraw.hi = ip[1];
    }
    // ... and so is the CRC weight of the unit's LAST position (GROUP): bit j of the block weighs x^(K-1-j) mod g, so position 16u + 15
    // weighs entry K - 16u - 16 (a short last unit of eight bits: the same expression, a negative exponent, its upper eight positions masked)
    const uint32_t crc_w0 = (GROUP && nv > 0) ? g.crc_tab[MI_CRC_TAB_BIAS + K - 16 * u - 16] : 0u;
    v2s            s0e[4], s0o[4]; // s0 = q(d0) + C1, the part of the vote that is not de-interleaved
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t *>(d12 + 2 * Kp) = 0u; // what a hole of the de-interleaver reads
    if (nv >= 0) {
        const UnitWords wa = load_unit_words(a.A1, tile_off, lane, u), wb = load_unit_words(a.B1, tile_off, lane, u),
                        wc = load_unit_words(a.B2, tile_off, lane, u);
        WordPairs       pa = word_pairs(wa.w[0]), pb = word_pairs(wb.w[0]), pc = word_pairs(wc.w[0]);
        const uint4    x0 = *reinterpret_cast<const uint4 *>(a.X0 + unit_off(tile_off, lane, u));
        const uint32_t x0w[4] = {x0.x, x0.y, x0.z, x0.w};
        uint32_t       dn[8]; // D1 + D2 of the unit in natural order, two int16 per word
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const WordPairs ca = word_pairs(wa.w[j + 1]), cb1 = word_pairs(wb.w[j + 1]), cb2 = word_pairs(wc.w[j + 1]);
            SX              fe, fo, ge, go, he, ho;
            feedback_sm(ca, pa, fe, fo);
            feedback_sm(cb1, pb, ge, go); // G  = soft_xor(B1[k-2], B1[k-3])
            feedback_sm(cb2, pc, he, ho); // G' = soft_xor(B2[k-2], B2[k-3])
            v2s d[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const SM  &A = h ? ca.o : ca.e, &B = h ? cb1.o : cb1.e, &B_ = h ? cb2.o : cb2.e;
                const SX  &G = h ? go : ge, &G_ = h ? ho : he;
                const v2s  At = h ? ca.ot : ca.et;
                // Step 10 (liblte_phy.cc:10778-10797), as selects: equal signs -> (|B|+|G|)>>1; B >= 0 > G -> -((A - G) >> 1);
                // B < 0 <= G -> -((-A + G) >> 1) -- the mixed-sign branches read in_act_1 (A), not int_act_1
                const v2s dAG = At - G.tc, t = as_v2s(as_u32(dAG) ^ B.s) - as_v2s(B.s); // (B >= 0) ? A - G : G - A
                const v2s v1  = bit_select(B.s ^ G.s, (v2s)(0) - (t >> 1), half_sum(B.m, G.m));
                // Step 11 (liblte_phy.cc:10800-10819): mixed signs -> -((B - G) >> 1) resp. -((-B - G) >> 1), i.e. -((|B| - G) >> 1)
                const v2s v2  = bit_select(B_.s ^ G_.s, (v2s)(0) - ((B_.m - G_.tc) >> 1), half_sum(B_.m, G_.m));
                d[h] = v1 + v2;
                const v2s c1 = sxor_tc(A, h ? fo : fe); // Steps 2-3
                (h ? s0o[j] : s0e[j]) = (h ? odd2(x0w[j]) : even2(x0w[j])) + c1;
            }
            pa = ca; pb = cb1; pc = cb2;
            const bool in = 4 * j < nv; // 0 past the block end (nv is 16, 8 or 0)
            dn[2 * j]     = in ? __builtin_amdgcn_perm(as_u32(d[1]), as_u32(d[0]), 0x05040100u) : 0u; // (e.lo, o.lo)
            dn[2 * j + 1] = in ? __builtin_amdgcn_perm(as_u32(d[1]), as_u32(d[0]), 0x07060302u) : 0u; // (e.hi, o.hi)
        }
        uint4 *dst = reinterpret_cast<uint4 *>(d12 + 32 * u);
        dst[0] = make_uint4(dn[0], dn[1], dn[2], dn[3]);
        dst[1] = make_uint4(dn[4], dn[5], dn[6], dn[7]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) s0e[j] = s0o[j] = (v2s)(0);
    }
    __syncthreads();
    uint32_t alloc = 0, tbs = 0, F = 0, crc = 0;
    if (GROUP) {
        alloc = g.desc[cb].alloc;
        tbs   = g.desc[cb].tbs;
        F     = K - tbs - 24;
    }
    if (nv > 0) {
        // Steps 12-14: de-interleave D1 + D2 (a hole, or a position past the block end, reads the zero slot), add, take the sign
        const uint32_t  iw[8] = {vraw.lo.x, vraw.lo.y, vraw.lo.z, vraw.lo.w, vraw.hi.x, vraw.hi.y, vraw.hi.z, vraw.hi.w};
        uint32_t        me[4], mo[4], bw[4]; // sign masks of the even / odd pairs (0xFFFF per negative sum), the bits one per byte
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i0 = iw[2 * j] & 0xFFFFu, i1 = iw[2 * j] >> 16, i2 = iw[2 * j + 1] & 0xFFFFu, i3 = iw[2 * j + 1] >> 16;
            const uint32_t ge = lds_u16(i0) | lds_u16(i2) << 16, go = lds_u16(i1) | lds_u16(i3) << 16; // D12 sits at LDS address 0
            me[j] = as_u32((s0e[j] + as_v2s(ge)) >> 15);
            mo[j] = as_u32((s0o[j] + as_v2s(go)) >> 15);
            bw[j] = (me[j] & 0x00010001u) | (mo[j] & 0x00010001u) << 8; // Step 14
        }
        if (GROUP) {
            // CRC24A over the block without its F filler bits: bit j of the block weighs x^(K-1-j) mod g (the 24 parity bits weigh
            // themselves), so the check "calc_crc(a) == p" is "XOR of the weights of the set bits == 0".  The sixteen weights of a unit are
            // x^0 .. x^15 times the weight of its last position: one table entry, then fifteen "times x" steps (shift, sign, conditional XOR
            // of g) -- the table itself, read sixteen entries per thread by every workgroup of the launch, cost the kernel a quarter of
            // its time in cache traffic.
            if (!(nv == 16 && 16 * u >= F)) { // a unit with filler or past the block end: clear the decisions that do not count
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    auto ok = [&](int k) { return (k < nv && 16 * u + k >= F) ? 0xFFFFu : 0u; };
                    me[j] &= ok(4 * j) | ok(4 * j + 2) << 16;
                    mo[j] &= ok(4 * j + 1) | ok(4 * j + 3) << 16;
                }
            }
            const uint32_t GS = 0x864CFBu << 8; // g without its x^24 term, in the table's left-aligned form
            uint32_t       w = crc_w0;
#pragma unroll
            for (int k = 15; k >= 0; k--) {
                const int      j = k >> 2;
                const uint32_t m = (k & 3) == 0 ? (uint32_t)__builtin_amdgcn_sbfe(me[j], 0, 1) : (k & 3) == 1 ? (uint32_t)__builtin_amdgcn_sbfe(mo[j], 0, 1)
                                 : (k & 3) == 2 ? (uint32_t)((int)me[j] >> 31) : (uint32_t)((int)mo[j] >> 31);
                crc ^= w & m;
                if (k > 0) w = (w << 1) ^ ((uint32_t)((int)w >> 31) & GS);
            }
        }
        const uint4 pk = make_uint4(bw[0], bw[1], bw[2], bw[3]);
        if (GROUP) *reinterpret_cast<uint4 *>(bits + 16 * u) = pk;
        else {
            uint2 *o = reinterpret_cast<uint2 *>(c_bits + (size_t)cb * K + 16 * u); // 8-byte aligned (K % 8 == 0)
            o[0] = make_uint2(pk.x, pk.y);
            if (nv > 8) o[1] = make_uint2(pk.z, pk.w);
        }
    }
    if (GROUP) {
        crc = wave_xor_u(crc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red_u[threadIdx.x >> 6] = crc;
        __syncthreads();
        crc = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) crc ^= red_u[w];
        if (threadIdx.x == 0) g.status[alloc] = (crc == 0) ? 0 /* LIBLTE_SUCCESS */ : 2 /* LIBLTE_ERROR_DECODE_FAIL */;
        // transport block = bits F .. F+tbs-1 of the code block, one bit per byte, 16 bytes per store where aligned
        uint8_t *o = g.out_bits + (size_t)alloc * g.out_stride;
        if (g.packed && ((F | tbs) & 7u) == 0) {
            // eight bits per byte (SURVEY 8d's K/8-byte output): the bytes in LDS are 0 / 1, so four of them become a nibble by one
            // multiplication (b0 b1 b2 b3 -> b0<<3 | b1<<2 | b2<<1 | b3 in the product's top byte); 32 bits per thread and store
            const uint32_t nby = tbs >> 3, nw = nby >> 2;
            auto byte_at = [&](uint32_t m) -> uint32_t { // output byte m = bits F + 8m .. F + 8m + 7
                const uint2 w = *reinterpret_cast<const uint2 *>(bits + F + 8 * m);
                return (((w.x * 0x08040201u) >> 24) & 0xFu) << 4 | (((w.y * 0x08040201u) >> 24) & 0xFu);
            };
            for (uint32_t m = threadIdx.x; m < nw; m += blockDim.x)
                reinterpret_cast<uint32_t *>(o)[m] = byte_at(4 * m) | byte_at(4 * m + 1) << 8 | byte_at(4 * m + 2) << 16 | byte_at(4 * m + 3) << 24;
            for (uint32_t m = 4 * nw + threadIdx.x; m < nby; m += blockDim.x) o[m] = (uint8_t)byte_at(m);
        } else if (g.packed) { // a transport block size that is not a multiple of 8 (none of TS 36.213's is): bit by bit, last byte zero-padded
            for (uint32_t m = threadIdx.x; m < (tbs + 7) >> 3; m += blockDim.x) {
                uint32_t v = 0;
                for (uint32_t b = 0; b < 8; b++) v = v << 1 | ((8 * m + b < tbs) ? ((uint32_t)bits[F + 8 * m + b] & 1u) : 0u);
                o[m] = (uint8_t)v;
            }
        } else if ((F & 3u) == 0) {
            const uint32_t nq = tbs >> 2;
            const uint32_t *bw = reinterpret_cast<const uint32_t *>(bits + F);
            for (uint32_t m = threadIdx.x; m < nq; m += blockDim.x) reinterpret_cast<uint32_t *>(o)[m] = bw[m];
            for (uint32_t m = 4 * nq + threadIdx.x; m < tbs; m += blockDim.x) o[m] = (uint8_t)bits[m + F];
        } else
            for (uint32_t m = threadIdx.x; m < tbs; m += blockDim.x) o[m] = (uint8_t)bits[m + F];
    }
}

// one thread per code block of the group: the descriptor the per-code-block kernels read (CbDesc)
__global__ __launch_bounds__(256) void k_cb_desc(GroupDesc g, uint32_t n_cb, const uint32_t *__restrict__ nnn, CbDesc *__restrict__ out)
{
    const uint32_t cb = blockIdx.x * blockDim.x + threadIdx.x;
    if (cb >= n_cb) return;
    const uint32_t a = g.cb_alloc[cb], txm = g.allocs[a].tx_mode;
    const uint32_t combo = g.ul ? 8u + (g.allocs[a].rv_idx & 3u)
                                : ((g.allocs[a].rv_idx & 3u) << 1) | ((txm == 3 || txm == 4 || txm == 8 || txm == 9) ? 1u : 0u);
    CbDesc d;
    d.alloc = a; d.e_off = g.e_off[a]; d.E = g.e_len[a]; d.combo = combo; d.Nnn = nnn[combo];
    d.hard  = (g.allocs[a].mod_type >= 2 && d.E <= d.Nnn) ? 1u : 0u; // the de-mapper's 16QAM / 64QAM soft bits are all +-127 (liblte_phy.cc:9573-9659), and no position is summed
    d.tbs   = g.allocs[a].tbs; d.pad = 0;
    out[cb] = d;
}

// the same for a merged decode: one thread per code-block slot of the whole batch, the slot's size found by bisection of the table
__global__ __launch_bounds__(256) void k_cb_desc_multi(GroupDesc g, uint32_t n_cb, const KSeg *__restrict__ segs, uint32_t n_seg, CbDesc *__restrict__ out)
{
    const uint32_t cb = blockIdx.x * blockDim.x + threadIdx.x;
    if (cb >= n_cb) return;
    uint32_t lo = 0, hi = n_seg; // segs are in slot order: the last one with cb_base <= cb
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].cb_base <= cb) lo = mid; else hi = mid;
    }
    const uint32_t a = g.cb_alloc[cb], txm = g.allocs[a].tx_mode;
    const uint32_t combo = g.ul ? 8u + (g.allocs[a].rv_idx & 3u)
                                : ((g.allocs[a].rv_idx & 3u) << 1) | ((txm == 3 || txm == 4 || txm == 8 || txm == 9) ? 1u : 0u);
    CbDesc d;
    d.alloc = a; d.e_off = g.e_off[a]; d.E = g.e_len[a]; d.combo = combo; d.Nnn = segs[lo].nnn[combo];
    d.hard  = (g.allocs[a].mod_type >= 2 && d.E <= d.Nnn) ? 1u : 0u;
    d.tbs   = g.allocs[a].tbs; d.pad = 0;
    out[cb] = d;
}

// rank tables for the fused rate un-matching: one launch per block size, cached in the context.
// combo 0..7 = rv*2 + (K_mimo == 2) with M_dl_harq = 8, N_soft = 250368, C = 1 as liblte_phy_pdsch_channel_decode passes them;
// combo 8..11 = UL-SCH, rv = combo - 8: ulsch_channel_decode passes chan_type ULSCH, so N_cb = K_w (liblte_phy.cc:12437-12449).
__global__ __launch_bounds__(256) void k_rm_rank_table(uint32_t K, uint16_t *__restrict__ tabs, uint32_t *__restrict__ nnn)
{
    const uint32_t combo = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    RmGeom rm;
    if (combo < 8) rm.init(K + 4, (combo & 1) ? 3 : 1, combo >> 1);
    else           rm.init(K + 4, 1, combo - 8, 1, 1, 1, false);
    if (t == 0) nnn[combo] = rm.Nnn;
    if (t >= 3 * K) return;
    const uint32_t x = t / K, i = t - x * K;
    uint32_t p, c;
    rm.pos_cnt(i, (int)x, p, c);
    uint32_t k = 0xFFFFu;
    if (p < rm.N_cb) k = (p >= rm.k0m) ? c - rm.cnt_k0 : rm.Nnn - rm.cnt_k0 + c;
    tabs[(size_t)combo * 3 * K + t] = (uint16_t)k;
}

// ------------------------------------------------------------------------------------------------
// stand-alone turbo rate un-matching with the reference's float interface
// (liblte_phy_rate_unmatch_turbo, liblte_phy.cc:11246-11490): d[i*3+x] interleaved, positions no
// e bit reaches read RX_NULL_BIT (10000.0f); repeats are added in transmission order, a repeat whose
// value is RX_NULL_BIT is skipped (:11407-11412).
struct RmParams { uint32_t D, E, C, tx_mode, N_soft, M_dl_harq, limited, rv; };

__global__ __launch_bounds__(256) void k_rate_unmatch_f32(const float *__restrict__ e, RmParams pr, uint32_t n_cb,
                                                          float *__restrict__ d)
{
    RmGeom rm;
    rm.init(pr.D, pr.tx_mode, pr.rv, pr.N_soft, pr.M_dl_harq, pr.C, pr.limited != 0);
    const uint32_t cb = blockIdx.y;
    const float   *eb = e + (size_t)cb * pr.E;
    float         *db = d + (size_t)cb * 3 * pr.D;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < 3 * pr.D; t += gridDim.x * blockDim.x) {
        const uint32_t i = t / 3;
        const int      x = (int)(t - 3 * i);
        const uint32_t p = rm.pos(i, x);
        float          v = (float)RX_NULL_AS_INT;
        if (p < rm.N_cb) {
            const uint32_t c = rm.cnt(p);
            uint32_t       k = (p >= rm.k0m) ? c - rm.cnt_k0 : rm.Nnn - rm.cnt_k0 + c;
            if (k < pr.E) {
                v = eb[k];
                for (k += rm.Nnn; k < pr.E; k += rm.Nnn)
                    if (eb[k] != (float)RX_NULL_AS_INT) v += eb[k];
            }
        }
        db[t] = v;
    }
}

// ---- BCJR mode of the PDSCH / PUSCH chains (MI_LTE_TURBO_BCJR): the max-log-MAP decoder (bcjr.hip) takes int8 channel values
// d[i*3+x], tail included, so the soft bits are rate-un-matched into that layout first (same walk as above, sums of repeats
// saturated to +-127, positions no bit reaches = 0), and the decoded block is finished the way dlsch_channel_decode does
// (liblte_phy.cc:12840-12869): filler removed, CRC24A checked, transport block written one bit per byte.
__global__ __launch_bounds__(256) void k_rm_to_i8(GroupDesc g, uint32_t K, uint32_t n_cb, const uint16_t *__restrict__ tabs, const uint32_t *__restrict__ nnn,
                                                  int8_t *__restrict__ d_soft, uint32_t e_cap)
{
    // the allocation's soft bits staged in LDS (16-byte loads; a zero behind index E, so that "rank the allocation does not reach" and
    // "never filled" both read 0 without a predicate), then four trellis positions per thread: their three rank words come in as
    // 8-byte loads, the twelve values leave as three dwords
    extern __shared__ __attribute__((aligned(16))) int8_t e_lds[];
    __shared__ RmGeom rm_s;
    const uint32_t cb = blockIdx.x, D = K + 4, a = g.cb_alloc[cb], txm = g.allocs[a].tx_mode, rv = g.allocs[a].rv_idx & 3u;
    if (threadIdx.x == 0) { // the geometry is the same for the whole code block; only the 12 tail values need it (the tables stop at K)
        if (g.ul) rm_s.init(D, 1, rv, 1, 1, 1, false);
        else      rm_s.init(D, txm, rv);
    }
    const uint32_t combo = g.ul ? 8u + rv : (rv << 1) | ((txm == 3 || txm == 4 || txm == 8 || txm == 9) ? 1u : 0u);
    const uint16_t *tab = tabs + (size_t)combo * 3 * K;
    const uint32_t Nnn = nnn[combo], E = g.e_len[a];
    const int8_t  *e = g.e_base + (size_t)g.e_off[a] * 64;
    int8_t        *db = d_soft + (size_t)cb * 3 * D;
    const bool     staged = E + 16 <= e_cap && E < 0xFFFFu;
    if (staged) {
        const uint32_t nq = (E + 15) >> 4; // the allocation's slot is padded to 64 bytes
        for (uint32_t w = threadIdx.x; w < nq; w += blockDim.x) reinterpret_cast<uint4 *>(e_lds)[w] = reinterpret_cast<const uint4 *>(e)[w];
        __syncthreads();
        for (uint32_t k = E + threadIdx.x; k < ((E + 16) & ~15u); k += blockDim.x) e_lds[k] = 0; // the zero behind the last soft bit
    }
    __syncthreads();
    auto sum_at = [&](uint32_t k) -> int { // sum over the laps of the circular buffer, saturated to int8 (first visit stores, repeats add: :11402-11416)
        int v = 0;
        if (staged) {
            v = e_lds[min(k, E)];
            for (uint32_t q = k + Nnn; q < E; q += Nnn) v += e_lds[q];
        } else if (k != 0xFFFFu) {
            for (; k < E; k += Nnn) v += e[k];
        }
        return max(-127, min(127, v));
    };
    // K is a multiple of 8: groups of four positions, 12 output bytes at byte offset 12 * (i / 4), 4-byte aligned (3 D is a multiple of 4)
    for (uint32_t i0 = 4 * threadIdx.x; i0 < K; i0 += 4 * blockDim.x) {
        uint32_t r[3][4];
#pragma unroll
        for (int x = 0; x < 3; x++) {
            const uint2 w = *reinterpret_cast<const uint2 *>(tab + (size_t)x * K + i0);
            r[x][0] = w.x & 0xFFFFu; r[x][1] = w.x >> 16; r[x][2] = w.y & 0xFFFFu; r[x][3] = w.y >> 16;
        }
        uint32_t o[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 12; j++) { // output byte j of the group = d[(i0 + j / 3) * 3 + j % 3]
            const int v = sum_at(r[j % 3][j / 3]);
            o[j >> 2] |= ((uint32_t)v & 0xFFu) << (8 * (j & 3));
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(db + 3 * (size_t)i0);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
    if (threadIdx.x < 12) { // the termination values d[K..K+3][x]
        const uint32_t t = 3 * K + threadIdx.x, i = t / 3;
        const int      x = (int)(t - 3 * i);
        const RmGeom  &rm = rm_s;
        const uint32_t p = rm.pos(i, x);
        uint32_t       k = 0xFFFFu;
        if (p < rm.N_cb) { const uint32_t c = rm.cnt(p); k = (p >= rm.k0m) ? c - rm.cnt_k0 : rm.Nnn - rm.cnt_k0 + c; }
        int v = 0;
        if (k != 0xFFFFu)
            for (; k < E; k += Nnn) v += e[k];
        db[t] = (int8_t)max(-127, min(127, v));
    }
}

// The BCJR chain's first stage in one kernel: turbo rate un-matching (the REF path's rank-table gather, sums of repeats saturated to +-127) straight
// into the max-log-MAP decoder's granule arrays (bcjr.hip: S1 P1 S2 P2, 16 steps = 16 bytes per lane) and termination records -- what k_rm_to_i8
// followed by k_bcjr_prep did with the interleaved int8 block written to HBM and read back in between.  One workgroup per code block, thread
// = 16 trellis steps, as in k_turbo_prep.  Dynamic LDS: staged e [e_cap] | S1 [Kp]; no static LDS (the gather addresses LDS from 0).
__global__ __launch_bounds__(384) void k_rm_bcjr_prep(SrcRateUnmatch src, uint32_t K, uint32_t n_cb, const uint16_t *__restrict__ pi, MiBcjrBufs B)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smr[];
    const uint32_t cb = xcd_cb(blockIdx.x, n_cb), tile = cb >> 6, lane = cb & 63, Kp = kpad64(K), n_units = Kp >> 4, u = threadIdx.x;
    if (cb >= n_cb) return; // uniform
    int8_t *s1_lds = smr + src.e_cap;
    src.init(cb, K);
    const bool e_in_lds = src.stage_e(smr);
    const int  nv = (u < n_units) ? min(16, max(0, (int)K - 16 * (int)u)) : -1;
    int v[3][16];
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int k = 0; k < 16; k++) v[x][k] = 0;
    if (nv > 0) src.load16(u, nv, v, 0u, e_in_lds);
    uint4 Q[3];
#pragma unroll
    for (int x = 0; x < 3; x++) {
        int q[16];
#pragma unroll
        for (int k = 0; k < 16; k++) q[k] = max(-127, min(127, v[x][k])); // what the decoder takes: int8, +-127
        Q[x] = pack16(q);
    }
    const size_t o8 = (size_t)tile * Kp * 64 + (size_t)u * 1024 + lane * 16; // granule u of this lane (bcjr.hip g8)
    if (nv >= 0) {
        *reinterpret_cast<uint4 *>(B.S1 + o8) = Q[0];
        *reinterpret_cast<uint4 *>(B.P1 + o8) = Q[1];
        *reinterpret_cast<uint4 *>(B.P2 + o8) = Q[2];
        *reinterpret_cast<uint4 *>(s1_lds + 16 * u) = Q[0];
    }
    if (u < 12) { // the termination values d[K..K+3][x] (the rank tables stop at K): element 3K + u -> x[u] of k_bcjr_prep's record
        const uint32_t a = src.g.desc[cb].alloc, txm = src.g.allocs[a].tx_mode, rv = src.g.allocs[a].rv_idx & 3u;
        RmGeom rm;
        if (src.g.ul) rm.init(K + 4, 1, rv, 1, 1, 1, false);
        else          rm.init(K + 4, txm, rv);
        const uint32_t i = K + u / 3;
        const int      x = (int)(u % 3);
        const uint32_t p = rm.pos(i, x);
        int            t = 0;
        if (p < rm.N_cb) {
            const uint32_t c = rm.cnt(p);
            for (uint32_t k = (p >= rm.k0m) ? c - rm.cnt_k0 : rm.Nnn - rm.cnt_k0 + c; k < src.E; k += src.Nnn) t += src.e[k];
        }
        const uint32_t slot = u < 6 ? ((u & 1u) ? 3 + (u >> 1) : (u >> 1)) : ((u & 1u) ? 9 + ((u - 7) >> 1) : 6 + ((u - 6) >> 1)); // t1s t1p t2s t2p
        B.tail[((size_t)cb << 4) + slot] = (int8_t)max(-127, min(127, t));
    }
    __syncthreads();
    if (nv >= 0) { // S2[i] = S1[pi[i]]
        uint32_t s2[4] = {0, 0, 0, 0};
        if (nv > 0) {
            uint32_t idx[16];
            load_idx16(pi, u, nv, idx, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) s2[k >> 2] |= ((k < nv) ? (uint32_t)(uint8_t)s1_lds[idx[k]] : 0u) << (8 * (k & 3));
        }
        *reinterpret_cast<uint4 *>(B.S2 + o8) = make_uint4(s2[0], s2[1], s2[2], s2[3]);
    }
}

__global__ __launch_bounds__(256) void k_crc_finish(const uint8_t *__restrict__ c_bits, uint32_t K, uint32_t n_cb, GroupDesc g)
{
    __shared__ uint32_t red[4];
    // (the block's allocation and size from its descriptor where k_cb_desc has written one: slot -> allocation -> its fields is two
    // dependent round trips at the start of a workgroup that lives for little more than three)
    uint32_t alloc, tbs;
    if (g.desc) { const uint4 d0 = reinterpret_cast<const uint4 *>(g.desc + blockIdx.x)[0], d1 = reinterpret_cast<const uint4 *>(g.desc + blockIdx.x)[1]; alloc = d0.x; tbs = d1.z; }
    else        { alloc = g.cb_alloc[blockIdx.x]; tbs = g.allocs[alloc].tbs; }
    const uint32_t cb = blockIdx.x, F = K - tbs - 24;
    const uint8_t *c = c_bits + (size_t)cb * K;
    uint8_t       *o = g.out_bits + (size_t)alloc * g.out_stride;
    uint32_t crc = 0;
    if (((F | tbs) & 7u) == 0) {
        // eight positions per thread and step: their bits as one 8-byte read, the weight of the LAST of them from the table and the other
        // seven by "times x" (k_turbo_vote's scheme: the table is the same 13-24 KB for every workgroup, read per bit it was most of this kernel),
        // the transport block's bytes (or its packed byte) as one store
        const uint32_t GS = 0x864CFBu << 8; // g without its x^24 term, in the table's left-aligned form
        for (uint32_t g8 = (F >> 3) + threadIdx.x; g8 < (K >> 3); g8 += blockDim.x) {
            const uint2    bb = *reinterpret_cast<const uint2 *>(c + 8 * g8); // c is 8-byte aligned: K is a multiple of 8
            const uint32_t lo = bb.x & 0x01010101u, hi = bb.y & 0x01010101u;   // positions 8 g8 .. + 3 | + 4 .. + 7, one bit per byte
            uint32_t       w  = g.crc_tab[MI_CRC_TAB_BIAS + K - 8 * g8 - 8];   // position 8 g8 + 7 weighs x^(K - 8 g8 - 8)
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                const uint32_t m = 0u - (((k < 4 ? lo : hi) >> (8 * (k & 3))) & 1u);
                crc ^= w & m;
                if (k > 0) w = (w << 1) ^ ((uint32_t)((int)w >> 31) & GS);
            }
            const uint32_t j = 8 * g8 - F; // position in the transport block
            if (j < tbs) {
                if (g.packed) o[j >> 3] = (uint8_t)((((lo * 0x08040201u) >> 24) & 0xFu) << 4 | (((hi * 0x08040201u) >> 24) & 0xFu)); // first bit most significant
                else          *reinterpret_cast<uint2 *>(o + j) = make_uint2(lo, hi); // out_stride and F are multiples of 8
            }
        }
    } else {
        for (uint32_t j = F + threadIdx.x; j < K; j += blockDim.x) {
            const uint32_t b = c[j] & 1u;
            crc ^= b ? g.crc_tab[MI_CRC_TAB_BIAS + K - 1 - j] : 0u; // bit j weighs x^(K-1-j) mod gCRC24A; the check is "XOR of the weights == 0"
            if (j < F + tbs && !g.packed) o[j - F] = (uint8_t)b;
        }
        if (g.packed)
            for (uint32_t m = threadIdx.x; m < (tbs + 7) >> 3; m += blockDim.x) {
                uint32_t v = 0;
                for (uint32_t b = 0; b < 8; b++) v = v << 1 | ((8 * m + b < tbs) ? (c[F + 8 * m + b] & 1u) : 0u);
                o[m] = (uint8_t)v;
            }
    }
    crc = wave_xor_u(crc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = crc;
    __syncthreads();
    if (threadIdx.x == 0) g.status[alloc] = ((red[0] ^ red[1] ^ red[2] ^ red[3]) == 0) ? 0 : 2;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host side

namespace {
constexpr int N_BYTE_ARRAYS = 11; // X0 X1 X2 I0 M1 M2 A1 I1 M3 B1 B2
enum { AX0, AX1, AX2, AI0, AM1, AM2, AA1, AI1, AM3, AB1, AB2 };
} // namespace

extern "C" size_t mi_lte_turbo_scratch_bytes(uint32_t K, uint32_t n_cb)
{
    const size_t n_tiles = (n_cb + 63) / 64, Kp = kpad64(K);
    return n_tiles * Kp * 64 * N_BYTE_ARRAYS + 3 * n_tiles * Kp * 32 + n_tiles * 64 * sizeof(CbDesc);
}

static bool no_static_lds(const void *kernel)
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, kernel) == hipSuccess && a.sharedSizeBytes == 0;
}

// The five launches of one REF decode over n_cb code blocks of size K.
template <typename Src, bool GROUP>
static int turbo_ref_run(mi_lte_ctx *ctx, Src src, uint32_t K, uint32_t n_cb, uint8_t *d_c_bits, GroupDesc gd, uint32_t e_cap = 0)
{
    TurboTables tb;
    int         rc = mi_ctx_turbo_tables(ctx, K, 0, &tb);
    if (rc != MI_LTE_OK) return rc;
    // the per-code-block kernels address their dynamic LDS block from 0 (lds_u8 and friends): that holds while they own no static LDS
    static const bool lds_ok = no_static_lds((const void *)k_turbo_prep<Src, 1>) && no_static_lds((const void *)k_turbo_perm<1>) &&
                               no_static_lds((const void *)k_turbo_vote<GROUP, 1>);
    if (!lds_ok) {
        ctx->err = "turbo kernels were built with static LDS: absolute LDS addressing is invalid";
        return MI_LTE_ERR_HIP;
    }
    const size_t n_tiles = (n_cb + 63) / 64, Kp = kpad64(K), arr_bytes = n_tiles * Kp * 64, dec_bytes = n_tiles * Kp * 32;
    rc = mi_ctx_reserve_scratch(ctx, mi_lte_turbo_scratch_bytes(K, n_cb));
    if (rc != MI_LTE_OK) return rc;
    uint8_t *base = (uint8_t *)ctx->scratch;
    uint8_t *arr[N_BYTE_ARRAYS];
    for (int a = 0; a < N_BYTE_ARRAYS; a++) arr[a] = base + a * arr_bytes;
    uint32_t *dec[3];
    for (int p = 0; p < 3; p++) dec[p] = (uint32_t *)(base + N_BYTE_ARRAYS * arr_bytes + p * dec_bytes);
    const bool small = n_cb <= ctx->siso_small_max; // a handful of code blocks (a per-call caller's transport block): k_turbo_siso_small
    if (n_cb % 64 && !small) // lanes past the batch end walk whatever the scratch holds; keep it defined (the state-parallel kernel has no such lanes)
        MI_HIP_CHECK(ctx, hipMemsetAsync(base, 0, N_BYTE_ARRAYS * arr_bytes, ctx->stream));

    if constexpr (GROUP) { // the per-block descriptors behind the tile arrays and the traceback words
        CbDesc *d_desc = (CbDesc *)(base + N_BYTE_ARRAYS * arr_bytes + 3 * dec_bytes);
        gd.desc = src.g.desc = d_desc;
        MI_LAUNCH(ctx, "k_cb_desc", k_cb_desc, dim3((n_cb + 255) / 256), dim3(256), 0, gd, n_cb, (const uint32_t *)src.nnn, d_desc);
    }
    PrepOut po;
    po.arr[0] = arr[AX0]; po.arr[1] = arr[AX1]; po.arr[2] = arr[AX2];
    po.arr[3] = arr[AI0]; po.arr[4] = arr[AM1]; po.arr[5] = arr[AM2];
    const uint32_t cb_threads = (uint32_t)(((Kp >> 4) + 63) & ~(size_t)63); // one thread per 16-step unit: 64..384
    MI_LAUNCH(ctx, "k_turbo_prep", (k_turbo_prep<Src, 1>), dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), prep_lds_bytes(Kp, e_cap), src, K, n_cb, tb.d_pi, po, MultiArgs{});

    SisoArgs s1;
    s1.p[0] = {arr[AX1], arr[AX0], arr[AM1], arr[AA1], dec[0]};
    s1.p[1] = s1.p[0];
    // a handful of code blocks: states on the lanes instead of code blocks
    auto gpw_of = [](uint32_t n_tr) { return n_tr <= 2048 ? 1u : n_tr <= 4096 ? 2u : n_tr <= 8192 ? 4u : SMALL_G; }; // one or two wavefronts per SIMD
    auto lds_of = [&](uint32_t gpw) { return sizeof(uint32_t) * gpw * (64 * 4 * 2 + (Kp >> 5) * 4); };
    if (small)
        MI_LAUNCH(ctx, "k_turbo_siso_small", k_turbo_siso_small<false>, dim3((n_cb + gpw_of(n_cb) - 1) / gpw_of(n_cb)), dim3(64), lds_of(gpw_of(n_cb)), s1, K, n_cb, 0u,
                  gpw_of(n_cb), MultiArgs{});
    else
        MI_LAUNCH(ctx, "k_turbo_siso", k_turbo_siso<false>, dim3(((n_tiles + 1) / 2 + 3) / 4), dim3(256), 0, s1, K, (uint32_t)n_tiles, 0u, MultiArgs{}, (const uint32_t *)nullptr); // two tiles per lane

    PermArgs pa;
    pa.A1 = arr[AA1]; pa.X2 = arr[AX2]; pa.out[0] = arr[AI1]; pa.out[1] = arr[AM3];
    const uint32_t perm_grid = ((8 * xcd_chunk(n_cb) + PERM_NB - 1) / PERM_NB + 7u) & ~7u; // a multiple of 8: b + i * grid stays on b's XCD
    MI_LAUNCH(ctx, "k_turbo_perm", k_turbo_perm<1>, dim3(perm_grid), dim3(cb_threads), MTAB_N + Kp + 32, pa, K, n_cb, tb.d_pi, MultiArgs{});

    SisoArgs s23;
    s23.p[0] = {arr[AX2], arr[AI0], arr[AM2], arr[AB1], dec[1]};
    s23.p[1] = {arr[AX2], arr[AI1], arr[AM3], arr[AB2], dec[2]};
    if (small)
        MI_LAUNCH(ctx, "k_turbo_siso_small", k_turbo_siso_small<false>, dim3((2 * n_cb + gpw_of(2 * n_cb) - 1) / gpw_of(2 * n_cb)), dim3(64), lds_of(gpw_of(2 * n_cb)), s23, K,
                  n_cb, 1u, gpw_of(2 * n_cb), MultiArgs{});
    else
        MI_LAUNCH(ctx, "k_turbo_siso", k_turbo_siso<false>, dim3((n_tiles + 3) / 4), dim3(256), 0, s23, K, (uint32_t)n_tiles, 1u, MultiArgs{}, (const uint32_t *)nullptr); // passes 2 and 3 of a tile per lane

    VoteArgs va = {arr[AX0], arr[AA1], arr[AB1], arr[AB2]};
    MI_LAUNCH(ctx, "k_turbo_vote", (k_turbo_vote<GROUP, 1>), dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), 3 * Kp + 64, va, K, n_cb, tb.d_inv2, d_c_bits, gd, MultiArgs{});
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_turbo_prep:1,k_turbo_siso:2,k_turbo_perm:1,k_turbo_vote:1";
    return MI_LTE_OK;
}

template <typename T>
static int turbo_ref_batch(mi_lte_ctx *ctx, const T *d_soft, uint32_t K, uint32_t n_cb, uint8_t *d_c_bits)
{
    SrcDirect<T> src{0, d_soft, nullptr};
    GroupDesc    none{};
    return turbo_ref_run<SrcDirect<T>, false>(ctx, src, K, n_cb, d_c_bits, none);
}

// per-K rank tables of the fused rate un-matching, built on first use and cached in the context
static int rm_rank_tables(mi_lte_ctx *ctx, uint32_t K, RmTables *out)
{
    auto it = ctx->rm_tables.find(K);
    if (it == ctx->rm_tables.end()) {
        RmTables t;
        MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_tabs, sizeof(uint16_t) * 12 * 3 * K));
        MI_HIP_CHECK(ctx, hipMalloc((void **)&t.d_nnn, sizeof(uint32_t) * 12));
        ctx->owned.push_back(t.d_tabs);
        ctx->owned.push_back(t.d_nnn);
        MI_LAUNCH(ctx, "k_rm_rank_table", k_rm_rank_table, dim3((3 * K + 255) / 256, 12), dim3(256), 0, K, t.d_tabs, t.d_nnn);
        MI_HIP_CHECK(ctx, hipGetLastError());
        it = ctx->rm_tables.emplace(K, t).first;
    }
    *out = it->second;
    return MI_LTE_OK;
}

// used by the PDSCH chain (chain.hip): decode the code blocks of one size K straight from the
// demodulator's soft bits
int mi_turbo_ref_group(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, const mi_lte_pdsch_alloc *d_allocs,
                       const uint32_t *d_cb_alloc, const int8_t *d_e, const uint32_t *d_e_off, const uint32_t *d_e_len,
                       uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status, uint32_t e_max_bytes, bool ul, bool packed)
{
    int rc = mi_ctx_crc_table(ctx);
    if (rc != MI_LTE_OK) return rc;
    GroupDesc gd{d_allocs, d_cb_alloc, d_e, d_e_off, d_e_len, d_out_bits, out_stride, d_status, ctx->d_crc_tab, ul ? 1u : 0u, packed ? 1u : 0u};
    RmTables rt;
    rc = rm_rank_tables(ctx, K, &rt);
    if (rc != MI_LTE_OK) return rc;
    SrcRateUnmatch src;
    src.g    = gd;
    src.tabs = rt.d_tabs;
    src.nnn  = rt.d_nnn;
    // stage e in LDS when the largest allocation of the group fits next to the block's own arrays
    const uint32_t cap = (e_max_bytes + 16u + 63u) & ~63u; // room for the zero slot behind the longest allocation
    src.e_cap          = (prep_lds_bytes(kpad64(K), cap) <= 48 * 1024) ? cap : 0;
    // every stream has at most 31 NULL slots, so a lap of the circular buffer consumes at least 3K - 81 soft bits: while the longest
    // allocation makes no more than 258 laps no sum of int8 values leaves int16, and the kernel may keep them in pairs
    const uint32_t laps = (e_max_bytes + (3 * K - 81) - 1) / (3 * K - 81);
    if (laps <= 258) {
        SrcRateUnmatchPk pk;
        static_cast<SrcRateUnmatch &>(pk) = src;
        return turbo_ref_run<SrcRateUnmatchPk, true>(ctx, pk, K, n_cb, nullptr, gd, src.e_cap);
    }
    return turbo_ref_run<SrcRateUnmatch, true>(ctx, src, K, n_cb, nullptr, gd, src.e_cap);
}

// Can the block-size group join a merged decode?  The merged kernels keep the rate un-matching sums as int16 pairs (SrcRateUnmatchPk): a
// group whose longest allocation makes more than 258 laps of the circular buffer takes the per-size path with 32-bit sums instead.
bool mi_turbo_ref_multi_takes(uint32_t K, uint32_t e_max_bytes) { return (e_max_bytes + (3 * K - 81) - 1) / (3 * K - 81) <= 258; }

// A whole PDSCH batch -- the code blocks of MANY sizes -- through the REF decoder with every kernel launched once (the per-code-block
// kernels once per workgroup width): see KSeg.  `groups` in ascending K, cb_base = the group's first slot in d_cb_alloc.  The tables the
// kernels read are rebuilt only when the groups differ from the ones `cache` was built for (a static plan: once; a dynamic plan: per
// assignment).
int mi_turbo_ref_multi(mi_lte_ctx *ctx, const MiKGroup *groups, uint32_t n_groups, const mi_lte_pdsch_alloc *d_allocs, const uint32_t *d_cb_alloc,
                       const int8_t *d_e, const uint32_t *d_e_off, const uint32_t *d_e_len, uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status,
                       bool ul, bool packed, MiMultiCache *cache)
{
    if (!groups || n_groups == 0 || n_groups > 0xFFFF || !cache) return MI_LTE_ERR_INVALID_ARG;
    int rc = mi_ctx_crc_table(ctx);
    if (rc != MI_LTE_OK) return rc;
    static const bool lds_ok = no_static_lds((const void *)k_turbo_prep<SrcRateUnmatchPk, 1, true>) && no_static_lds((const void *)k_turbo_perm<1, true>) &&
                               no_static_lds((const void *)k_turbo_vote<true, 1, true>);
    if (!lds_ok) { ctx->err = "turbo kernels were built with static LDS: absolute LDS addressing is invalid"; return MI_LTE_ERR_HIP; }
    constexpr int NCLS = 6; // workgroup widths 64 .. 384: one thread per 16-step unit
    auto cls_of = [](uint32_t K) { return (int)((((kpad64(K) >> 4) + 63) >> 6) - 1); };
    const bool same = cache->built_for.size() == n_groups && memcmp(cache->built_for.data(), groups, sizeof(MiKGroup) * n_groups) == 0;
    if (!same) {
        std::vector<KSeg>     segs(n_groups);
        std::vector<uint32_t> map;
        MiMultiGeom          &G = cache->geom;
        G = MiMultiGeom{};
        uint64_t arr = 0;
        uint32_t cbs = 0;
        for (uint32_t i = 0; i < n_groups; i++) {
            const MiKGroup &gr = groups[i];
            if ((i && gr.K <= groups[i - 1].K) || gr.n_cb == 0) { ctx->err = "merged decode: groups must be non-empty and in ascending block size"; return MI_LTE_ERR_INVALID_ARG; }
            TurboTables tb;
            RmTables    rt;
            if ((rc = mi_ctx_turbo_tables(ctx, gr.K, 0, &tb)) != MI_LTE_OK || (rc = rm_rank_tables(ctx, gr.K, &rt)) != MI_LTE_OK) return rc;
            KSeg &sg = segs[i];
            memset(&sg, 0, sizeof(sg));
            const uint32_t Kp = kpad64(gr.K), cap = (gr.e_max + 16u + 63u) & ~63u;
            sg.K = gr.K; sg.n_cb = gr.n_cb; sg.cb_base = gr.cb_base; sg.n_tiles = (gr.n_cb + 63) / 64;
            // LDS for an allocation's soft bits: room for the size's longest allocation, but no more than a lap and a quarter of the circular buffer
            // (3 (K + 4) positions) -- an allocation beyond that is staged lap by lap (gather_windowed_pk), and one repeated allocation of a size
            // no longer sets the occupancy of every workgroup of its width (width 64 of the mixed batch: 16 KB -> 9.7 KB per workgroup).
            // Only where the LDS is what limits the occupancy -- the 64-thread width, K <= 1024, one wavefront per workgroup: 1.02 -> 0.80 ms of the mixed
            // batch's prep; applied to every width it cost the 128- and 192-thread ones 0.05 and 0.11 ms (their blocks beyond a lap and a quarter pay
            // two barriers per lap and their occupancy is bound by registers anyway), gpurun_out/windowed.log
            const uint32_t cap_w = (uint32_t)((15 * (size_t)(gr.K + 4) / 4 + 64 + 63) & ~(size_t)63);
            sg.e_cap = (prep_lds_bytes(Kp, cap) <= 48 * 1024) ? (Kp <= 1024 ? std::min(cap, cap_w) : cap) : cap_w;
            sg.arr_off = arr;
            typedef __attribute__((address_space(1))) const uint16_t gl16_t;
            typedef __attribute__((address_space(1))) const uint32_t gl32_t;
            sg.pi = (gl16_t *)tb.d_pi; sg.inv2 = (gl16_t *)tb.d_inv2; sg.tabs = (gl16_t *)rt.d_tabs; sg.nnn = (gl32_t *)rt.d_nnn;
            arr += (uint64_t)sg.n_tiles * Kp * 64;
            cbs = std::max(cbs, gr.cb_base + gr.n_cb);
            const int c = cls_of(gr.K);
            // (a size whose last tile is partly filled stays with the table kernel: it zeroes the idle lanes the trellis kernel will walk)
            G.one_size[c] = (G.lds_prep[c] == 0 && gr.n_cb % 64 == 0) ? (int)i : -1; // the width's only size so far, or not the only one
            G.off_one[c] = sg.arr_off; G.e_cap_one[c] = sg.e_cap;
            G.lds_prep[c] = std::max(G.lds_prep[c], prep_lds_bytes(Kp, sg.e_cap));
            G.kp_max[c]   = std::max(G.kp_max[c], Kp);
        }
        G.arr_bytes = arr;
        G.n_slots   = cbs;
        // workgroup -> size maps of the per-code-block kernels, one entry per 512 (prep, vote) / 128 (perm) workgroups -- a few KB: they stay in the scalar cache --, class after class
        for (int c = 0; c < NCLS; c++) {
            G.map_cb[c] = (uint32_t)map.size();
            uint32_t wg = 0;
            for (uint32_t i = 0; i < n_groups; i++)
                if (cls_of(groups[i].K) == c) {
                    segs[i].wg_cb = wg;
                    const uint32_t g = 8 * xcd_chunk(groups[i].n_cb);
                    map.insert(map.end(), g / 512, i);
                    wg += g;
                }
            G.grid_cb[c] = wg;
        }
        for (int c = 0; c < NCLS; c++) {
            G.map_perm[c] = (uint32_t)map.size();
            uint32_t wg = 0;
            for (uint32_t i = 0; i < n_groups; i++)
                if (cls_of(groups[i].K) == c) {
                    segs[i].wg_perm   = wg;
                    segs[i].perm_grid = ((8 * xcd_chunk(groups[i].n_cb) + PERM_NB - 1) / PERM_NB + 7u) & ~7u; // a multiple of 8: b + i * grid stays on b's XCD
                    map.insert(map.end(), segs[i].perm_grid / 128, i);
                    wg += segs[i].perm_grid;
                }
            G.grid_perm[c] = wg;
        }
        // wavefront -> size maps of the trellis kernel, the largest sizes first (their walks are the longest: started first, the short ones fill in behind them)
        G.map_wv1 = (uint32_t)map.size();
        uint32_t wv = 0;
        for (uint32_t i = n_groups; i-- > 0;) {
            segs[i].wv1 = wv;
            map.insert(map.end(), (segs[i].n_tiles + 1) / 2, i);
            wv += (segs[i].n_tiles + 1) / 2;
        }
        G.n_wv1   = wv;
        G.map_wv23 = (uint32_t)map.size();
        wv = 0;
        for (uint32_t i = n_groups; i-- > 0;) {
            segs[i].wv23 = wv;
            map.insert(map.end(), segs[i].n_tiles, i);
            wv += segs[i].n_tiles;
        }
        G.n_wv23 = wv;
        // The order the trellis kernel's workgroups are LAUNCHED in.  A walk is as long as its block size, the device holds 1024 workgroups of four
        // walks at a time, and a mixed batch has little more than that (pass 1) or twice that (passes 2 + 3): with the longest walks simply first,
        // a compute unit's four resident workgroups are neighbours in the sorted order and the unit that got the four longest decides when the
        // launch ends.  Dealt out in rounds of 256 (one workgroup per compute unit and round), every other round backwards, each unit gets the
        // r-th longest of one round with the r-th shortest of the next: equal sums (0.07 ms of the mixed batch's 3.6; a scatter that gives up
        // "longest first" costs 1.3).  Batches of a few sizes (W4) keep the sorted order.  (MI_LTE_SISO_ORDER=0 / 1: A/B.)
        // And how many of them a compute unit holds at a time.  The registers allow four (16 walks per unit, 4096 in all): right for W4, whose
        // walks are equally long and come in more than two rounds of that.  A mixed batch of this size has 1.2 rounds (pass 1) and 2.3 (passes
        // 2 + 3) of walks between 44 and 4612 steps: the units that drew short ones run dry and nothing is left to hand them.  Half as many
        // resident workgroups are twice as many rounds -- the queue stays non-empty until close to the end -- at the price of fewer wavefronts to
        // hide latency behind; measured on the mixed batch (gpurun_out/occ*.log): 2 per unit for pass 1 and 3 for passes 2 + 3 take 0.12-0.15 ms
        // off the trellis kernel's 3.6, one per unit costs 0.15.  The limit is set with dynamic LDS that the kernel never touches.
        G.n_ord1 = G.n_ord23 = 0;
        G.siso_pad1 = G.siso_pad23 = 0;
        const bool many_sizes = n_groups >= 8;
        if (many_sizes) { G.siso_pad1 = 60000; G.siso_pad23 = 45000; } // (+ the kernel's own 8 KB: two / three of them in a unit's 160 KB)
        if (getenv("MI_LTE_SISO1_LDS")) G.siso_pad1 = (uint32_t)atoi(getenv("MI_LTE_SISO1_LDS"));
        if (getenv("MI_LTE_SISO23_LDS")) G.siso_pad23 = (uint32_t)atoi(getenv("MI_LTE_SISO23_LDS"));
        {
            const char *eo = getenv("MI_LTE_SISO_ORDER");
            const int   mode = eo ? atoi(eo) : (many_sizes ? 1 : 0);
            auto deal = [&](uint32_t n_wv, uint32_t *at, uint32_t *n_out) {
                const uint32_t n_wg = (n_wv + 3) / 4;
                if (mode == 0 || n_wg < 512) { *n_out = 0; return; }
                *at = (uint32_t)map.size();
                for (uint32_t j = 0; j < n_wg; j++) {
                    const uint32_t round = j / 256, c = j % 256, in_round = std::min(256u, n_wg - 256 * round);
                    uint32_t       src = j;
                    if (mode == 1 && (round & 1u)) src = 256 * round + (in_round - 1 - std::min(c, in_round - 1));
                    if (mode == 2) { // (a scatter by a stride coprime to the count, for comparison)
                        uint32_t st = (uint32_t)(0.618 * n_wg) | 1u;
                        auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
                        while (gcd(st, n_wg) != 1) st += 2;
                        src = (uint32_t)(((uint64_t)j * st) % n_wg);
                    }
                    for (uint32_t t = 0; t < 4; t++) map.push_back(4 * src + t < n_wv ? 4 * src + t : 0xFFFFFFFFu);
                }
                *n_out = 4 * n_wg;
            };
            deal(G.n_wv1, &G.ord_wv1, &G.n_ord1);
            deal(G.n_wv23, &G.ord_wv23, &G.n_ord23);
        }
        // ... and of the state-parallel trellis kernel (a handful of code blocks in all): workgroup = wavefront = up to gpw trellises of one size
        auto gpw_of = [](uint32_t n_tr) { return n_tr <= 2048 ? 1u : n_tr <= 4096 ? 2u : n_tr <= 8192 ? 4u : SMALL_G; };
        uint32_t tot = 0, kp_all = 0;
        for (uint32_t i = 0; i < n_groups; i++) { tot += groups[i].n_cb; kp_all = std::max(kp_all, kpad64(groups[i].K)); }
        G.gpw1 = gpw_of(tot); G.gpw23 = gpw_of(2 * tot); G.kp_all = kp_all;
        G.map_ws1 = (uint32_t)map.size();
        uint32_t wg = 0;
        for (uint32_t i = 0; i < n_groups; i++) {
            const uint32_t n = (groups[i].n_cb + G.gpw1 - 1) / G.gpw1;
            segs[i].ws1 = wg;
            map.insert(map.end(), n, i);
            wg += n;
        }
        G.n_ws1   = wg;
        G.map_ws23 = (uint32_t)map.size();
        wg = 0;
        for (uint32_t i = 0; i < n_groups; i++) {
            const uint32_t n = (2 * groups[i].n_cb + G.gpw23 - 1) / G.gpw23;
            segs[i].ws23 = wg;
            map.insert(map.end(), n, i);
            wg += n;
        }
        G.n_ws23 = wg;
        const size_t seg_bytes = sizeof(KSeg) * n_groups, map_bytes = (sizeof(uint32_t) * map.size() + 15) & ~(size_t)15, need = seg_bytes + map_bytes;
        if (need > cache->cap) {
            MI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (cache->d_tab) (void)hipFree(cache->d_tab);
            cache->d_tab = nullptr; cache->cap = 0;
            MI_HIP_CHECK(ctx, hipMalloc(&cache->d_tab, need + need / 4));
            cache->cap = need + need / 4;
        }
        std::vector<uint8_t> blob(need, 0);
        memcpy(blob.data(), segs.data(), seg_bytes);
        memcpy(blob.data() + seg_bytes, map.data(), sizeof(uint32_t) * map.size());
        MI_H2D(ctx, cache->d_tab, blob.data(), need); // (waits: the kernels of an earlier run that read the old tables are behind it on the stream)
        G.map_off = seg_bytes;
        cache->built_for.assign(groups, groups + n_groups);
    }
    const MiMultiGeom &G = cache->geom;
    const size_t A = G.arr_bytes, dec_bytes = A / 2;
    rc = mi_ctx_reserve_scratch(ctx, N_BYTE_ARRAYS * A + 3 * dec_bytes + (size_t)((G.n_slots + 63) & ~63u) * sizeof(CbDesc));
    if (rc != MI_LTE_OK) return rc;
    uint8_t *base = (uint8_t *)ctx->scratch;
    uint8_t *arr[N_BYTE_ARRAYS];
    for (int a = 0; a < N_BYTE_ARRAYS; a++) arr[a] = base + a * A;
    uint32_t *dec[3];
    for (int p = 0; p < 3; p++) dec[p] = (uint32_t *)(base + N_BYTE_ARRAYS * A + p * dec_bytes);
    CbDesc *d_desc = (CbDesc *)(base + N_BYTE_ARRAYS * A + 3 * dec_bytes);
    const KSeg     *d_segs = (const KSeg *)cache->d_tab;
    const uint32_t *d_map  = (const uint32_t *)((const uint8_t *)cache->d_tab + G.map_off);

    GroupDesc gd{d_allocs, d_cb_alloc, d_e, d_e_off, d_e_len, d_out_bits, out_stride, d_status, ctx->d_crc_tab, ul ? 1u : 0u, packed ? 1u : 0u};
    gd.desc = d_desc;
    MI_LAUNCH(ctx, "k_cb_desc", k_cb_desc_multi, dim3((G.n_slots + 255) / 256), dim3(256), 0, gd, G.n_slots, d_segs, n_groups, d_desc);
    SrcRateUnmatchPk src;
    src.g = gd; src.tabs = nullptr; src.nnn = nullptr; src.e_cap = 0;
    PrepOut po;
    po.arr[0] = arr[AX0]; po.arr[1] = arr[AX1]; po.arr[2] = arr[AX2];
    po.arr[3] = arr[AI0]; po.arr[4] = arr[AM1]; po.arr[5] = arr[AM2];
    for (int c = 0; c < NCLS; c++)
        if (G.grid_cb[c]) {
            // a width that holds ONE size (W4: each of its two sizes): the per-size kernel with kernel arguments, on the merged layout -- its
            // workgroups start two dependent scalar loads earlier, which is worth 0.2-0.4 ms of W4's 4.1-4.5 (see KSeg)
            if (G.one_size[c] >= 0) {
                const MiKGroup &gr = groups[G.one_size[c]];
                TurboTables tb;
                RmTables    rt;
                if ((rc = mi_ctx_turbo_tables(ctx, gr.K, 0, &tb)) != MI_LTE_OK || (rc = rm_rank_tables(ctx, gr.K, &rt)) != MI_LTE_OK) return rc;
                SrcRateUnmatchPk s1 = src;
                s1.tabs = rt.d_tabs; s1.nnn = rt.d_nnn; s1.g.desc = d_desc + gr.cb_base; s1.e_cap = G.e_cap_one[c];
                PrepOut p1 = po;
                for (int a = 0; a < 6; a++) p1.arr[a] += G.off_one[c];
                MI_LAUNCH(ctx, "k_turbo_prep", (k_turbo_prep<SrcRateUnmatchPk, 1, false>), dim3(G.grid_cb[c]), dim3(64 * (c + 1)), G.lds_prep[c], s1, gr.K, gr.n_cb, (const uint16_t *)tb.d_pi,
                          p1, MultiArgs{});
                continue;
            }
            MI_LAUNCH(ctx, "k_turbo_prep", (k_turbo_prep<SrcRateUnmatchPk, 1, true>), dim3(G.grid_cb[c]), dim3(64 * (c + 1)), G.lds_prep[c], src, 0u, 0u, (const uint16_t *)nullptr, po,
                      (MultiArgs{d_segs, d_map + G.map_cb[c]}));
        }
    SisoArgs s1;
    s1.p[0] = {arr[AX1], arr[AX0], arr[AM1], arr[AA1], dec[0]};
    s1.p[1] = s1.p[0];
    // a handful of code blocks in all (a per-call caller's subframe): the trellis's states on the lanes instead of code blocks, as in the per-size launches
    const bool small = G.n_slots <= ctx->siso_small_max;
    auto lds_small = [&](uint32_t gpw) { return sizeof(uint32_t) * gpw * (64 * 4 * 2 + (G.kp_all >> 5) * 4); };
    if (small)
        MI_LAUNCH(ctx, "k_turbo_siso_small", k_turbo_siso_small<true>, dim3(G.n_ws1), dim3(64), lds_small(G.gpw1), s1, 0u, 0u, 0u, G.gpw1, (MultiArgs{d_segs, d_map + G.map_ws1}));
    else
    MI_LAUNCH(ctx, "k_turbo_siso", k_turbo_siso<true>, dim3(G.n_ord1 ? G.n_ord1 / 4 : (G.n_wv1 + 3) / 4), dim3(256), G.siso_pad1, s1, 0u, G.n_wv1, 0u, (MultiArgs{d_segs, d_map + G.map_wv1}),
              G.n_ord1 ? d_map + G.ord_wv1 : (const uint32_t *)nullptr);
    PermArgs pa;
    pa.A1 = arr[AA1]; pa.X2 = arr[AX2]; pa.out[0] = arr[AI1]; pa.out[1] = arr[AM3];
    for (int c = 0; c < NCLS; c++)
        if (G.grid_perm[c])
            MI_LAUNCH(ctx, "k_turbo_perm", (k_turbo_perm<1, true>), dim3(G.grid_perm[c]), dim3(64 * (c + 1)), MTAB_N + G.kp_max[c] + 32, pa, 0u, 0u, (const uint16_t *)nullptr,
                      (MultiArgs{d_segs, d_map + G.map_perm[c]}));
    SisoArgs s23;
    s23.p[0] = {arr[AX2], arr[AI0], arr[AM2], arr[AB1], dec[1]};
    s23.p[1] = {arr[AX2], arr[AI1], arr[AM3], arr[AB2], dec[2]};
    if (small)
        MI_LAUNCH(ctx, "k_turbo_siso_small", k_turbo_siso_small<true>, dim3(G.n_ws23), dim3(64), lds_small(G.gpw23), s23, 0u, 0u, 1u, G.gpw23, (MultiArgs{d_segs, d_map + G.map_ws23}));
    else
    MI_LAUNCH(ctx, "k_turbo_siso", k_turbo_siso<true>, dim3(G.n_ord23 ? G.n_ord23 / 4 : (G.n_wv23 + 3) / 4), dim3(256), G.siso_pad23, s23, 0u, G.n_wv23, 1u, (MultiArgs{d_segs, d_map + G.map_wv23}),
              G.n_ord23 ? d_map + G.ord_wv23 : (const uint32_t *)nullptr);
    VoteArgs va = {arr[AX0], arr[AA1], arr[AB1], arr[AB2]};
    for (int c = 0; c < NCLS; c++)
        if (G.grid_cb[c])
            MI_LAUNCH(ctx, "k_turbo_vote", (k_turbo_vote<true, 1, true>), dim3(G.grid_cb[c]), dim3(64 * (c + 1)), 3 * G.kp_max[c] + 64, va, 0u, 0u, (const uint16_t *)nullptr,
                      (uint8_t *)nullptr, gd, (MultiArgs{d_segs, d_map + G.map_cb[c]}));
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_cb_desc:1,k_turbo_prep,k_turbo_siso:2,k_turbo_perm,k_turbo_vote per workgroup width over all block sizes";
    return MI_LTE_OK;
}

void mi_multi_cache_free(MiMultiCache *cache)
{
    if (cache && cache->d_tab) (void)hipFree(cache->d_tab);
    if (cache) { cache->d_tab = nullptr; cache->cap = 0; cache->built_for.clear(); }
}

int mi_turbo_bcjr_group(mi_lte_ctx *ctx, uint32_t K, uint32_t n_cb, const mi_lte_pdsch_alloc *d_allocs, const uint32_t *d_cb_alloc, const int8_t *d_e,
                        const uint32_t *d_e_off, const uint32_t *d_e_len, uint8_t *d_out_bits, uint32_t out_stride, int32_t *d_status, bool ul,
                        int8_t *d_soft, uint8_t *d_c_bits, uint32_t n_iter, int qpp_spec, bool packed, uint32_t e_max_bytes, bool block_mode, bool early)
{
    int rc = mi_ctx_crc_table(ctx);
    if (rc != MI_LTE_OK) return rc;
    GroupDesc gd{d_allocs, d_cb_alloc, d_e, d_e_off, d_e_len, d_out_bits, out_stride, d_status, ctx->d_crc_tab, ul ? 1u : 0u, packed ? 1u : 0u};
    RmTables  t;
    rc = rm_rank_tables(ctx, K, &t);
    if (rc != MI_LTE_OK) return rc;
    const uint32_t cap = (e_max_bytes + 16u + 63u) & ~63u, e_cap = cap <= 60 * 1024 ? cap : 0; // stage the allocation in LDS when it fits
    if (block_mode) { // the one-block-per-wavefront kernel takes the interleaved int8 block
        MI_LAUNCH(ctx, "k_rm_to_i8", k_rm_to_i8, dim3(n_cb), dim3(256), e_cap, gd, K, n_cb, (const uint16_t *)t.d_tabs, (const uint32_t *)t.d_nnn, d_soft, e_cap);
        MI_HIP_CHECK(ctx, hipGetLastError());
        rc = mi_turbo_bcjr_block_batch(ctx, d_soft, K, n_cb, n_iter, qpp_spec, d_c_bits);
    } else { // the batch kernels: rate un-matching writes their granule arrays itself (k_rm_bcjr_prep)
        static const bool lds_ok = no_static_lds((const void *)k_rm_bcjr_prep);
        if (!lds_ok) { ctx->err = "k_rm_bcjr_prep was built with static LDS"; return MI_LTE_ERR_HIP; }
        TurboTables tb;
        rc = mi_ctx_turbo_tables(ctx, K, qpp_spec ? 1 : 0, &tb);
        if (rc != MI_LTE_OK) return rc;
        MiBcjrBufs mb;
        rc = mi_turbo_bcjr_begin(ctx, K, n_cb, &mb);
        if (rc != MI_LTE_OK) return rc;
        const size_t   Kp = kpad64(K);
        const uint32_t cb_threads = (uint32_t)(((Kp >> 4) + 63) & ~(size_t)63), e_cap2 = (Kp + cap <= 60 * 1024) ? cap : 0;
        CbDesc *d_desc = (CbDesc *)mb.aux; // the per-block descriptors k_cb_desc writes
        gd.desc = d_desc;
        MI_LAUNCH(ctx, "k_cb_desc", k_cb_desc, dim3((n_cb + 255) / 256), dim3(256), 0, gd, n_cb, (const uint32_t *)t.d_nnn, d_desc);
        SrcRateUnmatch src;
        src.g = gd; src.tabs = t.d_tabs; src.nnn = t.d_nnn; src.e_cap = e_cap2;
        MI_LAUNCH(ctx, "k_rm_bcjr_prep", k_rm_bcjr_prep, dim3(8 * xcd_chunk(n_cb)), dim3(cb_threads), e_cap2 + Kp, src, K, n_cb, (const uint16_t *)tb.d_pi, mb);
        MI_HIP_CHECK(ctx, hipGetLastError());
        rc = mi_turbo_bcjr_iterate(ctx, K, n_cb, n_iter, qpp_spec, d_c_bits, early);
    }
    if (rc != MI_LTE_OK) return rc;
    MI_LAUNCH(ctx, "k_crc_finish", k_crc_finish, dim3(n_cb), dim3(256), 0, (const uint8_t *)d_c_bits, K, n_cb, gd);
    MI_HIP_CHECK(ctx, hipGetLastError());
    return MI_LTE_OK;
}

extern "C" int mi_lte_turbo_decode_batch(mi_lte_ctx *ctx, const void *d_soft, mi_lte_soft_type soft_type, uint32_t K,
                                         uint32_t n_cb, mi_lte_turbo_mode mode, uint32_t n_iter, int qpp_spec,
                                         uint8_t *d_c_bits)
{
    if (!ctx || !d_soft || !d_c_bits || n_cb == 0 || K < 40 || K > 6144) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (mode == MI_LTE_TURBO_REF) {
        (void)n_iter;
        (void)qpp_spec;
        switch (soft_type) {
        case MI_LTE_SOFT_F32: return turbo_ref_batch<float>(ctx, (const float *)d_soft, K, n_cb, d_c_bits);
        case MI_LTE_SOFT_I8:  return turbo_ref_batch<int8_t>(ctx, (const int8_t *)d_soft, K, n_cb, d_c_bits);
        case MI_LTE_SOFT_I16: return turbo_ref_batch<int16_t>(ctx, (const int16_t *)d_soft, K, n_cb, d_c_bits);
        }
        return MI_LTE_ERR_INVALID_ARG;
    }
    if (soft_type != MI_LTE_SOFT_I8) {
        ctx->err = "BCJR mode takes int8 LLRs (MI_LTE_SOFT_I8)";
        return MI_LTE_ERR_UNSUPPORTED;
    }
    if (mode == MI_LTE_TURBO_BCJR_BLOCK) return mi_turbo_bcjr_block_batch(ctx, (const int8_t *)d_soft, K, n_cb, n_iter, qpp_spec, d_c_bits);
    return mi_turbo_bcjr_batch(ctx, (const int8_t *)d_soft, K, n_cb, n_iter, qpp_spec, d_c_bits, mode == MI_LTE_TURBO_BCJR_EARLY);
}

extern "C" int mi_lte_rate_unmatch_turbo_batch(mi_lte_ctx *ctx, const float *d_e_bits, uint32_t N_e_bits, uint32_t D,
                                               uint32_t N_codeblocks, uint32_t tx_mode, uint32_t N_soft, uint32_t M_dl_harq,
                                               uint32_t chan_type, uint32_t rv_idx, uint32_t n_cb, float *d_d_bits)
{
    if (!ctx || !d_e_bits || !d_d_bits || D < 44 || D > 6148 || N_codeblocks == 0 || M_dl_harq == 0 || n_cb == 0 || rv_idx > 3)
        return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    RmParams pr{D, N_e_bits, N_codeblocks, tx_mode, N_soft, M_dl_harq, (chan_type == 0 || chan_type == 1) ? 1u : 0u, rv_idx};
    MI_LAUNCH(ctx, "k_rate_unmatch_f32", k_rate_unmatch_f32, dim3((3 * D + 255) / 256, n_cb), dim3(256), 0, d_e_bits, pr, n_cb, d_d_bits);
    MI_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernels = "k_rate_unmatch_f32:1";
    return MI_LTE_OK;
}
