// Per-call, host-pointer forms of the reference's entry points on the hot path (see include/mi_lte.h).  They exist for drop-in use
// through shim/liblte_phy_shim.cc: stage through HBM, run the same batch kernels with a batch of one, copy back.  No arithmetic
// happens on the host.
//
// A caller of the reference API makes several calls per subframe on the same LIBLTE_PHY_SUBFRAME_STRUCT (get_dl_subframe_and_ce ->
// pdcch_channel_decode -> pdsch_channel_decode per allocation; LTE_fdd_dl_fs_samp_buf.cc:445-515) and has 1 ms per subframe in the
// eNodeB's radio thread (LTE_fdd_enb_phy.cc:429-446).  So nothing here allocates per call: every context owns
//   * pinned host staging and device buffers that grow to the largest call seen and stay,
//   * ONE device subframe, tagged with the host arrays it mirrors and a fingerprint of their contents: the copy
//     get_dl_subframe_and_ce leaves in HBM is what the decoders called next read -- the 300-770 KB upload of the caller's struct only
//     happens when the caller hands in a subframe this context has not produced (or has changed since),
//   * plans (PDSCH, PDCCH, PUSCH, PRACH) cached by everything they were built from, least recently used evicted.
#include <cstring>
#include <list>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace {
constexpr size_t ROW = 16 * 1200; // floats per plane of a subframe struct (LIBLTE_PHY_SUBFRAME_STRUCT rows, liblte_phy.h:226-239)

// the LTE transform lengths and a grid that fits inside them: checked before any size arithmetic is done with fft_size
bool valid_fft(uint32_t fft_size, uint32_t N_rb)
{
    return (fft_size == 128 || fft_size == 256 || fft_size == 512 || fft_size == 1024 || fft_size == 2048) && N_rb >= 6 && N_rb * 12 < fft_size;
}
uint32_t fft_of(uint32_t N_rb) { return N_rb <= 6 ? 128 : N_rb <= 15 ? 256 : N_rb <= 25 ? 512 : N_rb <= 50 ? 1024 : 2048; }

bool prbs_on_carrier(const mi_lte_pdsch_alloc &a, uint32_t N_rb)
{
    if (a.N_prb > 112) return false;
    for (uint32_t s = 0; s < 2; s++)
        for (uint32_t i = 0; i < a.N_prb; i++)
            if (a.prb[s][i] >= N_rb) return false;
    return true;
}

template <typename Plan> struct PlanCache { // key -> plan, most recently used first
    typedef void (*Destroy)(mi_lte_ctx *, Plan *);
    std::list<std::pair<std::string, Plan *>> items;
    Destroy destroy;
    size_t  cap;
    PlanCache(Destroy d, size_t c) : destroy(d), cap(c) {}
    Plan *find(const std::string &k)
    {
        for (auto it = items.begin(); it != items.end(); ++it)
            if (it->first == k) { items.splice(items.begin(), items, it); return items.front().second; }
        return nullptr;
    }
    void put(mi_lte_ctx *ctx, const std::string &k, Plan *p)
    {
        items.emplace_front(k, p);
        while (items.size() > cap) { destroy(ctx, items.back().second); items.pop_back(); }
    }
    void clear(mi_lte_ctx *ctx)
    {
        for (auto &kv : items) destroy(ctx, kv.second);
        items.clear();
    }
};
template <typename T> void key_add(std::string &k, const T &v) { k.append(reinterpret_cast<const char *>(&v), sizeof(T)); }
uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

struct HostCache {
    uint8_t *h_pin = nullptr; size_t h_pin_bytes = 0; // pinned staging
    uint8_t *d_in = nullptr;  size_t d_in_bytes = 0;  // sample staging on the device
    float    *d_sub = nullptr;                        // the device subframe: (2 + 2*4) planes of 16 x 1200 floats
    // Parameters and results of a call live in PINNED HOST memory that the kernels read and write directly (h_* = the host's pointer, d_* = the
    // device's for the same bytes): a dozen bytes of parameters and a few hundred of results per call are not worth a copy command each --
    // hipMemcpyAsync costs ~20 us of API time per call on this platform, five of them per subframe were most of the 1.4 MHz scan's loop.
    // All of these blocks (and the staging buffer) are asked for as COHERENT host memory explicitly: what a kernel wrote must be there
    // when the stream reports it done, whatever HIP_HOST_COHERENT says about the default
    uint32_t *h_par = nullptr, *d_par = nullptr;      // 16 words of per-call parameters
    uint8_t  *h_res = nullptr, *d_res = nullptr;      // one decode's results: verdict at byte 0, decoded bits from byte 64
    uint8_t  *d_out = nullptr;                        // = d_res + 64
    int32_t  *d_st  = nullptr;                        // = d_res
    uint32_t  par_sf = ~0u, par_cell = ~0u;           // what d_par[4..5] hold (subframe number, cell): uploaded only when they change
    // what d_sub mirrors: the caller's rx_symb_re array, the port count its layout was written for, and a fingerprint of the contents
    const float *sub_host = nullptr;
    uint32_t     sub_n_ant = 0;
    bool         sub_ul = false;
    uint64_t     sub_fp = 0;
    uint64_t     n_reuse = 0, n_upload = 0; // statistics (mi_lte_host_cache_stats)
    bool         trust_address = false;     // mi_lte_host_cache_set_mode: the caller announces changes with mi_lte_host_cache_invalidate, nothing is hashed
    // the one-call-per-subframe form's own plan and result block (mi_lte_dl_subframe_decode_host)
    mi_lte_pdsch_plan *sf_plan = nullptr;
    mi_lte_dl_cfg      sf_cfg  = {0, 0, 0, 0};
    uint8_t           *h_sf_res = nullptr, *d_sf_res = nullptr; // MI_LTE_PDCCH_MAX_DCI verdicts at byte 0, the blocks' bits from byte 64 at the plan's stride
    uint8_t           *h_ul_res = nullptr, *d_ul_res = nullptr; // mi_lte_ul_subframe_decode_host: MI_LTE_UL_SUBFRAME_MAX_ALLOC verdicts at byte 0, the blocks' bits from byte 256
    PlanCache<mi_lte_pdsch_plan> pdsch{mi_lte_pdsch_plan_destroy, 64};
    PlanCache<mi_lte_pdcch_plan> pdcch{mi_lte_pdcch_plan_destroy, 8};
    PlanCache<mi_lte_pusch_plan> pusch{mi_lte_pusch_plan_destroy, 64};
    PlanCache<mi_lte_prach_plan> prach{mi_lte_prach_plan_destroy, 4};
};

void host_cache_free(mi_lte_ctx *ctx)
{
    HostCache *hc = (HostCache *)ctx->host_cache;
    if (!hc) return;
    (void)hipStreamSynchronize(ctx->stream);
    hc->pdsch.clear(ctx); hc->pdcch.clear(ctx); hc->pusch.clear(ctx); hc->prach.clear(ctx);
    if (hc->sf_plan) mi_lte_pdsch_plan_destroy(ctx, hc->sf_plan);
    if (hc->h_sf_res) (void)hipHostFree(hc->h_sf_res);
    if (hc->h_ul_res) (void)hipHostFree(hc->h_ul_res);
    if (hc->h_pin) (void)hipHostFree(hc->h_pin);
    (void)hipFree(hc->d_in); (void)hipFree(hc->d_sub);
    if (hc->h_par) (void)hipHostFree(hc->h_par);
    if (hc->h_res) (void)hipHostFree(hc->h_res);
    delete hc;
    ctx->host_cache = nullptr;
}

int host_cache(mi_lte_ctx *ctx, HostCache **out)
{
    if (!ctx->host_cache) {
        HostCache *hc = new HostCache();
        ctx->host_cache      = hc;
        ctx->host_cache_free = host_cache_free;
        auto guard = on_fail([&] { host_cache_free(ctx); }); // a half-built cache is not left behind: the next call starts over
        MI_HIP_CHECK(ctx, hipMalloc((void **)&hc->d_sub, 10 * ROW * sizeof(float)));
        MI_HIP_CHECK(ctx, hipMemsetAsync(hc->d_sub, 0, 10 * ROW * sizeof(float), ctx->stream));
        MI_HIP_CHECK(ctx, hipHostMalloc((void **)&hc->h_par, 64, hipHostMallocMapped | hipHostMallocCoherent));
        MI_HIP_CHECK(ctx, hipHostMalloc((void **)&hc->h_res, 64 + 6144 + 64, hipHostMallocMapped | hipHostMallocCoherent));
        MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&hc->d_par, hc->h_par, 0));
        MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&hc->d_res, hc->h_res, 0));
        memset(hc->h_par, 0, 64);
        hc->d_st  = (int32_t *)hc->d_res;
        hc->d_out = hc->d_res + 64;
        guard.armed = false;
    }
    *out = (HostCache *)ctx->host_cache;
    return MI_LTE_OK;
}
int need_pin(mi_lte_ctx *ctx, HostCache *hc, size_t bytes)
{
    if (bytes <= hc->h_pin_bytes) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    if (hc->h_pin) (void)hipHostFree(hc->h_pin);
    hc->h_pin = nullptr; hc->h_pin_bytes = 0;
    bytes = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    MI_HIP_CHECK(ctx, hipHostMalloc((void **)&hc->h_pin, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    hc->h_pin_bytes = bytes;
    return MI_LTE_OK;
}
int need_dev_in(mi_lte_ctx *ctx, HostCache *hc, size_t bytes)
{
    if (bytes <= hc->d_in_bytes) return MI_LTE_OK;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    (void)hipFree(hc->d_in);
    hc->d_in = nullptr; hc->d_in_bytes = 0;
    bytes = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    MI_HIP_CHECK(ctx, hipMalloc((void **)&hc->d_in, bytes));
    hc->d_in_bytes = bytes;
    return MI_LTE_OK;
}
// n floats of each of two host arrays -> (a | b) where a kernel can read them.  in_place: the pinned staging buffer itself, for a kernel
// that reads every sample once (each crosses the PCIe link once either way, and there is no copy command to pay for); otherwise d_in,
// through one copy, for a kernel that comes back to its samples (the PRACH correlator)
int stage_pair(mi_lte_ctx *ctx, HostCache *hc, const float *h_a, const float *h_b, size_t n, bool in_place, float **d_a, float **d_b)
{
    int rc = need_pin(ctx, hc, 2 * n * 4);
    if (rc == MI_LTE_OK && !in_place) rc = need_dev_in(ctx, hc, 2 * n * 4);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx)); // the staging buffer may still be in use by the previous call
    memcpy(hc->h_pin, h_a, n * 4);
    memcpy(hc->h_pin + n * 4, h_b, n * 4);
    float *d = (float *)hc->d_in;
    if (in_place) {
        MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&d, hc->h_pin, 0));
    } else {
        MI_HIP_CHECK(ctx, mi_pinned_to_device(ctx, hc->d_in, hc->h_pin, 2 * n * 4));
    }
    *d_a = d;
    *d_b = d + n;
    return MI_LTE_OK;
}

// Fingerprint of a host subframe: a 64-bit hash over EVERY value of rows 0-13 of every plane that carries data (four interleaved
// multiply-xor lanes over 8-byte words: ~15 us for a single-port 20 MHz subframe, a third of what uploading it costs).  It answers "is
// this still what the device copy was made from" for callers that rebuild a subframe in place between calls -- the uplink demo does,
// a sparse one at that -- so nothing short of the whole content will do.
uint64_t hash_words(const void *p, size_t n_bytes, uint64_t seed)
{
    const uint64_t *w = (const uint64_t *)p;
    const size_t    n = n_bytes / 8;
    uint64_t a = seed ^ 0x9E3779B97F4A7C15ull, b = seed + 0xC2B2AE3D27D4EB4Full, c = ~seed, d = seed * 0x165667B19E3779F9ull + 1;
    size_t   i = 0;
    for (; i + 4 <= n; i += 4) {
        a = (a ^ w[i]) * 0x9FB21C651E98DF25ull;     a ^= a >> 29;
        b = (b ^ w[i + 1]) * 0xD6E8FEB86659FD93ull; b ^= b >> 31;
        c = (c ^ w[i + 2]) * 0xA0761D6478BD642Full; c ^= c >> 27;
        d = (d ^ w[i + 3]) * 0xE7037ED1A0B428DBull; d ^= d >> 30;
    }
    for (; i < n; i++) { a = (a ^ w[i]) * 0x9FB21C651E98DF25ull; a ^= a >> 29; }
    return (a * 3 + b) ^ (c * 5 + d) ^ ((a ^ c) >> 32);
}
// The same job on AVX2 hosts (chosen at run time): 32 lanes of 32 bits, each lane h = rotl((h ^ w) * odd, 13) over its words.  Every step
// is a bijection of the lane's state, so two inputs that differ in the words of ONE lane always end in different lane states, and the
// fold below adds an injective 64-bit mix of every lane: a change confined to one lane always changes the result; changes spread over
// several lanes collide with probability ~2^-32 per lane.  ~4x the scalar loop's speed (the hash is three of a subframe's ~280 us).
#if defined(__x86_64__)
} // namespace
#include <immintrin.h>
namespace {
__attribute__((target("avx2"))) uint64_t hash_words_avx2(const void *p, size_t n_bytes, uint64_t seed)
{
    const __m256i *w = (const __m256i *)p;
    const size_t   n = n_bytes / 128; // four 32-byte vectors per round
    const __m256i  c0 = _mm256_set1_epi32((int)0x9E3779B1u), c1 = _mm256_set1_epi32((int)0x85EBCA77u), c2 = _mm256_set1_epi32((int)0xC2B2AE3Du),
                   c3 = _mm256_set1_epi32((int)0x27D4EB2Fu);
    __m256i a = _mm256_set1_epi32((int)(uint32_t)seed), b = _mm256_set1_epi32((int)(uint32_t)(seed >> 32)), c = _mm256_set1_epi32(0x165667B1),
            d = _mm256_set1_epi32((int)0xD6E8FEB8u);
#define MI_HASH_STEP(h, k, cst)                                                                     \
    do {                                                                                            \
        const __m256i t_ = _mm256_mullo_epi32(_mm256_xor_si256(h, _mm256_loadu_si256(w + 4 * i + k)), cst); \
        h = _mm256_or_si256(_mm256_slli_epi32(t_, 13), _mm256_srli_epi32(t_, 19));                   \
    } while (0)
    for (size_t i = 0; i < n; i++) {
        MI_HASH_STEP(a, 0, c0);
        MI_HASH_STEP(b, 1, c1);
        MI_HASH_STEP(c, 2, c2);
        MI_HASH_STEP(d, 3, c3);
    }
#undef MI_HASH_STEP
    uint32_t lanes[32];
    _mm256_storeu_si256((__m256i *)lanes, a); _mm256_storeu_si256((__m256i *)(lanes + 8), b);
    _mm256_storeu_si256((__m256i *)(lanes + 16), c); _mm256_storeu_si256((__m256i *)(lanes + 24), d);
    uint64_t h = seed;
    for (uint32_t k = 0; k < 32; k++) { // an injective mix of (lane index, lane state), summed
        uint64_t x = ((uint64_t)(k + 1) << 32) | lanes[k];
        x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
        h += x;
    }
    const size_t done = n * 128;
    return done < n_bytes ? hash_words((const char *)p + done, n_bytes - done, h) : h;
}
bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
// the same construction on 64 lanes where the host has AVX-512 (the rotate is one instruction there): 256 bytes per round
__attribute__((target("avx512f"))) uint64_t hash_words_avx512(const void *p, size_t n_bytes, uint64_t seed)
{
    const __m512i *w = (const __m512i *)p;
    const size_t   n = n_bytes / 256; // four 64-byte vectors per round
    const __m512i  c0 = _mm512_set1_epi32((int)0x9E3779B1u), c1 = _mm512_set1_epi32((int)0x85EBCA77u), c2 = _mm512_set1_epi32((int)0xC2B2AE3Du),
                   c3 = _mm512_set1_epi32((int)0x27D4EB2Fu);
    __m512i a = _mm512_set1_epi32((int)(uint32_t)seed), b = _mm512_set1_epi32((int)(uint32_t)(seed >> 32)), c = _mm512_set1_epi32(0x165667B1),
            d = _mm512_set1_epi32((int)0xD6E8FEB8u);
    for (size_t i = 0; i < n; i++) {
        a = _mm512_rol_epi32(_mm512_mullo_epi32(_mm512_xor_si512(a, _mm512_loadu_si512(w + 4 * i)), c0), 13);
        b = _mm512_rol_epi32(_mm512_mullo_epi32(_mm512_xor_si512(b, _mm512_loadu_si512(w + 4 * i + 1)), c1), 13);
        c = _mm512_rol_epi32(_mm512_mullo_epi32(_mm512_xor_si512(c, _mm512_loadu_si512(w + 4 * i + 2)), c2), 13);
        d = _mm512_rol_epi32(_mm512_mullo_epi32(_mm512_xor_si512(d, _mm512_loadu_si512(w + 4 * i + 3)), c3), 13);
    }
    uint32_t lanes[64];
    _mm512_storeu_si512((void *)lanes, a); _mm512_storeu_si512((void *)(lanes + 16), b);
    _mm512_storeu_si512((void *)(lanes + 32), c); _mm512_storeu_si512((void *)(lanes + 48), d);
    uint64_t h = seed;
    for (uint32_t k = 0; k < 64; k++) { // an injective mix of (lane index, lane state), summed
        uint64_t x = ((uint64_t)(k + 1) << 32) | lanes[k];
        x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
        h += x;
    }
    const size_t done = n * 256;
    return done < n_bytes ? hash_words_avx2((const char *)p + done, n_bytes - done, h) : h;
}
bool have_avx512() { static const bool v = __builtin_cpu_supports("avx512f"); return v; }
#else
uint64_t hash_words_avx2(const void *p, size_t n, uint64_t s) { return hash_words(p, n, s); }
uint64_t hash_words_avx512(const void *p, size_t n, uint64_t s) { return hash_words(p, n, s); }
bool have_avx2() { return false; }
bool have_avx512() { return false; }
#endif
uint64_t hash_plane(const void *p, size_t n_bytes, uint64_t seed)
{
    if (have_avx512() && n_bytes >= 1024) return hash_words_avx512(p, n_bytes, seed);
    return have_avx2() ? hash_words_avx2(p, n_bytes, seed) : hash_words(p, n_bytes, seed);
}

uint64_t subframe_fp(const float *re, const float *im, const float *ce_re, const float *ce_im, uint32_t n_ant, uint32_t n_sc, uint32_t rows)
{
    uint64_t h = ((uint64_t)n_ant << 32) | n_sc;
    // the first n_sc columns of each row are all that the decoders (and the reference's) ever read: a narrow carrier's fingerprint is
    // over those alone, row by row (1.4 MHz: 4 KB a plane instead of 67 KB)
    static thread_local std::vector<float> tmp; // a narrow carrier's rows side by side: one pass (and one fold of the lanes) per plane
    auto plane = [&](const float *p) {
        if (n_sc >= 1200) { h = hash_plane(p, (size_t)rows * 1200 * sizeof(float), h); return; } // rows are contiguous inside a plane
        tmp.resize((size_t)rows * n_sc);
        for (uint32_t r = 0; r < rows; r++) memcpy(tmp.data() + (size_t)r * n_sc, p + r * 1200, n_sc * sizeof(float));
        h = hash_plane(tmp.data(), (size_t)rows * n_sc * sizeof(float), h);
    };
    plane(re);
    plane(im);
    for (uint32_t p = 0; p < n_ant && ce_re; p++) {
        plane(ce_re + p * ROW);
        plane(ce_im + p * ROW);
    }
    return h;
}

// rows of d_sub (1200 wide) -> their first n_sc columns, packed, written straight into pinned host memory: what a narrow carrier's
// get_dl_subframe_and_ce hands back (1.4 MHz: 18 KB instead of 307 KB over the link, and no copy command)
__global__ void k_pack_cols(const float *__restrict__ sub, float *__restrict__ out, uint32_t n_sc)
{
    const float *src = sub + (size_t)blockIdx.x * 1200;
    float       *dst = out + (size_t)blockIdx.x * n_sc;
    for (uint32_t c = threadIdx.x; c < n_sc; c += blockDim.x) dst[c] = src[c];
}
int fetch_packed_rows(mi_lte_ctx *ctx, HostCache *hc, size_t planes, uint32_t n_sc, const float **st)
{
    int rc = need_pin(ctx, hc, planes * 16 * n_sc * 4);
    if (rc != MI_LTE_OK) return rc;
    float *d_pin;
    MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&d_pin, hc->h_pin, 0));
    k_pack_cols<<<dim3((uint32_t)planes * 16), dim3(n_sc >= 256 ? 256 : 64), 0, ctx->stream>>>(hc->d_sub, d_pin, n_sc);
    MI_HIP_CHECK(ctx, hipGetLastError());
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    *st = (const float *)hc->h_pin;
    return MI_LTE_OK;
}
// rows 0..rows-1 of packed plane `plane` -> the first n_sc columns of the caller's 1200-wide rows
void unpack_rows(float *dst, const float *st, size_t plane, uint32_t rows, uint32_t n_sc)
{
    for (uint32_t r = 0; r < rows; r++) memcpy(dst + r * 1200, st + (plane * 16 + r) * n_sc, n_sc * 4);
}

// (subframe number, cell) of the decoders' per-unit arrays at d_par[4], d_par[5]: the three or more decode calls of a subframe share them
int bind_params(mi_lte_ctx *ctx, HostCache *hc, uint32_t subfr_num, uint32_t N_id_cell)
{
    if (hc->par_sf == subfr_num && hc->par_cell == N_id_cell) return MI_LTE_OK;
    hc->h_par[4] = subfr_num; hc->h_par[5] = N_id_cell; // (every per-call form ends with a wait: no kernel of an earlier call is still reading them)
    hc->par_sf = subfr_num; hc->par_cell = N_id_cell;
    return MI_LTE_OK;
}

// An error after work was queued: the per-call forms share pinned parameter / result blocks and one device subframe on the assumption that
// every call ends with a wait, so a failing call must wait too (kernels of its earlier stages may still be reading h_par / h_pin) and
// must not leave the caches claiming that the device holds the caller's data
int fail_after_launch(mi_lte_ctx *ctx, HostCache *hc, int rc)
{
    (void)mi_stream_wait_polling(ctx);
    hc->sub_host = nullptr;
    hc->par_sf = hc->par_cell = ~0u;
    return rc;
}

// make d_sub hold the caller's subframe (downlink layout for n_ant ports, or the two uplink planes): nothing to do when it is the
// copy this context produced (or uploaded) last and the arrays have not changed since
int bind_subframe(mi_lte_ctx *ctx, HostCache *hc, const float *re, const float *im, const float *ce_re, const float *ce_im, uint32_t n_ant, uint32_t n_sc, bool ul)
{
    // explicit contract (mi_lte_host_cache_set_mode): same arrays, same layout, no invalidate since = the device copy is current
    if (hc->trust_address && hc->sub_host == re && hc->sub_n_ant == n_ant && hc->sub_ul == ul) { hc->n_reuse++; return MI_LTE_OK; }
    const uint64_t fp = hc->trust_address ? 0 : subframe_fp(re, im, ce_re, ce_im, ul ? 0 : n_ant, n_sc, 14);
    if (hc->sub_host == re && hc->sub_n_ant == n_ant && hc->sub_ul == ul && hc->sub_fp == fp) { hc->n_reuse++; return MI_LTE_OK; }
    const size_t planes = ul ? 2 : 2 + 2 * (size_t)n_ant, bytes = planes * ROW * 4;
    int rc = need_pin(ctx, hc, bytes);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    float *st = (float *)hc->h_pin;
    memcpy(st, re, ROW * 4);
    memcpy(st + ROW, im, ROW * 4);
    if (!ul) {
        memcpy(st + 2 * ROW, ce_re, n_ant * ROW * 4);
        memcpy(st + (2 + n_ant) * ROW, ce_im, n_ant * ROW * 4);
    }
    MI_HIP_CHECK(ctx, mi_pinned_to_device(ctx, hc->d_sub, st, bytes));
    hc->sub_host = re; hc->sub_n_ant = n_ant; hc->sub_ul = ul; hc->sub_fp = fp;
    hc->n_upload++;
    return MI_LTE_OK;
}
} // namespace

extern "C" {

int mi_lte_host_cache_stats(mi_lte_ctx *ctx, uint64_t *n_subframe_reuse, uint64_t *n_subframe_upload, uint32_t *n_plans)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    HostCache *hc = (HostCache *)ctx->host_cache;
    if (n_subframe_reuse) *n_subframe_reuse = hc ? hc->n_reuse : 0;
    if (n_subframe_upload) *n_subframe_upload = hc ? hc->n_upload : 0;
    if (n_plans) *n_plans = hc ? (uint32_t)(hc->pdsch.items.size() + hc->pdcch.items.size() + hc->pusch.items.size() + hc->prach.items.size()) : 0;
    return MI_LTE_OK;
}
int mi_lte_host_cache_invalidate(mi_lte_ctx *ctx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (ctx->host_cache) ((HostCache *)ctx->host_cache)->sub_host = nullptr;
    return MI_LTE_OK;
}
int mi_lte_host_cache_set_mode(mi_lte_ctx *ctx, uint32_t mode)
{
    if (!ctx || mode > MI_LTE_HOST_CACHE_EXPLICIT) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    hc->trust_address = mode == MI_LTE_HOST_CACHE_EXPLICIT;
    hc->sub_host      = nullptr; // whatever was mirrored under the other rule is not carried over
    return MI_LTE_OK;
}

// liblte_phy_get_dl_subframe_and_ce (liblte_phy.cc:5905-6200), argument checks as at :5937-5943
int mi_lte_get_dl_subframe_and_ce_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q,
                                       uint32_t frame_start_idx, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant,
                                       float *h_symb_re, float *h_symb_im, float *h_ce_re, float *h_ce_im)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !(N_ant == 1 || N_ant == 2 || N_ant == 4) || !h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || !valid_fft(fft_size, N_rb_dl))
        return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    const uint32_t sc = 2048 / fft_size;
    const size_t   per_sf = 30720 / sc, need = per_sf + 2 * fft_size + 160 / sc + 144 / sc - 1; // last sample symbol 15 reads, +1
    const size_t   start = (size_t)frame_start_idx + (size_t)subfr_num * per_sf;
    // the two sample arrays and the unit's parameters (start, subframe number, cell) in one staging buffer
    const size_t nb = 2 * need * 4, total = nb + 16;
    rc = need_pin(ctx, hc, total);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx)); // the staging buffer may still be in use by the previous call
    memcpy(hc->h_pin, h_i + start, need * 4);
    memcpy(hc->h_pin + need * 4, h_q + start, need * 4);
    struct { uint64_t start; uint32_t sf, cell; } par = {0, subfr_num, N_id_cell};
    memcpy(hc->h_pin + nb, &par, sizeof(par));
    // the FFT kernel reads the staging buffer itself (pinned host memory, mapped): each sample crosses the link once either way, and a
    // copy command's API time (~10 us) is a tenth of the whole call at 20 MHz, more on a narrow carrier
    uint8_t *d_src;
    MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&d_src, hc->h_pin, 0));
    float          *d_i = (float *)d_src, *d_q = d_i + need;
    const uint32_t *d_p = (const uint32_t *)(d_src + nb);
    hc->sub_host = nullptr; // d_sub is being rewritten
    mi_lte_dl_cfg cfg = {fft_size, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR | MI_LTE_IQ_ALL_ROWS}; // every row the reference's struct holds
    rc = mi_lte_dl_frontend_batch(ctx, &cfg, d_i, d_q, (const uint64_t *)d_p, d_p + 2, d_p + 3, 1, hc->d_sub);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    const size_t   planes = 2 + 2 * (size_t)N_ant;
    const uint32_t n_sc = 12 * N_rb_dl;
    // the reference writes columns 0..n_sc-1 of each row and nothing else (samples_to_symbols_dl, :8628-8632): so does this form
    const float *st;
    rc = fetch_packed_rows(ctx, hc, planes, n_sc, &st);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    unpack_rows(h_symb_re, st, 0, 16, n_sc);
    unpack_rows(h_symb_im, st, 1, 16, n_sc);
    for (uint32_t p = 0; p < N_ant; p++) { // estimate rows 14 and 15 are never written, as in the reference
        unpack_rows(h_ce_re + p * ROW, st, 2 + p, 14, n_sc);
        unpack_rows(h_ce_im + p * ROW, st, 2 + N_ant + p, 14, n_sc);
    }
    hc->sub_host = h_symb_re; hc->sub_n_ant = N_ant; hc->sub_ul = false;
    hc->sub_fp = hc->trust_address ? 0 : subframe_fp(h_symb_re, h_symb_im, h_ce_re, h_ce_im, N_ant, n_sc, 14);
    return 0;
}

// One subframe, one call: what LTE_fdd_dl_fs_samp_buf.cc:445-515 does with three or more calls of the reference API (get_dl_subframe_and_ce,
// pdcch_channel_decode, pdsch_channel_decode per DCI found) -- the subframe never leaves HBM, nothing is hashed, and the host waits twice
// (for the DCIs, which it needs to lay the PDSCH work out, and for the transport blocks) instead of 2 + N_dci times plus a 300-770 KB copy back.
int mi_lte_dl_subframe_decode_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t frame_start_idx,
                                   uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float phich_res, uint32_t phich_dur_extended, uint32_t flags,
                                   uint32_t *cfi, uint32_t *N_symbs, uint32_t *N_dci, mi_lte_pdcch_dci *dci, uint8_t *h_out_bits, uint32_t out_stride,
                                   uint32_t *N_out_bits, int32_t *status)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !(N_ant == 1 || N_ant == 2 || N_ant == 4) || N_id_cell > 503 || subfr_num > 9 || !cfi || !N_symbs || !N_dci || !dci || !h_out_bits ||
        !N_out_bits || !status || !valid_fft(fft_size, N_rb_dl))
        return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    const mi_lte_dl_cfg cfg = {fft_size, N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
    // plans: the control region's by what it was built from, ONE dynamic PDSCH plan per carrier configuration, re-planned per subframe
    std::string key;
    key_add(key, cfg); key_add(key, phich_res); key_add(key, phich_dur_extended); key_add(key, flags); key_add(key, N_id_cell);
    mi_lte_pdcch_plan *cplan = hc->pdcch.find(key);
    if (!cplan) {
        rc = mi_lte_pdcch_plan_create(ctx, &cfg, phich_res, phich_dur_extended, flags, &N_id_cell, 1, &cplan);
        if (rc != MI_LTE_OK) return rc;
        hc->pdcch.put(ctx, key, cplan);
    }
    if (!hc->sf_plan || memcmp(&hc->sf_cfg, &cfg, sizeof(cfg)) != 0) {
        MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
        if (hc->sf_plan) mi_lte_pdsch_plan_destroy(ctx, hc->sf_plan);
        hc->sf_plan = nullptr;
        const size_t soft = (size_t)MI_LTE_PDCCH_MAX_DCI * (((size_t)13 * N_rb_dl * 12 * 6 + 63) & ~(size_t)63);
        rc = mi_pdsch_plan_create_mapped(ctx, &cfg, MI_LTE_PDCCH_MAX_DCI, soft, &hc->sf_plan); // (descriptors in mapped host memory: no copy commands)
        if (rc != MI_LTE_OK) return rc;
        hc->sf_cfg = cfg;
    }
    const uint32_t stride = mi_lte_pdsch_plan_out_stride(hc->sf_plan);
    if (!hc->h_sf_res) {
        MI_HIP_CHECK(ctx, hipHostMalloc((void **)&hc->h_sf_res, 64 + (size_t)MI_LTE_PDCCH_MAX_DCI * 6144, hipHostMallocMapped | hipHostMallocCoherent));
        MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&hc->d_sf_res, hc->h_sf_res, 0));
    }
    if (stride > 6144 || out_stride < 6120) return MI_LTE_ERR_INVALID_ARG;

    // samples -> device subframe, as in mi_lte_get_dl_subframe_and_ce_host (the FFT kernel reads the mapped staging buffer)
    const uint32_t sc = 2048 / fft_size;
    const size_t   per_sf = 30720 / sc, need = per_sf + 2 * fft_size + 160 / sc + 144 / sc - 1;
    const size_t   start = (size_t)frame_start_idx + (size_t)subfr_num * per_sf;
    const size_t   nb = 2 * need * 4;
    rc = need_pin(ctx, hc, nb + 16);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    memcpy(hc->h_pin, h_i + start, need * 4);
    memcpy(hc->h_pin + need * 4, h_q + start, need * 4);
    struct { uint64_t start; uint32_t sf, cell; } par = {0, subfr_num, N_id_cell};
    memcpy(hc->h_pin + nb, &par, sizeof(par));
    uint8_t *d_src;
    MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&d_src, hc->h_pin, 0));
    const uint32_t *d_p = (const uint32_t *)(d_src + nb);
    hc->sub_host = nullptr; // d_sub is being rewritten, and mirrors no host struct afterwards
    rc = mi_lte_dl_frontend_batch(ctx, &cfg, d_src, d_src + need * 4, (const uint64_t *)d_p, d_p + 2, d_p + 3, 1, hc->d_sub);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    rc = bind_params(ctx, hc, subfr_num, N_id_cell);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    uint32_t h_rc = 1;
    *N_dci = 0;
    rc = mi_lte_pdcch_decode_run(ctx, cplan, hc->d_sub, hc->d_par + 4, hc->d_par + 5, 1, &h_rc, cfi, N_symbs, N_dci, dci); // (first wait)
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    if (h_rc != 0) return (int)h_rc; // no PCFICH / no DCI: what liblte_phy_pdcch_channel_decode reports
    // every DCI's transport block in one decode; the ones outside the single-code-block envelope report a decode failure, as the
    // per-allocation form does
    mi_lte_pdsch_alloc al[MI_LTE_PDCCH_MAX_DCI];
    uint32_t           slot[MI_LTE_PDCCH_MAX_DCI], n_al = 0;
    for (uint32_t k = 0; k < *N_dci && k < MI_LTE_PDCCH_MAX_DCI; k++) {
        status[k] = 2; N_out_bits[k] = 0;
        const mi_lte_pdsch_alloc &a = dci[k].alloc;
        if (a.tbs == 0 || !mi_lte_pdsch_alloc_decodable(&cfg, &a, *N_symbs)) continue;
        al[n_al] = a; al[n_al].unit = 0; al[n_al].n_pdcch_symbs = *N_symbs;
        slot[n_al++] = k;
    }
    if (n_al == 0) return 0;
    rc = mi_lte_pdsch_plan_assign(ctx, hc->sf_plan, *N_symbs, al, n_al);
    if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_UNSUPPORTED ? 0 : fail_after_launch(ctx, hc, rc);
    rc = mi_lte_pdsch_decode_run(ctx, hc->sf_plan, hc->d_sub, hc->d_par + 4, hc->d_par + 5, hc->d_sf_res + 64, (int32_t *)hc->d_sf_res);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx)); // (second wait)
    for (uint32_t j = 0; j < n_al; j++) {
        int32_t st;
        memcpy(&st, hc->h_sf_res + 4 * j, 4);
        status[slot[j]] = st;
        if (st == 0) {
            memcpy(h_out_bits + (size_t)slot[j] * out_stride, hc->h_sf_res + 64 + (size_t)j * stride, al[j].tbs);
            N_out_bits[slot[j]] = al[j].tbs;
        }
    }
    return 0;
}

// liblte_phy_pdsch_channel_decode (liblte_phy.cc:3690-3853)
int mi_lte_pdsch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im,
                                     const float *h_ce_re, const float *h_ce_im, uint32_t subfr_num, const mi_lte_pdsch_alloc *alloc,
                                     uint32_t N_pdcch_symbs, uint32_t N_id_cell, uint32_t N_ant, uint8_t *h_out_bits,
                                     uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || !alloc || N_id_cell > 503 || !h_out_bits || !N_out_bits || N_rb_dl < 6 || N_rb_dl > 100 ||
        !(N_ant == 1 || N_ant == 2 || N_ant == 4))
        return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg       cfg = {fft_of(N_rb_dl), N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
    mi_lte_pdsch_alloc  a   = *alloc;
    a.unit = 0; a.n_pdcch_symbs = 0;
    // a resource block past the carrier (a DCI that passed its CRC by chance): the reference demodulates whatever lies behind the row and
    // fails the transport block's CRC; the plans refuse such a list, so report the decode failure here
    if (!prbs_on_carrier(a, N_rb_dl)) return 2;
    std::string key;
    key_add(key, cfg); key_add(key, N_pdcch_symbs); key_add(key, a);
    mi_lte_pdsch_plan *plan = hc->pdsch.find(key);
    if (!plan) {
        rc = mi_lte_pdsch_plan_create(ctx, &cfg, N_pdcch_symbs, &a, 1, &plan);
        if (rc == MI_LTE_ERR_UNSUPPORTED) return 2; // outside the envelope the reference itself decodes: report a decode failure
        if (rc != MI_LTE_OK) return rc;
        hc->pdsch.put(ctx, key, plan);
    }
    rc = bind_subframe(ctx, hc, h_symb_re, h_symb_im, h_ce_re, h_ce_im, N_ant, 12 * N_rb_dl, false);
    if (rc != MI_LTE_OK) return rc;
    rc = bind_params(ctx, hc, subfr_num, N_id_cell);
    if (rc != MI_LTE_OK) return rc;
    rc = mi_lte_pdsch_decode_run(ctx, plan, hc->d_sub, hc->d_par + 4, hc->d_par + 5, hc->d_out, hc->d_st);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    // verdict and bits were written straight into pinned host memory (the bits are only handed over when the CRC matched, like the
    // reference, :12861-12869)
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    int32_t st;
    memcpy(&st, hc->h_res, 4);
    if (st == 0) {
        memcpy(h_out_bits, hc->h_res + 64, a.tbs);
        *N_out_bits = a.tbs;
    }
    return (int)st;
}

// liblte_phy_pdcch_channel_decode (liblte_phy.cc:4519-5135): PCFICH + common-search-space DCIs of one subframe
int mi_lte_pdcch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im, const float *h_ce_re,
                                     const float *h_ce_im, uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, float phich_res,
                                     uint32_t phich_dur_extended, uint32_t flags, uint32_t *cfi, uint32_t *N_symbs, uint32_t *N_dci,
                                     mi_lte_pdcch_dci *dci /*[MI_LTE_PDCCH_MAX_DCI]*/)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || N_id_cell > 503 || !cfi || !N_symbs || !N_dci || !dci || !(N_ant == 1 || N_ant == 2 || N_ant == 4)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg cfg = {fft_of(N_rb_dl), N_rb_dl, N_ant, MI_LTE_IQ_F32_PLANAR};
    std::string   key;
    key_add(key, cfg); key_add(key, phich_res); key_add(key, phich_dur_extended); key_add(key, flags); key_add(key, N_id_cell);
    mi_lte_pdcch_plan *plan = hc->pdcch.find(key);
    if (!plan) {
        rc = mi_lte_pdcch_plan_create(ctx, &cfg, phich_res, phich_dur_extended, flags, &N_id_cell, 1, &plan);
        if (rc != MI_LTE_OK) return rc;
        hc->pdcch.put(ctx, key, plan);
    }
    rc = bind_subframe(ctx, hc, h_symb_re, h_symb_im, h_ce_re, h_ce_im, N_ant, 12 * N_rb_dl, false);
    if (rc != MI_LTE_OK) return rc;
    uint32_t h_rc = 1;
    rc = bind_params(ctx, hc, subfr_num, N_id_cell);
    if (rc != MI_LTE_OK) return rc;
    rc = mi_lte_pdcch_decode_run(ctx, plan, hc->d_sub, hc->d_par + 4, hc->d_par + 5, 1, &h_rc, cfi, N_symbs, N_dci, dci);
    return rc != MI_LTE_OK ? fail_after_launch(ctx, hc, rc) : (int)h_rc;
}

// liblte_phy_bch_channel_decode (liblte_phy.cc:3968-4105)
int mi_lte_bch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_dl, const float *h_symb_re, const float *h_symb_im, const float *h_ce_re,
                                   const float *h_ce_im, uint32_t N_id_cell, uint8_t *N_ant, uint8_t *h_out_bits, uint32_t *N_out_bits, uint8_t *offset)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !h_ce_re || !h_ce_im || N_id_cell > 503 || !N_ant || !h_out_bits || !N_out_bits || !offset) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg cfg = {fft_of(N_rb_dl), N_rb_dl, 4, MI_LTE_IQ_F32_PLANAR};
    rc = bind_subframe(ctx, hc, h_symb_re, h_symb_im, h_ce_re, h_ce_im, 4, 12 * N_rb_dl, false); // all four ports' estimates are tried
    if (rc != MI_LTE_OK) return rc;
    hc->h_par[6] = N_id_cell;
    uint32_t n_ant = 0, off = 0, mib = 0;
    rc = mi_lte_pbch_decode_run(ctx, &cfg, hc->d_sub, hc->d_par + 6, 1, &n_ant, &off, &mib);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    *N_ant = (uint8_t)n_ant; // the reference zeroes it before trying (:4029)
    if (n_ant == 0) return 2;
    for (uint32_t i = 0; i < 24; i++) h_out_bits[i] = (uint8_t)((mib >> (23 - i)) & 1u);
    *N_out_bits = 24;
    *offset     = (uint8_t)off;
    return 0;
}

// liblte_phy_pucch_format_1_1a_1b_channel_decode (liblte_phy.cc:2961-3146); the sequences are the caller's (mi_lte.h)
int mi_lte_pucch_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_symb_re, const float *h_symb_im, uint32_t format, uint32_t N_ant,
                             uint32_t N_1_p_pucch, const float *h_tables, uint8_t *h_out_bits, uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || format > 2 || !h_tables || !h_out_bits || !N_out_bits) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    rc = bind_subframe(ctx, hc, h_symb_re, h_symb_im, nullptr, nullptr, 1, 12 * N_rb_ul, true);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_pucch_res r = {0, format, N_1_p_pucch};
    uint8_t  bits[2] = {0, 0};
    uint32_t nb = 0, rc2 = 1;
    rc = mi_lte_pucch_decode_run(ctx, N_rb_ul, N_ant, hc->d_sub, &r, h_tables, 1, bits, &nb, &rc2);
    if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_INVALID_ARG ? 1 : rc;
    h_out_bits[0] = bits[0];
    if (nb == 2) h_out_bits[1] = bits[1];
    *N_out_bits = nb;
    return (int)rc2;
}

// ---- initial synchronisation (liblte_phy.cc:5697-5852, :5306-5510, :5578-5687): stage n samples of each array, run the device search
int mi_lte_dl_find_coarse_timing_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t N_slots,
                                      mi_lte_coarse_timing *out)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !out || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_i, h_q, mi_lte_coarse_timing_samples(fft_size, N_slots), false, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    return mi_lte_coarse_timing_run(ctx, &cfg, d_i, d_q, 0, N_slots, out);
}

int mi_lte_find_pss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t *symb_starts,
                         uint32_t *N_id_2, uint32_t *pss_symb, float *pss_thresh, float *freq_offset)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !symb_starts || !N_id_2 || !pss_symb || !pss_thresh || !freq_offset || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg  cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    const uint32_t sc = 2048 / fft_size;
    uint32_t       last = 0;
    for (int j = 0; j < 7; j++) last = symb_starts[j] > last ? symb_starts[j] : last;
    const size_t n = (size_t)last + 11 * (15360 / sc) + 160 / sc + fft_size + 38; // last window of the 84, or of the +39 fine-timing trial
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_i, h_q, n, false, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    return mi_lte_find_pss_run(ctx, &cfg, d_i, d_q, 0, symb_starts, N_id_2, pss_symb, pss_thresh, freq_offset);
}

int mi_lte_find_sss_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_dl, const float *h_i, const float *h_q, uint32_t N_id_2, uint32_t *symb_starts,
                         float pss_thresh, uint32_t *N_id_1, uint32_t *frame_start_idx)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !symb_starts || !N_id_1 || !frame_start_idx || !valid_fft(fft_size, N_rb_dl)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg  cfg = {fft_size, N_rb_dl, 1, MI_LTE_IQ_F32_PLANAR};
    const uint32_t sc = 2048 / fft_size;
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_i, h_q, (size_t)symb_starts[5] + 160 / sc - 1 + fft_size, false, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    uint32_t found = 0;
    rc = mi_lte_find_sss_run(ctx, &cfg, d_i, d_q, 0, N_id_2, symb_starts, pss_thresh, N_id_1, frame_start_idx, &found);
    return rc != MI_LTE_OK ? rc : (found ? 0 : 1);
}

// liblte_phy_get_ul_subframe (liblte_phy.cc:6209-6236)
int mi_lte_get_ul_subframe_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const float *h_i, const float *h_q, float *h_symb_re,
                                float *h_symb_im)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !h_symb_re || !h_symb_im || !valid_fft(fft_size, N_rb_ul)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    const uint32_t sc = 2048 / fft_size;
    const size_t   need = 30720 / sc; // the last symbol's window ends one sample before the subframe does
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_i, h_q, need, true, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    // (the sample offset of the one unit is words 0..1 of the parameter block: zero since it was allocated)
    hc->sub_host = nullptr;
    mi_lte_dl_cfg cfg = {fft_size, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    rc = mi_lte_ul_frontend_batch(ctx, &cfg, d_i, d_q, (const uint64_t *)hc->d_par, 1, hc->d_sub);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    const float *st;
    rc = fetch_packed_rows(ctx, hc, 2, 12 * N_rb_ul, &st);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    // rows 14, 15 of the caller's struct and the columns past 12*N_rb_ul are left alone, as the reference leaves them (samples_to_symbols_ul, :8686-8691)
    unpack_rows(h_symb_re, st, 0, 14, 12 * N_rb_ul);
    unpack_rows(h_symb_im, st, 1, 14, 12 * N_rb_ul);
    hc->sub_host = h_symb_re; hc->sub_n_ant = 1; hc->sub_ul = true;
    hc->sub_fp = subframe_fp(h_symb_re, h_symb_im, nullptr, nullptr, 0, 12 * N_rb_ul, 14);
    return 0;
}

// liblte_phy_pusch_channel_decode (liblte_phy.cc:2801-2935).  The reference signals are the caller's: the arrays
// liblte_phy_ul_init stored in LIBLTE_PHY_STRUCT (pusch_dmrs_0/1_re/im[subframe][N_prb]).  A failed CRC is reported
// as 1 = LIBLTE_ERROR_INVALID_INPUTS, which is what the reference returns in that case (:2809, :2929).
int mi_lte_pusch_channel_decode_host(mi_lte_ctx *ctx, uint32_t N_rb_ul, const float *h_symb_re, const float *h_symb_im, uint32_t subfr_num,
                                     const mi_lte_pdsch_alloc *alloc, uint32_t N_id_cell, uint32_t N_ant, const float *h_dmrs_0_re,
                                     const float *h_dmrs_0_im, const float *h_dmrs_1_re, const float *h_dmrs_1_im, uint8_t *h_out_bits,
                                     uint32_t *N_out_bits)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_symb_re || !h_symb_im || !alloc || !h_out_bits || !N_out_bits || !h_dmrs_0_re || !h_dmrs_0_im || !h_dmrs_1_re || !h_dmrs_1_im || N_rb_ul < 6 || N_rb_ul > 100 ||
        alloc->N_prb == 0 || alloc->N_prb > 110)
        return 1;
    (void)N_ant;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg      cfg = {fft_of(N_rb_ul), N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    mi_lte_pdsch_alloc a   = *alloc;
    a.unit = 0; a.n_pdcch_symbs = 0;
    const size_t       M   = 12 * (size_t)a.N_prb;
    std::vector<float> dm(4 * M);
    memcpy(&dm[0], h_dmrs_0_re, M * 4); memcpy(&dm[M], h_dmrs_0_im, M * 4);
    memcpy(&dm[2 * M], h_dmrs_1_re, M * 4); memcpy(&dm[3 * M], h_dmrs_1_im, M * 4);
    // the plan carries the subframe number, the cell and the reference signals: all of them are part of its key
    std::string key;
    key_add(key, cfg); key_add(key, subfr_num); key_add(key, N_id_cell); key_add(key, a);
    key_add(key, fnv(dm.data(), dm.size() * 4));
    mi_lte_pusch_plan *plan = hc->pusch.find(key);
    if (!plan) {
        rc = mi_pusch_plan_create_impl(ctx, &cfg, nullptr, &subfr_num, &N_id_cell, 1, &a, 1, dm.data(), &plan);
        if (rc == MI_LTE_ERR_UNSUPPORTED || (rc == MI_LTE_ERR_INVALID_ARG && !prbs_on_carrier(a, N_rb_ul))) return 1; // (what the reference reports for an allocation it cannot decode)
        if (rc != MI_LTE_OK) return rc;
        hc->pusch.put(ctx, key, plan);
    }
    rc = bind_subframe(ctx, hc, h_symb_re, h_symb_im, nullptr, nullptr, 1, 12 * N_rb_ul, true);
    if (rc != MI_LTE_OK) return rc;
    rc = mi_lte_pusch_decode_run(ctx, plan, hc->d_sub, hc->d_out, hc->d_st);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    int32_t st;
    memcpy(&st, hc->h_res, 4);
    if (st == 0) {
        memcpy(h_out_bits, hc->h_res + 64, a.tbs);
        *N_out_bits = a.tbs;
    }
    return st == 0 ? 0 : 1;
}

// One UPLINK subframe, one call: what LTE_fdd_enb_phy.cc:832-917 does with 1 + N_pucch + N_pusch calls of the reference API
// (liblte_phy_get_ul_subframe, liblte_phy_pucch_format_1_1a_1b_channel_decode per resource, liblte_phy_pusch_channel_decode per scheduled UE)
// as ONE dependent launch chain with ONE polled wait: the received grid never leaves HBM (the per-call sequence brings its 134 KB back and
// fingerprints it before every decode), the PUCCH resources and all PUSCH transport blocks are queued behind the front end at once.
// The eNodeB's radio thread has 1 ms per subframe for this and the downlink encode (README).
int mi_lte_ul_subframe_decode_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const float *h_i, const float *h_q, uint32_t subfr_num,
                                   uint32_t N_id_cell, const mi_lte_ul_cfg *ul, const mi_lte_pdsch_alloc *allocs, uint32_t n_alloc, uint8_t *h_out_bits,
                                   uint32_t out_stride, uint32_t *N_out_bits, int32_t *status, const mi_lte_pucch_res *pucch, const float *h_pucch_tables,
                                   uint32_t n_pucch, uint8_t *h_pucch_bits, uint32_t *h_pucch_n_bits, uint32_t *h_pucch_rc)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    if (!h_i || !h_q || !ul || subfr_num > 9 || N_id_cell > 503 || !valid_fft(fft_size, N_rb_ul) || n_alloc > MI_LTE_UL_SUBFRAME_MAX_ALLOC ||
        (n_alloc && (!allocs || !h_out_bits || !N_out_bits || !status || out_stride < 6120)) || n_pucch > MI_PUCCH_STAGED_MAX ||
        (n_pucch && (!pucch || !h_pucch_tables || !h_pucch_bits || !h_pucch_n_bits || !h_pucch_rc)))
        return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    const mi_lte_dl_cfg cfg = {fft_size, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    // the transport blocks this library decodes (the reference's envelope: one code block, a width it has a transform plan for,
    // liblte_phy.cc:2360-2377); the others report the reference's failure value without stopping the rest
    mi_lte_pdsch_alloc al[MI_LTE_UL_SUBFRAME_MAX_ALLOC];
    uint32_t           slot[MI_LTE_UL_SUBFRAME_MAX_ALLOC], n_al = 0;
    for (uint32_t k = 0; k < n_alloc; k++) {
        status[k] = 1; N_out_bits[k] = 0;
        const mi_lte_pdsch_alloc &a = allocs[k];
        const bool planned = a.N_prb > 0 && a.N_prb < N_rb_ul && (a.N_prb % 2 == 0 || a.N_prb % 3 == 0 || a.N_prb % 5 == 0);
        if (!planned || a.tbs + 24 > 6144 || a.tbs + 24 < a.tbs || a.mod_type > 3 || !prbs_on_carrier(a, N_rb_ul)) continue;
        al[n_al] = a; al[n_al].unit = 0; al[n_al].n_pdcch_symbs = 0;
        slot[n_al++] = k;
    }
    mi_lte_pusch_plan *plan = nullptr;
    if (n_al) { // the plan carries subframe number, cell, reference signals and the list: all of them are its key (a grant pattern that recurs is planned once)
        std::string key;
        key_add(key, cfg); key_add(key, subfr_num); key_add(key, N_id_cell); key_add(key, *ul); key_add(key, n_al);
        for (uint32_t j = 0; j < n_al; j++) key_add(key, al[j]);
        plan = hc->pusch.find(key);
        if (!plan) {
            rc = mi_pusch_plan_create_impl(ctx, &cfg, ul, &subfr_num, &N_id_cell, 1, al, n_al, nullptr, &plan);
            if (rc != MI_LTE_OK) return rc;
            hc->pusch.put(ctx, key, plan);
        }
        if (mi_lte_pusch_plan_out_stride(plan) > 6144) return MI_LTE_ERR_INVALID_ARG;
        if (!hc->h_ul_res) {
            MI_HIP_CHECK(ctx, hipHostMalloc((void **)&hc->h_ul_res, 256 + (size_t)MI_LTE_UL_SUBFRAME_MAX_ALLOC * 6144, hipHostMallocMapped | hipHostMallocCoherent));
            MI_HIP_CHECK(ctx, hipHostGetDevicePointer((void **)&hc->d_ul_res, hc->h_ul_res, 0));
        }
    }
    // samples where the transform reads them (the mapped staging buffer), as in mi_lte_get_ul_subframe_host; this is the call's first and --
    // the stream being idle between calls -- free wait
    const uint32_t sc = 2048 / fft_size;
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_i, h_q, 30720 / sc, true, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    MiPucchStaged ps;
    if (n_pucch) {
        for (uint32_t r = 0; r < n_pucch; r++)
            if (pucch[r].unit != 0) return 1;
        rc = mi_pucch_stage(ctx, N_rb_ul, pucch, h_pucch_tables, n_pucch, &ps); // (the stream is idle: nothing reads the block)
        if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_INVALID_ARG ? 1 : rc;
    }
    hc->sub_host = nullptr; // d_sub is being rewritten, and mirrors no host struct afterwards
    rc = mi_lte_ul_frontend_batch(ctx, &cfg, d_i, d_q, (const uint64_t *)hc->d_par, 1, hc->d_sub);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    if (n_pucch && (rc = mi_pucch_launch(ctx, &ps, N_rb_ul, hc->d_sub)) != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    if (plan && (rc = mi_lte_pusch_decode_run(ctx, plan, hc->d_sub, hc->d_ul_res + 256, (int32_t *)hc->d_ul_res)) != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx)); // (the wait)
    if (n_pucch) mi_pucch_collect(&ps, h_pucch_bits, h_pucch_n_bits, h_pucch_rc);
    const uint32_t stride = plan ? mi_lte_pusch_plan_out_stride(plan) : 0;
    for (uint32_t j = 0; j < n_al; j++) {
        int32_t st;
        memcpy(&st, hc->h_ul_res + 4 * j, 4);
        status[slot[j]] = st == 0 ? 0 : 1; // a failed CRC is LIBLTE_ERROR_INVALID_INPUTS on this path (liblte_phy.cc:2809, :2929)
        if (st == 0) {
            memcpy(h_out_bits + (size_t)slot[j] * out_stride, hc->h_ul_res + 256 + (size_t)j * stride, al[j].tbs);
            N_out_bits[slot[j]] = al[j].tbs;
        }
    }
    return 0;
}

// liblte_phy_detect_prach (liblte_phy.cc:3299-3479): h_re / h_im point at the occasion's first cyclic-prefix sample; the root
// spectra are the caller's (LIBLTE_PHY_STRUCT::prach_x_u_fft_re/im, filled by liblte_phy_ul_init).  N_det_pre is always
// written; det_pre / det_ta only when a preamble was found, like the reference.
int mi_lte_detect_prach_host(mi_lte_ctx *ctx, uint32_t fft_size, uint32_t N_rb_ul, const mi_lte_prach_cfg *prach, const float *h_x_u_fft_re,
                             const float *h_x_u_fft_im, uint32_t n_roots, const float *h_re, const float *h_im, uint32_t *N_det_pre,
                             uint32_t *det_pre, uint32_t *det_ta)
{
    if (!ctx) return MI_LTE_ERR_INVALID_ARG;
    const bool own_roots = !h_x_u_fft_re && !h_x_u_fft_im; // no spectra handed over: the library's own root set for the cell's configuration
    if (!prach || (!own_roots && (!h_x_u_fft_re || !h_x_u_fft_im || n_roots == 0 || n_roots > 64)) || !h_re || !h_im || !N_det_pre || !det_pre || !det_ta || !valid_fft(fft_size, N_rb_ul)) return 1;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    mi_lte_dl_cfg cfg = {fft_size, N_rb_ul, 1, MI_LTE_IQ_F32_PLANAR};
    std::string   key;
    key_add(key, cfg); key_add(key, *prach); key_add(key, own_roots ? 0u : n_roots);
    if (!own_roots) key_add(key, fnv(h_x_u_fft_re, (size_t)n_roots * 839 * 4, fnv(h_x_u_fft_im, (size_t)n_roots * 839 * 4)));
    mi_lte_prach_plan *plan = hc->prach.find(key);
    if (!plan) {
        rc = own_roots ? mi_lte_prach_plan_create(ctx, &cfg, prach, &plan) : mi_lte_prach_plan_create_roots(ctx, &cfg, prach, h_x_u_fft_re, h_x_u_fft_im, n_roots, &plan);
        if (rc != MI_LTE_OK) return rc == MI_LTE_ERR_UNSUPPORTED ? 1 : rc;
        hc->prach.put(ctx, key, plan);
    }
    const size_t need = mi_lte_prach_occasion_samples(plan);
    float *d_i, *d_q;
    rc = stage_pair(ctx, hc, h_re, h_im, need, false, &d_i, &d_q);
    if (rc != MI_LTE_OK) return rc;
    // (the sample offset of the one unit is words 0..1 of the parameter block: zero since it was allocated)
    uint32_t n = 0, p = 0, ta = 0;
    rc = mi_lte_prach_detect_run(ctx, plan, d_i, d_q, (const uint64_t *)hc->d_par, 1, &n, &p, &ta);
    if (rc != MI_LTE_OK) return fail_after_launch(ctx, hc, rc);
    *N_det_pre = n;
    if (n) { *det_pre = p; *det_ta = ta; }
    return 0;
}

// liblte_phy_rate_unmatch_turbo (liblte_phy.cc:11246-11490)
int mi_lte_rate_unmatch_turbo_host(mi_lte_ctx *ctx, const float *h_e, uint32_t N_e, uint32_t N_dummy_bits, uint32_t C, uint32_t tx_mode,
                                   uint32_t N_soft, uint32_t M_dl_harq, uint32_t chan_type, uint32_t rv_idx, float *h_d, uint32_t *N_d)
{
    if (!ctx || !h_e || !h_d || !N_d) return MI_LTE_ERR_INVALID_ARG;
    MI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (N_dummy_bits < 44 || N_dummy_bits > 6148 || N_e == 0) return MI_LTE_ERR_INVALID_ARG;
    HostCache *hc;
    int        rc = host_cache(ctx, &hc);
    if (rc != MI_LTE_OK) return rc;
    const size_t e_bytes = (((size_t)N_e * 4) + 255) & ~(size_t)255, d_bytes = (size_t)3 * N_dummy_bits * 4;
    rc = need_pin(ctx, hc, e_bytes > d_bytes ? e_bytes : d_bytes);
    if (rc == MI_LTE_OK) rc = need_dev_in(ctx, hc, e_bytes + d_bytes);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    memcpy(hc->h_pin, h_e, (size_t)N_e * 4);
    float *d_e = (float *)hc->d_in, *d_d = (float *)(hc->d_in + e_bytes);
    MI_HIP_CHECK(ctx, mi_pinned_to_device(ctx, d_e, hc->h_pin, (size_t)N_e * 4));
    rc = mi_lte_rate_unmatch_turbo_batch(ctx, d_e, N_e, N_dummy_bits, C, tx_mode, N_soft, M_dl_harq, chan_type, rv_idx, 1, d_d);
    if (rc != MI_LTE_OK) return rc;
    MI_HIP_CHECK(ctx, mi_device_to_pinned(ctx, hc->h_pin, d_d, d_bytes));
    MI_HIP_CHECK(ctx, mi_stream_wait_polling(ctx));
    memcpy(h_d, hc->h_pin, d_bytes);
    *N_d = 3 * N_dummy_bits;
    return 0;
}

} // extern "C"
