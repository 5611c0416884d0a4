/*
 * lte_oracle.c -- CPU restatement of the reference's DL receive chain.  TEST INFRASTRUCTURE ONLY.
 * See lte_oracle.h for the rules on who may call this.  Reference line numbers are for
 * liblte/src/liblte_phy.cc unless another file is named.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared lte_oracle.c -o liblte_oracle.so -lm
 * (-ffp-contract=off: the reference is built for baseline x86-64, i.e. without FMA contraction,
 * and several results are truncated float->int8, where a fused multiply-add could flip a bit.)
 */
#include "lte_oracle.h"
#include "lte_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* numerology (liblte_phy.cc:2226-2277 sample-rate switch, :2592-2656 update_n_rb_dl)           */
int lo_cfg_init(lo_cfg_t *cfg, uint32_t fft_size, uint32_t N_rb_dl)
{
    uint32_t scale;
    if (fft_size != 128 && fft_size != 256 && fft_size != 512 && fft_size != 1024 && fft_size != 2048) return -1;
    scale = 2048 / fft_size; /* all CP/slot lengths are the 30.72 MHz ones divided by the FFT ratio */
    cfg->N_samps_per_symb  = fft_size;
    cfg->N_samps_cp_l_0    = 160 / scale;
    cfg->N_samps_cp_l_else = 144 / scale;
    cfg->N_samps_per_slot  = 15360 / scale;
    cfg->N_samps_per_subfr = 30720 / scale;
    cfg->N_rb_dl           = N_rb_dl;
    cfg->N_sc_rb_dl        = 12;
    cfg->FFT_size          = fft_size;
    cfg->FFT_pad_size      = (fft_size - N_rb_dl * 12) / 2;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* QPP interleaver (liblte_phy.cc:10934-11072): idx = (f1*i + f2*i*i) % K evaluated in uint32,  */
/* which wraps for 20 block sizes.  lo_qpp_map_spec is the 3GPP-exact alternative.  */
int lo_qpp_params(uint32_t K, uint32_t *f1, uint32_t *f2)
{
    for (int r = 0; r < LTE_QPP_N_SIZES; r++) {
        if (LTE_QPP_ROWS[r].K == K) {
            *f1 = LTE_QPP_ROWS[r].f1;
            *f2 = LTE_QPP_ROWS[r].f2;
            return 0;
        }
    }
    *f1 = 0; /* the reference falls through with f1 = f2 = 0 (every index maps to 0) */
    *f2 = 0;
    return -1;
}

void lo_qpp_map_ref(uint32_t K, uint16_t *pi)
{
    uint32_t f1, f2;
    lo_qpp_params(K, &f1, &f2);
    for (uint32_t i = 0; i < K; i++) {
        uint32_t idx = (f1 * i + f2 * i * i) % K; /* uint32 wrap-around is intentional */
        pi[i]        = (uint16_t)idx;
    }
}

void lo_qpp_map_spec(uint32_t K, uint16_t *pi)
{
    uint32_t f1, f2;
    lo_qpp_params(K, &f1, &f2);
    for (uint64_t i = 0; i < K; i++) pi[i] = (uint16_t)(((uint64_t)f1 * i + (uint64_t)f2 * i * i) % K);
}

/* ------------------------------------------------------------------------------------------ */
/* Gold sequence (liblte_phy.cc:9669-9704): x2 seeded with c_init and advanced 1600-31 times,  */
/* x1 starts from the pre-advanced constant 0x54D21B24; both 31-bit Fibonacci LFSRs.           */
void lo_prs_c(uint32_t c_init, uint32_t len, uint8_t *c)
{
    uint32_t x1 = 0x54D21B24u, x2 = c_init;
    for (uint32_t n = 0; n < 1600 - 31; n++) {
        uint32_t nb = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
        x2          = (x2 >> 1) | (nb << 30);
    }
    for (uint32_t n = 0; n < len; n++) {
        uint32_t nb1 = ((x1 >> 3) ^ x1) & 1u;
        uint32_t nb2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
        x1           = (x1 >> 1) | (nb1 << 30);
        x2           = (x2 >> 1) | (nb2 << 30);
        c[n]         = (uint8_t)(nb1 ^ nb2);
    }
}

/* CRS (liblte_phy.cc:8300-8333): 220 QPSK pilots per (slot, symbol). */
void lo_generate_crs(uint32_t N_s, uint32_t L, uint32_t N_id_cell, float *crs_re, float *crs_im)
{
    const float one_over_sqrt_2 = 1 / sqrt(2);
    uint8_t     c[440];
    uint32_t    c_init = 1024 * (7 * (N_s + 1) + L + 1) * (2 * N_id_cell + 1) + 2 * N_id_cell + 1;
    lo_prs_c(c_init, 440, c);
    for (int i = 0; i < 220; i++) {
        crs_re[i] = one_over_sqrt_2 * (1 - 2 * (float)c[2 * i]);
        crs_im[i] = one_over_sqrt_2 * (1 - 2 * (float)c[2 * i + 1]);
    }
}

/* CRC24A (liblte_phy.cc:9713-9743, polynomial :1376): bitwise long division, MSB first. */
void lo_crc24a(const uint8_t *bits, uint32_t n, uint8_t p[24])
{
    uint32_t rem = 0;
    for (uint32_t i = 0; i < n + 24; i++) {
        rem = (rem << 1) | (i < n ? bits[i] : 0u);
        if (rem & (1u << 24)) rem ^= 0x01864CFBu;
    }
    for (int i = 0; i < 24; i++) p[i] = (uint8_t)((rem >> (23 - i)) & 1u);
}

/* ------------------------------------------------------------------------------------------ */
/* REF turbo decoder                                                                          */

/* viterbi_decode_siso (liblte_phy.cc:10341-10529) for constraint_len 4, rate 2, g = {015,013}.
 * Path metrics are integer-valued floats in the reference; int32 is exact (|PM| <= 508*K). */
void lo_viterbi_siso(const int8_t *in, uint32_t K, int8_t *out)
{
    uint8_t  o0[8][2], o1[8][2];
    int32_t(*pm)[8] = (int32_t(*)[8])calloc((size_t)K + 1, sizeof(int32_t[8]));
    uint8_t *st     = (uint8_t *)malloc((size_t)K + 1);
    int32_t  W      = 0;

    /* expected encoder outputs for the transition prev -> s (liblte_phy.cc:10379-10408) */
    for (int s = 0; s < 8; s++) {
        for (int k = 0; k < 2; k++) {
            int prev = ((s << 1) + k) & 7, u = s >> 2;
            int p2 = (prev >> 2) & 1, p1 = (prev >> 1) & 1, p0 = prev & 1;
            o0[s][k] = (uint8_t)(u ^ p2 ^ p0); /* g = 015: taps on r0, r1, r3 */
            o1[s][k] = (uint8_t)(u ^ p1 ^ p0); /* g = 013: taps on r0, r2, r3 */
        }
    }

    /* add-compare-select; survivor picked on the HARD metric, accumulated metric is WEIGHTED
     * (liblte_phy.cc:10419-10465) */
    for (uint32_t t = 0; t < K; t++) {
        int x = in[2 * t], y = in[2 * t + 1];
        int b0 = x < 0, b1 = y < 0;
        int w = abs(x) + abs(y);
        if (w > W) W = w;
        for (int s = 0; s < 8; s++) {
            int pa = (s << 1) & 7, pb = pa + 1;
            int bra = (o0[s][0] == b0 ? -1 : 1) + (o1[s][0] == b1 ? -1 : 1);
            int brb = (o0[s][1] == b0 ? -1 : 1) + (o1[s][1] == b1 ? -1 : 1);
            if (bra + pm[t][pa] > brb + pm[t][pb]) pm[t + 1][s] = pm[t][pb] + w * brb;
            else                                   pm[t + 1][s] = pm[t][pa] + w * bra;
        }
    }

    /* end state: first strict minimum (liblte_phy.cc:10467-10481) */
    {
        int32_t best = 1000000;
        int     sel  = 0;
        for (int s = 0; s < 8; s++) {
            if (pm[K][s] < best) { best = pm[K][s]; sel = s; }
        }
        st[K] = (uint8_t)sel;
    }
    /* traceback compares the STORED metrics of the two predecessors (liblte_phy.cc:10483-10503) */
    for (int32_t t = (int32_t)K - 1; t >= 0; t--) {
        int pa = (st[t + 1] << 1) & 7, pb = pa + 1;
        st[t]  = (uint8_t)((pm[t][pa] > pm[t][pb]) ? pb : pa);
    }
    /* soft output (liblte_phy.cc:10506-10527); max_weight is the max branch weight on the path,
     * and the branch weight does not depend on the state, so it is max_t w_t */
    for (uint32_t t = 0; t < K; t++) {
        float wt  = (float)(abs(in[2 * t]) + abs(in[2 * t + 1]));
        int   pos = (st[t + 1] < st[t]) || (st[t + 1] == st[t] && st[t + 1] == 0);
        if (pos) out[t] = (int8_t)(127 * (wt / (float)W));
        else     out[t] = (int8_t)(-127 * (wt / (float)W));
    }
    free(pm);
    free(st);
}

/* conv_encode_soft with g = 03 over constraint length 3 (liblte_phy.cc:10070-10151) as called at
 * :10676-10685: fb[0] = 127, fb[i+1] = soft-xor of the two PREVIOUS inputs (x[i-1], x[i-2]),
 * register initialised to +127.  Only fb[0..K-1] is ever read afterwards. */
static int8_t soft_enc2(int a, int b)
{
    int mag = (a >= 0 ? a : -a) + (b >= 0 ? b : -b);
    int neg = (a < 0) + (b < 0);
    int8_t v = (int8_t)(mag >> 1);
    return (neg & 1) ? (int8_t)-v : v;
}
void lo_fb_soft(const int8_t *x, uint32_t K, int8_t *fb)
{
    fb[0] = 127;
    for (uint32_t i = 0; i + 1 < K; i++) {
        int a = (i >= 1) ? x[i - 1] : 127;
        int b = (i >= 2) ? x[i - 2] : 127;
        fb[i + 1] = soft_enc2(a, b);
    }
}

/* "soft xor" of steps 3 (liblte_phy.cc:10688-10707) */
static int8_t soft_xor(int a, int b)
{
    if (a >= 0 && b >= 0) return (int8_t)((a + b) >> 1);
    if (a < 0 && b < 0)   return (int8_t)((-a - b) >> 1);
    if (a >= 0 && b < 0)  return (int8_t)(-((a - b) >> 1));
    return (int8_t)(-((-a + b) >> 1));
}

/* turbo_decode, Steps 0-14 (liblte_phy.cc:10620-10845).  De-interleaver targets that are never
 * written ("holes", only for the uint32-overflow K) read as 0: the reference leaves stale scratch
 * there, and the parity harness zeroes that scratch before every call. */
void lo_turbo_decode_ref_taps(const float *d_in, uint32_t K, uint8_t *c_bits, int8_t *taps)
{
    const uint32_t D = K + 4;
    float   *d   = (float *)malloc(sizeof(float) * 3 * D);
    int8_t  *buf = (int8_t *)calloc(16 * (size_t)(K + 8), 1);
    int8_t  *q0 = buf, *q1 = q0 + K + 8, *q2 = q1 + K + 8, *A1 = q2 + K + 8, *F1 = A1 + K + 8,
            *C1 = F1 + K + 8, *I0 = C1 + K + 8, *I1 = I0 + K + 8, *B1 = I1 + K + 8, *B2 = B1 + K + 8,
            *G1 = B2 + K + 8, *G2 = G1 + K + 8, *D1 = G2 + K + 8, *D2 = D1 + K + 8, *C2 = D2 + K + 8,
            *C3 = C2 + K + 8;
    int8_t  *vin = (int8_t *)malloc(2 * (size_t)K);
    uint16_t *pi = (uint16_t *)malloc(sizeof(uint16_t) * K);
    float    max_value = 0;

    /* Step 0: punctured positions -> 0 (:10636-10642) */
    for (uint32_t i = 0; i < 3 * D; i++) d[i] = (d_in[i] == LO_RX_NULL) ? 0.0f : d_in[i];
    /* Step 1: scale to int8 by the max over the K systematic/parity triples (tail excluded) */
    for (uint32_t i = 0; i < K; i++)
        for (int x = 0; x < 3; x++)
            if (fabsf(d[i * 3 + x]) > max_value) max_value = fabsf(d[i * 3 + x]);
    for (uint32_t i = 0; i < K; i++) {
        q0[i] = (int8_t)(d[i * 3 + 0] * 127 / max_value);
        q1[i] = (int8_t)(d[i * 3 + 1] * 127 / max_value);
        q2[i] = (int8_t)(d[i * 3 + 2] * 127 / max_value);
    }
    lo_qpp_map_ref(K, pi);

    for (uint32_t i = 0; i < K; i++) { vin[2 * i] = q1[i]; vin[2 * i + 1] = q0[i]; }
    lo_viterbi_siso(vin, K, A1);                                              /* Step 1  */
    lo_fb_soft(A1, K, F1);                                                    /* Step 2  */
    for (uint32_t i = 0; i < K; i++) C1[i] = soft_xor(A1[i], F1[i]);          /* Step 3  */
    for (uint32_t i = 0; i < K; i++) I0[i] = q0[pi[i]];                       /* Step 4  */
    for (uint32_t i = 0; i < K; i++) I1[i] = C1[pi[i]];                       /* Step 5  */
    for (uint32_t i = 0; i < K; i++) { vin[2 * i] = q2[i]; vin[2 * i + 1] = I0[i]; }
    lo_viterbi_siso(vin, K, B1);                                              /* Step 6  */
    for (uint32_t i = 0; i < K; i++) { vin[2 * i] = q2[i]; vin[2 * i + 1] = I1[i]; }
    lo_viterbi_siso(vin, K, B2);                                              /* Step 7  */
    lo_fb_soft(B1, K, G1);                                                    /* Step 8  */
    lo_fb_soft(B2, K, G2);                                                    /* Step 9  */
    for (uint32_t i = 0; i < K; i++) {                                        /* Step 10 */
        int a = B1[i], b = G1[i], a1 = A1[i];
        if (a >= 0 && b >= 0)     D1[i] = (int8_t)((a + b) >> 1);
        else if (a < 0 && b < 0)  D1[i] = (int8_t)((-a - b) >> 1);
        else if (a >= 0 && b < 0) D1[i] = (int8_t)(-((a1 - b) >> 1));  /* A1, not B1 (:10791) */
        else                      D1[i] = (int8_t)(-((-a1 + b) >> 1)); /* A1, not B1 (:10794) */
    }
    for (uint32_t i = 0; i < K; i++) {                                        /* Step 11 */
        int a = B2[i], b = G2[i];
        if (a >= 0 && b >= 0)     D2[i] = (int8_t)((a + b) >> 1);
        else if (a < 0 && b < 0)  D2[i] = (int8_t)((-a - b) >> 1);
        else if (a >= 0 && b < 0) D2[i] = (int8_t)(-((a - b) >> 1));
        else                      D2[i] = (int8_t)(-((-a - b) >> 1));  /* second minus (:10816) */
    }
    for (uint32_t i = 0; i < K; i++) C2[pi[i]] = D1[i];                       /* Step 12 */
    for (uint32_t i = 0; i < K; i++) C3[pi[i]] = D2[i];                       /* Step 13 */
    for (uint32_t i = 0; i < K; i++) {                                        /* Step 14 */
        float s  = (float)(q0[i] + C1[i] + C2[i] + C3[i]);
        c_bits[i] = (s >= 0) ? 0 : 1;
    }
    if (taps) {
        int8_t *src[11] = {q0, A1, C1, I0, I1, B1, B2, D1, D2, C2, C3};
        for (int k = 0; k < 11; k++) memcpy(taps + (size_t)k * K, src[k], K);
    }
    free(d); free(buf); free(vin); free(pi);
}

void lo_turbo_decode_ref(const float *d_in, uint32_t K, uint8_t *c_bits)
{
    lo_turbo_decode_ref_taps(d_in, K, c_bits, NULL);
}

double lo_time_turbo_decode_ref(const float *d, uint32_t K, uint32_t n_cb, uint8_t *c_bits)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t b = 0; b < n_cb; b++) lo_turbo_decode_ref(d + (size_t)b * 3 * (K + 4), K, c_bits + (size_t)b * K);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------ */
/* turbo encoder (input synthesis): turbo_constituent_encoder :10855-10924, turbo_encode        */
/* :10541-10589.  Output is PLANAR d0[D] d1[D] d2[D].                                           */
static void rsc_encode(const uint8_t *in, uint32_t K, uint8_t *z, uint8_t *fb)
{
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (uint32_t i = 0; i < K + 4; i++) {
        s3 = s2; s2 = s1; s1 = s0;
        fb[i] = (uint8_t)(s2 ^ s3);
        s0    = (i < K) ? (fb[i] ^ in[i]) : 0; /* termination: the feedback bit is fed back in */
        z[i]  = (uint8_t)(s0 ^ s1 ^ s3);
    }
}
void lo_turbo_encode(const uint8_t *c, uint32_t K, uint8_t *d)
{
    const uint32_t D = K + 4;
    uint8_t  *z = (uint8_t *)malloc(4 * (size_t)D + K), *fb1 = z + D, *zp = fb1 + D, *xp = zp + D, *cp = xp + D;
    uint16_t *pi = (uint16_t *)malloc(sizeof(uint16_t) * K);
    lo_qpp_map_ref(K, pi);
    rsc_encode(c, K, z, fb1);
    for (uint32_t i = 0; i < K; i++) cp[i] = c[pi[i]];
    rsc_encode(cp, K, zp, xp);
    for (uint32_t i = 0; i < K; i++) { d[i] = c[i]; d[D + i] = z[i]; d[2 * D + i] = zp[i]; }
    d[K] = fb1[K];             d[K + 1] = z[K + 1];           d[K + 2] = xp[K];              d[K + 3] = zp[K + 1];
    d[D + K] = z[K];           d[D + K + 1] = fb1[K + 2];     d[D + K + 2] = zp[K];          d[D + K + 3] = xp[K + 2];
    d[2 * D + K] = fb1[K + 1]; d[2 * D + K + 1] = z[K + 2];   d[2 * D + K + 2] = xp[K + 1];  d[2 * D + K + 3] = zp[K + 2];
    free(z); free(pi);
}

/* ------------------------------------------------------------------------------------------ */
/* max-log-MAP ("BCJR") turbo decoder in 16-bit fixed point.  This is the specification of the    */
/* product's BCJR mode (bcjr.hip follows it operation for operation); the reference has no such   */
/* decoder, so parity for this mode is "the kernels equal this model bit for bit".    */
/*                                                                                                */
/* RSC of 36.212 5.1.3.2.1: feedback 1+D^2+D^3, parity 1+D+D^3.  State s = 4*r1 + 2*r2 + r3 (r1   */
/* newest).  With a = u^r2^r3 the next state is 4a + (s>>1) and the parity is z = a^r1^r3, so the  */
/* predecessors of next state n are 2(n&3) and 2(n&3)+1 with complementary (u,z) labels:           */
/*   n:      0      1      2      3      4      5      6      7                                    */
/*   from 2(n&3):   00     10     01     11     11     01     10     00                            */
/*   from 2(n&3)+1: 11     01     10     00     00     10     01     11                            */
/* Branch metric g(u,z) = [u==0]*(Ls+La) + [z==0]*Lp (positive LLR = bit 0; the common offset      */
/* against the +-L/2 form cancels in every comparison).                                            */
/*                                                                                                */
/* Number ranges (every intermediate fits int16, so the kernels run two code blocks per lane on    */
/* packed 16-bit arithmetic with no saturation anywhere):                                          */
/*   channel values Ls, Lp: int8, clipped to +-127;                                                */
/*   extrinsic: e = ((x*3)>>2) with x = clamp(llr - (Ls+La), +-340), clamped to +-254, STORED as   */
/*     q = e>>1 in int8 (the a-priori value the other decoder adds is La = 2q, |La| <= 254);       */
/*   alpha, beta: normalised (maximum 0, floor BCJR_NEG = -6000) every 8 steps; in between they    */
/*     move by at most 8*(127+254+127): [-10064, 4064]; alpha+beta+g >= -20636, llr within         */
/*     +-29272, llr - (Ls+La) within +-29653.                                                      */
/* Schedule: alpha runs continuously through a SEGMENT (n_seg = 8, 4, 2 or 1 per code block:       */
/* occupancy), starting from the value the previous segment reached in the previous iteration;     */
/* beta is initialised at the end of every 32-step BLOCK from the value the following block        */
/* reached at its start in the previous iteration of the same constituent decoder ("next           */
/* iteration initialisation"; all-zero = uniform before the first iteration; the block that ends   */
/* the code block starts from the termination bits).  That makes a block's forward and backward    */
/* pass one unit of work that needs nothing stored per trellis step.                               */
#define BCJR_NEG    (-6000)
#define BCJR_X_MAX  340 /* clamp of llr - (Ls+La) before the 3/4 scaling */
#define BCJR_LE_MAX 254
#define BCJR_W      8   /* alpha checkpoint spacing = backward window length; every QPP size is a multiple of 8 */
#define BCJR_BLK    32  /* beta block: next-iteration initialisation at every multiple of 32 steps */

static inline int bcjr_max(int a, int b) { return a > b ? a : b; }
static void bcjr_norm(int *v) /* subtract the maximum, floor at BCJR_NEG */
{
    int m = v[0];
    for (int s = 1; s < 8; s++) m = bcjr_max(m, v[s]);
    for (int s = 0; s < 8; s++) v[s] = bcjr_max(v[s] - m, BCJR_NEG);
}
static void bcjr_alpha_step(const int *a, int g00, int g01, int g10, int *o)
{
    o[0] = bcjr_max(a[0] + g00, a[1]);       o[4] = bcjr_max(a[0], a[1] + g00);
    o[1] = bcjr_max(a[2] + g10, a[3] + g01); o[5] = bcjr_max(a[2] + g01, a[3] + g10);
    o[2] = bcjr_max(a[4] + g01, a[5] + g10); o[6] = bcjr_max(a[4] + g10, a[5] + g01);
    o[3] = bcjr_max(a[6], a[7] + g00);       o[7] = bcjr_max(a[6] + g00, a[7]);
}
static void bcjr_beta_step(const int *b, int g00, int g01, int g10, int *o)
{
    o[0] = bcjr_max(b[0] + g00, b[4]);       o[1] = bcjr_max(b[0], b[4] + g00);
    o[2] = bcjr_max(b[1] + g10, b[5] + g01); o[3] = bcjr_max(b[1] + g01, b[5] + g10);
    o[4] = bcjr_max(b[2] + g01, b[6] + g10); o[5] = bcjr_max(b[2] + g10, b[6] + g01);
    o[6] = bcjr_max(b[3], b[7] + g00);       o[7] = bcjr_max(b[3] + g00, b[7]);
}
/* Segments: n_seg = 8, 4, 2 or 1 pieces of equal length (a multiple of 64 steps, at least 512) decoded by separate wavefronts */
uint32_t lo_bcjr_n_seg(uint32_t K)
{
    const uint32_t nblk = (K + 63) / 64;
    for (uint32_t n = 8; n > 1; n >>= 1)
        if (nblk % n == 0 && (nblk / n) * 64 >= 512) return n;
    return 1;
}
#define BCJR_MAX_BLKS 192 /* 6144 / 32 */
#define BCJR_MAX_SEGS 64 /* the one-block-per-wavefront mode: one segment per lane */
typedef struct { int16_t a[2][BCJR_MAX_SEGS][8], b[2][BCJR_MAX_BLKS][8]; } bcjr_bnd_t; /* [buffer][segment | block][state] */
/* Segment length of the "one code block per wavefront" mode (MI_LTE_TURBO_BCJR_BLOCK, k_bcjr_block): the block's 32-step beta blocks are
 * dealt to the 64 lanes of a wavefront, every lane a whole number of them -- 96 steps per lane at K = 6144, 32 at K <= 2048.  Same decoder,
 * same arithmetic; only the places where alpha restarts from the previous iteration's value are closer together. */
uint32_t lo_bcjr_block_seg_len(uint32_t K)
{
    const uint32_t n_blk = (K + BCJR_BLK - 1) / BCJR_BLK;
    return BCJR_BLK * ((n_blk + 63) / 64);
}
static int bcjr_range_ok = 1; /* cleared if any intermediate leaves int16 (checked by the tests through lo_bcjr_range_ok) */
int lo_bcjr_range_ok(void) { return bcjr_range_ok; }
#define BCJR_CHK(v) do { if ((v) > 32767 || (v) < -32768) bcjr_range_ok = 0; } while (0)

/* one SISO pass: systematic S, parity P (int8), stored a-priori halves Aq (int8; La = 2*Aq), the 3 termination pairs -> stored
 * extrinsic halves Eq and (optionally) the sign of the a-posteriori LLR (1 = negative = bit 1).  it = iteration number. */
static void bcjr_siso(const int8_t *S, const int8_t *P, const int8_t *Aq, const int8_t *tail_s, const int8_t *tail_p, uint32_t K,
                      int8_t *Eq, uint8_t *hard, bcjr_bnd_t *bnd, uint32_t it, uint32_t seg_len_mode)
{
    /* seg_len_mode = 0: the batch kernels' segmentation (n_seg = 8, 4, 2, 1); else the segment length in steps (a multiple of 32) */
    const uint32_t n_seg = seg_len_mode ? (K + seg_len_mode - 1) / seg_len_mode : lo_bcjr_n_seg(K);
    const uint32_t seg_len = seg_len_mode ? seg_len_mode : ((K + 63) / 64 / n_seg) * 64, rd = it & 1, wr = rd ^ 1, n_blk = (K + BCJR_BLK - 1) / BCJR_BLK;
    int t8[8];
    for (uint32_t sg = 0; sg < n_seg; sg++) {
        const uint32_t t_lo = sg * seg_len, t_hi = (t_lo + seg_len < K) ? t_lo + seg_len : K;
        int a[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
        if (sg > 0) for (int s = 0; s < 8; s++) a[s] = bnd->a[rd][sg][s];
        for (uint32_t b0 = t_lo; b0 < t_hi; b0 += BCJR_BLK) { /* one 32-step block: forward, then backward */
            const uint32_t b1 = (b0 + BCJR_BLK < t_hi) ? b0 + BCJR_BLK : t_hi, blk = b0 / BCJR_BLK, n_w = (b1 - b0) / BCJR_W;
            int chk[BCJR_BLK / BCJR_W][8];
            for (uint32_t t = b0; t < b1; t++) { /* alpha, normalised and checkpointed every W steps */
                if (t % BCJR_W == 0) {
                    bcjr_norm(a);
                    memcpy(chk[(t - b0) / BCJR_W], a, sizeof(a));
                }
                const int lsa = S[t] + 2 * Aq[t], lp = P[t];
                bcjr_alpha_step(a, lsa + lp, lsa, lp, t8);
                memcpy(a, t8, sizeof(a));
                for (int s = 0; s < 8; s++) BCJR_CHK(a[s]);
            }
            int b[8] = {0, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG, BCJR_NEG};
            if (blk + 1 == n_blk) { /* beta at step K from the termination: only the a = 0 edges exist, so state 2j+r3 continues to state j */
                for (int t = 2; t >= 0; t--) {
                    const int ls = tail_s[t], lp = tail_p[t], g00 = ls + lp, g01 = ls, g10 = lp;
                    t8[0] = b[0] + g00; t8[1] = b[0];       t8[2] = b[1] + g10; t8[3] = b[1] + g01;
                    t8[4] = b[2] + g01; t8[5] = b[2] + g10; t8[6] = b[3];       t8[7] = b[3] + g00;
                    memcpy(b, t8, sizeof(b));
                }
                bcjr_norm(b);
            } else
                for (int s = 0; s < 8; s++) b[s] = bnd->b[rd][blk][s]; /* what block blk+1 reached at its start in the previous iteration */
            for (int w = (int)n_w - 1; w >= 0; w--) { /* backward, one window at a time: re-run alpha from the checkpoint */
                const uint32_t t0 = b0 + (uint32_t)w * BCJR_W;
                int al[BCJR_W][8];
                memcpy(al[0], chk[w], sizeof(al[0]));
                for (int i = 1; i < BCJR_W; i++) {
                    const int lsa = S[t0 + i - 1] + 2 * Aq[t0 + i - 1], lp = P[t0 + i - 1];
                    bcjr_alpha_step(al[i - 1], lsa + lp, lsa, lp, al[i]);
                }
                for (int i = BCJR_W - 1; i >= 0; i--) {
                    const uint32_t t = t0 + i;
                    const int lsa = S[t] + 2 * Aq[t], lp = P[t], g00 = lsa + lp, g01 = lsa, g10 = lp;
                    const int *x = al[i];
                    /* u = 0 edges: (0->0) (1->4) (7->3) (6->7) carry g00, (3->1) (2->5) (4->2) (5->6) carry g01;
                     * u = 1 edges: (2->1) (3->5) (5->2) (4->6) carry g10, (1->0) (0->4) (6->3) (7->7) carry 0 */
                    const int m00 = bcjr_max(bcjr_max(x[0] + b[0], x[1] + b[4]), bcjr_max(x[7] + b[3], x[6] + b[7]));
                    const int m01 = bcjr_max(bcjr_max(x[3] + b[1], x[2] + b[5]), bcjr_max(x[4] + b[2], x[5] + b[6]));
                    const int m10 = bcjr_max(bcjr_max(x[2] + b[1], x[3] + b[5]), bcjr_max(x[5] + b[2], x[4] + b[6]));
                    const int m11 = bcjr_max(bcjr_max(x[1] + b[0], x[0] + b[4]), bcjr_max(x[6] + b[3], x[7] + b[7]));
                    const int llr = bcjr_max(m00 + g00, m01 + g01) - bcjr_max(m10 + g10, m11);
                    BCJR_CHK(m00 + g00); BCJR_CHK(m01 + g01); BCJR_CHK(m10 + g10); BCJR_CHK(m11); BCJR_CHK(llr); BCJR_CHK(llr - lsa);
                    int xx = llr - lsa;
                    xx     = xx > BCJR_X_MAX ? BCJR_X_MAX : xx < -BCJR_X_MAX ? -BCJR_X_MAX : xx;
                    int e  = (xx * 3) >> 2; /* extrinsic, scaled by 3/4 (arithmetic shift) */
                    e      = e > BCJR_LE_MAX ? BCJR_LE_MAX : e < -BCJR_LE_MAX ? -BCJR_LE_MAX : e;
                    Eq[t]  = (int8_t)(e >> 1);
                    if (hard) hard[t] = llr < 0 ? 1 : 0;
                    bcjr_beta_step(b, g00, g01, g10, t8);
                    memcpy(b, t8, sizeof(b));
                    for (int s = 0; s < 8; s++) BCJR_CHK(b[s]);
                }
                bcjr_norm(b);
            }
            if (blk > 0) for (int s = 0; s < 8; s++) bnd->b[wr][blk - 1][s] = (int16_t)b[s]; /* beta at this block's start, for block blk-1's next iteration */
        }
        if (sg + 1 < n_seg) {
            bcjr_norm(a);
            for (int s = 0; s < 8; s++) bnd->a[wr][sg + 1][s] = (int16_t)a[s];
        }
    }
}

static void turbo_decode_bcjr(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits, uint32_t seg_len_mode);
void lo_turbo_decode_bcjr(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits) { turbo_decode_bcjr(soft, K, n_iter, qpp_spec, c_bits, 0); }
/* the same decoder with the one-block-per-wavefront segmentation (lo_bcjr_block_seg_len) */
void lo_turbo_decode_bcjr_block(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits)
{
    turbo_decode_bcjr(soft, K, n_iter, qpp_spec, c_bits, lo_bcjr_block_seg_len(K));
}
static void turbo_decode_bcjr(const int16_t *soft, uint32_t K, uint32_t n_iter, int qpp_spec, uint8_t *c_bits, uint32_t seg_len_mode)
{
    int8_t   *S1 = (int8_t *)malloc(4 * (size_t)K), *P1 = S1 + K, *S2 = P1 + K, *P2 = S2 + K;
    int8_t   *A1 = (int8_t *)calloc(4 * (size_t)K, 1), *A2 = A1 + K, *E1 = A2 + K, *E2 = E1 + K;
    uint8_t  *hard = (uint8_t *)calloc(K, 1);
    uint16_t *pi = (uint16_t *)malloc(sizeof(uint16_t) * 2 * K), *inv = pi + K;
    int8_t    x[3 * 4];
    if (qpp_spec) lo_qpp_map_spec(K, pi); else lo_qpp_map_ref(K, pi);
    for (uint32_t j = 0; j < K; j++) inv[j] = 0xFFFF;
    for (uint32_t i = 0; i < K; i++) inv[pi[i]] = (uint16_t)i; /* last writer wins, like the reference's de-interleaver */
#define CLIP8(v) ((int8_t)((v) > 127 ? 127 : (v) < -127 ? -127 : (v)))
    for (uint32_t i = 0; i < K; i++) { S1[i] = CLIP8(soft[3 * i]); P1[i] = CLIP8(soft[3 * i + 1]); P2[i] = CLIP8(soft[3 * i + 2]); }
    for (uint32_t i = 0; i < K; i++) S2[i] = S1[pi[i]];
    for (uint32_t r = 0; r < 4; r++)
        for (uint32_t c = 0; c < 3; c++) x[3 * r + c] = CLIP8(soft[3 * (K + r) + c]); /* x[3r + stream] = d_stream[K + r] */
#undef CLIP8
    /* 36.212 5.1.3.2.2: d0 = x_K z_K+1 x'_K z'_K+1, d1 = z_K x_K+2 z'_K x'_K+2, d2 = x_K+1 z_K+2 x'_K+1 z'_K+2 */
    const int8_t t1s[3] = {x[0], x[2], x[4]}, t1p[3] = {x[1], x[3], x[5]};
    const int8_t t2s[3] = {x[6], x[8], x[10]}, t2p[3] = {x[7], x[9], x[11]};
    bcjr_bnd_t *bnd1 = (bcjr_bnd_t *)calloc(2, sizeof(bcjr_bnd_t)), *bnd2 = bnd1 + 1; /* all-zero (uniform) before the first iteration */
    for (uint32_t it = 0; it < n_iter; it++) {
        bcjr_siso(S1, P1, A1, t1s, t1p, K, E1, NULL, bnd1, it, seg_len_mode);
        for (uint32_t i = 0; i < K; i++) A2[i] = E1[pi[i]];
        bcjr_siso(S2, P2, A2, t2s, t2p, K, E2, it + 1 == n_iter ? hard : NULL, bnd2, it, seg_len_mode);
        for (uint32_t j = 0; j < K; j++) A1[j] = inv[j] != 0xFFFF ? E2[inv[j]] : 0;
    }
    for (uint32_t j = 0; j < K; j++) /* a hole of the (wrapped) de-interleaver falls back on the first decoder's view of that bit */
        c_bits[j] = inv[j] != 0xFFFF ? hard[inv[j]] : (uint8_t)(S1[j] + 2 * A1[j] < 0 ? 1 : 0);
    free(S1); free(A1); free(hard); free(pi); free(bnd1);
}

/* ------------------------------------------------------------------------------------------ */
/* rate matching (liblte_phy.cc:11081-11237 TX, :11246-11490 RX) as an index map over the       */
/* circular buffer w[0..3*K_pi): which (stream x, padded index n) sits at each w position.      */
typedef struct { uint32_t R, K_pi, N_d, K_w, N_cb, k0; } rm_geom_t;

static void rm_geometry(uint32_t D, uint32_t C, uint32_t tx_mode, uint32_t N_soft, uint32_t M, uint32_t chan,
                        uint32_t rv, rm_geom_t *g)
{
    uint32_t K_mimo = (tx_mode == 3 || tx_mode == 4 || tx_mode == 8 || tx_mode == 9) ? 2 : 1;
    uint32_t N_ir;
    g->R = 0;
    while (D > 32 * g->R) g->R++;
    g->K_pi = 32 * g->R;
    g->N_d  = g->K_pi - D;
    g->K_w  = 3 * g->K_pi;
    N_ir    = N_soft / (K_mimo * (M < 8 ? M : 8));
    if (chan == LO_CHAN_DLSCH || chan == LO_CHAN_PCH) g->N_cb = (N_ir / C < g->K_w) ? N_ir / C : g->K_w;
    else                                              g->N_cb = g->K_w;
    g->k0 = (uint32_t)((float)g->R * (2 * ceilf((float)g->N_cb / (float)(8 * g->R)) * (float)rv + 2));
}

/* stream and padded index held at circular-buffer position p */
static void rm_locate(const rm_geom_t *g, uint32_t p, uint32_t *x, uint32_t *n)
{
    if (p < g->K_pi) {
        *x = 0;
        *n = 32 * (p % g->R) + LTE_SUBBLOCK_COL_PERM[p / g->R];
    } else {
        uint32_t q = p - g->K_pi, i = q >> 1;
        if ((q & 1) == 0) { *x = 1; *n = 32 * (i % g->R) + LTE_SUBBLOCK_COL_PERM[i / g->R]; }
        else              { *x = 2; *n = (LTE_SUBBLOCK_COL_PERM[i / g->R] + 32 * (i % g->R) + 1) % g->K_pi; }
    }
}

uint32_t lo_rate_unmatch_turbo(const float *e, uint32_t E, uint32_t D, uint32_t C, uint32_t tx_mode,
                               uint32_t N_soft, uint32_t M, uint32_t chan, uint32_t rv, float *d)
{
    rm_geom_t g;
    uint32_t  k = 0, j = 0;
    rm_geometry(D, C, tx_mode, N_soft, M, chan, rv, &g);
    for (uint32_t i = 0; i < 3 * D; i++) d[i] = LO_RX_NULL;
    while (k < E) {
        uint32_t x, n;
        rm_locate(&g, (g.k0 + j) % g.N_cb, &x, &n);
        if (n >= g.N_d) { /* head padding is the only NULL the RX honours */
            float *slot = &d[(n - g.N_d) * 3 + x];
            if (*slot == LO_RX_NULL)      *slot = e[k];
            else if (e[k] != LO_RX_NULL) *slot += e[k];
            k++;
        }
        j++;
    }
    return 3 * D;
}

void lo_rate_match_turbo(const uint8_t *d_planar, uint32_t N_d_bits, uint32_t C, uint32_t tx_mode, uint32_t N_soft,
                         uint32_t M, uint32_t chan, uint32_t rv, uint32_t E, uint8_t *e)
{
    rm_geom_t g;
    uint32_t  D = N_d_bits / 3, k = 0, j = 0;
    rm_geometry(D, C, tx_mode, N_soft, M, chan, rv, &g);
    while (k < E) {
        uint32_t x, n;
        rm_locate(&g, (g.k0 + j) % g.N_cb, &x, &n);
        if (n >= g.N_d) {
            uint8_t v = d_planar[D * x + (n - g.N_d)];
            if (v != 100) e[k++] = v; /* TX_NULL_BIT fillers are skipped on the TX side only */
        }
        j++;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* PDSCH demodulation                                                                          */

/* get_soft_decision (liblte_phy.cc:13880-13900) */
static float soft_decision(float rx_re, float rx_im, float exp_re, float exp_im, float max_dist)
{
    float diff_re = rx_re - exp_re, diff_im = rx_im - exp_im;
    float dist    = sqrtf(diff_re * diff_re + diff_im * diff_im);
    if (dist >= (max_dist - (max_dist / 120))) dist = max_dist - (max_dist / 120);
    return max_dist - dist;
}

/* modulation_demapper (liblte_phy.cc:9502-9660).  16/64QAM are hard decisions (+-127). */
uint32_t lo_modulation_demapper(const float *d_re, const float *d_im, uint32_t M, uint32_t mod, int8_t *bits)
{
    const float r2   = 1 / sqrt(2);
    const float t10  = 2 / sqrt(10);
    const float t42  = 2 / sqrt(42);
    const float f42  = 4 / sqrt(42);
    const float s42  = 6 / sqrt(42);
    if (mod == LO_MOD_BPSK) {
        for (uint32_t i = 0; i < M; i++) {
            float ang = atan2f(d_im[i], d_re[i]);
            if ((ang > -M_PI / 4) && (ang < 3 * M_PI / 4)) bits[i] = +(int8_t)(127 * soft_decision(d_re[i], d_im[i], +r2, +r2, 1));
            else                                           bits[i] = -(int8_t)(127 * soft_decision(d_re[i], d_im[i], -r2, -r2, 1));
        }
        return M;
    }
    if (mod == LO_MOD_QPSK) {
        for (uint32_t i = 0; i < M; i++) {
            float ang = atan2f(d_im[i], d_re[i]);
            float er, ei;
            if ((ang >= 0) && (ang < M_PI / 2))        { er = +r2; ei = +r2; }
            else if ((ang >= -M_PI / 2) && (ang < 0))  { er = +r2; ei = -r2; }
            else if ((ang >= M_PI / 2) && (ang < M_PI)) { er = -r2; ei = +r2; }
            else                                       { er = -r2; ei = -r2; }
            int8_t m        = (int8_t)(127 * soft_decision(d_re[i], d_im[i], er, ei, 1));
            bits[i * 2 + 0] = (er > 0) ? m : (int8_t)-m;
            bits[i * 2 + 1] = (ei > 0) ? m : (int8_t)-m;
        }
        return 2 * M;
    }
    if (mod == LO_MOD_16QAM) {
        for (uint32_t i = 0; i < M; i++) {
            bits[i * 4 + 0] = (d_re[i] > 0) ? 127 : -127;
            bits[i * 4 + 1] = (d_im[i] > 0) ? 127 : -127;
            bits[i * 4 + 2] = (fabsf(d_re[i]) < t10) ? 127 : -127;
            bits[i * 4 + 3] = (fabsf(d_im[i]) < t10) ? 127 : -127;
        }
        return 4 * M;
    }
    for (uint32_t i = 0; i < M; i++) { /* 64QAM */
        float ar = fabsf(d_re[i]), ai = fabsf(d_im[i]);
        bits[i * 6 + 0] = (d_re[i] > 0) ? 127 : -127;
        bits[i * 6 + 1] = (d_im[i] > 0) ? 127 : -127;
        if (ar < f42) { bits[i * 6 + 2] = 127;  bits[i * 6 + 4] = (ar > t42) ? 127 : -127; }
        else          { bits[i * 6 + 2] = -127; bits[i * 6 + 4] = (ar < s42) ? 127 : -127; }
        if (ai < f42) { bits[i * 6 + 3] = 127;  bits[i * 6 + 5] = (ai > t42) ? 127 : -127; }
        else          { bits[i * 6 + 3] = -127; bits[i * 6 + 5] = (ai < s42) ? 127 : -127; }
    }
    return 6 * M;
}

/* pre_decoder_and_matched_filter_dl (liblte_phy.cc:7645-7796).  x is laid out [N_ant][M_ap/N_ant]
 * exactly as the reference's x_re_ptr[p] = &x_re[p*(M_ap_symb/N_ant)].  Returns M_layer_symb. */
uint32_t lo_pre_decoder_dl(const float *y_re, const float *y_im, const float *h_re, const float *h_im,
                           uint32_t h_len, uint32_t M_ap, uint32_t N_ant, float *x_re, float *x_im)
{
    const uint32_t xs = M_ap / N_ant;
#define H_RE(p, i) h_re[(p) * h_len + (i)]
#define H_IM(p, i) h_im[(p) * h_len + (i)]
    if (N_ant == 1) {
        for (uint32_t i = 0; i < M_ap; i++) {
            float hn = H_RE(0, i) * H_RE(0, i) + H_IM(0, i) * H_IM(0, i);
            x_re[i]  = (y_re[i] * H_RE(0, i) + y_im[i] * H_IM(0, i)) / hn;
            x_im[i]  = (y_im[i] * H_RE(0, i) - y_re[i] * H_IM(0, i)) / hn;
        }
        return M_ap;
    }
    if (N_ant == 2) {
        uint32_t M = M_ap / 2;
        for (uint32_t i = 0; i < M; i++) {
            float h0r = H_RE(0, 2 * i), h0i = H_IM(0, 2 * i), h1r = H_RE(1, 2 * i), h1i = H_IM(1, 2 * i);
            float y0r = y_re[2 * i], y0i = y_im[2 * i], y1r = y_re[2 * i + 1], y1i = y_im[2 * i + 1];
            float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i;
            float hn = sqrtf(a0 * a0 + a1 * a1); /* |h|^2 squared again: reference quirk Q4 (:7700) */
            x_re[i]      = (h0r * y0r + h0i * y0i + h1r * y1r + h1i * y1i) / hn;
            x_im[i]      = (h0r * y0i - h0i * y0r - h1r * y1i + h1i * y1r) / hn;
            x_re[xs + i] = (-h1r * y0r - h1i * y0i + h0r * y1r + h0i * y1i) / hn;
            x_im[xs + i] = (h1r * y0i - h1i * y0r + h0r * y1i - h0i * y1r) / hn;
        }
        return M;
    }
    { /* N_ant == 4 (:7718-7794) */
        uint32_t M = M_ap / 4, i;
        for (i = 0; i < M; i++) {
            float h0r = H_RE(0, 4 * i), h0i = H_IM(0, 4 * i), h2r = H_RE(2, 4 * i), h2i = H_IM(2, 4 * i);
            float h1r = H_RE(1, 4 * i + 2), h1i = H_IM(1, 4 * i + 2), h3r = H_RE(3, 4 * i + 2), h3i = H_IM(3, 4 * i + 2);
            float y0r = y_re[4 * i], y0i = y_im[4 * i], y1r = y_re[4 * i + 1], y1i = y_im[4 * i + 1];
            float y2r = y_re[4 * i + 2], y2i = y_im[4 * i + 2], y3r = y_re[4 * i + 3], y3i = y_im[4 * i + 3];
            float a0 = h0r * h0r + h0i * h0i, a1 = h1r * h1r + h1i * h1i, a2 = h2r * h2r + h2i * h2i,
                  a3 = h3r * h3r + h3i * h3i;
            float n02 = sqrtf(a0 * a0 + a2 * a2), n13 = sqrtf(a1 * a1 + a3 * a3);
            x_re[i]          = (h0r * y0r + h0i * y0i + h2r * y1r + h2i * y1i) / n02;
            x_im[i]          = (h0r * y0i - h0i * y0r - h2r * y1i + h2i * y1r) / n02;
            x_re[xs + i]     = (-h2r * y0r - h2i * y0i + h0r * y1r + h0i * y1i) / n02;
            x_im[xs + i]     = -(-h2r * y0i + h2i * y0r - h0r * y1i + h0i * y1r) / n02;
            x_re[2 * xs + i] = (h1r * y2r + h1i * y2i + h3r * y3r + h3i * y3i) / n13;
            x_im[2 * xs + i] = (h1r * y2i - h1i * y2r - h3r * y3i + h3i * y3r) / n13;
            x_re[3 * xs + i] = (-h3r * y2r - h3i * y2i + h1r * y3r + h1i * y3i) / n13;
            x_im[3 * xs + i] = -(-h3r * y2i + h3i * y2r - h1r * y3i + h1i * y3r) / n13;
        }
        if ((M_ap % 4) != 0) { /* asymmetric tail (:7766-7794) */
            float h0r = H_RE(0, 4 * i), h0i = H_IM(0, 4 * i), h2r = H_RE(2, 4 * i), h2i = H_IM(2, 4 * i);
            float y0r = y_re[4 * i], y0i = y_im[4 * i], y1r = y_re[4 * i + 1], y1i = y_im[4 * i + 1];
            float a0 = h0r * h0r + h0i * h0i, a2 = h2r * h2r + h2i * h2i, n02 = sqrtf(a0 * a0 + a2 * a2);
            x_re[i]      = (h0r * y0r + h0i * y0i + h2r * y1r + h2i * y1i) / n02;
            x_im[i]      = (h0r * y0i - h0i * y0r - h2r * y1i + h2i * y1r) / n02;
            x_re[xs + i] = (-h2r * y0r - h2i * y0i + h0r * y1r + h0i * y1i) / n02;
            x_im[xs + i] = (-h2r * y0i + h2i * y0r - h0r * y1r + h0i * y1i) / n02;
            x_re[2 * xs + i] = x_im[2 * xs + i] = x_re[3 * xs + i] = x_im[3 * xs + i] = LO_RX_NULL;
            return (M_ap + 2) / 4;
        }
        return M;
    }
#undef H_RE
#undef H_IM
}

/* layer_demapper_dl (liblte_phy.cc:7473-7514): NOTE the reference indexes x with stride
 * M_layer_symb here although the pre-decoder wrote it with stride M_ap_symb/N_ant; the two are
 * equal except in the 4-port tail case. */
uint32_t lo_layer_demapper_dl(const float *x_re, const float *x_im, uint32_t M_layer, uint32_t N_ant, float *d_re,
                              float *d_im)
{
    uint32_t M_symb = M_layer * N_ant;
    if (N_ant == 4 && x_re[2 * M_layer + M_layer - 1] == LO_RX_NULL && x_im[2 * M_layer + M_layer - 1] == LO_RX_NULL &&
        x_re[3 * M_layer + M_layer - 1] == LO_RX_NULL && x_im[3 * M_layer + M_layer - 1] == LO_RX_NULL)
        M_symb -= 2;
    for (uint32_t i = 0; i < M_layer; i++)
        for (uint32_t p = 0; p < N_ant; p++) {
            d_re[i * N_ant + p] = x_re[p * M_layer + i];
            d_im[i * N_ant + p] = x_im[p * M_layer + i];
        }
    return M_symb;
}

/* PBCH/PSS/SSS sub-carrier window (liblte_phy.cc:3722-3742) */
static void sync_window(uint32_t N_rb_dl, uint32_t *first_sc, uint32_t *last_sc)
{
    switch (N_rb_dl) {
    case 6:  *first_sc = 0;           *last_sc = 6 * 12 - 1;  break;
    case 15: *first_sc = 4 * 12 + 6;  *last_sc = 11 * 12 - 7; break;
    case 25: *first_sc = 9 * 12 + 6;  *last_sc = 16 * 12 - 7; break;
    case 50: *first_sc = 22 * 12;     *last_sc = 28 * 12 - 1; break;
    case 75: *first_sc = 34 * 12 + 6; *last_sc = 41 * 12 - 7; break;
    default: *first_sc = 47 * 12;     *last_sc = 53 * 12 - 1; break;
    }
}

/* is RE (symbol L, sub-carrier sc = prb*12 + j) excluded from PDSCH?  (liblte_phy.cc:3753-3789) */
static int pdsch_re_skipped(uint32_t N_ant, uint32_t N_id_cell, uint32_t sf_num, uint32_t L, uint32_t j, uint32_t sc,
                            uint32_t first_sc, uint32_t last_sc)
{
    if (N_ant == 1 && (L % 7) == 0 && (N_id_cell % 6) == (j % 6)) return 1;
    if (N_ant == 1 && (L % 7) == 4 && ((N_id_cell + 3) % 6) == (j % 6)) return 1;
    if ((N_ant == 2 || N_ant == 4) && ((L % 7) == 0 || (L % 7) == 4) && (N_id_cell % 3) == (j % 3)) return 1;
    if (N_ant == 4 && (L % 7) == 1 && (N_id_cell % 3) == (j % 3)) return 1;
    if (sf_num == 0 && sc >= first_sc && sc <= last_sc && L >= 7 && L <= 10) return 1;
    if ((sf_num == 0 || sf_num == 5) && sc >= first_sc && sc <= last_sc && (L == 6 || L == 5)) return 1;
    return 0;
}

uint32_t lo_pdsch_extract(const lo_cfg_t *cfg, const lo_subframe_t *sf, const lo_alloc_t *alloc, uint32_t N_pdcch_symbs,
                          uint32_t N_id_cell, uint32_t N_ant, float *y_re, float *y_im, float *c_re, float *c_im,
                          uint32_t cap)
{
    uint32_t first_sc, last_sc, idx = 0;
    sync_window(cfg->N_rb_dl, &first_sc, &last_sc);
    for (uint32_t L = N_pdcch_symbs; L < 14; L++)
        for (uint32_t pi = 0; pi < alloc->N_prb; pi++) {
            uint32_t prb = alloc->prb[pi];
            for (uint32_t j = 0; j < 12; j++) {
                uint32_t sc = prb * 12 + j;
                if (pdsch_re_skipped(N_ant, N_id_cell, sf->num, L, j, sc, first_sc, last_sc)) continue;
                if (idx >= cap) return idx;
                y_re[idx] = sf->rx_symb_re[L][sc];
                y_im[idx] = sf->rx_symb_im[L][sc];
                for (uint32_t p = 0; p < N_ant; p++) {
                    c_re[p * cap + idx] = sf->rx_ce_re[p][L][sc];
                    c_im[p * cap + idx] = sf->rx_ce_im[p][L][sc];
                }
                idx++;
            }
        }
    return idx;
}

/* ------------------------------------------------------------------------------------------ */
/* DL-SCH decode for the single-code-block envelope                                */

/* code block segmentation arithmetic (liblte_phy.cc:9779-9825; same in :9902-9946) */
int lo_segmentation_params(uint32_t B, uint32_t *C, uint32_t *F, uint32_t *K_plus, uint32_t *K_minus, uint32_t *C_minus)
{
    uint32_t L, Bp, Kp = 0, Km = 0, Cm = 0, Cc;
    if (B <= 6144) { L = 0; Cc = 1; Bp = B; }
    else           { L = 24; Cc = (uint32_t)ceilf((float)B / (float)(6144 - L)); Bp = B + Cc * L; }
    for (int r = 0; r < LTE_QPP_N_SIZES; r++)
        if (Cc * LTE_QPP_ROWS[r].K >= Bp) { Kp = LTE_QPP_ROWS[r].K; break; }
    if (Cc > 1) {
        for (int r = LTE_QPP_N_SIZES - 1; r >= 0; r--)
            if (LTE_QPP_ROWS[r].K < Kp) { Km = LTE_QPP_ROWS[r].K; break; }
        Cm = (Cc * Kp - Bp) / (Kp - Km);
    }
    *C = Cc; *K_plus = Kp; *K_minus = Km; *C_minus = Cm;
    *F = (Cc - Cm) * Kp + Cm * Km - Bp;
    return 0;
}

/* dlsch_channel_decode (liblte_phy.cc:12762-12872) restricted to C == 1: the C > 1 path of the
 * reference is broken (row-0 writes, stride literal 18432 vs 18528) and is outside the
 * envelope this oracle covers -- it returns LO_ERR_INVALID_INPUTS there.
 * With C == 1, code_block_deconcatenation (:11795-11891) always hands all N_in_bits to block 0. */
int lo_dlsch_channel_decode(const float *in_bits, uint32_t N_in_bits, uint32_t tbs, uint32_t tx_mode, uint32_t rv_idx,
                            uint32_t M_dl_harq, uint32_t N_soft, uint8_t *out_bits, uint32_t *N_out_bits,
                            uint8_t *c_bits_tap)
{
    uint32_t C, F, K, Km, Cm;
    uint8_t  p_calc[24];
    lo_segmentation_params(tbs + 24, &C, &F, &K, &Km, &Cm);
    if (C != 1 || K == 0) return LO_ERR_INVALID_INPUTS;
    float   *d = (float *)malloc(sizeof(float) * 3 * (K + 4));
    uint8_t *c = (uint8_t *)malloc(K);
    int      err = LO_ERR_INVALID_CRC;
    lo_rate_unmatch_turbo(in_bits, N_in_bits, K + 4, 1, tx_mode, N_soft, M_dl_harq, LO_CHAN_DLSCH, rv_idx, d);
    lo_turbo_decode_ref(d, K, c);
    if (c_bits_tap) memcpy(c_bits_tap, c, K);
    /* desegmentation (:9948-9986): drop the F leading filler positions; b = a (tbs) | p (24) */
    lo_crc24a(c + F, tbs, p_calc);
    if (0 == memcmp(p_calc, c + F + tbs, 24)) {
        memcpy(out_bits, c + F, tbs);
        *N_out_bits = tbs;
        err         = LO_SUCCESS;
    }
    free(d); free(c);
    return err;
}

/* ------------------------------------------------------------------------------------------ */
/* front end                                                                                   */

/* unnormalised forward DFT, float64 radix-2 (FFTW3f stand-in; the reference pins nothing tighter
 * than the DFT definition) */
static void dft_forward(const float *in_re, const float *in_im, uint32_t n, float *out_re, float *out_im)
{
    double  *xr = (double *)malloc(sizeof(double) * 2 * n), *xi = xr + n;
    uint32_t bits = 0;
    while ((1u << bits) < n) bits++;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = 0, v = i;
        for (uint32_t b = 0; b < bits; b++) { r = (r << 1) | (v & 1u); v >>= 1; }
        xr[r] = in_re[i];
        xi[r] = in_im[i];
    }
    for (uint32_t len = 2; len <= n; len <<= 1) {
        uint32_t half = len >> 1;
        for (uint32_t j = 0; j < half; j++) {
            double a = 2.0 * M_PI * (double)(j * (n / len)) / (double)n;
            double wr = cos(a), wi = -sin(a);
            for (uint32_t base = 0; base < n; base += len) {
                uint32_t p = base + j, q = p + half;
                double   tr = xr[q] * wr - xi[q] * wi, ti = xr[q] * wi + xi[q] * wr;
                xr[q] = xr[p] - tr; xi[q] = xi[p] - ti;
                xr[p] = xr[p] + tr; xi[p] = xi[p] + ti;
            }
        }
    }
    for (uint32_t i = 0; i < n; i++) { out_re[i] = (float)xr[i]; out_im[i] = (float)xi[i]; }
    free(xr);
}

/* samples_to_symbols_dl (liblte_phy.cc:8593-8644), scale = 0.  The FFT window starts at
 * index + CP_len - 1, one sample early (quirk Q1, :8621). */
void lo_samples_to_symbols_dl(const lo_cfg_t *cfg, const float *samps_re, const float *samps_im, uint32_t slot_start_idx,
                              uint32_t symbol_offset, float *symb_re, float *symb_im)
{
    const uint32_t N = cfg->N_samps_per_symb, half = cfg->FFT_size / 2 - cfg->FFT_pad_size;
    uint32_t CP_len = (symbol_offset % 7 == 0) ? cfg->N_samps_cp_l_0 : cfg->N_samps_cp_l_else;
    uint32_t index  = slot_start_idx + (N + cfg->N_samps_cp_l_else) * symbol_offset;
    float   *o_re = (float *)malloc(sizeof(float) * 2 * N), *o_im = o_re + N;
    if (symbol_offset > 0) index += cfg->N_samps_cp_l_0 - cfg->N_samps_cp_l_else;
    dft_forward(samps_re + index + CP_len - 1, samps_im + index + CP_len - 1, N, o_re, o_im);
    for (uint32_t i = 0; i < half; i++) {
        symb_re[i + half]     = o_re[i + 1];     /* positive spectrum: bins 1..half        */
        symb_im[i + half]     = o_im[i + 1];
        symb_re[half - i - 1] = o_re[N - i - 1]; /* negative spectrum: bins N-1 .. N-half  */
        symb_im[half - i - 1] = o_im[N - i - 1];
    }
    free(o_re);
}

/* wrap_phase (liblte_phy.cc:14105-14116): comparisons against the double constant M_PI,
 * the subtraction of 2*M_PI is done in double and rounded back to float. */
static void wrap_phase(float *phase_1, float phase_2)
{
    while ((*phase_1 - phase_2) >= M_PI) *phase_1 = *phase_1 - 2 * M_PI;
    while ((*phase_1 - phase_2) <= -M_PI) *phase_1 = *phase_1 + 2 * M_PI;
}

/* liblte_phy_get_dl_subframe_and_ce (liblte_phy.cc:5905-6200) */
int lo_get_dl_subframe_and_ce(const lo_cfg_t *cfg, const float *i_samps, const float *q_samps, uint32_t frame_start_idx,
                              uint32_t subfr_num, uint32_t N_id_cell, uint32_t N_ant, lo_subframe_t *sf)
{
    static const uint32_t SYM01[5] = {0, 4, 7, 11, 14}, SYM23[3] = {1, 8, 15};
    static const uint32_t V[4][5] = {{0, 3, 0, 3, 0}, {3, 0, 3, 0, 3}, {0, 3, 0, 0, 0}, {3, 6, 3, 0, 0}};
    const uint32_t N_sc = cfg->N_rb_dl * 12, v_shift = N_id_cell % 6;
    const uint32_t subfr_start = frame_start_idx + subfr_num * cfg->N_samps_per_subfr;
    float(*crs_re)[220] = (float(*)[220])malloc(sizeof(float) * 2 * 16 * 220), (*crs_im)[220] = crs_re + 16;
    float(*mag)[LO_N_SC_MAX] = (float(*)[LO_N_SC_MAX])malloc(sizeof(float) * 2 * 5 * LO_N_SC_MAX), (*ang)[LO_N_SC_MAX] = mag + 5;

    if (!(N_ant == 1 || N_ant == 2 || N_ant == 4)) { free(crs_re); free(mag); return LO_ERR_INVALID_INPUTS; }
    sf->num = subfr_num;
    for (uint32_t i = 0; i < 16; i++) /* 14 symbols + 2 look-ahead (:5946-5957) */
        lo_samples_to_symbols_dl(cfg, i_samps, q_samps, subfr_start + (i / 7) * cfg->N_samps_per_slot, i % 7,
                                 sf->rx_symb_re[i], sf->rx_symb_im[i]);

    { /* CRS for the symbols that carry pilots (:5960-5967) */
        static const uint32_t ls[8] = {0, 1, 4, 7, 8, 11, 14, 15};
        for (int q = 0; q < 8; q++)
            lo_generate_crs((subfr_num * 2 + ls[q] / 7) % 20, ls[q] % 7, N_id_cell, crs_re[ls[q]], crs_im[ls[q]]);
    }

    for (uint32_t p = 0; p < N_ant; p++) {
        const uint32_t *sym  = (p < 2) ? SYM01 : SYM23;
        const uint32_t N_sym = (p < 2) ? 5 : 3;
        /* frequency direction (:6016-6064) */
        for (uint32_t i = 0; i < N_sym; i++) {
            const float *s_re = sf->rx_symb_re[sym[i]], *s_im = sf->rx_symb_im[sym[i]];
            const float *r_re = crs_re[sym[i]], *r_im = crs_im[sym[i]];
            const uint32_t off = (V[p][i] + v_shift) % 6;
            float   frac_mag = 0, frac_ang = 0;
            uint32_t k = 0;
            for (uint32_t j = 0; j < 2 * cfg->N_rb_dl; j++) {
                uint32_t m_prime = j + 110 - cfg->N_rb_dl;
                k = 6 * j + off;
                float t_re = s_re[k] * r_re[m_prime] + s_im[k] * r_im[m_prime];
                float t_im = s_im[k] * r_re[m_prime] - s_re[k] * r_im[m_prime];
                mag[i][k] = sqrtf(t_re * t_re + t_im * t_im);
                ang[i][k] = atan2f(t_im, t_re);
                if (j > 0) {
                    wrap_phase(&ang[i][k], ang[i][k - 6]);
                    frac_mag = (mag[i][k] - mag[i][k - 6]) / 6;
                    frac_ang = (ang[i][k] - ang[i][k - 6]) / 6;
                    for (uint32_t z = 1; z < 6; z++) {
                        mag[i][k - z] = mag[i][k - (z - 1)] - frac_mag;
                        ang[i][k - z] = ang[i][k - (z - 1)] - frac_ang;
                    }
                }
                if (j == 1) /* below the first pilot: continue the first slope (quirk Q2) */
                    for (uint32_t z = 1; z < off + 1; z++) {
                        mag[i][k - 6 - z] = mag[i][k - 6 - (z - 1)] - frac_mag;
                        ang[i][k - 6 - z] = ang[i][k - 6 - (z - 1)] - frac_ang;
                    }
            }
            for (uint32_t z = 1; z < (5 - off) + 1; z++) { /* above the last pilot: last slope */
                mag[i][k + z] = mag[i][k + (z - 1)] - frac_mag;
                ang[i][k + z] = ang[i][k + (z - 1)] - frac_ang;
            }
        }
        /* time direction */
        for (uint32_t j = 0; j < N_sc; j++) {
            float(*ce_re)[LO_N_SC_MAX] = sf->rx_ce_re[p], (*ce_im)[LO_N_SC_MAX] = sf->rx_ce_im[p];
            float fm, fa, cm, ca;
#define EMIT(z, m, a) do { ce_re[z][j] = (m) * cosf(a); ce_im[z][j] = (m) * sinf(a); } while (0)
#define SLOPE(hi, lo, div) do { fm = (mag[hi][j] - mag[lo][j]) / (div); wrap_phase(&ang[hi][j], ang[lo][j]); \
                                fa = ang[hi][j] - ang[lo][j]; wrap_phase(&fa, 0); fa /= (div); } while (0)
            if (N_sym == 3) { /* ports 2/3 (:6067-6115) */
                EMIT(1, mag[0][j], ang[0][j]);
                EMIT(8, mag[1][j], ang[1][j]);
                SLOPE(1, 0, 7);
                cm = mag[1][j]; ca = ang[1][j];
                for (int z = 7; z > 1; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
                cm = mag[0][j] - fm; ca = ang[0][j] - fa; /* symbol 0: extrapolated with the 1->8 slope (Q3) */
                EMIT(0, cm, ca);
                SLOPE(2, 1, 7);
                cm = mag[2][j] - fm; ca = ang[2][j] - fa;
                for (int z = 13; z > 8; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
            } else { /* ports 0/1 (:6117-6192) */
                EMIT(0, mag[0][j], ang[0][j]);
                EMIT(4, mag[1][j], ang[1][j]);
                EMIT(7, mag[2][j], ang[2][j]);
                EMIT(11, mag[3][j], ang[3][j]);
                SLOPE(1, 0, 4);
                cm = mag[1][j]; ca = ang[1][j];
                for (int z = 3; z > 0; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
                SLOPE(2, 1, 3);
                cm = mag[2][j]; ca = ang[2][j];
                for (int z = 6; z > 4; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
                SLOPE(3, 2, 4);
                cm = mag[3][j]; ca = ang[3][j];
                for (int z = 10; z > 7; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
                SLOPE(4, 3, 3);
                cm = mag[4][j]; ca = ang[4][j];
                for (int z = 13; z > 11; z--) { cm -= fm; ca -= fa; EMIT(z, cm, ca); }
            }
#undef EMIT
#undef SLOPE
        }
    }
    free(crs_re);
    free(mag);
    return LO_SUCCESS;
}

/* liblte_phy_pdsch_channel_decode (liblte_phy.cc:3690-3853).  The reference's static buffers cap
 * one allocation at 5000 REs / 10000 soft bits (liblte_phy.h:349-363); larger allocations are
 * outside the envelope the unmodified reference can run. */
int lo_pdsch_channel_decode(const lo_cfg_t *cfg, const lo_subframe_t *sf, const lo_alloc_t *alloc, uint32_t N_pdcch_symbs,
                            uint32_t N_id_cell, uint32_t N_ant, uint8_t *out_bits, uint32_t *N_out_bits,
                            int8_t *soft_tap, uint32_t *N_soft_tap)
{
    /* the reference's scratch arrays hold 5000 resource elements (liblte_phy.h: LIBLTE_PHY_PDSCH... sizes); larger allocations
     * overrun them there.  The restatement sizes its scratch from the allocation so that it can also check the "big" variant
     * of SURVEY 8d W4 (one 100-PRB allocation), which the unmodified reference cannot run. */
    const uint32_t CAP = alloc->N_prb * 168u + 8u > 5000u ? alloc->N_prb * 168u + 8u : 5000u, NS = 2 * CAP, NB = 6 * CAP + 64;
    float   *buf = (float *)malloc(sizeof(float) * ((size_t)2 * CAP + 2 * 4 * CAP + 4 * NS + NB));
    float   *y_re = buf, *y_im = y_re + CAP, *c_re = y_im + CAP, *c_im = c_re + 4 * CAP, *x_re = c_im + 4 * CAP,
            *x_im = x_re + NS, *d_re = x_im + NS, *d_im = d_re + NS, *desc = d_im + NS;
    int8_t  *soft = (int8_t *)malloc(NB);
    uint8_t *c    = (uint8_t *)malloc(NB);
    uint32_t idx, M_layer, M_symb, N_bits, c_init;
    int      err = LO_ERR_DECODE_FAIL;

    if (N_id_cell > 503) { free(buf); free(soft); free(c); return LO_ERR_INVALID_INPUTS; }
    idx = lo_pdsch_extract(cfg, sf, alloc, N_pdcch_symbs, N_id_cell, N_ant, y_re, y_im, c_re, c_im, CAP);
    M_layer = lo_pre_decoder_dl(y_re, y_im, c_re, c_im, CAP, idx, N_ant, x_re, x_im);
    M_symb  = lo_layer_demapper_dl(x_re, x_im, M_layer, N_ant, d_re, d_im);
    N_bits  = lo_modulation_demapper(d_re, d_im, M_symb, alloc->mod_type, soft);
    if (soft_tap) { memcpy(soft_tap, soft, N_bits); *N_soft_tap = N_bits; }
    c_init = (alloc->rnti << 14) | (0 << 13) | (sf->num << 9) | N_id_cell; /* (:3831) */
    lo_prs_c(c_init, N_bits, c);
    for (uint32_t i = 0; i < N_bits; i++) desc[i] = (float)soft[i] * (1 - 2 * (float)c[i]);
    if (LO_SUCCESS == lo_dlsch_channel_decode(desc, N_bits, alloc->tbs, alloc->tx_mode, alloc->rv_idx, 8, 250368,
                                              out_bits, N_out_bits, NULL))
        err = LO_SUCCESS;
    free(buf); free(soft); free(c);
    return err;
}

/* the host libm's atan2f over arrays: what tests pin the product's restatement of it to (mi_lte_model_atan2f) */
void lo_libm_atan2f(const float *y, const float *x, float *out, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) out[i] = atan2f(y[i], x[i]);
}
