/*
 * fftw_shim_f32.c -- TEST INFRASTRUCTURE ONLY: the TIMING stand-in for FFTW3f (see fftw3.h in this directory).
 *
 * fftw_shim.c (float64 radix-2, the one every parity test and golden vector was made with) is slower than the single-precision FFTW3f
 * the reference links, which made bench.py's cpu_baseline pessimistic wherever the transform is a large share of the call (the front
 * end).  This file is what bench.py's cpu_baseline legs time instead (oracle/_ref/libref_oracle_f32fft.so): the same seven FFTW names
 * behind a single-precision transform of the kind FFTW itself would pick --
 *   power-of-two sizes: Stockham autosort, radix 4 (one radix-2 pass when log2 n is odd), twiddles from a table, no bit reversal;
 *   other sizes (12*N_prb DFTs of the uplink, the 839-point PRACH transform): decimation-in-time mixed radix 4 / 2 / 3 / 5 with a
 *   generic O(p^2) butterfly for any other prime factor.
 * Scalar C, gcc -O3 without -march (the library travels to another host).  No parity claim is made through this file: tests only
 * check that it agrees with fftw_shim.c to single-precision rounding (tests/test_oracle.py).
 */
#include "fftw3.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_FACTORS 32

struct oracle_fftwf_plan_s {
    int            n;
    int            sign;
    int            pow2;
    fftwf_complex *in;
    fftwf_complex *out;
    float         *tw;   /* n complex entries: exp(sign * 2*pi*i*k/n) */
    float         *work; /* n complex entries */
    int            factors[2 * MAX_FACTORS]; /* (radix, remaining length) pairs of the mixed-radix path */
};

void *fftwf_malloc(size_t n) { return calloc(1, n ? n : 1); }
void  fftwf_free(void *p) { free(p); }

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags)
{
    (void)flags;
    struct oracle_fftwf_plan_s *p = (struct oracle_fftwf_plan_s *)calloc(1, sizeof(*p));
    p->n    = n;
    p->sign = sign < 0 ? -1 : 1;
    p->in   = in;
    p->out  = out;
    p->pow2 = (n > 0) && ((n & (n - 1)) == 0);
    p->tw   = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    p->work = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    for (int k = 0; k < n; k++) {
        const double a = 2.0 * M_PI * (double)k / (double)n;
        p->tw[2 * k]     = (float)cos(a);
        p->tw[2 * k + 1] = (float)((double)p->sign * sin(a));
    }
    /* factor n: 4s first, then 2, 3, 5, then whatever is left */
    int m = n, nf = 0, r = 4;
    while (m > 1 && nf < MAX_FACTORS) {
        while (m % r) {
            if (r == 4) r = 2;
            else if (r == 2) r = 3;
            else r += 2;
            if ((long)r * r > m) r = m;
        }
        m /= r;
        p->factors[2 * nf]     = r;
        p->factors[2 * nf + 1] = m;
        nf++;
    }
    return p;
}

void fftwf_destroy_plan(fftwf_plan p)
{
    if (!p) return;
    free(p->tw);
    free(p->work);
    free(p);
}

/* ---- power of two: Stockham, x -> y -> x ..., s = stride of the already-transformed index ---- */
static void stockham_pow2(const fftwf_plan p)
{
    const int    n = p->n;
    const float  sg = (float)p->sign; /* -1 forward: multiply by -i is (re, im) -> (im, -re) */
    const float *tw = p->tw;
    float       *x = p->work, *y = (float *)p->out;
    /* the pass count decides where the result lands: start so that it ends in out */
    int passes = 0;
    for (int m = n; m > 1; m = (m % 4 == 0) ? m / 4 : m / 2) passes++;
    if (passes % 2 == 0) { x = (float *)p->out; y = p->work; }
    memcpy(x, p->in, sizeof(float) * 2 * (size_t)n);
    int m = n, s = 1;
    while (m > 1) {
        if (m % 4 == 0) {
            const int m1 = m / 4, tstep = n / m;
            for (int q = 0; q < m1; q++) {
                const float w1r = tw[2 * (q * tstep)], w1i = tw[2 * (q * tstep) + 1];
                const float w2r = tw[2 * (2 * q * tstep)], w2i = tw[2 * (2 * q * tstep) + 1];
                const float w3r = tw[2 * (3 * q * tstep)], w3i = tw[2 * (3 * q * tstep) + 1];
                const float *a = x + 2 * (size_t)s * q, *b = a + 2 * (size_t)s * m1, *c = b + 2 * (size_t)s * m1, *d = c + 2 * (size_t)s * m1;
                float       *o = y + 2 * (size_t)s * 4 * q;
                for (int k = 0; k < s; k++) {
                    const float ar = a[2 * k], ai = a[2 * k + 1], br = b[2 * k], bi = b[2 * k + 1];
                    const float cr = c[2 * k], ci = c[2 * k + 1], dr = d[2 * k], di = d[2 * k + 1];
                    const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
                    const float bpdr = br + dr, bpdi = bi + di;
                    /* sign * i * (b - d) */
                    const float jr = -sg * (bi - di), ji = sg * (br - dr);
                    const float t1r = amcr + jr, t1i = amci + ji, t2r = apcr - bpdr, t2i = apci - bpdi, t3r = amcr - jr, t3i = amci - ji;
                    o[2 * k]               = apcr + bpdr;
                    o[2 * k + 1]           = apci + bpdi;
                    o[2 * (s + k)]         = t1r * w1r - t1i * w1i;
                    o[2 * (s + k) + 1]     = t1r * w1i + t1i * w1r;
                    o[2 * (2 * s + k)]     = t2r * w2r - t2i * w2i;
                    o[2 * (2 * s + k) + 1] = t2r * w2i + t2i * w2r;
                    o[2 * (3 * s + k)]     = t3r * w3r - t3i * w3i;
                    o[2 * (3 * s + k) + 1] = t3r * w3i + t3i * w3r;
                }
            }
            m /= 4; s *= 4;
        } else {
            const int m1 = m / 2, tstep = n / m;
            for (int q = 0; q < m1; q++) {
                const float  wr = tw[2 * (q * tstep)], wi = tw[2 * (q * tstep) + 1];
                const float *a = x + 2 * (size_t)s * q, *b = a + 2 * (size_t)s * m1;
                float       *o = y + 2 * (size_t)s * 2 * q;
                for (int k = 0; k < s; k++) {
                    const float ar = a[2 * k], ai = a[2 * k + 1], br = b[2 * k], bi = b[2 * k + 1];
                    const float tr = ar - br, ti = ai - bi;
                    o[2 * k]           = ar + br;
                    o[2 * k + 1]       = ai + bi;
                    o[2 * (s + k)]     = tr * wr - ti * wi;
                    o[2 * (s + k) + 1] = tr * wi + ti * wr;
                }
            }
            m /= 2; s *= 2;
        }
        float *t = x; x = y; y = t;
    }
    if (x != (float *)p->out) memcpy(p->out, x, sizeof(float) * 2 * (size_t)n);
}

/* ---- any size: decimation in time, out[k + m*j] built from the `radix` sub-transforms of length m ---- */
static void mixed_work(const fftwf_plan p, float *out, const float *in, int fstride, const int *factors)
{
    const int radix = factors[0], m = factors[1], n = p->n;
    if (m == 1) {
        for (int j = 0; j < radix; j++) { out[2 * j] = in[2 * (size_t)j * fstride]; out[2 * j + 1] = in[2 * (size_t)j * fstride + 1]; }
    } else {
        for (int j = 0; j < radix; j++) mixed_work(p, out + 2 * (size_t)j * m, in + 2 * (size_t)j * fstride, fstride * radix, factors + 2);
    }
    /* butterflies: for every k < m, the radix values out[k + m*j] * tw^(j*k*fstride) go through a radix-point DFT */
    float        sr[64], si[64];
    float       *tr = sr, *ti = si, *heap = NULL;
    if (radix > 64) { heap = (float *)malloc(sizeof(float) * 2 * (size_t)radix); tr = heap; ti = heap + radix; }
    const float *tw = p->tw;
    for (int k = 0; k < m; k++) {
        for (int j = 0; j < radix; j++) {
            const float  xr = out[2 * (k + (size_t)m * j)], xi = out[2 * (k + (size_t)m * j) + 1];
            const size_t t = ((size_t)j * k * fstride) % (size_t)n;
            tr[j] = xr * tw[2 * t] - xi * tw[2 * t + 1];
            ti[j] = xr * tw[2 * t + 1] + xi * tw[2 * t];
        }
        if (radix == 2) {
            out[2 * k] = tr[0] + tr[1]; out[2 * k + 1] = ti[0] + ti[1];
            out[2 * (k + m)] = tr[0] - tr[1]; out[2 * (k + m) + 1] = ti[0] - ti[1];
        } else if (radix == 4) {
            const float sg = (float)p->sign;
            const float apcr = tr[0] + tr[2], apci = ti[0] + ti[2], amcr = tr[0] - tr[2], amci = ti[0] - ti[2];
            const float bpdr = tr[1] + tr[3], bpdi = ti[1] + ti[3];
            const float jr = -sg * (ti[1] - ti[3]), ji = sg * (tr[1] - tr[3]);
            out[2 * k] = apcr + bpdr;                     out[2 * k + 1] = apci + bpdi;
            out[2 * (k + m)] = amcr + jr;                 out[2 * (k + m) + 1] = amci + ji;
            out[2 * (k + 2 * (size_t)m)] = apcr - bpdr;   out[2 * (k + 2 * (size_t)m) + 1] = apci - bpdi;
            out[2 * (k + 3 * (size_t)m)] = amcr - jr;     out[2 * (k + 3 * (size_t)m) + 1] = amci - ji;
        } else {
            const size_t rstep = (size_t)n / (size_t)radix; /* exp(sign*2*pi*i/radix) = tw[rstep] */
            for (int q = 0; q < radix; q++) {
                float  ar = 0.f, ai = 0.f;
                size_t t  = 0;
                for (int j = 0; j < radix; j++) {
                    ar += tr[j] * tw[2 * t] - ti[j] * tw[2 * t + 1];
                    ai += tr[j] * tw[2 * t + 1] + ti[j] * tw[2 * t];
                    t += (size_t)q * rstep;
                    if (t >= (size_t)n) t -= (size_t)n;
                }
                out[2 * (k + (size_t)m * q)] = ar; out[2 * (k + (size_t)m * q) + 1] = ai;
            }
        }
    }
    free(heap);
}

void fftwf_execute(const fftwf_plan p)
{
    if (p->n <= 0) return;
    if (p->n == 1) { p->out[0][0] = p->in[0][0]; p->out[0][1] = p->in[0][1]; return; }
    if (p->pow2) { stockham_pow2(p); return; }
    if ((void *)p->in == (void *)p->out) { /* in place: through the work buffer */
        memcpy(p->work, p->in, sizeof(float) * 2 * (size_t)p->n);
        mixed_work(p, (float *)p->out, p->work, 1, p->factors);
    } else {
        mixed_work(p, (float *)p->out, (const float *)p->in, 1, p->factors);
    }
}
