/*
 * fftw_shim.c -- TEST INFRASTRUCTURE ONLY (see fftw3.h in this directory).
 *
 * Unnormalised 1-D complex DFT behind the seven FFTW names the reference uses.
 * Power-of-two sizes: iterative radix-2 in float64.  Other sizes (only reached by the
 * reference's UL/PRACH init, which the DL oracle never calls): O(n^2) float64 DFT with a
 * twiddle table.  Output is rounded to float32, so results agree with real FFTW3f to
 * float rounding (~1e-6 relative), which is all the reference itself pins.
 */
#include "fftw3.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct oracle_fftwf_plan_s {
    int            n;
    int            sign;
    int            pow2;
    fftwf_complex *in;
    fftwf_complex *out;
    double        *tw_re; /* n entries: cos(2*pi*k/n)            */
    double        *tw_im; /* n entries: sign*sin(2*pi*k/n)       */
    double        *wr;    /* work, n entries                      */
    double        *wi;
};

void *fftwf_malloc(size_t n) { return calloc(1, n ? n : 1); }
void  fftwf_free(void *p) { free(p); }

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags)
{
    (void)flags;
    struct oracle_fftwf_plan_s *p = (struct oracle_fftwf_plan_s *)calloc(1, sizeof(*p));
    p->n    = n;
    p->sign = sign;
    p->in   = in;
    p->out  = out;
    p->pow2 = (n > 0) && ((n & (n - 1)) == 0);
    p->tw_re = (double *)malloc(sizeof(double) * (size_t)n);
    p->tw_im = (double *)malloc(sizeof(double) * (size_t)n);
    p->wr    = (double *)malloc(sizeof(double) * (size_t)n);
    p->wi    = (double *)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        double a   = 2.0 * M_PI * (double)k / (double)n;
        p->tw_re[k] = cos(a);
        p->tw_im[k] = (double)sign * sin(a);
    }
    return p;
}

void fftwf_destroy_plan(fftwf_plan p)
{
    if (!p) return;
    free(p->tw_re);
    free(p->tw_im);
    free(p->wr);
    free(p->wi);
    free(p);
}

static void dft_pow2(const fftwf_plan p)
{
    const int n = p->n;
    double   *xr = p->wr, *xi = p->wi;
    int       bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        unsigned r = 0, v = (unsigned)i;
        for (int b = 0; b < bits; b++) { r = (r << 1) | (v & 1u); v >>= 1; }
        xr[r] = (double)p->in[i][0];
        xi[r] = (double)p->in[i][1];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int j = 0; j < half; j++) {
                const double wr = p->tw_re[j * step], wi = p->tw_im[j * step];
                const int    a = base + j, b = a + half;
                const double tr = xr[b] * wr - xi[b] * wi;
                const double ti = xr[b] * wi + xi[b] * wr;
                xr[b] = xr[a] - tr; xi[b] = xi[a] - ti;
                xr[a] = xr[a] + tr; xi[a] = xi[a] + ti;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        p->out[i][0] = (float)xr[i];
        p->out[i][1] = (float)xi[i];
    }
}

static void dft_any(const fftwf_plan p)
{
    const int n = p->n;
    for (int i = 0; i < n; i++) { p->wr[i] = (double)p->in[i][0]; p->wi[i] = (double)p->in[i][1]; }
    /* in and out may alias in principle; the reference never does that, but stay safe */
    float *tmp = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    for (int k = 0; k < n; k++) {
        double sr = 0.0, si = 0.0;
        long   idx = 0;
        for (int j = 0; j < n; j++) {
            const double wr = p->tw_re[idx], wi = p->tw_im[idx];
            sr += p->wr[j] * wr - p->wi[j] * wi;
            si += p->wr[j] * wi + p->wi[j] * wr;
            idx += k;
            if (idx >= n) idx -= n;
        }
        tmp[2 * k]     = (float)sr;
        tmp[2 * k + 1] = (float)si;
    }
    memcpy(p->out, tmp, sizeof(float) * 2 * (size_t)n);
    free(tmp);
}

void fftwf_execute(const fftwf_plan p)
{
    if (p->pow2) dft_pow2(p);
    else         dft_any(p);
}
