/*
 * Minimal single-precision FFTW3 API stand-in -- TEST INFRASTRUCTURE ONLY.
 *
 * FFTW3 is an un-vendored system dependency of the reference (link `fftw3f`,
 * liblte/CMakeLists.txt) and is not installed in this image.  The reference's
 * liblte_phy.cc only needs seven names from it (liblte_phy.h:93, liblte_phy.cc:2309-2330):
 * fftwf_complex, fftwf_plan, fftwf_malloc, fftwf_free, fftwf_plan_dft_1d,
 * fftwf_execute, fftwf_destroy_plan.  This header + fftw_shim.c provide them as a
 * mathematically defined unnormalised DFT (exp(sign*2*pi*i*jk/n)), evaluated in
 * float64 and rounded to float32 on output.
 *
 * Nothing in the product path includes this file; it exists so the reference's own
 * sources can be compiled in place into oracle/_ref/ (see Makefile).
 */
#ifndef ORACLE_FFTW3_SHIM_H
#define ORACLE_FFTW3_SHIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float fftwf_complex[2];

struct oracle_fftwf_plan_s;
typedef struct oracle_fftwf_plan_s *fftwf_plan;

#define FFTW_FORWARD  (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE  (0U)
#define FFTW_ESTIMATE (1U << 6)

void      *fftwf_malloc(size_t n);
void       fftwf_free(void *p);
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
void       fftwf_execute(const fftwf_plan p);
void       fftwf_destroy_plan(fftwf_plan p);

#ifdef __cplusplus
}
#endif

#endif
