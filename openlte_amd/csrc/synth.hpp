// Host-side input synthesis helpers (see synth.cc).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mi_lte.h"

namespace synth {
// what the generators divide by and index with: an LTE transform length, a grid that fits inside it (and the 112-entry PRB lists), and an
// allocation whose PRBs lie on the grid
inline bool valid_grid(uint32_t fft_size, uint32_t N_rb)
{
    return (fft_size == 128 || fft_size == 256 || fft_size == 512 || fft_size == 1024 || fft_size == 2048) && N_rb >= 6 && N_rb <= 110 && 12 * N_rb < fft_size;
}
inline bool valid_alloc(const mi_lte_pdsch_alloc &a, uint32_t N_rb)
{
    if (a.N_prb == 0 || a.N_prb > N_rb || a.mod_type > 3 || a.tbs == 0 || a.tbs + 24 > 6144 || a.rv_idx > 3) return false;
    for (uint32_t s = 0; s < 2; s++)
        for (uint32_t i = 0; i < a.N_prb; i++)
            if (a.prb[s][i] >= N_rb) return false;
    return true;
}
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next();
    double   uniform();
    double   normal();
};
void crc24a(const uint8_t *bits, uint32_t n, uint8_t p[24]);
bool qpp_params(uint32_t K, uint32_t *f1, uint32_t *f2);
void qpp_map(uint32_t K, bool ref_wrap, std::vector<uint16_t> &pi);
void turbo_encode(const uint8_t *c, uint32_t K, bool ref_wrap, uint8_t *d_planar);
void gold(uint32_t c_init, uint32_t len, uint8_t *c);
void rate_match(const uint8_t *d, uint32_t D, uint32_t N_cb_limit, uint32_t rv, uint32_t E, uint8_t *e);
void modulate(const uint8_t *b, uint32_t n_sym, uint32_t mod, float *re, float *im);
void idft(std::vector<double> &xr, std::vector<double> &xi);
} // namespace synth
