// Host-side input synthesis helpers (see synth.cc).
#pragma once
#include <cstdint>
#include <vector>

namespace synth {
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
