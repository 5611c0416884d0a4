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
} // namespace synth
