// pin_inputs.h -- the seeded input generator of the pinning probe (dependency-free; mirrored in compare.py, and the
// mirror is checked in this repo's CPU tests through inputs_dump.cpp).  32-bit LCG; every derived value is exactly
// representable, so C++ and Python produce the same bits.
#pragma once
#include <cmath>
#include <cstdint>

struct Lcg {
    uint32_t s;
    explicit Lcg(uint32_t seed) : s(seed) {}
    uint32_t next() { return s = s * 1664525u + 1013904223u; }
    // signed 24-bit integer times 2^(e - 23), e in [-12, 12]: cancellation-heavy sums, exact to construct
    float wide_float()
    {
        const int32_t m = (int32_t)(next() >> 8) - (1 << 23);
        const int e = (int)(next() >> 27) - 12; // [-12, 19]
        return std::ldexp((float)m, (e > 12 ? 12 : e) - 23);
    }
    double unit() { return (double)(next() >> 8) / 16777216.0; } // [0, 1), 24 bits
};
