/* k_reduce (groundgrid_amd/csrc/k2_reduce.hip) replaces the three binary32 divisions a / (c + 1) of the per-point recurrence
 * (src/GroundSegmentation.cpp:296, :302, :303; c + 1 = b, an integer in [1, 2^24]) by
 *     q = (float)((double)a * r),   r = 1.0 / (double)b,
 * falling back to the IEEE division when |q| < 2^-100.  This program checks the identity q == a / (float)b bit for bit on the
 * host (same binary64 multiply and conversions, SSE2) over: random operands of every exponent, numerators adjacent to the
 * products b * M of rounding boundaries M (the hardest cases), exact powers of two, and the extremes of the range.
 * Exit status 0 = identical everywhere it was tested.  Compile with -O1 -ffp-contract=off (no -ffast-math). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static float from_bits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static long checked = 0, fallbacks = 0, bad = 0;

static void check(float a, uint32_t b)
{
    const float bf = (float)b;
    const volatile double r = 1.0 / (double)bf;
    const volatile double p = (double)a * r;
    const float q = (float)p;
    const volatile float ref = a / bf;
    ++checked;
    if (!(fabsf(q) >= 0x1p-100f) && !isnan(q)) { /* the kernel's fallback: smallest(|q|) < 2^-100 */
        ++fallbacks;
        return;
    }
    if (isnan(q) && isnan(ref)) {
        if (bits(q) != bits(ref)) {
            if (bad++ < 10) printf("NaN bits differ a=%08x b=%u: %08x vs %08x\n", bits(a), b, bits(q), bits(ref));
        }
        return;
    }
    if (bits(q) != bits(ref)) {
        if (bad++ < 10) printf("MISMATCH a=%a (%08x) b=%u: fast %a ref %a\n", a, bits(a), b, q, ref);
    }
}

int main(void)
{
    /* 1. every divisor up to 4096 (the table) and a spread of larger ones, random numerators of every exponent */
    for (uint32_t b = 1; b <= 4096; ++b)
        for (int k = 0; k < 3000; ++k) check(from_bits((uint32_t)rng()), b);
    for (int t = 0; t < 2000000; ++t) {
        const uint32_t b = 1u + (uint32_t)(rng() % 16777216ull);
        check(from_bits((uint32_t)rng()), b);
    }
    /* 2. numerators next to b * M, M a midpoint of two adjacent floats (25-bit odd significand): the quotient is as
     *    close to a rounding boundary as it can get */
    for (int t = 0; t < 4000000; ++t) {
        const uint32_t b = (t & 1) ? 1u + (uint32_t)(rng() % 4096ull) : 1u + (uint32_t)(rng() % 16777216ull);
        const uint64_t mi = (1ull << 24) + (rng() % (1ull << 24)); /* 25-bit, made odd below */
        const int e = (int)(rng() % 180ull) - 90;
        const double m = ldexp((double)(mi | 1ull), e - 24);
        const double prod = (double)b * m; /* exact: < 2^49 significant bits */
        const float a0 = (float)prod;
        check(a0, b);
        check(nextafterf(a0, INFINITY), b);
        check(nextafterf(a0, -INFINITY), b);
        check(-a0, b);
    }
    /* 2b. the worst cases exactly: for odd b, a = b * Mi - eps with eps = +-1, +-3, +-5 and a representable (the low s
     *     bits of b * Mi - eps vanish): a / b = Mi - eps / b lies a relative ~2^-25 / b from the boundary Mi */
    {
        long worst = 0;
        for (uint64_t t = 0; t < 3000000ull; ++t) {
            const uint64_t b = (t < 2048 ? 2 * t + 1 : (rng() % 16777216ull)) | 1ull;
            uint64_t inv = b; /* b^-1 mod 2^64 (Newton) */
            for (int k = 0; k < 6; ++k) inv *= 2 - b * inv;
            int lb = 0;
            while ((b >> lb) > 1) ++lb; /* floor(log2 b) */
            for (int s = lb + 1; s <= lb + 2; ++s) {
                const uint64_t mod = 1ull << s;
                for (int eps = -5; eps <= 5; eps += 2) {
                    const uint64_t m0 = ((uint64_t)((int64_t)eps) * inv) & (mod - 1);
                    /* odd Mi = m0 + j 2^s in [2^24, 2^25) */
                    for (int tries = 0; tries < 2; ++tries) {
                        uint64_t mi;
                        if (mod >= (1ull << 25)) {
                            mi = m0;
                        } else {
                            const uint64_t span = (1ull << 24) / mod;
                            mi = m0 + ((1ull << 24) / mod + rng() % (span ? span : 1)) * mod;
                        }
                        if (mi < (1ull << 24) || mi >= (1ull << 25)) continue;
                        const int64_t num = (int64_t)(b * mi) - eps;
                        if (num <= 0 || (num & (int64_t)(mod - 1))) continue;
                        const uint64_t A = (uint64_t)num >> s;
                        if (A >= (1ull << 24)) continue;
                        const float a = ldexpf((float)A, s - 60 + (int)(rng() % 100ull));
                        check(a, (uint32_t)b);
                        check(-a, (uint32_t)b);
                        ++worst;
                    }
                }
            }
        }
        printf("worst-case numerators generated: %ld\n", worst);
    }
    /* 3. specials and the edges of the range */
    {
        const float sp[] = {0.0f, -0.0f, INFINITY, -INFINITY, NAN, -NAN, 3.4028234663852886e38f, -3.4028234663852886e38f, 0x1p-100f,
                            0x1.fffffep-101f, 0x1p-99f, 0x1p-126f, 0x1p-149f, 0x1.8p-149f, 1.0f, -1.0f, 0x1.fffffep127f, 0x1p-76f};
        for (unsigned k = 0; k < sizeof(sp) / sizeof(sp[0]); ++k)
            for (uint32_t b = 1; b <= 70000; ++b) check(sp[k], b);
        for (uint32_t b = 16777000; b <= 16777216; ++b)
            for (unsigned k = 0; k < sizeof(sp) / sizeof(sp[0]); ++k) check(sp[k], b);
        /* signalling / payload NaNs */
        check(from_bits(0x7fa12345u), 7);
        check(from_bits(0xffc00001u), 12345);
    }
    printf("checked %ld quotients, %ld below the 2^-100 threshold (IEEE division in the kernel), %ld mismatches\n", checked, fallbacks, bad);
    return bad ? 1 : 0;
}
