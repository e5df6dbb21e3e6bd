// Host-side replay of numpy's legacy random stream (no device code).  The reference draws its RANSAC hypotheses with
// np.random.permutation(n)[0:300] per plane (main.py:43, :78): to reproduce its draws the generator has to be advanced exactly as numpy
// advances it -- n - 1 bounded draws of a Fisher-Yates shuffle over ALL n pixels of the plane -- although only 300 entries are used.  That
// shuffle is the largest item of host time per frame (two planes of ~38 000 pixels: ~0.5 ms in numpy, whose generic shuffle swaps through
// memcpy); this is the same algorithm as a tight loop on the MT19937 state itself:
//   numpy/random/mtrand.pyx  RandomState.permutation(int) -> arange(n), _shuffle_raw: for i = n-1 .. 1: j = random_interval(i); swap(i, j)
//   numpy/random/src/distributions/distributions.c  random_interval(): mask = next power of two - 1, 32-bit draws, rejection
//   numpy/random/src/mt19937/mt19937.c  mt19937_gen / mt19937_next (the standard generator and tempering)
// tests/test_abi.py compares it with numpy draw by draw (state included) for sizes on both sides of the 624-word refill.
#include "common.h"
#include <cstdint>

namespace {

constexpr int kN = 624, kM = 397;

inline void mt_refill(uint32_t* mt) {
    int kk;
    uint32_t y;
    for (kk = 0; kk < kN - kM; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + kM] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
    }
    for (; kk < kN - 1; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (kM - kN)] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
    }
    y = (mt[kN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[kN - 1] = mt[kM - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
}

inline uint32_t mt_next(uint32_t* mt, int& pos) {
    if (pos == kN) { mt_refill(mt); pos = 0; }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

}  // namespace

extern "C" int vidc_host_mt19937_permutation_prefix(uint32_t* key624, int32_t* pos, long long n, int k, int32_t* idx_out, int32_t* scratch) {
    VIDC_REQUIRE(key624 && pos && (k == 0 || idx_out) && (n == 0 || scratch), VIDC_ERR_NULL, "vidc_host_mt19937_permutation_prefix: null pointer");
    VIDC_REQUIRE(n >= 0 && n < (1ll << 31) && k >= 0 && k <= n && *pos >= 0 && *pos <= kN, VIDC_ERR_SHAPE,
                 "vidc_host_mt19937_permutation_prefix: bad arguments (0 <= k <= n < 2^31, 0 <= pos <= 624)");
    int p = *pos;
    for (long long i = 0; i < n; ++i) scratch[i] = (int32_t)i;
    for (long long i = n - 1; i >= 1; --i) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t j;
        while ((j = (mt_next(key624, p) & mask)) > (uint32_t)i) {}
        const int32_t t = scratch[i];
        scratch[i] = scratch[j];
        scratch[j] = t;
    }
    for (int i = 0; i < k; ++i) idx_out[i] = scratch[i];
    *pos = p;
    return VIDC_OK;
}
