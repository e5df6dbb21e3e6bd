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

// Tempering of a run of state words (the output function of mt19937_next), written so that the compiler vectorises it.
inline void mt_temper(const uint32_t* __restrict__ mt, uint32_t* __restrict__ out, int from) {
    for (int k = from; k < kN; ++k) {
        uint32_t y = mt[k];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[k] = y;
    }
}

}  // namespace

// Round 4: 0.35 -> ~0.2 ms for the two planes of a 320x256 frame (the host's largest item per frame in the stream modes, where the mixed
// leg is host-bound).  Same draws, same state afterwards; what changed is how they are produced:
//   * a whole block of 624 state words is refilled and tempered at once (two vectorisable loops) instead of a branch + tempering per draw;
//   * the rejection loop of random_interval carries no unpredictable branch: a rejected draw performs the swap (i, i) -- nothing -- and
//     does not advance i (acceptance probability is between 1/2 and 1 and changes with i: a conditional branch here mispredicts on a
//     third of the elements);
//   * the mask is constant between powers of two: outer loop over the bit length of i.
extern "C" int vidc_host_mt19937_permutation_prefix(uint32_t* key624, int32_t* pos, long long n, int k, int32_t* idx_out, int32_t* scratch) {
    VIDC_REQUIRE(key624 && pos && (k == 0 || idx_out) && (n == 0 || scratch), VIDC_ERR_NULL, "vidc_host_mt19937_permutation_prefix: null pointer");
    VIDC_REQUIRE(n >= 0 && n < (1ll << 31) && k >= 0 && k <= n && *pos >= 0 && *pos <= kN, VIDC_ERR_SHAPE,
                 "vidc_host_mt19937_permutation_prefix: bad arguments (0 <= k <= n < 2^31, 0 <= pos <= 624)");
    int p = *pos;
    uint32_t tw[kN];
    if (p < kN) mt_temper(key624, tw, p);
    for (long long i = 0; i < n; ++i) scratch[i] = (int32_t)i;
    long long i = n - 1;
    while (i >= 1) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const long long lo = ((long long)mask + 1) >> 1;      // the mask holds for i in [lo, mask]; lo >= 1
        while (i >= lo) {
            if (p == kN) { mt_refill(key624); mt_temper(key624, tw, 0); p = 0; }
            const int avail = kN - p;
            int c = 0;
            for (; c < avail && i >= lo; ++c) {
                const uint32_t r = tw[p + c] & mask;
                const bool ok = r <= (uint32_t)i;
                const long long j = ok ? (long long)r : i;
                const int32_t t = scratch[i];
                scratch[i] = scratch[j];
                scratch[j] = t;
                i -= ok ? 1 : 0;
            }
            p += c;
        }
    }
    for (int q = 0; q < k; ++q) idx_out[q] = scratch[q];
    *pos = p;
    return VIDC_OK;
}
