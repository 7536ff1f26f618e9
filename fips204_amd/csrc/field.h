// Z_q arithmetic in 32-bit integer lanes (q = 8380417) for gfx950.
//
// Replaces the scalar helpers of the reference: mont_reduce (src/helpers.rs:156-165),
// partial_reduce32 (61-67), full_reduce32 (70-76), center_mod (88-95), to_mont (131-135).
// Only results mod q are observable (SURVEY.md appendix "Montgomery bookkeeping"), so the
// device code uses the 32-bit hi/lo formulation: hi32(a*b) - hi32(lo32(a*b*qinv)*q).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mldsa {

constexpr int32_t Q = 8380417;            // lib.rs:109
constexpr uint32_t QINV = 58728449u;      // q * QINV = 1 mod 2^32 (helpers.rs:157)
constexpr int32_t R_MOD_Q = 4193792;      // 2^32 mod q
constexpr int32_t R2_MOD_Q = 2365951;     // 2^64 mod q
constexpr uint32_t R2_MOD_Q_QINV = 2145647103u;
constexpr int32_t F_MONT = 16382;         // 256^-1 * 2^32 mod q (ntt.rs:88)
constexpr uint32_t F_MONT_QINV = 16777214u;
constexpr int32_t F_MONT2 = 41978;        // 256^-1 * 2^64 mod q (inverse NTT of R^-1-scaled input)
constexpr uint32_t F_MONT2_QINV = 4286571514u;
constexpr int N = 256;

// a * b * 2^-32 mod q, result in (-q, q); |a*b| < 2^31 * q.  Both operands variable.
__device__ __forceinline__ int32_t mont_mul(int32_t a, int32_t b) {
    uint32_t lo = (uint32_t)a * (uint32_t)b;
    int32_t t = (int32_t)(lo * QINV);
    return __mulhi(a, b) - __mulhi(t, Q);
}

// Same with a constant/twiddle b whose companion bq = b * QINV mod 2^32 is precomputed
// (saves one multiply per butterfly).
__device__ __forceinline__ int32_t mont_mul_c(int32_t a, int32_t b, uint32_t bq) {
    int32_t t = (int32_t)((uint32_t)a * bq);
    return __mulhi(a, b) - __mulhi(t, Q);
}

// helpers.rs:61-67: |a| < 2^31 - 2^22  ->  (-q, q)
__device__ __forceinline__ int32_t reduce32(int32_t a) {
    int32_t x = (a + (1 << 22)) >> 23;
    return a - x * Q;
}

// (-q, q) -> [0, q)
__device__ __forceinline__ int32_t caddq(int32_t x) { return x + ((x >> 31) & Q); }

// helpers.rs:70-76
__device__ __forceinline__ int32_t freeze(int32_t a) { return caddq(reduce32(a)); }

// helpers.rs:88-95: canonical representative in (-q/2, q/2]
__device__ __forceinline__ int32_t center(int32_t a) {
    int32_t t = freeze(a);
    return t - ((((Q / 2) - t) >> 31) & Q);
}

// x * 2^32 mod q (reference to_mont, helpers.rs:131-135), result in (-q, q)
__device__ __forceinline__ int32_t to_mont(int32_t x) { return mont_mul_c(x, R2_MOD_Q, R2_MOD_Q_QINV); }

}  // namespace mldsa
