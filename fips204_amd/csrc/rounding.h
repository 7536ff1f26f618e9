// Rounding helpers of src/high_low.rs for the device, shared by the codec / sign / arithmetic
// kernels: decompose (66-96), use_hint (155-192), and the w1Encode packing (encodings.rs:338-360)
// for a polynomial held in the NTT kernels' strided register layout.
#pragma once
#include "field.h"

namespace mldsa {

// decompose (high_low.rs:66-96) for canonical r in [0, q); G2HI = (gamma2 == (q-1)/32)
template <bool G2HI>
__device__ __forceinline__ void decompose(int32_t rp, int32_t& r1, int32_t& r0) {
    constexpr int32_t GAMMA2 = G2HI ? (Q - 1) / 32 : (Q - 1) / 88;
    int32_t x = (rp + 127) >> 7;
    if constexpr (!G2HI) {
        x = (x * 11275 + (1 << 23)) >> 24;
        x ^= ((43 - x) >> 31) & x;
    } else {
        x = (x * 1025 + (1 << 21)) >> 22;
        x &= 15;
    }
    int32_t y = rp - x * 2 * GAMMA2;
    y -= (((Q - 1) / 2 - y) >> 31) & Q;
    r1 = x;
    r0 = y;
}

// use_hint (high_low.rs:155-192), r canonical; h = 0 gives high_bits (104-111)
template <bool G2HI>
__device__ __forceinline__ int32_t use_hint(int32_t h, int32_t r) {
    int32_t r1, r0;
    decompose<G2HI>(r, r1, r0);
    if (h == 0) return r1;
    if constexpr (!G2HI) {
        if (r0 > 0) return r1 == 43 ? 0 : r1 + 1;
        return r1 == 0 ? 43 : r1 - 1;
    } else {
        return r0 > 0 ? (r1 + 1) & 15 : (r1 - 1) & 15;
    }
}

// simple_bit_pack of w1 (4 bits per coefficient for gamma2 = (q-1)/32, 6 bits otherwise) when lane
// holds coefficient 64 k + lane in v: coefficient pairs (4-bit) / quads (6-bit) sit in adjacent
// lanes, so the bytes are assembled with DPP quad permutes.  `dst` = first byte of the polynomial.
template <bool G2HI>
__device__ __forceinline__ void pack_w1_strided(uint32_t v, int k, uint8_t* dst, int lane) {
    if constexpr (G2HI) {
        const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // lane ^ 1
        if (!(lane & 1)) dst[32 * k + (lane >> 1)] = (uint8_t)(v | (nb << 4));
    } else {
        const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // lane ^ 1
        const uint32_t pair = (lane & 1) ? 0u : (v | (n1 << 6));                                         // 12 bits in even lanes
        const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pair, 0x4E, 0xF, 0xF, false);  // lane ^ 2
        if (!(lane & 3)) {
            const uint32_t q24 = pair | (n2 << 12);
            uint8_t* d = dst + 48 * k + 3 * (lane >> 2);
            d[0] = (uint8_t)q24; d[1] = (uint8_t)(q24 >> 8); d[2] = (uint8_t)(q24 >> 16);
        }
    }
}

}  // namespace mldsa
