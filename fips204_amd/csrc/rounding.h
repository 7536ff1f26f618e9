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

// simple_bit_pack of w1 (encodings.rs:338-360: 4 bits per coefficient for gamma2 = (q-1)/32, 6 bits
// otherwise) for a polynomial in the strided register layout: v[k] = field of coefficient 64 k + lane.
// `dst` = first byte of the polynomial's 32 * bits bytes (4-byte aligned for the 4-bit form).
template <bool G2HI>
__device__ __forceinline__ void pack_w1_strided(const uint32_t v[4], uint8_t* dst, int lane) {
    if constexpr (G2HI) {
        // 8 adjacent lanes make one dword (8 nibbles): OR-butterfly over lane ^ 1, lane ^ 2 and the mirror
        // of the 8-lane half row; afterwards every lane of the group holds dword 8 k + (lane >> 3) of
        // register k, and lanes 8 m + k (k < 4) store the polynomial's 128 bytes with ONE dword store
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = v[k] << (4 * (lane & 7));
            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
            x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);  // row_half_mirror
            d[k] = x;
        }
        const int k = lane & 3;
        const uint32_t mine = k == 0 ? d[0] : k == 1 ? d[1] : k == 2 ? d[2] : d[3];
        if (!(lane & 4)) store_row(reinterpret_cast<uint32_t*>(dst) + 8 * k + (lane >> 3), mine);
    } else {
        // coefficient quads sit in adjacent lanes: 4 x 6 bits = 3 bytes, assembled with DPP quad permutes
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[k], 0xB1, 0xF, 0xF, false);  // lane ^ 1
            const uint32_t pair = (lane & 1) ? 0u : (v[k] | (n1 << 6));                                       // 12 bits in even lanes
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pair, 0x4E, 0xF, 0xF, false);   // lane ^ 2
            if (!(lane & 3)) {
                const uint32_t q24 = pair | (n2 << 12);
                uint8_t* d = dst + 48 * k + 3 * (lane >> 2);
                d[0] = (uint8_t)q24; d[1] = (uint8_t)(q24 >> 8); d[2] = (uint8_t)(q24 >> 16);
            }
        }
    }
}

// BitPack / SimpleBitPack of four CONSECUTIVE coefficients per lane (encodings.rs, conversion.rs:143-196): f[i] = the `bits`-wide field
// of coefficient 4 lane + i; `dst` = first byte of the polynomial's 32 * bits bytes.  bits in {3, 4, 10, 13}.
__device__ __forceinline__ void store_fields(uint8_t* dst, const uint32_t f[4], int bits, int lane) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint64_t)f[i] << (i * bits);
    const int nbytes = bits / 2;  // 4 * bits / 8; bits in {3, 4, 10, 13} -> handle odd sizes below
    if (bits == 3) {  // 12 bits per lane: pair lanes -> 3 bytes per 2 lanes
        const uint32_t other = __shfl_xor((uint32_t)v, 1);
        if (!(lane & 1)) {
            const uint32_t both = (uint32_t)v | (other << 12);
            uint8_t* d = dst + (lane >> 1) * 3;
            d[0] = (uint8_t)both; d[1] = (uint8_t)(both >> 8); d[2] = (uint8_t)(both >> 16);
        }
    } else if (bits == 13) {  // 52 bits per lane: pair lanes -> 13 bytes per 2 lanes
        const uint64_t other = __shfl_xor((unsigned long long)v, 1);
        if (!(lane & 1)) {
            uint8_t* d = dst + (lane >> 1) * 13;
            const uint64_t lo = v | (other << 52);
            const uint64_t hi = other >> 12;
            // 8 + 4 + 1 bytes at whatever alignment the key row has (byte-granular stores of gfx950) instead of 13 byte stores
            *reinterpret_cast<u64_any*>(d) = lo;
            *reinterpret_cast<u32_any*>(d + 8) = (uint32_t)hi;
            d[12] = (uint8_t)(hi >> 32);
        }
    } else if (bits == 10) {  // 5 bytes per lane
        uint8_t* d = dst + lane * 5;
        *reinterpret_cast<u32_any*>(d) = (uint32_t)v;
        d[4] = (uint8_t)(v >> 32);
    } else {              // bits == 4: 2 bytes per lane
        *reinterpret_cast<u16_any*>(dst + lane * nbytes) = (uint16_t)v;
    }
}

}  // namespace mldsa
