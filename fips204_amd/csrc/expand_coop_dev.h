// ExpandA (24-bit form) by the wave-cooperative sponge, one polynomial per half-wave: the body shared by k_expand_a_coop
// (kernels_sample.hip) and the single-launch verify kernel of small calls (kernels_small.hip).
#pragma once
#include "keccak_coop2.h"
#include "sampler_dev.h"

namespace mldsa {

constexpr int EA_COOP_BLK_DWORDS = 44;  // 168 bytes + the read-ahead of the last candidate

// Polynomials g0 (lower half-wave) and g0 + 1 (upper) of the call's n_streams = n_ops * K * L: after every permutation the 21 lanes
// holding the rate words put the 168-byte block into the half's LDS row `blk`; the half's lanes then test its 56 candidates in two
// passes of 28 (coeff_from_three_bytes, conversion.rs:40-61), rank the accepted ones with a ballot and store each as its three bytes
// at 3 i of the polynomial's 768-byte row -- the same bytes k_expand_a<.., true> writes.  Wave-uniform control flow.
template <int K, int L>
__device__ __forceinline__ void expand_a_coop_pair(const uint8_t* __restrict__ rho, size_t rho_stride, const uint32_t* __restrict__ key_idx,
                                                   int32_t* __restrict__ a_hat, size_t g0, size_t n_streams, uint32_t n_keys, uint32_t* blk, int lane,
                                                   const CoopLane& c) {
    constexpr int ROW = PACKED_POLY_DWORDS * 4;
    const int half = lane >> 5, i = lane & 31;
    const size_t g = g0 + half;
    const bool valid = g < n_streams;
    const size_t gc = valid ? g : g0;
    const size_t op = gc / (K * L);
    const int rs = (int)(gc % (K * L)), r = rs / L, sidx = rs % L;
    size_t key = key_idx ? key_idx[op] : op;
    if (n_keys && key >= n_keys) key = 0;  // (refused beside this kernel: see k_expand_a)
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < 4) {
        const uint8_t* src = rho + key * rho_stride + 8 * c.word;
        lo = load_le32(src);
        hi = load_le32(src + 4);
    }
    if (c.active && c.word == 4) lo = (uint32_t)sidx | ((uint32_t)r << 8) | (0x1Fu << 16);
    if (c.active && c.word == SHAKE128_RATE / 8 - 1) hi = 0x80000000u;
    uint8_t* row = reinterpret_cast<uint8_t*>(a_hat) + g * (size_t)ROW;
    int count = valid ? 0 : N;  // coefficients stored so far (the same in every lane of the half)
    while (__any(count < N)) {
        keccak_f1600_coop(lo, hi, c);
        if (c.active && c.word < SHAKE128_RATE / 8) { blk[2 * c.word] = lo; blk[2 * c.word + 1] = hi; }
        wave_lds_sync();
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int cand = 28 * pass + i;
            bool acc = false;
            uint32_t z = 0;
            if (i < 28) {
                const int bo = 3 * cand;  // byte offset of the candidate: two aligned dwords hold it
                const uint64_t two = ((uint64_t)blk[(bo >> 2) + 1] << 32) | blk[bo >> 2];
                z = (uint32_t)(two >> (8 * (bo & 3))) & 0x7FFFFFu;
                acc = z < (uint32_t)Q;
            }
            const unsigned long long all = __ballot(acc);
            const uint32_t mine = half ? (uint32_t)(all >> 32) : (uint32_t)all;
            const int idx = count + __popc(mine & ((1u << i) - 1u));
            if (acc && idx < N) {
                uint8_t* dst = row + 3 * idx;
                dst[0] = (uint8_t)z;
                dst[1] = (uint8_t)(z >> 8);
                dst[2] = (uint8_t)(z >> 16);
            }
            count += __popc(mine);
        }
        wave_lds_sync();
    }
}

// The same for ONE polynomial per wave on the interleaved sponge (keccak_coop2.h): polynomial g of the call; the 56 candidates of a
// block are tested by lanes 0-55 in one pass.  Same bytes out.
template <int K, int L>
__device__ __forceinline__ void expand_a_coop2_poly(const uint8_t* __restrict__ rho, size_t rho_stride, const uint32_t* __restrict__ key_idx,
                                                    int32_t* __restrict__ a_hat, size_t g, uint32_t n_keys, uint32_t* blk, int lane, const Coop2Lane& c) {
    constexpr int ROW = PACKED_POLY_DWORDS * 4;
    const size_t op = g / (K * L);
    const int rs = (int)(g % (K * L)), r = rs / L, sidx = rs % L;
    size_t key = key_idx ? key_idx[op] : op;
    if (n_keys && key >= n_keys) key = 0;  // (refused beside this kernel: see k_expand_a)
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < 4) {
        const uint8_t* src = rho + key * rho_stride + 8 * c.word;
        lo = load_le32(src);
        hi = load_le32(src + 4);
    }
    if (c.active && c.word == 4) lo = (uint32_t)sidx | ((uint32_t)r << 8) | (0x1Fu << 16);
    if (c.active && c.word == SHAKE128_RATE / 8 - 1) hi = 0x80000000u;
    uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
    uint8_t* row = reinterpret_cast<uint8_t*>(a_hat) + g * (size_t)ROW;
    int count = 0;  // coefficients stored so far (wave-uniform)
    while (count < N) {
        keccak_f1600_coop2(v, c);
        coop2_to_lohi(v, lane, lo, hi);
        if (c.active && c.word < SHAKE128_RATE / 8) blk[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
        wave_lds_sync();
        bool acc = false;
        uint32_t z = 0;
        if (lane < 56) {
            const int bo = 3 * lane;  // byte offset of the candidate: two aligned dwords hold it
            const uint64_t two = ((uint64_t)blk[(bo >> 2) + 1] << 32) | blk[bo >> 2];
            z = (uint32_t)(two >> (8 * (bo & 3))) & 0x7FFFFFu;
            acc = z < (uint32_t)Q;
        }
        const unsigned long long all = __ballot(acc);
        const int idx = count + __popcll(all & ((1ull << lane) - 1ull));
        if (acc && idx < N) {
            uint8_t* dst = row + 3 * idx;
            dst[0] = (uint8_t)z;
            dst[1] = (uint8_t)(z >> 8);
            dst[2] = (uint8_t)(z >> 16);
        }
        count += __popcll(all);
        wave_lds_sync();
    }
}

}  // namespace mldsa
