// ExpandA (24-bit form) by the wave-cooperative sponge: the body shared by k_expand_a_coop (kernels_sample.hip) and the single-launch
// verify kernel of small calls (kernels_small.hip).
#pragma once
#include "keccak_coop2.h"
#include "sampler_dev.h"

namespace mldsa {

constexpr int EA_COOP_BLK_DWORDS = 44;  // 168 bytes + the read-ahead of the last candidate

// ONE polynomial per wave on the interleaved cooperative sponge (keccak_coop2.h): polynomial g of the call.  After every permutation the lanes
// holding the 21 rate words put the 168-byte block into the wave's LDS row `blk`; lanes 0-55 test its 56 candidates
// (coeff_from_three_bytes, conversion.rs:40-61), rank the accepted ones with a ballot and store each as its three bytes at 3 i of the
// polynomial's 768-byte row -- the same bytes k_expand_a<.., true> writes.  Wave-uniform control flow.
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
