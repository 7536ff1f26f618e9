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
// core: polynomial (r, sidx) of ExpandA(rho) into the 768-byte row `row`; `rho`: 32 bytes (any address space)
__device__ __forceinline__ void expand_a_coop2_core(const uint8_t* rho, int r, int sidx, uint8_t* __restrict__ row, uint32_t* blk, int lane, const Coop2Lane& c) {
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < 4) {
        lo = load_le32(rho + 8 * c.word);
        hi = load_le32(rho + 8 * c.word + 4);
    }
    if (c.active && c.word == 4) lo = (uint32_t)sidx | ((uint32_t)r << 8) | (0x1Fu << 16);
    if (c.active && c.word == SHAKE128_RATE / 8 - 1) hi = 0x80000000u;
    uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
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

template <int K, int L>
__device__ __forceinline__ void expand_a_coop2_poly(const uint8_t* __restrict__ rho, size_t rho_stride, const uint32_t* __restrict__ key_idx,
                                                    int32_t* __restrict__ a_hat, size_t g, uint32_t n_keys, uint32_t* blk, int lane, const Coop2Lane& c) {
    constexpr int ROW = PACKED_POLY_DWORDS * 4;
    const size_t op = g / (K * L);
    const int rs = (int)(g % (K * L));
    size_t key = key_idx ? key_idx[op] : op;
    if (n_keys && key >= n_keys) key = 0;  // (refused beside this kernel: see k_expand_a)
    expand_a_coop2_core(rho + key * rho_stride, rs / L, rs % L, reinterpret_cast<uint8_t*>(a_hat) + g * (size_t)ROW, blk, lane, c);
}

// coeff_from_half_byte (conversion.rs:80-111): the candidate's value and whether it is accepted
template <int ETA>
__device__ __forceinline__ bool half_byte(uint32_t b, int32_t& out) {
    if constexpr (ETA == 2) {
        out = 2 - (int32_t)(b - ((b * 13108u) >> 16) * 5u);  // b mod 5 for b < 16 (conversion.rs:91-93)
        return b < 15;
    } else {
        out = 4 - (int32_t)b;
        return b < 9;
    }
}

constexpr int ES_COOP_BLK_DWORDS = 36;  // 136 bytes + pad

// ExpandS (hashing.rs:225-268) for ONE polynomial by a wave: stream r of SHAKE256(rho' || r || 0), one BYTE per coefficient into `row`
// (256 bytes, coefficient order: what k_expand_s<ETA, true> writes).  The 136-byte block goes through the wave's LDS row `blk`; its 272
// half-byte candidates (low nibble first, hashing.rs:177-180) are tested in five passes of 64, ranked with a ballot and stored at their
// coefficient index.  `rho_prime`: the op's 64 bytes.
template <int ETA>
__device__ __forceinline__ void expand_s_coop2_poly(const uint8_t* __restrict__ rho_prime, uint32_t r, uint8_t* __restrict__ row, uint32_t* blk, int lane,
                                                    const Coop2Lane& c) {
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < 8) {
        lo = load_le32(rho_prime + 8 * c.word);
        hi = load_le32(rho_prime + 8 * c.word + 4);
    }
    if (c.active && c.word == 8) lo = r | (0x1Fu << 16);  // hashing.rs:260/266: rho' || r || 0  (then pad)
    if (c.active && c.word == SHAKE256_RATE / 8 - 1) hi = 0x80000000u;
    uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
    int count = 0;
    while (count < N) {  // wave-uniform
        keccak_f1600_coop2(v, c);
        coop2_to_lohi(v, lane, lo, hi);
        if (c.active && c.word < SHAKE256_RATE / 8) blk[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
        wave_lds_sync();
#pragma unroll
        for (int pass = 0; pass < 5; pass++) {
            const int cand = 64 * pass + lane;  // half-byte index: byte cand >> 1, low nibble first
            bool acc = false;
            int32_t val = 0;
            if (cand < 2 * SHAKE256_RATE) {
                const uint32_t b = (blk[cand >> 3] >> (4 * (cand & 7))) & 15u;
                acc = half_byte<ETA>(b, val);
            }
            const unsigned long long all = __ballot(acc);
            const int idx = count + __popcll(all & ((1ull << lane) - 1ull));
            if (acc && idx < N) row[idx] = (uint8_t)val;
            count += __popcll(all);
        }
        wave_lds_sync();
    }
}

// RAW ExpandMask (hashing.rs:281-313) for ONE polynomial by a wave: stream of SHAKE256(rho'' || kappa_r), the squeezed bytes themselves
// (32 c per polynomial = BitPack(y)) into `row` -- what k_expand_mask<GB, true> writes.  `rho_pp`: the op's 64 bytes; kr = kappa + r
// (u16 arithmetic, hashing.rs:293).
template <int GB>
__device__ __forceinline__ void expand_mask_coop2_poly(const uint8_t* __restrict__ rho_pp, uint32_t kr, uint8_t* __restrict__ row, int lane, const Coop2Lane& c) {
    constexpr int ROW_BYTES = 32 * (GB + 1);
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < 8) {
        lo = load_le32(rho_pp + 8 * c.word);
        hi = load_le32(rho_pp + 8 * c.word + 4);
    }
    if (c.active && c.word == 8) lo = (kr & 0xFFFFu) | (0x1Fu << 16);
    if (c.active && c.word == SHAKE256_RATE / 8 - 1) hi = 0x80000000u;
    uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        keccak_f1600_coop2(v, c);
        coop2_to_lohi(v, lane, lo, hi);
        const int off = blk * SHAKE256_RATE + 8 * c.word;
        if (c.active && c.word < SHAKE256_RATE / 8 && off < ROW_BYTES) *reinterpret_cast<uint32_t*>(row + off + 4 * (lane >> 5)) = lane < 32 ? lo : hi;  // (rows are 8-byte aligned: 576, 640)
    }
}

}  // namespace mldsa
