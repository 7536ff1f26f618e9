// Device cores shared by the stand-alone kernels (k_shake256_2, k_sample_in_ball) and the signer's fused challenge
// kernel (k_sign_challenge): the fixed-shape two-part SHAKE256 absorb of one op per lane with cooperative loads, and the
// Fisher-Yates body of SampleInBall on lane-private LDS rows.
#pragma once
#include "keccak_coop2.h"
#include "sampler_dev.h"

namespace mldsa {

constexpr int H_STRIDE = 35;        // dwords per tile row of the hash input (136 bytes + pad, odd stride)
constexpr int SIB_C_STRIDE = 65;    // dwords: 256 int8 + pad (odd: conflict-free)
constexpr int SIB_BLK_STRIDE = 35;  // dwords: 136 bytes + pad (= H_STRIDE: the hash tile can serve as the block rows)

// SHAKE256 over A (la bytes) | B (lb bytes) | up to 4 literal tail bytes for the wave's 64 ops, one op per lane; on return
// `st` has been permuted after the last (padded) block.  row_a / row_b: the wave's 64 source pointers per part in LDS
// (written by the caller, visible after a wave_lds_sync), tile: 64 rows of H_STRIDE dwords of the wave.
// For every rate block the wave reads the 64 ops' 136-byte pieces with consecutive lanes on consecutive dwords (coalesced)
// into the tile, and each lane then absorbs its own row.  Passes 0..31: lanes 0-31 fetch dwords 0..31 of row 2 i, lanes 32-63
// those of row 2 i + 1 (coalesced 128-byte runs); passes 32, 33: lane r fetches dword 32 / 33 of row r.  Branch-free: a
// dword past the end of the data is fetched from the row start and zeroed.  All 34 loads of a block are issued back to
// back, and those of block b + 1 before the permutation of block b, so their latency hides under it (the hash chain of
// one op is serial and the kernels run a single wave per SIMD).  la and lb are multiples of 4; ALIGNED = all pointers /
// strides are multiples of 4 (dword loads).
template <bool ALIGNED>
__device__ __forceinline__ void shake256_2_absorb(KeccakState& st, uint32_t* tile, const unsigned long long* row_a,
                                                  const unsigned long long* row_b, int la, int lb, uint32_t tail, int tail_len,
                                                  int lane) {
    const int data = la + lb;           // bytes that come from memory
    const int total = data + tail_len;  // message length
    const int n_blocks = total / SHAKE256_RATE + 1;
    keccak_zero(st);
    uint32_t pre[34];
    auto issue = [&](int blk) {
        const int base = blk * SHAKE256_RATE;
#pragma unroll
        for (int i = 0; i < 34; i++) {
            const int row = i < 32 ? 2 * i + (lane >> 5) : lane;
            const int off = base + 4 * (i < 32 ? (lane & 31) : i);
            const bool have = off + 4 <= data, in_a = off < la;
            const unsigned long long pp = in_a ? row_a[row] : row_b[row];
            const uint8_t* src = reinterpret_cast<const uint8_t*>(pp) + (have ? (in_a ? off : off - la) : 0);
            const uint32_t v = ALIGNED ? *reinterpret_cast<const uint32_t*>(src) : load_le32(src);
            pre[i] = have ? v : 0u;
        }
    };
    issue(0);
    for (int blk = 0; blk < n_blocks; blk++) {
        const int base = blk * SHAKE256_RATE;
#pragma unroll
        for (int i = 0; i < 34; i++) {
            const int row = i < 32 ? 2 * i + (lane >> 5) : lane;
            const int wd = i < 32 ? (lane & 31) : i;
            tile[row * H_STRIDE + wd] = pre[i];
        }
        wave_lds_sync();
        static_for<0, 17>([&](auto wc) {
            constexpr int W = decltype(wc)::value;
            uint32_t lo = tile[lane * H_STRIDE + 2 * W], hi = tile[lane * H_STRIDE + 2 * W + 1];
            const int off = base + 8 * W;
            if (off + 8 > data && off <= total) {  // word holds tail bytes and / or the 0x1F pad (lane-uniform)
                for (int i = 0; i < 8; i++) {
                    const int pos = off + i;
                    uint32_t v = 0;
                    if (pos >= data && pos < total) v = (tail >> (8 * (pos - data))) & 0xFF;
                    else if (pos == total) v = 0x1F;
                    if (i < 4) lo |= v << (8 * i); else hi |= v << (8 * (i - 4));
                }
            }
            st.lo[W] ^= lo;
            st.hi[W] ^= hi;
        });
        if (blk == n_blocks - 1) st.hi[16] ^= 0x80000000u;
        wave_lds_sync();
        if (blk + 1 < n_blocks) issue(blk + 1);
        keccak_f1600(st);
    }
}

// SampleInBall (hashing.rs:43-100) for the lane's op.  `st`: c_tilde absorbed and padded, not yet permuted.  c: the lane's
// row of 256 int8 (SIB_C_STRIDE dwords, zeroed here), bw: the lane's row for the squeezed block (SIB_BLK_STRIDE dwords).
// SHAKE256(c_tilde): first 8 bytes = sign bits h; for i = 256 - tau .. 255: draw bytes j until j <= i; c[i] = c[j];
// c[j] = 1 - 2 * bit(i + tau - 256) of h.  Wave-uniform control flow: every lane stays until the whole wave is done.
__device__ __forceinline__ void sample_in_ball_lane(KeccakState& st, int tau, bool valid, uint32_t* c_row, uint32_t* bw) {
    int8_t* c = reinterpret_cast<int8_t*>(c_row);
    const uint8_t* bb = reinterpret_cast<const uint8_t*>(bw);
#pragma unroll
    for (int i = 0; i < 64; i++) c_row[i] = 0;
    keccak_f1600(st);
    const uint64_t h64 = ((uint64_t)st.hi[0] << 32) | st.lo[0];  // hashing.rs:55-56
    static_for<0, 34>([&](auto wc) { constexpr int W = decltype(wc)::value; bw[W] = state_word<W>(st); });
    int pos = 8;
    int i = valid ? 256 - tau : 256;
    for (;;) {
        while (i < 256 && pos < SHAKE256_RATE) {
            const int j = bb[pos++];
            if (j <= i) {  // hashing.rs:68-83
                c[i] = c[j];
                const int index = i + tau - 256;
                const uint32_t bit = (uint32_t)((h64 >> index) & 1u);
                c[j] = (int8_t)(1 - 2 * (int)bit);
                i++;
            }
        }
        if (!__any(i < 256)) break;
        if (i < 256) {  // this lane used up its block (rare): squeeze the next one
            keccak_f1600(st);
            static_for<0, 34>([&](auto wc) { constexpr int W = decltype(wc)::value; bw[W] = state_word<W>(st); });
            pos = 0;
        }
    }
}

// SampleInBall (hashing.rs:43-100) for ONE op by a whole wave (small calls): the sponge on the interleaved cooperative form
// (keccak_coop2.h), the Fisher-Yates walk with everything wave-uniform and in registers -- the squeezed block as one dword per lane
// (fetched with v_readlane at the walk's byte position), c as the wave's output dword itself (byte k of lane l = c[64 k + l]: what
// k_sample_in_ball<.., C8> writes), read with v_readlane and updated by the one lane that owns the coefficient.  A step is ~15 scalar /
// vector instructions instead of four dependent LDS accesses (the walk of ~55 steps: 1.7 instead of 6 us).
// `ct`: the op's c_tilde (CT bytes); bw: 34 dwords of the wave's LDS (the block on its way from the state's lanes to "dword d in lane d").
template <int CT>
__device__ __forceinline__ uint32_t sample_in_ball_coop2(const uint8_t* __restrict__ ct, int tau, uint32_t* bw, int lane, const Coop2Lane& c) {
    uint32_t lo = 0, hi = 0;
    if (c.active && c.word < CT / 8) {
        lo = load_le32(ct + 8 * c.word);
        hi = load_le32(ct + 8 * c.word + 4);
    }
    if (c.active && c.word == CT / 8) lo ^= 0x1Fu;
    if (c.active && c.word == SHAKE256_RATE / 8 - 1) hi ^= 0x80000000u;
    uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
    uint32_t creg = 0;
    int pos = 8, i = 256 - tau;
    uint32_t h_lo = 0, h_hi = 0;
    bool first = true;
    auto set_byte = [&](int idx, uint32_t val) {  // c[idx] = val (idx wave-uniform)
        const int sh = 8 * (idx >> 6);
        if (lane == (idx & 63)) creg = (creg & ~(0xFFu << sh)) | (val << sh);
    };
    for (;;) {  // wave-uniform
        keccak_f1600_coop2(v, c);
        coop2_to_lohi(v, lane, lo, hi);
        if (c.active && c.word < SHAKE256_RATE / 8) bw[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
        wave_lds_sync();
        const uint32_t blk = lane < SHAKE256_RATE / 4 ? bw[lane] : 0u;  // dword d of the block in lane d
        wave_lds_sync();
        if (first) {  // hashing.rs:55-56: the first eight bytes are the sign bits
            h_lo = (uint32_t)__builtin_amdgcn_readlane((int)blk, 0);
            h_hi = (uint32_t)__builtin_amdgcn_readlane((int)blk, 1);
        }
        first = false;
        while (i < 256 && pos < SHAKE256_RATE) {
            const uint32_t dw = (uint32_t)__builtin_amdgcn_readlane((int)blk, __builtin_amdgcn_readfirstlane(pos >> 2));
            const int j = (int)((dw >> (8 * (pos & 3))) & 0xFFu);
            pos++;
            if (j <= i) {  // hashing.rs:68-83
                const uint32_t cj = ((uint32_t)__builtin_amdgcn_readlane((int)creg, __builtin_amdgcn_readfirstlane(j & 63)) >> (8 * (j >> 6))) & 0xFFu;
                set_byte(i, cj);
                const int index = i + tau - 256;
                const uint32_t bit = ((index < 32 ? h_lo >> index : h_hi >> (index - 32)) & 1u);
                set_byte(j, bit ? 0xFFu : 0x01u);
                i++;
            }
        }
        if (i >= 256) break;
        pos = 0;  // block used up (rare): squeeze the next one
    }
    return creg;
}

}  // namespace mldsa
