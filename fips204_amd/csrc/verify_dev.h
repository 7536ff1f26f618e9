// Device pieces of verify_internal shared by the batch kernel (k_verify_main, kernels_codec.hip) and the single-launch kernel of small
// calls (k_verify_small, kernels_small.hip): the wave-parallel hint decoder of sigDecode.
#pragma once
#include "keccak_coop2.h"
#include "sampler_dev.h"

namespace mldsa {

__device__ __forceinline__ void wave_lds_sync_c() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------
// sig_decode part 2: hint_bit_unpack (conversion.rs:340-414) by one wave, inside k_verify_main.  Output: a 256-bit
// mask per hint polynomial in the wave's LDS words mw[k * 8] and the validity flag (false = the reference returns Err).
// The reference walks the omega + k bytes serially; here every lane owns one position byte (two when omega > 64) and
// tests the reference's conditions for it independently:
//   limits y[omega + i] non-decreasing and <= omega; positions strictly increasing inside a polynomial; bytes past
//   the last limit zero.
// (As a lane-per-op kernel of its own the walk was ~60 dependent byte loads, ~190 us whatever the batch: the longest
// link of a small batch's chain.)
constexpr int HINT_LDS_DWORDS = 24 + 64;  // 96 hint bytes (omega + k <= 84) | k * 8 <= 64 mask words
struct HintBytes { uint32_t b0, b1; };       // bytes lane and 64 + lane of the hint section
// the loads are issued early (with the op's first A_hat row) and consumed after the forward transforms
__device__ __forceinline__ HintBytes hint_load(const uint8_t* __restrict__ y, int omega, int k, int lane) {
    HintBytes h;
    h.b0 = lane < omega + k ? y[lane] : 0u;  // ML-DSA-65: 61 bytes, the signature ends there
    h.b1 = lane + 64 < omega + k ? y[lane + 64] : 0u;
    return h;
}
template <int K>
__device__ __forceinline__ bool hint_unpack_wave(HintBytes h, int omega, uint32_t* __restrict__ hl, int lane) {
    uint8_t* yb = reinterpret_cast<uint8_t*>(hl);
    uint32_t* mw = hl + 24;
    yb[lane] = (uint8_t)h.b0;
    if (lane < 32) yb[lane + 64] = (uint8_t)h.b1;
    mw[lane] = 0;
    wave_lds_sync_c();
    // lane i < K keeps limit i; they are handed round with v_readlane
    const int lim = lane < K ? yb[omega + lane] : 0;
    int bad = 0;
    if (lane < K) {
        const int prev = lane ? yb[omega + lane - 1] : 0;
        bad = (lim < prev) | (lim > omega);
    }
    const int last = __builtin_amdgcn_readlane(lim, K - 1);
    for (int j = lane; j < omega; j += 64) {
        const int pos = yb[j];
        if (j < last) {
            int i = 0, first = 0;  // the polynomial position j belongs to = limits at or below j (at most K - 1: `last` is above)
#pragma unroll
            for (int t = 0; t < K - 1; t++) {
                const int lt = __builtin_amdgcn_readlane(lim, t);
                const bool le = lt <= j;
                i += le ? 1 : 0;
                first = le ? lt : first;
            }
            if (j > first && yb[j - 1] >= pos) bad = 1;
            atomicOr(&mw[i * 8 + (pos >> 5)], 1u << (pos & 31));
        } else if (pos != 0) {
            bad = 1;
        }
    }
    wave_lds_sync_c();
    return __ballot(bad) == 0ull;
}

// mu = H(tr || M', 64) (ml_dsa.rs:185-196 / 386-397) for ONE op by a whole wave on the interleaved cooperative sponge (keccak_coop2.h), any
// message length.  M' = M (internal), 0x00 | len(ctx) | ctx | M (pure) or 0x01 | len(ctx) | ctx | OID | PH(M) (pre-hash; the caller passes
// OID | PH(M) as the message).  The checks are k_mu's: offsets are the caller's and are never trusted -- the op is hashed only if its
// pairs lie inside [off[0], off[n_call]] in order; a ctx longer than 255 bytes or a malformed pair refuses the op before a byte of it is
// read.  `t` = the op's index in the CALL's offset tables.  Returns the refusal flag (0 = hashed; 1 = ctx too long; 2 = malformed
// offsets; else key_bad) and leaves the 64 bytes of mu in (lo, hi) of the lanes holding state words 0 .. 7 (zero when refused).
__device__ __forceinline__ int mu_coop2(const uint8_t* trp, int mode, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* ctxs,
                                        const uint64_t* ctx_off, size_t t, size_t n_call, int key_bad, uint32_t& lo_out, uint32_t& hi_out, int lane,
                                        const Coop2Lane& c) {
    const uint64_t m0 = msg_off[t], m1 = msg_off[t + 1];
    bool bad_off = !(msg_off[0] <= m0 && m0 <= m1 && m1 <= msg_off[n_call]);
    const uint8_t* mp = msgs + m0;
    size_t mlen = (size_t)(m1 - m0), clen = 0;
    bad_off |= mlen != 0 && msgs == nullptr;  // offsets that name bytes of a NULL array
    const uint8_t* cp = nullptr;
    if (ctx_off) {
        const uint64_t c0 = ctx_off[t], c1 = ctx_off[t + 1];
        bad_off |= !(ctx_off[0] <= c0 && c0 <= c1 && c1 <= ctx_off[n_call]);
        cp = ctxs + c0;
        clen = (size_t)(c1 - c0);
        bad_off |= clen != 0 && ctxs == nullptr;
    }
    // 2: malformed offsets or key index out of range; 1: ctx too long (lib.rs:274, 368, 589, 605: every entry point)
    const int flag = bad_off ? 2 : clen > 255 ? 1 : key_bad;
    const bool live = !bad_off && clen <= 255;
    if (!live) mlen = clen = 0;
    const size_t pre = (mode == MLDSA_MODE_INTERNAL) ? 0 : 2 + clen;
    const size_t total = live ? 64 + pre + mlen : 0;
    const size_t blocks = live ? total / SHAKE256_RATE + 1 : 0;  // the pad always fits in the last block
    auto byte_at = [&](size_t pos) -> uint32_t {
        if (pos < total) {
            if (pos < 64) return trp[pos];
            if (pos < 64 + pre) {
                const size_t q = pos - 64;
                return q == 0 ? (uint32_t)(mode == MLDSA_MODE_PREHASH ? 1 : 0) : q == 1 ? (uint32_t)clen : cp[q - 2];
            }
            return mp[pos - 64 - pre];
        }
        return pos == total ? 0x1Fu : 0u;
    };
    auto dword_at = [&](size_t pos) -> uint32_t {  // whole dwords of tr and of the message by one byte-granular load, boundaries from bytes
        if (pos + 4 <= 64) return load_le32(trp + pos);
        if (pos >= 64 + pre && pos + 4 <= total) return load_le32(mp + (pos - 64 - pre));
        if (pos > total) return 0u;
        return byte_at(pos) | (byte_at(pos + 1) << 8) | (byte_at(pos + 2) << 16) | (byte_at(pos + 3) << 24);
    };
    uint32_t v = 0;
    const bool absorbs = c.active && c.word < SHAKE256_RATE / 8;
    for (size_t b = 0; b < blocks; b++) {  // wave-uniform
        if (absorbs) {
            const size_t off = b * SHAKE256_RATE + 8 * (size_t)c.word;
            const uint32_t lo = dword_at(off);
            uint32_t hi = dword_at(off + 4);
            if (b == blocks - 1 && c.word == SHAKE256_RATE / 8 - 1) hi ^= 0x80000000u;
            v ^= coop2_from_lohi(lo, hi, c);
        }
        keccak_f1600_coop2(v, c);
    }
    coop2_to_lohi(v, lane, lo_out, hi_out);
    return flag;
}

}  // namespace mldsa
