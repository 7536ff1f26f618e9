// Device pieces of verify_internal shared by the batch kernel (k_verify_main, kernels_codec.hip) and the single-launch kernel of small
// calls (k_verify_small, kernels_small.hip): the wave-parallel hint decoder of sigDecode.
#pragma once
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

}  // namespace mldsa
