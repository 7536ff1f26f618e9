// Wave-cooperative Keccak-f[1600] for SMALL batches: the helpers of the lane layout (a state's words spread over the lanes of a wave).
//
// The pipelines' sponges are lane-per-state (keccak.h): 64 states per wavefront, 190 VALU instructions per round and no cross-lane
// traffic -- what makes the throughput of a large batch -- but ONE permutation is a chain of 4 560 dependent instructions, 9.4 us,
// and a hash of a few blocks is that many times 9.4 us however few operations a call has.  Spread over the lanes of a wave a
// permutation takes 2.2 us (keccak_coop2.h: one state per wave, one 32-bit half word per lane in the bit-interleaved form, five
// ds_bpermute in one level per round).  Round 4's first form -- two states per wave, (lo, hi) of a word in one lane, ten gathers in two
// levels, 3.8 us -- is kept in tools/ubench_keccak_coop.hip, which measures all of them against each other (EXPERIMENTS.md).
//
// Layout per 32-lane half-wave: planes y = 0, 1, 2 at lanes 5 y + x of the first 16-lane row, y = 3, 4 at lanes 16 + 5 (y - 3) + x of
// the second: theta's row-local parities are two DPP row shifts, chi's neighbours DPP shifts inside the plane's five lanes.
#pragma once
#include "keccak.h"

namespace mldsa {

__device__ __forceinline__ int coop_pos(int x, int y) {
    x %= 5;
    y %= 5;
    return y < 3 ? 5 * y + x : 16 + 5 * (y - 3) + x;
}

template <int CTRL>
__device__ __forceinline__ uint32_t coop_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t coop_gather(uint32_t v, int byte_addr) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }

}  // namespace mldsa
