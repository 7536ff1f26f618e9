// Wave-cooperative Keccak-f[1600] for SMALL batches: one 64-bit state word per lane, two states per wavefront.
//
// The pipelines' sponges are lane-per-state (keccak.h): 64 states per wavefront, 190 VALU instructions per round and no cross-lane
// traffic -- what makes the throughput of a large batch -- but ONE permutation is a chain of 4 560 dependent instructions, 9.4 us,
// and a hash of a few blocks is that many times 9.4 us however few operations a call has (the c~ hash is 71 of the 171 us of a
// one-op verification, tr = H(pk) 148 of the 310 us of a one-op key generation).  Spread over 25 lanes a permutation takes 3.8 us
// (tools/ubench_keccak_coop.hip; 0.5-0.8 G permutations/s against 9.3 G/s lane-per-state: for calls of a few thousand ops at most).
//
// Layout per 32-lane half-wave: planes y = 0, 1, 2 at lanes 5 y + x of the first 16-lane row, y = 3, 4 at lanes 16 + 5 (y - 3) + x of
// the second.  Per round: theta -- the parity of the planes of one row by two DPP row shifts (VALU rate), then each lane gathers both
// rows' parts of its two neighbouring columns (4 ds_bpermute); rho -- a per-lane rotate; pi -- one gather; chi -- the row's next two
// words by DPP shifts inside the plane's five lanes, with a select for the wrap-around.  10 gathers in two dependent levels per
// round (a first version with one gather per (x, y) neighbour had 16-18 in three to six levels: 5.8-6.25 us).
#pragma once
#include "keccak.h"

namespace mldsa {

struct CoopLane {
    int colm1, colp1, pi_src;  // ds_bpermute byte addresses
    int rot;                   // rho offset of the lane's word
    int word;                  // index x + 5 y of the lane's state word (valid when active)
    bool active, first, wrap1, wrap2;
};

__device__ __forceinline__ int coop_pos(int x, int y) {
    x %= 5;
    y %= 5;
    return y < 3 ? 5 * y + x : 16 + 5 * (y - 3) + x;
}

__device__ __forceinline__ CoopLane coop_lane(int lane) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // r[x + 5 y]
    CoopLane c;
    const int base = lane & 32, i = lane & 31;
    c.active = i < 15 || (i >= 16 && i < 26);
    int x = 0, y = 0;
    if (c.active) {
        const int r = i < 16 ? i : i - 16;
        x = r % 5;
        y = r / 5 + (i < 16 ? 0 : 3);
    }
    c.word = x + 5 * y;
    // the row shifts leave the parities of planes 0-2 in lanes 10 + x and of planes 3, 4 in lanes 21 + x (= 11 lanes up)
    c.colm1 = (base + 10 + (x + 4) % 5) << 2;
    c.colp1 = (base + 10 + (x + 1) % 5) << 2;
    c.pi_src = (base + coop_pos(x + 3 * y, x)) << 2;  // B[X, Y] = rot(A[x, y]) with x = X + 3 Y, y = X
    int rot = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) rot = (k == c.word) ? RHO[k] : rot;  // (no indexed constant array: a select chain, once per kernel)
    c.rot = c.active ? rot : 0;
    c.first = i == 0;
    c.wrap1 = x == 4;
    c.wrap2 = x >= 3;
    return c;
}

template <int CTRL>
__device__ __forceinline__ uint32_t coop_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t coop_gather(uint32_t v, int byte_addr) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }

// One round on the (lo, hi) halves of the lane's word.  Every cross-lane operation is executed by the WHOLE wave (no divergence
// around them): inactive lanes carry junk that no active lane ever reads.
__device__ __forceinline__ void coop_round(uint32_t& lo, uint32_t& hi, const CoopLane& c, uint32_t rc_lo, uint32_t rc_hi) {
    // theta: row-local column parities (row_shr:5 = 0x115, row_shr:10 = 0x11A) ...
    const uint32_t tl = lo ^ coop_dpp<0x115>(lo) ^ coop_dpp<0x11A>(lo), th = hi ^ coop_dpp<0x115>(hi) ^ coop_dpp<0x11A>(hi);
    // ... both rows' parts of both neighbouring columns
    const uint32_t ml = coop_gather(tl, c.colm1) ^ coop_gather(tl, c.colm1 + 44), mh = coop_gather(th, c.colm1) ^ coop_gather(th, c.colm1 + 44);
    const uint32_t pl = coop_gather(tl, c.colp1) ^ coop_gather(tl, c.colp1 + 44), ph = coop_gather(th, c.colp1) ^ coop_gather(th, c.colp1 + 44);
    lo ^= ml ^ __funnelshift_l(ph, pl, 1);
    hi ^= mh ^ __funnelshift_l(pl, ph, 1);
    // rho
    const int r = c.rot & 31;
    uint32_t rl = __funnelshift_l(hi, lo, r), rh = __funnelshift_l(lo, hi, r);
    if (c.rot & 32) {
        const uint32_t t = rl;
        rl = rh;
        rh = t;
    }
    // pi
    const uint32_t bl = coop_gather(rl, c.pi_src), bh = coop_gather(rh, c.pi_src);
    // chi: B[x + 1], B[x + 2] of the same plane (row_shl:1 / :2 = 0x101 / 0x102; the lanes that wrap take row_shr:4 / :3 = 0x114 / 0x113)
    const uint32_t s1l = coop_dpp<0x101>(bl), w1l = coop_dpp<0x114>(bl), s1h = coop_dpp<0x101>(bh), w1h = coop_dpp<0x114>(bh);
    const uint32_t s2l = coop_dpp<0x102>(bl), w2l = coop_dpp<0x113>(bl), s2h = coop_dpp<0x102>(bh), w2h = coop_dpp<0x113>(bh);
    lo = chi(bl, c.wrap1 ? w1l : s1l, c.wrap2 ? w2l : s2l);
    hi = chi(bh, c.wrap1 ? w1h : s1h, c.wrap2 ? w2h : s2h);
    // iota
    if (c.first) {
        lo ^= rc_lo;
        hi ^= rc_hi;
    }
}

__device__ __forceinline__ void keccak_f1600_coop(uint32_t& lo, uint32_t& hi, const CoopLane& c) {
#pragma unroll
    for (int r = 0; r < 24; r++) coop_round(lo, hi, c, KECCAK_RC_LO[r], ((KECCAK_RC_HI_BITS >> r) & 1u) << 31);
}

}  // namespace mldsa
