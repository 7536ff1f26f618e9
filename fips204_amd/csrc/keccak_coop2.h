// Wave-cooperative Keccak-f[1600], second form: ONE state per wavefront, one 32-bit HALF of a state word per lane, in the
// bit-interleaved representation -- and one level of five cross-lane gathers per round instead of two levels of ten.
//
// keccak_coop.h (two states per wavefront, the (lo, hi) halves of a word in two registers of one lane) spends a round on 36 VALU
// instructions and 10 ds_bpermute in two dependent levels: 3.8 us per permutation, and a one-op verification is 12 serial
// permutations (5 of ExpandA, 7 of the c~ hash).  Measured in round 4 (tools/ubench_keccak_coop.hip): a ds_bpermute costs ~16 cycles
// of issue even for a lone wave and each dependent level a full LDS round trip.  This form cuts all three:
//
//   * a 64-bit word W is held as E = its even bits and O = its odd bits in TWO LANES (lane i: E, lane 32 + i: O).  A 64-bit rotate
//     by 2 k is then a 32-bit rotate by k of both halves, a rotate by 2 k + 1 takes the halves from each other's lane
//     (E' = rot32(O, k + 1), O' = rot32(E, k)): no rotate ever needs both halves in one lane.  Every bitwise operation of the round
//     (theta's parities, chi, iota) works on E and O independently, so ONE instruction serves both halves: half the VALU count.
//   * pi moves every word anyway, so the half-swap of an odd rotate costs nothing: a lane gathers the half it needs.
//   * theta is applied AFTER the move: B[X, Y] = rot(A[x, y] ^ D[x]) -- the destination lane gathers the source word and the four
//     partial column parities D[x] is made of (both 16-lane rows' parts of columns x - 1 and x + 1, from the half that rot1
//     requires) in ONE level of five gathers, all issued back to back.
//
// Layout: half h = lane >> 5 (0: E, 1: O); inside a half, planes y = 0, 1, 2 at lanes 5 y + x of the first 16-lane row, y = 3, 4 at
// 16 + 5 (y - 3) + x of the second (as keccak_coop.h): theta's row-local parities are two DPP row shifts, chi's neighbours DPP shifts
// inside the plane's five lanes with a select for the wrap-around.  Per round: 5 ds_bpermute, ~18 VALU.
// Absorb / squeeze convert between (lo, hi) and (E, O): coop2_from_lohi / coop2_to_lohi.
#pragma once
#include "keccak_coop.h"

namespace mldsa {

struct Coop2Lane {
    int a_src;             // ds_bpermute byte address of the source word's half this lane needs
    int m_src, p_src;      // ... of the row-0 parts of the parities of columns x_src - 1 / x_src + 1 (the row-1 parts: + 44 bytes)
    int prot, rot;         // rotate of the x_src + 1 parity (1 or 0) and the lane's final rotate
    int word;              // x + 5 y of the lane's state word (valid when active)
    bool active, first, wrap1, wrap2, odd;
};

__device__ __forceinline__ Coop2Lane coop2_lane(int lane) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // r[x + 5 y]
    Coop2Lane c;
    const int h = lane >> 5, i = lane & 31;
    c.odd = h != 0;
    c.active = i < 15 || (i >= 16 && i < 26);
    int X = 0, Y = 0;
    if (c.active) {
        const int r = i < 16 ? i : i - 16;
        X = r % 5;
        Y = r / 5 + (i < 16 ? 0 : 3);
    }
    c.word = X + 5 * Y;
    // B[X, Y] = rot(A[xs, ys] ^ D[xs], r[xs, ys]) with xs = X + 3 Y, ys = X
    const int xs = (X + 3 * Y) % 5, ys = X;
    int r = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) r = (k == xs + 5 * ys) ? RHO[k] : r;  // (no indexed constant array: a select chain, once per kernel)
    if (!c.active) r = 0;
    const int hs = h ^ (r & 1);                                  // the half of W = A ^ D this lane's half of rot64(W, r) is made of
    c.rot = ((r + (r & 1) * (h == 0 ? 1 : -1)) >> 1) & 31;       // r even: r / 2; odd: E' = rot32(O, (r + 1) / 2), O' = rot32(E, (r - 1) / 2)
    c.a_src = (32 * hs + coop_pos(xs, ys)) << 2;
    // D_E[x] = C_E[x - 1] ^ rot32(C_O[x + 1], 1);  D_O[x] = C_O[x - 1] ^ C_E[x + 1]   (rot64 by 1 in the interleaved form)
    // partial parities after the two row shifts: planes 0-2 at lanes 10 + x of the half, planes 3-4 at lanes 21 + x (= + 11 lanes)
    c.m_src = (32 * hs + 10 + (xs + 4) % 5) << 2;
    c.p_src = (32 * (hs ^ 1) + 10 + (xs + 1) % 5) << 2;
    c.prot = hs == 0 ? 1 : 0;
    c.first = i == 0;
    c.wrap1 = X == 4;
    c.wrap2 = X >= 3;
    return c;
}

__device__ __forceinline__ uint32_t coop2_rotl(uint32_t v, int r) { return __builtin_amdgcn_alignbit(v, v, (32 - r) & 31); }

// bit-interleaved round constants: RC_E[r] = even bits, RC_O[r] = odd bits of RC[r]
__device__ __constant__ const uint32_t KECCAK_RC_E[24] = {
    0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000001u,
    0x00000000u, 0x00000000u, 0x00000001u, 0x00000000u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000001u,
    0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0x00000000u, 0x00000001u, 0x00000000u};
__device__ __constant__ const uint32_t KECCAK_RC_O[24] = {
    0x00000000u, 0x00000089u, 0x8000008bu, 0x80008080u, 0x0000008bu, 0x00008000u, 0x80008088u, 0x80000082u,
    0x0000000bu, 0x0000000au, 0x00008082u, 0x00008003u, 0x0000808bu, 0x8000000bu, 0x8000008au, 0x80000081u,
    0x80000081u, 0x80000008u, 0x00000083u, 0x80008003u, 0x80008088u, 0x80000088u, 0x00008000u, 0x80008082u};

// One round on the lane's half word.  Every cross-lane operation is executed by the WHOLE wave; inactive lanes carry junk that no
// active lane ever reads.
__device__ __forceinline__ void coop2_round(uint32_t& v, const Coop2Lane& c, uint32_t rc_e, uint32_t rc_o) {
    // theta, row-local part: parities of the planes of one 16-lane row (row_shr:5 = 0x115, row_shr:10 = 0x11A)
    const uint32_t t = v ^ coop_dpp<0x115>(v) ^ coop_dpp<0x11A>(v);
    // the one gather level: the source word's half and the four partial parities of its two neighbouring columns
    const uint32_t a = coop_gather(v, c.a_src);
    const uint32_t m0 = coop_gather(t, c.m_src), m1 = coop_gather(t, c.m_src + 44);
    const uint32_t p0 = coop_gather(t, c.p_src), p1 = coop_gather(t, c.p_src + 44);
    const uint32_t p = coop2_rotl(p0 ^ p1, c.prot);
    // rho + pi on A ^ D
    const uint32_t b = coop2_rotl(xor3(a, m0, m1) ^ p, c.rot);
    // chi: B[x + 1], B[x + 2] of the same plane (row_shl:1 / :2; the lanes that wrap take row_shr:4 / :3)
    const uint32_t s1 = coop_dpp<0x101>(b), w1 = coop_dpp<0x114>(b), s2 = coop_dpp<0x102>(b), w2 = coop_dpp<0x113>(b);
    v = chi(b, c.wrap1 ? w1 : s1, c.wrap2 ? w2 : s2);
    // iota
    if (c.first) v ^= c.odd ? rc_o : rc_e;
}

__device__ __forceinline__ void keccak_f1600_coop2(uint32_t& v, const Coop2Lane& c) {
#pragma unroll
    for (int r = 0; r < 24; r++) coop2_round(v, c, KECCAK_RC_E[r], KECCAK_RC_O[r]);
}

// ---- (lo, hi) <-> (E, O)
// even bits of x compressed into the low 16 bits
__device__ __forceinline__ uint32_t coop2_even16(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// the low 16 bits of x spread to the even bit positions
__device__ __forceinline__ uint32_t coop2_spread16(uint32_t x) {
    x &= 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// this lane's half (E for lanes 0-31, O for lanes 32-63) of the 64-bit word (hi:lo); both lanes of a word call it with the same (lo, hi)
__device__ __forceinline__ uint32_t coop2_from_lohi(uint32_t lo, uint32_t hi, const Coop2Lane& c) {
    const int sh = c.odd ? 1 : 0;
    return coop2_even16(lo >> sh) | (coop2_even16(hi >> sh) << 16);
}
// the word's (lo, hi), in both of its lanes (one gather: the partner half sits 32 lanes away)
__device__ __forceinline__ void coop2_to_lohi(uint32_t v, int lane, uint32_t& lo, uint32_t& hi) {
    const uint32_t other = coop_gather(v, (lane ^ 32) << 2);
    const uint32_t e = lane < 32 ? v : other, o = lane < 32 ? other : v;
    lo = coop2_spread16(e) | (coop2_spread16(o) << 1);
    hi = coop2_spread16(e >> 16) | (coop2_spread16(o >> 16) << 1);
}

}  // namespace mldsa
