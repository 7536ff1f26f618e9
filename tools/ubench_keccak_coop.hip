// Wave-cooperative Keccak-f[1600] against the lane-per-state form the library uses (csrc/keccak.h) -- MEASURED, to replace the
// estimate DESIGN.md carried for why `north_star`'s "one warp per polynomial Keccak" is applied to the output side only.
//
// Cooperative layout: one 64-bit lane word A[x, y] per SIMD lane (lane i = x + 5 y of a 32-lane half-wave: two states per
// wavefront, 25 of every 32 lanes active).  Per round: theta = 3 + 2 cross-lane gathers (column parity by adding the rows one,
// two and four steps away; the two neighbouring columns' parities), rho = a per-lane 64-bit rotate, pi = 1 gather,
// chi = 2 gathers (the row's next two words), iota on lane 0: 8 gathers x two 32-bit words = 16 ds_bpermute_b32 per round.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_keccak_coop tools/ubench_keccak_coop.hip && ./tools/ubench_keccak_coop
//
// Prints, per waves-per-SIMD, the time per permutation of one wave (latency) and the chip's permutations per second (throughput)
// for both forms, after checking on random states that they compute the same permutation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../fips204_amd/csrc/keccak.h"
#include "../fips204_amd/csrc/keccak_coop2.h"

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __constant__ const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // r[x + 5 y]

__device__ __forceinline__ uint32_t gather(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }

struct CoopIdx {  // per-lane source lanes of the round's gathers
    int row1, row2, row4, colm1, colp1, pi_src, rowp1, rowp2;
    int rot;
    bool active, first;
};

__device__ __forceinline__ CoopIdx coop_idx(int lane) {
    CoopIdx c;
    const int base = lane & 32, i = lane & 31;
    c.active = i < 25;
    const int ii = c.active ? i : 0, x = ii % 5, y = ii / 5;
    auto at = [&](int xx, int yy) { return base + (xx % 5) + 5 * (yy % 5); };
    c.row1 = at(x, y + 1); c.row2 = at(x, y + 2); c.row4 = at(x, y + 4);
    c.colm1 = at(x + 4, y); c.colp1 = at(x + 1, y);
    c.pi_src = at(x + 3 * y, x);  // B[X, Y] = rot(A[x, y]) with X = y, Y = 2 x + 3 y  =>  x = X + 3 Y, y = X
    c.rowp1 = at(x + 1, y); c.rowp2 = at(x + 2, y);
    c.rot = RHO[ii];
    c.first = i == 0;
    return c;
}

__device__ __forceinline__ void coop_round(uint32_t& lo, uint32_t& hi, const CoopIdx& c, int round) {
    // theta
    uint32_t tl = lo ^ gather(lo, c.row1), th = hi ^ gather(hi, c.row1);
    uint32_t ul = tl ^ gather(tl, c.row2), uh = th ^ gather(th, c.row2);
    const uint32_t cl = ul ^ gather(lo, c.row4), ch = uh ^ gather(hi, c.row4);  // column parity in every lane of the column
    const uint32_t ml = gather(cl, c.colm1), mh = gather(ch, c.colm1), pl = gather(cl, c.colp1), ph = gather(ch, c.colp1);
    lo ^= ml ^ __funnelshift_l(ph, pl, 1);  // rotl64(C[x + 1], 1)
    hi ^= mh ^ __funnelshift_l(pl, ph, 1);
    // rho: rotl64 by the lane's own amount
    const int r = c.rot & 31;
    uint32_t rl = __funnelshift_l(hi, lo, r), rh = __funnelshift_l(lo, hi, r);
    if (c.rot & 32) { const uint32_t t = rl; rl = rh; rh = t; }
    // pi
    const uint32_t bl = gather(rl, c.pi_src), bh = gather(rh, c.pi_src);
    // chi
    const uint32_t b1l = gather(bl, c.rowp1), b1h = gather(bh, c.rowp1), b2l = gather(bl, c.rowp2), b2h = gather(bh, c.rowp2);
    lo = mldsa::chi(bl, b1l, b2l);
    hi = mldsa::chi(bh, b1h, b2h);
    // iota
    if (c.first) {
        lo ^= mldsa::KECCAK_RC_LO[round];
        hi ^= ((mldsa::KECCAK_RC_HI_BITS >> round) & 1u) << 31;
    }
}

// ---- second cooperative form: the same layout, THREE dependent gather levels per round instead of six --
//   level 1: the four other words of the lane's column at once (theta's column parity);
//   level 2: the parities of the two neighbouring columns;
//   level 3: pi folded into chi's gathers: the lane fetches its own B word and the row's next two straight from the lanes that hold
//            them BEFORE pi (rho is applied at the source, where the rotation amount belongs)
// 18 gathers x 2 words per round (the first form: 16), but only three of them wait for each other.
struct Coop3Idx {
    int row[4], colm1, colp1, b0, b1, b2, rot;
    bool active, first;
};

__device__ __forceinline__ Coop3Idx coop3_idx(int lane) {
    Coop3Idx c;
    const int base = lane & 32, i = lane & 31;
    c.active = i < 25;
    const int ii = c.active ? i : 0, x = ii % 5, y = ii / 5;
    auto at = [&](int xx, int yy) { return (base + (xx % 5) + 5 * (yy % 5)) << 2; };  // byte address for ds_bpermute
    for (int k = 0; k < 4; k++) c.row[k] = at(x, y + 1 + k);
    c.colm1 = at(x + 4, y); c.colp1 = at(x + 1, y);
    auto pi_src = [&](int X, int Y) { return at(X + 3 * Y, X); };  // B[X, Y] = rot(A[x, y]) with x = X + 3 Y, y = X
    c.b0 = pi_src(x, y); c.b1 = pi_src((x + 1) % 5, y); c.b2 = pi_src((x + 2) % 5, y);
    c.rot = RHO[ii];
    c.first = i == 0;
    return c;
}

__device__ __forceinline__ uint32_t gather4(uint32_t v, int byte_addr) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }

__device__ __forceinline__ void coop3_round(uint32_t& lo, uint32_t& hi, const Coop3Idx& c, int round) {
    // theta, level 1
    const uint32_t l1 = gather4(lo, c.row[0]), l2 = gather4(lo, c.row[1]), l3 = gather4(lo, c.row[2]), l4 = gather4(lo, c.row[3]);
    const uint32_t h1 = gather4(hi, c.row[0]), h2 = gather4(hi, c.row[1]), h3 = gather4(hi, c.row[2]), h4 = gather4(hi, c.row[3]);
    const uint32_t cl = mldsa::xor3(mldsa::xor3(lo, l1, l2), l3, l4), ch = mldsa::xor3(mldsa::xor3(hi, h1, h2), h3, h4);
    // level 2
    const uint32_t ml = gather4(cl, c.colm1), mh = gather4(ch, c.colm1), pl = gather4(cl, c.colp1), ph = gather4(ch, c.colp1);
    lo ^= ml ^ __funnelshift_l(ph, pl, 1);
    hi ^= mh ^ __funnelshift_l(pl, ph, 1);
    // rho at the source
    const int r = c.rot & 31;
    uint32_t rl = __funnelshift_l(hi, lo, r), rh = __funnelshift_l(lo, hi, r);
    if (c.rot & 32) { const uint32_t t = rl; rl = rh; rh = t; }
    // level 3: pi + chi
    const uint32_t b0l = gather4(rl, c.b0), b1l = gather4(rl, c.b1), b2l = gather4(rl, c.b2);
    const uint32_t b0h = gather4(rh, c.b0), b1h = gather4(rh, c.b1), b2h = gather4(rh, c.b2);
    lo = mldsa::chi(b0l, b1l, b2l);
    hi = mldsa::chi(b0h, b1h, b2h);
    if (c.first) {
        lo ^= mldsa::KECCAK_RC_LO[round];
        hi ^= ((mldsa::KECCAK_RC_HI_BITS >> round) & 1u) << 31;
    }
}

__global__ __launch_bounds__(256) void k_coop3(uint32_t* __restrict__ io, int perms) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const Coop3Idx c = coop3_idx(lane);
    uint32_t* st = io + (wave * 2 + (lane >> 5)) * 50;
    uint32_t lo = c.active ? st[2 * (lane & 31)] : 0, hi = c.active ? st[2 * (lane & 31) + 1] : 0;
    for (int p = 0; p < perms; p++)
#pragma unroll 1
        for (int r = 0; r < 24; r++) coop3_round(lo, hi, c, r);
    if (c.active) { st[2 * (lane & 31)] = lo; st[2 * (lane & 31) + 1] = hi; }
}

// ---- third cooperative form: data movement with DPP where the layout allows it, ds_bpermute only where it does not -------------
// Layout per 32-lane half: planes y = 0, 1, 2 at lanes 5 y + x of the first 16-lane row, y = 3, 4 at lanes 16 + 5 (y - 3) + x of the
// second.  theta: the parity of the planes of one row by two DPP row shifts (VALU rate), one gather to add the two rows' parts,
// two gathers for the neighbouring columns; pi: one gather; chi: the row's next two words by DPP shifts inside the plane's five
// lanes (with a select for the wrap-around).  4 gathers x 2 words per round instead of 16 / 18.
struct Coop4Idx {
    int partner, colm1, colp1, pi_src, rot;
    bool active, first, wrap1, wrap2;   // wrap1: x == 4 (x + 1 wraps); wrap2: x >= 3 (x + 2 wraps)
};

__device__ __forceinline__ int pos4(int x, int y) { x %= 5; y %= 5; return y < 3 ? 5 * y + x : 16 + 5 * (y - 3) + x; }

__device__ __forceinline__ Coop4Idx coop4_idx(int lane) {
    Coop4Idx c;
    const int base = lane & 32, i = lane & 31;
    c.active = i < 15 || (i >= 16 && i < 26);
    int x = 0, y = 0;
    if (c.active) { const int r = i < 16 ? i : i - 16; x = r % 5; y = r / 5 + (i < 16 ? 0 : 3); }
    // where the column parities end up after the row shifts: lanes 10 + x (planes 0-2) and 21 + x (planes 3, 4); the full parity
    // is assembled in lanes 10 + x
    c.partner = (base + (i >= 10 && i < 15 ? 21 + (i - 10) : i)) << 2;   // only lanes 10 .. 14 need it
    c.colm1 = (base + 10 + (x + 4) % 5) << 2;
    c.colp1 = (base + 10 + (x + 1) % 5) << 2;
    c.pi_src = (base + pos4(x + 3 * y, x)) << 2;   // B[X, Y] = rot(A[x, y]) with x = X + 3 Y, y = X
    c.rot = c.active ? RHO[x + 5 * y] : 0;
    c.first = i == 0;
    c.wrap1 = x == 4;
    c.wrap2 = x >= 3;
    return c;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ void coop4_round(uint32_t& lo, uint32_t& hi, const Coop4Idx& c, int round) {
    // theta: row-local parity (row_shr:5 = 0x115, row_shr:10 = 0x11A), rows added in lanes 10 + x, neighbours fetched from there
    const uint32_t tl = lo ^ dpp0<0x115>(lo) ^ dpp0<0x11A>(lo), th = hi ^ dpp0<0x115>(hi) ^ dpp0<0x11A>(hi);
#ifdef COOP4_THETA_TWO_LEVELS
    const uint32_t cl = tl ^ gather4(tl, c.partner), ch = th ^ gather4(th, c.partner);   // complete in lanes 10 .. 14 (partner = self elsewhere: junk, unused)
    const uint32_t ml = gather4(cl, c.colm1), mh = gather4(ch, c.colm1), pl = gather4(cl, c.colp1), ph = gather4(ch, c.colp1);
#else
    // both rows' parts of both neighbouring columns at once (colm1 / colp1 name lanes 10 + x'; the other row's part sits 11 lanes up)
    const uint32_t ml = gather4(tl, c.colm1) ^ gather4(tl, c.colm1 + 44), mh = gather4(th, c.colm1) ^ gather4(th, c.colm1 + 44);
    const uint32_t pl = gather4(tl, c.colp1) ^ gather4(tl, c.colp1 + 44), ph = gather4(th, c.colp1) ^ gather4(th, c.colp1 + 44);
#endif
    lo ^= ml ^ __funnelshift_l(ph, pl, 1);
    hi ^= mh ^ __funnelshift_l(pl, ph, 1);
    // rho
    const int r = c.rot & 31;
    uint32_t rl = __funnelshift_l(hi, lo, r), rh = __funnelshift_l(lo, hi, r);
    if (c.rot & 32) { const uint32_t t = rl; rl = rh; rh = t; }
    // pi
    const uint32_t bl = gather4(rl, c.pi_src), bh = gather4(rh, c.pi_src);
    // chi: B[x + 1], B[x + 2] of the same plane: row_shl:1 / row_shl:2 (0x101, 0x102), wrapped lanes take row_shr:4 / row_shr:3 (0x114, 0x113)
    // (every shift is executed by the whole wave; the select comes afterwards)
    const uint32_t s1l = dpp0<0x101>(bl), w1l = dpp0<0x114>(bl), s1h = dpp0<0x101>(bh), w1h = dpp0<0x114>(bh);
    const uint32_t s2l = dpp0<0x102>(bl), w2l = dpp0<0x113>(bl), s2h = dpp0<0x102>(bh), w2h = dpp0<0x113>(bh);
    const uint32_t b1l = c.wrap1 ? w1l : s1l, b1h = c.wrap1 ? w1h : s1h;
    const uint32_t b2l = c.wrap2 ? w2l : s2l, b2h = c.wrap2 ? w2h : s2h;
    lo = mldsa::chi(bl, b1l, b2l);
    hi = mldsa::chi(bh, b1h, b2h);
    if (c.first) {
        lo ^= mldsa::KECCAK_RC_LO[round];
        hi ^= ((mldsa::KECCAK_RC_HI_BITS >> round) & 1u) << 31;
    }
}

// io: per wave two states x 25 words x (lo, hi) in the canonical order x + 5 y; the kernel maps them to its lanes
__global__ __launch_bounds__(256) void k_coop4(uint32_t* __restrict__ io, int perms) {
    const int lane = threadIdx.x & 63, i = lane & 31;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const Coop4Idx c = coop4_idx(lane);
    int w = 0;
    if (c.active) { const int r = i < 16 ? i : i - 16; w = (r % 5) + 5 * (r / 5 + (i < 16 ? 0 : 3)); }
    uint32_t* st = io + (wave * 2 + (lane >> 5)) * 50;
    uint32_t lo = c.active ? st[2 * w] : 0, hi = c.active ? st[2 * w + 1] : 0;
    for (int p = 0; p < perms; p++)
#pragma unroll
        for (int r = 0; r < 24; r++) coop4_round(lo, hi, c, r);
    if (c.active) { st[2 * w] = lo; st[2 * w + 1] = hi; }
}

// round 5: ONE state per wave, one 32-bit half per lane in the bit-interleaved form, one level of five gathers per round (keccak_coop2.h)
__global__ __launch_bounds__(256) void k_coop5(uint32_t* __restrict__ io, int perms) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const mldsa::Coop2Lane c = mldsa::coop2_lane(lane);
    uint32_t* st = io + wave * 50;
    uint32_t v = c.active ? mldsa::coop2_from_lohi(st[2 * c.word], st[2 * c.word + 1], c) : 0;
    for (int p = 0; p < perms; p++) mldsa::keccak_f1600_coop2(v, c);
    uint32_t lo, hi;
    mldsa::coop2_to_lohi(v, lane, lo, hi);
    if (c.active) st[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
}

__global__ __launch_bounds__(256) void k_coop(uint32_t* __restrict__ io, int perms) {
    // io: per wave two states x 25 words x (lo, hi); lane i of a half-wave owns word i
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const CoopIdx c = coop_idx(lane);
    uint32_t* st = io + (wave * 2 + (lane >> 5)) * 50;
    uint32_t lo = c.active ? st[2 * (lane & 31)] : 0, hi = c.active ? st[2 * (lane & 31) + 1] : 0;
    for (int p = 0; p < perms; p++)
#pragma unroll 1
        for (int r = 0; r < 24; r++) coop_round(lo, hi, c, r);
    if (c.active) { st[2 * (lane & 31)] = lo; st[2 * (lane & 31) + 1] = hi; }
}

__global__ __launch_bounds__(256) void k_lane(uint32_t* __restrict__ io, int perms) {
    // io: per lane 25 x (lo, hi)
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    mldsa::KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = io[t * 50 + 2 * i]; s.hi[i] = io[t * 50 + 2 * i + 1]; }
    for (int p = 0; p < perms; p++) mldsa::keccak_f1600(s);
#pragma unroll
    for (int i = 0; i < 25; i++) { io[t * 50 + 2 * i] = s.lo[i]; io[t * 50 + 2 * i + 1] = s.hi[i]; }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // ---- same permutation?
    {
        const int n_states = 2 * 4 * 8;  // 8 blocks of the cooperative kernel
        std::vector<uint32_t> h(n_states * 50), a(h.size()), b(h.size());
        srand(204);
        for (auto& v : h) v = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
        uint32_t *d1, *d2;
        CHECK(hipMalloc(&d1, 256 * 50 * 4 + h.size() * 4));
        CHECK(hipMalloc(&d2, h.size() * 4));
        CHECK(hipMemset(d1, 0, 256 * 50 * 4 + h.size() * 4));
        CHECK(hipMemcpy(d1, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_lane, dim3(1), dim3(256), 0, 0, d1, 3);  // threads 0 .. 63 hold the 64 states
        hipLaunchKernelGGL(k_coop, dim3(8), dim3(256), 0, 0, d2, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(a.data(), d1, h.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d2, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++)
            if (a[i] != b[i]) { fprintf(stderr, "MISMATCH at word %zu: lane-per-state %08x cooperative %08x\n", i, a[i], b[i]); return 1; }
        CHECK(hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_coop3, dim3(8), dim3(256), 0, 0, d2, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(b.data(), d2, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++)
            if (a[i] != b[i]) { fprintf(stderr, "MISMATCH at word %zu: lane-per-state %08x three-level cooperative %08x\n", i, a[i], b[i]); return 1; }
        CHECK(hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_coop4, dim3(8), dim3(256), 0, 0, d2, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(b.data(), d2, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++)
            if (a[i] != b[i]) { fprintf(stderr, "MISMATCH at word %zu: lane-per-state %08x DPP cooperative %08x\n", i, a[i], b[i]); return 1; }
        CHECK(hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_coop5, dim3(n_states / 4), dim3(256), 0, 0, d2, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(b.data(), d2, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++)
            if (a[i] != b[i]) { fprintf(stderr, "MISMATCH at word %zu: lane-per-state %08x interleaved cooperative %08x\n", i, a[i], b[i]); return 1; }
        printf("cooperative and lane-per-state Keccak-f[1600] agree on %d random states x 3 permutations\n", n_states);
        CHECK(hipFree(d1)); CHECK(hipFree(d2));
    }
    // ---- timing
    const int perms = 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device: %s, %d CUs\n", prop.name, cus);
    printf("%-18s %10s %12s %22s %20s\n", "form", "waves/SIMD", "ms", "us per permutation/wave", "G permutations/s");
    for (int form = 0; form < 5; form++)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;  // 4 waves per block = 1 per SIMD of a CU
            const size_t words = (size_t)blocks * 256 * 50;
            uint32_t* d;
            CHECK(hipMalloc(&d, words * 4));
            CHECK(hipMemset(d, 0x5A, words * 4));
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, d, perms);
                else if (form == 1) hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(256), 0, 0, d, perms);
                else if (form == 2) hipLaunchKernelGGL(k_coop3, dim3(blocks), dim3(256), 0, 0, d, perms);
                else if (form == 3) hipLaunchKernelGGL(k_coop4, dim3(blocks), dim3(256), 0, 0, d, perms);
                else hipLaunchKernelGGL(k_coop5, dim3(blocks), dim3(256), 0, 0, d, perms);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double states = (double)blocks * 4 * (form == 0 ? 64 : form == 4 ? 1 : 2);
            printf("%-18s %10d %12.3f %22.2f %20.3f\n", form == 0 ? "lane-per-state" : form == 1 ? "cooperative (2/wave)" : form == 2 ? "coop, 3 levels (2/wave)" : form == 3 ? "coop, DPP + 4 gathers" : "interleaved, 1/wave, 5 gathers", wps, best, best * 1e3 / perms,
                   states * perms / (best * 1e-3) / 1e9);
            CHECK(hipFree(d));
        }
    return 0;
}
