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
        printf("cooperative and lane-per-state Keccak-f[1600] agree on %d random states x 3 permutations\n", n_states);
        CHECK(hipFree(d1)); CHECK(hipFree(d2));
    }
    // ---- timing
    const int perms = 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device: %s, %d CUs\n", prop.name, cus);
    printf("%-18s %10s %12s %22s %20s\n", "form", "waves/SIMD", "ms", "us per permutation/wave", "G permutations/s");
    for (int form = 0; form < 3; form++)
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
                else hipLaunchKernelGGL(k_coop3, dim3(blocks), dim3(256), 0, 0, d, perms);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double states = (double)blocks * 4 * (form == 0 ? 64 : 2);
            printf("%-18s %10d %12.3f %22.2f %20.3f\n", form == 0 ? "lane-per-state" : form == 1 ? "cooperative (2/wave)" : "coop, 3 levels (2/wave)", wps, best, best * 1e3 / perms,
                   states * perms / (best * 1e-3) / 1e9);
            CHECK(hipFree(d));
        }
    return 0;
}
