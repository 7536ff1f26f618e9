// One wavefront = one polynomial: 256-point radix-2 NTT / inverse NTT over Z_q with the
// working polynomial in registers (4 coefficients per lane) and wave-level butterfly
// exchanges (v_permlane32_swap / v_permlane16_swap / DPP) instead of LDS round trips.
//
// Replaces src/ntt.rs:14-76 (ntt, FIPS 204 Alg 41, Cooley-Tukey, len 128 -> 1) and
// src/ntt.rs:85-161 (inv_ntt, Alg 42, Gentleman-Sande, len 1 -> 128, * 256^-1).
//
// Register layout.  A 256-point index j has 8 bits; 2 live in the register number
// ("reg bits") and 6 in the lane id.  A level that pairs j with j ^ (1 << s) needs bit s
// to be a reg bit, so before each cross-lane level one reg bit is exchanged with the lane
// bit that holds s (each lane trades half of its registers with its partner lane):
//
//   forward: load  r[k] = w[64 k + lane]   (4 coalesced 256-B dword loads per wave)
//            s = 7, 6 in-lane; s = 5..0 exchange with lane bits 5..0
//            end   r[k] = w_hat[4 lane + k] (one coalesced 1-KiB dwordx4 store per wave)
//   inverse: the mirror image (dwordx4 load, four dword stores).
//
// The forward output layout equals the inverse input layout, so NTT -> pointwise ->
// inverse NTT chains stay lane-local.  Twiddles: each lane needs 2 per cross-lane level;
// they are precomputed per lane on the host (tables.cpp simulates this exact exchange
// sequence) and stay in registers while a wave loops over many polynomials.
#pragma once
#include "field.h"

namespace mldsa {

// zeta * 2^32 mod q (forward) or -zeta * 2^32 mod q (inverse), helpers.rs:171-184
typedef int32_t Twiddle;

constexpr int FWD_TW = 12;  // levels s = 5..0, two butterflies per lane
constexpr int INV_TW = 14;  // levels s = 0..6, two butterflies per lane

// Uniform twiddles of the in-lane forward levels (ZETA_TABLE_MONT[1..3]) and of the last
// inverse level (-ZETA_TABLE_MONT[1]).
constexpr int32_t ZF1 = 25847, ZF2 = 5771523, ZF3 = 7861508;
constexpr int32_t ZI1 = -25847;
constexpr int64_t RINV_MOD_Q = 8265825;  // (2^32)^-1 mod q

// Twiddle providers.  FwdTw / InvTw keep the lane's twiddles in registers (standalone NTT
// kernels: a wave loops over many polynomials); LdsTw reads them from a block-shared LDS copy
// at the point of use (fused kernels, where registers are better spent on prefetched data).
struct FwdTw {
    Twiddle t[FWD_TW];
    __device__ __forceinline__ Twiddle get(int i) const { return t[i]; }
};
struct InvTw {
    Twiddle t[INV_TW];
    __device__ __forceinline__ Twiddle get(int i) const { return t[i]; }
};
struct LdsTw {
    const Twiddle* base;  // [n][64] in LDS
    int lane;
    __device__ __forceinline__ Twiddle get(int i) const { return base[i * 64 + lane]; }
};

__device__ __forceinline__ void load_fwd_tw(FwdTw& tw, const Twiddle* __restrict__ tab, int lane) {
#pragma unroll
    for (int i = 0; i < FWD_TW; i++) tw.t[i] = tab[i * 64 + lane];
}
__device__ __forceinline__ void load_inv_tw(InvTw& tw, const Twiddle* __restrict__ tab, int lane) {
#pragma unroll
    for (int i = 0; i < INV_TW; i++) tw.t[i] = tab[i * 64 + lane];
}

// ---- register exchange: the upper lane's x  <->  the lower lane's y  (lanes l, l ^ M) ----
template <int M>
__device__ __forceinline__ void swap_pair(int32_t& x, int32_t& y, int lane) {
    if constexpr (M == 32) {
        auto v = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        x = v[0];
        y = v[1];
    } else if constexpr (M == 16) {
        auto v = __builtin_amdgcn_permlane16_swap(x, y, false, false);
        x = v[0];
        y = v[1];
    } else if constexpr (M == 8) {
        // row_ror:8 pairs lane i with i ^ 8 inside a 16-lane row; bank_mask picks the writers
        int32_t ny = __builtin_amdgcn_update_dpp(y, x, 0x128, 0xF, 0x3, false);
        int32_t nx = __builtin_amdgcn_update_dpp(x, y, 0x128, 0xF, 0xC, false);
        x = nx;
        y = ny;
    } else if constexpr (M == 4) {
        // row_shl:4 -> lane i reads i + 4 (banks 0,2); row_shr:4 -> lane i reads i - 4 (banks 1,3)
        int32_t ny = __builtin_amdgcn_update_dpp(y, x, 0x104, 0xF, 0x5, false);
        int32_t nx = __builtin_amdgcn_update_dpp(x, y, 0x114, 0xF, 0xA, false);
        x = nx;
        y = ny;
    } else if constexpr (M == 2) {
        int32_t px = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
        int32_t py = __builtin_amdgcn_update_dpp(0, y, 0x4E, 0xF, 0xF, false);
        bool up = (lane & 2) != 0;
        x = up ? py : x;
        y = up ? y : px;
    } else {
        int32_t px = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
        int32_t py = __builtin_amdgcn_update_dpp(0, y, 0xB1, 0xF, 0xF, false);
        bool up = (lane & 1) != 0;
        x = up ? py : x;
        y = up ? y : px;
    }
}

// exchange reg bit 1 with lane bit M: afterwards the pairs are (r0,r2), (r1,r3)
template <int M>
__device__ __forceinline__ void xchg_hi(int32_t r[4], int lane) {
    swap_pair<M>(r[0], r[2], lane);
    swap_pair<M>(r[1], r[3], lane);
}
// exchange reg bit 0 with lane bit M: afterwards the pairs are (r0,r1), (r2,r3)
template <int M>
__device__ __forceinline__ void xchg_lo(int32_t r[4], int lane) {
    swap_pair<M>(r[0], r[1], lane);
    swap_pair<M>(r[2], r[3], lane);
}

// ntt.rs:47-54  t = zeta * b;  b = a - t;  a = a + t   (no reduction on add/sub)
__device__ __forceinline__ void bf_ct(int32_t& a, int32_t& b, int32_t z) {
    int32_t t = mont_mul(b, z);
    b = a - t;
    a = a + t;
}
// ntt.rs:122-132  a = t + b;  b = (-zeta) * (t - b)
__device__ __forceinline__ void bf_gs(int32_t& a, int32_t& b, int32_t z) {
    int32_t t = a;
    a = t + b;
    b = mont_mul(t - b, z);
}

// Forward NTT.  In: r[k] = w[64k + lane], |w| < q (callers reduce32 on load).
// Out: r[k] = w_hat[4 lane + k], |w_hat| < 9q, plain domain (ntt.rs:14-76).
template <class TW>
__device__ __forceinline__ void ntt_fwd_wave(int32_t r[4], const TW& tw, int lane) {
    bf_ct(r[0], r[2], ZF1);
    bf_ct(r[1], r[3], ZF1);
    bf_ct(r[0], r[1], ZF2);
    bf_ct(r[2], r[3], ZF3);
    xchg_hi<32>(r, lane);
    bf_ct(r[0], r[2], tw.get(0));
    bf_ct(r[1], r[3], tw.get(1));
    xchg_lo<16>(r, lane);
    bf_ct(r[0], r[1], tw.get(2));
    bf_ct(r[2], r[3], tw.get(3));
    xchg_hi<8>(r, lane);
    bf_ct(r[0], r[2], tw.get(4));
    bf_ct(r[1], r[3], tw.get(5));
    xchg_lo<4>(r, lane);
    bf_ct(r[0], r[1], tw.get(6));
    bf_ct(r[2], r[3], tw.get(7));
    xchg_hi<2>(r, lane);
    bf_ct(r[0], r[2], tw.get(8));
    bf_ct(r[1], r[3], tw.get(9));
    xchg_lo<1>(r, lane);
    bf_ct(r[0], r[1], tw.get(10));
    bf_ct(r[2], r[3], tw.get(11));
}

// Inverse NTT.  In: r[k] = w_hat[4 lane + k], |w_hat| < q.  Out: r[k] = w[64k + lane],
// canonical [0, q) after the final scaling by `f` (F_MONT: plain -> plain as ntt.rs:152-154;
// F_MONT2: input carries a factor 2^-32).  |intermediates| < 256 q < 2^31.
template <class TW>
__device__ __forceinline__ void ntt_inv_wave(int32_t r[4], const TW& tw, int lane, int32_t f) {
    bf_gs(r[0], r[1], tw.get(0));
    bf_gs(r[2], r[3], tw.get(1));
    bf_gs(r[0], r[2], tw.get(2));
    bf_gs(r[1], r[3], tw.get(3));
    xchg_lo<1>(r, lane);
    bf_gs(r[0], r[1], tw.get(4));
    bf_gs(r[2], r[3], tw.get(5));
    xchg_hi<2>(r, lane);
    bf_gs(r[0], r[2], tw.get(6));
    bf_gs(r[1], r[3], tw.get(7));
    xchg_lo<4>(r, lane);
    bf_gs(r[0], r[1], tw.get(8));
    bf_gs(r[2], r[3], tw.get(9));
    xchg_hi<8>(r, lane);
    bf_gs(r[0], r[2], tw.get(10));
    bf_gs(r[1], r[3], tw.get(11));
    xchg_lo<16>(r, lane);
    bf_gs(r[0], r[1], tw.get(12));
    bf_gs(r[2], r[3], tw.get(13));
    xchg_hi<32>(r, lane);
    // last level with the 1/256 scaling folded in: a' = (a + b) f, b' = (a - b) (zeta f); f is a
    // constant at every call site, so zf folds at compile time.  mont_mul lands in (-q, q): only the
    // conditional +q of full_reduce32 is left.
    const int32_t zf = (int32_t)((((int64_t)ZI1 * f) % Q) * RINV_MOD_Q % Q);  // ZI1 * f * 2^-32 mod q
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int32_t a = r[k], b = r[k + 2];
        r[k] = caddq(mont_mul(a + b, f));
        r[k + 2] = caddq(mont_mul(a - b, zf));
    }
}

// coalesced I/O in the two layouts
__device__ __forceinline__ void load_strided(int32_t r[4], const int32_t* __restrict__ p, int lane) {
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = p[64 * k + lane];
}
__device__ __forceinline__ void store_strided(const int32_t r[4], int32_t* __restrict__ p, int lane) {
#pragma unroll
    for (int k = 0; k < 4; k++) p[64 * k + lane] = r[k];
}
__device__ __forceinline__ void load_packed(int32_t r[4], const int32_t* __restrict__ p, int lane) {
    int4 v = reinterpret_cast<const int4*>(p)[lane];
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
}
__device__ __forceinline__ void store_packed(const int32_t r[4], int32_t* __restrict__ p, int lane) {
    reinterpret_cast<int4*>(p)[lane] = make_int4(r[0], r[1], r[2], r[3]);
}

}  // namespace mldsa
