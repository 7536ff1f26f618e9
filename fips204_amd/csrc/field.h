// Z_q arithmetic in 32-bit integer lanes (q = 8380417) for gfx950.
//
// Replaces the scalar helpers of the reference: mont_reduce (src/helpers.rs:156-165),
// partial_reduce32 (61-67), full_reduce32 (70-76), center_mod (88-95), to_mont (131-135).
// Only results mod q are observable (SURVEY.md appendix "Montgomery bookkeeping").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mldsa {

constexpr int32_t Q = 8380417;            // lib.rs:109
constexpr uint32_t QINV = 58728449u;      // q * QINV = 1 mod 2^32 (helpers.rs:157)
constexpr int32_t R_MOD_Q = 4193792;      // 2^32 mod q
constexpr int32_t R2_MOD_Q = 2365951;     // 2^64 mod q
constexpr int32_t F_MONT = 16382;         // 256^-1 * 2^32 mod q (ntt.rs:88)
constexpr int32_t F_MONT2 = 41978;        // 256^-1 * 2^64 mod q (inverse NTT of R^-1-scaled input)
constexpr int N = 256;

// The pipelines' private 24-bit form of a polynomial with canonical coefficients (A_hat from ExpandA, the signer's w): 768 bytes,
// four coefficients in three dwords.
struct Packed3 { uint32_t a, b, c; };
constexpr int PACKED_POLY_DWORDS = 192;  // 256 * 24 bits
__device__ __forceinline__ int4 unpack24(const Packed3& p) {
    return make_int4((int)(p.a & 0xFFFFFFu), (int)(__builtin_amdgcn_alignbit(p.b, p.a, 24) & 0xFFFFFFu),
                     (int)(__builtin_amdgcn_alignbit(p.c, p.b, 16) & 0xFFFFFFu), (int)(p.c >> 8));
}
// The signer's y between ExpandMask and its users: the squeezed bytes themselves, BitPack(y, gamma1 - 1, gamma1) with c = 18 / 20 bit
// fields (conversion.rs:227-262), 32 c bytes per polynomial.  Field i starts at bit c i: ONE dword load at its byte (no alignment
// needed on gfx950) holds it, shifted by (c i) & 7 -- which for i = 64 k + lane depends on the lane only.
typedef uint32_t __attribute__((aligned(1))) u32_any;
typedef uint64_t __attribute__((aligned(1))) u64_any;
typedef uint16_t __attribute__((aligned(1))) u16_any;
template <int CB, bool NT = false>
__device__ __forceinline__ uint32_t y_raw_dword(const uint8_t* poly, int k, int lane) {
    const u32_any* p = reinterpret_cast<const u32_any*>(poly + 8 * CB * k + ((lane * CB) >> 3));
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <int CB>
__device__ __forceinline__ int32_t y_from_raw(uint32_t raw, int lane) {
    return (1 << (CB - 1)) - (int32_t)((raw >> ((lane * CB) & 7)) & ((1u << CB) - 1u));
}
__device__ __forceinline__ Packed3 pack24(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {  // all < 2^24
    return Packed3{c0 | (c1 << 24), (c1 >> 8) | (c2 << 16), (c2 >> 16) | (c3 << 8)};
}

// ---- Cache policy of the streaming kernels (round 6; EXPERIMENTS.md "Round 6", profiles/r06_ab_memory_path_*.txt).
// Rows that a launch reads exactly ONCE -- A_hat, z / c and the signature bytes in k_verify_main and the config-2 kernel -- are loaded
// nontemporal (`global_load ... nt`): k_verify_main -8 %, the config-2 kernel 0.582 -> 0.635 of the HBM peak, same-box A/B.  The signer's
// rows are NOT: adjacent candidate rows share their A_hat through the XCD's L2 (nt there: sign_w +12 %), and y / w are read again by the
// tail kernels.  MLDSA_EXP (default 0 = the shipped library; `make variants` + tools/ab_variants.py for the A/Bs) switches the measured
// variants: the adopted policies OFF (bits 1, 8 and 9), and the rejected ones ON.
#ifndef MLDSA_EXP
#define MLDSA_EXP 0
#endif
constexpr bool NT_A_VERIFY = (MLDSA_EXP & 2) == 0;      // adopted: A_hat rows of k_verify_main / the config-2 kernel by nontemporal loads (bit 1: off)
constexpr bool NT_ZC = (MLDSA_EXP & 256) == 0;          // adopted: their read-once z, c and signature bytes too (bit 8: off)
constexpr bool NT_A_KG = (MLDSA_EXP & 512) == 0;        // adopted: key generation's A_hat rows (read once per key) as well: +1.0 ... 1.7 % keys/s for the three sets (bit 9: off)
constexpr bool EXP_NT_A_SIGN = (MLDSA_EXP & 1) != 0;    // rejected (sign_w +12 %): the signer's A_hat rows by nontemporal loads
constexpr bool EXP_NT_STORE = (MLDSA_EXP & 4) != 0;     // rejected (+-0.5 %, config 2 -1.5 %): every w / w1 row by nontemporal stores
constexpr bool EXP_LDSDMA = (MLDSA_EXP & 8) != 0;       // rejected (sign_w +3 %, 5 -> 3 waves per SIMD): the signer's A_hat rows by LDS-DMA
constexpr int EXP_PRIO = (MLDSA_EXP & 16) ? 1 : (MLDSA_EXP & 32) ? 3 : 0;  // rejected (0 / -1.6 %): s_setprio of the c~ hash / SampleInBall / NTT(c) waves
constexpr bool EXP_NT_DMA = (MLDSA_EXP & 64) != 0;      // with EXP_LDSDMA: the DMA with the nt policy (aux = 2; sign_w +5 %)
constexpr bool EXP_NT_STORE_W1 = (MLDSA_EXP & (4 | 128)) != 0;  // rejected (+-0.3 %): bit 7 = the packed w / w1 rows only (not config 2's int32 w')
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ Packed3 load_row(const Packed3* p) {
    if constexpr (NT) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);  // (three dword loads: the compiler merges them into one dwordx3 ... nt)
        return Packed3{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2)};
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ int4 load_row(const int4* p) {
    if constexpr (NT) {
        const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
        return make_int4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT = EXP_NT_STORE_W1, class T>
__device__ __forceinline__ void store_row(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT, class T>
__device__ __forceinline__ T load_once(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// a * b * 2^-32 mod q, result in (-q, q); |a*b| < 2^31 * q.  Same computation as the reference's
// mont_reduce (helpers.rs:156-165); hipcc lowers it to v_mad_i64_i32, v_mul_lo_u32,
// v_mad_i64_i32 -- three full-rate instructions on gfx950 (profiles/r01_ubench_valu.txt), with
// no precomputed zeta * qinv companion to keep in registers.
__device__ __forceinline__ int32_t mont_mul(int32_t a, int32_t b) {
    const int64_t p = (int64_t)a * b;
    const int32_t t = (int32_t)((uint32_t)p * QINV);
    return (int32_t)(((int64_t)t * (-Q) + p) >> 32);
}

// mont_reduce (helpers.rs:156-165) of a 64-bit sum of products: p * 2^-32 mod q in (-q, q) for |p| < 2^31 * q.  The fused
// arithmetic kernels accumulate a row's K * L products in 64 bits (one v_mad_i64_i32 per term) and reduce ONCE per
// coefficient instead of once per term: sum_j mont(a_j z_j) and mont(sum_j a_j z_j) agree mod q.
__device__ __forceinline__ int32_t mont_reduce64(int64_t p) {
    const int32_t t = (int32_t)((uint32_t)p * QINV);
    return (int32_t)(((int64_t)t * (-Q) + p) >> 32);
}

// helpers.rs:61-67: |a| < 2^31 - 2^22  ->  (-q, q)
__device__ __forceinline__ int32_t reduce32(int32_t a) {
    int32_t x = (a + (1 << 22)) >> 23;
    return a - x * Q;
}

// (-q, q) -> [0, q)
__device__ __forceinline__ int32_t caddq(int32_t x) { return x + ((x >> 31) & Q); }

// helpers.rs:70-76
__device__ __forceinline__ int32_t freeze(int32_t a) { return caddq(reduce32(a)); }

// helpers.rs:88-95: canonical representative in (-q/2, q/2]
__device__ __forceinline__ int32_t center(int32_t a) {
    int32_t t = freeze(a);
    return t - ((((Q / 2) - t) >> 31) & Q);
}

// x * 2^32 mod q (reference to_mont, helpers.rs:131-135), result in (-q, q)
__device__ __forceinline__ int32_t to_mont(int32_t x) { return mont_mul(x, R2_MOD_Q); }

// A kernel argument read from the kernarg segment where it is USED (an s_load that hits the scalar cache) instead of being held in
// scalar registers from the kernel's entry: the empty asm makes the segment pointer opaque at that point, so the load can neither be
// hoisted nor kept live across what lies in between.  byte_offset = offsetof(first by-value argument struct, field): relies on the
// code-object ABI putting that struct at byte 0 of the segment -- checked by a self-test at context creation (k_late_arg_selftest).
// A toolchain that lays the segment out differently fails that self-test; the way out is a build with -DMLDSA_NO_LATE_ARG (`make
// nolatearg`): every kernel that uses late arguments then copies its argument struct into LDS at entry (late_args_begin, through the
// compiler's own addressing of the by-value argument) and late_arg / reload_first_kernarg read that copy -- slower (the values arrive in
// vector registers), same results (tests/test_gpu_small_calls.py runs the suite's signing cases on that build).
#if defined(MLDSA_NO_LATE_ARG)
constexpr unsigned LATE_ARGS_MAX_DWORDS = 192;
static __shared__ uint32_t late_args_copy[LATE_ARGS_MAX_DWORDS];
template <class A>
__device__ __forceinline__ void late_args_begin(const A& a) {  // at the top of the kernel, before any return: the whole workgroup passes here
    static_assert(sizeof(A) % 4 == 0 && sizeof(A) <= 4 * LATE_ARGS_MAX_DWORDS, "argument struct fits the LDS copy");
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&a);
    for (unsigned i = threadIdx.x; i < sizeof(A) / 4; i += blockDim.x) late_args_copy[i] = w[i];
    __syncthreads();
}
template <class T>
__device__ __forceinline__ T late_arg(unsigned byte_offset) {
    static_assert(sizeof(T) % 4 == 0 || sizeof(T) < 4, "dword-sized fields (or smaller, inside one dword)");
    T v;
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = late_args_copy[byte_offset / 4], hi = late_args_copy[byte_offset / 4 + 1];
        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        __builtin_memcpy(&v, &u, 8);
    } else {
        const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)late_args_copy[byte_offset / 4]) >> (8 * (byte_offset & 3));
        __builtin_memcpy(&v, &u, sizeof(T));
    }
    return v;
}
template <class T>
__device__ __forceinline__ void reload_first_kernarg(T& out, const T& arg) { out = arg; }
// a whole sub-struct of the first argument, by reference (read where the reference is used)
template <class T>
__device__ __forceinline__ const T& late_ref(unsigned byte_offset) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(late_args_copy) + byte_offset);
}
#else
template <class A>
__device__ __forceinline__ void late_args_begin(const A&) {}
template <class T>
__device__ __forceinline__ T late_arg(unsigned byte_offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(const T __attribute__((address_space(4)))*)(ka + byte_offset);
}
// a whole sub-struct of the first argument, by reference: its fields are loaded from the kernarg segment where the reference is used,
// not at the kernel's entry
template <class T>
__device__ __forceinline__ const T& late_ref(unsigned byte_offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(const T*)(ka + byte_offset);
}
// the kernel's first by-value argument, read again from the kernarg segment (same ABI assumption, same self-test); `arg` = that argument
template <class T>
__device__ __forceinline__ void reload_first_kernarg(T& out, const T& arg) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    (void)arg;
    typedef const uint32_t __attribute__((address_space(4))) * kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    uint32_t* w = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = ka[i];
}
#endif

}  // namespace mldsa
