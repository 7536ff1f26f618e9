// Signing-side and key-side kernels: key expansion (try_from_bytes), the per-round pieces
// of the rejection loop of sign_internal (src/ml_dsa.rs:212-330), sig_encode, and the
// keygen tail (power2round, pk / sk encode).  Everything is integer, element-wise or one
// wave per polynomial; the NTT work goes through ntt_wave.h.
#include <algorithm>
#include <cstdlib>
#include "ctx.h"
#include "sign_slots_dev.h"
#include "keccak.h"
#include "rounding.h"
#include "sampler_dev.h"

namespace mldsa {

constexpr int GWAVES = 4;
constexpr int GBLOCK = 64 * GWAVES;

// `bits`-wide little-endian field starting at bit offset `bo`; never reads past the field
__device__ __forceinline__ uint32_t load_bits(const uint8_t* p, int bo, int bits) {
    const int sh = bo & 7, need = (sh + bits + 7) >> 3;
    const uint8_t* q = p + (bo >> 3);
    uint32_t v = q[0];
    if (need > 1) v |= (uint32_t)q[1] << 8;
    if (need > 2) v |= (uint32_t)q[2] << 16;
    if (need > 3) v |= (uint32_t)q[3] << 24;
    return (v >> sh) & ((1u << bits) - 1u);
}

// ------------------------------------------------------------------------------------
// bit_unpack / simple_bit_unpack (conversion.rs:198-262) -> ntt -> * scale, one wave per
// polynomial: the deserialisation half of expand_public / expand_private
// (ml_dsa.rs:445-498).  value = b - field (or the field itself when b < 0).
__global__ __launch_bounds__(GBLOCK) void k_unpack_ntt(const uint8_t* __restrict__ src, size_t key_stride, size_t poly_off,
                                                       int bits, int b, int32_t scale,
                                                       int32_t* __restrict__ out, int polys_per_key, size_t n_keys,
                                                       const Twiddle* __restrict__ fwd_tab) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * GWAVES + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * GWAVES;
    const size_t n_polys = n_keys * (size_t)polys_per_key;
    FwdTw tw;
    load_fwd_tw(tw, fwd_tab, lane);
    for (size_t p = wave; p < n_polys; p += n_waves) {
        const size_t key = p / polys_per_key;
        const int j = (int)(p % polys_per_key);
        const uint8_t* bytes = src + key * key_stride + poly_off + (size_t)j * (32 * bits);
        int32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t v = (int32_t)load_bits(bytes, (64 * k + lane) * bits, bits);
            r[k] = b < 0 ? v : b - v;
        }
        ntt_fwd_wave(r, tw, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = mont_mul(r[k], scale);
        store_packed(r, out + p * N, lane);
    }
}

struct Coef4 { int32_t v[4]; };
__device__ __forceinline__ Coef4 ld4(const int32_t* p, int u) {
    const int4 t = reinterpret_cast<const int4*>(p)[u];
    return Coef4{{t.x, t.y, t.z, t.w}};
}

// ------------------------------------------------------------------------------------
// Fused tail of one rejection-loop iteration (ml_dsa.rs:243-336): ONE WAVE PER CANDIDATE, looping over
// its polynomials, so there are no block barriers and no cross-wave reductions:
//   z_j  = y_j + inv_ntt(c_hat o s1_hat_j)            ||z||  <  gamma1 - beta ?
//   r_i  = w_i - inv_ntt(c_hat o s2_hat_i)            ||LowBits(r)|| < gamma2 - beta ?
//   -- only if both hold (about 1 candidate in 5):
//   ct0_i = inv_ntt(c_hat o t0_hat_i)                 ||ct0|| < gamma2 ?
//   h_i  = MakeHint(-ct0_i, r_i + ct0_i)              weight(h) <= omega ?
//   (ML-DSA-65 / 87: tau * 2^12 < gamma2, the ct0 test cannot fail, and h_i comes from ONE transform per row:
//    r_i + ct0_i = w_i + inv_ntt(c_hat o (t0_hat_i - s2_hat_i)) against HighBits(w_i) -- 13 / 17 instead of 17 / 23
//    inverse transforms per accepted attempt, and no r_i kept in LDS)
// (c_hat = ntt(c) comes from k_ntt.)  sigEncode (encodings.rs:238-276) is written EAGERLY while the
// polynomials stream through: z bytes in stage 1, hint bytes in stage 2.  If the attempt is then
// rejected the bytes are garbage, but the op's next attempt rewrites every byte, and only an
// accepted attempt sets done[] -- so the signature buffer of a finished op always holds the accepted
// attempt.  The reference's two `continue` stages (ml_dsa.rs:280, 312) stay two stages; stage 1 first
// transforms only the polynomials that CAN reject (flags from sign_w / ExpandMask, see below), the more
// selective LowBits test first, and leaves at the first rejection: a rejected ML-DSA-65 attempt costs
// about 1.5 of the 11 inverse NTTs of stage 1.
//
// tail_attempt is that iteration for one candidate, in two forms:
//   FULL  = the whole iteration with the signature bytes written to `sig` (k_sign_tail in rounds with one candidate
//           per op, k_resolve for the winner of a speculative round);
//   !FULL = the tests of the polynomials that can reject, nothing written (k_sign_tail in speculative rounds: most
//           candidates are rejected, and of the survivors only the FIRST one of each op is ever needed, so bytes,
//           the remaining z_j and the hint stage are left to k_resolve, which builds exactly one signature per op).
struct TailPtrs {
    const int32_t *c_hat, *y, *w;
    const uint8_t* ctilde;
    const int32_t *s1, *s2, *t0;
};

// The parameter set's scalars as compile-time constants (src/lib.rs:639-656, 681-698, 723-740): (K, L) names the set, so the tail
// kernels take none of them as arguments -- shifts and bounds become immediates, and the kernels fit their scalar registers
// (with gamma1's bit count, beta, omega, the c~ length and the signature length as kernel arguments k_sign_tail / k_resolve
// spilled 32-63 SGPRs into VGPR lanes).
template <int K, int L> struct TailConst;
template <> struct TailConst<4, 4> { static constexpr int GB = 17, BETA = 78, OMEGA = 80, CTILDE = 32; static constexpr size_t SIG_LEN = 2420; };
template <> struct TailConst<6, 5> { static constexpr int GB = 19, BETA = 196, OMEGA = 55, CTILDE = 48; static constexpr size_t SIG_LEN = 3309; };
template <> struct TailConst<8, 7> { static constexpr int GB = 19, BETA = 120, OMEGA = 75, CTILDE = 64; static constexpr size_t SIG_LEN = 4627; };

// Kernel arguments that are needed once per slot (the prefetch of the next slot, the verdict at the end) are NOT held in scalar
// registers across the slot: the kernels take one struct by value -- it sits at offset 0 of the kernarg segment -- and read
// those fields from the segment where they are used (an s_load that hits the scalar cache).  The empty asm makes the segment
// pointer opaque at that point, so the load can neither be hoisted to the kernel's entry nor kept live through the inverse
// transforms; with all ~20 pointers live the two kernels needed 32-63 more scalar registers than the 102 there are.
#define LATE(ARGS, field) late_arg<decltype(ARGS::field)>((unsigned)offsetof(ARGS, field))

struct SignTailArgs {
    TailPtrs a;          // used throughout a slot
    uint8_t* sigs;
    int ct0_exact, oor_by_op;
    const RoundCtl* ctl;  // prologue
    const Twiddle* inv_tab;
    // read where they are used (LATE):
    const uint32_t *slot_op, *slot_y, *key_idx;
    const uint8_t *wrisk, *yrisk, *key_oor;
    uint16_t* kappa;
    int32_t *done, *accept;
};

template <int K, int L, bool G2HI, bool FULL>
__device__ __forceinline__ bool tail_attempt(const TailPtrs& a, size_t slot, size_t yrow, size_t key, uint32_t r_risky, uint32_t z_risky,
                                             bool s2_oor, uint8_t* sig, int32_t* xp, const LdsTw& itw, int lane) {
    constexpr int gb = TailConst<K, L>::GB, beta = TailConst<K, L>::BETA, omega = TailConst<K, L>::OMEGA, ctilde_len = TailConst<K, L>::CTILDE;
    constexpr int32_t GAMMA2 = G2HI ? (Q - 1) / 32 : (Q - 1) / 88;
    // ||c t0||inf <= tau * 2^12: below gamma2 = (q-1)/32 for ML-DSA-65 / 87 (200 704, 245 760 < 261 888), not for ML-DSA-44
    constexpr bool CT0_CAN_FAIL = !G2HI;
    constexpr int32_t gamma1 = 1 << gb;
    constexpr int cb = gb + 1;
    constexpr int YCB = G2HI ? 20 : 18;  // = cb: gamma2 = (q - 1) / 32 goes with gamma1 = 2^19, (q - 1) / 88 with 2^17
    // everything a candidate owns -- y, w, c_hat, c~ and the risk flags -- lives in its ROW (k_make_slots)
    (void)slot;
    // (c_hat and c~ are read once, here: their pointers come from the kernarg segment -- TailPtrs opens both argument structs --
    //  and are not held in scalar registers through the transforms)
    const int4 cv = reinterpret_cast<const int4*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, c_hat)) + yrow * N)[lane];
    if (FULL && lane < ctilde_len) sig[lane] = late_arg<const uint8_t*>((unsigned)offsetof(TailPtrs, ctilde))[yrow * 64 + lane];
    // ---- stage 1.  ||c s1||inf and ||c s2||inf are at most beta = tau * eta, so a polynomial of w whose
    // LowBits all stay below gamma2 - 2 beta cannot fail ||LowBits(w - c s2)||inf < gamma2 - beta, and a
    // polynomial of y below gamma1 - 2 beta cannot fail ||y + c s1||inf < gamma1 - beta (ml_dsa.rs:280; with
    // |low| < gamma2 the decomposition of w - c s2 keeps the high part of w).  sign_w and ExpandMask flag the
    // few polynomials that are NOT below those margins (about 1 in 3 of w, 1 in 6 of y for ML-DSA-65):
    // pass 0 transforms only those -- the more selective LowBits test first -- and leaves at the first
    // rejection; pass 1 (FULL) computes the remaining z_j for the attempts that survived (1 in 5), which
    // need them for the signature bytes.
    bool ok = true;  // wave-uniform
    // work list of a pass: bit i < K = r_i (s2 / w), bit K + j = z_j (s1 / y), walked from the low end with the
    // next polynomial's loads issued before the current inverse NTT
    uint32_t todo = (1u << (K + L)) - 1u;
    const uint32_t risky = (r_risky & ((1u << K) - 1u)) | (z_risky << K);
    auto issue_loads = [&](int idx, int32_t(&v)[4], int32_t(&x)[4]) {
        const bool is_r = idx < K;
        // (the key and row tables' pointers come from the kernarg segment at this point, like c_hat above)
        const int32_t* sp = is_r ? late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, s2)) + (key * K + idx) * (size_t)N
                                 : late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, s1)) + (key * L + (idx - K)) * (size_t)N;
        // One load shape for both: w_i (sign_w's 24-bit form) is three planes of 64 dwords, y_j (ExpandMask's squeezed bytes, field.h)
        // four dwords at byte granularity; both are decoded where they are used.  (The fourth dword of a w_i is the next
        // polynomial's first plane -- the carve after w follows the last one -- and is not used.)
        const uint8_t* xb = is_r ? reinterpret_cast<const uint8_t*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, w))) + (yrow * K + idx) * (size_t)(PACKED_POLY_DWORDS * 4)
                                 : reinterpret_cast<const uint8_t*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, y))) + (yrow * L + (idx - K)) * (size_t)(32 * YCB);
        const int kstride = is_r ? 256 : 8 * YCB, lane_off = is_r ? 4 * lane : (lane * YCB) >> 3;
        load_packed(v, sp, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = (int32_t) * reinterpret_cast<const u32_any*>(xb + k * kstride + lane_off);
    };
#pragma unroll 1
    for (int pass = 0; pass < (FULL ? 2 : 1) && ok; pass++) {
        // pass 1 needs only the remaining z_j (signature bytes): stage 2 works from w itself
        uint32_t work = pass == 0 ? (risky & todo) : (todo & ~((1u << K) - 1u));
        todo &= ~work;
        int cur = work ? __ffs((int)work) - 1 : -1;
        int32_t nv[4] = {0, 0, 0, 0}, nx[4] = {0, 0, 0, 0};
        if (cur >= 0) issue_loads(cur, nv, nx);
#pragma unroll 1
        while (cur >= 0) {
            work &= work - 1u;
            const int nxt = work ? __ffs((int)work) - 1 : -1;
            const int32_t v[4] = {nv[0], nv[1], nv[2], nv[3]};
            int32_t x[4] = {nx[0], nx[1], nx[2], nx[3]};
            if (cur < K) {  // w_i arrives as three dwords of 24-bit fields
                const int4 w4 = unpack24(Packed3{(uint32_t)nx[0], (uint32_t)nx[1], (uint32_t)nx[2]});
                x[0] = w4.x; x[1] = w4.y; x[2] = w4.z; x[3] = w4.w;
            }
            if (nxt >= 0) issue_loads(nxt, nv, nx);
            int32_t r[4];
            r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
            ntt_inv_wave(r, itw, lane, F_MONT);
            bool bad = false;
            if (cur < K) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t rr = caddq(x[k] - r[k]);     // w - cs2, canonical (both operands are in [0, q))
                    int32_t r1, r0;
                    decompose<G2HI>(rr, r1, r0);
                    bad |= (r0 < 0 ? -r0 : r0) >= GAMMA2 - beta;
                }
                if (__ballot(bad) != 0ull) { ok = false; break; }
            } else {
                const int j = cur - K;
                int32_t zc4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // z mod+- q (ml_dsa.rs:264, 334): y in (-gamma1, gamma1], cs1 in [0, q), so one conditional
                    // subtraction lands in (-q/2, q/2]
                    const int32_t zs = y_from_raw<YCB>((uint32_t)x[k], lane) + r[k];
                    const int32_t zc = zs - ((((Q / 2) - zs) >> 31) & Q);
                    zc4[k] = zc;
                    bad |= (zc < 0 ? -zc : zc) >= gamma1 - beta;
                }
                if (__ballot(bad) != 0ull) { ok = false; break; }
                if constexpr (FULL) {
#pragma unroll
                    for (int k = 0; k < 4; k++) xp[64 * k + lane] = zc4[k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int4 z4 = reinterpret_cast<const int4*>(xp)[lane];
                    const int32_t zz[4] = {z4.x, z4.y, z4.z, z4.w};
                    uint64_t lo = 0;
                    uint32_t hi = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) {  // BitPack(z, gamma1 - 1, gamma1): field = gamma1 - z
                        const uint64_t f = (uint64_t)(uint32_t)(gamma1 - zz[t]);
                        const int sh = t * cb;
                        lo |= f << sh;
                        if (sh + cb > 64) hi |= (uint32_t)(f >> (64 - sh));
                    }
                    // the lane's 4 c bits = 9 or 10 bytes: one 8-byte store at whatever alignment the signature row has (SIG_LEN is
                    // odd) and one 1- or 2-byte store, instead of 9 / 10 byte stores
                    constexpr int NBYTES = YCB / 2;
                    uint8_t* dst = sig + ctilde_len + (size_t)j * (32 * YCB) + (size_t)lane * NBYTES;
                    *reinterpret_cast<u64_any*>(dst) = lo;
                    if constexpr (NBYTES == 10) *reinterpret_cast<u16_any*>(dst + 8) = (uint16_t)hi;
                    else dst[8] = (uint8_t)hi;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            cur = nxt;
        }
    }
    if constexpr (!FULL) return ok;
    // ---- stage 2: ct0, hints (HintBitPack, conversion.rs:277-328, written as they are found)
    if (ok) {
        uint8_t* hy = sig + ctilde_len + (size_t)L * (32 * cb);
        for (int i = lane; i < omega + K; i += 64) hy[i] = 0;
        int32_t dmax = 0;  // ML-DSA-44: largest |centred coefficient| of the rows' transform outputs
        int index = 0;     // running count of hints, wave-uniform
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            // The attempt passed the LowBits test, hence HighBits(w - cs2) = HighBits(w) (what makes verification recover
            // w1), and  h = [HighBits(w - cs2 + ct0) != HighBits(w)],  w - cs2 + ct0 = w + invNTT(c_hat o (t0_hat - s2_hat)):
            // ONE inverse transform per row instead of the reference's two (cs2 and ct0).
            int32_t v[4], v2[4], r[4], base[4];
            load_packed(v, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, t0)) + (key * K + i) * (size_t)N, lane);  // (t0: stage 2 only)
            load_packed(v2, a.s2 + (key * K + i) * (size_t)N, lane);
            // w_i stays in its three packed dwords across the transform
            const uint32_t* wq = reinterpret_cast<const uint32_t*>(a.w) + (yrow * K + i) * (size_t)PACKED_POLY_DWORDS;
            const Packed3 wp{wq[lane], wq[64 + lane], wq[128 + lane]};
            if (s2_oor) {
                // out-of-range s2: the identity is not guaranteed; r_i = w_i - c s2_i explicitly (it passed the LowBits
                // test in stage 1, where every polynomial counted as risky), then r_i + c t0_i as the reference does
                r[0] = mont_mul(cv.x, v2[0]); r[1] = mont_mul(cv.y, v2[1]); r[2] = mont_mul(cv.z, v2[2]); r[3] = mont_mul(cv.w, v2[3]);
                ntt_inv_wave(r, itw, lane, F_MONT);
                const int4 w4 = unpack24(wp);
                base[0] = caddq(w4.x - r[0]); base[1] = caddq(w4.y - r[1]); base[2] = caddq(w4.z - r[2]); base[3] = caddq(w4.w - r[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] -= v2[k];
            }
            r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
            ntt_inv_wave(r, itw, lane, F_MONT);  // ct0 - cs2 (ct0 for an out-of-range key), canonical
            if (!s2_oor) {
                const int4 w4 = unpack24(wp);
                base[0] = w4.x; base[1] = w4.y; base[2] = w4.z; base[3] = w4.w;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if constexpr (CT0_CAN_FAIL) {
                    int32_t tc = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);  // center_mod of a canonical value
                    tc = tc < 0 ? -tc : tc;
                    dmax = tc > dmax ? tc : dmax;
                }
                // make_hint(Q - ct0, partial_reduce32(w - cs2 + ct0)), ml_dsa.rs:298-306
                int32_t a1, a0, b1, b0;
                const int32_t sum = base[k] + r[k] - Q;  // both in [0, q)
                decompose<G2HI>(caddq(sum), a1, a0);
                decompose<G2HI>(base[k], b1, b0);  // (w - cs2 + ct0) + (Q - ct0) = w - cs2 (mod q); same high part as w
                const bool h = a1 != b1;
                const unsigned long long mask = __ballot(h);
                if (h) {
                    const int rank = index + __popcll(mask & ((1ull << lane) - 1ull));
                    if (rank < omega) hy[rank] = (uint8_t)(64 * k + lane);
                }
                index += __popcll(mask);
            }
            if (lane == 0) hy[omega + i] = (uint8_t)(index < 255 ? index : 255);
        }
        ok = index <= omega;  // ml_dsa.rs:313-315
        if constexpr (CT0_CAN_FAIL) {
            // ||ct0||inf < gamma2 (ml_dsa.rs:312).  The rows gave d = ct0 - cs2 with ||cs2||inf <= beta, so max|d| + beta bounds
            // it; an out-of-range key gave ct0 itself.  Only if that bound cannot decide (|d| within beta of gamma2: ~1e-7
            // of the attempts) are the K transforms of ct0 proper spent.
            auto wave_max = [](int32_t x) {
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const int32_t o = __shfl_xor(x, m);
                    x = o > x ? o : x;
                }
                return x;
            };
            dmax = wave_max(dmax);
            bool ct0_ok = dmax + (s2_oor ? 0 : beta) < GAMMA2;
            if (!s2_oor && (!ct0_ok || LATE(SignTailArgs, ct0_exact))) {  // (the test knob MLDSA_OPT_SIGN_CT0_EXACT: read here, once per accepted attempt)
                int32_t tmax = 0;
#pragma unroll 1
                for (int i = 0; i < K; i++) {
                    int32_t v[4], r[4];
                    load_packed(v, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, t0)) + (key * K + i) * (size_t)N, lane);  // (t0: stage 2 only)
                    r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
                    ntt_inv_wave(r, itw, lane, F_MONT);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int32_t tc = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);
                        tc = tc < 0 ? -tc : tc;
                        tmax = tc > tmax ? tc : tmax;
                    }
                }
                ct0_ok = wave_max(tmax) < GAMMA2;
            }
            ok = ok && ct0_ok;
        }
    }
    return ok;
}

template <int K, int L, bool G2HI>
__global__ __launch_bounds__(64 * GWAVES) __attribute__((amdgpu_waves_per_eu(5))) void k_sign_tail(
    SignTailArgs args) {
    typedef SignTailArgs A;
    late_args_begin(args);  // (a no-op unless built with -DMLDSA_NO_LATE_ARG, field.h)
    constexpr size_t sig_len = TailConst<K, L>::SIG_LEN;
    __shared__ Twiddle tw_lds[INV_TW * 64];
    __shared__ int32_t xpose[GWAVES][N];     // strided -> 4 consecutive coefficients per lane (z packing)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: slot indices and row pointers stay scalar
    // slots and candidates per op of this round: written by k_make_slots (the host never sees them)
    const uint32_t n_slots32 = args.ctl->ns;
    const int spec = (int)args.ctl->spec;
    if (n_slots32 == 0) return;
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * GWAVES) tw_lds[i] = args.inv_tab[i];
    __syncthreads();
    const LdsTw itw{tw_lds, lane};
    const uint32_t wid = blockIdx.x * GWAVES + wave, n_waves = gridDim.x * GWAVES;

    // op / key of the next slot are fetched one slot ahead (two dependent loads off the critical path)
    // ... and so are its risk flags (which polynomials stage 1 has to transform)
    // (y, w and their flags are addressed by the slot's ROW: the slot itself, or where a round that generated two candidates
    //  per op put this one)
    // one word per slot: bits 0 .. K-1 = which w_i can reject (wrisk), bits 8 .. 8+L-1 = which y_j (yrisk), bit 16 = the key's s2 is
    // out of range (one scalar register instead of three, for the current slot and for the prefetched one)
    uint32_t op_next = 0, key_next = 0, flags_next = 0, yrow_next = 0;
    auto fetch_slot = [&](uint32_t sl) {
        op_next = LATE(A, slot_op)[sl];
        yrow_next = LATE(A, slot_y)[sl];
        const uint32_t* kidx = LATE(A, key_idx);
        key_next = kidx ? kidx[op_next] : op_next;
        uint32_t f = (uint32_t)LATE(A, wrisk)[yrow_next];
        const uint8_t* yr = LATE(A, yrisk) + (size_t)yrow_next * L;
#pragma unroll
        for (int j = 0; j < L; j++) f |= (yr[j] ? 1u : 0u) << (8 + j);
        flags_next = f | (LATE(A, key_oor)[LATE(A, oor_by_op) ? op_next : key_next] ? 1u << 16 : 0u);
    };
    if (wid < n_slots32) fetch_slot(wid);
    for (uint32_t slot32 = wid; slot32 < n_slots32; slot32 += n_waves) {
        const size_t slot = slot32, op = op_next, key = key_next, yrow = yrow_next;
        // a key whose s2 leaves [-eta, eta] (k_key_range) voids ||c s2||inf <= beta: every polynomial can reject and the
        // hint stage takes the reference's two-transform form
        const uint32_t flags = flags_next;
        const bool s2_oor = (flags >> 16) != 0;  // wave-uniform
        const uint32_t r_risky = s2_oor ? (1u << K) - 1u : (flags & 0xFFu), z_risky = (flags >> 8) & 0xFFu;
        if (slot32 + n_waves < n_slots32) fetch_slot(slot32 + n_waves);
        if (spec == 1) {
            const bool ok = tail_attempt<K, L, G2HI, true>(args.a, slot, yrow, key, r_risky, z_risky, s2_oor, LATE(A, sigs) + op * sig_len, xpose[wave], itw,
                                                           lane);
            if (lane == 0) {
                if (ok) LATE(A, done)[op] = 1;
                else { uint16_t* kp = LATE(A, kappa) + op; *kp = (uint16_t)(*kp + L); }  // ml_dsa.rs:281 / 316
            }
        } else {
            const bool ok = tail_attempt<K, L, G2HI, false>(args.a, slot, yrow, key, r_risky, z_risky, s2_oor, nullptr, xpose[wave], itw, lane);
            if (lane == 0) LATE(A, accept)[slot] = ok ? 1 : 0;
        }
    }
}

// Speculative rounds.  When few ops are still unfinished the GPU would idle through a long tail
// of tiny rounds (geometric, p ~ 0.2 per iteration).  Instead each unfinished op gets `spec`
// slots that try kappa, kappa + l, ..., kappa + (spec-1) l in the same round; k_resolve keeps
// the FIRST accepted candidate, which is exactly the signature the sequential loop of
// ml_dsa.rs:212-330 would have produced (all earlier candidates were rejected).
//
// The loop is driven from the device.  RoundCtl (ctx.h) lives in the workspace: cnt[p] = unfinished ops
// entering a round of parity p.  k_make_slots opens round r: it reads m = cnt[r & 1], derives the
// candidates per op from m with the same rule the host used to apply (1 while the active set is wide,
// then spec_target / m, at most spec_max), publishes {m, spec, ns = m * spec} for the round's other
// kernels and zeroes cnt[(r + 1) & 1], which k_compact of this round counts the survivors into.  Every
// round kernel walks its units with a grid-stride loop bounded by those device values, so the host can
// enqueue any number of rounds ahead of time (or replay them from a hipGraph) with grids sized from the
// EXPECTED counts: a round that finds more work loops, a round that finds none exits at once.
//
// Two candidates per op generated at once (rounds with one candidate per op, i.e. the first four of a large batch).  There
// sign_w is bound by re-reading each op's A_hat from HBM (23 KB per op and round against 12 KB of y / w per candidate), so a round
// may produce the rows of TWO candidates per op -- kappa and kappa + l, adjacent rows that share the A_hat read -- while testing only
// the first: hash, SampleInBall and the tail run on row 2 i of op i.  The next round then generates nothing: k_compact recorded the
// position each surviving op had (ypos), its slot s tests row 2 ypos[s] + 1 (use_pre) and its ExpandMask / sign_w launches find
// ns_gen = 0.  A fifth of the second candidates belong to ops that finished and are never read.  Decided here, on the device:
// the host only says what its plan allows (may_gen2 / may_use_pre); gen_par[] carries what the previous round really did.
__global__ __launch_bounds__(256) void k_make_slots(RoundCtl* __restrict__ ctl, int parity, SpecRule rule, uint32_t spec_max, uint32_t ns_cap,
                                                    const uint32_t* __restrict__ act, const uint16_t* __restrict__ kappa, int l,
                                                    uint32_t* __restrict__ slot_op, uint16_t* __restrict__ slot_kappa,
                                                    const uint32_t* __restrict__ key_idx, uint32_t* __restrict__ gen_op,
                                                    uint16_t* __restrict__ gen_kappa, uint32_t* __restrict__ gen_key, int may_use_pre,
                                                    int may_gen2, const uint32_t* __restrict__ ypos, uint32_t* __restrict__ slot_y) {
    const uint32_t m = ctl->cnt[parity];
    const uint32_t spec = m ? rule.spec(m, spec_max) : 1u;
    make_slots_body(ctl, parity, m, spec, ctl->gen_par[parity ^ 1] /* not written here */, ns_cap, act, kappa, l, slot_op, slot_kappa, key_idx, gen_op,
                    gen_kappa, gen_key, may_use_pre, may_gen2, ypos, slot_y, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, false);
}

// Speculative rounds, second half: one wave per unfinished op.  accept[] holds the verdict of stage 1 for each of the op's
// candidates; the wave takes the FIRST survivor -- the candidate the sequential loop of ml_dsa.rs:212-330 would have reached --
// and runs the whole iteration for it with the bytes going straight to the op's signature (tail_attempt<FULL>: all z_j, the
// hint stage).  Should the hint stage reject it (weight(h) > omega, about 1 %), the next survivor is tried, as the loop would.
struct ResolveArgs {
    TailPtrs a;
    uint8_t* sigs;
    int ct0_exact, oor_by_op;
    const RoundCtl* ctl;
    const Twiddle* inv_tab;
    // LATE:
    const uint32_t *act, *slot_y, *key_idx;
    const int32_t* accept;
    const uint8_t* key_oor;
    uint16_t* kappa;
    int32_t* done;
};

static_assert(offsetof(ResolveArgs, ct0_exact) == offsetof(SignTailArgs, ct0_exact) && offsetof(ResolveArgs, a) == 0 && offsetof(SignTailArgs, a) == 0,
              "tail_attempt reads ct0_exact, c_hat and c~ at the same kernarg offsets for both kernels");

template <int K, int L, bool G2HI>
__global__ __launch_bounds__(64 * GWAVES) void k_resolve(ResolveArgs args) {
    typedef ResolveArgs A;
    late_args_begin(args);
    constexpr size_t sig_len = TailConst<K, L>::SIG_LEN;
    __shared__ Twiddle tw_lds[INV_TW * 64];
    __shared__ int32_t xpose[GWAVES][N];
    const int spec = (int)args.ctl->spec;
    if (spec == 1) return;  // k_sign_tail wrote done[] / kappa[] / the signature itself
    const uint32_t m = args.ctl->m;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * GWAVES) tw_lds[i] = args.inv_tab[i];
    __syncthreads();
    const LdsTw itw{tw_lds, lane};
    const uint32_t wid = blockIdx.x * GWAVES + wave, n_waves = gridDim.x * GWAVES;
    for (uint32_t i = wid; i < m; i += n_waves) {
        const uint32_t op = LATE(A, act)[i];
        const uint32_t* kidx = LATE(A, key_idx);
        const size_t key = kidx ? kidx[op] : op;
        const bool s2_oor = LATE(A, key_oor)[LATE(A, oor_by_op) ? op : key] != 0;
        const int mine = lane < spec ? LATE(A, accept)[(size_t)i * spec + lane] : 0;
        unsigned long long mask = __ballot(mine != 0);
        bool fin = false;
        while (mask && !fin) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            // every z_j is wanted (bytes); the r_i were tested by k_sign_tail and are not needed on their own
            const size_t slot = (size_t)i * spec + j;
            fin = tail_attempt<K, L, G2HI, true>(args.a, slot, (size_t)LATE(A, slot_y)[slot], key, 0u, (1u << L) - 1u, s2_oor,
                                                 LATE(A, sigs) + (size_t)op * sig_len, xpose[wave], itw, lane);
        }
        if (lane == 0) {
            if (fin) LATE(A, done)[op] = 1;
            else { uint16_t* kp = LATE(A, kappa) + op; *kp = (uint16_t)(*kp + spec * L); }
        }
    }
}

// keep the unfinished ops for the next round (order is irrelevant: ops are independent)
__global__ __launch_bounds__(256) void k_compact(RoundCtl* __restrict__ ctl, int parity, const uint32_t* __restrict__ act_in,
                                                 const int32_t* __restrict__ done, uint32_t* __restrict__ act_out,
                                                 uint32_t* __restrict__ ypos_out, uint32_t* __restrict__ exp_list) {
    const uint32_t m = ctl->m;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const uint32_t op = act_in[i];
        if (!done[op]) {
            const uint32_t j = atomicAdd(&ctl->cnt[parity ^ 1], 1u);
            act_out[j] = op;
            if (ypos_out) ypos_out[j] = i;  // where this round kept the op (= the row of a mask generated ahead for it)
        } else if (exp_list) {
            exp_list[atomicAdd(&ctl->exp_cnt, 1u)] = op;  // finished in this round: its signature is complete (k_export_done)
        }
    }
}

// The same for a SMALL call (at most 256 ops: ONE workgroup, so every count is the workgroup's own), with what would otherwise be two more
// launches behind it: (i) the counts the host's end-of-call check reads -- the survivors and the call's statistics -- go to device-visible
// host memory (host_ctl) instead of a copy of the control block behind the last round (a 4 us blit kernel and its gap on a 130 us call);
// every round overwrites them, the host looks after the last one.  (ii) With next_on the round that follows is opened here as well
// (make_slots_body for the other parity, on the list just built): no k_make_slots launch between two rounds of a small call.
__device__ __forceinline__ void compact_small_body(const CompactSmallArgs& A, uint32_t tid) {  // one workgroup of 256 threads
    RoundCtl* const ctl = A.ctl;
    const int parity = A.parity;
    const uint32_t m = ctl->m;
    for (uint32_t i = tid; i < m; i += 256) {
        const uint32_t op = A.act_in[i];
        if (!A.done[op]) {
            const uint32_t j = atomicAdd(&ctl->cnt[parity ^ 1], 1u);
            A.act_out[j] = op;
            if (A.ypos_out) A.ypos_out[j] = i;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's list entries (and counts) have reached L2 before the barrier
    __syncthreads();
    const uint32_t left = __hip_atomic_load(&ctl->cnt[parity ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (A.host_ctl && tid == 0) {
        A.host_ctl->cnt[parity ^ 1] = left;
        A.host_ctl->cnt[parity] = ctl->cnt[parity];
        A.host_ctl->slots_total = ctl->slots_total;
        A.host_ctl->ops_total = ctl->ops_total;
    }
    if (A.next_on) {
        const uint32_t spec = left ? A.rule.spec(left, A.spec_max) : 1u;
        make_slots_body(ctl, parity ^ 1, left, spec, ctl->gen_par[parity], A.ns_cap, A.act_out, A.kappa, A.l, A.slot_op, A.slot_kappa, A.key_idx, A.gen_op,
                        A.gen_kappa, A.gen_key, 0, 0, A.ypos_out, A.slot_y, tid, 256u, false);
    }
}
__global__ __launch_bounds__(256) void k_compact_small(CompactSmallArgs A) { compact_small_body(A, threadIdx.x); }

// ------------------------------------------------------------------------------------
// The SECOND half of a small signing round as ONE launch (calls of <= 256 ops; the first half is k_sign_front_small): the tests of every
// candidate (k_sign_tail), the one signature of each op's first survivor (k_resolve) and the bookkeeping that opens the next round
// (k_compact_small) hand over inside the launch, the way the single-launch verify / keygen / round-front kernels do.
//   * every op of the active list owns a cluster of ceil(spec / 4) workgroups, one wave per candidate: stage 1 of the iteration, the tests
//     that can reject (tail_attempt<!FULL>), verdict to accept[];
//   * the last workgroup of the cluster to arrive (arrival counter of the op's list position, release / acquire at agent scope, nobody
//     spins) builds the signature of the FIRST survivor with its four waves side by side: z_j by wave j mod 4 (norm test, bytes),
//     then the K hint rows by wave i mod 4 (one transform per row, hint bits as four ballot masks per row into LDS), then the verdict
//     (weight <= omega, ML-DSA-44: ||c t0||inf) and the hint bytes from the masks.  k_resolve runs the same iteration on ONE wave:
//     11 inverse transforms in a row, 13 us of a one-op call; here three or four.  A survivor the hint stage rejects (1 %) gives way to
//     the next one, as in the reference's loop (ml_dsa.rs:212-330);
//   * the workgroup that finishes the LAST op of the round compacts the active list, reports to the host's control block and opens the
//     next round (compact_small_body).
// Same rows in, same bytes out as the three kernels (tests/test_gpu_small_calls.py: fused = pipeline = oracle).
struct SmallSignBackArgs {
    TailPtrs a;          // (offset 0: tail_attempt reads these from the kernarg segment)
    uint8_t* sigs;
    int ct0_exact, oor_by_op;
    RoundCtl* ctl;
    const Twiddle* inv_tab;
    const uint32_t *act, *slot_y, *key_idx;
    const uint8_t *wrisk, *yrisk, *key_oor;
    uint16_t* kappa;
    int32_t *done, *accept;
    uint32_t* ctr;       // [ops_cap + 1] arrival counters (zero between launches): one per list position, one for the round
    uint32_t ops_cap, members, clusters;  // clusters: list positions the grid covers at once (a multiple of 8)
    CompactSmallArgs C;
};
static_assert(offsetof(SmallSignBackArgs, a) == 0 && offsetof(SmallSignBackArgs, ct0_exact) == offsetof(SignTailArgs, ct0_exact), "tail_attempt's kernarg offsets");

// the whole iteration for ONE candidate by the workgroup's four waves; every wave returns the same verdict
template <int K, int L, bool G2HI>
__device__ __forceinline__ bool resolve_coop4(int ct0_exact, size_t yrow, size_t key, bool s2_oor, uint8_t* sig, int32_t (*xpose)[N],
                                              unsigned long long (*hmask)[4], int32_t* row_max, int* s_zbad, const LdsTw& itw, int lane, int wave) {
    constexpr int gb = TailConst<K, L>::GB, beta = TailConst<K, L>::BETA, omega = TailConst<K, L>::OMEGA, ctilde_len = TailConst<K, L>::CTILDE;
    constexpr int32_t GAMMA2 = G2HI ? (Q - 1) / 32 : (Q - 1) / 88;
    constexpr bool CT0_CAN_FAIL = !G2HI;
    constexpr int32_t gamma1 = 1 << gb;
    constexpr int cb = gb + 1, YCB = cb;
    auto wave_max = [](int32_t x) {
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) {
            const int32_t o = __shfl_xor(x, mm);
            x = o > x ? o : x;
        }
        return x;
    };
    if (threadIdx.x == 0) *s_zbad = 0;
    const int4 cv = reinterpret_cast<const int4*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, c_hat)) + yrow * N)[lane];
    if (wave == 0 && lane < ctilde_len) sig[lane] = late_arg<const uint8_t*>((unsigned)offsetof(TailPtrs, ctilde))[yrow * 64 + lane];
    __syncthreads();
    // ---- z_j = y_j + invNTT(c_hat o s1_hat_j): norm test and bytes (ml_dsa.rs:243-280, encodings.rs:238-276)
#pragma unroll 1
    for (int j = wave; j < L; j += GWAVES) {
        int32_t v[4], r[4], zc4[4];
        load_packed(v, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, s1)) + (key * L + j) * (size_t)N, lane);
        const uint8_t* xb = reinterpret_cast<const uint8_t*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, y))) + (yrow * L + j) * (size_t)(32 * YCB);
        uint32_t x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = *reinterpret_cast<const u32_any*>(xb + k * (8 * YCB) + ((lane * YCB) >> 3));
        r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
        ntt_inv_wave(r, itw, lane, F_MONT);
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t zs = y_from_raw<YCB>(x[k], lane) + r[k];
            const int32_t zc = zs - ((((Q / 2) - zs) >> 31) & Q);
            zc4[k] = zc;
            bad |= (zc < 0 ? -zc : zc) >= gamma1 - beta;
        }
        if (__ballot(bad) != 0ull && lane == 0) *s_zbad = 1;
#pragma unroll
        for (int k = 0; k < 4; k++) xpose[wave][64 * k + lane] = zc4[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int4 z4 = reinterpret_cast<const int4*>(xpose[wave])[lane];
        const int32_t zz[4] = {z4.x, z4.y, z4.z, z4.w};
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {  // BitPack(z, gamma1 - 1, gamma1): field = gamma1 - z
            const uint64_t f = (uint64_t)(uint32_t)(gamma1 - zz[t]);
            const int sh = t * cb;
            lo |= f << sh;
            if (sh + cb > 64) hi |= (uint32_t)(f >> (64 - sh));
        }
        constexpr int NBYTES = YCB / 2;
        uint8_t* dst = sig + ctilde_len + (size_t)j * (32 * YCB) + (size_t)lane * NBYTES;
        *reinterpret_cast<u64_any*>(dst) = lo;
        if constexpr (NBYTES == 10) *reinterpret_cast<u16_any*>(dst + 8) = (uint16_t)hi;
        else dst[8] = (uint8_t)hi;
        __builtin_amdgcn_wave_barrier();
    }
    // ---- hint rows: h_i = [HighBits(w_i + invNTT(c_hat o (t0_hat_i - s2_hat_i))) != HighBits(w_i)]  (tail_attempt, stage 2)
#pragma unroll 1
    for (int i = wave; i < K; i += GWAVES) {
        int32_t v[4], v2[4], r[4], base[4];
        load_packed(v, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, t0)) + (key * K + i) * (size_t)N, lane);
        load_packed(v2, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, s2)) + (key * K + i) * (size_t)N, lane);
        const uint32_t* wq = reinterpret_cast<const uint32_t*>(late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, w))) + (yrow * K + i) * (size_t)PACKED_POLY_DWORDS;
        const Packed3 wp{wq[lane], wq[64 + lane], wq[128 + lane]};
        if (s2_oor) {
            r[0] = mont_mul(cv.x, v2[0]); r[1] = mont_mul(cv.y, v2[1]); r[2] = mont_mul(cv.z, v2[2]); r[3] = mont_mul(cv.w, v2[3]);
            ntt_inv_wave(r, itw, lane, F_MONT);
            const int4 w4 = unpack24(wp);
            base[0] = caddq(w4.x - r[0]); base[1] = caddq(w4.y - r[1]); base[2] = caddq(w4.z - r[2]); base[3] = caddq(w4.w - r[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] -= v2[k];
        }
        r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
        ntt_inv_wave(r, itw, lane, F_MONT);
        if (!s2_oor) {
            const int4 w4 = unpack24(wp);
            base[0] = w4.x; base[1] = w4.y; base[2] = w4.z; base[3] = w4.w;
        }
        int32_t dmax = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (CT0_CAN_FAIL) {
                int32_t tc = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);
                tc = tc < 0 ? -tc : tc;
                dmax = tc > dmax ? tc : dmax;
            }
            int32_t a1, a0, b1, b0;
            const int32_t sum = base[k] + r[k] - Q;
            decompose<G2HI>(caddq(sum), a1, a0);
            decompose<G2HI>(base[k], b1, b0);
            const unsigned long long mask = __ballot(a1 != b1);
            if (lane == 0) hmask[i][k] = mask;
        }
        if constexpr (CT0_CAN_FAIL) {
            dmax = wave_max(dmax);
            if (lane == 0) row_max[i] = dmax;
        }
    }
    __syncthreads();
    int index = 0;
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) index += __popcll(hmask[i][k]);
    bool ok = *s_zbad == 0 && index <= omega;  // ml_dsa.rs:280, 313-315
    if constexpr (CT0_CAN_FAIL) {
        int32_t dmax = 0;
#pragma unroll
        for (int i = 0; i < K; i++) dmax = row_max[i] > dmax ? row_max[i] : dmax;
        bool ct0_ok = dmax + (s2_oor ? 0 : beta) < GAMMA2;
        if (!s2_oor && (!ct0_ok || ct0_exact)) {  // (workgroup-uniform: every wave read the same LDS words) ||c t0||inf proper, ml_dsa.rs:312
            __syncthreads();
#pragma unroll 1
            for (int i = wave; i < K; i += GWAVES) {
                int32_t v[4], r[4], tmax = 0;
                load_packed(v, late_arg<const int32_t*>((unsigned)offsetof(TailPtrs, t0)) + (key * K + i) * (size_t)N, lane);
                r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
                ntt_inv_wave(r, itw, lane, F_MONT);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t tc = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);
                    tc = tc < 0 ? -tc : tc;
                    tmax = tc > tmax ? tc : tmax;
                }
                tmax = wave_max(tmax);
                if (lane == 0) row_max[i] = tmax;
            }
            __syncthreads();
            int32_t tmax = 0;
#pragma unroll
            for (int i = 0; i < K; i++) tmax = row_max[i] > tmax ? row_max[i] : tmax;
            ct0_ok = tmax < GAMMA2;
        }
        ok = ok && ct0_ok;
    }
    if (ok && wave == 0) {  // HintBitPack (conversion.rs:277-328) from the rows' masks
        uint8_t* hy = sig + ctilde_len + (size_t)L * (32 * cb);
        for (int i = lane; i < omega + K; i += 64) hy[i] = 0;
        int idx = 0;
#pragma unroll 1
        for (int i = 0; i < K; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long mask = hmask[i][k];
                if ((mask >> lane) & 1ull) {
                    const int rank = idx + __popcll(mask & ((1ull << lane) - 1ull));
                    if (rank < omega) hy[rank] = (uint8_t)(64 * k + lane);
                }
                idx += __popcll(mask);
            }
            if (lane == 0) hy[omega + i] = (uint8_t)(idx < 255 ? idx : 255);
        }
    }
    __syncthreads();  // the LDS words are free for the next candidate
    return ok;
}

template <int K, int L, bool G2HI>
__global__ __launch_bounds__(64 * GWAVES) void k_sign_back_small(SmallSignBackArgs A0) {
    typedef SmallSignBackArgs A;
    late_args_begin(A0);
    constexpr size_t sig_len = TailConst<K, L>::SIG_LEN;
    __shared__ Twiddle tw_lds[INV_TW * 64];
    __shared__ __attribute__((aligned(16))) int32_t xpose[GWAVES][N];
    __shared__ unsigned long long hmask[K][4];
    __shared__ int32_t row_max[K];
    __shared__ int s_last, s_zbad;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0 && LATE(A, ctl)->m == 0) {  // nothing left: one workgroup still reports and opens the (empty) next round, as k_compact_small would
        compact_small_body(late_ref<CompactSmallArgs>((unsigned)offsetof(A, C)), threadIdx.x);
        return;
    }
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)LATE(A, ctl)->m), spec = (uint32_t)__builtin_amdgcn_readfirstlane((int)LATE(A, ctl)->spec);
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t members = LATE(A, members);
    const uint32_t member0 = q % members;
    const uint32_t i0 = (q / members) * 8 + xcd;  // first position in the active list; the cluster's workgroups share an XCD (and its L2)
    const uint32_t n_members = (spec + GWAVES - 1) / GWAVES;  // workgroup-sized pieces of an op's candidates
    if (m == 0 || i0 >= m || member0 >= n_members) return;  // (whole workgroup)
    for (int t = threadIdx.x; t < INV_TW * 64; t += 64 * GWAVES) tw_lds[t] = LATE(A, inv_tab)[t];
    __syncthreads();
    const LdsTw itw{tw_lds, lane};
    // The grid is sized from the PLAN of the round (clusters = expected unfinished ops + 6 sigma, members = the candidates per op the rule
    // gives in that range); the device's own counts decide.  A round that turns out larger is still complete: a workgroup walks further
    // list positions (stride = the grid's clusters) and further pieces of an op's candidates (stride = the grid's members).
#pragma unroll 1
    for (uint32_t i = i0; i < m; i += LATE(A, clusters)) {
        // (the round's counts made opaque per iteration: what is derived from them -- lane masks, m - 1 ... -- is recomputed where it is used
        //  instead of being hoisted out of the loop and spilled)
        uint32_t ms = m, sp = spec;
        asm volatile("" : "+s"(ms), "+s"(sp));
        const uint32_t n_mem = (sp + GWAVES - 1) / GWAVES;
        const uint32_t op = LATE(A, act)[i];
        const size_t key = LATE(A, key_idx) ? LATE(A, key_idx)[op] : op;
        const bool s2_oor = LATE(A, key_oor)[LATE(A, oor_by_op) ? op : key] != 0;
        // ---------------------------------------------------------------- stage 1 of the op's candidates: one wave each
        uint32_t pieces = 0;
#pragma unroll 1
        for (uint32_t member = member0; member < n_mem; member += LATE(A, members), pieces++) {
            const uint32_t cand = member * GWAVES + (uint32_t)wave;
            if (cand < sp) {
                const size_t slot = (size_t)i * sp + cand;
                const size_t yrow = LATE(A, slot_y)[slot];
                const uint32_t f = (uint32_t)LATE(A, wrisk)[yrow];
                const uint8_t* yr = LATE(A, yrisk) + yrow * L;
                uint32_t zr = 0;
#pragma unroll
                for (int j = 0; j < L; j++) zr |= (yr[j] ? 1u : 0u) << j;
                const uint32_t r_risky = s2_oor ? (1u << K) - 1u : (f & 0xFFu);
                const bool ok = tail_attempt<K, L, G2HI, false>(A0.a, slot, yrow, key, r_risky, zr, s2_oor, nullptr, xpose[wave], itw, lane);
                if (lane == 0) __hip_atomic_store(&LATE(A, accept)[slot], ok ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---------------------------------------------------------------- hand-over to the last workgroup of the op to arrive
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's own stores have reached L2 (kernels_small.hip, hand-over)
        __syncthreads();
        if (n_mem > 1) {
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const uint32_t seen = __hip_atomic_fetch_add(&LATE(A, ctr)[i], pieces, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = seen + pieces == n_mem;
                if (last) __hip_atomic_store(&LATE(A, ctr)[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = last;
            }
            __syncthreads();
            if (!s_last) continue;  // (workgroup-uniform)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        // ---------------------------------------------------------------- the op's first survivor -> its signature (four waves)
        const int mine = (uint32_t)lane < sp ? __hip_atomic_load(&LATE(A, accept)[(size_t)i * sp + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        unsigned long long mask = __ballot(mine != 0);  // (the same in every wave)
        bool fin = false;
        while (mask && !fin) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const size_t slot = (size_t)i * sp + j;
            fin = resolve_coop4<K, L, G2HI>(LATE(A, ct0_exact), (size_t)LATE(A, slot_y)[slot], key, s2_oor, LATE(A, sigs) + (size_t)op * sig_len, xpose, hmask, row_max, &s_zbad, itw,
                                            lane, wave);
        }
        if (threadIdx.x == 0) {
            if (fin) LATE(A, done)[op] = 1;
            else { uint16_t* kp = LATE(A, kappa) + op; *kp = (uint16_t)(*kp + sp * L); }  // ml_dsa.rs:281 / 316 for every candidate of the round
        }
        // ---------------------------------------------------------------- the round's last op: compaction, report, next round
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t seen = __hip_atomic_fetch_add(&LATE(A, ctr)[LATE(A, ops_cap)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = seen == ms - 1;
            if (last) __hip_atomic_store(&LATE(A, ctr)[LATE(A, ops_cap)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            compact_small_body(late_ref<CompactSmallArgs>((unsigned)offsetof(A, C)), threadIdx.x);
            return;  // (the last op of the round: nothing follows)
        }
    }
}

// Signatures straight into the caller's HOST memory (mldsa_sign_host with page-locked buffers): after every round the ops that
// finished in it are copied from the device-side signature buffer (which the tail kernels write byte by byte) to `host` -- a
// device-visible pointer to the caller's page-locked array -- so the signatures cross PCIe while later rounds still run, instead
// of in one trailing 217 MB copy.  k_compact appends every op that finished to exp_list (completion order) and k_export_snap
// records how far the list had grown at the end of round r (exp_hi[r]); the export launch of round r, on a helper stream,
// copies the ops exp_list[exp_hi[r - 1] .. exp_hi[r]) -- it shares nothing with later rounds, so the round chain never waits
// for it.  One wave per row; 16-byte aligned stores (tools/ubench_d2h.hip: dwordx4 stores reach the DMA engines' 53 GB/s,
// dword stores a third of that), the source words funnel-shifted to the destination's alignment.
__global__ void k_export_snap(RoundCtl* __restrict__ ctl, int round) {
    ctl->exp_hi[round & (MLDSA_EXP_RING - 1)] = ctl->exp_cnt;
}

__global__ __launch_bounds__(256) void k_export_done(const RoundCtl* __restrict__ ctl, int round, const uint32_t* __restrict__ exp_list,
                                                     const uint8_t* __restrict__ sigs, uint8_t* __restrict__ host, size_t sig_len) {
    const uint32_t lo = round > 0 ? ctl->exp_hi[(round - 1) & (MLDSA_EXP_RING - 1)] : 0u, hi = ctl->exp_hi[round & (MLDSA_EXP_RING - 1)];
    const int lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (uint32_t i = lo + wid; i < hi; i += n_waves) {
        const uint32_t op = exp_list[i];
        const uint8_t* src = sigs + (size_t)op * sig_len;
        uint8_t* dst = host + (size_t)op * sig_len;
        const int head = (int)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15);  // bytes until dst is 16-byte aligned
        const int n_q = ((int)sig_len - head) / 16 - 1;                                  // the last quad goes byte-wise: no read past the row
        uint4* dst16 = reinterpret_cast<uint4*>(dst + head);
        const uintptr_t sa0 = reinterpret_cast<uintptr_t>(src + head);
        const uint32_t* src4 = reinterpret_cast<const uint32_t*>(sa0 & ~(uintptr_t)3);
        const int sh = (int)(sa0 & 3) * 8;
        for (int q = lane; q < n_q; q += 64) {
            const uint32_t* w = src4 + 4 * q;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            uint4 v = make_uint4(w0, w1, w2, w3);
            if (sh) {
                const uint32_t w4 = w[4];
                v = make_uint4(__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh), __builtin_amdgcn_alignbit(w3, w2, sh),
                               __builtin_amdgcn_alignbit(w4, w3, sh));
            }
            dst16[q] = v;
        }
        if (lane < head) dst[lane] = src[lane];
        const int tail0 = head + 16 * (n_q > 0 ? n_q : 0);
        for (int t = tail0 + lane; t < (int)sig_len; t += 64) dst[t] = src[t];
    }
}

// first active list: every op except those whose ctx is too long (lib.rs:274) or whose key index is out of
// range (bad: 0 = fine, 1 = ctx too long, 2 = bad key index); a refused op gets an all-zero signature (the signature buffer
// as a whole is NOT cleared: every other op's bytes are written by its accepted attempt).  ctl must be zeroed beforehand.
__global__ __launch_bounds__(256) void k_init_active(size_t n, const int32_t* __restrict__ bad_op, int32_t* __restrict__ done,
                                                     uint16_t* __restrict__ kappa, int32_t* __restrict__ status,
                                                     uint32_t* __restrict__ act_out, RoundCtl* __restrict__ ctl,
                                                     uint8_t* __restrict__ sigs, size_t sig_len) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    kappa[i] = 0;  // ml_dsa.rs:204
    const int bad = bad_op[i];
    done[i] = bad;
    if (status) status[i] = bad == 0 ? MLDSA_OK : bad == 1 ? MLDSA_ERR_CTX_LEN : MLDSA_ERR_PARAM;
    if (!bad) act_out[atomicAdd(&ctl->cnt[0], 1u)] = (uint32_t)i;
    else
        for (size_t b = 0; b < sig_len; b++) sigs[i * sig_len + b] = 0;  // rare: one thread per refused op
}

// mldsa_sign_async: ops that are still unfinished after the enqueued rounds (parity = rounds & 1) get status
// MLDSA_ERR_AGAIN and an all-zero signature (a rejected attempt may have left bytes there).  One block per op.
__global__ __launch_bounds__(256) void k_mark_unfinished(const RoundCtl* __restrict__ ctl, int parity, const uint32_t* __restrict__ act,
                                                         int32_t* __restrict__ status, uint8_t* __restrict__ sigs, size_t sig_len) {
    const uint32_t m = ctl->cnt[parity];
    for (uint32_t i = blockIdx.x; i < m; i += gridDim.x) {
        const uint32_t op = act[i];
        if (threadIdx.x == 0) status[op] = MLDSA_ERR_AGAIN;
        for (size_t b = threadIdx.x; b < sig_len; b += 256) sigs[(size_t)op * sig_len + b] = 0;
    }
}

// key_idx checked against n_keys (the C ABI's promise: an out-of-range index never reaches memory):
// safe[op] = key_idx[op] if it is in range, else 0; bad[op] = 2 for the ops that were out of range
__global__ __launch_bounds__(256) void k_sanitize_keys(const uint32_t* __restrict__ key_idx, uint32_t n_keys, size_t n_ops,
                                                       uint32_t* __restrict__ safe, int32_t* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ops) return;
    const uint32_t k = key_idx[i];
    safe[i] = k < n_keys ? k : 0u;
    bad[i] = k < n_keys ? 0 : 2;
}

// memset(dst, 0, bytes) as a kernel (any alignment): the captured pipelines consist of kernel nodes only -- memset and
// memcpy nodes of a replayed hipGraph were observed to run out of order with the kernels around them (ROCm 7.2).
__global__ __launch_bounds__(256) void k_zero(uint8_t* __restrict__ dst, size_t bytes) {
    const size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15);
    const size_t h = head < bytes ? head : bytes;
    const size_t n16 = (bytes - h) / 16;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    int4* mid = reinterpret_cast<int4*>(dst + h);
    for (size_t i = tid; i < n16; i += stride) mid[i] = make_int4(0, 0, 0, 0);
    const size_t tail0 = h + n16 * 16;
    if (tid < h) dst[tid] = 0;
    if (tid < bytes - tail0) dst[tail0 + tid] = 0;
}

// k_zero behind the last planned round of a SMALL synchronous signing call, enqueued before the host has seen the outcome (its launch costs
// the host ~10 us: better spent while the rounds still run than after them).  It clears only if no op is left unfinished -- the extra
// rounds the host would add need rho'', kappa and the lists -- which every workgroup reads for itself.  (The control block may lie in the
// span: a workgroup that finds it cleared already reads the same zero.)
__global__ __launch_bounds__(256) void k_zero_if_done(const RoundCtl* __restrict__ ctl, int parity, uint8_t* __restrict__ dst, size_t bytes) {
    if (__hip_atomic_load(&ctl->cnt[parity], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15);
    const size_t h = head < bytes ? head : bytes;
    const size_t n16 = (bytes - h) / 16;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    int4* mid = reinterpret_cast<int4*>(dst + h);
    for (size_t i = tid; i < n16; i += stride) mid[i] = make_int4(0, 0, 0, 0);
    const size_t tail0 = h + n16 * 16;
    if (tid < h) dst[tid] = 0;
    if (tid < bytes - tail0) dst[tail0 + tid] = 0;
}

// number of non-zero bytes of a device range (mldsa_debug_secret_residue: test support, any alignment)
__global__ __launch_bounds__(256) void k_count_nonzero(const uint8_t* __restrict__ src, size_t bytes, unsigned long long* __restrict__ out) {
    const size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15);
    const size_t h = head < bytes ? head : bytes;
    const size_t n16 = (bytes - h) / 16;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const uint4* mid = reinterpret_cast<const uint4*>(src + h);
    unsigned long long cnt = 0;
    auto nz = [](uint32_t v) { return (unsigned)((v & 0xFFu) != 0) + (unsigned)((v & 0xFF00u) != 0) + (unsigned)((v & 0xFF0000u) != 0) + (unsigned)((v >> 24) != 0); };
    for (size_t i = tid; i < n16; i += stride) {
        const uint4 v = mid[i];
        if (v.x | v.y | v.z | v.w) cnt += nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
    }
    const size_t tail0 = h + n16 * 16;
    if (tid < h) cnt += src[tid] != 0;
    if (tid < bytes - tail0) cnt += src[tail0 + tid] != 0;
    if (cnt) atomicAdd(out, cnt);
}

// dst[i][0 .. row_bytes) = src[i][0 .. row_bytes) for rows of different strides (row_bytes % 4 == 0; kernels
// instead of hipMemcpy2DAsync so that the pipelines consist of kernel and memset nodes only when captured)
__global__ __launch_bounds__(256) void k_copy_rows(uint8_t* __restrict__ dst, size_t dst_stride, const uint8_t* __restrict__ src,
                                                   size_t src_stride, int row_bytes, size_t n_rows) {
    const int per_row = row_bytes / 4;
    const size_t total = n_rows * (size_t)per_row;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / per_row;
        const int w = (int)(e % per_row);
        const uint8_t* sp = src + r * src_stride + 4 * w;
        uint8_t* dp = dst + r * dst_stride + 4 * w;
        const uint32_t v = load_le32(sp);
        dp[0] = (uint8_t)v; dp[1] = (uint8_t)(v >> 8); dp[2] = (uint8_t)(v >> 16); dp[3] = (uint8_t)(v >> 24);
    }
}

// ------------------------------------------------------------------------------------
// keygen tail (ml_dsa.rs:88-101 + encodings.rs:18-40, 94-152): t = inv_ntt(A s1_hat) + s2,
// (t1, t0) = power2round(t); pk = rho | SimpleBitPack(t1, 10 bits); and the s1 / s2 / t0
// sections of sk.  One wave per polynomial, 4 consecutive coefficients per lane.
// The seeds of a generated key pair: seeds[key] = rho (32) | rho' (64) | K (32); rho opens pk and sk, K follows in sk
// (encodings.rs:27-31, 118-121).  Everything else of pk / sk is written by the matrix-vector kernel's epilogue
// (kernels_poly.hip k_verify_arith<.., KG = true>: the s1 / s2 sections on the way in, t1 / t0 on the way out).
__global__ __launch_bounds__(256) void k_keygen_seeds(const uint8_t* __restrict__ seeds, uint8_t* __restrict__ pk, uint8_t* __restrict__ sk,
                                                      size_t pk_len, size_t sk_len, size_t n_keys) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t key = t >> 5;
    const int b = (int)(t & 31);
    if (key >= n_keys) return;
    const uint8_t r = seeds[key * 128 + b];
    pk[key * pk_len + b] = r;
    sk[key * sk_len + b] = r;
    sk[key * sk_len + 32 + b] = seeds[key * 128 + 96 + b];
}

// ------------------------------------------------------------------------------------
// The inverse direction of k_unpack_ntt, one wave per polynomial: a key polynomial held as NTT-domain
// Montgomery values (x_hat * 2^32, src/types.rs:19-41) back to coefficients -- mont_reduce, inv_ntt and the
// centring of SerDes::into_bytes (src/lib.rs:427-493) and private_to_public_key (src/ml_dsa.rs:510-541) --
// then either
//   bits > 0, b >= 0: BitPack field b - centred value       (skEncode: s1 / s2 with b = eta, t0 with b = 2^12)
//   bits > 0, b <  0: SimpleBitPack field canonical >> 13   (pkEncode of t1 from t1_d2_hat_mont, lib.rs:481-490)
//   bits == 0:        the centred coefficients as int32[256] (get_public_key's s1 / s2)
// into dst + key * key_stride + poly_off + j * 32 * bits (bytes) or out32[(key * out_ppk + out_off + j)][256].
__global__ __launch_bounds__(GBLOCK) void k_key_intt(const int32_t* __restrict__ src, int polys_per_key, size_t n_keys, int bits, int b,
                                                     uint8_t* __restrict__ dst, size_t key_stride, size_t poly_off,
                                                     int32_t* __restrict__ out32, int out_ppk, int out_off,
                                                     const Twiddle* __restrict__ inv_tab) {
    __shared__ int32_t xp[GWAVES][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * GWAVES + wave;
    const size_t n_waves = (size_t)gridDim.x * GWAVES;
    const size_t n_polys = n_keys * (size_t)polys_per_key;
    InvTw tw;
    load_inv_tw(tw, inv_tab, lane);
    for (size_t p = wid; p < n_polys; p += n_waves) {
        const size_t key = p / polys_per_key;
        const int j = (int)(p % polys_per_key);
        int32_t r[4];
        load_packed(r, src + p * N, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = mont_mul(reduce32(r[k]), 1);  // mont_reduce(x_hat_mont) = x_hat
        ntt_inv_wave(r, tw, lane, F_MONT);                                // canonical [0, q), r[k] = x[64 k + lane]
#pragma unroll
        for (int k = 0; k < 4; k++) xp[wave][64 * k + lane] = r[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int4 c4 = reinterpret_cast<const int4*>(&xp[wave][0])[lane];  // coefficients 4 lane .. 4 lane + 3
        const int32_t cc[4] = {c4.x, c4.y, c4.z, c4.w};
        if (bits == 0) {
            int32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = cc[k] - ((((Q / 2) - cc[k]) >> 31) & Q);  // > q/2 -> - q (ml_dsa.rs:521-527)
            store_packed(o, out32 + (key * out_ppk + out_off + j) * (size_t)N, lane);
        } else {
            uint32_t f[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int32_t cen = cc[k] - ((((Q / 2) - cc[k]) >> 31) & Q);
                f[k] = b < 0 ? (uint32_t)(cc[k] >> 13) : (uint32_t)(b - cen);
                f[k] &= (1u << bits) - 1u;
            }
            store_fields(dst + key * key_stride + poly_off + (size_t)j * (32 * bits), f, bits, lane);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Does a key's s2 lie in [-eta, eta]?  expand_private accepts every bit pattern (conversion.rs:259-260), so s2 may hold
// -5 (eta = 2) or -11 (eta = 4), and then ||c s2||inf can exceed beta = tau * eta -- the bound k_sign_tail's one-transform hint
// stage rests on.  One wave per (unit, polynomial): unit = key of the table, or op of the chunk when the table is larger than
// the chunk (then the key comes from kidx).  oor[unit] = 1 if any coefficient is out of range (zeroed by the caller).
__global__ __launch_bounds__(GBLOCK) void k_key_range(const int32_t* __restrict__ s2, int k, int eta, const uint32_t* __restrict__ kidx,
                                                      size_t n_units, uint8_t* __restrict__ oor, const Twiddle* __restrict__ inv_tab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * GWAVES + wave, n_waves = (size_t)gridDim.x * GWAVES;
    InvTw tw;
    load_inv_tw(tw, inv_tab, lane);
    for (size_t u = wid; u < n_units * (size_t)k; u += n_waves) {
        const size_t unit = u / k;
        const size_t key = kidx ? kidx[unit] : unit;
        int32_t r[4];
        load_packed(r, s2 + (key * k + u % k) * (size_t)N, lane);
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = mont_mul(reduce32(r[i]), 1);  // mont_reduce(x_hat_mont) = x_hat
        ntt_inv_wave(r, tw, lane, F_MONT);                                // canonical [0, q)
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int32_t cen = r[i] - ((((Q / 2) - r[i]) >> 31) & Q);
            bad |= (cen < 0 ? -cen : cen) > eta;
        }
        if (__ballot(bad) != 0ull && lane == 0) oor[unit] = 1;
    }
}

// get_public_key tail (ml_dsa.rs:543-556): t = as1 + s2 (full_reduce32), t1 = Power2Round(t).hi,
// t1_d2_hat_mont = NTT(t1) * 2^13 in Montgomery form.  One wave per polynomial.  as1[key][K] canonical coefficients
// (k_verify_arith<.., false> output), s1s2[key][L + K] centred coefficients (k_key_intt output).
__global__ __launch_bounds__(GBLOCK) void k_t1_hat(const int32_t* __restrict__ as1, const int32_t* __restrict__ s1s2, int k, int l,
                                                   int32_t* __restrict__ t1_out, size_t n_keys, const Twiddle* __restrict__ fwd_tab) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * GWAVES + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * GWAVES;
    FwdTw tw;
    load_fwd_tw(tw, fwd_tab, lane);
    for (size_t p = wave; p < n_keys * (size_t)k; p += n_waves) {
        const size_t key = p / k;
        const int i = (int)(p % k);
        int32_t a[4], s[4], r[4];
        load_strided(a, as1 + p * N, lane);
        load_strided(s, s1s2 + (key * (l + k) + l + i) * (size_t)N, lane);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int32_t tt = freeze(a[c] + s[c]);
            r[c] = (tt + (1 << 12) - 1) >> 13;  // power2round high part, high_low.rs:26-31
        }
        ntt_fwd_wave(r, tw, lane);
#pragma unroll
        for (int c = 0; c < 4; c++) r[c] = mont_mul(r[c], 6346488 /* 2^13 * 2^64 mod q */);
        store_packed(r, t1_out + p * N, lane);
    }
}

// ------------------------------------------------------------------------- launchers
int launch_unpack_ntt(mldsa_ctx* ctx, const uint8_t* src, size_t key_stride, size_t poly_off, int bits, int b, int32_t scale,
                      int32_t* out, int polys_per_key, size_t n_keys, hipStream_t s) {
    if (n_keys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_unpack_ntt, dim3(grid_for(ctx, n_keys * (size_t)polys_per_key, GWAVES, 8)), dim3(GBLOCK), 0, s, src,
                       key_stride, poly_off, bits, b, scale, out, polys_per_key, n_keys, ctx->d_fwd_tw);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// the tail kernels' compile-time scalars (TailConst) against the parameter table the rest of the library uses (capi.hip PARAMS)
template <int K, int L>
static bool tail_consts_are(const mldsa_params* p) {
    typedef TailConst<K, L> T;
    return p->k == K && p->l == L && p->gamma1 == (1 << T::GB) && p->beta == T::BETA && p->omega == T::OMEGA && p->ctilde_len == T::CTILDE &&
           (size_t)p->sig_len == T::SIG_LEN;
}
static bool tail_consts_match(const mldsa_params* p) {
    return p->set == MLDSA_44 ? tail_consts_are<4, 4>(p) : p->set == MLDSA_65 ? tail_consts_are<6, 5>(p) : p->set == MLDSA_87 && tail_consts_are<8, 7>(p);
}

// Self-test of late_arg (run by mldsa_ctx_create).  late_arg relies on the code-object ABI putting a kernel's first by-value argument
// at byte 0 of the kernarg segment with the host's struct layout.  That holds for every ROCm so far, but nothing in the language
// promises it, and a toolchain that changed it would give silently wrong signatures, not a build error.  This kernel receives a
// SignTailArgs the way k_sign_tail does, reads every field both ways -- as a normal argument and through late_arg at its
// offsetof() -- and reports the fields that differ; a context is not created on a mismatch.
__global__ void k_late_arg_selftest(SignTailArgs args, uint32_t* mismatch) {
    late_args_begin(args);
    uint32_t bad = 0;
    int bit = 0;
#define MLDSA_LATE_CHECK(ARGS, field, normal)                                    \
    do {                                                                         \
        if (LATE(ARGS, field) != (normal)) bad |= 1u << bit;                     \
        bit++;                                                                   \
    } while (0)
    MLDSA_LATE_CHECK(TailPtrs, c_hat, args.a.c_hat);
    MLDSA_LATE_CHECK(TailPtrs, y, args.a.y);
    MLDSA_LATE_CHECK(TailPtrs, w, args.a.w);
    MLDSA_LATE_CHECK(TailPtrs, ctilde, args.a.ctilde);
    MLDSA_LATE_CHECK(TailPtrs, s1, args.a.s1);
    MLDSA_LATE_CHECK(TailPtrs, s2, args.a.s2);
    MLDSA_LATE_CHECK(TailPtrs, t0, args.a.t0);
    MLDSA_LATE_CHECK(SignTailArgs, sigs, args.sigs);
    MLDSA_LATE_CHECK(SignTailArgs, ct0_exact, args.ct0_exact);
    MLDSA_LATE_CHECK(SignTailArgs, oor_by_op, args.oor_by_op);
    MLDSA_LATE_CHECK(SignTailArgs, ctl, args.ctl);
    MLDSA_LATE_CHECK(SignTailArgs, inv_tab, args.inv_tab);
    MLDSA_LATE_CHECK(SignTailArgs, slot_op, args.slot_op);
    MLDSA_LATE_CHECK(SignTailArgs, slot_y, args.slot_y);
    MLDSA_LATE_CHECK(SignTailArgs, key_idx, args.key_idx);
    MLDSA_LATE_CHECK(SignTailArgs, wrisk, args.wrisk);
    MLDSA_LATE_CHECK(SignTailArgs, yrisk, args.yrisk);
    MLDSA_LATE_CHECK(SignTailArgs, key_oor, args.key_oor);
    MLDSA_LATE_CHECK(SignTailArgs, kappa, args.kappa);
    MLDSA_LATE_CHECK(SignTailArgs, done, args.done);
    MLDSA_LATE_CHECK(SignTailArgs, accept, args.accept);
#undef MLDSA_LATE_CHECK
    if (threadIdx.x == 0 && blockIdx.x == 0) *mismatch = bad | 0x80000000u;  // (bit 31: the kernel ran)
}

int late_arg_selftest(hipStream_t s, uint32_t* d_word) {
    SignTailArgs args;
    // distinct, recognisable values in every field (never dereferenced)
    auto tag = [](uintptr_t i) { return (uintptr_t)0x5A5A000000000000ull + i * 0x0101010101ull; };
    args.a = TailPtrs{(const int32_t*)tag(1), (const int32_t*)tag(2), (const int32_t*)tag(3), (const uint8_t*)tag(4), (const int32_t*)tag(5),
                      (const int32_t*)tag(6), (const int32_t*)tag(7)};
    args.sigs = (uint8_t*)tag(8); args.ct0_exact = 0x1234567; args.oor_by_op = -0x7654321; args.ctl = (const RoundCtl*)tag(9);
    args.inv_tab = (const Twiddle*)tag(10); args.slot_op = (const uint32_t*)tag(11); args.slot_y = (const uint32_t*)tag(12);
    args.key_idx = (const uint32_t*)tag(13); args.wrisk = (const uint8_t*)tag(14); args.yrisk = (const uint8_t*)tag(15);
    args.key_oor = (const uint8_t*)tag(16); args.kappa = (uint16_t*)tag(17); args.done = (int32_t*)tag(18); args.accept = (int32_t*)tag(19);
    hipLaunchKernelGGL(k_late_arg_selftest, dim3(1), dim3(64), 0, s, args, d_word);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_sign_tail(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* c, const int32_t* y, const int32_t* w, const uint8_t* ctilde,
                     const uint32_t* slot_op, const uint32_t* key_idx, const int32_t* s1, const int32_t* s2, const int32_t* t0,
                     uint16_t* kappa, int32_t* done, uint8_t* sigs, const RoundCtl* ctl, int32_t* accept, size_t slots_hint,
                     hipStream_t s, const uint8_t* wrisk, const uint8_t* yrisk, const uint8_t* key_oor, int oor_by_op,
                     const uint32_t* slot_y) {
    if (!wrisk || !yrisk || !key_oor || !slot_y) return set_error(MLDSA_ERR_PARAM, "sign_tail: wrisk, yrisk, key_oor and slot_y are required");
    if (!tail_consts_match(p)) return set_error(MLDSA_ERR_PARAM, "sign_tail: parameter table and compiled constants disagree");
    SignTailArgs args;
    args.a = TailPtrs{c, y, w, ctilde, s1, s2, t0};
    args.sigs = sigs; args.ct0_exact = (int)ctx->opt_ct0_exact; args.oor_by_op = oor_by_op; args.ctl = ctl; args.inv_tab = ctx->d_inv_tw;
    args.slot_op = slot_op; args.slot_y = slot_y; args.key_idx = key_idx; args.wrisk = wrisk; args.yrisk = yrisk; args.key_oor = key_oor;
    args.kappa = kappa; args.done = done; args.accept = accept;
    dim3 grid(grid_for(ctx, slots_hint, GWAVES, 16));  // more, shorter blocks than fit at once: the dispatcher evens out the early exits
#define MLDSA_TAIL(KK, LL, G2) hipLaunchKernelGGL((k_sign_tail<KK, LL, G2>), grid, dim3(64 * GWAVES), 0, s, args)
    if (p->set == MLDSA_44) MLDSA_TAIL(4, 4, false);
    else if (p->set == MLDSA_65) MLDSA_TAIL(6, 5, true);
    else MLDSA_TAIL(8, 7, true);
#undef MLDSA_TAIL
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

static inline unsigned blocks256(size_t n) { return (unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1); }

int launch_key_range(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* s2, const uint32_t* kidx, size_t n_units, uint8_t* oor,
                     hipStream_t s) {
    if (n_units == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_key_range, dim3(grid_for(ctx, n_units * (size_t)p->k, GWAVES, 8)), dim3(GBLOCK), 0, s, s2, p->k, p->eta, kidx, n_units,
                       oor, ctx->d_inv_tw);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_make_slots(mldsa_ctx*, RoundCtl* ctl, int parity, const SpecRule& rule, uint32_t sp_max, uint32_t ns_cap, const uint32_t* act,
                      const uint16_t* kappa, int l, uint32_t* slot_op, uint16_t* slot_kappa, const uint32_t* key_idx,
                      uint32_t* gen_op, uint16_t* gen_kappa, uint32_t* gen_key, size_t slots_hint, hipStream_t s, int may_use_pre,
                      int may_gen2, const uint32_t* ypos, uint32_t* slot_y) {
    hipLaunchKernelGGL(k_make_slots, dim3(blocks256(slots_hint)), dim3(256), 0, s, ctl, parity, rule, sp_max, ns_cap, act, kappa, l,
                       slot_op, slot_kappa, key_idx, gen_op, gen_kappa, gen_key, may_use_pre, may_gen2, ypos, slot_y);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_resolve(mldsa_ctx* ctx, const mldsa_params* p, const RoundCtl* ctl, const uint32_t* act, const int32_t* accept,
                   const int32_t* c, const int32_t* y, const int32_t* w, const uint8_t* ctilde, const uint32_t* key_idx,
                   const int32_t* s1, const int32_t* s2, const int32_t* t0, uint8_t* sigs, int32_t* done, uint16_t* kappa,
                   size_t ops_hint, hipStream_t s, const uint8_t* key_oor, int oor_by_op, const uint32_t* slot_y) {
    if (!key_oor || !slot_y) return set_error(MLDSA_ERR_PARAM, "resolve: key_oor and slot_y are required");
    if (!tail_consts_match(p)) return set_error(MLDSA_ERR_PARAM, "resolve: parameter table and compiled constants disagree");
    ResolveArgs args;
    args.a = TailPtrs{c, y, w, ctilde, s1, s2, t0};
    args.sigs = sigs; args.ct0_exact = (int)ctx->opt_ct0_exact; args.oor_by_op = oor_by_op; args.ctl = ctl; args.inv_tab = ctx->d_inv_tw;
    args.act = act; args.slot_y = slot_y; args.key_idx = key_idx; args.accept = accept; args.key_oor = key_oor; args.kappa = kappa; args.done = done;
    dim3 grid(grid_for(ctx, ops_hint, GWAVES, 8));
#define MLDSA_RES(KK, LL, G2) hipLaunchKernelGGL((k_resolve<KK, LL, G2>), grid, dim3(64 * GWAVES), 0, s, args)
    if (p->set == MLDSA_44) MLDSA_RES(4, 4, false);
    else if (p->set == MLDSA_65) MLDSA_RES(6, 5, true);
    else MLDSA_RES(8, 7, true);
#undef MLDSA_RES
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_compact(mldsa_ctx*, RoundCtl* ctl, int parity, const uint32_t* act_in, const int32_t* done, uint32_t* act_out,
                   size_t ops_hint, hipStream_t s, uint32_t* ypos_out, uint32_t* exp_list) {
    hipLaunchKernelGGL(k_compact, dim3(blocks256(ops_hint)), dim3(256), 0, s, ctl, parity, act_in, done, act_out, ypos_out, exp_list);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_sign_back_small(mldsa_ctx* ctx, const mldsa_params* p, const SignBackSmall& B, const CompactSmallArgs& C, hipStream_t s) {
    if (!tail_consts_match(p)) return set_error(MLDSA_ERR_PARAM, "sign_back_small: parameter table and compiled constants disagree");
    if (!B.wrisk || !B.yrisk || !B.key_oor || !B.slot_y || !B.ctr || B.ops_cap == 0 || B.ops_cap > 256 || B.spec_cap == 0 || B.spec_cap > 64)
        return set_error(MLDSA_ERR_PARAM, "sign_back_small: bad arguments");
    SmallSignBackArgs A;
    A.a = TailPtrs{B.c_hat, B.y, B.w, B.ctilde, B.s1, B.s2, B.t0};
    A.sigs = B.sigs; A.ct0_exact = (int)ctx->opt_ct0_exact; A.oor_by_op = B.oor_by_op; A.ctl = B.ctl; A.inv_tab = ctx->d_inv_tw;
    A.act = B.act; A.slot_y = B.slot_y; A.key_idx = B.key_idx; A.wrisk = B.wrisk; A.yrisk = B.yrisk; A.key_oor = B.key_oor; A.kappa = B.kappa;
    A.done = B.done; A.accept = B.accept; A.ctr = B.ctr; A.ops_cap = B.ops_cap; A.C = C;
    // the grid follows the round's plan; the kernel walks whatever the device's own counts add to it
    const uint32_t ops_grid = std::min<uint32_t>(B.ops_cap, std::max<uint32_t>(B.ops_hint, 1u)), spec_grid = std::min<uint32_t>(B.spec_cap, std::max<uint32_t>(B.spec_hint, 1u));
    A.members = (spec_grid + GWAVES - 1) / GWAVES;
    A.clusters = 8u * ((ops_grid + 7u) / 8u);
    const unsigned grid = A.members * A.clusters;
#define MLDSA_BACK(KK, LL, G2) hipLaunchKernelGGL((k_sign_back_small<KK, LL, G2>), dim3(grid), dim3(64 * GWAVES), 0, s, A)
    if (p->set == MLDSA_44) MLDSA_BACK(4, 4, false);
    else if (p->set == MLDSA_65) MLDSA_BACK(6, 5, true);
    else MLDSA_BACK(8, 7, true);
#undef MLDSA_BACK
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_compact_small(mldsa_ctx*, const CompactSmallArgs& A, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(256), 0, s, A);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_export_done(mldsa_ctx* ctx, RoundCtl* ctl, int round, const uint32_t* exp_list, const uint8_t* sigs, uint8_t* host, size_t sig_len,
                       size_t ops_hint, hipStream_t snap_stream, hipStream_t s) {
    (void)snap_stream;
    // SIXTEEN workgroups, whatever the round's size: a kernel that stores to host memory slows everything that runs beside it
    // once it has more than that in flight (tools/ubench_d2h2.hip: 16 workgroups reach 44 GB/s and cost a neighbouring
    // kernel 0-15 %; 32 reach 52 GB/s and cost it 75 %; 128 make an HBM fill beside them 12 times slower)
    (void)ctx;
    const unsigned blocks = (unsigned)std::min<size_t>(16, std::max<size_t>(1, (ops_hint + 3) / 4));
    hipLaunchKernelGGL(k_export_done, dim3(blocks), dim3(256), 0, s, ctl, round, exp_list, sigs, host, sig_len);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_export_snap(mldsa_ctx*, RoundCtl* ctl, int round, hipStream_t s) {
    hipLaunchKernelGGL(k_export_snap, dim3(1), dim3(1), 0, s, ctl, round);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_init_active(mldsa_ctx*, size_t n, const int32_t* bad_op, int32_t* done, uint16_t* kappa, int32_t* status,
                       uint32_t* act_out, RoundCtl* ctl, uint8_t* sigs, size_t sig_len, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_init_active, dim3(blocks256(n)), dim3(256), 0, s, n, bad_op, done, kappa, status, act_out, ctl, sigs, sig_len);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_mark_unfinished(mldsa_ctx*, const RoundCtl* ctl, int parity, const uint32_t* act, int32_t* status, uint8_t* sigs,
                           size_t sig_len, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_unfinished, dim3(64), dim3(256), 0, s, ctl, parity, act, status, sigs, sig_len);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_sanitize_keys(mldsa_ctx*, const uint32_t* key_idx, size_t n_keys, size_t n_ops, uint32_t* safe, int32_t* bad, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    const uint32_t nk = n_keys > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_keys;
    hipLaunchKernelGGL(k_sanitize_keys, dim3(blocks256(n_ops)), dim3(256), 0, s, key_idx, nk, n_ops, safe, bad);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_zero(mldsa_ctx* ctx, void* dst, size_t bytes, hipStream_t s) {
    if (bytes == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_zero, dim3(grid_for(ctx, bytes / 16 + 1, 256, 8)), dim3(256), 0, s, static_cast<uint8_t*>(dst), bytes);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_zero_if_done(mldsa_ctx* ctx, const RoundCtl* ctl, int parity, void* dst, size_t bytes, hipStream_t s) {
    if (bytes == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_zero_if_done, dim3(grid_for(ctx, bytes / 16 + 1, 256, 8)), dim3(256), 0, s, ctl, parity, static_cast<uint8_t*>(dst), bytes);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int count_nonzero_dev(const void* dev, size_t bytes, size_t* nonzero) {
    *nonzero = 0;
    if (bytes == 0) return MLDSA_OK;
    unsigned long long* d_cnt = nullptr;
    if (malloc_quiesced((void**)&d_cnt, sizeof(*d_cnt)) != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "count_nonzero: counter allocation");
    unsigned long long h_cnt = 0;
    hipError_t e = memset_quiesced(d_cnt, 0, sizeof(*d_cnt));
    if (e == hipSuccess) {
        const size_t blocks = std::min<size_t>((bytes / 16 + 256) / 256, 8192);
        hipLaunchKernelGGL(k_count_nonzero, dim3((unsigned)blocks), dim3(256), 0, nullptr, static_cast<const uint8_t*>(dev), bytes, d_cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = memcpy_quiesced(&h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost);
    (void)free_quiesced(d_cnt);
    if (e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "count_nonzero", e);
    *nonzero = (size_t)h_cnt;
    return MLDSA_OK;
}

int launch_copy_rows(mldsa_ctx* ctx, void* dst, size_t dst_stride, const void* src, size_t src_stride, int row_bytes, size_t n_rows,
                     hipStream_t s) {
    if (n_rows == 0 || row_bytes == 0) return MLDSA_OK;
    if (row_bytes & 3) return set_error(MLDSA_ERR_PARAM, "copy_rows: row length must be a multiple of 4 bytes");
    hipLaunchKernelGGL(k_copy_rows, dim3(grid_for(ctx, n_rows * (size_t)(row_bytes / 4), 256, 8)), dim3(256), 0, s,
                       static_cast<uint8_t*>(dst), dst_stride, static_cast<const uint8_t*>(src), src_stride, row_bytes, n_rows);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_key_intt(mldsa_ctx* ctx, const int32_t* src, int polys_per_key, size_t n_keys, int bits, int b, uint8_t* dst,
                    size_t key_stride, size_t poly_off, int32_t* out32, int out_ppk, int out_off, hipStream_t s) {
    if (n_keys == 0 || polys_per_key == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_key_intt, dim3(grid_for(ctx, n_keys * (size_t)polys_per_key, GWAVES, 8)), dim3(GBLOCK), 0, s, src, polys_per_key,
                       n_keys, bits, b, dst, key_stride, poly_off, out32, out_ppk, out_off, ctx->d_inv_tw);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_t1_hat(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* as1, const int32_t* s1s2, int32_t* t1_out, size_t n_keys,
                  hipStream_t s) {
    if (n_keys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_t1_hat, dim3(grid_for(ctx, n_keys * (size_t)p->k, GWAVES, 8)), dim3(GBLOCK), 0, s, as1, s1s2, p->k, p->l, t1_out,
                       n_keys, ctx->d_fwd_tw);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_keygen_seeds(mldsa_ctx*, const mldsa_params* p, const uint8_t* seeds, uint8_t* pk, uint8_t* sk, size_t n_keys, hipStream_t s) {
    if (n_keys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_keygen_seeds, dim3(blocks256(n_keys * 32)), dim3(256), 0, s, seeds, pk, sk, (size_t)p->pk_len, (size_t)p->sk_len, n_keys);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
