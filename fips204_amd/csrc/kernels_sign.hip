// Signing-side and key-side kernels: key expansion (try_from_bytes), the per-round pieces
// of the rejection loop of sign_internal (src/ml_dsa.rs:212-330), sig_encode, and the
// keygen tail (power2round, pk / sk encode).  Everything is integer, element-wise or one
// wave per polynomial; the NTT work goes through ntt_wave.h.
#include <cstdlib>
#include "ctx.h"
#include "keccak.h"
#include "rounding.h"

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
// Fused tail of one rejection-loop iteration (ml_dsa.rs:243-336): ONE WAVE PER SLOT, looping over
// the slot's polynomials, so there are no block barriers and no cross-wave reductions:
//   z_j  = y_j + inv_ntt(c_hat o s1_hat_j)            ||z||  <  gamma1 - beta ?
//   r_i  = w_i - inv_ntt(c_hat o s2_hat_i)            ||LowBits(r)|| < gamma2 - beta ?
//   -- only if both hold (about 1 slot in 5):
//   ct0_i = inv_ntt(c_hat o t0_hat_i)                 ||ct0|| < gamma2 ?
//   h_i  = MakeHint(-ct0_i, r_i + ct0_i)              weight(h) <= omega ?
// (c_hat = ntt(c) comes from k_ntt.)  sigEncode (encodings.rs:238-276) is written EAGERLY while the
// polynomials stream through: z bytes in stage 1, hint bytes in stage 2.  If the attempt is then
// rejected the bytes are garbage, but the op's next attempt rewrites every byte, and only an
// accepted attempt sets done[] / accept[] -- so the signature buffer of a finished op always holds
// the accepted attempt.  The reference's two `continue` stages (ml_dsa.rs:280, 312) stay two stages;
// stage 1 first transforms only the polynomials that CAN reject (flags from sign_w / ExpandMask, see
// below), the more selective LowBits test first, and leaves at the first rejection: a rejected
// ML-DSA-65 attempt costs about 1.5 of the 11 inverse NTTs of stage 1.
template <int K, int L, bool G2HI>
__global__ __launch_bounds__(64 * GWAVES) void k_sign_tail(
    const int32_t* __restrict__ c_hat, const int32_t* __restrict__ y, const int32_t* __restrict__ w,
    const uint8_t* __restrict__ ctilde, const uint32_t* __restrict__ slot_op, const uint32_t* __restrict__ key_idx,
    const int32_t* __restrict__ s1, const int32_t* __restrict__ s2, const int32_t* __restrict__ t0,
    uint16_t* __restrict__ kappa, int32_t* __restrict__ done, uint8_t* __restrict__ sigs, int spec,
    uint8_t* __restrict__ stage, size_t stage_stride, int32_t* __restrict__ accept, int gb, int beta, int omega,
    int ctilde_len, size_t sig_len, size_t n_slots, const Twiddle* __restrict__ inv_tab,
    const uint8_t* __restrict__ wrisk, const uint8_t* __restrict__ yrisk) {
    constexpr int32_t GAMMA2 = G2HI ? (Q - 1) / 32 : (Q - 1) / 88;
    __shared__ Twiddle tw_lds[INV_TW * 64];
    __shared__ int32_t xpose[GWAVES][N];     // strided -> 4 consecutive coefficients per lane (z packing)
    __shared__ int32_t rr_lds[GWAVES][K][N]; // r_i = w_i - cs2_i, kept for the hint stage
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: slot indices and row pointers stay scalar
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * GWAVES) tw_lds[i] = inv_tab[i];
    __syncthreads();
    const LdsTw itw{tw_lds, lane};
    const int32_t gamma1 = 1 << gb;
    const int cb = gb + 1;
    const uint32_t wid = blockIdx.x * GWAVES + wave, n_waves = gridDim.x * GWAVES, n_slots32 = (uint32_t)n_slots;

    // op / key of the next slot are fetched one slot ahead (two dependent loads off the critical path)
    // ... and so are its risk flags (which polynomials stage 1 has to transform, see below)
    auto risk_flags = [&](uint32_t sl, uint32_t& rr, uint32_t& zr) {
        rr = wrisk ? (uint32_t)wrisk[sl] : (1u << K) - 1u;
        zr = (1u << L) - 1u;
        if (yrisk) {
            zr = 0;
#pragma unroll
            for (int j = 0; j < L; j++) zr |= (yrisk[(size_t)sl * L + j] ? 1u : 0u) << j;
        }
    };
    uint32_t op_next = 0, key_next = 0, rrisk_next = 0, zrisk_next = 0;
    if (wid < n_slots32) {
        op_next = slot_op[wid];
        key_next = key_idx ? key_idx[op_next] : op_next;
        risk_flags(wid, rrisk_next, zrisk_next);
    }
    for (uint32_t slot32 = wid; slot32 < n_slots32; slot32 += n_waves) {
        const size_t slot = slot32, op = op_next, key = key_next;
        const uint32_t r_risky = rrisk_next, z_risky = zrisk_next;
        if (slot32 + n_waves < n_slots32) {
            op_next = slot_op[slot32 + n_waves];
            key_next = key_idx ? key_idx[op_next] : op_next;
            risk_flags(slot32 + n_waves, rrisk_next, zrisk_next);
        }
        uint8_t* sig = spec == 1 ? sigs + op * sig_len : stage + slot * stage_stride;
        const int4 cv = reinterpret_cast<const int4*>(c_hat + slot * N)[lane];
        if (lane < ctilde_len) sig[lane] = ctilde[slot * 64 + lane];
        // ---- stage 1.  ||c s1||inf and ||c s2||inf are at most beta = tau * eta, so a polynomial of w whose
        // LowBits all stay below gamma2 - 2 beta cannot fail ||LowBits(w - c s2)||inf < gamma2 - beta, and a
        // polynomial of y below gamma1 - 2 beta cannot fail ||y + c s1||inf < gamma1 - beta (ml_dsa.rs:280; with
        // |low| < gamma2 the decomposition of w - c s2 keeps the high part of w).  sign_w and ExpandMask flag the
        // few polynomials that are NOT below those margins (about 1 in 3 of w, 1 in 6 of y for ML-DSA-65):
        // pass 0 transforms only those -- the more selective LowBits test first -- and leaves at the first
        // rejection; pass 1 computes the remaining r_i and z_j for the attempts that survived (1 in 5), which
        // need them for the hints and the signature bytes.
        bool ok = true;  // wave-uniform
        // work list of a pass: bit i < K = r_i (s2 / w), bit K + j = z_j (s1 / y), walked from the low end with the
        // next polynomial's loads issued before the current inverse NTT
        uint32_t todo = (1u << (K + L)) - 1u;
        const uint32_t risky = (r_risky & ((1u << K) - 1u)) | (z_risky << K);
        auto issue_loads = [&](int idx, int32_t(&v)[4], int32_t(&x)[4]) {
            const bool is_r = idx < K;
            const int32_t* sp = is_r ? s2 + (key * K + idx) * (size_t)N : s1 + (key * L + (idx - K)) * (size_t)N;
            const int32_t* xp = is_r ? w + (slot * K + idx) * (size_t)N : y + (slot * L + (idx - K)) * (size_t)N;
            load_packed(v, sp, lane);
            load_strided(x, xp, lane);
        };
#pragma unroll 1
        for (int pass = 0; pass < 2 && ok; pass++) {
            uint32_t work = pass == 0 ? (risky & todo) : todo;
            todo &= ~work;
            int cur = work ? __ffs((int)work) - 1 : -1;
            int32_t nv[4] = {0, 0, 0, 0}, nx[4] = {0, 0, 0, 0};
            if (cur >= 0) issue_loads(cur, nv, nx);
#pragma unroll 1
            while (cur >= 0) {
                work &= work - 1u;
                const int nxt = work ? __ffs((int)work) - 1 : -1;
                const int32_t v[4] = {nv[0], nv[1], nv[2], nv[3]}, x[4] = {nx[0], nx[1], nx[2], nx[3]};
                if (nxt >= 0) issue_loads(nxt, nv, nx);
                int32_t r[4];
                r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
                ntt_inv_wave(r, itw, lane, F_MONT);
                bool bad = false;
                if (cur < K) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int32_t rr = caddq(x[k] - r[k]);     // w - cs2, canonical (both operands are in [0, q))
                        rr_lds[wave][cur][64 * k + lane] = rr;      // kept for the hint stage
                        int32_t r1, r0;
                        decompose<G2HI>(rr, r1, r0);
                        bad |= (r0 < 0 ? -r0 : r0) >= GAMMA2 - beta;
                    }
                    if (__ballot(bad) != 0ull) { ok = false; break; }
                } else {
                    const int j = cur - K;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        // z mod+- q (ml_dsa.rs:264, 334): y in (-gamma1, gamma1], cs1 in [0, q), so one conditional
                        // subtraction lands in (-q/2, q/2]
                        const int32_t zs = x[k] + r[k];
                        const int32_t zc = zs - ((((Q / 2) - zs) >> 31) & Q);
                        xpose[wave][64 * k + lane] = zc;
                        bad |= (zc < 0 ? -zc : zc) >= gamma1 - beta;
                    }
                    if (__ballot(bad) != 0ull) { ok = false; break; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int4 z4 = reinterpret_cast<const int4*>(&xpose[wave][0])[lane];
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
                    const int nbytes = cb / 2;
                    uint8_t* dst = sig + ctilde_len + (size_t)j * (32 * cb) + (size_t)lane * nbytes;
                    for (int t = 0; t < 8; t++) dst[t] = (uint8_t)(lo >> (8 * t));
                    for (int t = 8; t < nbytes; t++) dst[t] = (uint8_t)(hi >> (8 * (t - 8)));
                    __builtin_amdgcn_wave_barrier();
                }
                cur = nxt;
            }
        }
        // ---- stage 2: ct0, hints (HintBitPack, conversion.rs:277-328, written as they are found)
        if (ok) {
            uint8_t* hy = sig + ctilde_len + (size_t)L * (32 * cb);
            for (int i = lane; i < omega + K; i += 64) hy[i] = 0;
            int32_t tmax = 0;
            int index = 0;  // running count of hints, wave-uniform
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                int32_t v[4], r[4];
                load_packed(v, t0 + (key * K + i) * (size_t)N, lane);
                r[0] = mont_mul(cv.x, v[0]); r[1] = mont_mul(cv.y, v[1]); r[2] = mont_mul(cv.z, v[2]); r[3] = mont_mul(cv.w, v[3]);
                ntt_inv_wave(r, itw, lane, F_MONT);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t tc = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);  // center_mod of a canonical value
                    tc = tc < 0 ? -tc : tc;
                    tmax = tc > tmax ? tc : tmax;
                    // make_hint(Q - ct0, partial_reduce32(w - cs2 + ct0)), ml_dsa.rs:298-306
                    const int32_t rr = rr_lds[wave][i][64 * k + lane];
                    int32_t a1, a0, b1, b0;
                    const int32_t sum = rr + r[k] - Q;  // both in [0, q)
                    decompose<G2HI>(caddq(sum), a1, a0);
                    decompose<G2HI>(rr, b1, b0);  // (w - cs2 + ct0) + (Q - ct0) = w - cs2 (mod q)
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
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const int32_t o = __shfl_xor(tmax, m);
                tmax = o > tmax ? o : tmax;
            }
            ok = (tmax < GAMMA2) && (index <= omega);  // ml_dsa.rs:312-315
        }
        if (lane == 0) {
            if (spec == 1) {
                if (ok) done[op] = 1;
                else kappa[op] = (uint16_t)(kappa[op] + L);  // ml_dsa.rs:281 / 316
            } else {
                accept[slot] = ok ? 1 : 0;
            }
        }
    }
}

// Speculative rounds.  When few ops are still unfinished the GPU would idle through a long tail
// of tiny rounds (geometric, p ~ 0.2 per iteration).  Instead each unfinished op gets `spec`
// slots that try kappa, kappa + l, ..., kappa + (spec-1) l in the same round; k_resolve keeps
// the FIRST accepted candidate, which is exactly the signature the sequential loop of
// ml_dsa.rs:212-330 would have produced (all earlier candidates were rejected).
__global__ __launch_bounds__(256) void k_make_slots(const uint32_t* __restrict__ act, size_t m, int spec,
                                                    const uint16_t* __restrict__ kappa, int l,
                                                    uint32_t* __restrict__ slot_op, uint16_t* __restrict__ slot_kappa,
                                                    const uint32_t* __restrict__ key_idx, uint32_t* __restrict__ slot_key,
                                                    uint32_t* __restrict__ counter) {
    const size_t sidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (sidx == 0) *counter = 0;  // k_compact of this round counts the survivors from zero
    if (sidx >= m * (size_t)spec) return;
    const uint32_t op = act[sidx / spec];
    slot_op[sidx] = op;
    if (slot_key) slot_key[sidx] = key_idx ? key_idx[op] : op;  // row of a per-key A_hat table
    slot_kappa[sidx] = (uint16_t)(kappa[op] + (uint32_t)(sidx % spec) * l);
}

__global__ __launch_bounds__(64) void k_resolve(const uint32_t* __restrict__ act, int spec, const int32_t* __restrict__ accept,
                                                const uint8_t* __restrict__ stage, size_t stage_stride,
                                                uint8_t* __restrict__ sigs, size_t sig_len, int32_t* __restrict__ done,
                                                uint16_t* __restrict__ kappa, int l) {
    const size_t i = blockIdx.x;
    const uint32_t op = act[i];
    const int lane = threadIdx.x;
    const int mine = lane < spec ? accept[i * spec + lane] : 0;
    const unsigned long long mask = __ballot(mine != 0);
    if (mask == 0) {
        if (lane == 0) kappa[op] = (uint16_t)(kappa[op] + spec * l);
        return;
    }
    const int j = __ffsll((long long)mask) - 1;
    const uint8_t* src = stage + (i * spec + j) * stage_stride;
    uint8_t* dst = sigs + (size_t)op * sig_len;
    for (size_t bidx = lane; bidx < sig_len; bidx += 64) dst[bidx] = src[bidx];
    if (lane == 0) done[op] = 1;
}

// keep the unfinished ops for the next round (order is irrelevant: ops are independent)
__global__ __launch_bounds__(256) void k_compact(const uint32_t* __restrict__ act_in, size_t n, const int32_t* __restrict__ done,
                                                 uint32_t* __restrict__ act_out, uint32_t* __restrict__ counter) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t op = act_in[i];
    if (!done[op]) act_out[atomicAdd(counter, 1u)] = op;
}

// first active list: every op except those whose ctx is too long (lib.rs:274)
__global__ __launch_bounds__(256) void k_init_active(size_t n, const int32_t* __restrict__ ctx_bad, int32_t* __restrict__ done,
                                                     uint16_t* __restrict__ kappa, int32_t* __restrict__ status,
                                                     uint32_t* __restrict__ act_out, uint32_t* __restrict__ counter) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    kappa[i] = 0;  // ml_dsa.rs:204
    const int bad = ctx_bad[i];
    done[i] = bad;
    if (status) status[i] = bad ? MLDSA_ERR_CTX_LEN : MLDSA_OK;
    if (!bad) act_out[atomicAdd(counter, 1u)] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------
// keygen tail (ml_dsa.rs:88-101 + encodings.rs:18-40, 94-152): t = inv_ntt(A s1_hat) + s2,
// (t1, t0) = power2round(t); pk = rho | SimpleBitPack(t1, 10 bits); and the s1 / s2 / t0
// sections of sk.  One wave per polynomial, 4 consecutive coefficients per lane.
__device__ __forceinline__ void store_fields(uint8_t* dst, const uint32_t f[4], int bits, int lane) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint64_t)f[i] << (i * bits);
    const int nbytes = bits / 2;  // 4 * bits / 8; bits in {3, 4, 10, 13} -> handle odd sizes below
    if (bits == 3) {  // 12 bits per lane: pair lanes -> 3 bytes per 2 lanes
        const uint32_t other = __shfl_xor((uint32_t)v, 1);
        if (!(lane & 1)) {
            const uint32_t both = (uint32_t)v | (other << 12);
            uint8_t* d = dst + (lane >> 1) * 3;
            d[0] = (uint8_t)both; d[1] = (uint8_t)(both >> 8); d[2] = (uint8_t)(both >> 16);
        }
    } else if (bits == 13) {  // 52 bits per lane: pair lanes -> 13 bytes per 2 lanes
        const uint64_t other = __shfl_xor((unsigned long long)v, 1);
        if (!(lane & 1)) {
            uint8_t* d = dst + (lane >> 1) * 13;
            const uint64_t lo = v | (other << 52);
            const uint64_t hi = other >> 12;
            for (int i = 0; i < 8; i++) d[i] = (uint8_t)(lo >> (8 * i));
            for (int i = 0; i < 5; i++) d[8 + i] = (uint8_t)(hi >> (8 * i));
        }
    } else {
        uint8_t* d = dst + lane * nbytes;
        for (int i = 0; i < nbytes; i++) d[i] = (uint8_t)(v >> (8 * i));
    }
}

// polys: s1s2[key][L + K] (ExpandS output), as1[key][K] = inv_ntt(A * ntt(s1)) canonical
// seeds[key] = rho (32) | rho' (64) | K (32): rho opens pk and sk, K follows in sk (encodings.rs:27-31, 118-121)
__global__ __launch_bounds__(GBLOCK) void k_keygen_encode(const int32_t* __restrict__ s1s2, const int32_t* __restrict__ as1,
                                                          const uint8_t* __restrict__ seeds,
                                                          uint8_t* __restrict__ pk, uint8_t* __restrict__ sk, int k, int l,
                                                          int eta, size_t pk_len, size_t sk_len, size_t n_keys) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * GWAVES + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * GWAVES;
    const int ebits = eta == 2 ? 3 : 4;
    const size_t per_key = (size_t)(l + 2 * k);  // l + k eta-polys, k t-polys
    for (size_t p = wave; p < n_keys * per_key; p += n_waves) {
        const size_t key = p / per_key;
        const int j = (int)(p % per_key);
        uint8_t* skp = sk + key * sk_len;
        if (j == 0 && lane < 32) {
            const uint8_t r = seeds[key * 128 + lane];
            pk[key * pk_len + lane] = r;
            skp[lane] = r;
            skp[32 + lane] = seeds[key * 128 + 96 + lane];
        }
        if (j < l + k) {  // skEncode: BitPack(s, eta, eta): field = eta - s   (encodings.rs:118-134)
            const Coef4 s = ld4(s1s2 + (key * (l + k) + j) * (size_t)N, lane);
            uint32_t f[4];
#pragma unroll
            for (int i = 0; i < 4; i++) f[i] = (uint32_t)(eta - s.v[i]);
            store_fields(skp + 128 + (size_t)j * (32 * ebits), f, ebits, lane);
        } else {
            const int i = j - (l + k);
            const Coef4 a = ld4(as1 + (key * k + i) * (size_t)N, lane);
            const Coef4 s2 = ld4(s1s2 + (key * (l + k) + l + i) * (size_t)N, lane);
            uint32_t f1[4], f0[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int32_t tt = freeze(a.v[c] + s2.v[c]);          // ml_dsa.rs:88-91
                const int32_t r1 = (tt + (1 << 12) - 1) >> 13;        // power2round, high_low.rs:26-31
                const int32_t r0 = tt - (r1 << 13);
                f1[c] = (uint32_t)r1;
                f0[c] = (uint32_t)((1 << 12) - r0);                   // BitPack(t0, 2^12 - 1, 2^12)
            }
            store_fields(pk + key * pk_len + 32 + (size_t)i * 320, f1, 10, lane);
            store_fields(skp + 128 + (size_t)(l + k) * (32 * ebits) + (size_t)i * 416, f0, 13, lane);
        }
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

int launch_sign_tail(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* c, const int32_t* y, const int32_t* w, const uint8_t* ctilde,
                     const uint32_t* slot_op, const uint32_t* key_idx, const int32_t* s1, const int32_t* s2, const int32_t* t0,
                     uint16_t* kappa, int32_t* done, uint8_t* sigs, int spec, uint8_t* stage, size_t stage_stride, int32_t* accept,
                     size_t n_slots, hipStream_t s, const uint8_t* wrisk, const uint8_t* yrisk) {
    if (n_slots == 0) return MLDSA_OK;
    const int gb = p->gamma1 == (1 << 17) ? 17 : 19;
    dim3 grid(grid_for(ctx, n_slots, GWAVES, 16));  // more, shorter blocks than fit at once: the dispatcher evens out the early exits
#define MLDSA_TAIL(KK, LL, G2)                                                                                              \
    hipLaunchKernelGGL((k_sign_tail<KK, LL, G2>), grid, dim3(64 * GWAVES), 0, s, c, y, w, ctilde, slot_op, key_idx, s1, s2, t0, kappa, \
                       done, sigs, spec, stage, stage_stride, accept, gb, p->beta, p->omega, p->ctilde_len, (size_t)p->sig_len,   \
                       n_slots, ctx->d_inv_tw, wrisk, yrisk)
    if (p->set == MLDSA_44) MLDSA_TAIL(4, 4, false);
    else if (p->set == MLDSA_65) MLDSA_TAIL(6, 5, true);
    else MLDSA_TAIL(8, 7, true);
#undef MLDSA_TAIL
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_make_slots(mldsa_ctx*, const uint32_t* act, size_t m, int spec, const uint16_t* kappa, int l, uint32_t* slot_op,
                      uint16_t* slot_kappa, hipStream_t s, const uint32_t* key_idx, uint32_t* slot_key, uint32_t* counter) {
    if (m == 0) return MLDSA_OK;
    const size_t n = m * (size_t)spec;
    hipLaunchKernelGGL(k_make_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, act, m, spec, kappa, l, slot_op, slot_kappa, key_idx,
                       slot_key, counter);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_resolve(mldsa_ctx*, const mldsa_params* p, const uint32_t* act, size_t m, int spec, const int32_t* accept,
                   const uint8_t* stage, size_t stage_stride, uint8_t* sigs, int32_t* done, uint16_t* kappa, hipStream_t s) {
    if (m == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)m), dim3(64), 0, s, act, spec, accept, stage, stage_stride, sigs,
                       (size_t)p->sig_len, done, kappa, p->l);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_compact(mldsa_ctx*, const uint32_t* act_in, size_t n, const int32_t* done, uint32_t* act_out, uint32_t* counter,
                   hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, act_in, n, done, act_out, counter);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_init_active(mldsa_ctx*, size_t n, const int32_t* ctx_bad, int32_t* done, uint16_t* kappa, int32_t* status,
                       uint32_t* act_out, uint32_t* counter, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_init_active, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ctx_bad, done, kappa, status, act_out,
                       counter);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_keygen_encode(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* s1s2, const int32_t* as1, const uint8_t* seeds,
                         uint8_t* pk, uint8_t* sk, size_t n_keys, hipStream_t s) {
    if (n_keys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_keygen_encode, dim3(grid_for(ctx, n_keys * (size_t)(p->l + 2 * p->k), GWAVES, 8)), dim3(GBLOCK), 0, s, s1s2,
                       as1, seeds, pk, sk, p->k, p->l, p->eta, (size_t)p->pk_len, (size_t)p->sk_len, n_keys);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
