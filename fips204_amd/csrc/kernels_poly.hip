// Polynomial-arithmetic kernels for gfx950: batched NTT / inverse NTT, K x L pointwise
// matrix-vector product, scalar-vector product, infinity norm, and the fused
// verify-arithmetic unit.  All integer (no MFMA); every kernel is a streaming kernel whose
// roofline is HBM (DESIGN.md "Kernels").
//
// Replaces src/ntt.rs (ntt, inv_ntt) and the poly helpers of src/helpers.rs (mat_vec_mul,
// to_mont, add_vector_ntt, infinity_norm) of the reference.
#include <cstdlib>

#include <type_traits>

#include "ctx.h"
#include "sampler_dev.h"
#include "rounding.h"

namespace mldsa {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

// ------------------------------------------------------------------ NTT (ntt.rs:14-76)
__global__ __launch_bounds__(BLOCK) void k_ntt(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                               size_t n_polys, const Twiddle *__restrict__ tab, const uint32_t *__restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    if (n_dev) n_polys = *n_dev;  // the signer's rounds: count known only on the device
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    FwdTw tw;
    load_fwd_tw(tw, tab, lane);
    // software prefetch: the next polynomial's loads are in flight while this one is transformed
    // (1 KiB per wave in flight is too little to cover HBM latency at 32 waves per CU)
    int32_t nxt[4] = {0, 0, 0, 0};
    if (wave < n_polys) load_strided(nxt, in + wave * N, lane);
    for (size_t p = wave; p < n_polys; p += n_waves) {
        int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        if (p + n_waves < n_polys) load_strided(nxt, in + (p + n_waves) * N, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = reduce32(r[k]);
        ntt_fwd_wave(r, tw, lane);
        store_packed(r, out + p * N, lane);
    }
}

// c_hat = NTT(c) for the signer's rounds: c arrives as k_sample_in_ball<.., C8>'s 256 bytes per row (the lane's four coefficients
// in its dword), c_hat leaves as int32[256] like k_ntt's output
__global__ __launch_bounds__(BLOCK) void k_ntt_c8(const uint32_t *__restrict__ in, int32_t *__restrict__ out, size_t n_polys,
                                                  const Twiddle *__restrict__ tab, const uint32_t *__restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    if (n_dev) n_polys = *n_dev;
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    if constexpr (EXP_PRIO != 0) __builtin_amdgcn_s_setprio(EXP_PRIO);
    FwdTw tw;
    load_fwd_tw(tw, tab, lane);
    uint32_t nxt = 0;
    if (wave < n_polys) nxt = in[wave * 64 + lane];
    for (size_t p = wave; p < n_polys; p += n_waves) {
        const uint32_t d = nxt;
        if (p + n_waves < n_polys) nxt = in[(p + n_waves) * 64 + lane];
        int32_t r[4] = {(int8_t)(d & 0xFF), (int8_t)((d >> 8) & 0xFF), (int8_t)((d >> 16) & 0xFF), (int8_t)(d >> 24)};
        ntt_fwd_wave(r, tw, lane);
        store_packed(r, out + p * N, lane);
    }
}

// -------------------------------------------------------- inverse NTT (ntt.rs:85-161)
__global__ __launch_bounds__(BLOCK) void k_inv_ntt(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                   size_t n_polys, const Twiddle *__restrict__ tab) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    InvTw tw;
    load_inv_tw(tw, tab, lane);
    int32_t nxt[4] = {0, 0, 0, 0};
    if (wave < n_polys) load_packed(nxt, in + wave * N, lane);
    for (size_t p = wave; p < n_polys; p += n_waves) {
        int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        if (p + n_waves < n_polys) load_packed(nxt, in + (p + n_waves) * N, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = reduce32(r[k]);
        ntt_inv_wave(r, tw, lane, F_MONT);
        store_strided(r, out + p * N, lane);
    }
}

// ------------------------------------------------------ element-wise helpers (16 B/lane)
template <int OP>  // 0: to_mont (helpers.rs:131-135)   1: add (helpers.rs:125-127)   2 / 3 / 4: partial_reduce32 / full_reduce32 / center_mod (61-95)
__global__ __launch_bounds__(BLOCK) void k_elementwise(const int4 *__restrict__ a, const int4 *__restrict__ b,
                                                       int4 *__restrict__ out, size_t n_vec) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (; i < n_vec; i += stride) {
        int4 x = a[i];
        if constexpr (OP == 0) {
            out[i] = make_int4(to_mont(x.x), to_mont(x.y), to_mont(x.z), to_mont(x.w));
        } else if constexpr (OP == 2) {
            out[i] = make_int4(reduce32(x.x), reduce32(x.y), reduce32(x.z), reduce32(x.w));
        } else if constexpr (OP == 3) {
            out[i] = make_int4(freeze(x.x), freeze(x.y), freeze(x.z), freeze(x.w));
        } else if constexpr (OP == 4) {
            out[i] = make_int4(center(x.x), center(x.y), center(x.z), center(x.w));
        } else {
            int4 y = b[i];
            out[i] = make_int4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        }
    }
}

// ------------------------------------------ rounding seams (high_low.rs), element-wise
// OP: 0 power2round (15-48), 1 decompose (66-96), 2 high_bits (104-111), 3 low_bits (119-126), 4 make_hint (134-144), 5 use_hint (155-192)
template <bool G2HI>
__global__ __launch_bounds__(BLOCK) void k_rounding(int op, const int32_t *__restrict__ a, const int32_t *__restrict__ b,
                                                    int32_t *__restrict__ out1, int32_t *__restrict__ out2, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (; i < n; i += stride) {
        const int32_t x = a[i];
        int32_t r1 = 0, r0 = 0;
        if (op == 0) {                      // input in [0, q) (the reference's debug_assert, high_low.rs:21-24)
            r1 = (x + (1 << 12) - 1) >> 13;
            r0 = x - (r1 << 13);
        } else if (op <= 3) {
            decompose<G2HI>(freeze(x), r1, r0);   // decompose starts with full_reduce32 (high_low.rs:76)
            if (op == 3) r1 = r0;
        } else if (op == 4) {               // a = z, b = r
            int32_t v1, t;
            decompose<G2HI>(freeze(b[i]), r1, t);
            decompose<G2HI>(freeze(b[i] + x), v1, t);
            r1 = r1 != v1 ? 1 : 0;
        } else {                            // a = h, b = r
            r1 = use_hint<G2HI>(x, freeze(b[i]));
        }
        out1[i] = r1;
        if (op <= 1 && out2) out2[i] = r0;
    }
}

int launch_rounding(mldsa_ctx *ctx, const mldsa_params *p, int op, const int32_t *a, const int32_t *b, int32_t *out1, int32_t *out2, size_t n_polys,
                    hipStream_t s) {
    if (n_polys == 0) return MLDSA_OK;
    const size_t n = n_polys * N;
    const dim3 grid(grid_for(ctx, n, BLOCK, 8)), block(BLOCK);
    if (p->gamma2 == (Q - 1) / 32) hipLaunchKernelGGL(k_rounding<true>, grid, block, 0, s, op, a, b, out1, out2, n);
    else hipLaunchKernelGGL(k_rounding<false>, grid, block, 0, s, op, a, b, out1, out2, n);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ------------------------------------------ mat_vec_mul (helpers.rs:100-114), one wave per row
// w_hat[op][i] = sum_j a_hat[op][i][j] o u_hat[op][j];  u is converted with to_mont first,
// exactly as the reference does, so each term is mont_reduce(a * u * 2^32) = a*u in (-q, q).
template <int K, int L>
__global__ __launch_bounds__(BLOCK) void k_mat_vec_mul(const int32_t *__restrict__ a_hat,
                                                       const int32_t *__restrict__ u_hat,
                                                       int32_t *__restrict__ w_hat, size_t n_ops) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    const size_t n_rows = n_ops * K;
    for (size_t row = wave; row < n_rows; row += n_waves) {
        const size_t op = row / K;
        const int32_t *a = a_hat + row * (size_t)L * N;
        const int32_t *u = u_hat + op * (size_t)L * N;
        int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < L; j++) {
            int32_t av[4], uv[4];
            load_packed(av, a + j * N, lane);
            load_packed(uv, u + j * N, lane);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] += mont_mul(av[k], to_mont(reduce32(uv[k])));
        }
        store_packed(acc, w_hat + row * N, lane);
    }
}

// ------------------------- c_hat o v_hat_mont (ml_dsa.rs:243-250 / 253-260 / 288-295)
__global__ __launch_bounds__(BLOCK) void k_pointwise_mont(const int32_t *__restrict__ c_hat,
                                                          const int32_t *__restrict__ v,
                                                          int32_t *__restrict__ out, size_t ppo, size_t n_ops) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    const size_t n_polys = n_ops * ppo;
    for (size_t p = wave; p < n_polys; p += n_waves) {
        int32_t cv[4], vv[4], r[4];
        load_packed(cv, c_hat + (p / ppo) * N, lane);
        load_packed(vv, v + p * N, lane);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = mont_mul(reduce32(cv[k]), vv[k]);
        store_packed(r, out + p * N, lane);
    }
}

// ------------------------------------------------ infinity_norm (helpers.rs:138-147)
__global__ __launch_bounds__(BLOCK) void k_infinity_norm(const int32_t *__restrict__ polys, size_t ppo,
                                                         size_t n_ops, int32_t *__restrict__ norms) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * WAVES_PER_BLOCK;
    for (size_t op = wave; op < n_ops; op += n_waves) {
        int32_t mx = 0;
        for (size_t p = 0; p < ppo; p++) {
            int32_t v[4];
            load_packed(v, polys + (op * ppo + p) * N, lane);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int32_t c = center(v[k]);
                c = c < 0 ? -c : c;
                mx = c > mx ? c : mx;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            int32_t o = __shfl_xor(mx, m);
            mx = o > mx ? o : mx;
        }
        if (lane == 0) norms[op] = mx;
    }
}

// -------------------------------------------------------------------------------------
// Fused verify-arithmetic unit (ml_dsa.rs:407-416):
//   w' = inv_ntt( A_hat * ntt(z)  -  ntt(c) o t1_d2_hat_mont )
// ONE WAVE PER OPERATION: the wave runs the L (+1) forward transforms one after the other and keeps
// z_hat (and c_hat) in lane-private LDS rows -- each lane reads back only what it wrote, so there is
// no barrier anywhere in the loop -- then walks the K rows of A_hat: row i + 1 is requested as soon
// as row i has been multiplied, so its HBM latency hides under row i's inverse NTT (row 0 is
// requested before the forward transforms).  Four waves (= four ops in flight) share the block's
// twiddle copy.  (A workgroup-per-op layout with one wave per polynomial and two barriers per op
// was 6-8 % slower at batch 65 536: this one streams at 95 % of the bandwidth the box delivers for
// the 6:1 read:write mix, profiles/r01_ubench_hbm.txt.)
// Domain bookkeeping: products mont_mul(a, z_hat) carry a factor 2^-32, so c_hat is stored
// as c_hat * 2^-32 (then mont_mul(c_hat', t1 * 2^32) carries the same factor) and the
// inverse NTT finishes with F_MONT2 = 256^-1 * 2^64.
// HBM traffic per op = algorithmic bytes: (K*L + L + 1 + K) KiB in, K KiB out.
// With HAS_C = false the kernel is the signer's w = inv_ntt(A_hat * ntt(y)) (ml_dsa.rs:218-222);
// a_idx then maps a candidate slot to the A_hat it uses.
// W1 = 1 / 2 additionally emits w1Encode(HighBits(w)) (6-bit / 4-bit fields) per op: the signer's
// commitment bytes (ml_dsa.rs:225-232), so no separate pass re-reads w.
// KG = true is key generation's t = A s1 + s2 (ml_dsa.rs:86-92): s1 and s2 are rows of one (L + K)-polynomial vector per key
// (`z`, z_polys_per_op = L + K), and instead of storing A s1 the epilogue adds s2_i, applies Power2Round (high_low.rs:15-48) and
// packs t1 into the key's pk and t0 into its sk (encodings.rs:18-40, 136-152): A s1 and t never travel through HBM.
struct KeygenOut {
    uint8_t *pk, *sk;
    size_t pk_len, sk_len, t0_off;  // t0_off: byte offset of the t0 section inside sk
    int eta, ebits;                 // BitPack(s, eta, eta): `ebits`-wide fields eta - s (the s1 / s2 sections of sk, from byte 128)
};
// YGB != 0 (the signer's rounds): y arrives as ExpandMask's squeezed bytes (field.h y_raw_dword, gamma1 = 2^YGB) and the kernel
// also flags the polynomials of y that can fail ||z||inf < gamma1 - beta (yr.flags: some |y| >= yr.bound = gamma1 - 2 beta).
struct YRisk {
    uint8_t *flags;
    int32_t bound;
};
constexpr int AW = 4;  // waves per block
template <int K, int L, bool HAS_C, int W1 = 0, bool APACK = false, bool KG = false, int YGB = 0>
__global__ __launch_bounds__(64 * AW) void k_verify_arith(
    const int32_t *__restrict__ a_hat, const uint32_t *__restrict__ a_idx, const int32_t *__restrict__ z,
    const int32_t *__restrict__ c, const int32_t *__restrict__ t1, const uint32_t *__restrict__ key_idx,
    int32_t *__restrict__ w_out, size_t n_ops, const Twiddle *__restrict__ fwd_tab, const Twiddle *__restrict__ inv_tab,
    uint8_t *__restrict__ w1, size_t w1_stride, size_t z_polys_per_op, uint8_t *__restrict__ wrisk, int32_t risk_bound,
    const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ z_idx, KeygenOut kg, YRisk yr) {
    constexpr int NZ = HAS_C ? L + 1 : L;
    constexpr bool YRAW = YGB != 0;
    constexpr int YCB = YGB + 1;
    __shared__ int4 zh[AW][NZ][64];
    __shared__ int32_t xp_kg[KG ? AW : 1][KG ? N : 1];  // strided -> four consecutive coefficients per lane (t1 / t0 packing)
    __shared__ Twiddle tw_lds[(FWD_TW + INV_TW) * 64];
    // (experiment, MLDSA_EXP bit 3: the A_hat row of the signer's kernel arrives by LDS-DMA, one buffer per wave, instead of in registers)
    constexpr bool DMA = EXP_LDSDMA && APACK && !HAS_C && !KG && W1 == 2 && K == 6;  // (the ML-DSA-65 signer only)
    constexpr bool NT_A = HAS_C ? NT_A_VERIFY : KG ? NT_A_KG : EXP_NT_A_SIGN;
    constexpr int ROW_BYTES = L * PACKED_POLY_DWORDS * 4, ROW_PIECES = (ROW_BYTES + 1023) / 1024;
    constexpr int DMA_STORES = 3 + (W1 == 2 ? 1 : 12);  // store instructions of one row's epilogue (w planes + w1Encode), issued behind the next row's DMA
    __shared__ uint32_t a_lds[DMA ? AW : 1][DMA ? ROW_PIECES * 256 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n_dev) n_ops = *n_dev;  // the signer's rounds: slots of this round, known only on the device
    if ((size_t)blockIdx.x * AW >= n_ops) return;
    for (int i = threadIdx.x; i < FWD_TW * 64; i += 64 * AW) tw_lds[i] = fwd_tab[i];
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * AW) tw_lds[FWD_TW * 64 + i] = inv_tab[i];
    __syncthreads();
    const LdsTw ftw{tw_lds, lane};
    const LdsTw itw{tw_lds + FWD_TW * 64, lane};
    const uint32_t wid = blockIdx.x * AW + wave, n_waves = gridDim.x * AW;

    // (the signer's rounds that generate two candidates per op: rows 2 p and 2 p + 1 use the same A_hat and go to two waves of one
    //  block, which read it at the same time and share the fetch in their XCD's L2; one wave taking both rows in turn was measured
    //  and is slower: sign_w 2.28-2.43 instead of 2.11-2.21 ms per ML-DSA-65 signing step)
    for (size_t op = wid; op < n_ops; op += n_waves) {
        const size_t aop = a_idx ? a_idx[op] : op;
        const size_t key = HAS_C ? (key_idx ? key_idx[op] : op) : 0;
        // APACK: A_hat in the pipelines' 24-bit form (768 bytes per polynomial, three dwords per lane)
        using ARow = std::conditional_t<APACK, Packed3, int4>;
        const ARow *arow = APACK ? reinterpret_cast<const ARow *>(reinterpret_cast<const uint32_t *>(a_hat) + (aop * K * (size_t)L) * PACKED_POLY_DWORDS)
                                 : reinterpret_cast<const ARow *>(a_hat + (aop * K * (size_t)L) * N);
        auto coeffs = [](const ARow &v) -> int4 {
            if constexpr (APACK) return unpack24(v); else return v;
        };
        // row 0 of A_hat (and of t1) is requested before the transforms
        ARow av[L];
        int4 tv = make_int4(0, 0, 0, 0);
        // row `i` of this op's A_hat -> the wave's LDS buffer: ROW_PIECES wave-instructions of 1 KiB, the last one EXEC-masked to the row
        auto dma_row = [&](int i) {
            const uint8_t *src = reinterpret_cast<const uint8_t *>(arow) + (size_t)i * ROW_BYTES;
#if defined(__HIP_DEVICE_COMPILE__)  // (gfx950 builtins and inline assembly: not for the host pass of hipcc)
#pragma unroll
            for (int t = 0; t < ROW_PIECES; t++)
                if (t * 1024 + lane * 16 < ROW_BYTES)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(src + t * 1024 + lane * 16), &a_lds[wave][t * 256], 16, 0, EXP_NT_DMA ? 2 : 0);
#else
            (void)src; (void)a_lds;
#endif
        };
        if constexpr (DMA) {
            dma_row(0);
        } else {
#pragma unroll
            for (int j = 0; j < L; j++) av[j] = load_row<NT_A>(&arow[j * 64 + lane]);
        }
        if constexpr (HAS_C) tv = reinterpret_cast<const int4 *>(t1 + (key * K) * (size_t)N)[lane];
        // ---- forward transforms, next polynomial loaded one ahead
        int32_t nr[4];
        const size_t zrow = (z_idx ? (size_t)z_idx[op] : op) * z_polys_per_op;  // first polynomial of the op's z / y vector
        auto load_z = [&](size_t poly) {
            if constexpr (KG) {  // ExpandS's byte rows (k_expand_s<.., S8>): coefficient i = byte i
                const uint32_t *src = reinterpret_cast<const uint32_t *>(z) + poly * (size_t)(N / 4);
#pragma unroll
                for (int k = 0; k < 4; k++) nr[k] = (int32_t)src[16 * k + (lane >> 2)];  // the dword that holds coefficient 64 k + lane
            } else if constexpr (YRAW) {
                const uint8_t *src = reinterpret_cast<const uint8_t *>(z) + poly * (size_t)(32 * YCB);
#pragma unroll
                for (int k = 0; k < 4; k++) nr[k] = (int32_t)y_raw_dword<YCB>(src, k, lane);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) nr[k] = load_once<NT_ZC && HAS_C>(z + poly * (size_t)N + 64 * k + lane);
            }
        };
        load_z(zrow);
#pragma unroll 1
        for (int j = 0; j < NZ; j++) {
            asm volatile("" ::: "memory");  // keep the LDS twiddle reads at their point of use (no hoisting into registers)
            int32_t r[4];
            if constexpr (YRAW) {
                bool near = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    r[k] = y_from_raw<YCB>((uint32_t)nr[k], lane);  // |y| <= gamma1 < q: no reduction
                    near |= (r[k] < 0 ? -r[k] : r[k]) >= yr.bound;
                }
                const bool any_near = __ballot(near) != 0ull;
                if (yr.flags && lane == 0) yr.flags[zrow + j] = any_near ? 1 : 0;
            } else if constexpr (KG) {
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = (int32_t)(int8_t)((uint32_t)nr[k] >> (8 * (lane & 3)));
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = reduce32(nr[k]);
            }
            if constexpr (KG) {  // s1_j passes through here: its section of sk (encodings.rs:118-134) is packed on the way,
                                 // from the lane's four CONSECUTIVE coefficients = dword `lane` of the byte row
                const uint32_t d = reinterpret_cast<const uint32_t *>(z)[(zrow + j) * (size_t)(N / 4) + lane];
                const uint32_t f[4] = {(uint32_t)(kg.eta - (int8_t)(d & 0xFF)), (uint32_t)(kg.eta - (int8_t)((d >> 8) & 0xFF)),
                                       (uint32_t)(kg.eta - (int8_t)((d >> 16) & 0xFF)), (uint32_t)(kg.eta - (int8_t)(d >> 24))};
                store_fields(kg.sk + op * kg.sk_len + 128 + (size_t)j * (32 * kg.ebits), f, kg.ebits, lane);
            }
            if (j + 1 < L) load_z(zrow + j + 1);
            else if (HAS_C && j + 1 == L) {
#pragma unroll
                for (int k = 0; k < 4; k++) nr[k] = load_once<NT_ZC>(c + op * (size_t)N + 64 * k + lane);
            }
            ntt_fwd_wave(r, ftw, lane);
            if (HAS_C && j == L) {
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = mont_mul(r[k], 1);  // c_hat * 2^-32
            }
            zh[wave][j][lane] = make_int4(r[0], r[1], r[2], r[3]);
        }
        // ---- rows
        uint32_t risk = 0;
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            asm volatile("" ::: "memory");
            // 64-bit accumulation, one Montgomery reduction per coefficient and row (field.h): |a| < 2^24 (what ExpandA
            // produces), |z_hat| < 9 q, at most L + 1 <= 8 terms: |sum| < 2^54 = the reduction's input bound
            int64_t acc64[4] = {0, 0, 0, 0};
            if constexpr (DMA) {  // the row's DMA has landed once nothing but the previous row's DMA_STORES stores is outstanding (VM operations retire in order)
                if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_STORES) : "memory");
            }
            if constexpr (DMA) {
                // read by hand: a compiler-visible LDS read of the DMA's buffer makes hipcc wait vmcnt(0) in front of it, i.e. for the
                // previous row's stores as well
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&a_lds[wave][lane * 3];
#pragma unroll
                for (int j = 0; j < L; j++)
                    asm volatile("ds_read_b32 %0, %3 offset:%4\n\tds_read_b32 %1, %3 offset:%5\n\tds_read_b32 %2, %3 offset:%6"
                                 : "=&v"(av[j].a), "=&v"(av[j].b), "=&v"(av[j].c)
                                 : "v"(la), "n"(j * 768), "n"(j * 768 + 4), "n"(j * 768 + 8)
                                 : "memory");
#pragma unroll
                for (int j = 0; j < L; j++) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[j].a), "+v"(av[j].b), "+v"(av[j].c)::"memory");
#endif
            }
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int4 zv = zh[wave][j][lane];
                const int4 a4 = coeffs(av[j]);
                acc64[0] += (int64_t)a4.x * zv.x;
                acc64[1] += (int64_t)a4.y * zv.y;
                acc64[2] += (int64_t)a4.z * zv.z;
                acc64[3] += (int64_t)a4.w * zv.w;
            }
            if constexpr (HAS_C) {
                const int4 cv = zh[wave][L][lane];  // c_hat * 2^-32 in (-q, q); t1 in Montgomery form: |product| < 2^46 ...
                acc64[0] -= (int64_t)cv.x * reduce32(tv.x);  // ... for any t1 representative the seam contract allows
                acc64[1] -= (int64_t)cv.y * reduce32(tv.y);
                acc64[2] -= (int64_t)cv.z * reduce32(tv.z);
                acc64[3] -= (int64_t)cv.w * reduce32(tv.w);
            }
            int32_t acc[4];
            if (i + 1 < K) {  // next row: in flight during this row's inverse transform
                if constexpr (DMA) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this row has been read out of the buffer
                    dma_row(i + 1);
                } else {
#pragma unroll
                    for (int j = 0; j < L; j++) av[j] = load_row<NT_A>(&arow[((i + 1) * L + j) * 64 + lane]);
                }
                if constexpr (HAS_C) tv = reinterpret_cast<const int4 *>(t1 + (key * K + i + 1) * (size_t)N)[lane];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = mont_reduce64(acc64[k]);  // (-q, q): the inverse transform's input range
            ntt_inv_wave(acc, itw, lane, F_MONT2);
            if constexpr (KG) {
                int32_t s2v[4];
                const uint32_t *s2row = reinterpret_cast<const uint32_t *>(z) + (zrow + L + i) * (size_t)(N / 4);  // s2_i: bytes in [-eta, eta]
#pragma unroll
                for (int k = 0; k < 4; k++) s2v[k] = (int32_t)(int8_t)(s2row[16 * k + (lane >> 2)] >> (8 * (lane & 3)));
                {   // ... and its section of sk
                    const uint32_t d = s2row[lane];
                    const uint32_t f[4] = {(uint32_t)(kg.eta - (int8_t)(d & 0xFF)), (uint32_t)(kg.eta - (int8_t)((d >> 8) & 0xFF)),
                                           (uint32_t)(kg.eta - (int8_t)((d >> 16) & 0xFF)), (uint32_t)(kg.eta - (int8_t)(d >> 24))};
                    store_fields(kg.sk + op * kg.sk_len + 128 + (size_t)(L + i) * (32 * kg.ebits), f, kg.ebits, lane);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) xp_kg[wave][64 * k + lane] = freeze(acc[k] + s2v[k]);  // t = A s1 + s2 (ml_dsa.rs:88-91)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int4 t4 = reinterpret_cast<const int4 *>(&xp_kg[wave][0])[lane];
                const int32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
                uint32_t f1[4], f0[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int32_t r1 = (tt[c] + (1 << 12) - 1) >> 13;       // power2round, high_low.rs:26-31
                    const int32_t r0 = tt[c] - (r1 << 13);
                    f1[c] = (uint32_t)r1;
                    f0[c] = (uint32_t)((1 << 12) - r0);                     // BitPack(t0, 2^12 - 1, 2^12)
                }
                store_fields(kg.pk + op * kg.pk_len + 32 + (size_t)i * 320, f1, 10, lane);
                store_fields(kg.sk + op * kg.sk_len + kg.t0_off + (size_t)i * 416, f0, 13, lane);
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            if constexpr (W1 != 0) {
                // the signer's w is read back by k_sign_tail / k_resolve only: 24-bit fields (canonical coefficients < 2^23), the
                // lane's four strided coefficients in three dwords, dword t of every lane in plane t (64 dwords) -- 768 bytes per
                // polynomial instead of 1 024, stored and loaded like the first three quarters of a strided polynomial
                const Packed3 pw = pack24((uint32_t)acc[0], (uint32_t)acc[1], (uint32_t)acc[2], (uint32_t)acc[3]);
                uint32_t *wp = reinterpret_cast<uint32_t *>(w_out) + (op * K + i) * (size_t)PACKED_POLY_DWORDS;
                store_row(wp + lane, pw.a);
                store_row(wp + 64 + lane, pw.b);
                store_row(wp + 128 + lane, pw.c);
            } else {
                int32_t *wq = w_out + (op * K + i) * (size_t)N;
#pragma unroll
                for (int k = 0; k < 4; k++) store_row<EXP_NT_STORE>(wq + 64 * k + lane, acc[k]);
            }
            if constexpr (W1 != 0) {
                constexpr bool G2HI = W1 == 2;
                uint8_t *dst = w1 + op * w1_stride + (size_t)i * (32 * (G2HI ? 4 : 6));
                uint32_t hb[4];
                bool near = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t r1, r0;
                    decompose<G2HI>(acc[k], r1, r0);  // HighBits = r1 (high_low.rs:104-111)
                    hb[k] = (uint32_t)r1;
                    near |= (r0 < 0 ? -r0 : r0) >= risk_bound;
                }
                pack_w1_strided<G2HI>(hb, dst, lane);
                // bit i: some |LowBits(w_i)| >= gamma2 - 2 beta.  Only such a polynomial can fail the signer's
                // ||LowBits(w - c s2)|| < gamma2 - beta test (||c s2|| <= beta), see k_sign_tail
                if (__ballot(near) != 0ull) risk |= 1u << i;
            }
        }
        if constexpr (W1 != 0) {
            if (wrisk && lane == 0) wrisk[op] = (uint8_t)risk;
        }
    }
}

// ------------------------------------------------------------------------- launchers
int launch_ntt(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n, hipStream_t s, const uint32_t *n_dev) {
    if (n == 0 && !n_dev) return MLDSA_OK;
    hipLaunchKernelGGL(k_ntt, dim3(grid_for(ctx, n, WAVES_PER_BLOCK, 8)), dim3(BLOCK), 0, s, in, out, n, ctx->d_fwd_tw, n_dev);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_ntt_c8(mldsa_ctx *ctx, const int32_t *c8, int32_t *out, size_t n, hipStream_t s, const uint32_t *n_dev) {
    if (n == 0 && !n_dev) return MLDSA_OK;
    hipLaunchKernelGGL(k_ntt_c8, dim3(grid_for(ctx, n, WAVES_PER_BLOCK, 8)), dim3(BLOCK), 0, s, reinterpret_cast<const uint32_t *>(c8), out, n,
                       ctx->d_fwd_tw, n_dev);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_inv_ntt(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_inv_ntt, dim3(grid_for(ctx, n, WAVES_PER_BLOCK, 8)), dim3(BLOCK), 0, s, in, out, n, ctx->d_inv_tw);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_to_mont(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    size_t n_vec = n * (N / 4);
    hipLaunchKernelGGL(k_elementwise<0>, dim3(grid_for(ctx, n_vec, BLOCK, 8)), dim3(BLOCK), 0, s,
                       reinterpret_cast<const int4 *>(in), (const int4 *)nullptr, reinterpret_cast<int4 *>(out), n_vec);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_reduce(mldsa_ctx *ctx, int kind, const int32_t *in, int32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    const size_t n_vec = n * (N / 4);
    const dim3 grid(grid_for(ctx, n_vec, BLOCK, 8)), block(BLOCK);
    const int4 *a = reinterpret_cast<const int4 *>(in);
    int4 *o = reinterpret_cast<int4 *>(out);
    if (kind == MLDSA_REDUCE_PARTIAL) hipLaunchKernelGGL(k_elementwise<2>, grid, block, 0, s, a, (const int4 *)nullptr, o, n_vec);
    else if (kind == MLDSA_REDUCE_FULL) hipLaunchKernelGGL(k_elementwise<3>, grid, block, 0, s, a, (const int4 *)nullptr, o, n_vec);
    else if (kind == MLDSA_REDUCE_CENTER) hipLaunchKernelGGL(k_elementwise<4>, grid, block, 0, s, a, (const int4 *)nullptr, o, n_vec);
    else return set_error(MLDSA_ERR_PARAM, "reduce: unknown kind");
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_add(mldsa_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return MLDSA_OK;
    size_t n_vec = n * (N / 4);
    hipLaunchKernelGGL(k_elementwise<1>, dim3(grid_for(ctx, n_vec, BLOCK, 8)), dim3(BLOCK), 0, s,
                       reinterpret_cast<const int4 *>(a), reinterpret_cast<const int4 *>(b), reinterpret_cast<int4 *>(out), n_vec);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_mat_vec_mul(mldsa_ctx *ctx, int k, int l, const int32_t *a, const int32_t *u, int32_t *w, size_t n_ops, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    dim3 grid(grid_for(ctx, n_ops * (size_t)k, WAVES_PER_BLOCK, 8)), block(BLOCK);
    if (k == 4 && l == 4) hipLaunchKernelGGL((k_mat_vec_mul<4, 4>), grid, block, 0, s, a, u, w, n_ops);
    else if (k == 6 && l == 5) hipLaunchKernelGGL((k_mat_vec_mul<6, 5>), grid, block, 0, s, a, u, w, n_ops);
    else if (k == 8 && l == 7) hipLaunchKernelGGL((k_mat_vec_mul<8, 7>), grid, block, 0, s, a, u, w, n_ops);
    else return set_error(MLDSA_ERR_PARAM, "mat_vec_mul: unsupported (K, L)");
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_pointwise_mont(mldsa_ctx *ctx, const int32_t *c, const int32_t *v, int32_t *out, size_t ppo, size_t n_ops, hipStream_t s) {
    if (n_ops == 0 || ppo == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_pointwise_mont, dim3(grid_for(ctx, n_ops * ppo, WAVES_PER_BLOCK, 8)), dim3(BLOCK), 0, s, c, v, out, ppo, n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_infinity_norm(mldsa_ctx *ctx, const int32_t *polys, size_t ppo, size_t n_ops, int32_t *norms, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_infinity_norm, dim3(grid_for(ctx, n_ops, WAVES_PER_BLOCK, 8)), dim3(BLOCK), 0, s, polys, ppo, n_ops, norms);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_verify_arith(mldsa_ctx *ctx, int set, const int32_t *a, const int32_t *z, const int32_t *c, const int32_t *t1,
                        const uint32_t *key_idx, int32_t *w, size_t n_ops, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    const uint32_t *no_idx = nullptr;
    dim3 gw(grid_for(ctx, n_ops, AW, (unsigned)ctx->opt_va_blocks));
    uint8_t *nw1 = nullptr;
    if (set == MLDSA_44) hipLaunchKernelGGL((k_verify_arith<4, 4, true>), gw, dim3(64 * AW), 0, s, a, no_idx, z, c, t1, key_idx, w, n_ops, ctx->d_fwd_tw, ctx->d_inv_tw, nw1, (size_t)0, (size_t)4, nw1, 0, no_idx, no_idx, KeygenOut{}, YRisk{});
    else if (set == MLDSA_65) hipLaunchKernelGGL((k_verify_arith<6, 5, true>), gw, dim3(64 * AW), 0, s, a, no_idx, z, c, t1, key_idx, w, n_ops, ctx->d_fwd_tw, ctx->d_inv_tw, nw1, (size_t)0, (size_t)5, nw1, 0, no_idx, no_idx, KeygenOut{}, YRisk{});
    else if (set == MLDSA_87) hipLaunchKernelGGL((k_verify_arith<8, 7, true>), gw, dim3(64 * AW), 0, s, a, no_idx, z, c, t1, key_idx, w, n_ops, ctx->d_fwd_tw, ctx->d_inv_tw, nw1, (size_t)0, (size_t)7, nw1, 0, no_idx, no_idx, KeygenOut{}, YRisk{});
    else return set_error(MLDSA_ERR_PARAM, "verify_arith: unknown parameter set");
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// w[slot] = inv_ntt(A_hat[a_idx[slot]] * ntt(y[slot]))   (ml_dsa.rs:218-222); with w1 != nullptr also
// w1Encode(HighBits(w)) (ml_dsa.rs:225-232)
int launch_sign_w(mldsa_ctx *ctx, int set, const int32_t *a, const uint32_t *a_idx, const int32_t *y, int32_t *w, uint8_t *w1,
                  size_t w1_stride, size_t n_ops, hipStream_t s, size_t y_polys_per_op, uint8_t *wrisk, bool a_packed,
                  const uint32_t *n_dev, const uint32_t *y_idx, bool y_raw, uint8_t *yrisk) {
    if (n_ops == 0 && !n_dev) return MLDSA_OK;
    const int32_t *none = nullptr;
    const uint32_t *no_idx = nullptr;
    const mldsa_params *pp = params_of(set);
    const int32_t risk_bound = pp ? pp->gamma2 - 2 * pp->beta : 0;
    const YRisk yr{yrisk, pp ? pp->gamma1 - 2 * pp->beta : 0};
    if (y_raw && !w1) return set_error(MLDSA_ERR_PARAM, "sign_w: raw y is the signer's form (with w1)");
    dim3 gw(grid_for(ctx, n_ops, AW, 16));
#define MLDSA_SW3(KK, LL, W1M, AP, YG)                                                                                                    \
    hipLaunchKernelGGL((k_verify_arith<KK, LL, false, W1M, AP, false, YG>), gw, dim3(64 * AW), 0, s, a, a_idx, y, none, none, no_idx, w, n_ops, \
                       ctx->d_fwd_tw, ctx->d_inv_tw, w1, w1_stride, y_polys_per_op ? y_polys_per_op : (size_t)LL, wrisk, risk_bound, n_dev, y_idx, KeygenOut{}, yr)
#define MLDSA_SW(KK, LL, W1M) do { if (a_packed) MLDSA_SW3(KK, LL, W1M, true, 0); else MLDSA_SW3(KK, LL, W1M, false, 0); } while (0)
#define MLDSA_SWR(KK, LL, W1M, YG) do { if (a_packed) MLDSA_SW3(KK, LL, W1M, true, YG); else MLDSA_SW3(KK, LL, W1M, false, YG); } while (0)
    if (y_raw) {
        if (set == MLDSA_44) MLDSA_SWR(4, 4, 1, 17);
        else if (set == MLDSA_65) MLDSA_SWR(6, 5, 2, 19);
        else if (set == MLDSA_87) MLDSA_SWR(8, 7, 2, 19);
        else return set_error(MLDSA_ERR_PARAM, "sign_w: unknown parameter set");
    }
    else if (set == MLDSA_44) { if (w1) MLDSA_SW(4, 4, 1); else MLDSA_SW(4, 4, 0); }
    else if (set == MLDSA_65) { if (w1) MLDSA_SW(6, 5, 2); else MLDSA_SW(6, 5, 0); }
    else if (set == MLDSA_87) { if (w1) MLDSA_SW(8, 7, 2); else MLDSA_SW(8, 7, 0); }
    else return set_error(MLDSA_ERR_PARAM, "sign_w: unknown parameter set");
#undef MLDSA_SWR
#undef MLDSA_SW3
#undef MLDSA_SW
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// key generation's t = A s1 + s2 with Power2Round and the t1 / t0 packing in the epilogue (k_verify_arith<.., KG = true>);
// s1s2[key][L + K] = ExpandS output, a_hat in the pipelines' 24-bit form
int launch_keygen_t(mldsa_ctx *ctx, const mldsa_params *p, const int32_t *a_hat, const int32_t *s1s2, uint8_t *pk, uint8_t *sk, size_t n_keys,
                    hipStream_t s) {
    if (n_keys == 0) return MLDSA_OK;
    const int32_t *none = nullptr;
    const uint32_t *no_idx = nullptr;
    uint8_t *nw1 = nullptr;
    const int eb = p->eta == 2 ? 3 : 4;
    const KeygenOut kg{pk, sk, (size_t)p->pk_len, (size_t)p->sk_len, (size_t)128 + (size_t)(p->l + p->k) * 32 * eb, p->eta, eb};
    dim3 gw(grid_for(ctx, n_keys, AW, 16));
#define MLDSA_KGT(KK, LL)                                                                                                              \
    hipLaunchKernelGGL((k_verify_arith<KK, LL, false, 0, true, true>), gw, dim3(64 * AW), 0, s, a_hat, no_idx, s1s2, none, none, no_idx,   \
                       (int32_t *)nullptr, n_keys, ctx->d_fwd_tw, ctx->d_inv_tw, nw1, (size_t)0, (size_t)(LL + KK), nw1, 0, no_idx, no_idx, kg, YRisk{})
    if (p->set == MLDSA_44) MLDSA_KGT(4, 4);
    else if (p->set == MLDSA_65) MLDSA_KGT(6, 5);
    else MLDSA_KGT(8, 7);
#undef MLDSA_KGT
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
