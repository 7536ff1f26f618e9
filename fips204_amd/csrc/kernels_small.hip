// verify_internal (src/ml_dsa.rs:351-437) for SMALL calls as ONE kernel launch.
//
// A verification call of a few operations is all latency: the batch pipeline (pipeline.hip verify_batch) is six dependent launches on
// three streams -- key check, mu, SampleInBall, ExpandA, the fused arithmetic, the c~ hash -- and its kernel timeline
// (profiles/r05_small_call_timeline_verify_n1.json) shows 100 us of device time for a chain whose kernels add up to 66 us on the
// critical path: ExpandA starts 28 us after the first kernel (the host enqueues the helper-stream kernels first), every cross-stream
// join costs ~10 us, and the arithmetic of one operation runs on one wave.  This kernel runs the same device code in one launch:
//
//   phase 1   every operation owns a CLUSTER of workgroups (4 waves each).  Each wave takes one role: ExpandA for one polynomial of
//             A_hat (expand_a_coop2_poly; absent when the caller keeps A_hat with its keys), mu = H(tr | M') (ml_dsa.rs:386-397) or
//             c = SampleInBall(c~) (ml_dsa.rs:400) -- all three on the one-state-per-wave cooperative sponge (keccak_coop2.h: 2.2 us
//             per permutation against 3.8 of the two-state form and 9.4 lane-per-state).
//             Results go to the call's workspace rows (24-bit A_hat, mu, c as bytes, the refusal flag).
//   hand-over a workgroup that has finished its roles releases its stores (agent scope) and bumps the operation's counter; the workgroup
//             that sees the count complete -- the LAST one to arrive -- acquires and carries on.  Nobody waits: no spinning, no
//             co-residency assumption.  The counter is reset by that workgroup, so the array stays zero between calls.
//             (A cluster of one workgroup -- A_hat kept by the caller -- needs no counter.)
//   tail      the last workgroup runs the rest with its four waves side by side: sigDecode + NTT(z_j), NTT(c), the hint decode,
//             then the rows  w'_i = invNTT(A_i o z_hat - c_hat o t1_hat_i), UseHint, w1Encode  (ml_dsa.rs:407-428) into LDS beside mu,
//             then c~' = H(mu | w1') on one wave straight from LDS and the verdict (ml_dsa.rs:429-436).
//
// Same device functions as the batch kernels (coeff_from_three_bytes ranking of expand_a_coop_pair, ntt_fwd_wave / ntt_inv_wave, hint_unpack_wave, use_hint,
// pack_w1_strided), same bytes in the workspace rows, same verdicts: tests/test_gpu_small_calls.py compares the two
// paths with each other and with the oracle over the ACVP sigVer vectors, damaged signatures, refused operations and every mode.
// Workgroup b belongs to XCD b mod 8 (round-robin dispatch): the clusters are laid out so that all workgroups of an operation share
// an XCD, i.e. the L2 that holds the A_hat rows they hand over.
#include "ctx.h"
#include "sign_slots_dev.h"
#include "challenge_dev.h"
#include "expand_coop_dev.h"
#include "keccak_coop2.h"
#include "ntt_wave.h"
#include "rounding.h"
#include "sampler_dev.h"
#include "verify_dev.h"

namespace mldsa {

constexpr int SMW = 4;  // waves per workgroup

struct SmallVerifyArgs {
    const uint8_t* rho;        // ExpandA input: rho rows of the keys (unused when a_keys is set)
    size_t rho_stride;
    const int32_t* a_keys;     // A_hat kept by the caller, int32 rows per key (mldsa_verify_cached_a), or nullptr
    const uint8_t* tr;         // [n_keys][64]
    const int32_t* t1;         // [n_keys][K][256]
    const uint32_t* key_idx;   // nullptr: op i uses key i
    uint32_t n_keys;
    int mode;
    const uint8_t* msgs;
    const uint64_t* msg_off;
    const uint8_t* ctxs;
    const uint64_t* ctx_off;
    const uint8_t* sigs;
    uint8_t* ok;
    uint32_t n_ops;
    // workspace rows of the call
    int32_t* a_ws;             // [n_ops][K * L] polynomials in the 24-bit form
    uint32_t* c_ws;            // [n_ops][64] dwords: c as one byte per coefficient
    uint8_t* mu_ws;            // [n_ops][64]
    int32_t* flag_ws;          // [n_ops]: 0 = hashed; 1 = ctx too long; 2 = malformed offsets / key index out of range
    uint32_t* ctr;             // [n_ops] arrival counters, zero between calls
    const Twiddle *fwd_tab, *inv_tab;
};

// s_waitcnt vmcnt(0): the wave's outstanding vector-memory operations (loads and stores) have completed -- for a store: written to L2
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- role: mu = H(tr | M', 64) by one wave (verify_dev.h mu_coop2: k_mu's checks, flags and bytes)
__device__ __forceinline__ void small_role_mu(const SmallVerifyArgs& A, size_t op, int lane, const Coop2Lane& c) {
    size_t key = A.key_idx ? A.key_idx[op] : op;
    int key_bad = 0;
    if (A.key_idx && key >= A.n_keys) { key = 0; key_bad = 2; }
    uint32_t lo, hi;
    const int flag = mu_coop2(A.tr + key * 64, A.mode, A.msgs, A.msg_off, A.ctxs, A.ctx_off, op, A.n_ops, key_bad, lo, hi, lane, c);
    if (c.active && c.word < 8) reinterpret_cast<uint32_t*>(A.mu_ws + op * 64)[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
    if (lane == 0) A.flag_ws[op] = flag;
}

template <int K, int L, int GB, bool G2HI, int CT, bool CACHED>
__global__ __launch_bounds__(64 * SMW) void k_verify_small(SmallVerifyArgs A0, int tau, int omega, int32_t zbound, size_t sig_len) {
    constexpr int CB = GB + 1;
    constexpr int BITS = G2HI ? 4 : 6;
    constexpr int W1_LEN = K * 32 * BITS;
    constexpr int NA = CACHED ? 0 : K * L;              // ExpandA roles (one polynomial each)
    constexpr int ROLES = NA + 2, NB = (ROLES + SMW - 1) / SMW;
    __shared__ uint32_t blk_lds[SMW * EA_COOP_BLK_DWORDS];                    // phase 1: the ExpandA waves' blocks
    __shared__ __attribute__((aligned(16))) int4 zh[L + 1][64];               // tail: z_hat rows and c_hat
    __shared__ uint32_t hint_lds[HINT_LDS_DWORDS];
    __shared__ Twiddle tw_lds[(FWD_TW + INV_TW) * 64];
    __shared__ __attribute__((aligned(16))) uint32_t msg_lds[(64 + W1_LEN) / 4];  // mu | w1Encode(w1')
    __shared__ int s_last, s_zbad, s_hint_ok;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // cluster layout: workgroup b -> XCD b & 7; the NB workgroups of an op are NB consecutive workgroups of one XCD
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t member = q % NB;
    const size_t op = (size_t)(q / NB) * 8 + xcd;
    if (op >= A0.n_ops) return;  // (whole workgroup)
    const Coop2Lane c = coop2_lane(lane);
    const int role = (int)member * SMW + wave;

    // ---------------------------------------------------------------- phase 1
    if (role < NA) {
        expand_a_coop2_poly<K, L>(A0.rho, A0.rho_stride, A0.key_idx, A0.a_ws, op * (K * L) + (size_t)role, A0.n_keys, blk_lds + wave * EA_COOP_BLK_DWORDS, lane, c);
    } else if (role == NA) {
        small_role_mu(A0, op, lane, c);
    } else if (role == NA + 1) {
        // c = SampleInBall(c~) (ml_dsa.rs:400; challenge_dev.h).  c~ opens the signature (encodings.rs:251); the block's LDS row lies in zh,
        // which nothing else uses before the tail
        A0.c_ws[op * 64 + lane] = sample_in_ball_coop2<CT>(A0.sigs + op * sig_len, tau, reinterpret_cast<uint32_t*>(&zh[0][0]), lane, c);
    }

    // ---------------------------------------------------------------- hand-over
    // Every wave waits for ITS OWN global stores to be acknowledged by L2 before the barrier: the barrier's workgroup-scope release does
    // not wait for outstanding global stores on this target (the ISA shows s_waitcnt lgkmcnt(0) only), and thread 0's agent-scope fence
    // below waits only for its own wave's -- without this the counter could be bumped while another wave's A_hat bytes are still on
    // their way to L2.  (A full agent-scope release per wave -- buffer_wbl2 four times per workgroup -- is correct too and was measured:
    // 256-op calls 123 -> 202 us.  Thread 0's buffer_wbl2 after the barrier writes back the whole L2, the other waves' lines included.)
    if (NB > 1) wait_own_stores();
    __syncthreads();
    if (NB > 1) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t seen = __hip_atomic_fetch_add(&A0.ctr[op], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = seen == (uint32_t)(NB - 1);
            if (last) __hip_atomic_store(&A0.ctr[op], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next call
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: the other workgroups' rows are visible from here on
    }

    // ---------------------------------------------------------------- tail (this workgroup's four waves)
    // The tail reads its arguments afresh from the kernarg segment (field.h late_arg): what phase 1 used dies with it, what the tail
    // uses is loaded here -- otherwise all ~25 scalars of the argument struct stay live across both and four of them spill.
    SmallVerifyArgs A;
    reload_first_kernarg(A, A0);
    for (int i = threadIdx.x; i < FWD_TW * 64; i += 64 * SMW) tw_lds[i] = A.fwd_tab[i];
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * SMW) tw_lds[FWD_TW * 64 + i] = A.inv_tab[i];
    if (threadIdx.x < 16) msg_lds[threadIdx.x] = reinterpret_cast<const uint32_t*>(A.mu_ws + op * 64)[threadIdx.x];
    if (threadIdx.x == 0) { s_zbad = 0; s_hint_ok = 1; }
    __syncthreads();
    const LdsTw ftw{tw_lds, lane};
    const LdsTw itw{tw_lds + FWD_TW * 64, lane};
    size_t key = A.key_idx ? (size_t)__builtin_amdgcn_readfirstlane((int)A.key_idx[op]) : op;
    if (A.key_idx && key >= A.n_keys) key = 0;  // (the op is refused through its flag)
    const uint8_t* zsrc = A.sigs + op * sig_len + CT;
    // forward transforms: z_0 .. z_{L-1} from the signature bytes (bit_unpack, conversion.rs:227-262, with the norm test of
    // ml_dsa.rs:434), then c; item j by wave j mod 4
    bool zbad = false;
#pragma unroll 1
    for (int j = wave; j <= L; j += SMW) {
        int32_t r[4];
        if (j < L) {
            const uint8_t* src = zsrc + (size_t)j * (32 * CB);
            int32_t mx = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = y_from_raw<CB>(y_raw_dword<CB>(src, k, lane), lane);
                const int32_t a = r[k] < 0 ? -r[k] : r[k];
                mx = a > mx ? a : mx;
            }
            zbad |= mx >= zbound;
        } else {
            const uint32_t d = A.c_ws[op * 64 + lane];
            r[0] = (int8_t)(d & 0xFF); r[1] = (int8_t)((d >> 8) & 0xFF); r[2] = (int8_t)((d >> 16) & 0xFF); r[3] = (int8_t)(d >> 24);
        }
        ntt_fwd_wave(r, ftw, lane);
        if (j == L) {
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = mont_mul(r[k], 1);  // c_hat * 2^-32
        }
        zh[j][lane] = make_int4(r[0], r[1], r[2], r[3]);
    }
    if (__ballot(zbad) != 0ull && lane == 0) atomicOr(&s_zbad, 1);
    if (wave == SMW - 1) {  // the wave with the fewest forward items also decodes the hint section
        const HintBytes hbytes = hint_load(zsrc + L * (32 * CB), omega, K, lane);
        const bool hint_ok = hint_unpack_wave<K>(hbytes, omega, hint_lds, lane);
        if (lane == 0) s_hint_ok = hint_ok ? 1 : 0;
    }
    __syncthreads();
    // rows: i by wave i mod 4
    using ARow = std::conditional_t<!CACHED, Packed3, int4>;
    const size_t aop = CACHED ? key : op;
    const ARow* arow = !CACHED ? reinterpret_cast<const ARow*>(reinterpret_cast<const uint32_t*>(A.a_ws) + (aop * K * (size_t)L) * PACKED_POLY_DWORDS)
                               : reinterpret_cast<const ARow*>(A.a_keys + (aop * K * (size_t)L) * N);
    auto coeffs = [](const ARow& v) -> int4 {
        if constexpr (!CACHED) return unpack24(v); else return v;
    };
    const int4* trow = reinterpret_cast<const int4*>(A.t1 + (key * K) * (size_t)N);
#pragma unroll 1
    for (int i = wave; i < K; i += SMW) {
        ARow av[L];
#pragma unroll
        for (int j = 0; j < L; j++) av[j] = arow[(unsigned)((i * L + j) * 64) + (unsigned)lane];
        const int4 tv = trow[(unsigned)(i * 64) + (unsigned)lane];
        int64_t acc64[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int4 zv = zh[j][lane];
            const int4 a4 = coeffs(av[j]);
            acc64[0] += (int64_t)a4.x * zv.x;
            acc64[1] += (int64_t)a4.y * zv.y;
            acc64[2] += (int64_t)a4.z * zv.z;
            acc64[3] += (int64_t)a4.w * zv.w;
        }
        const int4 cv = zh[L][lane];
        acc64[0] -= (int64_t)cv.x * tv.x;
        acc64[1] -= (int64_t)cv.y * tv.y;
        acc64[2] -= (int64_t)cv.z * tv.z;
        acc64[3] -= (int64_t)cv.w * tv.w;
        const uint32_t hword = hint_lds[24 + i * 8 + (lane & 7)];
        int32_t acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = mont_reduce64(acc64[k]);
        ntt_inv_wave(acc, itw, lane, F_MONT2);
        uint8_t* dst = reinterpret_cast<uint8_t*>(msg_lds) + 64 + (size_t)i * (32 * BITS);
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)hword, 2 * k), whi = (uint32_t)__builtin_amdgcn_readlane((int)hword, 2 * k + 1);
            const uint32_t hwk = lane < 32 ? wlo : whi;
            const uint32_t h = (hwk >> (lane & 31)) & 1u;
            v[k] = (uint32_t)use_hint<G2HI>((int32_t)h, acc[k]);
        }
        pack_w1_strided<G2HI>(v, dst, lane);
    }
    __syncthreads();
    if (wave != 0) return;
    // c~' = H(mu | w1Encode(w1'), lambda / 4) from LDS, both halves on the same state; the verdict of ml_dsa.rs:429-436
    {
        constexpr int DATA = 64 + W1_LEN, BLOCKS = DATA / SHAKE256_RATE + 1;
        auto msg_dword = [&](int off) -> uint32_t {
            if (off + 4 <= DATA) return msg_lds[off >> 2];
            uint32_t v = off == DATA ? 0x1Fu : 0u;  // (DATA is a multiple of 4)
            if (off + 4 == BLOCKS * SHAKE256_RATE) v |= 0x80000000u;
            return v;
        };
        uint32_t v = 0;
        const bool absorbs = c.active && c.word < SHAKE256_RATE / 8;
#pragma unroll 1
        for (int blk = 0; blk < BLOCKS; blk++) {
            if (absorbs) {
                const int off = blk * SHAKE256_RATE + 8 * c.word;
                v ^= coop2_from_lohi(msg_dword(off), msg_dword(off + 4), c);
            }
            keccak_f1600_coop2(v, c);
        }
        uint32_t lo, hi;
        coop2_to_lohi(v, lane, lo, hi);
        const bool digest = c.active && c.word < CT / 8;
        const uint8_t* c0 = A.sigs + op * sig_len + 8 * c.word + 4 * (lane >> 5);  // the E lane compares the word's low dword, the O lane the high one
        const bool differs = digest && ((lane < 32 ? lo : hi) ^ load_le32(c0)) != 0;
        const unsigned long long any = __ballot(differs);
        if (lane == 0) A.ok[op] = (uint8_t)(any == 0ull && !s_zbad && s_hint_ok && !A.flag_ws[op]);
    }
}

// ------------------------------------------------------------------------------------ launcher
// workspace rows of a call of n ops (the caller carves them from the context's workspace) and the launch itself
size_t verify_small_counter_bytes(size_t n_ops) { return ((n_ops + 7) / 8 * 8) * sizeof(uint32_t); }

int launch_verify_small(mldsa_ctx* ctx, const mldsa_params* p, int mode, const uint8_t* rho, size_t rho_stride, const int32_t* a_keys, const uint8_t* tr,
                        const int32_t* t1, size_t n_keys, const uint32_t* key_idx, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* ctxs,
                        const uint64_t* ctx_off, const uint8_t* sigs, uint8_t* ok, size_t n_ops, int32_t* a_ws, int32_t* c_ws, uint8_t* mu_ws,
                        int32_t* flag_ws, uint32_t* ctr, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    SmallVerifyArgs A;
    A.rho = rho; A.rho_stride = rho_stride; A.a_keys = a_keys; A.tr = tr; A.t1 = t1; A.key_idx = key_idx;
    A.n_keys = (uint32_t)std::min<size_t>(n_keys, 0xFFFFFFFFu);
    A.mode = mode; A.msgs = msgs; A.msg_off = msg_off; A.ctxs = ctxs; A.ctx_off = ctx_off; A.sigs = sigs; A.ok = ok; A.n_ops = (uint32_t)n_ops;
    A.a_ws = a_ws; A.c_ws = reinterpret_cast<uint32_t*>(c_ws); A.mu_ws = mu_ws; A.flag_ws = flag_ws; A.ctr = ctr;
    A.fwd_tab = ctx->d_fwd_tw; A.inv_tab = ctx->d_inv_tw;
    const bool cached = a_keys != nullptr;
    const int roles = (cached ? 0 : p->k * p->l) + 2, nb = (roles + SMW - 1) / SMW;
    const dim3 grid((unsigned)(((n_ops + 7) / 8) * 8 * (size_t)nb)), block(64 * SMW);
    const int32_t zbound = p->gamma1 - p->beta;
#define MLDSA_SMALL(KK, LL, GB, G2, CT)                                                                                                      \
    do {                                                                                                                                     \
        if (cached) hipLaunchKernelGGL((k_verify_small<KK, LL, GB, G2, CT, true>), grid, block, 0, s, A, p->tau, p->omega, zbound, (size_t)p->sig_len); \
        else hipLaunchKernelGGL((k_verify_small<KK, LL, GB, G2, CT, false>), grid, block, 0, s, A, p->tau, p->omega, zbound, (size_t)p->sig_len);       \
    } while (0)
    if (p->set == MLDSA_44) MLDSA_SMALL(4, 4, 17, false, 32);
    else if (p->set == MLDSA_65) MLDSA_SMALL(6, 5, 19, true, 48);
    else MLDSA_SMALL(8, 7, 19, true, 64);
#undef MLDSA_SMALL
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ====================================================================================================================
// key_gen_internal (src/ml_dsa.rs:57-134) + into_bytes for SMALL calls as ONE launch, same structure as k_verify_small.
//
// The batch pipeline is six launches (seed hash 6 us, ExpandA 15, ExpandS 7, the matrix-vector kernel 14, the seeds' copy 6, tr = H(pk)
// 32 us for one ML-DSA-65 key: profiles/r05_small_call_timeline_keygen_n1.json).  Here every key owns a cluster of workgroups:
//   phase 1   one wave per polynomial of A_hat (K * L) and of s1 / s2 (L + K).  Each derives (rho, rho', K) = H(xi | K | L, 128) itself --
//             one permutation costs less than handing the seed over -- and expands its polynomial (expand_a_coop2_core,
//             expand_s_coop2_poly) into the call's workspace rows; the first wave also stores the 128 seed bytes for the tail.
//   tail      the LAST workgroup to arrive (per-key counter, release / acquire at agent scope): NTT(s1_j) by wave j mod 4 with s1's
//             section of sk packed on the way; rows t_i = invNTT(A_i o s1_hat) + s2_i by wave i mod 4 with s2's section, Power2Round,
//             t1 -> pk (kept in LDS as well), t0 -> sk; rho, K; then tr = H(pk) by one wave straight from LDS -> sk.
// Same workspace rows as keygen_batch (so the caller's clearing of rho' / K / s1 / s2 is unchanged), same bytes out.
struct SmallKeygenArgs {
    const uint8_t* xi;   // [n][32]
    uint8_t *pk, *sk;    // wire formats out
    uint32_t n_keys;
    int32_t* a_ws;       // [n][K * L] polynomials, 24-bit form
    uint8_t* hbuf;       // [n][128]: rho | rho' | K
    uint8_t* s_ws;       // [n][L + K][256]: one byte per coefficient
    uint32_t* ctr;
    const Twiddle *fwd_tab, *inv_tab;
    int wipe;            // the tail clears the key's secret workspace rows (rho' / K, s1, s2) itself: no clearing launches behind a small call
};

template <int K, int L, int ETA>
__global__ __launch_bounds__(64 * SMW) void k_keygen_small(SmallKeygenArgs A0) {
    constexpr int EB = ETA == 2 ? 3 : 4;
    constexpr int PK_LEN = 32 + 320 * K, SK_LEN = 128 + 32 * EB * (K + L) + 416 * K;
    constexpr size_t T0_OFF = 128 + 32 * EB * (K + L);
    constexpr int NA = K * L, ROLES = K * L + K + L, NB = (ROLES + SMW - 1) / SMW;
    __shared__ uint32_t blk_lds[SMW * EA_COOP_BLK_DWORDS];
    __shared__ uint32_t seed_lds[SMW][32];
    __shared__ __attribute__((aligned(16))) int4 zh[L][64];
    __shared__ __attribute__((aligned(16))) int32_t xp[SMW][N];
    __shared__ Twiddle tw_lds[(FWD_TW + INV_TW) * 64];
    __shared__ __attribute__((aligned(16))) uint32_t pk_lds[PK_LEN / 4];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t member = q % NB;
    const size_t op = (size_t)(q / NB) * 8 + xcd;
    if (op >= A0.n_keys) return;  // (whole workgroup)
    const Coop2Lane c = coop2_lane(lane);
    const int role = (int)member * SMW + wave;

    // ---------------------------------------------------------------- phase 1
    if (role < ROLES) {
        // (rho, rho', K) <- H(xi || K || L, 128)                              ml_dsa.rs:68-74
        uint32_t lo = 0, hi = 0;
        if (c.active && c.word < 4) {
            lo = load_le32(A0.xi + op * 32 + 8 * c.word);
            hi = load_le32(A0.xi + op * 32 + 8 * c.word + 4);
        }
        if (c.active && c.word == 4) lo = (uint32_t)K | ((uint32_t)L << 8) | (0x1Fu << 16);
        if (c.active && c.word == SHAKE256_RATE / 8 - 1) hi = 0x80000000u;
        uint32_t v = c.active ? coop2_from_lohi(lo, hi, c) : 0u;
        keccak_f1600_coop2(v, c);
        coop2_to_lohi(v, lane, lo, hi);
        uint32_t* seed = seed_lds[wave];
        if (c.active && c.word < 16) seed[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
        wave_lds_sync();
        if (role == 0 && lane < 32) reinterpret_cast<uint32_t*>(A0.hbuf + op * 128)[lane] = seed[lane];
        const uint8_t* sb = reinterpret_cast<const uint8_t*>(seed);
        if (role < NA) {  // A_hat[r][s] <- RejNTTPoly(rho || s || r)                 ml_dsa.rs:85
            uint8_t* row = reinterpret_cast<uint8_t*>(A0.a_ws) + (op * NA + (size_t)role) * (size_t)(PACKED_POLY_DWORDS * 4);
            expand_a_coop2_core(sb, role / L, role % L, row, blk_lds + wave * EA_COOP_BLK_DWORDS, lane, c);
        } else {          // s1 / s2 <- ExpandS(rho')                                  ml_dsa.rs:79
            const uint32_t r = (uint32_t)(role - NA);
            expand_s_coop2_poly<ETA>(sb + 32, r, A0.s_ws + (op * (size_t)(K + L) + r) * N, blk_lds + wave * EA_COOP_BLK_DWORDS, lane, c);
        }
    }

    // ---------------------------------------------------------------- hand-over (as in k_verify_small: every wave releases its own stores first)
    if (NB > 1) wait_own_stores();
    __syncthreads();
    if (NB > 1) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t seen = __hip_atomic_fetch_add(&A0.ctr[op], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = seen == (uint32_t)(NB - 1);
            if (last) __hip_atomic_store(&A0.ctr[op], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }

    // ---------------------------------------------------------------- tail
    SmallKeygenArgs A;
    reload_first_kernarg(A, A0);
    for (int i = threadIdx.x; i < FWD_TW * 64; i += 64 * SMW) tw_lds[i] = A.fwd_tab[i];
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * SMW) tw_lds[FWD_TW * 64 + i] = A.inv_tab[i];
    uint8_t* pkb = reinterpret_cast<uint8_t*>(pk_lds);
    uint8_t* sk = A.sk + op * (size_t)SK_LEN;
    if (threadIdx.x < 32) {  // rho -> pk, sk; K -> sk                         encodings.rs:29, 110-116
        const uint8_t r = A.hbuf[op * 128 + threadIdx.x];
        pkb[threadIdx.x] = r;
        sk[threadIdx.x] = r;
        sk[32 + threadIdx.x] = A.hbuf[op * 128 + 96 + threadIdx.x];
    }
    __syncthreads();
    const LdsTw ftw{tw_lds, lane};
    const LdsTw itw{tw_lds + FWD_TW * 64, lane};
    const uint32_t* srows = reinterpret_cast<const uint32_t*>(A.s_ws) + op * (size_t)(K + L) * (N / 4);  // dword d of a row = coefficients 4 d .. 4 d + 3
    auto eta_fields = [&](uint32_t d, uint32_t (&f)[4]) {
        f[0] = (uint32_t)(ETA - (int8_t)(d & 0xFF)); f[1] = (uint32_t)(ETA - (int8_t)((d >> 8) & 0xFF));
        f[2] = (uint32_t)(ETA - (int8_t)((d >> 16) & 0xFF)); f[3] = (uint32_t)(ETA - (int8_t)(d >> 24));
    };
#pragma unroll 1
    for (int j = wave; j < L; j += SMW) {  // s1_j: its section of sk (encodings.rs:118-134) and its transform
        const uint32_t* src = srows + (size_t)j * (N / 4);
        uint32_t f[4];
        eta_fields(src[lane], f);
        store_fields(sk + 128 + (size_t)j * (32 * EB), f, EB, lane);
        int32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = (int32_t)(int8_t)(src[16 * k + (lane >> 2)] >> (8 * (lane & 3)));  // coefficient 64 k + lane
        ntt_fwd_wave(r, ftw, lane);
        zh[j][lane] = make_int4(r[0], r[1], r[2], r[3]);
    }
    __syncthreads();
    const Packed3* arow = reinterpret_cast<const Packed3*>(reinterpret_cast<const uint32_t*>(A.a_ws) + (op * NA) * (size_t)PACKED_POLY_DWORDS);
#pragma unroll 1
    for (int i = wave; i < K; i += SMW) {  // t_i = invNTT(A_i o s1_hat) + s2_i, Power2Round, t1 -> pk, t0 -> sk      ml_dsa.rs:86-92
        int64_t acc64[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int4 zv = zh[j][lane];
            const int4 a4 = unpack24(arow[(unsigned)((i * L + j) * 64) + (unsigned)lane]);
            acc64[0] += (int64_t)a4.x * zv.x;
            acc64[1] += (int64_t)a4.y * zv.y;
            acc64[2] += (int64_t)a4.z * zv.z;
            acc64[3] += (int64_t)a4.w * zv.w;
        }
        int32_t acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = mont_reduce64(acc64[k]);
        ntt_inv_wave(acc, itw, lane, F_MONT2);
        const uint32_t* s2row = srows + (size_t)(L + i) * (N / 4);
        {
            uint32_t f[4];
            eta_fields(s2row[lane], f);
            store_fields(sk + 128 + (size_t)(L + i) * (32 * EB), f, EB, lane);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t s2v = (int32_t)(int8_t)(s2row[16 * k + (lane >> 2)] >> (8 * (lane & 3)));
            xp[wave][64 * k + lane] = freeze(acc[k] + s2v);
        }
        wave_lds_sync();
        const int4 t4 = reinterpret_cast<const int4*>(&xp[wave][0])[lane];  // four consecutive coefficients per lane
        wave_lds_sync();
        const int32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
        uint32_t f1[4], f0[4];
#pragma unroll
        for (int cidx = 0; cidx < 4; cidx++) {
            const int32_t r1 = (tt[cidx] + (1 << 12) - 1) >> 13;  // power2round, high_low.rs:26-31
            const int32_t r0 = tt[cidx] - (r1 << 13);
            f1[cidx] = (uint32_t)r1;
            f0[cidx] = (uint32_t)((1 << 12) - r0);                // BitPack(t0, 2^12 - 1, 2^12)
        }
        store_fields(pkb + 32 + (size_t)i * 320, f1, 10, lane);
        store_fields(sk + T0_OFF + (size_t)i * 416, f0, 13, lane);
    }
    __syncthreads();
    {   // pk out
        uint32_t* pk_out = reinterpret_cast<uint32_t*>(A.pk + op * (size_t)PK_LEN);  // (PK_LEN is a multiple of 32: rows stay 4-byte aligned)
        for (int i = threadIdx.x; i < PK_LEN / 4; i += 64 * SMW) pk_out[i] = pk_lds[i];
    }
    if (A.wipe) {  // rho' / K, s1, s2 of this key are not needed any more (types.rs:19: zeroize on drop); every other workgroup of the cluster has left
        uint32_t* sw = reinterpret_cast<uint32_t*>(A.s_ws) + op * (size_t)(K + L) * (N / 4);
        for (int i = threadIdx.x; i < (K + L) * (N / 4); i += 64 * SMW) sw[i] = 0;
        if (threadIdx.x < 32) reinterpret_cast<uint32_t*>(A.hbuf + op * 128)[threadIdx.x] = 0;
    }
    if (wave != 0) return;
    {   // tr <- H(pk, 64)                                                       ml_dsa.rs:99-101
        constexpr int BLOCKS = PK_LEN / SHAKE256_RATE + 1;
        auto msg_dword = [&](int off) -> uint32_t {
            if (off + 4 <= PK_LEN) return pk_lds[off >> 2];
            uint32_t v = off == PK_LEN ? 0x1Fu : 0u;
            if (off + 4 == BLOCKS * SHAKE256_RATE) v |= 0x80000000u;
            return v;
        };
        uint32_t v = 0;
        const bool absorbs = c.active && c.word < SHAKE256_RATE / 8;
#pragma unroll 1
        for (int blk = 0; blk < BLOCKS; blk++) {
            if (absorbs) {
                const int off = blk * SHAKE256_RATE + 8 * c.word;
                v ^= coop2_from_lohi(msg_dword(off), msg_dword(off + 4), c);
            }
            keccak_f1600_coop2(v, c);
        }
        uint32_t lo, hi;
        coop2_to_lohi(v, lane, lo, hi);
        if (c.active && c.word < 8) *reinterpret_cast<uint32_t*>(sk + 64 + 8 * c.word + 4 * (lane >> 5)) = lane < 32 ? lo : hi;
    }
}

int launch_keygen_small(mldsa_ctx* ctx, const mldsa_params* p, const uint8_t* xi, uint8_t* pk, uint8_t* sk, size_t n_keys, int32_t* a_ws, uint8_t* hbuf,
                        int32_t* s_ws, uint32_t* ctr, hipStream_t s, bool wipe) {
    if (n_keys == 0) return MLDSA_OK;
    SmallKeygenArgs A;
    A.xi = xi; A.pk = pk; A.sk = sk; A.n_keys = (uint32_t)n_keys; A.a_ws = a_ws; A.hbuf = hbuf; A.s_ws = reinterpret_cast<uint8_t*>(s_ws); A.ctr = ctr;
    A.fwd_tab = ctx->d_fwd_tw; A.inv_tab = ctx->d_inv_tw;
    A.wipe = wipe ? 1 : 0;
    const int roles = p->k * p->l + p->k + p->l, nb = (roles + SMW - 1) / SMW;
    const dim3 grid((unsigned)(((n_keys + 7) / 8) * 8 * (size_t)nb)), block(64 * SMW);
    if (p->set == MLDSA_44) hipLaunchKernelGGL((k_keygen_small<4, 4, 2>), grid, block, 0, s, A);
    else if (p->set == MLDSA_65) hipLaunchKernelGGL((k_keygen_small<6, 5, 4>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_keygen_small<8, 7, 2>), grid, block, 0, s, A);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ====================================================================================================================
// The prologue of sign_internal (src/ml_dsa.rs:181-204) for SMALL calls as ONE launch.
//
// The batch pipeline enqueues nine kernels before the first round: the key check, ExpandA, a clearing of the round control block, mu,
// a copy of rnd, rho'' = H(K | rnd | mu), a clearing of the key-range flags, the key-range check and the first active list -- 43 us
// of kernels and eight boundaries for one op (profiles/r05_small_call_timeline_sign_n1.json: the first round starts 65 us into the
// call).  Here every op owns a cluster of workgroups: one wave per polynomial of A_hat (absent when the caller keeps A_hat with its
// keys) and ONE wave for everything else of the op, in order: key index check, mu (mu_coop2), rnd | mu row, rho'' (one more
// permutation), kappa / done / status (a refused op's signature zeroed), the key-range check of the op's unit (K inverse transforms).
// Every workgroup bumps ONE counter of the call; the last one to arrive builds the first active list (in op order) and the round
// control block.  Same workspace rows, same values as the nine kernels (the active list's order is not part of the contract: the
// batch kernel fills it with atomics).
template <int K, int L, bool CACHED>
__global__ __launch_bounds__(64 * SMW) void k_sign_prologue_small(SmallSignPrologueArgs A0) {
    constexpr int NA = CACHED ? 0 : K * L, ROLES = NA + 1, NB = (ROLES + SMW - 1) / SMW;
    __shared__ uint32_t blk_lds[SMW * EA_COOP_BLK_DWORDS];
    __shared__ uint32_t row_lds[24];  // rnd | mu of the op (one op wave per workgroup at most)
    __shared__ uint32_t wave_cnt[SMW];
    __shared__ uint32_t s_spec0;
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t member = q % NB;
    const size_t op = (size_t)(q / NB) * 8 + xcd;
    late_args_begin(A0);  // (a no-op unless built with -DMLDSA_NO_LATE_ARG, field.h)
    const uint32_t n = A0.n;
    if (op >= n) return;  // (whole workgroup)
    const Coop2Lane c = coop2_lane(lane);
    const int role = (int)member * SMW + wave;
    // (The argument struct has some forty scalars; every step below re-reads it from the kernarg segment -- reload_first_kernarg -- and
    //  the compiler keeps only that step's fields: held in registers from the kernel's entry, 19-26 of them spilled.)
    if (role < NA) {
        // 5: A_hat <- ExpandA(rho)                                            ml_dsa.rs:181
        expand_a_coop2_poly<K, L>(A0.rho, 32, A0.key_idx, A0.a_ws, op * NA + (size_t)role, A0.key_idx ? A0.n_keys : 0u, blk_lds + wave * EA_COOP_BLK_DWORDS,
                                  lane, c);
    } else if (role == NA) {
        size_t key = op;
        int bad = 0;
        uint32_t lo, hi;
        {   // the key index (the C ABI's promise: an out-of-range index never reaches memory), then 6: mu <- H(tr || M', 64)   ml_dsa.rs:185-196
            SmallSignPrologueArgs A;
            reload_first_kernarg(A, A0);
            int key_bad = 0;
            if (A.key_idx) {
                const uint32_t kraw = A.key_idx[op];
                key = kraw < A.n_keys ? kraw : 0u;
                key_bad = kraw < A.n_keys ? 0 : 2;
                if (lane == 0) A.kidx_out[op] = (uint32_t)key;
            }
            bad = mu_coop2(A.tr + key * 64, A.mode, A.msgs, A.msg_off, A.ctxs, A.ctx_off, A.op0 + op, A.n_call, key_bad, lo, hi, lane, c);
        }
        {   // the op's rnd | mu row; 7: rho'' <- H(K || rnd || mu, 64)          ml_dsa.rs:199-201   (128 bytes: one block)
            SmallSignPrologueArgs A;
            reload_first_kernarg(A, A0);
            if (lane < 8) row_lds[lane] = load_le32(A.rnd + op * 32 + 4 * lane);
            if (c.active && c.word < 8) row_lds[8 + 2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
            wave_lds_sync();
            if (lane < 24) reinterpret_cast<uint32_t*>(A.rnd_mu + op * 96)[lane] = row_lds[lane];
            uint32_t l2 = 0, h2 = 0;
            if (c.active && c.word < 4) {
                l2 = load_le32(A.cap_k + key * 32 + 8 * c.word);
                h2 = load_le32(A.cap_k + key * 32 + 8 * c.word + 4);
            } else if (c.active && c.word < 16) {
                l2 = row_lds[2 * (c.word - 4)];
                h2 = row_lds[2 * (c.word - 4) + 1];
            } else if (c.active && c.word == 16) {
                l2 = 0x1Fu;
                h2 = 0x80000000u;
            }
            uint32_t v = c.active ? coop2_from_lohi(l2, h2, c) : 0u;
            keccak_f1600_coop2(v, c);
            coop2_to_lohi(v, lane, l2, h2);
            if (c.active && c.word < 8) reinterpret_cast<uint32_t*>(A.rho_pp + op * 64)[2 * c.word + (lane >> 5)] = lane < 32 ? l2 : h2;
        }
        {   // 8: kappa <- 0; a refused op (ctx too long, lib.rs:274; bad key index / offsets) is done at once with an all-zero signature
            SmallSignPrologueArgs A;
            reload_first_kernarg(A, A0);
            if (lane == 0) {
                A.kappa[op] = 0;
                A.bad_op[op] = bad;
                A.done[op] = bad;
                if (A.status) A.status[op] = bad == 0 ? MLDSA_OK : bad == 1 ? MLDSA_ERR_CTX_LEN : MLDSA_ERR_PARAM;
            }
            if (bad)
                for (size_t b = lane; b < A.sig_len; b += 64) A.sigs[op * A.sig_len + b] = 0;
        }
        {   // is the unit's s2 within [-eta, eta]?  (k_key_range: a key that expand_private decoded out of range takes the reference's
            // two-transform hint stage)
            SmallSignPrologueArgs A;
            reload_first_kernarg(A, A0);
            if (op < A.units) {
                const size_t ukey = A.units_by_op ? key : op;
                InvTw tw;
                load_inv_tw(tw, A.inv_tab, lane);
                bool oor = false;
#pragma unroll 1
                for (int i = 0; i < K; i++) {
                    int32_t r[4];
                    load_packed(r, A.s2 + (ukey * K + i) * (size_t)N, lane);
#pragma unroll
                    for (int k = 0; k < 4; k++) r[k] = mont_mul(reduce32(r[k]), 1);  // mont_reduce(x_hat_mont) = x_hat
                    ntt_inv_wave(r, tw, lane, F_MONT);                                // canonical [0, q)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int32_t cen = r[k] - ((((Q / 2) - r[k]) >> 31) & Q);
                        oor |= (cen < 0 ? -cen : cen) > A.eta;
                    }
                }
                const bool any = __ballot(oor) != 0ull;
                if (lane == 0) A.key_oor[op] = any ? 1 : 0;
            } else if (lane == 0) {
                A.key_oor[op] = 0;
            }
        }
    }
    // ---------------------------------------------------------------- the call's counter: the last workgroup finishes
    wait_own_stores();
    __syncthreads();
    SmallSignPrologueArgs A;
    reload_first_kernarg(A, A0);
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t seen = __hip_atomic_fetch_add(&A.ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen == n * (uint32_t)NB - 1u;
        if (last) __hip_atomic_store(&A.ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // active = every op that was not refused, in op order (n <= 256: one thread per op); the round control block: zero but for the count
    {
        const uint32_t i = threadIdx.x;
        const bool live = i < n && A.done[i] == 0;
        const unsigned long long b = __ballot(live);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int w2 = 0; w2 < SMW; w2++) {
            if (w2 < wave) base += wave_cnt[w2];
            total += wave_cnt[w2];
        }
        if (live) A.act0[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = i;
        uint32_t* cw = reinterpret_cast<uint32_t*>(A.ctl);
        for (uint32_t d = threadIdx.x; d < sizeof(RoundCtl) / 4; d += 64 * SMW) cw[d] = d == 0 ? total : 0u;  // cnt[0] is the first field
        // ... and round 0 is opened right here (k_make_slots' work for parity 0 on the list just built; the rule's table is walked in the
        // kernarg segment by one thread)
        if (A.slots0) {
            if (threadIdx.x == 0) {
                uint32_t sp = 1;
                while (total && sp < A.spec_max && sp < 64 &&
                       total <= late_arg<uint32_t>((unsigned)(offsetof(SmallSignPrologueArgs, rule) + offsetof(SpecRule, thr)) + 4u * sp))
                    sp++;
                s_spec0 = sp;
            }
            wait_own_stores();  // the list and the cleared control block have reached L2 before any thread of the workgroup reads them
            __syncthreads();
            make_slots_body(A.ctl, 0, total, s_spec0, 0u, A.ns_cap, A.act0, A.kappa, A.l, A.slot_op, A.slot_kappa, A.key_idx ? A.kidx_out : nullptr,
                            A.gen_op, A.gen_kappa, A.gen_key, 0, 0, nullptr, A.slot_y, threadIdx.x, 64u * SMW, true);
        }
    }
}

int launch_sign_prologue_small(mldsa_ctx* ctx, const mldsa_params* p, const SmallSignPrologueArgs& A, bool cached, hipStream_t s) {
    if (A.n == 0) return MLDSA_OK;
    static_assert(offsetof(RoundCtl, cnt) == 0 && sizeof(RoundCtl) % 4 == 0, "k_sign_prologue_small writes the control block by dwords");
    const int roles = (cached ? 0 : p->k * p->l) + 1, nb = (roles + SMW - 1) / SMW;
    const dim3 grid((unsigned)((((size_t)A.n + 7) / 8) * 8 * (size_t)nb)), block(64 * SMW);
#define MLDSA_SP(KK, LL)                                                                                              \
    do {                                                                                                              \
        if (cached) hipLaunchKernelGGL((k_sign_prologue_small<KK, LL, true>), grid, block, 0, s, A);                    \
        else hipLaunchKernelGGL((k_sign_prologue_small<KK, LL, false>), grid, block, 0, s, A);                          \
    } while (0)
    if (p->set == MLDSA_44) MLDSA_SP(4, 4);
    else if (p->set == MLDSA_65) MLDSA_SP(6, 5);
    else MLDSA_SP(8, 7);
#undef MLDSA_SP
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ====================================================================================================================
// The first half of a signing round (ml_dsa.rs:215-240) for SMALL rounds as ONE launch: y = ExpandMask(rho'', kappa), w = invNTT(A_hat o
// NTT(y)), w1 = HighBits(w), c~ = H(mu | w1Encode(w1)), c = SampleInBall(c~), c_hat = NTT(c) -- five launches of the batch pipeline
// (ExpandMask 15.7, sign_w 11.1, the c~ hash 20.2, SampleInBall 11.4, NTT(c) 4.8 us for the 32 candidate rows of one op).  Every
// candidate ROW owns a cluster: one wave per polynomial of y (expand_mask_coop2_poly); the last workgroup to arrive transforms y_j
// (wave j mod 4, with the y risk flags), computes the K rows of w (wave i mod 4: 24-bit planes out, HighBits / w1Encode into LDS beside mu,
// the w risk bits), hashes mu | w1 from LDS, samples c and transforms it.  The rows it leaves -- y, w, w1, c~, c, c_hat, the risk flags
// -- are byte for byte those of the five kernels, so k_sign_tail / k_resolve / k_compact follow unchanged.
template <int K, int L, int GB, bool G2HI, int CT, bool APACK>
__global__ __launch_bounds__(64 * SMW) void k_sign_front_small(SmallSignFrontArgs A0) {
    constexpr int YCB = GB + 1, ROW_BYTES = 32 * YCB;
    constexpr int BITS = G2HI ? 4 : 6;
    constexpr int W1_LEN = K * 32 * BITS;
    constexpr int NB = (L + SMW - 1) / SMW;
    __shared__ __attribute__((aligned(16))) int4 zh[L][64];
    __shared__ Twiddle tw_lds[(FWD_TW + INV_TW) * 64];
    __shared__ __attribute__((aligned(16))) uint32_t msg_lds[(64 + W1_LEN) / 4];  // mu | w1Encode(w1)
    __shared__ uint32_t ct_lds[16], bw_lds[36];
    __shared__ int s_last, s_risk;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t member = q % NB;
    const size_t g = (size_t)(q / NB) * 8 + xcd;  // the row
    if (g >= *A0.ns_gen || g >= A0.rows_cap) return;  // (whole workgroup)
    const Coop2Lane c = coop2_lane(lane);
    const int role = (int)member * SMW + wave;
    const size_t op = A0.gen_op[g];
    // ---------------------------------------------------------------- phase 1: 11: y <- ExpandMask(rho'', kappa)      ml_dsa.rs:215
    if (role < L)
        expand_mask_coop2_poly<GB>(A0.rho_pp + op * 64, (uint32_t)A0.gen_kappa[g] + (uint32_t)role,
                                   reinterpret_cast<uint8_t*>(A0.y) + (g * L + (size_t)role) * (size_t)ROW_BYTES, lane, c);
    // ---------------------------------------------------------------- hand-over (as in k_verify_small)
    if (NB > 1) wait_own_stores();
    __syncthreads();
    if (NB > 1) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t seen = __hip_atomic_fetch_add(&A0.ctr[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = seen == (uint32_t)(NB - 1);
            if (last) __hip_atomic_store(&A0.ctr[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---------------------------------------------------------------- tail
    SmallSignFrontArgs A;
    reload_first_kernarg(A, A0);
    for (int i = threadIdx.x; i < FWD_TW * 64; i += 64 * SMW) tw_lds[i] = A.fwd_tab[i];
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * SMW) tw_lds[FWD_TW * 64 + i] = A.inv_tab[i];
    if (threadIdx.x < 16) msg_lds[threadIdx.x] = load_le32(A.mu + op * 96 + 4 * threadIdx.x);
    if (threadIdx.x == 0) s_risk = 0;
    __syncthreads();
    const LdsTw ftw{tw_lds, lane};
    const LdsTw itw{tw_lds + FWD_TW * 64, lane};
#pragma unroll 1
    for (int j = wave; j < L; j += SMW) {  // NTT(y_j), and: can this polynomial fail ||z||inf < gamma1 - beta?  (some |y| >= gamma1 - 2 beta)
        const uint8_t* src = reinterpret_cast<const uint8_t*>(A.y) + (g * L + (size_t)j) * (size_t)ROW_BYTES;
        int32_t r[4];
        bool near = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r[k] = y_from_raw<YCB>(y_raw_dword<YCB>(src, k, lane), lane);  // |y| <= gamma1 < q: no reduction
            near |= (r[k] < 0 ? -r[k] : r[k]) >= A.y_risk_bound;
        }
        const bool any_near = __ballot(near) != 0ull;
        if (lane == 0) A.yrisk[g * L + (size_t)j] = any_near ? 1 : 0;
        ntt_fwd_wave(r, ftw, lane);
        zh[j][lane] = make_int4(r[0], r[1], r[2], r[3]);
    }
    __syncthreads();
    using ARow = std::conditional_t<APACK, Packed3, int4>;
    const size_t aop = A.a_idx[g];
    const ARow* arow = APACK ? reinterpret_cast<const ARow*>(reinterpret_cast<const uint32_t*>(A.a_hat) + (aop * K * (size_t)L) * PACKED_POLY_DWORDS)
                             : reinterpret_cast<const ARow*>(A.a_hat + (aop * K * (size_t)L) * N);
    auto coeffs = [](const ARow& v) -> int4 {
        if constexpr (APACK) return unpack24(v); else return v;
    };
#pragma unroll 1
    for (int i = wave; i < K; i += SMW) {  // 12: w_i <- invNTT(A_hat[i] o y_hat); 13-14: w1_i = HighBits(w_i), w1Encode       ml_dsa.rs:218-232
        int64_t acc64[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int4 zv = zh[j][lane];
            const int4 a4 = coeffs(arow[(unsigned)((i * L + j) * 64) + (unsigned)lane]);
            acc64[0] += (int64_t)a4.x * zv.x;
            acc64[1] += (int64_t)a4.y * zv.y;
            acc64[2] += (int64_t)a4.z * zv.z;
            acc64[3] += (int64_t)a4.w * zv.w;
        }
        int32_t acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = mont_reduce64(acc64[k]);
        ntt_inv_wave(acc, itw, lane, F_MONT2);
        const Packed3 pw = pack24((uint32_t)acc[0], (uint32_t)acc[1], (uint32_t)acc[2], (uint32_t)acc[3]);
        uint32_t* wp = reinterpret_cast<uint32_t*>(A.w) + (g * K + (size_t)i) * (size_t)PACKED_POLY_DWORDS;
        wp[lane] = pw.a;
        wp[64 + lane] = pw.b;
        wp[128 + lane] = pw.c;
        uint32_t hb[4];
        bool near = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int32_t r1, r0;
            decompose<G2HI>(acc[k], r1, r0);  // HighBits = r1 (high_low.rs:104-111)
            hb[k] = (uint32_t)r1;
            near |= (r0 < 0 ? -r0 : r0) >= A.w_risk_bound;
        }
        pack_w1_strided<G2HI>(hb, reinterpret_cast<uint8_t*>(msg_lds) + 64 + (size_t)i * (32 * BITS), lane);
        if (__ballot(near) != 0ull && lane == 0) atomicOr(&s_risk, 1 << i);  // bit i: some |LowBits(w_i)| >= gamma2 - 2 beta
    }
    __syncthreads();
    if (threadIdx.x == 0) A.wrisk[g] = (uint8_t)s_risk;
    {   // w1 row out (the batch kernels keep it in the workspace too)
        uint32_t* w1o = reinterpret_cast<uint32_t*>(A.w1 + g * (size_t)W1_LEN);
        for (int i = threadIdx.x; i < W1_LEN / 4; i += 64 * SMW) w1o[i] = msg_lds[16 + i];
    }
    if (wave != 0) return;
    // 15: c~ <- H(mu || w1Encode(w1), lambda / 4)                          ml_dsa.rs:233
    {
        constexpr int DATA = 64 + W1_LEN, BLOCKS = DATA / SHAKE256_RATE + 1;
        auto msg_dword = [&](int off) -> uint32_t {
            if (off + 4 <= DATA) return msg_lds[off >> 2];
            uint32_t v = off == DATA ? 0x1Fu : 0u;
            if (off + 4 == BLOCKS * SHAKE256_RATE) v |= 0x80000000u;
            return v;
        };
        uint32_t v = 0;
        const bool absorbs = c.active && c.word < SHAKE256_RATE / 8;
#pragma unroll 1
        for (int blk = 0; blk < BLOCKS; blk++) {
            if (absorbs) {
                const int off = blk * SHAKE256_RATE + 8 * c.word;
                v ^= coop2_from_lohi(msg_dword(off), msg_dword(off + 4), c);
            }
            keccak_f1600_coop2(v, c);
        }
        uint32_t lo, hi;
        coop2_to_lohi(v, lane, lo, hi);
        if (c.active && c.word < CT / 8) {
            const uint32_t mine = lane < 32 ? lo : hi;
            ct_lds[2 * c.word + (lane >> 5)] = mine;
            *reinterpret_cast<uint32_t*>(A.ctilde + g * 64 + 8 * c.word + 4 * (lane >> 5)) = mine;
        }
        wave_lds_sync();
    }
    // 16: c <- SampleInBall(c~); 17: c_hat <- NTT(c)                      ml_dsa.rs:237-240
    const uint32_t creg = sample_in_ball_coop2<CT>(reinterpret_cast<const uint8_t*>(ct_lds), A.tau, bw_lds, lane, c);
    reinterpret_cast<uint32_t*>(A.c8)[g * 64 + lane] = creg;
    int32_t r[4] = {(int8_t)(creg & 0xFF), (int8_t)((creg >> 8) & 0xFF), (int8_t)((creg >> 16) & 0xFF), (int8_t)(creg >> 24)};
    ntt_fwd_wave(r, ftw, lane);
    store_packed(r, A.c_hat + g * (size_t)N, lane);
}

int launch_sign_front_small(mldsa_ctx* ctx, const mldsa_params* p, const SmallSignFrontArgs& A, bool a_packed, hipStream_t s) {
    (void)ctx;
    if (A.rows_cap == 0) return MLDSA_OK;
    const int nb = (p->l + SMW - 1) / SMW;
    const dim3 grid((unsigned)((((size_t)A.rows_cap + 7) / 8) * 8 * (size_t)nb)), block(64 * SMW);
#define MLDSA_SF(KK, LL, GB, G2, CT)                                                                                       \
    do {                                                                                                                   \
        if (a_packed) hipLaunchKernelGGL((k_sign_front_small<KK, LL, GB, G2, CT, true>), grid, block, 0, s, A);              \
        else hipLaunchKernelGGL((k_sign_front_small<KK, LL, GB, G2, CT, false>), grid, block, 0, s, A);                      \
    } while (0)
    if (p->set == MLDSA_44) MLDSA_SF(4, 4, 17, false, 32);
    else if (p->set == MLDSA_65) MLDSA_SF(6, 5, 19, true, 48);
    else MLDSA_SF(8, 7, 19, true, 64);
#undef MLDSA_SF
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
