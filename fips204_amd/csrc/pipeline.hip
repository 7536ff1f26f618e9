// Op-level sequencing: the batch orchestrator that reproduces the composition of the
// reference's callers on the device, chunk by chunk.
//   verify_batch  = verify_internal  (src/ml_dsa.rs:351-437)
// Intermediates live in one context-owned workspace that grows on demand (sized for HBM:
// a 65536-op chunk of ML-DSA-87 needs ~4.5 GiB, mostly A_hat; fixed per-kernel latencies are
// amortised over whole-batch launches).
#include <cstdlib>

#include "ctx.h"

namespace mldsa {

constexpr size_t CHUNK_OPS = 65536;

int ensure_workspace(mldsa_ctx *ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return MLDSA_OK;
    if (ctx->ws) {
        MLDSA_HIP_CHECK(hipDeviceSynchronize());
        (void)hipMemset(ctx->ws, 0, ctx->ws_bytes);  // may hold secrets of a previous sign call
        MLDSA_HIP_CHECK(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "workspace allocation", e);
    ctx->ws_bytes = bytes;
    return MLDSA_OK;
}

namespace {
struct Carver {
    uint8_t *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<uint8_t *>(p)) {}
    template <class T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct VerifyWs {
    int32_t *a_hat, *c, *znorm, *hvalid, *ctx_bad;
    uint32_t *hmask;
    uint8_t *mu_w1, *ctilde_p;
    size_t bytes;
    VerifyWs(void *base, const mldsa_params *p, size_t n, bool own_a_hat) {
        Carver cv(base);
        a_hat = cv.take<int32_t>(own_a_hat ? n * p->k * p->l * N : 0);
        c = cv.take<int32_t>(n * N);
        znorm = cv.take<int32_t>(n);
        hvalid = cv.take<int32_t>(n);
        ctx_bad = cv.take<int32_t>(n);
        hmask = cv.take<uint32_t>(n * p->k * 8);
        mu_w1 = cv.take<uint8_t>(n * (size_t)(64 + p->w1_len));  // mu || w1_encode(w1') per op
        ctilde_p = cv.take<uint8_t>(n * 64);
        bytes = cv.off + 256;
    }
};
}  // namespace

#define TRY(expr) do { int _rc = (expr); if (_rc != MLDSA_OK) return _rc; } while (0)
#define STAGE(name, expr) do { ProfScope _ps(ctx, s, name); TRY(expr); } while (0)

// verify_internal (ml_dsa.rs:351-437) for n_ops independent (key, message, signature) triples
int verify_batch(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr, const int32_t *t1,
                 const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
                 const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops, hipStream_t s,
                 const int32_t *a_hat_keys) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "verify: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    const size_t chunk = n_ops < CHUNK_OPS ? n_ops : CHUNK_OPS;
    TRY(ensure_workspace(ctx, VerifyWs(nullptr, p, chunk, a_hat_keys == nullptr).bytes));
    const size_t mw = (size_t)(64 + p->w1_len);
    const size_t kl_coeffs = (size_t)(p->k * p->l) * N;
    for (size_t o = 0; o < n_ops; o += chunk) {
        const size_t n = (n_ops - o) < chunk ? (n_ops - o) : chunk;
        VerifyWs w(ctx->ws, p, chunk, a_hat_keys == nullptr);
        const uint8_t *sg = sigs + o * (size_t)p->sig_len;
        const uint32_t *kidx = key_idx ? key_idx + o : nullptr;
        const size_t key_base = key_idx ? 0 : o;  // identity mapping: op i uses key i
        MLDSA_HIP_CHECK(hipMemsetAsync(w.znorm, 0, n * sizeof(int32_t), s));
        // fork: the small lane-per-op kernels are latency-bound (1-2 Keccak-f per op, <= 1 wave per SIMD) and
        // independent of ExpandA, so they run on the context's second stream underneath it
        hipStream_t aux = ctx->aux_stream;
        MLDSA_HIP_CHECK(hipEventRecord(ctx->fork_ev, s));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(aux, ctx->fork_ev, 0));
        {
            // 2: (c_tilde, z, h) <- sigDecode(sigma): hints here, z inside k_verify_main      ml_dsa.rs:368-376
            ProfScope ps(ctx, aux, "hint_unpack");
            TRY(launch_hint_unpack(ctx, p, sg, w.hmask, w.hvalid, n, aux));
        }
        {
            // 7: mu <- H(tr || M', 64)                                        ml_dsa.rs:386-397
            ProfScope ps(ctx, aux, "mu");
            TRY(launch_mu(ctx, tr + key_base * 64, 64, kidx, mode, msgs, msg_off + o, ctxs, ctx_off ? ctx_off + o : nullptr,
                          w.mu_w1, mw, w.ctx_bad, n, aux));
        }
        {
            // 8: c <- SampleInBall(c_tilde)                                   ml_dsa.rs:400
            ProfScope ps(ctx, aux, "sample_in_ball");
            TRY(launch_sample_in_ball(ctx, set, sg, (size_t)p->sig_len, w.c, n, aux));
        }
        MLDSA_HIP_CHECK(hipEventRecord(ctx->join_ev, aux));
        // 5: A_hat <- ExpandA(rho)                                        ml_dsa.rs:406
        // (skipped when the caller keeps A_hat with its keys: the optimisation the reference's benches/README.md
        //  names as missing; the rows are then looked up by key instead of by op)
        if (!a_hat_keys) STAGE("expand_a", launch_expand_a(ctx, set, rho + key_base * 32, 32, kidx, w.a_hat, n, s, true));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->join_ev, 0));  // join
        // 9-10: w1' <- UseHint(h, invNTT(A_hat o NTT(z) - NTT(c) o NTT(t1 2^d))), w1Encode   ml_dsa.rs:407-428
        STAGE("verify_main", launch_verify_main(ctx, p, a_hat_keys ? a_hat_keys + key_base * kl_coeffs : w.a_hat, sg, w.c,
                                                t1 + key_base * (size_t)p->k * N, kidx, w.hmask, w.mu_w1 + 64, mw, w.znorm, n, s,
                                                a_hat_keys != nullptr, a_hat_keys == nullptr));
        // 12: c_tilde' <- H(mu || w1Encode(w1'), lambda/4)                 ml_dsa.rs:429-431
        STAGE("ctilde_hash", launch_shake256_2(ctx, p->ctilde_len, w.mu_w1, mw, (int)mw, nullptr, nullptr, 0, 0, 0, 0, w.ctilde_p, 64, n, s));
        // 13: [[ ||z|| < gamma1 - beta ]] and [[ c_tilde = c_tilde' ]]      ml_dsa.rs:434-436
        STAGE("verdict", launch_verify_verdict(ctx, p, sg, w.ctilde_p, 64, w.znorm, w.hvalid, w.ctx_bad, ok + o, n, s));
    }
    return MLDSA_OK;
}


// ------------------------------------------------------------------------------------
// PublicKey::try_from_bytes -> expand_public (ml_dsa.rs:477-498)
int pk_expand_batch(mldsa_ctx *ctx, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1, size_t n, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "pk_expand: unknown parameter set");
    if (n == 0) return MLDSA_OK;
    MLDSA_HIP_CHECK(hipMemcpy2DAsync(rho, 32, pk, (size_t)p->pk_len, 32, n, hipMemcpyDeviceToDevice, s));
    TRY(launch_shake256_2(ctx, 64, pk, (size_t)p->pk_len, p->pk_len, nullptr, nullptr, 0, 0, 0, 0, tr, 64, n, s));  // tr = H(pk)
    // t1_d2_hat_mont = ntt(t1) * 2^13 * 2^32  (ml_dsa.rs:492-495)
    TRY(launch_unpack_ntt(ctx, pk, (size_t)p->pk_len, 32, 10, -1, 6346488 /* 2^13 * 2^64 mod q */, t1, p->k, n, s));
    return MLDSA_OK;
}

// PrivateKey::try_from_bytes -> expand_private (ml_dsa.rs:445-469)
int sk_expand_batch(mldsa_ctx *ctx, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k, uint8_t *tr, int32_t *s1,
                    int32_t *s2, int32_t *t0, size_t n, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "sk_expand: unknown parameter set");
    if (n == 0) return MLDSA_OK;
    const size_t skl = (size_t)p->sk_len;
    const int eb = p->eta == 2 ? 3 : 4;
    MLDSA_HIP_CHECK(hipMemcpy2DAsync(rho, 32, sk, skl, 32, n, hipMemcpyDeviceToDevice, s));
    MLDSA_HIP_CHECK(hipMemcpy2DAsync(cap_k, 32, sk + 32, skl, 32, n, hipMemcpyDeviceToDevice, s));
    MLDSA_HIP_CHECK(hipMemcpy2DAsync(tr, 64, sk + 64, skl, 64, n, hipMemcpyDeviceToDevice, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128, eb, p->eta, R2_MOD_Q, s1, p->l, n, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128 + (size_t)p->l * 32 * eb, eb, p->eta, R2_MOD_Q, s2, p->k, n, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128 + (size_t)(p->l + p->k) * 32 * eb, 13, 1 << 12, R2_MOD_Q, t0, p->k, n, s));
    return MLDSA_OK;
}

// KG::keygen_from_seed -> key_gen_internal (ml_dsa.rs:57-134) + into_bytes: xi -> (pk, sk) bytes
namespace {
struct KeygenWs {
    uint8_t *hbuf;
    int32_t *s1s2, *a_hat, *as1;
    size_t bytes;
    KeygenWs(void *base, const mldsa_params *p, size_t n) {
        Carver cv(base);
        hbuf = cv.take<uint8_t>(n * 128);
        s1s2 = cv.take<int32_t>(n * (size_t)(p->l + p->k) * N);
        a_hat = cv.take<int32_t>(n * (size_t)(p->k * p->l) * N);
        as1 = cv.take<int32_t>(n * (size_t)p->k * N);
        bytes = cv.off + 256;
    }
};
}  // namespace

int keygen_batch(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "keygen: unknown parameter set");
    if (n_keys == 0) return MLDSA_OK;
    const size_t chunk = n_keys < CHUNK_OPS ? n_keys : CHUNK_OPS;
    TRY(ensure_workspace(ctx, KeygenWs(nullptr, p, chunk).bytes));
    const size_t pkl = (size_t)p->pk_len, skl = (size_t)p->sk_len;
    for (size_t o = 0; o < n_keys; o += chunk) {
        const size_t n = (n_keys - o) < chunk ? (n_keys - o) : chunk;
        KeygenWs w(ctx->ws, p, chunk);
        uint8_t *pko = pk + o * pkl, *sko = sk + o * skl;
        // 1: (rho, rho', K) <- H(xi || k || l, 128)                          ml_dsa.rs:68-74
        STAGE("seed_hash", launch_shake256_2(ctx, 128, xi + o * 32, 32, 32, nullptr, nullptr, 0, 0, (uint32_t)p->k | ((uint32_t)p->l << 8), 2,
                                             w.hbuf, 128, n, s));
        STAGE("expand_s", launch_expand_s(ctx, set, w.hbuf + 32, 128, w.s1s2, n, s));                     // :79
        STAGE("expand_a", launch_expand_a(ctx, set, w.hbuf, 128, nullptr, w.a_hat, n, s, true));          // :85 (24-bit form)
        // :86-88 inv_ntt(A * ntt(s1)); s1 is read in place from the (s1, s2) rows ExpandS wrote
        STAGE("sign_w", launch_sign_w(ctx, set, w.a_hat, nullptr, w.s1s2, w.as1, nullptr, 0, n, s, (size_t)(p->l + p->k), nullptr, true));
        STAGE("keygen_encode", launch_keygen_encode(ctx, p, w.s1s2, w.as1, w.hbuf, pko, sko, n, s));  // :88-92, pk/sk encode incl. rho, K
        STAGE("tr_hash", launch_shake256_2(ctx, 64, pko, pkl, p->pk_len, nullptr, nullptr, 0, 0, 0, 0, sko + 64, skl, n, s));  // tr = H(pk), :99-101
    }
    return MLDSA_OK;
}

// ------------------------------------------------------------------------------------
// Signer::try_sign_* -> sign_internal (ml_dsa.rs:153-337) with the rejection loop re-batched:
// every round runs one loop iteration for all unfinished ops of a sub-batch, then compacts its
// active list.  Sub-batches run on several streams ("lanes") at different phases of the loop,
// so the short, latency-bound late rounds of one lane fill the machine under the wide early
// rounds of another; a lane that finishes takes the next sub-batch of the call.
namespace {
constexpr size_t SIGN_CHUNK_OPS = 65536;     // ops resident per call pass (workspace size), all lanes together
constexpr size_t SPEC_TARGET_SLOTS = 65536;  // upper bound of candidate slots per speculative round, all lanes together

struct SignWs {
    int32_t *a_hat, *y, *w, *c, *done, *ctx_bad, *accept;
    uint8_t *rnd_mu, *rho_pp, *w1, *ctilde, *stage, *wrisk, *yrisk;
    uint16_t *kappa, *slot_kappa;
    uint32_t *act0, *act1, *slot_op, *slot_key, *counter;
    size_t bytes = 0, stage_stride = 0;
    SignWs() = default;
    // n = ops of the sub-batch, spec_slots = cap of candidate slots in a speculative round
    SignWs(void *base, const mldsa_params *p, size_t n, size_t spec_slots, bool own_a_hat) {
        Carver cv(base);
        const size_t ns = n > spec_slots ? n : spec_slots;  // slots per round
        stage_stride = ((size_t)p->sig_len + 15) & ~(size_t)15;
        a_hat = cv.take<int32_t>(own_a_hat ? n * (size_t)(p->k * p->l) * N : 0);
        y = cv.take<int32_t>(ns * (size_t)p->l * N);
        w = cv.take<int32_t>(ns * (size_t)p->k * N);
        c = cv.take<int32_t>(ns * (size_t)N);
        done = cv.take<int32_t>(n);
        ctx_bad = cv.take<int32_t>(n);
        accept = cv.take<int32_t>(ns);
        rnd_mu = cv.take<uint8_t>(n * 96);  // rnd || mu per op: H(K || rnd || mu) input, ml_dsa.rs:199
        rho_pp = cv.take<uint8_t>(n * 64);
        w1 = cv.take<uint8_t>(ns * (size_t)p->w1_len);
        ctilde = cv.take<uint8_t>(ns * 64);
        wrisk = cv.take<uint8_t>(ns);
        yrisk = cv.take<uint8_t>(ns * (size_t)p->l);
        stage = cv.take<uint8_t>(spec_slots * stage_stride);
        kappa = cv.take<uint16_t>(n);
        slot_kappa = cv.take<uint16_t>(ns);
        act0 = cv.take<uint32_t>(n);
        act1 = cv.take<uint32_t>(n);
        slot_op = cv.take<uint32_t>(ns);
        slot_key = cv.take<uint32_t>(ns);
        counter = cv.take<uint32_t>(64);
        bytes = (cv.off + 511) & ~(size_t)255;
    }
};

struct SignLane {
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    volatile uint32_t *h_count = nullptr;
    SignWs w;
    size_t o = 0, n = 0, m = 0;
    uint32_t *act = nullptr, *act_next = nullptr;
    bool live = false, in_round = false;
};

int env_int(const char *name, long lo, long hi, long dflt) {
    if (const char *e = getenv(name)) {
        const long v = atol(e);
        if (v >= lo && v <= hi) return (int)v;
    }
    return (int)dflt;
}
}  // namespace

int sign_batch(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
               const int32_t *s1, const int32_t *s2, const int32_t *t0, const uint32_t *key_idx, const uint8_t *msgs,
               const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs,
               int32_t *status, size_t n_ops, hipStream_t s, const int32_t *a_hat_keys) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "sign: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    // lanes: sub-batches of at least MIN_LANE_OPS ops each (a narrower lane only adds launches)
    constexpr size_t MIN_LANE_OPS = 2048;
    int n_lanes = env_int("MLDSA_SIGN_LANES", 1, MLDSA_SIGN_MAX_LANES, 1);
    while (n_lanes > 1 && (n_ops + n_lanes - 1) / n_lanes < MIN_LANE_OPS) n_lanes--;
    const size_t resident = n_ops < SIGN_CHUNK_OPS ? n_ops : SIGN_CHUNK_OPS;
    const size_t lane_ops = (resident + n_lanes - 1) / n_lanes;
    // candidate slots per speculative round and candidates per op per round, per lane
    const size_t spec_target = (size_t)env_int("MLDSA_SPEC_TARGET", 1, SPEC_TARGET_SLOTS, 65536) / n_lanes;
    const int spec_max = env_int("MLDSA_SPEC_MAX", 1, 64, 32);  // k_resolve scans one wave of candidates
    const bool own_a = a_hat_keys == nullptr;
    const size_t kl_coeffs = (size_t)(p->k * p->l) * N;
    const size_t lane_bytes = SignWs(nullptr, p, lane_ops, spec_target, own_a).bytes;
    TRY(ensure_workspace(ctx, lane_bytes * n_lanes));

    SignLane lanes[MLDSA_SIGN_MAX_LANES];
    size_t next_op = 0;
    int rc = MLDSA_OK;
#define TRYC(expr) do { rc = (expr); if (rc != MLDSA_OK) return rc; } while (0)
#define STAGEC(name, expr) do { { ProfScope _ps(ctx, st, name); rc = (expr); } if (rc != MLDSA_OK) return rc; } while (0)
#define HIPC(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, #expr, _e); } while (0)

    // steps 1-8 of Algorithm 7 for the next sub-batch, on the lane's stream
    auto start_sub_batch = [&](SignLane &L) -> int {
        hipStream_t st = L.st;
        L.o = next_op;
        L.n = (n_ops - next_op) < lane_ops ? (n_ops - next_op) : lane_ops;
        next_op += L.n;
        const SignWs &w = L.w;
        const size_t o = L.o, n = L.n;
        const uint32_t *kidx = key_idx ? key_idx + o : nullptr;
        const size_t key_base = key_idx ? 0 : o;
        HIPC(hipMemsetAsync(sigs + o * (size_t)p->sig_len, 0, n * (size_t)p->sig_len, st));
        // 5: A_hat <- ExpandA(rho), once per signature                        ml_dsa.rs:181
        if (own_a) STAGEC("expand_a", launch_expand_a(ctx, set, rho + key_base * 32, 32, kidx, w.a_hat, n, st, true));
        // 6: mu <- H(tr || M', 64)                                            ml_dsa.rs:185-196
        STAGEC("mu", launch_mu(ctx, tr + key_base * 64, 64, kidx, mode, msgs, msg_off + o, ctxs, ctx_off ? ctx_off + o : nullptr,
                               w.rnd_mu + 32, 96, w.ctx_bad, n, st));
        HIPC(hipMemcpy2DAsync(w.rnd_mu, 96, rnd + o * 32, 32, 32, n, hipMemcpyDeviceToDevice, st));
        // 7: rho'' <- H(K || rnd || mu, 64)                                   ml_dsa.rs:199-201
        STAGEC("rho_pp_hash", launch_shake256_2(ctx, 64, cap_k + key_base * 32, 32, 32, kidx, w.rnd_mu, 96, 96, 0, 0, w.rho_pp, 64, n, st));
        // 8: kappa <- 0; active = all ops with a legal ctx
        HIPC(hipMemsetAsync(w.counter, 0, sizeof(uint32_t), st));
        TRYC(launch_init_active(ctx, n, w.ctx_bad, w.done, w.kappa, status ? status + o : nullptr, w.act0, w.counter, st));
        HIPC(hipMemcpyAsync((void *)L.h_count, w.counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC(hipEventRecord(L.ev, st));
        L.act = w.act0;
        L.act_next = w.act1;
        L.live = true;
        L.in_round = false;
        return MLDSA_OK;
    };

    // one pass of the rejection loop (steps 10-33) for the lane's L.m unfinished ops
    auto enqueue_round = [&](SignLane &L) -> int {
        hipStream_t st = L.st;
        const SignWs &w = L.w;
        const size_t o = L.o, m = L.m;
        const uint32_t *kidx = key_idx ? key_idx + o : nullptr;
        const size_t key_base = key_idx ? 0 : o;
        uint8_t *sg = sigs + o * (size_t)p->sig_len;
        // candidates per op this round (1 while the active set is wide)
        int spec = 1;
        if (m * 2 <= spec_target) {
            const size_t sp = spec_target / m;
            spec = sp > (size_t)spec_max ? spec_max : (int)sp;
        }
        const size_t ns = m * (size_t)spec;
        if (ctx->prof_on) ctx->prof_sign_slots += ns;
        STAGEC("make_slots", launch_make_slots(ctx, L.act, m, spec, w.kappa, p->l, w.slot_op, w.slot_kappa, st, kidx,
                                               own_a ? nullptr : w.slot_key, w.counter));
        // 11: y <- ExpandMask(rho'', kappa)                               :215
        STAGEC("expand_mask", launch_expand_mask(ctx, set, w.rho_pp, 64, w.slot_kappa, 1, w.slot_op, w.y, ns, st, w.yrisk));
        // 12: w <- invNTT(A_hat o NTT(y))                                 :218-222
        STAGEC("sign_w", launch_sign_w(ctx, set, own_a ? w.a_hat : a_hat_keys + key_base * kl_coeffs, own_a ? w.slot_op : w.slot_key,
                                       w.y, w.w, w.w1, (size_t)p->w1_len, ns, st, 0, w.wrisk, own_a));
        // 13-15: w1 <- HighBits(w), w1Encode: in the epilogue of sign_w; c_tilde <- H(mu || w1)   :225-234
        STAGEC("ctilde_hash", launch_shake256_2(ctx, p->ctilde_len, w.rnd_mu + 32, 96, 64, w.slot_op, w.w1, (size_t)p->w1_len, p->w1_len,
                                                0, 0, w.ctilde, 64, ns, st));
        // 16: c <- SampleInBall(c_tilde)                                  :237
        STAGEC("sample_in_ball", launch_sample_in_ball(ctx, set, w.ctilde, 64, w.c, ns, st));
        // 17: c_hat <- NTT(c), in place                                   :240
        STAGEC("ntt_c", launch_ntt(ctx, w.c, w.c, ns, st));
        // 18-33: <<c s1>>, <<c s2>>, z, r0, checks, <<c t0>>, h, checks, sigEncode   :240-336
        STAGEC("sign_tail", launch_sign_tail(ctx, p, w.c, w.y, w.w, w.ctilde, w.slot_op, kidx, s1 + key_base * (size_t)p->l * N,
                                             s2 + key_base * (size_t)p->k * N, t0 + key_base * (size_t)p->k * N, w.kappa, w.done, sg,
                                             spec, w.stage, w.stage_stride, w.accept, ns, st, w.wrisk, w.yrisk));
        if (spec > 1)
            STAGEC("resolve", launch_resolve(ctx, p, L.act, m, spec, w.accept, w.stage, w.stage_stride, sg, w.done, w.kappa, st));
        STAGEC("compact", launch_compact(ctx, L.act, m, w.done, L.act_next, w.counter, st));
        HIPC(hipMemcpyAsync((void *)L.h_count, w.counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC(hipEventRecord(L.ev, st));
        L.in_round = true;
        return MLDSA_OK;
    };

    auto run = [&]() -> int {
        // inputs may have been produced on the caller's stream
        HIPC(hipEventRecord(ctx->fork_ev, s));
        for (int i = 0; i < n_lanes; i++) {
            SignLane &L = lanes[i];
            L.st = ctx->lane_stream[i];
            L.ev = ctx->lane_ev[i];
            L.h_count = ctx->h_lane_count + i;
            L.w = SignWs(static_cast<uint8_t *>(ctx->ws) + lane_bytes * i, p, lane_ops, spec_target, own_a);
            HIPC(hipStreamWaitEvent(L.st, ctx->fork_ev, 0));
            if (next_op < n_ops) TRYC(start_sub_batch(L));
        }
        int n_live = 0;
        for (int i = 0; i < n_lanes; i++) n_live += lanes[i].live;
        while (n_live > 0) {  // 10: while (z, h) = bottom                      ml_dsa.rs:212
            bool progressed = false;
            for (int i = 0; i < n_lanes; i++) {
                SignLane &L = lanes[i];
                if (!L.live) continue;
                const hipError_t q = hipEventQuery(L.ev);
                if (q == hipErrorNotReady) continue;
                if (q != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "sign: lane event", q);
                progressed = true;
                L.m = *L.h_count;
                if (L.in_round) { uint32_t *t = L.act; L.act = L.act_next; L.act_next = t; }
                if (L.m > 0) {
                    TRYC(enqueue_round(L));
                } else if (next_op < n_ops) {
                    TRYC(start_sub_batch(L));
                } else {
                    L.live = false;
                    n_live--;
                }
            }
            if (!progressed) {
                // nothing finished yet: block on the first live lane instead of spinning on the queries
                for (int i = 0; i < n_lanes; i++)
                    if (lanes[i].live) { HIPC(hipEventSynchronize(lanes[i].ev)); break; }
            }
        }
        return MLDSA_OK;
    };
    rc = run();
#undef TRYC
#undef STAGEC
#undef HIPC
    // join every lane (also on the error path), then clear y, rho'', cs1/cs2: they are
    // secret-dependent (the reference zeroizes on drop, types.rs:19)
    // (A_hat = ExpandA(rho) is public and is the first and largest carve of each lane: skipped.)
    for (int i = 0; i < n_lanes; i++)
        if (lanes[i].st) (void)hipStreamSynchronize(lanes[i].st);
    for (int i = 0; i < n_lanes; i++) {
        uint8_t *lane_base = static_cast<uint8_t *>(ctx->ws) + lane_bytes * i;
        uint8_t *secrets = lanes[i].st ? reinterpret_cast<uint8_t *>(lanes[i].w.y) : lane_base;
        (void)hipMemsetAsync(secrets, 0, (size_t)(lane_base + lane_bytes - secrets), s);
    }
    (void)hipStreamSynchronize(s);
    return rc;
}

}  // namespace mldsa
