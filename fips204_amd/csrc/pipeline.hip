// Op-level sequencing: the batch orchestrator that reproduces the composition of the
// reference's callers on the device, chunk by chunk.
//   verify_batch  = verify_internal  (src/ml_dsa.rs:351-437)
// Intermediates live in one context-owned workspace that grows on demand (sized for HBM:
// a 16384-op chunk of ML-DSA-87 needs ~1.1 GiB, mostly A_hat).
#include "ctx.h"

namespace mldsa {

constexpr size_t CHUNK_OPS = 16384;

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
    int32_t *a_hat, *z, *c, *wp, *znorm, *hvalid, *ctx_bad;
    uint32_t *hmask;
    uint8_t *mu_w1, *ctilde_p;
    size_t bytes;
    VerifyWs(void *base, const mldsa_params *p, size_t n) {
        Carver cv(base);
        a_hat = cv.take<int32_t>(n * p->k * p->l * N);
        z = cv.take<int32_t>(n * p->l * N);
        c = cv.take<int32_t>(n * N);
        wp = cv.take<int32_t>(n * p->k * N);
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

// verify_internal (ml_dsa.rs:351-437) for n_ops independent (key, message, signature) triples
int verify_batch(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr, const int32_t *t1,
                 const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
                 const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "verify: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    const size_t chunk = n_ops < CHUNK_OPS ? n_ops : CHUNK_OPS;
    TRY(ensure_workspace(ctx, VerifyWs(nullptr, p, chunk).bytes));
    const size_t mw = (size_t)(64 + p->w1_len);
    for (size_t o = 0; o < n_ops; o += chunk) {
        const size_t n = (n_ops - o) < chunk ? (n_ops - o) : chunk;
        VerifyWs w(ctx->ws, p, chunk);
        const uint8_t *sg = sigs + o * (size_t)p->sig_len;
        const uint32_t *kidx = key_idx ? key_idx + o : nullptr;
        const size_t key_base = key_idx ? 0 : o;  // identity mapping: op i uses key i
        MLDSA_HIP_CHECK(hipMemsetAsync(w.znorm, 0, n * sizeof(int32_t), s));
        // 2: (c_tilde, z, h) <- sigDecode(sigma)                         ml_dsa.rs:368-376
        TRY(launch_sig_unpack_z(ctx, p, sg, w.z, w.znorm, n, s));
        TRY(launch_hint_unpack(ctx, p, sg, w.hmask, w.hvalid, n, s));
        // 7: mu <- H(tr || M', 64)                                        ml_dsa.rs:386-397
        TRY(launch_mu(ctx, tr + key_base * 64, 64, kidx, mode, msgs, msg_off + o, ctxs, ctx_off ? ctx_off + o : nullptr,
                      w.mu_w1, mw, w.ctx_bad, n, s));
        // 8: c <- SampleInBall(c_tilde)                                   ml_dsa.rs:400
        TRY(launch_sample_in_ball(ctx, set, sg, (size_t)p->sig_len, w.c, n, s));
        // 5: A_hat <- ExpandA(rho)                                        ml_dsa.rs:406
        TRY(launch_expand_a(ctx, set, rho + key_base * 32, 32, kidx, w.a_hat, n, s));
        // 9: w'_approx <- invNTT(A_hat o NTT(z) - NTT(c) o NTT(t1 2^d))   ml_dsa.rs:407-416
        TRY(launch_verify_arith(ctx, set, w.a_hat, w.z, w.c, t1 + key_base * (size_t)p->k * N, kidx, w.wp, n, s));
        // 10: w1' <- UseHint(h, w'_approx); w1Encode                       ml_dsa.rs:420-428
        TRY(launch_use_hint_w1(ctx, p, w.wp, w.hmask, w.mu_w1 + 64, mw, n, s));
        // 12: c_tilde' <- H(mu || w1Encode(w1'), lambda/4)                 ml_dsa.rs:429-431
        TRY(launch_shake256_2(ctx, p->ctilde_len, w.mu_w1, mw, (int)mw, nullptr, nullptr, 0, 0, 0, 0, w.ctilde_p, 64, n, s));
        // 13: [[ ||z|| < gamma1 - beta ]] and [[ c_tilde = c_tilde' ]]      ml_dsa.rs:434-436
        TRY(launch_verify_verdict(ctx, p, sg, w.ctilde_p, 64, w.znorm, w.hvalid, w.ctx_bad, ok + o, n, s));
    }
    return MLDSA_OK;
}

}  // namespace mldsa
