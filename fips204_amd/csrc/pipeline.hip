// Op-level sequencing: the batch orchestrator that reproduces the composition of the
// reference's callers on the device, chunk by chunk.
//   verify_batch  = verify_internal  (src/ml_dsa.rs:351-437)
//   sign_batch    = sign_internal    (src/ml_dsa.rs:153-337)
//   keygen_batch  = key_gen_internal (src/ml_dsa.rs:57-134) + into_bytes
// Intermediates live in one context-owned workspace (sized for HBM: a 65536-op chunk of ML-DSA-87 needs
// ~4.5 GiB, mostly A_hat; fixed per-kernel latencies are amortised over whole-batch launches).
//
// Every pipeline is a pure ENQUEUE function: no allocation, no host synchronisation, no host decision that
// depends on device data.  That makes a call capturable: run_op() replays a repeated call shape as a
// hipGraph (one graph launch instead of 10 ... 100 kernel launches) and falls back to direct launches for
// shapes it has not seen twice.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

#include <chrono>
#include <thread>

#include "ctx.h"

namespace mldsa {

// verify / keygen ops resident per pass: 131 072 instead of 65 536 is +0.8 / +3 / +6.5 % for ML-DSA-87 / 65 / 44 on calls of
// that size or more (the lane-per-op hash and mu kernels run two waves per SIMD instead of one; half as many launch tails)
// (the default of mldsa_ctx::pass_ops: a context whose device cannot hold that workspace halves it, reserve_workspace)
// MLDSA_OPT_GRAPHS = 1 replays signing calls of up to this many ops (run_op)
constexpr size_t GRAPH_AUTO_MAX_OPS = 16384;
// a context never shrinks its passes below this (reserve_workspace)
constexpr size_t MIN_PASS_OPS = 1024, MIN_SPEC_ROWS = 1024;

// Workspace for `n_ops` ops of operation `op`.  The pass sizes above are tuned for a whole MI355X (a 262 144-op ML-DSA-87 signing
// pass takes 15 GB of the 288); on a device that cannot give that much -- smaller, partitioned, or shared with other work --
// the context falls back to smaller passes (more of them per call, results identical) instead of failing the call.
int reserve_workspace(mldsa_ctx *ctx, const mldsa_params *p, int op, size_t n_ops, bool own_a, size_t wire_table_keys, int wire_mode) {
    // The shrinking below is TENTATIVE: the context's pass sizes change for good only when a smaller layout is actually obtained.  A
    // call that cannot be served at any size -- a caller-owned buffer or MLDSA_OPT_WORKSPACE_CAP_MB too small even for the smallest
    // pass -- leaves the context exactly as it found it (one oversized call must not turn every later call into 1 024-op passes),
    // and the user's MLDSA_OPT_SPEC_TARGET is never written here: the effective target is min(option, opt_spec_rows) where it is used.
    const size_t pass0 = ctx->pass_ops, pass_sign0 = ctx->pass_ops_sign;
    const long rows0 = ctx->opt_spec_rows, cap0 = ctx->spec_target_cap;
    unsigned long long shrinks = 0;
    auto undo = [&] { ctx->pass_ops = pass0; ctx->pass_ops_sign = pass_sign0; ctx->opt_spec_rows = rows0; ctx->spec_target_cap = cap0; };
    for (;;) {
        const size_t bytes = op == MLDSA_OP_SIGN ? sign_workspace_bytes(ctx, p, n_ops, own_a)
                             : op == MLDSA_OP_VERIFY ? verify_workspace_bytes(ctx, p, n_ops, own_a, wire_table_keys, wire_mode)
                                                     : keygen_workspace_bytes(ctx, p, n_ops);
        const int rc = ensure_workspace(ctx, bytes);
        size_t &pass = op == MLDSA_OP_SIGN ? ctx->pass_ops_sign : ctx->pass_ops;
        if (rc != MLDSA_ERR_NOMEM) {
            if (rc == MLDSA_OK) ctx->stats.workspace_shrinks += shrinks;  // committed
            else undo();
            return rc;
        }
        // A signing pass holds one row set (y, w, c ...) per candidate of a speculative round -- up to opt_spec_rows rows
        // whatever the batch size: those come down first, to the size of the pass.  (Speculation only trades rounds for width:
        // the signatures do not depend on it.)
        const size_t resident = std::min(pass, n_ops);
        if (op == MLDSA_OP_SIGN && (size_t)ctx->opt_spec_rows > std::max(resident, MIN_SPEC_ROWS)) {
            ctx->opt_spec_rows = (long)std::max<size_t>((size_t)ctx->opt_spec_rows / 2, MIN_SPEC_ROWS);
            ctx->spec_target_cap = std::min(ctx->spec_target_cap, ctx->opt_spec_rows);  // (rows = max(target, rows): the target has to follow)
        } else {
            // The workspace follows min(n_ops, pass): halving helps only while that minimum can still come down.  A call of at most
            // MIN_PASS_OPS ops asks for the same bytes at every pass size: it fails at once.
            if (pass <= MIN_PASS_OPS || n_ops <= MIN_PASS_OPS) {
                undo();
                return rc;
            }
            // the pass size is halved until it is BELOW the call's size (halving a pass that is still larger than the call would
            // ask for the same bytes again); the call then runs in two or more passes.
            do pass /= 2; while (pass > MIN_PASS_OPS && pass >= n_ops);
        }
        shrinks++;
    }
}

// A larger (or no) caller-owned buffer, a raised or removed cap: the passes go back to what the context was configured with; the next
// call that does not fit shrinks them again (mldsa_ctx_set_workspace, MLDSA_OPT_WORKSPACE_CAP_MB).
void restore_pass_sizes(mldsa_ctx *ctx) {
    ctx->pass_ops = ctx->pass_ops_cfg;
    ctx->pass_ops_sign = ctx->pass_ops_sign_cfg;
    ctx->opt_spec_rows = ctx->opt_spec_rows_cfg;
    ctx->spec_target_cap = LONG_MAX;
}

std::shared_mutex &capture_mutex() {
    static std::shared_mutex m;
    return m;
}

int ensure_workspace(mldsa_ctx *ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return MLDSA_OK;
    if (ctx->ws_external) return set_error(MLDSA_ERR_NOMEM, "workspace: the caller's buffer (mldsa_ctx_set_workspace) is too small for a pass");
    // MLDSA_OPT_WORKSPACE_CAP_MB: a host that shares the device bounds what the context may take
    if (ctx->opt_ws_cap_bytes && bytes > ctx->opt_ws_cap_bytes) return set_error(MLDSA_ERR_NOMEM, "workspace: above MLDSA_OPT_WORKSPACE_CAP_MB");
    // growing replaces the buffer every captured graph points into: wait for whatever still runs, then drop them
    MLDSA_HIP_CHECK(device_sync_quiesced());  // (no other context of the process captures a stream while the device is waited for)
    // (a background clearing of the previous call's secrets has finished with the device: nothing is pending on the old buffer)
    ctx->zero_pending = ctx->zero_head_valid = ctx->zero_wait_after_ea = false;
    drop_graphs(ctx);
    if (ctx->ws) {
        MLDSA_WIPE(memset_quiesced(ctx->ws, 0, ctx->ws_bytes));  // may hold secrets of a previous sign / keygen call
        MLDSA_HIP_CHECK(free_quiesced(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    ctx->secret_spans.clear();
    hipError_t e = malloc_quiesced(&ctx->ws, bytes);
    if (e != hipSuccess) {
        ctx->ws = nullptr;
        (void)hipGetLastError();
        return set_error(MLDSA_ERR_NOMEM, "workspace allocation", e);
    }
    ctx->ws_bytes = bytes;
    ctx->stats.workspace_growths++;
    return MLDSA_OK;
}

// Secrets are cleared on every path out of a call (the reference zeroizes on drop, types.rs:19).  On the successful path the
// clearing runs on a helper stream, off the caller's critical path: `head` (optional, small) first with its own event, then `rest`;
// the next op-level call of the context waits for the events on the device.  If any step of that plumbing fails, the spans are
// cleared on the call's own stream and waited for instead -- a failure never leaves secrets behind.
struct ZeroSpan { void *p; size_t bytes; };
static bool clear_in_background(mldsa_ctx *ctx, hipStream_t s, const ZeroSpan *head, const ZeroSpan *rest, int n_rest) {
    hipStream_t z = parallel_stream(ctx, s);
    ctx->zero_stream = z;
    bool ok = hipEventRecord(ctx->zero_fork_ev, s) == hipSuccess && hipStreamWaitEvent(z, ctx->zero_fork_ev, 0) == hipSuccess;
    if (ok && head) {
        MLDSA_WIPE(launch_zero(ctx, head->p, head->bytes, z));
        ok = hipEventRecord(ctx->zero_head_ev, z) == hipSuccess;
    }
    if (ok) {
        for (int i = 0; i < n_rest; i++) MLDSA_WIPE(launch_zero(ctx, rest[i].p, rest[i].bytes, z));
        ok = hipEventRecord(ctx->zero_ev, z) == hipSuccess;
    }
    if (ok) return true;
    (void)hipGetLastError();
    if (head) MLDSA_WIPE(launch_zero(ctx, head->p, head->bytes, s));
    for (int i = 0; i < n_rest; i++) MLDSA_WIPE(launch_zero(ctx, rest[i].p, rest[i].bytes, s));
    (void)hipStreamSynchronize(z);  // whatever part of the background clearing did start
    (void)hipStreamSynchronize(s);
    return false;
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
    int32_t *a_hat, *c, *znorm, *hvalid, *ctx_bad, *key_bad;
    uint32_t *kidx;
    uint8_t *mu_w1;
    // mldsa_verify_pk: the expanded fields of `wire_keys` wire-format public keys (tr = H(pk), t1_d2_hat_mont), expand_public's outputs
    uint8_t *tr_w;
    int32_t *t1_w;
    size_t bytes;
    VerifyWs(void *base, const mldsa_params *p, size_t n, bool own_a_hat, size_t wire_keys = 0) {
        Carver cv(base);
        a_hat = cv.take<int32_t>(own_a_hat ? n * p->k * p->l * N : 0);
        tr_w = cv.take<uint8_t>(wire_keys * 64);
        t1_w = cv.take<int32_t>(wire_keys * (size_t)p->k * N);
        c = cv.take<int32_t>(n * (N / 4));  // one byte per coefficient (k_sample_in_ball<.., C8>)
        znorm = cv.take<int32_t>(n);
        hvalid = cv.take<int32_t>(n);
        ctx_bad = cv.take<int32_t>(n);
        key_bad = cv.take<int32_t>(n);
        kidx = cv.take<uint32_t>(n);
        mu_w1 = cv.take<uint8_t>(n * (size_t)(64 + p->w1_len));  // mu || w1_encode(w1') per op
        bytes = cv.off + 256;
    }
};
}  // namespace

#define TRY(expr) do { int _rc = (expr); if (_rc != MLDSA_OK) return _rc; } while (0)
#define STAGE(name, expr) do { ProfScope _ps(ctx, s, name); TRY(expr); } while (0)

// wire_keys: 0, or the key count of an mldsa_verify_pk call whose keys arrive in wire format (identity mapping: one per op of a pass)
static size_t wire_keys_of_pass(const mldsa_ctx *ctx, size_t n_ops, size_t wire_table_keys, bool wire, bool by_table) {
    return !wire ? 0 : by_table ? wire_table_keys : std::min(n_ops, ctx->pass_ops);
}
size_t verify_workspace_bytes(const mldsa_ctx *ctx, const mldsa_params *p, size_t n_ops, bool own_a, size_t wire_table_keys, int wire_mode) {
    return VerifyWs(nullptr, p, std::min(n_ops, ctx->pass_ops), own_a, wire_keys_of_pass(ctx, n_ops, wire_table_keys, wire_mode != 0, wire_mode == 2)).bytes;
}

// verify_internal (ml_dsa.rs:351-437) for n_ops independent (key, message, signature) triples
int verify_batch(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr, const int32_t *t1, size_t n_keys,
                 const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
                 const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops, hipStream_t s,
                 const int32_t *a_hat_keys, const uint8_t *pk_wire) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "verify: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    const size_t chunk = std::min(n_ops, ctx->pass_ops);
    // mldsa_verify_pk: keys in wire format.  PublicKey::try_from_bytes (expand_public, ml_dsa.rs:477-498) becomes part of the call:
    // rho is read where it lies in the key bytes, tr = H(pk) and t1_d2_hat_mont = NTT(t1) 2^13 are produced on the helper stream
    // underneath ExpandA -- per pass for the identity mapping (key i belongs to op i), once for a key table
    const bool wire = pk_wire != nullptr;
    const size_t pkl = (size_t)p->pk_len;
    const size_t wire_keys = wire_keys_of_pass(ctx, n_ops, n_keys, wire, key_idx != nullptr);
    if (ctx->ws_bytes < VerifyWs(nullptr, p, chunk, a_hat_keys == nullptr, wire_keys).bytes)
        return set_error(MLDSA_ERR_NOMEM, "verify: workspace not reserved");
    const size_t mw = (size_t)(64 + p->w1_len);
    const size_t kl_coeffs = (size_t)(p->k * p->l) * N;
    // A SMALL call is all latency: it runs as ONE launch (kernels_small.hip: per op a cluster of workgroups for ExpandA | mu | SampleInBall,
    // the last one to finish carries on with the arithmetic, the c~ hash and the verdict) instead of the six launches on three
    // streams below.  Same device code, same workspace rows, same verdicts (MLDSA_OPT_SMALL_FUSED: the largest such call; 0 = never).
    if (!wire && ctx->opt_coop_hash && n_ops <= small_ops_limit(ctx->opt_small_fused, p) && n_ops <= chunk) {
        VerifyWs w(ctx->ws, p, chunk, a_hat_keys == nullptr, wire_keys);
        STAGE("verify_small", launch_verify_small(ctx, p, mode, rho, 32, a_hat_keys, tr, t1, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok,
                                                  n_ops, w.a_hat, w.c, w.mu_w1, w.ctx_bad, ctx->d_small_ctr, s));
        return MLDSA_OK;
    }
    for (size_t o = 0; o < n_ops; o += chunk) {
        const size_t n = (n_ops - o) < chunk ? (n_ops - o) : chunk;
        VerifyWs w(ctx->ws, p, chunk, a_hat_keys == nullptr, wire_keys);
        const uint8_t *sg = sigs + o * (size_t)p->sig_len;
        // key_idx is checked against n_keys on the device: the kernels below only see in-range indices, ops
        // with a bad index are flagged (ok = 0).  Identity mapping (key_idx == NULL) was checked by the caller.
        const uint32_t *kidx = nullptr;
        const int32_t *key_bad = nullptr;
        const size_t key_base = key_idx ? 0 : o;  // identity mapping: op i uses key i
        // fork: the small lane-per-op kernels are latency-bound (1-2 Keccak-f per op, <= 1 wave per SIMD) and
        // independent of ExpandA, so they run on the context's second stream underneath it
        hipStream_t aux = parallel_stream(ctx, s);
        MLDSA_HIP_CHECK(hipEventRecord(ctx->fork_ev, s));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(aux, ctx->fork_ev, 0));
        if (key_idx) {  // (on the second stream too: ExpandA starts at once and clamps the raw indices itself)
            TRY(launch_sanitize_keys(ctx, key_idx + o, n_keys, n, w.kidx, w.key_bad, aux));
            kidx = w.kidx;
            key_bad = w.key_bad;
        }
        const uint8_t *tr_keys = tr ? tr + key_base * 64 : nullptr;
        const int32_t *t1_keys = t1 ? t1 + key_base * (size_t)p->k * N : nullptr;
        if (wire) {
            const size_t nk = key_idx ? n_keys : n;            // keys to deserialise now
            const uint8_t *pkp = pk_wire + key_base * pkl;
            if (!key_idx || o == 0) {
                ProfScope ps(ctx, aux, "pk_expand");
                TRY(launch_shake256_2(ctx, 64, pkp, pkl, p->pk_len, nullptr, nullptr, 0, 0, 0, 0, w.tr_w, 64, nk, aux));                 // tr = H(pk)
                TRY(launch_unpack_ntt(ctx, pkp, pkl, 32, 10, -1, 6346488 /* 2^13 * 2^64 mod q */, w.t1_w, p->k, nk, aux));             // ml_dsa.rs:492-495
            }
            tr_keys = w.tr_w;   // rows of the keys just expanded: by op for the identity mapping, by key index for a table
            t1_keys = w.t1_w;
        }
        hipStream_t sib_stream = s;
        {
            // 7: mu <- H(tr || M', 64)                                        ml_dsa.rs:386-397
            ProfScope ps(ctx, aux, "mu");
            TRY(launch_mu(ctx, tr_keys, 64, kidx, mode, msgs, msg_off, ctxs, ctx_off, w.mu_w1, mw, w.ctx_bad, n, aux, key_bad, o, n_ops));
        }
        {
            // 8: c <- SampleInBall(c_tilde)                                   ml_dsa.rs:400
            // Needs nothing but the signature.  With A_hat kept by the caller the first stream is idle until the join and takes it,
            // beside mu.  Otherwise it follows mu on the second stream under ExpandA -- except in a SMALL call (the size the
            // cooperative ExpandA takes: 22 us), where that chain (key check, mu, SampleInBall: 38 us) had become the longest one: a
            // third stream then (one-op verify 116 -> 111 us; same-box A/B: from 1 024 ops up a third stream costs 4-5 %, at 65 536
            // ops 0.3 %: not used there).
            hipStream_t cs = s;
            const bool small_call = ctx->opt_coop_hash && n * (size_t)(p->k * p->l) <= ctx->coop_a_max;
            if (!a_hat_keys && !small_call) cs = aux;
            else if (!a_hat_keys) {
                cs = parallel_stream(ctx, s, aux);
                if (cs != aux && cs != s) MLDSA_HIP_CHECK(hipStreamWaitEvent(cs, ctx->fork_ev, 0));
                else cs = aux;  // (no third stream to be had: behind mu as before)
            }
            ProfScope ps(ctx, cs, "sample_in_ball");
            TRY(launch_sample_in_ball(ctx, set, sg, (size_t)p->sig_len, w.c, n, cs, nullptr, true));
            sib_stream = cs;
        }
        if (sib_stream != s && sib_stream != aux) MLDSA_HIP_CHECK(hipEventRecord(ctx->join2_ev, sib_stream));
        MLDSA_HIP_CHECK(hipEventRecord(ctx->join_ev, aux));
        // 5: A_hat <- ExpandA(rho)                                        ml_dsa.rs:406
        // (skipped when the caller keeps A_hat with its keys: the optimisation the reference's benches/README.md
        //  names as missing; the rows are then looked up by key instead of by op)
        if (!a_hat_keys) {
            const uint8_t *rho_rows = wire ? pk_wire + key_base * pkl : rho + key_base * 32;  // pkEncode puts rho first (encodings.rs:29)
            STAGE("expand_a", launch_expand_a(ctx, set, rho_rows, wire ? pkl : 32, key_idx ? key_idx + o : nullptr, w.a_hat, n, s, true, key_idx ? n_keys : 0));
        }
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->join_ev, 0));  // join
        if (sib_stream != s && sib_stream != aux) MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->join2_ev, 0));
        // 2: (c_tilde, z, h) <- sigDecode(sigma), inside k_verify_main                      ml_dsa.rs:368-376
        // 9-10: w1' <- UseHint(h, invNTT(A_hat o NTT(z) - NTT(c) o NTT(t1 2^d))), w1Encode   ml_dsa.rs:407-428
        STAGE("verify_main", launch_verify_main(ctx, p, a_hat_keys ? a_hat_keys + key_base * kl_coeffs : w.a_hat, sg, w.c,
                                                t1_keys, kidx, w.hvalid, w.mu_w1 + 64, mw, w.znorm, n, s,
                                                a_hat_keys != nullptr, a_hat_keys == nullptr));
        // 12-13: c_tilde' <- H(mu || w1Encode(w1'), lambda/4); [[ ||z|| < gamma1 - beta ]] and [[ c_tilde = c_tilde' ]]   ml_dsa.rs:429-436
        STAGE("ctilde_hash", launch_ctilde_verdict(ctx, p, w.mu_w1, mw, sg, w.znorm, w.hvalid, w.ctx_bad, ok + o, n, s));
    }
    return MLDSA_OK;
}


// ------------------------------------------------------------------------------------
// PublicKey::try_from_bytes -> expand_public (ml_dsa.rs:477-498)
int pk_expand_batch(mldsa_ctx *ctx, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1, size_t n, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "pk_expand: unknown parameter set");
    if (n == 0) return MLDSA_OK;
    ProfScope ps(ctx, s, "pk_expand");  // (one event pair around the three launches)
    TRY(launch_copy_rows(ctx, rho, 32, pk, (size_t)p->pk_len, 32, n, s));
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
    ProfScope ps(ctx, s, "sk_expand");
    TRY(launch_copy_rows(ctx, rho, 32, sk, skl, 32, n, s));
    TRY(launch_copy_rows(ctx, cap_k, 32, sk + 32, skl, 32, n, s));
    TRY(launch_copy_rows(ctx, tr, 64, sk + 64, skl, 64, n, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128, eb, p->eta, R2_MOD_Q, s1, p->l, n, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128 + (size_t)p->l * 32 * eb, eb, p->eta, R2_MOD_Q, s2, p->k, n, s));
    TRY(launch_unpack_ntt(ctx, sk, skl, 128 + (size_t)(p->l + p->k) * 32 * eb, 13, 1 << 12, R2_MOD_Q, t0, p->k, n, s));
    return MLDSA_OK;
}

// PublicKey::into_bytes (lib.rs:478-493): pk = rho | SimpleBitPack(inv_ntt(mont_reduce(t1_d2_hat_mont)) >> 13, 10 bits)
int pk_into_bytes_batch(mldsa_ctx *ctx, int set, const uint8_t *rho, const int32_t *t1, uint8_t *pk, size_t n, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "pk_into_bytes: unknown parameter set");
    if (n == 0) return MLDSA_OK;
    TRY(launch_copy_rows(ctx, pk, (size_t)p->pk_len, rho, 32, 32, n, s));
    TRY(launch_key_intt(ctx, t1, p->k, n, 10, -1, pk, (size_t)p->pk_len, 32, nullptr, 0, 0, s));
    return MLDSA_OK;
}

// PrivateKey::into_bytes (lib.rs:427-465): sk = rho | K | tr | BitPack(s1, eta) | BitPack(s2, eta) | BitPack(t0, 2^12)
int sk_into_bytes_batch(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1,
                        const int32_t *s2, const int32_t *t0, uint8_t *sk, size_t n, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "sk_into_bytes: unknown parameter set");
    if (n == 0) return MLDSA_OK;
    const size_t skl = (size_t)p->sk_len;
    const int eb = p->eta == 2 ? 3 : 4;
    TRY(launch_copy_rows(ctx, sk, skl, rho, 32, 32, n, s));
    TRY(launch_copy_rows(ctx, sk + 32, skl, cap_k, 32, 32, n, s));
    TRY(launch_copy_rows(ctx, sk + 64, skl, tr, 64, 64, n, s));
    TRY(launch_key_intt(ctx, s1, p->l, n, eb, p->eta, sk, skl, 128, nullptr, 0, 0, s));
    TRY(launch_key_intt(ctx, s2, p->k, n, eb, p->eta, sk, skl, 128 + (size_t)p->l * 32 * eb, nullptr, 0, 0, s));
    TRY(launch_key_intt(ctx, t0, p->k, n, 13, 1 << 12, sk, skl, 128 + (size_t)(p->l + p->k) * 32 * eb, nullptr, 0, 0, s));
    return MLDSA_OK;
}

// KG::keygen_from_seed -> key_gen_internal (ml_dsa.rs:57-134) + into_bytes: xi -> (pk, sk) bytes
namespace {
struct KeygenWs {
    uint8_t *hbuf;
    int32_t *s1s2, *a_hat, *as1;
    size_t bytes, secret_bytes;  // secret_bytes: hbuf .. end of as1 (rho' / K, s1, s2, A s1)
    // byte_rows: key generation's own carve (s1 / s2 as one byte per coefficient from k_expand_s<.., S8>, A s1 never stored);
    // otherwise get_public_key's (int32 rows from k_key_intt, A s1 stored).  Reservations use the larger, default form.
    KeygenWs(void *base, const mldsa_params *p, size_t n, bool byte_rows = false) {
        Carver cv(base);
        a_hat = cv.take<int32_t>(n * (size_t)(p->k * p->l) * N);  // public (ExpandA(rho)): first, outside the zeroised span
        hbuf = cv.take<uint8_t>(n * 128);
        const size_t secret_off = cv.off - n * 128;
        s1s2 = cv.take<int32_t>(n * (size_t)(p->l + p->k) * (byte_rows ? N / 4 : N));
        as1 = cv.take<int32_t>(byte_rows ? 0 : n * (size_t)p->k * N);
        secret_bytes = cv.off - secret_off;
        bytes = cv.off + 256;
    }
};
}  // namespace

size_t keygen_workspace_bytes(const mldsa_ctx *ctx, const mldsa_params *p, size_t n_keys) {
    return KeygenWs(nullptr, p, std::min(n_keys, ctx->pass_ops)).bytes;
}

int keygen_batch(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "keygen: unknown parameter set");
    if (n_keys == 0) return MLDSA_OK;
    const size_t chunk = std::min(n_keys, ctx->pass_ops);
    if (ctx->ws_bytes < KeygenWs(nullptr, p, chunk).bytes) return set_error(MLDSA_ERR_NOMEM, "keygen: workspace not reserved");
    const size_t pkl = (size_t)p->pk_len, skl = (size_t)p->sk_len;
    KeygenWs w(ctx->ws, p, chunk, true);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    // what this call will clear at its end: rho' / K (hbuf), s1, s2 -- A s1 is no longer stored
    const size_t z_lo = (size_t)(w.hbuf - static_cast<uint8_t *>(ctx->ws));
    const size_t z_mid = (size_t)(reinterpret_cast<uint8_t *>(w.s1s2) - static_cast<uint8_t *>(ctx->ws));
    const size_t z_hi = (size_t)(reinterpret_cast<uint8_t *>(w.as1) - static_cast<uint8_t *>(ctx->ws));
    // The previous call's clearing may still run on a helper stream (below).  A key-generation call of the SAME layout may start
    // beside it: its seed hash needs only the (small) seed block, cleared first (zero_head_ev); its ExpandA writes below the
    // cleared span; ExpandS, the first kernel that writes into the big part, waits for the rest (zero_ev).
    bool wait_head = false, wait_rest = false;
    if (ctx->zero_pending) {
        if (!capturing && ctx->zero_head_valid && ctx->zero_lo == z_lo && ctx->zero_mid == z_mid && ctx->zero_hi == z_hi) wait_head = wait_rest = true;
        else TRY(wait_zeroise(ctx, s));
    }
    int rc = MLDSA_OK;
    bool wiped_by_kernel = false;
    for (size_t o = 0; o < n_keys && rc == MLDSA_OK; o += chunk) {
        const size_t n = (n_keys - o) < chunk ? (n_keys - o) : chunk;
        uint8_t *pko = pk + o * pkl, *sko = sk + o * skl;
        rc = [&]() -> int {
            if (wait_head) { MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->zero_head_ev, 0)); wait_head = false; }
            // a SMALL call: one launch (kernels_small.hip k_keygen_small), the same workspace rows and the same bytes out
            if (ctx->opt_coop_hash && ctx->opt_small_fused > 0 && n <= small_ops_limit((long)ctx->small_keygen_max, p)) {
                if (wait_rest) {
                    MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->zero_ev, 0));
                    wait_rest = false;
                    ctx->zero_pending = false;
                }
#ifdef MLDSA_TEST_NO_ZEROISE
                const bool wipe_in_kernel = false;
#else
                const bool wipe_in_kernel = true;
#endif
                STAGE("keygen_small", launch_keygen_small(ctx, p, xi + o * 32, pko, sko, n, w.a_hat, w.hbuf, w.s1s2, ctx->d_small_ctr, s, wipe_in_kernel));
                wiped_by_kernel = wipe_in_kernel && n == n_keys;  // (a small call is one pass: everything secret is already cleared)
                return MLDSA_OK;
            }
            // 1: (rho, rho', K) <- H(xi || k || l, 128)                          ml_dsa.rs:68-74
            STAGE("seed_hash", launch_shake256_2(ctx, 128, xi + o * 32, 32, 32, nullptr, nullptr, 0, 0, (uint32_t)p->k | ((uint32_t)p->l << 8), 2,
                                                 w.hbuf, 128, n, s));
            STAGE("expand_a", launch_expand_a(ctx, set, w.hbuf, 128, nullptr, w.a_hat, n, s, true));          // :85 (24-bit form)
            if (wait_rest) {
                MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->zero_ev, 0));
                wait_rest = false;
                ctx->zero_pending = false;
            }
            STAGE("expand_s", launch_expand_s(ctx, set, w.hbuf + 32, 128, w.s1s2, n, s, true));               // :79 (byte rows)
            // :86-92 t = inv_ntt(A * ntt(s1)) + s2, Power2Round and the whole encoding of the polynomials: the matrix-vector kernel
            // packs s1 and s2 into sk as it reads them (in place, from the rows ExpandS wrote) and t1 -> pk, t0 -> sk in its
            // epilogue; A s1 and t never exist in HBM
            STAGE("keygen_t", launch_keygen_t(ctx, p, w.a_hat, w.s1s2, pko, sko, n, s));
            TRY(launch_keygen_seeds(ctx, p, w.hbuf, pko, sko, n, s));  // rho -> pk, sk; K -> sk
            STAGE("tr_hash", launch_shake256_2(ctx, 64, pko, pkl, p->pk_len, nullptr, nullptr, 0, 0, 0, 0, sko + 64, skl, n, s));  // tr = H(pk), :99-101
            return MLDSA_OK;
        }();
    }
    if (wait_head || wait_rest) (void)wait_zeroise(ctx, s);  // the call failed before it got there
    // rho' / K, s1 and s2 are secret: cleared on every path out, like the reference's zeroize-on-drop (types.rs:19).  Off the
    // caller's critical path: on a helper stream, the seed block first; the next op-level call of the context waits for it on
    // the device (OpGuard / above), destroy and regrow wait for the whole device.
    if (wiped_by_kernel && rc == MLDSA_OK) {
        // k_keygen_small's tail cleared each key's rows itself: no clearing launches (and no events) behind a small call.  What the
        // residue probe looks at is exactly those rows (the alignment padding between them never held anything of this call).
        ctx->secret_spans.push_back({z_lo, n_keys * 128});
        ctx->secret_spans.push_back({z_mid, n_keys * (size_t)(p->k + p->l) * N});
        return rc;
    }
    ctx->secret_spans.push_back({z_lo, z_hi - z_lo});
    if (capturing || rc != MLDSA_OK) {
        MLDSA_WIPE(launch_zero(ctx, w.hbuf, z_hi - z_lo, s));
    } else {
        const ZeroSpan head{w.hbuf, z_mid - z_lo}, rest{w.s1s2, z_hi - z_mid};
        if (clear_in_background(ctx, s, &head, &rest, 1)) {
            ctx->zero_pending = true;
            ctx->zero_head_valid = true;
            ctx->zero_lo = z_lo; ctx->zero_mid = z_mid; ctx->zero_hi = z_hi;
        }
    }
    return rc;
}

// Signer::get_public_key -> private_to_public_key (ml_dsa.rs:502-559)
int get_public_key_batch(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
                         uint8_t *pk_rho, uint8_t *pk_tr, int32_t *pk_t1, size_t n_keys, hipStream_t s) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "get_public_key: unknown parameter set");
    if (n_keys == 0) return MLDSA_OK;
    const size_t chunk = std::min(n_keys, ctx->pass_ops);
    if (ctx->ws_bytes < KeygenWs(nullptr, p, chunk).bytes) return set_error(MLDSA_ERR_NOMEM, "get_public_key: workspace not reserved");
    KeygenWs w(ctx->ws, p, chunk);
    const int k = p->k, l = p->l;
    int rc = MLDSA_OK;
    for (size_t o = 0; o < n_keys && rc == MLDSA_OK; o += chunk) {
        const size_t n = (n_keys - o) < chunk ? (n_keys - o) : chunk;
        rc = [&]() -> int {
            // s_1 / s_2 back to centred coefficients (ml_dsa.rs:512-527; the reference keeps s_1 in the NTT domain and
            // transforms only s_2 -- same values: ntt(inv_ntt(x)) = x)
            TRY(launch_key_intt(ctx, s1 + o * (size_t)l * N, l, n, 0, 0, nullptr, 0, 0, w.s1s2, l + k, 0, s));
            TRY(launch_key_intt(ctx, s2 + o * (size_t)k * N, k, n, 0, 0, nullptr, 0, 0, w.s1s2, l + k, l, s));
            TRY(launch_expand_a(ctx, set, rho + o * 32, 32, nullptr, w.a_hat, n, s, true));                      // :508
            TRY(launch_sign_w(ctx, set, w.a_hat, nullptr, w.s1s2, w.as1, nullptr, 0, n, s, (size_t)(l + k), nullptr, true));  // :544-545
            TRY(launch_t1_hat(ctx, p, w.as1, w.s1s2, pk_t1 + o * (size_t)k * N, n, s));                        // :546-556
            TRY(launch_copy_rows(ctx, pk_rho + o * 32, 32, rho + o * 32, 32, 32, n, s));
            TRY(launch_copy_rows(ctx, pk_tr + o * 64, 64, tr + o * 64, 64, 64, n, s));
            return MLDSA_OK;
        }();
    }
    ctx->secret_spans.push_back({(size_t)(w.hbuf - static_cast<uint8_t *>(ctx->ws)), w.secret_bytes});
    MLDSA_WIPE(launch_zero(ctx, w.hbuf, w.secret_bytes, s));
    return rc;
}

// ------------------------------------------------------------------------------------
// Signer::try_sign_* -> sign_internal (ml_dsa.rs:153-337) with the rejection loop re-batched and driven from
// the device: every round runs one loop iteration for all unfinished ops, then compacts the active list.  The
// round kernels take their counts from RoundCtl in the workspace (kernels_sign.hip), so the host enqueues a
// whole call -- prologue + a planned number of rounds -- without reading anything back in between.
namespace {
// sign ops resident per pass (workspace size).  The rounds' fixed costs (the lane-per-slot hash / SampleInBall at one wave per
// SIMD, launch tails) are paid per pass, so a larger pass signs faster: 7.7 M/s at 65 536 ops, 8.6 M/s at 131 072, 9.1 M/s at
// 262 144 (ML-DSA-65; the workspace of a 262 144-op ML-DSA-87 pass is 15 GB of the 288)
// = the default of mldsa_ctx::pass_ops_sign

struct SignWs {
    // y / w / w1 / wrisk / yrisk / c / ctilde are ROWS (one per generated candidate: `rows` = ns, or 2 n when two candidates per op
    // may be generated at once); accept is per TESTED candidate (slot)
    int32_t *a_hat, *y, *w, *c, *c8, *done, *bad_op, *key_bad, *accept;
    uint8_t *rnd_mu, *rho_pp, *w1, *ctilde, *wrisk, *yrisk, *key_oor;
    uint16_t *kappa, *slot_kappa, *gen_kappa;
    uint32_t *act[2], *ypos[2], *slot_op, *gen_op, *gen_key, *slot_y, *kidx, *exp_list;
    RoundCtl *ctl;
    uint32_t *small_ctr = nullptr;  // not a carve: this lane's set of arrival counters (ctx->d_small_ctr; set by sign_batch)
    RoundCtl *host_ctl = nullptr;  // not a carve: device-visible host memory k_compact of a SMALL call reports its counts in (launch_compact); set by sign_batch
    size_t bytes = 0;
    uint8_t *base = nullptr;
    SignWs() = default;
    // n = ops of the chunk, ns = most candidate slots of a round
    // two_rows: room for two generated candidates per op (MLDSA_OPT_SIGN_LOOKAHEAD)
    SignWs(void *base_, const mldsa_params *p, size_t n, size_t ns, bool own_a_hat, bool two_rows) : base(static_cast<uint8_t *>(base_)) {
        Carver cv(base_);
        const size_t rows = two_rows ? std::max(ns, 2 * n) : ns;
        a_hat = cv.take<int32_t>(own_a_hat ? n * (size_t)(p->k * p->l) * N : 0);
        key_bad = cv.take<int32_t>(n);
        kidx = cv.take<uint32_t>(n);
        // ExpandMask's squeezed bytes, 32 c per polynomial (k_expand_mask<.., RAW>); first secret-dependent carve: everything from here on is zeroised
        y = cv.take<int32_t>(rows * (size_t)p->l * (size_t)(8 * (p->gamma1 == (1 << 17) ? 18 : 20)));
        w = cv.take<int32_t>(rows * (size_t)p->k * PACKED_POLY_DWORDS);  // 24-bit fields (sign_w)
        c = cv.take<int32_t>(rows * (size_t)N);       // c_hat
        c8 = cv.take<int32_t>(rows * (size_t)(N / 4));  // c as SampleInBall leaves it: one byte per coefficient
        done = cv.take<int32_t>(n);
        bad_op = cv.take<int32_t>(n);
        accept = cv.take<int32_t>(ns);
        rnd_mu = cv.take<uint8_t>(n * 96);  // rnd || mu per op: H(K || rnd || mu) input, ml_dsa.rs:199
        rho_pp = cv.take<uint8_t>(n * 64);
        w1 = cv.take<uint8_t>(rows * (size_t)p->w1_len);
        ctilde = cv.take<uint8_t>(rows * 64);
        wrisk = cv.take<uint8_t>(rows);
        yrisk = cv.take<uint8_t>(rows * (size_t)p->l);
        key_oor = cv.take<uint8_t>(n);  // per key of the table, or per op when the table is larger than the chunk
        kappa = cv.take<uint16_t>(n);
        slot_kappa = cv.take<uint16_t>(ns);
        act[0] = cv.take<uint32_t>(n);
        act[1] = cv.take<uint32_t>(n);
        ypos[0] = cv.take<uint32_t>(n);
        ypos[1] = cv.take<uint32_t>(n);
        exp_list = cv.take<uint32_t>(n);
        slot_op = cv.take<uint32_t>(ns);
        gen_op = cv.take<uint32_t>(rows);
        gen_key = cv.take<uint32_t>(rows);
        gen_kappa = cv.take<uint16_t>(rows);
        slot_y = cv.take<uint32_t>(ns);
        ctl = cv.take<RoundCtl>(1);
        bytes = (cv.off + 511) & ~(size_t)255;
    }
};

// per-iteration acceptance probability of sign_internal's loop (FIPS 204 table 1: 4.25 / 5.1 / 3.85 expected iterations)
double accept_prob(int set) { return set == MLDSA_44 ? 1.0 / 4.25 : set == MLDSA_65 ? 1.0 / 5.1 : 1.0 / 3.85; }

// The a-priori round plan of one chunk: the device applies the speculation rule to the ACTUAL number of
// unfinished ops; the host replays the same rule on the EXPECTED number to size the grids and to know how
// many rounds to enqueue.  Rounds continue until the expected number of unfinished ops is below a threshold (by
// Markov's inequality the probability that an op is left is below it too).
struct SignPlan {
    uint32_t spec_target = 1, spec_rows = 1, spec_max = 1;
    SpecRule rule;  // the same table goes to k_make_slots
    size_t ns_max = 0;
    std::vector<size_t> m_hint, ns_hint;  // per round: ops / slots the grids are sized for
    std::vector<int> one_cand;            // per round: the plan expects one candidate per op (the device decides for itself)
    std::vector<uint32_t> spec_hi;        // per round: most candidates per op the rule gives over the range the count can fall in
};

static SignPlan plan_sign_compute(const mldsa_ctx *ctx, int set, size_t n, bool async_mode, double plan_stop) {
    SignPlan pl;
    pl.spec_max = (uint32_t)ctx->opt_spec_max;
    // a small batch cannot fill the target however many candidates each op gets: cap it so that the
    // workspace and the grids follow the batch
    // (effective target: the user's option under the cap reserve_workspace sets when it has to halve the rows of a round)
    const size_t tgt = std::min<size_t>((size_t)std::min(ctx->opt_spec_target, ctx->spec_target_cap), std::max<size_t>(n * pl.spec_max, 1));
    pl.spec_target = (uint32_t)tgt;
    // candidates per speculative round (>= the threshold above): rows = what such a round generates
    size_t rows = std::max(tgt, std::min<size_t>((size_t)ctx->opt_spec_rows, std::max<size_t>(n * pl.spec_max, 1)));
    // candidates per op in a speculative round = round((rows / m) ^ alpha): 1 fills every round to `rows`; 0.85 measured 1.5-2.2 % faster for
    // all three sets (mid-size rounds get fewer: fewer wasted candidates, ExpandMask launches that fit whole layers; EXPERIMENTS.md round 3)
    double alpha = 0.85;
    // A SMALL call (the ones whose prologue is one launch): a round of up to coop_mask_max / l candidate rows runs its first half as ONE
    // launch on the cooperative sponges (k_sign_front_small, ~45 us whatever the rows), a larger one on the five lane-per-state kernels
    // (~115 us for the 2 048 rows that 64 ops x 32 candidates make).  So such a call speculates only as far as the single launch reaches:
    // floor(rows_small / m) candidates per op -- 25 for 32 ML-DSA-65 ops (0.4 % of them need a second round) instead of 32 -- as long as
    // that leaves every op of round 0 a dozen candidates (ctx->small_sign_spec: with fewer, the extra rounds cost more than the five
    // kernels; EXPERIMENTS.md).  The signatures do not depend on it (the FIRST accepted candidate, ml_dsa.rs:212-330).
    const mldsa_params *pp = params_of(set);
    const size_t rows_small = (ctx->opt_coop_hash && ctx->opt_small_fused > 0 && ctx->small_sign_front && ctx->small_sign_spec && pp) ? ctx->coop_mask_max / (size_t)pp->l : 0;
    const bool small_rule = rows_small >= n * (size_t)std::max(1L, ctx->small_sign_spec) && n <= ctx->small_sign_max && n <= 256 && n <= small_ops_limit(ctx->opt_small_fused, pp) && rows > rows_small;
    size_t tgt_eff = tgt;
    if (small_rule) {
        rows = rows_small;
        tgt_eff = std::min(tgt, rows_small);
        alpha = 1.0;  // (floor: m * spec(m) never exceeds rows_small)
    }
    pl.spec_target = (uint32_t)tgt_eff;
    pl.spec_rows = (uint32_t)rows;
    pl.ns_max = std::max(n, rows);
    // candidates per op for m unfinished ops: 1 while m * 2 > tgt, then round((rows / m) ^ alpha), at most spec_max.  alpha = 1
    // fills every speculative round to `rows` candidates; alpha < 1 gives mid-size rounds fewer (more, smaller rounds: fewer
    // wasted candidates and ExpandMask launches that fit whole layers, against one more round's fixed cost)
    {
        auto spec_of = [&](double m) -> uint32_t {
            if (m < 1 || m * 2 > (double)tgt_eff) return 1;
            const double s = std::floor(std::pow((double)rows / m, alpha) + (alpha < 1.0 ? 0.5 : 0.0));
            return (uint32_t)std::max(1.0, std::min((double)pl.spec_max, s));
        };
        for (uint32_t s = 0; s < 64; s++) {
            // thr[s] = largest m that still gets more than s candidates (0 if none): spec_of is non-increasing in m
            uint32_t lo = 0, hi = (uint32_t)std::min<size_t>(n, 0xFFFFFFFFu);
            if (s == 0) { pl.rule.thr[0] = 0xFFFFFFFFu; continue; }
            if (spec_of(1) <= s) { pl.rule.thr[s] = 0; continue; }
            lo = 1;  // spec_of(lo) > s
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo + 1) / 2;
                if (spec_of((double)mid) > s) lo = mid; else hi = mid - 1;
            }
            pl.rule.thr[s] = lo;
        }
    }
    // the most slots a round can have = max over m <= n of m * spec(m): with the rounded rule a count just below a threshold has
    // MORE slots than `rows` (22 300 ops x 3 = 66 900 for rows = 65 536); spec is non-increasing in m, so the maximum sits at n or
    // at one of the thresholds.  Every slot / row array of the workspace is sized from this (k_make_slots clamps to it as well).
    {
        auto slots_at = [&](uint32_t mm) { return (size_t)mm * pl.rule.spec(mm, pl.spec_max); };
        size_t top = slots_at((uint32_t)std::min<size_t>(n, 0xFFFFFFFFu));
        for (uint32_t sidx = 1; sidx < 64; sidx++) {
            const uint32_t t = pl.rule.thr[sidx];
            if (t >= 1 && t <= n) top = std::max(top, slots_at(t));
        }
        pl.ns_max = std::max(pl.ns_max, top);
    }
    const double q = 1.0 - accept_prob(set);
    double m = (double)n;
    // a synchronous call looks at the device once anyway and adds rounds if an op is left, so its plan stops when that is
    // unlikely (< 5 % of the calls: a planned round for 0.002 expected ops costs every call the ~0.2 ms latency chain of an
    // empty round, two extra rounds cost the rare call ~0.5 ms); an asynchronous call cannot look, and plans until < 1e-9
    const double stop = plan_stop > 0.0 ? plan_stop : async_mode ? ctx->async_stop : 0.05;
    for (int r = 0; r < 64 && m > stop; r++) {
        // grids follow mean + 6 sigma of the binomial count (a round that still finds more just loops)
        const double m_hi = std::min((double)n, m + 6.0 * std::sqrt(m) + 1.0);
        const size_t mh = (size_t)std::ceil(m_hi);
        const size_t spec = pl.rule.spec((uint32_t)mh, pl.spec_max);
        const size_t mm = (size_t)std::max(1.0, std::floor(m));
        const double spec_mean = (double)pl.rule.spec((uint32_t)mm, pl.spec_max);  // the rule applied to the mean (what the device will mostly see)
        pl.m_hint.push_back(mh);
        pl.one_cand.push_back(spec_mean == 1.0 && spec == 1 ? 1 : 0);
        // slots the round may have: m * spec(m) over the whole range the count can fall in (spec steps UP as m falls below a
        // threshold, so the largest product sits at m_hi or just below one of the thresholds inside the range)
        const double m_lo = std::max(1.0, m - 6.0 * std::sqrt(m) - 1.0);
        size_t ns_top = std::max(mh * spec, (size_t)std::ceil(m_hi * spec_mean));
        for (uint32_t sidx = 1; sidx < 64 && sidx < pl.spec_max; sidx++) {
            const double t = (double)pl.rule.thr[sidx];
            if (t >= m_lo && t <= m_hi) ns_top = std::max(ns_top, (size_t)t * pl.rule.spec(pl.rule.thr[sidx], pl.spec_max));
        }
        pl.ns_hint.push_back(std::min(pl.ns_max, ns_top));
        pl.spec_hi.push_back(pl.rule.spec((uint32_t)std::max(1.0, std::floor(m_lo)), pl.spec_max));  // (spec is non-increasing in m)
        // progress is planned with the FEWER candidates of the two (the count sitting just above a threshold of the rule must not
        // leave the call short of rounds: a synchronous call would pay two extra rounds and a host round trip, an asynchronous
        // one would report MLDSA_ERR_AGAIN)
        m *= std::pow(q, (double)std::min<size_t>(spec, (size_t)spec_mean));
    }
    if (ctx->opt_sign_rounds > 0 && (size_t)ctx->opt_sign_rounds < pl.m_hint.size()) {
        pl.m_hint.resize((size_t)ctx->opt_sign_rounds);
        pl.ns_hint.resize((size_t)ctx->opt_sign_rounds);
        pl.one_cand.resize((size_t)ctx->opt_sign_rounds);
        pl.spec_hi.resize((size_t)ctx->opt_sign_rounds);
    }
    return pl;
}

// The plan of a call shape is asked for twice per call (the workspace size, then the call itself) and costs up to ~400 pow() -- 10 ... 15 us
// for a 64 ... 256-op call, before its first launch.  A service repeats its shapes: the last few plans are kept per thread, keyed by
// everything plan_sign_compute reads.
SignPlan plan_sign(const mldsa_ctx *ctx, int set, size_t n, bool async_mode, double plan_stop = 0.0) {
    struct Key {
        long v[16];
        double d[2];
    };
    struct Entry {
        Key key;
        SignPlan pl;
        bool valid = false;
    };
    Key k;
    memset(&k, 0, sizeof(k));
    k.v[0] = set; k.v[1] = (long)n; k.v[2] = async_mode ? 1 : 0; k.v[3] = ctx->opt_spec_max; k.v[4] = ctx->opt_spec_target; k.v[5] = ctx->spec_target_cap;
    k.v[6] = ctx->opt_spec_rows; k.v[8] = ctx->opt_coop_hash; k.v[9] = ctx->opt_small_fused; k.v[10] = ctx->small_sign_front;
    k.v[11] = ctx->small_sign_spec; k.v[12] = (long)ctx->coop_mask_max; k.v[13] = (long)ctx->small_sign_max; k.v[14] = ctx->opt_sign_rounds;
    k.d[0] = plan_stop; k.d[1] = ctx->async_stop;
    constexpr int SLOTS = 4;
    thread_local Entry cache[SLOTS];
    thread_local int next = 0;
    for (int i = 0; i < SLOTS; i++)
        if (cache[i].valid && memcmp(&cache[i].key, &k, sizeof(k)) == 0) return cache[i].pl;
    Entry &e = cache[next];
    next = (next + 1) % SLOTS;
    e.key = k;
    e.pl = plan_sign_compute(ctx, set, n, async_mode, plan_stop);
    e.valid = true;
    return e.pl;
}
}  // namespace

// Lanes of a chunk of `chunk` ops: two slices side by side (MLDSA_OPT_SIGN_LANES).  Measured per parameter set and size on three boxes
// (profiles/r06_ab_sign_lanes_grid*.txt): with slices of >= 65 536 ops -- full-size launches in both chains -- two lanes win every time
// (131 072 ... 262 144 ops: ML-DSA-65 +3.3 ... 4.9 %, 9.6 -> 9.9 ... 10.5 M/s; 44 +7 ... 12 %; 87 +2.5 ... 5.5 %); at 65 536 ops ML-DSA-44
// still gains (+1.6 / +2.9 % on two boxes), 65 and 87 are within +-1.5 % from box to box; below that the half-size slices speculate more
// and lose 4 ... 16 %.  0 (default) = two slices from that size on, 1 = never, 2 = for every call of >= 8 192 ops.
static size_t sign_lanes_auto_min_ops(int set) { return set == MLDSA_44 ? 65536 : 131072; }
static int sign_lanes_for(const mldsa_ctx *ctx, int set, size_t chunk) {
    if (ctx->opt_sign_lanes >= 2) return chunk >= 8192 ? 2 : 1;
    return ctx->opt_sign_lanes == 0 && chunk >= sign_lanes_auto_min_ops(set) ? 2 : 1;
}
static size_t lane_ops(size_t chunk, int n_lanes) { return n_lanes == 1 ? chunk : ((chunk + 1) / 2 + 255) & ~(size_t)255; }

static bool lookahead_on(const mldsa_ctx *ctx, const mldsa_params *p);

size_t sign_workspace_bytes(const mldsa_ctx *ctx, const mldsa_params *p, size_t n_ops, bool own_a) {
    const size_t chunk = std::min(n_ops, ctx->pass_ops_sign);
    auto layout = [&](int n_lanes) {
        const size_t n = lane_ops(chunk, n_lanes);
        return n_lanes * SignWs(nullptr, p, n, plan_sign(ctx, p->set, n, true).ns_max, own_a, lookahead_on(ctx, p)).bytes;
    };
    // (a call that exports its signatures round by round -- mldsa_sign_host's direct path -- runs as ONE lane whatever the policy says,
    //  sign_batch: room for either layout)
    return sign_lanes_for(ctx, p->set, chunk) == 1 ? layout(1) : std::max(layout(1), layout(2));
}

// batches below this size generate one candidate per op and round: their sign_w is not bound by re-reading A_hat
constexpr size_t LOOKAHEAD_MIN_OPS = 8192;
// Two candidates per op generated at once (k_make_slots).  Measured at 65 536 ops, same-box A/B: ML-DSA-65 7.81 / 7.93 ms against
// 8.00 / 8.18 ms per signing step (sign_w 2.11-2.21 instead of 2.44-2.61 ms; 6.8 instead of 6.5 generated candidates per signature);
// ML-DSA-44 and ML-DSA-87 within 1 % either way (their ExpandMask launches quantise worse at twice the rows), so the default
// (option value 1) applies it to ML-DSA-65 only
static bool lookahead_on(const mldsa_ctx *ctx, const mldsa_params *p) {
    return ctx->opt_lookahead == 2 || (ctx->opt_lookahead == 1 && p->set == MLDSA_65);
}

// One round of the rejection loop (steps 10-33 of Algorithm 7), counts read from the device
// mldsa_sign_host's direct export: round `round`'s finished signatures (k_export_done) on exp_stream, ordered after everything
// enqueued on `s` so far
static int enqueue_export(mldsa_ctx *ctx, const mldsa_params *p, const SignWs &w, int round, size_t ops_hint, const uint8_t *sg, uint8_t *export_sg,
                          hipStream_t s, hipStream_t exp_stream) {
    MLDSA_HIP_CHECK(hipEventRecord(ctx->exp_fork_ev, s));
    MLDSA_HIP_CHECK(hipStreamWaitEvent(exp_stream, ctx->exp_fork_ev, 0));
    {
        ProfScope ps(ctx, exp_stream, "export_to_host");
        TRY(launch_export_done(ctx, w.ctl, round, w.exp_list, sg, export_sg, (size_t)p->sig_len, ops_hint, s, exp_stream));
    }
    MLDSA_HIP_CHECK(hipEventRecord(ctx->exp_join_ev, exp_stream));
    return MLDSA_OK;
}

// pre_in: the plan lets this round test the rows the previous one generated; gen2: the plan lets this round generate two
// candidates per op (k_make_slots decides on the device)
static int enqueue_sign_round(mldsa_ctx *ctx, const mldsa_params *p, const SignWs &w, const SignPlan &pl, int round, size_t m_hint,
                              size_t ns_hint, const uint32_t *kidx, const int32_t *s1, const int32_t *s2, const int32_t *t0,
                              const int32_t *a_hat_keys, uint8_t *sg, hipStream_t s, bool oor_by_op, bool pre_in = false,
                              bool gen2 = false, uint8_t *export_sg = nullptr, hipStream_t exp_stream = nullptr,
                              bool exp_pending = false, bool slots_ready = false, bool slots_next = false) {
    // slots_ready / slots_next (small calls, w.host_ctl set): this round was opened by the previous round's k_compact_small already / this
    // round's k_compact_small opens the next one
    const int set = p->set, par = round & 1;
    const bool own_a = a_hat_keys == nullptr;
    const bool small_compact = w.host_ctl != nullptr;  // (sign_batch sets host_ctl only for calls that do not export)
    auto compact_small_args = [&]() {
        CompactSmallArgs C{};
        C.ctl = w.ctl; C.parity = par; C.act_in = w.act[par]; C.done = w.done; C.act_out = w.act[par ^ 1]; C.ypos_out = w.ypos[par ^ 1];
        C.host_ctl = w.host_ctl; C.next_on = slots_next ? 1 : 0;
        C.rule = pl.rule; C.spec_max = pl.spec_max; C.ns_cap = (uint32_t)std::min<size_t>(pl.ns_max, 0xFFFFFFFFu); C.kappa = w.kappa; C.l = p->l;
        C.slot_op = w.slot_op; C.slot_kappa = w.slot_kappa; C.key_idx = kidx; C.gen_op = w.gen_op; C.gen_kappa = w.gen_kappa;
        C.gen_key = own_a ? nullptr : w.gen_key; C.slot_y = w.slot_y;
        return C;
    };
    auto compact = [&]() -> int {
        if (small_compact) {
            STAGE("compact", launch_compact_small(ctx, compact_small_args(), s));
        } else {
            STAGE("compact", launch_compact(ctx, w.ctl, par, w.act[par], w.done, w.act[par ^ 1], m_hint, s, w.ypos[par ^ 1], export_sg ? w.exp_list : nullptr));
        }
        return MLDSA_OK;
    };
    // The second half of the round: the tests of every candidate, the winner's signature per op, compaction + the next round's slots.  A
    // SMALL call (<= 256 ops, its counts reported through host_ctl) runs it as ONE launch with the hand-overs inside (kernels_sign.hip
    // k_sign_back_small); otherwise three kernels.
    const bool back_small = small_compact && ctx->small_sign_back && ctx->opt_small_fused > 0 && w.small_ctr && !pre_in && !gen2 && !pl.m_hint.empty() &&
                            pl.m_hint[0] <= 256 && pl.spec_max <= 64 && ns_hint <= ctx->small_back_slots_max;
    auto second_half = [&](const int32_t *y_rows, const uint8_t *yrisk_rows) -> int {
        if (back_small) {
            SignBackSmall B{};
            B.c_hat = w.c; B.y = y_rows; B.w = w.w; B.ctilde = w.ctilde; B.s1 = s1; B.s2 = s2; B.t0 = t0; B.sigs = sg; B.oor_by_op = oor_by_op ? 1 : 0; B.ctl = w.ctl;
            B.act = w.act[par]; B.slot_y = w.slot_y; B.key_idx = kidx; B.wrisk = w.wrisk; B.yrisk = yrisk_rows; B.key_oor = w.key_oor; B.kappa = w.kappa; B.done = w.done;
            B.accept = w.accept; B.ctr = w.small_ctr; B.ops_cap = (uint32_t)pl.m_hint[0]; B.spec_cap = pl.spec_max;
            B.ops_hint = (uint32_t)std::min<size_t>(m_hint, 256); B.spec_hint = (size_t)round < pl.spec_hi.size() ? pl.spec_hi[(size_t)round] : pl.spec_max;
            STAGE("sign_back_small", launch_sign_back_small(ctx, p, B, compact_small_args(), s));
            return MLDSA_OK;
        }
        // 18-33: <<c s1>>, <<c s2>>, z, r0, checks, <<c t0>>, h, checks, sigEncode   :240-336
        // (in a speculative round: only the tests that can reject, one verdict per candidate)
        STAGE("sign_tail", launch_sign_tail(ctx, p, w.c, y_rows, w.w, w.ctilde, w.slot_op, kidx, s1, s2, t0, w.kappa, w.done, sg, w.ctl, w.accept,
                                            ns_hint, s, w.wrisk, yrisk_rows, w.key_oor, oor_by_op ? 1 : 0, w.slot_y));
        // speculative rounds only (the kernel leaves at once when the device chose one candidate per op): the whole iteration
        // for each op's first surviving candidate, bytes straight into the op's signature
        STAGE("resolve", launch_resolve(ctx, p, w.ctl, w.act[par], w.accept, w.c, y_rows, w.w, w.ctilde, kidx, s1, s2, t0, sg, w.done, w.kappa,
                                        m_hint, s, w.key_oor, oor_by_op ? 1 : 0, w.slot_y));
        TRY(compact());
        return MLDSA_OK;
    };
    const uint32_t *ns_gen_dev = &w.ctl->ns_gen;  // rows generated this round (the tail kernels read ctl->ns themselves)
    const size_t gen_hint = gen2 ? 2 * ns_hint : ns_hint;  // grids of the generating kernels (a round that tests ready rows finds ns_gen = 0)
    if (!(slots_ready && small_compact))
        STAGE("make_slots", launch_make_slots(ctx, w.ctl, par, pl.rule, pl.spec_max, (uint32_t)std::min<size_t>(pl.ns_max, 0xFFFFFFFFu), w.act[par], w.kappa, p->l, w.slot_op,
                                              w.slot_kappa, kidx, w.gen_op, w.gen_kappa, own_a ? nullptr : w.gen_key, gen_hint, s,
                                              pre_in ? 1 : 0, gen2 ? 1 : 0, w.ypos[par], w.slot_y));
    // A SMALL round (rows * L polynomials within the cooperative ExpandMask's range, one candidate per slot generated here): ExpandMask,
    // sign_w, the c~ hash, SampleInBall and NTT(c) as ONE launch (kernels_small.hip k_sign_front_small); the same rows come out.
    // (The launch covers every row the workspace can hold -- rows past the round's count leave at once -- so a round that turns out larger
    //  than its plan is still complete; the PLAN decides whether the round is small enough for the cooperative form to pay.)
    const size_t rows_cap = std::min<size_t>(pl.ns_max, 0xFFFFFFFFu);
    if (ctx->opt_coop_hash && ctx->opt_small_fused > 0 && ctx->small_sign_front && !gen2 && !pre_in && !export_sg && w.small_ctr && rows_cap <= SMALL_CTR_ENTRIES &&
        gen_hint * (size_t)p->l <= ctx->coop_mask_max) {
        SmallSignFrontArgs A{};
        A.ns_gen = ns_gen_dev; A.rho_pp = w.rho_pp; A.gen_kappa = w.gen_kappa; A.gen_op = w.gen_op; A.a_idx = own_a ? w.gen_op : w.gen_key;
        A.a_hat = own_a ? w.a_hat : a_hat_keys; A.mu = w.rnd_mu + 32; A.y = w.y; A.w = w.w; A.w1 = w.w1; A.ctilde = w.ctilde; A.c8 = w.c8; A.c_hat = w.c;
        A.wrisk = w.wrisk; A.yrisk = w.yrisk; A.ctr = w.small_ctr; A.fwd_tab = ctx->d_fwd_tw; A.inv_tab = ctx->d_inv_tw;
        A.w_risk_bound = p->gamma2 - 2 * p->beta; A.y_risk_bound = p->gamma1 - 2 * p->beta; A.tau = p->tau; A.rows_cap = (uint32_t)rows_cap;
        STAGE("sign_front_small", launch_sign_front_small(ctx, p, A, own_a, s));
        TRY(second_half(w.y, w.yrisk));
        return MLDSA_OK;
    }
    // 11: y <- ExpandMask(rho'', kappa)                               :215
    STAGE("expand_mask", launch_expand_mask(ctx, set, w.rho_pp, 64, w.gen_kappa, 1, w.gen_op, w.y, gen_hint, s, nullptr, ns_gen_dev, true));
    // the PREVIOUS round's finished signatures -> the caller's host memory, on a helper stream (a small, fixed number of
    // workgroups: see launch_export_done).  The launch reads only its own range of the completion-order list, which no later
    // round touches, so nothing of the round chain ever waits for it.
    if (export_sg && exp_pending) TRY(enqueue_export(ctx, p, w, round - 1, m_hint, sg, export_sg, s, exp_stream));
    // 12: w <- invNTT(A_hat o NTT(y))                                 :218-222   (one row per generated candidate; the two
    // candidates of an op are adjacent rows and share the A_hat read)
    STAGE("sign_w", launch_sign_w(ctx, set, own_a ? w.a_hat : a_hat_keys, own_a ? w.gen_op : w.gen_key, w.y, w.w, w.w1,
                                  (size_t)p->w1_len, gen_hint, s, 0, w.wrisk, own_a, ns_gen_dev, nullptr, true, w.yrisk));
    // 13-15: w1 <- HighBits(w), w1Encode: in the epilogue of sign_w; c_tilde <- H(mu || w1)   :225-234
    // The challenge of EVERY generated row, also of the second candidates a two-candidate round makes for the next round: these
    // lane-per-row kernels cost the same for 65 536 and 131 072 rows' worth of latency chains (one or two waves per SIMD), and
    // the next round then consists of the tail alone.
    STAGE("ctilde_hash", launch_shake256_2(ctx, p->ctilde_len, w.rnd_mu + 32, 96, 64, w.gen_op, w.w1, (size_t)p->w1_len, p->w1_len, 0, 0,
                                           w.ctilde, 64, gen_hint, s, ns_gen_dev));
    int32_t *y = w.y;
    uint8_t *yrisk = w.yrisk;
    // 16: c <- SampleInBall(c_tilde)                                  :237
    STAGE("sample_in_ball", launch_sample_in_ball(ctx, set, w.ctilde, 64, w.c8, gen_hint, s, ns_gen_dev, true));
    // 17: c_hat <- NTT(c)                                             :240
    STAGE("ntt_c", launch_ntt_c8(ctx, w.c8, w.c, gen_hint, s, ns_gen_dev));
    TRY(second_half(y, yrisk));
    if (export_sg) TRY(launch_export_snap(ctx, w.ctl, round, s));  // how far the completion-order list has grown: this round's range
    return MLDSA_OK;
}

namespace {
struct SignArgs {
    int set, mode;
    const uint8_t *rho, *cap_k, *tr;
    const int32_t *s1, *s2, *t0;
    size_t n_keys;
    const uint32_t *key_idx;
    const uint8_t *msgs;
    const uint64_t *msg_off;
    const uint8_t *ctxs;
    const uint64_t *ctx_off;
    const uint8_t *rnd;
    uint8_t *sigs;
    int32_t *status;
    const int32_t *a_hat_keys;
    hipEvent_t inputs_ev;  // optional (mldsa_sign_host): msgs / ctxs / rnd are on the device once this event has fired; waited for after ExpandA
    uint8_t *export_sigs;  // optional: the caller's page-locked host array (device-visible), finished signatures are copied there round by round
    size_t offset, n;   // this chunk: first op and number of ops
    size_t n_total;     // ops of the whole call (the offset tables have n_total + 1 entries)
    size_t chunk;       // ops the workspace / plan is laid out for
    int async_mode;
};

struct ChunkKeys {  // per-chunk views of the key tables (identity mapping walks with the chunk)
    const uint32_t *kidx;
    const int32_t *s1k, *s2k, *t0k, *ak;
    uint8_t *sg, *xsg;
};

ChunkKeys chunk_keys(const mldsa_params *p, const SignWs &w, const SignArgs &a) {
    const size_t key_base = a.key_idx ? 0 : a.offset;
    ChunkKeys c;
    c.kidx = a.key_idx ? w.kidx : nullptr;
    c.s1k = a.s1 + key_base * (size_t)p->l * N;
    c.s2k = a.s2 + key_base * (size_t)p->k * N;
    c.t0k = a.t0 + key_base * (size_t)p->k * N;
    c.ak = a.a_hat_keys ? a.a_hat_keys + key_base * (size_t)(p->k * p->l) * N : nullptr;
    c.sg = a.sigs + a.offset * (size_t)p->sig_len;
    c.xsg = a.export_sigs ? a.export_sigs + a.offset * (size_t)p->sig_len : nullptr;
    return c;
}

// the key-range flags of a slice are per op when the key table is larger than the slice (see sign_prologue)
bool oor_by_op(const SignArgs &a) { return a.key_idx && a.n_keys > a.n; }

void zeroise_sign_ws(mldsa_ctx *ctx, const SignWs &w, hipStream_t s) {
    // y, rho'', cs1 / cs2, staged signatures are secret-dependent (the reference zeroizes on drop, types.rs:19);
    // A_hat = ExpandA(rho) is public and is the first and largest carve: skipped.
    uint8_t *secrets = reinterpret_cast<uint8_t *>(w.y);
    MLDSA_WIPE(launch_zero(ctx, secrets, (size_t)(w.base + w.bytes - secrets), s));
}

// steps 1-8 of Algorithm 7 for one lane's slice of a chunk: enqueue only (capturable)
// (the lane-per-op kernels of the prologue -- mu, rho'', the key-range check, the first active list -- follow ExpandA on the same stream:
//  forking them onto the helper stream underneath ExpandA, like the verifier's, measured 8.25 against 8.19 ms per 65 536 ML-DSA-65
//  signatures and was removed in round 6 with its knob)
// slots0_plan / slots0_done (optional): when the prologue is the ONE launch of a small call, its last workgroup opens round 0 as well
// (k_make_slots' work, from this plan's rule) and *slots0_done says so
int sign_prologue(mldsa_ctx *ctx, const mldsa_params *p, const SignWs &w, const SignArgs &a, hipStream_t s,
                  const SignPlan *slots0_plan = nullptr, bool *slots0_done = nullptr) {
    if (slots0_done) *slots0_done = false;
    const bool own_a = a.a_hat_keys == nullptr;
    const size_t o = a.offset, n = a.n;
    int32_t *st = a.status ? a.status + o : nullptr;
    const int32_t *key_bad = nullptr;
    const bool small = ctx->opt_coop_hash && n <= small_ops_limit(ctx->opt_small_fused, p) && n <= ctx->small_sign_max && n <= 256;
    if (a.key_idx && !small) {
        TRY(launch_sanitize_keys(ctx, a.key_idx + o, a.n_keys, n, w.kidx, w.key_bad, s));
        key_bad = w.key_bad;
    }
    const ChunkKeys c = chunk_keys(p, w, a);
    const size_t key_base = a.key_idx ? 0 : o;
    // A SMALL call: the whole prologue -- key check, ExpandA, mu, rho'', key-range check, first active list, control block -- is ONE
    // launch (kernels_small.hip k_sign_prologue_small) instead of the nine below; same rows, same values.
    if (small) {
        if (ctx->zero_wait_after_ea) {  // (no ExpandA beside the previous call's clearing here: it is part of the one launch)
            MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->zero_ev, 0));
            ctx->zero_wait_after_ea = false;
            ctx->zero_pending = false;
        }
        if (a.inputs_ev) MLDSA_HIP_CHECK(hipStreamWaitEvent(s, a.inputs_ev, 0));
        const bool by_op = oor_by_op(a);
        SmallSignPrologueArgs A{};
        A.key_idx = a.key_idx ? a.key_idx + o : nullptr;
        A.n_keys = (uint32_t)std::min<size_t>(a.n_keys, 0xFFFFFFFFu);
        A.rho = a.rho + key_base * 32; A.cap_k = a.cap_k + key_base * 32; A.tr = a.tr + key_base * 64; A.s2 = c.s2k;
        A.mode = a.mode; A.msgs = a.msgs; A.msg_off = a.msg_off; A.ctxs = a.ctxs; A.ctx_off = a.ctx_off; A.op0 = o; A.n_call = a.n_total;
        A.rnd = a.rnd + o * 32; A.n = (uint32_t)n;
        A.a_ws = w.a_hat; A.kidx_out = w.kidx; A.rnd_mu = w.rnd_mu; A.rho_pp = w.rho_pp; A.bad_op = w.bad_op; A.done = w.done; A.status = st;
        A.kappa = w.kappa; A.act0 = w.act[0]; A.ctl = w.ctl; A.sigs = c.sg; A.sig_len = (size_t)p->sig_len; A.key_oor = w.key_oor;
        A.units = (uint32_t)(a.key_idx ? (by_op ? n : a.n_keys) : n); A.units_by_op = by_op ? 1 : 0; A.eta = p->eta;
        A.ctr = w.small_ctr; A.inv_tab = ctx->d_inv_tw;
        if (slots0_plan && slots0_done) {
            const SignPlan &pl = *slots0_plan;
            A.slots0 = 1; A.spec_max = pl.spec_max; A.ns_cap = (uint32_t)std::min<size_t>(pl.ns_max, 0xFFFFFFFFu); A.l = p->l; A.rule = pl.rule;
            A.slot_op = w.slot_op; A.slot_kappa = w.slot_kappa; A.gen_op = w.gen_op; A.gen_kappa = w.gen_kappa; A.gen_key = own_a ? nullptr : w.gen_key;
            A.slot_y = w.slot_y;
            *slots0_done = true;
        }
        STAGE("sign_prologue_small", launch_sign_prologue_small(ctx, p, A, !own_a, s));
        return MLDSA_OK;
    }
    // 5: A_hat <- ExpandA(rho), once per signature                        ml_dsa.rs:181
    if (own_a) STAGE("expand_a", launch_expand_a(ctx, a.set, a.rho + key_base * 32, 32, c.kidx, w.a_hat, n, s, true));
    // the previous signing call may still be clearing its secrets on a helper stream (sign_batch): ExpandA (public, below the
    // cleared span) was allowed to start beside it, everything from here on writes into that span
    if (ctx->zero_wait_after_ea) {
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->zero_ev, 0));
        ctx->zero_wait_after_ea = false;
        ctx->zero_pending = false;
    }
    if (a.inputs_ev) MLDSA_HIP_CHECK(hipStreamWaitEvent(s, a.inputs_ev, 0));  // the host path's uploads of everything but the keys ran beside ExpandA
    // the signature buffer is not cleared: every op's bytes come from its accepted attempt, a refused op (k_init_active) or
    // one an asynchronous call leaves unfinished (k_mark_unfinished) gets its zeros there
    TRY(launch_zero(ctx, w.ctl, sizeof(RoundCtl), s));
    // 6: mu <- H(tr || M', 64)                                            ml_dsa.rs:185-196
    STAGE("mu", launch_mu(ctx, a.tr + key_base * 64, 64, c.kidx, a.mode, a.msgs, a.msg_off, a.ctxs, a.ctx_off, w.rnd_mu + 32, 96, w.bad_op, n, s,
                          key_bad, o, a.n_total));
    TRY(launch_copy_rows(ctx, w.rnd_mu, 96, a.rnd + o * 32, 32, 32, n, s));
    // 7: rho'' <- H(K || rnd || mu, 64)                                   ml_dsa.rs:199-201
    STAGE("rho_pp_hash", launch_shake256_2(ctx, 64, a.cap_k + key_base * 32, 32, 32, c.kidx, w.rnd_mu, 96, 96, 0, 0, w.rho_pp, 64, n, s));
    // is every key's s2 within [-eta, eta]?  (k_sign_tail's one-transform hint stage needs ||c s2||inf <= beta; a key that
    // expand_private decoded out of range takes the reference's two-transform form.)  Per key of the table when it fits the
    // chunk-sized flag array, else per op.
    {
        const bool by_op = oor_by_op(a);
        const size_t units = a.key_idx ? (by_op ? n : a.n_keys) : n;
        TRY(launch_zero(ctx, w.key_oor, n, s));
        TRY(launch_key_range(ctx, p, c.s2k, by_op ? c.kidx : nullptr, units, w.key_oor, s));
    }
    // 8: kappa <- 0; active = all ops with a legal ctx and key index
    TRY(launch_init_active(ctx, n, w.bad_op, w.done, w.kappa, st, w.act[0], w.ctl, c.sg, (size_t)p->sig_len, s));
    return MLDSA_OK;
}

// One chunk = up to MLDSA_SIGN_MAX_LANES slices ("lanes"), each with its own workspace carve, control block and stream:
// lane 0 runs on the call's stream, lane 1 on the context's second stream, forked and joined with events.  The two
// kernel chains are independent, so the latency-bound kernels of one (the 7-block c~ hash, SampleInBall, the tiny
// bookkeeping kernels) run beside the throughput-bound kernels of the other.  Enqueue only (capturable).
struct SignLane {
    SignWs w;
    SignArgs a;
    hipStream_t st = nullptr;
};

int sign_chunk_enqueue(mldsa_ctx *ctx, const mldsa_params *p, const SignPlan &pl, SignLane *lanes, int n_lanes, hipStream_t s) {
    lanes[0].st = s;
    if (n_lanes > 1) {
        lanes[1].st = parallel_stream(ctx, s);
        MLDSA_HIP_CHECK(hipEventRecord(ctx->fork_ev, s));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(lanes[1].st, ctx->fork_ev, 0));
    }
    const int rounds = (int)pl.m_hint.size();
    bool slots0[2] = {false, false};  // round 0 opened by the (small) prologue's own launch
    for (int i = 0; i < n_lanes; i++) {
        const bool ask = rounds > 0 && lanes[i].w.host_ctl != nullptr && lanes[i].a.export_sigs == nullptr;
        TRY(sign_prologue(ctx, p, lanes[i].w, lanes[i].a, lanes[i].st, ask ? &pl : nullptr, ask ? &slots0[i] : nullptr));
    }
    // 10: while (z, h) = bottom                                            ml_dsa.rs:212
    // Two candidates per op generated at once (k_make_slots): where the plan expects two one-candidate rounds in a row, on a batch
    // large enough for sign_w to be bound by re-reading A_hat.  Rounds pair up: (generate two, test the first) then (test the second).
    const bool look = lookahead_on(ctx, p) && lanes[0].a.n >= LOOKAHEAD_MIN_OPS;
    bool gen2_prev = false;
    // export of finished signatures to host memory (mldsa_sign_host): one lane only (the lanes would share the events)
    const bool exporting = lanes[0].a.export_sigs != nullptr && n_lanes == 1;
    hipStream_t exp_stream = exporting ? parallel_stream(ctx, s) : nullptr;
    for (int round = 0; round < rounds; round++) {
        const bool gen2 = look && !gen2_prev && round + 1 < rounds && pl.one_cand[round] && pl.one_cand[round + 1];
        for (int i = 0; i < n_lanes; i++) {
            const SignLane &L = lanes[i];
            const ChunkKeys c = chunk_keys(p, L.w, L.a);
            // the plan is for a full slice; a short last one only makes its grids generous
            TRY(enqueue_sign_round(ctx, p, L.w, pl, round, std::min(pl.m_hint[round], L.a.n), pl.ns_hint[round], c.kidx, c.s1k, c.s2k, c.t0k,
                                   c.ak, c.sg, L.st, oor_by_op(L.a), gen2_prev, gen2, exporting ? c.xsg : nullptr, exp_stream, round > 0,
                                   /*slots_ready=*/round > 0 || slots0[i], /*slots_next=*/round + 1 < rounds));
        }
        gen2_prev = gen2;
    }
    if (exporting && rounds > 0) {  // the last round's export; the call's stream ends after it
        const SignLane &L = lanes[0];
        const ChunkKeys c = chunk_keys(p, L.w, L.a);
        TRY(enqueue_export(ctx, p, L.w, rounds - 1, std::min(pl.m_hint[rounds - 1], L.a.n), c.sg, c.xsg, s, exp_stream));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->exp_join_ev, 0));
    }
    for (int i = 0; i < n_lanes; i++) {
        const SignLane &L = lanes[i];
        if (L.a.async_mode) {
            // no host wait: what is (with probability < 1e-9) still unfinished is reported per op
            const ChunkKeys c = chunk_keys(p, L.w, L.a);
            TRY(launch_mark_unfinished(ctx, L.w.ctl, rounds & 1, L.w.act[rounds & 1], L.a.status ? L.a.status + L.a.offset : nullptr, c.sg,
                                       (size_t)p->sig_len, L.st));
            // (the exporting call of mldsa_sign_host is launched directly, never captured: sign_batch clears its workspace on
            //  the helper stream like a synchronous call's, 0.3 ms that the caller does not wait for)
            if (!exporting) zeroise_sign_ws(ctx, L.w, L.st);
        }
    }
    if (n_lanes > 1) {
        MLDSA_HIP_CHECK(hipEventRecord(ctx->join_ev, lanes[1].st));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->join_ev, 0));
    }
    return MLDSA_OK;
}

// the one host wait of a synchronous call; stragglers (practically never) get further rounds
int sign_chunk_finish(mldsa_ctx *ctx, const mldsa_params *p, const SignPlan &pl, SignLane *lanes, int n_lanes, hipStream_t s) {
    int round = (int)pl.m_hint.size();
    for (int i = 0; i < n_lanes; i++)
        if (!lanes[i].w.host_ctl)  // (a small call's last k_compact has written what is read below to h_ctl itself)
            MLDSA_HIP_CHECK(hipMemcpyAsync(&ctx->h_ctl[i], lanes[i].w.ctl, sizeof(RoundCtl), hipMemcpyDeviceToHost, s));
    // A small call (one lane, its counts arrive in h_ctl by themselves): the clearing of its secrets is enqueued HERE, behind an event the
    // host waits for instead of the stream -- the launch costs the host ~10 us, which it has while the rounds run and would otherwise spend
    // after them, with the caller waiting (timeline of a one-op call: 22 us between the end of the wait and the return).  The kernel clears
    // only if nothing is left unfinished (k_zero_if_done); otherwise the extra rounds below and sign_batch's background clearing follow.
    ctx->sign_wiped_inline = false;
    bool inline_wipe = false;
    if (n_lanes == 1 && lanes[0].w.host_ctl && !lanes[0].a.export_sigs && round > 0 && hipEventRecord(ctx->small_done_ev, s) == hipSuccess) {
        uint8_t *secrets = reinterpret_cast<uint8_t *>(lanes[0].w.y);
        MLDSA_WIPE(launch_zero_if_done(ctx, lanes[0].w.ctl, round & 1, secrets, (size_t)(lanes[0].w.base + lanes[0].w.bytes - secrets), s));
        inline_wipe = hipEventRecord(ctx->zero_ev, s) == hipSuccess;
        if (!inline_wipe) (void)hipGetLastError();
    } else {
        (void)hipGetLastError();
    }
    for (bool first = true;; first = false) {
        if (first && inline_wipe) {
            if (hipEventSynchronize(ctx->small_done_ev) != hipSuccess) {
                (void)hipGetLastError();
                MLDSA_HIP_CHECK(hipStreamSynchronize(s));
            }
        } else {
            MLDSA_HIP_CHECK(hipStreamSynchronize(s));
        }
        bool left = false;
        for (int i = 0; i < n_lanes; i++) {
            ctx->last_sign_slots += ctx->h_ctl[i].slots_total;
            if (ctx->prof_on) {
                ctx->prof_sign_slots += ctx->h_ctl[i].slots_total;
                ctx->prof_sign_op_rounds += ctx->h_ctl[i].ops_total;
            }
            left |= ctx->h_ctl[i].cnt[round & 1] != 0;
        }
        if (!left) {
            ctx->sign_wiped_inline = first && inline_wipe;  // (k_zero_if_done saw the same zero: the clearing is on its way)
            break;
        }
        for (int i = 0; i < n_lanes; i++) {
            const SignLane &L = lanes[i];
            const ChunkKeys c = chunk_keys(p, L.w, L.a);
            TRY(launch_zero(ctx, &L.w.ctl->slots_total, 2 * sizeof(unsigned long long), s));
            for (int e = 0; e < 2; e++) {
                ctx->stats.sign_extra_rounds++;
                const bool exporting = L.a.export_sigs != nullptr && n_lanes == 1;
                TRY(enqueue_sign_round(ctx, p, L.w, pl, round + e, 64, std::min<size_t>(pl.ns_max, 2048), c.kidx, c.s1k, c.s2k, c.t0k, c.ak,
                                       c.sg, s, oor_by_op(L.a), false, false, exporting ? c.xsg : nullptr,
                                       exporting ? parallel_stream(ctx, s) : nullptr, false, /*slots_ready=*/e == 1, /*slots_next=*/e == 0));
                if (exporting) {  // an extra round exports its own finishers right away
                    TRY(enqueue_export(ctx, p, L.w, round + e, 64, c.sg, c.xsg, s, parallel_stream(ctx, s)));
                    MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->exp_join_ev, 0));
                }
            }
            if (!L.w.host_ctl) MLDSA_HIP_CHECK(hipMemcpyAsync(&ctx->h_ctl[i], L.w.ctl, sizeof(RoundCtl), hipMemcpyDeviceToHost, s));
        }
        round += 2;
    }
    return MLDSA_OK;
}

}  // namespace

// would run_op replay / capture this call (then nothing may be waited for in the middle of its enqueue function)?
static bool op_uses_graph(const mldsa_ctx *ctx, hipStream_t s, int op, size_t n_ops) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return true;
    if (ctx->prof_on) return false;
    return ctx->opt_graphs == 2 || (ctx->opt_graphs == 1 && op == MLDSA_OP_SIGN && n_ops <= GRAPH_AUTO_MAX_OPS);
}

int wait_zeroise(mldsa_ctx *ctx, hipStream_t s) {
    if (!ctx->zero_pending) return MLDSA_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        // the caller is capturing `s` into a graph of its own: an event recorded outside the capture cannot become a dependency
        // of it, and the runtime refuses host waits from a capturing thread -- so the (sub-millisecond) clearing is polled for.
        // (run_op waits on the stream BEFORE it begins a capture of its own: this branch is for callers' captures only.)
        hipError_t e;
        while ((e = hipEventQuery(ctx->zero_ev)) == hipErrorNotReady) std::this_thread::yield();
        if (e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "waiting for the previous call's background clearing inside a stream capture", e);
    } else {
        hipError_t e = hipStreamWaitEvent(s, ctx->zero_ev, 0);
        if (e != hipSuccess) {
            // Seen when a capture of this context was invalidated by another thread's device-wide wait and the helper stream the
            // clearing ran on still counted as capturing for the runtime ("operation not permitted on an event last recorded in a
            // capturing stream").  The ordering this wait stands for is not optional -- the clearing must not run into this call's
            // fresh rows -- so it is established on the host instead (the clearing is sub-millisecond).
            (void)hipGetLastError();
            e = ctx->zero_stream ? hipStreamSynchronize(ctx->zero_stream) : hipErrorUnknown;
            if (e != hipSuccess) { (void)hipGetLastError(); e = device_sync_quiesced(); }
            if (e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "waiting for the previous call's background clearing", e);
        }
    }
    ctx->zero_pending = false;
    return MLDSA_OK;
}

int sign_batch(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
               const int32_t *s1, const int32_t *s2, const int32_t *t0, size_t n_keys, const uint32_t *key_idx,
               const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *rnd,
               uint8_t *sigs, int32_t *status, size_t n_ops, hipStream_t s, const int32_t *a_hat_keys, bool async_mode,
               double plan_stop, uint8_t *export_sigs, hipEvent_t inputs_ev) {
    const mldsa_params *p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "sign: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    const bool own_a = a_hat_keys == nullptr;
    const size_t chunk = std::min(n_ops, ctx->pass_ops_sign);
    const int n_lanes = export_sigs ? 1 : sign_lanes_for(ctx, set, chunk);  // (the export hangs off lane 0's rounds: one lane)
    const size_t per_lane = lane_ops(chunk, n_lanes);
    const SignPlan pl = plan_sign(ctx, set, per_lane, async_mode, plan_stop);
    SignLane lanes[2];
    size_t ws_off = 0;
    for (int i = 0; i < n_lanes; i++) {
        lanes[i].w = SignWs(static_cast<uint8_t *>(ctx->ws) + ws_off, p, per_lane, pl.ns_max, own_a, lookahead_on(ctx, p));
        static_assert(sizeof(lanes) / sizeof(lanes[0]) <= SMALL_CTR_SETS, "one set of arrival counters per lane");
        lanes[i].w.small_ctr = ctx->d_small_ctr + (size_t)i * SMALL_CTR_ENTRIES;
        // a small call's k_compact (one workgroup: at most 256 unfinished ops) reports to the host itself (see launch_compact)
        // (never for a call that exports its signatures round by round: its rounds end with the batch k_compact, which does not report --
        //  one predicate for "k_compact_small writes h_ctl" (enqueue_sign_round) and "the copy of the control block is skipped" (sign_chunk_finish))
        if (per_lane <= 256 && ctx->h_ctl_dev && !pl.m_hint.empty() && !export_sigs) lanes[i].w.host_ctl = ctx->h_ctl_dev + i;
        ws_off += lanes[i].w.bytes;
    }
    if (ctx->ws_bytes < ws_off) return set_error(MLDSA_ERR_NOMEM, "sign: workspace not reserved");
    for (int i = 0; i < n_lanes; i++) {  // where this call's secrets will live (mldsa_debug_secret_residue)
        const uint8_t *secrets = reinterpret_cast<const uint8_t *>(lanes[i].w.y);
        ctx->secret_spans.push_back({(size_t)(secrets - static_cast<uint8_t *>(ctx->ws)), (size_t)(lanes[i].w.base + lanes[i].w.bytes - secrets)});
    }
    if (ctx->zero_pending) {
        // The previous synchronous call is (perhaps) still clearing its secrets on a helper stream.  This call's ExpandA may run
        // beside that -- one is integer-issue-bound, the other a stream of stores -- when it is launched directly on one lane
        // and everything the prologue touches before its wait (A_hat, the key-index scratch) lies below the span being cleared.
        const size_t public_end = (size_t)(reinterpret_cast<uint8_t *>(lanes[0].w.y) - static_cast<uint8_t *>(ctx->ws));
        const bool defer = own_a && n_lanes == 1 && !op_uses_graph(ctx, s, MLDSA_OP_SIGN, chunk) && public_end <= ctx->zero_lo;
        if (defer) ctx->zero_wait_after_ea = true;
        else TRY(wait_zeroise(ctx, s));
    }
    ctx->last_sign_slots = 0;
    int rc = MLDSA_OK;
    for (size_t o = 0; o < n_ops && rc == MLDSA_OK; o += chunk) {
        const size_t n_chunk = (n_ops - o) < chunk ? (n_ops - o) : chunk;
        struct { int op, n_lanes; long spec_target, spec_rows, spec_max, rounds, ahead; SignArgs a[2]; } key;
        memset(&key, 0, sizeof(key));  // the struct is the graph key: no indeterminate padding
        key.op = MLDSA_OP_SIGN; key.n_lanes = n_lanes; key.spec_target = std::min(ctx->opt_spec_target, ctx->spec_target_cap); key.spec_rows = ctx->opt_spec_rows; key.spec_max = ctx->opt_spec_max;
        key.rounds = (long)pl.m_hint.size();  // the planned rounds (options, the asynchronous stop threshold) shape the launch sequence
        key.ahead = ctx->opt_lookahead;
        int live = 0;
        for (int i = 0; i < n_lanes; i++) {
            const size_t lo = std::min(n_chunk, (size_t)i * per_lane), hi = std::min(n_chunk, lo + per_lane);
            if (hi == lo) continue;
            SignArgs &a = key.a[live];
            a.set = set; a.mode = mode; a.rho = rho; a.cap_k = cap_k; a.tr = tr; a.s1 = s1; a.s2 = s2; a.t0 = t0; a.n_keys = n_keys;
            a.key_idx = key_idx; a.msgs = msgs; a.msg_off = msg_off; a.ctxs = ctxs; a.ctx_off = ctx_off; a.rnd = rnd; a.sigs = sigs;
            a.status = status; a.a_hat_keys = a_hat_keys; a.export_sigs = export_sigs; a.inputs_ev = inputs_ev; a.offset = o + lo; a.n = hi - lo; a.n_total = n_ops; a.chunk = per_lane;
            a.async_mode = async_mode ? 1 : 0;
            lanes[live].a = a;
            live++;
        }
        // a call that exports its signatures to host memory forks a helper launch off every round; replaying that shape crashed
        // inside hipGraphLaunch (hip::Graph::UpdateStreams, ROCm 7.0 runtime of this image) for some round counts: launched directly
        rc = run_op(ctx, s, MLDSA_OP_SIGN, n_chunk, &key, sizeof(key), [&](hipStream_t st) { return sign_chunk_enqueue(ctx, p, pl, lanes, live, st); },
                    export_sigs == nullptr);
        if (rc == MLDSA_OK && !async_mode) rc = sign_chunk_finish(ctx, p, pl, lanes, live, s);
    }
    if (ctx->zero_wait_after_ea) {  // the call failed before its prologue got there
        ctx->zero_wait_after_ea = false;
        (void)wait_zeroise(ctx, s);
    }
    if (rc != MLDSA_OK) {
        for (int i = 0; i < n_lanes; i++) zeroise_sign_ws(ctx, lanes[i].w, s);  // also on the error path
        (void)hipStreamSynchronize(s);
    } else if (!async_mode || export_sigs) {
        // Every signature is in place (sign_chunk_finish waited for the stream; mldsa_sign_host's exporting call is asynchronous
        // and ordered on the device: zero_fork_ev follows its last round and its last export).  y, w, c, rho'' ... are cleared on a helper
        // stream, off the caller's critical path (1.1 GB for a 65 536-op ML-DSA-65 call: ~0.2 ms of an 8 ms call): the next
        // op-level call on this context waits for zero_ev on the device before it touches the workspace, destroy / regrow
        // wait for the whole device.
        ZeroSpan spans[2];
        for (int i = 0; i < n_lanes; i++) {
            uint8_t *secrets = reinterpret_cast<uint8_t *>(lanes[i].w.y);
            spans[i] = {secrets, (size_t)(lanes[i].w.base + lanes[i].w.bytes - secrets)};
        }
        if (ctx->sign_wiped_inline) {  // a small call: sign_chunk_finish enqueued the clearing behind the last round, zero_ev follows it on `s`
            ctx->sign_wiped_inline = false;
            ctx->zero_stream = s;
            ctx->zero_pending = true;
            ctx->zero_head_valid = false;
            ctx->zero_lo = (size_t)(reinterpret_cast<uint8_t *>(lanes[0].w.y) - static_cast<uint8_t *>(ctx->ws));
            ctx->zero_hi = ws_off;
        } else if (clear_in_background(ctx, s, nullptr, spans, n_lanes)) {
            ctx->zero_pending = true;
            ctx->zero_head_valid = false;
            ctx->zero_lo = (size_t)(reinterpret_cast<uint8_t *>(lanes[0].w.y) - static_cast<uint8_t *>(ctx->ws));
            ctx->zero_hi = ws_off;
        }
    }
    return rc;
}

// ------------------------------------------------------------------------------------
// Helper streams that really run beside the caller's stream (see ctx.h).
__global__ void k_probe_spin(unsigned *p, int iters) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x2545F491u) p[0] = v;
}
__global__ void k_probe_touch(unsigned *p) {
    if (threadIdx.x == 0) p[1] = 1;
}

bool streams_serialise(mldsa_ctx *ctx, hipStream_t a, hipStream_t b) {
    if (a == b) return true;
    if (!ctx->d_probe && malloc_quiesced((void **)&ctx->d_probe, 256) != hipSuccess) return false;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    const double t0 = now();
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, ctx->d_probe, 40000);  // a few hundred microseconds on one wave
    hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, b, ctx->d_probe);
    (void)hipStreamSynchronize(b);
    const double tb = now() - t0;
    (void)hipStreamSynchronize(a);
    const double ta = now() - t0;
    (void)hipGetLastError();
    return tb > 0.5 * ta;
}

hipStream_t parallel_stream(mldsa_ctx *ctx, hipStream_t s, hipStream_t avoid) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return ctx->aux_stream;
    if (!avoid) {
        for (const auto &pr : ctx->parallel_of)
            if (pr.first == s) return pr.second;
    } else {
        for (const auto &t : ctx->parallel2_of)
            if (t.s == s && t.avoid == avoid) return t.found;
    }
    if (ctx->helper_streams.empty()) ctx->helper_streams.push_back(ctx->aux_stream);
    hipStream_t found = nullptr;
    for (size_t i = 0; i < 8 && !found; i++) {
        if (i == ctx->helper_streams.size()) {
            hipStream_t t = nullptr;
            if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) break;
            ctx->helper_streams.push_back(t);
        }
        hipStream_t c = ctx->helper_streams[i];
        if (c == s || c == avoid) continue;
        if (!streams_serialise(ctx, s, c) && (!avoid || !streams_serialise(ctx, avoid, c))) found = c;
    }
    if (!found) found = ctx->aux_stream;  // every candidate shares a queue: still correct, just not concurrent
    if (!avoid) ctx->parallel_of.emplace_back(s, found);
    else ctx->parallel2_of.push_back({s, avoid, found});
    return found;
}

// ------------------------------------------------------------------------------------
// hipGraph replay.  A call shape = the operation and every argument that ends up in a kernel parameter.
// callers make sure nothing of the context still runs (ensure_workspace and mldsa_ctx_destroy wait for the device,
// mldsa_set_option does the same before it gets here)
void drop_graphs(mldsa_ctx *ctx) {
    for (auto &g : ctx->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    ctx->graphs.clear();
}

int run_op(mldsa_ctx *ctx, hipStream_t s, int op, size_t n_ops, const void *key, size_t key_len,
           const std::function<int(hipStream_t)> &enqueue_fn, bool allow_graph) {
    // Every launcher checks hipGetLastError() after its launch, and that error is sticky per host thread: what a TOLERATED call before
    // this point left behind (destroying the graphs of a replaced workspace, best-effort event bookkeeping after an invalidated
    // capture ...) must not be taken for the failure of the first kernel launched here.  Found with the capture-under-fire test once
    // small signing calls had become short enough for it: "operation not permitted on an event last recorded in a capturing stream"
    // reported by a launch that had nothing to do with events.
    auto enqueue = [&](hipStream_t st) { (void)hipGetLastError(); return enqueue_fn(st); };
    // MLDSA_OPT_GRAPHS: 0 never; 1 signing calls of up to GRAPH_AUTO_MAX_OPS ops -- ~100 launches for a few ms of device
    // work, where the 0.15-0.4 ms of host time a directly launched call costs is a sizeable share of the call; 2 every
    // op-level call.  A replayed graph is NOT faster on the device (the device-driven loop never waits for the host):
    // it costs the call 20-50 us of launch latency and saves the host thread 5-25x of its time per call.
    const bool wanted = ctx->opt_graphs == 2 || (ctx->opt_graphs == 1 && op == MLDSA_OP_SIGN && n_ops <= GRAPH_AUTO_MAX_OPS);
    if (!wanted || !allow_graph || ctx->prof_on) {  // per-stage timing needs the individual launches
        ctx->stats.direct_calls++;
        return enqueue(s);
    }
    // a stream that is itself being captured by the caller: just add our nodes to the caller's graph
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        ctx->stats.direct_calls++;
        return enqueue(s);
    }
    const unsigned char *kb = static_cast<const unsigned char *>(key);
    GraphEntry *hit = nullptr;
    for (auto &g : ctx->graphs)
        if (g.key.size() == key_len && memcmp(g.key.data(), kb, key_len) == 0) { hit = &g; break; }
    const unsigned long long tick = ++ctx->graph_tick;
    if (!hit) {
        // first sighting: remember the shape, launch directly (most shapes of a varied workload never repeat)
        if ((long)ctx->graphs.size() >= ctx->opt_graph_cache) {
            auto lru = std::min_element(ctx->graphs.begin(), ctx->graphs.end(),
                                        [](const GraphEntry &a, const GraphEntry &b) { return a.last_use < b.last_use; });
            // the evicted graph may still be executing (asynchronous calls, the *_host paths): wait for its last launch
            if (lru->exec) {
                if (lru->done) (void)hipEventSynchronize(lru->done);
                (void)hipGraphExecDestroy(lru->exec);
            }
            if (lru->graph) (void)hipGraphDestroy(lru->graph);
            if (lru->done) (void)hipEventDestroy(lru->done);
            ctx->graphs.erase(lru);
        }
        GraphEntry e;
        e.key.assign(kb, kb + key_len);
        e.last_use = tick;
        ctx->graphs.push_back(std::move(e));
        ctx->stats.direct_calls++;
        return enqueue(s);
    }
    hit->last_use = tick;
    // The legacy default stream (NULL) can neither be captured nor take a graph launch that overlaps properly: graphs
    // of calls made on it run on a context-owned stream, ordered after and before the default stream by events.
    hipStream_t gs = s ? s : ctx->graph_stream;
    // a previous call's background clearing of the workspace: ordered here, on the real stream, before any capture begins
    // (an event recorded outside a capture cannot be waited for inside it)
    {
        const int rcz = wait_zeroise(ctx, s);
        if (rcz != MLDSA_OK) return rcz;
    }
    if (!s) {
        MLDSA_HIP_CHECK(hipEventRecord(ctx->graph_fork_ev, s));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(gs, ctx->graph_fork_ev, 0));
    }
    if (hit->never) {
        ctx->stats.direct_calls++;
        return enqueue(s);
    }
    if (!hit->exec) {
        // second sighting: capture and instantiate.  A capture that does not end well -- the application freed device memory on
        // another thread, which waits for every stream of the device and invalidates captures -- is not an error of the call:
        // the shape is launched directly, now and from then on.
        hipGraph_t graph = nullptr;
        int rc;
        hipError_t ee;
        {
            std::shared_lock<std::shared_mutex> cap(capture_mutex());
            ee = hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal);
            rc = ee == hipSuccess ? enqueue(gs) : MLDSA_ERR_DEVICE;
            if (ee == hipSuccess) ee = hipStreamEndCapture(gs, &graph);
        }
        hipGraphExec_t exec = nullptr;
        hipEvent_t done = nullptr;
        bool good = rc == MLDSA_OK && ee == hipSuccess && graph;
        if (good && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; good = false; }
        if (good && hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) { done = nullptr; good = false; }
        if (!good) {
            (void)hipGetLastError();
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            hit->never = true;
            ctx->stats.direct_calls++;
            if (!s) MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->graph_fork_ev, 0));
            return enqueue(s);  // (a call that failed for a reason of its own fails again here, with its own message)
        }
        hit->graph = graph;
        hit->exec = exec;
        hit->done = done;
        ctx->stats.graphs_captured++;
    } else {
        ctx->stats.graph_replays++;
    }
    MLDSA_HIP_CHECK(hipGraphLaunch(hit->exec, gs));
    MLDSA_HIP_CHECK(hipEventRecord(hit->done, gs));
    if (!s) {
        MLDSA_HIP_CHECK(hipEventRecord(ctx->graph_join_ev, gs));
        MLDSA_HIP_CHECK(hipStreamWaitEvent(s, ctx->graph_join_ev, 0));
    }
    return MLDSA_OK;
}

}  // namespace mldsa
