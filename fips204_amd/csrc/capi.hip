// extern "C" entry points of include/mldsa_hip.h: argument validation, context and
// device-memory helpers.  Kernels live in kernels_*.hip, op-level sequencing in pipeline.hip.
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>

#include "ctx.h"
#include "tables.h"

namespace mldsa {

static thread_local char g_err[256] = "";

int set_error(int code, const char *what, hipError_t e) {
    if (e != hipSuccess)
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

// src/lib.rs:639-656 (44), 681-698 (65), 723-740 (87); derived consts lib.rs:129-131
static const mldsa_params PARAMS[3] = {
    {44, 4, 4, 2, 39, 128, 1 << 17, (Q - 1) / 88, 80, 78, 32, 1312, 2560, 2420, 768},
    {65, 6, 5, 4, 49, 192, 1 << 19, (Q - 1) / 32, 55, 196, 48, 1952, 4032, 3309, 768},
    {87, 8, 7, 2, 60, 256, 1 << 19, (Q - 1) / 32, 75, 120, 64, 2592, 4896, 4627, 1024},
};

const mldsa_params *params_of(int set) {
    for (const auto &p : PARAMS)
        if (p.set == set) return &p;
    return nullptr;
}

}  // namespace mldsa

using namespace mldsa;

#define REQUIRE(cond, msg) \
    do { if (!(cond)) return set_error(MLDSA_ERR_PARAM, msg); } while (0)

extern "C" {

const char *mldsa_last_error(void) { return g_err; }

int mldsa_get_params(int set, mldsa_params *out) {
    const mldsa_params *p = params_of(set);
    REQUIRE(p && out, "mldsa_get_params: unknown parameter set");
    *out = *p;
    return MLDSA_OK;
}

int mldsa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mldsa_ctx_create(int device_id, mldsa_ctx **out) {
    REQUIRE(out, "mldsa_ctx_create: NULL out");
    *out = nullptr;
    MLDSA_HIP_CHECK(hipSetDevice(device_id));
    mldsa_ctx *ctx = new (std::nothrow) mldsa_ctx();
    if (!ctx) return set_error(MLDSA_ERR_NOMEM, "mldsa_ctx_create: host allocation failed");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cu = prop.multiProcessorCount;
    std::vector<HostTwiddle> f, i;
    try {
        f = gen_fwd_lane_twiddles();
        i = gen_inv_lane_twiddles();
    } catch (const std::exception &e) {
        delete ctx;
        return set_error(MLDSA_ERR_PARAM, e.what());
    }
    static_assert(sizeof(HostTwiddle) == sizeof(Twiddle), "twiddle layout");
    hipError_t e = hipMalloc((void **)&ctx->d_fwd_tw, f.size() * sizeof(Twiddle));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_inv_tw, i.size() * sizeof(Twiddle));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_fwd_tw, f.data(), f.size() * sizeof(Twiddle), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_inv_tw, i.data(), i.size() * sizeof(Twiddle), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming);
    for (int l = 0; l < MLDSA_SIGN_MAX_LANES; l++) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->lane_stream[l], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->lane_ev[l], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ws_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_lane_count, MLDSA_SIGN_MAX_LANES * sizeof(uint32_t));
    if (e != hipSuccess) {
        mldsa_ctx_destroy(ctx);
        return set_error(MLDSA_ERR_DEVICE, "mldsa_ctx_create: table upload / stream setup", e);
    }
    *out = ctx;
    return MLDSA_OK;
}

void mldsa_ctx_destroy(mldsa_ctx *ctx) {
    if (!ctx) return;
    if (ctx->ws) {
        (void)hipMemset(ctx->ws, 0, ctx->ws_bytes);  // secrets (y, rho'', s1..) live here: types.rs:19
        (void)hipFree(ctx->ws);
    }
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    for (int l = 0; l < MLDSA_SIGN_MAX_LANES; l++) {
        if (ctx->lane_ev[l]) (void)hipEventDestroy(ctx->lane_ev[l]);
        if (ctx->lane_stream[l]) (void)hipStreamDestroy(ctx->lane_stream[l]);
    }
    if (ctx->h_lane_count) (void)hipHostFree(ctx->h_lane_count);
    if (ctx->ws_ev) (void)hipEventDestroy(ctx->ws_ev);
    if (ctx->d_fwd_tw) (void)hipFree(ctx->d_fwd_tw);
    if (ctx->d_inv_tw) (void)hipFree(ctx->d_inv_tw);
    delete ctx;
}

int mldsa_malloc(void **dev_ptr, size_t bytes) {
    REQUIRE(dev_ptr, "mldsa_malloc: NULL out");
    *dev_ptr = nullptr;
    if (bytes == 0) return MLDSA_OK;
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "mldsa_malloc", e);
    return MLDSA_OK;
}

int mldsa_free(void *dev_ptr) {
    if (dev_ptr) MLDSA_HIP_CHECK(hipFree(dev_ptr));
    return MLDSA_OK;
}

int mldsa_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return MLDSA_OK;
    REQUIRE(dst && src, "mldsa_memcpy_h2d: NULL pointer");
    MLDSA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MLDSA_OK;
}

int mldsa_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return MLDSA_OK;
    REQUIRE(dst && src, "mldsa_memcpy_d2h: NULL pointer");
    MLDSA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MLDSA_OK;
}

int mldsa_memset(void *dst, int value, size_t bytes, void *stream) {
    if (bytes == 0) return MLDSA_OK;
    REQUIRE(dst, "mldsa_memset: NULL pointer");
    MLDSA_HIP_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return MLDSA_OK;
}

int mldsa_stream_sync(void *stream) {
    MLDSA_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return MLDSA_OK;
}

// ------------------------------------------------------------------ per-stage timing
int mldsa_profile_enable(mldsa_ctx *ctx, int on) {
    REQUIRE(ctx, "mldsa_profile_enable: NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    ctx->prof_on = on != 0;
    ctx->prof_used = 0;
    ctx->prof_sign_slots = 0;
    return MLDSA_OK;
}

int mldsa_profile_report(mldsa_ctx *ctx, char *buf, size_t buf_len) {
    REQUIRE(ctx && buf && buf_len > 2, "mldsa_profile_report: bad argument");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    MLDSA_HIP_CHECK(hipDeviceSynchronize());
    struct Acc { const char *name; double ms; size_t calls; };
    std::vector<Acc> acc;
    for (size_t i = 0; i < ctx->prof_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess) continue;
        bool found = false;
        for (auto &a : acc)
            if (strcmp(a.name, ctx->prof_name[i]) == 0) { a.ms += ms; a.calls++; found = true; break; }
        if (!found) acc.push_back({ctx->prof_name[i], ms, 1});
    }
    std::string out = "{";
    for (size_t i = 0; i < acc.size(); i++) {
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"ms\": %.6f, \"calls\": %zu}", i ? ", " : "", acc[i].name, acc[i].ms, acc[i].calls);
        out += tmp;
    }
    if (ctx->prof_sign_slots) {
        char tmp[96];
        snprintf(tmp, sizeof(tmp), "%s\"_sign_slots\": {\"ms\": 0, \"calls\": %llu}", acc.empty() ? "" : ", ", ctx->prof_sign_slots);
        out += tmp;
    }
    out += "}";
    if (out.size() + 1 > buf_len) return set_error(MLDSA_ERR_PARAM, "mldsa_profile_report: buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    ctx->prof_used = 0;
    ctx->prof_sign_slots = 0;
    return MLDSA_OK;
}

// ------------------------------------------------------------------ seam-level primitives
int mldsa_ntt(mldsa_ctx *ctx, const int32_t *w, int32_t *w_hat, size_t n_polys, void *stream) {
    REQUIRE(ctx && (n_polys == 0 || (w && w_hat)), "mldsa_ntt: NULL pointer");
    return launch_ntt(ctx, w, w_hat, n_polys, (hipStream_t)stream);
}

int mldsa_inv_ntt(mldsa_ctx *ctx, const int32_t *w_hat, int32_t *w, size_t n_polys, void *stream) {
    REQUIRE(ctx && (n_polys == 0 || (w && w_hat)), "mldsa_inv_ntt: NULL pointer");
    return launch_inv_ntt(ctx, w_hat, w, n_polys, (hipStream_t)stream);
}

int mldsa_to_mont(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n_polys, void *stream) {
    REQUIRE(ctx && (n_polys == 0 || (in && out)), "mldsa_to_mont: NULL pointer");
    return launch_to_mont(ctx, in, out, n_polys, (hipStream_t)stream);
}

int mldsa_mat_vec_mul(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *u_hat,
                      int32_t *w_hat, size_t n_ops, void *stream) {
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_mat_vec_mul: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (a_hat && u_hat && w_hat)), "mldsa_mat_vec_mul: NULL pointer");
    return launch_mat_vec_mul(ctx, p->k, p->l, a_hat, u_hat, w_hat, n_ops, (hipStream_t)stream);
}

int mldsa_pointwise_mont(mldsa_ctx *ctx, const int32_t *c_hat, const int32_t *v_hat_mont,
                         int32_t *out, size_t polys_per_op, size_t n_ops, void *stream) {
    REQUIRE(ctx && (n_ops * polys_per_op == 0 || (c_hat && v_hat_mont && out)), "mldsa_pointwise_mont: NULL pointer");
    return launch_pointwise_mont(ctx, c_hat, v_hat_mont, out, polys_per_op, n_ops, (hipStream_t)stream);
}

int mldsa_add_vector_ntt(mldsa_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out,
                         size_t n_polys, void *stream) {
    REQUIRE(ctx && (n_polys == 0 || (a && b && out)), "mldsa_add_vector_ntt: NULL pointer");
    return launch_add(ctx, a, b, out, n_polys, (hipStream_t)stream);
}

int mldsa_infinity_norm(mldsa_ctx *ctx, const int32_t *polys, size_t polys_per_op, size_t n_ops,
                        int32_t *norms, void *stream) {
    REQUIRE(ctx && polys_per_op > 0 && (n_ops == 0 || (polys && norms)), "mldsa_infinity_norm: bad argument");
    return launch_infinity_norm(ctx, polys, polys_per_op, n_ops, norms, (hipStream_t)stream);
}

int mldsa_verify_arith(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *z,
                       const int32_t *c, const int32_t *t1_d2_hat_mont, int32_t *w_out,
                       size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_verify_arith: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (a_hat && z && c && t1_d2_hat_mont && w_out)), "mldsa_verify_arith: NULL pointer");
    return launch_verify_arith(ctx, set, a_hat, z, c, t1_d2_hat_mont, nullptr, w_out, n_ops, (hipStream_t)stream);
}

// ------------------------------------------------------------------ samplers
int mldsa_expand_a(mldsa_ctx *ctx, int set, const uint8_t *rho, int32_t *a_hat, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_expand_a: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (rho && a_hat)), "mldsa_expand_a: NULL pointer");
    return launch_expand_a(ctx, set, rho, 32, nullptr, a_hat, n_ops, (hipStream_t)stream);
}

int mldsa_expand_s(mldsa_ctx *ctx, int set, const uint8_t *rho_prime, int32_t *s1s2, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_expand_s: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (rho_prime && s1s2)), "mldsa_expand_s: NULL pointer");
    return launch_expand_s(ctx, set, rho_prime, 64, s1s2, n_ops, (hipStream_t)stream);
}

int mldsa_expand_mask(mldsa_ctx *ctx, int set, const uint8_t *rho_pp, const uint16_t *kappa, int32_t *y,
                      size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_expand_mask: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (rho_pp && kappa && y)), "mldsa_expand_mask: NULL pointer");
    return launch_expand_mask(ctx, set, rho_pp, 64, kappa, 0, nullptr, y, n_ops, (hipStream_t)stream);
}

int mldsa_sample_in_ball(mldsa_ctx *ctx, int set, const uint8_t *c_tilde, int32_t *c, size_t n_ops, void *stream) {
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sample_in_ball: unknown parameter set");
    REQUIRE(ctx && (n_ops == 0 || (c_tilde && c)), "mldsa_sample_in_ball: NULL pointer");
    return launch_sample_in_ball(ctx, set, c_tilde, (size_t)p->ctilde_len, c, n_ops, (hipStream_t)stream);
}

// ------------------------------------------------------------------ op-level API
int mldsa_verify(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr,
                 const int32_t *t1_d2_hat_mont, const uint32_t *key_idx, const uint8_t *msgs,
                 const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                 const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_verify: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_verify: bad mode");
    REQUIRE(ctx && (n_ops == 0 || (rho && tr && t1_d2_hat_mont && msg_off && sigs && ok)), "mldsa_verify: NULL pointer");
    OpGuard guard(ctx, (hipStream_t)stream);
    return verify_batch(ctx, set, mode, rho, tr, t1_d2_hat_mont, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops,
                        (hipStream_t)stream);
}

int mldsa_pk_expand(mldsa_ctx *ctx, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1_d2_hat_mont,
                    size_t n_keys, void *stream) {
    REQUIRE(params_of(set), "mldsa_pk_expand: unknown parameter set");
    REQUIRE(ctx && (n_keys == 0 || (pk && rho && tr && t1_d2_hat_mont)), "mldsa_pk_expand: NULL pointer");
    return pk_expand_batch(ctx, set, pk, rho, tr, t1_d2_hat_mont, n_keys, (hipStream_t)stream);
}

int mldsa_sk_expand(mldsa_ctx *ctx, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k, uint8_t *tr,
                    int32_t *s_1_hat_mont, int32_t *s_2_hat_mont, int32_t *t_0_hat_mont, size_t n_keys, void *stream) {
    REQUIRE(params_of(set), "mldsa_sk_expand: unknown parameter set");
    REQUIRE(ctx && (n_keys == 0 || (sk && rho && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont)),
            "mldsa_sk_expand: NULL pointer");
    return sk_expand_batch(ctx, set, sk, rho, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, n_keys, (hipStream_t)stream);
}

int mldsa_keygen(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys, void *stream) {
    REQUIRE(params_of(set), "mldsa_keygen: unknown parameter set");
    REQUIRE(ctx && (n_keys == 0 || (xi && pk && sk)), "mldsa_keygen: NULL pointer");
    OpGuard guard(ctx, (hipStream_t)stream);
    return keygen_batch(ctx, set, xi, pk, sk, n_keys, (hipStream_t)stream);
}

int mldsa_sign(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
               const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, const int32_t *t_0_hat_mont,
               const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
               const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_sign: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_sign: bad mode");
    REQUIRE(ctx && (n_ops == 0 || (rho && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont && msg_off && rnd && sigs)),
            "mldsa_sign: NULL pointer");
    OpGuard guard(ctx, (hipStream_t)stream);
    return sign_batch(ctx, set, mode, rho, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, key_idx, msgs, msg_off, ctxs,
                      ctx_off, rnd, sigs, status, n_ops, (hipStream_t)stream);
}

int mldsa_verify_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *tr,
                          const int32_t *t1_d2_hat_mont, const uint32_t *key_idx, const uint8_t *msgs,
                          const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                          const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_verify_cached_a: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_verify_cached_a: bad mode");
    REQUIRE(ctx && (n_ops == 0 || (a_hat && tr && t1_d2_hat_mont && msg_off && sigs && ok)), "mldsa_verify_cached_a: NULL pointer");
    OpGuard guard(ctx, (hipStream_t)stream);
    return verify_batch(ctx, set, mode, nullptr, tr, t1_d2_hat_mont, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops,
                        (hipStream_t)stream, a_hat);
}

int mldsa_sign_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *cap_k,
                        const uint8_t *tr, const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont,
                        const int32_t *t_0_hat_mont, const uint32_t *key_idx, const uint8_t *msgs,
                        const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                        const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream) {
    REQUIRE(params_of(set), "mldsa_sign_cached_a: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_sign_cached_a: bad mode");
    REQUIRE(ctx && (n_ops == 0 || (a_hat && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont && msg_off && rnd && sigs)),
            "mldsa_sign_cached_a: NULL pointer");
    OpGuard guard(ctx, (hipStream_t)stream);
    return sign_batch(ctx, set, mode, nullptr, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, key_idx, msgs, msg_off, ctxs,
                      ctx_off, rnd, sigs, status, n_ops, (hipStream_t)stream, a_hat);
}

}  // extern "C"
