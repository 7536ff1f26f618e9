// extern "C" entry points of include/mldsa_hip.h: argument validation, context and
// device-memory helpers.  Kernels live in kernels_*.hip, op-level sequencing in pipeline.hip.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <algorithm>
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
// every entry that takes a context: argument check, then bind the calling thread to the context's device
#define ENTER(ctx, name)                      \
    REQUIRE(ctx, name ": NULL context");      \
    DeviceGuard _dev_guard((ctx)->device)

static bool mode_ok(int mode) { return mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH; }

// Measurement knobs from the environment (include/mldsa_hip.h "Environment"): read ONLY when the process asks for them with
// MLDSA_TUNING_ENV=1 (or in a -DMLDSA_TUNING build) -- a signing library must not be re-scheduled by whatever MLDSA_* variables a
// host's environment happens to hold.  Each knob sets the initial value of one per-context option; out-of-range values are ignored.
static bool tuning_env_on() {
#if defined(MLDSA_TUNING)
    return true;
#else
    const char *v = getenv("MLDSA_TUNING_ENV");  // (asked at every mldsa_ctx_create: the tests switch it per context)
    return v && v[0] == '1' && v[1] == 0;
#endif
}
static long env_long(const char *name, long lo, long hi, long dflt) {
    if (!tuning_env_on()) return dflt;
    if (const char *e = getenv(name)) {
        const long v = atol(e);
        if (v >= lo && v <= hi) return v;
    }
    return dflt;
}

extern "C" {

const char *mldsa_last_error(void) { return g_err; }

int mldsa_get_params(int set, mldsa_params *out) {
    const mldsa_params *p = params_of(set);
    REQUIRE(p && out, "mldsa_get_params: unknown parameter set");
    *out = *p;
    return MLDSA_OK;
}

int mldsa_abi_version(void) { return MLDSA_ABI_VERSION; }

// One O(n) pass over a caller's offset table: non-decreasing, so that every op's length is off[i + 1] - off[i] without
// wrapping and every op's bytes lie inside [off[0], off[n_ops]).  No device, no context: hosts call it on their own tables
// before the device-resident entry points if they want a call-level error instead of per-op refusals.
int mldsa_check_offsets(const uint64_t *off, size_t n_ops) {
    REQUIRE(off || n_ops == 0, "mldsa_check_offsets: NULL table");
    for (size_t i = 0; i < n_ops; i++)
        if (off[i + 1] < off[i]) {
            char msg[96];
            snprintf(msg, sizeof(msg), "offset table decreases at entry %zu (%llu after %llu)", i + 1, (unsigned long long)off[i + 1],
                     (unsigned long long)off[i]);
            return set_error(MLDSA_ERR_PARAM, msg);
        }
    return MLDSA_OK;
}

int mldsa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mldsa_ctx_device(const mldsa_ctx *ctx) { return ctx ? ctx->device : MLDSA_ERR_PARAM; }

int mldsa_ctx_create(int device_id, mldsa_ctx **out) {
    REQUIRE(out, "mldsa_ctx_create: NULL out");
    *out = nullptr;
    int n_dev = 0;
    MLDSA_HIP_CHECK(hipGetDeviceCount(&n_dev));
    REQUIRE(device_id >= 0 && device_id < n_dev, "mldsa_ctx_create: no such device");
    DeviceGuard guard(device_id);  // the caller's current device is left as it was
    mldsa_ctx *ctx = new (std::nothrow) mldsa_ctx();
    if (!ctx) return set_error(MLDSA_ERR_NOMEM, "mldsa_ctx_create: host allocation failed");
    ctx->device = device_id;
    // measurement knobs (MLDSA_TUNING_ENV=1 only; every name is listed in include/mldsa_hip.h): initial values of the per-context options
    ctx->opt_graphs = env_long("MLDSA_GRAPHS", 0, 2, ctx->opt_graphs);
    ctx->opt_coop_hash = env_long("MLDSA_COOP_HASH", 0, 1, ctx->opt_coop_hash);
    ctx->opt_small_fused = env_long("MLDSA_SMALL_FUSED", 0, (long)SMALL_FUSED_MAX, ctx->opt_small_fused);
    ctx->small_keygen_max = (size_t)env_long("MLDSA_SMALL_KEYGEN_MAX", 0, (long)SMALL_FUSED_MAX, (long)ctx->small_keygen_max);
    ctx->small_sign_max = (size_t)env_long("MLDSA_SMALL_SIGN_MAX", 0, 256, (long)ctx->small_sign_max);
    ctx->small_sign_front = env_long("MLDSA_SMALL_SIGN_FRONT", 0, 1, ctx->small_sign_front);
    ctx->small_sign_back = env_long("MLDSA_SMALL_SIGN_BACK", 0, 1, ctx->small_sign_back);
    ctx->small_back_slots_max = (size_t)env_long("MLDSA_SMALL_BACK_SLOTS_MAX", 0, 1 << 20, (long)ctx->small_back_slots_max);
    ctx->small_sign_spec = env_long("MLDSA_SMALL_SIGN_SPEC", 0, 32, ctx->small_sign_spec);
    ctx->coop_hash_max = (size_t)env_long("MLDSA_COOP_HASH_MAX", 0, 1 << 20, (long)ctx->coop_hash_max);
    ctx->coop_mask_max = (size_t)env_long("MLDSA_COOP_MASK_MAX", 0, 1 << 20, (long)ctx->coop_mask_max);
    ctx->coop_a_max = (size_t)env_long("MLDSA_COOP_A_MAX", 0, 1 << 20, (long)ctx->coop_a_max);
    ctx->coop_mu_max = (size_t)env_long("MLDSA_COOP_MU_MAX", 0, 1 << 20, (long)ctx->coop_mu_max);
    ctx->coop_sib_max = (size_t)env_long("MLDSA_COOP_SIB_MAX", 0, 1 << 20, (long)ctx->coop_sib_max);
    ctx->opt_spec_target = env_long("MLDSA_SPEC_TARGET", 1, 524288, ctx->opt_spec_target);
    ctx->opt_spec_max = env_long("MLDSA_SPEC_MAX", 1, 64, ctx->opt_spec_max);
    ctx->opt_spec_rows = env_long("MLDSA_SPEC_ROWS", 1, 524288, ctx->opt_spec_rows);
    ctx->opt_sign_lanes = env_long("MLDSA_SIGN_LANES", 0, 2, ctx->opt_sign_lanes);
    ctx->opt_lookahead = env_long("MLDSA_LOOKAHEAD", 0, 2, ctx->opt_lookahead);
    ctx->opt_va_blocks = env_long("MLDSA_VA_BLOCKS_PER_CU", 1, 64, ctx->opt_va_blocks);
    ctx->opt_host_sub_verify = env_long("MLDSA_HOST_SUB_VERIFY", 64, 65536, ctx->opt_host_sub_verify);
    ctx->opt_host_sub_sign = env_long("MLDSA_HOST_SUB_SIGN", 64, 65536, ctx->opt_host_sub_sign);
    ctx->opt_host_direct = env_long("MLDSA_HOST_DIRECT", 0, 1, ctx->opt_host_direct);
    ctx->opt_ws_cap_bytes = (size_t)env_long("MLDSA_WORKSPACE_CAP_MB", 0, 1L << 20, 0) << 20;
    ctx->pass_ops = (size_t)env_long("MLDSA_PASS_OPS", 256, 1 << 20, (long)ctx->pass_ops);
    ctx->pass_ops_sign = (size_t)env_long("MLDSA_PASS_OPS_SIGN", 256, 1 << 20, (long)ctx->pass_ops_sign);
    ctx->pass_ops_cfg = ctx->pass_ops;
    ctx->pass_ops_sign_cfg = ctx->pass_ops_sign;
    ctx->opt_spec_rows_cfg = ctx->opt_spec_rows;
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
    hipError_t e = malloc_quiesced((void **)&ctx->d_fwd_tw, f.size() * sizeof(Twiddle));
    if (e == hipSuccess) e = malloc_quiesced((void **)&ctx->d_inv_tw, i.size() * sizeof(Twiddle));
    if (e == hipSuccess) e = memcpy_quiesced(ctx->d_fwd_tw, f.data(), f.size() * sizeof(Twiddle), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = memcpy_quiesced(ctx->d_inv_tw, i.data(), i.size() * sizeof(Twiddle), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join2_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->exp_fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->exp_join_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ws_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->zero_fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->zero_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->zero_head_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->small_done_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->graph_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->graph_fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->graph_join_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = host_malloc_quiesced((void **)&ctx->h_ctl, 2 * sizeof(RoundCtl), hipHostMallocMapped);
    if (e == hipSuccess && hipHostGetDevicePointer((void **)&ctx->h_ctl_dev, ctx->h_ctl, 0) != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_ctl_dev = nullptr;  // (then every signing call copies its control block down, as large calls do)
    }
    if (e == hipSuccess) e = malloc_quiesced((void **)&ctx->d_small_ctr, SMALL_CTR_SETS * SMALL_CTR_ENTRIES * sizeof(uint32_t));
    if (e == hipSuccess) e = memset_quiesced(ctx->d_small_ctr, 0, SMALL_CTR_SETS * SMALL_CTR_ENTRIES * sizeof(uint32_t));
    if (e != hipSuccess) {
        mldsa_ctx_destroy(ctx);
        return set_error(MLDSA_ERR_DEVICE, "mldsa_ctx_create: table upload / stream setup", e);
    }
    memset(ctx->h_ctl, 0, 2 * sizeof(RoundCtl));
    {   // the signing tail reads some of its arguments straight from the kernarg segment (kernels_sign.hip late_arg): check, once per
        // context, that this toolchain and driver lay the segment out the way that code assumes -- a mismatch fails here, loudly
        uint32_t *d_word = nullptr, h_word = 0;
        e = malloc_quiesced((void **)&d_word, sizeof(uint32_t));
        // (no memset of the word: the kernel always writes it -- and a NULL-stream memset is not ordered against the non-blocking stream)
        int rc = e == hipSuccess ? late_arg_selftest(ctx->aux_stream, d_word) : MLDSA_ERR_DEVICE;
        if (rc == MLDSA_OK && hipStreamSynchronize(ctx->aux_stream) != hipSuccess) rc = MLDSA_ERR_DEVICE;
        if (rc == MLDSA_OK && memcpy_quiesced(&h_word, d_word, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = MLDSA_ERR_DEVICE;
        if (d_word) (void)free_quiesced(d_word);
        if (rc != MLDSA_OK || h_word != 0x80000000u) {
            (void)hipGetLastError();
            mldsa_ctx_destroy(ctx);
            return set_error(MLDSA_ERR_DEVICE, rc != MLDSA_OK ? "mldsa_ctx_create: kernel-argument self-test did not run"
                                                               : "mldsa_ctx_create: kernel arguments are not where late_arg() reads them (toolchain / code-object ABI changed): refusing to sign with this build");
        }
    }
    *out = ctx;
    return MLDSA_OK;
}

void mldsa_ctx_destroy(mldsa_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)device_sync_quiesced();  // (device-wide waits and frees: never while another context captures a stream, ctx.h)
    drop_graphs(ctx);
    host_stage_destroy(ctx);
    if (ctx->ws) {
        MLDSA_WIPE(memset_quiesced(ctx->ws, 0, ctx->ws_bytes));  // secrets (y, rho'', s1..) live here: types.rs:19
        if (!ctx->ws_external) (void)free_quiesced(ctx->ws);
    }
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    if (ctx->join2_ev) (void)hipEventDestroy(ctx->join2_ev);
    if (ctx->exp_fork_ev) (void)hipEventDestroy(ctx->exp_fork_ev);
    if (ctx->exp_join_ev) (void)hipEventDestroy(ctx->exp_join_ev);
    for (size_t i = 1; i < ctx->helper_streams.size(); i++) (void)hipStreamDestroy(ctx->helper_streams[i]);  // [0] is aux_stream
    if (ctx->d_probe) (void)free_quiesced(ctx->d_probe);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->graph_fork_ev) (void)hipEventDestroy(ctx->graph_fork_ev);
    if (ctx->graph_join_ev) (void)hipEventDestroy(ctx->graph_join_ev);
    if (ctx->graph_stream) (void)hipStreamDestroy(ctx->graph_stream);
    if (ctx->h_ctl) (void)host_free_quiesced(ctx->h_ctl);
    if (ctx->d_small_ctr) (void)free_quiesced(ctx->d_small_ctr);
    if (ctx->ws_ev) (void)hipEventDestroy(ctx->ws_ev);
    if (ctx->zero_fork_ev) (void)hipEventDestroy(ctx->zero_fork_ev);
    if (ctx->zero_ev) (void)hipEventDestroy(ctx->zero_ev);
    if (ctx->zero_head_ev) (void)hipEventDestroy(ctx->zero_head_ev);
    if (ctx->small_done_ev) (void)hipEventDestroy(ctx->small_done_ev);
    if (ctx->d_fwd_tw) (void)free_quiesced(ctx->d_fwd_tw);
    if (ctx->d_inv_tw) (void)free_quiesced(ctx->d_inv_tw);
    delete ctx;
}

int mldsa_set_option(mldsa_ctx *ctx, int option, long value) {
    REQUIRE(ctx, "mldsa_set_option: NULL context");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    switch (option) {
        case MLDSA_OPT_GRAPHS:
            REQUIRE(value >= 0 && value <= 2, "mldsa_set_option: MLDSA_OPT_GRAPHS is 0, 1 or 2");
            ctx->opt_graphs = value;
            return MLDSA_OK;
        case MLDSA_OPT_SPEC_TARGET:
            REQUIRE(value >= 1 && value <= 524288, "mldsa_set_option: MLDSA_OPT_SPEC_TARGET out of range");
            ctx->opt_spec_target = value;
            return MLDSA_OK;
        case MLDSA_OPT_SPEC_MAX:
            REQUIRE(value >= 1 && value <= 64, "mldsa_set_option: MLDSA_OPT_SPEC_MAX out of range");  // k_resolve scans one wave of candidates
            ctx->opt_spec_max = value;
            return MLDSA_OK;
        case MLDSA_OPT_VA_BLOCKS_PER_CU:
            REQUIRE(value >= 1 && value <= 64, "mldsa_set_option: MLDSA_OPT_VA_BLOCKS_PER_CU out of range");
            ctx->opt_va_blocks = value;
            return MLDSA_OK;
        case MLDSA_OPT_GRAPH_CACHE: {
            REQUIRE(value >= 1 && value <= 4096, "mldsa_set_option: MLDSA_OPT_GRAPH_CACHE out of range");
            ctx->opt_graph_cache = value;
            return MLDSA_OK;
        }
        case MLDSA_OPT_SIGN_ROUNDS:
            REQUIRE(value >= 0 && value <= 64, "mldsa_set_option: MLDSA_OPT_SIGN_ROUNDS out of range");
            ctx->opt_sign_rounds = value;
            return MLDSA_OK;
        case MLDSA_OPT_SIGN_LANES:
            REQUIRE(value >= 0 && value <= 2, "mldsa_set_option: MLDSA_OPT_SIGN_LANES is 0 (auto), 1 or 2");
            ctx->opt_sign_lanes = value;
            return MLDSA_OK;
        case MLDSA_OPT_SIGN_ASYNC_EXP:
            REQUIRE(value >= 1 && value <= 12, "mldsa_set_option: MLDSA_OPT_SIGN_ASYNC_EXP is 1 .. 12");
            ctx->async_stop = std::pow(10.0, -(double)value);
            return MLDSA_OK;
        case MLDSA_OPT_SIGN_LOOKAHEAD:
            REQUIRE(value >= 0 && value <= 2, "mldsa_set_option: MLDSA_OPT_SIGN_LOOKAHEAD is 0, 1 or 2");
            ctx->opt_lookahead = value;  // changes the workspace layout: the next signing call reserves for it
            return MLDSA_OK;
        case MLDSA_OPT_SIGN_CT0_EXACT:
            REQUIRE(value == 0 || value == 1, "mldsa_set_option: MLDSA_OPT_SIGN_CT0_EXACT is 0 or 1");
            if (ctx->opt_ct0_exact != value) {  // the flag is a kernel argument of captured launches
                DeviceGuard dg(ctx->device);
                MLDSA_HIP_CHECK(device_sync_quiesced());  // a replayed graph may still be running
                drop_graphs(ctx);
            }
            ctx->opt_ct0_exact = value;
            return MLDSA_OK;
        case MLDSA_OPT_COOP_HASH:
            REQUIRE(value == 0 || value == 1, "mldsa_set_option: MLDSA_OPT_COOP_HASH is 0 or 1");
            if (ctx->opt_coop_hash != value) {  // which kernel a captured call launches
                DeviceGuard dg(ctx->device);
                MLDSA_HIP_CHECK(device_sync_quiesced());
                drop_graphs(ctx);
            }
            ctx->opt_coop_hash = value;
            return MLDSA_OK;
        case MLDSA_OPT_SMALL_FUSED:
            REQUIRE(value >= 0 && value <= (long)SMALL_FUSED_MAX, "mldsa_set_option: MLDSA_OPT_SMALL_FUSED is 0 ... 1024");
            if (ctx->opt_small_fused != value) {  // which kernels a captured call launches
                DeviceGuard dg(ctx->device);
                MLDSA_HIP_CHECK(device_sync_quiesced());
                drop_graphs(ctx);
            }
            ctx->opt_small_fused = value;
            return MLDSA_OK;
        case MLDSA_OPT_WORKSPACE_CAP_MB:
            REQUIRE(value >= 0 && value <= (1L << 20), "mldsa_set_option: MLDSA_OPT_WORKSPACE_CAP_MB out of range");
            ctx->opt_ws_cap_bytes = (size_t)value << 20;
            restore_pass_sizes(ctx);  // a new cap: the passes start from the configured sizes again (reserve_workspace)
            return MLDSA_OK;
        default: return set_error(MLDSA_ERR_PARAM, "mldsa_set_option: unknown option");
    }
}

long mldsa_get_option(const mldsa_ctx *ctx, int option) {
    if (!ctx) return MLDSA_ERR_PARAM;
    // (the options are per-context state like everything else: read under the context's mutex -- a reader beside another thread's
    //  mldsa_set_option was a data race, found by the ThreadSanitizer leg of tests/cpp/fuzz_host.cpp)
    std::lock_guard<std::mutex> lk(const_cast<mldsa_ctx *>(ctx)->op_mutex);
    switch (option) {
        case MLDSA_OPT_GRAPHS: return ctx->opt_graphs;
        case MLDSA_OPT_SPEC_TARGET: return ctx->opt_spec_target;
        case MLDSA_OPT_SPEC_MAX: return ctx->opt_spec_max;
        case MLDSA_OPT_VA_BLOCKS_PER_CU: return ctx->opt_va_blocks;
        case MLDSA_OPT_GRAPH_CACHE: return ctx->opt_graph_cache;
        case MLDSA_OPT_SIGN_ROUNDS: return ctx->opt_sign_rounds;
        case MLDSA_OPT_SIGN_LANES: return ctx->opt_sign_lanes;
        case MLDSA_OPT_SIGN_CT0_EXACT: return ctx->opt_ct0_exact;
        case MLDSA_OPT_SIGN_LOOKAHEAD: return ctx->opt_lookahead;
        case MLDSA_OPT_SIGN_ASYNC_EXP: return std::lround(-std::log10(ctx->async_stop));
        case MLDSA_OPT_WORKSPACE_CAP_MB: return (long)(ctx->opt_ws_cap_bytes >> 20);
        case MLDSA_OPT_COOP_HASH: return ctx->opt_coop_hash;
        case MLDSA_OPT_SMALL_FUSED: return ctx->opt_small_fused;
        default: return MLDSA_ERR_PARAM;
    }
}

int mldsa_get_stats(mldsa_ctx *ctx, mldsa_stats *out) { return mldsa_get_stats_sized(ctx, out, sizeof(mldsa_stats)); }

// a client built against an older (shorter) or newer (longer) mldsa_stats names the size of ITS struct: only that much is written,
// fields this build does not know read as zero
int mldsa_get_stats_sized(mldsa_ctx *ctx, void *out, size_t out_bytes) {
    REQUIRE(ctx && out, "mldsa_get_stats: NULL pointer");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    memset(out, 0, out_bytes);
    memcpy(out, &ctx->stats, std::min(out_bytes, sizeof(mldsa_stats)));
    return MLDSA_OK;
}

int mldsa_ctx_set_workspace(mldsa_ctx *ctx, void *dev_buf, size_t bytes) {
    ENTER(ctx, "mldsa_ctx_set_workspace");
    REQUIRE((dev_buf != nullptr) == (bytes != 0), "mldsa_ctx_set_workspace: a buffer and its size, or NULL and 0");
    REQUIRE(((uintptr_t)dev_buf & 255) == 0, "mldsa_ctx_set_workspace: the buffer must be 256-byte aligned");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    MLDSA_HIP_CHECK(device_sync_quiesced());  // nothing of the context still runs in the old buffer
    ctx->zero_pending = ctx->zero_head_valid = ctx->zero_wait_after_ea = false;
    drop_graphs(ctx);                         // captured launches point into it
    if (ctx->ws) {
        MLDSA_WIPE(memset_quiesced(ctx->ws, 0, ctx->ws_bytes));
        if (!ctx->ws_external) MLDSA_HIP_CHECK(free_quiesced(ctx->ws));
    }
    ctx->secret_spans.clear();
    ctx->ws = dev_buf;
    ctx->ws_bytes = bytes;
    ctx->ws_external = dev_buf != nullptr;
    restore_pass_sizes(ctx);  // another buffer (or the context's own again): pass sizes a smaller one forced do not stick
    return MLDSA_OK;
}

int mldsa_reserve(mldsa_ctx *ctx, int set, int op, size_t n_ops) {
    ENTER(ctx, "mldsa_reserve");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_reserve: unknown parameter set");
    REQUIRE(op == MLDSA_OP_KEYGEN || op == MLDSA_OP_SIGN || op == MLDSA_OP_VERIFY, "mldsa_reserve: unknown operation");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    return n_ops ? reserve_workspace(ctx, p, op, n_ops, true) : MLDSA_OK;
}

// ---- test support: is anything secret left behind? --------------------------------------------------------------------
int mldsa_debug_count_nonzero(const void *dev_ptr, size_t bytes, size_t *nonzero) {
    REQUIRE(nonzero && (dev_ptr || bytes == 0), "mldsa_debug_count_nonzero: NULL pointer");
    return count_nonzero_dev(dev_ptr, bytes, nonzero);
}

int mldsa_debug_secret_residue(mldsa_ctx *ctx, size_t *scanned_bytes, size_t *nonzero_bytes) {
    REQUIRE(scanned_bytes && nonzero_bytes, "mldsa_debug_secret_residue: NULL pointer");
    ENTER(ctx, "mldsa_debug_secret_residue");
    std::lock_guard<std::mutex> host_lk(ctx->host_mutex);
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    // everything the context enqueued has run, the background clearing of the last call included; nothing is cleared HERE
    MLDSA_HIP_CHECK(device_sync_quiesced());
    size_t scanned = 0, nonzero = 0;
    for (const auto &sp : ctx->secret_spans) {
        if (!ctx->ws || sp.first + sp.second > ctx->ws_bytes) continue;  // the workspace was replaced since (cleared before it was freed)
        size_t nz = 0;
        const int rc = count_nonzero_dev(static_cast<const uint8_t *>(ctx->ws) + sp.first, sp.second, &nz);
        if (rc != MLDSA_OK) return rc;
        scanned += sp.second;
        nonzero += nz;
    }
    const int rc = host_stage_residue(ctx, &scanned, &nonzero);
    if (rc != MLDSA_OK) return rc;
    *scanned_bytes = scanned;
    *nonzero_bytes = nonzero;
    return MLDSA_OK;
}

int mldsa_malloc(void **dev_ptr, size_t bytes) {
    REQUIRE(dev_ptr, "mldsa_malloc: NULL out");
    *dev_ptr = nullptr;
    if (bytes == 0) return MLDSA_OK;
    hipError_t e = malloc_quiesced(dev_ptr, bytes);
    if (e != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "mldsa_malloc", e);
    return MLDSA_OK;
}

int mldsa_ctx_malloc(mldsa_ctx *ctx, void **dev_ptr, size_t bytes) {
    ENTER(ctx, "mldsa_ctx_malloc");
    return mldsa_malloc(dev_ptr, bytes);
}

int mldsa_free(void *dev_ptr) {
    if (dev_ptr) MLDSA_HIP_CHECK(free_quiesced(dev_ptr));  // hipFree waits for every stream of the device
    return MLDSA_OK;
}

int mldsa_host_alloc(void **host_ptr, size_t bytes) {
    REQUIRE(host_ptr, "mldsa_host_alloc: NULL out");
    *host_ptr = nullptr;
    if (bytes == 0) return MLDSA_OK;
    hipError_t e = host_malloc_quiesced(host_ptr, bytes, hipHostMallocPortable | hipHostMallocMapped);  // mapped: small batches of the batcher are read in place
    if (e != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "mldsa_host_alloc", e);
    return MLDSA_OK;
}

int mldsa_host_free(void *host_ptr) {
    if (host_ptr) MLDSA_HIP_CHECK(host_free_quiesced(host_ptr));
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
    ctx->prof_sign_op_rounds = 0;
    return MLDSA_OK;
}

int mldsa_profile_report(mldsa_ctx *ctx, char *buf, size_t buf_len) {
    REQUIRE(buf && buf_len > 2, "mldsa_profile_report: bad argument");
    ENTER(ctx, "mldsa_profile_report");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    MLDSA_HIP_CHECK(device_sync_quiesced());
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
        snprintf(tmp, sizeof(tmp), ", \"_sign_op_rounds\": {\"ms\": 0, \"calls\": %llu}", ctx->prof_sign_op_rounds);
        out += tmp;
    }
    out += "}";
    if (out.size() + 1 > buf_len) return set_error(MLDSA_ERR_PARAM, "mldsa_profile_report: buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    ctx->prof_used = 0;
    ctx->prof_sign_slots = 0;
    ctx->prof_sign_op_rounds = 0;
    return MLDSA_OK;
}

// ------------------------------------------------------------------ seam-level primitives
int mldsa_ntt(mldsa_ctx *ctx, const int32_t *w, int32_t *w_hat, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_ntt");
    REQUIRE(n_polys == 0 || (w && w_hat), "mldsa_ntt: NULL pointer");
    return launch_ntt(ctx, w, w_hat, n_polys, (hipStream_t)stream);
}

int mldsa_inv_ntt(mldsa_ctx *ctx, const int32_t *w_hat, int32_t *w, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_inv_ntt");
    REQUIRE(n_polys == 0 || (w && w_hat), "mldsa_inv_ntt: NULL pointer");
    return launch_inv_ntt(ctx, w_hat, w, n_polys, (hipStream_t)stream);
}

int mldsa_to_mont(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_to_mont");
    REQUIRE(n_polys == 0 || (in && out), "mldsa_to_mont: NULL pointer");
    return launch_to_mont(ctx, in, out, n_polys, (hipStream_t)stream);
}

int mldsa_reduce(mldsa_ctx *ctx, int kind, const int32_t *in, int32_t *out, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_reduce");
    REQUIRE(kind == MLDSA_REDUCE_PARTIAL || kind == MLDSA_REDUCE_FULL || kind == MLDSA_REDUCE_CENTER, "mldsa_reduce: unknown kind");
    REQUIRE(n_polys == 0 || (in && out), "mldsa_reduce: NULL pointer");
    return launch_reduce(ctx, kind, in, out, n_polys, (hipStream_t)stream);
}

int mldsa_rounding(mldsa_ctx *ctx, int set, int op, const int32_t *a, const int32_t *b, int32_t *out1, int32_t *out2, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_rounding");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_rounding: unknown parameter set");
    REQUIRE(op >= MLDSA_ROUND_POWER2ROUND && op <= MLDSA_ROUND_USE_HINT, "mldsa_rounding: unknown operation");
    REQUIRE(n_polys == 0 || (a && out1), "mldsa_rounding: NULL pointer");
    REQUIRE(n_polys == 0 || op < MLDSA_ROUND_MAKE_HINT || b, "mldsa_rounding: MakeHint / UseHint take two inputs");
    REQUIRE(n_polys == 0 || op > MLDSA_ROUND_DECOMPOSE || out2, "mldsa_rounding: Power2Round / Decompose have two outputs");
    return launch_rounding(ctx, p, op, a, b, out1, out2, n_polys, (hipStream_t)stream);
}

int mldsa_xof(mldsa_ctx *ctx, int bits, const uint8_t *data, const uint64_t *off, uint8_t *out, size_t out_len, uint8_t *bad, size_t n_ops,
              void *stream) {
    ENTER(ctx, "mldsa_xof");
    REQUIRE(bits == 128 || bits == 256, "mldsa_xof: bits must be 128 (g128_xof) or 256 (h256_xof)");
    REQUIRE(n_ops == 0 || out_len == 0 || (off && out), "mldsa_xof: NULL pointer");
    return launch_xof(ctx, bits, data, off, out, out_len, bad, n_ops, (hipStream_t)stream);
}

static int bit_length_of(int x) {  // helpers.rs bit_length: 32 - leading_zeros
    int n = 0;
    while (x > 0) { n++; x >>= 1; }
    return n;
}

int mldsa_bit_pack(mldsa_ctx *ctx, const int32_t *w, int a, int b, uint8_t *out, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_bit_pack");
    REQUIRE(a >= 0 && a < (1 << 20) && b >= 1 && b < (1 << 20), "mldsa_bit_pack: a in [0, 2^20), b in [1, 2^20) (conversion.rs:144-145)");
    REQUIRE(n_polys == 0 || (w && out), "mldsa_bit_pack: NULL pointer");
    return launch_bit_pack(ctx, w, a, b, bit_length_of(a + b), out, n_polys, (hipStream_t)stream);
}

int mldsa_bit_unpack(mldsa_ctx *ctx, const uint8_t *v, int a, int b, int32_t *w, uint8_t *ok, size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_bit_unpack");
    REQUIRE(a >= 0 && a < (1 << 20) && b >= 1 && b < (1 << 20), "mldsa_bit_unpack: a in [0, 2^20), b in [1, 2^20) (conversion.rs:228-229)");
    REQUIRE(n_polys == 0 || (v && w), "mldsa_bit_unpack: NULL pointer");
    return launch_bit_unpack(ctx, v, a, b, bit_length_of(a + b), w, ok, n_polys, (hipStream_t)stream);
}

int mldsa_hint_bit_pack(mldsa_ctx *ctx, int set, const int32_t *h, uint8_t *y, uint8_t *ok, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_hint_bit_pack");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_hint_bit_pack: unknown parameter set");
    REQUIRE(n_ops == 0 || (h && y), "mldsa_hint_bit_pack: NULL pointer");
    return launch_hint_pack(ctx, p, h, y, ok, n_ops, (hipStream_t)stream);
}

int mldsa_hint_bit_unpack(mldsa_ctx *ctx, int set, const uint8_t *y, int32_t *h, uint8_t *ok, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_hint_bit_unpack");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_hint_bit_unpack: unknown parameter set");
    REQUIRE(n_ops == 0 || (y && h && ok), "mldsa_hint_bit_unpack: NULL pointer");
    return launch_hint_unpack(ctx, p, y, h, ok, n_ops, (hipStream_t)stream);
}

int mldsa_sig_encode(mldsa_ctx *ctx, int set, const uint8_t *c_tilde, const int32_t *z, const int32_t *h, uint8_t *sigs, uint8_t *ok, size_t n_ops,
                     void *stream) {
    ENTER(ctx, "mldsa_sig_encode");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sig_encode: unknown parameter set");
    REQUIRE(n_ops == 0 || (c_tilde && z && h && sigs), "mldsa_sig_encode: NULL pointer");
    return launch_sig_encode(ctx, p, c_tilde, z, h, sigs, ok, n_ops, (hipStream_t)stream);
}

int mldsa_sig_decode(mldsa_ctx *ctx, int set, const uint8_t *sigs, uint8_t *c_tilde, int32_t *z, int32_t *h, uint8_t *ok, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_sig_decode");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sig_decode: unknown parameter set");
    REQUIRE(n_ops == 0 || (sigs && c_tilde && z && h && ok), "mldsa_sig_decode: NULL pointer");
    return launch_sig_decode(ctx, p, sigs, c_tilde, z, h, ok, n_ops, (hipStream_t)stream);
}

int mldsa_w1_encode(mldsa_ctx *ctx, int set, const int32_t *w1, uint8_t *out, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_w1_encode");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_w1_encode: unknown parameter set");
    REQUIRE(n_ops == 0 || (w1 && out), "mldsa_w1_encode: NULL pointer");
    return launch_w1_encode(ctx, p, w1, out, n_ops, (hipStream_t)stream);
}

int mldsa_mat_vec_mul(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *u_hat,
                      int32_t *w_hat, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_mat_vec_mul");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_mat_vec_mul: unknown parameter set");
    REQUIRE(n_ops == 0 || (a_hat && u_hat && w_hat), "mldsa_mat_vec_mul: NULL pointer");
    return launch_mat_vec_mul(ctx, p->k, p->l, a_hat, u_hat, w_hat, n_ops, (hipStream_t)stream);
}

int mldsa_pointwise_mont(mldsa_ctx *ctx, const int32_t *c_hat, const int32_t *v_hat_mont,
                         int32_t *out, size_t polys_per_op, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_pointwise_mont");
    REQUIRE(n_ops * polys_per_op == 0 || (c_hat && v_hat_mont && out), "mldsa_pointwise_mont: NULL pointer");
    return launch_pointwise_mont(ctx, c_hat, v_hat_mont, out, polys_per_op, n_ops, (hipStream_t)stream);
}

int mldsa_add_vector_ntt(mldsa_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out,
                         size_t n_polys, void *stream) {
    ENTER(ctx, "mldsa_add_vector_ntt");
    REQUIRE(n_polys == 0 || (a && b && out), "mldsa_add_vector_ntt: NULL pointer");
    return launch_add(ctx, a, b, out, n_polys, (hipStream_t)stream);
}

int mldsa_infinity_norm(mldsa_ctx *ctx, const int32_t *polys, size_t polys_per_op, size_t n_ops,
                        int32_t *norms, void *stream) {
    ENTER(ctx, "mldsa_infinity_norm");
    REQUIRE(polys_per_op > 0 && (n_ops == 0 || (polys && norms)), "mldsa_infinity_norm: bad argument");
    return launch_infinity_norm(ctx, polys, polys_per_op, n_ops, norms, (hipStream_t)stream);
}

int mldsa_verify_arith(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *z,
                       const int32_t *c, const int32_t *t1_d2_hat_mont, int32_t *w_out,
                       size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_verify_arith");
    REQUIRE(params_of(set), "mldsa_verify_arith: unknown parameter set");
    REQUIRE(n_ops == 0 || (a_hat && z && c && t1_d2_hat_mont && w_out), "mldsa_verify_arith: NULL pointer");
    return launch_verify_arith(ctx, set, a_hat, z, c, t1_d2_hat_mont, nullptr, w_out, n_ops, (hipStream_t)stream);
}

// ------------------------------------------------------------------ samplers
int mldsa_expand_a(mldsa_ctx *ctx, int set, const uint8_t *rho, int32_t *a_hat, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_expand_a");
    REQUIRE(params_of(set), "mldsa_expand_a: unknown parameter set");
    REQUIRE(n_ops == 0 || (rho && a_hat), "mldsa_expand_a: NULL pointer");
    return launch_expand_a(ctx, set, rho, 32, nullptr, a_hat, n_ops, (hipStream_t)stream);
}

int mldsa_expand_s(mldsa_ctx *ctx, int set, const uint8_t *rho_prime, int32_t *s1s2, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_expand_s");
    REQUIRE(params_of(set), "mldsa_expand_s: unknown parameter set");
    REQUIRE(n_ops == 0 || (rho_prime && s1s2), "mldsa_expand_s: NULL pointer");
    return launch_expand_s(ctx, set, rho_prime, 64, s1s2, n_ops, (hipStream_t)stream);
}

int mldsa_expand_mask(mldsa_ctx *ctx, int set, const uint8_t *rho_pp, const uint16_t *kappa, int32_t *y,
                      size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_expand_mask");
    REQUIRE(params_of(set), "mldsa_expand_mask: unknown parameter set");
    REQUIRE(n_ops == 0 || (rho_pp && kappa && y), "mldsa_expand_mask: NULL pointer");
    return launch_expand_mask(ctx, set, rho_pp, 64, kappa, 0, nullptr, y, n_ops, (hipStream_t)stream);
}

int mldsa_sample_in_ball(mldsa_ctx *ctx, int set, const uint8_t *c_tilde, int32_t *c, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_sample_in_ball");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sample_in_ball: unknown parameter set");
    REQUIRE(n_ops == 0 || (c_tilde && c), "mldsa_sample_in_ball: NULL pointer");
    return launch_sample_in_ball(ctx, set, c_tilde, (size_t)p->ctilde_len, c, n_ops, (hipStream_t)stream);
}

// ------------------------------------------------------------------ op-level API
// Shared by mldsa_verify and mldsa_verify_cached_a: reserve, then replay / capture / launch the pipeline.
static int verify_call(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const int32_t *a_hat, const uint8_t *tr,
                       const int32_t *t1, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off,
                       const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops, hipStream_t s,
                       const uint8_t *pk_wire = nullptr) {
    const mldsa_params *p = params_of(set);
    if (n_ops == 0) return MLDSA_OK;
    OpGuard guard(ctx, s);
    if (guard.rc != MLDSA_OK) return guard.rc;  // not ordered behind the previous call's clearing of its secrets
    int rc = reserve_workspace(ctx, p, MLDSA_OP_VERIFY, n_ops, a_hat == nullptr, n_keys, pk_wire ? (key_idx ? 2 : 1) : 0);
    if (rc != MLDSA_OK) return rc;
    struct { int op, set, mode; const void *rho, *a_hat, *tr, *t1, *pk; size_t n_keys; const void *key_idx, *msgs, *msg_off, *ctxs, *ctx_off, *sigs, *ok;
             size_t n_ops; } key;
    memset(&key, 0, sizeof(key));
    key.op = MLDSA_OP_VERIFY; key.set = set; key.mode = mode; key.rho = rho; key.a_hat = a_hat; key.tr = tr; key.t1 = t1; key.pk = pk_wire; key.n_keys = n_keys;
    key.key_idx = key_idx; key.msgs = msgs; key.msg_off = msg_off; key.ctxs = ctxs; key.ctx_off = ctx_off; key.sigs = sigs; key.ok = ok;
    key.n_ops = n_ops;
    return run_op(ctx, s, MLDSA_OP_VERIFY, n_ops, &key, sizeof(key), [&](hipStream_t st) {
        return verify_batch(ctx, set, mode, rho, tr, t1, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops, st, a_hat, pk_wire);
    });
}

int mldsa_verify(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr,
                 const int32_t *t1_d2_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                 const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                 const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_verify");
    REQUIRE(params_of(set), "mldsa_verify: unknown parameter set");
    REQUIRE(mode_ok(mode), "mldsa_verify: bad mode");
    REQUIRE(n_ops == 0 || (rho && tr && t1_d2_hat_mont && msg_off && sigs && ok), "mldsa_verify: NULL pointer");
    REQUIRE(n_ops == 0 || (key_idx ? n_keys > 0 : n_keys >= n_ops), "mldsa_verify: n_keys does not cover the batch");
    return verify_call(ctx, set, mode, rho, nullptr, tr, t1_d2_hat_mont, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops,
                       (hipStream_t)stream);
}

int mldsa_verify_pk(mldsa_ctx *ctx, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                    const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops,
                    void *stream) {
    ENTER(ctx, "mldsa_verify_pk");
    REQUIRE(params_of(set), "mldsa_verify_pk: unknown parameter set");
    REQUIRE(mode_ok(mode), "mldsa_verify_pk: bad mode");
    REQUIRE(n_ops == 0 || (pk && msg_off && sigs && ok), "mldsa_verify_pk: NULL pointer");
    REQUIRE(n_ops == 0 || (key_idx ? n_keys > 0 : n_keys >= n_ops), "mldsa_verify_pk: n_keys does not cover the batch");
    return verify_call(ctx, set, mode, nullptr, nullptr, nullptr, nullptr, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops,
                       (hipStream_t)stream, pk);
}

int mldsa_verify_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *tr,
                          const int32_t *t1_d2_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                          const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                          const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream) {
    ENTER(ctx, "mldsa_verify_cached_a");
    REQUIRE(params_of(set), "mldsa_verify_cached_a: unknown parameter set");
    REQUIRE(mode_ok(mode), "mldsa_verify_cached_a: bad mode");
    REQUIRE(n_ops == 0 || (a_hat && tr && t1_d2_hat_mont && msg_off && sigs && ok), "mldsa_verify_cached_a: NULL pointer");
    REQUIRE(n_ops == 0 || (key_idx ? n_keys > 0 : n_keys >= n_ops), "mldsa_verify_cached_a: n_keys does not cover the batch");
    return verify_call(ctx, set, mode, nullptr, a_hat, tr, t1_d2_hat_mont, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops,
                       (hipStream_t)stream);
}

int mldsa_pk_expand(mldsa_ctx *ctx, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1_d2_hat_mont,
                    size_t n_keys, void *stream) {
    ENTER(ctx, "mldsa_pk_expand");
    REQUIRE(params_of(set), "mldsa_pk_expand: unknown parameter set");
    REQUIRE(n_keys == 0 || (pk && rho && tr && t1_d2_hat_mont), "mldsa_pk_expand: NULL pointer");
    // (no workspace, but the profiling marks of the call live in the context: two threads expanding keys on one context while
    //  mldsa_profile_enable is on raced on them -- found by the host sanitizer leg, tests/cpp/fuzz_host.cpp)
    std::lock_guard<std::mutex> lk(ctx->op_mutex);
    return pk_expand_batch(ctx, set, pk, rho, tr, t1_d2_hat_mont, n_keys, (hipStream_t)stream);
}

int mldsa_sk_expand(mldsa_ctx *ctx, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k, uint8_t *tr,
                    int32_t *s_1_hat_mont, int32_t *s_2_hat_mont, int32_t *t_0_hat_mont, size_t n_keys, void *stream) {
    ENTER(ctx, "mldsa_sk_expand");
    REQUIRE(params_of(set), "mldsa_sk_expand: unknown parameter set");
    REQUIRE(n_keys == 0 || (sk && rho && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont), "mldsa_sk_expand: NULL pointer");
    std::lock_guard<std::mutex> lk(ctx->op_mutex);  // (the profiling marks: see mldsa_pk_expand)
    return sk_expand_batch(ctx, set, sk, rho, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, n_keys, (hipStream_t)stream);
}

int mldsa_pk_into_bytes(mldsa_ctx *ctx, int set, const uint8_t *rho, const int32_t *t1_d2_hat_mont, uint8_t *pk, size_t n_keys,
                        void *stream) {
    ENTER(ctx, "mldsa_pk_into_bytes");
    REQUIRE(params_of(set), "mldsa_pk_into_bytes: unknown parameter set");
    REQUIRE(n_keys == 0 || (rho && t1_d2_hat_mont && pk), "mldsa_pk_into_bytes: NULL pointer");
    return pk_into_bytes_batch(ctx, set, rho, t1_d2_hat_mont, pk, n_keys, (hipStream_t)stream);
}

int mldsa_sk_into_bytes(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
                        const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, const int32_t *t_0_hat_mont, uint8_t *sk,
                        size_t n_keys, void *stream) {
    ENTER(ctx, "mldsa_sk_into_bytes");
    REQUIRE(params_of(set), "mldsa_sk_into_bytes: unknown parameter set");
    REQUIRE(n_keys == 0 || (rho && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont && sk), "mldsa_sk_into_bytes: NULL pointer");
    return sk_into_bytes_batch(ctx, set, rho, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, sk, n_keys, (hipStream_t)stream);
}

int mldsa_get_public_key(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *tr, const int32_t *s_1_hat_mont,
                         const int32_t *s_2_hat_mont, uint8_t *pk_rho, uint8_t *pk_tr, int32_t *pk_t1_d2_hat_mont, size_t n_keys,
                         void *stream) {
    ENTER(ctx, "mldsa_get_public_key");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_get_public_key: unknown parameter set");
    REQUIRE(n_keys == 0 || (rho && tr && s_1_hat_mont && s_2_hat_mont && pk_rho && pk_tr && pk_t1_d2_hat_mont),
            "mldsa_get_public_key: NULL pointer");
    if (n_keys == 0) return MLDSA_OK;
    OpGuard guard(ctx, (hipStream_t)stream);
    if (guard.rc != MLDSA_OK) return guard.rc;
    int rc = reserve_workspace(ctx, p, MLDSA_OP_KEYGEN, n_keys, true);
    if (rc != MLDSA_OK) return rc;
    return get_public_key_batch(ctx, set, rho, tr, s_1_hat_mont, s_2_hat_mont, pk_rho, pk_tr, pk_t1_d2_hat_mont, n_keys, (hipStream_t)stream);
}

int mldsa_keygen(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys, void *stream) {
    ENTER(ctx, "mldsa_keygen");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_keygen: unknown parameter set");
    REQUIRE(n_keys == 0 || (xi && pk && sk), "mldsa_keygen: NULL pointer");
    if (n_keys == 0) return MLDSA_OK;
    hipStream_t s = (hipStream_t)stream;
    OpGuard guard(ctx, s, true);  // keygen_batch orders itself after a pending background clearing
    int rc = reserve_workspace(ctx, p, MLDSA_OP_KEYGEN, n_keys, true);
    if (rc != MLDSA_OK) return rc;
    struct { int op, set; const void *xi, *pk, *sk; size_t n; } key;
    memset(&key, 0, sizeof(key));
    key.op = MLDSA_OP_KEYGEN; key.set = set; key.xi = xi; key.pk = pk; key.sk = sk; key.n = n_keys;
    return run_op(ctx, s, MLDSA_OP_KEYGEN, n_keys, &key, sizeof(key), [&](hipStream_t st) { return keygen_batch(ctx, set, xi, pk, sk, n_keys, st); });
}

}  // extern "C"

int mldsa::sign_call(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const int32_t *a_hat, const uint8_t *cap_k, const uint8_t *tr,
                     const int32_t *s1, const int32_t *s2, const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                     const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs,
                     int32_t *status, size_t n_ops, hipStream_t s, bool async_mode, double plan_stop, uint8_t *export_sigs,
                     hipEvent_t inputs_ev) {
    const mldsa_params *p = params_of(set);
    if (n_ops == 0) return MLDSA_OK;
    OpGuard guard(ctx, s, true);  // sign_batch orders itself after a pending background clearing
    int rc = reserve_workspace(ctx, p, MLDSA_OP_SIGN, n_ops, a_hat == nullptr);
    if (rc != MLDSA_OK) return rc;
    return sign_batch(ctx, set, mode, rho, cap_k, tr, s1, s2, t0, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, rnd, sigs, status,
                      n_ops, s, a_hat, async_mode, plan_stop, export_sigs, inputs_ev);
}

extern "C" {

#define SIGN_CHECKS(name, first)                                                                                                  \
    ENTER(ctx, name);                                                                                                             \
    REQUIRE(params_of(set), name ": unknown parameter set");                                                                      \
    REQUIRE(mode_ok(mode), name ": bad mode");                                                                                    \
    REQUIRE(n_ops == 0 || (first && cap_k && tr && s_1_hat_mont && s_2_hat_mont && t_0_hat_mont && msg_off && rnd && sigs),       \
            name ": NULL pointer");                                                                                               \
    REQUIRE(n_ops == 0 || (key_idx ? n_keys > 0 : n_keys >= n_ops), name ": n_keys does not cover the batch")

int mldsa_sign(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
               const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, const int32_t *t_0_hat_mont, size_t n_keys,
               const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
               const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream) {
    SIGN_CHECKS("mldsa_sign", rho);
    return sign_call(ctx, set, mode, rho, nullptr, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, n_keys, key_idx, msgs, msg_off,
                     ctxs, ctx_off, rnd, sigs, status, n_ops, (hipStream_t)stream, false);
}

int mldsa_sign_async(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
                     const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, const int32_t *t_0_hat_mont, size_t n_keys,
                     const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
                     const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream) {
    SIGN_CHECKS("mldsa_sign_async", rho);
    REQUIRE(n_ops == 0 || status, "mldsa_sign_async: status must not be NULL");
    return sign_call(ctx, set, mode, rho, nullptr, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, n_keys, key_idx, msgs, msg_off,
                     ctxs, ctx_off, rnd, sigs, status, n_ops, (hipStream_t)stream, true);
}

int mldsa_sign_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *cap_k,
                        const uint8_t *tr, const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont,
                        const int32_t *t_0_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                        const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                        const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream) {
    SIGN_CHECKS("mldsa_sign_cached_a", a_hat);
    return sign_call(ctx, set, mode, nullptr, a_hat, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont, n_keys, key_idx, msgs, msg_off,
                     ctxs, ctx_off, rnd, sigs, status, n_ops, (hipStream_t)stream, false);
}

}  // extern "C"
