// Seeded random driver of the library's HOST code through the C ABI, over tests/cpp/stub_hip_full.cpp (no GPU): the counterpart of the
// reference's fuzz targets with debug-assertions + overflow-checks (/root/reference/fuzz/fuzz_targets/fuzz_all.rs:20-51,
// fuzz/Cargo.toml:31-35) for the ~3 400 lines of capi / pipeline / host_api / group .hip that no kernel test instruments.
//
//     fuzz_host <seconds> <seed> [threads]
//
// Every buffer handed to the library has EXACTLY the size the header promises (device buffers from mldsa_malloc = the stub's heap,
// host buffers from malloc or mldsa_host_alloc), so an off-by-one in a staging copy, a pass offset, a workspace carve or a clearing
// span is a sanitizer report.  Actions: device-resident verify / verify_pk / verify_cached_a / sign / sign_async / sign_cached_a /
// keygen, the *_host calls on pageable and page-locked buffers, malformed offset tables, out-of-range key indices, random options
// (graph replay with a small cache: eviction; speculation; lanes; small-call limits), workspace caps, caller-owned workspaces of random
// size, a device that runs out of memory (stub_set_mem_limit), groups of 1 ... 8 contexts (host-fed, device-resident, all-gather),
// key-expansion and seam calls, statistics / profiling / residue probes -- from `threads` threads, each with its own stream, on ONE
// shared context plus private ones (ThreadSanitizer build).  Results are not checked (the stub's kernels compute nothing): return
// codes must be MLDSA_OK or one of the documented refusals.
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../../include/mldsa_hip.h"

extern "C" {
void stub_set_mem_limit(size_t bytes);
void stub_exempt_from_limit(int on);
size_t stub_launches(void);
size_t stub_touches(void);
size_t stub_failed_allocs(void);
size_t stub_live_allocations(void);
int hipStreamCreateWithFlags(void **s, unsigned flags);
int hipStreamDestroy(void *s);
}

namespace {
std::atomic<long> g_calls{0}, g_refused{0};
std::atomic<bool> g_failed{false};

#define FAIL(...) do { std::fprintf(stderr, "fuzz_host: " __VA_ARGS__); std::fprintf(stderr, " (last error: %s)\n", mldsa_last_error()); g_failed = true; return; } while (0)

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    uint64_t u(uint64_t n) { return n ? g() % n : 0; }
    bool p(double q) { return (double)(g() >> 11) / 9007199254740992.0 < q; }
    size_t logu(size_t lo, size_t hi) { const double a = std::log((double)lo), b = std::log((double)hi + 1.0); return std::min(hi, (size_t)std::exp(a + (b - a) * ((double)(g() >> 11) / 9007199254740992.0))); }
};

// exactly-sized buffers; Dev = "device" memory through the library's own allocator, Host = pageable or page-locked host memory
struct Dev {
    void *p = nullptr; size_t n = 0;
    Dev() = default;
    explicit Dev(size_t bytes, mldsa_ctx *ctx = nullptr) : n(bytes) {
        stub_exempt_from_limit(1);  // (the caller's buffers exist; it is the LIBRARY's allocations that meet a full device)
        if ((ctx ? mldsa_ctx_malloc(ctx, &p, bytes ? bytes : 1) : mldsa_malloc(&p, bytes ? bytes : 1)) != MLDSA_OK) p = nullptr;
        stub_exempt_from_limit(0);
    }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    ~Dev() { if (p) mldsa_free(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
struct Host {
    void *p = nullptr; bool pinned = false;
    Host(size_t bytes, bool pin) : pinned(pin) { if (pin) { if (mldsa_host_alloc(&p, bytes ? bytes : 1) != MLDSA_OK) p = nullptr; } else p = std::calloc(1, bytes ? bytes : 1); }
    Host(const Host &) = delete;
    Host &operator=(const Host &) = delete;
    ~Host() { if (!p) return; if (pinned) mldsa_host_free(p); else std::free(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct Batch {  // the host image of one call's arguments
    int set, mode;
    mldsa_params P;
    size_t n, nk;
    bool identity, with_ctx, bad_tables;
    std::vector<uint32_t> key_idx;
    std::vector<uint8_t> msgs, ctxs;
    std::vector<uint64_t> msg_off, ctx_off;
};

Batch make_batch(Rng &r, size_t max_n) {
    Batch b;
    const int sets[3] = {MLDSA_44, MLDSA_65, MLDSA_87};
    b.set = sets[r.u(3)];
    mldsa_get_params(b.set, &b.P);
    b.mode = (int)r.u(3);
    b.n = r.p(0.1) ? 0 : r.logu(1, max_n);
    b.identity = r.p(0.3);
    b.nk = b.identity ? std::max<size_t>(b.n, 1) : 1 + r.u(std::min<size_t>(std::max<size_t>(b.n, 1), 48));
    b.with_ctx = b.mode != MLDSA_MODE_INTERNAL && r.p(0.6);
    b.bad_tables = r.p(0.08);
    if (!b.identity) {
        b.key_idx.resize(b.n);
        for (auto &k : b.key_idx) k = r.p(0.03) ? (uint32_t)(b.nk + r.u(1000)) : (uint32_t)r.u(b.nk);
    }
    const size_t max_len = r.p(0.1) ? 3000 : 120;
    b.msg_off.assign(b.n + 1, 0);
    for (size_t i = 0; i < b.n; i++) b.msg_off[i + 1] = b.msg_off[i] + r.u(max_len + 1);
    b.msgs.assign((size_t)b.msg_off[b.n], 0x4D);
    if (b.with_ctx) {
        b.ctx_off.assign(b.n + 1, 0);
        for (size_t i = 0; i < b.n; i++) b.ctx_off[i + 1] = b.ctx_off[i] + (r.p(0.02) ? 256 + r.u(50) : r.u(40));
        b.ctxs.assign((size_t)b.ctx_off[b.n], 0x43);
    }
    if (b.bad_tables && b.n >= 2) {  // an untrusted table: decreasing, overshooting or wrapping entries
        auto &t = (b.with_ctx && r.p(0.4)) ? b.ctx_off : b.msg_off;
        const size_t i = 1 + r.u(b.n - 1);
        const int kind = (int)r.u(3);
        t[i] = kind == 0 ? (t[i - 1] ? t[i - 1] - 1 : ~0ull) : kind == 1 ? t[b.n] + 1 + r.u(1 << 20) : ~0ull - r.u(100);
    }
    return b;
}

// MLDSA_ERR_NOMEM is a legitimate answer only while something bounds the workspace: the stub's device is short of memory, the context
// has a MLDSA_OPT_WORKSPACE_CAP_MB, or the workspace is a caller-owned buffer.  Anywhere else it means the library mis-sized its own
// workspace (the defensive "workspace not reserved" checks of the pipelines answer with it): a failure.
std::atomic<int> g_low_memory{0};
thread_local int t_own_workspace = 0;
thread_local bool t_cap_before = false;  // the context had a workspace cap when the action began (another thread may lift it meanwhile)
std::atomic<unsigned> g_cap_epoch{0};    // bumped around every change of a workspace cap: a cap that came AND went during an action still counts
thread_local unsigned t_epoch_before = 0;
bool acceptable(int rc, bool may_refuse, mldsa_ctx *ctx = nullptr) {
    g_calls++;
    if (rc == MLDSA_OK) return true;
    const bool bounded = g_low_memory.load() > 0 || t_own_workspace > 0 || t_cap_before || g_cap_epoch.load() != t_epoch_before || (ctx && mldsa_get_option(ctx, MLDSA_OPT_WORKSPACE_CAP_MB) != 0);
    if ((rc == MLDSA_ERR_NOMEM && (bounded || !ctx)) || (may_refuse && rc == MLDSA_ERR_PARAM)) { g_refused++; return true; }
    return false;
}

struct Up {  // a Batch on the "device"
    Dev key_idx, msgs, msg_off, ctxs, ctx_off;
    Up(const Batch &b, mldsa_ctx *ctx = nullptr)
        : key_idx(b.key_idx.size() * 4, ctx), msgs(b.msgs.size(), ctx), msg_off(b.msg_off.size() * 8, ctx), ctxs(b.ctxs.size(), ctx), ctx_off(b.ctx_off.size() * 8, ctx) {
        if (!b.key_idx.empty()) mldsa_memcpy_h2d(key_idx.p, b.key_idx.data(), b.key_idx.size() * 4, nullptr);
        if (!b.msgs.empty()) mldsa_memcpy_h2d(msgs.p, b.msgs.data(), b.msgs.size(), nullptr);
        mldsa_memcpy_h2d(msg_off.p, b.msg_off.data(), b.msg_off.size() * 8, nullptr);
        if (!b.ctxs.empty()) mldsa_memcpy_h2d(ctxs.p, b.ctxs.data(), b.ctxs.size(), nullptr);
        if (!b.ctx_off.empty()) mldsa_memcpy_h2d(ctx_off.p, b.ctx_off.data(), b.ctx_off.size() * 8, nullptr);
    }
};

void act_verify(mldsa_ctx *ctx, Rng &r, void *stream, size_t max_n) {
    const Batch b = make_batch(r, max_n);
    const size_t K = (size_t)b.P.k, L = (size_t)b.P.l;
    Up u(b);
    Dev sigs(b.n * (size_t)b.P.sig_len), ok(b.n);
    const uint32_t *ki = b.identity ? nullptr : u.key_idx.as<uint32_t>();
    const uint64_t *co = b.with_ctx ? u.ctx_off.as<uint64_t>() : nullptr;
    const uint8_t *cx = b.with_ctx ? u.ctxs.as<uint8_t>() : nullptr;
    int rc;
    const int form = (int)r.u(3);
    if (form == 0) {
        Dev rho(b.nk * 32), tr(b.nk * 64), t1(b.nk * K * 1024);
        rc = mldsa_verify(ctx, b.set, b.mode, rho.as<uint8_t>(), tr.as<uint8_t>(), t1.as<int32_t>(), b.nk, ki, u.msgs.as<uint8_t>(), u.msg_off.as<uint64_t>(), cx, co,
                          sigs.as<uint8_t>(), ok.as<uint8_t>(), b.n, stream);
    } else if (form == 1) {
        Dev pk(b.nk * (size_t)b.P.pk_len);
        rc = mldsa_verify_pk(ctx, b.set, b.mode, pk.as<uint8_t>(), b.nk, ki, u.msgs.as<uint8_t>(), u.msg_off.as<uint64_t>(), cx, co, sigs.as<uint8_t>(), ok.as<uint8_t>(), b.n, stream);
    } else {
        Dev a(b.nk * K * L * 1024), tr(b.nk * 64), t1(b.nk * K * 1024);
        rc = mldsa_verify_cached_a(ctx, b.set, b.mode, a.as<int32_t>(), tr.as<uint8_t>(), t1.as<int32_t>(), b.nk, ki, u.msgs.as<uint8_t>(), u.msg_off.as<uint64_t>(), cx, co,
                                   sigs.as<uint8_t>(), ok.as<uint8_t>(), b.n, stream);
    }
    mldsa_stream_sync(stream);
    if (!acceptable(rc, false, ctx)) FAIL("verify form %d set %d n %zu keys %zu: rc %d", form, b.set, b.n, b.nk, rc);
}

void act_sign(mldsa_ctx *ctx, Rng &r, void *stream, size_t max_n) {
    const Batch b = make_batch(r, max_n);
    const size_t K = (size_t)b.P.k, L = (size_t)b.P.l;
    Up u(b);
    Dev cap_k(b.nk * 32), tr(b.nk * 64), s1(b.nk * L * 1024), s2(b.nk * K * 1024), t0(b.nk * K * 1024), rnd(b.n * 32), sigs(b.n * (size_t)b.P.sig_len), status(b.n * 4);
    const uint32_t *ki = b.identity ? nullptr : u.key_idx.as<uint32_t>();
    const uint64_t *co = b.with_ctx ? u.ctx_off.as<uint64_t>() : nullptr;
    const uint8_t *cx = b.with_ctx ? u.ctxs.as<uint8_t>() : nullptr;
    const int form = (int)r.u(3);
    int32_t *st = (form == 1 || r.p(0.5)) ? status.as<int32_t>() : nullptr;
    int rc;
    if (form == 2) {
        Dev a(b.nk * K * L * 1024);
        rc = mldsa_sign_cached_a(ctx, b.set, b.mode, a.as<int32_t>(), cap_k.as<uint8_t>(), tr.as<uint8_t>(), s1.as<int32_t>(), s2.as<int32_t>(), t0.as<int32_t>(), b.nk, ki,
                                 u.msgs.as<uint8_t>(), u.msg_off.as<uint64_t>(), cx, co, rnd.as<uint8_t>(), sigs.as<uint8_t>(), st, b.n, stream);
    } else {
        Dev rho(b.nk * 32);
        auto fn = form == 1 ? mldsa_sign_async : mldsa_sign;
        rc = fn(ctx, b.set, b.mode, rho.as<uint8_t>(), cap_k.as<uint8_t>(), tr.as<uint8_t>(), s1.as<int32_t>(), s2.as<int32_t>(), t0.as<int32_t>(), b.nk, ki, u.msgs.as<uint8_t>(),
                u.msg_off.as<uint64_t>(), cx, co, rnd.as<uint8_t>(), sigs.as<uint8_t>(), st, b.n, stream);
    }
    mldsa_stream_sync(stream);
    if (!acceptable(rc, false, ctx)) FAIL("sign form %d set %d n %zu keys %zu: rc %d", form, b.set, b.n, b.nk, rc);
}

void act_keygen_and_keys(mldsa_ctx *ctx, Rng &r, void *stream, size_t max_n) {
    const int sets[3] = {MLDSA_44, MLDSA_65, MLDSA_87};
    const int set = sets[r.u(3)];
    mldsa_params P;
    mldsa_get_params(set, &P);
    const size_t n = r.p(0.1) ? 0 : r.logu(1, max_n), K = (size_t)P.k, L = (size_t)P.l;
    Dev xi(n * 32), pk(n * (size_t)P.pk_len), sk(n * (size_t)P.sk_len);
    int rc = mldsa_keygen(ctx, set, xi.as<uint8_t>(), pk.as<uint8_t>(), sk.as<uint8_t>(), n, stream);
    if (!acceptable(rc, false, ctx)) FAIL("keygen set %d n %zu: rc %d", set, n, rc);
    if (n && r.p(0.5)) {  // SerDes round trip on the same buffers: try_from_bytes, get_public_key, into_bytes
        Dev rho(n * 32), cap_k(n * 32), tr(n * 64), s1(n * L * 1024), s2(n * K * 1024), t0(n * K * 1024), t1(n * K * 1024);
        rc = mldsa_sk_expand(ctx, set, sk.as<uint8_t>(), rho.as<uint8_t>(), cap_k.as<uint8_t>(), tr.as<uint8_t>(), s1.as<int32_t>(), s2.as<int32_t>(), t0.as<int32_t>(), n, stream);
        if (!acceptable(rc, false)) FAIL("sk_expand: rc %d", rc);
        rc = mldsa_pk_expand(ctx, set, pk.as<uint8_t>(), rho.as<uint8_t>(), tr.as<uint8_t>(), t1.as<int32_t>(), n, stream);
        if (!acceptable(rc, false)) FAIL("pk_expand: rc %d", rc);
        {
            Dev rho2(n * 32), tr2(n * 64), t12(n * K * 1024);
            rc = mldsa_get_public_key(ctx, set, rho.as<uint8_t>(), tr.as<uint8_t>(), s1.as<int32_t>(), s2.as<int32_t>(), rho2.as<uint8_t>(), tr2.as<uint8_t>(), t12.as<int32_t>(), n, stream);
            if (!acceptable(rc, false)) FAIL("get_public_key: rc %d", rc);
            mldsa_stream_sync(stream);
        }
        rc = mldsa_pk_into_bytes(ctx, set, rho.as<uint8_t>(), t1.as<int32_t>(), pk.as<uint8_t>(), n, stream);
        if (!acceptable(rc, false)) FAIL("pk_into_bytes: rc %d", rc);
        rc = mldsa_sk_into_bytes(ctx, set, rho.as<uint8_t>(), cap_k.as<uint8_t>(), tr.as<uint8_t>(), s1.as<int32_t>(), s2.as<int32_t>(), t0.as<int32_t>(), sk.as<uint8_t>(), n, stream);
        if (!acceptable(rc, false)) FAIL("sk_into_bytes: rc %d", rc);
    }
    mldsa_stream_sync(stream);
}

void act_host(mldsa_ctx *ctx, mldsa_group *grp, Rng &r, size_t max_n) {
    const Batch b = make_batch(r, max_n);
    const bool pin = r.p(0.5);
    const uint32_t *ki = b.identity ? nullptr : b.key_idx.data();
    const uint64_t *co = b.with_ctx ? b.ctx_off.data() : nullptr;
    const uint8_t *cx = b.with_ctx ? b.ctxs.data() : nullptr;
    const int what = (int)r.u(3);
    int rc;
    if (what == 0) {
        Host pk(b.nk * (size_t)b.P.pk_len, pin), sigs(b.n * (size_t)b.P.sig_len, pin), ok(b.n, pin);
        rc = grp ? mldsa_verify_host_group(grp, b.set, b.mode, pk.as<uint8_t>(), b.nk, ki, b.msgs.data(), b.msg_off.data(), cx, co, sigs.as<uint8_t>(), ok.as<uint8_t>(), b.n)
                 : mldsa_verify_host(ctx, b.set, b.mode, pk.as<uint8_t>(), b.nk, ki, b.msgs.data(), b.msg_off.data(), cx, co, sigs.as<uint8_t>(), ok.as<uint8_t>(), b.n);
    } else if (what == 1) {
        Host sk(b.nk * (size_t)b.P.sk_len, pin), rnd(b.n * 32, pin), sigs(b.n * (size_t)b.P.sig_len, pin), status(b.n * 4, pin);
        rc = grp ? mldsa_sign_host_group(grp, b.set, b.mode, sk.as<uint8_t>(), b.nk, ki, b.msgs.data(), b.msg_off.data(), cx, co, rnd.as<uint8_t>(), sigs.as<uint8_t>(), status.as<int32_t>(), b.n)
                 : mldsa_sign_host(ctx, b.set, b.mode, sk.as<uint8_t>(), b.nk, ki, b.msgs.data(), b.msg_off.data(), cx, co, rnd.as<uint8_t>(), sigs.as<uint8_t>(), status.as<int32_t>(), b.n);
    } else {
        Host xi(b.n * 32, pin), pk(b.n * (size_t)b.P.pk_len, pin), sk(b.n * (size_t)b.P.sk_len, pin);
        rc = grp ? mldsa_keygen_host_group(grp, b.set, xi.as<uint8_t>(), pk.as<uint8_t>(), sk.as<uint8_t>(), b.n)
                 : mldsa_keygen_host(ctx, b.set, xi.as<uint8_t>(), pk.as<uint8_t>(), sk.as<uint8_t>(), b.n);
    }
    if (!acceptable(rc, /* malformed tables and out-of-range key tables fail a *_host call as a whole */ true, grp ? mldsa_group_ctx(grp, 0) : ctx))
        FAIL("host call %d (group %d) set %d n %zu keys %zu pinned %d: rc %d", what, grp ? mldsa_group_size(grp) : 0, b.set, b.n, b.nk, (int)pin, rc);
}

void act_options(mldsa_ctx *ctx, Rng &r) {
    struct { int opt; long lo, hi; } o[] = {{MLDSA_OPT_GRAPHS, 0, 2}, {MLDSA_OPT_SPEC_TARGET, 1, 200000}, {MLDSA_OPT_SPEC_MAX, 1, 64}, {MLDSA_OPT_VA_BLOCKS_PER_CU, 1, 64},
                                           {MLDSA_OPT_GRAPH_CACHE, 1, 4}, {MLDSA_OPT_SIGN_ROUNDS, 0, 6}, {MLDSA_OPT_SIGN_LANES, 0, 2}, {MLDSA_OPT_SIGN_CT0_EXACT, 0, 1},
                                           {MLDSA_OPT_SIGN_ASYNC_EXP, 1, 12}, {MLDSA_OPT_SIGN_LOOKAHEAD, 0, 2}, {MLDSA_OPT_COOP_HASH, 0, 1}, {MLDSA_OPT_SMALL_FUSED, 0, 1024}};
    const auto &c = o[r.u(sizeof(o) / sizeof(o[0]))];
    const long v = r.p(0.1) ? c.hi + 1 + (long)r.u(10) : c.lo + (long)r.u((uint64_t)(c.hi - c.lo + 1));
    const int rc = mldsa_set_option(ctx, c.opt, v);
    g_calls++;
    if (rc != MLDSA_OK && rc != MLDSA_ERR_PARAM) FAIL("set_option %d = %ld: rc %d", c.opt, v, rc);
    if (rc == MLDSA_OK && mldsa_get_option(ctx, c.opt) != v && c.opt != MLDSA_OPT_SIGN_ASYNC_EXP) FAIL("option %d does not read back", c.opt);
    if (r.p(0.3)) {
        const long caps[] = {0, 0, 1, 8, 64, 1024};
        g_cap_epoch++;
        const int rc_cap = mldsa_set_option(ctx, MLDSA_OPT_WORKSPACE_CAP_MB, caps[r.u(6)]);
        g_cap_epoch++;
        if (rc_cap != MLDSA_OK) FAIL("workspace cap");
    }
}

void act_misc(mldsa_ctx *ctx, Rng &r, void *stream) {
    mldsa_stats st;
    if (mldsa_get_stats(ctx, &st) != MLDSA_OK) FAIL("get_stats");
    unsigned char small[20];
    if (mldsa_get_stats_sized(ctx, small, sizeof(small)) != MLDSA_OK) FAIL("get_stats_sized");
    size_t a = 0, b = 0;
    if (mldsa_debug_secret_residue(ctx, &a, &b) != MLDSA_OK) FAIL("secret_residue");
    if (r.p(0.3)) {
        char buf[4096];
        mldsa_profile_enable(ctx, 1);
        act_verify(ctx, r, stream, 300);
        if (mldsa_profile_report(ctx, buf, r.p(0.3) ? 16 : sizeof(buf)) != MLDSA_OK) { /* a short buffer is refused, never overrun */ }
        mldsa_profile_enable(ctx, 0);
    }
    const int sets[3] = {MLDSA_44, MLDSA_65, MLDSA_87};
    const int ops[3] = {MLDSA_OP_KEYGEN, MLDSA_OP_SIGN, MLDSA_OP_VERIFY};
    const int rc = mldsa_reserve(ctx, sets[r.u(3)], ops[r.u(3)], r.logu(1, 20000));
    if (!acceptable(rc, false)) FAIL("reserve: rc %d", rc);
    // a few seams: launchers with exactly-sized polynomial buffers
    const size_t n = r.logu(1, 500);
    Dev x(n * 1024), y(n * 1024);
    if (mldsa_ntt(ctx, x.as<int32_t>(), y.as<int32_t>(), n, stream) != MLDSA_OK || mldsa_inv_ntt(ctx, y.as<int32_t>(), x.as<int32_t>(), n, stream) != MLDSA_OK ||
        mldsa_reduce(ctx, MLDSA_REDUCE_FULL, x.as<int32_t>(), y.as<int32_t>(), n, stream) != MLDSA_OK)
        FAIL("seam call");
    mldsa_stream_sync(stream);
}

void act_workspace(mldsa_ctx *ctx, Rng &r, void *stream) {  // a caller-owned workspace of random size for a few calls, then back
    const size_t bytes = r.logu(1 << 16, 96 << 20);
    Dev buf(bytes);
    if (!buf.p) return;
    int rc = mldsa_ctx_set_workspace(ctx, buf.p, bytes);
    if (rc != MLDSA_OK) FAIL("set_workspace: rc %d", rc);
    t_own_workspace++;
    for (int i = 0; i < 3 && !g_failed; i++) {
        switch (r.u(3)) { case 0: act_verify(ctx, r, stream, 3000); break; case 1: act_sign(ctx, r, stream, 1500); break; default: act_keygen_and_keys(ctx, r, stream, 1500); }
    }
    t_own_workspace--;
    rc = mldsa_ctx_set_workspace(ctx, nullptr, 0);
    if (rc != MLDSA_OK) FAIL("reset workspace: rc %d", rc);
}

void act_group(Rng &r) {
    const int n = 1 + (int)r.u(8);
    std::vector<int> devs((size_t)n);
    const bool distinct = r.p(0.5);
    for (int i = 0; i < n; i++) devs[(size_t)i] = distinct ? i : (int)r.u(8);
    mldsa_group *g = nullptr;
    if (mldsa_group_create(devs.data(), n, &g) != MLDSA_OK || !g) FAIL("group_create(%d)", n);
    for (int it = 0; it < 3 && !g_failed; it++) act_host(nullptr, g, r, 2500);
    if (!g_failed && r.p(0.7)) {  // device-resident slices + the verdict gather
        const int set = MLDSA_65;
        mldsa_params P;
        mldsa_get_params(set, &P);
        const size_t total = r.logu(1, 4000), per = (total + (size_t)n - 1) / (size_t)n, K = (size_t)P.k;
        std::vector<std::unique_ptr<Dev>> keep;
        auto mk = [&](mldsa_ctx *c, size_t bytes) { keep.emplace_back(new Dev(bytes, c)); return keep.back()->p; };
        std::vector<mldsa_verify_slice> sl((size_t)n);
        std::vector<uint8_t *> bufs((size_t)n);
        for (int i = 0; i < n; i++) {
            size_t first = 0, cnt = 0;
            mldsa_group_shard(total, n, i, &first, &cnt);
            mldsa_ctx *c = mldsa_group_ctx(g, i);
            mldsa_verify_slice &s = sl[(size_t)i];
            std::memset(&s, 0, sizeof(s));
            const size_t nk = std::max<size_t>(cnt, 1);
            s.rho = (const uint8_t *)mk(c, nk * 32); s.tr = (const uint8_t *)mk(c, nk * 64); s.t1_d2_hat_mont = (const int32_t *)mk(c, nk * K * 1024); s.n_keys = nk;
            s.msgs = (const uint8_t *)mk(c, 16); s.msg_off = (const uint64_t *)mk(c, (cnt + 1) * 8); s.sigs = (const uint8_t *)mk(c, cnt * (size_t)P.sig_len);
            bufs[(size_t)i] = (uint8_t *)mk(c, (size_t)n * per);
            s.ok = bufs[(size_t)i] + (size_t)i * per; s.n_ops = cnt; s.stream = nullptr;
        }
        const int wait = (int)r.u(2);
        int rc = mldsa_verify_group(g, set, MLDSA_MODE_PURE, sl.data(), wait);
        if (!acceptable(rc, false)) FAIL("verify_group: rc %d", rc);
        if (!wait && mldsa_group_sync(g) != MLDSA_OK) FAIL("group_sync");
        rc = mldsa_group_allgather(g, bufs.data(), total, 0);
        if (!acceptable(rc, false)) FAIL("allgather: rc %d", rc);
    }
    mldsa_group_destroy(g);
}

void worker(mldsa_ctx *shared, int id, uint64_t seed, double seconds) {
    Rng r(seed * 1000003ull + (uint64_t)id);
    void *stream = nullptr, *stream2 = nullptr;
    hipStreamCreateWithFlags(&stream, 1);
    hipStreamCreateWithFlags(&stream2, 1);
    mldsa_ctx *mine = nullptr;
    if (mldsa_ctx_create((int)r.u(8), &mine) != MLDSA_OK) { std::fprintf(stderr, "fuzz_host: ctx_create failed: %s\n", mldsa_last_error()); g_failed = true; return; }
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    while (std::chrono::steady_clock::now() < t_end && !g_failed) {
        mldsa_ctx *ctx = r.p(0.6) ? shared : mine;       // the shared context: calls of several threads interleave on it
        void *s = r.p(0.5) ? stream : stream2;
        t_epoch_before = g_cap_epoch.load();
        t_cap_before = mldsa_get_option(ctx, MLDSA_OPT_WORKSPACE_CAP_MB) != 0 || mldsa_get_option(mine, MLDSA_OPT_WORKSPACE_CAP_MB) != 0;
        const int a = (int)r.u(100);
        if (a < 22) act_verify(ctx, r, s, 4000);
        else if (a < 44) act_sign(ctx, r, s, 2500);
        else if (a < 56) act_keygen_and_keys(ctx, r, s, 2500);
        else if (a < 70) act_host(ctx, nullptr, r, 3000);
        else if (a < 80) act_options(ctx == shared && id != 0 ? mine : ctx, r);   // (one thread re-tunes the shared context, the others their own)
        else if (a < 87) act_misc(ctx, r, s);
        else if (a < 91 && ctx == mine) act_workspace(ctx, r, s);
        else if (a < 94 && id == 0) {  // the device runs short of memory for a while: the passes shrink, calls may be refused with MLDSA_ERR_NOMEM
            g_low_memory++;
            g_cap_epoch++;  // (a shortage that came AND went while another thread's call ran still explains that call's NOMEM)
            stub_set_mem_limit((size_t)r.logu(1 << 20, 512 << 20));
            for (int i = 0; i < 4 && !g_failed; i++) { if (r.p(0.5)) act_verify(mine, r, s, 4000); else act_sign(mine, r, s, 2500); }
            stub_set_mem_limit((size_t)1 << 40);
            g_cap_epoch++;
            g_low_memory--;
        } else if (a < 97) act_group(r);
        else if (ctx == mine) {  // a context comes and goes
            mldsa_ctx_destroy(mine);
            mine = nullptr;
            if (mldsa_ctx_create((int)r.u(8), &mine) != MLDSA_OK) { std::fprintf(stderr, "fuzz_host: ctx_create failed: %s\n", mldsa_last_error()); g_failed = true; break; }
        }
    }
    if (mine) mldsa_ctx_destroy(mine);
    hipStreamDestroy(stream);
    hipStreamDestroy(stream2);
}
}  // namespace

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 10.0;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    const int threads = argc > 3 ? std::atoi(argv[3]) : 2;
    if (mldsa_abi_version() != MLDSA_ABI_VERSION) { std::fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    const size_t live0 = stub_live_allocations();
    mldsa_ctx *shared = nullptr;
    if (mldsa_ctx_create(0, &shared) != MLDSA_OK) { std::fprintf(stderr, "fuzz_host: ctx_create failed: %s\n", mldsa_last_error()); return 1; }
    std::vector<std::thread> th;
    for (int i = 0; i < threads; i++) th.emplace_back(worker, shared, i, seed, seconds);
    for (auto &t : th) t.join();
    mldsa_ctx_destroy(shared);
    const size_t live1 = stub_live_allocations();
    std::printf("fuzz_host: seed %" PRIu64 " threads %d: %ld calls (%ld refused with NOMEM / PARAM), %zu kernel launches, %zu touch-modelled, %zu allocations refused by the stub, "
                "%zu allocations live before / %zu after\n", seed, threads, g_calls.load(), g_refused.load(), stub_launches(), stub_touches(), stub_failed_allocs(), live0, live1);
    if (g_failed) { std::printf("FAILED\n"); return 1; }
    if (live1 != live0) { std::printf("FAILED: device / page-locked allocations leaked\n"); return 1; }
    std::printf("OK\n");
    return 0;
}
