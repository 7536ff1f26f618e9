// In-library batch split over the GPUs of one node (SURVEY.md 8b(1) "mldsa_ctx_create(device_ids[], n)", 8e "one host thread +
// stream(s) per device"): a group owns one context and one worker thread per entry of device_ids; the *_host_group entry
// points cut a batch into contiguous slices of ceil(B / N) ops -- the same split as fips204_amd/multi_gpu.py shard() -- hand
// slice i to worker i, and every worker runs the ordinary host-memory entry point (host_api.hip) of its own context on its
// slice of the caller's arrays.  The operations are independent (src/traits.rs:118-308, 330-362 are pure functions of their
// arguments), results land directly in the caller's host buffers, so the host-output case needs no collective at all.
// For verdicts that stay on the devices mldsa_group_allgather offers the one exchange SURVEY 8e names: ncclAllGather of the
// per-op verdict bytes over xGMI (librccl.so, loaded on first use; plain device-to-device copies when a device appears in the
// group more than once, which RCCL refuses).
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>

#include "ctx.h"

struct mldsa_group {
    struct Worker {
        mldsa_ctx *ctx = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;  // set by the caller thread, cleared by the worker
        bool has_job = false, done = false, quit = false;
        int rc = MLDSA_OK;
        std::string err;
    };
    std::vector<std::unique_ptr<Worker>> w;
    std::mutex call_mu;  // one group call at a time
    // RCCL (optional, mldsa_group_allgather)
    void *rccl = nullptr;
    std::string rccl_how, rccl_path;  // "reused" (the copy the process had mapped already) / "loaded", and the file that was bound
    int rccl_version = 0;             // ncclGetVersion of that file
    std::vector<void *> comms;
    bool rccl_tried = false;
    std::vector<hipStream_t> gather_streams;
    std::vector<hipStream_t> last_streams;  // stream of slice i in the last device-resident group call (mldsa_group_sync)
    std::vector<char> last_used;
};

namespace mldsa {
namespace {

void worker_loop(mldsa_group::Worker *w) {
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const int rc = job();
        const char *e = mldsa_last_error();  // thread-local: this worker's message
        lk.lock();
        w->rc = rc;
        w->err = rc == MLDSA_OK ? "" : (e ? e : "");
        w->done = true;
        lk.unlock();
        w->cv.notify_all();
    }
}

// run job(i) on worker i for every i with a non-empty slice, wait for all, first error wins
int run_on_all(mldsa_group *g, const std::function<int(int)> &job) {
    const int n = (int)g->w.size();
    for (int i = 0; i < n; i++) {
        auto &w = *g->w[i];
        std::lock_guard<std::mutex> lk(w.mu);
        w.job = [i, &job] { return job(i); };
        w.has_job = true;
        w.done = false;
        w.cv.notify_all();
    }
    int rc = MLDSA_OK;
    std::string err;
    for (int i = 0; i < n; i++) {
        auto &w = *g->w[i];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        if (w.rc != MLDSA_OK && rc == MLDSA_OK) {
            rc = w.rc;
            err = "device slice " + std::to_string(i) + ": " + w.err;
        }
    }
    if (rc != MLDSA_OK) return set_error(rc, err.c_str());
    return MLDSA_OK;
}

inline void slice(size_t n_ops, int n_parts, int part, size_t &first, size_t &count) {
    const size_t per = (n_ops + (size_t)n_parts - 1) / (size_t)n_parts;
    first = std::min(n_ops, (size_t)part * per);
    count = std::min(n_ops, first + per) - first;
}

}  // namespace
}  // namespace mldsa

namespace mldsa {
namespace {
template <class Slice, class Call>
int run_slices(mldsa_group *g, const Slice *slices, int wait, const Call &call) {
    const int n = (int)g->w.size();
    g->last_streams.assign((size_t)n, nullptr);
    g->last_used.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        g->last_streams[(size_t)i] = (hipStream_t)slices[i].stream;
        g->last_used[(size_t)i] = 1;
    }
    return run_on_all(g, [&](int i) -> int {
        const int rc = call(g->w[i]->ctx, slices[i]);
        if (rc != MLDSA_OK || !wait) return rc;
        DeviceGuard dg(g->w[i]->ctx->device);
        MLDSA_HIP_CHECK(hipStreamSynchronize((hipStream_t)slices[i].stream));
        return MLDSA_OK;
    });
}
}  // namespace
}  // namespace mldsa

using namespace mldsa;

#define REQUIRE(cond, msg) \
    do { if (!(cond)) return set_error(MLDSA_ERR_PARAM, msg); } while (0)

// RCCL is loaded, not linked (hosts that never gather pay nothing for it).  ORDER MATTERS in a process that has an RCCL already: a
// Python host has mapped torch's own copy (torch/lib/librccl.so, SONAME librccl.so.1, built against torch's HIP runtime), and a second
// RCCL from /opt/rocm/lib beside it would put two collective runtimes on one HIP runtime.  So: (1) whatever the process has mapped
// under the SONAME (RTLD_NOLOAD: never loads anything), (2) the same question for the unversioned name, (3) the usual search --
// LD_LIBRARY_PATH, this library's rpath (/opt/rocm/lib) --, (4) the ROCm path spelled out.  Reports how and which file was bound.
static void *load_rccl(std::string *how, std::string *path, int *version) {
    static const struct { const char *name; int flags; const char *how; } order[] = {
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, "reused"},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, "reused"},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL, "loaded"},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL, "loaded"},
        {"/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL, "loaded"},
    };
    for (const auto &o : order) {
        void *h = dlopen(o.name, o.flags);
        if (!h) continue;
        void *sym = dlsym(h, "ncclAllGather");
        if (!sym) { dlclose(h); continue; }
        Dl_info info;
        if (how) *how = o.how;
        if (path) *path = (dladdr(sym, &info) && info.dli_fname) ? info.dli_fname : o.name;
        if (version) {
            typedef int (*ver_t)(int *);
            ver_t get = (ver_t)dlsym(h, "ncclGetVersion");
            int v = 0;
            *version = (get && get(&v) == 0) ? v : 0;
        }
        return h;
    }
    return nullptr;
}

extern "C" {

int mldsa_group_shard(size_t n_ops, int n_parts, int part, size_t *first, size_t *count) {
    REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts && first && count, "mldsa_group_shard: part outside the group");
    slice(n_ops, n_parts, part, *first, *count);
    return MLDSA_OK;
}

int mldsa_group_create(const int *device_ids, int n, mldsa_group **out) {
    REQUIRE(out, "mldsa_group_create: NULL out");
    *out = nullptr;
    REQUIRE(device_ids && n >= 1 && n <= 64, "mldsa_group_create: 1 ... 64 devices");
    std::unique_ptr<mldsa_group> g(new (std::nothrow) mldsa_group());
    if (!g) return set_error(MLDSA_ERR_NOMEM, "mldsa_group_create: host allocation failed");
    for (int i = 0; i < n; i++) {
        std::unique_ptr<mldsa_group::Worker> w(new (std::nothrow) mldsa_group::Worker());
        if (!w) { mldsa_group_destroy(g.release()); return set_error(MLDSA_ERR_NOMEM, "mldsa_group_create: host allocation failed"); }
        const int rc = mldsa_ctx_create(device_ids[i], &w->ctx);
        if (rc != MLDSA_OK) {
            mldsa_group_destroy(g.release());
            return rc;  // message of mldsa_ctx_create
        }
        g->w.push_back(std::move(w));
    }
    for (auto &w : g->w) w->th = std::thread(worker_loop, w.get());
    *out = g.release();
    return MLDSA_OK;
}

void mldsa_group_destroy(mldsa_group *g) {
    if (!g) return;
    for (auto &w : g->w) {
        if (w->th.joinable()) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            w->th.join();
        }
    }
    if (g->rccl) {
        typedef int (*destroy_t)(void *);
        destroy_t destroy = (destroy_t)dlsym(g->rccl, "ncclCommDestroy");
        for (void *c : g->comms)
            if (c && destroy) (void)destroy(c);
    }
    for (size_t i = 0; i < g->gather_streams.size(); i++) {
        if (!g->gather_streams[i]) continue;
        DeviceGuard dg(g->w[i]->ctx->device);
        (void)hipStreamDestroy(g->gather_streams[i]);
    }
    for (auto &w : g->w)
        if (w->ctx) mldsa_ctx_destroy(w->ctx);
    delete g;
}

int mldsa_group_size(const mldsa_group *g) { return g ? (int)g->w.size() : MLDSA_ERR_PARAM; }

mldsa_ctx *mldsa_group_ctx(mldsa_group *g, int i) { return (g && i >= 0 && i < (int)g->w.size()) ? g->w[i]->ctx : nullptr; }

int mldsa_verify_host_group(mldsa_group *g, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx,
                            const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                            const uint8_t *sigs, uint8_t *ok, size_t n_ops) {
    REQUIRE(g, "mldsa_verify_host_group: NULL group");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_verify_host_group: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    REQUIRE(pk && msg_off && sigs && ok, "mldsa_verify_host_group: NULL pointer");
    REQUIRE(key_idx ? n_keys > 0 : n_keys >= n_ops, "mldsa_verify_host_group: n_keys does not cover the batch");
    { const int rc = check_tables("mldsa_verify_host_group", msgs, msg_off, ctxs, ctx_off, n_ops); if (rc != MLDSA_OK) return rc; }
    std::lock_guard<std::mutex> lk(g->call_mu);
    const int n = (int)g->w.size();
    return run_on_all(g, [&](int i) -> int {
        size_t a, cnt;
        slice(n_ops, n, i, a, cnt);
        if (cnt == 0) return MLDSA_OK;
        // an explicit key table is shared by every slice; the identity mapping (op i uses key i) walks with the slice.
        // msgs / ctxs stay whole: the slice's offsets index into them.
        const size_t kb = key_idx ? 0 : a;
        return mldsa_verify_host(g->w[i]->ctx, set, mode, pk + kb * (size_t)p->pk_len, n_keys - kb, key_idx ? key_idx + a : nullptr, msgs,
                                 msg_off + a, ctxs, ctx_off ? ctx_off + a : nullptr, sigs + a * (size_t)p->sig_len, ok + a, cnt);
    });
}

int mldsa_sign_host_group(mldsa_group *g, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx,
                          const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                          const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops) {
    REQUIRE(g, "mldsa_sign_host_group: NULL group");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sign_host_group: unknown parameter set");
    if (n_ops == 0) return MLDSA_OK;
    REQUIRE(sk && msg_off && rnd && sigs, "mldsa_sign_host_group: NULL pointer");
    REQUIRE(key_idx ? n_keys > 0 : n_keys >= n_ops, "mldsa_sign_host_group: n_keys does not cover the batch");
    { const int rc = check_tables("mldsa_sign_host_group", msgs, msg_off, ctxs, ctx_off, n_ops); if (rc != MLDSA_OK) return rc; }
    std::lock_guard<std::mutex> lk(g->call_mu);
    const int n = (int)g->w.size();
    return run_on_all(g, [&](int i) -> int {
        size_t a, cnt;
        slice(n_ops, n, i, a, cnt);
        if (cnt == 0) return MLDSA_OK;
        const size_t kb = key_idx ? 0 : a;
        return mldsa_sign_host(g->w[i]->ctx, set, mode, sk + kb * (size_t)p->sk_len, n_keys - kb, key_idx ? key_idx + a : nullptr, msgs,
                               msg_off + a, ctxs, ctx_off ? ctx_off + a : nullptr, rnd + a * 32, sigs + a * (size_t)p->sig_len,
                               status ? status + a : nullptr, cnt);
    });
}

int mldsa_keygen_host_group(mldsa_group *g, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys) {
    REQUIRE(g, "mldsa_keygen_host_group: NULL group");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_keygen_host_group: unknown parameter set");
    if (n_keys == 0) return MLDSA_OK;
    REQUIRE(xi && pk && sk, "mldsa_keygen_host_group: NULL pointer");
    std::lock_guard<std::mutex> lk(g->call_mu);
    const int n = (int)g->w.size();
    return run_on_all(g, [&](int i) -> int {
        size_t a, cnt;
        slice(n_keys, n, i, a, cnt);
        if (cnt == 0) return MLDSA_OK;
        return mldsa_keygen_host(g->w[i]->ctx, set, xi + a * 32, pk + a * (size_t)p->pk_len, sk + a * (size_t)p->sk_len, cnt);
    });
}

// ---- device-resident slices ------------------------------------------------------------------------------------------
// Slice i of the caller's batch already lives on device i of the group (keys expanded there, inputs uploaded there): worker i
// makes the ordinary device-resident call of its context on the slice's own stream.  wait != 0: every worker also waits for its
// stream, so the results are complete when the call returns; wait == 0: the call returns once everything is enqueued
// (mldsa_sign_group then signs with mldsa_sign_async) and mldsa_group_sync waits later.

int mldsa_verify_group(mldsa_group *g, int set, int mode, const mldsa_verify_slice *slices, int wait) {
    REQUIRE(g && slices, "mldsa_verify_group: NULL pointer");
    REQUIRE(params_of(set), "mldsa_verify_group: unknown parameter set");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return run_slices(g, slices, wait, [&](mldsa_ctx *c, const mldsa_verify_slice &s) {
        return mldsa_verify(c, set, mode, s.rho, s.tr, s.t1_d2_hat_mont, s.n_keys, s.key_idx, s.msgs, s.msg_off, s.ctxs, s.ctx_off, s.sigs, s.ok,
                            s.n_ops, s.stream);
    });
}

int mldsa_sign_group(mldsa_group *g, int set, int mode, const mldsa_sign_slice *slices, int wait) {
    REQUIRE(g && slices, "mldsa_sign_group: NULL pointer");
    REQUIRE(params_of(set), "mldsa_sign_group: unknown parameter set");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return run_slices(g, slices, wait, [&](mldsa_ctx *c, const mldsa_sign_slice &s) {
        return (wait ? mldsa_sign : mldsa_sign_async)(c, set, mode, s.rho, s.cap_k, s.tr, s.s_1_hat_mont, s.s_2_hat_mont, s.t_0_hat_mont, s.n_keys,
                                                      s.key_idx, s.msgs, s.msg_off, s.ctxs, s.ctx_off, s.rnd, s.sigs, s.status, s.n_ops, s.stream);
    });
}

int mldsa_keygen_group(mldsa_group *g, int set, const mldsa_keygen_slice *slices, int wait) {
    REQUIRE(g && slices, "mldsa_keygen_group: NULL pointer");
    REQUIRE(params_of(set), "mldsa_keygen_group: unknown parameter set");
    std::lock_guard<std::mutex> lk(g->call_mu);
    return run_slices(g, slices, wait, [&](mldsa_ctx *c, const mldsa_keygen_slice &s) {
        return mldsa_keygen(c, set, s.xi, s.pk, s.sk, s.n_keys, s.stream);
    });
}

int mldsa_group_sync(mldsa_group *g) {
    REQUIRE(g, "mldsa_group_sync: NULL group");
    std::lock_guard<std::mutex> lk(g->call_mu);
    if (g->last_used.empty()) return MLDSA_OK;
    return run_on_all(g, [&](int i) -> int {
        if (!g->last_used[(size_t)i]) return MLDSA_OK;
        DeviceGuard dg(g->w[i]->ctx->device);
        MLDSA_HIP_CHECK(hipStreamSynchronize(g->last_streams[(size_t)i]));
        return MLDSA_OK;
    });
}

// ---- device-resident verdicts: all-gather ----------------------------------------------------------------------------
// bufs[i]: device pointer ON DEVICE i of the group, N * ceil(n_ops / N) bytes; slice i (mldsa_group_shard) of it holds what
// device i computed.  Afterwards every buffer holds all n_ops bytes.  use_rccl: 1 = ncclAllGather (distinct devices only),
// 0 = device-to-device copies, -1 = RCCL when the devices are distinct and librccl.so loads, copies otherwise.
int mldsa_group_allgather(mldsa_group *g, uint8_t *const *bufs, size_t n_ops, int use_rccl) {
    REQUIRE(g && bufs, "mldsa_group_allgather: NULL pointer");
    REQUIRE(use_rccl >= -1 && use_rccl <= 1, "mldsa_group_allgather: use_rccl is -1, 0 or 1");
    const int n = (int)g->w.size();
    for (int i = 0; i < n; i++) REQUIRE(bufs[i], "mldsa_group_allgather: NULL buffer");
    if (n_ops == 0) return MLDSA_OK;
    std::lock_guard<std::mutex> lk(g->call_mu);
    const size_t per = (n_ops + (size_t)n - 1) / (size_t)n;
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) distinct &= g->w[i]->ctx->device != g->w[j]->ctx->device;
    if (use_rccl == 1 && !distinct) return set_error(MLDSA_ERR_PARAM, "mldsa_group_allgather: RCCL needs distinct devices");
    if (g->gather_streams.empty()) {
        std::vector<hipStream_t> made((size_t)n, nullptr);
        hipError_t e = hipSuccess;
        for (int i = 0; i < n && e == hipSuccess; i++) {
            DeviceGuard dg(g->w[i]->ctx->device);
            e = hipStreamCreateWithFlags(&made[(size_t)i], hipStreamNonBlocking);
        }
        if (e != hipSuccess) {  // all or nothing: a later call must not find a half-filled table and use the NULL stream
            for (int i = 0; i < n; i++)
                if (made[(size_t)i]) { DeviceGuard dg(g->w[i]->ctx->device); (void)hipStreamDestroy(made[(size_t)i]); }
            return set_error(MLDSA_ERR_DEVICE, "mldsa_group_allgather: stream creation", e);
        }
        g->gather_streams = std::move(made);
    }
    // Ordering: the verdicts were produced by op-level calls of the group's contexts, on streams of the caller's choosing.  Every
    // context records an event behind its last op-level call (OpGuard); each gather stream waits for the events of ALL contexts
    // (with peer copies device i reads the other devices' buffers), so a gather enqueued right after mldsa_verify /
    // mldsa_verify_group needs no synchronisation by the caller.  Data written by other means is the caller's to order.
    for (int j = 0; j < n; j++) {
        mldsa_ctx *c = g->w[j]->ctx;
        std::lock_guard<std::mutex> clk(c->op_mutex);
        if (!c->ws_busy) continue;
        for (int i = 0; i < n; i++) {
            DeviceGuard dg(g->w[i]->ctx->device);
            MLDSA_HIP_CHECK(hipStreamWaitEvent(g->gather_streams[(size_t)i], c->ws_ev, 0));
        }
    }
    bool rccl = use_rccl != 0 && distinct;
    if (rccl && !g->rccl_tried) {
        g->rccl_tried = true;
        // single-process communicators over the group's devices (ncclCommInitAll); RCCL is loaded here, not linked: hosts that
        // never gather pay nothing for it
        void *h = load_rccl(&g->rccl_how, &g->rccl_path, &g->rccl_version);
        if (h) {
            typedef int (*init_all_t)(void **, int, const int *);
            init_all_t init_all = (init_all_t)dlsym(h, "ncclCommInitAll");
            std::vector<int> devs((size_t)n);
            for (int i = 0; i < n; i++) devs[i] = g->w[i]->ctx->device;
            g->comms.assign((size_t)n, nullptr);
            if (init_all && init_all(g->comms.data(), n, devs.data()) == 0) g->rccl = h;
            else { g->comms.clear(); dlclose(h); }
        }
    }
    if (rccl && !g->rccl) {
        if (use_rccl == 1) return set_error(MLDSA_ERR_DEVICE, "mldsa_group_allgather: librccl.so could not be loaded / initialised");
        rccl = false;
    }
    if (rccl) {
        typedef int (*ag_t)(const void *, void *, size_t, int, void *, hipStream_t);
        typedef int (*grp_t)(void);
        ag_t all_gather = (ag_t)dlsym(g->rccl, "ncclAllGather");
        grp_t group_start = (grp_t)dlsym(g->rccl, "ncclGroupStart"), group_end = (grp_t)dlsym(g->rccl, "ncclGroupEnd");
        if (!all_gather || !group_start || !group_end) return set_error(MLDSA_ERR_DEVICE, "mldsa_group_allgather: RCCL symbols missing");
        int rc = group_start();
        for (int i = 0; i < n && rc == 0; i++) {
            DeviceGuard dg(g->w[i]->ctx->device);
            rc = all_gather(bufs[i] + (size_t)i * per, bufs[i], per, 0 /* ncclInt8 */, g->comms[i], g->gather_streams[i]);  // in place
        }
        const int rc2 = group_end();
        if (rc != 0 || rc2 != 0) return set_error(MLDSA_ERR_DEVICE, "mldsa_group_allgather: ncclAllGather failed");
    } else {
        // every device pulls the other slices (hipMemcpyPeerAsync: over xGMI between GPUs, a plain copy on one)
        for (int i = 0; i < n; i++) {
            DeviceGuard dg(g->w[i]->ctx->device);
            for (int j = 0; j < n; j++) {
                if (j == i) continue;
                size_t a, cnt;
                slice(n_ops, n, j, a, cnt);
                if (cnt == 0 || bufs[i] == bufs[j]) continue;
                MLDSA_HIP_CHECK(hipMemcpyPeerAsync(bufs[i] + a, g->w[i]->ctx->device, bufs[j] + a, g->w[j]->ctx->device, cnt, g->gather_streams[i]));
            }
        }
    }
    for (int i = 0; i < n; i++) {
        DeviceGuard dg(g->w[i]->ctx->device);
        MLDSA_HIP_CHECK(hipStreamSynchronize(g->gather_streams[i]));
    }
    return MLDSA_OK;
}

// Which RCCL does / would this process gather with?  g != NULL: what the group's first RCCL gather bound (MLDSA_ERR_PARAM before it).
// g == NULL: a probe -- the loader's search is run without creating a communicator or touching a device (day-one preflight,
// tools/scale_preflight.py; the handle is released again).  buf <- "<reused|loaded> <file>"; returns ncclGetVersion's code (e.g. 22606),
// 0 if the library does not say, MLDSA_ERR_DEVICE if no RCCL can be loaded.
int mldsa_group_rccl_info(const mldsa_group *g, char *buf, size_t buf_len) {
    std::string how, path;
    int version = 0;
    if (g) {
        REQUIRE(g->rccl, "mldsa_group_rccl_info: this group has not gathered with RCCL");
        how = g->rccl_how; path = g->rccl_path; version = g->rccl_version;
    } else {
        void *h = load_rccl(&how, &path, &version);
        if (!h) return set_error(MLDSA_ERR_DEVICE, "mldsa_group_rccl_info: no librccl.so could be loaded");
        dlclose(h);
    }
    if (buf && buf_len) snprintf(buf, buf_len, "%s %s", how.c_str(), path.c_str());
    return version;
}

}  // extern "C"
