// Single-operation callers in front of the batched path.
//
// The reference's API is one operation per call -- `pk.verify(msg, sig, ctx)`, `sk.try_sign_with_rng(rng, msg, ctx)`,
// `KG::keygen_from_seed(xi)` (src/traits.rs:118-308, 330-362; src/lib.rs:247-296, 364-380) -- and a Rust shim that keeps that
// surface calls the C ABI with n_ops = 1, where a GPU call is all latency (0.18 ms per verification against 0.03 ms on a
// host core).  A batcher coalesces the calls of MANY host threads: each caller blocks in mldsa_batcher_verify / _sign /
// _keygen, its arguments are copied into the page-locked arrays of the batch that is filling, a dispatcher thread hands
// whatever has arrived to mldsa_verify_host / mldsa_sign_host / mldsa_keygen_host as ONE call, and every caller returns with
// its own result.  While a batch runs on the device the next one fills, so the batch size follows the load by itself
// (max_wait_us > 0 additionally holds a batch open for that long after its first request).  Keys are de-duplicated per batch
// (requests that carry the same key bytes share one try_from_bytes), results are exactly those of the batched entry points.
//
// Host code only (plain C++, no kernels): everything on the device goes through the ordinary host-memory entry points (host_api.hip).
#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "ctx.h"

using namespace mldsa;

namespace {

enum { OP_VERIFY = 0, OP_SIGN = 1, OP_KEYGEN = 2, N_OPS = 3 };
enum State { FREE, OPEN, SEALED, RUNNING, DONE };

// Synchronisation.  Hundreds of callers pass through the batcher per batch, each for a fraction of a microsecond (reserve a slot,
// look the key up): a std::mutex under that load turns into a convoy of futex hand-offs (measured with 256 callers: 29 k calls/s,
// against 166 k with what follows), so the shared state sits behind a spinlock, and everything that SLEEPS -- callers waiting for
// their batch, the idle dispatcher, callers waiting for a free batch -- sleeps on a futex word and is woken without any lock.
struct SpinLock {
    std::atomic<uint32_t> v{0};
    void lock() {
        int spins = 0;
        while (v.exchange(1, std::memory_order_acquire))
            while (v.load(std::memory_order_relaxed)) {
                if (++spins < 256) __builtin_ia32_pause();
                else { std::this_thread::yield(); spins = 0; }
            }
    }
    void unlock() { v.store(0, std::memory_order_release); }
};
using Lock = std::unique_lock<SpinLock>;

inline void futex_wait(std::atomic<uint32_t> *w, uint32_t seen, const timespec *timeout = nullptr) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, seen, timeout, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *w, int n) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

struct PinnedBuf {  // page-locked array (mldsa_host_alloc): the *_host entry points copy from / to it by DMA
    uint8_t *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n) {
        if (n <= bytes) return MLDSA_OK;
        void *q = nullptr;
        const int rc = mldsa_host_alloc(&q, n);
        if (rc != MLDSA_OK) return rc;
        release();
        p = static_cast<uint8_t *>(q);
        bytes = n;
        return MLDSA_OK;
    }
    void release() {
        if (p) { std::memset(p, 0, bytes); (void)mldsa_host_free(p); }  // may have held private keys / seeds
        p = nullptr;
        bytes = 0;
    }
    ~PinnedBuf() { release(); }
};

struct Batch {
    int op = OP_VERIFY, mode = 0;
    State state = FREE;
    size_t n = 0, n_keys = 0;
    size_t msg_used = 0, ctx_used = 0;
    std::atomic<size_t> copying{0}, readers{0};  // callers still copying their arguments in / their results out
    // bumped when the batch's results are ready.  Waking is a tree: the dispatcher wakes two callers, every caller that wakes up
    // wakes two more -- the dispatcher is back at the next batch after one system call, not after one wake-up per caller.
    std::atomic<uint32_t> done_gen{0};
    std::chrono::steady_clock::time_point first_arrival;
    int rc = MLDSA_OK;
    std::string err;
    // staging, all page-locked.  keys: distinct wire-format keys of the batch; in0: signatures (verify) / rnd (sign) / xi (keygen)
    PinnedBuf keys, msgs, ctxs, in0, out0, out1, kidx, moff, coff, status;
    std::unordered_multimap<uint64_t, uint32_t> key_slots;  // first 8 bytes of a key -> its slot (memcmp decides)
};

}  // namespace

struct mldsa_batcher {
    mldsa_ctx *ctx = nullptr;
    const mldsa_params *p = nullptr;
    size_t max_batch = 0;
    unsigned max_wait_us = 0;
    SpinLock mu;  // batch states, open[], slot / byte reservations, key tables, stats
    std::atomic<uint32_t> work_seq{0}, free_seq{0};  // futex words: "the dispatcher has something to look at", "a batch became free"
    std::thread th;
    bool quit = false, dispatcher_idle = false;
    static constexpr int NB = 3;  // per operation: one running, one filling, one being read out
    Batch batches[N_OPS][NB];
    Batch *open[N_OPS] = {nullptr, nullptr, nullptr};
    mldsa_batcher_stats stats{};
};

namespace {

size_t key_len(const mldsa_batcher *b, int op) { return op == OP_VERIFY ? (size_t)b->p->pk_len : op == OP_SIGN ? (size_t)b->p->sk_len : 0; }

int alloc_batch(mldsa_batcher *b, Batch &t, int op) {
    const size_t n = b->max_batch;
    t.op = op;
    int rc = MLDSA_OK;
    auto need = [&](PinnedBuf &buf, size_t bytes) { if (rc == MLDSA_OK) rc = buf.reserve(bytes); };
    if (op != OP_KEYGEN) {
        need(t.keys, n * key_len(b, op));
        need(t.msgs, std::max<size_t>(n * 256, 1 << 16));
        need(t.ctxs, n * 255);
        need(t.kidx, n * sizeof(uint32_t));
        need(t.moff, (n + 1) * sizeof(uint64_t));
        need(t.coff, (n + 1) * sizeof(uint64_t));
    }
    if (op == OP_VERIFY) { need(t.in0, n * (size_t)b->p->sig_len); need(t.out0, n); }
    if (op == OP_SIGN) { need(t.in0, n * 32); need(t.out0, n * (size_t)b->p->sig_len); need(t.status, n * sizeof(int32_t)); }
    if (op == OP_KEYGEN) { need(t.in0, n * 32); need(t.out0, n * (size_t)b->p->pk_len); need(t.out1, n * (size_t)b->p->sk_len); }
    return rc;
}

// with the lock held: tell the dispatcher there is something to look at; returns whether the caller must futex_wake(work_seq)
// once it has released the lock
bool poke_dispatcher(mldsa_batcher *b) {
    if (!b->dispatcher_idle) return false;
    b->work_seq.fetch_add(1, std::memory_order_release);
    return true;
}

enum { E_NONE = 0, E_QUIT, E_NOMEM };

// the batch a new request of (op, mode) with msg_len message bytes goes into; lock held on entry and on return, released while
// waiting for a free batch.  wake: the dispatcher must be woken after the lock is released.
Batch *open_batch(mldsa_batcher *b, Lock &lk, int op, int mode, size_t msg_len, int &err, bool &wake) {
    for (;;) {
        if (b->quit) { err = E_QUIT; return nullptr; }
        Batch *t = b->open[op];
        if (t) {
            const bool fits = t->n < b->max_batch && t->mode == mode && (op == OP_KEYGEN || t->msg_used + msg_len <= t->msgs.bytes);
            if (fits) return t;
            if (t->n == 0 && t->mode == mode) {  // one message larger than the whole staging array: grow it while nobody else is inside
                if (t->msgs.reserve(msg_len) != MLDSA_OK) { err = E_NOMEM; return nullptr; }  // (rare; the allocation holds the lock)
                return t;
            }
            if (t->n == 0) { t->mode = mode; continue; }
            t->state = SEALED;  // full, or a request of another mode: it runs as it is
            b->open[op] = nullptr;
            wake |= poke_dispatcher(b);
            continue;
        }
        for (Batch &c : b->batches[op])
            if (c.state == FREE) { t = &c; break; }
        if (!t) {
            const uint32_t seen = b->free_seq.load(std::memory_order_relaxed);
            lk.unlock();
            if (wake) { futex_wake(&b->work_seq, 1); wake = false; }
            futex_wait(&b->free_seq, seen);
            lk.lock();
            continue;
        }
        t->state = OPEN;
        t->mode = mode;
        t->n = t->n_keys = t->msg_used = t->ctx_used = 0;
        t->rc = MLDSA_OK;
        t->err.clear();
        t->key_slots.clear();
        b->open[op] = t;
    }
}

// slot of `key` in the batch's table of distinct keys (copied in on first sight); lock held
uint32_t key_slot(mldsa_batcher *b, Batch *t, const uint8_t *key) {
    const size_t kl = key_len(b, t->op);
    uint64_t tag;
    std::memcpy(&tag, key, sizeof(tag));
    auto range = t->key_slots.equal_range(tag);
    for (auto it = range.first; it != range.second; ++it)
        if (std::memcmp(t->keys.p + (size_t)it->second * kl, key, kl) == 0) return it->second;
    const uint32_t slot = (uint32_t)t->n_keys++;
    std::memcpy(t->keys.p + (size_t)slot * kl, key, kl);
    t->key_slots.emplace(tag, slot);
    return slot;
}

void run_batch(mldsa_batcher *b, Batch *t) {
    const mldsa_params *p = b->p;
    int rc;
    if (t->op == OP_KEYGEN) {
        rc = mldsa_keygen_host(b->ctx, p->set, t->in0.p, t->out0.p, t->out1.p, t->n);
    } else {
        uint64_t *moff = reinterpret_cast<uint64_t *>(t->moff.p), *coff = reinterpret_cast<uint64_t *>(t->coff.p);
        moff[t->n] = t->msg_used;
        coff[t->n] = t->ctx_used;
        const uint32_t *kidx = reinterpret_cast<const uint32_t *>(t->kidx.p);
        if (t->op == OP_VERIFY)
            rc = mldsa_verify_host(b->ctx, p->set, t->mode, t->keys.p, t->n_keys, kidx, t->msgs.p, moff, t->ctxs.p, coff, t->in0.p, t->out0.p, t->n);
        else
            rc = mldsa_sign_host(b->ctx, p->set, t->mode, t->keys.p, t->n_keys, kidx, t->msgs.p, moff, t->ctxs.p, coff, t->in0.p, t->out0.p,
                                 reinterpret_cast<int32_t *>(t->status.p), t->n);
    }
    t->rc = rc;
    if (rc != MLDSA_OK) { const char *e = mldsa_last_error(); t->err = e ? e : ""; }
    if (t->op != OP_VERIFY) {  // private keys and seeds do not outlive the call (types.rs:19)
        if (t->op == OP_SIGN) std::memset(t->keys.p, 0, t->n_keys * key_len(b, OP_SIGN));
        std::memset(t->in0.p, 0, t->n * 32);
    }
}

void dispatcher(mldsa_batcher *b) {
    Lock lk(b->mu);
    auto sleep_until_poked = [&](const timespec *timeout) {
        b->dispatcher_idle = true;
        const uint32_t seen = b->work_seq.load(std::memory_order_relaxed);
        lk.unlock();
        futex_wait(&b->work_seq, seen, timeout);  // a poke between the unlock and the wait has changed the word: returns at once
        lk.lock();
        b->dispatcher_idle = false;
    };
    for (;;) {
        // the oldest batch that has requests (sealed ones first: they were opened before the one that is filling)
        Batch *t = nullptr;
        for (int op = 0; op < N_OPS; op++)
            for (Batch &c : b->batches[op])
                if ((c.state == SEALED || (c.state == OPEN && c.n > 0)) && (!t || c.first_arrival < t->first_arrival)) t = &c;
        if (!t) {
            if (b->quit) return;
            sleep_until_poked(nullptr);
            continue;
        }
        if (t->state == OPEN) {
            if (b->max_wait_us && t->n < b->max_batch && !b->quit) {
                const auto left = t->first_arrival + std::chrono::microseconds(b->max_wait_us) - std::chrono::steady_clock::now();
                const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(left).count();
                if (ns > 0) {
                    const timespec ts{(time_t)(ns / 1000000000LL), (long)(ns % 1000000000LL)};
                    sleep_until_poked(&ts);
                    continue;  // look again: it may be full, sealed, or another batch may be older
                }
            }
            t->state = SEALED;
            b->open[t->op] = nullptr;
        }
        t->state = RUNNING;
        lk.unlock();
        // requests that reserved a slot and are still copying their bytes in: microseconds
        while (t->copying.load(std::memory_order_acquire)) std::this_thread::yield();
        run_batch(b, t);
        lk.lock();
        b->stats.batches++;
        b->stats.requests += t->n;
        if (t->n > b->stats.largest_batch) b->stats.largest_batch = t->n;
        b->stats.distinct_keys += t->n_keys;
        t->state = DONE;
        t->readers.store(t->n, std::memory_order_relaxed);
        t->done_gen.fetch_add(1, std::memory_order_release);
        lk.unlock();
        futex_wake(&t->done_gen, 2);
        lk.lock();
    }
}

struct Req {
    int op, mode;
    const uint8_t *key, *msg, *ctx, *in0;
    size_t msg_len, ctx_len;
    uint8_t *out0, *out1;
};

int submit(mldsa_batcher *b, const Req &r) {
    const mldsa_params *p = b->p;
    Lock lk(b->mu);
    int err = E_NONE;
    bool wake = false;
    Batch *t = open_batch(b, lk, r.op, r.mode, r.msg_len, err, wake);
    if (!t) {
        lk.unlock();
        if (wake) futex_wake(&b->work_seq, 1);
        return err == E_QUIT ? set_error(MLDSA_ERR_PARAM, "mldsa_batcher: destroyed while in use")
                             : set_error(MLDSA_ERR_NOMEM, "mldsa_batcher: no page-locked memory for a message of this size");
    }
    const size_t i = t->n++;
    size_t m0 = 0, c0 = 0;
    if (r.op != OP_KEYGEN) {
        reinterpret_cast<uint32_t *>(t->kidx.p)[i] = key_slot(b, t, r.key);
        m0 = t->msg_used;
        c0 = t->ctx_used;
        reinterpret_cast<uint64_t *>(t->moff.p)[i] = m0;
        reinterpret_cast<uint64_t *>(t->coff.p)[i] = c0;
        t->msg_used += r.msg_len;
        t->ctx_used += r.ctx_len;
    }
    t->copying.fetch_add(1, std::memory_order_relaxed);
    const uint32_t gen = t->done_gen.load(std::memory_order_relaxed);
    if (i == 0) {
        t->first_arrival = std::chrono::steady_clock::now();
        wake |= poke_dispatcher(b);  // it may be asleep with nothing to do
    }
    if (t->n == b->max_batch) {  // full: hand it over now
        t->state = SEALED;
        b->open[r.op] = nullptr;
        wake |= poke_dispatcher(b);
    }
    lk.unlock();
    if (wake) futex_wake(&b->work_seq, 1);
    // the request's own bytes, outside the lock (other callers fill their slots at the same time)
    if (r.op != OP_KEYGEN) {
        if (r.msg_len) std::memcpy(t->msgs.p + m0, r.msg, r.msg_len);
        if (r.ctx_len) std::memcpy(t->ctxs.p + c0, r.ctx, r.ctx_len);
    }
    const size_t in_len = r.op == OP_VERIFY ? (size_t)p->sig_len : 32;
    std::memcpy(t->in0.p + i * in_len, r.in0, in_len);
    t->copying.fetch_sub(1, std::memory_order_release);
    // results: the batch object is not reused before every caller of it has passed the countdown below
    bool slept = false;
    while (t->done_gen.load(std::memory_order_acquire) == gen) { futex_wait(&t->done_gen, gen); slept = true; }
    if (slept) futex_wake(&t->done_gen, 2);
    int rc = t->rc;
    if (rc != MLDSA_OK) set_error(rc, ("mldsa_batcher: " + t->err).c_str());
    else if (r.op == OP_VERIFY) *r.out0 = t->out0.p[i];
    else if (r.op == OP_SIGN) {
        std::memcpy(r.out0, t->out0.p + i * (size_t)p->sig_len, (size_t)p->sig_len);
        const int32_t st = reinterpret_cast<const int32_t *>(t->status.p)[i];
        if (st != MLDSA_OK) rc = set_error(st, st == MLDSA_ERR_CTX_LEN ? "mldsa_batcher_sign: ctx longer than 255 bytes" : "mldsa_batcher_sign: the operation was refused");
    } else {
        std::memcpy(r.out0, t->out0.p + i * (size_t)p->pk_len, (size_t)p->pk_len);
        std::memcpy(r.out1, t->out1.p + i * (size_t)p->sk_len, (size_t)p->sk_len);
    }
    if (t->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // the last caller out returns the batch
        if (t->op == OP_KEYGEN) std::memset(t->out1.p, 0, t->n * (size_t)p->sk_len);
        lk.lock();
        t->state = FREE;
        b->free_seq.fetch_add(1, std::memory_order_release);
        lk.unlock();
        futex_wake(&b->free_seq, (int)std::min<size_t>(b->max_batch, INT_MAX));  // no more callers than the batch can take
    }
    return rc;
}

}  // namespace

#define REQUIRE(cond, msg) \
    do { if (!(cond)) return set_error(MLDSA_ERR_PARAM, msg); } while (0)

extern "C" {

int mldsa_batcher_create(mldsa_ctx *ctx, int set, size_t max_batch, unsigned max_wait_us, mldsa_batcher **out) {
    REQUIRE(out, "mldsa_batcher_create: NULL out");
    *out = nullptr;
    REQUIRE(ctx, "mldsa_batcher_create: NULL context");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_batcher_create: unknown parameter set");
    REQUIRE(max_batch >= 1 && max_batch <= (1u << 20), "mldsa_batcher_create: max_batch in 1 ... 2^20");
    std::unique_ptr<mldsa_batcher> b(new (std::nothrow) mldsa_batcher());
    if (!b) return set_error(MLDSA_ERR_NOMEM, "mldsa_batcher_create: host allocation failed");
    b->ctx = ctx;
    b->p = p;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    for (int op = 0; op < N_OPS; op++)
        for (Batch &t : b->batches[op]) {
            const int rc = alloc_batch(b.get(), t, op);
            if (rc != MLDSA_OK) return rc;  // message of mldsa_host_alloc; the buffers made so far go with `b`
        }
    b->th = std::thread(dispatcher, b.get());
    *out = b.release();
    return MLDSA_OK;
}

void mldsa_batcher_destroy(mldsa_batcher *b) {
    if (!b) return;
    {
        Lock lk(b->mu);
        b->quit = true;  // batches that hold requests still run; new requests are refused
        b->work_seq.fetch_add(1, std::memory_order_release);
        b->free_seq.fetch_add(1, std::memory_order_release);
    }
    futex_wake(&b->work_seq, 1);
    futex_wake(&b->free_seq, INT_MAX);
    if (b->th.joinable()) b->th.join();
    for (;;) {  // callers still reading their results out
        bool busy = false;
        {
            Lock lk(b->mu);
            for (int op = 0; op < N_OPS; op++)
                for (Batch &t : b->batches[op]) busy |= t.state != FREE && !(t.state == OPEN && t.n == 0);
        }
        if (!busy) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    delete b;
}

int mldsa_batcher_verify(mldsa_batcher *b, int mode, const uint8_t *pk, const uint8_t *msg, size_t msg_len, const uint8_t *ctx, size_t ctx_len,
                         const uint8_t *sig, uint8_t *ok) {
    REQUIRE(b && pk && sig && ok, "mldsa_batcher_verify: NULL pointer");
    REQUIRE(mode >= MLDSA_MODE_PURE && mode <= MLDSA_MODE_PREHASH, "mldsa_batcher_verify: unknown mode");
    REQUIRE((msg || !msg_len) && (ctx || !ctx_len), "mldsa_batcher_verify: NULL bytes with a non-zero length");
    *ok = 0;
    if (ctx_len > 255 && mode != MLDSA_MODE_INTERNAL) return MLDSA_OK;  // lib.rs:368-370: false, before anything is hashed
    if (mode == MLDSA_MODE_INTERNAL) ctx_len = 0;                       // the internal interface has no ctx (ml_dsa.rs:386-395)
    const Req r{OP_VERIFY, mode, pk, msg, ctx, sig, msg_len, ctx_len, ok, nullptr};
    return submit(b, r);
}

int mldsa_batcher_sign(mldsa_batcher *b, int mode, const uint8_t *sk, const uint8_t *msg, size_t msg_len, const uint8_t *ctx, size_t ctx_len,
                       const uint8_t *rnd, uint8_t *sig) {
    REQUIRE(b && sk && rnd && sig, "mldsa_batcher_sign: NULL pointer");
    REQUIRE(mode >= MLDSA_MODE_PURE && mode <= MLDSA_MODE_PREHASH, "mldsa_batcher_sign: unknown mode");
    REQUIRE((msg || !msg_len) && (ctx || !ctx_len), "mldsa_batcher_sign: NULL bytes with a non-zero length");
    if (ctx_len > 255 && mode != MLDSA_MODE_INTERNAL) {  // lib.rs:274
        std::memset(sig, 0, (size_t)b->p->sig_len);
        return set_error(MLDSA_ERR_CTX_LEN, "mldsa_batcher_sign: ctx longer than 255 bytes");
    }
    if (mode == MLDSA_MODE_INTERNAL) ctx_len = 0;
    const Req r{OP_SIGN, mode, sk, msg, ctx, rnd, msg_len, ctx_len, sig, nullptr};
    return submit(b, r);
}

int mldsa_batcher_keygen(mldsa_batcher *b, const uint8_t *xi, uint8_t *pk, uint8_t *sk) {
    REQUIRE(b && xi && pk && sk, "mldsa_batcher_keygen: NULL pointer");
    const Req r{OP_KEYGEN, 0, nullptr, nullptr, nullptr, xi, 0, 0, pk, sk};
    return submit(b, r);
}

int mldsa_batcher_get_stats(mldsa_batcher *b, mldsa_batcher_stats *out) {
    REQUIRE(b && out, "mldsa_batcher_get_stats: NULL pointer");
    Lock lk(b->mu);
    *out = b->stats;
    return MLDSA_OK;
}

}  // extern "C"
