// Single-operation callers in front of the batched path.
//
// The reference's API is one operation per call -- `pk.verify(msg, sig, ctx)`, `sk.try_sign_with_rng(rng, msg, ctx)`,
// `KG::keygen_from_seed(xi)` (src/traits.rs:118-308, 330-362; src/lib.rs:247-296, 364-380) -- and a Rust shim that keeps that
// surface calls the C ABI with n_ops = 1, where a GPU call is all latency (0.18 ms per verification against 0.03 ms on a
// host core).  A batcher coalesces the calls of MANY host threads: each caller blocks in mldsa_batcher_verify / _sign /
// _keygen, its arguments are copied into the page-locked arrays of the batch that is filling, a dispatcher thread hands
// whatever has arrived to mldsa_verify_host / mldsa_sign_host / mldsa_keygen_host as ONE call, and every caller returns with
// its own result.  While a batch runs on the device the next one fills, so the batch size follows the load by itself
// (max_wait_us > 0 additionally holds a batch open for that long after its first request).
//
// Keys.  In the reference a key is deserialised once (`try_from_bytes` -> PublicKey / PrivateKey, src/ml_dsa.rs:445-498) and then
// used for many calls; callers of the batcher pass wire-format bytes with every call, and the batcher keeps what try_from_bytes
// produces -- the expanded fields AND A_hat = ExpandA(rho), the pre-compute benches/README.md:4-8 names -- in a device-resident
// table of `cache_keys` slots, found again by the key's bytes (a keyed 64-bit hash, then memcmp: a hit is exact).  A batch expands
// only the keys the table does not hold (FIFO replacement, never a key of the batch itself) and runs mldsa_verify_cached_a /
// mldsa_sign_cached_a with the table as its key array: results are bit for bit those of mldsa_verify / mldsa_sign.
//
// Host code only (plain C++, no kernels): the device work goes through the library's own entry points.
//
// Platform.  Written for the platform ROCm runs on -- Linux on x86-64: sleeping and waking go through raw futex words
// (FUTEX_WAIT_PRIVATE / FUTEX_WAKE_PRIVATE) and the spin loop uses the x86 `pause` hint.  Everything else is ISO C++17.  On another
// OS or architecture (or with -DMLDSA_BATCHER_PORTABLE, which the CPU test-suite also builds and runs) the same two primitives fall
// back to a mutex + condition variable per batcher and std::this_thread::yield: same semantics, more wake-up latency.
#if defined(__linux__) && !defined(MLDSA_BATCHER_PORTABLE)
#define MLDSA_BATCHER_FUTEX 1
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#else
#define MLDSA_BATCHER_FUTEX 0
#include <condition_variable>
#endif
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "host_common.h"

using namespace mldsa;

namespace {

enum { OP_VERIFY = 0, OP_SIGN = 1, OP_KEYGEN = 2, N_OPS = 3 };
constexpr size_t ZERO_COPY_MAX_OPS = 256;
enum State { FREE, OPEN, SEALED, RUNNING, DONE };

// Synchronisation.  Hundreds of callers pass through the batcher per batch, each for a fraction of a microsecond (reserve a slot,
// look the key up): a std::mutex under that load turns into a convoy of futex hand-offs (measured with 256 callers: 29 k calls/s,
// against 166 k with what follows), so the shared state sits behind a spinlock, and everything that SLEEPS -- callers waiting for
// their batch, the idle dispatcher, callers waiting for a free batch -- sleeps on a futex word and is woken without any lock.
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}

struct SpinLock {
    std::atomic<uint32_t> v{0};
    void lock() {
        int spins = 0;
        while (v.exchange(1, std::memory_order_acquire))
            while (v.load(std::memory_order_relaxed)) {
                if (++spins < 256) cpu_relax();
                else { std::this_thread::yield(); spins = 0; }
            }
    }
    void unlock() { v.store(0, std::memory_order_release); }
};
using Lock = std::unique_lock<SpinLock>;

#if MLDSA_BATCHER_FUTEX
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t seen, const timespec *timeout = nullptr) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, seen, timeout, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *w, int n) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}
inline void no_core_dump(void *p, size_t bytes) { (void)madvise(p, bytes, MADV_DONTDUMP); }
#else
// Portable stand-in for the two futex calls: one process-wide mutex + condition variable.  "Sleep while *w == seen" and "wake the
// sleepers of w" keep their meaning (every wake-up re-checks the word); n is ignored: everybody looks again.
inline std::mutex &park_mu() { static std::mutex m; return m; }
inline std::condition_variable &park_cv() { static std::condition_variable c; return c; }
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t seen, const timespec *timeout = nullptr) {
    std::unique_lock<std::mutex> lk(park_mu());
    auto changed = [&] { return w->load(std::memory_order_acquire) != seen; };
    if (timeout) park_cv().wait_for(lk, std::chrono::seconds(timeout->tv_sec) + std::chrono::nanoseconds(timeout->tv_nsec), changed);
    else park_cv().wait(lk, changed);
}
inline void futex_wake(std::atomic<uint32_t> *, int) {
    { std::lock_guard<std::mutex> lk(park_mu()); }
    park_cv().notify_all();
}
inline void no_core_dump(void *, size_t) {}
#endif

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
        if (p) { wipe_host(p, bytes); (void)mldsa_host_free(p); }  // may have held private keys / seeds (mldsa_host_free: Quiesce)
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
    size_t key_hits = 0, keys_expanded = 0;  // of this batch (dispatcher)
    std::atomic<size_t> copying{0}, readers{0};  // callers still copying their arguments in / their results out
    // bumped when the batch's results are ready.  Waking is a tree: the dispatcher wakes two callers, every caller that wakes up
    // wakes two more -- the dispatcher is back at the next batch after one system call, not after one wake-up per caller.
    std::atomic<uint32_t> done_gen{0};
    std::chrono::steady_clock::time_point first_arrival;
    int rc = MLDSA_OK;
    std::string err;
    // staging, all page-locked.  keys: distinct wire-format keys of the batch; in0: signatures (verify) / rnd (sign) / xi (keygen)
    PinnedBuf keys, msgs, ctxs, in0, out0, out1, kidx, moff, coff, status;
    PinnedBuf kslot;                                         // per op: the key's slot in the device-resident table (dispatcher)
    std::vector<uint64_t> key_hash;                          // per distinct key of the batch
    std::unordered_multimap<uint64_t, uint32_t> key_slots;   // keyed hash of a key -> its index in `keys` (memcmp decides)
};

// Device-resident table of expanded keys + A_hat (one for public, one for private keys): a ring of `cap` slots
struct KeyTable {
    size_t cap = 0, hand = 0;
    bool is_private = false;
    uint8_t *rho = nullptr, *cap_k = nullptr, *tr = nullptr;
    int32_t *f0 = nullptr, *f1 = nullptr, *f2 = nullptr;  // public: t1_d2_hat_mont; private: s_1_hat_mont, s_2_hat_mont, t_0_hat_mont
    int32_t *a_hat = nullptr;
    // host copy of every slot's wire bytes: what a lookup compares against.  Page-locked (never swapped), excluded from core dumps
    // (MADV_DONTDUMP), cleared slot by slot when a key leaves the table and as a whole before it is freed: a private key lives here
    // exactly as long as its expanded form lives in device memory, and mldsa_batcher_forget_key / _flush_keys /
    // _set_private_key_cache end both.
    PinnedBuf wire;
    std::vector<uint8_t> valid;
    std::vector<uint64_t> hash, last_batch;
    std::unordered_multimap<uint64_t, uint32_t> index;  // keyed hash -> slot
};

struct DevBuf {
    uint8_t *p = nullptr;
    size_t bytes = 0;
};

// One dispatcher: a context, a thread, the key tables and the device staging of the batch it runs.  Several lanes take batches
// from the same queues -- on one GPU two small batches overlap on the device (each is a chain of latency-bound kernels on a
// fraction of the SIMDs), on several GPUs every device has its own lane.
struct Lane {
    mldsa_ctx *ctx = nullptr;
    bool own_ctx = false;
    std::thread th;
    uint64_t batch_id = 0;
    KeyTable tables[2];  // [OP_VERIFY] public keys, [OP_SIGN] private keys
    std::mutex table_mu;  // the dispatcher holds it while a keyed batch runs; mldsa_batcher_forget_key / _flush_keys take it from outside
    hipStream_t stream = nullptr;
    DevBuf d_kslot, d_moff, d_coff, d_msgs, d_ctxs, d_in0, d_out0, d_status, d_kstage;
    PinnedBuf kstage;    // wire bytes of the keys a batch has to expand
};

}  // namespace

struct mldsa_batcher {
    const mldsa_params *p = nullptr;
    size_t max_batch = 0;
    unsigned max_wait_us = 0;
    SpinLock mu;  // batch states, open[], slot / byte reservations, key tables, stats
    std::atomic<uint32_t> work_seq{0}, free_seq{0};  // futex words: "the dispatcher has something to look at", "a batch became free"
    bool quit = false;
    int idle_dispatchers = 0;
    std::vector<std::unique_ptr<Lane>> lanes;
    // per operation: one batch per lane running, one filling, one being read out
    std::vector<std::unique_ptr<Batch>> batches[N_OPS];
    // the page-locked arrays of an operation's batches are made on its first request (tens of MB each for a large max_batch:
    // a verify-only host never pays for signing and key generation)
    std::once_flag alloc_once[N_OPS];
    int alloc_rc[N_OPS] = {MLDSA_OK, MLDSA_OK, MLDSA_OK};
    Batch *open[N_OPS] = {nullptr, nullptr, nullptr};
    mldsa_batcher_stats stats{};
    size_t cache_keys = 0;
    uint64_t hash_seed = 0;
    std::atomic<int> private_key_cache{1};  // 0: a private key leaves the table (host copy and device fields wiped) with the batch that used it
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
        need(t.kslot, n * sizeof(uint32_t));
        t.key_hash.resize(n);
        need(t.moff, (n + 1) * sizeof(uint64_t));
        need(t.coff, (n + 1) * sizeof(uint64_t));
    }
    if (op == OP_VERIFY) { need(t.in0, n * (size_t)b->p->sig_len); need(t.out0, n); }
    if (op == OP_SIGN) { need(t.in0, n * 32); need(t.out0, n * (size_t)b->p->sig_len); need(t.status, n * sizeof(int32_t)); }
    if (op == OP_KEYGEN) { need(t.in0, n * 32); need(t.out0, n * (size_t)b->p->pk_len); need(t.out1, n * (size_t)b->p->sk_len); }
    return rc;
}

// with the lock held: tell the dispatchers there is something to look at; returns whether the caller must futex_wake(work_seq)
// once it has released the lock.  EVERY sleeping dispatcher is woken (there are as many as lanes: a handful): waking one arbitrary
// sleeper could pick the lane that sits in a timed sleep on an open batch and goes back to sleep, while an idle lane sleeps on.
bool poke_dispatcher(mldsa_batcher *b) {
    if (b->idle_dispatchers == 0) return false;
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
            const bool fits = t->n < b->max_batch && t->mode == mode &&
                              (op == OP_KEYGEN || (msg_len <= t->msgs.bytes && t->msg_used <= t->msgs.bytes - msg_len));
            if (fits) return t;
            if (t->n == 0 && t->mode == mode) {
                // One message larger than the whole staging array: grow it while nobody else is inside.  The allocation (hipHostMalloc:
                // milliseconds, and it takes the capture lock) runs OUTSIDE the spinlock: the batch is taken out of circulation
                // (RUNNING: neither callers nor dispatchers touch it), grown, and put back as the open batch.
                b->open[op] = nullptr;
                t->state = RUNNING;
                lk.unlock();
                const int grow_rc = t->msgs.reserve(msg_len);
                lk.lock();
                t->state = FREE;
                b->free_seq.fetch_add(1, std::memory_order_release);
                lk.unlock();
                futex_wake(&b->free_seq, INT_MAX);
                lk.lock();
                if (grow_rc != MLDSA_OK) { err = E_NOMEM; return nullptr; }
                continue;  // (another caller may have opened a batch meanwhile: look again; the grown one is FREE and will be found)
            }
            if (t->n == 0) { t->mode = mode; continue; }
            t->state = SEALED;  // full, or a request of another mode: it runs as it is
            b->open[op] = nullptr;
            wake |= poke_dispatcher(b);
            continue;
        }
        for (auto &c : b->batches[op])
            if (c->state == FREE) { t = c.get(); break; }
        if (!t) {
            const uint32_t seen = b->free_seq.load(std::memory_order_relaxed);
            lk.unlock();
            if (wake) { futex_wake(&b->work_seq, INT_MAX); wake = false; }
            futex_wait(&b->free_seq, seen);
            lk.lock();
            continue;
        }
        t->state = OPEN;
        t->mode = mode;
        t->n = t->n_keys = t->msg_used = t->ctx_used = t->key_hits = t->keys_expanded = 0;
        t->rc = MLDSA_OK;
        t->err.clear();
        t->key_slots.clear();
        b->open[op] = t;
    }
}

// 64-bit hash of a whole key under the batcher's random seed (PK_LEN / SK_LEN are multiples of 8): only ever a shortcut to the
// memcmp that decides, so it needs to spread, not to resist -- the seed keeps a caller from aiming keys at one bucket
uint64_t hash_key(uint64_t seed, const uint8_t *key, size_t len) {
    uint64_t h = seed ^ (len * 0x9E3779B97F4A7C15ull);
    for (size_t i = 0; i + 8 <= len; i += 8) {
        uint64_t w;
        std::memcpy(&w, key + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    return h;
}

// index of `key` among the batch's distinct keys (copied in on first sight); lock held
uint32_t key_slot(mldsa_batcher *b, Batch *t, const uint8_t *key, uint64_t h) {
    const size_t kl = key_len(b, t->op);
    auto range = t->key_slots.equal_range(h);
    for (auto it = range.first; it != range.second; ++it)
        if (std::memcmp(t->keys.p + (size_t)it->second * kl, key, kl) == 0) return it->second;
    const uint32_t slot = (uint32_t)t->n_keys++;
    std::memcpy(t->keys.p + (size_t)slot * kl, key, kl);
    t->key_hash[slot] = h;
    t->key_slots.emplace(h, slot);
    return slot;
}

#define BCHECK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "mldsa_batcher: " #expr, e_); \
    } while (0)
#define BTRY(expr) do { const int rc_ = (expr); if (rc_ != MLDSA_OK) return rc_; } while (0)

int dev_reserve(DevBuf &d, size_t bytes, bool secret = false) {
    if (bytes <= d.bytes) return MLDSA_OK;
    if (d.p) {
        if (secret) (void)memset_quiesced(d.p, 0, d.bytes);
        BCHECK(free_quiesced(d.p));  // hipFree waits for every stream of the device: not while another lane or context captures one (host_common.h)
        d.p = nullptr;
        d.bytes = 0;
    }
    void *q = nullptr;
    if (malloc_quiesced(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return set_error(MLDSA_ERR_NOMEM, "mldsa_batcher: device staging allocation"); }
    d.p = static_cast<uint8_t *>(q);
    d.bytes = bytes;
    return MLDSA_OK;
}

void dev_release(DevBuf &d, bool secret = false) {
    if (d.p) {
        if (secret) (void)memset_quiesced(d.p, 0, d.bytes);
        (void)free_quiesced(d.p);
    }
    d.p = nullptr;
    d.bytes = 0;
}

int table_alloc(mldsa_batcher *b, KeyTable &kt, bool is_private) {
    if (kt.cap) return MLDSA_OK;
    const mldsa_params *p = b->p;
    const size_t n = b->cache_keys, K = (size_t)p->k, L = (size_t)p->l, kl = is_private ? (size_t)p->sk_len : (size_t)p->pk_len;
    kt.is_private = is_private;
    auto get = [&](void **q, size_t bytes) -> int {
        if (malloc_quiesced(q, bytes) != hipSuccess) { (void)hipGetLastError(); return set_error(MLDSA_ERR_NOMEM, "mldsa_batcher: key table allocation"); }
        return MLDSA_OK;
    };
    BTRY(get((void **)&kt.rho, n * 32));
    BTRY(get((void **)&kt.tr, n * 64));
    BTRY(get((void **)&kt.a_hat, n * K * L * 1024));
    if (is_private) {
        BTRY(get((void **)&kt.cap_k, n * 32));
        BTRY(get((void **)&kt.f0, n * L * 1024));
        BTRY(get((void **)&kt.f1, n * K * 1024));
        BTRY(get((void **)&kt.f2, n * K * 1024));
    } else {
        BTRY(get((void **)&kt.f0, n * K * 1024));
    }
    BTRY(kt.wire.reserve(n * kl));
    std::memset(kt.wire.p, 0, n * kl);
    if (is_private) no_core_dump(kt.wire.p, kt.wire.bytes);
    kt.valid.assign(n, 0);
    kt.hash.assign(n, 0);
    kt.last_batch.assign(n, 0);
    kt.cap = n;
    return MLDSA_OK;
}

void table_free(mldsa_batcher *b, KeyTable &kt) {
    const size_t n = kt.cap, K = (size_t)b->p->k, L = (size_t)b->p->l;
    if (kt.is_private && n) {  // expanded private keys (types.rs:19 ZeroizeOnDrop)
        if (kt.cap_k) (void)memset_quiesced(kt.cap_k, 0, n * 32);
        if (kt.f0) (void)memset_quiesced(kt.f0, 0, n * L * 1024);
        if (kt.f1) (void)memset_quiesced(kt.f1, 0, n * K * 1024);
        if (kt.f2) (void)memset_quiesced(kt.f2, 0, n * K * 1024);
        (void)device_sync_quiesced();
    }
    for (void *q : {(void *)kt.rho, (void *)kt.cap_k, (void *)kt.tr, (void *)kt.f0, (void *)kt.f1, (void *)kt.f2, (void *)kt.a_hat})
        if (q) (void)free_quiesced(q);
    kt.wire.release();  // (wipes before it frees)
    kt.rho = kt.cap_k = kt.tr = nullptr;
    kt.f0 = kt.f1 = kt.f2 = kt.a_hat = nullptr;
    kt.cap = kt.hand = 0;
    kt.valid.clear(); kt.hash.clear(); kt.last_batch.clear(); kt.index.clear();
}

// Take slot s out of the table.  For a private table the key is gone afterwards: the host copy of its wire bytes is cleared and the
// secret fields of the device slot (K, s1, s2, t0 in NTT form) are zeroed on the lane's stream (rho, tr and A_hat are public).
// The caller synchronises the stream before it reports the key as forgotten.
void table_drop(mldsa_batcher *b, Lane &ln, KeyTable &kt, uint32_t s) {
    auto range = kt.index.equal_range(kt.hash[s]);
    for (auto it = range.first; it != range.second; ++it)
        if (it->second == s) { kt.index.erase(it); break; }
    kt.valid[s] = 0;
    if (!kt.is_private) return;
    const size_t K = (size_t)b->p->k, L = (size_t)b->p->l, kl = (size_t)b->p->sk_len;
    wipe_host(kt.wire.p + (size_t)s * kl, kl);
    if (hipMemsetAsync(kt.cap_k + (size_t)s * 32, 0, 32, ln.stream) != hipSuccess || hipMemsetAsync(kt.f0 + (size_t)s * L * 256, 0, L * 1024, ln.stream) != hipSuccess ||
        hipMemsetAsync(kt.f1 + (size_t)s * K * 256, 0, K * 1024, ln.stream) != hipSuccess || hipMemsetAsync(kt.f2 + (size_t)s * K * 256, 0, K * 1024, ln.stream) != hipSuccess)
        (void)hipGetLastError();
}

// Every distinct key of the batch -> a slot of the device-resident table; the keys the table does not hold are expanded into
// free slots (try_from_bytes + ExpandA, once per key for as long as it stays in the table).  Dispatcher thread only.
int resolve_keys(mldsa_batcher *b, Lane &ln, Batch *t, std::vector<uint32_t> &slot_of) {
    const mldsa_params *p = b->p;
    const bool priv = t->op == OP_SIGN;
    KeyTable &kt = ln.tables[t->op];
    BTRY(table_alloc(b, kt, priv));
    const size_t kl = key_len(b, t->op), K = (size_t)p->k, L = (size_t)p->l;
    const uint64_t id = ++ln.batch_id;
    slot_of.assign(t->n_keys, 0);
    std::vector<uint32_t> miss;
    for (size_t j = 0; j < t->n_keys; j++) {
        const uint8_t *key = t->keys.p + j * kl;
        bool hit = false;
        auto range = kt.index.equal_range(t->key_hash[j]);
        for (auto it = range.first; it != range.second && !hit; ++it)
            if (std::memcmp(kt.wire.p + (size_t)it->second * kl, key, kl) == 0) {
                slot_of[j] = it->second;
                kt.last_batch[it->second] = id;
                hit = true;
            }
        if (!hit) miss.push_back((uint32_t)j);
    }
    t->key_hits = t->n_keys - miss.size();
    t->keys_expanded = miss.size();
    if (miss.empty()) return MLDSA_OK;
    // slots for the new keys, in ring order, skipping what this batch itself uses; consecutive slots form one expansion call
    struct Run { size_t first_slot, first_miss, count; };
    std::vector<Run> runs;
    auto drop = [&](uint32_t s) { table_drop(b, ln, kt, s); };
    BTRY(ln.kstage.reserve(miss.size() * kl));
    for (size_t m = 0; m < miss.size(); m++) {
        size_t s = kt.hand;
        while (kt.valid[s] && kt.last_batch[s] == id) s = (s + 1) % kt.cap;  // cap >= max_batch >= n_keys: ends
        kt.hand = (s + 1) % kt.cap;
        if (kt.valid[s]) drop((uint32_t)s);
        const uint8_t *key = t->keys.p + (size_t)miss[m] * kl;
        std::memcpy(kt.wire.p + s * kl, key, kl);
        std::memcpy(ln.kstage.p + m * kl, key, kl);
        kt.hash[s] = t->key_hash[miss[m]];
        kt.last_batch[s] = id;
        kt.valid[s] = 1;
        kt.index.emplace(kt.hash[s], (uint32_t)s);
        slot_of[miss[m]] = (uint32_t)s;
        if (!runs.empty() && runs.back().first_slot + runs.back().count == s) runs.back().count++;
        else runs.push_back({s, m, 1});
    }
    int rc = dev_reserve(ln.d_kstage, miss.size() * kl, priv);
    if (rc == MLDSA_OK && hipMemcpyAsync(ln.d_kstage.p, ln.kstage.p, miss.size() * kl, hipMemcpyHostToDevice, ln.stream) != hipSuccess)
        rc = set_error(MLDSA_ERR_DEVICE, "mldsa_batcher: key upload");
    for (size_t r = 0; r < runs.size() && rc == MLDSA_OK; r++) {
        const Run &u = runs[r];
        const uint8_t *src = ln.d_kstage.p + u.first_miss * kl;
        const size_t s = u.first_slot;
        if (priv) rc = mldsa_sk_expand(ln.ctx, p->set, src, kt.rho + s * 32, kt.cap_k + s * 32, kt.tr + s * 64, kt.f0 + s * L * 256, kt.f1 + s * K * 256,
                                       kt.f2 + s * K * 256, u.count, ln.stream);
        else rc = mldsa_pk_expand(ln.ctx, p->set, src, kt.rho + s * 32, kt.tr + s * 64, kt.f0 + s * K * 256, u.count, ln.stream);
        if (rc == MLDSA_OK) rc = mldsa_expand_a(ln.ctx, p->set, kt.rho + s * 32, kt.a_hat + s * K * L * 256, u.count, ln.stream);
    }
    if (priv) {  // the wire bytes of private keys leave the staging buffers with the batch
        if (ln.d_kstage.p && hipMemsetAsync(ln.d_kstage.p, 0, std::min(ln.d_kstage.bytes, miss.size() * kl), ln.stream) != hipSuccess) (void)hipGetLastError();
        if (hipStreamSynchronize(ln.stream) != hipSuccess) (void)hipGetLastError();  // the upload has read the page-locked copy
        wipe_host(ln.kstage.p, miss.size() * kl);
    }
    if (rc != MLDSA_OK)  // nothing half-expanded stays findable
        for (uint32_t j : miss) drop(slot_of[j]);
    return rc;
}

// verify / sign of one batch on the device-resident key table
int run_keyed(mldsa_batcher *b, Lane &ln, Batch *t) {
    const mldsa_params *p = b->p;
    const size_t n = t->n, sgl = (size_t)p->sig_len;
    DeviceGuard dg(mldsa_ctx_device(ln.ctx));
    if (!ln.stream) BCHECK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    std::vector<uint32_t> slot_of;
    BTRY(resolve_keys(b, ln, t, slot_of));
    KeyTable &kt = ln.tables[t->op];
    uint32_t *kslot = reinterpret_cast<uint32_t *>(t->kslot.p);
    const uint32_t *kidx = reinterpret_cast<const uint32_t *>(t->kidx.p);
    for (size_t i = 0; i < n; i++) kslot[i] = slot_of[kidx[i]];
    uint64_t *moff = reinterpret_cast<uint64_t *>(t->moff.p), *coff = reinterpret_cast<uint64_t *>(t->coff.p);
    moff[n] = t->msg_used;
    coff[n] = t->ctx_used;
    const size_t in_len = t->op == OP_VERIFY ? sgl : 32, out_len = t->op == OP_VERIFY ? 1 : sgl;
    hipStream_t st = ln.stream;
    // Small batches are all latency: six uploads and a download, one after the other on the stream, cost more than the 4 KB per op
    // they move.  Up to ZERO_COPY_MAX_OPS ops the kernels read the page-locked staging arrays in place and write the results
    // straight into them (hipHostMalloc memory is mapped into the device's address space; a batch of that size crosses PCIe in
    // microseconds either way); larger batches are staged through device memory by DMA.
    const bool zero_copy = n <= ZERO_COPY_MAX_OPS;
    const uint32_t *dk;
    const uint64_t *dm, *dc;
    const uint8_t *d_msgs, *d_ctxs, *d_in;
    uint8_t *d_out;
    int32_t *d_status = nullptr;
    if (zero_copy) {
        dk = kslot; dm = moff; dc = coff;
        d_msgs = t->msgs.p; d_ctxs = t->ctxs.p; d_in = t->in0.p; d_out = t->out0.p;
        d_status = reinterpret_cast<int32_t *>(t->status.p);
    } else {
        BTRY(dev_reserve(ln.d_kslot, n * 4));
        BTRY(dev_reserve(ln.d_moff, (n + 1) * 8));
        BTRY(dev_reserve(ln.d_coff, (n + 1) * 8));
        BTRY(dev_reserve(ln.d_msgs, std::max<size_t>(t->msg_used, 64)));
        BTRY(dev_reserve(ln.d_ctxs, std::max<size_t>(t->ctx_used, 64)));
        BTRY(dev_reserve(ln.d_in0, n * in_len, t->op == OP_SIGN));
        BTRY(dev_reserve(ln.d_out0, n * out_len));
        if (t->op == OP_SIGN) BTRY(dev_reserve(ln.d_status, n * 4));
        BCHECK(hipMemcpyAsync(ln.d_kslot.p, kslot, n * 4, hipMemcpyHostToDevice, st));
        BCHECK(hipMemcpyAsync(ln.d_moff.p, moff, (n + 1) * 8, hipMemcpyHostToDevice, st));
        BCHECK(hipMemcpyAsync(ln.d_coff.p, coff, (n + 1) * 8, hipMemcpyHostToDevice, st));
        if (t->msg_used) BCHECK(hipMemcpyAsync(ln.d_msgs.p, t->msgs.p, t->msg_used, hipMemcpyHostToDevice, st));
        if (t->ctx_used) BCHECK(hipMemcpyAsync(ln.d_ctxs.p, t->ctxs.p, t->ctx_used, hipMemcpyHostToDevice, st));
        BCHECK(hipMemcpyAsync(ln.d_in0.p, t->in0.p, n * in_len, hipMemcpyHostToDevice, st));
        dk = reinterpret_cast<const uint32_t *>(ln.d_kslot.p);
        dm = reinterpret_cast<const uint64_t *>(ln.d_moff.p);
        dc = reinterpret_cast<const uint64_t *>(ln.d_coff.p);
        d_msgs = ln.d_msgs.p; d_ctxs = ln.d_ctxs.p; d_in = ln.d_in0.p; d_out = ln.d_out0.p;
        d_status = reinterpret_cast<int32_t *>(ln.d_status.p);
    }
    if (t->op == OP_VERIFY) {
        BTRY(mldsa_verify_cached_a(ln.ctx, p->set, t->mode, kt.a_hat, kt.tr, kt.f0, kt.cap, dk, d_msgs, dm, d_ctxs, dc, d_in, d_out, n, st));
    } else {
        BTRY(mldsa_sign_cached_a(ln.ctx, p->set, t->mode, kt.a_hat, kt.cap_k, kt.tr, kt.f0, kt.f1, kt.f2, kt.cap, dk, d_msgs, dm, d_ctxs, dc, d_in, d_out,
                                 d_status, n, st));
        if (!zero_copy) {
            BCHECK(hipMemcpyAsync(t->status.p, ln.d_status.p, n * 4, hipMemcpyDeviceToHost, st));
            BCHECK(hipMemsetAsync(ln.d_in0.p, 0, n * 32, st));  // rnd
        }
    }
    if (!zero_copy) BCHECK(hipMemcpyAsync(t->out0.p, ln.d_out0.p, n * out_len, hipMemcpyDeviceToHost, st));
    BCHECK(hipStreamSynchronize(st));
    return MLDSA_OK;
}

void run_batch(mldsa_batcher *b, Lane &ln, Batch *t) {
    int rc;
    if (t->op == OP_KEYGEN) rc = mldsa_keygen_host(ln.ctx, b->p->set, t->in0.p, t->out0.p, t->out1.p, t->n);
    else {
        std::lock_guard<std::mutex> tl(ln.table_mu);  // (mldsa_batcher_forget_key / _flush_keys wait for the batch, not the other way round)
        rc = run_keyed(b, ln, t);
        if (rc != MLDSA_OK) { const char *e = mldsa_last_error(); t->err = e ? e : ""; }
        DeviceGuard dg(mldsa_ctx_device(ln.ctx));
        if (rc != MLDSA_OK && ln.stream) (void)hipStreamSynchronize(ln.stream);
        if (t->op == OP_SIGN) {
            // a staged (n > 256) batch that failed half-way left before its own clearing of the uploaded rnd values
            if (rc != MLDSA_OK && ln.d_in0.p && ln.stream && hipMemsetAsync(ln.d_in0.p, 0, ln.d_in0.bytes, ln.stream) != hipSuccess) (void)hipGetLastError();
            if (!b->private_key_cache.load(std::memory_order_relaxed)) {  // no private key outlives the batch that used it
                KeyTable &kt = ln.tables[OP_SIGN];
                for (size_t s = 0; s < kt.cap; s++)
                    if (kt.valid[s]) table_drop(b, ln, kt, (uint32_t)s);
            }
            if (ln.stream && hipStreamSynchronize(ln.stream) != hipSuccess) (void)hipGetLastError();
        }
    }
    t->rc = rc;
    if (rc != MLDSA_OK && t->err.empty()) { const char *e = mldsa_last_error(); t->err = e ? e : ""; }
    if (t->op != OP_VERIFY) {  // private keys and seeds do not outlive the call (types.rs:19)
        if (t->op == OP_SIGN) wipe_host(t->keys.p, t->n_keys * key_len(b, OP_SIGN));
        wipe_host(t->in0.p, t->n * 32);
    }
}

void dispatcher(mldsa_batcher *b, Lane *ln) {
    Lock lk(b->mu);
    auto sleep_until_poked = [&](const timespec *timeout) {
        b->idle_dispatchers++;
        const uint32_t seen = b->work_seq.load(std::memory_order_relaxed);
        lk.unlock();
        futex_wait(&b->work_seq, seen, timeout);  // a poke between the unlock and the wait has changed the word: returns at once
        lk.lock();
        b->idle_dispatchers--;
    };
    for (;;) {
        // What runs next: the oldest SEALED batch (full, or closed by a request of another mode) if there is one -- it is runnable
        // now, whatever open batch of another operation may be older and still inside its max_wait_us window -- else the oldest
        // open batch that has requests.
        Batch *t = nullptr, *open_t = nullptr;
        for (int op = 0; op < N_OPS; op++)
            for (auto &c : b->batches[op]) {
                if (c->state == SEALED && (!t || c->first_arrival < t->first_arrival)) t = c.get();
                if (c->state == OPEN && c->n > 0 && (!open_t || c->first_arrival < open_t->first_arrival)) open_t = c.get();
            }
        if (!t) t = open_t;
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
        run_batch(b, *ln, t);
        lk.lock();
        b->stats.batches++;
        b->stats.requests += t->n;
        if (t->n > b->stats.largest_batch) b->stats.largest_batch = t->n;
        b->stats.key_hits += t->key_hits;
        b->stats.keys_expanded += t->keys_expanded;
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
    std::call_once(b->alloc_once[r.op], [&] {
        for (auto &t : b->batches[r.op])
            if (b->alloc_rc[r.op] == MLDSA_OK) b->alloc_rc[r.op] = alloc_batch(b, *t, r.op);
    });
    if (b->alloc_rc[r.op] != MLDSA_OK) return set_error(b->alloc_rc[r.op], "mldsa_batcher: no page-locked memory for the staging arrays");
    const uint64_t kh = r.op == OP_KEYGEN ? 0 : hash_key(b->hash_seed, r.key, key_len(b, r.op));  // before the lock: 0.3 us of arithmetic
    Lock lk(b->mu);
    int err = E_NONE;
    bool wake = false;
    Batch *t = open_batch(b, lk, r.op, r.mode, r.msg_len, err, wake);
    if (!t) {
        lk.unlock();
        if (wake) futex_wake(&b->work_seq, INT_MAX);
        return err == E_QUIT ? set_error(MLDSA_ERR_PARAM, "mldsa_batcher: destroyed while in use")
                             : set_error(MLDSA_ERR_NOMEM, "mldsa_batcher: no page-locked memory for a message of this size");
    }
    const size_t i = t->n++;
    size_t m0 = 0, c0 = 0;
    if (r.op != OP_KEYGEN) {
        reinterpret_cast<uint32_t *>(t->kidx.p)[i] = key_slot(b, t, r.key, kh);
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
    if (wake) futex_wake(&b->work_seq, INT_MAX);
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
    if (rc != MLDSA_OK) {
        // the header's promise for a failed operation: ok = 0 (set by the entry point), an all-zero signature, all-zero keys
        set_error(rc, ("mldsa_batcher: " + t->err).c_str());
        if (r.op == OP_SIGN) std::memset(r.out0, 0, (size_t)p->sig_len);
        if (r.op == OP_KEYGEN) { std::memset(r.out0, 0, (size_t)p->pk_len); wipe_host(r.out1, (size_t)p->sk_len); }
    } else if (r.op == OP_VERIFY) *r.out0 = t->out0.p[i];
    else if (r.op == OP_SIGN) {
        std::memcpy(r.out0, t->out0.p + i * (size_t)p->sig_len, (size_t)p->sig_len);
        const int32_t st = reinterpret_cast<const int32_t *>(t->status.p)[i];
        if (st != MLDSA_OK) rc = set_error(st, st == MLDSA_ERR_CTX_LEN ? "mldsa_batcher_sign: ctx longer than 255 bytes" : "mldsa_batcher_sign: the operation was refused");
    } else {
        std::memcpy(r.out0, t->out0.p + i * (size_t)p->pk_len, (size_t)p->pk_len);
        std::memcpy(r.out1, t->out1.p + i * (size_t)p->sk_len, (size_t)p->sk_len);
    }
    if (t->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // the last caller out returns the batch
        if (t->op == OP_KEYGEN) wipe_host(t->out1.p, t->n * (size_t)p->sk_len);
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

static int batcher_make(const std::vector<mldsa_ctx *> &ctxs, bool own, int set, size_t max_batch, unsigned max_wait_us, size_t cache_keys,
                        mldsa_batcher **out) {
    const mldsa_params *p = params_of(set);
    std::unique_ptr<mldsa_batcher> b(new (std::nothrow) mldsa_batcher());
    if (!b) return set_error(MLDSA_ERR_NOMEM, "mldsa_batcher_create: host allocation failed");
    b->p = p;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    // a batch never evicts its own keys, so a table holds at least one batch's worth; default: 1 024 keys or one batch
    b->cache_keys = std::max(max_batch, cache_keys ? cache_keys : (size_t)1024);
    {   // the seed of the key hash: unpredictable to callers, nothing more (clocks, addresses, the thread)
        const uint64_t t0 = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
        const uint64_t t1 = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
        const uint64_t tid = (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id());
        b->hash_seed = (t0 * 0x9E3779B97F4A7C15ull) ^ (t1 << 32) ^ (uint64_t)(uintptr_t)b.get() ^ (tid << 48) ^ (tid >> 16);
    }
    for (mldsa_ctx *c : ctxs) {
        b->lanes.emplace_back(new Lane());
        b->lanes.back()->ctx = c;
        b->lanes.back()->own_ctx = own;
    }
    for (int op = 0; op < N_OPS; op++)
        for (size_t i = 0; i < ctxs.size() + 2; i++) {  // one running per lane, one filling, one being read out
            b->batches[op].emplace_back(new Batch());
            b->batches[op].back()->op = op;
        }
    for (auto &ln : b->lanes) ln->th = std::thread(dispatcher, b.get(), ln.get());
    *out = b.release();
    return MLDSA_OK;
}

int mldsa_batcher_create(mldsa_ctx *ctx, int set, size_t max_batch, unsigned max_wait_us, size_t cache_keys, mldsa_batcher **out) {
    REQUIRE(out, "mldsa_batcher_create: NULL out");
    *out = nullptr;
    REQUIRE(ctx, "mldsa_batcher_create: NULL context");
    REQUIRE(params_of(set), "mldsa_batcher_create: unknown parameter set");
    REQUIRE(max_batch >= 1 && max_batch <= (1u << 20), "mldsa_batcher_create: max_batch in 1 ... 2^20");
    REQUIRE(cache_keys <= (1u << 22), "mldsa_batcher_create: cache_keys up to 2^22");
    return batcher_make({ctx}, false, set, max_batch, max_wait_us, cache_keys, out);
}

int mldsa_batcher_create_on(const int *device_ids, int n, int set, size_t max_batch, unsigned max_wait_us, size_t cache_keys, mldsa_batcher **out) {
    REQUIRE(out, "mldsa_batcher_create_on: NULL out");
    *out = nullptr;
    REQUIRE(device_ids && n >= 1 && n <= 64, "mldsa_batcher_create_on: 1 ... 64 lanes");
    REQUIRE(params_of(set), "mldsa_batcher_create_on: unknown parameter set");
    REQUIRE(max_batch >= 1 && max_batch <= (1u << 20), "mldsa_batcher_create_on: max_batch in 1 ... 2^20");
    REQUIRE(cache_keys <= (1u << 22), "mldsa_batcher_create_on: cache_keys up to 2^22");
    std::vector<mldsa_ctx *> ctxs;
    int rc = MLDSA_OK;
    for (int i = 0; i < n && rc == MLDSA_OK; i++) {
        mldsa_ctx *c = nullptr;
        rc = mldsa_ctx_create(device_ids[i], &c);
        if (rc == MLDSA_OK) ctxs.push_back(c);
    }
    if (rc == MLDSA_OK) rc = batcher_make(ctxs, true, set, max_batch, max_wait_us, cache_keys, out);
    if (rc != MLDSA_OK)
        for (mldsa_ctx *c : ctxs) mldsa_ctx_destroy(c);
    return rc;
}

int mldsa_batcher_lanes(const mldsa_batcher *b) { return b ? (int)b->lanes.size() : MLDSA_ERR_PARAM; }

void mldsa_batcher_destroy(mldsa_batcher *b) {
    if (!b) return;
    {
        Lock lk(b->mu);
        b->quit = true;  // batches that hold requests still run; new requests are refused
        b->work_seq.fetch_add(1, std::memory_order_release);
        b->free_seq.fetch_add(1, std::memory_order_release);
    }
    futex_wake(&b->work_seq, INT_MAX);
    futex_wake(&b->free_seq, INT_MAX);
    for (auto &ln : b->lanes)
        if (ln->th.joinable()) ln->th.join();
    for (;;) {  // callers still reading their results out
        bool busy = false;
        {
            Lock lk(b->mu);
            for (int op = 0; op < N_OPS; op++)
                for (auto &t : b->batches[op]) busy |= t->state != FREE && !(t->state == OPEN && t->n == 0);
        }
        if (!busy) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    for (auto &lp : b->lanes) {
        Lane &ln = *lp;
        {
            DeviceGuard dg(mldsa_ctx_device(ln.ctx));
            if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
            for (KeyTable &kt : ln.tables) table_free(b, kt);
            for (DevBuf *d : {&ln.d_kslot, &ln.d_moff, &ln.d_coff, &ln.d_msgs, &ln.d_ctxs, &ln.d_out0, &ln.d_status}) dev_release(*d);
            dev_release(ln.d_in0, true);
            dev_release(ln.d_kstage, true);
        }
        if (ln.own_ctx) mldsa_ctx_destroy(ln.ctx);
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

// Key lifetime.  Take `key` (PK_LEN or SK_LEN wire bytes) out of every lane's table / empty the tables / stop keeping private keys
// between batches.  Each waits for the batch a lane is running (the table belongs to it for that long) and returns when the host
// copy of the wire bytes is cleared and the device slot's secret fields are zero.
static int forget_in_lane(mldsa_batcher *b, Lane &ln, int op, const uint8_t *key, size_t kl, uint64_t h) {
    std::lock_guard<std::mutex> tl(ln.table_mu);
    KeyTable &kt = ln.tables[op];
    if (!kt.cap) return MLDSA_OK;
    DeviceGuard dg(mldsa_ctx_device(ln.ctx));
    bool any = false;
    if (key) {
        auto range = kt.index.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (std::memcmp(kt.wire.p + (size_t)it->second * kl, key, kl) == 0) { table_drop(b, ln, kt, it->second); any = true; break; }
    } else {
        for (size_t s = 0; s < kt.cap; s++)
            if (kt.valid[s]) { table_drop(b, ln, kt, (uint32_t)s); any = true; }
    }
    if (any && kt.is_private && ln.stream) BCHECK(hipStreamSynchronize(ln.stream));
    return MLDSA_OK;
}

int mldsa_batcher_forget_key(mldsa_batcher *b, const uint8_t *key, size_t key_len_) {
    REQUIRE(b && key, "mldsa_batcher_forget_key: NULL pointer");
    const int op = key_len_ == (size_t)b->p->sk_len ? OP_SIGN : key_len_ == (size_t)b->p->pk_len ? OP_VERIFY : -1;
    REQUIRE(op >= 0, "mldsa_batcher_forget_key: key_len is neither PK_LEN nor SK_LEN of the batcher's parameter set");
    const uint64_t h = hash_key(b->hash_seed, key, key_len_);
    int rc = MLDSA_OK;
    for (auto &ln : b->lanes) {
        const int r = forget_in_lane(b, *ln, op, key, key_len_, h);
        if (rc == MLDSA_OK) rc = r;
    }
    return rc;
}

int mldsa_batcher_flush_keys(mldsa_batcher *b) {
    REQUIRE(b, "mldsa_batcher_flush_keys: NULL batcher");
    int rc = MLDSA_OK;
    for (auto &ln : b->lanes)
        for (int op : {OP_SIGN, OP_VERIFY}) {
            const int r = forget_in_lane(b, *ln, op, nullptr, 0, 0);
            if (rc == MLDSA_OK) rc = r;
        }
    return rc;
}

int mldsa_batcher_set_private_key_cache(mldsa_batcher *b, int on) {
    REQUIRE(b, "mldsa_batcher_set_private_key_cache: NULL batcher");
    b->private_key_cache.store(on ? 1 : 0, std::memory_order_relaxed);
    if (on) return MLDSA_OK;
    int rc = MLDSA_OK;  // switching it off also drops what is there now
    for (auto &ln : b->lanes) {
        const int r = forget_in_lane(b, *ln, OP_SIGN, nullptr, 0, 0);
        if (rc == MLDSA_OK) rc = r;
    }
    return rc;
}

int mldsa_batcher_get_stats(mldsa_batcher *b, mldsa_batcher_stats *out) {
    REQUIRE(b && out, "mldsa_batcher_get_stats: NULL pointer");
    Lock lk(b->mu);
    *out = b->stats;
    return MLDSA_OK;
}

}  // extern "C"
