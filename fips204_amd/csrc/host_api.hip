// Host-memory entry points of include/mldsa_hip.h (mldsa_verify_host / mldsa_sign_host / mldsa_keygen_host):
// the reference's API works on host slices (src/traits.rs:118-308, 330-362); these calls take host pointers and
// wire-format keys and stream the batch through the device in sub-batches:
//
//     up stream:    H2D(i + 1)            | H2D(i + 2) ...
//     comp stream:             kernels(i) | kernels(i + 1) ...
//     down stream:  D2H(i - 1)            | D2H(i) ...
//
// Three slots of context-owned device buffers (and page-locked bounce buffers for callers whose memory is
// pageable) rotate through the three stages; HIP events order the stages, the host thread only waits when it
// wants a slot back.  The device pointers of a slot never change, so the kernel sequence of a sub-batch is a
// repeated call shape and replays as a hipGraph (pipeline.hip run_op).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ctx.h"

namespace mldsa {

namespace {
constexpr int N_SLOTS = 3;
constexpr size_t HOST_DIRECT_MIN_OPS = 16384, HOST_DIRECT_MAX_OPS = 98304;  // mldsa_sign_host: calls that export round by round
// ops per sub-batch (ctx->opt_host_sub_*): verify is PCIe-bound on the way IN -- small sub-batches keep the pipeline fill
// and drain short.  Sign is compute-bound with its traffic on the way OUT: large sub-batches (its rounds are latency-bound on
// small batches) and a small LAST one, the only download nothing hides.
// Two properties of the platform shape the loops (tools/ubench_queues.hip):
//  * copies are served in the order they were SUBMITTED, across streams: an H2D copy submitted after a D2H copy that still
//    waits for its kernels waits with it -- so the uploads of sub-batch i + 1 are submitted before the download of sub-batch i;
//  * HIP streams share a few hardware queues (handed out 0 1 2 3 3 2 1 0 ...), and two streams on one queue take turns -- so
//    the three streams are probed against each other when the stage is created (streams_serialise).

struct Buf {  // a device buffer with an optional page-locked bounce twin, grown on demand
    uint8_t *dev = nullptr, *pin = nullptr;
    size_t cap = 0, pin_cap = 0;
};

struct Slot {
    Buf sigs, msgs, msg_off, ctxs, ctx_off, key_idx, rnd, out, status, xi, pk, sk, pack;
    // device views of the sub-batch's small inputs (inside `pack`, or the separate buffers above)
    const uint32_t *d_kidx = nullptr;
    const uint64_t *d_moff = nullptr, *d_coff = nullptr;
    const uint8_t *d_msgs = nullptr, *d_ctxs = nullptr;
    hipEvent_t up_done = nullptr, comp_done = nullptr, down_done = nullptr;
    bool busy = false;
    // pending bounce copies back to the caller's pageable memory, done when the slot is reclaimed
    struct Pending { void *user; const void *pin; size_t bytes; } pend[3];
    int n_pend = 0;
};
}  // namespace

struct HostStage {
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;
    Slot slot[N_SLOTS];
    Slot call;                              // verify: the side inputs of a whole call when they fit one pack (only .pack and the views)
    Buf key_bytes;                          // wire-format keys of the call
    Buf k_rho, k_capk, k_tr, k_a, k_b, k_c;  // expanded key fields (pk: rho, tr, t1; sk: rho, K, tr, s1, s2, t0)
    hipEvent_t keys_ready = nullptr;
    int secret_last = 0;  // which buffers held secrets during the last *_host call: 0 none (verify), MLDSA_OP_SIGN, MLDSA_OP_KEYGEN
};

namespace {
#define HCHECK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, #expr, _e);          \
    } while (0)
#define TRY(expr) do { int _rc = (expr); if (_rc != MLDSA_OK) return _rc; } while (0)

// Is [p, p + bytes) page-locked memory the runtime knows, over its WHOLE extent?  The first byte alone does not say so: a caller may
// have registered (hipHostRegister) only the head of an array, or handed a slice that runs off the end of a page-locked slab -- a
// kernel storing through the device view of such a buffer faults, and the process with it.  Both ends must be host memory of one
// mapping, and where the runtime can name the allocation the extent must lie inside it.  Anything else is treated as pageable
// (bounce copies / the sub-batch path): slower, never wrong.
bool is_pinned(const void *p, size_t bytes) {
    if (!p) return true;
    hipPointerAttribute_t a, b;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime: clear the sticky error
        return false;
    }
    if (a.type != hipMemoryTypeHost) return false;
    if (bytes <= 1) return true;
    const char *last = static_cast<const char *>(p) + (bytes - 1);
    if (hipPointerGetAttributes(&b, last) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (b.type != hipMemoryTypeHost) return false;
    if (static_cast<const char *>(b.devicePointer) - static_cast<const char *>(a.devicePointer) != (ptrdiff_t)(bytes - 1)) return false;
    void *base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, const_cast<void *>(p)) == hipSuccess && base && size)
        return static_cast<const char *>(p) >= static_cast<const char *>(base) && last < static_cast<const char *>(base) + size;
    (void)hipGetLastError();
    return true;  // the runtime cannot name the allocation: both ends are mapped host memory of one device view
}

int grow_dev(Buf &b, size_t bytes) {
    if (b.cap >= bytes) return MLDSA_OK;
    if (b.dev) { HCHECK(device_sync_quiesced()); MLDSA_WIPE(memset_quiesced(b.dev, 0, b.cap)); HCHECK(free_quiesced(b.dev)); b.dev = nullptr; b.cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (malloc_quiesced((void **)&b.dev, want) != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "host path: device staging allocation");
    b.cap = want;
    return MLDSA_OK;
}

int grow_pin(Buf &b, size_t bytes) {
    if (b.pin_cap >= bytes) return MLDSA_OK;
    if (b.pin) { HCHECK(device_sync_quiesced()); MLDSA_WIPE(wipe_host(b.pin, b.pin_cap)); HCHECK(host_free_quiesced(b.pin)); b.pin = nullptr; b.pin_cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (host_malloc_quiesced((void **)&b.pin, want) != hipSuccess) return set_error(MLDSA_ERR_NOMEM, "host path: page-locked staging allocation");
    b.pin_cap = want;
    return MLDSA_OK;
}

// host -> device on `st`: DMA straight from page-locked caller memory, through the bounce buffer otherwise
int upload(Buf &b, const void *user, size_t bytes, bool pinned, hipStream_t st) {
    if (bytes == 0) return grow_dev(b, 1);
    TRY(grow_dev(b, bytes));
    const void *src = user;
    if (!pinned) {
        TRY(grow_pin(b, bytes));
        memcpy(b.pin, user, bytes);
        src = b.pin;
    }
    HCHECK(hipMemcpyAsync(b.dev, src, bytes, hipMemcpyHostToDevice, st));
    return MLDSA_OK;
}

// device -> host on `st`; pageable destinations are filled from the bounce buffer when the slot is reclaimed
int download(Slot &sl, Buf &b, void *user, size_t bytes, bool pinned, hipStream_t st) {
    if (bytes == 0) return MLDSA_OK;
    void *dst = user;
    if (!pinned) {
        TRY(grow_pin(b, bytes));
        dst = b.pin;
        sl.pend[sl.n_pend++] = {user, b.pin, bytes};
    }
    HCHECK(hipMemcpyAsync(dst, b.dev, bytes, hipMemcpyDeviceToHost, st));
    return MLDSA_OK;
}

int reclaim(Slot &sl) {
    if (!sl.busy) return MLDSA_OK;
    HCHECK(hipEventSynchronize(sl.down_done));
    for (int i = 0; i < sl.n_pend; i++) memcpy(sl.pend[i].user, sl.pend[i].pin, sl.pend[i].bytes);
    sl.n_pend = 0;
    sl.busy = false;
    return MLDSA_OK;
}

int stage_get(mldsa_ctx *ctx, HostStage **out) {
    if (!ctx->host_stage) {
        HostStage *hs = new (std::nothrow) HostStage();
        if (!hs) return set_error(MLDSA_ERR_NOMEM, "host path: allocation failed");
        // Three streams that really run side by side: HIP hands its few hardware queues out in the order 0 1 2 3 3 2 1 0 ...,
        // so two streams created one after the other can land on ONE queue and then take turns (measured: with `comp` and
        // `down` on one queue the download of sub-batch i ran to the end before sub-batch i + 1 started signing).  Candidates
        // are probed against each other (ctx.h streams_serialise); the ones not chosen are released again.
        hipError_t e = hipStreamCreateWithFlags(&hs->comp, hipStreamNonBlocking);
        std::vector<hipStream_t> spare;
        for (int tries = 0; e == hipSuccess && tries < 10 && !(hs->up && hs->down); tries++) {
            hipStream_t c = nullptr;
            e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
            if (e != hipSuccess) break;
            const bool beside_comp = !streams_serialise(ctx, hs->comp, c);
            if (beside_comp && !hs->up) hs->up = c;
            else if (beside_comp && !hs->down && !streams_serialise(ctx, hs->up, c)) hs->down = c;
            else spare.push_back(c);
        }
        // fewer queues than hoped for: correctness does not depend on the choice
        for (hipStream_t *want : {&hs->up, &hs->down})
            if (e == hipSuccess && !*want) {
                if (!spare.empty()) { *want = spare.back(); spare.pop_back(); }
                else e = hipStreamCreateWithFlags(want, hipStreamNonBlocking);
            }
        for (hipStream_t c : spare) (void)hipStreamDestroy(c);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&hs->keys_ready, hipEventDisableTiming);
        for (auto &sl : hs->slot) {
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.up_done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.comp_done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.down_done, hipEventDisableTiming);
        }
        ctx->host_stage = hs;
        if (e != hipSuccess) return set_error(MLDSA_ERR_DEVICE, "host path: stream / event setup", e);
    }
    *out = ctx->host_stage;
    return MLDSA_OK;
}

// secrets must not outlive the call in the staging buffers (the reference zeroizes on drop, types.rs:19)
void wipe_buf(Buf &b, hipStream_t st) {
    if (b.dev) MLDSA_WIPE(hipMemsetAsync(b.dev, 0, b.cap, st));
    if (b.pin) MLDSA_WIPE(wipe_host(b.pin, b.pin_cap));
}

void free_buf(Buf &b) {
    if (b.dev) { MLDSA_WIPE(memset_quiesced(b.dev, 0, b.cap)); (void)free_quiesced(b.dev); }
    if (b.pin) { MLDSA_WIPE(wipe_host(b.pin, b.pin_cap)); (void)host_free_quiesced(b.pin); }
    b = Buf();
}

// bytes of the [a, b) slice of a concatenated byte-string array (the tables were validated: check_tables)
inline size_t span(const uint64_t *off, size_t a, size_t b) { return (size_t)(off[b] - off[a]); }
}  // namespace

// The host paths memcpy by the caller's offsets: one O(n) pass makes sure they are non-decreasing before anything is copied or
// uploaded (a decreasing pair would be a ~2^64-byte copy); a table that names bytes of a NULL array is refused as well.
int check_tables(const char *who, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, size_t n_ops) {
    int rc = mldsa_check_offsets(msg_off, n_ops);
    if (rc == MLDSA_OK && ctx_off) rc = mldsa_check_offsets(ctx_off, n_ops);
    if (rc != MLDSA_OK) {
        char msg[200];
        snprintf(msg, sizeof(msg), "%s: %s", who, mldsa_last_error());
        return set_error(rc, msg);
    }
    if ((!msgs && msg_off[n_ops] != msg_off[0]) || (ctx_off && !ctxs && ctx_off[n_ops] != ctx_off[0])) {
        char msg[200];
        snprintf(msg, sizeof(msg), "%s: the offsets name bytes of a NULL array", who);
        return set_error(MLDSA_ERR_PARAM, msg);
    }
    return MLDSA_OK;
}

void host_stage_destroy(mldsa_ctx *ctx) {
    HostStage *hs = ctx->host_stage;
    if (!hs) return;
    free_buf(hs->call.pack);
    for (auto &sl : hs->slot) {
        for (Buf *b : {&sl.sigs, &sl.msgs, &sl.msg_off, &sl.ctxs, &sl.ctx_off, &sl.key_idx, &sl.rnd, &sl.out, &sl.status, &sl.xi, &sl.pk, &sl.sk, &sl.pack})
            free_buf(*b);
        if (sl.up_done) (void)hipEventDestroy(sl.up_done);
        if (sl.comp_done) (void)hipEventDestroy(sl.comp_done);
        if (sl.down_done) (void)hipEventDestroy(sl.down_done);
    }
    for (Buf *b : {&hs->key_bytes, &hs->k_rho, &hs->k_capk, &hs->k_tr, &hs->k_a, &hs->k_b, &hs->k_c}) free_buf(*b);
    if (hs->keys_ready) (void)hipEventDestroy(hs->keys_ready);
    if (hs->up) (void)hipStreamDestroy(hs->up);
    if (hs->comp) (void)hipStreamDestroy(hs->comp);
    if (hs->down) (void)hipStreamDestroy(hs->down);
    delete hs;
    ctx->host_stage = nullptr;
}

// mldsa_debug_secret_residue: the staging buffers that held secrets during the last *_host call -- private keys (wire bytes and
// expanded fields) and rnd of a signing call, seeds and private keys of a key-generation call; device and page-locked twins
int host_stage_residue(mldsa_ctx *ctx, size_t *scanned, size_t *nonzero) {
    HostStage *hs = ctx->host_stage;
    if (!hs || hs->secret_last == 0) return MLDSA_OK;
    std::vector<Buf *> bufs;
    if (hs->secret_last == MLDSA_OP_SIGN) {
        for (Buf *b : {&hs->key_bytes, &hs->k_capk, &hs->k_a, &hs->k_b, &hs->k_c}) bufs.push_back(b);
        for (auto &sl : hs->slot) bufs.push_back(&sl.rnd);
    } else {
        for (auto &sl : hs->slot) { bufs.push_back(&sl.xi); bufs.push_back(&sl.sk); }
    }
    for (Buf *b : bufs) {
        if (b->dev) {
            size_t nz = 0;
            TRY(count_nonzero_dev(b->dev, b->cap, &nz));
            *scanned += b->cap;
            *nonzero += nz;
        }
        if (b->pin) {
            for (size_t i = 0; i < b->pin_cap; i++) *nonzero += b->pin[i] != 0;
            *scanned += b->pin_cap;
        }
    }
    return MLDSA_OK;
}

// shared by verify_host and sign_host: upload the per-op inputs of ops [a, b) into slot `sl` on the up stream
struct OpInputs {
    const uint32_t *key_idx;
    const uint8_t *msgs;
    const uint64_t *msg_off;
    const uint8_t *ctxs;
    const uint64_t *ctx_off;
    bool pin_kidx, pin_msgs, pin_moff, pin_ctxs, pin_coff;
};

// The per-op side inputs (key indices, offsets, messages, ctxs) are a few hundred KB per sub-batch in 3-5 arrays; every
// separate DMA costs ~15-20 us of link time whatever its size.  They are gathered into ONE page-locked buffer on the
// host (a sub-100-us memcpy that overlaps the previous sub-batch's DMA) and go up in one copy; only the bulk array
// (signatures / rnd) is copied straight from the caller's memory.
constexpr size_t PACK_LIMIT = 8u << 20;

static int upload_op_inputs(hipStream_t up, Slot &sl, const OpInputs &in, size_t a, size_t b) {
    const size_t n = b - a;
    const size_t kb = in.key_idx ? n * 4 : 0, mob = (n + 1) * 8, mb = in.msgs ? span(in.msg_off, a, b) : 0;
    const size_t cob = in.ctx_off ? (n + 1) * 8 : 0, cb = (in.ctx_off && in.ctxs) ? span(in.ctx_off, a, b) : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_k = 0, o_mo = al(kb), o_m = o_mo + al(mob), o_co = o_m + al(mb + 8), o_c = o_co + al(cob), total = o_c + al(cb + 8);
    if (total <= PACK_LIMIT) {
        TRY(grow_dev(sl.pack, total));
        TRY(grow_pin(sl.pack, total));
        uint8_t *h = sl.pack.pin;
        if (kb) memcpy(h + o_k, in.key_idx + a, kb);
        memcpy(h + o_mo, in.msg_off + a, mob);
        if (mb) memcpy(h + o_m, in.msgs + in.msg_off[a], mb);
        if (cob) memcpy(h + o_co, in.ctx_off + a, cob);
        if (cb) memcpy(h + o_c, in.ctxs + in.ctx_off[a], cb);
        HCHECK(hipMemcpyAsync(sl.pack.dev, h, total, hipMemcpyHostToDevice, up));
        uint8_t *d = sl.pack.dev;
        sl.d_kidx = kb ? reinterpret_cast<const uint32_t *>(d + o_k) : nullptr;
        sl.d_moff = reinterpret_cast<const uint64_t *>(d + o_mo);
        sl.d_msgs = d + o_m;
        sl.d_coff = cob ? reinterpret_cast<const uint64_t *>(d + o_co) : nullptr;
        sl.d_ctxs = d + o_c;
        return MLDSA_OK;
    }
    if (in.key_idx) TRY(upload(sl.key_idx, in.key_idx + a, kb, in.pin_kidx, up));
    TRY(upload(sl.msg_off, in.msg_off + a, mob, in.pin_moff, up));
    TRY(upload(sl.msgs, in.msgs ? in.msgs + in.msg_off[a] : nullptr, mb, in.pin_msgs, up));
    if (in.ctx_off) {
        TRY(upload(sl.ctx_off, in.ctx_off + a, cob, in.pin_coff, up));
        TRY(upload(sl.ctxs, in.ctxs ? in.ctxs + in.ctx_off[a] : nullptr, cb, in.pin_ctxs, up));
    }
    sl.d_kidx = in.key_idx ? reinterpret_cast<const uint32_t *>(sl.key_idx.dev) : nullptr;
    sl.d_moff = reinterpret_cast<const uint64_t *>(sl.msg_off.dev);
    sl.d_msgs = sl.msgs.dev;
    sl.d_coff = in.ctx_off ? reinterpret_cast<const uint64_t *>(sl.ctx_off.dev) : nullptr;
    sl.d_ctxs = in.ctx_off ? sl.ctxs.dev : nullptr;
    return MLDSA_OK;
}

}  // namespace mldsa

using namespace mldsa;

#define REQUIRE(cond, msg) \
    do { if (!(cond)) return set_error(MLDSA_ERR_PARAM, msg); } while (0)

extern "C" {

// declared in capi.hip's translation unit as ordinary exports
int mldsa_verify_host(mldsa_ctx *ctx, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx,
                      const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                      const uint8_t *sigs, uint8_t *ok, size_t n_ops) {
    REQUIRE(ctx, "mldsa_verify_host: NULL context");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_verify_host: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_verify_host: bad mode");
    if (n_ops == 0) return MLDSA_OK;
    REQUIRE(pk && msg_off && sigs && ok, "mldsa_verify_host: NULL pointer");
    REQUIRE(key_idx ? n_keys > 0 : n_keys >= n_ops, "mldsa_verify_host: n_keys does not cover the batch");
    TRY(check_tables("mldsa_verify_host", msgs, msg_off, ctxs, ctx_off, n_ops));
    DeviceGuard dg(ctx->device);
    std::lock_guard<std::mutex> host_lk(ctx->host_mutex);
    HostStage *hs;
    TRY(stage_get(ctx, &hs));
    hs->secret_last = 0;
    const size_t pkl = (size_t)p->pk_len, sgl = (size_t)p->sig_len, k = (size_t)p->k;
    // keys: upload once, expand once (PublicKey::try_from_bytes, ml_dsa.rs:477-498)
    TRY(upload(hs->key_bytes, pk, n_keys * pkl, is_pinned(pk, n_keys * pkl), hs->up));
    TRY(grow_dev(hs->k_rho, n_keys * 32));
    TRY(grow_dev(hs->k_tr, n_keys * 64));
    TRY(grow_dev(hs->k_a, n_keys * k * 1024));
    HCHECK(hipEventRecord(hs->keys_ready, hs->up));
    HCHECK(hipStreamWaitEvent(hs->comp, hs->keys_ready, 0));
    TRY(mldsa_pk_expand(ctx, set, hs->key_bytes.dev, hs->k_rho.dev, hs->k_tr.dev, reinterpret_cast<int32_t *>(hs->k_a.dev), n_keys, hs->comp));
    const size_t sub = std::min(n_ops, (size_t)ctx->opt_host_sub_verify);
    TRY(mldsa_reserve(ctx, set, MLDSA_OP_VERIFY, sub));
    OpInputs in{key_idx, msgs, msg_off, ctxs, ctx_off, is_pinned(key_idx, n_ops * 4), is_pinned(msgs ? msgs + msg_off[0] : nullptr, span(msg_off, 0, n_ops)), is_pinned(msg_off, (n_ops + 1) * 8),
                is_pinned(ctxs && ctx_off ? ctxs + ctx_off[0] : nullptr, ctx_off ? span(ctx_off, 0, n_ops) : 0), is_pinned(ctx_off, (n_ops + 1) * 8)};
    const bool pin_sigs = is_pinned(sigs, n_ops * sgl), pin_ok = is_pinned(ok, n_ops);
    int rc = MLDSA_OK;
    // Side inputs (key indices, offsets, messages, ctxs): when the whole call's fit one pack they go up ONCE, right after
    // the first sub-batch's signatures (the host-side gather then overlaps that DMA); a separate small DMA per
    // sub-batch idles the link for ~30 us out of every ~520.
    const size_t side_bytes = n_ops * 12 + 8 + span(msg_off, 0, n_ops) + (ctx_off ? n_ops * 8 + 8 + span(ctx_off, 0, n_ops) : 0);
    const bool whole = side_bytes + 2048 <= PACK_LIMIT;
    // Copies are served in the order they were SUBMITTED, across streams (tools/ubench_queues.hip: an H2D copy submitted
    // after a D2H copy that still waits for its kernels waits with it).  So the uploads of sub-batch i + 1 are submitted
    // BEFORE the download of sub-batch i.
    const size_t n_sub = (n_ops + sub - 1) / sub;
    auto stage_up = [&](size_t i) -> int {
        const size_t a = i * sub, b = std::min(n_ops, a + sub), n = b - a;
        Slot &sl = hs->slot[i % N_SLOTS];
        TRY(reclaim(sl));
        if (!whole) TRY(upload_op_inputs(hs->up, sl, in, a, b));
        TRY(upload(sl.sigs, sigs + a * sgl, n * sgl, pin_sigs, hs->up));
        if (whole && i == 0) TRY(upload_op_inputs(hs->up, hs->call, in, 0, n_ops));
        TRY(grow_dev(sl.out, n));
        HCHECK(hipEventRecord(sl.up_done, hs->up));
        return MLDSA_OK;
    };
    auto run = [&](size_t i) -> int {
        const size_t a = i * sub, b = std::min(n_ops, a + sub), n = b - a;
        Slot &sl = hs->slot[i % N_SLOTS];
        HCHECK(hipStreamWaitEvent(hs->comp, sl.up_done, 0));
        // the slice's byte strings start at offset msg_off[v0] of the caller's array: hand the kernels a base
        // pointer that makes the caller's own offsets land in the staging buffer
        const Slot &v = whole ? hs->call : sl;
        const size_t v0 = whole ? 0 : a, vo = a - v0;  // first op of the view, this sub-batch's position in it
        const uint32_t *d_kidx = v.d_kidx ? v.d_kidx + vo : nullptr;
        const uint64_t *d_moff = v.d_moff + vo, *d_coff = v.d_coff ? v.d_coff + vo : nullptr;
        const uint8_t *mbase = v.d_msgs - msg_off[v0];
        const uint8_t *cbase = ctx_off ? v.d_ctxs - ctx_off[v0] : nullptr;
        const size_t kb = key_idx ? 0 : a;  // identity mapping walks the key table with the batch
        TRY(mldsa_verify(ctx, set, mode, hs->k_rho.dev + kb * 32, hs->k_tr.dev + kb * 64,
                         reinterpret_cast<const int32_t *>(hs->k_a.dev) + kb * k * 256, n_keys - kb, d_kidx, mbase, d_moff, cbase,
                         d_coff, sl.sigs.dev, sl.out.dev, n, hs->comp));
        HCHECK(hipEventRecord(sl.comp_done, hs->comp));
        HCHECK(hipStreamWaitEvent(hs->down, sl.comp_done, 0));
        TRY(download(sl, sl.out, ok + a, n, pin_ok, hs->down));
        HCHECK(hipEventRecord(sl.down_done, hs->down));
        sl.busy = true;
        return MLDSA_OK;
    };
    rc = stage_up(0);
    for (size_t i = 0; i < n_sub && rc == MLDSA_OK; i++) {
        if (i + 1 < n_sub) rc = stage_up(i + 1);
        if (rc == MLDSA_OK) rc = run(i);
    }
    for (auto &sl : hs->slot) {
        const int r2 = reclaim(sl);
        if (rc == MLDSA_OK) rc = r2;
    }
    if (rc != MLDSA_OK) (void)device_sync_quiesced();
    return rc;
}

int mldsa_sign_host(mldsa_ctx *ctx, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx,
                    const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                    const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops) {
    REQUIRE(ctx, "mldsa_sign_host: NULL context");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_sign_host: unknown parameter set");
    REQUIRE(mode == MLDSA_MODE_PURE || mode == MLDSA_MODE_INTERNAL || mode == MLDSA_MODE_PREHASH, "mldsa_sign_host: bad mode");
    if (n_ops == 0) return MLDSA_OK;
    REQUIRE(sk && msg_off && rnd && sigs, "mldsa_sign_host: NULL pointer");
    REQUIRE(key_idx ? n_keys > 0 : n_keys >= n_ops, "mldsa_sign_host: n_keys does not cover the batch");
    TRY(check_tables("mldsa_sign_host", msgs, msg_off, ctxs, ctx_off, n_ops));
    DeviceGuard dg(ctx->device);
    std::lock_guard<std::mutex> host_lk(ctx->host_mutex);
    HostStage *hs;
    TRY(stage_get(ctx, &hs));
    hs->secret_last = MLDSA_OP_SIGN;
    // the body runs in a lambda so that EVERY way out -- early error returns included -- passes the clearing below
    const int rc_all = [&]() -> int {
    const size_t skl = (size_t)p->sk_len, sgl = (size_t)p->sig_len, k = (size_t)p->k, l = (size_t)p->l;
    // Page-locked signature buffer: the signing rounds write finished signatures straight into it (k_export_done after every
    // round, pipeline.hip) -- the 3.3 KB per signature cross PCIe while later rounds still run, so the whole batch is ONE signing
    // call (its rounds' fixed costs paid once) with nothing left to download but the statuses.  Pageable buffers, or a runtime
    // that cannot map the buffer, take the sub-batch path below.
    // Used for calls of 16 385 ... 98 304 ops (measured, ML-DSA-65: 5.7 instead of 7.3 ms at 32 768, 10.3 instead of 11.1 ms at 65 536, 20.5 instead of 19.6 ms at 131 072; smaller calls replay as
    // hipGraphs on the sub-batch path, larger ones amortise their rounds anyway and lose more to the export's interference).
    // (One lane only: the export hangs off lane 0's rounds.)
    const bool pin_sigs = is_pinned(sigs, n_ops * sgl);  // the WHOLE array: k_export_done stores across all of it
    uint8_t *sigs_dev_view = nullptr;
    const bool direct_size = n_ops > HOST_DIRECT_MIN_OPS && n_ops <= HOST_DIRECT_MAX_OPS;
    if (pin_sigs && direct_size && ctx->opt_host_direct && ctx->opt_sign_lanes < 2) {
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&sigs_dev_view), sigs, 0) != hipSuccess) {
            (void)hipGetLastError();
            sigs_dev_view = nullptr;
        }
    }
    // the direct path's signing call starts with ExpandA as soon as the keys and key_idx are there, the other inputs follow
    // beside it: key_idx goes up first, on its own
    const uint32_t *d_kidx_early = nullptr;
    if (sigs_dev_view && key_idx) {
        Slot &sl = hs->slot[0];
        TRY(reclaim(sl));
        TRY(upload(sl.key_idx, key_idx, n_ops * 4, is_pinned(key_idx, n_ops * 4), hs->up));
        d_kidx_early = reinterpret_cast<const uint32_t *>(sl.key_idx.dev);
    }
    // keys: upload once, expand once (PrivateKey::try_from_bytes, ml_dsa.rs:445-469)
    TRY(upload(hs->key_bytes, sk, n_keys * skl, is_pinned(sk, n_keys * skl), hs->up));
    TRY(grow_dev(hs->k_rho, n_keys * 32));
    TRY(grow_dev(hs->k_capk, n_keys * 32));
    TRY(grow_dev(hs->k_tr, n_keys * 64));
    TRY(grow_dev(hs->k_a, n_keys * l * 1024));
    TRY(grow_dev(hs->k_b, n_keys * k * 1024));
    TRY(grow_dev(hs->k_c, n_keys * k * 1024));
    HCHECK(hipEventRecord(hs->keys_ready, hs->up));
    HCHECK(hipStreamWaitEvent(hs->comp, hs->keys_ready, 0));
    int32_t *s1 = reinterpret_cast<int32_t *>(hs->k_a.dev), *s2 = reinterpret_cast<int32_t *>(hs->k_b.dev),
            *t0 = reinterpret_cast<int32_t *>(hs->k_c.dev);
    TRY(mldsa_sk_expand(ctx, set, hs->key_bytes.dev, hs->k_rho.dev, hs->k_capk.dev, hs->k_tr.dev, s1, s2, t0, n_keys, hs->comp));
    // Sub-batches: signing is compute-bound and its rounds are latency-bound on small batches, so they are as large as the
    // pipelines' chunk -- except the LAST one, whose signatures come down with nothing left to hide the transfer behind:
    // ctx->opt_host_sub_sign ops (49 152 + 16 384 for a 65 536-op call: 11.1 ms against 13.4 ms in one piece)
    std::vector<size_t> cut{0};
    {
        const size_t big = 65536, tail = std::max<size_t>(64, (size_t)ctx->opt_host_sub_sign);
        for (size_t rem = n_ops; rem > 0;) {
            const size_t take = rem > big + tail ? big : rem > 2 * tail ? rem - tail : rem;
            cut.push_back(cut.back() + take);
            rem -= take;
        }
    }
    {
        size_t largest = 0;  // a sub-batch can be larger than `big` (rem <= big + tail goes as rem - tail)
        for (size_t j = 0; j + 1 < cut.size(); j++) largest = std::max(largest, cut[j + 1] - cut[j]);
        if (sigs_dev_view) largest = n_ops;  // one signing call for the whole batch (see above)
        TRY(mldsa_reserve(ctx, set, MLDSA_OP_SIGN, largest));
    }
    OpInputs in{key_idx, msgs, msg_off, ctxs, ctx_off, is_pinned(key_idx, n_ops * 4), is_pinned(msgs ? msgs + msg_off[0] : nullptr, span(msg_off, 0, n_ops)), is_pinned(msg_off, (n_ops + 1) * 8),
                is_pinned(ctxs && ctx_off ? ctxs + ctx_off[0] : nullptr, ctx_off ? span(ctx_off, 0, n_ops) : 0), is_pinned(ctx_off, (n_ops + 1) * 8)};
    const bool pin_rnd = is_pinned(rnd, n_ops * 32);
    // per-op status always comes back: MLDSA_ERR_AGAIN marks the (practically never) ops that need another pass
    std::vector<int32_t> st_local;
    int32_t *st = status;
    if (!st) { st_local.resize(n_ops); st = st_local.data(); }
    const bool pin_st = status && is_pinned(status, n_ops * 4);
    int rc = MLDSA_OK;
    // the calls below cannot wait for the device, but an op they leave unfinished is signed again further down: plan the
    // rounds like a synchronous call (three or four empty ~0.2 ms rounds less per sub-batch than the 1e-9 plan).  Passed per
    // call: the context's own threshold stays what mldsa_sign_async callers on other threads configured.
    constexpr double HOST_PLAN_STOP = 0.05;
    if (sigs_dev_view) {
        Slot &sl = hs->slot[0];
        rc = [&]() -> int {
            TRY(reclaim(sl));
            OpInputs rest = in;  // key_idx is up already
            rest.key_idx = nullptr;
            TRY(upload_op_inputs(hs->up, sl, rest, 0, n_ops));
            sl.d_kidx = d_kidx_early;
            TRY(upload(sl.rnd, rnd, n_ops * 32, pin_rnd, hs->up));
            TRY(grow_dev(sl.out, n_ops * sgl));
            TRY(grow_dev(sl.status, n_ops * 4));
            HCHECK(hipEventRecord(sl.up_done, hs->up));  // the signing call waits for it itself, after its ExpandA
            const uint8_t *mbase = sl.d_msgs - msg_off[0];
            const uint8_t *cbase = ctx_off ? sl.d_ctxs - ctx_off[0] : nullptr;
            TRY(sign_call(ctx, set, mode, hs->k_rho.dev, nullptr, hs->k_capk.dev, hs->k_tr.dev, s1, s2, t0, n_keys, sl.d_kidx, mbase, sl.d_moff,
                          cbase, sl.d_coff, sl.rnd.dev, sl.out.dev, reinterpret_cast<int32_t *>(sl.status.dev), n_ops, hs->comp, true,
                          HOST_PLAN_STOP, sigs_dev_view, sl.up_done));
            HCHECK(hipMemcpyAsync(st, sl.status.dev, n_ops * 4, hipMemcpyDeviceToHost, hs->comp));
            HCHECK(hipStreamSynchronize(hs->comp));
            // a refused op (ctx too long, key index out of range) never entered a round: its signature is all zero
            for (size_t op = 0; op < n_ops; op++)
                if (st[op] != MLDSA_OK) memset(sigs + op * sgl, 0, sgl);
            return MLDSA_OK;
        }();
    }
    // uploads of sub-batch i + 1 are submitted before the download of sub-batch i (copies are served in submission order,
    // see mldsa_verify_host): otherwise the next sub-batch's few KB of inputs sit behind 100 MB of signatures that are not
    // even signed yet, and signing waits for both
    const size_t n_sub = sigs_dev_view ? 0 : cut.size() - 1;
    auto stage_up = [&](size_t j) -> int {
        const size_t a = cut[j], b = cut[j + 1], n = b - a;
        Slot &sl = hs->slot[j % N_SLOTS];
        TRY(reclaim(sl));
        TRY(upload_op_inputs(hs->up, sl, in, a, b));
        TRY(upload(sl.rnd, rnd + a * 32, n * 32, pin_rnd, hs->up));
        TRY(grow_dev(sl.out, n * sgl));
        TRY(grow_dev(sl.status, n * 4));
        HCHECK(hipEventRecord(sl.up_done, hs->up));
        return MLDSA_OK;
    };
    auto run = [&](size_t j) -> int {
        const size_t a = cut[j], n = cut[j + 1] - a;
        Slot &sl = hs->slot[j % N_SLOTS];
        HCHECK(hipStreamWaitEvent(hs->comp, sl.up_done, 0));
        const uint8_t *mbase = sl.d_msgs - msg_off[a];
        const uint8_t *cbase = ctx_off ? sl.d_ctxs - ctx_off[a] : nullptr;
        const size_t kb = key_idx ? 0 : a;
        TRY(sign_call(ctx, set, mode, hs->k_rho.dev + kb * 32, nullptr, hs->k_capk.dev + kb * 32, hs->k_tr.dev + kb * 64, s1 + kb * l * 256,
                      s2 + kb * k * 256, t0 + kb * k * 256, n_keys - kb, sl.d_kidx, mbase, sl.d_moff, cbase, sl.d_coff, sl.rnd.dev,
                      sl.out.dev, reinterpret_cast<int32_t *>(sl.status.dev), n, hs->comp, true, HOST_PLAN_STOP));
        HCHECK(hipEventRecord(sl.comp_done, hs->comp));
        HCHECK(hipStreamWaitEvent(hs->down, sl.comp_done, 0));
        TRY(download(sl, sl.out, sigs + a * sgl, n * sgl, pin_sigs, hs->down));
        TRY(download(sl, sl.status, st + a, n * 4, pin_st, hs->down));
        HCHECK(hipEventRecord(sl.down_done, hs->down));
        sl.busy = true;
        return MLDSA_OK;
    };
    if (n_sub) rc = stage_up(0);
    for (size_t j = 0; j < n_sub && rc == MLDSA_OK; j++) {
        if (j + 1 < n_sub) rc = stage_up(j + 1);
        if (rc == MLDSA_OK) rc = run(j);
    }
    for (auto &sl : hs->slot) {
        const int r2 = reclaim(sl);
        if (rc == MLDSA_OK) rc = r2;
    }
    if (rc != MLDSA_OK) return rc;
    // ops the enqueued rounds left unfinished (probability < 1e-9 per call): sign them again, waiting this time
    for (size_t op = 0; op < n_ops && rc == MLDSA_OK; op++) {
        if (st[op] != MLDSA_ERR_AGAIN) continue;
        Slot &sl = hs->slot[0];
        const uint32_t ki = key_idx ? key_idx[op] : (uint32_t)op;
        const uint64_t mo[2] = {0, msg_off[op + 1] - msg_off[op]};
        const uint64_t co[2] = {0, ctx_off ? ctx_off[op + 1] - ctx_off[op] : 0};
        rc = [&]() -> int {
            TRY(upload(sl.msg_off, mo, 16, false, hs->comp));
            TRY(upload(sl.ctx_off, co, 16, false, hs->comp));
            TRY(upload(sl.msgs, msgs ? msgs + msg_off[op] : nullptr, msgs ? (size_t)mo[1] : 0, false, hs->comp));
            TRY(upload(sl.ctxs, ctxs && ctx_off ? ctxs + ctx_off[op] : nullptr, ctxs && ctx_off ? (size_t)co[1] : 0, false, hs->comp));
            TRY(upload(sl.rnd, rnd + op * 32, 32, false, hs->comp));
            TRY(mldsa_sign(ctx, set, mode, hs->k_rho.dev + ki * 32, hs->k_capk.dev + ki * 32, hs->k_tr.dev + ki * 64, s1 + ki * l * 256,
                           s2 + ki * k * 256, t0 + ki * k * 256, 1, nullptr, sl.msgs.dev, reinterpret_cast<const uint64_t *>(sl.msg_off.dev),
                           sl.ctxs.dev, reinterpret_cast<const uint64_t *>(sl.ctx_off.dev), sl.rnd.dev, sl.out.dev,
                           reinterpret_cast<int32_t *>(sl.status.dev), 1, hs->comp));
            HCHECK(hipMemcpyAsync(sigs + op * sgl, sl.out.dev, sgl, hipMemcpyDeviceToHost, hs->comp));
            HCHECK(hipMemcpyAsync(st + op, sl.status.dev, 4, hipMemcpyDeviceToHost, hs->comp));
            HCHECK(hipStreamSynchronize(hs->comp));
            return MLDSA_OK;
        }();
    }
    return rc;
    }();
    // the private keys (wire bytes and expanded fields) and the per-signature randomness leave the staging buffers
    // (the reference zeroizes on drop, types.rs:19)
    if (rc_all != MLDSA_OK) (void)device_sync_quiesced();  // copies of a failed call may still be in flight
    for (Buf *b : {&hs->key_bytes, &hs->k_capk, &hs->k_a, &hs->k_b, &hs->k_c}) wipe_buf(*b, hs->comp);
    for (auto &sl : hs->slot) wipe_buf(sl.rnd, hs->comp);
    (void)hipStreamSynchronize(hs->comp);
    return rc_all;
}

int mldsa_keygen_host(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys) {
    REQUIRE(ctx, "mldsa_keygen_host: NULL context");
    const mldsa_params *p = params_of(set);
    REQUIRE(p, "mldsa_keygen_host: unknown parameter set");
    if (n_keys == 0) return MLDSA_OK;
    REQUIRE(xi && pk && sk, "mldsa_keygen_host: NULL pointer");
    DeviceGuard dg(ctx->device);
    std::lock_guard<std::mutex> host_lk(ctx->host_mutex);
    HostStage *hs;
    TRY(stage_get(ctx, &hs));
    hs->secret_last = MLDSA_OP_KEYGEN;
    const int rc_all = [&]() -> int {
    const size_t pkl = (size_t)p->pk_len, skl = (size_t)p->sk_len;
    const size_t sub = std::min(n_keys, (size_t)ctx->opt_host_sub_verify);
    TRY(mldsa_reserve(ctx, set, MLDSA_OP_KEYGEN, sub));
    const bool pin_xi = is_pinned(xi, n_keys * 32), pin_pk = is_pinned(pk, n_keys * pkl), pin_sk = is_pinned(sk, n_keys * skl);
    int rc = MLDSA_OK;
    // seeds of sub-batch i + 1 go up before the keys of sub-batch i come down (copies are served in submission order)
    const size_t n_sub = (n_keys + sub - 1) / sub;
    auto stage_up = [&](size_t j) -> int {
        const size_t a = j * sub, n = std::min(n_keys, a + sub) - a;
        Slot &sl = hs->slot[j % N_SLOTS];
        TRY(reclaim(sl));
        TRY(upload(sl.xi, xi + a * 32, n * 32, pin_xi, hs->up));
        TRY(grow_dev(sl.pk, n * pkl));
        TRY(grow_dev(sl.sk, n * skl));
        HCHECK(hipEventRecord(sl.up_done, hs->up));
        return MLDSA_OK;
    };
    auto run = [&](size_t j) -> int {
        const size_t a = j * sub, n = std::min(n_keys, a + sub) - a;
        Slot &sl = hs->slot[j % N_SLOTS];
        HCHECK(hipStreamWaitEvent(hs->comp, sl.up_done, 0));
        TRY(mldsa_keygen(ctx, set, sl.xi.dev, sl.pk.dev, sl.sk.dev, n, hs->comp));
        HCHECK(hipEventRecord(sl.comp_done, hs->comp));
        HCHECK(hipStreamWaitEvent(hs->down, sl.comp_done, 0));
        TRY(download(sl, sl.pk, pk + a * pkl, n * pkl, pin_pk, hs->down));
        TRY(download(sl, sl.sk, sk + a * skl, n * skl, pin_sk, hs->down));
        HCHECK(hipEventRecord(sl.down_done, hs->down));
        sl.busy = true;
        return MLDSA_OK;
    };
    rc = stage_up(0);
    for (size_t j = 0; j < n_sub && rc == MLDSA_OK; j++) {
        if (j + 1 < n_sub) rc = stage_up(j + 1);
        if (rc == MLDSA_OK) rc = run(j);
    }
    for (auto &sl : hs->slot) {
        const int r2 = reclaim(sl);
        if (rc == MLDSA_OK) rc = r2;
    }
    return rc;
    }();
    if (rc_all != MLDSA_OK) (void)device_sync_quiesced();
    for (auto &sl : hs->slot) { wipe_buf(sl.xi, hs->comp); wipe_buf(sl.sk, hs->comp); }  // seeds and private keys
    (void)hipStreamSynchronize(hs->comp);
    return rc_all;
}

}  // extern "C"
