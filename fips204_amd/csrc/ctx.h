// Internal context of the C ABI (include/mldsa_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mldsa_hip.h"
#include "ntt_wave.h"

namespace mldsa {
#define MLDSA_EXP_RING 64
// Device-resident control block of the signer's rejection loop (kernels_sign.hip: k_make_slots).
struct RoundCtl {
    uint32_t cnt[2];       // unfinished ops entering a round of parity 0 / 1
    uint32_t m, spec, ns;  // this round: unfinished ops, candidates per op, candidate slots = m * spec
    uint32_t rounds;       // rounds that found work (statistics)
    uint32_t m_par[2];     // unfinished ops that entered the last round of parity 0 / 1
    uint32_t exp_cnt;      // ops appended to the completion-order list so far (k_compact; mldsa_sign_host's direct export)
    uint32_t exp_hi[64];   // exp_cnt at the end of round r (ring, r mod 64): round r's export copies [exp_hi[r - 1], exp_hi[r])
    uint32_t spec_par[2];  // candidates per op of the last round of parity 0 / 1 (k_make_slots of the next round reads the other one)
    uint32_t use_pre;      // this round's masks were generated one round ahead (k_expand_mask role 2): slot s reads y row slot_y[s]
    unsigned long long slots_total;  // candidate slots of all rounds of the call (statistics)
    unsigned long long ops_total;    // sum over the rounds of the unfinished ops entering them (statistics)
};

// One captured op-level call shape (pipeline.hip "hipGraph replay").
struct GraphEntry {
    std::vector<unsigned char> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t last_stream = nullptr;  // where it was last launched (waited for before the graph is destroyed)
    unsigned long long last_use = 0;
};
struct HostStage;
}  // namespace mldsa

struct mldsa_ctx {
    int device = 0;
    int n_cu = 256;
    // tuning knobs (mldsa_set_option)
    long opt_graphs = 1, opt_spec_target = 65536, opt_spec_max = 32, opt_va_blocks = 16, opt_graph_cache = 24;
    long opt_sign_rounds = 0, opt_sign_lanes = 1, opt_ct0_exact = 0, opt_mask_ahead = 0;
    // expected number of unfinished ops at which an ASYNCHRONOUS sign call stops planning rounds (1e-9: practically never an
    // MLDSA_ERR_AGAIN); mldsa_sign_host, which re-signs such ops anyway, raises it to the synchronous plan's 0.05 for its calls
    double async_stop = 1e-9;
    long opt_side_prologue = 0;  // sign: mu / rho'' / key-range / first active list on the helper stream underneath ExpandA (experiment knob; measured: no gain, 8.25 vs 8.19 ms per 65 536 ML-DSA-65 signatures)
    long opt_host_direct = 1;  // mldsa_sign_host: finished signatures are written into a page-locked caller buffer round by round
    long opt_host_sub_verify = 8192, opt_host_sub_sign = 16384;  // *_host entry points: ops per sub-batch (verify), ops of the LAST sub-batch (sign)
    mldsa_stats stats = {};
    // hipGraph replay of repeated op-level call shapes
    std::vector<mldsa::GraphEntry> graphs;
    unsigned long long graph_tick = 0;
    // *_host entry points: copy streams, page-locked bounce buffers and device staging (host_api.hip)
    mldsa::HostStage *host_stage = nullptr;
    std::mutex host_mutex;  // one *_host call at a time per context (taken before op_mutex)
    mldsa::Twiddle *d_fwd_tw = nullptr;  // [FWD_TW][64]
    mldsa::Twiddle *d_inv_tw = nullptr;  // [INV_TW][64]
    // op-level pipeline workspace (grown on demand, pipeline.hip)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // ops resident per pass of a pipeline (= what the workspace is sized for): verify / keygen and sign.  Tuned for a whole
    // MI355X (pipeline.hip); halved by reserve_workspace when the device cannot hold the workspace of a full pass.
    size_t pass_ops = 131072, pass_ops_sign = 262144;
    // second stream: the small latency-bound lane-per-op kernels of verify (hint unpack, mu,
    // SampleInBall) run here, concurrently with the VALU-bound ExpandA on the caller's stream
    hipStream_t aux_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    hipEvent_t exp_fork_ev = nullptr, exp_join_ev = nullptr;  // sign: the per-round export of finished signatures to host memory
    hipEvent_t pre_fork_ev = nullptr, pre_join_ev = nullptr;  // sign: the helper ExpandMask launch of a round (pipeline.hip)
    // HIP maps streams onto a handful of hardware queues (four here, handed out 0 1 2 3 3 2 1 0 ...: tools/ubench_queues.hip);
    // two streams on one queue run their kernels strictly one after the other.  parallel_stream() therefore PROBES which of
    // the context's helper streams really runs beside a given stream (a spin kernel on one, an empty kernel on the other)
    // and remembers the answer per stream handle.
    std::vector<hipStream_t> helper_streams;             // aux_stream first, more created on demand
    std::vector<std::pair<hipStream_t, hipStream_t>> parallel_of;  // (stream, helper that does not share its queue)
    std::vector<hipStream_t> prio_streams;                          // high-priority helpers (priority_stream)
    std::vector<std::pair<hipStream_t, hipStream_t>> prio_of;
    unsigned *d_probe = nullptr;
    // graphs of calls made on the legacy default stream run here (that stream cannot be captured)
    hipStream_t graph_stream = nullptr;
    hipEvent_t graph_fork_ev = nullptr, graph_join_ev = nullptr;
    // sign: page-locked copy of the loop's control block, read once after the enqueued rounds
    mldsa::RoundCtl *h_ctl = nullptr;
    // The op-level calls share the workspace and the helper streams: they serialise here.  The mutex
    // covers the host side of a call; ws_ev orders the device side when the next call comes on another
    // stream (device-side wait, no host synchronisation).
    std::mutex op_mutex;
    hipEvent_t ws_ev = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_busy = false;
    // A synchronous signing call clears its secrets (y, w, c, rho'' ...: 1.1 GB for 65 536 ML-DSA-65 ops) on a helper stream
    // after the signatures are complete; zero_ev marks the end of that, and the next op-level call waits for it on the device
    // before it touches the workspace (OpGuard) -- a signing call only after its ExpandA, which writes below [zero_lo, zero_hi).
    hipEvent_t zero_fork_ev = nullptr, zero_ev = nullptr;
    bool zero_pending = false, zero_wait_after_ea = false;
    size_t zero_lo = 0, zero_hi = 0;
    // optional per-stage timing: HIP event pairs recorded on the launch stream, resolved
    // only when the caller asks for the report (no synchronisation in the timed region)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;      // 2 events per recorded stage launch
    std::vector<const char *> prof_name;  // stage of pair i
    size_t prof_used = 0;                 // pairs in use
    unsigned long long prof_sign_slots = 0;  // candidate slots run by mldsa_sign while profiling
    unsigned long long prof_sign_op_rounds = 0;  // sum over rounds of unfinished ops (A_hat is needed once per op and round)
    unsigned long long last_sign_slots = 0;  // candidate slots of the last synchronous mldsa_sign call
};

namespace mldsa {

const mldsa_params *params_of(int set);
int set_error(int code, const char *what, hipError_t e = hipSuccess);

#define MLDSA_HIP_CHECK(expr)                                                        \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) return mldsa::set_error(MLDSA_ERR_DEVICE, #expr, _e);  \
    } while (0)

// persistent-style grid: enough workgroups to fill 256 CUs several times over, capped so
// that waves loop over units and keep their twiddles in registers
inline unsigned grid_for(const mldsa_ctx *ctx, size_t units, unsigned units_per_block, unsigned blocks_per_cu) {
    size_t need = (units + units_per_block - 1) / units_per_block;
    size_t cap = (size_t)ctx->n_cu * blocks_per_cu;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// Every extern "C" entry that touches HIP runs under one of these: the calling thread is bound to the
// context's device for the duration of the call and its previous device is restored afterwards, so a
// context created for device N works from any host thread and next to contexts of other devices.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// orders `s` after the background clearing of the previous signing call's secrets, if one is pending (pipeline.hip)
int wait_zeroise(mldsa_ctx *, hipStream_t s);

// RAII guard of one op-level call (capi.hip)
struct OpGuard {
    mldsa_ctx *c;
    hipStream_t s;
    std::unique_lock<std::mutex> lk;
    // defer_zero: the caller (sign_call) orders itself after the previous call's background clearing
    OpGuard(mldsa_ctx *ctx, hipStream_t stream, bool defer_zero = false) : c(ctx), s(stream), lk(ctx->op_mutex) {
        if (c->ws_busy && c->ws_stream != s) (void)hipStreamWaitEvent(s, c->ws_ev, 0);
        if (!defer_zero) (void)wait_zeroise(c, s);
    }
    ~OpGuard() {
        (void)hipEventRecord(c->ws_ev, s);
        c->ws_stream = s;
        c->ws_busy = true;
    }
};

// RAII stage marker used by pipeline.hip: records an event pair around one kernel launch
struct ProfScope {
    mldsa_ctx *c;
    hipStream_t s;
    long idx = -1;
    ProfScope(mldsa_ctx *ctx, hipStream_t stream, const char *name) : c(ctx), s(stream) {
        if (!c->prof_on) return;
        if (c->prof_used * 2 + 2 > c->prof_ev.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                c->prof_ev.push_back(e);
            }
            c->prof_name.push_back(name);
        } else {
            c->prof_name[c->prof_used] = name;
        }
        idx = (long)c->prof_used++;
        (void)hipEventRecord(c->prof_ev[2 * idx], s);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(c->prof_ev[2 * idx + 1], s);
    }
};

// ---- launchers (kernels_poly.hip) ----
// n_dev (where present): the unit count is read from device memory by the kernel (the signer's device-driven rounds);
// the host-side count then only sizes the grid
int launch_ntt(mldsa_ctx *, const int32_t *, int32_t *, size_t n_polys, hipStream_t, const uint32_t *n_dev = nullptr);
int launch_inv_ntt(mldsa_ctx *, const int32_t *, int32_t *, size_t n_polys, hipStream_t);
int launch_to_mont(mldsa_ctx *, const int32_t *, int32_t *, size_t n_polys, hipStream_t);
int launch_mat_vec_mul(mldsa_ctx *, int k, int l, const int32_t *, const int32_t *, int32_t *, size_t n_ops, hipStream_t);
int launch_pointwise_mont(mldsa_ctx *, const int32_t *, const int32_t *, int32_t *, size_t ppo, size_t n_ops, hipStream_t);
int launch_add(mldsa_ctx *, const int32_t *, const int32_t *, int32_t *, size_t n_polys, hipStream_t);
int launch_infinity_norm(mldsa_ctx *, const int32_t *, size_t ppo, size_t n_ops, int32_t *, hipStream_t);
int launch_verify_arith(mldsa_ctx *, int set, const int32_t *, const int32_t *, const int32_t *, const int32_t *, const uint32_t *key_idx, int32_t *, size_t n_ops, hipStream_t);


// ---- launchers (kernels_sample.hip) ----
// pack24: write A_hat as 24-bit fields, 768 bytes per polynomial (the pipelines' private form, sampler_dev.h)
int launch_expand_a(mldsa_ctx *, int set, const uint8_t *rho, size_t rho_stride, const uint32_t *key_idx, int32_t *a_hat, size_t n_ops, hipStream_t,
                    bool pack24 = false);
int launch_expand_s(mldsa_ctx *, int set, const uint8_t *rho_prime, size_t rho_stride, int32_t *s12, size_t n_ops, hipStream_t);
// yrisk (optional): one byte per polynomial, 1 = some |y| >= gamma1 - 2 beta (see k_sign_tail)
// ctl / role / kappa_add: the signer's two launches per round, see k_expand_mask
int launch_expand_mask(mldsa_ctx *, int set, const uint8_t *rho_pp, size_t rho_stride, const uint16_t *kappa, int kappa_by_slot,
                       const uint32_t *op_idx, int32_t *y, size_t n_ops, hipStream_t, uint8_t *yrisk = nullptr,
                       const uint32_t *n_dev = nullptr, const RoundCtl *ctl = nullptr, int role = 0, uint32_t kappa_add = 0);
int launch_sample_in_ball(mldsa_ctx *, int set, const uint8_t *c_tilde, size_t ct_stride, int32_t *c, size_t n_ops, hipStream_t,
                          const uint32_t *n_dev = nullptr);


// ---- launchers (kernels_codec.hip) ----
int launch_verify_main(mldsa_ctx *, const mldsa_params *, const int32_t *a_hat, const uint8_t *sigs, const int32_t *c, const int32_t *t1,
                       const uint32_t *key_idx, int32_t *hvalid, uint8_t *w1, size_t w1_stride, int32_t *znorm, size_t n_ops,
                       hipStream_t, bool a_by_key = false, bool a_packed = false);
int launch_mu(mldsa_ctx *, const uint8_t *tr, size_t tr_stride, const uint32_t *key_idx, int mode, const uint8_t *msgs,
              const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, uint8_t *mu, size_t mu_stride, int32_t *ctx_bad,
              size_t n_ops, hipStream_t, const int32_t *key_bad = nullptr);
int launch_shake256_2(mldsa_ctx *, int out_len, const uint8_t *a, size_t sa, int la, const uint32_t *a_idx, const uint8_t *b, size_t sb,
                      int lb, uint32_t tail, int tail_len, uint8_t *out, size_t so, size_t n_ops, hipStream_t,
                      const uint32_t *n_dev = nullptr);
int launch_ctilde_verdict(mldsa_ctx *, const mldsa_params *, const uint8_t *mu_w1, size_t mw, const uint8_t *sigs, const int32_t *znorm,
                          const int32_t *hvalid, const int32_t *ctx_bad, uint8_t *ok, size_t n_ops, hipStream_t);

// ---- launchers (kernels_sign.hip, kernels_poly.hip) ----
// y_polys_per_op: distance between consecutive ops' y vectors in polynomials (0 = L, contiguous)
// y_idx (optional): slot -> row of y (the signer's rounds may use masks generated one round ahead, laid out by the previous
// round's slots)
int launch_sign_w(mldsa_ctx *, int set, const int32_t *a_hat, const uint32_t *a_idx, const int32_t *y, int32_t *w, uint8_t *w1,
                  size_t w1_stride, size_t n_ops, hipStream_t, size_t y_polys_per_op = 0, uint8_t *wrisk = nullptr,
                  bool a_packed = false, const uint32_t *n_dev = nullptr, const uint32_t *y_idx = nullptr);
int launch_unpack_ntt(mldsa_ctx *, const uint8_t *src, size_t key_stride, size_t poly_off, int bits, int b, int32_t scale,
                      int32_t *out, int polys_per_key, size_t n_keys, hipStream_t);
// the round kernels of the signer: counts come from `ctl` on the device, the *_hint arguments only size the grids
int launch_sign_tail(mldsa_ctx *, const mldsa_params *, const int32_t *c, const int32_t *y, const int32_t *w, const uint8_t *ctilde,
                     const uint32_t *slot_op, const uint32_t *key_idx, const int32_t *s1, const int32_t *s2, const int32_t *t0,
                     uint16_t *kappa, int32_t *done, uint8_t *sigs, const RoundCtl *ctl, int32_t *accept, size_t slots_hint,
                     hipStream_t, const uint8_t *wrisk = nullptr, const uint8_t *yrisk = nullptr, const uint8_t *key_oor = nullptr,
                     int oor_by_op = 0, const uint32_t *slot_y = nullptr);
int launch_key_range(mldsa_ctx *, const mldsa_params *, const int32_t *s2, const uint32_t *kidx, size_t n_units, uint8_t *oor, hipStream_t);
// pre_enqueued: the previous round launched k_expand_mask role 2 (its rows are indexed by ypos = the op's position in that round)
int launch_make_slots(mldsa_ctx *, RoundCtl *ctl, int parity, uint32_t spec_target, uint32_t spec_max, const uint32_t *act,
                      const uint16_t *kappa, int l, uint32_t *slot_op, uint16_t *slot_kappa, const uint32_t *key_idx,
                      uint32_t *slot_key, size_t slots_hint, hipStream_t, int pre_enqueued = 0, const uint32_t *ypos = nullptr,
                      uint32_t *slot_y = nullptr);
// speculative rounds: builds the signature of each op's first surviving candidate (the candidates' c / y / w / c~ rows)
int launch_resolve(mldsa_ctx *, const mldsa_params *, const RoundCtl *ctl, const uint32_t *act, const int32_t *accept,
                   const int32_t *c, const int32_t *y, const int32_t *w, const uint8_t *ctilde, const uint32_t *key_idx,
                   const int32_t *s1, const int32_t *s2, const int32_t *t0, uint8_t *sigs, int32_t *done, uint16_t *kappa,
                   size_t ops_hint, hipStream_t, const uint8_t *key_oor = nullptr, int oor_by_op = 0, const uint32_t *slot_y = nullptr);
// ypos_out (optional): position each surviving op had in act_in (= its slot in a one-candidate round)
int launch_compact(mldsa_ctx *, RoundCtl *ctl, int parity, const uint32_t *act_in, const int32_t *done, uint32_t *act_out,
                   size_t ops_hint, hipStream_t, uint32_t *ypos_out = nullptr, uint32_t *exp_list = nullptr);
// mldsa_sign_host's direct export (kernels_sign.hip k_export_done): snapshot of the completion-order list at the end of a round
// (on the call's stream), then the copy of that round's finished signatures to host memory (device-visible pointer; helper stream)
int launch_export_snap(mldsa_ctx *, RoundCtl *ctl, int round, hipStream_t);
int launch_export_done(mldsa_ctx *, RoundCtl *ctl, int round, const uint32_t *exp_list, const uint8_t *sigs, uint8_t *host, size_t sig_len,
                       size_t ops_hint, hipStream_t snap_stream, hipStream_t);
int launch_init_active(mldsa_ctx *, size_t n, const int32_t *bad_op, int32_t *done, uint16_t *kappa, int32_t *status,
                       uint32_t *act_out, RoundCtl *ctl, uint8_t *sigs, size_t sig_len, hipStream_t);
int launch_mark_unfinished(mldsa_ctx *, const RoundCtl *ctl, int parity, const uint32_t *act, int32_t *status, uint8_t *sigs,
                           size_t sig_len, hipStream_t);
int launch_sanitize_keys(mldsa_ctx *, const uint32_t *key_idx, size_t n_keys, size_t n_ops, uint32_t *safe, int32_t *bad, hipStream_t);
int launch_zero(mldsa_ctx *, void *dst, size_t bytes, hipStream_t);
// true if a kernel on `b` only starts once a kernel on `a` has finished (same hardware queue); ~0.3 ms, waits for both streams
bool streams_serialise(mldsa_ctx *, hipStream_t a, hipStream_t b);
// a context-owned stream whose kernels run beside those of `s` (probed once per stream handle; `avoid`: a second stream it
// must not share a queue with either, or nullptr).  Falls back to aux_stream while `s` is being captured.
hipStream_t parallel_stream(mldsa_ctx *, hipStream_t s, hipStream_t avoid = nullptr);
hipStream_t priority_stream(mldsa_ctx *, hipStream_t s);  // the same, with the device's highest stream priority
int launch_copy_rows(mldsa_ctx *, void *dst, size_t dst_stride, const void *src, size_t src_stride, int row_bytes, size_t n_rows,
                     hipStream_t);
int launch_key_intt(mldsa_ctx *, const int32_t *src, int polys_per_key, size_t n_keys, int bits, int b, uint8_t *dst, size_t key_stride,
                    size_t poly_off, int32_t *out32, int out_ppk, int out_off, hipStream_t);
int launch_t1_hat(mldsa_ctx *, const mldsa_params *, const int32_t *as1, const int32_t *s1s2, int32_t *t1_out, size_t n_keys, hipStream_t);
int launch_keygen_encode(mldsa_ctx *, const mldsa_params *, const int32_t *s1s2, const int32_t *as1, const uint8_t *seeds,
                         uint8_t *pk, uint8_t *sk, size_t n_keys, hipStream_t);

// ---- op-level pipelines (pipeline.hip): pure enqueue functions (no allocation, no host wait) unless noted ----
int pk_expand_batch(mldsa_ctx *, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1, size_t n, hipStream_t);
int sk_expand_batch(mldsa_ctx *, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k, uint8_t *tr, int32_t *s1, int32_t *s2,
                    int32_t *t0, size_t n, hipStream_t);
int pk_into_bytes_batch(mldsa_ctx *, int set, const uint8_t *rho, const int32_t *t1, uint8_t *pk, size_t n, hipStream_t);
int sk_into_bytes_batch(mldsa_ctx *, int set, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1,
                        const int32_t *s2, const int32_t *t0, uint8_t *sk, size_t n, hipStream_t);
int get_public_key_batch(mldsa_ctx *, int set, const uint8_t *rho, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
                         uint8_t *pk_rho, uint8_t *pk_tr, int32_t *pk_t1, size_t n_keys, hipStream_t);
int keygen_batch(mldsa_ctx *, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys, hipStream_t);
// a_hat_keys != nullptr: per-key A_hat supplied by the caller (ExpandA is skipped).
// sign_batch replays / captures its chunks itself (run_op) and, unless async_mode, waits for the stream.
int sign_batch(mldsa_ctx *, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1,
               const int32_t *s2, const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
               const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs,
               int32_t *status, size_t n_ops, hipStream_t, const int32_t *a_hat_keys, bool async_mode, double plan_stop = 0.0,
               uint8_t *export_sigs = nullptr);
// mldsa_sign / mldsa_sign_async / mldsa_sign_cached_a behind their argument checks (capi.hip).  plan_stop > 0: expected number of
// unfinished ops at which the round plan stops, instead of the context's default for the mode (mldsa_sign_host re-signs
// leftovers itself and plans like a synchronous call).
int sign_call(mldsa_ctx *, int set, int mode, const uint8_t *rho, const int32_t *a_hat, const uint8_t *cap_k, const uint8_t *tr,
              const int32_t *s1, const int32_t *s2, const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
              const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *rnd, uint8_t *sigs,
              int32_t *status, size_t n_ops, hipStream_t, bool async_mode, double plan_stop = 0.0, uint8_t *export_sigs = nullptr);
int verify_batch(mldsa_ctx *, int set, int mode, const uint8_t *rho, const uint8_t *tr, const int32_t *t1_d2_hat_mont, size_t n_keys,
                 const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs,
                 const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok, size_t n_ops, hipStream_t,
                 const int32_t *a_hat_keys = nullptr);
// workspace: sized before a pipeline is enqueued (growing waits for the device and drops the captured graphs)
int ensure_workspace(mldsa_ctx *, size_t bytes);
size_t verify_workspace_bytes(const mldsa_ctx *, const mldsa_params *, size_t n_ops, bool own_a);
size_t sign_workspace_bytes(const mldsa_ctx *, const mldsa_params *, size_t n_ops, bool own_a);
size_t keygen_workspace_bytes(const mldsa_ctx *, const mldsa_params *, size_t n_keys);
// ensure_workspace for n_ops ops of MLDSA_OP_*; on MLDSA_ERR_NOMEM the context's pass size is halved and the request retried
int reserve_workspace(mldsa_ctx *, const mldsa_params *, int op, size_t n_ops, bool own_a);
// hipGraph replay of repeated call shapes: `key` = every value that ends up in a kernel parameter
int run_op(mldsa_ctx *, hipStream_t, int op, size_t n_ops, const void *key, size_t key_len, const std::function<int(hipStream_t)> &enqueue,
           bool allow_graph = true);
void drop_graphs(mldsa_ctx *);
void host_stage_destroy(mldsa_ctx *);  // host_api.hip

}  // namespace mldsa
