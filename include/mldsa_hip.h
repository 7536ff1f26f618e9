/*
 * mldsa_hip.h -- C ABI of the MI355X-native batched ML-DSA hot path.
 *
 * Drop-in boundary for the crate-private seams of integritychain/fips204 v0.4.6 (the
 * reference has no FFI of its own).  Each entry point names the
 * reference function it replaces (file:line relative to the crate root).  One call =
 * `n` independent units (polynomials or sign/verify operations).
 *
 * Conventions
 *  - Plain pointers and sizes only.  All data pointers are DEVICE pointers (hipMalloc'd
 *    by anyone: this library's mldsa_malloc, PyTorch, a Rust hip-sys binding ...) unless
 *    the name ends in _host (those take HOST pointers and do their own staging, see the end
 *    of this file).  `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    calls are asynchronous on it.
 *  - Devices.  A context belongs to the device it was created for.  Every call that takes a
 *    context binds the calling thread to that device for the duration of the call and restores
 *    the thread's previous device before it returns, so contexts of several GPUs can be driven
 *    from one process and from any thread.  The context-free memory helpers (mldsa_malloc ...)
 *    act on the calling thread's current device; mldsa_ctx_malloc allocates on the context's.
 *  - Polynomials are the reference's `R` / `T` (src/types.rs:45-55): 256 contiguous
 *    int32_t, arrays of polynomials contiguous and row-major ([[T; L]; K] = K*L*256).
 *    Inputs may be any signed representative inside the contract the reference's
 *    debug_asserts state (|x| < 2^31 - 2^22, helpers.rs:62); outputs documented per call.
 *  - Key material is passed by FIELD pointer, never as a struct: the reference's
 *    PublicKey / PrivateKey are not repr(C) (src/types.rs:19-41).
 *  - Return value: 0 = MLDSA_OK, negative = error (never aborts).  A failed signature
 *    verification is ok[i] = 0, not an error (src/lib.rs:368-370, ml_dsa.rs:368-376).
 *  - Randomness is always supplied by the caller (src/traits.rs:228-232).
 *  - Threads and streams.  The reference's functions are re-entrant and its keys Send + Sync.  Here
 *    the seam-level primitives keep no state in the context and may be called from any thread on
 *    any stream.  The op-level calls (mldsa_keygen / mldsa_sign / mldsa_verify) share the context's
 *    workspace: they may also be called from any thread on any stream, and the context serialises
 *    them (a mutex on the host side, a device-side event wait when consecutive calls use different
 *    streams -- never a host synchronisation).  For op-level calls that should overlap on the device,
 *    use one context per stream.  mldsa_sign returns after its last rejection round completed
 *    (mldsa_sign_async does not wait); everything else is asynchronous on `stream`.
 *  - Workspaces.  The op-level calls use a context-owned device workspace.  mldsa_reserve sizes it
 *    ahead of time; a call that finds it too small grows it, which waits for the device once.
 *  - hipGraphs.  An op-level call whose shape -- operation, parameter set, mode, n_ops and the pointer
 *    arguments -- repeats can be captured into a hipGraph the second time it is seen and replayed from then
 *    on: one graph launch instead of ~100 kernel launches for a signing call, 5 ... 25 times less host time
 *    per call (27 us instead of 150 us for a 4096-op ML-DSA-65 signing call).  The device is not faster for
 *    it -- the signing loop is driven from device memory and never waits for the host either way -- and a graph
 *    launch adds 40 ... 60 us of latency: measured (round 5, every size from 1 to 262144 ops) a replayed signing
 *    call is 3 ... 20 % SLOWER than the same call launched directly, waited for or issued back to back
 *    (profiles/r05_sweep_batch_sizes_extras.json), so MLDSA_OPT_GRAPHS defaults to 0: every call is launched directly.
 *    A host that is short of CPU time, not of latency, sets 1 (signing calls of up to 16384 ops replay) or 2
 *    (every call).  Results are identical either way.
 */
#ifndef MLDSA_HIP_H
#define MLDSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLDSA_OK 0
#define MLDSA_ERR_PARAM (-1)    /* unknown parameter set / NULL pointer / bad length   */
#define MLDSA_ERR_CTX_LEN (-2)  /* ctx longer than 255 bytes (src/lib.rs:274)           */
#define MLDSA_ERR_DEVICE (-3)   /* HIP runtime error; see mldsa_last_error()            */
#define MLDSA_ERR_NOMEM (-4)    /* device workspace allocation failed                   */
#define MLDSA_ERR_AGAIN (-5)    /* mldsa_sign_async only: op still unfinished, sign it again */

/* parameter sets: src/lib.rs:639-656, 681-698, 723-740 */
#define MLDSA_44 44
#define MLDSA_65 65
#define MLDSA_87 87

/* message-representative modes of sign_internal / verify_internal
 * (src/ml_dsa.rs:185-194, 386-395) */
#define MLDSA_MODE_PURE 0      /* mu = H(tr | 0x00 | len(ctx) | ctx | M)      ML-DSA.Sign/Verify   */
#define MLDSA_MODE_INTERNAL 1  /* mu = H(tr | M)   (`nist = true`, the ACVP "internal" interface)   */
#define MLDSA_MODE_PREHASH 2   /* mu = H(tr | 0x01 | len(ctx) | ctx | OID | PH(M)); msg = OID|PH(M) */

/* Bumped whenever a struct of this header grows or an entry point changes its meaning (4: mldsa_stats has 6 fields, the group
 * calls take device-resident slices, offset tables are validated; 5: key-lifetime calls of the batcher, MLDSA_OPT_SMALL_FUSED;
 * 6: the environment is read only under MLDSA_TUNING_ENV=1, three closed knobs are gone, mldsa_group_rccl_info).
 * mldsa_abi_version() reports the library's. */
#define MLDSA_ABI_VERSION 6
int mldsa_abi_version(void);

typedef struct mldsa_ctx mldsa_ctx;

typedef struct {
    int set, k, l, eta, tau, lambda, gamma1, gamma2, omega, beta;
    int ctilde_len, pk_len, sk_len, sig_len, w1_len;
} mldsa_params;

/* ---- context: device, twiddle tables, workspaces ------------------------------------ */
int mldsa_ctx_create(int device_id, mldsa_ctx **out);
void mldsa_ctx_destroy(mldsa_ctx *ctx);
const char *mldsa_last_error(void);
int mldsa_get_params(int set, mldsa_params *out);
int mldsa_device_count(void);
int mldsa_ctx_device(const mldsa_ctx *ctx); /* device id the context is bound to (negative: NULL ctx) */

/* Size the context's workspace for op-level calls of up to n_ops operations of `op` on `set`, so that
 * no later call of that size has to grow it (growing waits for the device).  op: MLDSA_OP_*. */
#define MLDSA_OP_KEYGEN 1
#define MLDSA_OP_SIGN 2
#define MLDSA_OP_VERIFY 3
int mldsa_reserve(mldsa_ctx *ctx, int set, int op, size_t n_ops);

/* Caller-owned workspace.  The reference allocates nothing (`#![no_std]`, no alloc: README.md:15-16); a host that manages device
 * memory itself -- a pool, a fixed budget next to other work -- hands the context its workspace instead of letting it hipMalloc
 * one: dev_buf (256-byte aligned, on the context's device, `bytes` long) is used for every op-level call from now on, is never
 * freed or regrown by the context, and is still cleared of secrets at the end of every call and (entirely) on mldsa_ctx_destroy /
 * when it is replaced.  A call whose full pass does not fit runs in smaller passes, like under MLDSA_OPT_WORKSPACE_CAP_MB; a
 * buffer too small for a 1024-op pass fails the call with MLDSA_ERR_NOMEM.  (NULL, 0) returns to a context-owned workspace.
 * Waits for the device. */
int mldsa_ctx_set_workspace(mldsa_ctx *ctx, void *dev_buf, size_t bytes);

/* Tuning knobs (per context).  Defaults are the measured best; none changes any result. */
#define MLDSA_OPT_GRAPHS 1          /* hipGraph replay of repeated call shapes: 0 (default) never, 1 signing calls of <= 16384 ops, 2 every call */
#define MLDSA_OPT_SPEC_TARGET 2     /* sign: candidate slots per speculative round (1 ... 524288, default 65536)       */
#define MLDSA_OPT_SPEC_MAX 3        /* sign: most speculative candidates per op and round (1 ... 64, default 32)        */
#define MLDSA_OPT_VA_BLOCKS_PER_CU 4 /* mldsa_verify_arith: workgroups per CU of the persistent grid (default 16)      */
#define MLDSA_OPT_GRAPH_CACHE 5     /* graphs kept per context before the least recently used one is dropped (default 24) */
#define MLDSA_OPT_SIGN_ROUNDS 6     /* sign: rounds enqueued before the host looks at the device; 0 (default) = as many as the
                                       plan says finish the batch (see mldsa_sign); a small value exercises the extra-round path */
#define MLDSA_OPT_SIGN_LANES 7      /* sign: a batch as two slices whose round chains run side by side on two streams.  0 (default): for calls
                                       (passes) of >= 131 072 ops (ML-DSA-44: 65 536), where it measured +2.5 ... 12 %; 1: never; 2: for every call of >= 8 192 ops.
                                       Signatures do not depend on it */
#define MLDSA_OPT_SIGN_CT0_EXACT 8  /* sign, ML-DSA-44 (test knob): 1 = always compute ||c t0||inf for the test of ml_dsa.rs:312, 0 (default) = only
                                       when the bound the hint stage gets for free cannot decide; signatures are identical */
#define MLDSA_OPT_SIGN_ASYNC_EXP 9   /* mldsa_sign_async: rounds are planned until the expected number of unfinished ops of the call is below
                                       10^-value (1..12, default 9: practically never an MLDSA_ERR_AGAIN).  A caller that re-signs such
                                       ops anyway can lower it: 2 plans like the synchronous call (three ~0.2 ms rounds less, an op
                                       left over in about 1 call in 500) */
#define MLDSA_OPT_SIGN_LOOKAHEAD 10  /* sign: in the early rounds of a batch of >= 8192 ops (one candidate tested per op) a round may generate the
                                       masks and w = A y of TWO candidates per op, which share the read of the op's A_hat -- the kernel is bound
                                       by re-reading A_hat from HBM there --, and the next round tests the second one without generating
                                       anything.  0 = never, 1 (default) = for the parameter sets where it measured faster (ML-DSA-65: +2.7 %),
                                       2 = for every set.  Signatures are identical */
#define MLDSA_OPT_WORKSPACE_CAP_MB 11 /* most MiB of device memory the context's workspace may take (0 = default: whatever the device gives).  A call
                                       whose full pass does not fit runs in smaller passes (see workspace_shrinks: the context lowers its candidates
                                       per speculative round and then its ops per pass, and keeps the smaller values until the cap is raised or
                                       removed; the options the caller set -- MLDSA_OPT_SPEC_TARGET ... -- are never rewritten, and a call that
                                       cannot be served at any size leaves the context as it found it); results are identical.  For hosts
                                       that share the GPU with other work; a cap too small even for a 1024-op pass fails the call with MLDSA_ERR_NOMEM */
#define MLDSA_OPT_COOP_HASH 12 /* 1 (default): the sponges of SMALL calls -- the fixed-shape SHAKE256 hashes (c_tilde, rho'', tr, the keygen seed) up to
                                * 4 096 ops, mu and SampleInBall up to 1 024 ops, ExpandA / ExpandS / the signer's ExpandMask up to 4 096 polynomials -- run
                                * wave-cooperatively, ONE state per wave in bit-interleaved form (csrc/keccak_coop2.h): 2.2 instead of 9.4 us per
                                * permutation of a latency-bound call.  0: always the lane-per-state form of the large batches.  Results are
                                * identical. */
#define MLDSA_OPT_SMALL_FUSED 13 /* most ops of a call that runs on the single-launch kernels of csrc/kernels_small.hip, counted in ML-DSA-65 ops: the
                                    limit of another set scales with the polynomials of A_hat per op -- value * 30 / (k l): 256 means 480 ML-DSA-44
                                    ops, 256 ML-DSA-65 ops, 137 ML-DSA-87 ops (measured crossovers per set: profiles/r05_ab_small_limits_per_set.txt).
                                    Verification: ONE launch
                                    (every op owns a cluster of workgroups for ExpandA, mu and SampleInBall; the last one to finish carries on with
                                    the arithmetic, the c_tilde hash and the verdict).  Key generation: ONE launch (one wave per polynomial of A_hat,
                                    s1, s2; the last workgroup does the arithmetic, the packing and tr); at most 256 ML-DSA-65 keys' worth whatever the value.
                                    Signing: the prologue is one launch (calls of at most 256 ops; it opens round 0 as well, and one launch between
                                    rounds compacts the active list and opens the next) and so is the first half of every round planned at <= 819
                                    candidate rows (ExpandMask, w = A y, the c_tilde hash, SampleInBall, NTT(c)).  Default 256 (measured
                                    crossovers against the batch pipeline: ~350 verifications, ~400 keys); 0 = always the batch pipeline; at most
                                    1024.  Needs MLDSA_OPT_COOP_HASH = 1.  Results are bit-identical either way; one ML-DSA-65 op: verify 50 instead
                                    of 105 us, key generation 73 instead of 166, signing 127 instead of 227. */
int mldsa_set_option(mldsa_ctx *ctx, int option, long value);
long mldsa_get_option(const mldsa_ctx *ctx, int option);

/* Environment.  By DEFAULT THE LIBRARY READS NOTHING FROM THE ENVIRONMENT THAT CHANGES WHAT IT LAUNCHES: a signing library is not
 * re-scheduled by whatever MLDSA_* variables a host process happens to carry.  The measurement knobs below are honoured only when the
 * process also sets MLDSA_TUNING_ENV=1 (or the library was built with -DMLDSA_TUNING); mldsa_ctx_create then takes them as the INITIAL
 * value of the option / context field beside them (out-of-range values are ignored; mldsa_set_option still overrides).  They exist for
 * same-box A/Bs (tools/ab_*.sh, tools/coop_thresholds.sh, the soak of tests/test_gpu_sign_schedule.py); results never depend on them.
 * tests/test_source_guards_cpu.py: every getenv() of the library is named here.
 *   MLDSA_TUNING_ENV            1 = read the knobs below (anything else, or unset: none of them is read)
 *   MLDSA_GRAPHS                MLDSA_OPT_GRAPHS (0 ... 2)
 *   MLDSA_COOP_HASH             MLDSA_OPT_COOP_HASH (0 / 1)
 *   MLDSA_SMALL_FUSED           MLDSA_OPT_SMALL_FUSED (0 ... 1024)
 *   MLDSA_SMALL_KEYGEN_MAX      most keys of a single-launch key generation (ML-DSA-65 keys; default 256, under MLDSA_OPT_SMALL_FUSED)
 *   MLDSA_SMALL_SIGN_MAX        most ops of a signing call whose prologue is one launch (default 256)
 *   MLDSA_SMALL_SIGN_FRONT      0 = small signing rounds on the five batch kernels instead of the single-launch round front (default 1)
 *   MLDSA_SMALL_SIGN_BACK       0 = the second half of a small signing round (tests, the winner's signature, bookkeeping) as three kernels instead of one launch (default 1)
 *   MLDSA_SMALL_BACK_SLOTS_MAX  ... in rounds planned at up to this many candidate slots (default 2560)
 *   MLDSA_SMALL_SIGN_SPEC       candidates per op a small call must keep in round 0 to speculate only as far as that launch reaches (default 12; 0 = never)
 *   MLDSA_COOP_HASH_MAX, MLDSA_COOP_MASK_MAX, MLDSA_COOP_A_MAX, MLDSA_COOP_MU_MAX, MLDSA_COOP_SIB_MAX
 *                               largest launch (ops resp. polynomials) that takes the wave-cooperative sponge: fixed-shape hashes, ExpandMask,
 *                               ExpandA (default 4096 each), mu, SampleInBall (default 1024 each); all under MLDSA_OPT_COOP_HASH
 *   MLDSA_SPEC_TARGET           MLDSA_OPT_SPEC_TARGET
 *   MLDSA_SPEC_MAX              MLDSA_OPT_SPEC_MAX
 *   MLDSA_SPEC_ROWS             sign: candidates generated per speculative round (default 65536; beside MLDSA_OPT_SPEC_TARGET)
 *   MLDSA_SIGN_LANES            MLDSA_OPT_SIGN_LANES
 *   MLDSA_LOOKAHEAD             MLDSA_OPT_SIGN_LOOKAHEAD
 *   MLDSA_VA_BLOCKS_PER_CU      MLDSA_OPT_VA_BLOCKS_PER_CU
 *   MLDSA_HOST_SUB_VERIFY, MLDSA_HOST_SUB_SIGN
 *                               *_host calls: ops per sub-batch (verify, keygen; default 8192) and of the last sub-batch (sign; default 16384)
 *   MLDSA_HOST_DIRECT           0 = mldsa_sign_host never exports signatures round by round into a page-locked caller buffer (default 1)
 *   MLDSA_WORKSPACE_CAP_MB      MLDSA_OPT_WORKSPACE_CAP_MB
 *   MLDSA_PASS_OPS, MLDSA_PASS_OPS_SIGN
 *                               ops resident per pass of a pipeline = what the workspace is sized for (default 131072 verify / keygen, 262144 sign)
 * One more variable is read regardless (it changes what is PRINTED, nothing that runs):
 *   MLDSA_DEBUG_IGNORED         1 = report on stderr every HIP error the library tolerates and clears (host_common.h)
 * Removed in ABI 6 with their code paths (A/Bs closed, EXPERIMENTS.md): MLDSA_SIB_THIRD_STREAM, MLDSA_SIDE_PROLOGUE, MLDSA_SPEC_ALPHA. */
/* counters for tests and bench.py: graphs captured / replayed, direct (un-captured) op-level calls, workspace growths,
 * extra signing rounds, and workspace_shrinks = how often the context made its passes smaller (the speculative rows of a signing pass
 * first, then the ops per pass) because the workspace of a full pass did not fit -- the device, MLDSA_OPT_WORKSPACE_CAP_MB or a caller-owned
 * buffer; the call then runs in more passes, results are identical */
typedef struct {
    unsigned long long graphs_captured, graph_replays, direct_calls, workspace_growths, sign_extra_rounds, workspace_shrinks;
} mldsa_stats;
int mldsa_get_stats(mldsa_ctx *ctx, mldsa_stats *out);
/* the same with the size of the CALLER's struct: at most out_bytes are written (a client built against a header whose mldsa_stats
 * was shorter is not overrun; fields the library does not know read as zero) */
int mldsa_get_stats_sized(mldsa_ctx *ctx, void *out, size_t out_bytes);

/* Per-stage timing of the op-level calls (bench.py's roofline figure): while enabled, every
 * kernel launch of mldsa_verify / mldsa_sign is bracketed by a HIP event pair on the launch
 * stream.  mldsa_profile_report synchronises the device, writes a JSON object
 * {"stage": {"ms": total, "calls": n}, ...} into buf and resets the counters. */
int mldsa_profile_enable(mldsa_ctx *ctx, int on);
int mldsa_profile_report(mldsa_ctx *ctx, char *buf, size_t buf_len);

/* Test support for the mirror of the reference's `Zeroize, ZeroizeOnDrop` key structs (src/types.rs:19, 45): secret-dependent
 * intermediates (y, w, c s1, rho'', rho', s1, s2, K ...) and staged private keys / seeds / rnd must not outlive the call that used
 * them.  mldsa_debug_secret_residue waits for the device (and with it for the call's background clearing), then counts the non-zero
 * bytes of (a) the part of the workspace the LAST op-level call used for secret-dependent data and (b) the staging buffers that
 * held secrets during the last *_host call.  scanned_bytes says how much was looked at (0 after a verify-only history);
 * nonzero_bytes must be 0.  Nothing is cleared by the probe itself.  mldsa_debug_count_nonzero: the same count over any device
 * range (synchronous). */
int mldsa_debug_secret_residue(mldsa_ctx *ctx, size_t *scanned_bytes, size_t *nonzero_bytes);
int mldsa_debug_count_nonzero(const void *dev_ptr, size_t bytes, size_t *nonzero);

/* ---- device memory helpers for hosts without their own HIP binding ------------------ */
int mldsa_malloc(void **dev_ptr, size_t bytes);
int mldsa_ctx_malloc(mldsa_ctx *ctx, void **dev_ptr, size_t bytes); /* on the context's device */
int mldsa_free(void *dev_ptr);
/* page-locked host memory: buffers handed to the *_host entry points stream at the PCIe rate when
 * they come from here (pageable memory works too, through a bounce copy) */
int mldsa_host_alloc(void **host_ptr, size_t bytes);
int mldsa_host_free(void *host_ptr);
int mldsa_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int mldsa_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int mldsa_memset(void *dst_dev, int value, size_t bytes, void *stream);
int mldsa_stream_sync(void *stream);

/* ---- seam-level batched primitives (hot path) ------------------ */

/* ntt::<KL>() src/ntt.rs:14-76 (FIPS 204 Alg 41).  w_hat may alias w.
 * Output: plain domain, bit-reversed order as the reference, representative in (-9q, 9q). */
int mldsa_ntt(mldsa_ctx *ctx, const int32_t *w, int32_t *w_hat, size_t n_polys, void *stream);

/* inv_ntt::<KL>() src/ntt.rs:85-161 (Alg 42).  Output canonical [0, q) like the
 * reference's full_reduce32 (ntt.rs:152-154).  w may alias w_hat. */
int mldsa_inv_ntt(mldsa_ctx *ctx, const int32_t *w_hat, int32_t *w, size_t n_polys, void *stream);

/* to_mont() src/helpers.rs:131-135: x * 2^32 mod q, output in (-q, q). */
int mldsa_to_mont(mldsa_ctx *ctx, const int32_t *in, int32_t *out, size_t n_polys, void *stream);

/* partial_reduce32 / full_reduce32 / center_mod, src/helpers.rs:61-67, 70-76, 88-95, element-wise over n_polys polynomials (the
 * reference applies them coefficient by coefficient at ml_dsa.rs:89-91, 263-272, 302-303, 334-335 and ntt.rs:153; inside the fused
 * kernels they are single instructions of an epilogue -- these entry points exist so that the seam is testable on its own).
 * Input: any representative with |a| < 2^31 - 2^22 (helpers.rs:62).  Output: PARTIAL (-q, q), FULL [0, q), CENTER (-q/2, q/2]. */
#define MLDSA_REDUCE_PARTIAL 0
#define MLDSA_REDUCE_FULL 1
#define MLDSA_REDUCE_CENTER 2
int mldsa_reduce(mldsa_ctx *ctx, int kind, const int32_t *in, int32_t *out, size_t n_polys, void *stream);

/* The rounding functions of src/high_low.rs, element-wise over n_polys polynomials (inside the pipelines they are epilogues
 * of k_verify_main / sign_w / k_sign_tail -- these entry points make the seam testable on its own).  gamma2 comes from `set`.
 *   POWER2ROUND (high_low.rs:15-48)  a in [0, q)                 -> out1 = r1, out2 = r0
 *   DECOMPOSE   (66-96)              a any representative         -> out1 = r1, out2 = r0
 *   HIGH_BITS / LOW_BITS (104-126)   a                            -> out1
 *   MAKE_HINT   (134-144)            a = z, b = r                 -> out1 = 0 / 1
 *   USE_HINT    (155-192)            a = h (0 / 1), b = r         -> out1 */
#define MLDSA_ROUND_POWER2ROUND 0
#define MLDSA_ROUND_DECOMPOSE 1
#define MLDSA_ROUND_HIGH_BITS 2
#define MLDSA_ROUND_LOW_BITS 3
#define MLDSA_ROUND_MAKE_HINT 4
#define MLDSA_ROUND_USE_HINT 5
int mldsa_rounding(mldsa_ctx *ctx, int set, int op, const int32_t *a, const int32_t *b, int32_t *out1, int32_t *out2, size_t n_polys,
                   void *stream);

/* h256_xof / g128_xof (src/hashing.rs:13-27): SHAKE256 (bits = 256) / SHAKE128 (bits = 128) of one byte string per op --
 * data[off[i] .. off[i + 1]), the reference's list of slices concatenated by the caller -- first out_len bytes to out[i * out_len].
 * `off` is a device array of n_ops + 1 entries and is untrusted like msg_off: an op with a malformed pair is not read, its output
 * is zero and bad[i] = 1 (bad may be NULL).  A seam for tests and for callers that need the XOF itself; the pipelines' hashes
 * (mu, rho'', c_tilde, tr, the samplers) are fixed-shape kernels of their own. */
int mldsa_xof(mldsa_ctx *ctx, int bits, const uint8_t *data, const uint64_t *off, uint8_t *out, size_t out_len, uint8_t *bad, size_t n_ops,
              void *stream);

/* The wire-format codecs as seams.  Inside mldsa_verify / mldsa_sign / mldsa_keygen they are fused
 * into the arithmetic kernels (no int32 z, h or w1 ever reaches HBM); these entry points run the same device code on int32
 * polynomials (16-byte aligned) so that every codec of src/conversion.rs and src/encodings.rs can be checked on its own.  `ok`
 * arrays hold one byte per polynomial / operation: 1 = the reference returns Ok, 0 = it returns Err (or, for the encoders,
 * trips a debug_assert: the output is then well-formed but lossy).  `ok` may be NULL for the encoders.
 *   mldsa_bit_pack         bit_pack (conversion.rs:143-186); a = 0 is simple_bit_pack (120-132).  out: 32 * bitlen(a + b) bytes / poly
 *   mldsa_bit_unpack       bit_unpack (227-262); a = 0 is simple_bit_unpack (198-213)
 *   mldsa_hint_bit_pack    hint_bit_pack (277-328): h [n_ops][K][256] of 0 / 1 -> y [n_ops][omega + K]; ok = 0 if weight(h) > omega
 *   mldsa_hint_bit_unpack  hint_bit_unpack (340-414): ok = 0 for a malformed section (h is then all zero)
 *   mldsa_sig_encode       sig_encode (encodings.rs:238-280): c_tilde [n_ops][lambda/4], z [n_ops][L][256] in (-gamma1, gamma1],
 *                          h [n_ops][K][256] -> sigs [n_ops][sig_len]
 *   mldsa_sig_decode       sig_decode (290-328): the reverse; ok = 0 where the reference returns Err
 *   mldsa_w1_encode        w1_encode (338-360): w1 [n_ops][K][256] in [0, (q-1)/(2 gamma2)) -> [n_ops][w1_len] */
int mldsa_bit_pack(mldsa_ctx *ctx, const int32_t *w, int a, int b, uint8_t *out, size_t n_polys, void *stream);
int mldsa_bit_unpack(mldsa_ctx *ctx, const uint8_t *v, int a, int b, int32_t *w, uint8_t *ok, size_t n_polys, void *stream);
int mldsa_hint_bit_pack(mldsa_ctx *ctx, int set, const int32_t *h, uint8_t *y, uint8_t *ok, size_t n_ops, void *stream);
int mldsa_hint_bit_unpack(mldsa_ctx *ctx, int set, const uint8_t *y, int32_t *h, uint8_t *ok, size_t n_ops, void *stream);
int mldsa_sig_encode(mldsa_ctx *ctx, int set, const uint8_t *c_tilde, const int32_t *z, const int32_t *h, uint8_t *sigs, uint8_t *ok,
                     size_t n_ops, void *stream);
int mldsa_sig_decode(mldsa_ctx *ctx, int set, const uint8_t *sigs, uint8_t *c_tilde, int32_t *z, int32_t *h, uint8_t *ok, size_t n_ops,
                     void *stream);
int mldsa_w1_encode(mldsa_ctx *ctx, int set, const int32_t *w1, uint8_t *out, size_t n_ops, void *stream);

/* mat_vec_mul::<K,L>() src/helpers.rs:100-114: w_hat[i] = sum_j a_hat[i][j] o u_hat[j]
 * for n_ops independent (a_hat, u_hat) pairs.  a_hat: n_ops*K*L polys, u_hat: n_ops*L,
 * w_hat: n_ops*K; output representative in (-L q, L q). */
int mldsa_mat_vec_mul(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *u_hat,
                      int32_t *w_hat, size_t n_ops, void *stream);

/* scalar-vector product c_hat o v_hat_mont inlined at src/ml_dsa.rs:243-250, 253-260,
 * 288-295 (FIPS 204 Alg 45/47): out[op][p] = mont_reduce(c_hat[op] * v_hat_mont[op][p]),
 * output in (-q, q).  c_hat: n_ops polys, v_hat_mont/out: n_ops*polys_per_op polys. */
int mldsa_pointwise_mont(mldsa_ctx *ctx, const int32_t *c_hat, const int32_t *v_hat_mont,
                         int32_t *out, size_t polys_per_op, size_t n_ops, void *stream);

/* add_vector_ntt() src/helpers.rs:125-127: element-wise a + b (no reduction). */
int mldsa_add_vector_ntt(mldsa_ctx *ctx, const int32_t *a, const int32_t *b, int32_t *out,
                         size_t n_polys, void *stream);

/* infinity_norm::<ROW>() src/helpers.rs:138-147: norms[op] = max |center_mod(w)| over the
 * op's polys_per_op polynomials. */
int mldsa_infinity_norm(mldsa_ctx *ctx, const int32_t *polys, size_t polys_per_op, size_t n_ops,
                        int32_t *norms, void *stream);

/* The verify-arithmetic unit (BASELINE config 2), src/ml_dsa.rs:407-416 fused in one
 * kernel: w'[op] = inv_ntt(a_hat[op] * ntt(z[op]) - ntt(c[op]) o t1_d2_hat_mont[op]).
 * a_hat: n_ops*K*L (|a| < 2^24: what mldsa_expand_a produces; the K*L products of a row are accumulated in 64 bits
 * and reduced once), z: n_ops*L, c: n_ops, t1_d2_hat_mont: n_ops*K (Montgomery form as in PublicKey,
 * src/types.rs:40; z, c, t1 any representative of the general contract), w_out: n_ops*K polys, canonical [0, q). */
int mldsa_verify_arith(mldsa_ctx *ctx, int set, const int32_t *a_hat, const int32_t *z,
                       const int32_t *c, const int32_t *t1_d2_hat_mont, int32_t *w_out,
                       size_t n_ops, void *stream);

/* ---- SHAKE-driven samplers (hot path, src/hashing.rs) -------------------------------- */

/* expand_a::<K,L>() src/hashing.rs:225-239 (Alg 32; rej_ntt_poly 111-146): one rho (32 B)
 * per op -> a_hat[op][K][L][256], canonical [0, q), NTT domain. */
int mldsa_expand_a(mldsa_ctx *ctx, int set, const uint8_t *rho, int32_t *a_hat, size_t n_ops,
                   void *stream);

/* expand_s::<K,L>() src/hashing.rs:252-272 (Alg 33; rej_bounded_poly 158-213): one rho'
 * (64 B) per op -> s1s2[op][L + K][256] (s1 = first L polys, s2 = next K), in [-eta, eta]. */
int mldsa_expand_s(mldsa_ctx *ctx, int set, const uint8_t *rho_prime, int32_t *s1s2,
                   size_t n_ops, void *stream);

/* expand_mask::<L>() src/hashing.rs:281-313 (Alg 34): rho'' (64 B) and counter kappa per op
 * -> y[op][L][256] in [-gamma1 + 1, gamma1]. */
int mldsa_expand_mask(mldsa_ctx *ctx, int set, const uint8_t *rho_pp, const uint16_t *kappa,
                      int32_t *y, size_t n_ops, void *stream);

/* sample_in_ball() src/hashing.rs:43-100 (Alg 29): c_tilde (lambda/4 bytes) per op ->
 * c[op][256] with tau coefficients +-1. */
int mldsa_sample_in_ball(mldsa_ctx *ctx, int set, const uint8_t *c_tilde, int32_t *c,
                         size_t n_ops, void *stream);

/* ---- op-level batched API (mirrors the Verifier / Signer / KeyGen / SerDes traits) ---- */

/* Verifier::verify / _internal_verify / hash_verify (src/lib.rs:364-411, 605-612) ->
 * verify_internal (src/ml_dsa.rs:351-437) for n_ops independent operations.
 *   Public keys are passed EXPANDED, field by field, as the reference's PublicKey holds them
 *   (src/types.rs:35-41): rho[n_keys][32], tr[n_keys][64], t1_d2_hat_mont[n_keys][K][256]
 *   (produce them with mldsa_pk_expand).  key_idx[op] < n_keys selects the key of op (NULL: op i
 *   uses key i, n_keys >= n_ops).  An op whose key_idx is out of range gets ok = 0 and never touches
 *   memory outside the n_keys rows.  A_hat is re-derived from rho for every op, as the reference does
 *   (ml_dsa.rs:406).
 *   msgs/msg_off: concatenated messages and n_ops + 1 byte offsets; ctxs/ctx_off likewise
 *   (ctx_off NULL = every ctx empty).  mode: MLDSA_MODE_*.  sigs: n_ops * SIG_LEN bytes.
 *   ok[op] = 1 iff the reference returns true; malformed hints, |ctx| > 255, z too large and
 *   c_tilde mismatch all give 0 (ml_dsa.rs:368-376, 434-436; lib.rs:368-370).
 *   Offsets are untrusted input like everything else (the reference never panics on it, fuzz_all.rs:25-37): the call vouches
 *   for the bytes [off[0], off[n_ops]) of msgs / ctxs and nothing outside them is read; an op whose pair is not in order inside
 *   that range (off[0] <= off[i] <= off[i + 1] <= off[n_ops]: a decreasing, wrapping or overshooting entry) is refused on its own
 *   -- ok = 0 here, status MLDSA_ERR_PARAM and an all-zero signature in mldsa_sign -- and the other ops are unaffected; an op
 *   whose ctx is longer than 255 bytes is refused before a byte of its ctx or message is read (lib.rs:274, 368).  The *_host
 *   entry points, which copy by these offsets on the host, check the tables first (mldsa_check_offsets) and fail the whole call
 *   with MLDSA_ERR_PARAM. */
int mldsa_verify(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *tr,
                 const int32_t *t1_d2_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                 const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                 const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream);

/* PublicKey::try_from_bytes (src/lib.rs:471-475 -> expand_public, src/ml_dsa.rs:477-498) and Verifier::verify in ONE call: the public
 * keys arrive in FIPS 204 wire format, pk[n_keys][PK_LEN] on the device, and are deserialised inside the call -- rho is read where it
 * lies in the key bytes, tr = H(pk) and NTT(t1) 2^13 are produced on the context's helper stream underneath ExpandA -- so a batch in
 * which every op carries its own key (key_idx NULL: op i uses key i, n_keys >= n_ops) costs little more than a batch on
 * expanded keys; with key_idx the n_keys keys of the table are deserialised once per call.  Everything else -- arguments,
 * verdicts, refusal rules -- is mldsa_verify's; the results are identical to mldsa_pk_expand followed by mldsa_verify.
 * Workspace: 64 + 1024 K bytes per key of a pass (identity mapping) or of the table. */
int mldsa_verify_pk(mldsa_ctx *ctx, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                    const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off, const uint8_t *sigs, uint8_t *ok,
                    size_t n_ops, void *stream);

/* n_ops + 1 byte offsets, non-decreasing?  MLDSA_OK or MLDSA_ERR_PARAM (mldsa_last_error names the entry).  Pure host code:
 * needs no context and no device. */
int mldsa_check_offsets(const uint64_t *off, size_t n_ops);

/* SerDes::try_from_bytes for PublicKey (src/lib.rs:471-475) -> expand_public
 * (src/ml_dsa.rs:477-498): pk[n][PK_LEN] -> rho[n][32], tr[n][64] = H(pk),
 * t1_d2_hat_mont[n][K][256] = NTT(t1) * 2^13 in Montgomery form.  Never fails (the
 * reference's range check is vacuous for 10-bit fields, conversion.rs:259-260). */
int mldsa_pk_expand(mldsa_ctx *ctx, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr,
                    int32_t *t1_d2_hat_mont, size_t n_keys, void *stream);

/* SerDes::try_from_bytes for PrivateKey (src/lib.rs:421-424) -> expand_private
 * (src/ml_dsa.rs:445-469): sk[n][SK_LEN] -> rho, cap_k [n][32], tr [n][64],
 * s_1_hat_mont [n][L][256], s_2_hat_mont / t_0_hat_mont [n][K][256] (Montgomery form). */
int mldsa_sk_expand(mldsa_ctx *ctx, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k,
                    uint8_t *tr, int32_t *s_1_hat_mont, int32_t *s_2_hat_mont,
                    int32_t *t_0_hat_mont, size_t n_keys, void *stream);

/* SerDes::into_bytes for PublicKey (src/lib.rs:478-493): the expanded fields back to pk[n][PK_LEN]:
 * t1 = inv_ntt(mont_reduce(t1_d2_hat_mont)) >> 13, pk = pkEncode(rho, t1).  tr is not needed. */
int mldsa_pk_into_bytes(mldsa_ctx *ctx, int set, const uint8_t *rho, const int32_t *t1_d2_hat_mont,
                        uint8_t *pk, size_t n_keys, void *stream);

/* SerDes::into_bytes for PrivateKey (src/lib.rs:427-465): s1, s2, t0 = centred inv_ntt of the Montgomery
 * fields, sk[n][SK_LEN] = skEncode(rho, K, tr, s1, s2, t0). */
int mldsa_sk_into_bytes(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr,
                        const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, const int32_t *t_0_hat_mont,
                        uint8_t *sk, size_t n_keys, void *stream);

/* Signer::get_public_key (src/lib.rs:345-349) -> private_to_public_key (src/ml_dsa.rs:502-559):
 * expanded private keys -> expanded public keys.  t = inv_ntt(A_hat o s_1_hat) + s_2 with A_hat = ExpandA(rho),
 * t1 = Power2Round(t), t1_d2_hat_mont = NTT(t1) * 2^13 (Montgomery form); rho and tr are copied.
 * (cap_k and t_0_hat_mont are not inputs: the reference only uses t_0 in a debug assertion.) */
int mldsa_get_public_key(mldsa_ctx *ctx, int set, const uint8_t *rho, const uint8_t *tr,
                         const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont, uint8_t *pk_rho, uint8_t *pk_tr,
                         int32_t *pk_t1_d2_hat_mont, size_t n_keys, void *stream);

/* KeyGen::keygen_from_seed (src/lib.rs:247-250) -> key_gen_internal (src/ml_dsa.rs:57-134)
 * followed by SerDes::into_bytes: xi[n][32] -> pk[n][PK_LEN], sk[n][SK_LEN] (FIPS 204 wire
 * format; expand with mldsa_pk_expand / mldsa_sk_expand to use them). */
int mldsa_keygen(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk,
                 size_t n_keys, void *stream);

/* Signer::try_sign_with_seed / _internal_sign / try_hash_sign_with_seed (src/lib.rs:268-342,
 * 586-600) -> sign_internal (src/ml_dsa.rs:153-337) for n_ops independent operations.
 *   Private keys EXPANDED, field by field (src/types.rs:19-28), from mldsa_sk_expand;
 *   n_keys / key_idx / msgs / ctxs / mode as in mldsa_verify.  rnd[n_ops][32]: the caller's per-
 *   signature randomness (all zero = deterministic variant), src/lib.rs:282-283.
 *   sigs[n_ops][SIG_LEN].  status (may be NULL): per-op MLDSA_OK, MLDSA_ERR_CTX_LEN (|ctx| > 255,
 *   src/lib.rs:274) or MLDSA_ERR_PARAM (key_idx out of range); the signature of a failed op is all zero.
 *   The rejection loop (ml_dsa.rs:212-330) runs in rounds over the unfinished ops and is driven from the
 *   device: the round kernels read the number of unfinished ops and the candidates per op from device
 *   memory, so a whole call is enqueued without the host looking at the device.
 *   mldsa_sign enqueues as many rounds as finish the batch in more than 999 of 1000 calls, waits for
 *   `stream` once, and enqueues further rounds in the rare case that an op is left: when it returns every
 *   op is signed, exactly like the reference's loop.
 *   mldsa_sign_async does not wait: it enqueues rounds until the probability that any op is left is below
 *   1e-9; such an op gets status MLDSA_ERR_AGAIN and an all-zero signature (status must not be NULL). */
int mldsa_sign(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k,
               const uint8_t *tr, const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont,
               const int32_t *t_0_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
               const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
               const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream);
int mldsa_sign_async(mldsa_ctx *ctx, int set, int mode, const uint8_t *rho, const uint8_t *cap_k,
                     const uint8_t *tr, const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont,
                     const int32_t *t_0_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                     const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                     const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream);

/* ---- A_hat kept with the keys --------------------------------------------------------
 * The reference re-derives A_hat = ExpandA(rho) inside every sign and verify (src/ml_dsa.rs:181,
 * 406) and names "the cap_a_hat pre-compute ... put into both PublicKey and PrivateKey structs"
 * as an open optimisation (benches/README.md:4-8).  These two entry points are that
 * optimisation: identical to mldsa_verify / mldsa_sign (same arguments, same results, bit for
 * bit) except that ExpandA is skipped and row key_idx[op] (or row op when key_idx is NULL) of
 * `a_hat` is used instead.  a_hat[n_keys][K][L][256] = mldsa_expand_a(set, rho of the keys), any
 * representative mldsa_expand_a produces.  bench.py reports them as separate workloads
 * (verify65_cached_a, sign65_cached_a), never as the headline. */
int mldsa_verify_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *tr,
                          const int32_t *t1_d2_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                          const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                          const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *stream);
int mldsa_sign_cached_a(mldsa_ctx *ctx, int set, int mode, const int32_t *a_hat, const uint8_t *cap_k,
                        const uint8_t *tr, const int32_t *s_1_hat_mont, const int32_t *s_2_hat_mont,
                        const int32_t *t_0_hat_mont, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                        const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                        const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *stream);

/* ---- host-memory entry points ----------------------------------------------------------
 * The reference's API works on host slices (src/traits.rs:118-308 Signer, 330-362 Verifier:
 * `verify(&self, message: &[u8], sig: &Signature, ctx: &[u8])`).  These variants take HOST pointers, keys
 * in FIPS 204 wire format, and do the staging themselves: the keys are uploaded and expanded once
 * (try_from_bytes), then the batch streams through the device in sub-batches with the upload of
 * sub-batch i + 1 and the download of sub-batch i - 1 running beside the kernels of sub-batch i (three
 * HIP streams, context-owned device and page-locked staging buffers, so the kernel sequence of a
 * sub-batch replays as a hipGraph).  Buffers obtained from mldsa_host_alloc (or otherwise page-locked)
 * are copied by DMA directly; pageable buffers go through a page-locked bounce buffer first.
 * Arguments are those of mldsa_verify / mldsa_sign / mldsa_keygen with every pointer on the host;
 * the calls return when the results are in the caller's buffers. */
int mldsa_verify_host(mldsa_ctx *ctx, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx,
                      const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                      const uint8_t *sigs, uint8_t *ok, size_t n_ops);
int mldsa_sign_host(mldsa_ctx *ctx, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx,
                    const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                    const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops);
int mldsa_keygen_host(mldsa_ctx *ctx, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys);

/* ---- several GPUs of one node -------------------------------------------------------------
 * The reference's operations are independent pure functions of their arguments (src/traits.rs:118-308,
 * 330-362), so a batch shards trivially: contiguous slices of ceil(n_ops / N) ops, one per device (SURVEY.md
 * 8e), no exchange on the data path.  A group owns one context and one host worker thread per entry of
 * device_ids (a device may be listed more than once: several contexts on one GPU).  The *_host_group calls
 * take exactly the arguments of mldsa_verify_host / mldsa_sign_host / mldsa_keygen_host, hand slice i to
 * worker i -- which runs the ordinary host-memory entry point of its context on its part of the caller's
 * arrays -- and return when every slice's results are in the caller's buffers.  Results are byte-identical
 * to the single-context call.  mldsa_group_ctx(g, i) gives the i-th context (options, reserve, the device-
 * resident entry points on the caller's own streams). */
typedef struct mldsa_group mldsa_group;
int mldsa_group_create(const int *device_ids, int n, mldsa_group **out);
void mldsa_group_destroy(mldsa_group *g);
int mldsa_group_size(const mldsa_group *g);
mldsa_ctx *mldsa_group_ctx(mldsa_group *g, int i);
/* the slice of an n_ops batch that part `part` of `n_parts` owns: first = min(n_ops, part * ceil(n_ops / n_parts)) */
int mldsa_group_shard(size_t n_ops, int n_parts, int part, size_t *first, size_t *count);
int mldsa_verify_host_group(mldsa_group *g, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx,
                            const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                            const uint8_t *sigs, uint8_t *ok, size_t n_ops);
int mldsa_sign_host_group(mldsa_group *g, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx,
                          const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *ctxs, const uint64_t *ctx_off,
                          const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops);
int mldsa_keygen_host_group(mldsa_group *g, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n_keys);
/* Device-resident slices: the batch is already sharded, slice i lives in the memory of device i of the group (keys expanded there
 * with mldsa_pk_expand / mldsa_sk_expand on mldsa_group_ctx(g, i), inputs uploaded there) -- the HBM-resident form of the calls
 * above, driven from ONE host thread of ONE process.  slices[i] holds exactly the arguments of mldsa_verify / mldsa_sign /
 * mldsa_keygen for slice i (offset tables are the slice's own: n_ops + 1 entries; n_ops may be 0); worker i makes that call on
 * its context.  wait != 0: the call returns when every slice's results are complete (mldsa_sign semantics for signing);
 * wait == 0: it returns when everything is enqueued on the slices' streams (mldsa_sign_async semantics: status is required and
 * may carry MLDSA_ERR_AGAIN) and mldsa_group_sync waits for those streams later.  Results are byte-identical to the
 * single-context calls on the same ops. */
typedef struct {
    const uint8_t *rho, *tr;               /* expanded public keys on the slice's device (mldsa_pk_expand) */
    const int32_t *t1_d2_hat_mont;
    size_t n_keys;
    const uint32_t *key_idx;
    const uint8_t *msgs;
    const uint64_t *msg_off;
    const uint8_t *ctxs;
    const uint64_t *ctx_off;
    const uint8_t *sigs;
    uint8_t *ok;
    size_t n_ops;
    void *stream;                          /* hipStream_t of the slice's device (NULL = its default stream) */
} mldsa_verify_slice;
typedef struct {
    const uint8_t *rho, *cap_k, *tr;       /* expanded private keys on the slice's device (mldsa_sk_expand) */
    const int32_t *s_1_hat_mont, *s_2_hat_mont, *t_0_hat_mont;
    size_t n_keys;
    const uint32_t *key_idx;
    const uint8_t *msgs;
    const uint64_t *msg_off;
    const uint8_t *ctxs;
    const uint64_t *ctx_off;
    const uint8_t *rnd;
    uint8_t *sigs;
    int32_t *status;
    size_t n_ops;
    void *stream;
} mldsa_sign_slice;
typedef struct {
    const uint8_t *xi;
    uint8_t *pk, *sk;
    size_t n_keys;
    void *stream;
} mldsa_keygen_slice;
int mldsa_verify_group(mldsa_group *g, int set, int mode, const mldsa_verify_slice *slices /* [mldsa_group_size] */, int wait);
int mldsa_sign_group(mldsa_group *g, int set, int mode, const mldsa_sign_slice *slices, int wait);
int mldsa_keygen_group(mldsa_group *g, int set, const mldsa_keygen_slice *slices, int wait);
int mldsa_group_sync(mldsa_group *g); /* waits for the streams of the last device-resident group call (they must still exist) */

/* Device-resident verdicts (each device ran mldsa_verify on its slice): the one exchange SURVEY 8e names.
 * bufs[i] = device pointer on device i of the group, N * ceil(n_ops / N) bytes, slice i of it filled; afterwards
 * every buffer holds all n_ops bytes.  use_rccl: 1 = ncclAllGather over xGMI (librccl.so is loaded on first
 * use; the devices must be distinct), 0 = device-to-device copies, -1 = RCCL if possible, copies otherwise.
 * Ordering: the gather runs after the last op-level call of every context of the group (a device-side wait on the event each
 * context records behind its calls -- no host synchronisation is needed between mldsa_verify / mldsa_verify_group and this
 * call); buffers filled by anything else must be complete before the call.  Returns when every buffer is gathered. */
int mldsa_group_allgather(mldsa_group *g, uint8_t *const *bufs, size_t n_ops, int use_rccl);
/* Which RCCL the gather binds.  The library is loaded on first use in this order: the copy the process has ALREADY mapped under the
 * SONAME librccl.so.1 (a Python host: torch's own torch/lib/librccl.so -- two RCCLs on one HIP runtime are avoided), then the usual
 * search (LD_LIBRARY_PATH, the rpath /opt/rocm/lib).  g != NULL: what this group's first RCCL gather bound (MLDSA_ERR_PARAM before it);
 * g == NULL: a probe that runs the same search without creating a communicator or touching a device.  buf <- "<reused|loaded> <file>";
 * returns ncclGetVersion's code (e.g. 22606), 0 if unknown, MLDSA_ERR_DEVICE if no RCCL can be loaded. */
int mldsa_group_rccl_info(const mldsa_group *g, char *buf, size_t buf_len);

/* ---- single-operation callers: a batcher in front of the batched path ---------------------------
 * The reference's API is ONE operation per call (src/traits.rs:118-308 Signer, 330-362 Verifier, 28-104 KeyGen); a shim that keeps
 * that surface would call this library with n_ops = 1, where a GPU call is all latency.  A batcher coalesces the calls of many
 * host threads: each of the three calls below BLOCKS its caller, copies the arguments into the batch that is currently filling
 * (page-locked arrays), a dispatcher thread of the batcher hands every batch to mldsa_verify_host / mldsa_sign_host /
 * mldsa_keygen_host as one call on `ctx`, and the caller returns with its own result -- the same bytes and verdicts as the
 * batched entry points give.  While one batch runs on the device the next one fills, so batch sizes follow the load;
 * max_wait_us > 0 additionally keeps a batch open that long after its first request (0: never wait for company).
 * Keys: callers pass wire-format bytes with every call; what try_from_bytes makes of them (src/ml_dsa.rs:445-498) -- and A_hat =
 * ExpandA(rho), the pre-compute benches/README.md:4-8 names -- stays in a device-resident table of `cache_keys` slots (0: 1 024; at
 * least max_batch) and is found again by the key's bytes (keyed hash, then memcmp), so a key is expanded once for as long as it stays
 * in the table (FIFO replacement); batches run mldsa_verify_cached_a / mldsa_sign_cached_a on the table.  Private keys in the table
 * are cleared on replacement and when the batcher is destroyed.  Thread-safe; any number of threads may call concurrently.  `ctx`
 * must outlive the batcher; destroy it only when no call is in flight.
 *   mldsa_batcher_verify   PublicKey::try_from_bytes(pk)?.verify(msg, sig, ctx) (lib.rs:364-380, 471-475): *ok = 1 / 0
 *   mldsa_batcher_sign     PrivateKey::try_from_bytes(sk)?.try_sign_with_seed(rnd, msg, ctx) (lib.rs:268-296): MLDSA_ERR_CTX_LEN for a
 *                          ctx longer than 255 bytes (the signature is then all zero)
 *   mldsa_batcher_keygen   KG::keygen_from_seed(xi) (lib.rs:247-250)
 * mode as in mldsa_verify / mldsa_sign (MLDSA_MODE_PREHASH: msg = OID | PH(M)).
 * Platform: the batcher's parking and spinning use Linux futex words and the x86 `pause` hint (the platform ROCm runs on); built with
 * -DMLDSA_BATCHER_PORTABLE (or on another OS / architecture) it falls back to a condition variable and std::this_thread::yield --
 * same semantics, more wake-up latency (csrc/batcher.cpp; both builds run the same CPU test). */
typedef struct mldsa_batcher mldsa_batcher;
typedef struct {
    uint64_t batches, requests, largest_batch; /* batches run, requests served, largest batch */
    uint64_t keys_expanded, key_hits;          /* per batch and distinct key: expanded (try_from_bytes + ExpandA) / found in the table */
} mldsa_batcher_stats;
int mldsa_batcher_create(mldsa_ctx *ctx, int set, size_t max_batch, unsigned max_wait_us, size_t cache_keys, mldsa_batcher **out);
/* Several dispatchers ("lanes") behind the same calls: one context, thread and key table per entry of device_ids, all created and owned
 * by the batcher; whichever lane is idle takes the next batch.  A device may be listed more than once: TWO lanes on one GPU overlap
 * small batches on the device (each is a chain of latency-bound kernels on a fraction of the SIMDs: 64 callers, ML-DSA-65: 364 k
 * instead of 278 k verifications/s, p50 0.17 instead of 0.23 ms; more lanes than that share the device's four hardware queues and
 * lose again; signing gains nothing); one lane per GPU of a node spreads the callers' operations over the GPUs (a key is then
 * expanded once per lane that meets it). */
int mldsa_batcher_create_on(const int *device_ids, int n, int set, size_t max_batch, unsigned max_wait_us, size_t cache_keys, mldsa_batcher **out);
int mldsa_batcher_lanes(const mldsa_batcher *b);
void mldsa_batcher_destroy(mldsa_batcher *b);
int mldsa_batcher_verify(mldsa_batcher *b, int mode, const uint8_t *pk, const uint8_t *msg, size_t msg_len, const uint8_t *ctx, size_t ctx_len,
                         const uint8_t *sig, uint8_t *ok);
int mldsa_batcher_sign(mldsa_batcher *b, int mode, const uint8_t *sk, const uint8_t *msg, size_t msg_len, const uint8_t *ctx, size_t ctx_len,
                       const uint8_t *rnd, uint8_t *sig);
int mldsa_batcher_keygen(mldsa_batcher *b, const uint8_t *xi, uint8_t *pk, uint8_t *sk);
int mldsa_batcher_get_stats(mldsa_batcher *b, mldsa_batcher_stats *out);
/* Key lifetime (the reference's caller owns a ZeroizeOnDrop PrivateKey, src/types.rs:19: dropping it ends the key's life; here the
 * table would keep it).  What the table holds of a key: its wire bytes in page-locked host memory that is excluded from core dumps
 * (what a lookup compares against) and the expanded fields + A_hat in device memory.
 *   mldsa_batcher_forget_key(b, key, key_len)    takes that key (PK_LEN or SK_LEN bytes) out of every lane's table;
 *   mldsa_batcher_flush_keys(b)                  empties the tables;
 *   mldsa_batcher_set_private_key_cache(b, 0)    from now on no private key stays in the table beyond the batch that used it
 *                                                (every signing call then pays try_from_bytes + ExpandA); 1 (default) keeps them.
 * Each returns once the host copy is cleared and, for private keys, the device slot's secret fields (K, s1, s2, t0) are zero; a call
 * waits for the batch a lane is running.  Forgetting a key that is not in the table is not an error. */
int mldsa_batcher_forget_key(mldsa_batcher *b, const uint8_t *key, size_t key_len);
int mldsa_batcher_flush_keys(mldsa_batcher *b);
int mldsa_batcher_set_private_key_cache(mldsa_batcher *b, int on);

#ifdef __cplusplus
}
#endif
#endif /* MLDSA_HIP_H */
