/* A plain C99 host of the C ABI (include/mldsa_hip.h): what a cgo / JNI / Rust -sys binding does, without any of them.
 * Generates keys, signs and verifies a small batch through the host-memory entry points (wire-format keys, host arrays:
 * the shape of the reference's own API, src/traits.rs:118-308, 330-362), damages one signature, hands one op a malformed
 * offset pair through the device-resident entry points, and prints what came back.
 *
 *   gcc -std=c99 -I include tests/cpp/c_host.c -L fips204_amd/csrc -lmldsa_hip -Wl,-rpath,$PWD/fips204_amd/csrc -o c_host && ./c_host
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mldsa_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != MLDSA_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mldsa_last_error());       \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    enum { N = 100, NK = 4 };
    mldsa_params p;
    mldsa_ctx *ctx = NULL;
    if (mldsa_abi_version() != MLDSA_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    CHECK(mldsa_get_params(MLDSA_65, &p));
    CHECK(mldsa_ctx_create(0, &ctx));

    /* KeyGen::keygen_from_seed for NK seeds (randomness is the caller's: here a fixed pattern) */
    uint8_t xi[NK * 32];
    for (int i = 0; i < NK * 32; i++) xi[i] = (uint8_t)(i * 7 + 1);
    uint8_t *pk = malloc((size_t)NK * p.pk_len), *sk = malloc((size_t)NK * p.sk_len);
    CHECK(mldsa_keygen_host(ctx, MLDSA_65, xi, pk, sk, NK));

    /* N messages of different lengths, concatenated, with N + 1 offsets; op i signs with key i mod NK */
    uint64_t off[N + 1];
    uint32_t key_idx[N];
    uint8_t msgs[N * 40], rnd[N * 32];
    off[0] = 0;
    for (int i = 0; i < N; i++) {
        off[i + 1] = off[i] + (uint64_t)(i % 40);
        key_idx[i] = (uint32_t)(i % NK);
    }
    for (size_t i = 0; i < sizeof msgs; i++) msgs[i] = (uint8_t)(i * 13);
    memset(rnd, 0, sizeof rnd); /* all-zero rnd = the deterministic variant (src/lib.rs:282-283) */
    uint8_t *sig = malloc((size_t)N * p.sig_len), ok[N];
    int32_t status[N];
    CHECK(mldsa_sign_host(ctx, MLDSA_65, MLDSA_MODE_PURE, sk, NK, key_idx, msgs, off, NULL, NULL, rnd, sig, status, N));

    sig[7 * (size_t)p.sig_len + 100] ^= 1; /* one damaged signature: verify says false for it, and only for it */
    CHECK(mldsa_verify_host(ctx, MLDSA_65, MLDSA_MODE_PURE, pk, NK, key_idx, msgs, off, NULL, NULL, sig, ok, N));
    int good = 0;
    for (int i = 0; i < N; i++) good += ok[i];
    printf("verified %d of %d (op 7 damaged: ok[7] = %d)\n", good, N, ok[7]);
    if (good != N - 1 || ok[7]) return 2;

    /* a malformed offset table fails a host-memory call as a whole, before anything is copied ... */
    uint64_t bad[N + 1];
    memcpy(bad, off, sizeof bad);
    bad[50] = bad[49] - 1;
    int rc = mldsa_verify_host(ctx, MLDSA_65, MLDSA_MODE_PURE, pk, NK, key_idx, msgs, bad, NULL, NULL, sig, ok, N);
    printf("malformed table through mldsa_verify_host: %d (%s)\n", rc, mldsa_last_error());
    if (rc != MLDSA_ERR_PARAM) return 3;

    /* ... and refuses only the ops it touches in a device-resident call (wire-format keys on the device: mldsa_verify_pk) */
    void *d_pk, *d_msgs, *d_off, *d_kidx, *d_sig, *d_ok;
    CHECK(mldsa_ctx_malloc(ctx, &d_pk, (size_t)NK * p.pk_len));
    CHECK(mldsa_ctx_malloc(ctx, &d_msgs, sizeof msgs));
    CHECK(mldsa_ctx_malloc(ctx, &d_off, sizeof bad));
    CHECK(mldsa_ctx_malloc(ctx, &d_kidx, sizeof key_idx));
    CHECK(mldsa_ctx_malloc(ctx, &d_sig, (size_t)N * p.sig_len));
    CHECK(mldsa_ctx_malloc(ctx, &d_ok, N));
    CHECK(mldsa_memcpy_h2d(d_pk, pk, (size_t)NK * p.pk_len, NULL));
    CHECK(mldsa_memcpy_h2d(d_msgs, msgs, sizeof msgs, NULL));
    CHECK(mldsa_memcpy_h2d(d_off, bad, sizeof bad, NULL));
    CHECK(mldsa_memcpy_h2d(d_kidx, key_idx, sizeof key_idx, NULL));
    CHECK(mldsa_memcpy_h2d(d_sig, sig, (size_t)N * p.sig_len, NULL));
    CHECK(mldsa_verify_pk(ctx, MLDSA_65, MLDSA_MODE_PURE, d_pk, NK, d_kidx, d_msgs, d_off, NULL, NULL, d_sig, d_ok, N, NULL));
    CHECK(mldsa_memcpy_d2h(ok, d_ok, N, NULL));
    CHECK(mldsa_stream_sync(NULL));
    good = 0;
    for (int i = 0; i < N; i++) good += ok[i];
    printf("device-resident call with the same table: %d of %d verified (ops 7, 49 and 50 refused or false)\n", good, N);
    if (good != N - 3 || ok[7] || ok[49] || ok[50]) return 4;

    /* nothing secret is left in the context's workspace or staging buffers (src/types.rs:19: ZeroizeOnDrop) */
    size_t scanned = 0, nonzero = 0;
    CHECK(mldsa_debug_secret_residue(ctx, &scanned, &nonzero));
    printf("secret residue: %zu non-zero bytes of %zu scanned\n", nonzero, scanned);
    if (nonzero) return 5;

    /* the reference's own call shape, one operation per call (src/lib.rs:268-296, 364-380): through a batcher, which would coalesce the
     * calls of concurrent threads; here one thread: op 3 signed again (the same bytes as in the batch: rnd is all-zero) and verified */
    {
        mldsa_batcher *bt = NULL;
        mldsa_batcher_stats st;
        uint8_t *one = malloc((size_t)p.sig_len), ok1 = 0, ok2 = 1;
        const uint8_t *m3 = msgs + off[3];
        const size_t l3 = (size_t)(off[4] - off[3]);
        CHECK(mldsa_batcher_create(ctx, MLDSA_65, 64, 0, 0, &bt));
        CHECK(mldsa_batcher_sign(bt, MLDSA_MODE_PURE, sk + (size_t)key_idx[3] * p.sk_len, m3, l3, NULL, 0, rnd + 3 * 32, one));
        CHECK(mldsa_batcher_verify(bt, MLDSA_MODE_PURE, pk + (size_t)key_idx[3] * p.pk_len, m3, l3, NULL, 0, one, &ok1));
        CHECK(mldsa_batcher_verify(bt, MLDSA_MODE_PURE, pk + (size_t)key_idx[2] * p.pk_len, m3, l3, NULL, 0, one, &ok2));
        CHECK(mldsa_batcher_get_stats(bt, &st));
        printf("batcher: signature %s the batch's, verified %d, under another key %d, %llu requests in %llu batches\n",
               memcmp(one, sig + 3 * (size_t)p.sig_len, (size_t)p.sig_len) ? "differs from" : "equals", ok1, ok2,
               (unsigned long long)st.requests, (unsigned long long)st.batches);
        if (memcmp(one, sig + 3 * (size_t)p.sig_len, (size_t)p.sig_len) || !ok1 || ok2 || st.requests != 3) return 6;
        mldsa_batcher_destroy(bt);
        free(one);
    }

    mldsa_free(d_pk); mldsa_free(d_msgs); mldsa_free(d_off); mldsa_free(d_kidx); mldsa_free(d_sig); mldsa_free(d_ok);
    mldsa_ctx_destroy(ctx);
    free(pk); free(sk); free(sig);
    printf("OK\n");
    return 0;
}
