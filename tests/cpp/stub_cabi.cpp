// A CPU stand-in for the C ABI (include/mldsa_hip.h) used ONLY to run the C++ host mirror (fips204_amd/host/*.hpp) under
// AddressSanitizer / UndefinedBehaviorSanitizer (sanitizers run on the CPU build only; the GPU pool has none).
// It computes nothing cryptographic.  What it does: every "device" buffer is an exactly-sized heap allocation, every entry
// point READS every input byte its contract entitles it to and WRITES every output byte it owes -- so a wrong length, offset,
// stride or key index in the mirror's argument packing is an ASan report -- and outputs are a deterministic digest of the
// inputs, so the test can check that the right bytes reached the right call (sign -> verify round trips, corruption and
// wrong-key detection, per-op status codes).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mldsa_hip.h"

struct mldsa_ctx { int device; };
struct mldsa_group { std::vector<mldsa_ctx *> c; };

namespace {
thread_local std::string g_err;
int fail(int code, const char *m) { g_err = m; return code; }

const mldsa_params P[3] = {
    {44, 4, 4, 2, 39, 128, 1 << 17, (8380417 - 1) / 88, 80, 78, 32, 1312, 2560, 2420, 768},
    {65, 6, 5, 4, 49, 192, 1 << 19, (8380417 - 1) / 32, 55, 196, 48, 1952, 4032, 3309, 768},
    {87, 8, 7, 2, 60, 256, 1 << 19, (8380417 - 1) / 32, 75, 120, 64, 2592, 4896, 4627, 1024},
};
const mldsa_params *pp(int set) {
    for (auto &p : P) if (p.set == set) return &p;
    return nullptr;
}

struct Fnv {  // 64-bit FNV-1a; reading through it touches every byte
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    }
    template <class T> void val(T v) { add(&v, sizeof v); }
};
void expand(uint64_t seed, uint8_t *out, size_t n) {  // splitmix64 stream: writes every output byte
    for (size_t i = 0; i < n; i++) {
        if (i % 8 == 0) { seed += 0x9E3779B97F4A7C15ull; }
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = (uint8_t)(z >> (8 * (i % 8)));
    }
}
// "expanded" key fields hold the wire bytes, two per coefficient, so that into_bytes can give them back
void bytes_to_polys(const uint8_t *b, size_t nb, int32_t *polys, size_t n_coef) {
    for (size_t i = 0; i < n_coef; i++) {
        const size_t j = 2 * i;
        polys[i] = (int32_t)((j < nb ? b[j] : 0) | ((j + 1 < nb ? b[j + 1] : 0) << 8));
    }
}
void polys_to_bytes(const int32_t *polys, size_t n_coef, uint8_t *b, size_t nb) {
    for (size_t i = 0; i < n_coef; i++) {
        const size_t j = 2 * i;
        if (j < nb) b[j] = (uint8_t)polys[i];
        if (j + 1 < nb) b[j + 1] = (uint8_t)(polys[i] >> 8);
    }
}
uint64_t tr_of_pk(const uint8_t *pk, size_t n) { Fnv f; f.add(pk, n); return f.h; }

// the signature is a digest of what BOTH sides of a key pair know (rho, tr) and of the op's inputs
void make_sig(const mldsa_params *p, int mode, const uint8_t *rho, const uint8_t *tr, const uint8_t *msg, size_t mlen, const uint8_t *ctx,
              size_t clen, uint8_t *sig) {
    Fnv f;
    f.val(p->set); f.val(mode); f.add(rho, 32); f.add(tr, 64); f.val((uint64_t)mlen); f.add(msg, mlen); f.val((uint64_t)clen); f.add(ctx, clen);
    expand(f.h, sig, (size_t)p->sig_len);
}
}  // namespace

extern "C" {
const char *mldsa_last_error(void) { return g_err.c_str(); }
int mldsa_get_params(int set, mldsa_params *out) { const mldsa_params *p = pp(set); if (!p || !out) return fail(MLDSA_ERR_PARAM, "params"); *out = *p; return 0; }
int mldsa_device_count(void) { return 2; }
int mldsa_ctx_create(int device_id, mldsa_ctx **out) {
    if (!out || device_id < 0 || device_id >= 2) return fail(MLDSA_ERR_PARAM, "mldsa_ctx_create: no such device");
    *out = new mldsa_ctx{device_id};
    return 0;
}
void mldsa_ctx_destroy(mldsa_ctx *c) { delete c; }
int mldsa_ctx_device(const mldsa_ctx *c) { return c ? c->device : MLDSA_ERR_PARAM; }
int mldsa_malloc(void **p, size_t n) { *p = n ? std::malloc(n) : nullptr; return (n && !*p) ? fail(MLDSA_ERR_NOMEM, "malloc") : 0; }
int mldsa_ctx_malloc(mldsa_ctx *c, void **p, size_t n) { if (!c) return fail(MLDSA_ERR_PARAM, "NULL ctx"); return mldsa_malloc(p, n); }
int mldsa_free(void *p) { std::free(p); return 0; }
int mldsa_memcpy_h2d(void *d, const void *s, size_t n, void *) { if (n) std::memcpy(d, s, n); return 0; }
int mldsa_memcpy_d2h(void *d, const void *s, size_t n, void *) { if (n) std::memcpy(d, s, n); return 0; }
int mldsa_memset(void *d, int v, size_t n, void *) { if (n) std::memset(d, v, n); return 0; }
int mldsa_stream_sync(void *) { return 0; }

int mldsa_keygen(mldsa_ctx *c, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "keygen");
    for (size_t i = 0; i < n; i++) {
        Fnv f; f.val(set); f.add(xi + 32 * i, 32);
        uint8_t *pki = pk + i * (size_t)p->pk_len, *ski = sk + i * (size_t)p->sk_len;
        expand(f.h, pki, (size_t)p->pk_len);
        expand(f.h ^ 0x55, ski, (size_t)p->sk_len);
        std::memcpy(ski, pki, 32);                                        // rho opens both (encodings.rs:27, 118)
        uint8_t tr[64]; expand(tr_of_pk(pki, (size_t)p->pk_len), tr, 64);
        std::memcpy(ski + 64, tr, 64);                                    // tr = H(pk) sits in sk
    }
    return 0;
}
int mldsa_pk_expand(mldsa_ctx *c, int set, const uint8_t *pk, uint8_t *rho, uint8_t *tr, int32_t *t1, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "pk_expand");
    for (size_t i = 0; i < n; i++) {
        const uint8_t *pki = pk + i * (size_t)p->pk_len;
        std::memcpy(rho + 32 * i, pki, 32);
        expand(tr_of_pk(pki, (size_t)p->pk_len), tr + 64 * i, 64);
        bytes_to_polys(pki + 32, (size_t)p->pk_len - 32, t1 + i * (size_t)p->k * 256, (size_t)p->k * 256);
    }
    return 0;
}
int mldsa_sk_expand(mldsa_ctx *c, int set, const uint8_t *sk, uint8_t *rho, uint8_t *cap_k, uint8_t *tr, int32_t *s1, int32_t *s2, int32_t *t0,
                    size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "sk_expand");
    const size_t body = (size_t)p->sk_len - 128, c1 = (size_t)p->l * 256, c2 = (size_t)p->k * 256;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *ski = sk + i * (size_t)p->sk_len;
        std::memcpy(rho + 32 * i, ski, 32); std::memcpy(cap_k + 32 * i, ski + 32, 32); std::memcpy(tr + 64 * i, ski + 64, 64);
        std::vector<int32_t> all(c1 + 2 * c2);
        bytes_to_polys(ski + 128, body, all.data(), all.size());
        std::memcpy(s1 + i * c1, all.data(), c1 * 4); std::memcpy(s2 + i * c2, all.data() + c1, c2 * 4); std::memcpy(t0 + i * c2, all.data() + c1 + c2, c2 * 4);
    }
    return 0;
}
int mldsa_pk_into_bytes(mldsa_ctx *c, int set, const uint8_t *rho, const int32_t *t1, uint8_t *pk, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "pk_into_bytes");
    for (size_t i = 0; i < n; i++) {
        uint8_t *pki = pk + i * (size_t)p->pk_len;
        std::memcpy(pki, rho + 32 * i, 32);
        polys_to_bytes(t1 + i * (size_t)p->k * 256, (size_t)p->k * 256, pki + 32, (size_t)p->pk_len - 32);
    }
    return 0;
}
int mldsa_sk_into_bytes(mldsa_ctx *c, int set, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
                        const int32_t *t0, uint8_t *sk, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "sk_into_bytes");
    const size_t body = (size_t)p->sk_len - 128, c1 = (size_t)p->l * 256, c2 = (size_t)p->k * 256;
    for (size_t i = 0; i < n; i++) {
        uint8_t *ski = sk + i * (size_t)p->sk_len;
        std::memcpy(ski, rho + 32 * i, 32); std::memcpy(ski + 32, cap_k + 32 * i, 32); std::memcpy(ski + 64, tr + 64 * i, 64);
        std::vector<int32_t> all(c1 + 2 * c2);
        std::memcpy(all.data(), s1 + i * c1, c1 * 4); std::memcpy(all.data() + c1, s2 + i * c2, c2 * 4); std::memcpy(all.data() + c1 + c2, t0 + i * c2, c2 * 4);
        polys_to_bytes(all.data(), all.size(), ski + 128, body);
    }
    return 0;
}
int mldsa_get_public_key(mldsa_ctx *c, int set, const uint8_t *rho, const uint8_t *tr, const int32_t *s1, const int32_t *s2, uint8_t *pk_rho,
                         uint8_t *pk_tr, int32_t *pk_t1, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "get_public_key");
    for (size_t i = 0; i < n; i++) {
        Fnv f; f.add(s1 + i * (size_t)p->l * 256, (size_t)p->l * 1024); f.add(s2 + i * (size_t)p->k * 256, (size_t)p->k * 1024);
        std::memcpy(pk_rho + 32 * i, rho + 32 * i, 32); std::memcpy(pk_tr + 64 * i, tr + 64 * i, 64);
        for (size_t j = 0; j < (size_t)p->k * 256; j++) pk_t1[i * (size_t)p->k * 256 + j] = (int32_t)((f.h + j) & 0x3FFFFF);
    }
    return 0;
}

static int sign_core(const mldsa_params *p, int mode, const uint8_t *rho, const uint8_t *tr, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                     const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, uint8_t *sigs, int32_t *status, size_t n_ops) {
    for (size_t i = 0; i < n_ops; i++) {
        const size_t k = key_idx ? key_idx[i] : i;
        const size_t clen = coff ? (size_t)(coff[i + 1] - coff[i]) : 0;
        int32_t st = MLDSA_OK;
        if (k >= n_keys) st = MLDSA_ERR_PARAM;
        else if (clen > 255) st = MLDSA_ERR_CTX_LEN;
        uint8_t *sig = sigs + i * (size_t)p->sig_len;
        if (st != MLDSA_OK) std::memset(sig, 0, (size_t)p->sig_len);
        else make_sig(p, mode, rho + 32 * k, tr + 64 * k, msgs + moff[i], (size_t)(moff[i + 1] - moff[i]), coff ? ctxs + coff[i] : nullptr, clen, sig);
        if (status) status[i] = st;
    }
    return 0;
}
int mldsa_sign(mldsa_ctx *c, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
               const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff, const uint8_t *ctxs,
               const uint64_t *coff, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "sign");
    Fnv touch;  // the whole key table and the per-op randomness belong to the call
    touch.add(rho, n_keys * 32); touch.add(cap_k, n_keys * 32); touch.add(tr, n_keys * 64); touch.add(s1, n_keys * (size_t)p->l * 1024);
    touch.add(s2, n_keys * (size_t)p->k * 1024); touch.add(t0, n_keys * (size_t)p->k * 1024); touch.add(rnd, n_ops * 32);
    if (key_idx) touch.add(key_idx, n_ops * 4);
    (void)touch.h;
    return sign_core(p, mode, rho, tr, n_keys, key_idx, msgs, moff, ctxs, coff, sigs, status, n_ops);
}
int mldsa_sign_async(mldsa_ctx *c, int set, int mode, const uint8_t *rho, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
                     const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff, const uint8_t *ctxs,
                     const uint64_t *coff, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *s) {
    return mldsa_sign(c, set, mode, rho, cap_k, tr, s1, s2, t0, n_keys, key_idx, msgs, moff, ctxs, coff, rnd, sigs, status, n_ops, s);
}
static int verify_core(const mldsa_params *p, int mode, const uint8_t *rho, const uint8_t *tr, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                       const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops) {
    std::vector<uint8_t> want((size_t)p->sig_len);
    for (size_t i = 0; i < n_ops; i++) {
        const size_t k = key_idx ? key_idx[i] : i;
        const size_t clen = coff ? (size_t)(coff[i + 1] - coff[i]) : 0;
        Fnv t; t.add(sigs + i * (size_t)p->sig_len, (size_t)p->sig_len);
        if (k >= n_keys || clen > 255) { ok[i] = 0; continue; }  // lib.rs:368-370: every failure is `false`
        make_sig(p, mode, rho + 32 * k, tr + 64 * k, msgs + moff[i], (size_t)(moff[i + 1] - moff[i]), coff ? ctxs + coff[i] : nullptr, clen, want.data());
        ok[i] = std::memcmp(want.data(), sigs + i * (size_t)p->sig_len, want.size()) == 0;
    }
    return 0;
}
int mldsa_verify(mldsa_ctx *c, int set, int mode, const uint8_t *rho, const uint8_t *tr, const int32_t *t1, size_t n_keys, const uint32_t *key_idx,
                 const uint8_t *msgs, const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "verify");
    Fnv touch; touch.add(rho, n_keys * 32); touch.add(tr, n_keys * 64); touch.add(t1, n_keys * (size_t)p->k * 1024);
    if (key_idx) touch.add(key_idx, n_ops * 4);
    (void)touch.h;
    return verify_core(p, mode, rho, tr, n_keys, key_idx, msgs, moff, ctxs, coff, sigs, ok, n_ops);
}

int mldsa_verify_pk(mldsa_ctx *c, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff,
                    const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops, void *) {
    return mldsa_verify_host(c, set, mode, pk, n_keys, key_idx, msgs, moff, ctxs, coff, sigs, ok, n_ops);  // same bytes, same verdicts
}

// ---- host-memory entry points: wire-format keys; expanded exactly like the device-side calls above
int mldsa_keygen_host(mldsa_ctx *c, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n) { return mldsa_keygen(c, set, xi, pk, sk, n, nullptr); }
int mldsa_sign_host(mldsa_ctx *c, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff,
                    const uint8_t *ctxs, const uint64_t *coff, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "sign_host");
    if (!(key_idx ? n_keys > 0 : n_keys >= n_ops)) return fail(MLDSA_ERR_PARAM, "mldsa_sign_host: n_keys does not cover the batch");
    if (mldsa_check_offsets(moff, n_ops) || (coff && mldsa_check_offsets(coff, n_ops))) return MLDSA_ERR_PARAM;  // before anything is copied
    std::vector<uint8_t> rho(n_keys * 32), tr(n_keys * 64);
    Fnv touch; touch.add(sk, n_keys * (size_t)p->sk_len); touch.add(rnd, n_ops * 32);
    for (size_t i = 0; i < n_keys; i++) { std::memcpy(&rho[32 * i], sk + i * (size_t)p->sk_len, 32); std::memcpy(&tr[64 * i], sk + i * (size_t)p->sk_len + 64, 64); }
    return sign_core(p, mode, rho.data(), tr.data(), n_keys, key_idx, msgs, moff, ctxs, coff, sigs, status, n_ops);
}
int mldsa_verify_host(mldsa_ctx *c, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff,
                      const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "verify_host");
    if (!(key_idx ? n_keys > 0 : n_keys >= n_ops)) return fail(MLDSA_ERR_PARAM, "mldsa_verify_host: n_keys does not cover the batch");
    if (mldsa_check_offsets(moff, n_ops) || (coff && mldsa_check_offsets(coff, n_ops))) return MLDSA_ERR_PARAM;
    std::vector<uint8_t> rho(n_keys * 32), tr(n_keys * 64);
    for (size_t i = 0; i < n_keys; i++) {
        std::memcpy(&rho[32 * i], pk + i * (size_t)p->pk_len, 32);
        expand(tr_of_pk(pk + i * (size_t)p->pk_len, (size_t)p->pk_len), &tr[64 * i], 64);
    }
    return verify_core(p, mode, rho.data(), tr.data(), n_keys, key_idx, msgs, moff, ctxs, coff, sigs, ok, n_ops);
}

// ---- what csrc/batcher.cpp calls (tests/cpp/test_batcher_tsan.cpp runs the REAL batcher over these stand-ins under ThreadSanitizer):
// page-locked memory = heap memory; "A_hat" of a key = its rho in the first eight coefficients, so that the *_cached_a calls can
// form the same digests as mldsa_sign / mldsa_verify and a stale or foreign table slot shows as a wrong signature / verdict
// (every release is shown to stub_free_hook first, when the test program defines one: tests/cpp/test_batcher_tsan.cpp looks for key bytes
// in what the batcher hands back)
extern "C" void stub_free_hook(const void *p, size_t bytes) __attribute__((weak));
static std::mutex g_host_mu;
static std::map<void *, size_t> g_host_sizes;
int mldsa_host_alloc(void **p, size_t n) {
    *p = n ? std::calloc(1, n) : nullptr;
    if (n && !*p) return fail(MLDSA_ERR_NOMEM, "host_alloc");
    if (*p) { std::lock_guard<std::mutex> lk(g_host_mu); g_host_sizes[*p] = n; }
    return 0;
}
int mldsa_host_free(void *p) {
    size_t n = 0;
    if (p) { std::lock_guard<std::mutex> lk(g_host_mu); auto it = g_host_sizes.find(p); if (it != g_host_sizes.end()) { n = it->second; g_host_sizes.erase(it); } }
    if (p && n && stub_free_hook) stub_free_hook(p, n);
    std::free(p);
    return 0;
}
int mldsa_expand_a(mldsa_ctx *c, int set, const uint8_t *rho, int32_t *a_hat, size_t n, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "expand_a");
    const size_t per = (size_t)p->k * (size_t)p->l * 256;
    for (size_t i = 0; i < n; i++) {
        Fnv f; f.add(rho + 32 * i, 32);
        for (size_t j = 0; j < per; j++) a_hat[i * per + j] = (int32_t)((f.h + j * 2654435761u) & 0x7FFFFF);
        std::memcpy(a_hat + i * per, rho + 32 * i, 32);
    }
    return 0;
}
static void rho_of_a(const mldsa_params *p, const int32_t *a_hat, size_t n_keys, std::vector<uint8_t> &rho) {
    const size_t per = (size_t)p->k * (size_t)p->l * 256;
    rho.resize(n_keys * 32);
    for (size_t i = 0; i < n_keys; i++) std::memcpy(&rho[32 * i], a_hat + i * per, 32);
}
int mldsa_verify_cached_a(mldsa_ctx *c, int set, int mode, const int32_t *a_hat, const uint8_t *tr, const int32_t *t1, size_t n_keys, const uint32_t *key_idx,
                          const uint8_t *msgs, const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops,
                          void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "verify_cached_a");
    (void)t1;  // (the table has slots that were never filled: only the slots the ops name are read)
    std::vector<uint8_t> rho;
    rho_of_a(p, a_hat, n_keys, rho);
    return verify_core(p, mode, rho.data(), tr, n_keys, key_idx, msgs, moff, ctxs, coff, sigs, ok, n_ops);
}
int mldsa_sign_cached_a(mldsa_ctx *c, int set, int mode, const int32_t *a_hat, const uint8_t *cap_k, const uint8_t *tr, const int32_t *s1, const int32_t *s2,
                        const int32_t *t0, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs, const uint64_t *moff, const uint8_t *ctxs,
                        const uint64_t *coff, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops, void *) {
    const mldsa_params *p = pp(set);
    if (!c || !p) return fail(MLDSA_ERR_PARAM, "sign_cached_a");
    (void)cap_k; (void)s1; (void)s2; (void)t0;
    Fnv touch; touch.add(rnd, n_ops * 32); (void)touch.h;
    std::vector<uint8_t> rho;
    rho_of_a(p, a_hat, n_keys, rho);
    return sign_core(p, mode, rho.data(), tr, n_keys, key_idx, msgs, moff, ctxs, coff, sigs, status, n_ops);
}

// ---- groups: the real library's slice arithmetic (csrc/group.hip), run sequentially
int mldsa_group_shard(size_t n_ops, int n_parts, int part, size_t *first, size_t *count) {
    if (n_parts < 1 || part < 0 || part >= n_parts || !first || !count) return fail(MLDSA_ERR_PARAM, "shard");
    const size_t per = (n_ops + (size_t)n_parts - 1) / (size_t)n_parts;
    *first = per * (size_t)part < n_ops ? per * (size_t)part : n_ops;
    *count = (*first + per < n_ops ? *first + per : n_ops) - *first;
    return 0;
}
int mldsa_group_create(const int *ids, int n, mldsa_group **out) {
    if (!ids || n < 1 || !out) return fail(MLDSA_ERR_PARAM, "group_create");
    mldsa_group *g = new mldsa_group();
    for (int i = 0; i < n; i++) {
        mldsa_ctx *c = nullptr;
        if (mldsa_ctx_create(ids[i], &c) != 0) { for (auto *x : g->c) delete x; delete g; return MLDSA_ERR_PARAM; }
        g->c.push_back(c);
    }
    *out = g;
    return 0;
}
void mldsa_group_destroy(mldsa_group *g) { if (!g) return; for (auto *x : g->c) delete x; delete g; }
int mldsa_group_size(const mldsa_group *g) { return g ? (int)g->c.size() : MLDSA_ERR_PARAM; }
mldsa_ctx *mldsa_group_ctx(mldsa_group *g, int i) { return (g && i >= 0 && i < (int)g->c.size()) ? g->c[(size_t)i] : nullptr; }
int mldsa_keygen_host_group(mldsa_group *g, int set, const uint8_t *xi, uint8_t *pk, uint8_t *sk, size_t n) {
    const mldsa_params *p = pp(set);
    if (!g || !p) return fail(MLDSA_ERR_PARAM, "keygen_host_group");
    for (int i = 0; i < (int)g->c.size(); i++) {
        size_t a, cnt; mldsa_group_shard(n, (int)g->c.size(), i, &a, &cnt);
        if (cnt && mldsa_keygen_host(g->c[(size_t)i], set, xi + 32 * a, pk + a * (size_t)p->pk_len, sk + a * (size_t)p->sk_len, cnt)) return MLDSA_ERR_PARAM;
    }
    return 0;
}
int mldsa_sign_host_group(mldsa_group *g, int set, int mode, const uint8_t *sk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                          const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, const uint8_t *rnd, uint8_t *sigs, int32_t *status, size_t n_ops) {
    const mldsa_params *p = pp(set);
    if (!g || !p) return fail(MLDSA_ERR_PARAM, "sign_host_group");
    for (int i = 0; i < (int)g->c.size(); i++) {
        size_t a, cnt; mldsa_group_shard(n_ops, (int)g->c.size(), i, &a, &cnt);
        const size_t kb = key_idx ? 0 : a;
        if (cnt && mldsa_sign_host(g->c[(size_t)i], set, mode, sk + kb * (size_t)p->sk_len, n_keys - kb, key_idx ? key_idx + a : nullptr, msgs, moff + a, ctxs,
                                   coff ? coff + a : nullptr, rnd + 32 * a, sigs + a * (size_t)p->sig_len, status ? status + a : nullptr, cnt)) return MLDSA_ERR_PARAM;
    }
    return 0;
}
int mldsa_verify_host_group(mldsa_group *g, int set, int mode, const uint8_t *pk, size_t n_keys, const uint32_t *key_idx, const uint8_t *msgs,
                            const uint64_t *moff, const uint8_t *ctxs, const uint64_t *coff, const uint8_t *sigs, uint8_t *ok, size_t n_ops) {
    const mldsa_params *p = pp(set);
    if (!g || !p) return fail(MLDSA_ERR_PARAM, "verify_host_group");
    for (int i = 0; i < (int)g->c.size(); i++) {
        size_t a, cnt; mldsa_group_shard(n_ops, (int)g->c.size(), i, &a, &cnt);
        const size_t kb = key_idx ? 0 : a;
        if (cnt && mldsa_verify_host(g->c[(size_t)i], set, mode, pk + kb * (size_t)p->pk_len, n_keys - kb, key_idx ? key_idx + a : nullptr, msgs, moff + a, ctxs,
                                     coff ? coff + a : nullptr, sigs + a * (size_t)p->sig_len, ok + a, cnt)) return MLDSA_ERR_PARAM;
    }
    return 0;
}
// ---- device-resident group calls: worker i = a plain call on context i, sequentially
int mldsa_verify_group(mldsa_group *g, int set, int mode, const mldsa_verify_slice *s, int) {
    if (!g || !s) return fail(MLDSA_ERR_PARAM, "verify_group");
    for (size_t i = 0; i < g->c.size(); i++)
        if (mldsa_verify(g->c[i], set, mode, s[i].rho, s[i].tr, s[i].t1_d2_hat_mont, s[i].n_keys, s[i].key_idx, s[i].msgs, s[i].msg_off, s[i].ctxs, s[i].ctx_off,
                         s[i].sigs, s[i].ok, s[i].n_ops, s[i].stream)) return MLDSA_ERR_PARAM;
    return 0;
}
int mldsa_sign_group(mldsa_group *g, int set, int mode, const mldsa_sign_slice *s, int) {
    if (!g || !s) return fail(MLDSA_ERR_PARAM, "sign_group");
    for (size_t i = 0; i < g->c.size(); i++)
        if (mldsa_sign(g->c[i], set, mode, s[i].rho, s[i].cap_k, s[i].tr, s[i].s_1_hat_mont, s[i].s_2_hat_mont, s[i].t_0_hat_mont, s[i].n_keys, s[i].key_idx,
                       s[i].msgs, s[i].msg_off, s[i].ctxs, s[i].ctx_off, s[i].rnd, s[i].sigs, s[i].status, s[i].n_ops, s[i].stream)) return MLDSA_ERR_PARAM;
    return 0;
}
int mldsa_keygen_group(mldsa_group *g, int set, const mldsa_keygen_slice *s, int) {
    if (!g || !s) return fail(MLDSA_ERR_PARAM, "keygen_group");
    for (size_t i = 0; i < g->c.size(); i++)
        if (s[i].n_keys && mldsa_keygen(g->c[i], set, s[i].xi, s[i].pk, s[i].sk, s[i].n_keys, s[i].stream)) return MLDSA_ERR_PARAM;
    return 0;
}
int mldsa_group_sync(mldsa_group *g) { return g ? 0 : fail(MLDSA_ERR_PARAM, "group_sync"); }
int mldsa_group_allgather(mldsa_group *g, uint8_t *const *bufs, size_t n_ops, int) {
    if (!g || !bufs) return fail(MLDSA_ERR_PARAM, "allgather");
    const int n = (int)g->c.size();
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            size_t a, cnt; mldsa_group_shard(n_ops, n, j, &a, &cnt);
            if (j != i && cnt && bufs[i] != bufs[j]) std::memcpy(bufs[i] + a, bufs[j] + a, cnt);
        }
    return 0;
}
int mldsa_check_offsets(const uint64_t *off, size_t n_ops) {
    if (!off && n_ops) return fail(MLDSA_ERR_PARAM, "offsets");
    for (size_t i = 0; i < n_ops; i++) if (off[i + 1] < off[i]) return fail(MLDSA_ERR_PARAM, "offset table decreases");
    return 0;
}
int mldsa_abi_version(void) { return MLDSA_ABI_VERSION; }
}  // extern "C"
