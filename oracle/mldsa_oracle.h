/*
 * oracle/mldsa_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the hot path of integritychain/fips204 v0.4.6 and of
 * the callers needed to reach the reference's known-answer byte strings.  It exists
 * only to check the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing under fips204_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * algorithm it restates.  Parity status: PINNED -- see tests/test_oracle_kat.py
 * (75 keyGen + 60 sigGen + 45 sigVer ACVP cases from the reference's
 * tests/nist_vectors, tests/messages.rs, tests/integration.rs::bad_sig, the
 * helpers.rs/conversion.rs/lib.rs unit pins).
 *
 * Third-party dependency restated here because its source is not under
 * /root/reference: RustCrypto `sha3 = "0.10.2"` (Cargo.toml:30; SHAKE128/256 =
 * FIPS 202 Keccak-f[1600]); pinned transitively by every KAT byte string.
 */
#ifndef MLDSA_ORACLE_H
#define MLDSA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_Q 8380417
#define ORC_D 13
#define ORC_N 256
#define ORC_KMAX 8
#define ORC_LMAX 7

/* src/lib.rs:639-656, 681-698, 723-740 and derived consts lib.rs:129-131 */
typedef struct {
    int set;          /* 44, 65, 87 */
    int k, l, eta, tau, lambda, gamma1, gamma2, omega, beta;
    int ctilde_len;   /* LAMBDA / 4 */
    int pk_len, sk_len, sig_len, w1_len;
} orc_params;

const orc_params *orc_get_params(int set);

/* types.rs:35-41 / 19-28: expanded keys exactly as the reference holds them */
typedef struct {
    uint8_t rho[32];
    uint8_t tr[64];
    int32_t t1_d2_hat_mont[ORC_KMAX][ORC_N];
} orc_pubkey;

typedef struct {
    uint8_t rho[32];
    uint8_t cap_k[32];
    uint8_t tr[64];
    int32_t s_1_hat_mont[ORC_LMAX][ORC_N];
    int32_t s_2_hat_mont[ORC_KMAX][ORC_N];
    int32_t t_0_hat_mont[ORC_KMAX][ORC_N];
} orc_privkey;

/* ---- FIPS 202 (sha3 crate) ---- */
void orc_shake(int bits /*128|256*/, const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen);
void orc_keccak_f1600(uint64_t s[25]);

/* ---- helpers.rs ---- */
int32_t orc_mont_reduce(int64_t a);
int32_t orc_partial_reduce64(int64_t a);
int32_t orc_partial_reduce32(int32_t a);
int32_t orc_full_reduce32(int32_t a);
int32_t orc_center_mod(int32_t a);
void orc_zeta_table(int32_t out[256]);
void orc_to_mont(const int32_t *in, int32_t *out, size_t n_polys);
void orc_mat_vec_mul(int k, int l, const int32_t *a_hat, const int32_t *u_hat, int32_t *w_hat);
void orc_pointwise_mont(const int32_t *c_hat, const int32_t *v_hat_mont, int32_t *out, size_t n_polys);
int32_t orc_infinity_norm(const int32_t *polys, size_t n_polys);

/* ---- ntt.rs ---- */
void orc_ntt(const int32_t *in, int32_t *out, size_t n_polys);
void orc_inv_ntt(const int32_t *in, int32_t *out, size_t n_polys);

/* ---- hashing.rs ---- */
void orc_sample_in_ball(int tau, const uint8_t *rho, size_t rho_len, int32_t c[256]);
void orc_rej_ntt_poly(const uint8_t seed34[34], int32_t a_hat[256]);
int  orc_rej_bounded_poly(int eta, const uint8_t seed66[66], int32_t a[256]);
void orc_expand_a(int k, int l, const uint8_t rho[32], int32_t *a_hat /* [k][l][256] */);
void orc_expand_s(int k, int l, int eta, const uint8_t rho[64], int32_t *s1, int32_t *s2);
void orc_expand_mask(int l, int gamma1, const uint8_t rho[64], uint16_t mu, int32_t *y);

/* ---- high_low.rs ---- */
void orc_power2round(const int32_t *r, int32_t *r1, int32_t *r0, size_t n_coeffs);
void orc_decompose(int gamma2, int32_t r, int32_t *r1, int32_t *r0);
int32_t orc_high_bits(int gamma2, int32_t r);
int32_t orc_low_bits(int gamma2, int32_t r);
int orc_make_hint(int gamma2, int32_t z, int32_t r);
int32_t orc_use_hint(int gamma2, int32_t h, int32_t r);

/* ---- conversion.rs ---- */
int  orc_coeff_from_three_bytes(const uint8_t b[3], int32_t *out);
int  orc_coeff_from_half_byte(int eta, uint8_t b, int32_t *out);
void orc_bit_pack(const int32_t w[256], int a, int b, uint8_t *out);
int  orc_bit_unpack(const uint8_t *v, size_t vlen, int a, int b, int32_t w[256]);
void orc_hint_bit_pack(int k, int omega, const int32_t *h, uint8_t *y);
int  orc_hint_bit_unpack(int k, int omega, const uint8_t *y, int32_t *h);

/* ---- encodings.rs ---- */
void orc_pk_encode(int set, const uint8_t rho[32], const int32_t *t1, uint8_t *pk);
int  orc_pk_decode(int set, const uint8_t *pk, uint8_t rho[32], int32_t *t1);
void orc_sk_encode(int set, const uint8_t rho[32], const uint8_t k[32], const uint8_t tr[64],
                   const int32_t *s1, const int32_t *s2, const int32_t *t0, uint8_t *sk);
int  orc_sk_decode(int set, const uint8_t *sk, uint8_t rho[32], uint8_t k[32], uint8_t tr[64],
                   int32_t *s1, int32_t *s2, int32_t *t0);
void orc_sig_encode(int set, const uint8_t *c_tilde, const int32_t *z, const int32_t *h, uint8_t *sig);
int  orc_sig_decode(int set, const uint8_t *sig, uint8_t *c_tilde, int32_t *z, int32_t *h);
void orc_w1_encode(int set, const int32_t *w1, uint8_t *out);

/* ---- ml_dsa.rs / lib.rs ---- */
void orc_keygen_from_seed(int set, const uint8_t xi[32], orc_pubkey *pk, orc_privkey *sk);
int  orc_pk_try_from_bytes(int set, const uint8_t *pk_bytes, orc_pubkey *pk);
int  orc_sk_try_from_bytes(int set, const uint8_t *sk_bytes, orc_privkey *sk);
void orc_pk_into_bytes(int set, const orc_pubkey *pk, uint8_t *out);
void orc_sk_into_bytes(int set, const orc_privkey *sk, uint8_t *out);
void orc_get_public_key(int set, const orc_privkey *sk, orc_pubkey *pk);

/* mode: 0 = external pure (0x00|len(ctx)|ctx|M), 1 = nist/internal (tr|M),
 *       2 = pre-hash (0x01|len(ctx)|ctx|OID|PHM; msg holds OID|PHM)            */
int  orc_sign_internal(int set, const orc_privkey *sk, const uint8_t *msg, size_t mlen,
                       const uint8_t *ctx, size_t ctxlen, const uint8_t rnd[32], int mode,
                       uint8_t *sig, int *iterations);
int  orc_verify_internal(int set, const orc_pubkey *pk, const uint8_t *msg, size_t mlen,
                         const uint8_t *ctx, size_t ctxlen, const uint8_t *sig, int mode);

/* verify-arithmetic unit of BASELINE config 2 (ml_dsa.rs:406-417 without ExpandA):
 * w' = inv_ntt(A_hat * ntt(z) - ntt(c) o t1_d2_hat_mont) */
void orc_verify_arith(int k, int l, const int32_t *a_hat, const int32_t *z, const int32_t *c,
                      const int32_t *t1_d2_hat_mont, int32_t *w_out);

/* batch legs used only for the cpu_baseline timing in bench.py */
void orc_verify_batch(int set, const orc_pubkey *pks, size_t n_keys, const uint8_t *msgs,
                      size_t mlen, const uint8_t *sigs, size_t n_ops, int mode, uint8_t *ok);
void orc_sign_batch(int set, const orc_privkey *sks, size_t n_keys, const uint8_t *msgs,
                    size_t mlen, const uint8_t *rnds, size_t n_ops, int mode, uint8_t *sigs);
void orc_verify_arith_batch(int k, int l, const int32_t *a_hat, const int32_t *z, const int32_t *c,
                            const int32_t *t1, int32_t *w_out, size_t n_ops);

/* multi-threaded (pthreads) versions: ops dealt round-robin to n_threads; `repeat` passes */
void orc_verify_batch_mt(int set, const orc_pubkey *pks, const uint32_t *key_idx, const uint8_t *msgs, size_t mlen,
                         const uint8_t *sigs, size_t n_ops, int mode, uint8_t *ok, int n_threads, size_t repeat);
void orc_sign_batch_mt(int set, const orc_privkey *sks, const uint32_t *key_idx, const uint8_t *msgs, size_t mlen,
                       const uint8_t *rnds, size_t n_ops, int mode, uint8_t *sigs, int n_threads, size_t repeat);

/* keygen_from_seed + into_bytes per seed; try_from_bytes + verify / sign per op (the "from wire bytes" units) */
void orc_keygen_batch_mt(int set, const uint8_t *xi, size_t n_keys, uint8_t *pk_out, uint8_t *sk_out, int n_threads, size_t repeat);
void orc_verify_wire_batch_mt(int set, const uint8_t *pk_bytes, const uint32_t *key_idx, const uint8_t *msgs, size_t mlen,
                              const uint8_t *sigs, size_t n_ops, int mode, uint8_t *ok, int n_threads, size_t repeat);
void orc_sign_wire_batch_mt(int set, const uint8_t *sk_bytes, const uint32_t *key_idx, const uint8_t *msgs, size_t mlen,
                            const uint8_t *rnds, size_t n_ops, int mode, uint8_t *sigs, int n_threads, size_t repeat);

#ifdef __cplusplus
}
#endif
#endif
