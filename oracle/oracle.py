"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (as the checker, never as the thing measured or shipped).  The product path in
fips204_amd/ never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MLDSA_ORACLE_LIB: load another build of the same source instead -- tests/test_sanitizers_cpu.py points it at
# liboracle_asan.so (oracle/Makefile) in a subprocess that has the ASan runtime preloaded
_LIB_PATH = os.environ.get("MLDSA_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

Q = 8380417
N = 256
KMAX, LMAX = 8, 7


def build(force=False):
    """Compile oracle/mldsa_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "mldsa_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "set", "k", "l", "eta", "tau", "lambda_", "gamma1", "gamma2", "omega", "beta",
        "ctilde_len", "pk_len", "sk_len", "sig_len", "w1_len")]


class PubKey(C.Structure):
    _fields_ = [("rho", C.c_uint8 * 32), ("tr", C.c_uint8 * 64),
                ("t1_d2_hat_mont", (C.c_int32 * N) * KMAX)]


class PrivKey(C.Structure):
    _fields_ = [("rho", C.c_uint8 * 32), ("cap_k", C.c_uint8 * 32), ("tr", C.c_uint8 * 64),
                ("s_1_hat_mont", (C.c_int32 * N) * LMAX),
                ("s_2_hat_mont", (C.c_int32 * N) * KMAX),
                ("t_0_hat_mont", (C.c_int32 * N) * KMAX)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_get_params.restype = C.POINTER(Params)
        _lib.orc_get_params.argtypes = [C.c_int]
        _lib.orc_mont_reduce.restype = C.c_int32
        _lib.orc_mont_reduce.argtypes = [C.c_int64]
        _lib.orc_partial_reduce64.restype = C.c_int32
        _lib.orc_partial_reduce64.argtypes = [C.c_int64]
        for f in ("orc_partial_reduce32", "orc_full_reduce32", "orc_center_mod"):
            getattr(_lib, f).restype = C.c_int32
            getattr(_lib, f).argtypes = [C.c_int32]
        _lib.orc_infinity_norm.restype = C.c_int32
        _lib.orc_infinity_norm.argtypes = [C.c_void_p, C.c_size_t]
        for f in ("orc_high_bits", "orc_low_bits"):
            getattr(_lib, f).restype = C.c_int32
            getattr(_lib, f).argtypes = [C.c_int, C.c_int32]
        _lib.orc_use_hint.restype = C.c_int32
        _lib.orc_use_hint.argtypes = [C.c_int, C.c_int32, C.c_int32]
        _lib.orc_decompose.restype = None
        _lib.orc_decompose.argtypes = [C.c_int, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        _lib.orc_power2round.restype = None
        _lib.orc_power2round.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib.orc_make_hint.restype = C.c_int
        _lib.orc_make_hint.argtypes = [C.c_int, C.c_int32, C.c_int32]
    return _lib


def params(pset):
    p = lib().orc_get_params(pset)
    if not p:
        raise ValueError(f"unknown parameter set {pset}")
    return p.contents


def _u8(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else (C.c_uint8 * 1)()


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


# ---- FIPS 202 -------------------------------------------------------------
def shake(bits, data, outlen):
    out = (C.c_uint8 * outlen)()
    lib().orc_shake(C.c_int(bits), _u8(data), C.c_size_t(len(data)), out, C.c_size_t(outlen))
    return bytes(out)


def keccak_f1600(state25):
    s = np.ascontiguousarray(state25, dtype=np.uint64).copy()
    lib().orc_keccak_f1600(s.ctypes.data_as(C.c_void_p))
    return s


# ---- helpers.rs / ntt.rs --------------------------------------------------
def zeta_table():
    out = np.zeros(256, dtype=np.int32)
    lib().orc_zeta_table(out.ctypes.data_as(C.c_void_p))
    return out


def ntt(polys):
    a, p = _i32(polys)
    out = np.empty_like(a)
    lib().orc_ntt(p, out.ctypes.data_as(C.c_void_p), C.c_size_t(a.size // N))
    return out


def inv_ntt(polys):
    a, p = _i32(polys)
    out = np.empty_like(a)
    lib().orc_inv_ntt(p, out.ctypes.data_as(C.c_void_p), C.c_size_t(a.size // N))
    return out


def to_mont(polys):
    a, p = _i32(polys)
    out = np.empty_like(a)
    lib().orc_to_mont(p, out.ctypes.data_as(C.c_void_p), C.c_size_t(a.size // N))
    return out


def mat_vec_mul(k, l, a_hat, u_hat):
    a, pa = _i32(a_hat)
    u, pu = _i32(u_hat)
    assert a.size == k * l * N and u.size == l * N
    out = np.empty((k, N), dtype=np.int32)
    lib().orc_mat_vec_mul(C.c_int(k), C.c_int(l), pa, pu, out.ctypes.data_as(C.c_void_p))
    return out


def pointwise_mont(c_hat, v_hat_mont):
    c, pc = _i32(c_hat)
    v, pv = _i32(v_hat_mont)
    out = np.empty_like(v)
    lib().orc_pointwise_mont(pc, pv, out.ctypes.data_as(C.c_void_p), C.c_size_t(v.size // N))
    return out


def infinity_norm(polys):
    a, p = _i32(polys)
    return int(lib().orc_infinity_norm(p, C.c_size_t(a.size // N)))


def verify_arith(k, l, a_hat, z, c, t1):
    a, pa = _i32(a_hat)
    zz, pz = _i32(z)
    cc, pc = _i32(c)
    tt, pt = _i32(t1)
    n_ops = cc.size // N
    out = np.empty((n_ops, k, N), dtype=np.int32)
    lib().orc_verify_arith_batch(C.c_int(k), C.c_int(l), pa, pz, pc, pt,
                                 out.ctypes.data_as(C.c_void_p), C.c_size_t(n_ops))
    return out


# ---- hashing.rs -----------------------------------------------------------
def sample_in_ball(tau, seed):
    out = np.zeros(N, dtype=np.int32)
    lib().orc_sample_in_ball(C.c_int(tau), _u8(seed), C.c_size_t(len(seed)),
                             out.ctypes.data_as(C.c_void_p))
    return out


def expand_a(k, l, rho):
    out = np.zeros((k, l, N), dtype=np.int32)
    lib().orc_expand_a(C.c_int(k), C.c_int(l), _u8(rho), out.ctypes.data_as(C.c_void_p))
    return out


def expand_s(k, l, eta, rho64):
    s1 = np.zeros((l, N), dtype=np.int32)
    s2 = np.zeros((k, N), dtype=np.int32)
    lib().orc_expand_s(C.c_int(k), C.c_int(l), C.c_int(eta), _u8(rho64),
                       s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p))
    return s1, s2


def expand_mask(l, gamma1, rho64, mu):
    y = np.zeros((l, N), dtype=np.int32)
    lib().orc_expand_mask(C.c_int(l), C.c_int(gamma1), _u8(rho64), C.c_uint16(mu),
                          y.ctypes.data_as(C.c_void_p))
    return y


def rej_bounded_poly(eta, seed66):
    out = np.zeros(N, dtype=np.int32)
    used = lib().orc_rej_bounded_poly(C.c_int(eta), _u8(seed66), out.ctypes.data_as(C.c_void_p))
    return out, int(used)


# ---- codecs ---------------------------------------------------------------
def bit_pack(w, a, b, outlen):
    ww, p = _i32(w)
    out = (C.c_uint8 * outlen)()
    lib().orc_bit_pack(p, C.c_int(a), C.c_int(b), out)
    return bytes(out)


def bit_unpack(v, a, b):
    out = np.zeros(N, dtype=np.int32)
    ok = lib().orc_bit_unpack(_u8(v), C.c_size_t(len(v)), C.c_int(a), C.c_int(b),
                              out.ctypes.data_as(C.c_void_p))
    return bool(ok), out


def hint_bit_unpack(k, omega, y):
    h = np.zeros((k, N), dtype=np.int32)
    ok = lib().orc_hint_bit_unpack(C.c_int(k), C.c_int(omega), _u8(y), h.ctypes.data_as(C.c_void_p))
    return bool(ok), h


def sig_decode(pset, sig):
    p = params(pset)
    ct = (C.c_uint8 * p.ctilde_len)()
    z = np.zeros((p.l, N), dtype=np.int32)
    h = np.zeros((p.k, N), dtype=np.int32)
    ok = lib().orc_sig_decode(C.c_int(pset), _u8(sig), ct, z.ctypes.data_as(C.c_void_p),
                              h.ctypes.data_as(C.c_void_p))
    return bool(ok), bytes(ct), z, h


def w1_encode(pset, w1):
    p = params(pset)
    a, pa = _i32(w1)
    out = (C.c_uint8 * p.w1_len)()
    lib().orc_w1_encode(C.c_int(pset), pa, out)
    return bytes(out)


def use_hint_vec(gamma2, h, r):
    L = lib()
    return np.array([L.orc_use_hint(gamma2, int(a), int(b)) for a, b in zip(np.ravel(h), np.ravel(r))],
                    dtype=np.int32).reshape(np.shape(r))


# ---- scheme ---------------------------------------------------------------
def keygen_from_seed(pset, xi):
    pk, sk = PubKey(), PrivKey()
    lib().orc_keygen_from_seed(C.c_int(pset), _u8(xi), C.byref(pk), C.byref(sk))
    return pk, sk


def pk_into_bytes(pset, pk):
    out = (C.c_uint8 * params(pset).pk_len)()
    lib().orc_pk_into_bytes(C.c_int(pset), C.byref(pk), out)
    return bytes(out)


def sk_into_bytes(pset, sk):
    out = (C.c_uint8 * params(pset).sk_len)()
    lib().orc_sk_into_bytes(C.c_int(pset), C.byref(sk), out)
    return bytes(out)


def pk_try_from_bytes(pset, b):
    assert len(b) == params(pset).pk_len
    pk = PubKey()
    if not lib().orc_pk_try_from_bytes(C.c_int(pset), _u8(b), C.byref(pk)):
        raise ValueError("pk decode failed")
    return pk


def sk_try_from_bytes(pset, b):
    assert len(b) == params(pset).sk_len
    sk = PrivKey()
    if not lib().orc_sk_try_from_bytes(C.c_int(pset), _u8(b), C.byref(sk)):
        raise ValueError("sk decode failed")
    return sk


def get_public_key(pset, sk):
    pk = PubKey()
    lib().orc_get_public_key(C.c_int(pset), C.byref(sk), C.byref(pk))
    return pk


MODE_PURE, MODE_INTERNAL, MODE_PREHASH = 0, 1, 2


def hash_message(message, ph):
    """hash_message (src/hashing.rs:316-354): (oid, PH(M)) of HashML-DSA; `ph` in "SHA256" / "SHA512" / "SHAKE128".
    The three hashes are third-party crates in the reference (sha2, sha3); hashlib supplies them here."""
    import hashlib
    oid = bytes([0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02])
    if ph == "SHA256":  # :318-329
        return oid + b"\x01", hashlib.sha256(message).digest()
    if ph == "SHA512":  # :330-340
        return oid + b"\x03", hashlib.sha512(message).digest()
    if ph == "SHAKE128":  # :341-352
        return oid + b"\x0b", hashlib.shake_128(message).digest(32)
    raise ValueError(ph)


def hash_sign(pset, sk, message, rnd, ctx, ph):
    """try_hash_sign_with_rng (src/lib.rs:310-342): M' = 0x01 | len(ctx) | ctx | OID | PH(M) under sign_internal."""
    oid, phm = hash_message(message, ph)
    return sign_internal(pset, sk, oid + phm, rnd, ctx=ctx, mode=MODE_PREHASH)


def hash_verify(pset, pk, message, sig, ctx, ph):
    """hash_verify (src/lib.rs:391-411)"""
    oid, phm = hash_message(message, ph)
    return verify_internal(pset, pk, oid + phm, sig, ctx=ctx, mode=MODE_PREHASH)


def sign_internal(pset, sk, msg, rnd, ctx=b"", mode=MODE_INTERNAL, want_iters=False):
    sig = (C.c_uint8 * params(pset).sig_len)()
    iters = C.c_int(0)
    rc = lib().orc_sign_internal(C.c_int(pset), C.byref(sk), _u8(msg), C.c_size_t(len(msg)),
                                 _u8(ctx), C.c_size_t(len(ctx)), _u8(rnd), C.c_int(mode), sig,
                                 C.byref(iters))
    if rc != 0:
        raise ValueError("ML-DSA.Sign: ctx too long" if rc == -2 else f"sign failed {rc}")
    return (bytes(sig), iters.value) if want_iters else bytes(sig)


def verify_internal(pset, pk, msg, sig, ctx=b"", mode=MODE_INTERNAL):
    if len(sig) != params(pset).sig_len:
        return False
    return bool(lib().orc_verify_internal(C.c_int(pset), C.byref(pk), _u8(msg), C.c_size_t(len(msg)),
                                          _u8(ctx), C.c_size_t(len(ctx)), _u8(sig), C.c_int(mode)))


# ---- multi-threaded batch legs (cpu_baseline of bench.py) ------------------------------
def verify_batch_mt(pset, pks, key_idx, msgs, sigs, n_threads, repeat=1, mode=MODE_PURE):
    """pks: list of PubKey; msgs: list of equal-length byte strings; sigs: list of SIG_LEN byte strings."""
    n = len(msgs)
    arr = (PubKey * len(pks))(*pks)
    kidx = np.ascontiguousarray(key_idx, dtype=np.uint32)
    mlen = len(msgs[0])
    mb, sb = b"".join(msgs), b"".join(sigs)
    ok = (C.c_uint8 * n)()
    lib().orc_verify_batch_mt(C.c_int(pset), arr, kidx.ctypes.data_as(C.c_void_p), mb, C.c_size_t(mlen), sb, C.c_size_t(n),
                              C.c_int(mode), ok, C.c_int(n_threads), C.c_size_t(repeat))
    return np.frombuffer(bytes(ok), dtype=np.uint8).astype(bool)


def sign_batch_mt(pset, sks, key_idx, msgs, rnds, n_threads, repeat=1, mode=MODE_PURE):
    n = len(msgs)
    arr = (PrivKey * len(sks))(*sks)
    kidx = np.ascontiguousarray(key_idx, dtype=np.uint32)
    mlen = len(msgs[0])
    mb, rb = b"".join(msgs), b"".join(rnds)
    out = (C.c_uint8 * (n * params(pset).sig_len))()
    lib().orc_sign_batch_mt(C.c_int(pset), arr, kidx.ctypes.data_as(C.c_void_p), mb, C.c_size_t(mlen), rb, C.c_size_t(n),
                            C.c_int(mode), out, C.c_int(n_threads), C.c_size_t(repeat))
    sl = params(pset).sig_len
    raw = bytes(out)
    return [raw[i * sl:(i + 1) * sl] for i in range(n)]


def keygen_batch_mt(pset, xis, n_threads, repeat=1):
    """keygen_from_seed + into_bytes for every 32-byte seed -> (pk bytes [n, PK_LEN], sk bytes [n, SK_LEN]) as numpy arrays"""
    p = params(pset)
    n = len(xis)
    pk = np.zeros((max(n, 1), p.pk_len), dtype=np.uint8)
    sk = np.zeros((max(n, 1), p.sk_len), dtype=np.uint8)
    lib().orc_keygen_batch_mt(C.c_int(pset), b"".join(xis), C.c_size_t(n), pk.ctypes.data_as(C.c_void_p), sk.ctypes.data_as(C.c_void_p),
                              C.c_int(n_threads), C.c_size_t(repeat))
    return pk[:n], sk[:n]


def verify_wire_batch_mt(pset, pk_bytes, key_idx, msgs, sigs, n_threads, repeat=1, mode=MODE_PURE):
    """PublicKey::try_from_bytes (ml_dsa.rs:477-498) + verify per op: pk_bytes = uint8 [n_keys, PK_LEN]"""
    n = len(msgs)
    kidx = np.ascontiguousarray(key_idx, dtype=np.uint32)
    pkb = np.ascontiguousarray(pk_bytes, dtype=np.uint8)
    ok = (C.c_uint8 * n)()
    lib().orc_verify_wire_batch_mt(C.c_int(pset), pkb.ctypes.data_as(C.c_void_p), kidx.ctypes.data_as(C.c_void_p), b"".join(msgs),
                                   C.c_size_t(len(msgs[0])), b"".join(sigs), C.c_size_t(n), C.c_int(mode), ok, C.c_int(n_threads), C.c_size_t(repeat))
    return np.frombuffer(bytes(ok), dtype=np.uint8).astype(bool)


def sign_wire_batch_mt(pset, sk_bytes, key_idx, msgs, rnds, n_threads, repeat=1, mode=MODE_PURE):
    """PrivateKey::try_from_bytes (ml_dsa.rs:445-469) + sign per op"""
    n = len(msgs)
    kidx = np.ascontiguousarray(key_idx, dtype=np.uint32)
    skb = np.ascontiguousarray(sk_bytes, dtype=np.uint8)
    out = (C.c_uint8 * (n * params(pset).sig_len))()
    lib().orc_sign_wire_batch_mt(C.c_int(pset), skb.ctypes.data_as(C.c_void_p), kidx.ctypes.data_as(C.c_void_p), b"".join(msgs),
                                 C.c_size_t(len(msgs[0])), b"".join(rnds), C.c_size_t(n), C.c_int(mode), out, C.c_int(n_threads), C.c_size_t(repeat))
    sl = params(pset).sig_len
    raw = bytes(out)
    return [raw[i * sl:(i + 1) * sl] for i in range(n)]
