"""Pins oracle/ (the CPU restatement) against every known-answer vector the reference's
own tests hold for the path.  CPU only.

Mirrors: tests/nist_vectors/mod.rs (test_keygen 56-92, test_siggen 94-146, test_sigver
148-203), tests/messages.rs:10-21, tests/integration.rs:63-74 (bad_sig),
src/helpers.rs:187-216, src/conversion.rs:417-488, src/lib.rs:543-545.
"""
import hashlib

import numpy as np
import pytest

from conftest import PSET
from oracle import oracle as orc
from chacha8rng import ChaCha8Rng


def test_shake_matches_hashlib():
    rng = np.random.default_rng(1)
    for n in [0, 1, 33, 34, 66, 135, 136, 137, 167, 168, 169, 500, 2592]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert orc.shake(128, data, 777) == hashlib.shake_128(data).digest(777)
        assert orc.shake(256, data, 700) == hashlib.shake_256(data).digest(700)


def test_check_zeta():  # helpers.rs:190-197
    z = orc.zeta_table()
    assert (z[0], z[1], z[2]) == (4_193_792, 25_847, 5_771_523)


def test_reductions():  # helpers.rs contracts
    L = orc.lib()
    rng = np.random.default_rng(2)
    for a in rng.integers(-2_143_289_343, 2_143_289_343, 2000):
        a = int(a)
        r = L.orc_partial_reduce32(a)
        assert abs(r) < orc.Q and (r - a) % orc.Q == 0
        f = L.orc_full_reduce32(a)
        assert 0 <= f < orc.Q and (f - a) % orc.Q == 0
        c = L.orc_center_mod(a)
        assert -orc.Q // 2 <= c <= orc.Q // 2 and (c - a) % orc.Q == 0
    for a in rng.integers(-(2**31) * orc.Q, (2**31) * orc.Q, 2000):
        a = int(a)
        r = L.orc_mont_reduce(a)
        assert -orc.Q < r < orc.Q and (r * (1 << 32) - a) % orc.Q == 0
    for a in (0, orc.Q, -orc.Q, 12_345_678, -12_345_678):  # helpers.rs:360-375
        r = L.orc_partial_reduce64(a)
        assert abs(r) < 2 * orc.Q and (r - a) % orc.Q == 0


def test_coeff_leaf_pins():  # conversion.rs:422-488
    import ctypes as C
    L = orc.lib()
    out = C.c_int32()
    assert L.orc_coeff_from_three_bytes((C.c_uint8 * 3)(0x12, 0x34, 0x56), C.byref(out)) and out.value == 0x563412
    assert L.orc_coeff_from_three_bytes((C.c_uint8 * 3)(0x12, 0x34, 0x80), C.byref(out)) and out.value == 0x003412
    assert L.orc_coeff_from_three_bytes((C.c_uint8 * 3)(0x01, 0xe0, 0x80), C.byref(out)) and out.value == 0x00e001
    assert not L.orc_coeff_from_three_bytes((C.c_uint8 * 3)(0x01, 0xe0, 0x7f), C.byref(out))
    assert L.orc_coeff_from_half_byte(2, 3, C.byref(out)) and out.value == -1
    assert L.orc_coeff_from_half_byte(4, 8, C.byref(out)) and out.value == -4
    assert not L.orc_coeff_from_half_byte(4, 10, C.byref(out))
    assert not L.orc_coeff_from_half_byte(2, 15, C.byref(out))


def test_ntt_roundtrip_and_definition():
    rng = np.random.default_rng(3)
    w = rng.integers(0, orc.Q, (3, 256), dtype=np.int32)
    back = orc.inv_ntt(orc.ntt(w))
    assert np.array_equal(back, w)
    # Alg 41 definition: w_hat[i] = w(zeta^(2*brv8(i)+1)) -- check a few points by big-int evaluation
    hat = orc.ntt(w[:1])[0].astype(np.int64) % orc.Q
    for i in (0, 1, 77, 255):
        brv = int(f"{i:08b}"[::-1], 2)
        x = pow(1753, 2 * brv + 1, orc.Q)
        acc = 0
        for j in range(255, -1, -1):
            acc = (acc * x + int(w[0, j])) % orc.Q
        assert acc == int(hat[i])


def test_pk0_pins():  # lib.rs:541-545: keygen_from_seed([0x11;32]) -> pk_bytes[0]
    for pset, want in ((44, 197), (65, 177), (87, 16)):
        pk, sk = orc.keygen_from_seed(pset, bytes([0x11] * 32))
        assert orc.pk_into_bytes(pset, pk)[0] == want
        # lib.rs:519: pk == sk.get_public_key()
        assert orc.pk_into_bytes(pset, orc.get_public_key(pset, sk)) == orc.pk_into_bytes(pset, pk)


def test_acvp_keygen(acvp_keygen):  # nist_vectors/mod.rs:56-92
    n = 0
    for g in acvp_keygen["testGroups"]:
        pset = PSET[g["parameterSet"]]
        for t in g["tests"]:
            pk, sk = orc.keygen_from_seed(pset, bytes.fromhex(t["seed"]))
            assert orc.pk_into_bytes(pset, pk) == bytes.fromhex(t["pk"]), t["tcId"]
            assert orc.sk_into_bytes(pset, sk) == bytes.fromhex(t["sk"]), t["tcId"]
            n += 1
    assert n == 75


def test_acvp_siggen(acvp_siggen):  # nist_vectors/mod.rs:94-146
    n, iters_seen = 0, []
    for g in acvp_siggen["testGroups"]:
        pset = PSET[g["parameterSet"]]
        for t in g["tests"]:
            sk = orc.sk_try_from_bytes(pset, bytes.fromhex(t["sk"]))
            rnd = bytes.fromhex(t["rnd"]) if "rnd" in t else bytes(32)
            sig, it = orc.sign_internal(pset, sk, bytes.fromhex(t["message"]), rnd, want_iters=True)
            assert sig == bytes.fromhex(t["signature"]), t["tcId"]
            iters_seen.append(it)
            n += 1
    assert n == 60
    assert max(iters_seen) > 10  # the vectors exercise long rejection loops (max 22)


def test_acvp_sigver(acvp_sigver):  # nist_vectors/mod.rs:148-203
    n = 0
    for g in acvp_sigver["testGroups"]:
        pset = PSET[g["parameterSet"]]
        pk = orc.pk_try_from_bytes(pset, bytes.fromhex(g["pk"]))
        for t in g["tests"]:
            got = orc.verify_internal(pset, pk, bytes.fromhex(t["message"]), bytes.fromhex(t["signature"]))
            assert got == t["testPassed"], (t["tcId"], t["reason"])
            n += 1
    assert n == 45


def test_messages_rs(ref_hex):  # tests/messages.rs:10-21, external (ctx-prefixed) interface
    v = ref_hex["messages_rs"]
    rng = ChaCha8Rng(123)
    xi, rnd = rng.fill_bytes(32), rng.fill_bytes(32)
    pk, sk = orc.keygen_from_seed(44, xi)
    assert orc.sk_into_bytes(44, sk) == bytes.fromhex(v["sk"])
    assert orc.pk_into_bytes(44, pk) == bytes.fromhex(v["pk"])
    sig = orc.sign_internal(44, sk, b"asdf", rnd, ctx=b"", mode=orc.MODE_PURE)
    assert sig == bytes.fromhex(v["sig"])
    assert orc.verify_internal(44, pk, b"asdf", sig, ctx=b"", mode=orc.MODE_PURE)


def test_bad_sig(ref_hex):  # tests/integration.rs:63-74
    v = ref_hex["integration_bad_sig"]
    pk = orc.pk_try_from_bytes(44, bytes.fromhex(v["pk"]))
    msg = bytes.fromhex(v["msg"])
    assert orc.verify_internal(44, pk, msg, bytes.fromhex(v["good_sig"]))
    assert not orc.verify_internal(44, pk, msg, bytes.fromhex(v["bad_sig"]))


def test_44_no_verif():  # tests/integration.rs:79-119
    msg = bytes(range(8))
    rng = ChaCha8Rng(123)
    pk, sk = orc.keygen_from_seed(44, rng.fill_bytes(32))
    sig = orc.sign_internal(44, sk, msg, rng.fill_bytes(32), ctx=b"\x00", mode=orc.MODE_PURE)
    assert orc.verify_internal(44, pk, msg, sig, ctx=b"\x00", mode=orc.MODE_PURE)
    for i in range(8):
        bad = bytearray(msg); bad[i] ^= 0x08
        assert not orc.verify_internal(44, pk, bytes(bad), sig, ctx=b"\x00", mode=orc.MODE_PURE)
    skb = orc.sk_into_bytes(44, sk)
    for i in range(8):
        bad = bytearray(skb); bad[70 + i * 10] ^= 0x08
        sk_bad = orc.sk_try_from_bytes(44, bytes(bad))
        s2 = orc.sign_internal(44, sk_bad, msg, rng.fill_bytes(32), ctx=b"\x00", mode=orc.MODE_PURE)
        assert not orc.verify_internal(44, pk, msg, s2, ctx=b"\x00", mode=orc.MODE_PURE)
    pkb = orc.pk_into_bytes(44, pk)
    for i in range(8):
        bad = bytearray(pkb); bad[i * 10] ^= 0x08
        assert not orc.verify_internal(44, orc.pk_try_from_bytes(44, bytes(bad)), msg, sig, ctx=b"\x00", mode=orc.MODE_PURE)
    for i in range(8):
        bad = bytearray(sig); bad[i * 10] ^= 0x08
        assert not orc.verify_internal(44, pk, msg, bytes(bad), ctx=b"\x00", mode=orc.MODE_PURE)


def test_ctx_too_long():  # lib.rs:274, 368-370, smoke_test 527-528
    pk, sk = orc.keygen_from_seed(44, bytes(32))
    sig = orc.sign_internal(44, sk, b"m", bytes(32), mode=orc.MODE_PURE)
    assert not orc.verify_internal(44, pk, b"m", sig, ctx=bytes(257), mode=orc.MODE_PURE)
    with pytest.raises(ValueError):
        orc.sign_internal(44, sk, b"m", bytes(32), ctx=bytes(257), mode=orc.MODE_PURE)
