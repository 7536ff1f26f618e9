"""mldsa_batcher_*: many host threads each making the reference's ONE-operation calls (src/traits.rs:118-308 Signer, 330-362 Verifier,
28-104 KeyGen; src/lib.rs:247-296, 364-380), coalesced inside the library into batched calls.  Every caller must get exactly what the
reference gives for ITS arguments -- byte-exact keys and signatures, the verdict of its own signature -- whatever it was batched with."""
import ctypes as C
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from fips204_amd import _lib
from fips204_amd.ml_dsa import MODE_INTERNAL, MODE_PREHASH, MODE_PURE, MlDsaBatcher
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    from fips204_amd.hotpath import HotPath
    h = HotPath(0)
    yield h
    h.close()


def requests_for(pset, n, n_keys, seed):
    rng = np.random.default_rng(seed)
    xis = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_keys)]
    keys = [orc.keygen_from_seed(pset, xi) for xi in xis]
    pkb = [orc.pk_into_bytes(pset, pk) for pk, _ in keys]
    skb = [orc.sk_into_bytes(pset, sk) for _, sk in keys]
    reqs = []
    for i in range(n):
        k = int(rng.integers(0, n_keys))
        msg = rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes()
        ctx = rng.integers(0, 256, int(rng.integers(0, 40)) if i % 3 else 0, dtype=np.uint8).tobytes()
        rnd = rng.integers(0, 256, 32, dtype=np.uint8).tobytes() if i % 4 else bytes(32)
        reqs.append((k, msg, ctx, rnd))
    return xis, keys, pkb, skb, reqs


@pytest.mark.parametrize("pset,lanes", [(44, 1), (65, 3), (87, 2)])
def test_many_threads_single_op_calls(hp, pset, lanes):
    """48 threads, 240 keygen + sign + verify requests with their own keys, message and ctx lengths: keys and signatures byte-exact
    with the oracle's, verdicts of good, damaged and wrong-ctx signatures as the oracle gives them; the library ran fewer batches
    than requests (it coalesced) and expanded every key once (the device-resident key table).  lanes > 1: mldsa_batcher_create_on([0] * lanes)
    -- that many dispatchers with a context and a key table each, on the one GPU -- behind the same calls."""
    xis, keys, pkb, skb, reqs = requests_for(pset, 240, 6, 300 + pset)
    b = MlDsaBatcher(pset, hotpath=hp, max_batch=64) if lanes == 1 else MlDsaBatcher(pset, max_batch=64, device_ids=[0] * lanes)
    assert b.lib.mldsa_batcher_lanes(b._b) == lanes
    try:
        with ThreadPoolExecutor(48) as pool:
            made = list(pool.map(b.keygen_from_seed, xis * 8))
            for (pk, sk), i in zip(made, list(range(6)) * 8):
                assert pk == pkb[i] and sk == skb[i]
            sigs = list(pool.map(lambda r: b.sign(skb[r[0]], r[1], r[3], ctx=r[2]), reqs))
            for (k, msg, ctx, rnd), sig in zip(reqs[:60], sigs):
                assert sig == orc.sign_internal(pset, keys[k][1], msg, rnd, ctx=ctx, mode=orc.MODE_PURE)

            def check(j):
                k, msg, ctx, _ = reqs[j]
                sig = bytearray(sigs[j])
                kind = j % 4
                if kind == 1:
                    sig[(j * 131) % len(sig)] ^= 1 << (j % 8)
                use_ctx = ctx + b"x" if kind == 2 else ctx
                use_pk = pkb[(k + 1) % 6] if kind == 3 else pkb[k]
                return b.verify(use_pk, msg, bytes(sig), ctx=use_ctx), kind, (use_pk, msg, bytes(sig), use_ctx)
            res = list(pool.map(check, range(len(reqs))))
        for j, (got, kind, (pk, msg, sig, ctx)) in enumerate(res):
            want = orc.verify_internal(pset, orc.pk_try_from_bytes(pset, pk), msg, sig, ctx=ctx, mode=orc.MODE_PURE)
            assert got == want, (j, kind)
            assert got == (kind == 0), (j, kind)
        st = b.stats()
        # it coalesced.  (How much depends on how long a batch keeps the device busy while Python's threads queue up behind it: with the
        # single-launch kernels of round 5 a batch is two to three times shorter than before and three lanes drain the queue faster still.)
        assert st["requests"] == 48 + 240 + 240 and st["batches"] <= st["requests"] * 3 // 4 and st["largest_batch"] >= 4
        # six public and six private keys: each expanded ONCE per lane (try_from_bytes + ExpandA), found in the table from then on
        assert 12 <= st["keys_expanded"] <= 12 * lanes and st["key_hits"] > 0
    finally:
        b.close()


def test_acvp_vectors_through_single_op_calls(hp, acvp_keygen, acvp_siggen, acvp_sigver):
    """The reference's NIST ACVP vectors (tests/nist_vectors/mod.rs:56-203: 75 keyGen, 60 sigGen, 45 sigVer) through the ONE-operation
    calls of the batcher -- KG::keygen_from_seed, _internal_sign (MLDSA_MODE_INTERNAL, rnd as given or all-zero), _internal_verify --
    issued by 16 threads at once, so that the vectors of one group meet in the same batches.  Byte-exact keys and signatures, the
    expected verdict of every sigVer case (the three failure classes included)."""
    from conftest import PSET
    n = {"keygen": 0, "siggen": 0, "sigver": 0}
    batchers = {ps: MlDsaBatcher(ps, hotpath=hp, max_batch=32) for ps in (44, 65, 87)}
    try:
        with ThreadPoolExecutor(16) as pool:
            for g in acvp_keygen["testGroups"]:
                b = batchers[PSET[g["parameterSet"]]]
                got = list(pool.map(lambda t: b.keygen_from_seed(bytes.fromhex(t["seed"])), g["tests"]))
                for t, (pk, sk) in zip(g["tests"], got):
                    assert pk == bytes.fromhex(t["pk"]) and sk == bytes.fromhex(t["sk"]), t["tcId"]
                    n["keygen"] += 1
            for g in acvp_siggen["testGroups"]:
                b = batchers[PSET[g["parameterSet"]]]
                got = list(pool.map(lambda t: b.sign(bytes.fromhex(t["sk"]), bytes.fromhex(t["message"]),
                                                     bytes.fromhex(t["rnd"]) if "rnd" in t else bytes(32), mode=MODE_INTERNAL), g["tests"]))
                for t, sig in zip(g["tests"], got):
                    assert sig == bytes.fromhex(t["signature"]), t["tcId"]
                    n["siggen"] += 1
            for g in acvp_sigver["testGroups"]:
                b = batchers[PSET[g["parameterSet"]]]
                pk = bytes.fromhex(g["pk"])
                got = list(pool.map(lambda t: b.verify(pk, bytes.fromhex(t["message"]), bytes.fromhex(t["signature"]), mode=MODE_INTERNAL), g["tests"]))
                for t, ok in zip(g["tests"], got):
                    assert ok == t["testPassed"], (t["tcId"], t.get("reason"))
                    n["sigver"] += 1
        assert n == {"keygen": 75, "siggen": 60, "sigver": 45}
        st = {ps: b.stats() for ps, b in batchers.items()}
        assert sum(s_["requests"] for s_ in st.values()) == 180 and all(s_["batches"] < s_["requests"] for s_ in st.values())
    finally:
        for b in batchers.values():
            b.close()


def test_modes_long_ctx_and_large_messages(hp):
    """ML-DSA-65: the three message modes mixed by concurrent callers (a batch holds one mode; the others wait for the next), a ctx
    of 256 bytes (lib.rs:274: Err for sign; 368: false for verify, nothing runs), a 300 000-byte message (larger than the staging
    array of a fresh batch), an empty message."""
    pset = 65
    xis, keys, pkb, skb, _ = requests_for(pset, 1, 2, 99)
    b = MlDsaBatcher(pset, hotpath=hp, max_batch=16)
    rng = np.random.default_rng(4)
    try:
        big = rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
        oid_ph = b"".join(orc.hash_message(b"abc", "SHA512"))
        jobs = []
        for i in range(36):
            mode = (MODE_PURE, MODE_INTERNAL, MODE_PREHASH)[i % 3]
            msg = oid_ph if mode == MODE_PREHASH else big if i == 6 else b"" if i == 9 else bytes([i]) * (i + 1)
            jobs.append((i % 2, mode, msg, b"ctx%d" % i if mode != MODE_INTERNAL else b"", bytes([i]) * 32))
        with ThreadPoolExecutor(12) as pool:
            sigs = list(pool.map(lambda j: b.sign(skb[j[0]], j[2], j[4], ctx=j[3], mode=j[1]), jobs))
            for (k, mode, msg, ctx, rnd), sig in zip(jobs, sigs):
                assert sig == orc.sign_internal(pset, keys[k][1], msg, rnd, ctx=ctx, mode=mode), (mode, len(msg))
            oks = list(pool.map(lambda js: b.verify(pkb[js[0][0]], js[0][2], js[1], ctx=js[0][3], mode=js[0][1]), zip(jobs, sigs)))
            assert all(oks)
            # a signature made in one mode does not verify in another
            assert not b.verify(pkb[jobs[0][0]], jobs[0][2], sigs[0], ctx=jobs[0][3], mode=MODE_INTERNAL)
        with pytest.raises(_lib.MldsaError) as e:
            b.sign(skb[0], b"m", bytes(32), ctx=bytes(256))
        assert e.value.code == _lib.ERR_CTX_LEN
        assert b.verify(pkb[0], b"m", sigs[0], ctx=bytes(256)) is False
        n_before = b.stats()["requests"]
        assert b.verify(pkb[0], b"m", sigs[0], ctx=bytes(300)) is False and b.stats()["requests"] == n_before
    finally:
        b.close()


def test_max_wait_collects_a_batch(hp):
    """max_wait_us = 20 000: eight callers that arrive within a few milliseconds of each other leave in ONE batch"""
    pset = 44
    xis, keys, pkb, skb, reqs = requests_for(pset, 8, 1, 5)
    b = MlDsaBatcher(pset, hotpath=hp, max_batch=64, max_wait_us=20_000)
    try:
        b.keygen_from_seed(xis[0])  # warm-up: the context's workspace
        s0 = b.stats()
        start = threading.Barrier(8)

        def one(r):
            start.wait()
            return b.sign(skb[0], r[1], r[3], ctx=r[2])
        with ThreadPoolExecutor(8) as pool:
            sigs = list(pool.map(one, reqs))
        s1 = b.stats()
        assert s1["requests"] - s0["requests"] == 8 and s1["batches"] - s0["batches"] == 1 and s1["keys_expanded"] - s0["keys_expanded"] == 1
        for r, sig in zip(reqs, sigs):
            assert sig == orc.sign_internal(pset, keys[0][1], r[1], r[3], ctx=r[2], mode=orc.MODE_PURE)
    finally:
        b.close()


def test_key_table_replacement(hp):
    """A table of FOUR slots (cache_keys = max_batch = 4) and twelve keys taking turns: every key is expanded again after it has
    been replaced, a batch never loses a key it uses itself, and every result still equals the oracle's -- verdicts included for
    signatures checked against ANOTHER key that sits in the table (a stale slot would accept or reject wrongly)."""
    pset = 44
    xis, keys, pkb, skb, _ = requests_for(pset, 1, 12, 77)
    b = MlDsaBatcher(pset, hotpath=hp, max_batch=4, cache_keys=1)
    rng = np.random.default_rng(8)
    try:
        jobs = [(int(k), bytes([j]) * (1 + j % 50), rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
                for j, k in enumerate(list(range(12)) * 3 + list(rng.integers(0, 12, 60)))]
        with ThreadPoolExecutor(6) as pool:
            sigs = list(pool.map(lambda j: b.sign(skb[j[0]], j[1], j[2]), jobs))
            for (k, msg, rnd), sig in zip(jobs, sigs):
                assert sig == orc.sign_internal(pset, keys[k][1], msg, rnd, mode=orc.MODE_PURE)
            oks = list(pool.map(lambda js: b.verify(pkb[js[0][0]], js[0][1], js[1]), zip(jobs, sigs)))
            wrong = list(pool.map(lambda js: b.verify(pkb[(js[0][0] + 1) % 12], js[0][1], js[1]), zip(jobs, sigs)))
        assert all(oks) and not any(wrong)
        st = b.stats()
        assert st["keys_expanded"] > 24 and st["largest_batch"] <= 4   # 12 + 12 keys, more than once each
        # one key over and over: expanded once
        s0 = b.stats()
        for i in range(6):
            assert b.verify(pkb[3], jobs[3][1], sigs[3])
        s1 = b.stats()
        assert s1["keys_expanded"] - s0["keys_expanded"] <= 1 and s1["key_hits"] - s0["key_hits"] >= 5
    finally:
        b.close()


def test_batcher_soak(hp):
    """MLDSA_SOAK_S seconds (default 10) of 32 threads making random single-operation calls through ONE batcher: signing, verifying
    (good signatures, signatures of another request, another key), key generation; three modes, messages of 0 ... 3 000 bytes, 40
    keys over a key table of 16 slots (replacement all the time), batches of at most 16.  Every result is compared with what the
    oracle gave for the same request before the threads started."""
    import os
    import random
    import time
    pset = 44
    seconds = float(os.environ.get("MLDSA_SOAK_S", "10"))
    rng = np.random.default_rng(2024)
    n_keys, n_req = 40, 240
    xis = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_keys)]
    keys = [orc.keygen_from_seed(pset, xi) for xi in xis]
    pkb = [orc.pk_into_bytes(pset, pk) for pk, _ in keys]
    skb = [orc.sk_into_bytes(pset, sk) for _, sk in keys]
    oid_ph = b"".join(orc.hash_message(b"soak", "SHA256"))
    reqs = []
    for i in range(n_req):
        mode = (MODE_PURE, MODE_PURE, MODE_INTERNAL, MODE_PREHASH)[i % 4]
        k = int(rng.integers(0, n_keys))
        msg = oid_ph if mode == MODE_PREHASH else rng.integers(0, 256, int(rng.choice([0, 1, 32, 135, 136, 137, 500, 3000])), dtype=np.uint8).tobytes()
        ctx = b"" if mode == MODE_INTERNAL else rng.integers(0, 256, int(rng.choice([0, 0, 7, 255])), dtype=np.uint8).tobytes()
        rnd = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        reqs.append((k, mode, msg, ctx, rnd, orc.sign_internal(pset, keys[k][1], msg, rnd, ctx=ctx, mode=mode)))
    b = MlDsaBatcher(pset, hotpath=hp, max_batch=16, cache_keys=16)
    stop = time.time() + seconds
    errors, counts = [], [0] * 32

    def worker(tid):
        r = random.Random(tid)
        try:
            while time.time() < stop and not errors:
                k, mode, msg, ctx, rnd, sig = reqs[r.randrange(n_req)]
                what = r.randrange(10)
                if what < 4:
                    assert b.sign(skb[k], msg, rnd, ctx=ctx, mode=mode) == sig, "signature differs from the oracle's"
                elif what < 7:
                    assert b.verify(pkb[k], msg, sig, ctx=ctx, mode=mode) is True, "a good signature was rejected"
                elif what < 8:
                    other = reqs[r.randrange(n_req)]
                    same = other[5] == sig or (other[0] == k and other[1] == mode and other[2] == msg and other[3] == ctx)
                    assert b.verify(pkb[k], msg, other[5], ctx=ctx, mode=mode) is same, "a signature of another request was accepted"
                elif what < 9:
                    assert b.verify(pkb[(k + 1) % n_keys], msg, sig, ctx=ctx, mode=mode) is False, "accepted under another key"
                else:
                    j = r.randrange(n_keys)
                    assert b.keygen_from_seed(xis[j]) == (pkb[j], skb[j]), "key pair differs from the oracle's"
                counts[tid] += 1
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    try:
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(32)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:3]
        st = b.stats()
        assert st["requests"] == sum(counts) and st["requests"] > 50 * seconds and st["keys_expanded"] > 2 * n_keys and st["key_hits"] > 0
        print(f"batcher soak: {sum(counts)} calls in {seconds:.0f} s, {st}")
    finally:
        b.close()


def test_batcher_argument_errors(hp):
    lib = hp.lib
    h = C.c_void_p()
    assert lib.mldsa_batcher_create(hp._h, 50, 16, 0, 0, C.byref(h)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_create(hp._h, 65, 0, 0, 0, C.byref(h)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_create(None, 65, 16, 0, 0, C.byref(h)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_create(hp._h, 65, 16, 0, 0, C.byref(h)) == _lib.OK
    ok = C.c_uint8(7)
    assert lib.mldsa_batcher_verify(h, 0, None, b"", 0, b"", 0, b"x", C.byref(ok)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_verify(h, 9, b"x", b"", 0, b"", 0, b"x", C.byref(ok)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_verify(h, 0, b"x", None, 5, b"", 0, b"x", C.byref(ok)) == _lib.ERR_PARAM
    assert lib.mldsa_batcher_keygen(h, None, None, None) == _lib.ERR_PARAM
    lib.mldsa_batcher_destroy(h)
    lib.mldsa_batcher_destroy(None)
