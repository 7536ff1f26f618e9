"""Small verification calls as ONE launch (csrc/kernels_small.hip, MLDSA_OPT_SMALL_FUSED) against the batch pipeline and the oracle.

The single-launch kernel runs the same device functions as the six-launch pipeline; what is new is the orchestration (roles per wave,
the hand-over through a per-op counter, a four-wave tail) and the wave-cooperative forms of mu and SampleInBall.  Everything is
therefore checked three ways: fused = unfused = oracle (verify_internal, src/ml_dsa.rs:351-437), for every parameter set, call sizes
around the cluster layout's boundaries (1, 7, 8, 9, 64 ops), every mode, ragged messages up to many rate blocks, every kind of
refusal the boundary knows (malformed offsets, ctx > 255 bytes, key index out of range), damaged signatures in each section (c~, z,
hints), keys shared and distinct, A_hat kept by the caller (mldsa_verify_cached_a), and many calls back to back (the counters must
return to zero).  Reference tests mirrored: tests/nist_vectors/mod.rs:148-203, tests/integration.rs:63-119, fuzz/fuzz_targets/fuzz_all.rs:25-37."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

OPT_SMALL_FUSED = 13
FUSED_ON = 512   # (counted in ML-DSA-65 ops: 274 ML-DSA-87 ops -- every size below stays on the single-launch kernels for all three sets)


@pytest.fixture(scope="module")
def env():
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    hp = HotPath(0)
    assert hp.get_option(OPT_SMALL_FUSED) == 256           # the default
    yield hp, {s: MlDsa(s, hotpath=hp) for s in (44, 65, 87)}
    hp.close()


def _keys(m, pset, n_keys, seed):
    rng = np.random.default_rng(seed)
    keys = [orc.keygen_from_seed(pset, rng.integers(0, 256, 32, dtype=np.uint8).tobytes()) for _ in range(n_keys)]
    pkb = [orc.pk_into_bytes(pset, pk) for pk, _ in keys]
    pks = m.public_keys_from_bytes(torch.frombuffer(bytearray(b"".join(pkb)), dtype=torch.uint8).cuda().view(n_keys, -1))
    return keys, pks


def _batch(pset, keys, n_ops, seed, modes=(0,)):
    """oracle-signed ops with ragged messages / contexts, a third of them damaged in c~, z, the hint section or the message"""
    rng = np.random.default_rng(seed)
    p = orc.params(pset)
    ops = []
    for i in range(n_ops):
        ki = int(rng.integers(0, len(keys)))
        mode = int(modes[i % len(modes)])
        mlen = int(rng.choice([0, 1, 31, 32, 69, 70, 71, 72, 73, 205, 206, 207, 208, 500, 1500]))
        msg = rng.integers(0, 256, mlen, dtype=np.uint8).tobytes()
        ctx = b"" if mode == 1 else rng.integers(0, 256, int(rng.choice([0, 0, 1, 17, 255])), dtype=np.uint8).tobytes()
        sig = orc.sign_internal(pset, keys[ki][1], msg, rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), ctx=ctx, mode=mode)
        kind = i % 6
        if kind in (1, 2, 3):
            b = bytearray(sig)
            lo, hi = {1: (0, p.ctilde_len), 2: (p.ctilde_len, p.sig_len - p.omega - p.k), 3: (p.sig_len - p.omega - p.k, p.sig_len)}[kind]
            b[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
            sig = bytes(b)
        elif kind == 4 and mlen:
            b = bytearray(msg)
            b[int(rng.integers(0, mlen))] ^= 0x10
            msg = bytes(b)
        want = orc.verify_internal(pset, keys[ki][0], msg, sig, ctx=ctx, mode=mode)
        ops.append(dict(key=ki, mode=mode, msg=msg, ctx=ctx, sig=sig, want=want))
    return ops


def _verify(m, hp, pks, ops, fused, mode, a_hat=None):
    hp.set_option(OPT_SMALL_FUSED, FUSED_ON if fused else 0)
    from fips204_amd.ml_dsa import _cat_with_offsets
    n = len(ops)
    mb, mo = _cat_with_offsets([o["msg"] for o in ops], m.device)
    cb, co = _cat_with_offsets([o["ctx"] for o in ops], m.device)
    sg = torch.frombuffer(bytearray(b"".join(o["sig"] for o in ops)), dtype=torch.uint8).cuda().view(n, -1)
    kidx = torch.tensor([o["key"] for o in ops], dtype=torch.int32, device="cuda")
    ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    m.verify_device(pks, mb, mo, sg, ok, n, cb, co, kidx, mode, a_hat=a_hat)
    torch.cuda.synchronize()
    return ok.cpu().numpy().tolist()


@pytest.mark.parametrize("pset", [44, 65, 87])
@pytest.mark.parametrize("n_ops", [1, 7, 8, 9, 64, 200])
def test_fused_equals_pipeline_equals_oracle(env, pset, n_ops):
    hp, sets = env
    m = sets[pset]
    keys, pks = _keys(m, pset, 3, 900 + pset)
    for mode in (0, 1, 2):
        ops = _batch(pset, keys, n_ops, 31 * pset + n_ops + mode, modes=(mode,))
        want = [int(o["want"]) for o in ops]
        got_f = _verify(m, hp, pks, ops, True, mode)
        got_u = _verify(m, hp, pks, ops, False, mode)
        assert got_f == want, (pset, n_ops, mode, "fused vs oracle")
        assert got_u == want, (pset, n_ops, mode, "pipeline vs oracle")
    hp.set_option(OPT_SMALL_FUSED, 256)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_cached_a_hat_path(env, pset):
    """mldsa_verify_cached_a (what the batcher calls): clusters of ONE workgroup, no counter, int32 A_hat rows looked up by key"""
    hp, sets = env
    m = sets[pset]
    keys, pks = _keys(m, pset, 4, 1200 + pset)
    a_hat = m.expand_a_for_keys(pks)
    for n_ops in (1, 5, 64):
        ops = _batch(pset, keys, n_ops, 77 + pset + n_ops)
        want = [int(o["want"]) for o in ops]
        assert _verify(m, hp, pks, ops, True, 0, a_hat=a_hat) == want
        assert _verify(m, hp, pks, ops, False, 0, a_hat=a_hat) == want
    hp.set_option(OPT_SMALL_FUSED, 256)


def test_acvp_sigver_one_op_per_call(env, acvp_sigver):
    """the reference's own call shape: every ACVP sigVer vector as a call of ONE operation (nist_vectors/mod.rs:148-203)"""
    from conftest import PSET
    hp, sets = env
    hp.set_option(OPT_SMALL_FUSED, 256)
    n = 0
    for g in acvp_sigver["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        pkb = bytes.fromhex(g["pk"])
        pks = m.public_keys_from_bytes(torch.frombuffer(bytearray(pkb), dtype=torch.uint8).cuda().view(1, -1))
        for t in g["tests"]:
            got = m.verify(pks, [bytes.fromhex(t["message"])], [bytes.fromhex(t["signature"])], mode=1)
            assert got.tolist() == [t["testPassed"]], (t["tcId"], t["reason"])
            n += 1
    assert n == 45


def test_refusals_are_per_op_and_identical(env):
    """malformed offset pairs, a ctx of 256 bytes and key indices out of range refuse their own op only, the same way on both paths"""
    hp, sets = env
    m = sets[65]
    keys, pks = _keys(m, 65, 2, 4321)
    from fips204_amd.ml_dsa import _cat_with_offsets
    rng = np.random.default_rng(9)
    n = 12
    msgs = [rng.integers(0, 256, 40 + i, dtype=np.uint8).tobytes() for i in range(n)]
    ctxs = [b"c" * (256 if i == 3 else i) for i in range(n)]
    sigs = [orc.sign_internal(65, keys[i % 2][1], msgs[i], bytes(32), ctx=ctxs[i][:255], mode=0) for i in range(n)]
    mb, mo = _cat_with_offsets(msgs, m.device)
    cb, co = _cat_with_offsets(ctxs, m.device)
    sg = torch.frombuffer(bytearray(b"".join(sigs)), dtype=torch.uint8).cuda().view(n, -1)
    kidx = torch.tensor([i % 2 for i in range(n)], dtype=torch.int32, device="cuda")
    kidx[5] = 2                      # out of range (n_keys = 2)
    kidx[6] = -1                     # 0xFFFFFFFF
    mo_bad = mo.clone()
    mo_bad[9] = mo_bad[8] - 1        # a decreasing pair: ops 8 and 9
    want = [1] * n
    for i in (3, 5, 6, 8, 9):
        want[i] = 0
    res = {}
    for fused in (True, False):
        hp.set_option(OPT_SMALL_FUSED, FUSED_ON if fused else 0)
        ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        m.verify_device(pks, mb, mo_bad, sg, ok, n, cb, co, kidx, 0)
        torch.cuda.synchronize()
        res[fused] = ok.cpu().numpy().tolist()
    assert res[True] == want and res[False] == want
    hp.set_option(OPT_SMALL_FUSED, 256)


def test_long_messages_many_rate_blocks(env):
    """mu over hundreds of SHAKE256 blocks on the cooperative sponge (and the block boundaries around the 136-byte rate)"""
    hp, sets = env
    m = sets[44]
    keys, pks = _keys(m, 44, 1, 55)
    rng = np.random.default_rng(3)
    ops = []
    for mlen in (0, 1, 69, 70, 71, 205, 206, 207, 341, 342, 343, 4096, 40000):
        msg = rng.integers(0, 256, mlen, dtype=np.uint8).tobytes()
        ops.append(dict(key=0, mode=0, msg=msg, ctx=b"", sig=orc.sign_internal(44, keys[0][1], msg, bytes(32), ctx=b"", mode=0), want=True))
    assert _verify(m, hp, pks, ops, True, 0) == [1] * len(ops)
    ops[4]["msg"] = ops[4]["msg"][:-1] + bytes([ops[4]["msg"][-1] ^ 1])
    assert _verify(m, hp, pks, ops, True, 0) == [1] * 4 + [0] + [1] * (len(ops) - 5)


def test_counters_return_to_zero_and_calls_interleave(env):
    """300 calls back to back, the three parameter sets and the call sizes interleaved, then the counter array is all zero again"""
    hp, sets = env
    hp.set_option(OPT_SMALL_FUSED, 256)
    prepared = []
    for pset in (44, 65, 87):
        m = sets[pset]
        keys, pks = _keys(m, pset, 2, 7000 + pset)
        for n_ops in (1, 3, 8, 13, 64):
            ops = _batch(pset, keys, n_ops, 5 * pset + n_ops)
            prepared.append((m, pks, ops))
    rng = np.random.default_rng(1)
    for it in range(300):
        m, pks, ops = prepared[int(rng.integers(0, len(prepared)))]
        assert _verify(m, hp, pks, ops, True, 0) == [int(o["want"]) for o in ops], it
    # the library's own view: a debug count over nothing but the workspace would not see the counters, so verify once more
    # with every op -- a dirty counter would make a cluster finish early (wrong verdict) or never (all verdicts stay at the fill value)
    for m, pks, ops in prepared:
        assert _verify(m, hp, pks, ops, True, 0) == [int(o["want"]) for o in ops]


def test_option_bounds(env):
    hp, _ = env
    lib, h = hp.lib, hp._h
    assert lib.mldsa_set_option(h, OPT_SMALL_FUSED, 1025) != 0 and lib.mldsa_set_option(h, OPT_SMALL_FUSED, -1) != 0
    assert lib.mldsa_set_option(h, OPT_SMALL_FUSED, 0) == 0 and hp.get_option(OPT_SMALL_FUSED) == 0
    assert lib.mldsa_set_option(h, OPT_SMALL_FUSED, 256) == 0


@pytest.mark.parametrize("pset,limit", [(44, 480), (65, 256), (87, 137)])
def test_the_limit_counts_ml_dsa_65_ops(env, pset, limit):
    """MLDSA_OPT_SMALL_FUSED counts ML-DSA-65 ops; another set's limit scales with the A_hat polynomials per op (value * 30 / (k l)): with
    the default 256 a verification call of `limit` ops is one launch (stage "verify_small" of the library's own profile), one op more
    runs the batch pipeline -- measured crossovers per set: profiles/r05_ab_small_limits_per_set.txt.  Same verdicts either way."""
    hp, sets = env
    m = sets[pset]
    hp.set_option(OPT_SMALL_FUSED, 256)
    keys, pks = _keys(m, pset, 2, 77 + pset)
    ops = _batch(pset, keys, 6, 78 + pset)
    from fips204_amd.ml_dsa import _cat_with_offsets
    for n, small in ((limit, True), (limit + 1, False)):
        many = [ops[i % len(ops)] for i in range(n)]
        mb, mo = _cat_with_offsets([o["msg"] for o in many], m.device)
        cb, co = _cat_with_offsets([o["ctx"] for o in many], m.device)
        sg = torch.frombuffer(bytearray(b"".join(o["sig"] for o in many)), dtype=torch.uint8).cuda().view(n, -1)
        kidx = torch.tensor([o["key"] for o in many], dtype=torch.int32, device="cuda")
        ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        hp.profile_enable(True)
        try:
            m.verify_device(pks, mb, mo, sg, ok, n, cb, co, kidx, 0)
            torch.cuda.synchronize()
            stages = str(hp.profile_report())
        finally:
            hp.profile_enable(False)
        assert ("verify_small" in stages) == small, (pset, n, stages[:300])
        assert ok.cpu().numpy().tolist() == [int(o["want"]) for o in many], (pset, n)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_keygen_single_launch_equals_pipeline_equals_oracle(env, pset):
    """key_gen_internal + into_bytes (src/ml_dsa.rs:57-134, src/lib.rs:247-250) for small calls as one launch (k_keygen_small): the keys of
    1 ... 64-key calls are byte-identical to the six-launch pipeline's and to the oracle's, call sizes around the cluster layout's
    boundaries, many calls back to back (the per-key counters return to zero), and no secret stays behind in the workspace."""
    hp, sets = env
    m = sets[pset]
    rng = np.random.default_rng(60 + pset)
    for n in (1, 2, 7, 8, 9, 33, 64):
        xi = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
        hp.set_option(OPT_SMALL_FUSED, 256)
        pk_f, sk_f = m.keygen_from_seed(xi)
        torch.cuda.synchronize()
        scanned, nonzero = hp.secret_residue()
        assert scanned > 0 and nonzero == 0, (n, nonzero)
        hp.set_option(OPT_SMALL_FUSED, 0)
        pk_u, sk_u = m.keygen_from_seed(xi)
        torch.cuda.synchronize()
        hp.set_option(OPT_SMALL_FUSED, 256)
        assert torch.equal(pk_f, pk_u) and torch.equal(sk_f, sk_u), n
        pkb, skb = pk_f.cpu().numpy(), sk_f.cpu().numpy()
        for i in sorted({0, n // 2, n - 1}):
            pk_o, sk_o = orc.keygen_from_seed(pset, xi[i])
            assert pkb[i].tobytes() == orc.pk_into_bytes(pset, pk_o) and skb[i].tobytes() == orc.sk_into_bytes(pset, sk_o), (n, i)
    xi = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(5)]
    first = m.keygen_from_seed(xi)
    for _ in range(100):
        again = m.keygen_from_seed(xi)
        assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])


def test_acvp_keygen_one_key_per_call(env, acvp_keygen):
    """the reference's own call shape: every ACVP keyGen vector as a call of ONE key (nist_vectors/mod.rs)"""
    from conftest import PSET
    hp, sets = env
    hp.set_option(OPT_SMALL_FUSED, 256)
    n = 0
    for g in acvp_keygen["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        for t in g["tests"]:
            pk, sk = m.keygen_from_seed([bytes.fromhex(t["seed"])])
            assert pk[0].cpu().numpy().tobytes() == bytes.fromhex(t["pk"]) and sk[0].cpu().numpy().tobytes() == bytes.fromhex(t["sk"]), t["tcId"]
            n += 1
    assert n == 75


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_sign_fused_prologue_and_round_front_equal_pipeline_equal_oracle(env, pset):
    """sign_internal (src/ml_dsa.rs:153-337) for small calls: the prologue as one launch (k_sign_prologue_small, calls of <= 256 ops) and
    the first half of a round as one launch (k_sign_front_small, rounds planned at <= 819 candidate rows), with the bookkeeping between
    rounds folded in (round 0 opened by the prologue, k_compact_small, k_zero_if_done) and the small calls' own speculation rule.  Signatures of
    the fused paths = the batch pipeline's (MLDSA_OPT_SMALL_FUSED = 0) = the oracle's, byte for byte, at call sizes either side of
    both limits, for the three modes, with refused ops (ctx of 256 bytes, key index out of range, a malformed offset pair) in the
    batch, with A_hat kept by the caller (mldsa_sign_cached_a), and no secret left in the workspace afterwards."""
    from fips204_amd.ml_dsa import _cat_with_offsets
    hp, sets = env
    m = sets[pset]
    rng = np.random.default_rng(300 + pset)
    nk = 3
    xi = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
    sk_o = [orc.keygen_from_seed(pset, x)[1] for x in xi]
    a_hat = m.expand_a_for_keys(pks)
    for n in (1, 3, 25, 26, 64, 256, 257):
        for mode in ((0, 1, 2) if n in (3, 26) else (0,)):
            msgs = [rng.integers(0, 256, int(rng.choice([0, 1, 32, 70, 71, 300])), dtype=np.uint8).tobytes() for _ in range(n)]
            ctxs = [b"" if mode == 1 else rng.integers(0, 256, int(rng.choice([0, 0, 5, 255])), dtype=np.uint8).tobytes() for _ in range(n)]
            rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            kidx_h = rng.integers(0, nk, n).astype(np.int32)
            refuse_ctx, refuse_key = (1, 2) if n >= 3 else (None, None)
            if refuse_ctx is not None and mode != 1:
                ctxs[refuse_ctx] = b"z" * 256
            if refuse_key is not None:
                kidx_h[refuse_key] = nk + 5
            mb, mo = _cat_with_offsets(msgs, m.device)
            cb, co = _cat_with_offsets(ctxs, m.device)
            d_rnd, kidx = torch.from_numpy(rnd).cuda(), torch.from_numpy(kidx_h).cuda()
            out = {}
            for label, fused, ah in (("fused", FUSED_ON, None), ("pipeline", 0, None), ("fused_cached_a", FUSED_ON, a_hat)):
                hp.set_option(OPT_SMALL_FUSED, fused)
                sig = torch.full((n, m.SIG_LEN), 0x5A, dtype=torch.uint8, device="cuda")
                st = torch.full((n,), 77, dtype=torch.int32, device="cuda")
                m.sign_device(sks, mb, mo, d_rnd, sig, n, cb, co, kidx, mode, status=st, a_hat=ah)
                torch.cuda.synchronize()
                out[label] = (sig.cpu().numpy(), st.cpu().numpy())
                scanned, nonzero = hp.secret_residue()
                assert scanned > 0 and nonzero == 0, (label, n, nonzero)
            hp.set_option(OPT_SMALL_FUSED, 256)
            for label in ("pipeline", "fused_cached_a"):
                assert np.array_equal(out["fused"][0], out[label][0]) and np.array_equal(out["fused"][1], out[label][1]), (n, mode, label)
            sig, st = out["fused"]
            for i in range(n):
                bad_ctx = len(ctxs[i]) > 255
                bad_key = kidx_h[i] >= nk
                if bad_ctx or bad_key:
                    assert st[i] == (-1 if bad_key else -2) and not sig[i].any(), (n, i, st[i])   # MLDSA_ERR_PARAM / MLDSA_ERR_CTX_LEN, all-zero signature
                elif i < 6 or i == n - 1:
                    assert st[i] == 0 and sig[i].tobytes() == orc.sign_internal(pset, sk_o[kidx_h[i]], msgs[i], rnd[i].tobytes(), ctx=ctxs[i], mode=mode), (n, mode, i)
            good = [i for i in range(n) if len(ctxs[i]) <= 255 and kidx_h[i] < nk]
            ver = m.verify(pks, [msgs[i] for i in good], torch.from_numpy(sig[good]).cuda(), ctxs=[ctxs[i] for i in good], key_idx=kidx_h[good].astype(np.uint32), mode=mode)
            assert ver.all(), (n, mode)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_small_signing_calls_whose_plan_leaves_ops_unfinished(env, pset):
    """A small synchronous call enqueues the clearing of its secrets BEFORE the host has looked at the outcome (k_zero_if_done, behind
    the event the host waits for).  With one planned round of one candidate per op most ops are still unfinished when the host
    looks: the kernel must have left rho'', kappa and the lists alone, the extra rounds (opened by k_make_slots launches again, then by
    k_compact_small) finish every op, the signatures are the oracle's, and nothing secret stays behind afterwards.  The asynchronous
    call (no host wait: MLDSA_ERR_AGAIN for what is left) and the planned-rounds call agree op by op."""
    hp, sets = env
    m = sets[pset]
    rng = np.random.default_rng(900 + pset)
    nk = 2
    xi = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    sks = m.private_keys_from_bytes(sk)
    sk_o = [orc.keygen_from_seed(pset, x)[1] for x in xi]
    old = {o: hp.get_option(o) for o in (3, 6)}
    try:
        for n in (1, 5, 40, 256):
            msgs = [rng.integers(0, 256, 33, dtype=np.uint8).tobytes() for _ in range(n)]
            rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
            kidx = rng.integers(0, nk, n).astype(np.uint32)
            hp.set_option(3, old[3]); hp.set_option(6, old[6])
            want = host_bytes(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx))
            s0 = hp.stats()["sign_extra_rounds"]
            hp.set_option(3, 1)   # MLDSA_OPT_SPEC_MAX: one candidate per op
            hp.set_option(6, 1)   # MLDSA_OPT_SIGN_ROUNDS: one planned round
            for rep in range(3):
                got = host_bytes(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx))
                assert np.array_equal(got, want), (pset, n, rep)
                scanned, nonzero = hp.secret_residue()
                assert scanned > 0 and nonzero == 0, (pset, n, nonzero)
            if n >= 5:
                assert hp.stats()["sign_extra_rounds"] > s0, (pset, n)   # (one op may well be accepted at once)
            for i in range(min(n, 4)):
                assert want[i].tobytes() == orc.sign_internal(pset, sk_o[int(kidx[i])], msgs[i], rnd[i], mode=0), (pset, n, i)
    finally:
        for o, v in old.items():
            hp.set_option(o, v)


def host_bytes(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def test_acvp_siggen_one_op_per_call(env, acvp_siggen):
    """the reference's own call shape: every ACVP sigGen vector as a signing call of ONE operation (nist_vectors/mod.rs:94-146)"""
    from conftest import PSET
    hp, sets = env
    hp.set_option(OPT_SMALL_FUSED, 256)
    n = 0
    for g in acvp_siggen["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        for t in g["tests"]:
            skb = bytes.fromhex(t["sk"])
            sks = m.private_keys_from_bytes(torch.frombuffer(bytearray(skb), dtype=torch.uint8).cuda().view(1, -1))
            rnd = bytes.fromhex(t["rnd"]) if "rnd" in t and t["rnd"] else bytes(32)
            sig = m.try_sign_with_seed(sks, [bytes.fromhex(t["message"])], [rnd], mode=1)
            assert sig[0].cpu().numpy().tobytes() == bytes.fromhex(t["signature"]), t["tcId"]
            n += 1
    assert n == 60


# ------------------------------------------------------------------------------ the -DMLDSA_NO_LATE_ARG build (field.h)
def test_the_build_without_late_arguments_gives_the_same_bytes():
    """A toolchain that lays the kernarg segment out differently fails mldsa_ctx_create's self-test (field.h late_arg); the fallback build
    (`make nolatearg`: late arguments from an LDS copy of the kernel's argument struct) must then be a drop-in: keys, signatures and
    verdicts of one-op, small and 20 000-op calls of all three parameter sets equal the oracle's (ml_dsa.rs:57-134, 153-337, 351-437)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fips204_amd", "csrc")
    lib = os.path.join(root, "tests", "_build", "libmldsa_hip_nolatearg.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(csrc, f)) for f in ("kernels_sign.hip", "kernels_small.hip", "field.h", "pipeline.hip")):
        subprocess.check_call(["make", "-C", csrc, "-j8", "nolatearg"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "nolatearg_scenarios.py"), lib], capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(res) >= 12 and all(res.values()), res


# ------------------------------------------------------------------------------ the fused second half of a small round (k_sign_back_small)
def hashlib_shake(tag, i, n):
    import hashlib
    return hashlib.shake_256(tag + int(i).to_bytes(8, "little")).digest(n)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_small_signing_calls_with_out_of_range_keys_match_the_oracle(env, pset):
    """k_sign_back_small builds the winner's signature on four waves (resolve_coop4); a key whose s2 leaves [-eta, eta] voids the bound the
    one-transform hint rows rest on and takes the reference's two-transform form there as well (ml_dsa.rs:288-306).  Coherent worst-case
    keys (every eta field all-ones) make that matter in about 1 signature in 500: 3 000 signatures in small calls of 1 ... 250 ops, every
    byte against the oracle, and the same calls through the batch pipeline (MLDSA_OPT_SMALL_FUSED = 0)."""
    hp, sets = env
    m = sets[pset]
    rng = np.random.default_rng(4100 + pset)
    pk_o, sk_o = orc.keygen_from_seed(pset, bytes(range(7, 39)))
    good = np.frombuffer(orc.sk_into_bytes(pset, sk_o), dtype=np.uint8)
    p = m.params
    eta_bits = 3 if p.eta == 2 else 4
    s_off, s_len = 128, (p.k + p.l) * 32 * eta_bits
    nk = 4
    sk = np.tile(good, (nk, 1)).copy()
    for i in range(nk):
        sk[i, s_off:s_off + s_len] = 0xFF
        sk[i, 0] ^= i
    sks = m.private_keys_from_bytes(torch.from_numpy(sk).cuda())
    sk_or = [orc.sk_try_from_bytes(pset, sk[i].tobytes()) for i in range(nk)]
    total, done = 3000, 0
    sizes = [1, 2, 5, 26, 64, 250]
    call = 0
    while done < total:
        n = min(sizes[call % len(sizes)], total - done)
        msgs = [hashlib_shake(b"oor-small-msg", done + i, 24) for i in range(n)]
        rnd = [hashlib_shake(b"oor-small-rnd", done + i, 32) for i in range(n)]
        kidx = ((np.arange(n) + call) % nk).astype(np.uint32)
        want = orc.sign_batch_mt(pset, sk_or, kidx, msgs, rnd, 8, 1, mode=1)
        for fused in ((FUSED_ON, 0) if call % 7 == 0 else (FUSED_ON,)):
            hp.set_option(OPT_SMALL_FUSED, fused)
            sig = m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx, mode=1).cpu().numpy()
            bad = [i for i in range(n) if sig[i].tobytes() != want[i]]
            assert not bad, (pset, n, fused, bad[:5])
        hp.set_option(OPT_SMALL_FUSED, 256)
        done += n
        call += 1


@pytest.mark.parametrize("pset,n,keep", [(65, 200, 12), (44, 256, 5), (87, 120, 1)])
def test_a_round_with_far_fewer_ops_than_planned_walks_the_rest_itself(env, pset, n, keep):
    """k_sign_back_small's grid is sized from the PLAN of the round (unfinished ops + 6 sigma, the candidates per op the rule gives there);
    the device's own counts decide.  A call in which most ops are refused at the prologue (a ctx of 256 bytes: lib.rs:274) enters round 0
    with far fewer ops than planned, so every op gets MORE candidates than the grid has workgroups for: the workgroups walk the further
    pieces themselves.  (The other direction -- more ops than the grid's clusters -- is what the extra rounds of
    test_small_signing_calls_whose_plan_leaves_ops_unfinished run.)  Signatures = the batch pipeline's = the oracle's."""
    from fips204_amd.ml_dsa import _cat_with_offsets
    hp, sets = env
    m = sets[pset]
    rng = np.random.default_rng(5200 + pset)
    xi = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(2)]
    pk, sk = m.keygen_from_seed(xi)
    sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
    sk_o = [orc.keygen_from_seed(pset, x)[1] for x in xi]
    live = set(int(i) for i in rng.choice(n, keep, replace=False))
    msgs = [rng.integers(0, 256, 20, dtype=np.uint8).tobytes() for _ in range(n)]
    ctxs = [b"ok" if i in live else b"z" * 256 for i in range(n)]
    rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    kidx_h = rng.integers(0, 2, n).astype(np.int32)
    mb, mo = _cat_with_offsets(msgs, m.device)
    cb, co = _cat_with_offsets(ctxs, m.device)
    d_rnd, kidx = torch.from_numpy(rnd).cuda(), torch.from_numpy(kidx_h).cuda()
    out = {}
    for label, fused in (("fused", FUSED_ON), ("pipeline", 0)):
        hp.set_option(OPT_SMALL_FUSED, fused)
        sig = torch.full((n, m.SIG_LEN), 0x5A, dtype=torch.uint8, device="cuda")
        st = torch.full((n,), 77, dtype=torch.int32, device="cuda")
        m.sign_device(sks, mb, mo, d_rnd, sig, n, cb, co, kidx, 0, status=st)
        torch.cuda.synchronize()
        out[label] = (sig.cpu().numpy(), st.cpu().numpy())
    hp.set_option(OPT_SMALL_FUSED, 256)
    assert np.array_equal(out["fused"][0], out["pipeline"][0]) and np.array_equal(out["fused"][1], out["pipeline"][1])
    sig, st = out["fused"]
    for i in range(n):
        if i in live:
            assert st[i] == 0 and sig[i].tobytes() == orc.sign_internal(pset, sk_o[kidx_h[i]], msgs[i], rnd[i].tobytes(), ctx=ctxs[i], mode=0), i
        else:
            assert st[i] == -2 and not sig[i].any(), i
