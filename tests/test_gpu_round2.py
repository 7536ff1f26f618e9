"""GPU parity for what round 2 added behind the C ABI: get_public_key / into_bytes (ACVP keyGen KATs),
key_idx bounds, the device-driven signing loop (planned rounds, extra-round path, mldsa_sign_async),
hipGraph replay, the host-memory entry points, contexts used from other threads, and the BASELINE shapes at
full size (ML-DSA-65 sign at 65 536, ML-DSA-87 verify at 131 072 = config 4's per-GPU slice)."""
import hashlib
import threading

import numpy as np
import pytest
import torch

from conftest import PSET
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    from fips204_amd.hotpath import HotPath
    h = HotPath(0)
    yield h
    h.close()


@pytest.fixture(scope="module")
def sets(hp):
    from fips204_amd.ml_dsa import MlDsa
    return {s: MlDsa(s, hotpath=hp) for s in (44, 65, 87)}


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def shake(tag, i, n=32):
    return hashlib.shake_256(tag + i.to_bytes(8, "little")).digest(n)


def make_batch(m, n_ops, n_keys, tag):
    """n_keys key pairs, n_ops 32-byte messages + rnd, keys dealt round-robin: device-resident inputs"""
    from fips204_amd.ml_dsa import _cat_with_offsets
    xi = [shake(tag + b"key", i) for i in range(n_keys)]
    pk, sk = m.keygen_from_seed(xi)
    msgs = [shake(tag + b"msg", i) for i in range(n_ops)]
    rnd = [shake(tag + b"rnd", i) for i in range(n_ops)]
    mb, mo = _cat_with_offsets(msgs, m.device)
    rn = torch.frombuffer(bytearray(b"".join(rnd)), dtype=torch.uint8).cuda().view(n_ops, 32)
    kidx_host = (np.arange(n_ops) % n_keys).astype(np.uint32)
    kidx = torch.from_numpy(kidx_host.view(np.int32)).cuda()
    return dict(xi=xi, pk=pk, sk=sk, pks=m.public_keys_from_bytes(pk), sks=m.private_keys_from_bytes(sk), msgs=msgs, rnd=rnd,
                mb=mb, mo=mo, rn=rn, kidx=kidx, kidx_host=kidx_host, n=n_ops)


def oracle_sigs(pset, b, idx):
    skb = host(b["sk"])
    sks = {}
    out = []
    for i in idx:
        ki = int(b["kidx_host"][i])
        if ki not in sks:
            sks[ki] = orc.sk_try_from_bytes(pset, skb[ki].tobytes())
        out.append(orc.sign_internal(pset, sks[ki], b["msgs"][i], b["rnd"][i], mode=0))
    return out


# ------------------------------------------------------------------------------ SerDes / get_public_key
def test_get_public_key_and_into_bytes_on_every_acvp_keygen_case(sets, acvp_keygen):
    """Signer::get_public_key (lib.rs:345-349 -> ml_dsa.rs:502-559) and SerDes::into_bytes (lib.rs:427-493) from the
    EXPANDED fields: for every ACVP keyGen case the public key derived from sk, re-encoded, is the KAT's pk, and
    both keys survive try_from_bytes -> into_bytes."""
    n_cases = 0
    for g in acvp_keygen["testGroups"]:
        pset = PSET[g["parameterSet"]]
        m = sets[pset]
        sk_b = [bytes.fromhex(t["sk"]) for t in g["tests"]]
        pk_b = [bytes.fromhex(t["pk"]) for t in g["tests"]]
        sks = m.private_keys_from_bytes(sk_b)
        pks_from_sk = m.get_public_key(sks)
        got_pk = host(m.public_keys_into_bytes(pks_from_sk))
        got_sk = host(m.private_keys_into_bytes(sks))
        pks = m.public_keys_from_bytes(pk_b)
        again_pk = host(m.public_keys_into_bytes(pks))
        for i in range(len(sk_b)):
            assert got_pk[i].tobytes() == pk_b[i], (pset, i, "pk from sk")
            assert got_sk[i].tobytes() == sk_b[i], (pset, i, "sk round trip")
            assert again_pk[i].tobytes() == pk_b[i], (pset, i, "pk round trip")
        # the expanded fields themselves: rho / tr copied from sk (ml_dsa.rs:558), t1_d2_hat_mont equal mod q
        assert torch.equal(pks_from_sk.rho, pks.rho) and torch.equal(pks_from_sk.tr, sks.tr)
        q = orc.Q
        assert torch.equal(pks_from_sk.t1_d2_hat_mont % q, pks.t1_d2_hat_mont % q)
        # and against the oracle's restatement of private_to_public_key for one case per group
        sk_o = orc.sk_try_from_bytes(pset, sk_b[0])
        assert orc.pk_into_bytes(pset, orc.get_public_key(pset, sk_o)) == pk_b[0]
        n_cases += len(sk_b)
    assert n_cases == 75


def test_malformed_secret_key_bytes_round_trip_like_the_reference(sets):
    """expand_private never rejects (conversion.rs:259-260 is vacuous): out-of-range eta fields (a 4-bit field 15 is
    s = -11 for eta = 4) are used as they are, and into_bytes re-encodes eta - s: random bytes must come back the
    way the oracle's restatement of lib.rs:427-465 returns them."""
    m = sets[65]
    rng = np.random.default_rng(5)
    sk = rng.integers(0, 256, (3, m.SK_LEN), dtype=np.uint8)
    sks = m.private_keys_from_bytes(torch.from_numpy(sk).cuda())
    got = host(m.private_keys_into_bytes(sks))
    for i in range(3):
        want = orc.sk_into_bytes(65, orc.sk_try_from_bytes(65, sk[i].tobytes()))
        assert got[i].tobytes() == want


# ------------------------------------------------------------------------------ key_idx bounds
@pytest.mark.parametrize("pset", [44, 87])
def test_out_of_range_key_index_is_refused_per_op(sets, pset):
    m = sets[pset]
    b = make_batch(m, 40, 3, b"bounds%d" % pset)
    sig = torch.empty((b["n"], m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(b["n"], dtype=torch.int32, device="cuda")
    bad = b["kidx"].clone()
    bad[5], bad[17] = 3, 0x7FFFFFF0  # n_keys = 3: both out of range (the second far outside any allocation)
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, b["n"], key_idx=bad, status=st)
    st_h, sig_h = host(st), host(sig)
    assert st_h[5] == -1 and st_h[17] == -1 and (np.delete(st_h, [5, 17]) == 0).all()  # MLDSA_ERR_PARAM for those two only
    assert not sig_h[5].any() and not sig_h[17].any()
    good = [i for i in range(b["n"]) if i not in (5, 17)]
    for i, want in zip(good[:6], oracle_sigs(pset, b, good[:6])):
        assert sig_h[i].tobytes() == want
    # verify: valid signatures everywhere, same two bad indices -> exactly those two are rejected
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, b["n"], key_idx=b["kidx"], status=st)
    ok = torch.zeros(b["n"], dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, b["n"], key_idx=bad)
    ok_h = host(ok)
    assert not ok_h[5] and not ok_h[17] and np.delete(ok_h, [5, 17]).all()
    # the Python wrapper also refuses up front
    with pytest.raises(IndexError):
        m.verify(b["pks"], b["msgs"], sig, key_idx=[3] * b["n"])
    # identity mapping needs a key per op
    from fips204_amd import _lib
    with pytest.raises(_lib.MldsaError):
        m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, b["n"], key_idx=None)


# ------------------------------------------------------------------------------ device-driven loop, graphs
@pytest.mark.parametrize("pset", [44, 65, 87])
def test_sign_graph_replay_async_and_extra_rounds_are_bit_identical(hp, sets, pset):
    """One batch signed (a) with direct launches, (b) three times through the same buffers so that the call is captured
    and replayed as a hipGraph, (c) with mldsa_sign_async, (d) with only 2 rounds enqueued before the host looks (the
    extra-round path): all byte-identical and equal to the oracle."""
    m = sets[pset]
    n = 700
    b = make_batch(m, n, 5, b"loop%d" % pset)
    want = oracle_sigs(pset, b, range(0, n, 7))
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")

    def run(wait=True):
        sig.zero_()
        m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st, wait=wait)
        s = host(sig).copy()
        assert int(host(st).min()) == 0 and int(host(st).max()) == 0
        return s

    hp.set_option(1, 0)  # MLDSA_OPT_GRAPHS off
    s0 = hp.stats()
    direct = run()
    assert hp.stats()["graph_replays"] == s0["graph_replays"]
    for got, w in zip(direct[::7], want):
        assert got.tobytes() == w
    hp.set_option(1, 1)
    s0 = hp.stats()
    for _ in range(3):
        assert np.array_equal(run(), direct)
    s1 = hp.stats()
    assert s1["graphs_captured"] - s0["graphs_captured"] == 1 and s1["graph_replays"] - s0["graph_replays"] == 1
    assert s1["sign_extra_rounds"] == s0["sign_extra_rounds"]  # the planned rounds finished the batch
    for _ in range(3):
        assert np.array_equal(run(wait=False), direct)  # mldsa_sign_async: its own call shape, captured + replayed too
    assert hp.stats()["graph_replays"] - s1["graph_replays"] == 1
    # (d) two planned rounds of one candidate per op cannot finish 700 ops: the synchronous call adds rounds until every
    # op is signed ...
    hp.set_option(6, 2)  # MLDSA_OPT_SIGN_ROUNDS
    hp.set_option(3, 1)  # MLDSA_OPT_SPEC_MAX: no speculation
    try:
        s0 = hp.stats()
        assert np.array_equal(run(), direct)
        assert hp.stats()["sign_extra_rounds"] > s0["sign_extra_rounds"]
        # ... and the asynchronous call reports the unfinished ops instead (status MLDSA_ERR_AGAIN, zero signature)
        sig.zero_()
        m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st, wait=False)
        s, stat = host(sig), host(st)
        again = stat == -5
        assert 0 < again.sum() < n and (stat[~again] == 0).all()
        assert not s[again].any()
        assert np.array_equal(s[~again], direct[~again])
    finally:
        hp.set_option(6, 0)
        hp.set_option(3, 32)


def test_verify_and_keygen_replay_as_graphs(hp, sets):
    m = sets[65]
    b = make_batch(m, 300, 4, b"vgraph")
    sig = m.try_sign_with_seed(b["sks"], b["msgs"], b["rnd"], key_idx=b["kidx_host"])
    sig[11, 40] ^= 1
    ok = torch.zeros(b["n"], dtype=torch.uint8, device="cuda")
    hp.set_option(1, 2)  # MLDSA_OPT_GRAPHS = every op-level call (the default, 1, replays signing calls only)
    s0 = hp.stats()
    outs = []
    for _ in range(4):
        ok.zero_()
        m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, b["n"], key_idx=b["kidx"])
        outs.append(host(ok).copy())
    s1 = hp.stats()
    assert s1["graphs_captured"] - s0["graphs_captured"] == 1 and s1["graph_replays"] - s0["graph_replays"] == 2
    want = np.ones(b["n"], dtype=np.uint8)
    want[11] = 0
    assert all(np.array_equal(o, want) for o in outs)
    xi = torch.frombuffer(bytearray(b"".join(b["xi"])), dtype=torch.uint8).cuda().view(-1, 32)
    pk = torch.empty((4, m.PK_LEN), dtype=torch.uint8, device="cuda")
    sk = torch.empty((4, m.SK_LEN), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        pk.zero_()
        m.keygen_from_seed(xi, out=(pk, sk))
        assert torch.equal(pk, b["pk"]) and torch.equal(sk, b["sk"])
    assert hp.stats()["graphs_captured"] - s1["graphs_captured"] == 1
    hp.set_option(1, 1)
    s2 = hp.stats()
    for _ in range(3):
        m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, b["n"], key_idx=b["kidx"])
    assert hp.stats()["graph_replays"] == s2["graph_replays"]  # default policy: verify is launched directly


def test_reserve_makes_later_calls_allocation_free(sets):
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    h = HotPath(0)
    try:
        m = MlDsa(44, hotpath=h)
        h.reserve(44, 2, 3000)  # MLDSA_OP_SIGN
        g0 = h.stats()["workspace_growths"]
        assert g0 == 1
        b = make_batch(m, 3000, 2, b"reserve")
        sig = m.try_sign_with_seed(b["sks"], b["msgs"], b["rnd"], key_idx=b["kidx_host"])
        assert m.verify(b["pks"], b["msgs"], sig, key_idx=b["kidx_host"]).all()
        assert h.stats()["workspace_growths"] == g0  # keygen, sign and verify of that size all fit
    finally:
        h.close()


# ------------------------------------------------------------------------------ contexts, threads, devices
def test_contexts_keep_their_device_and_work_from_a_fresh_thread(sets):
    """the device was bound only in mldsa_ctx_create.  Two contexts created back to back keep their
    own device id, and an op-level call issued from a brand-new host thread (whose current device the runtime
    initialises to 0, not to the context's) gives the right answer."""
    from fips204_amd import _lib
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    n_dev = torch.cuda.device_count()
    a, c = HotPath(0), HotPath(n_dev - 1)
    try:
        lib = _lib.load()
        assert lib.mldsa_ctx_device(a._h) == 0 and lib.mldsa_ctx_device(c._h) == n_dev - 1
        m = MlDsa(44, hotpath=c)
        res = {}

        def work():
            try:
                xi = [bytes([7]) * 32]
                pk, sk = m.keygen_from_seed(xi)
                sig = m.try_sign_with_seed(m.private_keys_from_bytes(sk), [b"thread"], [bytes(32)])
                res["ok"] = bool(m.verify(m.public_keys_from_bytes(pk), [b"thread"], sig)[0])
                pk_o, sk_o = orc.keygen_from_seed(44, xi[0])
                res["same"] = sig[0].cpu().numpy().tobytes() == orc.sign_internal(44, sk_o, b"thread", bytes(32), mode=0)
            except Exception as e:  # noqa: BLE001
                res["err"] = repr(e)

        t = threading.Thread(target=work)
        t.start()
        t.join(300)
        assert res == {"ok": True, "same": True}, res
        import ctypes as C
        h = C.c_void_p()
        assert lib.mldsa_ctx_create(n_dev + 3, C.byref(h)) < 0 and not h.value  # no such device: an error, not a crash
    finally:
        a.close()
        c.close()


# ------------------------------------------------------------------------------ host-memory entry points
@pytest.mark.parametrize("pinned", [False, True])
def test_host_entry_points_match_the_device_resident_path(sets, pinned):
    """mldsa_keygen_host / mldsa_sign_host / mldsa_verify_host (wire-format keys, host buffers, three sub-batches of which
    the last is ragged) against the device-pointer API: identical keys, signatures and verdicts."""
    m = sets[65]
    n, nk = 2 * 16384 + 77, 6
    b = make_batch(m, n, nk, b"host")
    keep = []

    def mk(a):
        a = np.ascontiguousarray(a)
        if not pinned:
            return a.copy()
        t = torch.empty(max(a.nbytes, 1), dtype=torch.uint8, pin_memory=True)  # page-locked: the DMA path of the library
        keep.append(t)
        v = t.numpy()[:a.nbytes].view(a.dtype).reshape(a.shape)
        v[...] = a
        return v
    xi = mk(np.frombuffer(b"".join(b["xi"]), dtype=np.uint8).reshape(nk, 32))
    pk_h, sk_h = m.keygen_host(xi)
    assert np.array_equal(pk_h, host(b["pk"])) and np.array_equal(sk_h, host(b["sk"]))
    # variable-length messages and ctxs so that the per-sub-batch offsets matter
    rng = np.random.default_rng(11)
    msgs = [rng.integers(0, 256, int(l), dtype=np.uint8).tobytes() for l in rng.integers(0, 200, n)]
    ctxs = [rng.integers(0, 256, int(l), dtype=np.uint8).tobytes() for l in rng.integers(0, 9, n)]
    rnd = mk(np.frombuffer(b"".join(b["rnd"]), dtype=np.uint8).reshape(n, 32))
    kidx = mk(b["kidx_host"])
    mflat, moff = m._cat_host(msgs)
    cflat, coff = m._cat_host(ctxs)
    mflat, moff, cflat, coff = mk(mflat), mk(moff), mk(cflat), mk(coff)
    sig_h = m.sign_host(mk(sk_h), (mflat, moff), rnd, ctxs=(cflat, coff), key_idx=kidx)
    sig_d = host(m.try_sign_with_seed(b["sks"], msgs, b["rnd"], ctxs=ctxs, key_idx=b["kidx_host"]))
    assert np.array_equal(sig_h, sig_d)
    sk_o = orc.sk_try_from_bytes(65, sk_h[int(b["kidx_host"][n - 1])].tobytes())
    assert sig_h[n - 1].tobytes() == orc.sign_internal(65, sk_o, msgs[n - 1], b["rnd"][n - 1], ctx=ctxs[n - 1], mode=0)
    bad = [3, 16383, 16384, n - 1]
    sig_c = mk(sig_h)
    for i in bad:
        sig_c[i, 100 + i % 50] ^= 0x20
    ok = m.verify_host(mk(pk_h), (mflat, moff), sig_c, ctxs=(cflat, coff), key_idx=kidx)
    want = np.ones(n, dtype=bool)
    want[bad] = False
    assert np.array_equal(ok, want)
    # ctx too long is an error for sign, a plain False for verify (lib.rs:274, 368)
    long_ctx = [b"\x01" * 256] + [b""] * 9
    with pytest.raises(ValueError):
        m.sign_host(sk_h, msgs[:10], rnd[:10], ctxs=long_ctx, key_idx=kidx[:10])
    assert not m.verify_host(pk_h, msgs[:10], sig_h[:10], ctxs=long_ctx, key_idx=kidx[:10])[0]


# ------------------------------------------------------------------------------ BASELINE shapes at full size
def test_config3_ml_dsa_65_sign_at_65536(sets):
    """BASELINE config 3 / SURVEY row C3: ML-DSA-65, B = 65 536 (sk, 32-byte msg, hedged rnd) triples: signatures
    byte-exact against the oracle on the first 1024 ops, every signature accepted by the verifier."""
    m = sets[65]
    n = 65536
    b = make_batch(m, n, 1024, b"c3")
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
    assert int(host(st).max()) == 0 and int(host(st).min()) == 0
    skb = host(b["sk"])
    sk_o = [orc.sk_try_from_bytes(65, skb[i].tobytes()) for i in range(1024)]
    want = orc.sign_batch_mt(65, sk_o, b["kidx_host"][:1024], b["msgs"][:1024], b["rnd"][:1024], 8, 1)
    got = host(sig[:1024])
    for i in range(1024):
        assert got[i].tobytes() == want[i], i
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, n, key_idx=b["kidx"])
    assert bool(host(ok).all())
    # default graph policy: a call of this size is launched directly (replay only pays for calls of <= 16384 ops);
    # MLDSA_OPT_GRAPHS = 2 replays it too, with the same signatures
    hp = m.hp
    s0 = hp.stats()
    for _ in range(3):
        m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
    assert hp.stats()["graph_replays"] == s0["graph_replays"]
    first = sig[:4096].clone()
    hp.set_option(1, 2)
    try:
        for _ in range(3):
            m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
        assert hp.stats()["graph_replays"] - s0["graph_replays"] == 1 and torch.equal(sig[:4096], first)
    finally:
        hp.set_option(1, 1)


def test_config4_slice_ml_dsa_87_verify_at_131072(sets):
    """BASELINE config 4's per-GPU slice: 131 072 ML-DSA-87 verifies (one pipeline pass), 1 % of the
    signatures corrupted in a known pattern; the oracle agrees on a sample from both halves."""
    m = sets[87]
    n = 131072
    b = make_batch(m, n, 1024, b"c4")
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"])
    bad = np.arange(37, n, 100)
    cols = (bad * 7919) % m.SIG_LEN
    sig[torch.from_numpy(bad).cuda(), torch.from_numpy(cols).cuda()] ^= 0x10
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, n, key_idx=b["kidx"])
    ok_h = host(ok)
    want = np.ones(n, dtype=np.uint8)
    want[bad] = 0
    # a flipped bit in the hint section can leave a signature valid only if it hits padding that must be zero -> it
    # cannot: every corrupted signature must be rejected, every other accepted
    assert np.array_equal(ok_h, want)
    pkb = host(b["pk"])
    sig_h = host(sig[[5, 37, 65535, 65536, 65637, n - 1]])
    for row, i in enumerate([5, 37, 65535, 65536, 65637, n - 1]):
        pk_o = orc.pk_try_from_bytes(87, pkb[int(b["kidx_host"][i])].tobytes())
        assert orc.verify_internal(87, pk_o, b["msgs"][i], sig_h[row].tobytes(), mode=0) == bool(want[i])


@pytest.mark.parametrize("pset", [65, 87])
def test_multichunk_sign(sets, pset):
    """a signing batch larger than one pipeline pass (262 144 ops): second-pass signatures byte-exact, all verify"""
    m = sets[pset]
    n = 262144 + 1500
    b = make_batch(m, n, 64, b"mc%d" % pset)
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
    assert int(host(st).max()) == 0
    idx = [0, 65535, 65536, 262143, 262144, 262145, n - 1]
    got = host(sig[idx])
    for row, w in enumerate(oracle_sigs(pset, b, idx)):
        assert got[row].tobytes() == w
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, n, key_idx=b["kidx"])
    assert bool(host(ok).all())


# ------------------------------------------------------------------------------ non-canonical hint encodings
@pytest.mark.parametrize("pset", [44, 65, 87])
def test_non_canonical_hint_encodings_are_rejected_like_the_reference(sets, pset):
    """hint_bit_unpack (conversion.rs:340-414) refuses encodings that describe the SAME hint set differently: positions
    out of order inside a polynomial, a non-zero byte behind the last position, limits that run backwards or past
    omega.  The hint masks (and therefore c_tilde') are unchanged by the first two, so only the decoder's own checks
    stand between such a signature and `true` -- the wave-cooperative decoder in k_verify_main tests every position
    independently instead of walking the bytes, and must refuse exactly what the reference refuses."""
    m = sets[pset]
    n = 512
    b = make_batch(m, n, 8, b"hint%d" % pset)
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"])
    base = host(sig).copy()
    p = m.params
    k, omega = p.k, p.omega
    hoff = m.SIG_LEN - omega - k
    rng = np.random.default_rng(pset)
    variants, kinds = [], []
    for i in range(n):
        h = base[i, hoff:].copy()
        lim = [0] + [int(x) for x in h[omega:]]
        total = lim[-1]
        made = []
        # (a) two positions of one polynomial swapped (same set, no longer strictly increasing)
        polys = [j for j in range(k) if lim[j + 1] - lim[j] >= 2]
        if polys:
            j = polys[int(rng.integers(len(polys)))]
            a = lim[j] + int(rng.integers(lim[j + 1] - lim[j] - 1))
            v = h.copy(); v[a], v[a + 1] = v[a + 1], v[a]
            made.append(("swap", v))
            v = h.copy(); v[a + 1] = v[a]  # equal neighbours: >= must refuse, not only >
            made.append(("dup", v))
        # (b) non-zero padding behind the last position
        if total < omega:
            v = h.copy(); v[total + int(rng.integers(omega - total))] = 1 + int(rng.integers(255))
            made.append(("pad", v))
        # (c) limits: one runs backwards / one exceeds omega
        v = h.copy(); jj = int(rng.integers(k)); v[omega + jj] = omega + 1 + int(rng.integers(255 - omega))
        made.append(("limit>omega", v))
        if k >= 2 and total >= 1:
            cand = [j for j in range(1, k) if lim[j] >= 1]
            if cand:
                j = cand[int(rng.integers(len(cand)))]
                v = h.copy(); v[omega + j] = lim[j] - 1  # polynomial j's limit below polynomial j-1's
                made.append(("limit backwards", v))
        for kind, v in made:
            s = base[i].copy(); s[hoff:] = v
            variants.append((i, s)); kinds.append(kind)
    assert {"swap", "dup", "pad", "limit>omega", "limit backwards"} <= set(kinds)
    nv = len(variants)
    sig_v = torch.from_numpy(np.stack([s for _, s in variants])).cuda()
    src = np.array([i for i, _ in variants])
    from fips204_amd.ml_dsa import _cat_with_offsets
    mb, mo = _cat_with_offsets([b["msgs"][i] for i in src], m.device)
    kidx = torch.from_numpy(b["kidx_host"][src].view(np.int32)).cuda()
    ok = torch.ones(nv, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], mb, mo, sig_v, ok, nv, key_idx=kidx)
    got = host(ok)
    # the untouched signatures verify
    ok0 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok0, n, key_idx=b["kidx"])
    assert bool(host(ok0).all())
    # the oracle (serial walk of the reference) on a sample of every kind, the product on all of them
    pkb = host(b["pk"])
    seen = {}
    for row, kind in enumerate(kinds):
        if seen.get(kind, 0) >= 12:
            continue
        seen[kind] = seen.get(kind, 0) + 1
        i = int(src[row])
        pk_o = orc.pk_try_from_bytes(pset, pkb[int(b["kidx_host"][i])].tobytes())
        want = orc.verify_internal(pset, pk_o, b["msgs"][i], variants[row][1].tobytes(), mode=0)
        assert bool(got[row]) == want, (kind, row)
    # 'limit backwards' may by chance still be a well-formed (different) hint -> c_tilde mismatch; every kind is refused
    assert not got.any(), [kinds[r] for r in np.nonzero(got)[0][:5]]


# ------------------------------------------------------------------------------ HashML-DSA (pre-hash) front-end
@pytest.mark.parametrize("pset", [44, 65, 87])
def test_hash_sign_and_hash_verify_match_the_oracle(sets, pset):
    """try_hash_sign_with_seed / hash_verify (src/lib.rs:310-342, 391-411) with Ph = SHA256 / SHA512 / SHAKE128: the
    pre-hash on the host, M' = 0x01 | len(ctx) | ctx | OID | PH(M) on the device; byte-exact against the oracle."""
    m = sets[pset]
    rng = np.random.default_rng(50 + pset)
    pk_o, sk_o = orc.keygen_from_seed(pset, bytes(range(7, 39)))
    pks = m.public_keys_from_bytes([orc.pk_into_bytes(pset, pk_o)])
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(pset, sk_o)])
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 1, 64, 200, 5000)]
    ctxs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 255, 3, 17, 1)]
    rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in msgs]
    for ph in ("SHA256", "SHA512", "SHAKE128"):
        sig = host(m.try_hash_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, ph=ph))
        for i in range(len(msgs)):
            assert sig[i].tobytes() == orc.hash_sign(pset, sk_o, msgs[i], rnd[i], ctxs[i], ph), (ph, i)
            assert orc.hash_verify(pset, pk_o, msgs[i], sig[i].tobytes(), ctxs[i], ph)
        sig_t = torch.from_numpy(sig).cuda()
        assert m.hash_verify(pks, msgs, sig_t, ctxs=ctxs, ph=ph).all()
        other = "SHAKE128" if ph == "SHA256" else "SHA256"
        assert not m.hash_verify(pks, msgs, sig_t, ctxs=ctxs, ph=other).any()  # the OID is part of M'
        assert not m.verify(pks, msgs, sig_t, ctxs=ctxs).any()                 # and 0x01 != 0x00
    with pytest.raises(ValueError):
        m.try_hash_sign_with_seed(sks, msgs[:1], rnd[:1], ctxs=[bytes(256)], ph="SHA512")  # lib.rs:316
    assert not m.hash_verify(pks, msgs[:1], torch.from_numpy(sig[:1]).cuda(), ctxs=[bytes(256)], ph="SHAKE128").any()  # lib.rs:395


def test_os_rng_entry_points(sets):
    """try_keygen / try_sign / try_hash_sign (src/traits.rs:44-46, 156-158, 247-251): xi and rnd from the operating
    system; hedged signatures of one message differ and both verify; an over-long ctx is refused before the
    generator is touched (src/lib.rs:274 precedes 282)."""
    m = sets[44]
    pk, sk = m.try_keygen(2)
    assert not torch.equal(pk[0], pk[1])
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    msgs = [b"one", b"one"]
    kidx = np.zeros(2, dtype=np.uint32)
    sig = m.try_sign(sks, msgs, key_idx=kidx)
    assert not torch.equal(sig[0], sig[1]) and m.verify(pks, msgs, sig, key_idx=kidx).all()
    hsig = m.try_hash_sign(sks, msgs, ph="SHAKE128", key_idx=kidx)
    assert m.hash_verify(pks, msgs, hsig, ph="SHAKE128", key_idx=kidx).all()

    class Counting:
        calls = 0

        def fill_bytes(self, n):
            self.calls += 1
            return bytes(n)

    rng = Counting()
    with pytest.raises(ValueError):
        m.try_sign_with_rng(rng, sks, msgs, ctxs=[b"", bytes(256)], key_idx=kidx)
    with pytest.raises(ValueError):
        m.try_hash_sign_with_rng(rng, sks, msgs, ctxs=[bytes(256), b""], key_idx=kidx)
    assert rng.calls == 0


# ------------------------------------------------------------------------------ signing with out-of-range secret keys
@pytest.mark.parametrize("pset", [44, 65, 87])
@pytest.mark.parametrize("kind", ["random_bytes", "extreme_fields", "few_flips"])
def test_signing_with_out_of_range_secret_keys_matches_the_reference(sets, pset, kind):
    """expand_private accepts every bit pattern (conversion.rs:259-260): an eta field may decode to s = -5 (eta = 2) or
    -11 (eta = 4), and then ||c s||inf can exceed beta = tau * eta -- the bound the signer's short cuts (which
    polynomials can reject at all, HighBits(w - c s2) = HighBits(w) in the hint stage) rest on.  The reference just
    computes with what it decoded; the signatures must still be byte-identical."""
    m = sets[pset]
    rng = np.random.default_rng(100 * pset + len(kind))
    pk_o, sk_o = orc.keygen_from_seed(pset, bytes(range(3, 35)))
    good = np.frombuffer(orc.sk_into_bytes(pset, sk_o), dtype=np.uint8)
    p = m.params
    eta_bits = 3 if p.eta == 2 else 4
    s_off, s_len = 128, (p.k + p.l) * 32 * eta_bits
    n_keys = 4
    sk = np.tile(good, (n_keys, 1)).copy()
    for i in range(n_keys):
        if kind == "random_bytes":
            sk[i, s_off:] = rng.integers(0, 256, m.SK_LEN - s_off, dtype=np.uint8)
        elif kind == "extreme_fields":  # every s1 / s2 field all-ones: s = eta - (2^bits - 1)
            sk[i, s_off:s_off + s_len] = 0xFF
            sk[i, 0] ^= i  # distinct rho per key
        else:
            for pos in rng.integers(s_off, s_off + s_len, 6):
                sk[i, pos] ^= 1 << int(rng.integers(8))
    # coherent extreme keys make ||c s2||inf > beta common enough to matter in about 1 signature in 500 (ML-DSA-65):
    # enough of them that a short cut resting on the bound shows
    n = 3000 if kind == "extreme_fields" else 64
    msgs = [shake(b"oor-msg", i, 40) for i in range(n)]
    rnd = [shake(b"oor-rnd", i) for i in range(n)]
    kidx = (np.arange(n) % n_keys).astype(np.uint32)
    sks = m.private_keys_from_bytes(torch.from_numpy(sk).cuda())
    sig = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx, mode=1))
    sk_or = [orc.sk_try_from_bytes(pset, sk[i].tobytes()) for i in range(n_keys)]
    want = orc.sign_batch_mt(pset, sk_or, kidx, msgs, rnd, 8, 1, mode=1)
    bad = [i for i in range(n) if sig[i].tobytes() != want[i]]
    assert not bad, (kind, len(bad), bad[:5])
    # the same keys as a table larger than the batch (flags per op instead of per key) and one key per op (identity mapping)
    if kind == "extreme_fields":
        big = np.tile(sk, (40, 1))  # 160 keys
        sks_big = m.private_keys_from_bytes(torch.from_numpy(big).cuda())
        k2 = ((np.arange(100) * 7) % 160).astype(np.uint32)
        sig2 = host(m.try_sign_with_seed(sks_big, msgs[:100], rnd[:100], key_idx=k2, mode=1))
        want2 = orc.sign_batch_mt(pset, sk_or, (k2 % n_keys).astype(np.uint32), msgs[:100], rnd[:100], 8, 1, mode=1)
        assert all(sig2[i].tobytes() == want2[i] for i in range(100))
        sig3 = host(m.try_sign_with_seed(sks_big, msgs[:160], rnd[:160], mode=1))  # key_idx None: op i uses key i
        want3 = orc.sign_batch_mt(pset, sk_or, (np.arange(160) % n_keys).astype(np.uint32), msgs[:160], rnd[:160], 8, 1, mode=1)
        assert all(sig3[i].tobytes() == want3[i] for i in range(160))


def test_ml_dsa_44_ct0_bound_and_exact_test_agree(hp, sets):
    """ML-DSA-44: ||c t0||inf < gamma2 (ml_dsa.rs:312) CAN fail (tau * 2^12 > gamma2).  The hint stage's single transform per
    row gives ct0 - cs2, whose maximum + beta bounds ||ct0||inf; only if that bound cannot decide is ct0 transformed on its
    own.  MLDSA_OPT_SIGN_CT0_EXACT = 1 takes the exact test for every surviving attempt: same signatures, same oracle."""
    m = sets[44]
    n = 4096
    b = make_batch(m, n, 16, b"ct0")
    sig0 = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    sig1 = torch.empty_like(sig0)
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig0, n, key_idx=b["kidx"])
    assert hp.get_option(8) == 0
    hp.set_option(8, 1)
    try:
        m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig1, n, key_idx=b["kidx"])
    finally:
        hp.set_option(8, 0)
    assert torch.equal(sig0, sig1)
    got = host(sig0)
    for row, want in zip(range(0, n, 97), oracle_sigs(44, b, range(0, n, 97))):
        assert got[row].tobytes() == want


def test_async_plan_exponent_option(hp, sets):
    """MLDSA_OPT_SIGN_ASYNC_EXP: an asynchronous call plans until the expected number of unfinished ops is below 10^-value.
    With 1 (plan stops early) some ops of a large batch may come back MLDSA_ERR_AGAIN with an all-zero signature; every other
    signature is the one the synchronous call produces."""
    m = sets[44]
    n = 20000
    b = make_batch(m, n, 8, b"aexp")
    ref = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], ref, n, key_idx=b["kidx"])
    assert hp.get_option(9) == 9
    for exp in (1, 2, 12):
        hp.set_option(9, exp)
        try:
            assert hp.get_option(9) == exp
            sig = torch.full((n, m.SIG_LEN), 7, dtype=torch.uint8, device="cuda")
            st = torch.zeros(n, dtype=torch.int32, device="cuda")
            m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st, wait=False)
            st_h = host(st)
            assert set(np.unique(st_h)) <= {0, -5}
            done = torch.from_numpy(st_h == 0).cuda()
            assert torch.equal(sig[done], ref[done])
            assert not sig[~done].any()
            if exp == 12:
                assert (st_h == 0).all()
        finally:
            hp.set_option(9, 9)
    with pytest.raises(Exception):
        hp.set_option(9, 0)


# ------------------------------------------------------------------------------ every op of a full-size batch vs the oracle
@pytest.mark.parametrize("pset", [44, 65, 87])
def test_full_batch_every_signature_and_every_verdict_match_the_oracle(sets, pset):
    """BASELINE configs 2-3 at full size, checked in full: all 65 536 signatures of a batch byte-identical to the oracle's
    (16 host threads, a few seconds), and the verdicts on the same batch with one random bit flipped in 10 % of the
    signatures identical to the oracle's verdicts (not merely to the corruption pattern)."""
    m = sets[pset]
    n, nk = 65536, 512
    xi = [shake(b"full-key%d" % pset, i) for i in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    msgs = [shake(b"full-msg", i) for i in range(n)]
    rnd = [shake(b"full-rnd", i) for i in range(n)]
    kidx = (np.arange(n) * 7 % nk).astype(np.uint32)
    sig_h = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx, mode=0))
    skb, pkb = host(sk), host(pk)
    sk_o = [orc.sk_try_from_bytes(pset, skb[i].tobytes()) for i in range(nk)]
    want = orc.sign_batch_mt(pset, sk_o, kidx, msgs, rnd, 16, 1, mode=0)
    bad = [i for i in range(n) if sig_h[i].tobytes() != want[i]]
    assert not bad, (len(bad), bad[:5])
    rng = np.random.default_rng(pset)
    sig2 = sig_h.copy()
    idx = rng.choice(n, n // 10, replace=False)
    sig2[idx, rng.integers(0, m.SIG_LEN, idx.size)] ^= (1 << rng.integers(0, 8, idx.size)).astype(np.uint8)
    got = m.verify(pks, msgs, torch.from_numpy(sig2).cuda(), key_idx=kidx, mode=0)
    pk_o = [orc.pk_try_from_bytes(pset, pkb[i].tobytes()) for i in range(nk)]
    want_v = orc.verify_batch_mt(pset, pk_o, kidx, msgs, [sig2[i].tobytes() for i in range(n)], 16, 1, mode=0)
    assert np.array_equal(got, np.asarray(want_v, dtype=bool))
    assert int((~got).sum()) == idx.size  # a flipped bit never leaves a signature valid


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_get_public_key_of_arbitrary_secret_key_bytes(sets, pset):
    """private_to_public_key (ml_dsa.rs:502-559) computes t = A s1 + s2 from whatever expand_private decoded, out-of-range
    eta fields and a tr that is not H(pk) included: the public key bytes must be the ones the oracle derives."""
    m = sets[pset]
    rng = np.random.default_rng(300 + pset)
    sk = rng.integers(0, 256, (6, m.SK_LEN), dtype=np.uint8)
    sk[3, 128:] = 0xFF  # every field all-ones
    sks = m.private_keys_from_bytes(torch.from_numpy(sk).cuda())
    got = host(m.public_keys_into_bytes(m.get_public_key(sks)))
    for i in range(len(sk)):
        want = orc.pk_into_bytes(pset, orc.get_public_key(pset, orc.sk_try_from_bytes(pset, sk[i].tobytes())))
        assert got[i].tobytes() == want, i
