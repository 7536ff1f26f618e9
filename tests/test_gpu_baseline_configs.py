"""BASELINE.json configs at full size against the oracle: config 3 (ML-DSA-65 sign, 65 536 ops), config 4's per-GPU slice (ML-DSA-87 verify, 131 072 ops),
every signature and verdict of a full batch per set, config 5's request mix.  (Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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
    # graph policy 1: a call of this size is launched directly (only calls of <= 16384 ops replay);
    # MLDSA_OPT_GRAPHS = 2 replays it too, with the same signatures
    hp = m.hp
    old_graphs = hp.get_option(1)
    hp.set_option(1, 1)
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
        hp.set_option(1, old_graphs)


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


# ------------------------------------------------------------------------------ BASELINE config 5 at its specified request mix
def test_config5_request_mix_against_the_oracle(hp):
    """SURVEY 8(d') C5: request i -> set (44, 65, 87)[i mod 3], keygen / sign / verify by i mod 10 (10 / 40 / 50 %), bucketed into
    per-set calls on one context with asynchronous signing (bench.py MixedStream = `--workload mixed`).  EVERY generated key,
    signature and verdict of a step is compared with the oracle (ml_dsa.rs:57, 153, 351 are the three callers)."""
    import bench
    old = hp.get_option(9)
    try:
        wl = bench.MixedStream(hp, 1200, 0)
        n = wl.ops_per_step
        assert n == 3600 and wl.count == {"keygen": 360, "sign": 1440, "verify": 1800}
        for pset in (44, 65, 87):
            r = wl.req[pset]
            ids = np.concatenate([r["keygen"], r["sign"], r["verify"]])
            assert len(ids) == 1200 and (ids % 3 == (44, 65, 87).index(pset)).all()
            assert (r["keygen"] % 10 == 0).all() and ((r["sign"] % 10 >= 1) & (r["sign"] % 10 <= 4)).all() and (r["verify"] % 10 >= 5).all()
        wl.check(n_oracle=10 ** 9)
        # a second step on the same buffers (graph replay for the shapes the policy covers) gives the same bytes
        before = [d["sign"]["sig"].clone() for d in wl.sets]
        for d in wl.sets:
            d["sign"]["sig"].zero_()
        wl.step(1)
        wl.finish_steps()
        for d, b in zip(wl.sets, before):
            assert torch.equal(d["sign"]["sig"], b)
    finally:
        hp.set_option(9, old)
