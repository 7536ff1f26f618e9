"""How a signing call is scheduled -- hipGraph replay, asynchronous calls, planned and extra rounds, speculation tables, passes, lanes -- never
changes a signature (sign_internal, src/ml_dsa.rs:153-337: the FIRST accepted candidate).  Includes the seeded soak over shapes, knobs and modes.
(Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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


# ------------------------------------------------------------------------------ two lanes in their small rounds
def test_two_signing_lanes_in_small_rounds_side_by_side(hp, sets):
    """Two lanes (MLDSA_OPT_SIGN_LANES = 2, calls of >= 8 192 ops) run their rounds side by side on two streams.  With one candidate
    per op the late rounds of BOTH lanes are small enough for the single-launch round front (k_sign_front_small), whose workgroups hand
    over through arrival counters: every lane has its own (found by the soak, seed 5151: with one array for both, a row's last workgroup
    could be named early and read half-written masks -- a different, later candidate was signed about once in a thousand calls).  The same
    batch 60 times: every signature every time, a sample against the oracle (ml_dsa.rs:212-330: the FIRST accepted candidate)."""
    m = sets[65]  # (l = 5: a row's masks come from TWO workgroups -- ML-DSA-44's four fit one, which hands over to nobody)
    n = 8200
    b = make_batch(m, n, 3, b"lanes2small")
    want = oracle_sigs(65, b, range(0, n, 401))
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    old = {o: hp.get_option(o) for o in (1, 3, 7)}
    hp.set_option(1, 0)
    hp.set_option(3, 1)  # MLDSA_OPT_SPEC_MAX: one candidate per op, ~35 rounds, most of them small
    hp.set_option(7, 2)  # MLDSA_OPT_SIGN_LANES
    try:
        first = None
        for it in range(int(os.environ.get("MLDSA_LANES_CALLS", "60"))):
            sig.zero_()
            m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
            got = host(sig).copy()
            assert int(host(st).min()) == 0 and int(host(st).max()) == 0
            if first is None:
                first = got
                for g, w in zip(first[::401], want):
                    assert g.tobytes() == w
            else:
                assert np.array_equal(got, first), f"call {it}: ops {np.nonzero((got != first).any(axis=1))[0][:8]} differ"
    finally:
        for o, v in old.items():
            hp.set_option(o, v)


def test_default_policy_takes_two_lanes_from_131072_ops_and_signs_the_same(hp, sets):
    """MLDSA_OPT_SIGN_LANES = 0 (the default): a call of >= 131 072 ops runs as two slices side by side (measured +3.5 ... 7 %,
    profiles/r06_ab_sign_lanes_grid.txt), a smaller one as one.  The signatures do not depend on it: 131 072 ML-DSA-44 ops under the default
    are byte for byte those of the one-lane call, a sample of them the oracle's, every one verifies on the device."""
    m = sets[44]
    n = 131072
    assert hp.get_option(7) == 0
    b = make_batch(m, n, 5, b"lanes-auto")
    idx = range(0, n, 4099)
    want = oracle_sigs(44, b, idx)
    sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    res = {}
    try:
        for lanes in (0, 1):
            hp.set_option(7, lanes)
            sig.zero_()
            m.sign_device(b["sks"], b["mb"], b["mo"], b["rn"], sig, n, key_idx=b["kidx"], status=st)
            assert int(host(st).min()) == 0 and int(host(st).max()) == 0
            res[lanes] = host(sig).copy()
    finally:
        hp.set_option(7, 0)
    assert np.array_equal(res[0], res[1]), f"ops {np.nonzero((res[0] != res[1]).any(axis=1))[0][:8]} differ between the default and one lane"
    for g, w in zip(res[0][::4099], want):
        assert g.tobytes() == w
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    m.verify_device(b["pks"], b["mb"], b["mo"], sig, ok, n, key_idx=b["kidx"])
    assert int(host(ok).min()) == 1


# ------------------------------------------------------------------------------ soak (tests/integration.rs:22-53 `forever`)
def test_soak_random_shapes_knobs_and_modes_against_the_oracle(hp, sets):
    """A seeded, time-bounded version of the reference's `forever` loop: random parameter set, batch size (1 ... 70 000,
    log-uniform), key count, message / ctx lengths, interface mode and library knobs (graph replay, speculation target and
    width, planned rounds -> the extra-round path, signing lanes, synchronous / asynchronous signing, the size limit of the
    single-launch small-call kernels), keygen -> sign ->
    verify -> flip -> verify on the device, with a sample of every iteration's keys, signatures and verdicts compared with
    the oracle.  Shapes repeat and alternate on ONE context, so workspace regrowth, the graph cache and the loop's control
    block are exercised the way a long-running service would."""
    seconds = float(os.environ.get("MLDSA_SOAK_SECONDS", "60"))
    seed = int(os.environ.get("MLDSA_SOAK_SEED", "20260203"))
    max_n = int(os.environ.get("MLDSA_SOAK_MAX_N", "70000"))  # (a small limit keeps the run on the small-call kernels: thousands of calls)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    defaults = {o: hp.get_option(o) for o in (1, 2, 3, 6, 7, 10, 13)}  # 10 = MLDSA_OPT_SIGN_LOOKAHEAD, 13 = MLDSA_OPT_SMALL_FUSED
    it = 0
    shapes = []
    try:
        while time.time() < t_end:
            it += 1
            pset = int(rng.choice([44, 65, 87]))
            m = sets[pset]
            if shapes and rng.random() < 0.3:
                n, nk = shapes[int(rng.integers(len(shapes)))]  # a shape seen before: graph-cache hits
            else:
                n = int(np.exp(rng.uniform(0, np.log(max_n))))
                nk = int(min(n, np.exp(rng.uniform(0, np.log(600)))))
                shapes.append((n, nk))
            mode = int(rng.choice([0, 0, 1, 2]))
            knobs = {1: int(rng.choice([0, 1, 2])), 2: int(rng.choice([1024, 8192, 40000, 65536, 150000])),
                     3: int(rng.choice([1, 4, 32, 64])), 6: int(rng.choice([0, 0, 1, 3])), 7: int(rng.choice([0, 1, 1, 2])),
                     10: int(rng.choice([0, 1, 2])), 13: int(rng.choice([256, 256, 0, 40, 1024]))}
            for o, v in knobs.items():
                hp.set_option(o, v)
            tag = b"soak%d-" % it
            what = f"seed {seed} iteration {it}: set {pset} n {n} keys {nk} mode {mode} knobs {knobs}"
            xi = [shake(tag + b"k", i) for i in range(nk)]
            pk, sk = m.keygen_from_seed(xi)
            pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
            max_len = int(rng.choice([0, 32, 300, 3000])) if n < 5000 else 48
            msgs = [shake(tag + b"m", i, int(rng.integers(0, max_len + 1))) if mode != 2 else
                    b"".join(orc.hash_message(shake(tag + b"m", i, 20), "SHA512")) for i in range(n)]
            ctxs = None if (mode == 1 or rng.random() < 0.5) else [shake(tag + b"c", i, int(rng.integers(0, 256))) for i in range(n)]
            rnd = [shake(tag + b"r", i) for i in range(n)]
            kidx = rng.integers(0, nk, n).astype(np.uint32)
            if rng.random() < 0.5:
                sig = m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, key_idx=kidx, mode=mode)
            else:  # asynchronous: re-sign what the enqueued rounds left (status -5), as a service would
                from fips204_amd.ml_dsa import _cat_with_offsets
                mb, mo = _cat_with_offsets(msgs, m.device)
                cb = co = None
                if ctxs is not None:
                    cb, co = _cat_with_offsets(ctxs, m.device)
                rn = dev(np.frombuffer(b"".join(rnd), dtype=np.uint8).reshape(n, 32))
                sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
                st = torch.zeros(n, dtype=torch.int32, device="cuda")
                kd = dev(kidx.view(np.int32))
                m.sign_device(sks, mb, mo, rn, sig, n, cb, co, kd, mode, st, wait=False)
                st_h = host(st)
                assert set(np.unique(st_h)) <= {0, -5}, what
                for i in np.nonzero(st_h == -5)[0]:
                    assert not host(sig[i]).any(), what
                    one = m.try_sign_with_seed(sks, [msgs[i]], [rnd[i]], ctxs=None if ctxs is None else [ctxs[i]],
                                               key_idx=kidx[i:i + 1], mode=mode)
                    sig[i] = one[0]
            torch.cuda.synchronize()
            assert hp.secret_residue()[1] == 0, what          # round 4: nothing secret outlives the signing call, whatever the knobs
            ok = m.verify(pks, msgs, sig, ctxs=ctxs, key_idx=kidx, mode=mode)
            assert ok.all(), what
            if n >= 4 and rng.random() < 0.3:                  # round 4: a few damaged entries in the message offset table
                from fips204_amd.ml_dsa import _cat_with_offsets
                mb, mo = _cat_with_offsets(msgs, m.device)
                cb = co = None
                if ctxs is not None:
                    cb, co = _cat_with_offsets(ctxs, m.device)
                off = host(mo).view(np.uint64).copy()
                for k in rng.choice(np.arange(1, n), min(3, n - 1), replace=False):
                    cands = [0, int(off[k - 1]) - 1 if off[k - 1] else 0, 2 ** 63, 2 ** 64 - 1, int(off[-1]) + 1]
                    off[k] = np.uint64(cands[int(rng.integers(len(cands)))])
                lo, hi = int(off[0]), int(off[-1])
                valid = np.array([lo <= int(a) <= int(b) <= hi for a, b in zip(off[:-1], off[1:])])
                same = valid & (off[:-1] == host(mo).view(np.uint64)[:-1]) & (off[1:] == host(mo).view(np.uint64)[1:])
                sg2 = torch.full((n, m.SIG_LEN), 0x33, dtype=torch.uint8, device="cuda")
                st2 = torch.full((n,), 9, dtype=torch.int32, device="cuda")
                m.sign_device(sks, mb, dev(off.view(np.int64)), dev(np.frombuffer(b"".join(rnd), dtype=np.uint8).reshape(n, 32)), sg2, n, cb, co,
                              dev(kidx.view(np.int32)), mode, st2)
                st2_h, sg2_h = host(st2), host(sg2)
                clen_bad = np.array([ctxs is not None and len(ctxs[i]) > 255 for i in range(n)])
                assert ((st2_h == -1) == ~valid).all() and (st2_h[valid & ~clen_bad] == 0).all(), what
                assert not sg2_h[~valid].any() and np.array_equal(sg2_h[same & ~clen_bad], host(sig)[same & ~clen_bad]), what
            sig_h = host(sig).copy()
            flip = rng.random(n) < 0.3
            rows = np.nonzero(flip)[0]
            sig_h[rows, rng.integers(0, m.SIG_LEN, rows.size)] ^= (1 << rng.integers(0, 8, rows.size)).astype(np.uint8)
            ok2 = m.verify(pks, msgs, dev(sig_h), ctxs=ctxs, key_idx=kidx, mode=mode)
            assert np.array_equal(ok2, ~flip), what
            if rng.random() < 0.5:                             # round 4: the same verdicts from wire-format keys in one call
                assert np.array_equal(m.verify_pk(pk, msgs, dev(sig_h), ctxs=ctxs, key_idx=kidx, mode=mode), ok2), what
            # the oracle on a sample: keys, signatures (before the flips), verdicts (after)
            pkb, skb = host(pk), host(sk)
            sig_good = host(sig)
            for i in rng.choice(n, min(n, 12), replace=False):
                ki = int(kidx[i])
                pk_o, sk_o = orc.keygen_from_seed(pset, xi[ki])
                assert pkb[ki].tobytes() == orc.pk_into_bytes(pset, pk_o) and skb[ki].tobytes() == orc.sk_into_bytes(pset, sk_o), what
                c = b"" if ctxs is None else ctxs[i]
                assert sig_good[i].tobytes() == orc.sign_internal(pset, sk_o, msgs[i], rnd[i], ctx=c, mode=mode), (what, int(i))
                assert bool(ok2[i]) == orc.verify_internal(pset, pk_o, msgs[i], sig_h[i].tobytes(), ctx=c, mode=mode), (what, int(i))
    finally:
        for o, v in defaults.items():
            hp.set_option(o, v)
    assert it >= 3, f"only {it} iterations in {seconds} s"
    print(f"soak: {it} iterations in {seconds:.0f} s, stats {hp.stats()}")


# ------------------------------------------------------------------------------ slot capacity at the speculation rule's thresholds
@pytest.mark.parametrize("n,env", [(22300, {}), (22200, {}), (65536, {"MLDSA_SPEC_ROWS": "81920"}), (50800, {"MLDSA_SPEC_ROWS": "81920", "MLDSA_SPEC_TARGET": "131072"})])
def test_rounds_just_below_a_speculation_threshold_fit_the_workspace(n, env):
    """Candidates per op = round((rows / m) ^ 0.85): a count just below a threshold of that rule has more slots than `rows`
    (22 300 ops x 3 = 66 900 for rows = 65 536).  The workspace is carved for the rule's true maximum and k_make_slots never makes
    more: every signature of such a batch verifies and a sample equals the oracle's (ml_dsa.rs:212-330: the FIRST accepted
    candidate, whatever the speculation)."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    env = dict(env, MLDSA_TUNING_ENV="1") if env else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h2 = HotPath(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        m = MlDsa(65, hotpath=h2)
        nk = 16
        pk, sk = m.keygen_from_seed([shake(b"cap-key", i) for i in range(nk)])
        sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
        msgs = [shake(b"cap-msg", i, 32) for i in range(n)]
        rnd = [shake(b"cap-rnd", i) for i in range(n)]
        kidx = (np.arange(n) % nk).astype(np.uint32)
        sig = m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx)          # ONE device-resident call of n ops
        assert np.asarray(m.verify(pks, msgs, sig, key_idx=kidx)).all()
        sigb, skb = host(sig), host(sk)
        sk_o = [orc.sk_try_from_bytes(65, skb[i].tobytes()) for i in range(nk)]
        for i in list(range(8)) + [n // 2, n - 1]:
            assert sigb[i].tobytes() == orc.sign_internal(65, sk_o[kidx[i]], msgs[i], rnd[i], ctx=b"", mode=0)
    finally:
        h2.close()


# ------------------------------------------------------------------------------ two candidates per op generated at once
@pytest.mark.parametrize("pset", [44, 65, 87])
def test_two_candidate_generation_and_the_speculation_table_do_not_change_a_signature(hp, sets, pset):
    """MLDSA_OPT_SIGN_LOOKAHEAD (rounds that generate kappa and kappa + l of every op at once and test them in two rounds,
    rows addressed through slot_y) and the candidates-per-op rule are scheduling only: the same 20 000 signatures, byte for
    byte, whatever they are set to -- and those are the oracle's (ml_dsa.rs:212-330: the FIRST accepted kappa wins)."""
    m = sets[pset]
    n, nk = 20000, 50
    xi = [shake(b"look-key%d" % pset, i) for i in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    sks = m.private_keys_from_bytes(sk)
    msgs = [shake(b"look-msg", i, 33) for i in range(n)]
    rnd = [shake(b"look-rnd", i) for i in range(n)]
    kidx = (np.arange(n) * 11 % nk).astype(np.uint32)
    old = {o: hp.get_option(o) for o in (2, 3, 10)}
    got = {}
    try:
        for look in (0, 1, 2):
            for target, smax in ((65536, 32), (30000, 5)):
                hp.set_option(10, look); hp.set_option(2, target); hp.set_option(3, smax)
                got[(look, target)] = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx)).copy()
    finally:
        for o, v in old.items():
            hp.set_option(o, v)
    ref = got[(0, 65536)]
    for k, v in got.items():
        assert np.array_equal(v, ref), k
    skb = host(sk)
    sk_o = [orc.sk_try_from_bytes(pset, skb[i].tobytes()) for i in range(nk)]
    want = orc.sign_batch_mt(pset, sk_o, kidx, msgs, rnd, 16, 1, mode=0)
    assert all(ref[i].tobytes() == want[i] for i in range(n))


def test_graph_cache_outlives_the_stream_it_was_filled_on(sets):
    """Small signing calls replay as hipGraphs, kept in a cache of 24 shapes (MLDSA_OPT_GRAPH_CACHE).  A caller may DESTROY the
    stream it made those calls on; shapes replaced later must not touch that stream again (the library used to wait for the
    entry's last stream before destroying its graph: a dangling handle -- found as hangs and a crash in 3 of 10 runs of
    tests/test_gpu_batcher.py, whose every batcher owns and destroys a stream; 0 of 31 runs since each entry carries its own
    event).  What a dangling handle does is up to the allocator, so this test exercises the path rather than proving the fix:
    30 shapes on a stream, stream destroyed, 30 more on a second one; signatures against the oracle."""
    m = sets[44]
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(12)
    xi = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    sks = m.private_keys_from_bytes(m.keygen_from_seed(dev(xi))[1])
    _, sk_o = orc.keygen_from_seed(44, xi[0].tobytes())
    captured0 = m.hp.stats()["graphs_captured"]
    raws = [C.c_void_p(), C.c_void_p()]  # both made first: the second must not get the handle the first one had
    for raw in raws:
        assert hip.hipStreamCreate(C.byref(raw)) == 0
    for leg, raw in enumerate(raws):
        with torch.cuda.stream(torch.cuda.ExternalStream(raw.value)):
            for n in range(1 + 30 * leg, 31 + 30 * leg):
                msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
                rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
                d_msg, d_off, d_rnd = dev(msgs.reshape(-1)), dev_off(np.arange(n + 1) * 32), dev(rnd)
                kidx = torch.zeros(n, dtype=torch.int32, device="cuda")
                sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
                for _ in range(3):  # first sighting: direct; second: captured; third: replayed
                    m.sign_device(sks, d_msg, d_off, d_rnd, sig, n, key_idx=kidx)
                torch.cuda.current_stream().synchronize()
                got = sig.cpu().numpy()
                j = int(rng.integers(0, n))
                assert got[j].tobytes() == orc.sign_internal(44, sk_o, msgs[j].tobytes(), rnd[j].tobytes(), mode=orc.MODE_PURE)
        assert hip.hipStreamDestroy(raw) == 0
    st = m.hp.stats()
    assert st["graphs_captured"] - captured0 >= 50   # more shapes than the cache holds: entries of the dead stream were replaced


def test_captures_survive_device_wide_waits_on_other_threads(sets):
    """hipDeviceSynchronize / hipFree wait for every stream of the device, and waiting for a stream that is being captured invalidates
    the capture.  While one thread signs small batches of ever new shapes (second sighting = capture, third = replay) two others
    keep waiting for the device: a second CONTEXT growing its workspace over and over (ensure_workspace: kept apart from captures
    by the library's lock), and raw hipDeviceSynchronize + hipMalloc / hipFree (invisible to the library: a capture that does not
    end well falls back to direct launches).  Every call succeeds and every signature equals the oracle's.  (Before: MLDSA_ERR_DEVICE
    "operation failed due to a previous error during capture" for the call and every later one of its shape -- found with four
    batcher lanes on one GPU.)"""
    import threading
    from fips204_amd.hotpath import HotPath
    m = sets[44]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(31)
    xi = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    sks = m.private_keys_from_bytes(m.keygen_from_seed(dev(xi))[1])
    _, sk_o = orc.keygen_from_seed(44, xi[0].tobytes())
    stop = threading.Event()
    waits = [0, 0]
    errors = []

    def grow_another_context():
        try:
            torch.cuda.set_device(0)
            while not stop.is_set():
                other = HotPath(0)
                for n in (64, 256, 1024, 4096):   # every step replaces the workspace: device-wide wait + hipFree + hipMalloc
                    other.reserve(65, 2, n)
                    waits[0] += 1
                other.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def wait_raw():
        torch.cuda.set_device(0)
        while not stop.is_set():
            p = C.c_void_p()
            hip.hipDeviceSynchronize()   # (the runtime may refuse it beside a capture: hipErrorStreamCaptureUnsupported -- the caller's problem)
            if hip.hipMalloc(C.byref(p), 1 << 20) == 0:
                hip.hipFree(p)
            waits[1] += 1

    threads = [threading.Thread(target=grow_another_context), threading.Thread(target=wait_raw)]
    for t in threads:
        t.start()
    try:
        for n in range(1, 41):
            msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            d_msg, d_off, d_rnd = dev(msgs.reshape(-1)), dev_off(np.arange(n + 1) * 32), dev(rnd)
            kidx = torch.zeros(n, dtype=torch.int32, device="cuda")
            sig = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
            for _ in range(3):
                sig.zero_()
                m.sign_device(sks, d_msg, d_off, d_rnd, sig, n, key_idx=kidx)   # raises on any error return
                got = host(sig)
                j = int(rng.integers(0, n))
                assert got[j].tobytes() == orc.sign_internal(44, sk_o, msgs[j].tobytes(), rnd[j].tobytes(), mode=orc.MODE_PURE), (n, j)
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors
    assert waits[0] > 8 and waits[1] > 20
