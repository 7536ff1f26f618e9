"""The C-ABI boundary under untrusted input: offset tables, over-long ctx, NULL arrays, argument errors, caller-owned and capped workspaces,
non-canonical hint encodings, fuzz-derived batches (the reference never panics: fuzz/fuzz_targets/fuzz_all.rs:25-37; src/lib.rs:274, 368-370).
(Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_fuzzed_signatures_and_arbitrary_public_keys_full_batch(sets, pset):
    """65 536 verifications per parameter set: good signatures XOR masks of every density, masks confined to each
    section of the encoding, random signatures, and random PUBLIC-KEY bytes through mldsa_pk_expand -- all verdicts
    equal the oracle's (fuzz_all.rs:14-37, fuzz_verify.rs:17-31)."""
    m = sets[pset]
    n, nk = 65536, 256
    pk_all, kidx, msgs, sig, cls, changed = fuzz_batch(m, pset, n, nk, 7000 + pset)
    pks = m.public_keys_from_bytes(dev(pk_all))
    got = m.verify(pks, msgs, dev(sig), key_idx=kidx, mode=0)
    pk_o = [orc.pk_try_from_bytes(pset, pk_all[i].tobytes()) for i in range(2 * nk)]
    want = np.asarray(orc.verify_batch_mt(pset, pk_o, kidx, msgs, [sig[i].tobytes() for i in range(n)], 16, 1, mode=0), dtype=bool)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(int(i), CLASSES[cls[i]]) for i in bad[:8]]
    assert got[cls == 0].all()  # the untouched signatures verify
    # the same batch through the host-memory entry point (wire-format keys expanded inside the call)
    got_h = m.verify_host(pk_all, msgs, sig, key_idx=kidx, mode=0)
    assert np.array_equal(got_h, want)
    # strong unforgeability as a sanity check of the batch itself: whatever was altered is rejected, the rest still verifies
    assert np.array_equal(got, ~changed)
    assert changed.sum() > n * 0.85 and (~changed).sum() >= n // len(CLASSES)


def test_small_passes_give_identical_results(sets):
    """A context whose device cannot hold the workspace of a full pass falls back to smaller passes (reserve_workspace);
    MLDSA_PASS_OPS / MLDSA_PASS_OPS_SIGN force that: 3 000 ops in passes of 512 must give the bytes of the one-pass call."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    m = sets[65]
    n, nk = 3000, 40
    xi = [shake(b"pass-key", i) for i in range(nk)]
    msgs = [shake(b"pass-msg", i, 40) for i in range(n)]
    rnd = [shake(b"pass-rnd", i) for i in range(n)]
    kidx = (np.arange(n) * 3 % nk).astype(np.uint32)
    def run(mm):
        pk, sk = mm.keygen_from_seed(xi)
        sig = mm.try_sign_with_seed(mm.private_keys_from_bytes(sk), msgs, rnd, key_idx=kidx)
        bad = sig.clone()
        bad[::3, 100] ^= 1
        return host(pk), host(sk), host(sig), mm.verify(mm.public_keys_from_bytes(pk), msgs, bad, key_idx=kidx)
    want = run(m)
    old = {k: os.environ.get(k) for k in ("MLDSA_PASS_OPS", "MLDSA_PASS_OPS_SIGN", "MLDSA_TUNING_ENV")}
    os.environ["MLDSA_TUNING_ENV"] = "1"  # the knobs are read only when the process asks for them (include/mldsa_hip.h "Environment")
    os.environ["MLDSA_PASS_OPS"] = "512"
    os.environ["MLDSA_PASS_OPS_SIGN"] = "512"
    try:
        h2 = HotPath(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        got = run(MlDsa(65, hotpath=h2))
    finally:
        h2.close()
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    assert (~want[3][::3]).all() and want[3][1::3].all()


@pytest.mark.parametrize("pset,mode", [(44, 0), (65, 1), (87, 2)])
def test_malformed_offset_tables_refuse_only_their_ops(sets, pset, mode):
    """a decreasing / overshooting / wrapping msg_off or ctx_off pair used to be a ~2^64-byte length inside
    k_mu (endless loop, out-of-bounds reads).  Now: the op is refused (verify: ok = 0; sign and sign_async: status
    MLDSA_ERR_PARAM, all-zero signature), the call returns in the usual time, and every op whose pair is in order gets exactly the
    oracle's result for the bytes its pair names.  4 096 ops, damage in the middle, all three modes."""
    m = sets[pset]
    n, nk = 4096, 16
    rng = np.random.default_rng(4000 + pset)
    xi = [shake(b"off-key%d" % pset, i) for i in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    sk_o = [orc.sk_try_from_bytes(pset, bytes(r)) for r in host(sk)]
    if mode == 2:   # pre-hash mode: the message is OID || PH(M), a fixed 43 / 75 bytes
        from fips204_amd.ml_dsa import hash_message
        msgs = [hash_message(shake(b"off-msg", i, 50), "SHA256" if i % 2 else "SHA512") for i in range(n)]
    else:
        msgs = [shake(b"off-msg", i, int(rng.integers(0, 300))) for i in range(n)]
    ctxs = [shake(b"off-ctx", i, i % 7) for i in range(n)]
    rnd = np.frombuffer(b"".join(shake(b"off-rnd", i) for i in range(n)), dtype=np.uint8).reshape(n, 32)
    kidx_h = (np.arange(n) * 3 % nk).astype(np.uint32)
    kidx = dev(kidx_h.view(np.int32))
    mflat, moff = table(msgs)
    cflat, coff = table(ctxs)
    d_m, d_c, d_rnd = dev(mflat), dev(cflat), dev(rnd)

    def sign(mo, co, wait):
        sig = torch.full((n, m.SIG_LEN), 0xAB, dtype=torch.uint8, device="cuda")
        st = torch.full((n,), 77, dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        m.sign_device(sks, d_m, dev_off(mo), d_rnd, sig, n, d_c, dev_off(co), kidx, mode, st, wait=wait)
        torch.cuda.synchronize()
        return host(sig), host(st), time.perf_counter() - t0

    def verify(mo, co, sig):
        ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        t0 = time.perf_counter()
        m.verify_device(pks, d_m, dev_off(mo), dev(sig), ok, n, d_c, dev_off(co), kidx, mode)
        torch.cuda.synchronize()
        return host(ok).astype(bool), time.perf_counter() - t0

    sign(moff, coff, True)  # warm-up (workspace growth, stream probing)
    good, st0, t_sign = sign(moff, coff, True)
    assert (st0 == 0).all()
    v0, t_ver = verify(moff, coff, good)
    assert v0.all()
    for i in rng.choice(n, 24, replace=False):   # the clean batch against the oracle
        assert good[i].tobytes() == orc.sign_internal(pset, sk_o[kidx_h[i]], msgs[i], rnd[i].tobytes(), ctx=ctxs[i], mode=mode), int(i)

    def named(flat, off, i):  # the bytes a (valid) pair names
        return flat[int(off[i]):int(off[i + 1])].tobytes()

    for which in ("msg", "ctx"):
        base = moff if which == "msg" else coff
        for name, bad in corruptions(base, rng).items():
            mo, co = (bad, coff) if which == "msg" else (moff, bad)
            valid = pairs_ok(mo) & pairs_ok(co)
            clen_ok = np.array([(int(co[i + 1]) - int(co[i])) <= 255 if valid[i] else True for i in range(n)])
            same = valid & (mo[:-1] == moff[:-1]) & (mo[1:] == moff[1:]) & (co[:-1] == coff[:-1]) & (co[1:] == coff[1:])
            for wait in (True, False):
                sig, st, dt = sign(mo, co, wait)
                assert dt < 20 * t_sign + 0.5, (which, name, wait, dt, t_sign)     # no endless loop
                assert ((st == -1) == ~valid).all(), (which, name, wait, np.nonzero((st == -1) != ~valid)[0][:5])
                assert ((st == -2) == (valid & ~clen_ok)).all(), (which, name)
                assert (st[valid & clen_ok] == 0).all()
                assert not sig[st != 0].any(), (which, name, "refused ops get all-zero signatures")
                assert np.array_equal(sig[same], good[same]), (which, name, wait)  # untouched ops: byte-identical
                shifted = np.nonzero(valid & clen_ok & ~same)[0]                   # in order, but naming other bytes: the oracle on THOSE bytes
                for i in shifted[:12]:
                    want = orc.sign_internal(pset, sk_o[kidx_h[i]], named(mflat, mo, i), rnd[i].tobytes(), ctx=named(cflat, co, i), mode=mode)
                    assert sig[i].tobytes() == want, (which, name, int(i))
            ok, dt = verify(mo, co, good)
            assert dt < 20 * t_ver + 0.5, (which, name, dt, t_ver)
            # valid exactly where the pair still names the bytes that were signed (an op shifted onto other bytes fails its c~ check)
            # (the internal interface hashes no ctx, ml_dsa.rs:386-388: there only its length matters)
            same_bytes = np.array([bool(valid[i] and clen_ok[i]) and named(mflat, mo, i) == msgs[i] and (mode == 1 or named(cflat, co, i) == ctxs[i])
                                   for i in range(n)])
            assert np.array_equal(ok, same_bytes), (which, name, np.nonzero(ok != same_bytes)[0][:5])
            assert ok.sum() >= n - 100 or name == "short_last_entry", (which, name, int(ok.sum()))


def test_over_long_ctx_is_refused_before_it_is_read(sets):
    """lib.rs:274 / 368 return before touching the message: an op whose ctx is 100 MB costs what an op with an empty ctx costs
    (the lane used to hash all of it: ~0.7 M serial permutations), and the other ops of the batch are unaffected."""
    m = sets[44]
    n = 256
    xi = [shake(b"ctx-key", 0)]
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    msgs = [shake(b"ctx-msg", i) for i in range(n)]
    mflat, moff = table(msgs)
    big = 100 * 1000 * 1000
    ctx_bytes = torch.randint(0, 256, (big + 4096,), dtype=torch.uint8, device="cuda")
    lens = np.array([i % 4 for i in range(n)], dtype=np.uint64)
    lens[100] = big
    lens[200] = 256
    coff = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=coff[1:])
    small = np.zeros(n + 1, dtype=np.uint64)  # the same batch with the two long ctxs emptied
    lens2 = lens.copy(); lens2[[100, 200]] = 0
    np.cumsum(lens2, out=small[1:])
    kidx = dev(np.zeros(n, dtype=np.int32))
    rnd = dev(np.zeros((n, 32), dtype=np.uint8))
    d_m = dev(mflat)

    def run(co):
        sig = torch.zeros((n, m.SIG_LEN), dtype=torch.uint8, device="cuda")
        st = torch.zeros(n, dtype=torch.int32, device="cuda")
        ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
        t0 = time.perf_counter()
        m.sign_device(sks, d_m, dev_off(moff), rnd, sig, n, ctx_bytes, dev_off(co), kidx, 0, st)
        m.verify_device(pks, d_m, dev_off(moff), sig, ok, n, ctx_bytes, dev_off(co), kidx, 0)
        torch.cuda.synchronize()
        return host(sig), host(st), host(ok), time.perf_counter() - t0

    run(small)
    _, st_s, ok_s, t_small = run(small)
    sig, st, ok, t_big = run(coff)
    assert (st_s == 0).all() and ok_s.all()
    assert st[100] == -2 and st[200] == -2 and (np.delete(st, [100, 200]) == 0).all()
    assert not ok[100] and not ok[200] and np.delete(ok, [100, 200]).all()
    assert not sig[100].any() and not sig[200].any()
    assert t_big < 3 * t_small + 0.05, (t_big, t_small)
    cb = host(ctx_bytes[:int(coff[99]) + 8])
    sk_o = orc.sk_try_from_bytes(44, bytes(host(sk)[0]))
    for i in (0, 1, 2, 3, 99):
        assert sig[i].tobytes() == orc.sign_internal(44, sk_o, msgs[i], bytes(32), ctx=cb[int(coff[i]):int(coff[i + 1])].tobytes(), mode=0)


# ------------------------------------------------------------------------------ workspace that does not fit (pipeline.hip:40)
@pytest.mark.parametrize("pset", [44, 87])
def test_workspace_cap_shrinks_the_passes_and_changes_nothing(sets, pset):
    """A device that cannot hold the workspace of a full pass: reserve_workspace used to give up whenever the call was smaller
    than half a pass (every signing call up to 131 072 ops).  With MLDSA_OPT_WORKSPACE_CAP_MB standing in for the allocation
    failure the context halves its pass size until the workspace fits, runs the call in several passes, and keygen / sign /
    verify output is byte-identical to an uncapped context's."""
    from fips204_amd import _lib
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    m = sets[pset]
    n, nk = 6000, 40
    xi = [shake(b"cap-key%d" % pset, i) for i in range(n)]
    msgs = [shake(b"cap-msg", i, i % 50) for i in range(n)]
    rnd = [shake(b"cap-rnd", i) for i in range(n)]
    kidx = (np.arange(n) * 7 % nk).astype(np.uint32)
    pk0, sk0 = m.keygen_from_seed(xi)
    sks0 = m.private_keys_from_bytes(sk0[:nk])
    pks0 = m.public_keys_from_bytes(pk0[:nk])
    sig0 = m.try_sign_with_seed(sks0, msgs, rnd, key_idx=kidx)
    bad = host(sig0).copy()
    bad[::9, 40] ^= 1
    v0 = m.verify(pks0, msgs, dev(bad), key_idx=kidx)
    hp2 = HotPath(0)
    try:
        cap = {44: 100, 87: 200}[pset]   # MiB: below what a 6 000-op pass of keygen / sign / verify needs
        hp2.set_option(_lib.OPT_WORKSPACE_CAP_MB, cap)
        assert hp2.get_option(_lib.OPT_WORKSPACE_CAP_MB) == cap
        m2 = MlDsa(pset, hotpath=hp2)
        pk1, sk1 = m2.keygen_from_seed(xi)
        s_keygen = hp2.stats()["workspace_shrinks"]
        assert torch.equal(pk1, pk0) and torch.equal(sk1, sk0)
        sks1, pks1 = m2.private_keys_from_bytes(sk1[:nk]), m2.public_keys_from_bytes(pk1[:nk])
        sig1 = m2.try_sign_with_seed(sks1, msgs, rnd, key_idx=kidx)
        s_sign = hp2.stats()["workspace_shrinks"]
        assert torch.equal(sig1, sig0)
        v1 = m2.verify(pks1, msgs, dev(bad), key_idx=kidx)
        assert np.array_equal(v1, v0) and not v1[::9].any() and v1[1::9].all()
        st = hp2.stats()
        assert s_keygen > 0 and s_sign > s_keygen, st          # both pass sizes had to come down below the call's size
        # a cap too small even for the smallest pass is an error, not a crash -- and the context recovers when it is lifted
        hp3 = HotPath(0)
        try:
            hp3.set_option(_lib.OPT_WORKSPACE_CAP_MB, 1)
            with pytest.raises(_lib.MldsaError) as e:
                MlDsa(pset, hotpath=hp3).keygen_from_seed(xi)
            assert e.value.code == _lib.ERR_NOMEM
            hp3.set_option(_lib.OPT_WORKSPACE_CAP_MB, 0)
            pk3, _ = MlDsa(pset, hotpath=hp3).keygen_from_seed(xi[:100])
            assert torch.equal(pk3, pk0[:100])
        finally:
            hp3.close()
    finally:
        hp2.close()


def test_offsets_that_name_bytes_of_a_null_array_refuse_the_op(sets):
    """Device-resident calls cannot check the caller's tables on the host.  msgs = NULL is legal when every message is empty; an op
    whose offsets name bytes of the NULL array is refused like any other malformed pair (no read through the NULL pointer)."""
    from fips204_amd import _lib
    m = sets[44]
    lib, h = m.lib, m.hp._h
    n = 64
    pk, sk = m.keygen_from_seed([shake(b"null-key", 0)])
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    lens = np.zeros(n, dtype=np.uint64)
    lens[[5, 40]] = 9                       # two ops claim nine bytes of a message array that is not there
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    d_off = dev_off(off)
    kidx = dev(np.zeros(n, dtype=np.int32))
    rnd = dev(np.zeros((n, 32), dtype=np.uint8))
    sig = torch.full((n, m.SIG_LEN), 7, dtype=torch.uint8, device="cuda")
    st = torch.full((n,), 9, dtype=torch.int32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.mldsa_sign(h, 44, 0, P(sks.rho), P(sks.cap_k), P(sks.tr), P(sks.s_1_hat_mont), P(sks.s_2_hat_mont), P(sks.t_0_hat_mont), 1,
                              P(kidx), None, P(d_off), None, None, P(rnd), P(sig), P(st), n, s))
    st_h, sig_h = host(st), host(sig)
    bad = np.zeros(n, dtype=bool)
    bad[[5, 40]] = True
    assert (st_h[bad] == _lib.ERR_PARAM).all() and (st_h[~bad] == 0).all() and not sig_h[bad].any()
    ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    _lib.check(lib.mldsa_verify(h, 44, 0, P(pks.rho), P(pks.tr), P(pks.t1_d2_hat_mont), 1, P(kidx), None, P(d_off), None, None, P(sig), P(ok), n, s))
    assert np.array_equal(host(ok).astype(bool), ~bad)
    sk_o = orc.sk_try_from_bytes(44, bytes(host(sk)[0]))
    assert sig_h[0].tobytes() == orc.sign_internal(44, sk_o, b"", bytes(32), mode=0)


def test_caller_owned_workspace_and_sized_stats(sets):
    """mldsa_ctx_set_workspace: the caller's buffer is the workspace (the reference allocates nothing, README.md:15-16).  A buffer too
    small for a full pass makes the context run smaller passes -- same keys, signatures and verdicts --, one too small for any pass
    is MLDSA_ERR_NOMEM, and the context never grows or frees it.  mldsa_get_stats_sized writes only what the caller's struct holds."""
    from fips204_amd import _lib
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    m = sets[65]
    n, nk = 5000, 8
    xi = [shake(b"ws-key", i) for i in range(n)]
    msgs = [shake(b"ws-msg", i, i % 40) for i in range(n)]
    rnd = [shake(b"ws-rnd", i) for i in range(n)]
    kidx = (np.arange(n) % nk).astype(np.uint32)
    pk0, sk0 = m.keygen_from_seed(xi)
    sig0 = m.try_sign_with_seed(m.private_keys_from_bytes(sk0[:nk]), msgs, rnd, key_idx=kidx)
    hp2 = HotPath(0)
    try:
        ws = torch.full((192 << 20,), 0x5A, dtype=torch.uint8, device="cuda")   # 192 MiB: no 5 000-op ML-DSA-65 pass fits
        guard = ws[-4096:].clone()
        hp2.set_workspace(ws[:-4096])
        m2 = MlDsa(65, hotpath=hp2)
        pk1, sk1 = m2.keygen_from_seed(xi)
        sig1 = m2.try_sign_with_seed(m2.private_keys_from_bytes(sk1[:nk]), msgs, rnd, key_idx=kidx)
        ok1 = m2.verify(m2.public_keys_from_bytes(pk1[:nk]), msgs, sig1, key_idx=kidx)
        assert torch.equal(pk1, pk0) and torch.equal(sk1, sk0) and torch.equal(sig1, sig0) and ok1.all()
        st = hp2.stats()
        assert st["workspace_shrinks"] > 0 and st["workspace_growths"] == 0, st     # smaller passes, and never a hipMalloc of its own
        torch.cuda.synchronize()
        assert torch.equal(ws[-4096:], guard), "the context wrote past the end of the caller's buffer"
        # a client built against the five-field mldsa_stats of round 2: 40 bytes are written, the next 8 are left alone
        buf = (C.c_ulonglong * 7)(*([0xDEADBEEF] * 7))
        _lib.check(m.lib.mldsa_get_stats_sized(hp2._h, buf, 40))
        assert buf[5] == 0xDEADBEEF and buf[6] == 0xDEADBEEF and buf[2] == st["direct_calls"]
        # too small for the smallest pass
        small = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        hp2.set_workspace(small)
        with pytest.raises(_lib.MldsaError) as e:
            m2.keygen_from_seed(xi[:2000])
        assert e.value.code == _lib.ERR_NOMEM
        hp2.set_workspace(None)                                                      # back to a context-owned workspace
        pk3, _ = m2.keygen_from_seed(xi[:300])
        assert torch.equal(pk3, pk0[:300])
        # misaligned or half-specified buffers are argument errors
        assert m.lib.mldsa_ctx_set_workspace(hp2._h, C.c_void_p(ws.data_ptr() + 8), 1 << 20) == _lib.ERR_PARAM
        assert m.lib.mldsa_ctx_set_workspace(hp2._h, C.c_void_p(ws.data_ptr()), 0) == _lib.ERR_PARAM
    finally:
        hp2.close()


def test_argument_errors_never_abort(sets):
    """include/mldsa_hip.h: "0 = MLDSA_OK, negative = error (never aborts)".  Every entry point with NULL pointers, unknown parameter
    sets, unknown modes, n_keys that do not cover the batch and zero-sized batches: an error code (or MLDSA_OK for an empty batch)
    and a message, never a fault -- and the context still signs and verifies afterwards (src/lib.rs:274, 368: the reference returns
    Err / false on every malformed argument it can be handed)."""
    from fips204_amd import _lib
    m = sets[44]
    lib, h = m.lib, m.hp._h
    E, OK = _lib.ERR_PARAM, 0
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    off = dev_off(np.zeros(9, dtype=np.uint64))
    p, z, o = C.c_void_p(buf.data_ptr()), None, C.c_void_p(off.data_ptr())
    calls = [
        # seam level: NULL pointers with n > 0, n = 0 with NULLs, unknown sets
        (lib.mldsa_ntt, (h, z, p, 4, z), E), (lib.mldsa_ntt, (h, z, z, 0, z), OK), (lib.mldsa_ntt, (None, p, p, 4, z), E),
        (lib.mldsa_inv_ntt, (h, p, z, 1, z), E), (lib.mldsa_to_mont, (h, z, z, 1, z), E),
        (lib.mldsa_mat_vec_mul, (h, 45, p, p, p, 1, z), E), (lib.mldsa_mat_vec_mul, (h, 44, z, p, p, 1, z), E),
        (lib.mldsa_pointwise_mont, (h, z, p, p, 4, 1, z), E), (lib.mldsa_add_vector_ntt, (h, p, z, p, 1, z), E),
        (lib.mldsa_infinity_norm, (h, p, 0, 1, p, z), E), (lib.mldsa_infinity_norm, (h, z, 4, 1, p, z), E),
        (lib.mldsa_verify_arith, (h, 44, p, p, z, p, p, 1, z), E), (lib.mldsa_verify_arith, (h, 0, p, p, p, p, p, 1, z), E),
        (lib.mldsa_expand_a, (h, 66, p, p, 1, z), E), (lib.mldsa_expand_a, (h, 65, z, p, 1, z), E), (lib.mldsa_expand_s, (h, 44, p, z, 1, z), E),
        (lib.mldsa_expand_mask, (h, 44, p, z, p, 1, z), E), (lib.mldsa_sample_in_ball, (h, 44, z, p, 1, z), E),
        # op level
        (lib.mldsa_verify, (h, 44, 0, p, p, p, 1, z, p, o, z, z, p, z, 1, z), E),                 # ok = NULL
        (lib.mldsa_verify, (h, 44, 7, p, p, p, 1, z, p, o, z, z, p, p, 1, z), E),                 # mode 7
        (lib.mldsa_verify, (h, 44, 0, p, p, p, 1, z, p, o, z, z, p, p, 8, z), E),                 # 1 key, 8 ops, no key_idx
        (lib.mldsa_verify, (h, 44, 0, p, p, p, 0, p, p, o, z, z, p, p, 8, z), E),                 # key_idx with n_keys = 0
        (lib.mldsa_verify, (h, 44, 0, z, z, z, 0, z, z, z, z, z, z, z, 0, z), OK),                # empty batch
        (lib.mldsa_verify, (h, 44, 0, p, p, p, 1, z, p, z, z, z, p, p, 1, z), E),                 # msg_off = NULL
        (lib.mldsa_verify_cached_a, (h, 44, 0, z, p, p, 1, z, p, o, z, z, p, p, 1, z), E),
        (lib.mldsa_sign, (h, 44, 0, p, p, p, p, p, p, 1, z, p, o, z, z, z, p, p, 1, z), E),       # rnd = NULL
        (lib.mldsa_sign, (h, 99, 0, p, p, p, p, p, p, 1, z, p, o, z, z, p, p, p, 1, z), E),
        (lib.mldsa_sign, (h, 44, 0, z, z, z, z, z, z, 0, z, z, z, z, z, z, z, z, 0, z), OK),
        (lib.mldsa_sign_async, (h, 44, 0, p, p, p, p, p, p, 1, z, p, o, z, z, p, p, z, 1, z), E),  # the asynchronous call needs `status`
        (lib.mldsa_sign_cached_a, (h, 44, 3, p, p, p, p, p, p, 1, z, p, o, z, z, p, p, p, 1, z), E),
        (lib.mldsa_keygen, (h, 44, z, p, p, 1, z), E), (lib.mldsa_keygen, (h, 45, p, p, p, 1, z), E), (lib.mldsa_keygen, (h, 44, z, z, z, 0, z), OK),
        (lib.mldsa_pk_expand, (h, 44, p, p, z, p, 1, z), E), (lib.mldsa_sk_expand, (h, 44, p, p, p, p, p, z, p, 1, z), E),
        (lib.mldsa_pk_into_bytes, (h, 44, p, z, p, 1, z), E), (lib.mldsa_sk_into_bytes, (h, 44, p, p, p, p, p, p, z, 1, z), E),
        (lib.mldsa_get_public_key, (h, 44, p, p, p, p, p, p, z, 1, z), E),
        # host-memory entry points and groups
        (lib.mldsa_verify_host, (h, 44, 0, z, 1, z, z, z, z, z, z, z, 1), E), (lib.mldsa_verify_host, (None, 44, 0, z, 0, z, z, z, z, z, z, z, 0), E),
        (lib.mldsa_sign_host, (h, 44, 0, z, 1, z, z, z, z, z, z, z, z, 1), E), (lib.mldsa_keygen_host, (h, 44, z, z, z, 3), E),
        (lib.mldsa_keygen_host, (h, 44, z, z, z, 0), OK),
        (lib.mldsa_verify_group, (None, 44, 0, z, 1), E), (lib.mldsa_group_sync, (None,), E), (lib.mldsa_group_allgather, (None, z, 1, 0), E),
        # housekeeping
        (lib.mldsa_reserve, (h, 44, 9, 100), E), (lib.mldsa_reserve, (h, 43, 2, 100), E), (lib.mldsa_set_option, (h, 99, 1), E),
        (lib.mldsa_set_option, (h, _lib.OPT_SPEC_MAX, 65), E), (lib.mldsa_get_stats, (h, None), E),
        (lib.mldsa_profile_report, (h, None, 0), E), (lib.mldsa_debug_secret_residue, (h, None, None), E),
        (lib.mldsa_debug_count_nonzero, (None, 16, None), E),
    ]
    for fn, a, want in calls:
        rc = fn(*a)
        assert rc == want, (fn.__name__, a[1:4], rc, lib.mldsa_last_error())
        if want != OK:
            assert lib.mldsa_last_error(), fn.__name__
    g = C.c_void_p()
    assert lib.mldsa_group_create(None, 2, C.byref(g)) == E and lib.mldsa_group_create((C.c_int * 1)(0), 0, C.byref(g)) == E
    assert lib.mldsa_group_create((C.c_int * 1)(77), 1, C.byref(g)) < 0 and not g.value
    hh = C.c_void_p()
    assert lib.mldsa_ctx_create(-1, C.byref(hh)) == E and lib.mldsa_ctx_create(0, None) == E
    lib.mldsa_ctx_destroy(None)
    lib.mldsa_group_destroy(None)
    # the context is still in working order
    pk, sk = m.keygen_from_seed([shake(b"err-key", 0)])
    sig = m.try_sign_with_seed(m.private_keys_from_bytes(sk), [b"still works"], [bytes(32)])
    assert m.verify(m.public_keys_from_bytes(pk), [b"still works"], sig).all()


def test_environment_knobs_are_read_only_when_the_process_asks_for_them():
    """include/mldsa_hip.h "Environment": a host's MLDSA_* variables do not re-schedule the library; MLDSA_TUNING_ENV=1 turns the
    measurement knobs on, per mldsa_ctx_create.  (Results never depend on them; what is checked is the option each knob shadows.)"""
    from fips204_amd.hotpath import HotPath
    names = ("MLDSA_TUNING_ENV", "MLDSA_SIGN_LANES", "MLDSA_SMALL_FUSED", "MLDSA_GRAPHS", "MLDSA_SPEC_TARGET")
    old = {k: os.environ.get(k) for k in names}
    OPT_GRAPHS, OPT_SPEC_TARGET, OPT_SIGN_LANES, OPT_SMALL_FUSED = 1, 2, 7, 13
    try:
        os.environ.update({"MLDSA_SIGN_LANES": "2", "MLDSA_SMALL_FUSED": "0", "MLDSA_GRAPHS": "2", "MLDSA_SPEC_TARGET": "4096"})
        got = {}
        for switch in (None, "0", "yes", "1"):
            os.environ.pop("MLDSA_TUNING_ENV", None)
            if switch is not None:
                os.environ["MLDSA_TUNING_ENV"] = switch
            h = HotPath(0)
            try:
                got[switch] = tuple(h.get_option(o) for o in (OPT_GRAPHS, OPT_SPEC_TARGET, OPT_SIGN_LANES, OPT_SMALL_FUSED))
            finally:
                h.close()
        assert got[None] == got["0"] == got["yes"] == (0, 65536, 0, 256), got   # the defaults: nothing was read
        assert got["1"] == (2, 4096, 2, 0), got
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
