"""Host-memory entry points (mldsa_*_host: wire-format keys, pageable and page-locked buffers), HashML-DSA and OS-RNG entry points of the mirrors
(src/traits.rs:118-308, 330-362).  (Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("n,nk,passes,lanes", [(40000, 300, 0, 1), (700, 9, 0, 1), (20000, 64, 4096, 1), (20000, 64, 0, 2)])
def test_sign_host_writes_page_locked_signatures_directly(sets, n, nk, passes, lanes):
    """With a page-locked signature buffer mldsa_sign_host signs the whole batch in ONE call and k_export_done copies the
    signatures that finished in each round to the caller's memory (lib.rs:268-296 is the per-op contract: same bytes).
    Cases: a call in the direct path's size range (16 385 ... 131 072 ops), a small one (sub-batch path, captured and replayed
    as a hipGraph: three calls on the same buffers), a direct call cut into several passes (export offsets per pass), and a
    context set to two signing lanes (the export hangs off ONE lane's rounds: such a context takes the sub-batch path);
    refused ops get zero rows and their status."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    env = {"MLDSA_PASS_OPS_SIGN": str(passes)} if passes else {}
    if lanes != 1:
        env["MLDSA_SIGN_LANES"] = str(lanes)
    if env:
        env["MLDSA_TUNING_ENV"] = "1"  # the knobs are read only when the process asks for them (include/mldsa_hip.h "Environment")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h2 = HotPath(0)
        h2.set_option(1, 1)  # MLDSA_OPT_GRAPHS = 1 (not the default since round 5): signing calls of <= 16384 ops replay as hipGraphs
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        m = MlDsa(65, hotpath=h2)
        rng = np.random.default_rng(n)
        xi = np.frombuffer(b"".join(shake(b"dir-key", i) for i in range(nk)), dtype=np.uint8)
        pk, sk = m.keygen_host(xi)
        msgs = [shake(b"dir-msg", i, int(rng.integers(0, 120))) for i in range(n)]
        ctxs = [shake(b"dir-ctx", i, i % 5) for i in range(n)]
        rnd = np.frombuffer(b"".join(shake(b"dir-rnd", i) for i in range(n)), dtype=np.uint8)
        kidx = rng.integers(0, nk, n).astype(np.uint32)
        want = m.sign_host(sk, msgs, rnd, ctxs=ctxs, key_idx=kidx)          # pageable output: the sub-batch path
        keep_s, sig = _pinned((n, m.SIG_LEN), np.uint8)
        keep_t, st = _pinned((n,), np.int32)
        for rep in range(3):
            sig[...] = 0xA5
            got = m.sign_host(sk, msgs, rnd, ctxs=ctxs, key_idx=kidx, out=(sig, st))
            assert got.ctypes.data == sig.ctypes.data and np.array_equal(got, want), rep
            assert not st.any()
        stats = h2.stats()
        if n <= 16384:
            assert stats["graph_replays"] >= 1
        sk_o = [orc.sk_try_from_bytes(65, sk[i].tobytes()) for i in range(nk)]
        for i in rng.choice(n, 8, replace=False):
            assert sig[i].tobytes() == orc.sign_internal(65, sk_o[kidx[i]], msgs[i], rnd[32 * i:32 * i + 32].tobytes(), ctx=ctxs[i], mode=0)
        assert m.verify_host(pk, msgs, sig, ctxs=ctxs, key_idx=kidx).all()
        # refused ops: an over-long ctx in the middle of the batch
        ctxs2 = list(ctxs)
        bad = [1, n // 2, n - 1]
        for i in bad:
            ctxs2[i] = b"z" * 256
        sig[...] = 0xA5
        with pytest.raises(ValueError):
            m.sign_host(sk, msgs, rnd, ctxs=ctxs2, key_idx=kidx, out=(sig, st))
        good = np.ones(n, dtype=bool)
        good[bad] = False
        assert np.array_equal(sig[good], want[good]) and not sig[bad].any()
        assert (st[bad] == -2).all() and not st[good].any()
    finally:
        h2.close()


def test_sign_host_direct_with_one_key_per_op(sets):
    """The direct path without key_idx (op i signs with key i; nothing goes up ahead of the keys) and with it naming the same
    keys: the same signatures, equal to the device-resident call's."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    n = 16500
    h2 = HotPath(0)
    try:
        m = MlDsa(44, hotpath=h2)
        xi = np.frombuffer(b"".join(shake(b"d1-key", i) for i in range(n)), dtype=np.uint8)
        pk, sk = m.keygen_host(xi)
        msgs = [shake(b"d1-msg", i, i % 77) for i in range(n)]
        rnd = np.frombuffer(b"".join(shake(b"d1-rnd", i) for i in range(n)), dtype=np.uint8)
        keep_s, sig = _pinned((n, m.SIG_LEN), np.uint8)
        keep_t, st = _pinned((n,), np.int32)
        m.sign_host(sk, msgs, rnd, out=(sig, st))
        assert not st.any() and m.verify_host(pk, msgs, sig).all()
        first = sig.copy()
        sig[...] = 0xA5
        m.sign_host(sk, msgs, rnd, key_idx=np.arange(n, dtype=np.uint32), out=(sig, st))
        assert np.array_equal(sig, first)
        sk_o = [orc.sk_try_from_bytes(44, sk[i].tobytes()) for i in (0, 1, n - 1)]
        for o, i in zip(sk_o, (0, 1, n - 1)):
            assert sig[i].tobytes() == orc.sign_internal(44, o, msgs[i], rnd[32 * i:32 * i + 32].tobytes(), ctx=b"", mode=0)
    finally:
        h2.close()


@pytest.mark.parametrize("group", [False, True])
def test_host_entry_points_refuse_malformed_tables(sets, group):
    """mldsa_verify_host / mldsa_sign_host (and the group forms) memcpy by the caller's offsets: a decreasing pair fails the whole
    call with MLDSA_ERR_PARAM before anything is copied or uploaded; a table naming bytes of a NULL array likewise."""
    from fips204_amd import _lib
    from fips204_amd.ml_dsa import MlDsaGroup
    m = sets[44]
    lib = m.lib
    n = 300
    xi = np.frombuffer(b"".join(shake(b"hoff-key", i) for i in range(4)), dtype=np.uint8)
    pk, sk = m.keygen_host(xi)
    msgs = [shake(b"hoff-msg", i, i % 90) for i in range(n)]
    rnd = np.zeros(n * 32, dtype=np.uint8)
    kidx = (np.arange(n) % 4).astype(np.uint32)
    sig = m.sign_host(sk, msgs, rnd, key_idx=kidx)
    mflat, moff = table(msgs)
    g = MlDsaGroup(44, [0, 0]) if group else None
    handle = g._g if group else m.hp._h
    vfn = lib.mldsa_verify_host_group if group else lib.mldsa_verify_host
    sfn = lib.mldsa_sign_host_group if group else lib.mldsa_sign_host
    vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
    try:
        ok = np.zeros(n, dtype=np.uint8)
        out = np.zeros((n, m.SIG_LEN), dtype=np.uint8)
        st = np.zeros(n, dtype=np.int32)
        assert vfn(handle, 44, 0, vp(pk), 4, vp(kidx), vp(mflat), vp(moff), None, None, vp(sig), vp(ok), n) == 0 and ok.all()
        for k, val in ((150, None), (1, 2 ** 64 - 1), (299, 0)):
            bad = moff.copy()
            bad[k] = np.uint64(val) if val is not None else bad[k - 1] - np.uint64(1) if bad[k - 1] else np.uint64(0)
            if (bad[1:] >= bad[:-1]).all():
                continue
            t0 = time.perf_counter()
            assert vfn(handle, 44, 0, vp(pk), 4, vp(kidx), vp(mflat), vp(bad), None, None, vp(sig), vp(ok), n) == _lib.ERR_PARAM
            assert b"decreases at entry" in lib.mldsa_last_error()
            assert sfn(handle, 44, 0, vp(sk), 4, vp(kidx), vp(mflat), vp(bad), None, None, vp(rnd), vp(out), vp(st), n) == _lib.ERR_PARAM
            # the ctx table is checked the same way
            assert vfn(handle, 44, 0, vp(pk), 4, vp(kidx), vp(mflat), vp(moff), vp(mflat), vp(bad), vp(sig), vp(ok), n) == _lib.ERR_PARAM
            assert time.perf_counter() - t0 < 0.5
        # offsets that name bytes of a NULL array
        assert vfn(handle, 44, 0, vp(pk), 4, vp(kidx), None, vp(moff), None, None, vp(sig), vp(ok), n) == _lib.ERR_PARAM
        assert sfn(handle, 44, 0, vp(sk), 4, vp(kidx), vp(mflat), vp(moff), None, vp(moff), vp(rnd), vp(out), vp(st), n) == _lib.ERR_PARAM
        # and the context still works afterwards
        assert vfn(handle, 44, 0, vp(pk), 4, vp(kidx), vp(mflat), vp(moff), None, None, vp(sig), vp(ok), n) == 0 and ok.all()
    finally:
        if g:
            g.close()


# ------------------------------------------------------------------------------ page-locked extents (host_api.hip:382)
def test_partially_registered_signature_buffer_takes_the_safe_path(sets):
    """mldsa_sign_host writes finished signatures straight into a page-locked caller buffer (k_export_done).  The decision used to
    look at the FIRST byte only: a buffer whose head alone is registered (hipHostRegister of a sub-range) would make the GPU store
    into unmapped host memory -- a fault that aborts the process.  The whole extent is checked now; such a buffer goes through
    the sub-batch path and the signatures are the same."""
    m = sets[44]
    hip = C.CDLL("libamdhip64.so")
    n, nk = 20000, 8        # > 16 384 ops: the size range of the direct export
    xi = np.frombuffer(b"".join(shake(b"pin-key", i) for i in range(nk)), dtype=np.uint8)
    pk, sk = m.keygen_host(xi)
    msgs = [shake(b"pin-msg", i) for i in range(n)]
    rnd = np.frombuffer(b"".join(shake(b"pin-rnd", i) for i in range(n)), dtype=np.uint8)
    kidx = (np.arange(n) % nk).astype(np.uint32)
    want = m.sign_host(sk, msgs, rnd, key_idx=kidx)          # pageable output
    # fully page-locked output: the direct path
    pinned = C.c_void_p()
    assert m.lib.mldsa_host_alloc(C.byref(pinned), n * m.SIG_LEN) == 0
    try:
        full = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(n * m.SIG_LEN,)).reshape(n, m.SIG_LEN)
        full[:] = 0
        st = np.zeros(n, dtype=np.int32)
        got = m.sign_host(sk, msgs, rnd, key_idx=kidx, out=(full, st))
        assert np.array_equal(got, want)
    finally:
        m.lib.mldsa_host_free(pinned)
    # head registered, tail pageable
    raw = np.zeros(n * m.SIG_LEN + 8192, dtype=np.uint8)
    base = (raw.ctypes.data + 4095) & ~4095
    view = raw[base - raw.ctypes.data:][:n * m.SIG_LEN].reshape(n, m.SIG_LEN)
    reg_bytes = (n * m.SIG_LEN // 3) & ~4095
    assert hip.hipHostRegister(C.c_void_p(base), C.c_size_t(reg_bytes), C.c_uint(0)) == 0
    try:
        st = np.zeros(n, dtype=np.int32)
        got = m.sign_host(sk, msgs, rnd, key_idx=kidx, out=(view, st))
        assert np.array_equal(got, want) and (st == 0).all()
        assert m.verify_host(pk, msgs, got, key_idx=kidx).all()
    finally:
        hip.hipHostUnregister(C.c_void_p(base))
