"""Boundary hardening (round 4): untrusted offset tables, workspace shrinking, page-locked extents, device-resident group calls.

The reference never panics on untrusted input (fuzz/fuzz_targets/fuzz_all.rs:25-37) and returns a plain error / `false`
(src/lib.rs:274, 368-370).  Its slices carry their own lengths; the C ABI's offset tables are the caller's and can be wrong:
an op with a malformed pair is refused on its own, every other op of the batch still equals the oracle's result."""
import ctypes as C
import hashlib
import time

import numpy as np
import pytest
import torch

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
    return hashlib.shake_256(tag + int(i).to_bytes(8, "little")).digest(n)


def dev(a):
    return torch.from_numpy(np.array(a, copy=True)).cuda()


def dev_off(off):
    return torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).cuda()


def table(items):
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in items], out=off[1:])
    return np.frombuffer(b"".join(items) + bytes(16), dtype=np.uint8), off


def pairs_ok(off):
    """per op: the pair lies in order inside [off[0], off[n]] (the rule of k_mu / include/mldsa_hip.h)"""
    off = [int(x) for x in off]
    lo, hi = off[0], off[-1]
    return np.array([lo <= a <= b <= hi for a, b in zip(off[:-1], off[1:])], dtype=bool)


def corruptions(off, rng):
    """name -> corrupted copy of a monotonic table (n + 1 entries), the damage in the middle of the batch"""
    n = off.size - 1
    k = n // 2
    out = {}
    assert off[k] > 0
    t = off.copy(); t[k + 1] = t[k] - np.uint64(1)
    out["decreasing"] = t
    t = off.copy(); t[k:k + 5] = t[k]
    out["equal_run"] = t                                    # legal: four empty byte strings
    t = off.copy(); t[k + 1] = np.uint64(1) << np.uint64(63)
    out["overshoot"] = t                                    # far past the end: two ops refused
    t = off.copy(); t[k] = np.uint64(2 ** 64 - 8); t[k + 1] = np.uint64(2 ** 64 - 1)
    out["near_2_64"] = t
    t = off.copy(); t[k + 1] = t[0]
    out["back_to_start"] = t
    t = off.copy(); t[-1] = t[n // 4]
    out["short_last_entry"] = t                             # the call vouches for fewer bytes than the table names
    t = off.copy(); idx = rng.choice(np.arange(1, n), 40, replace=False); t[idx] = rng.integers(0, 2 ** 63, 40, dtype=np.uint64)
    out["forty_random_entries"] = t
    return out


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


# ------------------------------------------------------------------------------ device-resident group calls
@pytest.mark.parametrize("devices", [[0, 0, 0], [0]])
def test_device_resident_group_calls_match_the_single_context(sets, devices):
    """mldsa_keygen_group / mldsa_sign_group / mldsa_verify_group: slice i of the batch already lives on device
    i; one host thread drives all of them.  Byte-identical to the single-context calls on the same ops, ragged split and an empty
    slice included; wait=0 + mldsa_group_sync; and the verdicts gathered with mldsa_group_allgather WITHOUT a host
    synchronisation in between (the gather waits for each context's last call on the device).  src/traits.rs:118-308, 330-362."""
    from fips204_amd.ml_dsa import MlDsaGroup
    pset = 65
    m = sets[pset]
    g = MlDsaGroup(pset, devices)
    N = len(devices)
    try:
        for n, nk in ((1001, 37), (2, 2), (4099, 64)):
            rng = np.random.default_rng(n)
            xi = np.frombuffer(b"".join(shake(b"dg-key", i) for i in range(nk)), dtype=np.uint8).reshape(nk, 32)
            msgs = [shake(b"dg-msg", i, int(rng.integers(0, 120))) for i in range(n)]
            ctxs = [shake(b"dg-ctx", i, i % 5) for i in range(n)]
            rnd = np.frombuffer(b"".join(shake(b"dg-rnd", i) for i in range(n)), dtype=np.uint8).reshape(n, 32)
            kidx = rng.integers(0, nk, n).astype(np.uint32)
            # single context
            pk0, sk0 = m.keygen_from_seed(dev(xi))
            sks0, pks0 = m.private_keys_from_bytes(sk0), m.public_keys_from_bytes(pk0)
            sig0 = host(m.try_sign_with_seed(sks0, msgs, [bytes(r) for r in rnd], ctxs=ctxs, key_idx=kidx)).copy()
            bad = sig0.copy()
            bad[::4, 9] ^= 0x10
            v0 = m.verify(pks0, msgs, dev(bad), ctxs=ctxs, key_idx=kidx)
            # keygen: the seeds sharded
            ks = []
            for i in range(N):
                a, c = g.shard(nk, i)
                ks.append(dict(xi=dev(xi[a:a + c]) if c else torch.zeros(32, dtype=torch.uint8, device="cuda"),
                               pk=torch.zeros((max(c, 1), m.PK_LEN), dtype=torch.uint8, device="cuda"),
                               sk=torch.zeros((max(c, 1), m.SK_LEN), dtype=torch.uint8, device="cuda"), n_keys=c))
            g.keygen_group(ks, wait=False)
            g.sync()
            pk1 = np.concatenate([host(s["pk"])[:s["n_keys"]] for s in ks])
            sk1 = np.concatenate([host(s["sk"])[:s["n_keys"]] for s in ks])
            assert np.array_equal(pk1, host(pk0)) and np.array_equal(sk1, host(sk0))
            # sign + verify: every device holds the whole (small) key table, the ops are sharded
            per = -(-n // N)
            oks = [torch.full((per * N,), 7, dtype=torch.uint8, device="cuda") for _ in range(N)]
            ss, vs = [], []
            for i in range(N):
                a, c = g.shard(n, i)
                mi = g.on_device(i)
                sks_i, pks_i = mi.private_keys_from_bytes(sk0), mi.public_keys_from_bytes(pk0)
                mf, mo = table(msgs[a:a + c])
                cf, co = table(ctxs[a:a + c])
                common = dict(msg_buf=dev(mf), msg_off=dev_off(mo), ctx_buf=dev(cf), ctx_off=dev_off(co),
                              key_idx=dev(kidx[a:a + c].view(np.int32)) if c else None, n_ops=c)
                ss.append(dict(common, sks=sks_i, rnd=dev(rnd[a:a + c]) if c else torch.zeros(32, dtype=torch.uint8, device="cuda"),
                               sigs=torch.zeros((max(c, 1), m.SIG_LEN), dtype=torch.uint8, device="cuda"),
                               status=torch.full((max(c, 1),), 5, dtype=torch.int32, device="cuda")))
                vs.append(dict(common, pks=pks_i, sigs=dev(bad[a:a + c]) if c else torch.zeros(m.SIG_LEN, dtype=torch.uint8, device="cuda"),
                               ok=oks[i][a:a + c] if c else oks[i][:0]))
            torch.cuda.synchronize()
            for wait in (True, False):
                for s in ss:
                    s["sigs"].zero_()
                g.sign_group(ss, wait=wait)
                if not wait:
                    g.sync()
                sig1 = np.concatenate([host(s["sigs"])[:s["n_ops"]] for s in ss])
                assert np.array_equal(sig1, sig0), (n, wait)
                assert all((host(s["status"])[:s["n_ops"]] == 0).all() for s in ss)
            # verify without waiting, then gather: no host synchronisation between the two
            for i in range(N):   # (an empty slice's view cannot carry a device pointer: give the call a dummy)
                if vs[i]["n_ops"] == 0:
                    vs[i]["ok"] = torch.zeros(1, dtype=torch.uint8, device="cuda")
            g.verify_group(vs, wait=False)
            g.allgather(oks, n, use_rccl=0)
            for b in oks:
                assert np.array_equal(host(b)[:n].astype(bool), v0), n
            g.verify_group(vs, wait=True)
            assert np.array_equal(np.concatenate([host(oks[i])[g.shard(n, i)[0]:sum(g.shard(n, i))] for i in range(N)]).astype(bool), v0)
    finally:
        g.close()


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


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_verify_pk_equals_try_from_bytes_plus_verify(sets, pset):
    """mldsa_verify_pk = PublicKey::try_from_bytes (src/ml_dsa.rs:477-498) + Verifier::verify (351-437) in one call.  On a fuzzed batch
    (good and damaged signatures, arbitrary public-key bytes: fuzz_all.rs:25-37, fuzz_verify.rs:17-31) its verdicts equal those of
    mldsa_pk_expand + mldsa_verify AND the oracle's -- with a key table + key_idx, with one key per op (identity mapping), and
    when the call runs in several passes (a capped workspace)."""
    from test_gpu_round3 import fuzz_batch
    from fips204_amd import _lib
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    m = sets[pset]
    n, nk = 8192, 128
    pk_all, kidx, msgs, sig, cls, changed = fuzz_batch(m, pset, n, nk, 9100 + pset)
    d_pk, d_sig = dev(pk_all), dev(sig)
    pks = m.public_keys_from_bytes(d_pk)
    want = m.verify(pks, msgs, d_sig, key_idx=kidx, mode=0)
    pk_o = [orc.pk_try_from_bytes(pset, pk_all[i].tobytes()) for i in range(2 * nk)]
    want_o = np.asarray(orc.verify_batch_mt(pset, pk_o, kidx, msgs, [sig[i].tobytes() for i in range(n)], 16, 1, mode=0), dtype=bool)
    assert np.array_equal(want, want_o) and np.array_equal(want, ~changed)
    # a key table + key_idx
    assert np.array_equal(m.verify_pk(d_pk, msgs, d_sig, key_idx=kidx, mode=0), want)
    # one wire-format key per op
    per_op = dev(pk_all[kidx])
    assert np.array_equal(m.verify_pk(per_op, msgs, d_sig, mode=0), want)
    # ragged tail, ctxs, internal mode on a prefix
    ctxs = [shake(b"vpk-ctx", i, i % 9) for i in range(1001)]
    got = m.verify_pk(per_op[:1001], msgs[:1001], d_sig[:1001], ctxs=ctxs, mode=1)
    assert np.array_equal(got, m.verify(m.public_keys_from_bytes(per_op[:1001]), msgs[:1001], d_sig[:1001], ctxs=ctxs, mode=1))
    # several passes: a workspace cap that does not hold 8 192 ops' A_hat and keys
    hp2 = HotPath(0)
    try:
        hp2.set_option(_lib.OPT_WORKSPACE_CAP_MB, {44: 48, 65: 96, 87: 160}[pset])
        m2 = MlDsa(pset, hotpath=hp2)
        assert np.array_equal(m2.verify_pk(per_op, msgs, d_sig, mode=0), want)
        assert np.array_equal(m2.verify_pk(d_pk, msgs, d_sig, key_idx=kidx, mode=0), want)
        assert hp2.stats()["workspace_shrinks"] > 0
    finally:
        hp2.close()
    # argument errors
    lib, h = m.lib, m.hp._h
    assert lib.mldsa_verify_pk(h, pset, 0, None, 1, None, None, None, None, None, None, None, 1, None) == _lib.ERR_PARAM
    assert lib.mldsa_verify_pk(h, pset, 0, C.c_void_p(d_pk.data_ptr()), 4, None, None, C.c_void_p(d_pk.data_ptr()), None, None,
                               C.c_void_p(d_sig.data_ptr()), C.c_void_p(d_sig.data_ptr()), 8, None) == _lib.ERR_PARAM   # 4 keys, 8 ops, no key_idx


def test_scalar_reductions_as_a_seam(hp):
    """partial_reduce32 / full_reduce32 / center_mod (src/helpers.rs:61-95) element-wise through mldsa_reduce, bit-exact
    with the oracle's scalar functions over the whole input contract |a| < 2^31 - 2^22 (helpers.rs:62): edge values and 2^16 random ones."""
    lib = orc.lib()
    lim = 2 ** 31 - 2 ** 22
    rng = np.random.default_rng(55)
    edges = [0, 1, -1, orc.Q - 1, orc.Q, orc.Q + 1, -orc.Q, -orc.Q + 1, orc.Q // 2, orc.Q // 2 + 1, -(orc.Q // 2), -(orc.Q // 2) - 1, lim - 1, -lim + 1,
             8 * orc.Q, -8 * orc.Q, 2 ** 23, -2 ** 23, 2 ** 22, 255 * orc.Q]
    vals = np.concatenate([np.array(edges, dtype=np.int64), rng.integers(-lim + 1, lim, 65536 - len(edges))]).astype(np.int32)
    d = dev(vals.reshape(-1, 256))
    got = {"partial": host(hp.partial_reduce32(d)).ravel(), "full": host(hp.full_reduce32(d)).ravel(), "center": host(hp.center_mod(d)).ravel()}
    for name, fn in (("partial", lib.orc_partial_reduce32), ("full", lib.orc_full_reduce32), ("center", lib.orc_center_mod)):
        want = np.array([fn(int(v)) for v in vals], dtype=np.int32)
        assert np.array_equal(got[name], want), (name, np.nonzero(got[name] != want)[0][:5])
    assert (np.abs(got["partial"].astype(np.int64)) < orc.Q).all() and (got["full"] >= 0).all() and (got["full"] < orc.Q).all()
    assert (got["center"] > -(orc.Q // 2) - 1).all() and (got["center"] <= orc.Q // 2).all()
    assert (got["full"].astype(np.int64) - vals) .__mod__(orc.Q).max() == 0
    from fips204_amd import _lib
    assert hp.lib.mldsa_reduce(hp._h, 7, C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 1, None) == _lib.ERR_PARAM


@pytest.mark.parametrize("pset", [44, 65])
def test_rounding_functions_as_a_seam(hp, pset):
    """Power2Round / Decompose / HighBits / LowBits / MakeHint / UseHint (src/high_low.rs:15-192) element-wise through
    mldsa_rounding, bit-exact with the oracle's scalar functions -- both gamma2 values (ML-DSA-44: (q-1)/88, ML-DSA-65/87: (q-1)/32), the
    wrap-around points of Decompose (r1 = 0 when r - r0 = q - 1) and of UseHint (m = 44 / 16), non-canonical representatives."""
    from fips204_amd import _lib
    lib = orc.lib()
    g2 = orc.params(pset).gamma2
    q = orc.Q
    rng = np.random.default_rng(61 + pset)
    edges = [0, 1, q - 1, q - 2, g2, g2 + 1, g2 - 1, 2 * g2, 2 * g2 + 1, 2 * g2 - 1, q - 1 - g2, q - g2, q - g2 + 1, q // 2, q // 2 + 1,
             (q - 1) - 2 * g2, 4096, 4095, 4097, 8191, 8192, 8193]
    canon = np.concatenate([np.array(edges, dtype=np.int64), rng.integers(0, q, 32768 - len(edges))]).astype(np.int32)
    # power2round: [0, q)
    r1, r0 = (host(x).ravel() for x in hp.rounding(pset, _lib.ROUND_POWER2ROUND, dev(canon.reshape(-1, 256))))
    w1, w0 = np.zeros_like(canon), np.zeros_like(canon)
    lib.orc_power2round(canon.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p), w0.ctypes.data_as(C.c_void_p), C.c_size_t(canon.size))
    assert np.array_equal(r1, w1) and np.array_equal(r0, w0)
    assert np.array_equal((r1.astype(np.int64) << 13) + r0, canon) and (r0 > -4096).all() and (r0 <= 4096).all()
    # decompose / high_bits / low_bits: any representative full_reduce32 accepts
    lim = 2 ** 31 - 2 ** 22
    anyrep = np.concatenate([canon[:16384], canon[16384:24576] - q, rng.integers(-lim + 1, lim, 8192).astype(np.int32)]).astype(np.int32)
    d = dev(anyrep.reshape(-1, 256))
    r1, r0 = (host(x).ravel() for x in hp.rounding(pset, _lib.ROUND_DECOMPOSE, d))
    hb, lb = host(hp.rounding(pset, _lib.ROUND_HIGH_BITS, d)).ravel(), host(hp.rounding(pset, _lib.ROUND_LOW_BITS, d)).ravel()
    a1, a0 = C.c_int32(), C.c_int32()
    want = np.zeros((anyrep.size, 2), dtype=np.int32)
    for i, v in enumerate(anyrep):
        lib.orc_decompose(g2, int(v), C.byref(a1), C.byref(a0))
        want[i] = a1.value, a0.value
    assert np.array_equal(r1, want[:, 0]) and np.array_equal(r0, want[:, 1]) and np.array_equal(hb, r1) and np.array_equal(lb, r0)
    m = (q - 1) // (2 * g2)
    assert r1.min() == 0 and r1.max() == m - 1 and (np.abs(r0) <= g2).all()
    assert ((r1.astype(np.int64) * 2 * g2 + r0 - anyrep) % q == 0).all()
    # make_hint(z, r) / use_hint(h, r): the signer's ranges (z = -c t0 as q - ct0, r = w - c s2 + c t0) and the verifier's
    z = np.concatenate([rng.integers(-g2, g2 + 1, 16384), q - rng.integers(0, g2, 16384)]).astype(np.int32)
    r = np.concatenate([rng.integers(-q + 1, q, 16384), rng.integers(0, q, 16384)]).astype(np.int32)
    h = host(hp.rounding(pset, _lib.ROUND_MAKE_HINT, dev(z.reshape(-1, 256)), dev(r.reshape(-1, 256)))).ravel()
    assert np.array_equal(h, np.array([lib.orc_make_hint(g2, int(a), int(b)) for a, b in zip(z, r)], dtype=np.int32)) and 0 < h.sum() < h.size
    hin = rng.integers(0, 2, canon.size).astype(np.int32)
    u = host(hp.rounding(pset, _lib.ROUND_USE_HINT, dev(hin.reshape(-1, 256)), dev(canon.reshape(-1, 256)))).ravel()
    assert np.array_equal(u, orc.use_hint_vec(g2, hin, canon)) and u.min() == 0 and u.max() == m - 1
    # FIPS 204 lemma: use_hint(make_hint(z, r), r) == high_bits(r + z) for |z| <= gamma2
    zz, rr = z[:16384], canon[:16384]
    hh = hp.rounding(pset, _lib.ROUND_MAKE_HINT, dev(zz.reshape(-1, 256)), dev(rr.reshape(-1, 256)))
    lhs = host(hp.rounding(pset, _lib.ROUND_USE_HINT, hh, dev(rr.reshape(-1, 256)))).ravel()
    rhs = host(hp.rounding(pset, _lib.ROUND_HIGH_BITS, dev(((rr.astype(np.int64) + zz) % q).astype(np.int32).reshape(-1, 256)))).ravel()
    assert np.array_equal(lhs, rhs)
    # argument errors
    p = C.c_void_p(d.data_ptr())
    assert hp.lib.mldsa_rounding(hp._h, pset, 9, p, p, p, p, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_rounding(hp._h, 50, 0, p, p, p, p, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_rounding(hp._h, pset, _lib.ROUND_MAKE_HINT, p, None, p, None, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_rounding(hp._h, pset, _lib.ROUND_DECOMPOSE, p, None, p, None, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_rounding(hp._h, pset, _lib.ROUND_DECOMPOSE, None, None, None, None, 0, None) == _lib.OK


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


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_cooperative_and_lane_per_state_sponges_agree(hp, sets, pset):
    """MLDSA_OPT_COOP_HASH: small calls run ExpandA, ExpandMask and the fixed-shape hashes wave-cooperatively (csrc/keccak_coop.h), large
    ones lane-per-state.  Same bytes either way: keys, signatures and verdicts of 1, 2, 7, 64, 137 and 700 ops (the last beyond the
    cooperative ExpandA's range, inside the hashes') with the option off and on, and against the oracle."""
    from fips204_amd import _lib
    m = sets[pset]
    rng = np.random.default_rng(90 + pset)
    old = hp.get_option(_lib.OPT_COOP_HASH)
    assert old == 1
    try:
        for n in (1, 2, 7, 64, 137, 700):
            xi = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            msgs = [rng.integers(0, 256, int(rng.integers(0, 120)), dtype=np.uint8).tobytes() for _ in range(n)]
            rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            out = {}
            for coop in (0, 1):
                hp.set_option(_lib.OPT_COOP_HASH, coop)
                assert hp.get_option(_lib.OPT_COOP_HASH) == coop
                pk, sk = m.keygen_from_seed(dev(xi))
                sig = m.try_sign_with_seed(m.private_keys_from_bytes(sk), msgs, [r.tobytes() for r in rnd])
                bad = sig.clone()
                bad[::3, 40] ^= 1
                pks = m.public_keys_from_bytes(pk)
                as_np = lambda t: host(t) if hasattr(t, "cpu") else np.asarray(t)
                out[coop] = (host(pk), host(sk), host(sig), as_np(m.verify(pks, msgs, sig)), as_np(m.verify(pks, msgs, bad)))
            for a, b in zip(out[0], out[1]):
                assert np.array_equal(a, b), n
            assert out[1][3].all() and not out[1][4][::3].any() and out[1][4][1::3].all()
            j = int(rng.integers(0, n))
            pk_o, sk_o = orc.keygen_from_seed(pset, xi[j].tobytes())
            assert out[1][0][j].tobytes() == orc.pk_into_bytes(pset, pk_o) and out[1][1][j].tobytes() == orc.sk_into_bytes(pset, sk_o)
            assert out[1][2][j].tobytes() == orc.sign_internal(pset, sk_o, msgs[j], rnd[j].tobytes(), mode=orc.MODE_PURE)
    finally:
        hp.set_option(_lib.OPT_COOP_HASH, old)
    assert hp.lib.mldsa_set_option(hp._h, _lib.OPT_COOP_HASH, 2) == _lib.ERR_PARAM
