"""Several contexts / devices behind one call: mldsa_group_* host-fed and device-resident, the verdict all-gather, contexts bound to
their device from any host thread.  (Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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


# ------------------------------------------------------------------------------ in-library batch split (mldsa_group_*)
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_over_two_contexts_matches_the_single_context_call(sets, devices):
    """mldsa_*_host_group over a group of contexts (here: all on GPU 0, the only device of the box) gives byte-identical
    keys, signatures and verdicts to the single-context host entry points, ragged split included (n % N != 0, and a batch
    smaller than the group).  Mirrors src/traits.rs:118-308, 330-362: host slices in, arrays out."""
    from fips204_amd.ml_dsa import MlDsaGroup
    m = sets[65]
    g = MlDsaGroup(65, devices)
    try:
        assert len(g) == len(devices)
        rng = np.random.default_rng(11)
        for n, nk in ((1001, 37), (2, 2), (4099, 4099)):
            xi = np.frombuffer(b"".join(shake(b"grp-key", i) for i in range(nk)), dtype=np.uint8)
            pk1, sk1 = m.keygen_host(xi)
            pk2, sk2 = g.keygen_host(xi)
            assert np.array_equal(pk1, pk2) and np.array_equal(sk1, sk2)
            msgs = [shake(b"grp-msg", i, int(rng.integers(0, 200))) for i in range(n)]
            ctxs = [shake(b"grp-ctx", i, i % 11) for i in range(n)]
            rnd = np.frombuffer(b"".join(shake(b"grp-rnd", i) for i in range(n)), dtype=np.uint8)
            kidx = None if nk == n else rng.integers(0, nk, n).astype(np.uint32)   # identity mapping walks with the slice
            s1 = m.sign_host(sk1, msgs, rnd, ctxs=ctxs, key_idx=kidx)
            s2 = g.sign_host(sk1, msgs, rnd, ctxs=ctxs, key_idx=kidx)
            assert np.array_equal(s1, s2)
            bad = s1.copy()
            bad[::5, 17] ^= 0x40
            v1 = m.verify_host(pk1, msgs, bad, ctxs=ctxs, key_idx=kidx)
            v2 = g.verify_host(pk1, msgs, bad, ctxs=ctxs, key_idx=kidx)
            assert np.array_equal(v1, v2) and not v2[::5].any() and v2[1::5].all()
        # an over-long ctx in ONE slice: that op's status is MLDSA_ERR_CTX_LEN, the mirror raises like the single call
        with pytest.raises(ValueError):
            g.sign_host(sk1[:3], [b"a", b"b", b"c"], rnd[:96], ctxs=[b"", b"", b"x" * 256])
        # errors of a slice surface with the slice's message
        with pytest.raises(Exception):
            g.verify_host(pk1[:2], [b"a"] * 4, s1[:4], key_idx=None)  # 2 keys, 4 ops, no key_idx
    finally:
        g.close()


def test_group_allgather_of_device_resident_verdicts(hp):
    """mldsa_group_allgather: every context's slice of verdict bytes ends up in every buffer.  On this 1-GPU box the group
    lists GPU 0 twice (device-to-device copies; RCCL refuses duplicate devices and is reported as such) and once (the
    ncclAllGather path with a world of one)."""
    import ctypes as C
    from fips204_amd import _lib
    lib = _lib.load()
    for devices, use_rccl in (([0, 0], 0), ([0, 0], -1), ([0], 1), ([0, 0, 0], 0)):
        ids = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        _lib.check(lib.mldsa_group_create(ids, len(devices), C.byref(g)))
        try:
            n = 1000 + len(devices)
            per = -(-n // len(devices))
            want = torch.arange(n, dtype=torch.int64, device="cuda").remainder(251).to(torch.uint8)
            bufs = []
            for i in range(len(devices)):
                b = torch.full((per * len(devices),), 255, dtype=torch.uint8, device="cuda")
                a, c = C.c_size_t(), C.c_size_t()
                _lib.check(lib.mldsa_group_shard(n, len(devices), i, C.byref(a), C.byref(c)))
                b[a.value:a.value + c.value] = want[a.value:a.value + c.value]
                bufs.append(b)
            torch.cuda.synchronize()
            arr = (C.c_void_p * len(devices))(*[b.data_ptr() for b in bufs])
            _lib.check(lib.mldsa_group_allgather(g, arr, n, use_rccl))
            for b in bufs:
                assert torch.equal(b[:n], want), (devices, use_rccl)
            if len(devices) > 1:
                assert lib.mldsa_group_allgather(g, arr, n, 1) != 0  # RCCL + duplicate devices: refused, not attempted
        finally:
            lib.mldsa_group_destroy(g)


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
