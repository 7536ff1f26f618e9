"""GPU parity derived from the reference's fuzz targets and soak test (round 3):

  * fuzz/fuzz_targets/fuzz_all.rs:14-15   a good signature XOR an arbitrary mask is verified
  * fuzz/fuzz_targets/fuzz_all.rs:25-37   arbitrary PUBLIC-KEY bytes are deserialised and garbage signatures verified under them
  * fuzz/fuzz_targets/fuzz_verify.rs:17-31 arbitrary (pk, sig) byte strings through try_from_bytes + verify
  * fuzz/fuzz_targets/fuzz_sign.rs         arbitrary SECRET-KEY bytes sign
  * tests/integration.rs:22-53 (`forever`) randomized keygen / sign / verify / flip loop

Every verdict and every signature is compared with the oracle's (bit-exact), at BASELINE batch sizes for the verifier."""
import hashlib
import os
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
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# mask classes of the fuzzed verify batch (op i gets class i % len(CLASSES))
CLASSES = ("good", "bit", "byte", "dense1pct", "dense50pct", "random_sig", "mask_ctilde", "mask_z", "mask_hints",
           "random_pk", "random_pk_random_sig")


def fuzz_batch(m, pset, n, nk, seed):
    """(pk bytes [2 nk], key_idx, msgs, sigs [n]) -- keys 0 .. nk-1 are generated keys, nk .. 2 nk-1 are random bytes."""
    rng = np.random.default_rng(seed)
    p = m.params
    xi = [shake(b"fuzz-key%d" % pset, i) for i in range(nk)]
    pk, sk = m.keygen_from_seed(xi)
    sks = m.private_keys_from_bytes(sk)
    msgs = [shake(b"fuzz-msg", i) for i in range(n)]
    rnd = [shake(b"fuzz-rnd", i) for i in range(n)]
    kidx = (np.arange(n) * 5 % nk).astype(np.uint32)
    sig = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx, mode=0)).copy()
    good = sig.copy()
    pk_all = np.concatenate([host(pk), rng.integers(0, 256, (nk, m.PK_LEN), dtype=np.uint8)])
    cls = np.arange(n) % len(CLASSES)
    L = m.SIG_LEN
    cb = 18 if p.gamma1 == (1 << 17) else 20
    z0, h0 = p.ctilde_len, p.ctilde_len + p.l * 32 * cb  # sigEncode sections: c~ | z | hints (encodings.rs:238-276)
    def xor_mask(rows, lo, hi, density):
        mask = (rng.random((rows.size, hi - lo, 8)) < density)
        sig[rows, lo:hi] ^= np.packbits(mask, axis=2, bitorder="little")[:, :, 0]
    for c, name in enumerate(CLASSES):
        rows = np.nonzero(cls == c)[0]
        if name == "bit":
            sig[rows, rng.integers(0, L, rows.size)] ^= (1 << rng.integers(0, 8, rows.size)).astype(np.uint8)
        elif name == "byte":
            sig[rows, rng.integers(0, L, rows.size)] ^= rng.integers(1, 256, rows.size).astype(np.uint8)
        elif name == "dense1pct":
            xor_mask(rows, 0, L, 0.01)
        elif name == "dense50pct":
            xor_mask(rows, 0, L, 0.5)
        elif name in ("random_sig", "random_pk_random_sig"):
            sig[rows] = rng.integers(0, 256, (rows.size, L), dtype=np.uint8)
        elif name == "mask_ctilde":
            xor_mask(rows, 0, z0, 0.02)
        elif name == "mask_z":
            xor_mask(rows, z0, h0, 0.0005)
        elif name == "mask_hints":
            xor_mask(rows, h0, L, 0.01)
        if name.startswith("random_pk"):
            kidx[rows] += nk  # verified under arbitrary public-key bytes
    changed = (sig != good).any(axis=1) | (kidx >= nk)  # a sparse mask may leave a signature as it was
    return pk_all, kidx, msgs, sig, cls, changed


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


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_arbitrary_secret_key_bytes_sign_like_the_oracle(sets, pset):
    """fuzz_sign.rs: PrivateKey::try_from_bytes accepts any bytes (conversion.rs:259-260 never rejects) and the signer computes
    with them -- out-of-range eta fields, a tr that is no hash of anything.  Signatures must be the oracle's byte for byte,
    and the verdict under the matching get_public_key() must be the oracle's too."""
    m = sets[pset]
    rng = np.random.default_rng(8000 + pset)
    nk, n = 48, 1536
    sk = rng.integers(0, 256, (nk, m.SK_LEN), dtype=np.uint8)
    sk[0, 128:] = 0xFF
    sk[1, 128:] = 0x00
    sks = m.private_keys_from_bytes(dev(sk))
    msgs = [shake(b"rsk-msg", i, 1 + i % 90) for i in range(n)]
    rnd = [shake(b"rsk-rnd", i) if i % 3 else bytes(32) for i in range(n)]
    ctxs = [shake(b"rsk-ctx", i, i % 7) for i in range(n)]
    kidx = (np.arange(n) % nk).astype(np.uint32)
    sig = host(m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, key_idx=kidx, mode=0))
    sk_o = [orc.sk_try_from_bytes(pset, sk[i].tobytes()) for i in range(nk)]
    for i in range(n):
        want = orc.sign_internal(pset, sk_o[kidx[i]], msgs[i], rnd[i], ctx=ctxs[i], mode=0)
        assert sig[i].tobytes() == want, i
    pks = m.get_public_key(sks)
    pkb = host(m.public_keys_into_bytes(pks))
    got = m.verify(pks, msgs, dev(sig), ctxs=ctxs, key_idx=kidx, mode=0)
    for i in range(n):
        pk_o = orc.get_public_key(pset, sk_o[kidx[i]])
        if i < nk:
            assert pkb[i].tobytes() == orc.pk_into_bytes(pset, pk_o), i
        assert bool(got[i]) == orc.verify_internal(pset, pk_o, msgs[i], sig[i].tobytes(), ctx=ctxs[i], mode=0), i


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
    old = {k: os.environ.get(k) for k in ("MLDSA_PASS_OPS", "MLDSA_PASS_OPS_SIGN")}
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


# ------------------------------------------------------------------------------ soak (tests/integration.rs:22-53 `forever`)
def test_soak_random_shapes_knobs_and_modes_against_the_oracle(hp, sets):
    """A seeded, time-bounded version of the reference's `forever` loop: random parameter set, batch size (1 ... 70 000,
    log-uniform), key count, message / ctx lengths, interface mode and library knobs (graph replay, speculation target and
    width, planned rounds -> the extra-round path, signing lanes, synchronous / asynchronous signing), keygen -> sign ->
    verify -> flip -> verify on the device, with a sample of every iteration's keys, signatures and verdicts compared with
    the oracle.  Shapes repeat and alternate on ONE context, so workspace regrowth, the graph cache and the loop's control
    block are exercised the way a long-running service would."""
    seconds = float(os.environ.get("MLDSA_SOAK_SECONDS", "60"))
    seed = int(os.environ.get("MLDSA_SOAK_SEED", "20260203"))
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    defaults = {o: hp.get_option(o) for o in (1, 2, 3, 6, 7, 10)}  # 10 = MLDSA_OPT_SIGN_LOOKAHEAD
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
                n = int(np.exp(rng.uniform(0, np.log(70000))))
                nk = int(min(n, np.exp(rng.uniform(0, np.log(600)))))
                shapes.append((n, nk))
            mode = int(rng.choice([0, 0, 1, 2]))
            knobs = {1: int(rng.choice([0, 1, 2])), 2: int(rng.choice([1024, 8192, 40000, 65536, 150000])),
                     3: int(rng.choice([1, 4, 32, 64])), 6: int(rng.choice([0, 0, 1, 3])), 7: int(rng.choice([1, 1, 2])),
                     10: int(rng.choice([0, 1, 2]))}
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


# ------------------------------------------------------------------------------ mldsa_sign_host: signatures written to host memory round by round
def _pinned(shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    t = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=True)
    a = t.numpy()[:n].view(dtype).reshape(shape)
    a[...] = 0xA5 if dtype == np.uint8 else -7   # stale bytes: every row must be overwritten
    return t, a


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
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h2 = HotPath(0)
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


# ------------------------------------------------------------------------------ slot capacity at the speculation rule's thresholds
@pytest.mark.parametrize("n,env", [(22300, {}), (22200, {}), (65536, {"MLDSA_SPEC_ROWS": "81920"}), (50800, {"MLDSA_SPEC_ROWS": "81920", "MLDSA_SPEC_TARGET": "131072"})])
def test_rounds_just_below_a_speculation_threshold_fit_the_workspace(n, env):
    """Candidates per op = round((rows / m) ^ 0.85): a count just below a threshold of that rule has more slots than `rows`
    (22 300 ops x 3 = 66 900 for rows = 65 536).  The workspace is carved for the rule's true maximum and k_make_slots never makes
    more: every signature of such a batch verifies and a sample equals the oracle's (ml_dsa.rs:212-330: the FIRST accepted
    candidate, whatever the speculation)."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
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
