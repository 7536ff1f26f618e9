"""GPU stress / edge-case parity for the op-level API: multi-chunk batches, message lengths around
the SHAKE256 rate boundaries, maximum ctx, identity key mapping, rejection-loop speculation paths."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sets():
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    hp = HotPath(0)
    yield {s: MlDsa(s, hotpath=hp) for s in (44, 65, 87)}
    hp.close()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_message_and_ctx_length_edges_vs_oracle(sets, pset):
    """mu = H(tr | 0 | len(ctx) | ctx | M): lengths that put the pad byte at every position class of
    the 136-byte rate (64 + 2 + |ctx| + |M| = 135, 136, 137, 271, 272, 273 ...), empty message, ctx 255."""
    m = sets[pset]
    rng = np.random.default_rng(900 + pset)
    pk_o, sk_o = orc.keygen_from_seed(pset, bytes(range(32)))
    pks = m.public_keys_from_bytes([orc.pk_into_bytes(pset, pk_o)])
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(pset, sk_o)])
    cases = []
    for total in (66, 67, 135, 136, 137, 138, 271, 272, 273, 407, 408, 409, 1000):
        for clen in (0, 1, 255):
            mlen = total - 66 - clen
            if mlen >= 0:
                cases.append((mlen, clen))
    msgs = [rng.integers(0, 256, a, dtype=np.uint8).tobytes() for a, _ in cases]
    ctxs = [rng.integers(0, 256, b, dtype=np.uint8).tobytes() for _, b in cases]
    rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in cases]
    sig = host(m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs))
    for i in range(len(cases)):
        assert sig[i].tobytes() == orc.sign_internal(pset, sk_o, msgs[i], rnd[i], ctx=ctxs[i], mode=0), cases[i]
    assert m.verify(pks, msgs, torch.from_numpy(sig).cuda(), ctxs=ctxs).all()
    # pre-hash domain separation (mode 2): msg = OID | PH(M), must differ from pure mode
    sig2 = host(m.try_sign_with_seed(sks, msgs[:3], rnd[:3], ctxs=ctxs[:3], mode=2))
    for i in range(3):
        assert sig2[i].tobytes() == orc.sign_internal(pset, sk_o, msgs[i], rnd[i], ctx=ctxs[i], mode=2)
    assert m.verify(pks, msgs[:3], torch.from_numpy(sig2).cuda(), ctxs=ctxs[:3], mode=2).all()
    assert not m.verify(pks, msgs[:3], torch.from_numpy(sig2).cuda(), ctxs=ctxs[:3], mode=0).any()


def test_identity_key_mapping_and_multichunk(sets):
    """key_idx = None with one key per op, and a batch larger than one pipeline pass (131 072 ops for keygen and verify): verdicts must match a known corruption pattern."""
    m = sets[44]
    g = torch.Generator(device="cuda").manual_seed(17)
    n = 131072 + 7001  # two verify / keygen passes with a ragged end (sign: one pass, see test_multichunk_sign)
    xi = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    msgs = [i.to_bytes(8, "little") for i in range(n)]
    rnd = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")  # deterministic variant
    sig = m.try_sign_with_seed(sks, msgs, rnd)                      # key_idx None: op i uses key i
    bad = np.zeros(n, dtype=bool)
    bad[::97] = True
    sig_h = host(sig).copy()
    sig_h[bad, 40] ^= 0x10
    got = m.verify(pks, msgs, torch.from_numpy(sig_h).cuda())
    assert np.array_equal(got, ~bad)
    # spot-check signatures against the oracle on both sides of the chunk boundary
    skh = host(sk)
    for i in (0, 1, 65535, 65536, 131071, 131072, n - 1):
        want = orc.sign_internal(44, orc.sk_try_from_bytes(44, skh[i].tobytes()), msgs[i], bytes(32), mode=0)
        assert host(sig)[i].tobytes() == want, i


@pytest.mark.parametrize("n_ops", [1, 2, 63, 64, 65, 1000, 40000])
def test_sign_speculation_regimes_match_oracle(sets, n_ops):
    """active-set sizes that start in different speculation regimes (spec = 32 ... 1)"""
    m = sets[65]
    rng = np.random.default_rng(n_ops)
    pk_o, sk_o = orc.keygen_from_seed(65, bytes([7] * 32))
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(65, sk_o)])
    pks = m.public_keys_from_bytes([orc.pk_into_bytes(65, pk_o)])
    msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    sig = m.try_sign_with_seed(sks, msgs, rnd)
    sig_h = host(sig)
    for i in sorted(set([0, n_ops // 2, n_ops - 1] + list(range(0, n_ops, max(1, n_ops // 12))))):
        assert sig_h[i].tobytes() == orc.sign_internal(65, sk_o, msgs[i], rnd[i], mode=0), i
    assert m.verify(pks, msgs, sig).all()


@pytest.mark.parametrize("target,spec_max,rounds,lanes", [(65536, 32, 0, 1), (4096, 7, 0, 1), (1, 1, 0, 1), (65536, 64, 0, 1), (65536, 32, 1, 1),
                                                           (512, 3, 4, 1), (65536, 32, 0, 2), (2048, 5, 2, 2)])
def test_sign_schedule_knobs_do_not_change_signatures(sets, target, spec_max, rounds, lanes):
    """The rejection loop's scheduling (candidate slots per round, candidates per op, rounds enqueued before the
    host looks, one or two slices on two streams) is invisible in the output: the first accepted kappa wins,
    exactly as in ml_dsa.rs:212-336."""
    m = sets[44]
    hp = m.hp
    n_ops = 9001
    rng = np.random.default_rng(5)
    pk_o, sk_o = orc.keygen_from_seed(44, bytes([3] * 32))
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(44, sk_o)])
    msgs = [rng.integers(0, 256, 20, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    defaults = {o: hp.get_option(o) for o in (2, 3, 6, 7)}  # MLDSA_OPT_SPEC_TARGET, _SPEC_MAX, _SIGN_ROUNDS, _SIGN_LANES
    assert defaults == {2: 65536, 3: 32, 6: 0, 7: 0}  # (lanes: 0 = two slices from 131 072 ops on, ML-DSA-44 from 65 536)
    base = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=[0] * n_ops)).copy()
    try:
        hp.set_option(2, target)
        hp.set_option(3, spec_max)
        hp.set_option(6, rounds)
        hp.set_option(7, lanes)
        got = host(m.try_sign_with_seed(sks, msgs, rnd, key_idx=[0] * n_ops))
    finally:
        for o, v in defaults.items():
            hp.set_option(o, v)
    assert np.array_equal(got, base)
    for i in (0, 4500, 9000):
        assert base[i].tobytes() == orc.sign_internal(44, sk_o, msgs[i], rnd[i], mode=0), i


def _verify_job(m, pset, seed, n_ops):
    """keys / messages / signatures (from the oracle, a few corrupted) and the expected verdicts"""
    rng = np.random.default_rng(seed)
    pk_o, sk_o = orc.keygen_from_seed(pset, bytes(rng.integers(0, 256, 32, dtype=np.uint8)))
    pks = m.public_keys_from_bytes([orc.pk_into_bytes(pset, pk_o)])
    base_msgs = [rng.integers(0, 256, 24, dtype=np.uint8).tobytes() for _ in range(8)]
    base_sigs = [orc.sign_internal(pset, sk_o, mm, bytes(32), mode=0) for mm in base_msgs]
    msgs = [base_msgs[i % 8] for i in range(n_ops)]
    want = np.ones(n_ops, dtype=bool)
    sigs = []
    for i in range(n_ops):
        sg = bytearray(base_sigs[i % 8])
        if i % 5 == 3:
            sg[100 + i % 64] ^= 1 << (i % 8)
            want[i] = False
        sigs.append(bytes(sg))
    return pks, msgs, sigs, want


def test_one_context_two_streams_and_two_threads(sets):
    """include/mldsa_hip.h threading contract: op-level calls on one context may come from several host
    threads and several streams; the context serialises their use of its workspace."""
    import threading
    jobs = {pset: _verify_job(sets[pset], pset, pset, 3000 + pset) for pset in (44, 65)}
    # (a) two streams, alternating calls, nothing synchronised in between
    st = {44: torch.cuda.Stream(), 65: torch.cuda.Stream()}
    torch.cuda.synchronize()
    # (the convenience wrapper synchronises; the device-level entry does not)
    from fips204_amd.ml_dsa import _cat_with_offsets
    staged = {}
    for pset in (44, 65):
        pks, msgs, sigs, _ = jobs[pset]
        mb, mo = _cat_with_offsets(msgs, "cuda")
        sg = torch.frombuffer(bytearray(b"".join(sigs)), dtype=torch.uint8).cuda().view(len(sigs), -1)
        kidx = torch.zeros(len(sigs), dtype=torch.int32, device="cuda")
        staged[pset] = (pks, mb, mo, sg, kidx)
    torch.cuda.synchronize()
    oks = {44: [], 65: []}
    for rep in range(4):
        for pset in (44, 65):
            pks, mb, mo, sg, kidx = staged[pset]
            ok = torch.zeros(sg.shape[0], dtype=torch.uint8, device="cuda")
            with torch.cuda.stream(st[pset]):
                sets[pset].verify_device(pks, mb, mo, sg, ok, sg.shape[0], key_idx=kidx)
            oks[pset].append(ok)
    torch.cuda.synchronize()
    for pset in (44, 65):
        for ok in oks[pset]:
            assert np.array_equal(ok.cpu().numpy().astype(bool), jobs[pset][3]), pset

    # (b) two host threads, each on its own stream, hammering the same context (ctypes drops the GIL)
    errors = []

    def worker(pset):
        try:
            pks, mb, mo, sg, kidx = staged[pset]
            s = torch.cuda.Stream()
            for _ in range(6):
                ok = torch.zeros(sg.shape[0], dtype=torch.uint8, device="cuda")
                with torch.cuda.stream(s):
                    sets[pset].verify_device(pks, mb, mo, sg, ok, sg.shape[0], key_idx=kidx)
                s.synchronize()
                if not np.array_equal(ok.cpu().numpy().astype(bool), jobs[pset][3]):
                    errors.append(pset)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(p,)) for p in (44, 65)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_cached_a_hat_entry_points_are_bit_identical(sets, pset):
    """mldsa_verify_cached_a / mldsa_sign_cached_a (A_hat kept with the keys, benches/README.md:4-8) give
    exactly the results of the per-op ExpandA path, with and without a key index."""
    from fips204_amd.ml_dsa import _cat_with_offsets
    m = sets[pset]
    g = torch.Generator(device="cuda").manual_seed(pset)
    n_keys, n_ops = 5, 700
    xi = torch.randint(0, 256, (n_keys, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    a_pk, a_sk = m.expand_a_for_keys(pks), m.expand_a_for_keys(sks)
    assert torch.equal(a_pk, a_sk)
    msgs = [bytes([i & 255, i >> 8]) * 9 for i in range(n_ops)]
    mb, mo = _cat_with_offsets(msgs, "cuda")
    rnd = torch.randint(0, 256, (n_ops, 32), dtype=torch.uint8, device="cuda", generator=g)
    kidx = (torch.arange(n_ops, device="cuda") % n_keys).to(torch.int32)
    sig0 = torch.empty((n_ops, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    sig1 = torch.empty_like(sig0)
    m.sign_device(sks, mb, mo, rnd, sig0, n_ops, key_idx=kidx)
    m.sign_device(sks, mb, mo, rnd, sig1, n_ops, key_idx=kidx, a_hat=a_sk)
    assert torch.equal(sig0, sig1)
    bad = sig0.clone()
    bad[::7, 50] ^= 4
    want = np.ones(n_ops, dtype=bool)
    want[::7] = False
    for a in (None, a_pk):
        ok = torch.zeros(n_ops, dtype=torch.uint8, device="cuda")
        m.verify_device(pks, mb, mo, bad, ok, n_ops, key_idx=kidx, a_hat=a)
        assert np.array_equal(host(ok).astype(bool), want)
    # identity mapping (key_idx = NULL): op i uses key i and A_hat row i
    sig2 = torch.empty((n_keys, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    sig3 = torch.empty_like(sig2)
    mb5, mo5 = _cat_with_offsets(msgs[:n_keys], "cuda")
    m.sign_device(sks, mb5, mo5, rnd[:n_keys].contiguous(), sig2, n_keys)
    m.sign_device(sks, mb5, mo5, rnd[:n_keys].contiguous(), sig3, n_keys, a_hat=a_sk)
    assert torch.equal(sig2, sig3)
    ok = torch.zeros(n_keys, dtype=torch.uint8, device="cuda")
    m.verify_device(pks, mb5, mo5, sig3, ok, n_keys, a_hat=a_pk)
    assert host(ok).all()
    skh = host(sk)
    for i in (0, 4):
        want_sig = orc.sign_internal(pset, orc.sk_try_from_bytes(pset, skh[i].tobytes()), msgs[i], host(rnd)[i].tobytes(), mode=0)
        assert host(sig3)[i].tobytes() == want_sig


@pytest.mark.parametrize("pset,n_ops", [(44, 6000), (65, 4000), (87, 3000)])
def test_every_signature_of_a_batch_matches_the_oracle(sets, pset, n_ops):
    """Not a sample: all signatures of a few-thousand-op batch (several keys, hedged rnd) are compared
    byte for byte with the oracle, so a wrong accept / reject of any single candidate -- e.g. in the
    margin shortcut of k_sign_tail -- would show as a different kappa."""
    import os
    from fips204_amd.ml_dsa import _cat_with_offsets
    m = sets[pset]
    rng = np.random.default_rng(1000 + pset)
    n_keys = 7
    keys = [orc.keygen_from_seed(pset, bytes(rng.integers(0, 256, 32, dtype=np.uint8))) for _ in range(n_keys)]
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(pset, k[1]) for k in keys])
    msgs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    rnds = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_ops)]
    kidx_h = (np.arange(n_ops) * 5 % n_keys).astype(np.uint32)
    mb, mo = _cat_with_offsets(msgs, "cuda")
    rnd = torch.frombuffer(bytearray(b"".join(rnds)), dtype=torch.uint8).cuda().view(n_ops, 32)
    kidx = torch.from_numpy(kidx_h.view(np.int32)).cuda()
    sig = torch.empty((n_ops, m.SIG_LEN), dtype=torch.uint8, device="cuda")
    m.sign_device(sks, mb, mo, rnd, sig, n_ops, key_idx=kidx)
    got = host(sig)
    want = orc.sign_batch_mt(pset, [k[1] for k in keys], kidx_h, msgs, rnds, n_threads=min(16, os.cpu_count() or 1))
    bad = [i for i in range(n_ops) if got[i].tobytes() != want[i]]
    assert not bad, (len(bad), bad[:5])


def test_mixed_parameter_set_stream(sets):
    """BASELINE config 5 as a parity case: ML-DSA-44 / 65 / 87 keygen -> sign -> verify issued back to back
    on one context with no synchronisation in between; all signatures verify and match the oracle."""
    from fips204_amd.ml_dsa import _cat_with_offsets
    n_ops, n_keys = 600, 6
    runs = []
    for rep in range(2):
        for pset in (87, 44, 65):
            m = sets[pset]
            g = torch.Generator(device="cuda").manual_seed(100 * rep + pset)
            xi = torch.randint(0, 256, (n_keys, 32), dtype=torch.uint8, device="cuda", generator=g)
            msgs = [bytes([pset, rep, i & 255, i >> 8]) * 5 for i in range(n_ops)]
            mb, mo = _cat_with_offsets(msgs, "cuda")
            rnd = torch.randint(0, 256, (n_ops, 32), dtype=torch.uint8, device="cuda", generator=g)
            kidx = (torch.arange(n_ops, device="cuda") % n_keys).to(torch.int32)
            pk, sk = m.keygen_from_seed(xi)
            pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
            sig = torch.empty((n_ops, m.SIG_LEN), dtype=torch.uint8, device="cuda")
            ok = torch.zeros(n_ops, dtype=torch.uint8, device="cuda")
            m.sign_device(sks, mb, mo, rnd, sig, n_ops, key_idx=kidx)
            m.verify_device(pks, mb, mo, sig, ok, n_ops, key_idx=kidx)
            runs.append((pset, xi, msgs, rnd, sk, sig, ok))
    torch.cuda.synchronize()
    for pset, xi, msgs, rnd, sk, sig, ok in runs:
        assert host(ok).all(), pset
        skh, sgh, rh, xh = host(sk), host(sig), host(rnd), host(xi)
        for i in (0, 7, n_ops - 1):
            ki = i % n_keys
            pk_o, sk_o = orc.keygen_from_seed(pset, xh[ki].tobytes())
            assert skh[ki].tobytes() == orc.sk_into_bytes(pset, sk_o)
            assert sgh[i].tobytes() == orc.sign_internal(pset, sk_o, msgs[i], rh[i].tobytes(), mode=0), (pset, i)


def test_hint_weight_and_z_bound_rejections(sets, acvp_sigver):
    """the ACVP 'too many hints' / 'z too large' signatures stay rejected inside a large mixed batch"""
    g = [x for x in acvp_sigver["testGroups"] if x["parameterSet"] == "ML-DSA-87"][0]
    m = sets[87]
    pks = m.public_keys_from_bytes([bytes.fromhex(g["pk"])])
    tests = g["tests"] * 300  # 4500 ops
    msgs = [bytes.fromhex(t["message"]) for t in tests]
    sigs = [bytes.fromhex(t["signature"]) for t in tests]
    got = m.verify(pks, msgs, sigs, mode=1)
    assert got.tolist() == [t["testPassed"] for t in tests]


def test_very_long_and_empty_messages_in_one_batch(sets):
    """mu = H(tr || M') absorbs messages of any length (ml_dsa.rs:185-196): a 300 000-byte message, an empty one and a
    135 / 136 / 137-byte boundary group share one batch with ragged offsets; signatures byte-identical to the oracle, in all
    three message modes, and the verifier accepts them only under the mode they were made in."""
    m = sets[65]
    rng = np.random.default_rng(77)
    pk_o, sk_o = orc.keygen_from_seed(65, bytes(range(9, 41)))
    pks = m.public_keys_from_bytes([orc.pk_into_bytes(65, pk_o)])
    sks = m.private_keys_from_bytes([orc.sk_into_bytes(65, sk_o)])
    lens = [300000, 0, 135, 136, 137, 1, 70001]
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    ctxs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 255, 1, 0, 17, 3, 200)]
    rnd = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in lens]
    for mode in (0, 1, 2):
        sig = host(m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, mode=mode))
        for i in range(len(lens)):
            assert sig[i].tobytes() == orc.sign_internal(65, sk_o, msgs[i], rnd[i], ctx=ctxs[i], mode=mode), (mode, lens[i])
        st = torch.from_numpy(sig).cuda()
        assert m.verify(pks, msgs, st, ctxs=ctxs, mode=mode).all()
        assert not m.verify(pks, msgs, st, ctxs=ctxs, mode=(mode + 1) % 3).any()
