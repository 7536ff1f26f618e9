"""Keys on the device: SerDes (try_from_bytes / into_bytes, src/lib.rs:421-493, src/ml_dsa.rs:445-498), get_public_key (lib.rs:345-349), arbitrary and
out-of-range key bytes signing like the reference, key-index tables, mldsa_verify_pk.  (Tests re-filed by component in round 5 from
test_gpu_round2/3/4.py: same tests, none lost.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_verify_pk_equals_try_from_bytes_plus_verify(sets, pset):
    """mldsa_verify_pk = PublicKey::try_from_bytes (src/ml_dsa.rs:477-498) + Verifier::verify (351-437) in one call.  On a fuzzed batch
    (good and damaged signatures, arbitrary public-key bytes: fuzz_all.rs:25-37, fuzz_verify.rs:17-31) its verdicts equal those of
    mldsa_pk_expand + mldsa_verify AND the oracle's -- with a key table + key_idx, with one key per op (identity mapping), and
    when the call runs in several passes (a capped workspace)."""
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
