"""Seam-level entry points added in round 4: scalar reductions (helpers.rs:33-98), the rounding functions (high_low.rs:66-192), and the
cooperative against the lane-per-state sponges.  (Re-filed by component in round 5.)"""
from gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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
