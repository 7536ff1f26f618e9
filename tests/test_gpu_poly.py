"""GPU parity: polynomial-arithmetic kernels through the C ABI vs the CPU oracle.
Bit-exact mod q (integer work); canonical where the reference's output is canonical."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

Q = orc.Q
SETS = {44: (4, 4, 2, 1 << 17), 65: (6, 5, 4, 1 << 19), 87: (8, 7, 2, 1 << 19)}


@pytest.fixture(scope="module")
def hp():
    from fips204_amd.hotpath import HotPath
    h = HotPath(0)
    yield h
    h.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def ntt_inputs(rng):
    cases = {
        "uniform": rng.integers(0, Q, (37, 256)),
        "zero": np.zeros((1, 256)),
        "q_minus_1": np.full((1, 256), Q - 1),
        "eta": rng.integers(-4, 5, (3, 256)),
        "gamma1": rng.integers(-(1 << 19) + 1, (1 << 19) + 1, (5, 256)),
        "gamma1_edges": np.where(rng.integers(0, 2, (2, 256)) == 1, 1 << 19, -(1 << 19) + 1),
        "t0": rng.integers(-(1 << 12) + 1, (1 << 12) + 1, (2, 256)),
        "signed_lazy": rng.integers(-9 * Q, 9 * Q, (4, 256)),
    }
    c = np.zeros((2, 256), dtype=np.int64)
    for r in range(2):
        pos = rng.choice(256, 49, replace=False)
        c[r, pos] = rng.choice([-1, 1], 49)
    cases["challenge"] = c
    return cases


def test_ntt_matches_oracle(hp):
    rng = np.random.default_rng(41)
    for name, w in ntt_inputs(rng).items():
        got = host(hp.ntt(dev(w))).astype(np.int64) % Q
        want = orc.ntt(w.astype(np.int32)).astype(np.int64) % Q
        assert np.array_equal(got, want), name


def test_inv_ntt_matches_oracle_canonical(hp):
    rng = np.random.default_rng(42)
    for name, w in ntt_inputs(rng).items():
        got = host(hp.inv_ntt(dev(w)))
        want = orc.inv_ntt(w.astype(np.int32))
        assert got.min() >= 0 and got.max() < Q, name  # full_reduce32, ntt.rs:153
        assert np.array_equal(got, want), name


def test_ntt_roundtrip_and_linearity_large(hp):
    # size-independent properties at a BASELINE-sized batch (4096 ops x 5 polys)
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randint(0, Q, (20480, 256), dtype=torch.int32, device="cuda", generator=g)
    b = torch.randint(0, Q, (20480, 256), dtype=torch.int32, device="cuda", generator=g)
    assert torch.equal(hp.inv_ntt(hp.ntt(a)), a)
    lhs = hp.ntt(a + b).to(torch.int64) % Q
    rhs = (hp.ntt(a).to(torch.int64) + hp.ntt(b).to(torch.int64)) % Q
    assert torch.equal(lhs, rhs)
    # in-place aliasing is allowed by the ABI
    c = a.clone()
    hp.ntt(c, out=c)
    assert torch.equal(c.to(torch.int64) % Q, hp.ntt(a).to(torch.int64) % Q)


def test_ntt_empty_and_ragged(hp):
    assert hp.ntt(torch.empty((0, 256), dtype=torch.int32, device="cuda")).numel() == 0
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 5, 63, 257):
        w = rng.integers(0, Q, (n, 256))
        assert np.array_equal(host(hp.ntt(dev(w))).astype(np.int64) % Q,
                              orc.ntt(w.astype(np.int32)).astype(np.int64) % Q)
    with pytest.raises(ValueError):
        hp.ntt(torch.zeros(100, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_mat_vec_mul_matches_oracle(hp, pset):
    k, l, _, _ = SETS[pset]
    rng = np.random.default_rng(100 + pset)
    n_ops = 9
    a = rng.integers(0, Q, (n_ops, k, l, 256))
    u = rng.integers(-8 * Q, 8 * Q, (n_ops, l, 256))  # ntt output range, SURVEY appendix B
    got = host(hp.mat_vec_mul(pset, dev(a), dev(u))).astype(np.int64) % Q
    for i in range(n_ops):
        want = orc.mat_vec_mul(k, l, a[i].astype(np.int32), u[i].astype(np.int32)).astype(np.int64) % Q
        assert np.array_equal(got[i], want)


def test_pointwise_to_mont_add_norm(hp):
    rng = np.random.default_rng(9)
    c = rng.integers(-8 * Q, 8 * Q, (6, 256))
    v = rng.integers(-2 * Q + 1, 2 * Q, (6, 5, 256))
    got = host(hp.pointwise_mont(dev(c), dev(v))).astype(np.int64) % Q
    for i in range(6):
        want = orc.pointwise_mont(c[i].astype(np.int32), v[i].astype(np.int32)).astype(np.int64) % Q
        assert np.array_equal(got[i], want)
    x = rng.integers(-67_058_538, 67_058_539, (7, 256))  # partial_reduce64 contract, helpers.rs:35
    assert np.array_equal(host(hp.to_mont(dev(x))).astype(np.int64) % Q, orc.to_mont(x.astype(np.int32)).astype(np.int64) % Q)
    y = rng.integers(-(1 << 29), 1 << 29, (7, 256))
    assert np.array_equal(host(hp.add_vector_ntt(dev(x), dev(y))), (x + y).astype(np.int32))
    w = rng.integers(-(1 << 30), 1 << 30, (11, 4, 256))
    w[3] = 0
    w[4, 2, 17] = Q // 2 + 1  # centre boundary
    got = host(hp.infinity_norm(dev(w), 4))
    want = [orc.infinity_norm(w[i].astype(np.int32)) for i in range(11)]
    assert got.tolist() == want


@pytest.mark.parametrize("pset,n_ops", [(44, 4096), (65, 257), (87, 131)])
def test_verify_arith_matches_oracle(hp, pset, n_ops):
    # BASELINE config 2 is (44, 4096); parity on ALL ops against the oracle
    k, l, _, gamma1 = SETS[pset]
    tau = {44: 39, 65: 49, 87: 60}[pset]
    rng = np.random.default_rng(200 + pset)
    a = rng.integers(0, Q, (n_ops, k, l, 256), dtype=np.int32)
    z = rng.integers(-gamma1 + 1, gamma1 + 1, (n_ops, l, 256), dtype=np.int32)
    c = np.zeros((n_ops, 256), dtype=np.int32)
    for i in range(n_ops):
        pos = rng.choice(256, tau, replace=False)
        c[i, pos] = rng.choice([-1, 1], tau)
    t1 = rng.integers(-2 * Q + 1, 2 * Q, (n_ops, k, 256), dtype=np.int32)  # partial_reduce64 range
    got = host(hp.verify_arith(pset, dev(a), dev(z), dev(c), dev(t1)))
    want = orc.verify_arith(k, l, a, z, c, t1)
    assert got.min() >= 0 and got.max() < Q
    assert np.array_equal(got, want)


def test_verify_arith_composes_from_seam_primitives(hp):
    # the fused kernel == the reference's own sequence of seam calls (ml_dsa.rs:407-416)
    k, l = 6, 5
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 512
    a = torch.randint(0, Q, (n, k, l, 256), dtype=torch.int32, device="cuda", generator=g)
    z = torch.randint(-(1 << 19) + 1, (1 << 19) + 1, (n, l, 256), dtype=torch.int32, device="cuda", generator=g)
    c = torch.randint(-1, 2, (n, 256), dtype=torch.int32, device="cuda", generator=g)
    t1 = torch.randint(0, Q, (n, k, 256), dtype=torch.int32, device="cuda", generator=g)
    az = hp.mat_vec_mul(65, a, hp.ntt(z))
    ct = hp.pointwise_mont(hp.ntt(c), t1)
    ref = hp.inv_ntt(az - ct.view(n, k, 256))
    assert torch.equal(hp.verify_arith(65, a, z, c, t1), ref)
