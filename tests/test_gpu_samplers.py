"""GPU parity: SHAKE-driven samplers through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SETS = {44: (4, 4, 2, 1 << 17, 39, 32), 65: (6, 5, 4, 1 << 19, 49, 48), 87: (8, 7, 2, 1 << 19, 60, 64)}


@pytest.fixture(scope="module")
def hp():
    from fips204_amd.hotpath import HotPath
    h = HotPath(0)
    yield h
    h.close()


def devb(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("pset", [44, 65, 87])
@pytest.mark.parametrize("n_ops", [1, 3, 70])
def test_expand_a(hp, pset, n_ops):
    k, l = SETS[pset][:2]
    rng = np.random.default_rng(1000 + pset + n_ops)
    rho = rng.integers(0, 256, (n_ops, 32), dtype=np.uint8)
    got = host(hp.expand_a(pset, devb(rho)))
    assert got.min() >= 0 and got.max() < orc.Q
    for i in range(n_ops):
        assert np.array_equal(got[i], orc.expand_a(k, l, rho[i].tobytes())), i


def test_expand_a_rejection_paths(hp):
    # ~22 % of polys see at least one rejected candidate; check a large sample so that
    # streams with 0, 1, 2, 3 rejections (768..777 bytes consumed) all occur
    rng = np.random.default_rng(77)
    rho = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    got = host(hp.expand_a(87, devb(rho)))
    for i in range(0, 300, 7):
        assert np.array_equal(got[i], orc.expand_a(8, 7, rho[i].tobytes())), i


def test_expand_a_kat_seed(hp, acvp_keygen):
    # rho of the first ACVP keyGen case of each set (pk[0:32])
    for g in acvp_keygen["testGroups"]:
        pset = {"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87}[g["parameterSet"]]
        k, l = SETS[pset][:2]
        rho = bytes.fromhex(g["tests"][0]["pk"])[:32]
        got = host(hp.expand_a(pset, devb(np.frombuffer(rho, dtype=np.uint8))))[0]
        assert np.array_equal(got, orc.expand_a(k, l, rho))


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_expand_s(hp, pset):
    k, l, eta = SETS[pset][:3]
    rng = np.random.default_rng(2000 + pset)
    n_ops = 23
    rho = rng.integers(0, 256, (n_ops, 64), dtype=np.uint8)
    s1, s2 = hp.expand_s(pset, devb(rho))
    s1, s2 = host(s1), host(s2)
    assert s1.min() >= -eta and s1.max() <= eta and s2.min() >= -eta and s2.max() <= eta
    for i in range(n_ops):
        w1, w2 = orc.expand_s(k, l, eta, rho[i].tobytes())
        assert np.array_equal(s1[i], w1) and np.array_equal(s2[i], w2), i


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_expand_mask(hp, pset):
    k, l, eta, gamma1 = SETS[pset][:4]
    rng = np.random.default_rng(3000 + pset)
    n_ops = 41
    rho = rng.integers(0, 256, (n_ops, 64), dtype=np.uint8)
    kappa = rng.integers(0, 3000, n_ops).astype(np.uint16) * l
    kappa[0] = 0
    kappa[1] = 65535 - l + 1  # top of the u16 range without wrapping
    kd = torch.from_numpy(kappa.view(np.int16)).cuda()
    got = host(hp.expand_mask(pset, devb(rho), kd))
    assert got.min() >= -gamma1 + 1 and got.max() <= gamma1
    for i in range(n_ops):
        assert np.array_equal(got[i], orc.expand_mask(l, gamma1, rho[i].tobytes(), int(kappa[i]))), i


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_sample_in_ball(hp, pset):
    tau, ct = SETS[pset][4:6]
    rng = np.random.default_rng(4000 + pset)
    n_ops = 200
    seeds = rng.integers(0, 256, (n_ops, ct), dtype=np.uint8)
    got = host(hp.sample_in_ball(pset, devb(seeds)))
    for i in range(n_ops):
        want = orc.sample_in_ball(tau, seeds[i].tobytes())
        assert np.array_equal(got[i], want), i
        assert np.count_nonzero(got[i]) == tau  # hashing.rs:89-96


def test_samplers_large_batch_properties(hp):
    # full-size property checks: determinism + per-op independence of batch position
    g = torch.Generator(device="cuda").manual_seed(11)
    rho = torch.randint(0, 256, (4096, 32), dtype=torch.uint8, device="cuda", generator=g)
    a = hp.expand_a(65, rho)
    assert int(a.min()) >= 0 and int(a.max()) < orc.Q
    perm = torch.randperm(4096, device="cuda", generator=g)
    b = hp.expand_a(65, rho[perm].contiguous())
    assert torch.equal(a[perm], b)
