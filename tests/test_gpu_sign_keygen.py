"""GPU parity for key expansion, keygen and sign through the C ABI.  Mirrors
tests/nist_vectors/mod.rs::test_keygen / test_siggen, tests/messages.rs and the
keygen -> sign -> verify smoke tests of src/lib.rs:497-552."""
import numpy as np
import pytest
import torch

from chacha8rng import ChaCha8Rng
from conftest import PSET
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
Q = orc.Q


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


def test_acvp_keygen(sets, acvp_keygen):  # nist_vectors/mod.rs:56-92
    n = 0
    for g in acvp_keygen["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        seeds = [bytes.fromhex(t["seed"]) for t in g["tests"]]
        pk, sk = m.keygen_from_seed(seeds)
        pk, sk = host(pk), host(sk)
        for i, t in enumerate(g["tests"]):
            assert pk[i].tobytes() == bytes.fromhex(t["pk"]), t["tcId"]
            assert sk[i].tobytes() == bytes.fromhex(t["sk"]), t["tcId"]
            n += 1
    assert n == 75


def test_pk0_pins(sets):  # lib.rs:541-545
    for pset, want in ((44, 197), (65, 177), (87, 16)):
        pk, _ = sets[pset].keygen_from_seed([bytes([0x11] * 32)])
        assert int(host(pk)[0, 0]) == want


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_key_expansion_matches_reference_structs(sets, pset):
    m = sets[pset]
    k, l = m.params.k, m.params.l
    rng = np.random.default_rng(600 + pset)
    keys = [orc.keygen_from_seed(pset, rng.integers(0, 256, 32, dtype=np.uint8).tobytes()) for _ in range(7)]
    pkb = [orc.pk_into_bytes(pset, pk) for pk, _ in keys]
    skb = [orc.sk_into_bytes(pset, sk) for _, sk in keys]
    pks, sks = m.public_keys_from_bytes(pkb), m.private_keys_from_bytes(skb)
    for i, (pk, sk) in enumerate(keys):
        assert host(pks.rho)[i].tobytes() == bytes(pk.rho) and host(pks.tr)[i].tobytes() == bytes(pk.tr)
        want = np.ctypeslib.as_array(pk.t1_d2_hat_mont)[:k].astype(np.int64) % Q
        assert np.array_equal(host(pks.t1_d2_hat_mont)[i].astype(np.int64) % Q, want)
        assert host(sks.rho)[i].tobytes() == bytes(sk.rho) and host(sks.cap_k)[i].tobytes() == bytes(sk.cap_k)
        assert host(sks.tr)[i].tobytes() == bytes(sk.tr)
        for got, ref, cnt in ((sks.s_1_hat_mont, sk.s_1_hat_mont, l), (sks.s_2_hat_mont, sk.s_2_hat_mont, k),
                              (sks.t_0_hat_mont, sk.t_0_hat_mont, k)):
            assert np.array_equal(host(got)[i].astype(np.int64) % Q, np.ctypeslib.as_array(ref)[:cnt].astype(np.int64) % Q)
    with pytest.raises(ValueError):
        m.public_keys_from_bytes([pkb[0][:-1]])


def test_acvp_siggen(sets, acvp_siggen):  # nist_vectors/mod.rs:94-146 (internal interface, nist = true)
    n = 0
    for g in acvp_siggen["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        sks = m.private_keys_from_bytes([bytes.fromhex(t["sk"]) for t in g["tests"]])
        msgs = [bytes.fromhex(t["message"]) for t in g["tests"]]
        rnd = [bytes.fromhex(t["rnd"]) if "rnd" in t else bytes(32) for t in g["tests"]]
        sigs = host(m.try_sign_with_seed(sks, msgs, rnd, mode=1))
        for i, t in enumerate(g["tests"]):
            assert sigs[i].tobytes() == bytes.fromhex(t["signature"]), t["tcId"]
            n += 1
    assert n == 60


def test_messages_rs(sets, ref_hex):  # tests/messages.rs:10-21: keygen + sign("asdf", ctx = []) + verify, external API
    v = ref_hex["messages_rs"]
    m = sets[44]
    rng = ChaCha8Rng(123)
    pk, sk = m.try_keygen_with_rng(rng)
    assert host(sk)[0].tobytes() == bytes.fromhex(v["sk"])
    assert host(pk)[0].tobytes() == bytes.fromhex(v["pk"])
    sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
    sig = m.try_sign_with_seed(sks, [b"asdf"], [rng.fill_bytes(32)], ctxs=[b""])
    assert host(sig)[0].tobytes() == bytes.fromhex(v["sig"])
    assert m.verify(pks, [b"asdf"], sig, ctxs=[b""]).tolist() == [True]


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_smoke_keygen_sign_verify(sets, pset):  # lib.rs:497-552 smoke_test, batched
    m = sets[pset]
    rng = ChaCha8Rng(123)
    n = 32
    pk, sk = m.try_keygen_with_rng(rng, n)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    m1, m2 = bytes(range(8)), bytes([7] * 8)
    rnd = [rng.fill_bytes(32) for _ in range(n)]
    sig = m.try_sign_with_seed(sks, [m1] * n, rnd)
    assert m.verify(pks, [m1] * n, sig).all()
    assert not m.verify(pks, [m2] * n, sig).any()
    # bit-exact against the oracle's signer on the same keys / randomness
    skh = host(sk)
    for i in range(0, n, 5):
        want = orc.sign_internal(pset, orc.sk_try_from_bytes(pset, skh[i].tobytes()), m1, rnd[i], mode=0)
        assert host(sig)[i].tobytes() == want
    # ctx too long: verify -> False, sign -> error (lib.rs:527-528)
    assert not m.verify(pks, [m1], sig[:1], ctxs=[bytes(257)]).any()
    with pytest.raises(ValueError):
        m.try_sign_with_seed(sks, [m1], rnd[:1], ctxs=[bytes(257)])


def test_sign_large_batch_roundtrip(sets):
    # many ops on few keys: exercises the rejection-loop re-batching down to an empty active
    # list; every signature must verify, and hedged signatures of one message differ
    m = sets[65]
    g = torch.Generator(device="cuda").manual_seed(5)
    n_keys, n_ops = 8, 2048
    xi = torch.randint(0, 256, (n_keys, 32), dtype=torch.uint8, device="cuda", generator=g)
    pk, sk = m.keygen_from_seed(xi)
    pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
    rnd = torch.randint(0, 256, (n_ops, 32), dtype=torch.uint8, device="cuda", generator=g)
    msgs = [i.to_bytes(4, "little") * 8 for i in range(n_ops)]
    sig = m.try_sign_with_seed(sks, msgs, rnd)
    assert m.verify(pks, msgs, sig).all()
    same = m.try_sign_with_seed(sks, [msgs[0]] * 16, rnd[:16], key_idx=[0] * 16)
    assert len({host(same)[i].tobytes() for i in range(16)}) == 16
    again = m.try_sign_with_seed(sks, msgs, rnd)
    assert torch.equal(sig, again)  # deterministic for fixed rnd, independent of compaction order
