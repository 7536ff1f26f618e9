"""GPU parity for whole verify (verify_internal, src/ml_dsa.rs:351-437) through the C ABI.
Mirrors tests/nist_vectors/mod.rs::test_sigver, tests/integration.rs::bad_sig and the
bit-flip checks of test_44_no_verif."""
import numpy as np
import pytest
import torch

from conftest import PSET
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sets():
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    hp = HotPath(0)
    yield {s: MlDsa(s, hotpath=hp) for s in (44, 65, 87)}
    hp.close()


def upload_oracle_pks(m, pk_bytes_list):
    """expanded keys computed by the oracle (isolates verify from pk_expand)"""
    from fips204_amd.ml_dsa import PublicKeys
    k = m.params.k
    rho, tr, t1 = [], [], []
    for b in pk_bytes_list:
        pk = orc.pk_try_from_bytes(m.pset, b)
        rho.append(np.frombuffer(bytes(pk.rho), dtype=np.uint8))
        tr.append(np.frombuffer(bytes(pk.tr), dtype=np.uint8))
        t1.append(np.ctypeslib.as_array(pk.t1_d2_hat_mont)[:k].copy())
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.stack(a), dtype=dt)).cuda()
    return PublicKeys(m.pset, d(rho, np.uint8), d(tr, np.uint8), d(t1, np.int32))


def test_acvp_sigver(sets, acvp_sigver):  # nist_vectors/mod.rs:148-203
    n = 0
    for g in acvp_sigver["testGroups"]:
        m = sets[PSET[g["parameterSet"]]]
        pks = upload_oracle_pks(m, [bytes.fromhex(g["pk"])])
        msgs = [bytes.fromhex(t["message"]) for t in g["tests"]]
        sigs = [bytes.fromhex(t["signature"]) for t in g["tests"]]
        got = m.verify(pks, msgs, sigs, mode=1)
        want = [t["testPassed"] for t in g["tests"]]
        assert got.tolist() == want, [(t["tcId"], t["reason"]) for t, a, b in zip(g["tests"], got, want) if a != b]
        n += len(want)
    assert n == 45


def test_bad_sig(sets, ref_hex):  # tests/integration.rs:63-74
    v = ref_hex["integration_bad_sig"]
    m = sets[44]
    pks = upload_oracle_pks(m, [bytes.fromhex(v["pk"])])
    msg = bytes.fromhex(v["msg"])
    got = m.verify(pks, [msg, msg], [bytes.fromhex(v["good_sig"]), bytes.fromhex(v["bad_sig"])], mode=1)
    assert got.tolist() == [True, False]


def test_messages_rs_external_interface(sets, ref_hex):  # tests/messages.rs:10-21 (ctx-prefixed path)
    v = ref_hex["messages_rs"]
    m = sets[44]
    pks = upload_oracle_pks(m, [bytes.fromhex(v["pk"])])
    sig = bytes.fromhex(v["sig"])
    assert m.verify(pks, [b"asdf"], [sig], ctxs=[b""], mode=0).tolist() == [True]
    assert m.verify(pks, [b"asdf"], [sig], ctxs=None, mode=0).tolist() == [True]
    assert m.verify(pks, [b"asdf"], [sig], ctxs=[b"\x00"], mode=0).tolist() == [False]
    assert m.verify(pks, [b"asdg"], [sig], mode=0).tolist() == [False]
    assert m.verify(pks, [b"asdf"], [sig], mode=1).tolist() == [False]  # internal interface: no prefix
    assert m.verify(pks, [b"asdf"], [sig], ctxs=[bytes(256)], mode=0).tolist() == [False]  # lib.rs:368


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_sign_verify_batch_with_corruptions(sets, pset):
    """oracle-signed batch, many keys, ragged messages and contexts; 1 in 4 corrupted in the
    ways tests/integration.rs:79-119 corrupts (message / signature bit flips)"""
    m = sets[pset]
    rng = np.random.default_rng(500 + pset)
    n_keys, n_ops = 5, 64
    keys = [orc.keygen_from_seed(pset, rng.integers(0, 256, 32, dtype=np.uint8).tobytes()) for _ in range(n_keys)]
    pks = upload_oracle_pks(m, [orc.pk_into_bytes(pset, pk) for pk, _ in keys])
    msgs, ctxs, sigs, want, kidx = [], [], [], [], []
    for i in range(n_ops):
        ki = int(rng.integers(0, n_keys))
        mlen = int(rng.choice([0, 1, 31, 32, 70, 71, 72, 200, 207, 208, 500]))
        msg = rng.integers(0, 256, mlen, dtype=np.uint8).tobytes()
        ctx = rng.integers(0, 256, int(rng.choice([0, 0, 1, 17, 255])), dtype=np.uint8).tobytes()
        sig = orc.sign_internal(pset, keys[ki][1], msg, rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), ctx=ctx, mode=0)
        good = True
        if i % 4 == 1:
            b = bytearray(sig); b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8)); sig = bytes(b)
        elif i % 4 == 3 and mlen > 0:
            b = bytearray(msg); b[int(rng.integers(0, mlen))] ^= 0x08; msg = bytes(b)
        good = orc.verify_internal(pset, keys[ki][0], msg, sig, ctx=ctx, mode=0)
        msgs.append(msg); ctxs.append(ctx); sigs.append(sig); want.append(good); kidx.append(ki)
    got = m.verify(pks, msgs, sigs, ctxs=ctxs, key_idx=kidx, mode=0)
    assert got.tolist() == want
    assert 20 <= sum(want) <= 60


def test_wrong_length_signature_and_empty_batch(sets):
    m = sets[44]
    pk, sk = orc.keygen_from_seed(44, bytes(32))
    pks = upload_oracle_pks(m, [orc.pk_into_bytes(44, pk)])
    sig = orc.sign_internal(44, sk, b"m", bytes(32), mode=0)
    assert m.verify(pks, [b"m", b"m"], [sig, sig[:-1]], mode=0).tolist() == [True, False]
    assert m.verify(pks, [], [], mode=0).tolist() == []
