"""Fixtures and helpers shared by the component-wise GPU test files (test_gpu_keys / _boundary / _group / _host_api / _sign_schedule / _seams /
_baseline_configs): one context per test module, the three parameter sets on it, byte-string derivation, device upload helpers, the
offset-table corruptions and the fuzz-derived batches.  Import with `from gpu_common import *`."""
import ctypes as C  # noqa: F401
import hashlib
import os  # noqa: F401
import threading  # noqa: F401
import time  # noqa: F401

import numpy as np
import pytest
import torch

from conftest import PSET  # noqa: F401
from oracle import oracle as orc

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

def make_batch(m, n_ops, n_keys, tag):
    """n_keys key pairs, n_ops 32-byte messages + rnd, keys dealt round-robin: device-resident inputs"""
    from fips204_amd.ml_dsa import _cat_with_offsets
    xi = [shake(tag + b"key", i) for i in range(n_keys)]
    pk, sk = m.keygen_from_seed(xi)
    msgs = [shake(tag + b"msg", i) for i in range(n_ops)]
    rnd = [shake(tag + b"rnd", i) for i in range(n_ops)]
    mb, mo = _cat_with_offsets(msgs, m.device)
    rn = torch.frombuffer(bytearray(b"".join(rnd)), dtype=torch.uint8).cuda().view(n_ops, 32)
    kidx_host = (np.arange(n_ops) % n_keys).astype(np.uint32)
    kidx = torch.from_numpy(kidx_host.view(np.int32)).cuda()
    return dict(xi=xi, pk=pk, sk=sk, pks=m.public_keys_from_bytes(pk), sks=m.private_keys_from_bytes(sk), msgs=msgs, rnd=rnd,
                mb=mb, mo=mo, rn=rn, kidx=kidx, kidx_host=kidx_host, n=n_ops)

def oracle_sigs(pset, b, idx):
    skb = host(b["sk"])
    sks = {}
    out = []
    for i in idx:
        ki = int(b["kidx_host"][i])
        if ki not in sks:
            sks[ki] = orc.sk_try_from_bytes(pset, skb[ki].tobytes())
        out.append(orc.sign_internal(pset, sks[ki], b["msgs"][i], b["rnd"][i], mode=0))
    return out

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

# ------------------------------------------------------------------------------ mldsa_sign_host: signatures written to host memory round by round
def _pinned(shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    t = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=True)
    a = t.numpy()[:n].view(dtype).reshape(shape)
    a[...] = 0xA5 if dtype == np.uint8 else -7   # stale bytes: every row must be overwritten
    return t, a


__all__ = ['C', 'hashlib', 'os', 'threading', 'time', 'np', 'pytest', 'torch', 'PSET', 'orc', 'hp', 'sets', 'host', 'shake', 'dev', 'dev_off', 'table', 'pairs_ok', 'corruptions', 'make_batch', 'oracle_sigs', 'CLASSES', 'fuzz_batch', '_pinned']
