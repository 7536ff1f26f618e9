"""The call paths whose secrets must be gone afterwards, shared by tests/test_gpu_zeroise.py (the shipped library: residue 0
everywhere) and its negative control (a build of the same sources with every clearing compiled out, MLDSA_TEST_NO_ZEROISE: the
probe must FIND the residue there, otherwise a green test would prove nothing).

Mirror of the reference's `Zeroize, ZeroizeOnDrop` key structs (src/types.rs:19, 45): what the reference keeps on the stack and
drops -- s1, s2, t0, K, rho', rho'', y, w, c s1, c s2 -- lives here in the context's workspace and in the host paths' staging
buffers, and is cleared at the end of the call that used it.

Run as a script it prints {"scenario": [scanned_bytes, nonzero_bytes], ...} as JSON:  python tests/zeroise_scenarios.py [lib.so]
"""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def shake(tag, i, n=32):
    return hashlib.shake_256(tag + int(i).to_bytes(8, "little")).digest(n)


def run_all(lib_path=None):
    from fips204_amd import _lib
    if lib_path:
        _lib.LIB_PATH = lib_path  # before the first load(): the negative control's build
    import numpy as np
    import torch
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa, _cat_with_offsets

    out = {}
    hp = HotPath(0)
    m = MlDsa(65, hotpath=hp)
    n, nk = 3000, 64
    xi = [shake(b"z-key", i) for i in range(nk)]
    msgs = [shake(b"z-msg", i, i % 70) for i in range(n)]
    rnd = [shake(b"z-rnd", i) for i in range(n)]
    kidx = (np.arange(n) % nk).astype(np.uint32)

    # KeyGen::keygen_from_seed: rho', K, s1, s2 (ml_dsa.rs:68-92)
    pk, sk = m.keygen_from_seed(xi)
    out["keygen"] = hp.secret_residue()
    sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
    # Signer::get_public_key: s1, s2, A s1 in coefficient form (ml_dsa.rs:512-545)
    m.get_public_key(sks)
    out["get_public_key"] = hp.secret_residue()
    # synchronous signing (cleared on a helper stream after the last round)
    sig = m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx)
    out["sign"] = hp.secret_residue()
    # mldsa_sign_async (cleared at the end of the enqueued call)
    msg_buf, msg_off = _cat_with_offsets(msgs, m.device)
    d_rnd = m._key_bytes(rnd, 32, "rnd")
    d_kidx = torch.as_tensor(kidx.view(np.int32)).to(m.device)
    sigs = torch.empty((n, m.SIG_LEN), dtype=torch.uint8, device=m.device)
    status = torch.zeros(n, dtype=torch.int32, device=m.device)
    m.sign_device(sks, msg_buf, msg_off, d_rnd, sigs, n, key_idx=d_kidx, status=status, wait=False)
    torch.cuda.synchronize()
    out["sign_async"] = hp.secret_residue()
    # the same call shape three more times: captured, then replayed as a hipGraph
    before = hp.stats()["graph_replays"]
    old_graphs = hp.get_option(1)
    hp.set_option(1, 1)  # MLDSA_OPT_GRAPHS = 1: signing calls of this size replay (the default launches directly)
    for _ in range(3):
        m.sign_device(sks, msg_buf, msg_off, d_rnd, sigs, n, key_idx=d_kidx, status=status, wait=True)
    torch.cuda.synchronize()
    hp.set_option(1, old_graphs)
    out["sign_graph_replay"] = hp.secret_residue()
    out["_graph_replays"] = [hp.stats()["graph_replays"] - before, 0]
    # a call with refused ops (ctx of 256 bytes: MLDSA_ERR_CTX_LEN, lib.rs:274) next to good ones
    ctxs = [b"x" * (256 if i % 100 == 0 else i % 3) for i in range(n)]
    try:
        m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, key_idx=kidx)
        out["_ctx_len_raised"] = [0, 0]
    except ValueError:
        out["_ctx_len_raised"] = [1, 0]
    out["sign_with_refused_ops"] = hp.secret_residue()
    # host-memory entry points: staged private keys (wire bytes + expanded fields), rnd, seeds
    sk_h, pk_h = sk.cpu().numpy(), pk.cpu().numpy()
    rnd_h = np.frombuffer(b"".join(rnd), dtype=np.uint8)
    m.sign_host(sk_h, msgs, rnd_h, key_idx=kidx)                   # pageable buffers: sub-batches + bounce copies
    out["sign_host"] = hp.secret_residue()
    m.keygen_host(np.frombuffer(b"".join(xi), dtype=np.uint8))
    out["keygen_host"] = hp.secret_residue()
    # the direct export of mldsa_sign_host: > 16 384 ops into a page-locked signature buffer
    m44 = MlDsa(44, hotpath=hp)
    nb = 17000
    pk44, sk44 = m44.keygen_host(np.frombuffer(b"".join(xi), dtype=np.uint8))
    pinned = C.c_void_p()
    _lib.check(m44.lib.mldsa_host_alloc(C.byref(pinned), nb * m44.SIG_LEN))
    try:
        buf = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(nb * m44.SIG_LEN,)).reshape(nb, m44.SIG_LEN)
        st = np.zeros(nb, dtype=np.int32)
        big_msgs = [shake(b"z-big", i) for i in range(nb)]
        m44.sign_host(sk44, big_msgs, np.zeros(nb * 32, dtype=np.uint8), key_idx=(np.arange(nb) % nk).astype(np.uint32), out=(buf, st))
        out["sign_host_direct_export"] = hp.secret_residue()
        out["_direct_export_verifies"] = [int(m44.verify_host(pk44, big_msgs, buf, key_idx=(np.arange(nb) % nk).astype(np.uint32)).all()), 0]
    finally:
        m44.lib.mldsa_host_free(pinned)
    # a verify-only call leaves nothing secret behind: nothing to scan in the workspace
    m.verify(pks, msgs, sig, key_idx=kidx)
    out["after_verify"] = hp.secret_residue()
    hp.close()

    # mldsa_ctx_destroy with a caller-owned workspace: the whole buffer is cleared before the context lets go of it
    hp2 = HotPath(0)
    ws = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    hp2.set_workspace(ws)
    m2 = MlDsa(65, hotpath=hp2)
    sks2 = m2.private_keys_from_bytes(sk)
    m2.sign_device(sks2, msg_buf, msg_off, d_rnd, sigs, n, key_idx=d_kidx, status=status, wait=False)
    used = int((ws != 0).sum().item())
    hp2.close()
    torch.cuda.synchronize()
    out["_workspace_bytes_used_before_destroy"] = [used, 0]
    out["destroy"] = [ws.numel(), int((ws != 0).sum().item())]
    # ... and when the caller takes the buffer back (mldsa_ctx_set_workspace(NULL))
    hp3 = HotPath(0)
    hp3.set_workspace(ws)
    m3 = MlDsa(65, hotpath=hp3)
    m3.keygen_from_seed(xi)
    m3.try_sign_with_seed(m3.private_keys_from_bytes(sk), msgs[:500], rnd[:500], key_idx=kidx[:500])
    hp3.set_workspace(None)
    out["replace_workspace"] = [ws.numel(), int((ws != 0).sum().item())]
    hp3.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run_all(sys.argv[1] if len(sys.argv) > 1 else None)))
