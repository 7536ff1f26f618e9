"""Signing, verification and key generation through a given build of the library against the CPU oracle; run as a script:

    python tests/nolatearg_scenarios.py <lib.so>

prints {"case": true / false, ...} as JSON.  tests/test_gpu_small_calls.py runs it on the -DMLDSA_NO_LATE_ARG build (`make nolatearg`,
field.h): every kernel that reads late arguments (k_sign_tail, k_resolve, the single-launch small-call kernels) takes them from an LDS copy
of its argument struct there, and must produce the same bytes -- one-op calls, calls either side of the small-call limits, a 20 000-op
call (speculative rounds, k_resolve), all three parameter sets.  The oracle is the checker only (oracle/liboracle.so)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def shake(tag, i, n=32):
    return hashlib.shake_256(tag + int(i).to_bytes(8, "little")).digest(n)


def run_all(lib_path):
    from fips204_amd import _lib
    _lib.LIB_PATH = lib_path  # before the first load()
    import numpy as np
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    from oracle import oracle as orc

    out = {}
    hp = HotPath(0)
    for pset, sizes in ((44, (1, 300)), (65, (1, 7, 64, 300, 20000)), (87, (1, 150))):
        m = MlDsa(pset, hotpath=hp)
        nk = 5
        xi = [shake(b"nla-key%d" % pset, i) for i in range(nk)]
        pk, sk = m.keygen_from_seed(xi)
        pk_h, sk_h = pk.cpu().numpy(), sk.cpu().numpy()
        ok = True
        okeys = [orc.keygen_from_seed(pset, xi[i]) for i in range(nk)]
        for i in range(nk):
            ok &= bytes(pk_h[i]) == orc.pk_into_bytes(pset, okeys[i][0]) and bytes(sk_h[i]) == orc.sk_into_bytes(pset, okeys[i][1])
        out[f"keygen{pset}"] = bool(ok)
        sks, pks = m.private_keys_from_bytes(sk), m.public_keys_from_bytes(pk)
        for n in sizes:
            msgs = [shake(b"nla-msg", i, i % 90) for i in range(n)]
            ctxs = [shake(b"nla-ctx", i, i % 4) for i in range(n)]
            rnd = [shake(b"nla-rnd", i) for i in range(n)]
            kidx = (np.arange(n) * 7 % nk).astype(np.uint32)
            sig = m.try_sign_with_seed(sks, msgs, rnd, ctxs=ctxs, key_idx=kidx)
            sig_h = sig.cpu().numpy()
            step = max(1, n // 48)  # every signature of the small calls, a spread sample of the large one
            good = all(bytes(sig_h[i]) == orc.sign_internal(pset, okeys[kidx[i]][1], msgs[i], rnd[i], ctx=ctxs[i], mode=orc.MODE_PURE) for i in range(0, n, step))
            bad = sig.clone()
            bad[::3, 40] ^= 1
            v = np.asarray(m.verify(pks, msgs, bad, ctxs=ctxs, key_idx=kidx))
            good &= bool((~v[::3]).all() and v[1::3].all() and v[2::3].all())
            out[f"sign_verify{pset}_n{n}"] = bool(good)
    hp.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run_all(sys.argv[1])))
