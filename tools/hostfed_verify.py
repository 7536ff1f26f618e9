"""mldsa_verify_host on page-locked buffers, ML-DSA-65:   python tools/hostfed_verify.py [n_ops] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fips204_amd.hotpath import HotPath
from fips204_amd.ml_dsa import MlDsa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hp = HotPath(0); m = MlDsa(65, hotpath=hp)
nk = 1024
def pin(a):
    t = torch.empty(max(a.nbytes, 1), dtype=torch.uint8, pin_memory=True)
    v = t.numpy()[:a.nbytes].view(a.dtype).reshape(a.shape); v[...] = a
    return t, v
rng = np.random.default_rng(1)
xi = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
pk, sk = m.keygen_host(xi)
k1, pk = pin(pk)
k2, mflat = pin(rng.integers(0, 256, n * 32, dtype=np.uint8))
k3, moff = pin((np.arange(n + 1, dtype=np.uint64) * 32))
rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
k5, kidx = pin((np.arange(n) % nk).astype(np.uint32))
k6, sig = pin(m.sign_host(sk, (mflat, moff), rnd, key_idx=kidx))
k7, ok = pin(np.zeros(n, dtype=np.uint8))
assert m.verify_host(pk, (mflat, moff), sig, key_idx=kidx, out=ok).all()
t0 = time.perf_counter()
for _ in range(reps):
    m.verify_host(pk, (mflat, moff), sig, key_idx=kidx, out=ok)
dt = (time.perf_counter() - t0) / reps
print(f"verify_host {n} ops: {dt*1e3:.3f} ms per call, {n/dt/1e6:.3f} M verifies/s")
assert ok.all()
sig[5, 100] ^= 1
assert not m.verify_host(pk, (mflat, moff), sig, key_idx=kidx, out=ok)[5] and ok.sum() == n - 1
hp.close()
