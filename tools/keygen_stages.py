import sys, json, torch
sys.path.insert(0, '/root/repo')
from fips204_amd.hotpath import HotPath
from fips204_amd.ml_dsa import MlDsa
hp = HotPath(0)
for pset in (44, 65, 87):
    m = MlDsa(pset, hotpath=hp)
    n = 65536
    xi = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device='cuda')
    pk = torch.empty((n, m.PK_LEN), dtype=torch.uint8, device='cuda'); sk = torch.empty((n, m.SK_LEN), dtype=torch.uint8, device='cuda')
    for _ in range(3): m.keygen_from_seed(xi, out=(pk, sk))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10): m.keygen_from_seed(xi, out=(pk, sk))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    hp.profile_enable(True)
    for _ in range(10): m.keygen_from_seed(xi, out=(pk, sk))
    st = hp.profile_report(); hp.profile_enable(False)
    print(pset, f"{dt*1e3:.3f} ms per 65536 keys = {n/dt/1e6:.2f} M keys/s", {k: round(v['ms']/10, 3) for k, v in st.items()})
