#!/usr/bin/env python3
"""Where does the time of a SMALL call go?  tools/latency_probe.py <verify|sign|keygen> <n_ops> [calls]
Device-resident ML-DSA-65 (environment SET = 44 / 65 / 87) call of n_ops ops, repeated with a stream synchronisation after each: prints the median wall time per call.
Run it under `rocprofv3 --kernel-trace --stats` to get the kernels' own durations next to it (the difference is launch + wait)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from fips204_amd.hotpath import HotPath  # noqa: E402

op, n = sys.argv[1], int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 300
hp = HotPath(0)
hp.set_option(1, int(__import__("os").environ.get("GRAPHS", "0")))
hp.set_option(12, int(__import__("os").environ.get("COOP", "1")))  # MLDSA_OPT_COOP_HASH
wl = bench.WholeOp(hp, int(__import__("os").environ.get("SET", "65")), "verify", max(n, 64), 0)
ml = wl.ml
xi = torch.randint(0, 256, (max(n, 64), 32), dtype=torch.uint8, device="cuda")
pk = torch.empty((max(n, 64), ml.PK_LEN), dtype=torch.uint8, device="cuda")
sk = torch.empty((max(n, 64), ml.SK_LEN), dtype=torch.uint8, device="cuda")
sig2 = torch.empty_like(wl.sigs)
call = {"verify": lambda: ml.verify_device(wl.pks, wl.msg_buf, wl.msg_off, wl.sigs, wl.ok, n, key_idx=wl.key_idx),
        "sign": lambda: ml.sign_device(wl.sks, wl.msg_buf, wl.msg_off, wl.rnd, sig2, n, key_idx=wl.key_idx, status=wl.status),
        "keygen": lambda: ml.keygen_from_seed(xi[:n], out=(pk[:n], sk[:n]))}[op]
for _ in range(10):
    call()
torch.cuda.synchronize()
lat = []
gap_s = float(__import__("os").environ.get("CALL_GAP_US", "0")) * 1e-6  # a pause between calls: lets tools/small_call_timeline.py tell calls apart
for _ in range(calls):
    if gap_s:
        time.sleep(gap_s)
    t0 = time.perf_counter()
    call()
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
if __import__("os").environ.get("WALL_JSON"):
    __import__("json").dump({"op": op, "n_ops": n, "calls": calls, "wall_us": [x * 1e6 for x in lat]}, open(__import__("os").environ["WALL_JSON"], "w"))
print(f"{op} n={n}: median {np.median(lat) * 1e6:.1f} us per call (min {min(lat) * 1e6:.1f}), {calls} calls")
