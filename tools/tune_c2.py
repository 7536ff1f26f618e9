"""Quick A/B harness for the config-2 kernel: interleaved rounds in one process per variant
(the knob is a per-context option)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fips204_amd.hotpath import HotPath

hp = HotPath(0)
pset = int(os.environ.get("PSET", "44")); batch = int(os.environ.get("BATCH", "4096"))
wl = bench.VerifyArith(hp, pset, batch, 0)
variants = [v for v in os.environ.get("VARIANTS", "6,3,4,8,12,16").split(",")]
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        hp.set_option(4, int(v))  # MLDSA_OPT_VA_BLOCKS_PER_CU
        for i in range(3): wl.step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): wl.step(i)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
for v in variants:
    us = np.median(res[v]); gbs = wl.bytes_per_op * wl.batch / us / 1e3
    print(f"blocks_per_cu={v:>3s}  median {us:7.2f} us  min {min(res[v]):7.2f}  -> {gbs:7.1f} GB/s ({gbs/8000:.1%} of 8 TB/s)")
