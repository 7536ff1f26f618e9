#!/usr/bin/env python3
"""Why do the first steps of bench.py's timed region run slowly?  One process, the headline workload (verify65), several timed regions in a
row (ramp -> warm-up -> K steps with an event mark per step), optionally after the rocprofv3 --pmc child passes of the default run:
    PMC=1 [SCLK=1] [STATS=1] python tools/marks_probe.py 20 100 100 50 200      (SCLK: read the shader clock half way through the ramp, as bench.py does)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("IMPORT_FIRST") == "1":  # (as bench.py does: torch is imported -- the GPU not touched -- before the children run)
    import torch  # noqa: F401
if os.environ.get("PMC") == "1":
    from benchlib.pmc import measure_pmc_traffic
    print("pmc children:", bool(measure_pmc_traffic("verify65")), bool(measure_pmc_traffic("verify_arith44")), flush=True)
import torch  # noqa: E402
from benchlib import runner  # noqa: E402
from benchlib.workloads import make_workload  # noqa: E402
from fips204_amd.hotpath import HotPath  # noqa: E402

hp = HotPath(0)
wl = make_workload("verify65", hp, 0, 0, 1)
wl.check()
for k in [int(a) for a in sys.argv[1:]] or [20, 100]:
    ramp, clk = runner.clock_ramp(wl, 0, midway=runner.read_sclk if os.environ.get("SCLK") == "1" else None)
    if os.environ.get("STATS") == "1":
        hp.stats()
    for i in range(5):
        wl.step(i)
    torch.cuda.synchronize()
    dt, ev = runner.timed_steps(wl, 1, k, 5, marks=k)
    m = runner.timed_steps.marks_ms
    print(f"K={k:4d} ms/step {dt / k * 1e3:.4f}  host enqueue {runner.timed_steps.enqueue_ms:.2f} ms  first marks {m[:8]}  last {m[-3:]}", flush=True)
hp.close()
