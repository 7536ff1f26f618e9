#!/usr/bin/env python3
"""bench.py -- throughput of the batched ML-DSA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--full]

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  For N > 1 the driver launches one rank per GPU (torch.distributed.run);
the batch is sharded per rank with no data-path collective (independent ops),
so scaling is "weak": every rank processes its own full-size batch.

Workloads (BASELINE.json configs):
  verify65        the metric's headline: whole ML-DSA-65 verifies/s, batch 65536 (default)
  sign65          config[2]: whole ML-DSA-65 signs/s, batch 65536
  verify_arith44  config[1]: ml_dsa_44, batch 4096, NTT/INTT + pointwise kernels only
                  (the fused verify-arithmetic unit; the HBM-roofline kernel)

Rank 0 prints ONE JSON line of at most 6 000 bytes (benchlib/line.py): the contract's keys,
"roofline" (dominant kernel, HIP-event timed on the launch stream inside the timed region),
"cpu_baseline" (the KAT-pinned CPU oracle timed on this box's host cores, rank 0, N = 1) and,
in the default run, a compact "also" for sign65 and verify_arith44.  Everything else -- stage
tables, by-stage rooflines, the Keccak-ceiling derivation, host-fed legs and, with --full, the
corrupted / from-wire-bytes variants and the batch-size sweep with its CPU crossover -- goes to
the side file named in the line ("extras_file": bench_extras.json beside this script).
The parts live in benchlib/ (workloads, cpu, pmc, runner, sweep, group, line); this file is the CLI.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import line as bline  # noqa: E402  (no torch in there)
from benchlib.pmc import LIVE_PMC, measure_pmc_traffic, pmc_traffic  # noqa: E402,F401


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("MLDSA_BENCH_WORKLOAD", "verify65"))
    ap.add_argument("--batch", type=int, default=0, help="ops per GPU (0 = the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only --workload: no also.sign65 / also.verify_arith44, no live PMC passes")
    ap.add_argument("--full", action="store_true",
                    help="default run + the 1 %-corrupted verify batch, the from-wire-bytes units, host-fed legs and the batch-size sweep; "
                         "all of it in the side file, the printed line stays the compact one")
    ap.add_argument("--extras-file", default=bline.EXTRAS_FILE, help="name of the side file (beside bench.py; '' = do not write one)")
    ap.add_argument("--backend", default=os.environ.get("MLDSA_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL) or gloo (CPU rendezvous; lets several ranks "
                         "share one GPU for a functional check on a 1-GPU box)")
    ap.add_argument("--graphs", type=int, default=-1, help="override MLDSA_OPT_GRAPHS (hipGraph replay) of the context: 0 / 1")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic live for --workload (two child rocprofv3 counter passes); "
                                                       "the default run does this for verify65 unless --no-pmc")
    ap.add_argument("--no-pmc", action="store_true", help="never start the profiler: roofline.traffic from this round's file under profiles/ (null when absent)")
    ap.add_argument("--resident", action="store_true",
                    help="with --inproc: slices resident per device and the device-resident group calls (mldsa_verify_group / mldsa_sign_group): "
                         "the contract's HBM-resident value from one process")
    ap.add_argument("--inproc", action="store_true",
                    help="ONE process driving --gpus N devices through the library's own batch split (mldsa_group_create + "
                         "mldsa_*_host_group): host-memory inputs, so the figure is PCIe-inclusive and is NOT the contract's `value` path")
    return ap.parse_args(argv)



def early_pmc(args):
    """The live HBM-traffic counters: child runs of this script under `rocprofv3 --pmc` (benchlib/pmc.py), single-GPU runs only -- started
    before this process has touched the GPU, and before torch (with it the HIP runtime) is even LOADED: with the runtime mapped while a
    profiler session ran in another process, the first deep burst of launches of this process ran up to 25 % slow for ~25 ms -- the
    first dozen steps of a 100-step timed region (profiles/r06_bench_step_marks_before_after.txt)."""
    if args.inproc or (args.gpus > 1 and "RANK" not in os.environ):
        return
    single = args.gpus == 1 and "RANK" not in os.environ
    if single and not args.no_pmc and (args.pmc or (args.workload == "verify65" and not args.no_extras)):
        # the headline workload, and in the default run config[1]'s kernel too (also.verify_arith44.traffic_ratio: PMC bytes / algorithmic bytes)
        for name in (args.workload,) + (("verify_arith44",) if args.workload == "verify65" and not args.no_extras else ()):
            got = measure_pmc_traffic(name)
            if got:
                LIVE_PMC[name] = got


_ARGS = None
_T_START = time.perf_counter()
if __name__ == "__main__":
    _ARGS = parse()
    early_pmc(_ARGS)

import torch  # noqa: E402

from benchlib.constants import *  # noqa: E402,F401,F403  (Q, peaks, SETS, REFERENCE_PUBLISHED: tools/ and tests/ read them through `bench`)
from benchlib.cpu import _shake, oracle_keygen_rates, usable_cores  # noqa: E402,F401
from benchlib.dist import barrier, dist_setup, max_over_ranks  # noqa: E402,F401
from benchlib.group import run_inproc, run_inproc_resident  # noqa: E402
from benchlib.hostfed import host_fed, measure_h2d_GBs  # noqa: E402,F401
from benchlib.runner import run_one, timed_steps  # noqa: E402,F401
from benchlib.sweep import SWEEP_SIZES, run_single_op_callers, run_small_calls, run_sweep  # noqa: E402,F401
from benchlib.workloads import MixedStream, SeamKernel, VerifyArith, WholeOp, config5_requests, make_workload  # noqa: E402,F401


# (workload, steps, warmup, CPU-baseline budget in seconds) of the default run's `also` objects
# (the GPU idles for seconds while the previous workload's CPU baseline runs and its clocks drop: the warm-ups are long enough to bring
#  them back before a timed region starts -- config[1]'s kernel runs 26 us a step, so its 500 warm-up steps are 13 ms)
ALSO_DEFAULT = (("sign65", 30, 6, 6.0), ("verify_arith44", 2000, 500, 3.0))
ALSO_FULL = (("verify65_corrupt1", 20, 3, 0.0), ("verify65_wire", 20, 3, 3.0), ("sign65_wire", 10, 2, 3.0))


def run_sweep_workload(args, hp, multi_gpu):
    """`--workload sweep`: the batch-size curve (side file) with a compact line whose value is the 65 536-op verify point"""
    sw = run_sweep(hp, cpu=not args.no_cpu_baseline)
    sw["concurrent_small_verify_calls"] = run_small_calls()
    sw["single_op_callers"] = run_single_op_callers()  # (a process of its own, with its own context)
    v = next(pt for pt in sw["ops"]["verify"]["points"] if pt["n_ops"] == 65536)
    line = {"metric": "ML-DSA-65 verifies/sec per GPU (batched); batch-size sweep through the C ABI", "value": v["best_ops_per_s"], "unit": "verifies/s",
            "n_gpus": 1, "steps": v["direct"]["calls_timed"], "warmup": 3, "ms_per_step": v["best_ms_per_call"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "ml_dsa_65 verify / sign / keygen, n_ops = 1 ... 262144 per call, inputs resident in HBM", "batch_per_gpu": 65536,
                       "parallelism": "batch-split x1"},
            "roofline": None,
            "one_op_ms": {op: bline._num(o["points"][0]["best_ms_per_call"]) for op, o in sw["ops"].items()},
            "break_even_n_vs_reference_published": {op: o.get("crossover", {}).get("break_even_n_vs_reference_published") for op, o in sw["ops"].items()}}
    if args.extras_file:
        bline.write_extras(ROOT, {"sweep": sw, "library_stats": hp.stats()}, args.extras_file)
        line["extras_file"] = args.extras_file
    bline.emit(line)
    hp.close()
    return multi_gpu.finish()


def main():
    args = _ARGS if _ARGS is not None else parse()
    if args.inproc:
        return run_inproc_resident(args) if args.resident else run_inproc(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher.  It has not touched the GPU
        # (importing torch does not), starts N fresh rank processes of this script and relays rank 0's line.
        from fips204_amd import multi_gpu
        raise SystemExit(multi_gpu.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    t_start = _T_START  # (the side file's wall_s includes the live PMC passes)
    rank, local_rank, world = dist_setup(args)
    from fips204_amd import multi_gpu
    from fips204_amd.hotpath import HotPath
    hp = HotPath(local_rank)
    if args.graphs >= 0:
        hp.set_option(1, args.graphs)
    default_run = world == 1 and args.workload == "verify65" and not args.no_extras
    if args.workload == "sweep":
        if world != 1:
            raise SystemExit("bench.py: --workload sweep is a single-GPU measurement")
        return run_sweep_workload(args, hp, multi_gpu)
    cb = not args.no_cpu_baseline
    full = run_one(args, hp, rank, world, args.workload, args.steps, args.warmup, cb,
                   with_host_fed=(default_run and args.full) or os.environ.get("MLDSA_BENCH_HOST_FED") == "1")
    # the default single-GPU run also carries BASELINE config[2] (whole sign) and config[1] (the HBM-roofline kernel) as compact
    # `also` objects; --full adds the SURVEY 8(d) variants and the sweep -- to the side file only
    also, extras = {}, {}
    if default_run:
        for name, st, wu, budget in ALSO_DEFAULT + (ALSO_FULL if args.full else ()):
            sub = run_one(args, hp, rank, world, name, st, wu, cb and budget > 0, with_host_fed=(args.full and name == "sign65"), cpu_budget_s=budget or None)
            (also if (name, st, wu, budget) in ALSO_DEFAULT else extras)[name] = sub
        if args.full:
            extras["sweep"] = run_sweep(hp, cpu=cb)
    if rank == 0:
        line = bline.compact_line(full, also, extras_file=args.extras_file or None)
        if args.extras_file:
            side = {"headline": full, "also": also, **extras, "library_stats": hp.stats(), "argv": sys.argv[1:],
                    "wall_s": time.perf_counter() - t_start}
            bline.write_extras(ROOT, side, args.extras_file)
        bline.emit(line)
    hp.close()
    multi_gpu.finish()


if __name__ == "__main__":
    main()
