"""Batch-size sweep, concurrent small calls, single-operation callers (side-file measurements, never in the bench line)."""
import json
import os
import time

import numpy as np
import torch

from .constants import REFERENCE_PUBLISHED, ROOT
from .cpu import oracle_keygen_rates, usable_cores
from .workloads import WholeOp


SWEEP_SIZES = (1, 64, 1024, 4096, 16384, 65536, 262144)


def run_sweep(hp, pset=65, sizes=SWEEP_SIZES, cpu=True, target_s=0.25):
    """Batch-size curve through the C ABI (the reference's only published metric is single-op latency,
    benches/benchmark.rs:28-62): verify / sign / keygen of n_ops = 1 ... 262 144 ML-DSA-`pset` ops, device-resident inputs, each
    point as (a) ms per call when the caller waits for every call -- the latency an integrator with n ops in hand sees -- and
    (b) ops/s of calls issued back to back, both launched directly and replayed as hipGraphs.  Beside every point: what the
    KAT-pinned oracle needs for the same n ops on one host thread and on all of them, and where the GPU path starts to win."""
    from fips204_amd import _lib
    from fips204_amd.ml_dsa import MlDsa  # noqa: F401
    big = max(sizes)
    wl = WholeOp(hp, pset, "verify", big, 0)
    ml = wl.ml
    g = torch.Generator(device="cuda").manual_seed(4)
    xi = torch.randint(0, 256, (big, 32), dtype=torch.uint8, device="cuda", generator=g)
    kg_pk = torch.empty((big, ml.PK_LEN), dtype=torch.uint8, device="cuda")
    kg_sk = torch.empty((big, ml.SK_LEN), dtype=torch.uint8, device="cuda")
    sig2 = torch.empty_like(wl.sigs)
    hp.reserve(pset, 1, big)
    hp.reserve(pset, 3, big)
    calls = {
        "verify": lambda n: ml.verify_device(wl.pks, wl.msg_buf, wl.msg_off, wl.sigs, wl.ok, n, key_idx=wl.key_idx),
        "sign": lambda n: ml.sign_device(wl.sks, wl.msg_buf, wl.msg_off, wl.rnd, sig2, n, key_idx=wl.key_idx, status=wl.status),
        "keygen": lambda n: ml.keygen_from_seed(xi[:n], out=(kg_pk[:n], kg_sk[:n])),
    }
    old_graphs = hp.get_option(_lib.OPT_GRAPHS)
    out = {"parameter_set": pset, "sizes": list(sizes), "ops": {}}
    try:
        for op, call in calls.items():
            pts = []
            for n in sizes:
                pt = {"n_ops": n}
                for label, gopt in (("direct", 0), ("graph", 2)):
                    hp.set_option(_lib.OPT_GRAPHS, gopt)
                    for _ in range(3):  # first sighting, capture, first replay
                        call(n)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    call(n)
                    torch.cuda.synchronize()
                    reps = int(min(200, max(5, target_s / max(time.perf_counter() - t0, 1e-6))))
                    lat = []
                    for _ in range(reps):
                        t0 = time.perf_counter()
                        call(n)
                        torch.cuda.synchronize()
                        lat.append(time.perf_counter() - t0)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        call(n)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    pt[label] = {"ms_per_call": float(np.median(lat)) * 1e3, "ops_per_s_back_to_back": n * reps / dt, "calls_timed": reps}
                pt["best_ms_per_call"] = min(pt["direct"]["ms_per_call"], pt["graph"]["ms_per_call"])
                pt["best_ops_per_s"] = max(pt["direct"]["ops_per_s_back_to_back"], pt["graph"]["ops_per_s_back_to_back"])
                pts.append(pt)
            out["ops"][op] = {"points": pts}
        assert bool(wl.ok.all()) and int(wl.status.min()) == 0, "sweep: a call failed"
    finally:
        hp.set_option(_lib.OPT_GRAPHS, old_graphs)
    if cpu:
        cores = usable_cores()
        rates = {}
        for kind in ("verify", "sign"):
            wl.kind = kind
            cb = wl.cpu_baseline(budget_s=2.5)
            rates[kind] = (cb["single_thread_value"], cb["value"])
        wl.kind = "verify"
        r1, rt, _, _, _ = oracle_keygen_rates(pset, [bytes(x) for x in xi[:2048].cpu().numpy()], 2.0)
        rates["keygen"] = (r1, rt)
        for op, (r1, rt) in rates.items():
            o = out["ops"][op]
            o["cpu_oracle"] = {"single_thread_ops_per_s": r1, "all_threads_ops_per_s": rt, "threads": cores,
                               "us_per_op_single_thread": 1e6 / r1,
                               "model": "n ops take n / r1 on one thread and max(ceil(n / T) / r1, n / rT) on T threads"}
            win1 = winT = None
            for pt in o["points"]:
                n = pt["n_ops"]
                t1 = n / r1 * 1e3
                tT = max(-(-n // cores) / r1, n / rt) * 1e3
                pt["cpu_ms_one_thread"], pt["cpu_ms_all_threads"] = t1, tT
                pt["gpu_speedup_vs_one_thread"], pt["gpu_speedup_vs_all_threads"] = t1 / pt["best_ms_per_call"], tT / pt["best_ms_per_call"]
                if win1 is None and pt["best_ms_per_call"] < t1:
                    win1 = n
                if winT is None and pt["best_ms_per_call"] < tT:
                    winT = n
            # between the swept sizes: the GPU call time interpolated in log n, the CPU model evaluated exactly
            ns = np.array([pt["n_ops"] for pt in o["points"]], dtype=float)
            ms = np.array([pt["best_ms_per_call"] for pt in o["points"]])

            def break_even(cpu_ms):
                for n in np.unique(np.round(np.logspace(0, np.log10(ns[-1]), 600)).astype(np.int64)):
                    if float(np.interp(np.log(n), np.log(ns), ms)) < cpu_ms(int(n)):
                        return int(n)
                return None
            o["crossover"] = {"first_swept_n_where_gpu_call_beats_one_thread": win1, "first_swept_n_where_gpu_call_beats_all_threads": winT,
                              "break_even_n_vs_one_thread": break_even(lambda n: n / r1 * 1e3),
                              "break_even_n_vs_all_threads": break_even(lambda n: max(-(-n // cores) / r1, n / rt) * 1e3),
                              "note": "a call of fewer ops than the break-even is faster on the CPU path: one GPU call costs about the same few hundred "
                                      "microseconds for every n up to a few thousand (launch- and latency-bound); GPU time interpolated in log n "
                                      "between the swept sizes"}
    # The reference's OWN published single-core latencies (benches/README.md:16-26: 28.0 / 352.9 / 194.8 us for ML-DSA-65 verify / sign /
    # keygen on an i7-7700K) beside every point, and the call size from which one GPU call beats one such core doing the ops in turn.
    for op, o in out["ops"].items():
        us = REFERENCE_PUBLISHED[f"{op}_us"][pset]
        for pt in o["points"]:
            pt["reference_published_ms"] = pt["n_ops"] * us * 1e-3
            pt["gpu_speedup_vs_reference_published_core"] = pt["reference_published_ms"] / pt["best_ms_per_call"]
        ns = np.array([pt["n_ops"] for pt in o["points"]], dtype=float)
        ms = np.array([pt["best_ms_per_call"] for pt in o["points"]])
        be = None
        for n in np.unique(np.round(np.logspace(0, np.log10(ns[-1]), 600)).astype(np.int64)):
            if float(np.interp(np.log(n), np.log(ns), ms)) < n * us * 1e-3:
                be = int(n)
                break
        o.setdefault("crossover", {})
        o["crossover"]["reference_published_us_per_op"] = us
        o["crossover"]["one_op_call_vs_reference_published"] = o["points"][0]["best_ms_per_call"] * 1e3 / us  # > 1: the CPU core wins a single op
        o["crossover"]["break_even_n_vs_reference_published"] = be
    out["reference_published_source"] = REFERENCE_PUBLISHED["source"]
    out["note"] = ("device-resident inputs (expanded keys, messages, signatures in HBM); ms_per_call includes the launch and the wait for the "
                   "result; sign = mldsa_sign (waits inside), verify / keygen = enqueue + stream synchronisation")
    del wl
    torch.cuda.empty_cache()
    return out


def run_small_calls(pset=65, sizes=(64, 1024), contexts=(1, 2, 4, 8, 16), calls=200, graphs=None):
    """Many INDEPENDENT small calls that cannot be coalesced into one batch (a service with per-request latency bounds): C contexts on
    one GPU, each with its own stream and worker thread (mldsa_group_create([0] * C)), every step = one n-op verify call per context,
    enqueued without waiting (mldsa_verify_group, wait = 0), one mldsa_group_sync at the end.  A small call occupies a fraction of
    the SIMDs for ~0.2 ms of latency chains, so calls of different contexts overlap on the device; what one context cannot do -- keep
    the machine busy with 64-op calls -- several can.  Returns {n: {C: ops/s}}."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsaGroup
    out = {}
    for n in sizes:
        out[str(n)] = {}
        for C_ in contexts:
            g = MlDsaGroup(pset, [torch.cuda.current_device()] * C_)
            if graphs is not None:
                g.set_option(1, graphs)  # MLDSA_OPT_GRAPHS
            wls, slices = [], []
            for i in range(C_):
                wl = WholeOp(HotPath.from_handle(g.ctx(i), torch.cuda.current_device()), pset, "verify", n, i, world=C_)
                wls.append(wl)
                slices.append(dict(pks=wl.pks, msg_buf=wl.msg_buf, msg_off=wl.msg_off, key_idx=wl.key_idx, n_ops=n, sigs=wl.sigs, ok=wl.ok,
                                   stream=torch.cuda.Stream().cuda_stream))
            for _ in range(10):
                g.verify_group(slices, wait=False)
            g.sync()
            t0 = time.perf_counter()
            for _ in range(calls):
                g.verify_group(slices, wait=False)
            g.sync()
            dt = time.perf_counter() - t0
            assert all(bool(wl.ok.all()) for wl in wls), "small calls: a valid signature was rejected"
            out[str(n)][str(C_)] = {"ops_per_s": C_ * n * calls / dt, "calls_per_s": C_ * calls / dt, "us_per_step": dt / calls * 1e6}
            del wls, slices
            g.close()
    return out


def run_single_op_callers(pset=65, seconds=1.5, threads="1,8,32,64"):
    """The reference's own call shape -- ONE operation per call (benches/benchmark.rs:28-62 times exactly that) -- from T host threads
    through mldsa_batcher_* (the library coalesces concurrent calls into batches and keeps expanded keys + A_hat in a device-resident
    table), next to the same calls made one at a time with n_ops = 1.  The load generator is tools/batcher_bench.cpp (host threads in
    C++: Python's GIL would be the bottleneck), built here with g++; returns its JSON object or {"skipped": reason}."""
    import shutil
    import subprocess
    import tempfile
    root = ROOT
    if not shutil.which("g++"):
        return {"skipped": "g++ not found"}
    libdir = os.path.join(root, "fips204_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "batcher_bench")
        try:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "batcher_bench.cpp"),
                                   "-o", exe, f"-L{libdir}", "-lmldsa_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
            res = {}
            for lanes in (1, 2):  # one dispatcher, and two on the one GPU (mldsa_batcher_create_on: small batches overlap on the device)
                out = subprocess.run([exe, str(pset), str(seconds), "0", threads, str(lanes)], capture_output=True, text=True, timeout=600)
                if out.returncode != 0:
                    return {"skipped": "tools/batcher_bench.cpp failed: " + out.stderr[-300:]}
                res[f"lanes_{lanes}"] = json.loads(out.stdout)
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired) as e:
            return {"skipped": f"tools/batcher_bench.cpp: {e}"}
    res["usable_cores"] = usable_cores()
    return res

