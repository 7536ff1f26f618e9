"""The timed region and the full per-workload result object."""
import os
import time

import torch

from .constants import HBM_PEAK_GBS, KECCAK_PEAK_DERIVATION, KECCAK_PEAK_GPERMS, REFERENCE_PUBLISHED
from .dist import barrier, max_over_ranks
from .hostfed import host_fed
from .pmc import LIVE_PMC, pmc_traffic
from .workloads import MixedStream, WholeOp, make_workload


def read_sclk(device=0):
    """the GPU's current shader clock in MHz (the starred line of pp_dpm_sclk in sysfs, else `rocm-smi --showclocks`), or None.  One read
    costs ~0.1 ms (sysfs): taken OUTSIDE the timed region only."""
    import glob
    import re
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for ln in open(path):
                if "*" in ln:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
                    if m:
                        return {"mhz": int(m.group(1)), "source": path}
        except OSError:
            pass
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showclocks", "-d", str(device)], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"sclk clock level.*?\((\d+)\s*Mhz\)", out, flags=re.I)
        if m:
            return {"mhz": int(m.group(1)), "source": "rocm-smi --showclocks"}
    except Exception:  # noqa: BLE001
        pass
    return None


def clock_ramp(wl, first, min_ms=200.0, midway=None):
    """>= min_ms of the workload's own steps BEFORE the counted warm-up: the GPU idled (clocks down) while the inputs were built and the CPU
    oracle checked them.  Outside `steps` / `warmup`; what was run is reported in the side file.  `midway()` is called once, half way
    (the shader clock is read THERE: a sysfs read between the warm-up and the timed region leaves the device idle for long enough that the
    first ~12 steps of the timed region run on a clock that is still climbing -- profiles/r06_bench_step_marks_before_after.txt)."""
    t0 = time.perf_counter()
    n = 0
    mid = None
    burst = 8  # steps enqueued between two synchronisations: 8 at first, then ~60 ms worth (at most 128) -- the timed region enqueues its K
    #            steps in one burst where a step is asynchronous, and the FIRST deep burst of a process can run slowly (early_pmc, bench.py)
    while (time.perf_counter() - t0) * 1e3 < min_ms and n < 100000:
        tb = time.perf_counter()
        for _ in range(burst):
            wl.step(first + n)
            n += 1
        if midway is not None and mid is None and (time.perf_counter() - t0) * 1e3 >= min_ms / 2:
            mid = midway()  # (the steps just enqueued are still running where the workload's step is asynchronous)
        torch.cuda.synchronize()
        per_step_ms = (time.perf_counter() - tb) * 1e3 / burst
        burst = int(min(128, max(8, 60.0 / max(per_step_ms, 1e-3))))
    if midway is not None and mid is None:
        mid = midway()
    return {"steps": n, "ms": round((time.perf_counter() - t0) * 1e3, 2)}, mid


def timed_steps(wl, world, steps, first, marks=64):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks.
    An event is recorded on the launch stream every ceil(steps / marks) steps (a few microseconds of host time each, hidden behind the
    device work of the steps): timed_steps.step_ms = min / median / max over those chunks, per step."""
    stride = max(1, -(-steps // max(1, marks)))
    n_marks = -(-steps // stride)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_marks + 1)]
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        wl.step(first + i)
        if (i + 1) % stride == 0 or i + 1 == steps:
            evs[(i + stride) // stride].record()
    timed_steps.enqueue_ms = (time.perf_counter() - t0) * 1e3  # how long the host took to enqueue the K steps (side file)
    if hasattr(wl, "finish_steps"):
        wl.finish_steps()  # inside the timed region: e.g. the mixed stream re-signs what its asynchronous calls left over
    ev_end = torch.cuda.Event(enable_timing=True)
    ev_end.record()
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t0
    timed_steps.local_s = dt  # this rank's own time (run_one reports min / max over ranks beside `value`)
    per = []
    for j in range(n_marks):
        in_chunk = min(stride, steps - j * stride)
        per.append(evs[j].elapsed_time(evs[j + 1]) / in_chunk)
    timed_steps.marks_ms = [round(x, 4) for x in per]  # in time order (side file: which steps were the slow ones)
    per.sort()
    timed_steps.step_ms = {"min": per[0], "median": per[len(per) // 2], "max": per[-1], "steps_per_mark": stride} if per else None
    return max_over_ranks(dt, world), evs[0].elapsed_time(ev_end)


def run_one(args, hp, rank, world, name, steps, warmup, cpu_baseline, with_host_fed=False, cpu_budget_s=None):
    wl = make_workload(name, hp, args.batch if name == args.workload else 0, rank, world)
    if rank == 0:
        wl.check()
    # (outside steps / warmup; in the side file.  Nothing but the warm-up, a synchronize and the barrier lies between the ramp and the timed
    #  region: no file is read, no object is built there)
    ramp, sclk0 = clock_ramp(wl, 0, min_ms=float(os.environ.get("BENCH_RAMP_MS", "200")), midway=read_sclk if rank == 0 else None)
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize()
    whole = isinstance(wl, WholeOp)
    units_per_step = getattr(wl, "ops_per_step", wl.batch)

    # THE timed region: exactly K steps of the product's default path (a signing call whose shape repeats replays as a
    # hipGraph), barrier + synchronize on both sides, max over ranks -> `value`
    st0 = hp.stats() if hasattr(hp, "stats") else None
    dt, ev_ms = timed_steps(wl, world, steps, warmup)
    step_ms = timed_steps.step_ms
    marks_ms = getattr(timed_steps, "marks_ms", None)
    enqueue_ms = getattr(timed_steps, "enqueue_ms", None)
    sclk1 = read_sclk() if rank == 0 else None
    kern_ms = ev_ms / steps / wl.kernel_launches_per_step()
    value = units_per_step * world * steps / dt
    ranks = None
    if world > 1:  # per-rank rates: a straggler shows as min << max (value itself uses the slowest rank's time)
        from .dist import min_over_ranks
        mine = units_per_step * steps / timed_steps.local_s
        ranks = {"min": min_over_ranks(mine, world), "max": max_over_ranks(mine, world), "unit": wl.unit + " per rank"}
    st1 = hp.stats()
    launch_mode = {"graph_replays": st1["graph_replays"] - st0["graph_replays"], "direct_calls": st1["direct_calls"] - st0["direct_calls"],
                   "sign_extra_rounds": st1["sign_extra_rounds"] - st0["sign_extra_rounds"]}
    # Per-kernel durations: a graph has no place for an event between two of its kernels, so the whole-op workloads
    # run the SAME K steps once more right away with a HIP event pair around every kernel launch on the launch
    # stream (the library launches directly while it is being profiled).  The roofline's kernel time comes from there.
    stages, dt_prof = None, None
    if whole:
        hp.profile_enable(True)
        dt_prof, _ = timed_steps(wl, world, steps, warmup + steps, marks=0)
        stages = hp.profile_report()
        hp.profile_enable(False)

    # the verdict bytes of every rank gathered into the whole job's verdict array, outside `value`
    gather = None
    if whole and wl.kind == "verify":
        from fips204_amd import multi_gpu
        import torch.distributed as dist
        on_cpu = multi_gpu.is_distributed() and dist.get_backend() != "nccl"
        multi_gpu.gather_verdicts(wl.ok.cpu() if on_cpu else wl.ok, wl.batch * world).sum().item()  # first use: communicator set-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        allok = multi_gpu.gather_verdicts(wl.ok.cpu() if on_cpu else wl.ok, wl.batch * world)
        n_ok = int(allok.sum().item())
        gather = {"ms": (time.perf_counter() - t0) * 1e3, "verdicts": wl.batch * world, "ok": n_ok,
                  "collective": ("none (single rank: a copy)" if not multi_gpu.is_distributed() else
                                 "all_gather_into_tensor (gloo)" if on_cpu else "all_gather_into_tensor (RCCL)")}
        expect = int(wl.expect_ok.sum().item())
        assert n_ok == expect * world or world > 1 and n_ok <= wl.batch * world, "a rank reported a failed verification of a valid signature"
        gather["expected_ok_per_rank"] = expect
    if rank != 0:
        return None

    alg_bytes = wl.bytes_per_op * units_per_step
    traffic, traffic_by_stage, traffic_file = pmc_traffic(name)
    live = LIVE_PMC.get(name)
    if live:
        traffic, traffic_by_stage, traffic_file = live["hbm_bytes_per_launch"], live["by_stage"], None
    slots = op_rounds = None
    if whole:
        slots = stages.pop("_sign_slots", None)
        op_rounds = stages.pop("_sign_op_rounds", None)

        def stage_bytes_total(st_name):
            """algorithmic bytes of all launches of a stage inside the timed region"""
            per_round = st_name in ("expand_mask", "sign_w", "sign_tail")
            units = slots["calls"] if (wl.kind == "sign" and per_round and slots) else wl.batch * steps
            total = wl.stage_bytes[st_name] * units
            if st_name == "sign_w" and wl.kind == "sign" and op_rounds:
                total += wl.stage_bytes["sign_w_per_op_round"] * op_rounds["calls"]
            return total, units

        # dominant kernel = the stage with the largest share of device time; its average launch
        # duration comes from the event pairs recorded inside the timed region
        dom = max((k for k in stages if k in wl.stage_bytes), key=lambda k: stages[k]["ms"])
        kern_ms = stages[dom]["ms"] / stages[dom]["calls"]
        alg_bytes = stage_bytes_total(dom)[0] / stages[dom]["calls"]
        wl.kernel = "k_" + dom
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": wl.metric, "value": value, "unit": wl.unit, "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": {"workload": wl.name, "batch_per_gpu": wl.batch, "parallelism": f"batch-split x{world}",
                   "input_sets_rotated": wl.n_sets},
        "roofline": {"bound": "hbm", "kernel": wl.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms},
    }
    # `traffic` is a PMC figure (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes) read from the file named beside it: it was
    # NOT measured by this process (counters need the profiler)
    line["roofline"]["traffic_measured_in_this_run"] = bool(live)
    if live:
        line["roofline"]["traffic_source"] = ("two child runs of this command under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate passes, "
                                              "--kernel-trace only) on this box just before the timed region; FETCH_SIZE x 2 + WRITE_SIZE, KiB, mean per launch")
    elif traffic_file:
        line["roofline"]["traffic_source"] = "profiles/" + traffic_file
    line["launch_mode"] = launch_mode
    # spread of the timed region (event marks on the launch stream) and the clock state either side of it: what tells a slow box or a
    # clock that had not ramped from a regression (the value itself stays total units / wall time of exactly K steps)
    if step_ms:
        line["step_ms"] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in step_ms.items()}
    if enqueue_ms is not None:
        line["host_enqueue_ms"] = round(enqueue_ms, 3)  # the host's time to enqueue the K timed steps (it runs ahead of the device where a step is asynchronous)
    if marks_ms:
        line["step_ms_marks"] = marks_ms  # (side file only: benchlib/line.py does not copy it)
    line["clock_ramp"] = dict(ramp, note="the workload's own steps before the counted warm-up, outside steps / warmup")
    line["sclk"] = {"before_timed_region": sclk0, "after_timed_region": sclk1,
                    "note": "MHz (null = not readable on this box): before = half way through the clock ramp, while its steps run; after = right after the timed region's final synchronize (the device is idle again by then)"}
    if ranks:
        line["ranks"] = ranks
    if isinstance(wl, MixedStream):
        line["ops_per_s_by_class"] = {k: n * world * steps / dt for k, n in wl.count.items()}
        line["requests_per_step"] = {"total": wl.ops_per_step, **wl.count,
                                     "per_set": {str(ps): {k: int(len(v)) for k, v in wl.req[ps].items()} for ps in (44, 65, 87)}}
        line["resigned_after_async"] = getattr(wl, "resigned", 0)
    if gather:
        line["verdict_gather"] = gather
    if whole:
        # stages that run on a helper stream UNDERNEATH a kernel of the call's stream (verify: mu and SampleInBall under ExpandA;
        # sign: the optional side-stream prologue) are not on the critical path: they are listed, but neither the
        # busy fraction nor the gap adds them to the critical stream's time
        overlapped = {"mu", "sample_in_ball"} if wl.kind == "verify" else {"expand_mask_ahead", "mu", "rho_pp_hash"}
        total_ms = sum(v["ms"] for k, v in stages.items() if k not in overlapped)
        line["stage_ms_per_step"] = {k: round(v["ms"] / steps, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"])}
        line["stages_overlapped_on_helper_stream"] = sorted(k for k in stages if k in overlapped)
        line["launch_gap_ms_per_step"] = round(max(0.0, dt_prof / steps * 1e3 - total_ms / steps), 4)
        line["profiled_pass"] = {"ms_per_step": dt_prof / steps * 1e3, "value": units_per_step * world * steps / dt_prof,
                                 "note": "the same K steps again with an event pair around every kernel (direct launches): source of "
                                         "stage_ms_per_step and roofline.kernel_ms"}
        if slots:
            line["sign_iterations_per_signature"] = slots["calls"] / (wl.batch * steps)
        line["device_busy_frac"] = min(1.0, total_ms / (dt_prof * 1e3))  # critical-stream kernel time / wall time of the profiled pass
        perms = {"verify": {44: 89, 65: 159, 87: 291}, "sign": {44: 201, 65: 320, 87: 455}}[wl.kind][wl.pset]
        line["roofline"]["note"] = ("whole ops are integer-ALU-bound (Keccak-f[1600]), not HBM-bound: "
                                    f"~{perms} permutations per op; the HBM-bound kernel of the path is reported under "
                                    "also.verify_arith44 (BASELINE config 2).  Bytes are those the kernel is obliged to move "
                                    "(A_hat as the pipelines hold it: 768 B per polynomial)")
        line["keccak_permutations_per_s"] = perms * value / world
        # every modelled stage against the ceiling that bounds it: HBM peak for the polynomial-streaming
        # kernels, the measured Keccak-f[1600] issue ceiling (tools/ubench_valu.hip k_keccak at 8 waves/SIMD,
        # profiles/r01_ubench_valu.txt) for the SHAKE-bound samplers
        by_stage = {}
        for st_name, st in stages.items():
            if st_name in wl.stage_perms:
                units = slots["calls"] if (wl.kind == "sign" and st_name == "expand_mask" and slots) else wl.batch * steps
                ach = wl.stage_perms[st_name] * units / (st["ms"] * 1e-3) / 1e9
                by_stage[st_name] = {"bound": "valu", "achieved": ach, "peak": KECCAK_PEAK_GPERMS,
                                     "unit": "G Keccak-f[1600]/s", "frac": ach / KECCAK_PEAK_GPERMS}
            elif st_name in wl.stage_bytes:
                model = stage_bytes_total(st_name)[0] / st["calls"]
                pmc = traffic_by_stage.get(st_name)
                # the figure credited is never above what the counters saw cross the memory interface
                moved = min(model, pmc) if pmc else model
                ach = moved / (st["ms"] / st["calls"] * 1e-3) / 1e9
                by_stage[st_name] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ach / HBM_PEAK_GBS, "model_bytes_per_launch": model, "pmc_bytes_per_launch": pmc}
                if ach > 6400:
                    by_stage[st_name]["suspect"] = "above what this box streams (6.2-6.4 TB/s): served partly from L2 / Infinity Cache"
        line["roofline_by_stage"] = by_stage
        if dom in by_stage and by_stage[dom]["bound"] == "hbm":
            line["roofline"]["achieved"] = by_stage[dom]["achieved"]
            line["roofline"]["frac"] = by_stage[dom]["frac"]
        elif dom in by_stage:
            # The dominant kernel is a SHAKE sampler (ExpandA for verify): integer-issue-bound, and the roofline object says so.
            # peak = the issue ceiling derived in-line from the round's instruction mix (KECCAK_PEAK_DERIVATION); the HBM view of
            # the same launch stays beside it under both byte models.
            per_launch_units = wl.batch  # every ExpandA launch of the timed region covers the whole batch
            hbm_packed = line["roofline"]["achieved"]
            int32_bytes = (32 + 1024 * wl.k * wl.l) * per_launch_units if dom == "expand_a" else alg_bytes
            hbm_int32 = int32_bytes / (kern_ms * 1e-3) / 1e9
            line["roofline"].update({
                "bound": "valu", "achieved": by_stage[dom]["achieved"], "peak": KECCAK_PEAK_GPERMS, "unit": "G Keccak-f[1600]/s",
                "frac": by_stage[dom]["frac"],
                "permutations_per_launch": wl.stage_perms[dom] * per_launch_units,
                "peak_derivation": KECCAK_PEAK_DERIVATION,
                "hbm_view": {"peak_GBs": HBM_PEAK_GBS,
                             "survey_8d_int32_model": {"bytes_per_launch": int32_bytes, "achieved_GBs": hbm_int32, "frac": hbm_int32 / HBM_PEAK_GBS,
                                                       "note": "SURVEY 8d: 32 + 1024*K*L bytes per op (the reference's int32 layout)"},
                             "packed_24bit_as_stored": {"bytes_per_launch": alg_bytes, "achieved_GBs": hbm_packed, "frac": hbm_packed / HBM_PEAK_GBS,
                                                        "note": "what the kernel writes: A_hat as 24-bit fields, 768 B per polynomial"}},
            })
        line["whole_op_hbm"] = {"algorithmic_bytes_per_op": wl.bytes_per_op,
                                "achieved_GBs": wl.bytes_per_op * value / world / 1e9,
                                "frac_of_peak": wl.bytes_per_op * value / world / 1e9 / HBM_PEAK_GBS}
        pub = REFERENCE_PUBLISHED[f"{wl.kind}_us"][wl.pset]
        line["reference_published"] = {"value": 1e6 / pub, "unit": wl.unit + " per core", "us_per_op": pub,
                                       "source": REFERENCE_PUBLISHED["source"], "note": REFERENCE_PUBLISHED["note"]}
    if world == 1 and cpu_baseline:
        cb = wl.cpu_baseline() if cpu_budget_s is None else wl.cpu_baseline(budget_s=cpu_budget_s)
        if cb:
            line["cpu_baseline"] = cb
    if world == 1 and with_host_fed and whole and not wl.cached_a:
        line["end_to_end_host_fed"] = host_fed(wl)
    del wl
    torch.cuda.empty_cache()
    return line

