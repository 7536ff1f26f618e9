#!/usr/bin/env python3
"""Day-one check of a multi-GPU node (the single-op shape this stands behind: /root/reference/src/traits.rs:330-362).

Nothing in this repository has ever run on a device other than 0, and the RCCL all-gather of csrc/group.hip has never run across two
devices: the pool leases one GPU.  This script is what to run FIRST on an 8-GPU node, before any scaling number is believed:

    python tools/scale_preflight.py                 every visible device, then bench.py --gpus N
    python tools/scale_preflight.py --dry-run --gpus 2   no GPU: the rank plumbing only (gloo), as the CPU test-suite runs it

Steps (real mode), each compared with the KAT-pinned oracle or with a second way of computing the same bytes:
  1. per device d: a context on d, 256 ML-DSA-65 signatures and 1 024 verifications (1 % damaged) -- every signature and verdict
     against the oracle.
  2. one mldsa_group over ALL devices: mldsa_verify_group on device-resident slices, then mldsa_group_allgather of the verdict bytes
     with use_rccl = 1 (ncclAllGather over xGMI; needs distinct devices) and with use_rccl = 0 (peer copies): byte-equal, and equal
     to the verdicts step 1's oracle gives.
  3. `python bench.py --gpus N` (N = all devices; one rank per GPU over RCCL): the compact line parses, is <= 6 000 bytes, names
     n_gpus = N and carries ranks.min / ranks.max (a straggler shows as min << max).
Prints ONE JSON object: per step "ok" / "failed: ..." / "skipped: ..." (on the 1-GPU lease: the RCCL leg and ranks are skipped and
said so).  Exit code 0 iff nothing failed."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def shake(tag, i, n=32):
    import hashlib
    return hashlib.shake_256(tag + i.to_bytes(8, "little")).digest(n)


def check_device(dev, pset=65, n_sign=256, n_verify=1024):
    """step 1 on one device; returns (report, data for step 2)"""
    import numpy as np
    import torch
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa
    from oracle import oracle as orc
    torch.cuda.set_device(dev)
    hp = HotPath(dev)
    try:
        m = MlDsa(pset, hotpath=hp)
        nk = 8
        xi = [shake(b"preflight-key" + bytes([dev]), i) for i in range(nk)]
        pk, sk = m.keygen_from_seed(xi)
        pks, sks = m.public_keys_from_bytes(pk), m.private_keys_from_bytes(sk)
        keys_o = [orc.keygen_from_seed(pset, x) for x in xi]
        pkb = pk.cpu().numpy()
        for i in range(nk):
            assert pkb[i].tobytes() == orc.pk_into_bytes(pset, keys_o[i][0]), f"device {dev}: generated key {i} differs from the oracle"
        msgs = [shake(b"preflight-msg", 1000 * dev + i, 40 + i % 17) for i in range(n_verify)]
        rnd = [shake(b"preflight-rnd", 1000 * dev + i) for i in range(n_verify)]
        kidx = (np.arange(n_verify) % nk).astype(np.uint32)
        sig = m.try_sign_with_seed(sks, msgs, rnd, key_idx=kidx)
        sigb = sig.cpu().numpy()
        for i in range(n_sign):
            want = orc.sign_internal(pset, keys_o[int(kidx[i])][1], msgs[i], rnd[i], ctx=b"", mode=0)
            assert sigb[i].tobytes() == want, f"device {dev}: signature {i} differs from the oracle"
        damaged = sig.clone()
        rows = torch.arange(7, n_verify, 100, device=sig.device)
        damaged[rows, (rows * 31) % m.SIG_LEN] ^= 0x04
        got = m.verify(pks, msgs, damaged, key_idx=kidx)
        dam = damaged.cpu().numpy()
        want = np.array([orc.verify_internal(pset, keys_o[int(kidx[i])][0], msgs[i], dam[i].tobytes(), ctx=b"", mode=0) for i in range(n_verify)])
        assert np.array_equal(np.asarray(got), want), f"device {dev}: verdicts differ from the oracle"
        assert int(want.sum()) == n_verify - len(rows), "the damaged signatures were not all rejected by the oracle"
        return "ok", dict(xi=xi, msgs=msgs, sigs=dam, kidx=kidx, want=want)
    finally:
        hp.close()


def check_group(devices, per_dev, pset=65):
    """step 2: one group over all devices, device-resident slices, all-gather with and without RCCL"""
    import numpy as np
    import torch
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsa, MlDsaGroup, _cat_with_offsets
    g = MlDsaGroup(pset, devices)
    report = {}
    try:
        slices, oks, wants = [], [], []
        for i, d in enumerate(devices):
            data = per_dev[d]
            with torch.cuda.device(d):
                m = MlDsa(pset, hotpath=HotPath.from_handle(g.ctx(i), d))
                pk, _ = m.keygen_from_seed(data["xi"])
                pks = m.public_keys_from_bytes(pk)
                mb, mo = _cat_with_offsets(data["msgs"], m.device)
                sg = torch.from_numpy(np.ascontiguousarray(data["sigs"])).to(m.device)
                kidx = torch.from_numpy(data["kidx"].view(np.int32)).to(m.device)
                ok = torch.zeros(len(data["msgs"]), dtype=torch.uint8, device=m.device)
                slices.append(dict(pks=pks, msg_buf=mb, msg_off=mo, key_idx=kidx, n_ops=len(data["msgs"]), sigs=sg, ok=ok))
                oks.append(ok)
                wants.append(data["want"])
        g.verify_group(slices, wait=True)
        for i, (ok, want) in enumerate(zip(oks, wants)):
            assert np.array_equal(ok.cpu().numpy().astype(bool), want), f"group slice {i}: verdicts differ from the oracle"
        report["verify_group"] = "ok"
        n = len(wants[0])
        total = n * len(devices)
        gathered = {}
        distinct = len(set(devices)) == len(devices) and len(devices) > 1
        for use_rccl in ((1, 0) if distinct else (0,)):
            bufs = []
            for i, d in enumerate(devices):
                with torch.cuda.device(d):
                    b = torch.zeros(total, dtype=torch.uint8, device=f"cuda:{d}")
                    b[i * n:(i + 1) * n] = oks[i]
                    bufs.append(b)
            for d in set(devices):
                torch.cuda.synchronize(d)
            g.allgather(bufs, total, use_rccl=use_rccl)
            for d in set(devices):
                torch.cuda.synchronize(d)
            gathered[use_rccl] = [b.cpu().numpy() for b in bufs]
            want_all = np.concatenate(wants).astype(np.uint8)
            for i, b in enumerate(gathered[use_rccl]):
                assert np.array_equal(b, want_all), f"all-gather (use_rccl={use_rccl}): device {i}'s buffer differs from the oracle's verdicts"
        if distinct:
            assert all(np.array_equal(a, b) for a, b in zip(gathered[1], gathered[0])), "RCCL and peer-copy all-gather disagree"
            report["allgather_rccl_vs_copies"] = "ok"
        else:
            report["allgather_copies"] = "ok"
            report["allgather_rccl_vs_copies"] = "skipped: needs two or more DISTINCT devices (this box shows %d)" % len(set(devices))
    finally:
        g.close()
    return report


def rccl_report(import_torch=True):
    """Which RCCL the library's group gather binds in THIS process (mldsa_group_rccl_info(NULL): the loader's search without a communicator or a
    device) beside the one torch has mapped: in a Python host they must be the same file ("reused ..."), never two RCCLs on one HIP runtime."""
    import ctypes as C
    rep = {}
    if import_torch:
        import torch  # noqa: F401  (maps torch/lib/librccl.so, SONAME librccl.so.1)
        try:
            v = torch.cuda.nccl.version()
            rep["torch_rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as e:  # noqa: BLE001
            rep["torch_rccl_version"] = "unknown: %r" % (e,)
    mapped = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
    rep["rccl_mapped_before_probe"] = mapped
    from fips204_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(1024)
    ver = lib.mldsa_group_rccl_info(None, buf, len(buf))
    if ver < 0:
        rep["library_rccl"] = "failed: no librccl.so could be loaded"
        return rep
    how, _, path = buf.value.decode().partition(" ")
    rep["library_rccl"] = {"how": how, "file": path, "version_code": ver}
    if mapped:
        same = os.path.realpath(path) in {os.path.realpath(m) for m in mapped}
        rep["library_rccl_is_the_mapped_one"] = "ok" if (how == "reused" and same) else "failed: the process had %r mapped, the library bound %s %r" % (mapped, how, path)
    return rep


def run_bench(n, extra=()):
    """step 3: bench.py --gpus n as a child process (never an exec from a process that has touched the GPU)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-pmc",
           "--no-extras", "--extras-file", ""] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return "failed: rc %d: %s" % (out.returncode, out.stderr[-400:]), None
    return check_line(lines[-1], n)


def check_line(text, n):
    if len(text) > 6000:
        return "failed: the line is %d bytes" % len(text), None
    line = json.loads(text)
    if line.get("n_gpus") != n:
        return "failed: n_gpus = %r, expected %d" % (line.get("n_gpus"), n), line
    if not line.get("value", 0) > 0:
        return "failed: value = %r" % line.get("value"), line
    if n > 1:
        r = line.get("ranks")
        if not r or not (0 < r["min"] <= r["max"]):
            return "failed: ranks = %r" % (r,), line
        if r["min"] < 0.8 * r["max"]:
            return "ok (STRAGGLER: slowest rank at %.0f %% of the fastest)" % (100 * r["min"] / r["max"]), line
    return "ok", line


# ------------------------------------------------------------------ dry run: the rank plumbing without a GPU (gloo)
def dry_run_worker():
    """one rank of the dry run: shard the job, gather fake verdicts, min / max over ranks, rank 0 builds and prints the compact line"""
    import torch
    from benchlib import line as bline
    from fips204_amd import multi_gpu
    rank, _, world = multi_gpu.init_process_group("gloo")
    n_total = 1000 + world                          # ragged on purpose
    start, count = multi_gpu.shard(n_total, rank, world)
    ok = torch.tensor([0 if (start + i) % 100 == 7 else 1 for i in range(count)], dtype=torch.uint8)
    allok = multi_gpu.gather_verdicts(ok, n_total)
    mine = 1.0e6 * (1.0 + 0.01 * rank)              # this rank's (fake) rate
    lo, hi = multi_gpu.min_over_ranks(mine), multi_gpu.max_over_ranks(mine)
    multi_gpu.barrier()
    if rank == 0:
        assert allok.tolist() == [0 if i % 100 == 7 else 1 for i in range(n_total)]
        full = {"metric": "ML-DSA-65 verifies/sec per GPU (batched); % HBM roofline", "value": lo * world, "unit": "verifies/s", "n_gpus": world, "steps": 5,
                "warmup": 2, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": "dry run (no GPU): rank plumbing only", "batch_per_gpu": count, "parallelism": f"batch-split x{world}"},
                "roofline": {"bound": "hbm", "kernel": "none (dry run)", "achieved": 0.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0, "traffic": None,
                             "algorithmic_bytes_per_launch": 0, "kernel_ms": 0.0, "traffic_measured_in_this_run": False},
                "ranks": {"min": lo, "max": hi}}
        bline.emit(bline.compact_line(full, extras_file=None))
    multi_gpu.finish()


def dry_run(n):
    from fips204_amd import multi_gpu
    import contextlib
    import io
    import tempfile
    with tempfile.TemporaryFile(mode="w+") as f:
        # launch_ranks passes rank 0's stdout through: capture it at the file-descriptor level
        saved = os.dup(1)
        sys.stdout.flush()
        os.dup2(f.fileno(), 1)
        try:
            rc = multi_gpu.launch_ranks(n, [os.path.abspath(__file__), "--rank-worker"], timeout=300)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        f.seek(0)
        text = f.read()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if rc != 0 or not lines:
        return {"dry_run": "failed: rc %d, output %r" % (rc, text[-300:])}
    verdict, line = check_line(lines[-1], n)
    return {"dry_run": verdict, "line": line, "skipped": "no GPU work: device contexts, the group all-gather and bench.py were not run"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="devices to check (0 = all visible)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: N gloo ranks exercise the shard / gather / min-max / line path")
    ap.add_argument("--rank-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--rccl-report", action="store_true", help="only: which librccl the library would bind in this process (no GPU needed)")
    ap.add_argument("--no-torch", action="store_true", help="with --rccl-report: do not import torch first (a C / C++ / Rust host)")
    args = ap.parse_args()
    if args.rank_worker:
        return dry_run_worker()
    if args.rccl_report:
        rep = rccl_report(import_torch=not args.no_torch)
        print(json.dumps(rep))
        return 1 if any(isinstance(v, str) and v.startswith("failed") for v in rep.values()) else 0
    t0 = time.time()
    if args.dry_run:
        rep = dry_run(args.gpus or 2)
    else:
        import torch
        n_vis = torch.cuda.device_count()
        if n_vis < 1:
            print(json.dumps({"failed": "no GPU visible (use --dry-run on a CPU box)"}))
            return 1
        n = args.gpus or n_vis
        devices = list(range(min(n, n_vis)))
        rep = {"devices_visible": n_vis, "devices_checked": devices}
        rep.update(rccl_report())
        per_dev = {}
        for d in devices:
            try:
                rep[f"device_{d}"], per_dev[d] = check_device(d)
            except AssertionError as e:
                rep[f"device_{d}"] = "failed: %s" % e
        if len(per_dev) == len(devices):
            try:
                rep.update(check_group(devices, per_dev))
            except AssertionError as e:
                rep["group"] = "failed: %s" % e
        if not args.skip_bench:
            rep["bench_gpus_%d" % len(devices)], line = run_bench(len(devices))
            if line:
                rep["bench_line"] = {k: line.get(k) for k in ("value", "unit", "n_gpus", "ms_per_step", "ranks")}
        if len(devices) == 1:
            rep["skipped"] = ("one GPU visible: the RCCL all-gather across distinct devices, devices other than 0 and ranks.min / ranks.max "
                              "(N > 1) were NOT exercised")
    rep["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(rep))
    return 1 if any(isinstance(v, str) and v.startswith("failed") for v in rep.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
