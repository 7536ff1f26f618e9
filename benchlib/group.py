"""`--inproc`: one process driving N devices through the library's own batch split (mldsa_group_*)."""
import json
import time

import numpy as np
import torch

from .cpu import _shake
from .workloads import WholeOp


def run_inproc_resident(args):
    """`--inproc --resident`: the contract's HBM-resident `value` from ONE process.  One mldsa_group over N devices (devices reused
    round-robin when fewer GPUs are visible: a functional run, labelled), slice i of the job resident on device i -- expanded keys,
    messages, signatures -- and one mldsa_verify_group / mldsa_sign_group call per step, enqueued without waiting; the timed
    region ends with mldsa_group_sync.  No collective on the data path; the verdict all-gather is timed separately."""
    from fips204_amd.hotpath import HotPath
    from fips204_amd.ml_dsa import MlDsaGroup
    kind = "sign" if args.workload.startswith("sign") else "verify"
    digits = "".join(ch for ch in args.workload if ch.isdigit())
    pset = int(digits) if digits in ("44", "65", "87") else 65
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py --inproc --resident: no GPU visible")
    devices = [i % n_dev for i in range(args.gpus)]
    per_gpu = args.batch or 65536
    N = args.gpus
    g = MlDsaGroup(pset, devices)
    wls, slices = [], []
    for i, d in enumerate(devices):
        with torch.cuda.device(d):
            hp_i = HotPath.from_handle(g.ctx(i), d)
            wl = WholeOp(hp_i, pset, kind, per_gpu, i, world=N)
            if i == 0:
                wl.check()
            wls.append(wl)
            common = dict(msg_buf=wl.msg_buf, msg_off=wl.msg_off, key_idx=wl.key_idx, n_ops=per_gpu)
            if kind == "verify":
                slices.append(dict(common, pks=wl.pks, sigs=wl.sigs, ok=wl.ok))
            else:
                slices.append(dict(common, sks=wl.sks, rnd=wl.rnd, sigs=wl.sigs, status=wl.status))
    step = (lambda: g.verify_group(slices, wait=False)) if kind == "verify" else (lambda: g.sign_group(slices, wait=False))
    for _ in range(args.warmup):
        step()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    g.sync()
    dt = time.perf_counter() - t0
    for wl in wls:
        with torch.cuda.device(wl.ok.device):
            if kind == "verify":
                assert bool(wl.ok.all()), "a valid signature was rejected"
            else:
                assert int(wl.status.abs().max()) == 0, "an op was refused or left unfinished"
    line = {"metric": wls[0].metric, "value": per_gpu * N * args.steps / dt, "unit": wls[0].unit, "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": wls[0].name + f"; ONE process, mldsa_{kind}_group over {N} contexts (one worker thread each), slices resident per device",
                       "batch_per_gpu": per_gpu, "parallelism": f"in-library batch-split x{N}, device-resident", "devices": devices,
                       "distinct_gpus": len(set(devices))},
            "note": ("devices reused round-robin: a functional run of the N-context device-resident path on fewer GPUs, NOT a scaling measurement"
                     if len(set(devices)) < N else "one context per GPU")}
    if kind == "verify":  # the verdict bytes of every slice into every device's buffer, outside `value`
        per = per_gpu
        bufs = []
        for i, wl in enumerate(wls):
            with torch.cuda.device(wl.ok.device):
                b = torch.zeros(per * N, dtype=torch.uint8, device=wl.ok.device)
                b[i * per:(i + 1) * per] = wl.ok
                bufs.append(b)
        for d in set(devices):
            torch.cuda.synchronize(d)
        g.allgather(bufs, per * N, use_rccl=-1)
        t0 = time.perf_counter()
        g.allgather(bufs, per * N, use_rccl=-1)
        ms = (time.perf_counter() - t0) * 1e3
        assert all(bool(b.all()) for b in bufs)
        line["verdict_gather"] = {"ms": ms, "verdicts": per * N, "collective": "mldsa_group_allgather (RCCL ncclAllGather on distinct devices, device-to-device copies otherwise)"}
    print(json.dumps(line), flush=True)
    del wls, slices
    g.close()


def run_inproc(args):
    """`--inproc`: the C ABI's in-library multi-GPU path.  One process, one mldsa_group over N devices (one context + one worker
    thread each; when fewer than N GPUs are visible the devices are reused round-robin -- a functional run, labelled as such),
    the host-memory entry points on page-locked buffers, contiguous ceil(B / N) slices, no collective.  Prints one JSON line whose
    value is host-fed (PCIe-inclusive) throughput: beside the contract's `value`, never instead of it."""
    from fips204_amd.ml_dsa import MlDsaGroup
    kind = "sign" if args.workload.startswith("sign") else "verify"
    digits = "".join(ch for ch in args.workload if ch.isdigit())
    pset = int(digits) if digits in ("44", "65", "87") else 65
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py --inproc: no GPU visible")
    devices = [i % n_dev for i in range(args.gpus)]
    per_gpu = args.batch or 65536
    n = per_gpu * args.gpus
    g = MlDsaGroup(pset, devices)
    nk = min(n, 1024)

    def pin(a):
        t = torch.empty(max(a.nbytes, 1), dtype=torch.uint8, pin_memory=True)
        v = t.numpy()[:a.nbytes].view(a.dtype).reshape(a.shape)
        v[...] = a
        return t, v
    keep = []
    def P(a):
        t, v = pin(np.ascontiguousarray(a)); keep.append(t); return v
    xi = P(np.frombuffer(b"".join(_shake(b"mldsa-bench-key" + bytes([pset]), i, 4) for i in range(nk)), dtype=np.uint8).reshape(nk, 32))
    pk, sk = g.keygen_host(xi)
    pk, sk = P(pk), P(sk)
    msgs = P(np.frombuffer(b"".join(_shake(b"mldsa-bench-msg", i, 8) for i in range(n)), dtype=np.uint8))
    moff = P(np.arange(n + 1, dtype=np.uint64) * 32)
    rnd = P(np.frombuffer(b"".join(_shake(b"mldsa-bench-rnd", i, 8) for i in range(n)), dtype=np.uint8).reshape(n, 32))
    kidx = P((np.arange(n) % nk).astype(np.uint32))
    sig, st, ok = P(np.zeros((n, g.SIG_LEN), np.uint8)), P(np.zeros(n, np.int32)), P(np.zeros(n, np.uint8))
    g.sign_host(sk, (msgs, moff), rnd, key_idx=kidx, out=(sig, st))
    step = (lambda: g.sign_host(sk, (msgs, moff), rnd, key_idx=kidx, out=(sig, st))) if kind == "sign" else \
           (lambda: g.verify_host(pk, (msgs, moff), sig, key_idx=kidx, out=ok))
    # parity of a sample against the oracle, and the whole batch against the verifier
    from oracle import oracle as orc
    for i in (0, n // 2, n - 1):
        sk_o = orc.sk_try_from_bytes(pset, sk[kidx[i]].tobytes())
        assert sig[i].tobytes() == orc.sign_internal(pset, sk_o, msgs[32 * i:32 * i + 32].tobytes(), rnd[i].tobytes(), mode=0), "group signature differs from the oracle"
    assert g.verify_host(pk, (msgs, moff), sig, key_idx=kidx, out=ok).all(), "group verify rejected a valid signature"
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    p = g.params
    up, down = (32 + 32 + 12, p.sig_len + 4) if kind == "sign" else (p.sig_len + 32 + 12, 1)
    line = {"metric": f"ML-DSA-{pset} {kind}s/sec, host-fed through the in-library group (PCIe-inclusive; not the contract's HBM-resident value)",
            "value": n * args.steps / dt, "unit": f"{kind}s/s" if kind == "sign" else "verifies/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"ml_dsa_{pset} {kind}, {per_gpu} ops per device x {args.gpus}, wire-format keys and page-locked host buffers, "
                                   "mldsa_*_host_group (one process, one worker thread and context per device)",
                       "batch_per_gpu": per_gpu, "parallelism": f"in-library batch-split x{args.gpus}", "devices": devices,
                       "distinct_gpus": len(set(devices))},
            "pcie_GBs_used": n * max(up, down) * args.steps / dt / 1e9,
            "note": ("devices reused round-robin: functional run of the N-context path on fewer GPUs, not a scaling measurement"
                     if len(set(devices)) < args.gpus else "one context per GPU")}
    print(json.dumps(line), flush=True)
    g.close()

