#!/usr/bin/env python3
"""bench.py -- throughput of the batched ML-DSA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  For N > 1 the driver launches one rank per GPU (torch.distributed.run);
the batch is sharded per rank with no data-path collective (independent ops),
so scaling is "weak": every rank processes its own full-size batch.

Workloads (BASELINE.json configs):
  verify_arith44  config[1]: ml_dsa_44, batch 4096, NTT/INTT + pointwise kernels only
                  (the fused verify-arithmetic unit; HBM-roofline kernel)
  verify65        the metric's headline: whole ML-DSA-65 verifies/s, batch 65536
  sign65          config[2]: whole ML-DSA-65 signs/s, batch 65536

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
"roofline" (dominant kernel, HIP-event timed inside the timed region's stream) and
"cpu_baseline" (the KAT-pinned CPU oracle timed on this box's host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q = 8380417
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
SETS = {44: dict(k=4, l=4, gamma1=1 << 17, tau=39), 65: dict(k=6, l=5, gamma1=1 << 19, tau=49),
        87: dict(k=8, l=7, gamma1=1 << 19, tau=60)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("MLDSA_BENCH_WORKLOAD", "verify_arith44"))
    ap.add_argument("--batch", type=int, default=0, help="ops per GPU (0 = the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    return rank, local_rank, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------- workloads
class VerifyArith:
    """BASELINE config[1]: w' = INTT(A_hat * NTT(z) - NTT(c) o t1_hat) for every op; inputs
    i.i.d. uniform polys pre-staged in HBM."""

    def __init__(self, hp, pset, batch, rank):
        self.hp, self.pset, self.batch = hp, pset, batch
        p = SETS[pset]
        self.k, self.l = p["k"], p["l"]
        self.bytes_per_op = 1024 * (self.k * self.l + self.l + 1 + 2 * self.k)
        # rotate over enough distinct input sets that nothing is served from the 256 MiB
        # Infinity Cache (a step's inputs must have been evicted before they are reused)
        in_bytes = batch * 1024 * (self.k * self.l + self.l + 1 + self.k)
        self.n_sets = max(2, int(np.ceil(640e6 / in_bytes)) + 1)
        g = torch.Generator(device="cuda").manual_seed(204 + rank)
        self.inputs = []
        for _ in range(self.n_sets):
            a = torch.randint(0, Q, (batch, self.k, self.l, 256), dtype=torch.int32, device="cuda", generator=g)
            z = torch.randint(-p["gamma1"] + 1, p["gamma1"] + 1, (batch, self.l, 256), dtype=torch.int32, device="cuda", generator=g)
            c = torch.zeros((batch, 256), dtype=torch.int32, device="cuda")
            pos = torch.rand((batch, 256), device="cuda", generator=g).argsort(dim=1)[:, :p["tau"]]
            sign = torch.randint(0, 2, (batch, p["tau"]), device="cuda", generator=g, dtype=torch.int32) * 2 - 1
            c.scatter_(1, pos, sign)
            t1 = torch.randint(0, Q, (batch, self.k, 256), dtype=torch.int32, device="cuda", generator=g)
            self.inputs.append((a, z, c, t1))
        self.out = torch.empty((batch, self.k, 256), dtype=torch.int32, device="cuda")
        self.kernel = f"k_verify_arith<{self.k},{self.l}>"
        self.name = f"ml_dsa_{pset} batch={batch} verify arithmetic (NTT/INTT + pointwise kernels only, inputs resident in HBM)"
        self.unit = "verifies/s"
        self.metric = f"ML-DSA-{pset} verify-arithmetic units/sec per GPU (batched); % HBM roofline"
        self.dtype = "int32"

    def step(self, i):
        a, z, c, t1 = self.inputs[i % self.n_sets]
        self.hp.verify_arith(self.pset, a, z, c, t1, out=self.out)

    def kernel_launches_per_step(self):
        return 1

    def check(self):
        from oracle import oracle as orc
        a, z, c, t1 = self.inputs[0]
        n = min(16, self.batch)
        self.step(0)
        torch.cuda.synchronize()
        want = orc.verify_arith(self.k, self.l, a[:n].cpu().numpy(), z[:n].cpu().numpy(), c[:n].cpu().numpy(), t1[:n].cpu().numpy())
        assert np.array_equal(self.out[:n].cpu().numpy(), want), "bench output differs from the oracle"

    def cpu_baseline(self, budget_s=12.0):
        from oracle import oracle as orc
        a, z, c, t1 = [x[:256].cpu().numpy() for x in self.inputs[0]]
        orc.verify_arith(self.k, self.l, a[:4], z[:4], c[:4], t1[:4])
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < budget_s:
            n = min(256, self.batch)
            orc.verify_arith(self.k, self.l, a[:n], z[:n], c[:n], t1[:n])
            done += n
        dt = time.perf_counter() - t0
        return dict(value=done / dt, unit=self.unit, cores=1, kind="port",
                    sample=f"{done} verify-arithmetic units of the same synthetic batch (256-op slice repeated), "
                           f"oracle/liboracle.so single thread, {dt:.1f} s")


def make_workload(name, hp, batch, rank):
    if name.startswith("verify_arith"):
        pset = int(name[len("verify_arith"):])
        return VerifyArith(hp, pset, batch or 4096, rank)
    raise SystemExit(f"unknown workload {name!r}")


def main():
    args = parse()
    rank, local_rank, world = dist_setup(args.gpus)
    from fips204_amd.hotpath import HotPath
    hp = HotPath(local_rank)
    wl = make_workload(args.workload, hp, args.batch, rank)
    if rank == 0:
        wl.check()

    for i in range(args.warmup):
        wl.step(i)
    torch.cuda.synchronize()

    # timed region: exactly K steps, barrier + synchronize on both sides; per-step HIP events
    # on the launch stream give the dominant kernel's average duration for the roofline
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        wl.step(args.warmup + i)
        ev[i][1].record()
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world)

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / wl.kernel_launches_per_step()
    total_ops = wl.batch * world * args.steps
    value = total_ops / dt
    if rank != 0:
        return

    alg_bytes = wl.bytes_per_op * wl.batch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
    if os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
    line = {
        "metric": wl.metric, "value": value, "unit": wl.unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": {"workload": wl.name, "batch_per_gpu": wl.batch, "parallelism": f"batch-split x{world}",
                   "input_sets_rotated": wl.n_sets},
        "roofline": {"bound": "hbm", "kernel": wl.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = wl.cpu_baseline()
    print(json.dumps(line))


if __name__ == "__main__":
    main()
