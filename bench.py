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
KECCAK_PEAK_GPERMS = 9.26  # measured: tools/ubench_valu.hip k_keccak, 8 waves/SIMD (profiles/r01_ubench_valu.txt)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
SETS = {44: dict(k=4, l=4, gamma1=1 << 17, tau=39), 65: dict(k=6, l=5, gamma1=1 << 19, tau=49),
        87: dict(k=8, l=7, gamma1=1 << 19, tau=60)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("MLDSA_BENCH_WORKLOAD", "verify65"))
    ap.add_argument("--batch", type=int, default=0, help="ops per GPU (0 = the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra config[1] / config[2] objects of the default run")
    return ap.parse_args()


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MLDSA_BENCH_FORCE_DIST=1: take the RCCL path even with one rank (to exercise it on a 1-GPU box)
    global _DIST
    _DIST = world > 1 or (os.environ.get("MLDSA_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if _DIST:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    return rank, local_rank, world


_DIST = False


def barrier(world):
    if _DIST:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world):
    if not _DIST:
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
        self._calls = None
        self.kernel = f"k_verify_arith<{self.k},{self.l}>"
        self.name = f"ml_dsa_{pset} batch={batch} verify arithmetic (NTT/INTT + pointwise kernels only, inputs resident in HBM)"
        self.unit = "verifies/s"
        self.metric = f"ML-DSA-{pset} verify-arithmetic units/sec per GPU (batched); % HBM roofline"
        self.dtype = "int32"

    def step(self, i):
        # the kernel runs ~25 us: go through a pre-bound C call so the host keeps the stream's queue full
        if self._calls is None:
            import ctypes as C
            import functools
            lib, h = self.hp.lib, self.hp._h
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._calls = [functools.partial(lib.mldsa_verify_arith, h, self.pset, C.c_void_p(a.data_ptr()), C.c_void_p(z.data_ptr()),
                                             C.c_void_p(c.data_ptr()), C.c_void_p(t1.data_ptr()), C.c_void_p(self.out.data_ptr()),
                                             self.batch, stream) for a, z, c, t1 in self.inputs]
        rc = self._calls[i % self.n_sets]()
        if rc != 0:
            raise RuntimeError(f"mldsa_verify_arith failed: {rc}")

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


def usable_cores():
    """Host threads this process may actually run concurrently: the cgroup CPU quota when there is
    one (the GPU box exposes 256 logical CPUs but caps the container), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


class SeamKernel:
    """One seam-level primitive of the hot path on its own (SURVEY.md 8d per-kernel figures):
    ntt / inv_ntt (2 048 B per polynomial), mat_vec_mul65 (41 984 B per op), expand_a65 (30 752 B
    written per op; Keccak-bound), expand_mask65 ((66 + 1024) * L per op; Keccak-bound)."""

    def __init__(self, hp, name, batch, rank):
        self.hp, self.name_, self.rank = hp, name, rank
        g = torch.Generator(device="cuda").manual_seed(204 + rank)
        self.dtype = "int32"
        self.unit = "polys/s"
        if name in ("ntt", "inv_ntt"):
            self.batch = batch or 393216  # config-3 size: 65 536 ops x 6 polys
            self.bytes_per_op = 2048
            self.n_sets = 3               # 3 x 403 MB in + out buffers: nothing survives in the 256 MiB Infinity Cache
            self.inputs = [torch.randint(0, Q, (self.batch, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(self.n_sets)]
            self.out = torch.empty_like(self.inputs[0])
            fn = hp.ntt if name == "ntt" else hp.inv_ntt
            self.call = lambda i: fn(self.inputs[i % self.n_sets], out=self.out)
            self.kernel = "k_" + name
        elif name == "mat_vec_mul65":
            self.batch = batch or 16384
            self.unit = "ops/s"
            self.bytes_per_op = 1024 * (30 + 5 + 6)
            self.n_sets = 2
            self.inputs = [(torch.randint(0, Q, (self.batch, 6, 5, 256), dtype=torch.int32, device="cuda", generator=g),
                            torch.randint(0, Q, (self.batch, 5, 256), dtype=torch.int32, device="cuda", generator=g)) for _ in range(self.n_sets)]
            self.call = lambda i: hp.mat_vec_mul(65, *self.inputs[i % self.n_sets])
            self.kernel = "k_mat_vec_mul<6,5>"
        elif name == "expand_a65":
            self.batch = batch or 65536
            self.unit = "ops/s"
            self.bytes_per_op = 32 + 1024 * 30
            self.n_sets = 1
            self.rho = torch.randint(0, 256, (self.batch, 32), dtype=torch.uint8, device="cuda", generator=g)
            self.call = lambda i: hp.expand_a(65, self.rho)
            self.kernel = "k_expand_a<6,5>"
        elif name == "expand_mask65":
            self.batch = batch or 65536
            self.unit = "ops/s"
            self.bytes_per_op = (66 + 1024) * 5
            self.n_sets = 1
            self.rho = torch.randint(0, 256, (self.batch, 64), dtype=torch.uint8, device="cuda", generator=g)
            self.kappa = torch.zeros(self.batch, dtype=torch.int16, device="cuda")
            self.call = lambda i: hp.expand_mask(65, self.rho, self.kappa)
            self.kernel = "k_expand_mask<19>"
        elif name in ("keygen44", "keygen65", "keygen87"):
            # KeyGen::keygen_from_seed (ml_dsa.rs:31-150) as one pipeline: xi -> pk, sk wire bytes
            from fips204_amd.ml_dsa import MlDsa
            pset = int(name[-2:])
            ml = MlDsa(pset, hotpath=hp)
            self.batch = batch or 65536
            self.unit = "keys/s"
            self.bytes_per_op = 32 + ml.PK_LEN + ml.SK_LEN  # 3 904 / 6 016 / 7 520
            self.n_sets = 1
            self.xi = torch.randint(0, 256, (self.batch, 32), dtype=torch.uint8, device="cuda", generator=g)
            self.call = lambda i: ml.keygen_from_seed(self.xi)
            self.kernel = "keygen_batch (ExpandA + ExpandS + NTT/mat-vec + encode pipeline)"
        else:
            raise SystemExit(f"unknown seam kernel {name!r}")
        self.name = f"{name} batch={self.batch} (seam-level primitive, inputs resident in HBM)"
        self.metric = f"{name} {self.unit} per GPU (batched); % HBM roofline"

    def step(self, i):
        self.call(i)

    def kernel_launches_per_step(self):
        return 1

    def check(self):
        pass  # parity of every seam primitive is covered by tests/test_gpu_poly.py / test_gpu_samplers.py

    def cpu_baseline(self, budget_s=0):
        return None


def _shake(tag, i, width):
    import hashlib
    return hashlib.shake_256(tag + i.to_bytes(width, "little")).digest(32)


class WholeOp:
    """Whole ML-DSA verify or sign on wire-format inputs resident in HBM:
    n_keys = min(B, 1024) keys from xi_i = SHAKE256("mldsa-bench-key" | set | i_le32), round-robin;
    32-byte messages m_i = SHAKE256("mldsa-bench-msg" | i_le64); hedged rnd_i =
    SHAKE256("mldsa-bench-rnd" | i_le64); empty ctx, external interface.  A_hat is re-derived
    from rho inside every op (no cross-op reuse), like the reference (ml_dsa.rs:181, 406)."""

    def __init__(self, hp, pset, kind, batch, rank, cached_a=False):
        from fips204_amd.ml_dsa import MlDsa, _cat_with_offsets
        self.cached_a = cached_a
        self.hp, self.pset, self.kind, self.batch, self.rank = hp, pset, kind, batch, rank
        self.ml = ml = MlDsa(pset, hotpath=hp)
        p = ml.params
        self.k, self.l = p.k, p.l
        n_keys = min(batch, 1024)
        base = rank * batch  # global op index of this rank's first op (weak scaling: distinct data per rank)
        xi = [_shake(b"mldsa-bench-key" + bytes([pset]), base + i, 4) for i in range(n_keys)]
        self.msgs = [_shake(b"mldsa-bench-msg", base + i, 8) for i in range(batch)]
        self.rnd_host = [_shake(b"mldsa-bench-rnd", base + i, 8) for i in range(batch)]
        self.pk_bytes, self.sk_bytes = ml.keygen_from_seed(xi)
        self.pks = ml.public_keys_from_bytes(self.pk_bytes)
        self.sks = ml.private_keys_from_bytes(self.sk_bytes)
        self.key_idx_host = np.arange(batch, dtype=np.uint32) % n_keys
        self.key_idx = torch.from_numpy(self.key_idx_host.view(np.int32)).cuda()
        self.msg_buf, self.msg_off = _cat_with_offsets(self.msgs, ml.device)
        self.rnd = torch.frombuffer(bytearray(b"".join(self.rnd_host)), dtype=torch.uint8).cuda().view(batch, 32)
        self.sigs = torch.empty((batch, ml.SIG_LEN), dtype=torch.uint8, device="cuda")
        self.ok = torch.zeros(batch, dtype=torch.uint8, device="cuda")
        self.status = torch.zeros(batch, dtype=torch.int32, device="cuda")
        ml.sign_device(self.sks, self.msg_buf, self.msg_off, self.rnd, self.sigs, batch, key_idx=self.key_idx, status=self.status)
        torch.cuda.synchronize()
        self.n_sets = 1
        kl = self.k * self.l
        if kind == "verify":
            self.bytes_per_op = p.pk_len + p.sig_len + 32 + 1          # whole verify
            self.unit = "verifies/s"
            self.metric = f"ML-DSA-{pset} verifies/sec per GPU (batched); % HBM roofline"
        else:
            self.bytes_per_op = p.sk_len + 32 + 32 + p.sig_len          # whole sign
            self.unit = "signs/s"
            self.metric = f"ML-DSA-{pset} signs/sec per GPU (batched); % HBM roofline"
        # algorithmic bytes per unit of each stage's kernel (DESIGN.md "Kernels")
        self.stage_bytes = {
            "expand_a": 32 + 1024 * kl,
            "verify_arith": 1024 * (kl + self.l + 1 + 2 * self.k),
            "sign_w": 1024 * (kl + self.l + self.k),
            "expand_mask": 66 * self.l + 1024 * self.l,
            # A_hat + signature bytes + c + t1 row block + hint masks in, w1 bytes out
            "verify_main": 1024 * (kl + 1 + self.k) + p.sig_len + 32 * self.k + p.w1_len,
        }
        # Keccak-f[1600] permutations per unit of the SHAKE-bound stages (5 SHAKE128 blocks per A_hat
        # polynomial, 5 SHAKE256 blocks per mask polynomial): their ceiling is integer-ALU issue
        self.stage_perms = {"expand_a": 5 * kl, "expand_mask": 5 * self.l}
        if cached_a:
            # the n_keys A_hat tables (n_keys * K * L KiB, 30 MB for 1 024 ML-DSA-65 keys) are re-read from
            # L2 / Infinity Cache, not from HBM: they are not algorithmic HBM bytes of these workloads
            self.stage_bytes["verify_main"] -= 1024 * kl
            self.stage_bytes["sign_w"] -= 1024 * kl
        self.name = (f"ml_dsa_{pset} batch={batch} whole {kind} on FIPS 204 wire formats, "
                     + (f"A_hat KEPT WITH THE {n_keys} KEYS (no per-op ExpandA: not the reference's per-op cost, reported separately)"
                        if cached_a else "GPU ExpandA")
                     + ("/ExpandMask + rejection-loop re-batch" if kind == "sign" else "")
                     + ", 32-byte messages, inputs resident in HBM")
        self.a_hat = ml.expand_a_for_keys(self.pks) if cached_a else None
        self.dtype = "int32"
        self.kernel = None

    def step(self, i):
        if self.kind == "verify":
            self.ml.verify_device(self.pks, self.msg_buf, self.msg_off, self.sigs, self.ok, self.batch, key_idx=self.key_idx,
                                  a_hat=self.a_hat)
        else:
            self.ml.sign_device(self.sks, self.msg_buf, self.msg_off, self.rnd, self.sigs, self.batch, key_idx=self.key_idx,
                                status=self.status, a_hat=self.a_hat)

    def kernel_launches_per_step(self):
        return 1

    def _oracle_keys(self, n):
        from oracle import oracle as orc
        pkb, skb = self.pk_bytes.cpu().numpy(), self.sk_bytes.cpu().numpy()
        pk = [orc.pk_try_from_bytes(self.pset, pkb[i].tobytes()) for i in range(n)]
        sk = [orc.sk_try_from_bytes(self.pset, skb[i].tobytes()) for i in range(n)]
        return pk, sk

    def check(self):
        from oracle import oracle as orc
        n = min(8, self.batch)
        pk, sk = self._oracle_keys(min(n, self.pk_bytes.shape[0]))
        if self.cached_a and self.kind == "sign":
            self.sigs.zero_()
            self.step(0)  # the signatures checked below come from the cached-A_hat entry point
            torch.cuda.synchronize()
        sig = self.sigs[:n].cpu().numpy()
        for i in range(n):
            ki = int(self.key_idx_host[i])
            want = orc.sign_internal(self.pset, sk[ki], self.msgs[i], self.rnd_host[i], mode=0)
            assert sig[i].tobytes() == want, "GPU signature differs from the oracle"
            assert orc.verify_internal(self.pset, pk[ki], self.msgs[i], want, mode=0)
        self.ml.verify_device(self.pks, self.msg_buf, self.msg_off, self.sigs, self.ok, self.batch, key_idx=self.key_idx,
                              a_hat=self.a_hat)
        torch.cuda.synchronize()
        assert bool(self.ok.all()), "GPU verify rejected a GPU-made signature"

    def cpu_baseline(self, budget_s=10.0):
        """The KAT-pinned oracle (C, gcc -O3 -march=native) on this box's host cores: the same
        synthetic ops dealt round-robin to one pthread per logical core (oracle/mldsa_oracle.c,
        orc_*_batch_mt); the single-thread rate is reported next to it."""
        from oracle import oracle as orc
        n_ops = min(4096, self.batch)
        pk, sk = self._oracle_keys(self.pk_bytes.shape[0])
        sig = [x.tobytes() for x in self.sigs[:n_ops].cpu().numpy()]
        kidx = self.key_idx_host[:n_ops]
        msgs, rnds = self.msgs[:n_ops], self.rnd_host[:n_ops]

        def run(n, threads, repeat):
            t0 = time.perf_counter()
            if self.kind == "verify":
                ok = orc.verify_batch_mt(self.pset, pk, kidx[:n], msgs[:n], sig[:n], threads, repeat)
                assert ok.all(), "oracle rejected a GPU-made signature"
            else:
                out = orc.sign_batch_mt(self.pset, sk, kidx[:n], msgs[:n], rnds[:n], threads, repeat)
                assert out[0] == sig[0], "oracle signature differs from the GPU signature"
            return n * repeat / (time.perf_counter() - t0)

        one_rate = run(min(256, n_ops), 1, 1)
        one_rate = run(min(n_ops, max(64, int(one_rate * 1.5))), 1, 1)       # ~1.5 s single thread
        cores = usable_cores()
        run(n_ops, cores, 1)
        pilot = run(n_ops, cores, max(1, int(one_rate * cores * 1.0 / n_ops)))  # ~1 s pilot at the sustained rate
        repeat = max(1, int(pilot * budget_s / n_ops))
        t0 = time.perf_counter()
        rate = run(n_ops, cores, repeat)
        dt = time.perf_counter() - t0
        return dict(value=rate, unit=self.unit, cores=cores, kind="port", single_thread_value=one_rate,
                    sample=f"{n_ops * repeat} whole {self.kind} ops (the batch's first {n_ops} ops x {repeat} passes) on {cores} "
                           f"host threads (pthreads; = the container's CPU quota on a {os.cpu_count()}-CPU host), oracle/liboracle.so = KAT-pinned C "
                           f"restatement with per-op ExpandA, {dt:.1f} s")


class MixedStream:
    """BASELINE config[4] on one GPU: a stream of ML-DSA-44 / 65 / 87 work -- per step and parameter set
    keygen of B/8 keys, B signatures under those keys, B verifications -- issued back to back on one
    context.  value = (keys + signatures + verifications) per second."""

    def __init__(self, hp, batch, rank):
        from fips204_amd.ml_dsa import MlDsa, _cat_with_offsets
        self.hp, self.batch, self.rank = hp, batch or 16384, rank
        self.unit = "ops/s"
        self.dtype = "int32"
        self.n_sets = 1
        self.kernel = "keygen + sign + verify pipelines of the three parameter sets"
        g = torch.Generator(device="cuda").manual_seed(4465 + rank)
        B, nk = self.batch, max(1, self.batch // 8)
        self.sets = []
        for pset in (44, 65, 87):
            ml = MlDsa(pset, hotpath=hp)
            xi = torch.randint(0, 256, (nk, 32), dtype=torch.uint8, device="cuda", generator=g)
            msgs = [_shake(b"mldsa-bench-mixed" + bytes([pset]), rank * B + i, 8) for i in range(B)]
            mb, mo = _cat_with_offsets(msgs, ml.device)
            rnd = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device="cuda", generator=g)
            kidx = (torch.arange(B, device="cuda") % nk).to(torch.int32)
            sig = torch.empty((B, ml.SIG_LEN), dtype=torch.uint8, device="cuda")
            ok = torch.zeros(B, dtype=torch.uint8, device="cuda")
            self.sets.append(dict(ml=ml, xi=xi, mb=mb, mo=mo, rnd=rnd, kidx=kidx, sig=sig, ok=ok, msgs=msgs))
        self.ops_per_step = 3 * (nk + 2 * B)
        p = [s["ml"] for s in self.sets]
        self.bytes_per_op = sum(nk * (32 + m.PK_LEN + m.SK_LEN) + B * (m.SK_LEN + 64 + m.SIG_LEN) + B * (m.PK_LEN + m.SIG_LEN + 33)
                                for m in p) / self.ops_per_step
        self.name = (f"mixed ml_dsa_44/65/87 stream: per set keygen x{nk} + sign x{B} + verify x{B} per step, one context, "
                     "wire formats resident in HBM")
        self.metric = "mixed ML-DSA-44/65/87 keygen+sign+verify ops/sec per GPU (batched); % HBM roofline"

    def step(self, i):
        for s in self.sets:
            ml = s["ml"]
            pk, sk = ml.keygen_from_seed(s["xi"])
            pks, sks = ml.public_keys_from_bytes(pk), ml.private_keys_from_bytes(sk)
            ml.sign_device(sks, s["mb"], s["mo"], s["rnd"], s["sig"], self.batch, key_idx=s["kidx"])
            ml.verify_device(pks, s["mb"], s["mo"], s["sig"], s["ok"], self.batch, key_idx=s["kidx"])
            s["pk"], s["sk"] = pk, sk

    def kernel_launches_per_step(self):
        return 1

    def check(self):
        from oracle import oracle as orc
        self.step(0)
        torch.cuda.synchronize()
        for s in self.sets:
            ml = s["ml"]
            assert bool(s["ok"].all()), "mixed stream: a GPU signature did not verify"
            sk0 = orc.sk_try_from_bytes(ml.pset, s["sk"][0].cpu().numpy().tobytes())
            want = orc.sign_internal(ml.pset, sk0, s["msgs"][0], s["rnd"][0].cpu().numpy().tobytes(), mode=0)
            assert s["sig"][0].cpu().numpy().tobytes() == want, "mixed stream: GPU signature differs from the oracle"

    def cpu_baseline(self, budget_s=0):
        return None


def make_workload(name, hp, batch, rank):
    if name == "mixed":
        return MixedStream(hp, batch, rank)
    if name.startswith("verify_arith"):
        pset = int(name[len("verify_arith"):])
        return VerifyArith(hp, pset, batch or 4096, rank)
    for kind in ("verify", "sign"):
        core = name[:-len("_cached_a")] if name.endswith("_cached_a") else name
        if core.startswith(kind) and core[len(kind):].isdigit():
            return WholeOp(hp, int(core[len(kind):]), kind, batch or 65536, rank, cached_a=name.endswith("_cached_a"))
    if name in ("ntt", "inv_ntt", "mat_vec_mul65", "expand_a65", "expand_mask65", "keygen44", "keygen65", "keygen87"):
        return SeamKernel(hp, name, batch, rank)
    raise SystemExit(f"unknown workload {name!r}")


def run_one(args, hp, rank, world, name, steps, warmup, cpu_baseline):
    wl = make_workload(name, hp, args.batch if name == args.workload else 0, rank)
    if rank == 0:
        wl.check()
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize()
    whole = isinstance(wl, WholeOp)
    if whole:
        hp.profile_enable(True)  # event pairs around every kernel launch, resolved after the timed region

    # timed region: exactly K steps, barrier + synchronize on both sides; HIP events on the launch
    # stream give the dominant kernel's average duration for the roofline
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        wl.step(warmup + i)
    ev1.record()
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world)

    # single-kernel workloads: average launch duration over the back-to-back launches of the timed region
    # (HIP events on the launch stream, first launch -> last completion)
    kern_ms = ev0.elapsed_time(ev1) / steps / wl.kernel_launches_per_step()
    total_ops = getattr(wl, "ops_per_step", wl.batch) * world * steps
    value = total_ops / dt
    stages = None
    if whole:
        stages = hp.profile_report()
        hp.profile_enable(False)
    if rank != 0:
        return None

    alg_bytes = wl.bytes_per_op * getattr(wl, "ops_per_step", wl.batch)
    if whole:
        # dominant kernel = the stage with the largest share of device time; its average launch
        # duration comes from the event pairs recorded inside the timed region
        dom = max((k for k in stages if k in wl.stage_bytes), key=lambda k: stages[k]["ms"])
        kern_ms = stages[dom]["ms"] / stages[dom]["calls"]
        per_round = dom in ("expand_mask", "sign_w", "sign_tail")
        units_total = stages["_sign_slots"]["calls"] if (wl.kind == "sign" and per_round) else wl.batch * steps
        alg_bytes = wl.stage_bytes[dom] * units_total / stages[dom]["calls"]
        wl.kernel = "k_" + dom
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_{name}.json")
    if os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
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
    if whole:
        slots = stages.pop("_sign_slots", None)
        total_ms = sum(v["ms"] for v in stages.values())
        line["stage_ms_per_step"] = {k: round(v["ms"] / steps, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"])}
        if slots:
            line["sign_iterations_per_signature"] = slots["calls"] / (wl.batch * steps)
        line["device_busy_frac"] = total_ms / (dt * 1e3)
        perms = {"verify": {44: 89, 65: 159, 87: 291}, "sign": {44: 201, 65: 320, 87: 455}}[wl.kind][wl.pset]
        line["roofline"]["note"] = ("whole ops are integer-ALU-bound (Keccak-f[1600]), not HBM-bound: "
                                    f"~{perms} permutations per op; the HBM-bound kernel of the path is reported under "
                                    "also.verify_arith44 (BASELINE config 2)")
        line["roofline"]["note"] += ("; algorithmic bytes are SURVEY 8d's int32-polynomial counts: the pipelines keep A_hat as 24-bit "
                                     "fields and candidates of one op share its rows through L2, so the PMC traffic of "
                                     "sign_w / verify_main is below them")
        line["keccak_permutations_per_s"] = perms * value / world
        # every modelled stage against the ceiling that bounds it: HBM peak for the polynomial-streaming
        # kernels, the measured Keccak-f[1600] issue ceiling (tools/ubench_valu.hip k_keccak at 8 waves/SIMD,
        # profiles/r01_ubench_valu.txt) for the SHAKE-bound samplers
        by_stage = {}
        n_slots = slots["calls"] if slots else None
        for st_name, st in stages.items():
            per_round = st_name in ("expand_mask", "sign_w", "sign_tail")
            units = n_slots if (wl.kind == "sign" and per_round) else wl.batch * steps
            if st_name in wl.stage_perms:
                ach = wl.stage_perms[st_name] * units / (st["ms"] * 1e-3) / 1e9
                by_stage[st_name] = {"bound": "valu", "achieved": ach, "peak": KECCAK_PEAK_GPERMS,
                                     "unit": "G Keccak-f[1600]/s", "frac": ach / KECCAK_PEAK_GPERMS}
            elif st_name in wl.stage_bytes:
                ach = wl.stage_bytes[st_name] * units / (st["ms"] * 1e-3) / 1e9
                by_stage[st_name] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ach / HBM_PEAK_GBS}
        line["roofline_by_stage"] = by_stage
        line["whole_op_hbm"] = {"algorithmic_bytes_per_op": wl.bytes_per_op,
                                "achieved_GBs": wl.bytes_per_op * value / world / 1e9,
                                "frac_of_peak": wl.bytes_per_op * value / world / 1e9 / HBM_PEAK_GBS}
    if world == 1 and cpu_baseline:
        cb = wl.cpu_baseline()
        if cb:
            line["cpu_baseline"] = cb
    del wl
    torch.cuda.empty_cache()
    return line


def main():
    args = parse()
    rank, local_rank, world = dist_setup(args.gpus)
    from fips204_amd.hotpath import HotPath
    hp = HotPath(local_rank)
    line = run_one(args, hp, rank, world, args.workload, args.steps, args.warmup, not args.no_cpu_baseline)
    # the default single-GPU run also carries the other two BASELINE configs as extra objects
    # (same JSON line): config[1] = the HBM-roofline kernel, config[2] = whole sign
    if world == 1 and args.workload == "verify65" and not args.no_extras:
        also = {}
        for name, st, wu in (("verify_arith44", 50, 5), ("sign65", 3, 1)):
            sub = run_one(args, hp, rank, world, name, st, wu, False)
            also[name] = {k: sub[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline") if k in sub}
            for k in ("stage_ms_per_step", "sign_iterations_per_signature"):
                if k in sub:
                    also[name][k] = sub[k]
        line["also"] = also
    if rank == 0:
        print(json.dumps(line))
    if _DIST:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
