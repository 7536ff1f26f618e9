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
KECCAK_PEAK_MEASURED_GPERMS = 9.26  # tools/ubench_valu.hip k_keccak at 8 waves/SIMD (profiles/r01_ubench_valu.txt): cross-check only
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
# The integer-issue ceiling of the lane-per-state Keccak-f[1600] (csrc/keccak.h), DERIVED from its instruction mix and the
# measured issue cost of each instruction class on this chip (profiles/r01_ubench_valu.txt, cycles per wave64 instruction per
# SIMD at the measured clock): one round = 70 v_bitop3_b32 (chi, theta parities) + 58 v_alignbit_b32 (rotates) + 62 v_xor_b32.
KECCAK_ROUND_MIX = {"v_bitop3_b32": (70, 4.4), "v_alignbit_b32": (58, 4.4), "v_xor_b32": (62, 2.7)}  # (count per round, cycles)
GPU_SIMDS, GPU_CLOCK_GHZ = 256 * 4, 2.4


def keccak_issue_ceiling():
    """G permutations/s if every SIMD issued nothing but Keccak rounds, with the arithmetic spelled out"""
    cyc_round = sum(n * c for n, c in KECCAK_ROUND_MIX.values())
    cyc_perm = 24 * cyc_round            # per wave = per 64 states
    peak = GPU_SIMDS * GPU_CLOCK_GHZ * 64 / cyc_perm
    return peak, {
        "instruction_mix_per_round": {k: {"count": n, "issue_cycles_per_wave64_instruction": c} for k, (n, c) in KECCAK_ROUND_MIX.items()},
        "issue_costs_source": "profiles/r01_ubench_valu.txt (tools/ubench_valu.hip, column cyc/instr@clk; v_bitop3_b32 issues like v_bfi_b32 / v_and_or_b32)",
        "cycles_per_round_per_wave": cyc_round, "rounds": 24, "cycles_per_permutation_per_wave": cyc_perm, "states_per_wave": 64,
        "simds": GPU_SIMDS, "clock_GHz": GPU_CLOCK_GHZ,
        "formula": "simds * clock_GHz * states_per_wave / cycles_per_permutation_per_wave",
        "G_permutations_per_s": peak,
        "measured_pure_keccak_kernel_G_per_s": KECCAK_PEAK_MEASURED_GPERMS,
    }


KECCAK_PEAK_GPERMS, KECCAK_PEAK_DERIVATION = keccak_issue_ceiling()
SETS = {44: dict(k=4, l=4, gamma1=1 << 17, tau=39), 65: dict(k=6, l=5, gamma1=1 << 19, tau=49),
        87: dict(k=8, l=7, gamma1=1 << 19, tau=60)}


# the reference's own published single-core figures (benches/README.md:16-26; i7-7700K @ 4.2 GHz, Rust 1.81,
# RUSTFLAGS="-C target-cpu=native" cargo bench), printed beside the CPU baseline measured here
REFERENCE_PUBLISHED = {
    "source": "/root/reference/benches/README.md:16-26 (Intel i7-7700K @ 4.20 GHz, one core, Oct 2024)",
    "keygen_us": {44: 104.89, 65: 194.80, 87: 290.24},
    "sign_us": {44: 226.32, 65: 352.89, 87: 385.05},
    "verify_us": {44: 21.016, 65: 27.996, 87: 36.468},
    "note": ("published figures of another machine, quoted for context only.  They are not mutually consistent by operation count: "
             "keygen (about 190 Keccak-f incl. ExpandA) is listed at 194.8 us, verify (159 Keccak-f incl. the same ExpandA, "
             "ml_dsa.rs:406) at 28.0 us.  The oracle timed here spends 318 us per keygen at 2.1 GHz against the published 194.8 us "
             "at 4.2-4.5 GHz."),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("MLDSA_BENCH_WORKLOAD", "verify65"))
    ap.add_argument("--batch", type=int, default=0, help="ops per GPU (0 = the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra config[1] / config[2] objects of the default run")
    ap.add_argument("--backend", default=os.environ.get("MLDSA_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL) or gloo (CPU rendezvous; lets several ranks "
                         "share one GPU for a functional check on a 1-GPU box)")
    ap.add_argument("--graphs", type=int, default=-1, help="override MLDSA_OPT_GRAPHS (hipGraph replay) of the context: 0 / 1")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic live for --workload (two child rocprofv3 counter passes); "
                                                       "the default run does this for verify65 unless --no-pmc")
    ap.add_argument("--no-pmc", action="store_true", help="never start the profiler: roofline.traffic from the file kept under profiles/")
    ap.add_argument("--resident", action="store_true",
                    help="with --inproc: slices resident per device and the device-resident group calls (mldsa_verify_group / mldsa_sign_group): "
                         "the contract's HBM-resident value from one process")
    ap.add_argument("--inproc", action="store_true",
                    help="ONE process driving --gpus N devices through the library's own batch split (mldsa_group_create + "
                         "mldsa_*_host_group): host-memory inputs, so the figure is PCIe-inclusive and is NOT the contract's `value` path")
    return ap.parse_args(argv)


def dist_setup(args):
    """One process per GPU.  Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment) or by this script's own parent (main(): --gpus N without that environment)."""
    from fips204_amd import multi_gpu
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and "RANK" in os.environ:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if world > 1 and n_dev < world and args.backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPUs visible (RCCL needs one GPU per rank; "
                         "--backend gloo shares GPUs for a functional check)")
    dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev)
    rank, _, world = multi_gpu.init_process_group(args.backend, device_index=dev)
    return rank, dev, world


def barrier(world):
    from fips204_amd import multi_gpu
    multi_gpu.barrier()


def max_over_ranks(x, world):
    from fips204_amd import multi_gpu
    if not multi_gpu.is_distributed():
        return x
    import torch.distributed as dist
    return multi_gpu.max_over_ranks(x, "cuda" if dist.get_backend() == "nccl" else "cpu")


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
        """the first ops of the very buffers the timed region uses, against the oracle (bit-exact)"""
        from oracle import oracle as orc
        nm, n = self.name_, 8
        host = lambda t: t.cpu().numpy()
        if nm in ("ntt", "inv_ntt"):
            got = self.call(0)
            torch.cuda.synchronize()
            want = (orc.ntt if nm == "ntt" else orc.inv_ntt)(host(self.inputs[0][:n]))
            got = host(got[:n]).astype(np.int64) % Q
            assert np.array_equal(got, np.asarray(want, dtype=np.int64) % Q), f"{nm}: bench output differs from the oracle"
        elif nm == "mat_vec_mul65":
            a, u = self.inputs[0]
            got = self.call(0)
            torch.cuda.synchronize()
            for i in range(n):
                want = orc.mat_vec_mul(6, 5, host(a[i]), host(u[i]))
                assert np.array_equal(host(got[i]).astype(np.int64) % Q, np.asarray(want, dtype=np.int64) % Q), "mat_vec_mul: differs from the oracle"
        elif nm == "expand_a65":
            got = self.call(0)
            torch.cuda.synchronize()
            rho = host(self.rho[:n])
            for i in range(n):
                assert np.array_equal(host(got[i]), orc.expand_a(6, 5, rho[i].tobytes())), "expand_a: differs from the oracle"
        elif nm == "expand_mask65":
            got = self.call(0)
            torch.cuda.synchronize()
            rho = host(self.rho[:n])
            for i in range(n):
                assert np.array_equal(host(got[i]), orc.expand_mask(5, 1 << 19, rho[i].tobytes(), 0)), "expand_mask: differs from the oracle"
        else:  # keygen
            pset = int(nm[-2:])
            pk, sk = self.call(0)
            torch.cuda.synchronize()
            xi = host(self.xi[:n])
            for i in range(n):
                pk_o, sk_o = orc.keygen_from_seed(pset, xi[i].tobytes())
                assert host(pk[i]).tobytes() == orc.pk_into_bytes(pset, pk_o) and host(sk[i]).tobytes() == orc.sk_into_bytes(pset, sk_o), \
                    "keygen: differs from the oracle"

    def cpu_baseline(self, budget_s=6.0):
        if not self.name_.startswith("keygen"):
            return None
        pset = int(self.name_[-2:])
        r1, rt, cores, done, dt = oracle_keygen_rates(pset, [bytes(x) for x in self.xi[:2048].cpu().numpy()], budget_s)
        return dict(value=rt, unit=self.unit, cores=cores, kind="port", single_thread_value=r1,
                    sample=f"{done} keygen_from_seed + into_bytes of the batch's first seeds on {cores} host threads (pthreads), "
                           f"oracle/liboracle.so, {dt:.1f} s")


def oracle_keygen_rates(pset, xis, budget_s):
    """(single-thread keys/s, all-core keys/s, cores, keys generated in the timed multi-thread pass, its seconds)"""
    from oracle import oracle as orc
    cores = usable_cores()
    t0 = time.perf_counter()
    orc.keygen_batch_mt(pset, xis[:64], 1)
    r1 = 64 / (time.perf_counter() - t0)
    n = min(len(xis), max(cores * 8, int(r1 * cores * 0.5)))
    t0 = time.perf_counter()
    orc.keygen_batch_mt(pset, xis[:n], cores)
    pilot = n / (time.perf_counter() - t0)
    repeat = max(1, int(pilot * budget_s / n))
    t0 = time.perf_counter()
    orc.keygen_batch_mt(pset, xis[:n], cores, repeat)
    dt = time.perf_counter() - t0
    return r1, n * repeat / dt, cores, n * repeat, dt


def _shake(tag, i, width):
    import hashlib
    return hashlib.shake_256(tag + i.to_bytes(width, "little")).digest(32)


class WholeOp:
    """Whole ML-DSA verify or sign on wire-format inputs resident in HBM:
    n_keys = min(B, 1024) keys from xi_i = SHAKE256("mldsa-bench-key" | set | i_le32), round-robin;
    32-byte messages m_i = SHAKE256("mldsa-bench-msg" | i_le64); hedged rnd_i =
    SHAKE256("mldsa-bench-rnd" | i_le64); empty ctx, external interface.  A_hat is re-derived
    from rho inside every op (no cross-op reuse), like the reference (ml_dsa.rs:181, 406)."""

    def __init__(self, hp, pset, kind, batch, rank, cached_a=False, world=1, corrupt_every=0, wire=False):
        """corrupt_every = 100: every 100th signature of a verify batch is damaged (SURVEY 8d "1 % corrupted mix for a
        correctness-under-load run").  wire = True: the "from wire bytes" unit of SURVEY 8d -- every op deserialises its key
        first (PublicKey / PrivateKey::try_from_bytes, ml_dsa.rs:477-498 / 445-469: tr = H(pk) and the key NTTs), B wire-format
        keys resident in HBM, mldsa_pk_expand / mldsa_sk_expand + the op as one timed unit."""
        from fips204_amd import multi_gpu
        from fips204_amd.ml_dsa import MlDsa, _cat_with_offsets
        self.cached_a, self.corrupt_every, self.wire = cached_a, corrupt_every, wire
        self.hp, self.pset, self.kind, self.batch, self.rank, self.world = hp, pset, kind, batch, rank, world
        self.ml = ml = MlDsa(pset, hotpath=hp)
        p = ml.params
        self.k, self.l = p.k, p.l
        n_keys = min(batch, 1024)
        # the job is batch * world ops, contiguous slices per rank (weak scaling: distinct data per rank)
        base, n_mine = multi_gpu.shard(batch * world, rank, world)
        assert n_mine == batch
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
        hp.reserve(pset, 2, batch)  # MLDSA_OP_SIGN: the largest workspace of the three pipelines
        ml.sign_device(self.sks, self.msg_buf, self.msg_off, self.rnd, self.sigs, batch, key_idx=self.key_idx, status=self.status)
        torch.cuda.synchronize()
        self.expect_ok = torch.ones(batch, dtype=torch.bool, device="cuda")
        if corrupt_every:  # one flipped bit in every corrupt_every-th signature, walking through c~ | z | hints
            assert kind == "verify"
            rows = torch.arange(corrupt_every // 3, batch, corrupt_every, device="cuda")
            cols = (rows * 2654435761 % ml.SIG_LEN)
            self.sigs[rows, cols] ^= (1 << (rows % 8)).to(torch.uint8)
            self.expect_ok[rows] = False
            self.corrupt_rows = rows.cpu().numpy()
        if wire:  # one wire-format key per op, gathered once at set-up
            kb = self.pk_bytes if kind == "verify" else self.sk_bytes
            self.key_op = kb[self.key_idx.long()].contiguous()
            self.keys_op = ml.empty_public_keys(batch) if kind == "verify" else ml.empty_private_keys(batch)
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
        # Bytes each stage's kernel is OBLIGED to move per unit (DESIGN.md "Kernels").  The pipelines keep their own
        # A_hat as 24-bit fields: 768 bytes per polynomial, written once by expand_a and read once by verify_main;
        # sign_w needs an op's A_hat once per ROUND (its speculative candidates share the rows), so its A_hat term is
        # counted per op-round, the y / w / w1 terms per candidate slot (see run_one).
        self.a_poly_bytes = 768
        self.y_poly_bytes = 32 * (18 if pset == 44 else 20)                 # the signer's y as ExpandMask squeezed it
        self.stage_bytes = {
            "expand_a": 32 + self.a_poly_bytes * kl,
            "verify_arith": 1024 * (kl + self.l + 1 + 2 * self.k),
            # per candidate slot: y in (the squeezed bytes, 32 c per polynomial), w (24-bit fields) + w1 + the risk flags out
            "sign_w": self.y_poly_bytes * self.l + 768 * self.k + p.w1_len + 1 + self.l,
            "sign_w_per_op_round": self.a_poly_bytes * kl,                  # per unfinished op and round: A_hat in
            "expand_mask": 66 * self.l + self.y_poly_bytes * self.l,
            # A_hat + signature bytes + c + t1 row block + hint masks in, w1 bytes out
            "verify_main": self.a_poly_bytes * kl + 256 + 1024 * self.k + p.sig_len + 32 * self.k + p.w1_len,   # (c: one byte per coefficient)
        }
        # Keccak-f[1600] permutations per unit of the SHAKE-bound stages (5 SHAKE128 blocks per A_hat
        # polynomial, 5 SHAKE256 blocks per mask polynomial): their ceiling is integer-ALU issue
        self.stage_perms = {"expand_a": 5 * kl, "expand_mask": 5 * self.l}
        if cached_a:
            # the n_keys A_hat tables (n_keys * K * L KiB, 30 MB for 1 024 ML-DSA-65 keys) are re-read from
            # L2 / Infinity Cache, not from HBM: they are not algorithmic HBM bytes of these workloads
            self.stage_bytes["verify_main"] -= self.a_poly_bytes * kl
            self.stage_bytes["sign_w_per_op_round"] = 0
        self.name = (f"ml_dsa_{pset} batch={batch} whole {kind} on FIPS 204 wire formats, "
                     + (f"A_hat KEPT WITH THE {n_keys} KEYS (no per-op ExpandA: not the reference's per-op cost, reported separately)"
                        if cached_a else "GPU ExpandA")
                     + ("/ExpandMask + rejection-loop re-batch" if kind == "sign" else "")
                     + ", 32-byte messages, inputs resident in HBM"
                     + (f", every {corrupt_every}th signature corrupted (one flipped bit)" if corrupt_every else "")
                     + (", FROM WIRE BYTES: try_from_bytes of the op's key (tr = H(pk) / key NTTs) inside the timed unit"
                        + (" (mldsa_verify_pk: one call)" if kind == "verify" else " (mldsa_sk_expand + mldsa_sign)") if wire else ""))
        self.a_hat = ml.expand_a_for_keys(self.pks) if cached_a else None
        self.dtype = "int32"
        self.kernel = None

    def step(self, i):
        if self.wire and self.kind == "verify":
            if os.environ.get("MLDSA_BENCH_WIRE_UNFUSED") == "1":   # the two calls a host without mldsa_verify_pk would make
                self.ml.public_keys_from_bytes(self.key_op, out=self.keys_op)
                self.ml.verify_device(self.keys_op, self.msg_buf, self.msg_off, self.sigs, self.ok, self.batch)
            else:
                self.ml.verify_pk_device(self.key_op, self.msg_buf, self.msg_off, self.sigs, self.ok, self.batch)
        elif self.wire:
            self.ml.private_keys_from_bytes(self.key_op, out=self.keys_op)
            self.ml.sign_device(self.keys_op, self.msg_buf, self.msg_off, self.rnd, self.sigs, self.batch, status=self.status)
        elif self.kind == "verify":
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
        if (self.cached_a or self.wire) and self.kind == "sign":
            self.sigs.zero_()
            self.step(0)  # the signatures checked below come from the entry points this workload times
            torch.cuda.synchronize()
        sig = self.sigs[:n].cpu().numpy()
        for i in range(n):
            ki = int(self.key_idx_host[i])
            want = orc.sign_internal(self.pset, sk[ki], self.msgs[i], self.rnd_host[i], mode=0)
            assert sig[i].tobytes() == want, "GPU signature differs from the oracle"
            assert orc.verify_internal(self.pset, pk[ki], self.msgs[i], want, mode=0)
        if self.kind == "verify":
            self.ok.zero_()
            self.step(0)
        else:
            self.ml.verify_device(self.pks, self.msg_buf, self.msg_off, self.sigs, self.ok, self.batch, key_idx=self.key_idx, a_hat=self.a_hat)
        torch.cuda.synchronize()
        assert torch.equal(self.ok.bool(), self.expect_ok), "GPU verdicts differ from the expected ones (valid signature rejected or damaged one accepted)"
        if self.corrupt_every:  # the damaged signatures (and their neighbours) through the oracle as well
            pk_all, _ = self._oracle_keys(self.pk_bytes.shape[0])
            rows = np.concatenate([self.corrupt_rows[:24], self.corrupt_rows[:24] + 1])
            sg = self.sigs[torch.from_numpy(rows).cuda()].cpu().numpy()
            ok = self.ok.cpu().numpy()
            for j, i in enumerate(rows):
                assert bool(ok[i]) == orc.verify_internal(self.pset, pk_all[int(self.key_idx_host[i])], self.msgs[i], sg[j].tobytes(), mode=0), \
                    "verdict of a damaged signature differs from the oracle's"

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

        pkb, skb = self.pk_bytes.cpu().numpy(), self.sk_bytes.cpu().numpy()
        expect = self.expect_ok[:n_ops].cpu().numpy()

        def run(n, threads, repeat):
            t0 = time.perf_counter()
            if self.kind == "verify":
                ok = (orc.verify_wire_batch_mt(self.pset, pkb, kidx[:n], msgs[:n], sig[:n], threads, repeat) if self.wire else
                      orc.verify_batch_mt(self.pset, pk, kidx[:n], msgs[:n], sig[:n], threads, repeat))
                assert np.array_equal(ok, expect[:n]), "oracle verdicts differ from the GPU's"
            else:
                out = (orc.sign_wire_batch_mt(self.pset, skb, kidx[:n], msgs[:n], rnds[:n], threads, repeat) if self.wire else
                       orc.sign_batch_mt(self.pset, sk, kidx[:n], msgs[:n], rnds[:n], threads, repeat))
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
                    sample=f"{n_ops * repeat} whole {self.kind} ops{' incl. try_from_bytes of the key' if self.wire else ''} (the batch's first {n_ops} ops x {repeat} passes) on {cores} "
                           f"host threads (pthreads; = the container's CPU quota on a {os.cpu_count()}-CPU host), oracle/liboracle.so = KAT-pinned C "
                           f"restatement with per-op ExpandA, {dt:.1f} s")


def measure_h2d_GBs(n_bytes=256 << 20):
    """PCIe host->device rate of this box from page-locked memory (what bounds the host-fed path)."""
    h = torch.empty(n_bytes, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    return 4 * n_bytes / (time.perf_counter() - t0) / 1e9


def host_fed(wl, reps=3):
    """The same batch handed over in HOST memory (mldsa_verify_host / mldsa_sign_host: wire-format keys,
    page-locked buffers, sub-batches with upload | kernels | download overlapped).  PCIe-inclusive, so it is
    reported beside `value`, never as `value`."""
    ml, p, n = wl.ml, wl.ml.params, wl.batch
    pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t).numpy()
    keys = pin(wl.pk_bytes if wl.kind == "verify" else wl.sk_bytes)
    msgs = pin(wl.msg_buf)
    moff = pin(wl.msg_off).view(np.uint64)
    kidx = pin(wl.key_idx).view(np.uint32)
    if wl.kind == "verify":
        sigs = pin(wl.sigs)
        ok_out = pin(wl.ok)
        run = lambda: ml.verify_host(keys, (msgs, moff), sigs, key_idx=kidx, out=ok_out)
        up, down = p.sig_len + 32 + 8 + 4, 1
    else:
        rnd = pin(wl.rnd)
        outs = (pin(wl.sigs), pin(wl.status))
        run = lambda: ml.sign_host(keys, (msgs, moff), rnd, key_idx=kidx, out=outs)
        up, down = 32 + 32 + 8 + 4, p.sig_len + 4
    res = run()  # warm-up: staging buffers, graphs
    if wl.kind == "verify":
        assert bool(res.all()), "host-fed verify rejected a valid signature"
    else:
        assert np.array_equal(res[:64], wl.sigs[:64].cpu().numpy()), "host-fed signatures differ from the device-resident path"
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    h2d = measure_h2d_GBs()
    bound = h2d * 1e9 / max(up, down)  # full duplex: the busier direction bounds
    return {"value": n / dt, "unit": wl.unit, "ms_per_batch": dt * 1e3, "bytes_up_per_op": up, "bytes_down_per_op": down,
            "pcie_GBs_used": n * max(up, down) / dt / 1e9, "pcie_h2d_GBs_measured": h2d, "pcie_bound_ops_per_s": bound,
            "frac_of_pcie_bound": n / dt / bound,
            "note": "inputs and outputs in page-locked host memory, wire-format keys uploaded and expanded once per call; "
                    "includes H2D + kernels + D2H + the host-side call overhead of the ctypes wrapper"}


def config5_requests(n_requests, first=0):
    """SURVEY.md 8(d') C5: request i has set = (44, 65, 87)[i mod 3]; it is a keygen if i mod 10 == 0, a signature if
    i mod 10 in {1..4}, a verification otherwise (10 % / 40 % / 50 %).  Returns {pset: {"keygen": ids, "sign": ids, "verify": ids}}
    with the global request ids of each bucket (numpy int64, ascending)."""
    i = np.arange(first, first + n_requests, dtype=np.int64)
    out = {}
    for r, pset in enumerate((44, 65, 87)):
        mine = i[i % 3 == r]
        d = mine % 10
        out[pset] = {"keygen": mine[d == 0], "sign": mine[(d >= 1) & (d <= 4)], "verify": mine[d >= 5]}
    return out


class MixedStream:
    """BASELINE config[4] on one GPU: a stream of requests, request i with parameter set (44, 65, 87)[i mod 3] and
    operation keygen / sign / verify by i mod 10 (10 % / 40 % / 50 %), inputs derived from the request id as in SURVEY 8d.
    A step = `batch` requests per parameter set (3 * batch in all), bucketed into ONE keygen, ONE sign and ONE verify call
    per set -- nine op-level calls issued back to back on one context, nothing waits for the device (mldsa_sign_async), every
    call writes the same buffers each step, so repeated shapes replay as hipGraphs where the library's policy says so.
    Signatures use a table of min(1024, .) resident keys per set, verifications check signatures made at set-up.
    value = requests per second; ops/s per class beside it."""

    def __init__(self, hp, batch, rank, world=1):
        from fips204_amd.ml_dsa import MlDsa, _cat_with_offsets
        self.hp, self.batch, self.rank = hp, batch or 65536, rank
        self.unit = "ops/s"
        self.dtype = "int32"
        self.n_sets = 1
        self.kernel = "keygen + sign + verify pipelines of the three parameter sets"
        B = self.batch
        first = rank * 3 * B
        self.req = config5_requests(3 * B, first)
        hp.set_option(9, 2)  # MLDSA_OPT_SIGN_ASYNC_EXP: see finish_steps
        self.sets = []
        self.count = {"keygen": 0, "sign": 0, "verify": 0}
        for pset in (87, 65, 44):  # largest workspace first: reserved once
            ml = MlDsa(pset, hotpath=hp)
            r = self.req[pset]
            nk = max(1, min(1024, len(r["sign"])))
            hp.reserve(pset, 2, max(1, len(r["sign"])))
            hp.reserve(pset, 3, max(1, len(r["verify"])))
            tag = bytes([pset])
            pk, sk = ml.keygen_from_seed([_shake(b"mldsa-bench-key" + tag, i, 4) for i in range(nk)])
            pks, sks = ml.public_keys_from_bytes(pk), ml.private_keys_from_bytes(sk)
            d = dict(ml=ml, pk=pk, sk=sk, pks=pks, sks=sks, nk=nk)
            # keygen requests: fresh seeds -> wire-format keys
            d["kg_xi_host"] = [_shake(b"mldsa-bench-xi" + tag, int(i), 8) for i in r["keygen"]]
            d["kg_xi"] = torch.frombuffer(bytearray(b"".join(d["kg_xi_host"]) or b"\0" * 32), dtype=torch.uint8).cuda().view(-1, 32)
            d["kg_pk"] = torch.empty((max(1, len(r["keygen"])), ml.PK_LEN), dtype=torch.uint8, device="cuda")
            d["kg_sk"] = torch.empty((max(1, len(r["keygen"])), ml.SK_LEN), dtype=torch.uint8, device="cuda")
            for kind in ("sign", "verify"):
                ids = r[kind]
                msgs = [_shake(b"mldsa-bench-msg", int(i), 8) for i in ids]
                rnd = [_shake(b"mldsa-bench-rnd", int(i), 8) for i in ids]
                mb, mo = _cat_with_offsets(msgs, ml.device)
                kidx_h = (ids % nk).astype(np.uint32)
                d[kind] = dict(n=len(ids), msgs=msgs, rnd_host=rnd, mb=mb, mo=mo, kidx_host=kidx_h,
                               kidx=torch.from_numpy(kidx_h.view(np.int32)).cuda(),
                               rnd=torch.frombuffer(bytearray(b"".join(rnd) or b"\0" * 32), dtype=torch.uint8).cuda().view(-1, 32),
                               sig=torch.empty((max(1, len(ids)), ml.SIG_LEN), dtype=torch.uint8, device="cuda"),
                               st=torch.zeros(max(1, len(ids)), dtype=torch.int32, device="cuda"),
                               ok=torch.zeros(max(1, len(ids)), dtype=torch.uint8, device="cuda"))
            v = d["verify"]  # the signatures the verify requests carry: made once, here
            if v["n"]:
                ml.sign_device(sks, v["mb"], v["mo"], v["rnd"], v["sig"], v["n"], key_idx=v["kidx"], status=v["st"])
            torch.cuda.synchronize()
            for kind in self.count:
                self.count[kind] += len(r[kind])
            self.sets.append(d)
        self.ops_per_step = sum(self.count.values())
        assert self.ops_per_step == 3 * B
        by = {d["ml"].pset: d["ml"] for d in self.sets}
        self.bytes_per_op = sum(len(self.req[ps]["keygen"]) * (32 + m.PK_LEN + m.SK_LEN) + len(self.req[ps]["sign"]) * (m.SK_LEN + 64 + m.SIG_LEN)
                                + len(self.req[ps]["verify"]) * (m.PK_LEN + m.SIG_LEN + 33) for ps, m in by.items()) / self.ops_per_step
        self.name = (f"config 5 request stream: {3 * B} requests per step, set = (44,65,87)[i mod 3], keygen / sign / verify by i mod 10 "
                     f"(10/40/50 %): {self.count['keygen']} keygens + {self.count['sign']} signatures + {self.count['verify']} verifications, "
                     "one context, nine op-level calls per step, wire formats resident in HBM, no host wait inside a step")
        self.metric = "mixed ML-DSA-44/65/87 keygen+sign+verify requests/sec per GPU (batched); % HBM roofline"

    def step(self, i):
        for d in self.sets:
            ml, s, v = d["ml"], d["sign"], d["verify"]
            if len(d["kg_xi_host"]):
                ml.keygen_from_seed(d["kg_xi"], out=(d["kg_pk"], d["kg_sk"]))
            if s["n"]:
                ml.sign_device(d["sks"], s["mb"], s["mo"], s["rnd"], s["sig"], s["n"], key_idx=s["kidx"], status=s["st"], wait=False)
            if v["n"]:
                ml.verify_device(d["pks"], v["mb"], v["mo"], v["sig"], v["ok"], v["n"], key_idx=v["kidx"])

    def kernel_launches_per_step(self):
        return 1

    def finish_steps(self):
        """mldsa_sign_async plans its rounds until an unfinished op is unlikely and reports one as MLDSA_ERR_AGAIN; the stream
        plans like a synchronous call (MLDSA_OPT_SIGN_ASYNC_EXP = 2: three empty rounds less per call, a left-over op in about
        1 call in 500) and signs such ops again here -- inside the timed region, one look at the statuses per K steps (the
        inputs repeat every step, so what the last step left over is what every step left over)."""
        torch.cuda.synchronize()
        self.resigned = 0
        for d in self.sets:
            s, ml = d["sign"], d["ml"]
            if not s["n"]:
                continue
            again = torch.nonzero(s["st"][:s["n"]] == -5).flatten()  # MLDSA_ERR_AGAIN
            if again.numel():
                idx = again.cpu().tolist()
                self.resigned += len(idx)
                sig = ml.try_sign_with_seed(d["sks"], [s["msgs"][i] for i in idx], s["rnd"][again], key_idx=s["kidx_host"][idx])
                s["sig"][again] = sig
                s["st"][again] = 0

    def check(self, n_oracle=6):
        """one step, then against the oracle: a sample of every bucket (keys, signatures, verdicts), all statuses and verdicts"""
        from oracle import oracle as orc
        self.step(0)
        self.finish_steps()
        host = lambda t: t.cpu().numpy()
        for d in self.sets:
            ml, s, v = d["ml"], d["sign"], d["verify"]
            ps = ml.pset
            assert s["n"] == 0 or int(s["st"][:s["n"]].min()) == 0, "config 5: an op was left unfinished by the enqueued rounds"
            assert v["n"] == 0 or bool(v["ok"][:v["n"]].all()), "config 5: a valid signature was rejected"
            skb, pkb = host(d["sk"]), host(d["pk"])
            for j in range(min(n_oracle, len(d["kg_xi_host"]))):
                pk_o, sk_o = orc.keygen_from_seed(ps, d["kg_xi_host"][j])
                assert host(d["kg_pk"][j]).tobytes() == orc.pk_into_bytes(ps, pk_o) and host(d["kg_sk"][j]).tobytes() == orc.sk_into_bytes(ps, sk_o), \
                    "config 5: generated key differs from the oracle"
            for j in range(min(n_oracle, s["n"])):
                sk_o = orc.sk_try_from_bytes(ps, skb[s["kidx_host"][j]].tobytes())
                assert host(s["sig"][j]).tobytes() == orc.sign_internal(ps, sk_o, s["msgs"][j], s["rnd_host"][j], mode=0), \
                    "config 5: GPU signature differs from the oracle"
            for j in range(min(n_oracle, v["n"])):
                pk_o = orc.pk_try_from_bytes(ps, pkb[v["kidx_host"][j]].tobytes())
                assert orc.verify_internal(ps, pk_o, v["msgs"][j], host(v["sig"][j]).tobytes(), mode=0), "config 5: oracle rejects a GPU signature"

    def cpu_baseline(self, budget_s=9.0):
        """the same request mix on the host cores: every bucket's first ops through the oracle on all threads, the stream's rate =
        requests of a step / sum over the nine buckets of (requests / bucket rate)"""
        from oracle import oracle as orc
        cores = usable_cores()
        per = budget_s / 9.0
        t_step, rates, sampled = 0.0, {}, 0
        host = lambda t: t.cpu().numpy()
        for d in self.sets:
            ml, ps = d["ml"], d["ml"].pset
            pkb, skb = host(d["pk"]), host(d["sk"])
            pk_o = [orc.pk_try_from_bytes(ps, pkb[i].tobytes()) for i in range(d["nk"])]
            sk_o = [orc.sk_try_from_bytes(ps, skb[i].tobytes()) for i in range(d["nk"])]

            def timed(fn, n):
                fn(min(n, 64), 1)  # touch
                t0 = time.perf_counter()
                fn(n, 1)
                pilot = n / (time.perf_counter() - t0)
                rep = max(1, int(pilot * per / n))
                t0 = time.perf_counter()
                fn(n, rep)
                return n * rep / (time.perf_counter() - t0), n * rep
            s_, v_ = d["sign"], d["verify"]
            legs = {}
            if len(d["kg_xi_host"]):
                n = min(len(d["kg_xi_host"]), 512)
                legs["keygen"] = (timed(lambda m, rep: orc.keygen_batch_mt(ps, d["kg_xi_host"][:m], cores, rep), n), len(d["kg_xi_host"]))
            if s_["n"]:
                n = min(s_["n"], 1024)
                legs["sign"] = (timed(lambda m, rep: orc.sign_batch_mt(ps, sk_o, s_["kidx_host"][:m], s_["msgs"][:m], s_["rnd_host"][:m], cores, rep), n), s_["n"])
            if v_["n"]:
                n = min(v_["n"], 2048)
                sg = [x.tobytes() for x in host(v_["sig"][:n])]
                legs["verify"] = (timed(lambda m, rep: orc.verify_batch_mt(ps, pk_o, v_["kidx_host"][:m], v_["msgs"][:m], sg[:m], cores, rep), n), v_["n"])
            for kind, ((rate, done), count) in legs.items():
                rates[f"{kind}{ps}"] = rate
                t_step += count / rate
                sampled += done
        return dict(value=self.ops_per_step / t_step, unit=self.unit, cores=cores, kind="port", ops_per_s_by_bucket=rates,
                    sample=f"{sampled} oracle operations over the nine (set, class) buckets of the step's request mix on {cores} host threads; "
                           "value = requests per step / sum(bucket requests / bucket rate)")


def make_workload(name, hp, batch, rank, world=1):
    if name == "mixed":
        return MixedStream(hp, batch, rank, world)
    if name.startswith("verify_arith"):
        pset = int(name[len("verify_arith"):])
        return VerifyArith(hp, pset, batch or 4096, rank)
    for kind in ("verify", "sign"):
        core, suffix = name, ""
        for sfx in ("_cached_a", "_corrupt1", "_wire"):
            if name.endswith(sfx):
                core, suffix = name[:-len(sfx)], sfx
        if core.startswith(kind) and core[len(kind):].isdigit():
            return WholeOp(hp, int(core[len(kind):]), kind, batch or 65536, rank, cached_a=suffix == "_cached_a", world=world,
                           corrupt_every=100 if suffix == "_corrupt1" else 0, wire=suffix == "_wire")
    if name in ("ntt", "inv_ntt", "mat_vec_mul65", "expand_a65", "expand_mask65", "keygen44", "keygen65", "keygen87"):
        return SeamKernel(hp, name, batch, rank)
    raise SystemExit(f"unknown workload {name!r}")


def pmc_traffic(name):
    """HBM bytes per launch from the PMC passes kept under profiles/ (tools/collect_profiles.sh): the dominant
    kernel's figure, and per stage where collected.  None when the file is absent."""
    for fn in (f"r04_pmc_{name}.json", f"r03_pmc_{name}.json", f"r02_pmc_{name}.json", f"pmc_{name}.json"):
        path = os.path.join(ROOT, "profiles", fn)
        if os.path.exists(path):
            d = json.load(open(path))
            return d.get("hbm_bytes_per_launch"), d.get("by_stage", {}), fn
    return None, {}, None


def measure_pmc_traffic(workload, timeout_s=300):
    """HBM traffic of `workload`'s kernels from the hardware counters, measured NOW on this box: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, the program itself after `--`, as
    MI355X_MICROARCH.md prescribes), started before this process has touched the GPU.  Returns tools/pmc_summary.py's object
    (bytes per launch per stage, FETCH_SIZE doubled for gfx950) or None when the profiler is not available / fails / times out --
    the line then falls back to the figure kept under profiles/ and says so."""
    import importlib.util
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None  # this process is itself being profiled: no nested profiler
    dirs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix=f"mldsa_pmc_{counter}_", dir="/tmp")
            dirs[counter] = d
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", os.path.abspath(__file__),
                   "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-pmc"]
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)  # the exact process group this function started
                return None
            if rc != 0:
                return None
        spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out = mod.hbm_compute(workload, dirs["FETCH_SIZE"], dirs["WRITE_SIZE"])
        return out if out.get("hbm_bytes_per_launch") else None
    except Exception:
        return None
    finally:
        for d in dirs.values():
            shutil.rmtree(d, ignore_errors=True)


LIVE_PMC = {}  # workload -> measure_pmc_traffic() object of this run


def timed_steps(wl, world, steps, first):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        wl.step(first + i)
    if hasattr(wl, "finish_steps"):
        wl.finish_steps()  # inside the timed region: e.g. the mixed stream re-signs what its asynchronous calls left over
    ev1.record()
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, world), ev0.elapsed_time(ev1)


def run_one(args, hp, rank, world, name, steps, warmup, cpu_baseline, with_host_fed=False, cpu_budget_s=None):
    wl = make_workload(name, hp, args.batch if name == args.workload else 0, rank, world)
    if rank == 0:
        wl.check()
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize()
    whole = isinstance(wl, WholeOp)
    units_per_step = getattr(wl, "ops_per_step", wl.batch)

    # THE timed region: exactly K steps of the product's default path (a signing call whose shape repeats replays as a
    # hipGraph), barrier + synchronize on both sides, max over ranks -> `value`
    st0 = hp.stats() if hasattr(hp, "stats") else None
    dt, ev_ms = timed_steps(wl, world, steps, warmup)
    kern_ms = ev_ms / steps / wl.kernel_launches_per_step()
    value = units_per_step * world * steps / dt
    st1 = hp.stats()
    launch_mode = {"graph_replays": st1["graph_replays"] - st0["graph_replays"], "direct_calls": st1["direct_calls"] - st0["direct_calls"],
                   "sign_extra_rounds": st1["sign_extra_rounds"] - st0["sign_extra_rounds"]}
    # Per-kernel durations: a graph has no place for an event between two of its kernels, so the whole-op workloads
    # run the SAME K steps once more right away with a HIP event pair around every kernel launch on the launch
    # stream (the library launches directly while it is being profiled).  The roofline's kernel time comes from there.
    stages, dt_prof = None, None
    if whole:
        hp.profile_enable(True)
        dt_prof, _ = timed_steps(wl, world, steps, warmup + steps)
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
    root = os.path.dirname(os.path.abspath(__file__))
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


def main():
    args = parse()
    if args.inproc:
        return run_inproc_resident(args) if args.resident else run_inproc(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher.  It has not touched the GPU
        # (importing torch does not), starts N fresh rank processes of this script and relays rank 0's line.
        from fips204_amd import multi_gpu
        raise SystemExit(multi_gpu.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    # live HBM-traffic counters (rocprofv3 child passes) BEFORE this process touches the GPU: single-GPU runs only
    single = args.gpus == 1 and "RANK" not in os.environ
    if single and not args.no_pmc and (args.pmc or (args.workload == "verify65" and not args.no_extras)):
        got = measure_pmc_traffic(args.workload)
        if got:
            LIVE_PMC[args.workload] = got
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
        sw = run_sweep(hp, cpu=not args.no_cpu_baseline)
        sw["concurrent_small_verify_calls"] = run_small_calls()
        sw["single_op_callers"] = run_single_op_callers()  # (a process of its own, with its own context)
        v = next(pt for pt in sw["ops"]["verify"]["points"] if pt["n_ops"] == 65536)
        line = {"metric": "ML-DSA-65 verifies/sec per GPU (batched); batch-size sweep through the C ABI", "value": v["best_ops_per_s"], "unit": "verifies/s",
                "n_gpus": 1, "steps": v["direct"]["calls_timed"], "warmup": 3, "ms_per_step": v["best_ms_per_call"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": "ml_dsa_65 verify / sign / keygen at n_ops = " + ", ".join(str(n) for n in SWEEP_SIZES) +
                                       " (value = back-to-back verify calls of 65 536 ops), inputs resident in HBM"},
                "sweep": sw, "library_stats": hp.stats()}
        print(json.dumps(line), flush=True)
        hp.close()
        return multi_gpu.finish()
    line = run_one(args, hp, rank, world, args.workload, args.steps, args.warmup, not args.no_cpu_baseline,
                   with_host_fed=default_run or os.environ.get("MLDSA_BENCH_HOST_FED") == "1")
    # the default single-GPU run also carries the other BASELINE configs and the SURVEY 8(d) variants as extra objects (same JSON
    # line): config[1] = the HBM-roofline kernel, config[2] = whole sign, the 1 %-corrupted verify batch, the "from wire bytes"
    # units, and the batch-size sweep with the CPU crossover
    if default_run:
        also = {}
        cb = not args.no_cpu_baseline
        for name, st, wu, want_cb, budget in (("verify_arith44", 200, 10, cb, 3.0), ("sign65", 30, 3, cb, None), ("verify65_corrupt1", 20, 3, False, None),
                                              ("verify65_wire", 20, 3, cb, 3.0), ("sign65_wire", 10, 2, cb, 3.0)):
            sub = run_one(args, hp, rank, world, name, st, wu, want_cb, with_host_fed=(name == "sign65"), cpu_budget_s=budget)
            also[name] = {k: sub[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline") if k in sub}
            for k in ("stage_ms_per_step", "launch_gap_ms_per_step", "sign_iterations_per_signature", "launch_mode", "profiled_pass", "cpu_baseline",
                      "reference_published", "end_to_end_host_fed", "roofline_by_stage", "verdict_gather"):
                if k in sub and (name in ("verify_arith44", "sign65") or k in ("stage_ms_per_step", "cpu_baseline", "verdict_gather")):
                    also[name][k] = sub[k]
        sw = run_sweep(hp, cpu=cb)
        for o in sw["ops"].values():  # compact form inside the default line; `--workload sweep` prints everything
            for pt in o["points"]:
                for label in ("direct", "graph"):
                    pt[label] = {k: pt[label][k] for k in ("ms_per_call", "ops_per_s_back_to_back")}
        also["sweep"] = sw
        line["also"] = also
    if rank == 0:
        line["library_stats"] = hp.stats()
        print(json.dumps(line), flush=True)
    hp.close()
    multi_gpu.finish()


if __name__ == "__main__":
    main()
