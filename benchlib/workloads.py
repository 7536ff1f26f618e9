"""Workloads of bench.py: BASELINE.json's configs as objects with step() / check() / cpu_baseline()."""
import os
import time

import numpy as np
import torch

from .constants import Q, SETS
from .cpu import _shake, oracle_keygen_rates, usable_cores


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
                    sample_short=f"{n_ops * repeat} {self.kind} ops (first {n_ops} of the batch x {repeat}), {cores} threads, {dt:.1f} s",
                    sample=f"{n_ops * repeat} whole {self.kind} ops{' incl. try_from_bytes of the key' if self.wire else ''} (the batch's first {n_ops} ops x {repeat} passes) on {cores} "
                           f"host threads (pthreads; = the container's CPU quota on a {os.cpu_count()}-CPU host), oracle/liboracle.so = KAT-pinned C "
                           f"restatement with per-op ExpandA, {dt:.1f} s")


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

