#!/usr/bin/env python3
"""BASELINE config 4 at FULL size on the one GPU a gpurun box has: 2^20 ML-DSA-87 verifications split 8 ways
by the library itself -- mldsa_group_create([0] * 8): eight contexts and worker threads, 131 072 ops each (SURVEY 8(d') C4) --

  1. host-fed:          mldsa_verify_host_group on the whole batch (wire-format keys, host arrays)
  2. device-resident:   slice i resident in HBM under context i, ONE mldsa_verify_group call, then the 2^20 verdict bytes gathered
                        into every context's buffer with mldsa_group_allgather (device-to-device copies: RCCL refuses a device
                        listed twice)

with 1 % of the signatures damaged (SURVEY 8(d) "correctness-under-load" mix) and EVERY verdict compared with the oracle's
(orc.verify_batch_mt on the host cores).  Eight contexts share one GPU here, so the rates say nothing about scaling: the run
shows that the partition arithmetic, eight contexts' workspaces and the 2^20-byte gather work at the size config 4 names.
Writes one JSON object (stdout and, if given, argv[1]).  Mirrors /root/reference/src/ml_dsa.rs:351-437 x 2^20."""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fips204_amd.ml_dsa import MlDsaGroup  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def shake(tag, i, width=8):  # SHAKE256(tag | i_le32 or i_le64, 32)
    return hashlib.shake_256(tag + int(i).to_bytes(width, "little")).digest(32)


def main():
    pset, N = 87, 8
    n = int(os.environ.get("CONFIG4_OPS", 1 << 20))
    nk = 1024
    per = -(-n // N)
    res = {"config": f"ML-DSA-87, {n} verifications, {N} contexts on GPU 0 ({per} ops each), 1 % damaged signatures", "functional_run": True,
           "note": "eight contexts share ONE GPU: not a scaling measurement; distinct-GPU scaling stays unmeasured (no multi-GPU box)"}
    g = MlDsaGroup(pset, [0] * N)
    t0 = time.perf_counter()
    xi = np.frombuffer(b"".join(shake(b"mldsa-bench-key" + bytes([pset]), i, 4) for i in range(nk)), dtype=np.uint8).reshape(nk, 32)
    pk, sk = g.keygen_host(xi)
    msgs = np.frombuffer(b"".join(shake(b"mldsa-bench-msg", i) for i in range(n)), dtype=np.uint8).copy()
    moff = np.arange(n + 1, dtype=np.uint64) * 32
    rnd = np.frombuffer(b"".join(shake(b"mldsa-bench-rnd", i) for i in range(n)), dtype=np.uint8).reshape(n, 32)
    kidx = (np.arange(n) % nk).astype(np.uint32)
    sig = g.sign_host(sk, (msgs, moff), rnd, key_idx=kidx)
    res["setup_s"] = time.perf_counter() - t0
    # 1 % damaged: one flipped bit, positions walking through c~ | z | hints
    rows = np.arange(37, n, 100)
    sig = np.ascontiguousarray(sig)
    sig[rows, (rows * 2654435761) % g.SIG_LEN] ^= (1 << (rows % 8)).astype(np.uint8)
    damaged = np.zeros(n, dtype=bool)
    damaged[rows] = True

    # ---- the oracle's verdicts for all n ops, in chunks (the CPU baseline of this workload at the same time)
    pk_o = [orc.pk_try_from_bytes(pset, pk[i].tobytes()) for i in range(nk)]
    import bench
    threads = bench.usable_cores()  # the container's CPU quota, not the host's 256 logical CPUs
    want = np.zeros(n, dtype=bool)
    t0 = time.perf_counter()
    for a in range(0, n, 65536):
        b = min(n, a + 65536)
        want[a:b] = orc.verify_batch_mt(pset, pk_o, kidx[a:b], [msgs[32 * i:32 * i + 32].tobytes() for i in range(a, b)],
                                        [sig[i].tobytes() for i in range(a, b)], threads, 1, mode=0)
    dt = time.perf_counter() - t0
    res["oracle"] = {"seconds": dt, "verifies_per_s": n / dt, "threads": threads}
    assert np.array_equal(want, ~damaged), "the oracle accepts a damaged signature or rejects a good one"

    # ---- 1. host-fed through the group
    ok = np.zeros(n, dtype=np.uint8)
    g.verify_host(pk, (msgs, moff), sig, key_idx=kidx, out=ok)  # warm-up: staging buffers
    t0 = time.perf_counter()
    got = g.verify_host(pk, (msgs, moff), sig, key_idx=kidx, out=ok)
    dt = time.perf_counter() - t0
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"host-fed group verdicts differ from the oracle at {bad[:8]}"
    res["host_fed_group"] = {"seconds": dt, "verifies_per_s": n / dt, "verdicts_equal_oracle": n, "rejected": int((~got).sum()),
                             "entry_point": "mldsa_verify_host_group", "pageable_host_arrays": True}

    # ---- 2. device-resident slices, one mldsa_verify_group call, verdict all-gather
    slices, bufs = [], []
    for i in range(N):
        a, c = g.shard(n, i)
        mi = g.on_device(i)
        pks_i = mi.public_keys_from_bytes(torch.from_numpy(pk).cuda())
        buf = torch.full((per * N,), 9, dtype=torch.uint8, device="cuda")
        bufs.append(buf)
        slices.append(dict(pks=pks_i, msg_buf=torch.from_numpy(msgs[32 * a:32 * (a + c)]).cuda(),
                           msg_off=torch.from_numpy((np.arange(c + 1, dtype=np.uint64) * 32).view(np.int64)).cuda(),
                           key_idx=torch.from_numpy(kidx[a:a + c].view(np.int32)).cuda(), sigs=torch.from_numpy(sig[a:a + c]).cuda(),
                           ok=buf[a:a + c], n_ops=c))
    torch.cuda.synchronize()
    g.verify_group(slices, wait=True)  # warm-up: eight workspaces of a 131 072-op ML-DSA-87 pass
    free, total = torch.cuda.mem_get_info()
    res["device_memory_GB"] = {"used": (total - free) / 1e9, "total": total / 1e9}
    for b in bufs:
        b.fill_(9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.verify_group(slices, wait=False)        # enqueued on eight contexts ...
    g.allgather(bufs, n, use_rccl=0)          # ... and gathered right behind it: ordered on the device, returns when complete
    dt = time.perf_counter() - t0
    for i, b in enumerate(bufs):
        got = b[:n].cpu().numpy().astype(bool)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"device-resident verdicts in buffer {i} differ from the oracle at {bad[:8]}"
    t0 = time.perf_counter()
    g.allgather(bufs, n, use_rccl=0)
    res["device_resident_group"] = {"seconds_verify_plus_gather": dt, "verifies_per_s": n / dt, "verdicts_equal_oracle_in_every_buffer": n,
                                    "gather_ms": (time.perf_counter() - t0) * 1e3, "gather_bytes_per_buffer": n, "buffers": N,
                                    "entry_points": "mldsa_verify_group(wait = 0) + mldsa_group_allgather(use_rccl = 0)"}
    g.close()
    text = json.dumps(res)
    print(text, flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
