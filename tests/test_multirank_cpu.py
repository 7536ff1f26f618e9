"""Multi-rank path on CPU (gloo, world_size 2): the batch split of SURVEY.md 8e has no data-path
collective -- each rank derives its own slice of the synthetic workload from the global op
index -- so what must hold is: slices are disjoint and cover the job, and the timing reduction
bench.py performs (barrier + MAX over ranks) works across processes."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    batch = 64
    base = rank * batch
    # the same derivation bench.WholeOp uses for this rank's slice
    msgs = [bench._shake(b"mldsa-bench-msg", base + i, 8) for i in range(batch)]
    keys = [bench._shake(b"mldsa-bench-key" + bytes([65]), base + i, 4) for i in range(8)]
    digest = hashlib.sha256(b"".join(msgs + keys)).digest()
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, base, digest.hex(), msgs[0].hex(), msgs[-1].hex()))
    # bench.py's timing reduction: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"gathered": gathered, "tmax": float(t.item())}))
    dist.destroy_process_group()
""" % ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_batch_split_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    g = sorted(res["gathered"])
    assert [x[0] for x in g] == [0, 1] and [x[1] for x in g] == [0, 64]  # contiguous, disjoint slices
    assert g[0][2] != g[1][2]                                              # distinct data per rank
    import bench
    assert g[1][3] == bench._shake(b"mldsa-bench-msg", 64, 8).hex()      # rank 1 starts where rank 0 ends
    assert g[0][4] == bench._shake(b"mldsa-bench-msg", 63, 8).hex()
    assert res["tmax"] == 2.0                                              # MAX over ranks


def test_bench_reads_torchrun_environment(monkeypatch):
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.max_over_ranks(3.5, 1) == 3.5
    args_default = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
    assert args_default is None or args_default.gpus == 1
