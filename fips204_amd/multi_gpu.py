"""Batch split over the GPUs of one node (BASELINE configs 4 and 5).

Independent sign / verify operations shard trivially: contiguous slices of the batch, one process per
GPU, no collective on the data path.  RCCL (torch.distributed backend "nccl") is used only for the
launch barrier, the max-over-ranks of the elapsed time and the gather of the per-op verdict bytes into
rank 0.  Everything here also runs on CPU tensors with the "gloo" backend (tests/test_multirank_cpu.py).

    shard(n_ops, rank, world)        the slice of a batch one rank owns (ragged tails allowed)
    launch_ranks(n, argv)            start n fresh rank processes of a script BEFORE any GPU call
    init_process_group(backend)      rank / local_rank / world from the torchrun-style environment
    gather_verdicts(ok, n_total)     per-rank verdict bytes -> the whole batch's verdicts on rank 0
"""
import os
import socket
import subprocess
import sys


def shard(n_ops, rank, world):
    """Contiguous slice [start, start + count) of an n_ops batch owned by `rank` of `world`:
    ceil(n_ops / world) ops per rank, the last ranks take what is left (possibly nothing)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("shard: rank outside the world")
    per = -(-n_ops // world)
    start = min(n_ops, rank * per)
    return start, min(n_ops, start + per) - start


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, extra_env=None, timeout=None):
    """Run `python argv...` as n rank processes of one node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set like torch.distributed.run does) and wait for them.  The caller must not have touched
    the GPU: children are fresh interpreters, nothing is exec'ed from an initialised process.  Rank 0's
    stdout is passed through; returns the first non-zero exit code (0 if every rank succeeded)."""
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            code = p.wait(timeout=timeout)
            if code != 0 and rc == 0:
                rc = code
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:  # exactly the processes started here, never a pattern
            if p.poll() is None:
                p.kill()
    return rc


_DIST = False


def init_process_group(backend=None, device_index=None):
    """(rank, local_rank, world) from the environment; joins the process group when world > 1.
    backend None = "nccl" (RCCL) when a device is given, else "gloo"."""
    global _DIST
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MLDSA_BENCH_FORCE_DIST=1: join a process group even with one rank (exercises the RCCL path on a 1-GPU box)
    if world > 1 or (os.environ.get("MLDSA_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if device_index is not None else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        _DIST = True
    return rank, local_rank, world


def is_distributed():
    return _DIST


def barrier():
    if _DIST:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, device="cpu"):
    if not _DIST:
        return x
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(x, device="cpu"):
    """the slowest rank sets `value` (max_over_ranks of the time); this shows how far the fastest one was ahead"""
    return -max_over_ranks(-x, device)


def gather_verdicts(ok_local, n_total):
    """The verdict bytes of this rank's shard (uint8 tensor, shard(n_total, rank, world)[1] entries) -> the
    n_total verdicts of the whole batch in batch order (returned on every rank; SURVEY 8e: <= 1 MiB even for
    2^20 ops).  One all_gather of equally sized (padded) pieces -- RCCL on device tensors, gloo on CPU."""
    import torch
    if not _DIST:
        return ok_local[:n_total].clone()
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_total // world)
    start, count = shard(n_total, rank, world)
    if ok_local.numel() < count:
        raise ValueError("gather_verdicts: shard shorter than shard(n_total, rank, world)")
    piece = torch.zeros(per, dtype=torch.uint8, device=ok_local.device)
    piece[:count] = ok_local[:count]
    out = torch.empty(per * world, dtype=torch.uint8, device=ok_local.device)
    dist.all_gather_into_tensor(out, piece)
    return out[:n_total]  # pieces are contiguous slices of ceil(n/world): padding only at the very end


def finish():
    global _DIST
    if _DIST:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        _DIST = False
