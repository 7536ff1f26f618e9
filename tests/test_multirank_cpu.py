"""Multi-rank path on CPU (gloo, world_size 2 and 3): the batch split of SURVEY.md 8e has no data-path
collective, so what must hold is that the PRODUCT code bench.py and the host layers use --
fips204_amd.multi_gpu.shard / launch_ranks / init_process_group / gather_verdicts / max_over_ranks --
partitions a batch into disjoint contiguous slices that cover it (ragged B % N != 0 included), starts one
fresh process per rank before any GPU call, and reassembles the per-rank verdict bytes in batch order."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from fips204_amd import multi_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(0, 1), (1, 1), (10, 3), (65536, 8), (1 << 20, 8), (7, 8), (1000003, 7), (5, 2)])
def test_shard_partitions_the_batch(n, world):
    pieces = [multi_gpu.shard(n, r, world) for r in range(world)]
    pos = 0
    for start, count in pieces:
        assert start == pos and count >= 0  # contiguous, in rank order
        pos += count
    assert pos == n
    per = -(-n // world) if world else 0
    assert all(c <= per for _, c in pieces)  # ceil(B / N) per GPU
    assert all(c == per for _, c in pieces[:n // per if per else 0])
    with pytest.raises(ValueError):
        multi_gpu.shard(n, world, world)


def test_config4_slice():
    """BASELINE config 4: 2^20 ML-DSA-87 verifies over 8 GPUs = 131072 per GPU."""
    assert [multi_gpu.shard(1 << 20, r, 8) for r in (0, 7)] == [(0, 131072), (7 * 131072, 131072)]


WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, %r)
    import torch
    from fips204_amd import multi_gpu
    import bench
    rank, local_rank, world = multi_gpu.init_process_group("gloo")
    n_total = int(sys.argv[1])
    start, count = multi_gpu.shard(n_total, rank, world)
    # this rank's slice of the synthetic job, derived from the GLOBAL op index like bench.WholeOp does
    msgs = [bench._shake(b"mldsa-bench-msg", start + i, 8) for i in range(count)]
    # a verdict pattern that depends on the global index only: op i fails iff i %% 7 == 3
    ok = torch.tensor([0 if (start + i) %% 7 == 3 else 1 for i in range(count)], dtype=torch.uint8)
    allok = multi_gpu.gather_verdicts(ok, n_total)
    tmax = multi_gpu.max_over_ranks(1.0 + rank)
    multi_gpu.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "verdicts": allok.tolist(), "tmax": tmax,
                          "first_msg": msgs[0].hex() if msgs else None}), flush=True)
    multi_gpu.finish()
""" % ROOT)


@pytest.mark.parametrize("world,n_total", [(2, 64), (2, 65), (3, 10)])
def test_launch_shard_gather(tmp_path, world, n_total, capfd):
    """launch_ranks starts `world` fresh processes (RANK / WORLD_SIZE / MASTER_* set), they shard the job,
    and the gathered verdict array is the whole batch in order -- also when B %% N != 0."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    rc = multi_gpu.launch_ranks(world, [str(script), str(n_total)], timeout=240)
    out = capfd.readouterr().out
    assert rc == 0, out[-2000:]
    res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert res["world"] == world
    assert res["verdicts"] == [0 if i % 7 == 3 else 1 for i in range(n_total)]
    assert res["tmax"] == float(world)  # MAX over ranks of 1 + rank
    import bench
    assert res["first_msg"] == bench._shake(b"mldsa-bench-msg", 0, 8).hex()


def test_launch_ranks_reports_failure(tmp_path):
    script = tmp_path / "bad.py"
    script.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert multi_gpu.launch_ranks(2, [str(script)], timeout=60) == 3


def test_bench_gpus_flag_launches_ranks(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment must start N ranks itself (the flag was
    parsed and ignored).  The launcher is intercepted here: no GPU in this container."""
    import bench
    seen = {}

    def fake_launch(n, argv, **kw):
        seen["n"], seen["argv"] = n, list(argv)
        return 0

    monkeypatch.setattr(multi_gpu, "launch_ranks", fake_launch)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--workload", "verify87", "--batch", "131072"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["n"] == 4
    assert seen["argv"][0].endswith("bench.py") and seen["argv"][1:] == ["--gpus", "4", "--workload", "verify87", "--batch", "131072"]


def test_bench_refuses_a_world_that_disagrees_with_gpus(monkeypatch):
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        bench.dist_setup(bench.parse(["--gpus", "8"]))


def test_library_slice_math_matches_the_python_split():
    """mldsa_group_shard (the C ABI's in-library batch split, csrc/group.hip) = multi_gpu.shard for every (n, world, rank):
    contiguous ceil(n / world) slices, ragged and empty tails included.  Pure host arithmetic: runs without a GPU."""
    import ctypes as C
    from fips204_amd import _lib
    from fips204_amd.multi_gpu import shard
    lib = _lib.load()
    for world in (1, 2, 3, 7, 8):
        for n in (0, 1, 5, 8, 63, 64, 65, 65536, 65537, 1 << 20, (1 << 20) + 3):
            covered = 0
            for rank in range(world):
                a, c = C.c_size_t(), C.c_size_t()
                assert lib.mldsa_group_shard(n, world, rank, C.byref(a), C.byref(c)) == 0
                assert (a.value, c.value) == shard(n, rank, world), (n, world, rank)
                assert a.value == covered or c.value == 0
                covered += c.value
            assert covered == n
    a, c = C.c_size_t(), C.c_size_t()
    assert lib.mldsa_group_shard(10, 2, 2, C.byref(a), C.byref(c)) != 0   # part outside the group
    assert lib.mldsa_group_shard(10, 0, 0, C.byref(a), C.byref(c)) != 0
