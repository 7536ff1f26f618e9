"""One process per GPU: rendezvous, barrier, max over ranks (torch.distributed; RCCL on GPUs, gloo for CPU tests)."""
import os

import torch


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



def min_over_ranks(x, world):
    return -max_over_ranks(-x, world)
