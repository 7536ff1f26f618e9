"""The CPU side of the measurement: host threads this process may use, oracle keygen rates, the synthetic-input hash."""
import os
import time


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

