"""HBM traffic from the hardware counters: figures kept under profiles/ and the live child passes under rocprofv3."""
import json
import os

from .constants import ROOT

PROFILE_ROUND = "r06"


def pmc_traffic(name):
    """HBM bytes per launch from the PMC passes kept under profiles/ (tools/collect_profiles.sh): the dominant
    kernel's figure, and per stage where collected.  Only THIS round's file counts (profiles/r06_MANIFEST.json ties it to the
    library it was taken on): when it is absent the answer is None and the line's `traffic` is null -- no older file stands in."""
    fn = f"{PROFILE_ROUND}_pmc_{name}.json"
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
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py"),
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

