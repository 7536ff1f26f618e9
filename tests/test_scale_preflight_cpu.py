"""tools/scale_preflight.py, the day-one check of a multi-GPU node: its N = 2 and N = 3 dry-run path on CPU -- fresh rank
processes over gloo run the shard / gather / min-max-over-ranks plumbing and rank 0 prints the compact bench line with
ranks{min, max}; the parent applies the same checks it applies to `bench.py --gpus N` on a real node."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "scale_preflight.py")


@pytest.mark.parametrize("n", [2, 3])
def test_dry_run(n):
    out = subprocess.run([sys.executable, SCRIPT, "--dry-run", "--gpus", str(n)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["dry_run"] == "ok" and "skipped" in rep
    line = rep["line"]
    assert line["n_gpus"] == n and line["ranks"]["min"] <= line["ranks"]["max"] and line["value"] == pytest.approx(line["ranks"]["min"] * n)
    assert len(json.dumps(line)) <= 6000


def test_line_checks_catch_what_a_real_run_could_show():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_preflight as sp
    ok = {"n_gpus": 8, "value": 3.0e8, "ranks": {"min": 3.7e7, "max": 3.8e7}}
    assert sp.check_line(json.dumps(ok), 8)[0] == "ok"
    assert sp.check_line(json.dumps(dict(ok, n_gpus=4)), 8)[0].startswith("failed")
    assert sp.check_line(json.dumps({k: v for k, v in ok.items() if k != "ranks"}), 8)[0].startswith("failed")
    assert "STRAGGLER" in sp.check_line(json.dumps(dict(ok, ranks={"min": 2.0e7, "max": 3.8e7})), 8)[0]
    assert sp.check_line(json.dumps(dict(ok, value=0)), 8)[0].startswith("failed")
    assert sp.check_line("{" + " " * 7000 + "}", 8)[0].startswith("failed")


# ------------------------------------------------------------------------------ which RCCL the group gather binds (csrc/group.hip load_rccl)
_STANDIN = r"""
/* a stand-in for librccl.so.1: the symbols the loader looks for, a recognisable version */
int ncclGetVersion(int *v) { *v = 99999; return 0; }
int ncclAllGather(const void *s, void *r, unsigned long n, int t, void *c, void *st) { (void)s; (void)r; (void)n; (void)t; (void)c; (void)st; return 0; }
int ncclCommInitAll(void **c, int n, const int *d) { (void)c; (void)n; (void)d; return 1; }
"""
_PROBE = r"""
import ctypes as C, json, sys
pre, lib_path = sys.argv[1], sys.argv[2]
if pre != "-":
    C.CDLL(pre)                      # the process has an RCCL mapped already (what `import torch` does in a Python host)
lib = C.CDLL(lib_path)
lib.mldsa_group_rccl_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
buf = C.create_string_buffer(1024)
ver = lib.mldsa_group_rccl_info(None, buf, len(buf))
print(json.dumps({"version": ver, "text": buf.value.decode()}))
"""


def test_group_gather_reuses_the_rccl_the_process_has_mapped(tmp_path):
    """VERDICT r5 "What's weak" 6: the loader asked for "librccl.so" first, which resolves through the rpath to /opt/rocm/lib -- a SECOND RCCL
    beside the one torch has mapped.  Now: whatever is mapped under the SONAME librccl.so.1 is reused (RTLD_NOLOAD), then the usual search
    (LD_LIBRARY_PATH before the rpath), then /opt/rocm/lib; mldsa_group_rccl_info(NULL) reports the file and ncclGetVersion.  Here with a
    stand-in library, without torch and without a GPU."""
    lib = os.path.join(ROOT, "fips204_amd", "csrc", "libmldsa_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    src = tmp_path / "standin.c"
    src.write_text(_STANDIN)
    d = tmp_path / "fake"
    d.mkdir()
    standin = d / "librccl.so.1"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname,librccl.so.1", "-o", str(standin), str(src)])
    probe = tmp_path / "probe.py"
    probe.write_text(_PROBE)

    def run(pre, env_extra):
        env = {k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"}
        env.update(env_extra)
        out = subprocess.run([sys.executable, str(probe), pre, lib], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    # (1) mapped already -> reused, whatever the search path would find
    got = run(str(standin), {})
    assert got["version"] == 99999 and got["text"] == "reused " + str(standin), got
    # (2) nothing mapped, LD_LIBRARY_PATH names one -> loaded from there (before the library's own rpath)
    got = run("-", {"LD_LIBRARY_PATH": str(d)})
    assert got["version"] == 99999 and got["text"] == "loaded " + str(standin), got
    # (3) nothing mapped, nothing in the search path -> ROCm's copy through the rpath (when this image has one)
    got = run("-", {})
    if os.path.exists("/opt/rocm/lib/librccl.so.1") or os.path.exists("/opt/rocm/lib/librccl.so"):
        assert got["version"] > 20000 and got["text"].startswith("loaded /opt/rocm") and "librccl.so" in got["text"], got
    else:
        assert got["version"] < 0


def test_rccl_report_of_the_preflight_names_one_file():
    """tools/scale_preflight.py --rccl-report in a Python host: torch's RCCL is mapped, the library reuses exactly that file"""
    out = subprocess.run([sys.executable, SCRIPT, "--rccl-report"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["library_rccl"]["how"] == "reused" and rep["library_rccl_is_the_mapped_one"] == "ok", rep
    assert rep["library_rccl"]["file"] in rep["rccl_mapped_before_probe"]
