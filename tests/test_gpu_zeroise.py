"""Proof that secrets are zeroised (the reference: `Zeroize, ZeroizeOnDrop`, src/types.rs:19, 45).

mldsa_debug_secret_residue counts the non-zero bytes of the workspace span the LAST op-level call used for secret-dependent data
(recorded where the call carves its workspace, independently of the clearing code) and of the staging buffers that held
secrets during the last *_host call.  tests/zeroise_scenarios.py drives every path out of a signing / key-generation call; here
the shipped library must leave 0 bytes on each, and a build of the SAME sources with the clearing compiled out
(-DMLDSA_TEST_NO_ZEROISE, `make nozero`) must leave plenty -- the negative control that makes the zero meaningful."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fips204_amd", "csrc")
NOZERO = os.path.join(ROOT, "tests", "_build", "libmldsa_hip_nozero.so")

CALLS = ("keygen", "get_public_key", "sign", "sign_async", "sign_graph_replay", "sign_with_refused_ops", "sign_host", "keygen_host",
         "sign_host_direct_export", "destroy", "replace_workspace")


def test_no_secret_outlives_its_call():
    import zeroise_scenarios
    res = zeroise_scenarios.run_all()
    for name in CALLS:
        scanned, nonzero = res[name]
        assert scanned > 100_000, (name, "nothing was scanned", res)
        assert nonzero == 0, (name, f"{nonzero} non-zero bytes of {scanned} left behind", res)
    assert res["after_verify"] == [0, 0] or res["after_verify"][1] == 0
    assert res["_graph_replays"][0] >= 1, "the replayed-graph path was not exercised"
    assert res["_ctx_len_raised"][0] == 1
    assert res["_direct_export_verifies"][0] == 1
    assert res["_workspace_bytes_used_before_destroy"][0] > 1_000_000   # the caller-owned buffer really was the workspace


def test_probe_finds_the_secrets_when_nothing_clears_them():
    """negative control: the same scenarios against the no-zeroise build"""
    if not os.path.exists(NOZERO) or os.path.getmtime(NOZERO) < os.path.getmtime(os.path.join(CSRC, "pipeline.hip")):
        subprocess.check_call(["make", "-C", CSRC, "-j8", "nozero"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "zeroise_scenarios.py"), NOZERO], capture_output=True, text=True,
                         timeout=1200, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    res = json.loads(out.stdout.strip().splitlines()[-1])
    for name in CALLS:
        scanned, nonzero = res[name]
        assert nonzero > 10_000, (name, "the probe is blind on this path: its zero above would prove nothing", res)
    # the functional results do not depend on the clearing
    assert res["_direct_export_verifies"][0] == 1 and res["_graph_replays"][0] >= 1
