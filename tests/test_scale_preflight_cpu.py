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
