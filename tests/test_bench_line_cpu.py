"""The bench line's size and shape contract (the round-4 line grew to 29.8 KB and the harness kept only its
last 8 KB -- nothing of it was parsed).  benchlib/line.py builds the one printed line from the full result objects; these tests
feed it worst-case objects -- the round-4 line itself (profiles/r04_bench_default.json, 29 830 bytes) and a synthetic object whose
every string is an essay and every number a 17-digit float -- and check the limit, the required keys and that nothing long
survives.  Reference counterpart: nine criterion lines, /root/reference/benches/benchmark.rs:28-62."""
import json
import os

import pytest

from benchlib import line as bline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worst_case():
    essay = "x" * 1200
    roof = {"bound": "valu", "kernel": "k_" + "expand_a" * 30, "achieved": 8.503271234567891, "peak": 8.972345678912345, "unit": "G Keccak-f[1600]/s",
            "frac": 0.9476543219876543, "traffic": 1794123456.7891234, "algorithmic_bytes_per_launch": 1512046592.123456, "kernel_ms": 1.1561234567891234,
            "traffic_measured_in_this_run": True, "traffic_source": essay, "note": essay, "peak_derivation": {"formula": essay, "mix": {str(i): essay for i in range(20)}},
            "hbm_view": {"survey_8d_int32_model": {"frac": 0.21812345678912345, "note": essay}, "packed_24bit_as_stored": {"note": essay}}}
    full = {"metric": "ML-DSA-65 verifies/sec per GPU (batched); % HBM roofline", "value": 37912345.67891234, "unit": "verifies/s", "n_gpus": 8, "steps": 100,
            "warmup": 5, "ms_per_step": 1.7312345678912345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "ml_dsa_65 batch=65536 whole verify on FIPS 204 wire formats, GPU ExpandA, " + essay, "batch_per_gpu": 65536,
                       "parallelism": "batch-split x8", "input_sets_rotated": 1},
            "roofline": roof, "cpu_baseline": {"value": 143123.45678912345, "unit": "verifies/s", "cores": 16, "kind": "port", "single_thread_value": 9312.345678912345,
                                               "sample": essay},
            "reference_published": {"value": 35719.38848406915, "us_per_op": 27.996, "unit": "verifies/s per core", "source": essay, "note": essay},
            "ranks": {"min": 4712345.678912345, "max": 4798765.432198765, "unit": essay},
            "stage_ms_per_step": {f"stage{i}": 0.123456789 for i in range(40)}, "roofline_by_stage": {f"stage{i}": dict(roof) for i in range(40)},
            "library_stats": {f"s{i}": i for i in range(60)}, "profiled_pass": {"note": essay}}
    sub = dict(full, roofline=dict(roof, bound="hbm"), end_to_end_host_fed={"note": essay})
    also = {"sign65": sub, "verify_arith44": sub, "sign65_wire": sub, "verify65_wire": sub, "verify65_corrupt1": sub}
    return full, also


def test_worst_case_object_fits_and_has_the_contract_keys():
    full, also = _worst_case()
    assert len(json.dumps(full)) > 50000           # the input really is oversized
    line = bline.compact_line(full, also)
    text = bline.check_line(line)
    assert len(text) <= bline.MAX_LINE_BYTES == 6000
    got = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in got, k
    assert set(got["config"]) == {"workload", "batch_per_gpu", "parallelism"} and "model" not in got["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "hbm_frac_survey_8d", "traffic", "algorithmic_bytes_per_launch", "kernel_ms",
              "traffic_measured_in_this_run"):
        assert k in got["roofline"], k
    assert "peak_derivation" not in got["roofline"] and "note" not in got["roofline"] and "hbm_view" not in got["roofline"]
    assert got["roofline"]["hbm_frac_survey_8d"] == pytest.approx(0.218123, rel=1e-5)
    for k in ("value", "unit", "cores", "single_thread_value", "kind"):
        assert k in got["cpu_baseline"], k
    assert set(got["reference_published"]) == {"value", "us_per_op"}
    assert set(got["ranks"]) == {"min", "max"}
    assert got["also"]["sign65"].keys() >= {"value", "ms_per_step", "roofline_frac", "roofline_kernel", "cpu_value", "cpu_cores"}
    assert got["also"]["verify_arith44"].keys() >= {"value", "hbm_frac", "traffic_ratio"}
    assert got["extras_file"] == "bench_extras.json"
    assert got["value"] == pytest.approx(full["value"], rel=1e-5)   # rounding to 6 significant figures only


def test_round4_line_would_now_fit():
    """the line that broke BENCH_r04.json, re-built through the compact builder"""
    path = os.path.join(ROOT, "profiles", "r04_bench_default.json")
    old = json.loads(open(path).read().strip().splitlines()[-1])
    assert len(json.dumps(old)) > 8192
    also = {k: v for k, v in old.get("also", {}).items() if k in ("sign65", "verify_arith44")}
    text = bline.check_line(bline.compact_line(old, also))
    assert len(text) < 3000, len(text)
    got = json.loads(text)
    assert got["roofline"]["bound"] == "valu" and 0.2 < got["roofline"]["hbm_frac_survey_8d"] < 0.25
    assert got["also"]["verify_arith44"]["traffic_ratio"] == pytest.approx(1.0, abs=0.03)


def test_check_line_refuses_what_the_harness_cannot_read():
    full, also = _worst_case()
    line = bline.compact_line(full, also)
    with pytest.raises(AssertionError):
        bline.check_line(dict(line, blob="y" * 7000))
    with pytest.raises(AssertionError):
        bline.check_line(dict(line, note="y" * 101))
    missing = dict(line)
    del missing["roofline"]
    with pytest.raises(AssertionError):
        bline.check_line(missing)


def test_hbm_bound_headline_reports_its_own_frac_as_the_survey_view():
    full, _ = _worst_case()
    full["roofline"] = {"bound": "hbm", "kernel": "k_verify_arith<4,4>", "achieved": 4590.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.574, "traffic": None,
                        "algorithmic_bytes_per_launch": 121634816, "kernel_ms": 0.0265, "traffic_measured_in_this_run": False}
    r = bline.compact_line(full)["roofline"]
    assert r["hbm_frac_survey_8d"] == 0.574 and r["traffic"] is None


def test_no_profile_fallback_to_older_rounds():
    """VERDICT r4 'weak' 6: roofline.traffic must come from THIS round's PMC file or be null"""
    from benchlib import pmc
    t, by, fn = pmc.pmc_traffic("no_such_workload")
    assert t is None and by == {} and fn is None
    src = open(os.path.join(ROOT, "benchlib", "pmc.py")).read()
    assert "r04_pmc" not in src and "r03_pmc" not in src and "r02_pmc" not in src


def test_side_file_is_written_beside_bench_py(tmp_path):
    paths = bline.write_extras(str(tmp_path), {"a": {"b": 1.5}})
    assert paths == [str(tmp_path / "bench_extras.json")] and json.load(open(paths[0])) == {"a": {"b": 1.5}}
    os.mkdir(tmp_path / "gpurun_out")
    assert len(bline.write_extras(str(tmp_path), {"a": 1})) == 2


def test_line_carries_the_spread_of_the_timed_steps_and_the_side_object_the_clocks():
    """the one driver-timed number came without spread or clock state.  runner.run_one now runs >= 200 ms of the
    workload before the counted warm-up, marks the timed steps with events and reads sclk either side; the line gets three numbers
    (step_ms min / median / max), the side file the rest -- and the worst case still fits."""
    full, also = _worst_case()
    full["step_ms"] = {"min": 1.7012345678, "median": 1.7312345678, "max": 1.9912345678, "steps_per_mark": 2}
    full["clock_ramp"] = {"steps": 120, "ms": 207.5, "note": "x" * 300}
    full["sclk"] = {"before_timed_region": {"mhz": 2400, "source": "/sys/class/drm/card1/device/pp_dpm_sclk"}, "after_timed_region": None, "note": "x" * 300}
    line = bline.compact_line(full, also)
    text = bline.check_line(line)
    assert len(text) <= bline.MAX_LINE_BYTES
    assert line["step_ms"] == {"min": 1.7012, "median": 1.7312, "max": 1.9912}
    assert "clock_ramp" not in line and "sclk" not in line   # side file only
    # the runner's helpers without a GPU: the sysfs reader returns None or a dict, never raises
    from benchlib import runner
    got = runner.read_sclk()
    assert got is None or (isinstance(got["mhz"], int) and got["source"])
    import inspect
    src = inspect.getsource(runner.run_one)
    assert src.index("clock_ramp(wl, 0") < src.index("for i in range(warmup)") < src.index("timed_steps(wl, world, steps, warmup)")
    # nothing that leaves the device idle for long (a sysfs read, a subprocess) between the warm-up and the timed region: the clock is read
    # half way through the ramp, while its steps run
    between = src[src.index("for i in range(warmup)"):src.index("timed_steps(wl, world, steps, warmup)")]
    assert "read_sclk" not in between and "subprocess" not in between
    # ... and the live PMC passes run before torch -- the HIP runtime -- is loaded (bench.py early_pmc: with the runtime mapped while a
    # profiler session ran in another process, the first deep burst of launches ran slowly: the first dozen steps of the timed region)
    bsrc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert bsrc.index("early_pmc(_ARGS)") < bsrc.index("\nimport torch")
