"""The ONE JSON line bench.py prints, kept small enough for the harness that reads it.

The full result objects (every stage, every by-stage roofline, derivations, host-fed legs, sweeps ...) go to a side file
(bench_extras.json beside bench.py, named in the line as `extras_file`); the line itself carries only the contract's keys, the
`roofline` and `cpu_baseline` objects and a compact `also` for the other BASELINE configs.  MAX_LINE_BYTES is enforced
before the print (tests/test_bench_line_cpu.py feeds the builder a worst-case object).  The reference's counterpart is nine
criterion lines: /root/reference/benches/benchmark.rs:28-62."""
import json
import os

MAX_LINE_BYTES = 6000
MAX_STRING = 100
EXTRAS_FILE = "bench_extras.json"

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "hbm_frac_survey_8d", "traffic", "algorithmic_bytes_per_launch",
                 "kernel_ms", "traffic_measured_in_this_run")
CPU_KEYS = ("value", "unit", "cores", "single_thread_value", "kind", "sample")
REQUIRED = CONTRACT_KEYS + ("config", "roofline")


def _num(x, digits=6):
    """numbers to `digits` significant figures (a float's repr is up to 19 characters), everything else unchanged"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _short(s, limit=MAX_STRING):
    s = str(s)
    return s if len(s) <= limit else s[:limit - 1] + "~"


def _pick(obj, keys):
    out = {}
    for k in keys:
        if k in obj:
            v = obj[k]
            out[k] = _short(v) if isinstance(v, str) else _num(v)
    return out


def short_workload(name):
    """`config.workload` of the line: the workload's long description cut at its first clause"""
    return _short(name.split(", ")[0] if len(name) > MAX_STRING else name)


def compact_roofline(full):
    r = dict(full.get("roofline") or {})
    hv = (r.get("hbm_view") or {}).get("survey_8d_int32_model") or {}
    # the HBM view of the same launch under SURVEY 8(d)'s bytes: for an HBM-bound kernel that IS frac
    r.setdefault("hbm_frac_survey_8d", hv.get("frac", r.get("frac") if r.get("bound") == "hbm" else None))
    return _pick(r, ROOFLINE_KEYS)


def compact_cpu(full):
    cb = full.get("cpu_baseline")
    if not cb:
        return None
    return _pick(dict(cb, sample=cb.get("sample_short", cb.get("sample", ""))), CPU_KEYS)


def compact_line(full, also=None, extras_file=EXTRAS_FILE):
    """full = runner.run_one()'s object of the headline workload; also = {name: run_one() object} of the other configs."""
    line = _pick(full, CONTRACT_KEYS)
    cfg = full.get("config", {})
    line["config"] = {"workload": short_workload(cfg.get("workload", "")), "batch_per_gpu": cfg.get("batch_per_gpu"),
                      "parallelism": _short(cfg.get("parallelism", ""))}
    line["roofline"] = compact_roofline(full)
    cb = compact_cpu(full)
    if cb:
        line["cpu_baseline"] = cb
    pub = full.get("reference_published")
    if pub:
        line["reference_published"] = _pick(pub, ("value", "us_per_op"))
    if "ranks" in full:
        line["ranks"] = _pick(full["ranks"], ("min", "max"))
    if full.get("step_ms"):  # spread over the K timed steps (event marks on the launch stream): three numbers
        line["step_ms"] = {k: _num(full["step_ms"].get(k), 5) for k in ("min", "median", "max")}
    if also:
        out = {}
        for name, sub in also.items():
            rf = sub.get("roofline") or {}
            if name.startswith("verify_arith"):
                tr = rf.get("traffic")
                alg = rf.get("algorithmic_bytes_per_launch")
                o = {"value": sub.get("value"), "ms_per_step": sub.get("ms_per_step"), "kernel": rf.get("kernel"), "hbm_frac": rf.get("frac"),
                     "traffic_ratio": (tr / alg) if tr and alg else None}
            else:
                cbs = sub.get("cpu_baseline") or {}
                o = {"value": sub.get("value"), "ms_per_step": sub.get("ms_per_step"), "roofline_frac": rf.get("frac"), "roofline_bound": rf.get("bound"),
                     "roofline_kernel": rf.get("kernel"), "cpu_value": cbs.get("value"), "cpu_cores": cbs.get("cores")}
            out[name] = {k: (_short(v) if isinstance(v, str) else _num(v)) for k, v in o.items()}
        line["also"] = out
    if extras_file:
        line["extras_file"] = extras_file
    return line


def check_line(line):
    """the size and shape rules of the line; raises AssertionError"""
    text = json.dumps(line)
    assert len(text) <= MAX_LINE_BYTES, f"bench line is {len(text)} bytes (limit {MAX_LINE_BYTES})"
    assert "\n" not in text
    for k in REQUIRED:
        assert k in line, f"bench line lacks {k!r}"

    def walk(o, path):
        if isinstance(o, dict):
            for k, v in o.items():
                walk(v, f"{path}.{k}")
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                walk(v, f"{path}[{i}]")
        elif isinstance(o, str):
            assert len(o) <= MAX_STRING or path == ".metric", f"string of {len(o)} characters at {path}"
    walk(line, "")
    return text


def write_extras(root, extras, name=EXTRAS_FILE):
    """the side file: everything the line does not carry.  Written beside bench.py and, when the run is a gpurun call, under
    gpurun_out/ as well (the only directory that travels back).  Never fails the run."""
    paths = [os.path.join(root, name)]
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        paths.append(os.path.join(root, "gpurun_out", name))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(extras, f, indent=1, default=str)
            written.append(p)
        except OSError:
            pass
    return written


def emit(line):
    text = check_line(line)
    print(text, flush=True)
    return text
