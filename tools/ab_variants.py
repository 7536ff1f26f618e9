#!/usr/bin/env python3
"""Same-box A/B of library variants (round 6: the MLDSA_EXP memory-path experiments, `make -C fips204_amd/csrc variants`).

    python tools/ab_variants.py [--reps 2] [--variants base,1,2,...] [--workloads sign65,verify65,verify_arith44] [--out FILE]

Every (rep, variant, workload) is one `python bench.py --workload W --no-extras --no-pmc --no-cpu-baseline` process with the variant's
.so copied over fips204_amd/csrc/libmldsa_hip.so (restored at the end); the variants alternate inside a rep so that clock and box drift
hit all of them alike.  bench.py's own check against the CPU oracle runs in every process: a variant that changes one byte fails here.
Prints one line per run and a summary (median over reps) with the stage that should move; writes everything to --out.
"""
import argparse
import json
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fips204_amd", "csrc", "libmldsa_hip.so")
VAR = os.path.join(ROOT, "build", "variants")
STEPS = {"keygen65": (60, 10), "keygen44": (60, 10), "keygen87": (40, 10), "sign65": (40, 5), "verify65": (100, 10), "verify_arith44": (2000, 500), "sign44": (40, 5), "sign87": (30, 5), "verify87": (60, 10)}
STAGES = {"keygen65": (), "keygen44": (), "keygen87": (), "sign65": ("sign_w", "ctilde_hash", "sample_in_ball", "ntt_c", "sign_tail"), "verify65": ("verify_main", "expand_a", "ctilde_hash"), "verify_arith44": (),
          "sign44": ("sign_w", "ctilde_hash"), "sign87": ("sign_w", "ctilde_hash"), "verify87": ("verify_main", "expand_a")}


def run(workload, side):
    st, wu = STEPS[workload]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-extras", "--no-pmc", "--no-cpu-baseline",
           "--steps", str(st), "--warmup", str(wu), "--extras-file", side]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stderr or p.stdout)[-600:], "rc": p.returncode}
    d = json.loads(lines[-1])
    out = {"value": d["value"], "ms_per_step": d["ms_per_step"], "roofline_frac": (d.get("roofline") or {}).get("frac"),
           "kernel_ms": (d.get("roofline") or {}).get("kernel_ms")}
    try:
        h = json.load(open(os.path.join(ROOT, side)))["headline"]
        out["stages"] = {k: v for k, v in (h.get("stage_ms_per_step") or {}).items()}
        out["gap_ms"] = h.get("launch_gap_ms_per_step")
    except Exception as e:  # noqa: BLE001
        out["stages"] = {}
        out["side_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--variants", default="base,1,2,4,7,8,72,16,32")
    ap.add_argument("--workloads", default="sign65,verify65,verify_arith44")
    ap.add_argument("--out", default="gpurun_out/ab_variants.json")
    a = ap.parse_args()
    variants = a.variants.split(",")
    workloads = a.workloads.split(",")
    keep = LIB + ".ab_keep"
    shutil.copy2(LIB, keep)
    res = []
    try:
        for rep in range(a.reps):
            order = variants if rep % 2 == 0 else variants[::-1]
            for v in order:
                src = keep if v == "base" else os.path.join(VAR, f"libmldsa_hip_exp{v}.so")
                if not os.path.exists(src):
                    print(f"missing {src}", flush=True)
                    continue
                shutil.copy2(src, LIB)
                for w in workloads:
                    r = run(w, "ab_extras.json")
                    r.update(rep=rep, variant=v, workload=w)
                    res.append(r)
                    if "error" in r:
                        print(f"rep {rep} {v:>5} {w:<15} FAILED rc={r['rc']}: {r['error'][-300:]}", flush=True)
                        continue
                    st = " ".join(f"{k} {r['stages'].get(k, float('nan')):.4f}" for k in STAGES[w])
                    fr = f" frac {r['roofline_frac']:.4f} kernel_ms {r['kernel_ms']:.5f}" if r.get("roofline_frac") else ""
                    print(f"rep {rep} {v:>5} {w:<15} {r['value'] / 1e6:9.3f} M/s  {r['ms_per_step']:.4f} ms/step  {st}{fr}", flush=True)
    finally:
        shutil.copy2(keep, LIB)
        os.remove(keep)
        try:
            os.remove(os.path.join(ROOT, "ab_extras.json"))
        except OSError:
            pass
    # summary: median over reps, relative to base
    summ = {}
    for w in workloads:
        base = [r["value"] for r in res if r["workload"] == w and r["variant"] == "base" and "value" in r]
        for v in variants:
            vals = [r for r in res if r["workload"] == w and r["variant"] == v and "value" in r]
            if not vals:
                continue
            med = statistics.median(r["value"] for r in vals)
            e = {"median_value": med, "runs": len(vals), "vs_base": med / statistics.median(base) if base else None}
            for k in STAGES[w]:
                xs = [r["stages"][k] for r in vals if k in r.get("stages", {})]
                if xs:
                    e[k + "_ms"] = statistics.median(xs)
            if vals[0].get("roofline_frac"):
                e["roofline_frac"] = statistics.median(r["roofline_frac"] for r in vals)
            summ.setdefault(w, {})[v] = e
    print("\nsummary (median over reps)")
    for w in workloads:
        for v, e in summ.get(w, {}).items():
            rest = " ".join(f"{k} {x:.4f}" for k, x in e.items() if k.endswith("_ms") or k == "roofline_frac")
            print(f"{w:<15} {v:>5}  {e['median_value'] / 1e6:9.3f} M/s  x{e['vs_base']:.4f}  {rest}")
    os.makedirs(os.path.dirname(os.path.join(ROOT, a.out)), exist_ok=True)
    json.dump({"runs": res, "summary": summ}, open(os.path.join(ROOT, a.out), "w"), indent=1)


if __name__ == "__main__":
    main()
