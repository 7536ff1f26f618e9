"""Summarise rocprofv3 --pmc runs of bench.py into profiles/.

  python tools/pmc_summary.py hbm <workload> <fetch_dir> <write_dir> <round>
      FETCH_SIZE / WRITE_SIZE (separate passes) -> profiles/<round>_pmc_<workload>.json with the HBM bytes per launch
      of every pipeline stage's kernel ("by_stage", mean over the dispatches: the signer's per-round launches differ
      in size, and bench.py's model is total bytes / launches too) and of the workload's dominant kernel.
      Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a
      wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
      Both counters are in KiB.
  python tools/pmc_summary.py sq <name> <dir> <round>
      SQ_* counters -> profiles/<round>_sq_<name>.json: per kernel, mean per dispatch, and the derived ratios
      (VALU instructions per wave-cycle, share of wave cycles with a VALU instruction active / waiting to issue).
"""
import csv, glob, json, os, re, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_KERNELS = {  # pipeline stage -> substring of the kernel name that runs it
    "expand_a": "k_expand_a<", "verify_main": "k_verify_main<", "expand_mask": "k_expand_mask<", "sign_tail": "k_sign_tail<",
    "ctilde_hash": "k_shake256_2<", "sample_in_ball": "k_sample_in_ball<",
}
DOMINANT = {"verify65": "expand_a", "verify44": "expand_a", "verify87": "expand_a", "sign65": "sign_w", "sign44": "sign_w",
            "sign87": "sign_w", "verify_arith44": "verify_arith"}


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))


def stage_of(kernel):
    m = re.search(r"k_verify_arith<\d+, \d+, (true|false)(?:, \d+, (?:true|false), (true|false))?", kernel)
    if m:  # <K, L, HAS_C, W1, APACK, KG>: HAS_C = the config-2 / verify form, KG = key generation's t = A s1 + s2
        return "verify_arith" if m.group(1) == "true" else "keygen_t" if m.group(2) == "true" else "sign_w"
    for st, sub in STAGE_KERNELS.items():
        if sub in kernel:
            return st
    return None


def hbm_compute(workload, fdir, wdir):
    """the summary object of one FETCH_SIZE and one WRITE_SIZE pass (bench.py's live measurement uses it too)"""
    per = {}
    for d, counter in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
        for r in rows(d):
            st = stage_of(r.get("Kernel_Name", ""))
            if st and r.get("Counter_Name") == counter:
                per.setdefault(st, {}).setdefault(counter, []).append(float(r["Counter_Value"]))
    by_stage, detail = {}, {}
    for st, c in per.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        f_kib, w_kib = statistics.fmean(c["FETCH_SIZE"]), statistics.fmean(c["WRITE_SIZE"])
        by_stage[st] = (2 * f_kib + w_kib) * 1024
        detail[st] = {"dispatches": [len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"])], "FETCH_SIZE_KiB_raw_mean": f_kib,
                      "WRITE_SIZE_KiB_mean": w_kib, "hbm_read_bytes_per_launch": 2 * f_kib * 1024,
                      "hbm_write_bytes_per_launch": w_kib * 1024}
    dom = DOMINANT.get(workload)
    out = {"workload": workload, "statistic": "mean over the dispatches of each kernel",
           "fetch_correction": "x2 (gfx950 counts 128-B requests as 64 B for wide streaming reads; dwordx3 reads of the 24-bit A_hat "
                               "are booked like dwordx4, so read figures of kernels that stream it are upper bounds)",
           "dominant_stage": dom, "hbm_bytes_per_launch": by_stage.get(dom), "by_stage": by_stage, "detail": detail}
    return out


def hbm(workload, fdir, wdir, rnd):
    out = hbm_compute(workload, fdir, wdir)
    by_stage = out["by_stage"]
    path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps({k: round(v / 1e6, 2) for k, v in by_stage.items()}), "MB per launch")


def sq(name, d, rnd):
    per = {}
    for r in rows(d):
        k = r.get("Kernel_Name", "")
        per.setdefault(k, {}).setdefault(r.get("Counter_Name"), []).append(float(r["Counter_Value"]))
    out = {"name": name, "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over waves resp. SQs "
                                  "(MI355X_MICROARCH.md, cycle constants table); SQ_INSTS_VALU counts wave-level instructions",
           "kernels": {}}
    for k, c in per.items():
        if not k.startswith("void mldsa::") and "mldsa" not in k:
            continue
        m = {n: statistics.fmean(v) for n, v in c.items()}
        m["dispatches"] = max(len(v) for v in c.values())
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in m:
                    m[n + "_per_WAVE_CYCLE"] = m[n] / wc
            if "SQ_INSTS_VALU" in m:
                m["VALU_insts_per_wave_quad_cycle"] = m["SQ_INSTS_VALU"] / wc
        out["kernels"][k] = m
    path = os.path.join(ROOT, "profiles", f"{rnd}_sq_{name}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    for k, m in out["kernels"].items():
        print("  ", k[:70], {a: (round(b, 4) if b < 100 else round(b)) for a, b in m.items()})


if __name__ == "__main__":
    if sys.argv[1] == "hbm":
        hbm(*sys.argv[2:6])
    elif sys.argv[1] == "sq":
        sq(*sys.argv[2:5])
    else:
        raise SystemExit(__doc__)
