"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into profiles/pmc_<workload>.json.

Usage: python tools/pmc_summary.py <workload> <kernel substring> <fetch_dir> <write_dir> [mean]
("mean" averages over the dispatches instead of taking the median: for kernels whose launches differ
in size, e.g. the signer's per-round kernels, matching bench.py's average bytes per launch)
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores.  Both counters are in KiB.  Values are per launch (median
over the dispatches of the named kernel)."""
import csv, glob, json, os, statistics, sys


def counter_values(d, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


def main():
    workload, kernel, fdir, wdir = sys.argv[1:5]
    stat = statistics.fmean if (len(sys.argv) > 5 and sys.argv[5] == "mean") else statistics.median
    fetch = counter_values(fdir, kernel, "FETCH_SIZE")
    write = counter_values(wdir, kernel, "WRITE_SIZE")
    if not fetch or not write:
        raise SystemExit(f"no counter rows for {kernel!r}: fetch={len(fetch)} write={len(write)}")
    f_kib, w_kib = stat(fetch), stat(write)
    out = {
        "workload": workload, "kernel": kernel, "dispatches": [len(fetch), len(write)], "statistic": stat.__name__,
        "FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB": w_kib,
        "fetch_correction": "x2 (gfx950 counts 128-B requests as 64 B for wide streaming reads)",
        "hbm_read_bytes_per_launch": 2 * f_kib * 1024, "hbm_write_bytes_per_launch": w_kib * 1024,
        "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024,
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", f"pmc_{workload}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
