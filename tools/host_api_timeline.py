#!/usr/bin/env python3
"""Host-side view of ONE small call: the HIP API calls the library makes (rocprofv3 --hip-trace) with their start offsets and durations,
beside the kernels (--kernel-trace).  usage (on the GPU box):
    cd /tmp && CALL_GAP_US=400 rocprofv3 --hip-trace --kernel-trace --output-format csv -d DIR -o t -- python3 $REPO/tools/latency_probe.py sign 1 40
    python3 tools/host_api_timeline.py DIR > timeline.txt
Picks the call of median device span (calls are separated by the probe's 400 us pause)."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    api = list(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
    ker = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]) for r in api]
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL", r["Kernel_Name"][:50]) for r in ker]
    ev.sort()
    # group by gaps > 200 us between consecutive kernel launches
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ker)
    groups = [[ks[0]]]
    for a, b in ks[1:]:
        if a - groups[-1][-1][1] > 200_000:
            groups.append([])
        groups[-1].append((a, b))
    groups = [g for g in groups[len(groups) // 4:] if g]  # (past the warm-up calls)
    g = sorted(groups, key=lambda g: g[-1][1] - g[0][0])[len(groups) // 2]  # the call of median device span
    t0, t1 = g[0][0] - 120_000, g[-1][1] + 60_000
    first = None
    for a, b, kind, name in ev:
        if a < t0 or a > t1:
            continue
        if first is None:
            first = a
        print(f"{(a - first) / 1e3:9.2f} us  +{(b - a) / 1e3:7.2f}  {kind:6s} {name}")


if __name__ == "__main__":
    main()
