#!/usr/bin/env python3
"""Where does a host-fed signing call (mldsa_sign_host, page-locked buffers, direct export) spend its time?

    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d D -o t -- python3 tools/hostfed_sign.py 65536 4
    python3 tools/hostfed_timeline.py D

The kernels (and copies, when traced) of the LAST call are taken (calls are separated by the host-side work between them: groups are cut at
idle stretches of > 300 us).  Printed as JSON: span of the call on the device, busy time of the round chain (union of every kernel but the
export), time under k_export_done, how much of the export lies behind the last round kernel (the unhidden tail), the gaps of the round
chain, and the per-kernel totals.  The device-resident call (bench.py --workload sign65) is the yardstick: same kernels, no export."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mldsa::", "")))
    copies = []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "copy"))))
    rows.sort()
    copies.sort()
    return rows, copies


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    rows, copies = load(sys.argv[1])
    groups, cur, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[0] - last_end > 300_000 and any("k_expand_a" in x[2] for x in cur):
            groups.append(cur)
            cur = []
        cur.append(r)
        last_end = max(last_end or 0, r[1])
    if cur:
        groups.append(cur)
    groups = [g for g in groups if sum(1 for x in g if "k_verify_arith" in x[2]) >= 5]  # signing calls (several rounds of sign_w)
    g = groups[-1]
    t0, t1 = g[0][0], max(x[1] for x in g)
    chain = [(s, e) for s, e, n in g if "k_export_done" not in n]
    export = [(s, e) for s, e, n in g if "k_export_done" in n]
    chain_end = max(e for s, e in chain)
    per = defaultdict(lambda: [0, 0.0])
    for s, e, n in g:
        per[n][0] += 1
        per[n][1] += (e - s) / 1e6
    # gaps of the round chain: idle stretches between consecutive chain kernels (by end-time order)
    ch = sorted(chain)
    gaps, end = [], ch[0][1]
    for s, e in ch[1:]:
        if s > end:
            gaps.append((s - end) / 1e3)
        end = max(end, e)
    cp = [(s, e, d) for s, e, d in copies if t0 - 2_000_000 <= s <= t1 + 2_000_000]
    out = {
        "calls_seen": len(groups),
        "span_ms": (t1 - t0) / 1e6,
        "round_chain_busy_ms": union(chain) / 1e6,
        "round_chain_end_ms": (chain_end - t0) / 1e6,
        "export_busy_ms": union(export) / 1e6,
        "export_launches": len(export),
        "export_after_last_round_kernel_ms": max(0.0, (max((e for s, e in export), default=chain_end) - chain_end) / 1e6),
        "chain_gaps_ms_total": sum(gaps) / 1e3,
        "chain_gaps_over_20us": sorted((round(x, 1) for x in gaps if x > 20), reverse=True)[:12],
        "copies_near_call": [{"ms_from_start": round((s - t0) / 1e6, 3), "ms": round((e - s) / 1e6, 3), "what": d} for s, e, d in cp][:20],
        "kernels": {n: {"launches": c, "ms": round(ms, 3)} for n, (c, ms) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
