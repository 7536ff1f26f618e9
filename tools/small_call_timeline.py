#!/usr/bin/env python3
"""Decompose the latency of SMALL calls from a rocprofv3 kernel trace.

    tools/small_call_timeline.py <kernel_trace.csv> <wall.json> [skip_calls]

wall.json = what tools/latency_probe.py wrote (WALL_JSON=...; CALL_GAP_US=400 so that consecutive calls are separated on the
device by more than any gap inside a call).  The trace's kernels are grouped into calls at every idle stretch of > 150 us; the
first `skip_calls` groups (warm-up, set-up kernels) are dropped and the LAST len(wall_us) groups are matched with the wall times.
Per call: span = first kernel start .. last kernel end; busy = length of the union of kernel intervals (kernels of helper streams
overlap); gaps = span - busy (launch / boundary gaps on the device); host = wall - span (API call, doorbell, completion signal,
synchronize wake-up).  Prints one JSON object: medians, the per-kernel table (median duration, launches per call) and the
critical chain in launch order."""
import csv
import json
import statistics as st
import sys
from collections import defaultdict


def load(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    return rows


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    trace, wall_path = sys.argv[1], sys.argv[2]
    wall = json.load(open(wall_path))
    rows = load(trace)
    groups, cur, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[0] - last_end > 150_000:
            groups.append(cur)
            cur = []
        cur.append(r)
        last_end = max(last_end or 0, r[1])
    if cur:
        groups.append(cur)
    n = len(wall["wall_us"])
    groups = groups[-n:]
    per_call = []
    kern = defaultdict(list)
    count = defaultdict(list)
    for g, w in zip(groups, wall["wall_us"][-len(groups):]):
        span = (max(r[1] for r in g) - g[0][0]) / 1e3
        busy, end = 0, None
        for s, e, *_ in g:
            if end is None or s > end:
                busy += e - s
                end = e
            elif e > end:
                busy += e - end
                end = e
        busy /= 1e3
        per_call.append(dict(wall_us=w, span_us=span, busy_us=busy, gaps_us=span - busy, host_us=w - span, kernels=len(g),
                             sum_kernel_us=sum(r[1] - r[0] for r in g) / 1e3))
        c = defaultdict(int)
        for s, e, name, *_ in g:
            kern[short(name)].append((e - s) / 1e3)
            c[short(name)] += 1
        for k, v in c.items():
            count[k].append(v)
    med = lambda key: round(st.median(x[key] for x in per_call), 2)
    # the chain of a typical call: the group whose span is the median one
    typical = sorted(zip((x["span_us"] for x in per_call), range(len(groups))))[len(groups) // 2][1]
    t0 = groups[typical][0][0]
    chain = [dict(kernel=short(nm), start_us=round((s - t0) / 1e3, 2), dur_us=round((e - s) / 1e3, 2), queue=q, stream=sid) for s, e, nm, q, sid in groups[typical]]
    out = dict(op=wall["op"], n_ops=wall["n_ops"], calls=len(groups),
               median=dict(wall_us=med("wall_us"), device_span_us=med("span_us"), device_busy_us=med("busy_us"), device_gaps_us=med("gaps_us"),
                           host_and_sync_us=med("host_us"), kernels_per_call=med("kernels"), sum_of_kernel_durations_us=med("sum_kernel_us")),
               kernels={k: dict(median_us=round(st.median(v), 2), per_call=st.median(count[k])) for k, v in sorted(kern.items(), key=lambda kv: -st.median(kv[1]))},
               typical_call=chain)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
