#!/usr/bin/env python
"""Summary of a rocprofv3 --kernel-trace of bench.py (tools/shard_timeline.sh): device-busy time against wall time of the
last traced step, the idle gaps between consecutive kernels (launch latency of the Python-driven update shows up
here) and the kernels by total time.  A "step" is found from the GAE kernel, which runs once per step."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src = sys.argv[1]
    f = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "gae" in r[2].lower() and "scan" not in r[2].lower()]
    if len(marks) < 3:
        marks = [i for i, r in enumerate(rows) if "gae" in r[2].lower()]
    lo, hi = marks[-2], marks[-1]                   # one full step: GAE launch to the next GAE launch
    step = rows[lo:hi]
    wall = rows[hi][0] - rows[lo][0]
    busy = 0
    cur_end = step[0][0]
    gaps = []
    for s, e, name in step:
        if s > cur_end:
            gaps.append((s - cur_end, name))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, name in step:
        k = name.split("(")[0][-70:]
        by[k][0] += e - s
        by[k][1] += 1
    top = sorted(by.items(), key=lambda kv: -kv[1][0])
    small = sum(v[0] for k, v in by.items() if v[0] / v[1] < 100e3)
    after = collections.defaultdict(lambda: [0, 0])
    for g, name in gaps:
        k = name.split("(")[0][-60:]
        after[k][0] += g
        after[k][1] += 1
    out = {"trace": os.path.relpath(f, src), "kernels_in_step": len(step), "wall_ms": wall / 1e6, "busy_ms": busy / 1e6,
           "idle_ms": (wall - busy) / 1e6, "gaps": len(gaps), "gaps_over_20us": sum(1 for g, _ in gaps if g > 20e3),
           "idle_ms_in_gaps_over_20us": sum(g for g, _ in gaps if g > 20e3) / 1e6,
           "device_ms_of_kernels_under_100us_avg": small / 1e6,
           "top_kernels_ms": [[k, round(v[0] / 1e6, 3), v[1]] for k, v in top[:25]],
           "idle_before_kernel_ms": [[k, round(v[0] / 1e6, 3), v[1]] for k, v in sorted(after.items(), key=lambda kv: -kv[1][0])[:25]]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
