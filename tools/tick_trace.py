#!/usr/bin/env python3
"""Durations of the tick kernels of one analysis, in launch order, from a rocprofv3 kernel trace (tools/tick_trace.sh)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "pipe_tick" in r["Kernel_Name"] or "pipe_reset" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs, cur = [], []
for r in rows:
    if "pipe_reset" in r["Kernel_Name"]:
        if cur: runs.append(cur)
        cur = []
    else:
        cur.append(r)
if cur: runs.append(cur)
for k, run in enumerate(runs[-4:]):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in run]
    span = (int(run[-1]["End_Timestamp"]) - int(run[0]["Start_Timestamp"])) / 1e3
    print("analysis %d: %d ticks, sum %.1f us, first start to last end %.1f us" % (k, len(d), sum(d), span))
    print(" ".join("%.1f" % x for x in d))
