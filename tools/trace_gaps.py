#!/usr/bin/env python3
"""Durations of the pipeline tick kernels and the gaps between them from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/pipeline_stamps.py
    python3 tools/trace_gaps.py gpurun_out/trace"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ticks = [r for r in rows if "pipe_tick" in r["Kernel_Name"]]
ticks.sort(key=lambda r: int(r["Start_Timestamp"]))
# last launch sequence only (the timed one)
seq = ticks[len(ticks)//2:]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seq[:-1], seq[1:])]
print("ticks", len(seq), "mean dur us %.2f" % (sum(durs)/len(durs)), "mean gap us %.2f" % (sum(gaps)/len(gaps)), "total ms %.3f" % ((int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"]))/1e6))
print("durs", " ".join("%.0f" % d for d in durs[:40]))
print("gaps", " ".join("%.1f" % g for g in gaps[:40]))
