#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a profiling round into the files kept under profiles/:

    python3 tools/pmc_summary.py <stats dir> <FETCH_SIZE dir> <WRITE_SIZE dir> <tag> <mc steps per bench step> <bench steps in pmc runs>

writes profiles/<tag>_pipeline_kernel_stats.csv (copy of the --stats kernel table),
profiles/<tag>_pmc_traffic.json (bytes per MC step, read by bench.py for roofline.traffic) and prints
the markdown table for profiles/<tag>_pmc_summary.md.  FETCH_SIZE / WRITE_SIZE are in KB; gfx950
tallies 128-B read requests as 64 B, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return f[-1]


def counter_sum(d, name, kernel="pipe_tick_kernel"):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n


def main():
    stats, fetch, write, tag, mc_steps, pmc_bench_steps = sys.argv[1:7]
    mc_steps, pmc_bench_steps = int(mc_steps), int(pmc_bench_steps)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    shutil.copy(find(stats, "*kernel_stats.csv"), os.path.join(out, tag + "_pipeline_kernel_stats.csv"))
    f_kb, f_n = counter_sum(fetch, "FETCH_SIZE")
    w_kb, w_n = counter_sum(write, "WRITE_SIZE")
    f_b = 2.0 * f_kb * 1024 / pmc_bench_steps
    w_b = w_kb * 1024 / pmc_bench_steps
    js = {"fetch_bytes_per_mc_step": f_b / mc_steps, "write_bytes_per_mc_step": w_b / mc_steps}
    json.dump(js, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"))
    print("| counter | tick-kernel dispatches | sum (KB) | per bench step, corrected (bytes) | per MC step (bytes) |")
    print("|---|---|---|---|---|")
    print("| FETCH_SIZE | %d | %.0f | %.3e (x2) | %.0f |" % (f_n, f_kb, f_b, f_b / mc_steps))
    print("| WRITE_SIZE | %d | %.0f | %.3e | %.0f |" % (w_n, w_kb, w_b, w_b / mc_steps))
    print("traffic per MC step: %.0f B" % ((f_b + w_b) / mc_steps))
    for r in csv.DictReader(open(os.path.join(out, tag + "_pipeline_kernel_stats.csv"))):
        if "pipe_tick" in r["Name"]:
            print("stats:", r["Name"], "calls", r["Calls"], "total ns", r["TotalDurationNs"], "avg ns", r["AverageNs"])


if __name__ == "__main__":
    main()
