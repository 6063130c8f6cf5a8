#!/usr/bin/env python3
"""Turns the raw rocprofv3 outputs of tools/profile.sh <round> (gpurun_out/<round>prof/) into the files kept under profiles/:

    python3 tools/pmc_summary.py <round> [commit]           e.g.  python3 tools/pmc_summary.py r04 abc1234

  profiles/<round>_<tag>_kernel_stats.csv      copy of the --stats kernel table of each workload
  profiles/<round>_pmc_traffic.json            config 2: FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE bytes per MC step
  profiles/<round>_valu_per_step.json          per config: VALU wave-instructions per MC step (SQ_INSTS_VALU)
  profiles/<round>_summary.md                  the tables
"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"
RAW = os.path.join(ROOT, "gpurun_out", ROUND + "prof")
OUT = os.path.join(ROOT, "profiles")


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return f[-1] if f else None


def bench_line(tag):
    """The JSON line bench.py printed under the --stats pass (MC steps per launch, launch time)."""
    p = os.path.join(RAW, tag + "_stats.log")
    for l in reversed(open(p).read().splitlines()):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no bench line in " + p)


def counters(tag, group, kernel_rx, skip_init=False):
    """Sums per counter over the dispatches of the named kernel.  skip_init (pipeline): the first tick dispatch
    behind every pipe_reset_kernel dispatch evaluates the N initial rows of every chain — chain initialisation, not
    MC steps — and is left out (its sum is returned separately)."""
    f = find(os.path.join(RAW, "%s_%s" % (tag, group)), "*counter_collection.csv")
    if not f:
        return {}, 0, {}
    rows = list(csv.DictReader(open(f)))
    order = sorted(set((int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in rows))
    init_ids, after_reset = set(), False
    for did, name in order:
        if "pipe_reset_kernel" in name:
            after_reset = True
        elif after_reset and re.search(kernel_rx, name):
            init_ids.add(did); after_reset = False
    tot, init, disp = {}, {}, set()
    for r in rows:
        if not re.search(kernel_rx, r["Kernel_Name"]):
            continue
        tgt = init if (skip_init and int(r["Dispatch_Id"]) in init_ids) else tot
        tgt[r["Counter_Name"]] = tgt.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp.add(r["Dispatch_Id"])
    return tot, len(disp), init


def main():
    commit = sys.argv[2] if len(sys.argv) > 2 else "?"
    md = ["# Profiles %s (MI355X, rocprofv3; raw passes by tools/profile.sh, this file by tools/pmc_summary.py)" % ROUND, "",
          "Counter passes run with `--kernel-trace --pmc ...` only, the bench program directly after `--`; every pass is its own run.",
          "SQ_* counters are sums over all shader engines of the dispatches of the named kernel; *_CYCLES in quad-cycles.",
          "Kernel durations under `--kernel-trace` run ~6 %% above the un-profiled ones (config 2: 107 ticks x the average below = the launch time "
          "the bench reports UNDER the same pass; profiles/%s_bench_line.json is bench.py without a profiler)." % ROUND, ""]
    valu = {}
    works = [("c2", 2, r"pipe_tick_kernel"), ("c3", 3, r"pipe_tick_kernel"), ("c4", 4, r"pipe_tick_kernel"), ("c5", 5, r"pipe_tick_kernel"),
             ("wave8192", None, r"chain_wave_kernel")]
    for tag, cfg, rx in works:
        if not os.path.exists(os.path.join(RAW, tag + "_stats.log")):
            continue
        b = bench_line(tag)
        launches, steps = b["launch_ms"]["n"], b["value"] * b["timed_region_s"]
        st = find(os.path.join(RAW, tag + "_stats"), "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(OUT, ROUND + "_%s_kernel_stats.csv" % tag))
        md += ["## %s — %s" % (tag, b["config"]["workload"]), "",
               "bench under the --stats pass: %.3e MC steps/s, launch %.3f ms mean (min %.3f, max %.3f), exec mode %s, window %s"
               % (b["value"], b["launch_ms"]["mean"], b["launch_ms"]["min"], b["launch_ms"]["max"], b["config"]["exec_mode"], b["config"]["window"]), ""]
        if st:
            md += ["| kernel | calls | total ms | average us |", "|---|---|---|---|"]
            for r in csv.DictReader(open(st)):
                if float(r["Percentage"]) > 0.5:
                    md.append("| %s | %s | %.3f | %.2f |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) * 1e-6, float(r["AverageNs"]) * 1e-3))
            md.append("")
        # counters are taken over ALL launches of the counter run: warm-up included; bench line of that run gives its step count
        rows = []
        for group in ("sq1", "sq2"):
            logp = os.path.join(RAW, "%s_%s.log" % (tag, group))
            if not os.path.exists(logp):
                continue
            bl = None
            for l in reversed(open(logp).read().splitlines()):
                if l.startswith("{"):
                    bl = json.loads(l); break
            if bl is None:
                continue
            tot, nd, init = counters(tag, group, rx, skip_init=cfg is not None)
            if group == "sq1" and init.get("SQ_INSTS_VALU"):
                md += ["(chain initialisation, left out of the table: %.3e VALU wave-instructions = %.1f %% of the run's)"
                       % (init["SQ_INSTS_VALU"], 100 * init["SQ_INSTS_VALU"] / (init["SQ_INSTS_VALU"] + tot.get("SQ_INSTS_VALU", 0))), ""]
            # MC steps executed in the counter run = (timed + warm-up launches) x steps per launch
            per_launch = bl["value"] * bl["timed_region_s"] / max(bl["launch_ms"]["n"], 1)
            n_launch = (bl["steps"] + bl["warmup"]) * bl["config"]["launches_per_step"]
            mc = per_launch * n_launch
            for k in sorted(tot):
                rows.append((k, tot[k], tot[k] / mc))
            if group == "sq1" and "SQ_INSTS_VALU" in tot and cfg is not None:
                valu[str(cfg)] = {"valu_wave_instr_per_mc_step": tot["SQ_INSTS_VALU"] / mc, "commit": commit}
            if group == "sq1" and "SQ_INSTS_VALU" in tot and cfg is None:
                valu[tag] = {"valu_wave_instr_per_mc_step": tot["SQ_INSTS_VALU"] / mc, "commit": commit}
        if rows:
            md += ["| counter | sum over the run | per MC step |", "|---|---|---|"]
            md += ["| %s | %.4e | %.2f |" % r for r in rows]
            d = dict((r[0], r[1]) for r in rows)
            if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
                md += ["", "shares of SQ_WAVE_CYCLES: ACTIVE_INST_ANY %.1f %%, WAIT_INST_ANY %.1f %%, WAIT_ANY %.1f %%; ACTIVE_INST_VALU %.1f %%, WAIT_INST_LDS %.1f %%"
                       % tuple(100 * d.get(k, 0) / d["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"))]
            md.append("")
    # memory-side traffic per config (FETCH_SIZE and WRITE_SIZE each in a pass of its own)
    traffic = {}
    md += ["## memory-side traffic (FETCH_SIZE x2: gfx950 tallies 128-B requests at 64 B; WRITE_SIZE as is)", "",
           "| config | counter | dispatches | sum (KB) | bytes per MC step |", "|---|---|---|---|---|"]
    for tag, cfg, rx in works:
        fb = os.path.join(RAW, tag + "_fetch.log")
        if cfg is None or not os.path.exists(fb):
            continue
        bl = [json.loads(l) for l in open(fb).read().splitlines() if l.startswith("{")][-1]
        per_launch = bl["value"] * bl["timed_region_s"] / max(bl["launch_ms"]["n"], 1)
        mc = per_launch * (bl["steps"] + bl["warmup"]) * bl["config"]["launches_per_step"]
        f, nf, _ = counters(tag, "fetch", rx)
        w, nw, _ = counters(tag, "write", rx)
        fbytes = 2.0 * f.get("FETCH_SIZE", 0) * 1024 / mc
        wbytes = w.get("WRITE_SIZE", 0) * 1024 / mc
        traffic[str(cfg)] = {"fetch_bytes_per_mc_step": fbytes, "write_bytes_per_mc_step": wbytes, "commit": commit,
                             "note": "chain initialisation included (all dispatches of the run)"}
        md += ["| %s | FETCH_SIZE | %d | %.0f | %.0f |" % (tag, nf, f.get("FETCH_SIZE", 0), fbytes),
               "| %s | WRITE_SIZE | %d | %.0f | %.0f |" % (tag, nw, w.get("WRITE_SIZE", 0), wbytes)]
    md += ["", "algorithmic figure of SURVEY 8d: 40 Q bytes per MC step = 20480 B at 512 q, 40960 B at 1024 q", ""]
    # the kernel sources these counters describe (bench.py compares: a kernel change without a profile refresh shows in the bench line)
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_sha16
    sha = kernel_sources_sha16()
    for d in (traffic, valu):
        for v in d.values():
            v["csrc_sha16"] = sha
    json.dump(traffic, open(os.path.join(OUT, ROUND + "_pmc_traffic.json"), "w"), indent=1)
    json.dump(valu, open(os.path.join(OUT, ROUND + "_valu_per_step.json"), "w"), indent=1)
    open(os.path.join(OUT, ROUND + "_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
