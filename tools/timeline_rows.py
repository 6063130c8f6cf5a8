#!/usr/bin/env python3
"""Per-row marks of the producer waves in a stamps-build timeline (tools/r3_tl.sh): for waves 0 and 4 of every producer block,
the times (us from the wave's start) of: start-up marks 18..21, row l evaluated / row l done (marks 4..11, l = 0..3), rows of
sub-window 0 done, Gram 0 done, rows 1, Gram 1, end."""
import sys
import numpy as np
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = [l.split()[2:] for l in open(sys.argv[1]) if l.startswith("[mcsas timeline]") and l.split()[2].isdigit()]
out = {0: [], 4: []}
for r in rows:
    blk, wave = int(r[0]), int(r[1])
    if blk < R or wave not in out:
        continue
    t0, t1 = float(r[2]), float(r[3])
    m = [float(x) for x in r[6:]]                    # marks 0..25
    rel = lambda i: (m[i] - t0) if m[i] else np.nan
    out[wave].append([rel(22), rel(23), rel(25), rel(18), rel(19), rel(20), rel(21)] + [rel(4 + i) for i in range(8)] + [rel(0), rel(1), rel(2), rel(3), t1 - t0])
names = ["entry", "snap", "tables", "st0", "props", "stale", "loop"] + ["r%d%s" % (i // 2, "e" if i % 2 == 0 else "d") for i in range(8)] + ["rows0", "gram0", "rows1", "gram1", "end"]
for w in (0, 4):
    a = np.array(out[w])
    print("wave %d of %d producer blocks: median (p10 .. p90) us from the wave's start" % (w, len(a)))
    for k, n in enumerate(names):
        c = a[:, k]; c = c[~np.isnan(c)]
        if len(c):
            print("  %-7s %6.2f  (%5.2f .. %5.2f)" % (n, np.median(c), np.percentile(c, 10), np.percentile(c, 90)))
