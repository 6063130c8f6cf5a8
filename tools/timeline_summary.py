#!/usr/bin/env python3
"""Summarise the '[mcsas timeline]' lines of a stamps-build run (MCSAS_TIMELINE_TICK=<tick>): when the waves of one
tick start and end, per block kind and per XCD."""
import sys, collections
import numpy as np
rows = [l.split()[2:] for l in open(sys.argv[1]) if l.startswith("[mcsas timeline]") and l.split()[2].isdigit()]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
a = np.array([[int(r[0]), int(r[1]), float(r[2]), float(r[3]), int(r[4], 16), int(r[5])] for r in rows])
blk, wave, t0, t1, hw, xcc = a.T
xcc = xcc.astype(int) & 15
scan = blk < R
def desc(x): return "min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
print("waves", len(a), " kernel span %.1f us" % (t1.max() - t0.min()))
print("scan  blocks: start", desc(t0[scan]), "| end", desc(t1[scan]), "| duration", desc((t1 - t0)[scan]))
print("prod  blocks: start", desc(t0[~scan]), "| end", desc(t1[~scan]), "| duration", desc((t1 - t0)[~scan]))
pb = collections.defaultdict(list)
for b, e in zip(blk[~scan], t1[~scan]): pb[int(b)].append(e)
ends = np.array([max(v) for v in pb.values()]); firsts = np.array([min(v) for v in pb.values()])
print("prod block end (slowest wave):", desc(ends), "| fastest wave of the block:", desc(firsts))
for x in sorted(set(xcc)):
    m = xcc == x
    print("xcc %d: blocks %d (scan %d)  last end %.1f  mean prod end %.1f" % (x, len(set(blk[m])), len(set(blk[m & scan])), t1[m].max(), t1[m & ~scan].mean() if (m & ~scan).any() else 0))
cu = collections.Counter((int(x), int(h) >> 8 & 0xf, int(h) >> 13 & 0x7) for x, h, w in zip(xcc, hw, wave) if w == 0)   # (xcc, cu_id, se_id) of wave 0
print("distinct (xcc, cu, se) of wave 0:", len(cu), " max blocks on one:", max(cu.values()))
late = np.argsort(-t1)[:8]
print("latest waves:", [(int(blk[i]), int(wave[i]), round(float(t1[i]), 1), int(xcc[i])) for i in late])
