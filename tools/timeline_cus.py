#!/usr/bin/env python3
"""Which compute unit did every block of the dumped tick run on?  (HW_ID of wave 0 of each block: cu [11:8], sh [12], se [15:13];
XCC_ID [3:0])  usage: tools/timeline_cus.py log [scan blocks R]"""
import re, sys, collections
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
blocks = {}
for line in open(sys.argv[1]):
    m = re.match(r"\[mcsas timeline\] (\d+) (\d+) ([\d.]+) ([\d.]+) ([0-9a-f]+) (\d+)", line)
    if m and int(m[2]) == 0:
        hw = int(m[5], 16)
        blocks[int(m[1])] = (int(m[6]) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, float(m[3]), float(m[4]))
percu = collections.defaultdict(list)
for b, (x, se, sh, cu, s, e) in blocks.items(): percu[(x, se, sh, cu)].append(b)
print("%d blocks on %d distinct CUs" % (len(blocks), len(percu)))
print("CUs seen per XCC:", sorted(collections.Counter(k[0] for k in percu).items()))
multi = {k: v for k, v in percu.items() if len(v) > 1}
print("CUs holding more than one block: %d" % len(multi))
for k, v in sorted(multi.items()):
    print("  xcc %d se %d sh %d cu %d:" % k, ["%s%d (%.1f-%.1f us)" % ("scan " if b < R else "prod ", b, blocks[b][4], blocks[b][5]) for b in v])
se_cu = collections.Counter((k[1], k[3]) for k in percu)
print("distinct (se, cu) ids:", len(se_cu), sorted(se_cu)[:48])
