#!/usr/bin/env python3
"""tools/cmp_dumps.py a.npz b.npz ... : are the arrays of tools/time_c2.py's dumps identical to the first one's?"""
import sys, numpy as np
a = np.load(sys.argv[1])
for f in sys.argv[2:]:
    b = np.load(f)
    print(f, {k: bool(np.array_equal(a[k], b[k])) for k in a.files}, "moves", int(b["moves"].sum()))
