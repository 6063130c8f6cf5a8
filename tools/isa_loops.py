#!/usr/bin/env python3
"""Innermost loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): instruction counts, fp64 VALU and scratch
(spill) instructions per loop body — where a kernel's spills sit relative to its hot loops.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ibuild/csrc -DMCSAS_M=1 -S --cuda-device-only -o /tmp/pipe_m1.s mcsas_amd/csrc/kern_pipe.hip
    tools/isa_loops.py /tmp/pipe_m1.s 'Li1ELi8ELb1' [min VALU per loop]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
inner = [(a, b) for (a, b) in loops if not any(a <= c and d <= b and (c, d) != (a, b) for (c, d) in loops)]
tot_s = sum("scratch_" in l for l in body)
print("%s: %d lines, %d scratch instructions in all, %d loops (%d innermost)" % (body[0].split(":")[0], len(body), tot_s, len(loops), len(inner)))
for a, b in sorted(inner):
    seg = body[a:b + 1]
    nv = sum(re.match(r"\s+v_", x) is not None for x in seg)
    nf = sum(re.search(r"\sv_(fma|mul|add)_f64", x) is not None for x in seg)
    ns = sum("scratch_" in x for x in seg)
    if nv >= minv:
        print("  lines %6d-%6d: %5d instr, %5d VALU (%5d f64 fma/mul/add), %3d scratch" % (a, b, b - a, nv, nf, ns))
