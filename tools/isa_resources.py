#!/usr/bin/env python3
"""ISA resource table of every chain-kernel instantiation (VGPRs, AGPRs, SGPRs, spills, scratch, occupancy): compiles the kernel
translation units of mcsas_amd/csrc with the Makefile's flags plus -Rpass-analysis=kernel-resource-usage (hipcc cross-compiles for
gfx950 without a GPU) and writes profiles/rNN_isa_resources.md.

    python3 tools/isa_resources.py r04 [-j 4]
"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcsas_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result".split()
MODEL_NAMES = {0: "Sphere", 1: "CylindersIsotropic", 2: "EllipsoidalCoreShell", 3: "Kholodenko", 4: "EllipsoidsIsotropic",
               5: "SphericalCoreShell", 6: "GaussianChain", 7: "LMADenseSphere"}


def models():
    txt = open(os.path.join(CSRC, "model_list.h")).read()
    line = [l for l in txt.splitlines() if l.startswith("#define MCSAS_FOR_MODELS")][0]
    return [int(x) for x in re.findall(r"X\((\d+)\)", line)]


def compile_one(args):
    fam, m, extra = args
    cache = os.path.join(ROOT, "build", "isa", "%s_m%d.txt" % (fam, m))
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    if "--reuse" in sys.argv and os.path.exists(cache) and os.path.getmtime(cache) > max(os.path.getmtime(f) for f in srcs):
        return fam, m, open(cache).read()
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-DMCSAS_M=%d" % m, "-Rpass-analysis=kernel-resource-usage", "-c", "-o",
               os.path.join(td, "x.o"), os.path.join(CSRC, "kern_%s.hip" % fam)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        open(cache, "w").write(r.stderr)
        return fam, m, r.stderr


def parse(text):
    rows, cur = [], None
    for l in text.splitlines():
        mm = re.search(r"Function Name: (\S+)", l)
        if mm:
            cur = {"name": mm.group(1)}; rows.append(cur); continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            mm = re.search(pat, l)
            if mm and cur is not None:
                cur[key] = int(mm.group(1))
    return rows


def demangle(names):
    """_ZN5mcsas16pipe_tick_kernelILi0ELi8EEEv... -> pipe_tick_kernel<0, 8> (integer / bool template arguments only)"""
    out = []
    for n in names:
        mm = re.match(r"_ZN5mcsas(\d+)", n)
        ln = int(mm.group(1)); start = mm.end()
        base, rest = n[start:start + ln], n[start + ln:]
        args = []
        if rest.startswith("I"):
            for kind, val in re.findall(r"L([ib])(n?\d+)E", rest.split("EE")[0] + "E"):
                args.append(("true" if val == "1" else "false") if kind == "b" else val.replace("n", "-"))
        out.append("%s<%s>" % (base, ", ".join(args)))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 4
    work = [(fam, m, []) for fam in ("pipe", "wave", "wg", "wide") for m in models()]
    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(compile_one, work))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = ["# ISA resources of the chain kernels (%s, tree at %s%s)" % (tag, commit, "+" if subprocess.run(["git", "-C", ROOT, "diff", "--quiet"]).returncode else ""), "",
           "`hipcc %s -Rpass-analysis=kernel-resource-usage` on `mcsas_amd/csrc/kern_{pipe,wave,wg,wide}.hip` per model (tools/isa_resources.py)." % " ".join(FLAGS),
           "gfx950: 512 registers per lane and SIMD shared by a wave's VGPRs + AGPRs; the pipeline / workgroup kernels run 8 waves per",
           "workgroup (2 waves per SIMD: up to 256 registers each), LDS is dynamic (not in this table: see DESIGN.md). `scratch` > 0 means",
           "spilled vector registers (bytes per lane); SGPR spills go to VGPR lanes (v_writelane / v_readlane), not to memory.", ""]
    for fam, title in (("pipe", "pipe_tick_kernel<M, QPL> — whole-chip pipeline"), ("wave", "chain_wave_kernel<M, QPL, CACHE> — one wavefront per chain"),
                       ("wg", "chain_wg_kernel<M, QPL> — one workgroup per chain"), ("wide", "chain_wide_kernel<M, QPL> — more than 1024 q-points")):
        out += ["## " + title, "", "| model | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | occupancy waves/SIMD |", "|---|---|---|---|---|---|---|---|---|"]
        for f, m, text in res:
            if f != fam:
                continue
            rows = [r for r in parse(text) if "reset" not in r["name"]]
            for r, nm in zip(rows, demangle([r["name"] for r in rows])):
                out.append("| %d %s | `%s` | %d | %d | %d | %d | %d | %d | %d |" % (m, MODEL_NAMES.get(m, "?"), nm, r.get("vgpr", -1), r.get("agpr", -1),
                           r.get("sgpr", -1), r.get("scratch", -1), r.get("vspill", -1), r.get("sspill", -1), r.get("occ", -1)))
        out.append("")
    path = os.path.join(ROOT, "profiles", "%s_isa_resources.md" % tag)
    open(path, "w").write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
