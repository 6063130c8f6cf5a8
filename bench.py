#!/usr/bin/env python3
"""bench.py — MC accept/reject steps per second of the McSAS hot path on MI355X.

Headline workload (BASELINE.json configs[1]): Sphere model, synthetic 512 q-points x 400 contributions,
50 repetitions per GPU, 20000 Monte-Carlo steps per chain per launch with convergenceCriterion = 0 (no chain
leaves early; SURVEY 8d "throughput runs").  One bench "step" = `--launches-per-step` back-to-back launches of
all chains of the rank (re-seeded each time, data and workspaces resident in HBM), so that the K timed steps
cover seconds of steady-state clocks; per-launch device times (HIP events on the launch stream) are reported as
min / median / max.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config 2|3|4|5]

`value` is config 2 AS NAMED: one 50-repetition analysis on the chip at a time (one plan, one stream; the next analysis is queued
while the previous one's results are fetched).  The throughput of a SERIES of data sets — two plans on streams of their own whose
analyses share the chip — is reported beside it as `series_two_streams`.

--gpus N > 1 without a torch.distributed environment: this process starts the N ranks itself
(`python -m torch.distributed.run`, one process per GPU, 127.0.0.1 rendezvous) BEFORE anything touches the GPU,
forwards their output and exits with their code; under an external launcher (WORLD_SIZE set) the world size must
equal --gpus.  Weak scaling: every rank runs the config's per-GPU repetitions; strong scaling: the config's total
repetitions (50 / 200 / 400 / 100) are sharded over the ranks (mcsas_amd.dist.shard_reps); chain id = global
repetition index either way; nothing in an analysis needs another rank (no data-path collective): one all-gather (RCCL)
after the timed region puts the last analysis together.

Prints ONE JSON line (rank 0); besides the contract's keys: `roofline` (the SURVEY 8d byte model — 40 Q bytes per MC step —
against the HBM peak for the tick kernels of ONE analysis by themselves: HIP-event time of analyses run alone on one stream,
`launch_ms_solo`; with the memory-side traffic of the committed FETCH/WRITE passes as `traffic`), `launch_ms` /
`launch_ms_effective` (per-analysis event times in the timed region, where two streams share the chip, and the timed region
divided by its analyses), `roofline_valu` (the
resource the counters name for these kernels: fp64 vector issue), `launch_ms` (min / median / max), `configs`
(configs 3-5 at their per-GPU repetition counts, each sustained over >= 1 s of back-to-back launches),
`convergence_run` (criterion 1 as BASELINE names it), `quickstart` (the reference's published workload end to end) and
`cpu_baseline`.

Round 5: BASELINE names configs 3-5 as TOTALS over 8 GPUs (200 / 400 / 100 repetitions).  With the default --config 2 every run
— one rank or N — also runs those totals strong-sharded over the ranks (`shard_reps`, chain id = global repetition index, no
collective on the data path, all-reduce of the timings only): `configs[k].strong_total` in the N = 1 line (all repetitions on
one GPU), `configs[k]` in an N > 1 line, so that a 1 -> N ratio exists for each.  Per config 2-4 the line also carries
`cpu_baseline` (the C oracle timed on the same workload shape) and `chisq_rel_diff_vs_cpu`: the final chi² of chains run on the GPU
and by the C oracle on identical Philox streams (max over chains of the relative difference).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q, NCONTRIB, REPS_PER_GPU = 512, 400, 50
HBM_PEAK = 8.0e12                     # B/s, MI355X_MICROARCH.md
FP64_VECTOR_PEAK_INSTR = 256 * 4 * 2.4e9 / 4.0   # wave-instructions/s: 1024 SIMDs, one v_fma_f64 per 4 cycles (78.6 TFLOP/s)


def synthetic_data(nq=Q, seed=20250101):
    """q = logspace(1e7, 3e9) 1/m, tri-modal sphere population (8/40/100 nm),
    sigma = 1 % of I, multiplicative Gaussian noise, flat background 0."""
    q = np.logspace(7, np.log10(3e9), nq)
    rs = np.random.RandomState(seed)
    radii = np.abs(np.concatenate([rs.normal(8, 3, 300), rs.normal(40, 10, 150), rs.normal(100, 10, 50)])) + 0.5
    radii *= 1e-9
    I = np.zeros(nq)
    for R in radii:
        x = q * R
        F = 3 * (np.sin(x) - x * np.cos(x)) / x**3
        I += (4 * np.pi / 3 * R**3)**2 * F**2
    I *= 1e3 / I.max()
    sigma = 0.01 * I
    I = I + sigma * rs.normal(size=nq)
    return q, I, sigma


def truth_population(config, rs):
    """"a known tri-modal population mirroring quickstart.rst:195-199 ... for the model under test".  The three
    Gaussian modes (8 / 40 / 100 nm, widths 3 / 10 / 10 nm, 300 / 150 / 50 members) at HALF size, so that they sit inside the
    radius ranges of configs 3 and 4 (1-100 nm); the remaining shape parameters uniform inside their ranges."""
    r = np.abs(np.concatenate([rs.normal(4, 1.5, 300), rs.normal(20, 5, 150), rs.normal(50, 5, 50)])) + 0.5
    r *= 1e-9
    if config == 3:                                           # (radius, aspect)
        return np.stack([r, rs.uniform(1.0, 8.0, len(r))], axis=1)
    if config == 4:                                           # (a, b, t): b = a x (1..3), shell 0.5-5 nm
        return np.stack([r, r * rs.uniform(1.0, 3.0, len(r)), rs.uniform(5e-10, 5e-9, len(r))], axis=1)
    raise ValueError(config)


def model_truth_data(config, model, nq, device, seed=20250101):
    """Ground-truth I(q) of `model` for the population above, by the library's own ScatteringModel.calc (mcsas_hip_model_calc, outside
    any timed region), normalised to a maximum of 1e3, sigma = 1 % of I, multiplicative Gaussian noise, flat background 0 — the
    recipe of synthetic_data() with the model under test in place of the sphere."""
    from mcsas_amd import engine
    q = np.logspace(7, np.log10(3e9), nq)
    rs = np.random.RandomState(seed)
    pset = truth_population(config, rs)
    I = engine.model_calc(model.setup(), q, pset, 0.6666666, device=device)[0]
    I = I * (1e3 / I.max())
    sigma = 0.01 * I
    z = rs.normal(size=nq)
    return q, I + sigma * z, sigma, float(np.mean(z * z))


def kholodenko_file_data():
    """BASELINE config 5's data: testdata/sasfit_kho-1-10-1000.dat as the reference's loader prepares it (1 %
    uncertainty floor), brought to 512 q-points; the vectors are the ones the reference itself was run on for the
    fixture tests/golden/g9_kho_q512.npz (oracle/make_golden.py: gen_kholodenko_config5)."""
    path = os.path.join(ROOT, "tests", "golden", "g9_kho_q512.npz")
    g = np.load(path)
    return np.array(g["data_q"]), np.array(g["data_I"]), np.array(g["data_sigma"])


# BASELINE.json configs 2-5: model, data, contributions, repetitions (total over 8 GPUs / per GPU), instructions
# per form-factor point (fp64 wave-instructions per 64 points, from the SQ_INSTS_VALU passes in profiles/)
def workload(config, device=-1, dry=False):
    """-> dict(name, model, q, I, sigma, n, reps_total, reps_gpu, K, chisq_of_truth).  chisq_of_truth: reduced chi² of the
    noise-free ground truth against the noisy curve (mean of the squared noise draws) — what "final chi²" is to be read against:
    a criterion below it asks the chains to fit the noise."""
    import mcsas_amd
    if config == 2:
        q, I, s = synthetic_data(512)
        rs = np.random.RandomState(20250101); rs.normal(size=500); z = rs.normal(size=512)     # (the draws of synthetic_data)
        m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
        return dict(name="Sphere, synthetic 512 q-points x 400 contribs", model=m, q=q, I=I, sigma=s, n=400,
                    reps_total=50, reps_gpu=50, K=1, chisq_of_truth=float(np.mean(z * z)))
    if config == 3:
        m = mcsas_amd.CylindersIsotropic()
        m.radius.setActive(True); m.aspect.setActive(True)
        m.radius.setActiveRange((1e-9, 1e-7)); m.aspect.setActiveRange((0.5, 20.0))
        q, I, s, c2 = (synthetic_data(512) + (None,)) if dry else model_truth_data(3, m, 512, device)
        return dict(name="Isotropic cylinders (radius + aspect, intDiv 100), synthetic 512 q x 400 contribs, ground truth: tri-modal "
                         "cylinder population", model=m, q=q, I=I, sigma=s, n=400, reps_total=200, reps_gpu=25, K=100, chisq_of_truth=c2)
    if config == 4:
        m = mcsas_amd.EllipsoidalCoreShell()
        for name, rng in (("a", (1e-9, 1e-7)), ("b", (2e-9, 2e-7)), ("t", (2e-10, 1e-8))):
            getattr(m, name).setActive(True); getattr(m, name).setActiveRange(rng)
        q, I, s, c2 = (synthetic_data(1024) + (None,)) if dry else model_truth_data(4, m, 1024, device)
        return dict(name="Core-shell ellipsoid (a, b, t, intDiv 100), synthetic 1024 q x 1000 contribs, ground truth: tri-modal "
                         "core-shell ellipsoid population", model=m, q=q, I=I, sigma=s, n=1000, reps_total=400, reps_gpu=50, K=200,
                    chisq_of_truth=c2)
    if config == 5:
        q, I, s = kholodenko_file_data()
        m = mcsas_amd.Kholodenko()
        return dict(name="Kholodenko worm, testdata/sasfit_kho-1-10-1000.dat at 512 q x 600 contribs", model=m,
                    q=q, I=I, sigma=s, n=600, reps_total=100, reps_gpu=13, K=1, chisq_of_truth=None)
    raise SystemExit("unknown --config %r" % config)


def cpu_baseline(q, I, sigma, lo, hi, seconds_target=12.0):
    """The compiled CPU oracle (oracle/c/mcsas_oracle.c: plain-C restatement of mcFit with the closed-form
    fit and cached rows, libm sin/cos) on the GPU's share of the host cores, one chain per thread at a time, same
    workload shape, bounded sample; beside it the numpy restatement on the same number of cores, one chain per
    process."""
    from oracle import mcsas_oracle as O
    from oracle import c_oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, int(os.environ.get("MCSAS_BENCH_CPU_THREADS", "16"))))   # one GPU's CPU share of the host
    c_oracle.load()
    probe = 20000
    t0 = time.time()
    c_oracle.analyse_sphere(q, I, sigma, lo, hi, NCONTRIB, threads, probe, 0.0, seed=1, threads=threads)
    t_probe = time.time() - t0
    steps = int(max(probe, min(2000000, probe * seconds_target / max(t_probe, 1e-3))))
    t0 = time.time()
    r = c_oracle.analyse_sphere(q, I, sigma, lo, hi, NCONTRIB, threads, steps, 0.0, seed=1, threads=threads)
    dt = time.time() - t0
    out = {"value": float(r.num_iter.sum()) / dt, "unit": "MC steps/s", "cores": threads, "kind": "port",
           "host_cores_total": os.cpu_count(), "host_cores_available": avail,
           "sample": "%d chains x %d MC steps (incl. %d-contribution init each), Sphere %dq x %d contribs, C oracle "
                     "(oracle/c, gcc -O2, libm), %d threads" % (threads, steps, NCONTRIB, Q, NCONTRIB, threads)}
    # the numpy restatement, one chain per process on the same number of cores
    import multiprocessing as mp
    nsteps = 2500
    ctx = mp.get_context("spawn")          # this process holds a GPU context: no fork
    t0 = time.time()
    with ctx.Pool(threads) as pool:
        done = pool.map(_numpy_chain, [(q, I, sigma, lo, hi, nsteps, c) for c in range(threads)])
    dtn = time.time() - t0
    out["numpy_port"] = {"value": float(sum(done)) / dtn, "cores": threads,
                         "sample": "%d processes x %d MC steps (incl. init), oracle/mcsas_oracle.py" % (threads, nsteps)}
    return out


def oracle_spec(setup):
    """The checker's description (oracle.mcsas_oracle.ModelSpec) of a flattened product model (engine.ModelSetup)."""
    from oracle import mcsas_oracle as O
    return O.ModelSpec(int(setup.model_id), tuple(int(i) for i in setup.active_index), np.array(setup.gen_lo, float),
                       np.array(setup.gen_hi, float), tuple(int(k) for k in setup.gen_kind), np.array(setup.params, float))


def cpu_vs_gpu_chains(wl, steps, dev_index, seed=1):
    """BASELINE.md §3, per config: `threads` chains of the workload run (a) on the GPU through the C ABI and (b) by the plain-C
    oracle (oracle/c) on the SAME counter-based random streams for `steps` steps each — the oracle's wall time is the CPU baseline
    of this config (one chain per thread), the chains' final chi² are compared (max over chains of |gpu - cpu| / cpu), and so are
    their move counts.  Only the checker is timed here; the GPU side is untimed."""
    from oracle import c_oracle
    from mcsas_amd import engine
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, int(os.environ.get("MCSAS_BENCH_CPU_THREADS", "16"))))
    setup = wl["model"].setup()
    st = engine.Settings(n_contrib=wl["n"], n_reps=threads, max_iter=steps, conv_crit=0.0, max_retries=0, seed=seed, device=dev_index)
    res = engine.analyse(setup, wl["q"], wl["I"], wl["sigma"], st)
    c_oracle.load()
    t0 = time.time()
    ref = c_oracle.analyse(oracle_spec(setup), wl["q"], wl["I"], wl["sigma"], wl["n"], threads, steps, 0.0, seed=seed, threads=threads)
    dt = time.time() - t0
    rel = np.abs(res.chisq - ref.chisq) / np.abs(ref.chisq)
    return {"value": float(ref.num_iter.sum()) / dt, "unit": "MC steps/s", "cores": threads, "kind": "port",
            "host_cores_total": os.cpu_count(), "host_cores_available": avail,
            "sample": "%d chains x %d MC steps (incl. %d-contribution init each), %s, C oracle (oracle/c, gcc -O2, libm, Cephes J1), "
                      "%d threads, %.1f s" % (threads, steps, wl["n"], wl["name"], threads, dt),
            "chisq_rel_diff_vs_cpu": float(rel.max()), "chisq_rel_diff_target": 1e-5,
            "moves_equal": bool(np.array_equal(res.num_moves, ref.num_moves)),
            "chisq_gpu_median": float(np.median(res.chisq)), "chisq_cpu_median": float(np.median(ref.chisq)),
            "compared": "%d chains, %d steps each, identical Philox streams (seed %d), GPU exec mode auto" % (threads, steps, seed)}


def _numpy_rows(args):
    """`n` Kholodenko rows at the workload's q grid through the numpy / QUADPACK restatement (one MC step of the reference = one such
    row for the proposal, the `old` row being cached): seconds taken."""
    q, lo, hi, n, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import mcsas_oracle as O
    spec = O.ModelSpec.make("kholodenko", ["radius", "lenKuhn", "lenContour"], lo, hi)
    rs = np.random.RandomState(seed)
    t0 = time.time()
    for _ in range(n):
        row = [O.transform(g, rs.rand()) * (h - l) + l for g, l, h in zip(spec.gen, spec.lo, spec.hi)]
        O.calc_intensity(spec, q, row, 0.6666666)
    return time.time() - t0


def config5_cpu_and_reference(wl, dev_index, rows_per_proc=6):
    """Config 5 has no C port (the reference integrates the worm's form factor with QUADPACK): (a) CPU baseline = the numpy / scipy
    restatement's row rate, one process per core, a bounded sample (a step of the cached-row algorithm costs one row); (b) final
    chi² against the REFERENCE ITSELF: tests/golden/g9_kho_q512_long.npz — the reference's own chain at this config's shape, 1300
    steps — replayed here through the C ABI on the uniform stream the reference consumed."""
    import multiprocessing as mp
    from mcsas_amd import engine
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, int(os.environ.get("MCSAS_BENCH_CPU_THREADS", "16"))))
    setup = wl["model"].setup()
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(threads) as pool:
        pool.map(_numpy_rows, [(wl["q"], list(setup.gen_lo), list(setup.gen_hi), rows_per_proc, 100 + c) for c in range(threads)])
    dt = time.time() - t0
    out = {"value": threads * rows_per_proc / dt, "unit": "MC steps/s", "cores": threads, "kind": "port",
           "sample": "%d processes x %d form-factor rows of %d q (numpy / scipy QUADPACK restatement, oracle/mcsas_oracle.py; one row per "
                     "MC step with cached `old` rows), %.1f s" % (threads, rows_per_proc, len(wl["q"]), dt)}
    path = os.path.join(ROOT, "tests", "golden", "g9_kho_q512_long.npz")
    if os.path.exists(path):
        g = np.load(path)
        st = engine.Settings(n_contrib=int(g["spec_n_contrib"]), n_reps=1, max_iter=int(g["spec_max_iter"]), conv_crit=float(g["spec_conv_crit"]),
                             comp_exp=float(g["spec_comp_exp"]), max_retries=0, device=dev_index)
        res = engine.analyse(setup, g["data_q"], g["data_I"], g["data_sigma"], st, replay=g["stream"][None, :])
        out["chisq_rel_diff_vs_reference"] = abs(float(res.chisq[0]) - float(g["res_conval"])) / float(g["res_conval"])
        out["moves_equal_reference"] = bool(int(res.num_moves[0]) == int(g["res_num_moves"]))
        out["compared"] = "the reference's own chain (g9_kho_q512_long: 512 q x 600 contributions x %d steps, %d accepted) replayed on its uniform stream" % (
            int(g["res_num_iter"]), int(g["res_num_moves"]))
    return out


def _numpy_chain(args):
    q, I, sigma, lo, hi, nsteps, chain = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import mcsas_oracle as O
    spec = O.ModelSpec.make("sphere", ["radius"], [lo], [hi])
    st = O.Settings(n_contrib=NCONTRIB, n_reps=1, max_iter=nsteps, conv_crit=0.0)
    r = O.mc_fit(spec, q, I, sigma, [I.min(), I.max()], [q.min(), q.max()], st, O.PhiloxStream(1, chain), method="closed")
    return r.num_iter


# --------------------------------------------------------------------------------------------- multi-process
def self_launch(args, argv):
    """--gpus N > 1 with no launcher around us: become the launcher (nothing here touches the GPU)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


class DryPlan(object):
    """Stand-in for engine.Plan in the CPU tests of THIS FILE's multi-rank code path (tests/test_dist_gloo.py,
    MCSAS_BENCH_DRY=1): no Monte Carlo, the per-repetition payload is a hash of (seed, global repetition index),
    which is what makes a sharded run comparable with a 1-rank run repetition for repetition.  Never timed,
    never reported as a measurement (`data` says "dry-run", `value` is null)."""

    def __init__(self, n_contrib, n_active, nq, reps):
        self.shape = (n_contrib, n_active, nq, reps)
        self.seed, self.first = 0, 0
        self.last_ms, self.total_steps = 1.0, 0
        self.info = dict(exec_mode="dry", waves_per_chain=0, q_per_lane=0, window=0, launches=0, cached_rows=False)

    def reseed(self, seed, first):
        self.seed, self.first = seed, first

    def launch(self):
        pass

    def fetch(self):
        N, P, nq, R = self.shape

        class Res(object):
            pass
        res = Res()
        res.contribs = np.zeros((N, P, R)); res.fit = np.zeros((nq, R))
        res.chisq = np.zeros(R); res.scaling = np.zeros(R); res.background = np.zeros(R)
        for r in range(R):
            rs = np.random.RandomState((self.seed * 1000003 + self.first + r) % (2**31 - 1))
            res.contribs[:, :, r] = rs.rand(N, P); res.fit[:, r] = rs.rand(nq)
            res.chisq[r], res.scaling[r], res.background[r] = rs.rand(3)
        return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config 2..5 (headline: 2)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: weak for config 2 (its 50 repetitions on EVERY GPU; the strong figure — 50 in all — is reported beside "
                         "it as `strong_scaling`), strong for configs 3-5, whose repetition counts are named as totals over 8 GPUs")
    ap.add_argument("--launches-per-step", type=int, default=0,
                    help="launches per bench step (0: enough for ~0.15 s per step on the headline config, 1 otherwise)")
    ap.add_argument("--mc-steps", type=int, default=0, help="MC iterations per chain per launch (0: 20000 for config 2)")
    ap.add_argument("--reps", type=int, default=0, help="repetitions (chains) per GPU, weak scaling (0: the config's)")
    ap.add_argument("--waves", type=int, default=0, help="waves per chain (0 = library default)")
    ap.add_argument("--mode", type=int, default=0, help="exec_mode: 0 auto, 1 wave, 2 workgroup, 3 pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence-run", action="store_true",
                    help="skip the untimed run-to-convergence (profiling: only the timed launches reach the profiler)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of configs 3-5")
    ap.add_argument("--totals-only", action="store_true", help="of configs 3-5 run only the named totals (200 / 400 / 100 repetitions sharded over the "
                    "ranks), not the per-GPU-share runs a one-rank line carries beside them")
    ap.add_argument("--debug-flags", type=int, default=0, help="diagnostic role ablation (invalid results)")
    ap.add_argument("--streams", type=int, default=1, help="plans on streams of their own whose analyses overlap on the chip (default 1: "
                    "config 2 as named, one analysis on the chip at a time; the two-stream series figure is reported beside it)")
    ap.add_argument("--no-series", action="store_true", help="skip the two-stream series measurement behind the timed region")
    ap.add_argument("--no-many-chains", action="store_true", help="skip the wave-per-chain run (8192 chains)")
    ap.add_argument("--inflight", type=int, default=2, help="result slots used per plan (analyses of one plan in flight); 1 = fetch before the next launch")
    ap.add_argument("--dump", default="", help="rank 0: write the gathered arrays of the last launch to this .npz (tests)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # MCSAS_BENCH_FORCE_DIST=1: the multi-rank code path — process group, RCCL all-gather of the results, all-reduce of the timings —
    # also at world size 1 (tests/test_parity_gpu.py runs it once on the one-GPU box, so that path has executed on hardware)
    force_dist = os.environ.get("MCSAS_BENCH_FORCE_DIST") == "1"
    if world_env is None and (args.gpus > 1 or force_dist):
        self_launch(args, sys.argv[1:])                      # does not return
    rank = int(os.environ.get("RANK", "0")); world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: refusing to report a different number of GPUs than asked for"
                         % (args.gpus, world))

    dry = os.environ.get("MCSAS_BENCH_DRY") == "1"
    import torch
    import mcsas_amd
    from mcsas_amd import engine, dist as mdist

    # one process per GPU.  MCSAS_BENCH_BACKEND=gloo lets several ranks share one card for a dry run of
    # the multi-process path on a single-GPU box (results gathered through host memory)
    backend = os.environ.get("MCSAS_BENCH_BACKEND", "nccl")
    dev_index = 0
    if not dry:
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < world:
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, ndev))
        dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as tdist
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            tdist.init_process_group(backend)

    wl = workload(args.config, dev_index, dry)
    q, I, sigma, model, ncontrib = wl["q"], wl["I"], wl["sigma"], wl["model"], wl["n"]
    if args.scaling is None:
        args.scaling = "weak" if args.config == 2 else "strong"
    if args.scaling == "weak":
        reps = args.reps or wl["reps_gpu"]
        n_total, first = reps * world, rank * reps
    else:
        n_total = args.reps * world if args.reps else wl["reps_total"]
        if n_total < world:
            raise SystemExit("strong scaling: %d repetitions cannot be sharded over %d ranks" % (n_total, world))
        first, reps = mdist.shard_reps(n_total, world, rank)
    mc_steps = args.mc_steps or {2: 20000, 3: 2000, 4: 1000, 5: 1000}[args.config]
    lps = args.launches_per_step or (40 if args.config == 2 and not dry else 1)
    setup = model.setup()
    if dry:
        plan = DryPlan(ncontrib, setup.n_active, len(q), reps)
    else:
        st = engine.Settings(n_contrib=ncontrib, n_reps=reps, max_iter=mc_steps, conv_crit=0.0,
                             max_retries=0, seed=20250101, rep_offset=first, device=dev_index,
                             waves_per_chain=args.waves, exec_mode=args.mode, debug_flags=args.debug_flags)
        plan = engine.Plan(setup, q, I, sigma, st)
    # Analyses in flight: every launch is a complete analyse() of this rank's repetitions whose results come back to the host.
    #  * --streams S (default 2): S plans, each on a stream of its own.  Their tick kernels share the chip: an analysis' first
    #    ~45 ticks are bound by its scan blocks (50 of 256 CUs busy for twice as long as the producers), its last ~60 by the
    #    producers — two analyses at different stages fill each other's idle CUs (3.2e8 against 2.9e8 steps/s on one stream).
    #  * --inflight (default 2): result slots per plan (mcsas_hip_plan_launch_slot) — the plan's next analysis is already queued
    #    while the previous one's results are fetched and unpacked.
    # A series of data sets (mcsas_amd.run_series; the reference's gui/calc.py:271-330 runs them one after the other) is the use
    # case.  --streams 1 --inflight 1: strictly one analysis after the other.
    nslots = 1 if dry else max(1, min(args.inflight, 2))
    plans = [plan]
    streams = [None]
    if not dry and args.streams > 1:
        plans += [engine.Plan(setup, q, I, sigma, st) for _ in range(args.streams - 1)]
        streams = [torch.cuda.Stream() for _ in plans]
    lanes = [(k, sl) for sl in range(nslots) for k in range(len(plans))]      # launch order: A0 B0 A1 B1 ...

    launch_ms = []
    gathered = {}
    pending = []
    state = {"mc": 0, "res": None, "n": 0, "gather_ms": None}

    def retire():
        k, slot = pending.pop(0)
        pl = plans[k]
        res = pl.fetch() if dry else pl.fetch(slot=slot)
        launch_ms.append(pl.last_ms)
        state["mc"] += pl.total_steps
        state["res"] = res
        return res

    def one_launch(seed):
        k, slot = lanes[state["n"] % len(lanes)]
        state["n"] += 1
        while (k, slot) in pending or len(pending) >= len(lanes):   # (a slot's previous analysis must be home before it is written again)
            retire()
        pl = plans[k]
        pl.reseed(seed, first)
        if dry:
            pl.launch()
        else:
            pl.launch(stream=streams[k].cuda_stream if streams[k] is not None else None, slot=slot)
        pending.append((k, slot))

    def drain():
        while pending:
            retire()

    def gather_last():
        # the ranks' blocks of the LAST analysis put together (one RCCL all-gather, outside the timed region: the repetitions
        # are independent and nothing in an analysis needs another rank's results); timed by itself -> `gather_ms`
        res = state["res"]
        if res is not None and (use_dist or args.dump):
            local = dict(contribs=np.moveaxis(res.contribs, 2, 0), chisq=res.chisq[:, None],
                         scaling=res.scaling[:, None], background=res.background[:, None], fit=res.fit.T)
            gathered.clear()
            if use_dist:
                ms = []
                for _ in range(1 if dry else 3):               # (the first call pays RCCL's lazy connection set-up)
                    barrier()
                    t0g = time.perf_counter()
                    got = mdist.gather_results(local, n_total, force=True)
                    barrier()
                    ms.append((time.perf_counter() - t0g) * 1e3)
                gathered.update(got)
                state["gather_ms"] = ms
            else:
                gathered.update({k: np.asarray(v) for k, v in local.items()})

    def barrier():
        if use_dist:
            tdist.barrier()
        if not dry:
            torch.cuda.synchronize()

    seed = 1000
    for w in range(args.warmup):
        for _ in range(lps):
            one_launch(seed); seed += 1
    drain()
    barrier()
    del launch_ms[:]
    state["mc"] = 0
    sampler = ClockSampler(dev_index) if (not dry and rank == 0) else None
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    for k in range(args.steps):
        for _ in range(lps):
            one_launch(seed); seed += 1
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
    dt_local = dt
    mc_total, res = state["mc"], state["res"]
    gather_last()
    # The dominant kernel by itself, for the roofline objects.  One stream (the default): an analysis' HIP events bracket its own
    # tick kernels only, the per-analysis event times ARE the kernel's.  Several streams (--streams 2): the events bracket the other
    # streams' kernels too, so a few analyses are run ALONE behind the timed region (that is also what the committed rocprofv3
    # passes characterise: tools/profile.sh runs --streams 1 --inflight 1).
    solo_ms = []
    if not dry and len(plans) > 1:
        for i in range(8):
            plan.reseed(seed + i, first); plan.launch(); plan.fetch(want_arrays=False)
            if i >= 2:
                solo_ms.append(plan.last_ms)
    # A SERIES of data sets (gui/calc.py:271-330; mcsas_amd.run_series(overlap=True)): two plans on streams of their own, two result
    # slots each, launch order A0 B0 A1 B1 — an analysis' scan-bound first ticks run beside another's producer-bound last ones.
    series = None
    if not dry and len(plans) == 1 and world == 1 and args.config == 2 and not args.no_series:
        series = series_two_streams(engine, torch, setup, q, I, sigma, st, first, seed + 100)
    # Strong scaling of config 2 beside the weak headline: its 50 repetitions IN ALL, sharded over the ranks (6-7 chains per GPU)
    strong = None
    if not dry and world > 1 and args.config == 2 and args.scaling == "weak":
        f2, r2 = mdist.shard_reps(wl["reps_total"], world, rank)
        st2 = engine.Settings(n_contrib=ncontrib, n_reps=r2, max_iter=mc_steps, conv_crit=0.0, max_retries=0, seed=20250101,
                              rep_offset=f2, device=dev_index, waves_per_chain=args.waves, exec_mode=args.mode)
        pl2 = engine.Plan(setup, q, I, sigma, st2)
        nl = max(4, lps)
        for i in range(3):
            pl2.reseed(5000 + i, f2); pl2.launch(); pl2.fetch(want_arrays=False)
        barrier()
        t0s = time.perf_counter()
        steps2 = 0
        for i in range(nl):
            pl2.reseed(5100 + i, f2); pl2.launch(slot=i & 1)
            if i >= 1:
                pl2.fetch(want_arrays=False, slot=(i - 1) & 1); steps2 += pl2.total_steps
        pl2.fetch(want_arrays=False, slot=(nl - 1) & 1); steps2 += pl2.total_steps
        barrier()
        dts = time.perf_counter() - t0s
        tt = torch.tensor([dts, float(steps2)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tmx = tt.clone(); tdist.all_reduce(tmx, op=tdist.ReduceOp.MAX)
        tsm = tt.clone(); tdist.all_reduce(tsm, op=tdist.ReduceOp.SUM)
        strong = {"value": float(tsm[1]) / float(tmx[0]), "unit": "MC steps/s", "scaling": "strong", "reps_total": wl["reps_total"],
                  "reps_this_rank": r2, "launches": nl, "timed_region_s": float(tmx[0]), "window": pl2.info["window"],
                  "note": "config 2's 50 repetitions in all, sharded over the ranks: 6-7 chains per GPU, where an analysis is bound "
                          "by the per-tick latency of the pipeline (3.1 ms for 7 chains against 3.4 for 50 on one GPU), not by throughput"}
        pl2.close()
    totals = None
    if args.config == 2 and not args.no_configs and (not dry or os.environ.get("MCSAS_BENCH_DRY_CONFIGS") == "1"):
        totals = named_totals(dev_index, world, rank, use_dist, backend, dry, barrier)
    ranks_seen = 1
    if use_dist:
        dev = "cuda" if backend == "nccl" and not dry else "cpu"
        t = torch.tensor([dt, float(mc_total), 1.0], dtype=torch.float64, device=dev)
        tmax = t.clone(); tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        tsum = t.clone(); tdist.all_reduce(tsum, op=tdist.ReduceOp.SUM)
        dt, total_steps, ranks_seen = float(tmax[0]), float(tsum[1]), int(round(float(tsum[2])))
    else:
        total_steps = float(mc_total)

    if rank == 0:
        if args.dump:
            np.savez(args.dump, **gathered)
        lm = np.array(launch_ms) if launch_ms else np.zeros(1)
        overlapped = len(plans) > 1
        launch_eff_s = dt_local / max(len(launch_ms), 1)                               # this rank's timed region / its analyses
        launch_s = float(np.mean(solo_ms)) * 1e-3 if solo_ms else float(lm.mean()) * 1e-3     # one analysis alone on the chip
        steps_per_launch = mc_total / max(len(launch_ms), 1)
        nq = len(q)
        achieved = 40 * nq * steps_per_launch / max(launch_s, 1e-12)
        info = plan.info
        out = {
            "metric": "MC accept/reject steps/sec (whole node)",
            "value": None if dry else total_steps / dt, "unit": "MC steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "dry-run" if dry else ("synthetic" if args.config != 5 else "testdata/sasfit_kho-1-10-1000.dat (reference data file)"),
            "config": {"workload": "%s, %d reps on this rank (%d in all), %d MC steps per chain per launch, convergenceCriterion=0, "
                                   "%s" % (wl["name"], reps, n_total, mc_steps,
                                           "one analysis on the chip at a time" if not overlapped else "%d analyses side by side" % len(plans)),
                       "baseline_config": args.config, "reps_total": n_total, "reps_per_gpu": reps, "launches_per_step": lps,
                       "ranks_seen": ranks_seen, "streams": len(plans), "result_slots": nslots, **info},
            "timed_region_s": dt,
            "launch_ms": {"n": int(len(lm)), "min": float(lm.min()), "median": float(np.median(lm)), "max": float(lm.max()),
                          "mean": float(lm.mean()),
                          "note": "HIP events around each analysis on its stream" + ("; %d streams share the chip, so an analysis' events "
                                  "span the other streams' kernels as well: effective_ms is the timed region / analyses" % len(plans) if overlapped else
                                  " (one stream: the analysis' own tick kernels)")},
            "launch_ms_effective": launch_eff_s * 1e3,
            "launch_ms_solo": {"n": len(solo_ms), "mean": float(np.mean(solo_ms)), "min": float(np.min(solo_ms)), "max": float(np.max(solo_ms)),
                               "note": "analyses run alone on one stream behind the timed region: the kernel time the roofline objects use"} if solo_ms else
                              {"n": int(len(lm)), "mean": float(lm.mean()), "min": float(lm.min()), "max": float(lm.max()),
                               "note": "= launch_ms: one stream, an analysis' events bracket its own kernels only"},
        }
        if sampler:
            out["clocks"] = sampler.summary()
        if state["gather_ms"]:
            gm = state["gather_ms"]
            out["gather_ms"] = {"first": gm[0], "steady": float(np.min(gm[1:])) if len(gm) > 1 else None, "backend": backend, "ranks": world,
                                "bytes_per_rank": int(8 * reps * (ncontrib * setup.n_active + len(q) + 3)),
                                "note": "one packed all-gather of the last analysis' per-repetition results (mcsas_amd/dist.py), outside the timed "
                                        "region: nothing in an analysis needs another rank"}
        if series:
            out["series_two_streams"] = series
        if strong:
            out["strong_scaling"] = strong
        if not dry:
            out["final_chisq_median"] = float(np.median(res.chisq))
            out["chisq_of_truth"] = wl["chisq_of_truth"]
            # SURVEY 8d streaming model: 40*Q bytes per MC step against the HBM peak.  Chain state (q, I, sigma, ft)
            # actually stays on chip, so this is an algorithmic figure; `traffic` is what the memory-side
            # counters saw for the same command in the committed profile named in `traffic_source`.
            traffic, source = None, None
            tj, pmt = latest_profile("pmc_traffic.json")
            pm = pmt.get(str(args.config))
            if pm and info["exec_mode"] == "pipeline" and reps == wl["reps_gpu"]:
                traffic = (pm["fetch_bytes_per_mc_step"] + pm["write_bytes_per_mc_step"]) * steps_per_launch / launch_s / 1e9
                source = "from_profile: %s (commit %s)" % (tj, pm.get("commit", "?"))
            algorithmic = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                           "measured_in_this_run": True,
                           "note": "40*Q B per MC step (SURVEY 8d streaming model, no on-chip reuse credit) x MC steps per launch / mean "
                                   "HIP-event time of a launch; q, I, sigma and ft stay on chip, so this is not memory traffic"}
            ij, pvt = latest_profile("valu_per_step.json")
            pv = pvt.get(str(args.config))
            if pv and info["exec_mode"] == "pipeline":
                rate = pv["valu_wave_instr_per_mc_step"] * steps_per_launch / launch_s
                out["roofline_valu"] = {"bound": "valu", "achieved": rate / 1e9, "peak": FP64_VECTOR_PEAK_INSTR / 1e9,
                                        "unit": "G wave-instr/s", "frac": rate / FP64_VECTOR_PEAK_INSTR,
                                        "instr_per_mc_step": pv["valu_wave_instr_per_mc_step"], "measured_in_this_run": False,
                                        "profile_taken_on_these_kernel_sources": pv.get("csrc_sha16") == kernel_sources_sha16(),
                                        "source": "from_profile: %s (SQ_INSTS_VALU pass, commit %s)" % (ij, pv.get("commit", "?")),
                                        "note": "fp64 vector issue: 1024 SIMDs x one wave-instruction per 4 cycles at 2.4 GHz; achieved = "
                                                "SQ_INSTS_VALU per MC step (committed counter pass) x MC steps per launch / mean HIP-event time of a launch"}
            out["roofline"] = dict(algorithmic, traffic=traffic, traffic_unit="GB/s", traffic_source=source, traffic_measured_in_this_run=False,
                                   profile_taken_on_these_kernel_sources=bool(pm) and pm.get("csrc_sha16") == kernel_sources_sha16())
        # outside the timed region: the same repetitions run the way McSAS.analyse runs them — convergenceCriterion 1 (BASELINE),
        # maxIterations 1e5, one attempt — -> final chi² and how many got there, to be read against chisq_of_truth.
        if not dry and not args.no_convergence_run:
            out["convergence_run"] = convergence_run(engine, setup, q, I, sigma, ncontrib, reps, first, dev_index, wl["chisq_of_truth"], args.mode)
        if not dry and not args.no_convergence_run and world == 1 and args.config == 2:
            out["calc_breakdown"] = calc_breakdown(wl, dev_index)
            out["quickstart"] = quickstart(dev_index)
            out["python_only_model"] = python_only_model(wl, dev_index)
        if not dry and not args.no_configs and not args.totals_only and world == 1 and args.config == 2:
            out["configs"] = other_configs(dev_index)
            for k, e in (totals or {}).items():                  # the named totals on ONE GPU: the denominator of a 1 -> N ratio
                out["configs"][k]["strong_total"] = e
        elif totals:                                             # N > 1 (or the CPU dry run): configs 3-5 as named, whole node
            out["configs"] = totals
        if not dry and not args.no_many_chains and not args.no_configs and world == 1 and args.config == 2:
            out["many_chains"] = many_chains(wl, dev_index)
        if not dry and not args.no_cpu_baseline and world == 1 and args.config == 2:   # the CPU baseline is timed at N = 1 only
            lo, hi = np.pi / q.max(), np.pi / q.min()
            out["cpu_baseline"] = cpu_baseline(q, I, sigma, lo, hi)
            # BASELINE.md §3 per config: the C oracle timed on each workload's shape and the chains' final chi² GPU vs CPU on identical
            # streams; the budgets are the ones tests/test_parity_gpu.py checks chain by chain (20 000 / 4000 / 3000 steps)
            cmp2 = cpu_vs_gpu_chains(wl, 20000, dev_index)
            out["cpu_baseline"].update({k: cmp2[k] for k in ("chisq_rel_diff_vs_cpu", "chisq_rel_diff_target", "moves_equal", "compared")})
            out["chisq_rel_diff_vs_cpu"] = cmp2["chisq_rel_diff_vs_cpu"]
            if "configs" in out and not args.no_configs and not args.totals_only:
                for cfg, steps_c in ((3, 4000), (4, 3000)):
                    e = cpu_vs_gpu_chains(workload(cfg, dev_index), steps_c, dev_index)
                    out["configs"][str(cfg)]["cpu_baseline"] = e
                    out["configs"][str(cfg)]["chisq_rel_diff_vs_cpu"] = e["chisq_rel_diff_vs_cpu"]
                e5 = config5_cpu_and_reference(workload(5, dev_index), dev_index)
                out["configs"]["5"]["cpu_baseline"] = e5
                out["configs"]["5"]["chisq_rel_diff_vs_reference"] = e5.get("chisq_rel_diff_vs_reference")
        print(json.dumps(out))
    if use_dist:
        tdist.destroy_process_group()


def named_totals(dev_index, world, rank, use_dist, backend, dry, barrier):
    """BASELINE configs 3-5 AS NAMED: 200 / 400 / 100 repetitions in all, sharded over the ranks (mcsas.py:214 is the loop being
    sharded; mcsas_amd.dist.shard_reps; chain id = global repetition index).  Every rank runs its block — one warm-up launch, then
    `launches` timed ones between barriers — and the timings meet in two all-reduces (MAX of the time, SUM of the steps): whole-job
    MC steps/s per config.  At world size 1 the same code runs all repetitions on the one GPU, which is the denominator of a 1 -> N
    ratio.  Called on EVERY rank (collectives inside)."""
    import torch
    from mcsas_amd import engine, dist as mdist
    out = {}
    for cfg, budget, launches in ((3, 10000, 2), (4, 15000, 2), (5, 10000, 2)):
        wl = workload(cfg, dev_index, dry)
        total = wl["reps_total"]
        first, reps = mdist.shard_reps(total, world, rank)
        setup = wl["model"].setup()
        steps, dt, info = 0, 0.0, {}
        if dry:
            plan = DryPlan(wl["n"], setup.n_active, len(wl["q"]), reps)
            plan.reseed(1, first); plan.launch(); plan.fetch()
            info = plan.info
        elif reps > 0:
            st = engine.Settings(n_contrib=wl["n"], n_reps=reps, max_iter=budget, conv_crit=0.0, max_retries=0, seed=20250101,
                                 rep_offset=first, device=dev_index)
            plan = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st)
            plan.reseed(600, first); plan.launch(); plan.fetch(want_arrays=False)
        barrier()
        t0 = time.perf_counter()
        if not dry and reps > 0:
            for i in range(launches):
                plan.reseed(601 + i, first); plan.launch(slot=i & 1)
                if i >= 1:
                    plan.fetch(want_arrays=False, slot=(i - 1) & 1); steps += plan.total_steps
            plan.fetch(want_arrays=False, slot=(launches - 1) & 1); steps += plan.total_steps
            info = plan.info
        barrier()
        dt = time.perf_counter() - t0
        if not dry and reps > 0:
            plan.close()
            engine.release_cached_memory()
        ranks_seen, tot_steps, tmax = 1, float(steps), dt
        if use_dist:
            import torch.distributed as tdist
            dev = "cuda" if backend == "nccl" and not dry else "cpu"
            t = torch.tensor([dt, float(steps), 1.0], dtype=torch.float64, device=dev)
            tm = t.clone(); tdist.all_reduce(tm, op=tdist.ReduceOp.MAX)
            ts = t.clone(); tdist.all_reduce(ts, op=tdist.ReduceOp.SUM)
            tmax, tot_steps, ranks_seen = float(tm[0]), float(ts[1]), int(round(float(ts[2])))
        out[str(cfg)] = {"workload": "%s, %d reps IN ALL sharded over %d rank(s) (%d on rank 0), %d MC steps per chain per launch"
                                     % (wl["name"], total, world, reps, budget),
                         "value": None if dry else tot_steps / max(tmax, 1e-12), "unit": "MC steps/s", "scaling": "strong",
                         "reps_total": total, "reps_rank0": reps, "ranks_seen": ranks_seen, "n_gpus": world, "launches": launches,
                         "mc_steps": tot_steps, "timed_region_s": tmax, "exec_mode": info.get("exec_mode"), "window": info.get("window")}
    return out


def kernel_sources_sha16():
    """sha256 (16 hex digits) over the kernel sources (mcsas_amd/csrc/*.h, *.hip, in name order): tools/pmc_summary.py stores it beside
    the per-step counters it derives, and the bench line says whether the counters it quotes were taken on THESE sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mcsas_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def latest_profile(suffix):
    """(relative path, parsed JSON) of the newest committed profiles/rNN_<suffix>, ({} when there is none)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not c:
        return None, {}
    return os.path.relpath(c[-1], ROOT), json.load(open(c[-1]))


def convergence_run(engine, setup, q, I, sigma, ncontrib, reps, first, dev_index, chisq_of_truth, mode=0, max_iter=100000):
    """The workload's repetitions the way McSAS.analyse runs them (mcsas.py:191-285): criterion 1 (BASELINE / SURVEY 8d), maxIterations
    1e5, one attempt per repetition.  Second leg when criterion 1 is below what the data allow (it asks for a chi² under the ground
    truth's own): the same at 1.15 x chisq_of_truth, i.e. the criterion a user of the reference would set for this curve."""
    def run(crit):
        stc = engine.Settings(n_contrib=ncontrib, n_reps=reps, max_iter=max_iter, conv_crit=crit, max_retries=0,
                              seed=20250101, rep_offset=first, device=dev_index, exec_mode=mode)
        t0 = time.perf_counter()
        conv = engine.analyse(setup, q, I, sigma, stc)
        wall = time.perf_counter() - t0
        return {"criterion": crit, "max_iterations": max_iter, "wall_s": wall, "converged": int(conv.converged.sum()), "reps": reps,
                "chisq_min": float(conv.chisq.min()), "chisq_median": float(np.median(conv.chisq)), "chisq_max": float(conv.chisq.max()),
                "steps_mean": float(conv.num_iter.mean()), "mc_steps_per_s_incl_setup": float(conv.num_iter.sum()) / wall}
    out = run(1.0)
    out["chisq_of_truth"] = chisq_of_truth
    if chisq_of_truth is not None and out["converged"] < reps:
        out["at_reachable_criterion"] = run(round(1.15 * chisq_of_truth, 3))
    return out


def series_two_streams(engine, torch, setup, q, I, sigma, st, first, seed, seconds=1.0):
    """Whole-job rate of back-to-back analyses with TWO plans on streams of their own, two result slots each."""
    pls = [engine.Plan(setup, q, I, sigma, st) for _ in range(2)]
    sts = [torch.cuda.Stream() for _ in pls]
    lanes2 = [(k, sl) for sl in range(2) for k in range(2)]
    pend, steps, n = [], 0, 0

    def retire():
        k, sl = pend.pop(0)
        pls[k].fetch(slot=sl)
        return pls[k].total_steps

    for i in range(8):                                        # warm-up: every slot twice
        k, sl = lanes2[i % 4]
        while (k, sl) in pend:
            retire()
        pls[k].reseed(seed + i, first); pls[k].launch(stream=sts[k].cuda_stream, slot=sl); pend.append((k, sl))
    while pend:
        retire()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while True:
        k, sl = lanes2[n % 4]
        while (k, sl) in pend or len(pend) >= 4:
            steps += retire()
        pls[k].reseed(seed + 50 + n, first); pls[k].launch(stream=sts[k].cuda_stream, slot=sl); pend.append((k, sl)); n += 1
        if time.perf_counter() - t0 >= seconds:
            break
    while pend:
        steps += retire()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for pl in pls:
        pl.close()
    return {"value": steps / dt, "unit": "MC steps/s", "streams": 2, "result_slots": 2, "analyses": n, "timed_region_s": dt,
            "ms_per_analysis_effective": dt / n * 1e3,
            "note": "throughput of a series of data sets (mcsas_amd.run_series(overlap=True)): two analyses share the chip; not the "
                    "latency of one analysis"}


def calc_breakdown(wl, dev_index, trials=5):
    """McSAS.calc() of config 2 through the front end (mcsas_amd.McSAS), split into its two halves: analyse() (mcsas.py:191-285) and
    histogram() (:445-615) with one 50-bin volume-weighted log histogram — fixed budget of 20 000 steps, showIncomplete."""
    import mcsas_amd
    q = wl["q"]
    ta, th = [], []
    for t in range(trials):
        m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
        lo, hi = m.radius.activeRange()
        m.radius.histograms().append(mcsas_amd.Histogram(m.radius, lo, hi, binCount=50, xscale='log', yweight='vol'))
        algo = mcsas_amd.McSAS(seed=900 + t, device=dev_index)
        algo.numContribs.setValue(wl["n"]); algo.numReps.setValue(wl["reps_gpu"]); algo.maxIterations.setValue(20000)
        algo.convergenceCriterion.setValue(0.0); algo.showIncomplete.setValue(True)
        algo.maxRetries = mcsas_amd.mcsas._Setting("maxRetries", 0)          # one attempt: the fixed budget of the timed workload
        algo.model = m
        algo.data = mcsas_amd.SASData(q, wl["I"], wl["sigma"])
        algo.result = []; algo.stop = False
        logging_off = __import__("logging"); logging_off.disable(logging_off.WARNING)
        t0 = time.perf_counter(); algo.analyse(); t1 = time.perf_counter(); algo.histogram(); t2 = time.perf_counter()
        logging_off.disable(logging_off.NOTSET)
        ta.append((t1 - t0) * 1e3); th.append((t2 - t1) * 1e3)
    ta, th = np.array(ta[1:]), np.array(th[1:])
    return {"analyse_ms": float(np.median(ta)), "histogram_ms": float(np.median(th)), "calc_ms": float(np.median(ta + th)),
            "histogram_over_analyse": float(np.median(th) / np.median(ta)),
            "workload": "config 2 through mcsas_amd.McSAS: 50 reps x 400 contribs x 20000 steps, one 50-bin log histogram (vol); wall times"}


class ClockSampler(object):
    """sclk / socket power of the GPU while a region runs, read from sysfs (hwmon freq1_input / power1_average|power1_input of the
    device's card) every 50 ms on a thread: backs the statement that the wave-per-chain rate follows the chip's power state."""

    def __init__(self, dev_index):
        import glob
        self.freq, self.power, self.how = None, None, None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input"))]
        pick = None
        try:                                                 # the card whose PCI address is the HIP device's
            import torch
            pr = torch.cuda.get_device_properties(max(dev_index, 0))
            addr = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in cards:
                if addr in os.path.realpath(os.path.join(c, "device")):
                    pick, self.how = c, "pci " + addr + "0"
        except Exception:
            pass
        self.cards = cards if pick is None else [pick]       # (no match: sample every card, report the busiest)
        self.samples, self._stop, self._th = [], False, None

    def _paths(self, c):
        p = None
        for name in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(c, name)):
                p = os.path.join(c, name)
                break
        return os.path.join(c, "freq1_input"), p

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().split()[0])
        except Exception:
            return None

    def _loop(self):
        paths = [self._paths(c) for c in self.cards]
        while not self._stop:
            self.samples.append([(self._read(f), self._read(p) if p else None) for f, p in paths])
            time.sleep(0.05)

    def __enter__(self):
        import threading
        if self.cards and not os.environ.get("MCSAS_BENCH_NO_CLOCKS"):     # (measurement knob: is the sampling itself visible in the rate?)
            self._th = threading.Thread(target=self._loop, daemon=True); self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self):
        if not self.samples:
            return {"available": False, "note": "no hwmon freq1_input under /sys/class/drm: clocks not sampled"}
        best = None
        for i, c in enumerate(self.cards):
            f = np.array([s[i][0] for s in self.samples if s[i][0] is not None]) / 1e6          # Hz -> MHz
            p = np.array([s[i][1] for s in self.samples if s[i][1] is not None]) / 1e6          # uW -> W
            if len(f) and (best is None or (len(p) and np.median(p) > best[3])):
                best = (c, f, p, float(np.median(p)) if len(p) else 0.0)
        if best is None:
            return {"available": False, "note": "hwmon entries unreadable"}
        c, f, p, _ = best
        out = {"available": True, "samples": int(len(f)), "sclk_mhz": {"min": float(f.min()), "median": float(np.median(f)), "max": float(f.max())},
               "source": os.path.join(c, "freq1_input"), "card_chosen_by": self.how or "highest median power of %d cards sampled" % len(self.cards)}
        if len(p):
            out["power_w"] = {"min": float(p.min()), "median": float(np.median(p)), "max": float(p.max())}
        for name in ("power1_cap", "power1_cap_max", "power1_cap_default"):
            v = self._read(os.path.join(c, name))
            if v is not None:
                out[name + "_w"] = v / 1e6
        return out


def many_chains(wl, dev_index, reps=8192, mc_steps=20000, seconds=1.5):
    """The kernel north_star describes literally — one wavefront per chain, thousands of chains: Sphere 512 q x 400 contributions,
    8192 repetitions, MCSAS_EXEC_WAVE, back-to-back launches of 20 000 steps per chain (the headline workload's budget per launch;
    chain initialisation included) sustained over >= `seconds`, clocks and power sampled beside it."""
    import torch
    from mcsas_amd import engine
    setup = wl["model"].setup()
    st = engine.Settings(n_contrib=wl["n"], n_reps=reps, max_iter=mc_steps, conv_crit=0.0, max_retries=0, seed=20250101,
                         device=dev_index, exec_mode=engine.EXEC_WAVE)
    plan = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st)
    for i in range(2):
        plan.reseed(40 + i, 0); plan.launch(); plan.fetch(want_arrays=False)
    torch.cuda.synchronize()
    ms, steps, n = [], 0, 0
    with ClockSampler(dev_index) as cs:
        t0 = time.perf_counter()
        while True:
            plan.reseed(50 + n, 0); plan.launch(); plan.fetch(want_arrays=False)
            ms.append(plan.last_ms); steps += plan.total_steps; n += 1
            if time.perf_counter() - t0 >= seconds:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    info = plan.info
    plan.close()
    engine.release_cached_memory()
    rate = steps / dt
    e = {"workload": "Sphere 512 q x 400 contribs, %d reps, one wavefront per chain, %d MC steps per chain per launch" % (reps, mc_steps),
         "value": rate, "unit": "MC steps/s", "timed_region_s": dt, "launches": n, "exec_mode": info["exec_mode"],
         "waves_per_chain": info["waves_per_chain"], "q_per_lane": info["q_per_lane"],
         "launch_ms": {"min": float(np.min(ms)), "median": float(np.median(ms)), "max": float(np.max(ms))},
         "rate_of_fastest_launch": reps * mc_steps / (float(np.min(ms)) * 1e-3), "rate_of_slowest_launch": reps * mc_steps / (float(np.max(ms)) * 1e-3),
         "clocks": cs.summary(),
         # the chain state of this kernel never leaves the chip (ft in registers, tables in LDS): SURVEY 8d's 40 Q bytes per step are an
         # algorithmic figure that no memory system carries, so there is no `frac` against HBM; what bounds the kernel is fp64 issue
         "algorithmic_gbps": 40 * len(wl["q"]) * rate / 1e9,
         "algorithmic_note": "40*Q B per MC step x steps/s (SURVEY 8d streaming model); NOT memory traffic and not a roofline fraction"}
    ij, pvt = latest_profile("valu_per_step.json")
    pv = pvt.get("wave8192")
    if pv:
        r = pv["valu_wave_instr_per_mc_step"] * rate
        e["roofline"] = {"bound": "valu", "achieved": r / 1e9, "peak": FP64_VECTOR_PEAK_INSTR / 1e9, "unit": "G wave-instr/s",
                         "frac": r / FP64_VECTOR_PEAK_INSTR, "instr_per_mc_step": pv["valu_wave_instr_per_mc_step"],
                         "measured_in_this_run": False, "profile_taken_on_these_kernel_sources": pv.get("csrc_sha16") == kernel_sources_sha16(),
                         "source": "from_profile: %s (SQ_INSTS_VALU pass, commit %s); rate measured in this run" % (ij, pv.get("commit", "?"))}
    return e


def other_configs(dev_index, seconds=1.0):
    """BASELINE configs 3-5 at their per-GPU repetition counts (200 / 400 / 100 repetitions over 8 GPUs -> 25 / 50 / 13),
    SUSTAINED: back-to-back launches of a fixed budget until at least `seconds` of wall time have passed (device
    synchronised on both sides), value = all MC steps / that time — chain initialisation (N form-factor rows per chain and
    launch) included.  The initialisation alone (a zero-step launch of the same plan, median of three) is reported beside
    it, and the rate with it taken out as a secondary figure."""
    import torch
    from mcsas_amd import engine
    out = {}
    ij, prof = latest_profile("valu_per_step.json")
    tj, traf = latest_profile("pmc_traffic.json")
    # budgets per launch: config 4 reaches BASELINE's criterion after 1.5e4 steps on average (its convergence_run): that many; the
    # other two do not get there within 1e5 steps: 1e4 (chain initialisation — N rows per chain — is then 4-6 % of a launch)
    for cfg, budget in ((3, 10000), (4, 15000), (5, 10000)):
        wl = workload(cfg, dev_index)
        setup = wl["model"].setup()
        st0 = engine.Settings(n_contrib=wl["n"], n_reps=wl["reps_gpu"], max_iter=0, conv_crit=0.0, max_retries=0,
                              seed=20250101, device=dev_index)
        plan0 = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st0)
        init_ms = []
        for rep in range(4):
            plan0.reseed(70 + rep, 0); plan0.launch(); plan0.fetch(want_arrays=False); init_ms.append(plan0.last_ms)
        plan0.close()
        init = float(np.median(init_ms[1:]))
        st = engine.Settings(n_contrib=wl["n"], n_reps=wl["reps_gpu"], max_iter=budget, conv_crit=0.0, max_retries=0,
                             seed=20250101, device=dev_index)
        plan = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st)
        plan.reseed(76, 0); plan.launch(); plan.fetch(want_arrays=False)          # warm-up
        torch.cuda.synchronize()
        with ClockSampler(dev_index) as cs1:
            t0 = time.perf_counter()
            steps, ms, n, res = 0, [], 0, None
            while True:
                plan.reseed(77 + n, 0); plan.launch(); res = plan.fetch()
                steps += plan.total_steps; ms.append(plan.last_ms); n += 1
                if time.perf_counter() - t0 >= seconds:
                    break
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        info = plan.info
        rate = steps / dt
        # ... and the same with a second plan on a stream of its own, two result slots each (the main workload's scheme): the
        # analyses fill each other's tails — a tick of these kernels ends with its slowest row, and 13 Kholodenko chains load the
        # chip in one uneven round
        plan_b = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st)
        pls, sts = [plan, plan_b], [torch.cuda.Stream(), torch.cuda.Stream()]
        lanes2 = [(k, sl) for sl in range(2) for k in range(2)]
        pend, steps2, n2 = [], 0, 0

        def retire2():
            k, sl = pend.pop(0)
            pls[k].fetch(slot=sl)
            return pls[k].total_steps

        for i in range(4):                                    # warm-up: every slot once
            k, sl = lanes2[i]
            pls[k].reseed(300 + i, 0); pls[k].launch(stream=sts[k].cuda_stream, slot=sl); pend.append((k, sl))
        while pend:
            retire2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while True:
            k, sl = lanes2[n2 % 4]
            while (k, sl) in pend or len(pend) >= 4:
                steps2 += retire2()
            pls[k].reseed(400 + n2, 0); pls[k].launch(stream=sts[k].cuda_stream, slot=sl); pend.append((k, sl)); n2 += 1
            if time.perf_counter() - t0 >= seconds:
                break
        while pend:
            steps2 += retire2()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        plan.close(); plan_b.close()
        rate_one, rate = rate, steps2 / dt2
        dev_s = float(np.sum(ms)) * 1e-3
        nq = len(wl["q"])
        pts = nq * wl["K"]                                    # form-factor points evaluated per MC step: the `new` row (`old` comes from the row cache)
        e = {"workload": "%s, %d reps (per-GPU share of %d), %d MC steps per chain per launch" % (wl["name"], wl["reps_gpu"], wl["reps_total"], budget),
             "value": rate, "unit": "MC steps/s", "timed_region_s": dt2, "launches": n2, "mc_steps": steps2,
             "streams": 2, "result_slots": 2,
             "one_stream": {"value": rate_one, "timed_region_s": dt, "launches": n, "mc_steps": steps,
                            "launch_ms": {"min": float(np.min(ms)), "median": float(np.median(ms)), "max": float(np.max(ms))},
                            "init_ms": init, "value_excl_init": steps / max(dev_s - n * init * 1e-3, 1e-9), "clocks": cs1.summary()},
             "exec_mode": info["exec_mode"], "window": info["window"], "final_chisq_median": float(np.median(res.chisq)),
             "chisq_of_truth": wl["chisq_of_truth"],
             "ff_points_per_s": rate * pts, "ff_points_per_mc_step": pts}
        # the same repetitions run to BASELINE's criterion (1, maxIterations 1e5, one attempt): final chi² and how many got there
        e["convergence_run"] = convergence_run(engine, setup, wl["q"], wl["I"], wl["sigma"], wl["n"], wl["reps_gpu"], 0, dev_index,
                                               wl["chisq_of_truth"])
        # (roofline objects: the kernels of one analysis by themselves, like the main workload's)
        alg = {"bound": "hbm", "achieved": 40 * nq * rate_one / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": 40 * nq * rate_one / HBM_PEAK}
        pv = prof.get(str(cfg))
        tr = traf.get(str(cfg))
        traffic = (tr["fetch_bytes_per_mc_step"] + tr["write_bytes_per_mc_step"]) * rate_one / 1e9 if tr else None
        e["roofline"] = dict(alg, traffic=traffic, traffic_unit="GB/s", measured_in_this_run=True, traffic_measured_in_this_run=False,
                             traffic_source="from_profile: %s" % tj if tr else None)
        if pv:
            r = pv["valu_wave_instr_per_mc_step"] * rate_one
            e["roofline_valu"] = {"bound": "valu", "achieved": r / 1e9, "peak": FP64_VECTOR_PEAK_INSTR / 1e9, "unit": "G wave-instr/s",
                                  "frac": r / FP64_VECTOR_PEAK_INSTR, "instr_per_mc_step": pv["valu_wave_instr_per_mc_step"],
                                  "measured_in_this_run": False, "profile_taken_on_these_kernel_sources": pv.get("csrc_sha16") == kernel_sources_sha16(),
                                  "source": "from_profile: %s / %s (commit %s)" % (ij, tj, pv.get("commit", "?"))}
        out[str(cfg)] = e
    return out


def python_only_model(wl, dev_index, chains=16, steps=2000, window=64):
    """A ScatteringModel that exists only as Python (the reference's plug-in contract: numpy formfactor / volume, no kernel id, no HIP
    text) on config 2's data: the library draws the proposals and calls back into the model's calcIntensity for the rows of a window
    of steps, the device does the rest (mcsas_hip_analyse_host_rows).  Host-bound by construction: the rate is what the model's own
    Python costs per row (the reference: the same row evaluation TWICE per step plus a MINPACK fit, 2.3-2.9e3 steps/s per core)."""
    import mcsas_amd
    from mcsas_amd import engine
    from mcsas_amd.scatteringmodels import host_model_calc

    class PythonOnlySphere(mcsas_amd.SASModel):
        shortName = "Sphere (Python only)"
        parameters = mcsas_amd.Sphere.parameters

        def __init__(self):
            super().__init__()
            self.radius.setActive(True)

        def volume(self):
            return (np.pi * 4. / 3.) * self.radius()**3

        def absVolume(self):
            return self.volume() * self.sld()**2

        def formfactor(self, dataset):
            qr = self.getQ(dataset) * self.radius()
            return 3. * (np.sin(qr) - qr * np.cos(qr)) / (qr**3.)

    q = wl["q"]
    m = PythonOnlySphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    data = mcsas_amd.SASData(q, wl["I"], wl["sigma"])
    st = engine.Settings(n_contrib=wl["n"], n_reps=chains, max_iter=steps, conv_crit=0.0, max_retries=0, seed=20250101, device=dev_index)
    spent = [0.0, 0]

    def rows(pset):
        t0 = time.perf_counter()
        out = host_model_calc(m, data, pset, st.comp_exp, want_rows=True)[4]
        spent[0] += time.perf_counter() - t0; spent[1] += len(pset)
        return out
    t0 = time.perf_counter()
    res = engine.analyse_host_rows(m.setup(data), q, wl["I"], wl["sigma"], st, rows, window=window)
    dt = time.perf_counter() - t0
    return {"workload": "Sphere typed as a Python-only model (numpy formfactor / volume), config 2's data: %d q x %d contribs, %d reps x %d "
                        "steps, window %d" % (len(q), wl["n"], chains, steps, window),
            "value": float(res.num_iter.sum()) / dt, "unit": "MC steps/s", "wall_s": dt, "rows_evaluated": spent[1],
            "callback_share": spent[0] / dt, "python_us_per_row": spent[0] / max(spent[1], 1) * 1e6,
            "final_chisq_median": float(np.median(res.chisq)),
            "note": "host-bound: the device waits for the model's own Python; everything but the rows (row cache, fit, chi², decisions) is on the GPU"}


def quickstart(dev_index):
    """The one workload the reference publishes a time for (doc/source/quickstart.rst:66-107: Sphere on
    testdata/quickstartdemo1.csv, 10 repetitions x 300 contributions, convergence criterion 1, one 50-bin log histogram:
    "36 seconds on a 3.4 GHz intel i7 iMac"): McSAS.calc() — analyse() AND histogram() — on the data vectors of the fixture
    the reference itself was run on in the build container (tests/golden/g13_quickstart.npz; its calc() took 23.6 s there)."""
    import mcsas_amd
    path = os.path.join(ROOT, "tests", "golden", "g13_quickstart.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    lo, hi = float(g["lo"]), float(g["hi"])
    walls, chis, iters = [], None, None
    for trial in range(4):                                     # the first call pays library / device start-up
        m = mcsas_amd.Sphere(); m.radius.setActiveRange((lo, hi))
        m.radius.histograms().append(mcsas_amd.Histogram(m.radius, lo, hi, binCount=50, xscale='log', yweight='vol'))
        algo = mcsas_amd.McSAS(seed=100 + trial, device=dev_index)
        algo.numContribs.setValue(300); algo.numReps.setValue(10); algo.convergenceCriterion.setValue(1.0)
        algo.model = m
        algo.data = mcsas_amd.SASData(g["data_q"], g["data_I"], g["data_sigma"], f_limit=g["data_f_limit"])
        t0 = time.perf_counter()
        algo.calc()
        walls.append(time.perf_counter() - t0)
        chis, iters = algo.details.chisq, algo.details.num_iter
    return {"workload": "Sphere, testdata/quickstartdemo1.csv (100 q after the reference's rebinning), 10 reps x 300 contribs, "
                        "convergenceCriterion 1, McSAS.calc() incl. histogram()",
            "quickstart_wall_s": float(np.median(walls[1:])), "first_call_wall_s": float(walls[0]),
            "chisq_max": float(np.max(chis)), "converged": int((chis <= 1.0).sum()), "steps_mean": float(np.mean(iters)),
            "reference_doc_wall_s": 36.0, "reference_doc_source": "doc/source/quickstart.rst:106-107 (3.4 GHz i7 iMac, 2012)",
            "reference_here_wall_s": float(g["wall_s"]),
            "reference_here_source": "oracle/make_golden.py gen_quickstart: the reference's calc() in the build container (1 core)"}


if __name__ == "__main__":
    main()
