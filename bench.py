#!/usr/bin/env python3
"""bench.py — MC accept/reject steps per second of the McSAS hot path on MI355X.

Workload (BASELINE.json configs[1]): Sphere model, synthetic 512 q-points x 400 contributions,
50 repetitions per GPU, fixed Monte-Carlo budget per chain (convergenceCriterion = 0 so no chain
leaves early; SURVEY §8d "throughput runs").  One bench "step" = one launch of all chains of the
rank for `--mc-steps` MC iterations each, data and workspaces already resident in HBM.
With --gpus N (one process per GPU, launched by torch.distributed.run) every rank runs its own
50 reps (weak scaling) and one RCCL all-gather per step assembles the results.

Prints ONE JSON line; see the module-level keys `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q, NCONTRIB, REPS_PER_GPU = 512, 400, 50
HBM_PEAK = 8.0e12                     # B/s, MI355X_MICROARCH.md
BYTES_PER_MC_STEP = 40 * Q            # read q, I, sigma, ft + write ft_test, fp64


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


def cpu_baseline(q, I, sigma, lo, hi, seconds_target=12.0):
    """The compiled CPU oracle (oracle/c/mcsas_oracle.c: plain-C restatement of mcFit with the closed-form
    fit and cached rows, libm sin/cos) on every host core this process may use, one chain per thread at
    a time, same workload shape, bounded sample.  The numpy restatement's single-core rate is reported
    beside it (`numpy_port_1core`)."""
    from oracle import mcsas_oracle as O
    from oracle import c_oracle
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, int(os.environ.get("MCSAS_BENCH_CPU_THREADS", "16"))))   # one GPU's CPU share of the host
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
           "sample": "%d chains x %d MC steps (incl. %d-contribution init each), Sphere %dq x %d contribs, C oracle "
                     "(oracle/c, gcc -O2, libm), %d threads" % (threads, steps, NCONTRIB, Q, NCONTRIB, threads)}
    spec = O.ModelSpec.make("sphere", ["radius"], [lo], [hi])
    st = O.Settings(n_contrib=NCONTRIB, n_reps=1, max_iter=3000, conv_crit=0.0)
    t0 = time.time()
    rn = O.mc_fit(spec, q, I, sigma, [I.min(), I.max()], [q.min(), q.max()], st, O.PhiloxStream(1, 0), method="closed")
    out["numpy_port_1core"] = rn.num_iter / (time.time() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mc-steps", type=int, default=20000, help="MC iterations per chain per launch")
    ap.add_argument("--reps", type=int, default=REPS_PER_GPU, help="repetitions (chains) per GPU")
    ap.add_argument("--waves", type=int, default=0, help="waves per chain (0 = library default)")
    ap.add_argument("--mode", type=int, default=0, help="exec_mode: 0 auto, 1 wave, 2 workgroup, 3 pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence-run", action="store_true",
                    help="skip the untimed run-to-convergence (profiling: only the timed launches reach the profiler)")
    ap.add_argument("--debug-flags", type=int, default=0, help="diagnostic role ablation (invalid results)")
    args = ap.parse_args()

    import torch
    import mcsas_amd
    from mcsas_amd import engine, dist as mdist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    # one process per GPU.  MCSAS_BENCH_BACKEND=gloo lets several ranks share one card for a dry run of
    # the multi-process path on a single-GPU box (results gathered through host memory)
    backend = os.environ.get("MCSAS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as tdist
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            tdist.init_process_group(backend)

    q, I, sigma = synthetic_data()
    lo, hi = np.pi / q.max(), np.pi / q.min()
    model = mcsas_amd.Sphere()
    model.radius.setActiveRange((lo, hi))
    n_total = args.reps * world
    first = rank * args.reps
    st = engine.Settings(n_contrib=NCONTRIB, n_reps=args.reps, max_iter=args.mc_steps, conv_crit=0.0,
                         max_retries=0, seed=20250101, rep_offset=first, device=dev_index,
                         waves_per_chain=args.waves, exec_mode=args.mode, debug_flags=args.debug_flags)
    plan = engine.Plan(model.setup(), q, I, sigma, st)

    def one_step(seed):
        plan.reseed(seed, first)
        plan.launch()
        res = plan.fetch()
        if use_dist:
            mdist.gather_results(dict(contribs=np.moveaxis(res.contribs, 2, 0), chisq=res.chisq[:, None],
                                      scaling=res.scaling[:, None], background=res.background[:, None],
                                      fit=res.fit.T), n_total)
        return res

    def barrier():
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        one_step(1000 + w)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, mc_steps, res = 0.0, 0, None
    for k in range(args.steps):
        res = one_step(2000 + k)
        kernel_ms += plan.last_ms
        mc_steps += plan.total_steps
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt, float(mc_steps)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tmax = t.clone(); tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        tsum = t.clone(); tdist.all_reduce(tsum, op=tdist.ReduceOp.SUM)
        dt, total_steps = float(tmax[0]), float(tsum[1])
    else:
        total_steps = float(mc_steps)

    if rank == 0:
        launch_s = kernel_ms * 1e-3 / args.steps
        achieved = BYTES_PER_MC_STEP * (mc_steps / args.steps) / launch_s
        # HBM-side bytes per MC step from the committed PMC passes of this same command
        # (profiles/r01_pmc_summary.md: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); pipeline mode only
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_c_pmc_traffic.json")
        if plan.info["exec_mode"] == "pipeline" and os.path.exists(tj):
            pm = json.load(open(tj))
            traffic = (pm["fetch_bytes_per_mc_step"] + pm["write_bytes_per_mc_step"]) * (mc_steps / args.steps) / launch_s / 1e9
        out = {
            "metric": "MC accept/reject steps/sec (whole node)", "value": total_steps / dt, "unit": "MC steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Sphere, synthetic %d q-points x %d contribs, %d reps/GPU, %d MC steps per chain per launch, convergenceCriterion=0"
                                   % (Q, NCONTRIB, args.reps, args.mc_steps),
                       "reps_total": n_total, **plan.info},
            "final_chisq_median": float(np.median(res.chisq)),
            "kernel_ms_per_launch": launch_s * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "note": "achieved = 40*Q B per MC step (SURVEY 8d streaming model, chain state actually stays on chip) x MC steps per launch / HIP-event time of the launch sequence"},
        }
        # outside the timed region: the same 50 repetitions run the way McSAS.analyse runs them
        # (maxIterations = 1e5, maxRetries = 5) -> final chi² and steps to converge.  Criterion 2: with 1 %
        # noise on the synthetic curve 400 spheres plateau at chi² ~1.15 after 1e5 steps, so the default
        # criterion 1 is never met on this data set (every repetition then burns all 6 attempts)
        CRIT = 2.0
        if not args.no_convergence_run:
            stc = engine.Settings(n_contrib=NCONTRIB, n_reps=args.reps, max_iter=100000, conv_crit=CRIT, max_retries=5,
                                  seed=20250101, rep_offset=first, device=dev_index, exec_mode=args.mode)
            t0 = time.perf_counter()
            conv = engine.analyse(model.setup(), q, I, sigma, stc)
            out["convergence_run"] = {"criterion": CRIT, "wall_s": time.perf_counter() - t0,
                                      "converged": int(conv.converged.sum()), "reps": args.reps,
                                      "chisq_max": float(conv.chisq.max()), "chisq_mean": float(conv.chisq.mean()),
                                      "steps_mean": float(conv.num_iter.mean()), "attempts_max": int(conv.attempts.max())}
        if not args.no_cpu_baseline and world == 1:           # the CPU baseline is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(q, I, sigma, lo, hi)
        print(json.dumps(out))
    if use_dist:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
