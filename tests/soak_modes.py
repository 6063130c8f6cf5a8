#!/usr/bin/env python3
"""Soak version of test_random_configurations_all_modes_identical (not collected by pytest): the same seeded sweep
for a range of case numbers, reporting — not asserting — where the execution modes part ways.  A numerically tied
decision shows up as a mismatch of the parameter sets in single cases; anything systematic is a bug.

    python tests/soak_modes.py 18 400        # cases 18 .. 399, on a GPU box"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_parity_gpu as T
from mcsas_amd import engine
import mcsas_amd

lo_case, hi_case = int(sys.argv[1]), int(sys.argv[2])
bad = []
for case in range(lo_case, hi_case):
    rs = np.random.RandomState(1000 + case)
    tag = list(T.RANDOM_RANGES)[case % len(T.RANDOM_RANGES)]
    heavy = tag in ("cyl_aspect", "cyl_length", "ellcs", "kholodenko", "elliso")
    nq = int(rs.choice([5, 33, 64, 100, 257] if heavy else [5, 33, 64, 100, 257, 512, 700]))
    n = int(rs.choice([16, 24, 50, 130] if heavy else [16, 24, 50, 130, 300, 400]))
    reps = int(rs.randint(1, 5))
    steps = int(rs.randint(1, 4 * n))
    lo, hi = T.RANDOM_RANGES[tag]
    m, spec = T.make_models(tag, lo, hi)
    q = np.sort(10 ** rs.uniform(7.2, 9.3, nq))
    truth = np.array([10 ** rs.uniform(np.log10(a), np.log10(b), 12) for a, b in zip(lo, hi)]).T
    I = T.O.model_calc(spec, q, truth, 0.6666666)[0]
    I = I * (1 + 0.03 * rs.standard_normal(nq)) + 0.02 * I.mean()
    sig = 0.03 * np.abs(I) + 1e-3 * np.abs(I).mean()
    kw = dict(find_background=bool(rs.randint(2)), positive_background=bool(rs.randint(2)),
              start_from_minimum=bool(rs.randint(4) == 0), max_retries=int(rs.randint(0, 3)),
              conv_crit=float(rs.choice([1e-9, 5.0, 200.0])))
    outs = []
    for mode in (engine.EXEC_WAVE, engine.EXEC_WORKGROUP, engine.EXEC_PIPELINE):
        st = engine.Settings(n_contrib=n, n_reps=reps, max_iter=steps, seed=77 + case, exec_mode=mode, **kw)
        try:
            outs.append((mode, engine.analyse(m.setup(T.FakeData(q)), q, I, sig, st)))
        except mcsas_amd._lib.McSASHipError as e:
            if not (e.code == -1 and mode != engine.EXEC_WAVE):
                bad.append((case, tag, nq, n, mode, "error %s" % e))
    ref = outs[0][1]
    for mode, res in outs[1:]:
        for f in ("contribs", "num_iter", "num_moves", "attempts", "converged", "draws"):
            if not np.array_equal(getattr(res, f), getattr(ref, f)):
                bad.append((case, tag, nq, n, mode, f, int(np.sum(getattr(res, f) != getattr(ref, f)))))
                break
    if (case - lo_case) % 50 == 49:
        print("case %d done, mismatches so far: %d" % (case, len(bad)), flush=True)
print("cases %d..%d: %d mismatches" % (lo_case, hi_case - 1, len(bad)))
for b in bad:
    print("  ", b)
