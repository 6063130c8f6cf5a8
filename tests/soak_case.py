#!/usr/bin/env python3
"""One case of tests/soak_modes.py against the numpy oracle on the same Philox streams (not collected by pytest):
which execution mode follows the oracle where they part ways.   python tests/soak_case.py <case>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_parity_gpu as T
from mcsas_amd import engine
O = T.O
case = int(sys.argv[1])
rs = np.random.RandomState(1000 + case)
tag = list(T.RANDOM_RANGES)[case % len(T.RANDOM_RANGES)]
heavy = tag in ("cyl_aspect", "cyl_length", "ellcs", "kholodenko", "elliso")
nq = int(rs.choice([5, 33, 64, 100, 257] if heavy else [5, 33, 64, 100, 257, 512, 700]))
n = int(rs.choice([16, 24, 50, 130] if heavy else [16, 24, 50, 130, 300, 400]))
reps = int(rs.randint(1, 5)); steps = int(rs.randint(1, 4 * n))
lo, hi = T.RANDOM_RANGES[tag]
m, spec = T.make_models(tag, lo, hi)
q = np.sort(10 ** rs.uniform(7.2, 9.3, nq))
truth = np.array([10 ** rs.uniform(np.log10(a), np.log10(b), 12) for a, b in zip(lo, hi)]).T
I = O.model_calc(spec, q, truth, 0.6666666)[0]
I = I * (1 + 0.03 * rs.standard_normal(nq)) + 0.02 * I.mean()
sig = 0.03 * np.abs(I) + 1e-3 * np.abs(I).mean()
kw = dict(find_background=bool(rs.randint(2)), positive_background=bool(rs.randint(2)),
          start_from_minimum=bool(rs.randint(4) == 0), max_retries=int(rs.randint(0, 3)),
          conv_crit=float(rs.choice([1e-9, 5.0, 200.0])))
print(tag, "nq", nq, "n", n, "reps", reps, "steps", steps, kw)
for mode in (engine.EXEC_WAVE, engine.EXEC_WORKGROUP, engine.EXEC_PIPELINE):
    st = engine.Settings(n_contrib=n, n_reps=reps, max_iter=steps, seed=77 + case, exec_mode=mode, **kw)
    res = engine.analyse(m.setup(T.FakeData(q)), q, I, sig, st)
    print("mode", mode, "moves", res.num_moves.tolist(), "iter", res.num_iter.tolist(), "attempts", res.attempts.tolist(), "chisq", res.chisq.tolist())
ost = O.Settings(n_contrib=n, n_reps=1, max_iter=steps, conv_crit=kw["conv_crit"], find_bg=kw["find_background"], pos_bg=kw["positive_background"],
                 start_from_min=kw["start_from_minimum"], max_retries=0)
if kw["max_retries"] == 0 or True:
    for r in range(reps):
        ref = O.mc_fit(spec, q, I, sig, [I.min(), I.max()], [q.min(), q.max()], ost, O.PhiloxStream(77 + case, r), method="closed")
        print("oracle rep", r, "first attempt: moves", ref.num_moves, "iter", ref.num_iter, "chisq", ref.conval)
