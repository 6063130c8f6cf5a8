"""What the GPU acceptance test will see: the numpy oracle free-running on the Philox streams of the device (HIP == oracle chain
for chain with the same counter-based stream), against the reference's own free run (G16)."""
import sys, os, numpy as np, multiprocessing as mp
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import mcsas_oracle as O
import helpers
tag = sys.argv[1]; R = int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 77
g = {k: v for k, v in helpers.load("g16_%s_free.npz" % tag).items()}
model = str(g["spec_model"])
def build():
    extra = {}
    if model == "cyl_aspect": extra = dict(sld=float(g["spec_sld"]), intDiv=float(g["spec_int_div"]))
    if model == "ellcs": extra = dict(eta_c=float(g["spec_eta_c"]), eta_s=float(g["spec_eta_s"]), eta_sol=float(g["spec_eta_sol"]), intDiv=float(g["spec_int_div"]))
    return helpers.make_models(model, g["spec_lo"], g["spec_hi"], [int(x) for x in g["spec_gen"]], **extra)
def chain(r):
    os.environ["OMP_NUM_THREADS"] = "1"
    m, spec = build()
    st = O.Settings(n_contrib=int(g["n_contrib"]), n_reps=1, max_iter=int(g["max_iter"]), conv_crit=float(g["crit"]), comp_exp=float(g["spec_comp_exp"]))
    att = 0
    stream = O.PhiloxStream(seed, r)
    while True:
        res = O.mc_fit(spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"], g["data_x0_limit"], st, stream, method="closed")
        att += 1
        if res.conval <= st.conv_crit or att > 6: break
    return res.rset, res.conval, res.num_iter, att
if __name__ == "__main__":
    with mp.Pool(6) as pool:
        out = pool.map(chain, range(R))
    contribs = np.stack([o[0] for o in out], axis=2)
    chis = np.array([o[1] for o in out]); iters = np.array([o[2] for o in out])
    print("chisq max", chis.max(), "iters mean", iters.mean(), "ref", float(g["numIter"]), "attempts", [o[3] for o in out])
    m, spec = build()
    st = O.Settings(n_contrib=int(g["n_contrib"]), n_reps=R, conv_crit=float(g["crit"]), comp_exp=float(g["spec_comp_exp"]))
    frac, sc = O.fractions(spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"], st, contribs, method="closed")
    Rr = int(g["reps"])
    for k in range(int(g["n_hist"])):
        pre = "h%d_" % k
        pidx = list(helpers.CASES[model]["active"]).index(str(g[pre + "param"]))
        h = O.histogram_calc(contribs, pidx, frac, float(g[pre + "lo"]), float(g[pre + "hi"]), int(g[pre + "nbin"]), str(g[pre + "xscale"]), str(g[pre + "yweight"]))
        ours, ours_se = np.asarray(h["bins_mean"]), np.asarray(h["bins_std"]) / np.sqrt(R)
        ref, ref_se = g[pre + "bins_mean"], g[pre + "bins_std"] / np.sqrt(Rr)
        for floor in (0.01, 0.02, 0.03):
            se = np.sqrt(ours_se**2 + ref_se**2) + floor * ref.max()
            z = (ours - ref) / se
            print(pre, str(g[pre+"param"]), "floor", floor, "zmax %.2f zrms %.2f" % (np.abs(z).max(), np.sqrt(np.mean(z**2))))
        mo = np.asarray(h["moments"], dtype=float)
        print("   moments ours", mo[[0, 2]], "ref", g[pre + "moments"][[0, 2]], "ratio", mo[[0,2]] / g[pre + "moments"][[0, 2]])
    pass
