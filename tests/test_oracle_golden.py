"""Pins the CPU oracle (oracle/mcsas_oracle.py) against fixtures produced by the real reference
(oracle/make_golden.py) and against the SASfit known-answer files the reference's own tests name."""
import os
import numpy as np
import pytest

from oracle import mcsas_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


from helpers import CASES as MODEL_CASES, make_models, traj_setup as _traj_setup, SMEAR_CASES, oracle_smearing, traj_smearing


def spec_for(tag, lo=None, hi=None, gen=None, **extra):
    return make_models(tag, lo, hi, gen, **extra)[1]


@pytest.mark.parametrize("tag", [t for t in MODEL_CASES if t != "cylradiso"])      # (cylradiso: G18, below)
def test_g1_g2_model_vectors(tag):
    g = load("g12_models.npz")
    spec = spec_for(tag)
    q, pset = g[tag + "_q"], g[tag + "_pset"]
    c = float(g["comp_exp"])
    # kholodenko: QUADPACK result is deterministic for identical scipy -> expect (near) bit equality
    tol = 1e-13
    for row, ff_ref, it_ref in zip(pset, g[tag + "_ff"], g[tag + "_rows"]):
        it, v, w, s = O.calc_intensity(spec, q, row, c)
        np.testing.assert_allclose(it, it_ref, rtol=tol, atol=0)
        np.testing.assert_allclose(np.sqrt(it / w), np.abs(ff_ref), rtol=1e-12, atol=0)
    cum, vset, wset, sset = O.model_calc(spec, q, pset, c)
    np.testing.assert_allclose(cum, g[tag + "_cumInt"], rtol=tol)
    np.testing.assert_allclose(vset, g[tag + "_vset"], rtol=tol)
    np.testing.assert_allclose(wset, g[tag + "_wset"], rtol=tol)
    np.testing.assert_allclose(sset, g[tag + "_sset"], rtol=tol)


def test_g18_radially_isotropic_cylinders_vectors():
    """models/cylindersradiallyisotropic.py (G18, oracle/make_golden.py gen_cyl_radially_isotropic): formfactor and calc of the
    reference against the restatement."""
    g = load("g18_cylradiso_models.npz")
    spec = spec_for("cylradiso", aspect=float(g["aspect"]), sld=float(g["sld"]), psiAngleDivisions=float(g["divisions"]))
    np.testing.assert_allclose(g["psi_range"], O.PARAM_VALUE_RANGE[O.CYL_RAD_ISO][2], rtol=0)
    q, pset, c = g["q"], g["pset"], float(g["comp_exp"])
    for row, ff_ref, it_ref in zip(pset, g["ff"], g["rows"]):
        it, v, w, s = O.calc_intensity(spec, q, row, c)
        np.testing.assert_allclose(it, it_ref, rtol=1e-13)
        np.testing.assert_allclose(np.sqrt(it / w), np.abs(ff_ref), rtol=1e-12)
    cum, vset, wset, sset = O.model_calc(spec, q, pset, c)
    for got, name in ((cum, "cumInt"), (vset, "vset"), (wset, "wset"), (sset, "sset")):
        np.testing.assert_allclose(got, g[name], rtol=1e-13)


def test_g6_generators():
    g = load("g6_generators.npz")
    u = g["u"]
    for name, kind in (("uniform", O.GEN_UNIFORM), ("exp1", O.GEN_EXP1), ("exp2", O.GEN_EXP2), ("exp3", O.GEN_EXP3)):
        np.testing.assert_array_equal(O.transform(kind, u), g[name])
    spec = spec_for("cyl_aspect", g["cyl_lo"], g["cyl_hi"])
    out = O.generate_parameters(spec, O.ReplayStream(u), 32)
    np.testing.assert_array_equal(out, g["cyl_params"])
    spec = spec_for("sphere", g["sph_lo"], g["sph_hi"])
    np.testing.assert_array_equal(O.generate_parameters(spec, O.ReplayStream(u), 64), g["sph_params"])


def test_g3_bgfit_leastsq_and_closed():
    g = load("g3_bgfit.npz")
    I, sig = g["I"], g["sigma"]
    for C, (ci, fb, pb), sc0, simplex, lm in zip(g["C"], g["flags"], g["sc0"], g["simplex"], g["lm"]):
        sc1, cv1, ag1 = O.bgfit_calc(I, sig, C, sc0, bool(fb), bool(pb), ver=1, num_params=1)
        np.testing.assert_allclose([sc1[0], sc1[1], cv1, ag1], simplex, rtol=1e-12)
        sc2, cv2, ag2 = O.bgfit_calc(I, sig, C, sc1, bool(fb), bool(pb), num_params=1)
        np.testing.assert_allclose([sc2[0], sc2[1], cv2, ag2], lm, rtol=1e-12)
        # closed form == what MINPACK converges to (chi² to LM's own termination noise)
        sc3, cv3, ag3 = O.bgfit_calc(I, sig, C, sc0, bool(fb), bool(pb), num_params=1, method="closed")
        assert abs(cv3 - lm[2]) <= 1e-11 * lm[2]
        assert cv3 <= lm[2] * (1 + 1e-14)            # the closed form is the true minimum
        np.testing.assert_allclose(sc3[0], lm[0], rtol=1e-6)
        np.testing.assert_allclose(ag3, lm[3], rtol=1e-6)
    # negative free background: positiveBackground must land on the b = 0 boundary
    In, sn, C = g["neg_I"], g["neg_sigma"], g["neg_C"]
    free = O.bgfit_calc(In, sn, C, g["neg_sc0"], True, False, num_params=1, method="closed")
    assert free[0][1] < 0
    np.testing.assert_allclose(free[1], g["neg_free"][2], rtol=1e-11)
    pos = O.bgfit_calc(In, sn, C, g["neg_sc0"], True, True, num_params=1, method="closed")
    assert pos[0][1] == 0.0
    # LM on |b| stalls close to, not at, the kink: agree to 1e-6 and never be worse
    assert pos[1] <= g["neg_pos"][2] * (1 + 1e-12)
    np.testing.assert_allclose(pos[1], g["neg_pos"][2], rtol=1e-6)


def traj_setup(name):
    g, m, spec, st, ost = _traj_setup(name)
    return g, spec, ost


TRAJ = ["g4_sphere_q100_fixed.npz", "g4_sphere_q100_converge.npz", "g4_sphere_q512_fixed.npz",
        "g4_sphere_q100_nobg.npz", "g4_sphere_q100_posbg.npz", "g4_sphere_q100_frommin.npz",
        "g4_cyl_q40.npz", "g4_ellcs_q40.npz", "g4_kho_q24.npz", "g4_elliso_q40.npz", "g4_sphcs_q40.npz",
        "g4_gausschain_q40.npz", "g4_lmasphere_q40.npz",
        "g9_cyl_q512.npz", "g9_ellcs_q1024.npz", "g9_kho_q64.npz", "g9_kho_q512.npz",
        # round 3: config 2's shape (512 q x 400) over long budgets — 25 000 fixed steps (62 sweeps over the contributions) and a
        # chain that the reference ends by convergence (criterion 2, 5509 steps)
        "g14_sphere_q512_long.npz", "g14_sphere_q512_converge.npz",
        # round 4: chains the reference ENDS BY CONVERGENCE (criterion 1) for models with an orientation integral (cylinders 6228
        # steps, core-shell ellipsoids 1887 steps; 100 q x 200 contributions) and with positiveBackground (sphere, criterion 2, 5768 steps)
        "g17_cyl_q100_converge.npz", "g17_ellcs_q100_converge.npz", "g17_sphere_q100_posbg_converge.npz",
        # round 4: radially isotropic cylinders (the reference's "not verified" variant that runs as written), 40 q x 40 x 250 steps
        "g18_cylradiso_q40.npz",
        # round 5: config 5 as named over 1300 steps (two sweeps over the 600 contributions); ~50 min of numpy: MCSAS_SLOW_TESTS
        "g9_kho_q512_long.npz"]


def test_g17_positive_background_chain_that_only_minpack_follows():
    """A positiveBackground chain run to criterion 1 on data whose background is ~0 (g17_..._posbg_minpack, oracle/make_golden.py
    gen_converging_trajectories): the fit sits next to the |b| kink and the reference's MINPACK stops up to 1e-6 above the minimum,
    which decides a late step.  The call-for-call leastsq restatement replays the whole chain; the closed-form minimiser (what the
    kernels use) follows it for the first 584 of 606 accepted moves and then keeps the better chi² of the two."""
    g, spec, st = traj_setup("g17_sphere_q100_posbg_minpack.npz")
    args = (spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"], g["data_x0_limit"], st)
    res = O.mc_fit(*args, O.ReplayStream(g["stream"]), method="leastsq")
    assert res.num_iter == int(g["res_num_iter"]) and res.num_moves == int(g["res_num_moves"])
    np.testing.assert_array_equal(np.array(res.accepted), g["res_accepted"])
    np.testing.assert_allclose(res.rset, g["res_rset"], rtol=1e-15)
    np.testing.assert_allclose(res.conval, float(g["res_conval"]), rtol=1e-12)
    longer = np.concatenate([g["stream"], np.random.RandomState(1).random_sample(20000)])
    clo = O.mc_fit(*args, O.ReplayStream(longer), method="closed")
    acc, ref = np.array(clo.accepted), g["res_accepted"]
    n = min(len(acc), len(ref))
    first = int(np.nonzero(acc[:n] != ref[:n])[0][0])
    assert first == 584 and ref[first] < acc[first]           # the reference took a step the exact minimum does not justify
    assert clo.conval <= 1.0


@pytest.mark.parametrize("name", TRAJ)
@pytest.mark.parametrize("method", ["leastsq", "closed"])
def test_g4_replay_trajectories(name, method):
    """Replaying the uniform stream the reference consumed reproduces its accept/reject decisions,
    final parameter set and chi² (leastsq: call-for-call restatement; closed: the kernels' fit)."""
    if method == "leastsq" and (name in ("g4_sphere_q100_converge.npz",) or name.startswith("g9_") or name.startswith("g14_") or name.startswith("g17_")):
        pytest.skip("covered by the closed-form run (thousands of leastsq steps are slow)")
    if name in ("g9_kho_q512.npz", "g9_kho_q512_long.npz") and not os.environ.get("MCSAS_SLOW_TESTS"):
        pytest.skip("config 5 as named through the QUADPACK oracle takes ~10 min: set MCSAS_SLOW_TESTS=1 "
                    "(run once per oracle change in the build container; result recorded in DESIGN.md)")
    if not os.path.exists(os.path.join(G, name)):
        pytest.skip("fixture %s not generated yet" % name)
    g, spec, st = traj_setup(name)
    res = O.mc_fit(spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"],
                   g["data_x0_limit"], st, O.ReplayStream(g["stream"]), method=method)
    assert res.num_iter == int(g["res_num_iter"])
    np.testing.assert_array_equal(np.array(res.accepted), g["res_accepted"])
    assert res.num_moves == int(g["res_num_moves"])
    np.testing.assert_allclose(res.rset, g["res_rset"], rtol=1e-15)
    rtol = 1e-12 if method == "leastsq" else 1e-9
    if "posbg" in name and method == "closed":
        rtol = 1e-5
    if name == "g17_sphere_q100_posbg_converge.npz" and method == "closed":
        # the reference's LAST fit of this chain ran into MINPACK's maxfev next to the |b| kink and reports a chi² 0.4 % above the
        # minimum of the very same parameter set (leastsq restatement: 1e-12); the closed form finds the minimum
        assert float(g["res_conval"]) * (1 - 1e-2) < res.conval <= float(g["res_conval"]) * (1 + 1e-9)
        return
    np.testing.assert_allclose(res.conval, float(g["res_conval"]), rtol=rtol)
    # (atol: scale * model + background crosses zero on the worm data file, whose intensity spans 9 decades)
    np.testing.assert_allclose(res.fit, g["res_fit"], rtol=1e-6, atol=1e-12 * np.abs(g["res_fit"]).max())
    np.testing.assert_allclose(res.scaling, float(g["res_scaling"]), rtol=1e-6)


def test_g45_analyse_and_histogram():
    g = load("g45_analyse.npz")
    q, I, sig = g["data_q"], g["data_I"], g["data_sigma"]
    spec = spec_for("sphere", [float(g["A_lo"])], [float(g["A_hi"])])
    st = O.Settings(n_contrib=150, n_reps=3, max_iter=100000, conv_crit=5.0)
    stream = O.ReplayStream(g["A_stream"])
    res, info = O.analyse(spec, q, I, sig, g["data_f_limit"], g["data_x0_limit"], st, stream, method="closed")
    assert stream.pos == int(g["A_consumed"])
    np.testing.assert_allclose(res["contribs"], g["A_contribs"], rtol=1e-15)
    np.testing.assert_allclose(res["fitMeasValMean"], g["A_fitMean"], rtol=1e-6)
    np.testing.assert_allclose(res["fitMeasValStd"], g["A_fitStd"], rtol=1e-4, atol=1e-9 * g["A_fitMean"].max())
    np.testing.assert_allclose(res["scaling"], g["A_scaling"], rtol=1e-6)
    np.testing.assert_allclose(res["background"], g["A_background"], rtol=1e-4)
    assert res["numIter"] == float(g["A_numIter"])
    # G5: histogram of those contribs
    frac, scaling = O.fractions(spec, q, I, sig, g["data_f_limit"], st, g["A_contribs"], method="leastsq")
    for hi, (bc, xlog, yw) in enumerate(g["A_h_spec"]):
        h = O.histogram_calc(g["A_contribs"], 0, frac, float(g["A_lo"]), float(g["A_hi"]), int(bc),
                             "log" if xlog else "lin", O.YWEIGHTS[int(yw)])
        p = "A_h%d_" % hi
        np.testing.assert_allclose(h["edges"], g[p + "edges"], rtol=1e-15)
        np.testing.assert_allclose(h["bins_full"], g[p + "bins_full"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(h["bins_mean"], g[p + "bins_mean"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(h["bins_std"], g[p + "bins_std"], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(h["cdf_mean"], g[p + "cdf_mean"], rtol=1e-9)
        np.testing.assert_allclose(h["cdf_std"], g[p + "cdf_std"], rtol=1e-7, atol=1e-15)
        np.testing.assert_allclose(h["observability"], g[p + "obs"], rtol=1e-9)
        mom, ref = h["moments"], g[p + "moments"]
        np.testing.assert_allclose(mom[0::2], ref[0::2], rtol=1e-9)                  # means over reps
        for k in range(5):      # stds over reps: rounding noise of the mean is the floor
            assert abs(mom[2 * k + 1] - ref[2 * k + 1]) <= 1e-7 * abs(ref[2 * k + 1]) + 1e-13 * abs(ref[2 * k])


def test_g4_analyse_retries():
    g = load("g45_analyse.npz")
    q, I, sig = g["data_q"], g["data_I"], g["data_sigma"]
    spec = spec_for("sphere", [float(g["A_lo"])], [float(g["A_hi"])])
    st = O.Settings(n_contrib=50, n_reps=2, max_iter=60, conv_crit=1e-9, max_retries=2, show_incomplete=True)
    stream = O.ReplayStream(g["B_stream"])
    res, info = O.analyse(spec, q, I, sig, g["data_f_limit"], g["data_x0_limit"], st, stream, method="closed")
    assert stream.pos == int(g["B_consumed"])            # 2 reps x 3 attempts x (50 + 60) draws
    assert [i["attempts"] for i in info] == [3, 3]
    np.testing.assert_allclose(res["contribs"], g["B_contribs"], rtol=1e-15)
    np.testing.assert_allclose(res["fitMeasValMean"], g["B_fitMean"], rtol=1e-6)
    np.testing.assert_allclose(res["scaling"], g["B_scaling"], rtol=1e-6)
    assert res["numIter"] == float(g["B_numIter"])
    # without showIncomplete the reference returns without a result (mcsas.py:227-230)
    st.show_incomplete = False
    res, _ = O.analyse(spec, q, I, sig, g["data_f_limit"], g["data_x0_limit"], st,
                       O.ReplayStream(g["B_stream"]), method="closed")
    assert res is None


@pytest.mark.parametrize("R", [2, 10, 20, 50, 100])
def test_sasfit_sphere_known_answers(R):
    """sphere.py:68-75 (Sphere.testRelErr = 1e-4 on the MEAN relative error of (V·F)²)."""
    d = np.loadtxt(os.path.join(G, "ref_testdata", "sasfit_sphere-%d-1.dat" % R))
    q, Iref = d[:, 0], d[:, 1]                         # nm^-1, nm^6
    V = 4 * np.pi / 3 * R**3
    Icalc = (V * O.ff_sphere(q, float(R)))**2
    assert np.mean(np.abs((Iref - Icalc) / Iref)) < 1e-4


def test_sasfit_kholodenko_known_answer():
    """kholodenko.py:98-102 (testVolExp = 0, default testRelErr 1e-5); sampled q to bound run time."""
    d = np.loadtxt(os.path.join(G, "ref_testdata", "sasfit_kho-1-10-1000.dat"))[::25]
    q, Iref = d[:, 0], d[:, 1]
    Icalc = O.ff_kholodenko(q, 1.0, 10.0, 1000.0)**2
    assert np.mean(np.abs((Iref - Icalc) / Iref)) < 1e-5


def test_philox_known_answer():
    """Random123 kat_vectors: philox4x32-10, counter 0 key 0 and the pi-digits vector."""
    r = O.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = O.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = O.philox_uniform(12345, 7, np.arange(1000))
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 0.05


# ----------------------------------------------------------------------------- G7: beam-profile smearing
def _widths(g, pre):
    return {k: float(g[pre + k]) for k in ("umbra", "penumbra", "variance") if pre + k in g}


@pytest.mark.parametrize("tag,kind,two_d", SMEAR_CASES)
def test_g7_smearing_preparation_and_smeared_intensities(tag, kind, two_d):
    """SmearingConfig.setIntPoints, SASConfig.prepareSmearing and the smeared branch of
    SASModel.calcIntensity (sasconfig.py:122-149,209-233,308-339; sasmodel.py:56-73)."""
    g = load("g7_smearing.npz"); pre = tag + "_"
    q = g[pre + "q"]
    sm = oracle_smearing(kind, two_d, int(g[pre + "n_steps"]), q, **_widths(g, pre))
    np.testing.assert_allclose(sm.q_offset, g[pre + "q_offset"], rtol=1e-15)
    np.testing.assert_allclose(sm.weights, g[pre + "weights"], rtol=1e-14)
    np.testing.assert_allclose(sm.locs, g[pre + "locs"], rtol=1e-15)
    spec = spec_for("sphere", [1e-10], [1e-6]); spec.smear = sm
    for r, it in zip(g[pre + "sphere_radii"], g[pre + "sphere_it"]):
        np.testing.assert_allclose(O.calc_intensity(spec, q, [r], 0.6666666)[0], it, rtol=1e-13)
    np.testing.assert_allclose(O.model_calc(spec, q, g[pre + "sphere_pset"], 0.6666666)[0], g[pre + "sphere_cum"], rtol=1e-13)
    mf, sld = g[pre + "lma_fixed"]
    lspec = spec_for("lmasphere", [1e-10, 0.001], [1e-6, 0.9], mf=float(mf), sld=float(sld)); lspec.smear = sm
    for row, it in zip(g[pre + "lma_params"], g[pre + "lma_it"]):
        np.testing.assert_allclose(O.calc_intensity(lspec, q, row, 0.6666666)[0], it, rtol=1e-12)
    # canSmear = False: the configuration is ignored (sasmodel.py:57)
    gspec = spec_for("gausschain"); gspec.smear = sm
    gspec_plain = spec_for("gausschain")
    row = [gspec.values[i] for i in gspec.active]
    np.testing.assert_array_equal(O.calc_intensity(gspec, q, row, 0.6666666)[0], O.calc_intensity(gspec_plain, q, row, 0.6666666)[0])
    np.testing.assert_allclose(O.calc_intensity(gspec, q, row, 0.6666666)[0], g[pre + "gauss_chain_it"], rtol=1e-12)


@pytest.mark.parametrize("method", ["leastsq", "closed"])
def test_g7_smeared_trajectory(method):
    g, spec, st = traj_setup("g7_sphere_q100_smeared.npz")
    spec.smear, _ = traj_smearing(g)
    res = O.mc_fit(spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"],
                   g["data_x0_limit"], st, O.ReplayStream(g["stream"]), method=method)
    assert res.num_iter == int(g["res_num_iter"])
    np.testing.assert_array_equal(np.array(res.accepted), g["res_accepted"])
    np.testing.assert_allclose(res.rset, g["res_rset"], rtol=1e-15)
    np.testing.assert_allclose(res.conval, float(g["res_conval"]), rtol=1e-12 if method == "leastsq" else 1e-9)
    np.testing.assert_allclose(res.fit, g["res_fit"], rtol=1e-6)


# ----------------------------------------------------------------------------- G8: input preparation
@pytest.mark.parametrize("tag", ["demo", "dense"])
def test_g8_uncertainty_floor_and_rebin(tag):
    """DataObj._prepareUncertainty and DataObj._reBin (dataobj/dataobj.py:204-227, 288-345)."""
    g = load("g8_input_prep.npz")
    fu = O.prepare_uncertainty(g[tag + "_raw_f"], g[tag + "_raw_fu"], float(g[tag + "_fu_min"]))
    np.testing.assert_array_equal(fu, g[tag + "_si_fu"])
    xb, fb, ub = O.rebin(g[tag + "_san_x"], g[tag + "_san_f"], g[tag + "_san_fu"], int(g[tag + "_nbin"]))
    np.testing.assert_array_equal(xb, g[tag + "_bin_x"])
    np.testing.assert_array_equal(fb, g[tag + "_bin_f"])
    np.testing.assert_array_equal(ub, g[tag + "_bin_fu"])
    no_col = O.prepare_uncertainty(g[tag + "_raw_f"], None, 0.05)
    np.testing.assert_array_equal(no_col, 0.05 * g[tag + "_raw_f"])


# ----------------------------------------------------------------------------- the C restatement (oracle/c)
C_TRAJ = ["g4_sphere_q100_fixed.npz", "g4_sphere_q100_converge.npz", "g4_sphere_q512_fixed.npz",
          "g4_sphere_q100_nobg.npz", "g4_sphere_q100_posbg.npz", "g4_sphere_q100_frommin.npz",
          "g14_sphere_q512_long.npz", "g14_sphere_q512_converge.npz", "g17_sphere_q100_posbg_converge.npz",
          # round 5: the cylinder and core-shell-ellipsoid rows (BASELINE configs 3 and 4): the reference's short chains, its chains
          # at the configs' shapes (512 q x 400 x 2000 steps, 1024 q x 1000 x 1500 steps) and the ones it ends by convergence
          "g4_cyl_q40.npz", "g4_ellcs_q40.npz", "g9_cyl_q512.npz", "g9_ellcs_q1024.npz",
          "g17_cyl_q100_converge.npz", "g17_ellcs_q100_converge.npz"]


@pytest.mark.parametrize("name", C_TRAJ)
def test_c_oracle_replays_reference_trajectories(name):
    """oracle/c/mcsas_oracle.c (plain C, libm, Cephes J1) on the uniform stream the reference consumed: the
    reference's accept/reject sequence, parameter set, chi² and fit."""
    from oracle import c_oracle
    g, spec, st = traj_setup(name)
    start = []
    for c in range(spec.n_active):                            # mcsas.py:310-315
        mb = min(spec.lo[c], spec.hi[c]); mb = mb if mb != 0 else np.pi / g["data_x0_limit"][1]
        start.append(0.5 * mb)
    r = c_oracle.analyse(spec, g["data_q"], g["data_I"], g["data_sigma"], st.n_contrib, 1,
                         st.max_iter, st.conv_crit, comp_exp=st.comp_exp, find_bg=st.find_bg, pos_bg=st.pos_bg,
                         start_from_min=st.start_from_min, start_value=start, replay=g["stream"][None, :],
                         want_accepted=int(g["res_num_moves"]) + 4)
    assert r.num_iter[0] == int(g["res_num_iter"]) and r.num_moves[0] == int(g["res_num_moves"])
    np.testing.assert_array_equal(r.accepted[0, :r.num_moves[0]], g["res_accepted"])
    np.testing.assert_allclose(r.contribs[:, :, 0], g["res_rset"], rtol=1e-15 if spec.model_id == O.SPHERE else 1e-13)
    if name == "g17_sphere_q100_posbg_converge.npz":           # (the reference's last fit hit MINPACK's maxfev: see test_g4_replay_trajectories)
        assert float(g["res_conval"]) * (1 - 1e-2) < r.chisq[0] <= float(g["res_conval"]) * (1 + 1e-9)
        return
    np.testing.assert_allclose(r.chisq[0], float(g["res_conval"]), rtol=1e-5 if "posbg" in name else 1e-9)
    np.testing.assert_allclose(r.fit[:, 0], g["res_fit"], rtol=1e-6)


def test_c_oracle_j1_is_the_cephes_j1_scipy_runs():
    """The C oracle's Bessel function (restated from the published Cephes algorithm) against scipy.special.j1 — the function the
    reference calls (cylindersisotropic.py:73,79) — over both ranges of the algorithm and across its seam at 5."""
    from oracle import c_oracle
    from scipy.special import j1
    x = np.concatenate([np.linspace(1e-6, 5.0, 2001), np.linspace(5.0, 60.0, 4001), np.logspace(-8, 4, 1201), [5.0, np.nextafter(5.0, 6.0)]])
    np.testing.assert_allclose(c_oracle.j1(x), j1(x), rtol=0, atol=4e-16)
    np.testing.assert_array_equal(c_oracle.j1(-x[:50]), -c_oracle.j1(x[:50]))


@pytest.mark.parametrize("tag", ["sphere", "cyl_aspect", "cyl_length", "ellcs"])
def test_c_oracle_model_vectors(tag):
    """The C oracle's rows against the reference's own ScatteringModel.calc vectors (G2) and against the numpy restatement on
    seeded parameter sets across the models' ranges."""
    from oracle import c_oracle
    g = load("g12_models.npz")
    spec = spec_for(tag)
    q, pset, c = g[tag + "_q"], g[tag + "_pset"], float(g["comp_exp"])
    np.testing.assert_allclose(c_oracle.model_calc(spec, q, pset, c), g[tag + "_cumInt"], rtol=2e-13)
    for row, it_ref in zip(pset, g[tag + "_rows"]):
        np.testing.assert_allclose(c_oracle.model_calc(spec, q, row[None, :], c), it_ref, rtol=2e-12, atol=1e-300)
    rs = np.random.RandomState(3)
    lo, hi = {"sphere": ([1e-9], [3e-7]), "cyl_aspect": ([1e-9, 0.5], [1e-7, 20.]), "cyl_length": ([1e-9, 1e-9], [1e-7, 1e-6]),
              "ellcs": ([1e-9, 2e-9, 2e-10], [1e-7, 2e-7, 1e-8])}[tag]
    spec2 = spec_for(tag, lo, hi)
    q2 = np.logspace(7, np.log10(3e9), 48)
    sets = np.exp(rs.uniform(np.log(lo), np.log(hi), size=(12, len(lo))))
    for row in sets:
        want = O.calc_intensity(spec2, q2, row, c)[0]
        np.testing.assert_allclose(c_oracle.model_calc(spec2, q2, row[None, :], c), want, rtol=5e-12, atol=1e-14 * want.max())


def test_c_oracle_philox_and_threads_match_numpy_oracle():
    from oracle import c_oracle
    lib = c_oracle.load()
    idx = np.array([0, 1, 2, 3, 1000, 2**33 + 5], dtype=np.uint64)
    ref = O.philox_uniform(20250101, 7, idx)
    got = np.array([lib.mcsas_c_philox_uniform(20250101, 7, int(i)) for i in idx])
    np.testing.assert_array_equal(got, ref)
    g = load("g4_sphere_q100_fixed.npz")
    q, I, sig = g["data_q"], g["data_I"], g["data_sigma"]
    lo, hi = float(g["spec_lo"][0]), float(g["spec_hi"][0])
    spec = spec_for("sphere", [lo], [hi])
    one = c_oracle.analyse_sphere(q, I, sig, lo, hi, 60, 5, 300, 1e-9, seed=11, rep_offset=2, threads=1)
    par = c_oracle.analyse_sphere(q, I, sig, lo, hi, 60, 5, 300, 1e-9, seed=11, rep_offset=2, threads=3)
    np.testing.assert_array_equal(one.contribs, par.contribs)
    ost = O.Settings(n_contrib=60, n_reps=1, max_iter=300, conv_crit=1e-9)
    for r in range(5):
        ref = O.mc_fit(spec, q, I, sig, g["data_f_limit"], g["data_x0_limit"], ost, O.PhiloxStream(11, 2 + r), method="closed")
        assert one.num_moves[r] == ref.num_moves
        np.testing.assert_allclose(one.contribs[:, 0, r], ref.rset[:, 0], rtol=1e-15)
        np.testing.assert_allclose(one.chisq[r], ref.conval, rtol=1e-9)


def test_g15_series_two_data_sets_from_one_stream():
    """The series Calculator's core (gui/calc.py:271-349): calc() on one data set after the other, ONE global random stream.
    The oracle, fed the reference's stream, reproduces each data set's parameter sets and leaves the stream where the
    reference's next calc() picked it up (maxRetries is clipped to its valueRange's minimum of 1: two attempts per repetition)."""
    g = load("g15_series.npz")
    m, spec = make_models("sphere", [float(g["lo"])], [float(g["hi"])])
    ost = O.Settings(n_contrib=60, n_reps=2, max_iter=200, conv_crit=1e-9, max_retries=1, show_incomplete=True)
    stream = O.ReplayStream(g["stream"])
    for i in range(2):
        pre = "d%d_" % i
        assert stream.pos == int(g["starts"][i])
        res, info = O.analyse(spec, g[pre + "q"], g[pre + "I"], g[pre + "sigma"], g[pre + "f_limit"], g[pre + "x0_limit"], ost,
                              stream, method="closed")
        np.testing.assert_allclose(res["contribs"], g[pre + "contribs"], rtol=1e-12)
        assert res["numIter"] == float(g[pre + "numIter"])
        np.testing.assert_allclose(res["scaling"], g[pre + "scaling"], rtol=1e-6)
    assert stream.pos == int(g["starts"][2])
