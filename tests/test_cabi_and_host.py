"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/mcsas_hip.h
declares, the ctypes structs match the C layout, and the host-side mirror of the reference's
plugin API flattens models the way the kernels expect."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import mcsas_amd
from mcsas_amd import _lib, engine
from mcsas_amd.scatteringmodels import setup_from_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcsas_hip.h")


def test_library_exports_every_declared_symbol():
    lib = _lib.load()                                   # raises if the .so is missing: no fallback
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(mcsas_hip_[a-z_0-9]+)\s*\(", text))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mcsas_hip_abi_version() == _lib.ABI_VERSION
    assert lib.mcsas_hip_device_count() >= 0            # 0 here: no GPU in the build container


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\n'
                   'int main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(mcsas_problem), sizeof(mcsas_result),'
                   ' offsetof(mcsas_problem, n_contrib), offsetof(mcsas_problem, seed), offsetof(mcsas_problem, stop),'
                   ' offsetof(mcsas_result, draws)); return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.Problem) and out[1] == C.sizeof(_lib.Result)
    assert out[2] == _lib.Problem.n_contrib.offset and out[3] == _lib.Problem.seed.offset
    assert out[4] == _lib.Problem.stop.offset and out[5] == _lib.Result.draws.offset


def test_errors_without_gpu_are_loud():
    """No device in this container: the product path reports MCSAS_ENODEV, it does not fall back."""
    if _lib.load().mcsas_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    m = mcsas_amd.Sphere()
    with pytest.raises(_lib.McSASHipError) as e:
        engine.analyse(m.setup(), np.logspace(7, 9, 16), np.ones(16), np.ones(16),
                       engine.Settings(n_contrib=4, n_reps=1, max_iter=1))
    assert e.value.code == -2


def test_model_flattening_follows_reference_rules():
    m = mcsas_amd.CylindersIsotropic()
    m.radius.setActive(True); m.aspect.setActive(True)
    m.radius.setActiveRange((1e-12, 5e-8))            # below valueRange min 0.1 nm -> clipped (utils/parameter.py:615-624)
    m.aspect.setActiveRange((20.0, 0.5))              # reversed -> (min, max)
    s = m.setup()
    assert s.model_id == engine.MODEL_CYL_ISO
    assert s.active_index == (0, 3)                    # radius, aspect in `parameters` order
    np.testing.assert_allclose(s.gen_lo, [1e-10, 0.5]); np.testing.assert_allclose(s.gen_hi, [5e-8, 20.0])
    assert s.gen_kind == (1, 1)                        # RandomExponential
    np.testing.assert_allclose(s.clip_lo, [1e-10, 1e-3])
    assert s.params[1] == 1.0 and s.params[4] == 100.0
    m.useAspect.setValue(False)
    assert m.setup().params[1] == 0.0
    k = mcsas_amd.Kholodenko().setup()
    assert k.active_index == (0, 1, 2) and k.gen_kind == (1, 0, 0)
    # startFromMinimum fill: min(activeRange)/2, or pi/qmax/2 when the range starts at 0 (mcsas.py:311-315)
    sp = mcsas_amd.Sphere(); sp.radius.setActiveRange((0.0, 1e-7))
    class D: pass
    d = D(); d.x0 = D(); d.x0.limit = [1e7, 2e9]
    np.testing.assert_allclose(setup_from_model(sp, d).start_value, [np.pi / 2e9 * .5])
    class Foreign: pass
    with pytest.raises(NotImplementedError):
        setup_from_model(Foreign())


def test_parameter_api_mirrors_reference_semantics():
    p = mcsas_amd.Sphere().radius
    p.setValue(-1.0)
    assert p() == 0.0                                   # clipped into valueRange (parameter.py:405-414)
    p.setActive(False); p.setActiveVal(np.ones(3), index=0)
    assert p.activeValues() == []                       # inactive parameters ignore setActiveVal
    p.setActive(True); p.setActiveVal(np.ones(3), index=2)
    assert len(p.activeValues()) == 3 and p.activeValues()[0] is None
    np.random.seed(3)
    u = np.random.uniform(size=5)
    np.random.seed(3)
    p.setActiveRange((2e-9, 4e-9))
    np.testing.assert_allclose(p.generate(count=5), u * 2e-9 + 2e-9)
    algo = mcsas_amd.McSAS.factory()()
    assert algo.numContribs() == 300 and algo.compensationExponent() == 0.6666666 and algo.maxRetries() == 5
    algo.maxRetries.setValue(500)
    assert algo.maxRetries() == 100                     # mcsasparameters.json valueRange [1, 100]
    algo.stop = True
    assert algo.stop is True


def test_sasdata_from_reference_csv(golden_dir):
    d = mcsas_amd.SASData.fromCsv(os.path.join(golden_dir, "ref_testdata", "quickstartdemo1.csv"))
    assert d.count == 101 and abs(d.q[0] - 1e7) < 1 and d.f.binnedDataU[0] == 1.89e8
    np.testing.assert_allclose(d.sphericalSizeEst(), [np.pi / 1e9, np.pi / 1e7])


@pytest.mark.parametrize("tag,kind,two_d", [("trapz_slit", "trapezoid", False), ("trapz_pinhole", "trapezoid", True),
                                            ("gauss_slit", "gaussian", False)])
def test_smearing_config_mirror_prepares_what_the_reference_prepares(tag, kind, two_d, golden_dir):
    """data.config.smearing / data.locs of the host mirror (mcsas_amd/dataobj.py) against the
    reference's SmearingConfig.setIntPoints + SASConfig.prepareSmearing (fixture g7_smearing.npz)."""
    from helpers import product_smearing
    g = np.load(os.path.join(golden_dir, "g7_smearing.npz")); pre = tag + "_"
    widths = {k: float(g[pre + k]) for k in ("umbra", "penumbra", "variance") if pre + k in g}
    d, args = product_smearing(kind, two_d, int(g[pre + "n_steps"]), g[pre + "q"], **widths)
    np.testing.assert_allclose(args.q_offset, g[pre + "q_offset"], rtol=1e-15)
    np.testing.assert_allclose(args.weights, g[pre + "weights"], rtol=1e-14)
    np.testing.assert_allclose(args.locs, g[pre + "locs"], rtol=1e-15)
    np.testing.assert_array_equal(d.locs, args.locs)
    # models that cannot smear, a disabled or an invalid configuration: the plain branch
    assert d.smearArgs(mcsas_amd.GaussianChain()) is None
    d.config.smearing.doSmear.setValue(False)
    assert d.smearArgs(mcsas_amd.Sphere()) is None
    d.updateConfig()
    np.testing.assert_array_equal(d.locs, g[pre + "q"])
    plain = mcsas_amd.SASData(g[pre + "q"], np.ones(len(g[pre + "q"])), np.ones(len(g[pre + "q"])))
    plain.config.smearing.doSmear.setValue(True)            # umbra = penumbra = lower limit: not valid
    assert not plain.config.smearing.inputValid() and plain.smearArgs(mcsas_amd.Sphere()) is None


def test_smearing_value_ranges_follow_the_data():
    """updateSmearingLimits / onUmbraUpdate (sasconfig.py:169-182): widths are clipped into
    [min |dq|, 2 q_max] and penumbra cannot go below umbra."""
    q = np.array([1e7, 1.5e7, 3e7, 1e8])
    d = mcsas_amd.SASData(q, np.ones(4), np.ones(4))
    sm = d.config.smearing
    sm.umbra.setValue(1.0)
    assert sm.umbra() == 5e6
    sm.penumbra.setValue(1e12)
    assert sm.penumbra() == 2e8
    sm.umbra.setValue(6e7)
    sm.penumbra.setValue(1e7)
    assert sm.penumbra() == 6e7 and not sm.inputValid()


def test_histogram_host_binning_matches_reference(golden_dir):
    """mcsas_amd.Histogram.calc / Moments (the vectorised host half of McSAS.histogram(), utils/parameter.py:
    20-122, 349-479) on fractions from the oracle, against the reference's bins, CDF, observability and
    moments (fixture G5)."""
    from oracle import mcsas_oracle as O
    from helpers import make_models
    g = np.load(os.path.join(golden_dir, "g45_analyse.npz"))
    lo, hi = float(g["A_lo"]), float(g["A_hi"])
    m, spec = make_models("sphere", [lo], [hi])
    st = O.Settings(n_contrib=150, n_reps=3, max_iter=100000, conv_crit=5.0)
    frac, _ = O.fractions(spec, g["data_q"], g["data_I"], g["data_sigma"], g["data_f_limit"], st, g["A_contribs"], method="leastsq")
    for hi_, (bc, xlog, yw) in enumerate(g["A_h_spec"]):
        h = mcsas_amd.Histogram(m.radius, lo, hi, binCount=int(bc), xscale="log" if xlog else "lin", yweight=O.YWEIGHTS[int(yw)])
        h.calc(g["A_contribs"], 0, frac)
        p = "A_h%d_" % hi_
        np.testing.assert_allclose(h.xLowerEdge, g[p + "edges"], rtol=1e-15)
        np.testing.assert_allclose(h.bins.full, g[p + "bins_full"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(h.bins.mean, g[p + "bins_mean"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(h.bins.std, g[p + "bins_std"], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(h.cdf.mean, g[p + "cdf_mean"], rtol=1e-9)
        np.testing.assert_allclose(h.observability, g[p + "obs"], rtol=1e-9)
        np.testing.assert_allclose(np.array(h.moments.fields)[0::2], g[p + "moments"][0::2], rtol=1e-9)


@pytest.mark.parametrize("case", ["ordinary", "zero_weight", "one_rep_zero", "empty_range", "one_member"])
def test_moments_edge_cases_match_the_reference(golden_dir, case):
    """Moments (utils/parameter.py:84-122) where a weighting sums to zero, a repetition has nothing in range or a single member
    (fixture G19, oracle/make_golden.py gen_moments_edge_cases, from the reference's own class): only an EXACT zero of
    sum(frac) * sigma is skipped — an all-zero weighting gives NaN variance AND NaN skew / kurtosis, as in the reference."""
    from mcsas_amd.parameter import Moments
    g = np.load(os.path.join(golden_dir, "g19_moments_edge.npz"))
    contribs = g["contribs_one_member"] if case == "one_member" else g["contribs"]
    with np.errstate(all="ignore"):
        m = Moments(contribs, 0, tuple(g[case + "_range"]), g[case + "_fraction"])
    np.testing.assert_allclose(np.array(m.fields, dtype=float), g[case + "_fields"], rtol=1e-9, atol=1e-300, equal_nan=True)
    assert np.isnan(g["zero_weight_fields"][4:]).all()          # (what the fixture pins)


def test_shard_rule_of_the_library_is_the_one_of_dist():
    """mcsas_hip_shard (how mcsas_hip_analyse splits repetitions over a device list) == mcsas_amd.dist.shard_reps (how
    bench.py's ranks split them): contiguous blocks in order, sizes within one of each other, empty blocks allowed."""
    from mcsas_amd import dist, engine
    for n in (0, 1, 5, 13, 50, 400):
        for g in (1, 2, 3, 8, 16):
            blocks = [engine.shard(n, g, i) for i in range(g)]
            assert blocks == [dist.shard_reps(n, g, i) for i in range(g)]
            assert sum(c for _, c in blocks) == n and all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(g - 1))


def test_release_library_refuses_a_nonzero_reserved_word():
    """mcsas_problem.reserved0 "must be 0": the tuning / ablation variants of the pipeline mode exist in the measurement
    build only (libmcsas_hip_tuning.so, -DMCSAS_TUNING); the release library answers MCSAS_EINVAL before it touches a
    device — a caller that fails to zero the field gets an error, not a silently different fit."""
    import ctypes as C
    from mcsas_amd import _lib, engine
    import mcsas_amd
    lib = _lib.load()
    assert lib.mcsas_hip_is_tuning_build() == 0
    q = np.logspace(7, 9, 64)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((1e-9, 1e-7))
    st = engine.Settings(n_contrib=40, n_reps=2, max_iter=10)
    prob = engine.HipProblem(m.setup(), q, np.ones(64), np.ones(64), st)
    prob.c.reserved0 = 1 << 16
    h = C.c_void_p()
    rc = lib.mcsas_hip_plan_create(C.byref(prob.c), C.byref(h))
    assert rc == -1 and b"reserved0" in lib.mcsas_hip_last_error()
    res = engine.ChainResults(40, 1, 2, 64)
    assert lib.mcsas_hip_analyse(C.byref(prob.c), C.byref(res.c)) == -1
    tl = _lib.load(tuning=True)
    assert tl.mcsas_hip_is_tuning_build() == 1 and tl is not lib


def test_model_plugins_compile_without_a_gpu_and_report_compiler_errors():
    """mcsas_hip_plugin_compile: hiprtc builds the plug-in text against the kernel headers the library carries — no GPU, no source
    tree; the same text gives the same id; a text that does not compile is refused with the compiler's message."""
    from helpers import PLUGIN_SOURCES, make_models, plugin_twin
    a = engine.compile_plugin(PLUGIN_SOURCES["gausschain"])
    b = engine.compile_plugin(PLUGIN_SOURCES["sphcs"])
    assert a >= engine.MODEL_PLUGIN0 and b >= engine.MODEL_PLUGIN0 and a != b
    assert engine.compile_plugin(PLUGIN_SOURCES["gausschain"]) == a
    with pytest.raises(engine.PluginCompileError) as e:
        engine.compile_plugin(PLUGIN_SOURCES["gausschain"].replace("p[3] * (p[0] * p[0])", "p[3] * rg_squared"))
    assert "rg_squared" in e.value.log and "plugin:" in e.value.log          # (line numbers refer to the plug-in text)
    # the row class / canSmear flag: any spelling the preprocessor accepts on a `#define` line is read the same way by the host,
    # and a text where host and preprocessor would disagree (the define sits in a dead `#if 0` block) is refused
    base = PLUGIN_SOURCES["gausschain"]
    for spelled in ("#define MCSAS_PLUGIN_ROW_CLASS (1)\n", "#  define\tMCSAS_PLUGIN_ROW_CLASS\t1\n", "  # define MCSAS_PLUGIN_ROW_CLASS 0x1\n",
                    "// #define MCSAS_PLUGIN_ROW_CLASS 1 (comment: not a define)\n#define MCSAS_PLUGIN_CAN_SMEAR 1\n"):
        assert engine.compile_plugin(spelled + base) >= engine.MODEL_PLUGIN0
    for dead in ("#if 0\n#define MCSAS_PLUGIN_ROW_CLASS 1\n#endif\n", "/*\n#define MCSAS_PLUGIN_CAN_SMEAR 1\n*/\n",
                 "#define MCSAS_PLUGIN_ROW_CLASS 1\n#undef MCSAS_PLUGIN_ROW_CLASS\n#define MCSAS_PLUGIN_ROW_CLASS (2 - 2)\n"):
        with pytest.raises(engine.PluginCompileError) as e:
            engine.compile_plugin(dead + base)
        assert "host side read another value" in e.value.log
    # host mirror: a model class of the user's own with a `hipSource` attribute flattens like a built-in one
    m, _ = make_models("gausschain")
    s0 = m.setup()
    s1 = plugin_twin(m, "gausschain").setup()
    assert s1.model_id == a and s0.model_id == engine.MODEL_GAUSS_CHAIN
    assert np.array_equal(s0.params, s1.params) and s0.active_index == s1.active_index
    # ... and one without it is still a loud error
    class Nameless(mcsas_amd.scatteringmodels.SASModel):
        parameters = ()
    with pytest.raises(NotImplementedError):
        Nameless().setup()


def test_committed_counter_profiles_describe_the_committed_kernels():
    """bench.py quotes per-step counters (instructions, memory-side bytes) from the newest profiles/rNN_*.json; tools/pmc_summary.py
    stores a hash of the kernel sources they were taken on, and the bench line reports whether it matches the tree.  Committed state:
    it matches — a kernel change without a profile refresh fails here (and shows as `profile_taken_on_these_kernel_sources: false`)."""
    import bench
    for suffix in ("valu_per_step.json", "pmc_traffic.json"):
        name, prof = bench.latest_profile(suffix)
        assert name and prof, suffix
        for key, entry in prof.items():
            assert entry.get("csrc_sha16") == bench.kernel_sources_sha16(), (name, key)
