"""mcsas_amd — MI355X-native Monte-Carlo core for McSAS (BAMresearch/McSAS hot path).

Host-side mirror of the reference's plugin/operator interface for ONE path — McSAS.analyse /
McSAS.mcFit and what it calls — over hand-written HIP kernels for gfx950 behind a C ABI
(include/mcsas_hip.h).  Importing the package does not need a GPU; running anything does, and
fails loudly if libmcsas_hip.so is not built (there is no CPU fallback).
"""
from .engine import (ModelSetup, Settings, analyse, model_calc, bgfit, observability, Plan,     # noqa: F401
                     device_count, GEN_UNIFORM, GEN_EXP1, GEN_EXP2, GEN_EXP3)
from .parameter import (Parameter, FitParameter, RandomUniform, RandomExponential,               # noqa: F401
                        RandomExponential1, RandomExponential2, RandomExponential3, Histogram)
from .scatteringmodels import (ScatteringModel, SASModel, SASModelData, Sphere,                   # noqa: F401
                               CylindersIsotropic, EllipsoidalCoreShell, Kholodenko, EllipsoidsIsotropic,
                               SphericalCoreShell, GaussianChain, LMADenseSphere, CylindersRadiallyIsotropic, setup_from_model)
from .dataobj import SASData, SASConfig, TrapezoidSmearing, GaussianSmearing, SmearArgs                                                                      # noqa: F401
from .series import run_series                                                                    # noqa: F401
from .mcsas import McSAS                                                                          # noqa: F401

__version__ = "0.1.0"
