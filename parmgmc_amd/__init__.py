"""parmgmc_amd -- MI355X (gfx950) implementation of ParMGMC's Gibbs/SOR sweep hot path.

The product is the C-ABI shared library ``libparmgmc_hip.so`` (``include/parmgmc_hip.h``): host code in C,
hand-written HIP kernels for gfx950.  This package only loads it through ctypes and offers thin wrappers for
tests and benchmarks; PyTorch is used by callers for device memory and streams, never for the arithmetic.
There is no CPU fallback: importing :mod:`parmgmc_amd.capi` fails loudly when the library is missing.
"""
from .capi import lib, library_path, check, PMGError  # noqa: F401
from .wrappers import MCSOR, GridMCSOR, CholSampler, MGMC, vec_set_random_standard_normal, autocorrelation, iact, estimate_covariance_errors, make_observation_mats  # noqa: F401
from .capi import SOR_FORWARD_SWEEP, SOR_BACKWARD_SWEEP, SOR_SYMMETRIC_SWEEP, COLORING_GREEDY, COLORING_LEXLEVELS, COLORING_USER, COLORING_ITERATED  # noqa: F401
