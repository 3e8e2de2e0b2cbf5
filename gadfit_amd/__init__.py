"""gadfit_amd -- MI355X-native hot path of gadfit's Levenberg-Marquardt global fit.

Host-side mirror of the reference's fitting-function API (``ad``, ``fitfunction``) and
driver API (``gadfit``); the per-point AD sweep, J^T J / J^T r and chi2 run as HIP kernels
on gfx950 behind the C ABI in include/gadfit_hip.h.
"""
from .ad import advar, Real, INFINITY, integrate  # noqa: F401
from .fitfunction import fitfunc  # noqa: F401
