"""Training objective of the GP quasar model on the GPU.

Mirrors ``objective.m`` / ``spectrum_loss.m``: ``objective(x, centered_rest_fluxes, lya_1pzs,
rest_noise_variances)`` returns ``(f, g)`` for ``x = [vec M; log omega; log c0; log tau0; log beta]``.
:class:`TrainingSet` keeps the three [num_quasars x num_pixels] matrices resident in HBM so an
L-BFGS driver (the reference uses the third-party minFunc, learn_qso_model.m:100-101) pays only for
``x`` and ``g`` per iteration; :func:`fit` is that driver on ``scipy.optimize``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_dp = C.POINTER(C.c_double)


class TrainingSet:
    def __init__(self, centered_rest_fluxes, lya_1pzs, rest_noise_variances, device: int = 0):
        self.lib = _lib.load()
        F = np.asfortranarray(centered_rest_fluxes, dtype=np.float64)
        L = np.asfortranarray(lya_1pzs, dtype=np.float64)
        N = np.asfortranarray(rest_noise_variances, dtype=np.float64)
        if not (F.shape == L.shape == N.shape and F.ndim == 2):
            raise _lib.GpdlaError(-1, "the three training matrices must share one [num_quasars, num_pixels] shape")
        self.num_quasars, self.num_pixels = F.shape
        self._h = C.c_void_p()
        _lib.check(self.lib.gpdla_training_create(int(device), self.num_quasars, self.num_pixels,
                                                  F.ctypes.data_as(_dp), L.ctypes.data_as(_dp),
                                                  N.ctypes.data_as(_dp), C.byref(self._h)))

    def objective(self, x):
        """(f, g) of objective.m:12-75 at x."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        k, rem = divmod(x.size - 3, self.num_pixels)
        k -= 1
        if rem or k < 1:
            raise _lib.GpdlaError(-1, f"x has {x.size} entries, expected num_pixels*(k+1)+3")
        f = C.c_double()
        g = np.empty_like(x)
        _lib.check(self.lib.gpdla_training_objective(self._h, x.ctypes.data_as(_dp), int(k),
                                                     C.byref(f), g.ctypes.data_as(_dp)))
        return f.value, g

    def close(self):
        if self._h:
            self.lib.gpdla_training_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def objective(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, device: int = 0):
    """``[f, g] = objective(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances)`` (objective.m:12)."""
    t = TrainingSet(centered_rest_fluxes, lya_1pzs, rest_noise_variances, device)
    try:
        return t.objective(x)
    finally:
        t.close()


def fit(initial_x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, max_iter: int = 2000,
        max_fun_evals: int = 4000, device: int = 0):
    """learn_qso_model.m:100-101 with scipy's L-BFGS-B in place of minFunc (set_parameters.m:43-45
    gives MaxIter 2000, MaxFunEvals 4000).  Returns (x, f, scipy result)."""
    from scipy.optimize import minimize
    t = TrainingSet(centered_rest_fluxes, lya_1pzs, rest_noise_variances, device)
    try:
        res = minimize(t.objective, np.asarray(initial_x, dtype=np.float64), jac=True, method="L-BFGS-B",
                       options=dict(maxiter=max_iter, maxfun=max_fun_evals))
    finally:
        t.close()
    return res.x, res.fun, res
