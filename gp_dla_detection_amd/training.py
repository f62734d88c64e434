"""Training objective of the GP quasar model on the GPU.

Mirrors ``objective.m`` / ``spectrum_loss.m``: ``objective(x, centered_rest_fluxes, lya_1pzs,
rest_noise_variances)`` returns ``(f, g)`` for ``x = [vec M; log omega; log c0; log tau0; log beta]``;
``objective_lyseries`` is the mean-flux model's form (multi_dlas/objective_lyseries.m).
:class:`TrainingSet` keeps the three [num_quasars x num_pixels] matrices resident in HBM so an
L-BFGS driver (the reference uses the third-party minFunc, learn_qso_model.m:100-101) pays only for
``x`` and ``g`` per iteration; :func:`fit` is that driver, on :func:`minimize_lbfgs` (minFunc's
default L-BFGS restated; numpy only).
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

    def set_lyseries(self, num_forest_lines: int, all_transition_wavelengths=None, all_oscillator_strengths=None):
        """Switch to the mean-flux model's objective (multi_dlas/objective_lyseries.m over
        spectrum_loss_lyseries.m): the optical depth sums the first ``num_forest_lines`` Lyman lines,
        each counted where its redshift does not exceed the quasar's.  The two tables default to
        set_parameters_multi.m:76-143; ``num_forest_lines <= 1`` switches back to objective.m."""
        wl = fs = None
        if all_transition_wavelengths is not None or all_oscillator_strengths is not None:
            wl = np.ascontiguousarray(all_transition_wavelengths, dtype=np.float64).reshape(-1)
            fs = np.ascontiguousarray(all_oscillator_strengths, dtype=np.float64).reshape(-1)
            if wl.size < num_forest_lines or fs.size < num_forest_lines:
                raise _lib.GpdlaError(-1, f"{num_forest_lines} lines asked for, tables hold {wl.size} / {fs.size}")
        _lib.check(self.lib.gpdla_training_set_lyseries(self._h, int(num_forest_lines),
                                                        None if wl is None else wl.ctypes.data_as(_dp),
                                                        None if fs is None else fs.ctypes.data_as(_dp)))

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


def objective_lyseries(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, num_forest_lines,
                       all_transition_wavelengths=None, all_oscillator_strengths=None, device: int = 0):
    """``[f, g] = objective_lyseries(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances,
    num_forest_lines, all_transition_wavelengths, all_oscillator_strengths)``
    (multi_dlas/objective_lyseries.m:12-14)."""
    t = TrainingSet(centered_rest_fluxes, lya_1pzs, rest_noise_variances, device)
    try:
        t.set_lyseries(num_forest_lines, all_transition_wavelengths, all_oscillator_strengths)
        return t.objective(x)
    finally:
        t.close()


class FitResult:
    """What minFunc's fourth output carries (iterations, funcCount, firstorderopt, message, trace)."""

    def __init__(self):
        self.x = None
        self.fun = np.nan
        self.nit = 0
        self.nfev = 0
        self.firstorderopt = np.nan
        self.message = ""
        self.trace_fval = []
        self.trace_nfev = []

    def __repr__(self):
        return (f"FitResult(fun={self.fun!r}, nit={self.nit}, nfev={self.nfev}, "
                f"firstorderopt={self.firstorderopt:.3e}, message={self.message!r})")


def _cubic_min(x1, f1, g1, x2, f2, g2, lo, hi):
    """Minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clamped to [lo, hi]; the midpoint
    when the cubic has no real minimiser (minFunc's polyinterp for two points with derivatives)."""
    if x1 > x2:
        x1, f1, g1, x2, f2, g2 = x2, f2, g2, x1, f1, g1
    d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
    disc = d1 * d1 - g1 * g2
    if np.isfinite(disc) and disc >= 0.0:
        d2 = np.sqrt(disc)
        den = g2 - g1 + 2.0 * d2
        if den != 0.0:
            t = x2 - (x2 - x1) * ((g2 + d2 - d1) / den)
            if np.isfinite(t):
                return min(max(t, lo), hi)
    return 0.5 * (lo + hi)


def _wolfe_line_search(fun, x, t, d, f, g, gtd, c1, c2, max_ls, prog_tol, budget):
    """Strong-Wolfe bracketing line search with cubic interpolation (minFunc's WolfeLineSearch with
    LS_interp = 2, its default for L-BFGS; Nocedal & Wright alg. 3.5/3.6).  Returns
    (t, f_new, g_new, evaluations)."""
    evals = 0

    def phi(step):
        nonlocal evals
        evals += 1
        fn, gn = fun(x + step * d)
        return fn, gn, float(gn @ d)

    f_new, g_new, gtd_new = phi(t)
    t_prev, f_prev, g_prev, gtd_prev = 0.0, f, g, gtd
    bracket = None
    it = 0
    while it < max_ls and evals < budget:
        if not (np.isfinite(f_new) and np.isfinite(gtd_new)):
            # stepped out of the objective's domain (B not positive definite gives nan): halve
            t = 0.5 * (t_prev + t)
            f_new, g_new, gtd_new = phi(t)
            it += 1
            continue
        if f_new > f + c1 * t * gtd or (it > 0 and f_new >= f_prev):
            bracket = [(t_prev, f_prev, g_prev, gtd_prev), (t, f_new, g_new, gtd_new)]
            break
        if abs(gtd_new) <= -c2 * gtd:
            return t, f_new, g_new, evals
        if gtd_new >= 0.0:
            bracket = [(t_prev, f_prev, g_prev, gtd_prev), (t, f_new, g_new, gtd_new)]
            break
        lo, hi = t + 0.01 * (t - t_prev), t * 10.0
        t_next = _cubic_min(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, lo, hi)
        t_prev, f_prev, g_prev, gtd_prev = t, f_new, g_new, gtd_new
        t = t_next
        f_new, g_new, gtd_new = phi(t)
        it += 1
    if bracket is None:
        return t, f_new, g_new, evals
    insufficient = False
    while it < max_ls and evals < budget:
        (ta, fa, ga, gtda), (tb, fb, gb, gtdb) = bracket
        lo_i = 0 if fa <= fb else 1
        lo_pt, hi_pt = bracket[lo_i], bracket[1 - lo_i]
        t = _cubic_min(ta, fa, gtda, tb, fb, gtdb, min(ta, tb), max(ta, tb))
        span = max(ta, tb) - min(ta, tb)
        # keep the trial point away from the ends of the bracket (minFunc's 10 % rule)
        if min(max(ta, tb) - t, t - min(ta, tb)) / span < 0.1:
            if insufficient or t >= max(ta, tb) or t <= min(ta, tb):
                t = max(ta, tb) - 0.1 * span if abs(t - max(ta, tb)) < abs(t - min(ta, tb)) else min(ta, tb) + 0.1 * span
                insufficient = False
            else:
                insufficient = True
        else:
            insufficient = False
        f_new, g_new, gtd_new = phi(t)
        it += 1
        if not np.isfinite(f_new) or f_new > f + c1 * t * gtd or f_new >= lo_pt[1]:
            bracket[1 - lo_i] = (t, f_new, g_new, gtd_new)  # replaces the high point
        else:
            if abs(gtd_new) <= -c2 * gtd:
                return t, f_new, g_new, evals
            if gtd_new * (hi_pt[0] - lo_pt[0]) >= 0.0:
                bracket[1 - lo_i] = lo_pt
            bracket[lo_i] = (t, f_new, g_new, gtd_new)
        (ta, fa, ga, gtda), (tb, fb, gb, gtdb) = bracket
        if abs(ta - tb) * max(abs(gtda), abs(gtdb)) < prog_tol:
            break
    best = min((pt for pt in bracket if np.isfinite(pt[1])), key=lambda pt: pt[1])
    return best[0], best[1], best[2], evals


def minimize_lbfgs(fun, x0, max_iter: int = 2000, max_fun_evals: int = 4000, corrections: int = 100,
                   opt_tol: float = 1e-5, prog_tol: float = 1e-9, c1: float = 1e-4, c2: float = 0.9,
                   max_line_search: int = 25, callback=None) -> FitResult:
    """L-BFGS as the reference's optimiser runs it.  learn_qso_model.m:100-101 calls the third-party
    minFunc (M. Schmidt, 2012 release; not in the reference tree) with only MaxIter = 2000 and
    MaxFunEvals = 4000 set (set_parameters.m:43-45), so everything else is minFunc's default for
    Method 'lbfgs', restated here from its documentation: 100 corrections, initial Hessian scaling
    y's / y'y, curvature pairs skipped when y's <= 1e-10, first step min(1, 1/sum|g|) and 1
    thereafter, strong-Wolfe line search (c1 1e-4, c2 0.9, cubic interpolation, at most 25
    evaluations), and its four stopping tests: max|g| <= optTol (1e-5), max|t d| <= progTol (1e-9),
    |f - f_old| <= progTol, and the two budgets.  ``fun(x) -> (f, g)``."""
    x = np.array(x0, dtype=np.float64)
    res = FitResult()
    f, g = fun(x)
    g = np.asarray(g, dtype=np.float64)
    nfev = 1
    res.trace_fval.append(f)
    res.trace_nfev.append(nfev)
    S, Y, rho = [], [], []
    gamma = 1.0
    message = "Reached Maximum Number of Iterations"
    if np.abs(g).max() <= opt_tol:
        message = "Optimality Condition below optTol"
        max_iter = 0
    it = 0
    d = None
    while it < max_iter:
        if it == 0:
            d = -g
        else:
            # two-loop recursion over the stored (s, y) pairs
            q = g.copy()
            alpha = [0.0] * len(S)
            for i in range(len(S) - 1, -1, -1):
                alpha[i] = rho[i] * float(S[i] @ q)
                q -= alpha[i] * Y[i]
            q *= gamma
            for i in range(len(S)):
                beta = rho[i] * float(Y[i] @ q)
                q += (alpha[i] - beta) * S[i]
            d = -q
        gtd = float(g @ d)
        if not np.isfinite(gtd):
            message = "Search direction is not finite"
            break
        if gtd > -prog_tol:
            message = "Directional Derivative below progTol"
            break
        t = min(1.0, 1.0 / np.abs(g).sum()) if it == 0 else 1.0
        f_old, g_old = f, g
        t, f, g, used = _wolfe_line_search(fun, x, t, d, f_old, g_old, gtd, c1, c2, max_line_search,
                                           prog_tol, max_fun_evals - nfev)
        nfev += used
        if not np.isfinite(f) or f > f_old:
            f, g = f_old, g_old  # the line search found nothing better: keep the iterate
            message = "Line search failed to decrease the objective"
            break
        g = np.asarray(g, dtype=np.float64)
        step = t * d
        x = x + step
        it += 1
        y = g - g_old
        ys = float(y @ step)
        if ys > 1e-10:
            if len(S) == corrections:
                S.pop(0), Y.pop(0), rho.pop(0)
            S.append(step)
            Y.append(y)
            rho.append(1.0 / ys)
            gamma = ys / float(y @ y)
        res.trace_fval.append(f)
        res.trace_nfev.append(nfev)
        if callback is not None:
            callback(x, f, g)
        if np.abs(g).max() <= opt_tol:
            message = "Optimality Condition below optTol"
            break
        if np.abs(step).max() <= prog_tol:
            message = "Step Size below progTol"
            break
        if abs(f - f_old) < prog_tol:
            message = "Function Value changing by less than progTol"
            break
        if nfev >= max_fun_evals:
            message = "Reached Maximum Number of Function Evaluations"
            break
    res.x, res.fun, res.nit, res.nfev = x, f, it, nfev
    res.firstorderopt = float(np.abs(g).max())
    res.message = message
    return res


# Kim et al. (2007) priors of objective.m:59-71
_TAU_0_MU, _TAU_0_SIGMA, _BETA_MU, _BETA_SIGMA = 0.0023, 0.0007, 3.65, 0.21


def prior_value(x):
    """The value whose gradient objective.m:59-71 adds to g.  The reference adds the two Gaussian
    priors on tau_0 and beta to the GRADIENT only -- its f carries no prior term, so (f, g) are not
    a consistent pair and a line search along a prior-dominated direction cannot succeed.
    ``fit(..., prior_in_value=True)`` adds this to f; the default keeps the reference's f."""
    tau_0, beta = np.exp(x[-2]), np.exp(x[-1])
    return 0.5 * ((tau_0 - _TAU_0_MU) / _TAU_0_SIGMA) ** 2 + 0.5 * ((beta - _BETA_MU) / _BETA_SIGMA) ** 2


def fit(initial_x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, max_iter: int = 2000,
        max_fun_evals: int = 4000, device: int = 0, prior_in_value: bool = False,
        num_forest_lines: int = 0, all_transition_wavelengths=None, all_oscillator_strengths=None):
    """``[x, log_likelihood, ~, minFunc_output] = minFunc(objective_function, initial_x,
    minFunc_options)`` of learn_qso_model.m:100-101 with the objective evaluated on the GPU
    (set_parameters.m:43-45: MaxIter 2000, MaxFunEvals 4000).  Returns (x, f, FitResult); f is the
    reference's objective value (without the prior term) in either mode.  ``num_forest_lines > 1``:
    the mean-flux model's objective (multi_dlas/learn_qso_model_meanflux.m:140-147)."""
    t = TrainingSet(centered_rest_fluxes, lya_1pzs, rest_noise_variances, device)
    if num_forest_lines > 1:
        try:
            t.set_lyseries(num_forest_lines, all_transition_wavelengths, all_oscillator_strengths)
        except BaseException:
            t.close()
            raise

    def safe(x):
        try:
            f, g = t.objective(x)
        except _lib.GpdlaError as e:
            # a trial step that leaves the domain (B not positive definite) is an infinite value to
            # the line search, as a nan/inf from chol would be to minFunc
            if e.code == _lib.ERR_NOT_POSITIVE_DEFINITE:
                return np.inf, np.full(x.size, np.nan)
            raise
        return (f + prior_value(x) if prior_in_value else f), g

    try:
        res = minimize_lbfgs(safe, np.asarray(initial_x, dtype=np.float64), max_iter=max_iter,
                             max_fun_evals=max_fun_evals)
        if prior_in_value:
            res.fun -= prior_value(res.x)
    finally:
        t.close()
    return res.x, res.fun, res
