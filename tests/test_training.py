"""Training objective (objective.m / spectrum_loss.m, "next" row N3).

CPU: the oracle's gradient against central finite differences of its own value (the priors of
objective.m:59-71 enter the gradient only, exactly as in the reference).  GPU: value and gradient
against the oracle, and a short L-BFGS run that must decrease the objective."""
import numpy as np
import pytest


def training_problem(nq=40, G=64, k=5, seed=0, missing=0.1):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((G, k)) * 0.3 * 0.8 ** np.arange(k)
    lo = rng.uniform(-3, -2, G)
    x = np.concatenate([M.ravel(order="F"), lo, [np.log(0.1), np.log(0.0023), np.log(3.65)]])
    z = rng.standard_normal((nq, k))
    L1 = 1 + rng.uniform(1.5, 3.0, (nq, G))
    NV = 10 ** rng.uniform(-3, -1, (nq, G))
    F = z @ M.T + np.sqrt(NV) * rng.standard_normal((nq, G))
    F[rng.uniform(size=F.shape) < missing] = np.nan
    F[3] = np.nan  # a quasar with no valid pixel contributes nothing
    return x, F, L1, NV


def test_oracle_gradient_vs_finite_differences(oracle):
    x, F, L1, NV = training_problem(nq=12, G=40, k=4)
    f, g = oracle.objective(x, F, L1, NV)
    t0, b0 = np.exp(x[-2]), np.exp(x[-1])
    prior = np.zeros(x.size)
    prior[-2] = t0 * (t0 - 0.0023) / 0.0007 ** 2
    prior[-1] = b0 * (b0 - 3.65) / 0.21 ** 2
    rng = np.random.default_rng(1)
    for i in list(rng.choice(x.size - 3, 8, replace=False)) + [x.size - 3, x.size - 2, x.size - 1]:
        e = np.zeros(x.size)
        e[i] = 1e-6
        fd = (oracle.objective(x + e, F, L1, NV)[0] - oracle.objective(x - e, F, L1, NV)[0]) / 2e-6
        assert abs(g[i] - prior[i] - fd) < 1e-6 * max(1.0, abs(fd)), i


def test_oracle_threads_agree(oracle):
    x, F, L1, NV = training_problem()
    f1, g1 = oracle.objective(x, F, L1, NV, num_threads=1)
    f4, g4 = oracle.objective(x, F, L1, NV, num_threads=4)
    assert abs(f1 - f4) < 1e-9 * abs(f1)
    np.testing.assert_allclose(g1, g4, rtol=1e-10, atol=1e-9)


@pytest.mark.gpu
def test_gpu_objective_matches_oracle(oracle):
    from gp_dla_detection_amd import training
    # (40 pixels: shorter than one staged chunk; 300: a ragged last chunk; k = 33, 40: several
    # entries of B per thread)
    for (nq, G, k) in ((40, 64, 5), (64, 1217, 20), (24, 300, 33), (16, 513, 40), (8, 40, 3)):
        x, F, L1, NV = training_problem(nq=nq, G=G, k=k, seed=k)
        f_ref, g_ref = oracle.objective(x, F, L1, NV)
        f, g = training.objective(x, F, L1, NV)
        assert abs(f - f_ref) < 1e-9 * abs(f_ref), (f, f_ref)
        scale = np.abs(g_ref).max()
        assert np.abs(g - g_ref).max() < 1e-9 * scale, (np.abs(g - g_ref).max(), scale)


def _perturbed_start(x, G, k):
    rng = np.random.default_rng(2)
    x0 = x.copy()
    x0[: G * k] += 0.05 * rng.standard_normal(G * k)
    return x0


def test_lbfgs_driver_on_known_minima():
    """minimize_lbfgs (minFunc's default L-BFGS restated) on problems with known answers: a convex
    quadratic, Rosenbrock, and a function with a wall of non-finite values (as chol failing gives)."""
    from scipy.optimize import rosen, rosen_der
    from gp_dla_detection_amd.training import minimize_lbfgs
    rng = np.random.default_rng(0)
    A = rng.standard_normal((30, 30))
    A = A @ A.T + 30 * np.eye(30)
    b = rng.standard_normal(30)
    r = minimize_lbfgs(lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b), np.zeros(30))
    # (stops on minFunc's progTol test, g'd > -1e-9, which this well-conditioned problem reaches
    # while max|g| is still ~1e-4)
    assert np.abs(r.x - np.linalg.solve(A, b)).max() < 1e-5 and r.firstorderopt < 1e-3
    assert r.message in ("Optimality Condition below optTol", "Directional Derivative below progTol")
    r = minimize_lbfgs(lambda x: (rosen(x), rosen_der(x)), np.full(10, -1.2))
    assert np.abs(r.x - 1).max() < 1e-4 and r.fun < 1e-8
    assert r.trace_fval[0] > r.trace_fval[-1] and all(b <= a for a, b in zip(r.trace_fval, r.trace_fval[1:]))

    def walled(x):  # log-barrier: not finite for x <= 0; minimum at x = 1/3
        if (x <= 0).any():
            return np.inf, np.full(x.size, np.nan)
        return float(np.sum(3 * x - np.log(x))), 3 - 1 / x

    r = minimize_lbfgs(walled, np.full(4, 5.0))
    assert np.abs(r.x - 1 / 3).max() < 1e-5
    # budgets are respected
    r = minimize_lbfgs(lambda x: (rosen(x), rosen_der(x)), np.full(10, -1.2), max_iter=5)
    assert r.nit == 5 and r.message == "Reached Maximum Number of Iterations"
    r = minimize_lbfgs(lambda x: (rosen(x), rosen_der(x)), np.full(10, -1.2), max_fun_evals=12)
    assert r.nfev <= 12 + 1


def test_prior_value_is_what_the_gradient_prior_terms_differentiate(oracle):
    """objective.m:59-71 adds the tau_0 / beta priors to g only.  prior_value() is the missing value
    term: f + prior_value has exactly g as its gradient (central differences), f alone does not."""
    from gp_dla_detection_amd.training import prior_value
    x, F, L1, NV = training_problem(nq=30, G=48, k=3, seed=4)
    x[-2] += 0.3  # away from the prior means, where the prior terms vanish
    x[-1] -= 0.1
    _, g = oracle.objective(x, F, L1, NV)
    for i in (-2, -1):
        e = np.zeros_like(x)
        e[i] = 1e-6
        fp, fm = oracle.objective(x + e, F, L1, NV)[0], oracle.objective(x - e, F, L1, NV)[0]
        d_ref = (fp - fm) / 2e-6
        d_full = d_ref + (prior_value(x + e) - prior_value(x - e)) / 2e-6
        assert abs(d_full - g[i]) < 1e-5 * abs(g[i])
        assert abs(d_ref - g[i]) > 1e-3 * abs(g[i])


def test_lbfgs_driver_fits_the_oracle_objective(oracle):
    from gp_dla_detection_amd.training import minimize_lbfgs, prior_value
    x, F, L1, NV = training_problem(nq=80, G=96, k=4, seed=9)
    x0 = _perturbed_start(x, 96, 4)

    def consistent(xx):
        f, g = oracle.objective(xx, F, L1, NV)
        return f + prior_value(xx), g

    f0 = consistent(x0)[0]
    r = minimize_lbfgs(consistent, x0, max_iter=60, max_fun_evals=120)
    assert r.fun < f0 - 5000 and r.nit == 60 and np.isfinite(r.x).all()
    # the reference's own (f, g) pair: the driver still descends, and stops when the line search
    # along a prior-dominated direction can find no decrease
    r2 = minimize_lbfgs(lambda xx: oracle.objective(xx, F, L1, NV), x0, max_iter=60, max_fun_evals=120)
    assert r2.fun < oracle.objective(x0, F, L1, NV)[0] - 5000


@pytest.mark.gpu
def test_gpu_fit_decreases_objective(oracle):
    from gp_dla_detection_amd import training
    x, F, L1, NV = training_problem(nq=80, G=96, k=4, seed=9)
    x0 = _perturbed_start(x, 96, 4)
    f0 = training.objective(x0, F, L1, NV)[0]
    for prior_in_value in (False, True):
        x1, f1, res = training.fit(x0, F, L1, NV, max_iter=30, max_fun_evals=60, prior_in_value=prior_in_value)
        assert f1 < f0 - 5000 and np.isfinite(x1).all() and res.nfev <= 61
        # the value the driver reports is the reference's objective at the returned x
        assert abs(f1 - oracle.objective(x1, F, L1, NV)[0]) < 1e-9 * abs(f1)
    # same driver on the oracle's objective: identical iteration count, matching value (the two
    # objectives agree to ~1e-12, so 30 iterations of a deterministic driver stay together)
    r = training.minimize_lbfgs(lambda xx: (oracle.objective(xx, F, L1, NV)[0] + training.prior_value(xx),
                                            oracle.objective(xx, F, L1, NV)[1]), x0, max_iter=30, max_fun_evals=60)
    assert r.nit == res.nit and abs((r.fun - training.prior_value(r.x)) - f1) < 1e-6 * abs(f1)


@pytest.mark.gpu
def test_gpu_objective_is_deterministic_and_matches_the_oracle_on_ragged_shapes(oracle):
    """The matrix-core path sums every partial in a fixed order: two evaluations agree bit for
    bit, and -- on shapes that are not multiples of the 16-row / 4-step / 64-pixel tilings -- with
    the oracle to 1e-9 relative (spectrum_loss.m:31-74)."""
    from gp_dla_detection_amd import training
    for (nq, G, k) in ((37, 203, 20), (130, 70, 7), (5, 17, 2), (21, 90, 23), (9, 130, 40)):
        x, F, L1, NV = training_problem(nq=nq, G=G, k=k, seed=100 + k)
        t = training.TrainingSet(F, L1, NV)
        f1, g1 = t.objective(x)
        f2, g2 = t.objective(x)
        t.close()
        assert f1 == f2 and np.array_equal(g1, g2), (nq, G, k)  # (k > 20: slot-ordered sums, no atomics)
        f_ref, g_ref = oracle.objective(x, F, L1, NV)
        assert abs(f1 - f_ref) < 1e-9 * abs(f_ref), (nq, G, k, f1, f_ref)
        assert np.abs(g1 - g_ref).max() < 1e-9 * np.abs(g_ref).max(), (nq, G, k)


@pytest.mark.gpu
def test_gpu_training_handle_survives_rank_changes(oracle):
    """One TrainingSet evaluated at k = 4, 20, 7, 20: the captured graph is dropped whenever the
    buffers it points at are replaced (growth) or the rank changes, never replayed stale."""
    from gp_dla_detection_amd import training
    rng = np.random.default_rng(8)
    nq, G = 30, 75
    L1 = 1 + rng.uniform(1.5, 3.0, (nq, G))
    NV = 10 ** rng.uniform(-3, -1, (nq, G))
    F = 0.1 * rng.standard_normal((nq, G))
    t = training.TrainingSet(F, L1, NV)
    try:
        for k in (4, 20, 33, 7, 40, 20, 4):  # (k <= 20 and k <= 40 are two workspace classes)
            x = np.concatenate([(rng.standard_normal((G, k)) * 0.3 * 0.8 ** np.arange(k)).ravel(order="F"),
                                rng.uniform(-3, -2, G), [np.log(0.1), np.log(0.0023), np.log(3.65)]])
            f, g = t.objective(x)
            f_ref, g_ref = oracle.objective(x, F, L1, NV)
            assert abs(f - f_ref) < 1e-9 * abs(f_ref), k
            assert np.abs(g - g_ref).max() < 1e-9 * np.abs(g_ref).max(), k
    finally:
        t.close()


@pytest.mark.gpu
def test_gpu_objective_not_positive_definite_is_reported():
    from gp_dla_detection_amd import _lib, training
    x, F, L1, NV = training_problem(nq=20, G=48, k=4, seed=5)
    NV = NV.copy()
    NV[2] = -0.5  # negative variances make B = I + M'D^-1 M indefinite for that quasar (chol throws, :42)
    with pytest.raises(_lib.GpdlaError) as e:
        training.objective(x, F, L1, NV)
    assert e.value.code == -4


# ---------------------------------------------------------------------------------------------
# the mean-flux model's objective: multi_dlas/objective_lyseries.m over spectrum_loss_lyseries.m
# ---------------------------------------------------------------------------------------------

def lyseries_problem(nq=24, G=72, k=5, seed=3):
    """A training set in which the Lyman-series cut-offs bite: rest wavelengths from the Lyman limit
    to Lyman alpha, so the higher lines of a pixel fall beyond the quasar for part of the grid."""
    rng = np.random.default_rng(seed)
    x, F, _, NV = training_problem(nq=nq, G=G, k=k, seed=seed)
    rest = np.linspace(911.75, 1215.6701, G)
    L1 = (1 + rng.uniform(2.1, 4.5, nq))[:, None] * rest[None, :] / 1215.6701  # lya_1pzs; last column = 1 + z_qso
    return x, F, L1, NV


def series_tables():
    from gp_dla_detection_amd import _lyman_data as ld  # (wavelength [cm], oscillator strength, ...) per line
    wl = np.array([line[0] * 1e8 for line in ld.LINES])   # set_parameters_multi.m:77-108: the same numbers, in Angstrom
    fs = np.array([line[1] for line in ld.LINES])
    return wl, fs


def test_oracle_lyseries_objective(oracle):
    """One line is the plain objective, bit for bit; with six lines the value moves and the gradient
    agrees with finite differences for M, log omega, log c0 and log tau0 (the log beta entry is the
    reference's as-written expression, spectrum_loss_lyseries.m:90, which uses log(lya_1pz) for
    every line -- not a derivative of the value when more than one line is on)."""
    x, F, L1, NV = lyseries_problem()
    wl, fs = series_tables()
    f1, g1 = oracle.objective(x, F, L1, NV, num_threads=1)  # (one thread: the oracle's sums over quasars are then ordered)
    f1b, g1b = oracle.objective_lyseries(x, F, L1, NV, 1, wl, fs, num_threads=1)
    assert f1 == f1b and np.array_equal(g1, g1b)
    f6, g6 = oracle.objective_lyseries(x, F, L1, NV, 6, wl, fs)
    assert abs(f6 - f1) > 1e-3
    t0 = np.exp(x[-2])
    prior_t = t0 * (t0 - 0.0023) / 0.0007 ** 2
    rng = np.random.default_rng(1)
    for i in list(rng.choice(x.size - 3, 8, replace=False)) + [x.size - 3, x.size - 2]:
        e = np.zeros(x.size)
        e[i] = 1e-6
        fd = (oracle.objective_lyseries(x + e, F, L1, NV, 6, wl, fs)[0]
              - oracle.objective_lyseries(x - e, F, L1, NV, 6, wl, fs)[0]) / 2e-6
        assert abs(g6[i] - (prior_t if i == x.size - 2 else 0.0) - fd) < 1e-6 * max(1.0, abs(fd)), i


@pytest.mark.gpu
def test_gpu_lyseries_objective_matches_the_oracle(oracle):
    """gpdla_training_set_lyseries: 1 line = the plain objective bit for bit; 6 and 31 lines (the
    library's own table and a caller's) agree with the oracle to 1e-9 relative on ragged shapes in
    both rank classes; switching back restores the plain objective; bad tables are refused."""
    from gp_dla_detection_amd import _lib, training
    wl, fs = series_tables()
    for (nq, G, k) in ((24, 72, 5), (37, 203, 20), (9, 130, 33)):
        x, F, L1, NV = lyseries_problem(nq=nq, G=G, k=k, seed=40 + k)
        t = training.TrainingSet(F, L1, NV)
        try:
            f0, g0 = t.objective(x)
            t.set_lyseries(1)
            f1, g1 = t.objective(x)
            assert f0 == f1 and np.array_equal(g0, g1), (nq, G, k)
            for nfl, tables in ((6, (None, None)), (31, (wl, fs)), (3, (wl[:3] * 7.0, fs[:3]))):  # (any unit)
                t.set_lyseries(nfl, *tables)
                f, g = t.objective(x)
                f2, g2 = t.objective(x)
                assert f == f2 and np.array_equal(g, g2)  # deterministic
                f_ref, g_ref = oracle.objective_lyseries(x, F, L1, NV, nfl, wl, fs)
                assert abs(f - f_ref) < 1e-9 * abs(f_ref), (nq, G, k, nfl, f, f_ref)
                assert np.abs(g - g_ref).max() < 1e-9 * np.abs(g_ref).max(), (nq, G, k, nfl)
                assert abs(f - f0) > 1e-6 * abs(f0)  # the extra lines do something on this grid
            t.set_lyseries(0)
            f3, g3 = t.objective(x)
            assert f3 == f0 and np.array_equal(g3, g0)
            with pytest.raises(_lib.GpdlaError):
                t.set_lyseries(3, wl[[0, 2, 1]], fs[:3])  # not decreasing
            with pytest.raises(_lib.GpdlaError):
                t.set_lyseries(40)
        finally:
            t.close()
    x, F, L1, NV = lyseries_problem()
    f, g = training.objective_lyseries(x, F, L1, NV, 6, wl, fs)
    f_ref, g_ref = oracle.objective_lyseries(x, F, L1, NV, 6, wl, fs)
    assert abs(f - f_ref) < 1e-9 * abs(f_ref) and np.abs(g - g_ref).max() < 1e-9 * np.abs(g_ref).max()
