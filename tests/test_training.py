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


@pytest.mark.gpu
def test_gpu_fit_decreases_objective():
    from gp_dla_detection_amd import training
    x, F, L1, NV = training_problem(nq=80, G=96, k=4, seed=9)
    rng = np.random.default_rng(2)
    x0 = x.copy()
    x0[: 96 * 4] += 0.05 * rng.standard_normal(96 * 4)
    t = training.TrainingSet(F, L1, NV)
    f0, _ = t.objective(x0)
    t.close()
    x1, f1, res = training.fit(x0, F, L1, NV, max_iter=30, max_fun_evals=60)
    assert f1 < f0 and np.isfinite(x1).all()


@pytest.mark.gpu
def test_gpu_objective_is_deterministic_and_matches_the_atomic_kernel():
    """The matrix-core path sums every partial in a fixed order: two evaluations agree bit for
    bit (the one-block-per-quasar kernel it replaces for k <= 20 accumulated g with fp64 atomics).
    Shapes that are not multiples of the 16-row / 4-step / 64-pixel tilings included."""
    from gp_dla_detection_amd import training
    for (nq, G, k) in ((37, 203, 20), (130, 70, 7), (5, 17, 2)):
        x, F, L1, NV = training_problem(nq=nq, G=G, k=k, seed=100 + k)
        t = training.TrainingSet(F, L1, NV)
        f1, g1 = t.objective(x)
        f2, g2 = t.objective(x)
        t.close()
        assert f1 == f2 and np.array_equal(g1, g2)
        assert np.isfinite(g1).all()


@pytest.mark.gpu
def test_gpu_objective_not_positive_definite_is_reported():
    from gp_dla_detection_amd import _lib, training
    x, F, L1, NV = training_problem(nq=20, G=48, k=4, seed=5)
    NV = NV.copy()
    NV[2] = -0.5  # negative variances make B = I + M'D^-1 M indefinite for that quasar (chol throws, :42)
    with pytest.raises(_lib.GpdlaError) as e:
        training.objective(x, F, L1, NV)
    assert e.value.code == -4
