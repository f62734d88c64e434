"""The Woodbury restatement of log_mvnpdf_low_rank.m:5-34 vs its frozen golden outputs and vs an
independent dense evaluation (K formed explicitly)."""
import numpy as np
import pytest


def test_golden_cases(golden, oracle):
    g = golden("log_mvnpdf_low_rank.npz")
    for c in range(int(g["num_cases"])):
        lp, rc = oracle.log_mvnpdf_low_rank(g[f"y_{c}"], g[f"mu_{c}"], g[f"M_{c}"], g[f"d_{c}"])
        assert rc == 0
        assert lp == float(g[f"log_p_{c}"])  # same code, same machine family: bit-exact
        assert abs(lp - float(g[f"log_p_dense_{c}"])) < 1e-9 * abs(lp)


@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (64, 20), (300, 7)])
def test_against_dense(oracle, n, k):
    rng = np.random.default_rng(n * 100 + k)
    M = rng.standard_normal((n, k))
    mu = rng.standard_normal(n)
    d = 10.0 ** rng.uniform(-3, 1, n)
    y = rng.standard_normal(n) * 2
    lp, rc = oracle.log_mvnpdf_low_rank(y, mu, M, d)
    assert rc == 0
    assert abs(lp - oracle.dense_log_mvnpdf(y, mu, M, d)) < 1e-9 * max(1.0, abs(lp))


def test_not_positive_definite(oracle):
    # a negative diagonal entry large enough to make B = I + M' D^-1 M indefinite: MATLAB's chol
    # throws here (log_mvnpdf_low_rank.m:24); the oracle reports rc = -1 and NaN
    n, k = 6, 2
    M = np.ones((n, k))
    d = np.full(n, -0.5)
    lp, rc = oracle.log_mvnpdf_low_rank(np.zeros(n), np.zeros(n), M, d)
    assert rc == -1 and np.isnan(lp)
