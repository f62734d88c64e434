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


def test_oracle_against_exact_arithmetic(golden, oracle):
    """The anchor for the unpinned MATLAB half: log N(y; mu, M M' + diag d) in 50-digit mpmath at
    the same fp64 inputs (tests/golden/make_exact.py).  A correctly rounded MATLAB evaluation of
    log_mvnpdf_low_rank.m:11-32 lies within its own rounding error of that value, so this bounds
    the oracle against ANY faithful MATLAB to 1e-9 -- not merely against itself."""
    g, e = golden("log_mvnpdf_low_rank.npz"), golden("exact_log_mvnpdf.npz")
    for c in range(int(e["num_cases"])):
        lp, rc = oracle.log_mvnpdf_low_rank(g[f"y_{c}"], g[f"mu_{c}"], g[f"M_{c}"], g[f"d_{c}"])
        assert rc == 0 and abs(lp - float(e[f"log_p_exact_{c}"])) < 1e-9, c
        assert abs(oracle.dense_log_mvnpdf(g[f"y_{c}"], g[f"mu_{c}"], g[f"M_{c}"], g[f"d_{c}"])
                   - float(e[f"log_p_exact_{c}"])) < 1e-9, c


def exact_case_inputs(golden):
    """The fp64 inputs process_qsos.m:190-198 hands to log_mvnpdf_low_rank for the 32 picked samples
    of the config-1 quasar, from the stored absorption vectors."""
    s, e = golden("spectrum_config1.npz"), golden("exact_log_mvnpdf.npz")
    mask = s["pixel_mask"].astype(bool)
    rest = s["wavelengths"] / (1 + float(s["z_qso"]))
    ind = (rest >= 911.75) & (rest <= 1215.75) & ~mask
    y, nv = s["flux"][ind], s["noise_variance"][ind]
    mu, M, om2 = s["this_mu"], s["this_M"], s["this_omega2"]
    yield "null", y, mu, M, om2 + nv, float(e["null_log_p_exact"])
    for a, i, ex in zip(e["absorption"], e["sample_indices"], e["sample_log_p_exact"]):
        yield int(i), y, mu * a, M * a[:, None], om2 * a ** 2 + nv, float(ex)


def test_oracle_sweep_inputs_against_exact_arithmetic(golden, oracle):
    """Same anchor on the path's own operating point: the null model and 32 absorbed samples of
    the BASELINE config-1 quasar (n = 800, k = 20), log-likelihoods down to -3.3e4."""
    s = golden("spectrum_config1.npz")
    worst = 0.0
    for tag, y, mu, M, d, exact in exact_case_inputs(golden):
        lp, rc = oracle.log_mvnpdf_low_rank(y, mu, M, d)
        assert rc == 0
        worst = max(worst, abs(lp - exact))
        frozen = float(s["log_likelihood_no_dla"]) if tag == "null" else float(s["sample_log_likelihoods_dla"][tag])
        assert abs(frozen - exact) < 1e-9, tag   # the frozen driver output too
    assert worst < 1e-9, worst
