"""Pins the Voigt half of the CPU oracle against the reference's own numbers.

Golden vectors come from importing /root/reference/CDDF_analysis/voigt.py in the build container
(tests/golden/make_golden.py) and from 50-digit mpmath.  Tolerances: the reference's scipy wofz is
itself within 1.9e-14 of mpmath in this regime, so the oracle is held to
1e-13 relative on the line function and on the optical depth.
"""
import numpy as np

from gp_dla_detection_amd import _lyman


def test_line_tables_bit_identical(golden):
    g = golden("lyman_tables.npz")
    lines = np.array(_lyman.LINES)
    assert np.array_equal(lines[:, 0], g["transition_wavelengths"])
    assert np.array_equal(lines[:, 1], g["oscillator_strengths"])
    assert np.array_equal(lines[:, 2], g["Gammas"])
    assert np.array_equal(lines[:, 3], g["leading_constants"])
    assert np.array_equal(lines[:, 4], g["gammas"])
    assert np.array_equal(np.array(_lyman.INSTRUMENT_PROFILE), g["instrument_profile"])
    assert _lyman.C_CGS == float(g["c"]) and _lyman.SIGMA_CGS == float(g["sigma"])


def test_faddeeva_vs_mpmath(golden, oracle):
    g = golden("faddeeva.npz")
    for a, y in enumerate(g["y"]):
        got = oracle.faddeeva_re(g["x"], y)
        rel = np.abs(got - g["re_w"][a]) / g["re_w"][a]
        assert rel.max() < 1e-14, (y, g["x"][rel.argmax()], rel.max())


def test_line_function_vs_reference(golden, oracle):
    g = golden("voigt_profiles.npz")
    for row, j in enumerate(g["line_indices"]):
        gam = _lyman.LINES[int(j)][4]
        got = np.array([oracle.voigt_line(v, _lyman.SIGMA_CGS, gam) for v in g["line_velocities"]])
        rel = np.abs(got - g["line_values"][row]) / g["line_values"][row]
        assert rel.max() < 1e-13, (j, rel.max())


def test_profiles_vs_reference(golden, oracle):
    g = golden("voigt_profiles.npz")
    for c in range(int(g["num_cases"])):
        z, N, nl = g[f"args_{c}"]
        lam = g[f"lambdas_{c}"]
        raw = oracle.voigt(lam, z, N, int(nl), raw=True)
        prof = oracle.voigt(lam, z, N, int(nl))
        assert prof.size == lam.size - 6
        # compare optical depths where the profile is not fully saturated, and absorptions
        # everywhere: d(exp(-tau)) = exp(-tau) dtau <= 0.37 dtau/tau
        assert np.abs(raw - g[f"raw_{c}"]).max() < 2e-13, c
        assert np.abs(prof - g[f"profile_{c}"]).max() < 2e-13, c
        ok = g[f"raw_{c}"] > 1e-200
        tau_ref = -np.log(g[f"raw_{c}"][ok])
        tau = -np.log(raw[ok])
        big = tau_ref > 1e-2  # below that -log(1-tau) is rounding noise
        assert (np.abs(tau - tau_ref)[big] / tau_ref[big]).max() < 1e-12, c


def test_voigt_rejects_bad_arguments(oracle):
    import pytest
    lam = np.linspace(4000, 4001, 6)
    with pytest.raises(ValueError):
        oracle.voigt(lam, 2.0, 1e20, 3)
    with pytest.raises(ValueError):
        oracle.voigt(np.linspace(4000, 4100, 50), 2.0, 1e20, 32)
