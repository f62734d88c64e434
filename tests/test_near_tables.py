"""The sweep kernel's accurate tier: per-line piecewise polynomials of Re w(x + i y_line) for
|x| < 32 (csrc/near_tables.hpp), which stand where the reference calls libcerf's voigt()
(voigt.c:288).  The library evaluates the same coefficient tables on the host through
gpdla_debug_near_poly; here they are checked against 40-digit mpmath.  No GPU needed."""
import ctypes as C
import os

import mpmath as mp
import numpy as np
import pytest

from gp_dla_detection_amd import _lib

REL_TOL = 3e-15  # interpolation (<= 1.4e-15) + Horner rounding in fp64


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def rew_mp(x, y):
    z = mp.mpc(mp.mpf(x), mp.mpf(y))
    return mp.re(mp.exp(-z * z) * mp.erfc(-1j * z))


def near_poly(lib, line, x):
    v, y = C.c_double(), C.c_double()
    assert lib.gpdla_debug_near_poly(line, float(x), C.byref(v), C.byref(y)) == 0
    return v.value, y.value


def test_damping_parameters_are_those_of_the_line_tables(lib):
    tables = np.load(os.path.join(os.path.dirname(__file__), "golden", "lyman_tables.npz"))
    gammas, sigma = tables["gammas"], float(tables["sigma"])  # voigt.c:146, 185-220 via voigt.py
    for line in (0, 1, 2, 30):
        _, y = near_poly(lib, line, 1.0)
        assert y == pytest.approx(gammas[line] / np.sqrt(2.0) / sigma, rel=1e-15)


@pytest.mark.parametrize("line", [0, 1, 2, 7, 30])
def test_tables_against_mpmath(lib, line):
    mp.mp.dps = 40
    rng = np.random.default_rng(1234 + line)
    xs = np.concatenate([
        rng.uniform(0.0, 8.0, 120),           # Gaussian core, width-1/8 intervals
        rng.uniform(8.0, 30.0, 80),           # damping wing, width-1/2 intervals
        np.arange(0, 64) / 8.0,               # interval edges (left ends)
        np.nextafter(np.arange(1, 65) / 8.0, 0.0),  # ... and right ends
        8.0 + np.arange(0, 44) / 2.0,
        np.nextafter(8.0 + np.arange(1, 45) / 2.0, 0.0),
        [0.0, 1e-300, 7.999999999999999, 8.0, 29.999999999999996, 30.0, 31.9],
    ])
    worst = 0.0
    for x in xs:
        v, y = near_poly(lib, line, x)
        ref = rew_mp(x, y)
        worst = max(worst, float(abs((mp.mpf(v) - ref) / ref)))
    assert worst < REL_TOL, f"line {line}: worst relative error {worst:.2e}"


def test_even_in_x_and_argument_checks(lib):
    a, _ = near_poly(lib, 0, 3.25)
    b, _ = near_poly(lib, 0, -3.25)
    assert a == b
    v = C.c_double()
    assert lib.gpdla_debug_near_poly(31, 1.0, C.byref(v), None) == -1   # 31 lines: 0..30
    assert lib.gpdla_debug_near_poly(0, 32.0, C.byref(v), None) == -1   # outside the tables
    assert lib.gpdla_debug_near_poly(0, float("nan"), C.byref(v), None) == -1
