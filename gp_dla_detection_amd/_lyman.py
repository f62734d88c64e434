"""Lyman-series line data parsed from include/gpdla_lyman_series.h (single source of the numbers)."""
from __future__ import annotations

import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include",
                    "gpdla_lyman_series.h")


def _parse():
    text = open(_HDR).read()
    rows = re.findall(
        r"GPDLA_LINE\(\s*(\d+),\s*([^,]+),\s*([^,]+),\s*([^,]+),\s*([^,]+),\s*([^)]+)\)", text)
    lines = [tuple(float(x) for x in r[1:]) for r in rows]
    c = float(re.search(r"GPDLA_SPEED_OF_LIGHT_CGS\s+(\S+)", text).group(1))
    sigma = float(re.search(r"GPDLA_GAUSS_SIGMA_CGS\s+(\S+)", text).group(1))
    taps = re.search(r"GPDLA_INSTRUMENT_PROFILE\s*\{([^}]*)\}", text).group(1)
    prof = [float(x) for x in taps.split(",")]
    return lines, c, sigma, prof


#: rows of (transition_wavelength_cm, oscillator_strength, Gamma, leading_constant, gamma)
LINES, C_CGS, SIGMA_CGS, INSTRUMENT_PROFILE = _parse()
assert len(LINES) == 31 and len(INSTRUMENT_PROFILE) == 7
