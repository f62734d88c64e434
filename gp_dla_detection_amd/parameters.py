"""Constants of the DLA pipeline that the inference sweep reads.

Restates the values of the reference's ``set_parameters.m`` (single-DLA) and
``multi_dlas/set_parameters_multi.m`` + ``process_qsos_multiple_dlas_meanflux.m:31-37`` (multi-DLA)
as dataclasses instead of MATLAB workspace variables (the workspace mechanism
itself is out of scope; the *values* are inputs to the hot path).
"""
from __future__ import annotations

from dataclasses import dataclass

SPEED_OF_LIGHT = 299792458.0  # m/s, set_parameters.m:8


def kms_to_z(kms: float) -> float:
    """set_parameters.m:11"""
    return (kms * 1000) / SPEED_OF_LIGHT


@dataclass(frozen=True)
class Parameters:
    lya_wavelength: float = 1215.6701      # set_parameters.m:5
    lyb_wavelength: float = 1025.7223      # :6
    lyman_limit: float = 911.7633          # :7
    min_lambda: float = 911.75             # :33
    max_lambda: float = 1215.75            # :34
    dlambda: float = 0.25                  # :35
    k: int = 20                            # :36
    num_dla_samples: int = 10000           # :48
    prior_z_qso_increase: float = kms_to_z(30000)  # :56
    width: int = 3                         # :59
    pixel_spacing: float = 1e-4            # :60
    num_lines: int = 3                     # :63
    max_z_cut: float = kms_to_z(3000)      # :65
    min_z_cut: float = kms_to_z(3000)      # :69
    # 0: fp64 contraction (parity-grade, default); 1: fp32-matrix-core study variant (BASELINE config 5)
    contraction_precision: int = 0
    # HBM budget of a batch's per-K-step records; a larger batch is swept group by group through one
    # pool of this size (results do not depend on it); 0 = 16 GiB
    record_pool_bytes: int = 0

    def min_z_dla(self, wavelengths, z_qso):
        """set_parameters.m:70-73"""
        return max(min(wavelengths) / self.lya_wavelength - 1,
                   self.lyman_limit * (1 + z_qso) / self.lya_wavelength - 1 + self.min_z_cut)

    def max_z_dla(self, wavelengths, z_qso):
        """set_parameters.m:66-67"""
        return (max(wavelengths) / self.lya_wavelength - 1) - self.max_z_cut


@dataclass(frozen=True)
class MultiParameters(Parameters):
    max_dlas: int = 4                          # process_qsos_multiple_dlas_meanflux.m:32
    min_z_separation: float = kms_to_z(3000)   # :33
    prev_tau_0: float = 0.0023                 # :36
    prev_beta: float = 3.65                    # :37
    num_forest_lines: int = 31                 # set_parameters_multi.m:75
    rng_seed: int = 0x9E3779B97F4A7C15         # GPU resampling stream (the reference: rng('default'))
    first_quasar_index: int = 0                # global index of this batch's first quasar (sharding)
    multi_profile_bytes: int = 0               # HBM budget of the Voigt profile table; 0 = 16 GiB
