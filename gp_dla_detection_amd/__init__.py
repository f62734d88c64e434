"""gp_dla_detection_amd -- MI355X-native GP marginal-likelihood sweep for DLA detection.

One hot path of jibanCat/gp_dla_detection (the per-spectrum sweep of ``process_qsos.m``) rebuilt
for gfx950: hand-written HIP kernels behind a C-ABI (``include/gpdla.h``), with this package as the
Python host side mirroring the reference's call surface.  See DESIGN.md.
"""
from .api import (Batch, Context, dla_existence_prior, dla_existence_prior_multi,
                  log_mvnpdf_low_rank, prepare_prior, process_qsos,
                  process_qsos_multiple_dlas_meanflux, spectra_to_csr, voigt)
from .parameters import MultiParameters, Parameters, kms_to_z

__all__ = ["Batch", "Context", "dla_existence_prior", "dla_existence_prior_multi",
           "log_mvnpdf_low_rank", "prepare_prior", "process_qsos", "process_qsos_multiple_dlas_meanflux",
           "spectra_to_csr", "voigt", "Parameters", "MultiParameters", "kms_to_z"]
