"""The slim step records (vech(m m') formed inside the sweep) against the pre-expanded records they
replace: k_sweep_slim vs k_sweep and k_sweep_multi_slim vs k_sweep_multi (k <= 20),
k_sweep_split_slim vs k_sweep_split and k_sweep_multi_split (20 < k <= 40).  Same products, same MFMA sequence per
column, same epilogue: every output must be bit-identical.  The superseded kernels are not in the
product library: they live in libgpdla_legacy.so (csrc/gpdla.hip built with -DGPDLA_WITH_LEGACY,
__graft_entry__.build()), which a clean child process loads through GPDLA_LIB_PATH with the
diagnostic switch GPDLA_EXPANDED_RECORDS=1 set (hot loops: process_qsos.m:185-199,
process_qsos_multiple_dlas_meanflux.m:340-381)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import record_class_worker as rcw

pytestmark = pytest.mark.gpu


def expanded(kind, k, num_lines, tmp_path, extra_env=None):
    out = tmp_path / f"{kind}_{k}_{num_lines}.npz"
    from gp_dla_detection_amd import _lib
    assert os.path.exists(_lib.LEGACY_LIB_PATH), "libgpdla_legacy.so is missing: __graft_entry__.build() makes it"
    env = {"GPDLA_EXPANDED_RECORDS": "1", "GPDLA_LIB_PATH": _lib.LEGACY_LIB_PATH}
    env.update(extra_env or {})
    pr = mp.get_context("forkserver").Process(target=rcw.run_child, args=(kind, k, num_lines, env, str(out)))
    pr.start()
    pr.join(600)
    if pr.is_alive():  # our own child, by handle
        pr.kill()
        pr.join()
    assert pr.exitcode == 0
    return np.load(out)


@pytest.mark.parametrize("kind,k,num_lines", [("single", 20, 3), ("single", 40, 3), ("single", 33, 3),
                                              ("multi", 40, 3), ("multi", 23, 3), ("multi", 20, 3), ("multi", 17, 3)])
def test_slim_records_reproduce_the_expanded_records_bit_for_bit(kind, k, num_lines, tmp_path):
    want = expanded(kind, k, num_lines, tmp_path)
    got = rcw.run_case(kind, k, num_lines)
    checked = 0
    for name in want.files:
        a, b = np.asarray(got[name]), want[name]
        assert a.shape == b.shape, name
        if a.dtype.kind == "f":
            assert np.array_equal(a, b, equal_nan=True), (name, float(np.nanmax(np.abs(a - b))))
        else:
            assert np.array_equal(a, b), name
        checked += 1
    assert checked >= 10
    table = "sample_log_likelihoods_dla"
    assert np.isfinite(np.asarray(got[table])).any()


@pytest.mark.parametrize("k,num_lines", [(20, 5), (20, 31), (13, 1), (27, 5), (40, 31)])
def test_run_time_line_count_on_slim_records_matches_the_expanded_records(k, num_lines, tmp_path):
    """A line count other than set_parameters.m:63's three (voigt.c:16, 266 default to 31): the slim
    kernels (k_sweep_slim<0>, k_sweep_split_slim<0, 0>) keep no per-sample table of the line multipliers
    and form x_j = lambda / (1 + z_DLA) * kms_j - c / (sqrt2 sigma), four lines at a time
    (wing_sum_runtime), instead of (lambda mult_j - c) / (sqrt2 sigma) line by line, so they are held to
    the pre-expanded kernels' table-based arithmetic at 1e-9 (not bit for bit: the velocity and the sum
    over lines are rounded in a different order), beside the oracle tests in test_gpu_parity.py
    (test_num_lines_31_and_1 and the randomised shapes)."""
    want = expanded("single", k, num_lines, tmp_path)
    got = rcw.run_case("single", k, num_lines)
    table = "sample_log_likelihoods_dla"
    a, b = np.asarray(got[table]), want[table]
    assert a.shape == b.shape and np.isfinite(a).any()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.nanmax(np.abs(a - b)) < 1e-9 * max(1.0, float(np.nanmax(np.abs(b))))
    for name in ("log_likelihoods_no_dla", "log_posteriors_dla", "MAP_z_dlas", "MAP_log_nhis"):
        if name in want.files:
            x, y = np.asarray(got[name]), want[name]
            assert np.allclose(x, y, rtol=1e-9, atol=1e-8, equal_nan=True), name
