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
                                              ("single", 27, 5), ("multi", 40, 3), ("multi", 23, 3), ("multi", 20, 3), ("multi", 17, 3)])
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
