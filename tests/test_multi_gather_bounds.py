"""Host-side address model of the multi-DLA sweeps' profile gathers (tools/multi_gather_address_model.py):
the shipped, clamped index never leaves its row of the profile table -- for every pixel count a
quasar can have -- and the unclamped index of the experiment abandoned in round 4 does, by up to 12
entries, past the END of the table for the last row of a sub-batch (DESIGN.md section 4 records what
that means for the two aborted runs).  No GPU, no library."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("multi_gather_address_model",
                                                  os.path.join(ROOT, "tools", "multi_gather_address_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stride_formula_matches_the_library_source():
    """The model restates multi_alloc's stride; hold it to the line in csrc/gpdla.hip."""
    src = open(os.path.join(ROOT, "gp_dla_detection_amd", "csrc", "gpdla.hip")).read()
    assert "const int64_t stride = ((4 * ((b->max_pix + 3) / 4) + 4 + 15) / 16) * 16;" in src
    assert "const size_t need = (size_t)nq_sub * 2 * S * stride;" in src
    kern = open(os.path.join(ROOT, "gp_dla_detection_amd", "csrc", "sweep_multi_slim_kernel.hpp")).read()
    assert "const int p_last = 4 * m.steps + jj;" in kern
    assert "gather(min(4 * t + jj, p_last), raw[t]);" in kern and "gather(min(4 * (rn + kAhead) + jj, p_last), raw[tt % kAhead]);" in kern
    m = _model()
    assert [m.stride_of(n) for n in (1, 12, 13, 404, 1500)] == [16, 16, 32, 416, 1504]


def test_shipped_gather_index_stays_inside_its_row():
    m = _model()
    for n in list(range(1, 1300)) + [1499, 1500, 1501, 4999]:
        for clamped_n_u in (n, max(1, n - 7)):
            a = m.model([n], 5, clamped=True, n_u=[clamped_n_u])
            assert not a["beyond_row"] and a["beyond_allocation_bytes"] == 0, n
    a = m.model([320, 211, 402], 160, clamped=True)          # a ragged sub-batch: the stride is the longest quasar's
    assert not a["beyond_row"] and a["beyond_allocation_bytes"] == 0


def test_unclamped_index_of_the_abandoned_experiment_leaves_the_table():
    """The golden multi-DLA spectrum (404 pixels, S = 256): 2 S 8 B = 4096 B per entry of stride, so the
    table ends exactly on a 4-KiB page, and the unclamped request of the last K-steps runs 32 B past it
    unless the rows are padded by 12 more entries (and the sub-batch sizing follows the larger stride)."""
    m = _model()
    g = np.load(os.path.join(ROOT, "tests", "golden", "spectrum_multi.npz"))
    n = int(g["wavelengths"].size)
    b = m.model([n], 256, clamped=False)
    assert b["allocation_ends_on_page"] and b["beyond_row"] and b["beyond_allocation_bytes"] == 32
    c = m.model([n], 256, clamped=False, pad=12)
    assert not c["beyond_row"] and c["beyond_allocation_bytes"] == 0
