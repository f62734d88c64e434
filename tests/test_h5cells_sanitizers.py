"""csrc/h5cells.c under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only): every cell of
a contiguous and of a deflate-compressed file, garbage object addresses, a file truncated at every
37th byte of its first 6 KB, a wrong element count -- no report, no leak, and what cannot be read is
refused, not guessed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from gp_dla_detection_amd import io

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "gp_dla_detection_amd", "csrc", "h5cells.c")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_native_cell_reader_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "harness")
    build = subprocess.run(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fopenmp",
                            os.path.join(HERE, "h5cells_harness.c"), SRC, "-lz", "-o", exe], capture_output=True, text=True)
    if build.returncode:
        pytest.skip("this gcc has no sanitizer runtime: " + build.stderr[-200:])
    rng = np.random.default_rng(0)
    lens = [2000, 150000, 0, 30, 1, 700, 513]
    cells = {k: [rng.standard_normal((m, 1)) for m in lens] for k in ("all_wavelengths", "all_flux", "all_noise_variance")}
    cells["all_pixel_mask"] = [rng.uniform(size=(m, 1)) < 0.2 for m in lens]
    for compress in (False, True):
        p = str(tmp_path / f"x{int(compress)}.mat")
        io.savemat73(p, cells, compress=compress)
        with io.PreloadedReader(p) as r:
            assert r._f.userblock_size == 512  # (the harness passes 512)
            addrs = np.concatenate([np.asarray(r._refs[k], dtype=np.uint64) for k in r.KEYS]
                                   + [np.array([0, 8, 96, 12345, 2 ** 40, 2 ** 63], dtype=np.uint64)])
        addrs.tofile(str(tmp_path / "addrs.bin"))
        run = subprocess.run([exe, p, str(tmp_path / "addrs.bin")], capture_output=True, text=True,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
        assert run.returncode == 0 and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-2000:]
        assert "ok 24, refused 10 of 34" in run.stdout, run.stdout  # 4 empty cells + 6 garbage addresses refused
