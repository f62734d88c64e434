"""Round trips of the .mat readers/writers through scipy (v5 files; -v7.3 needs h5py)."""
import numpy as np
import pytest
from scipy.io import loadmat, savemat

from gp_dla_detection_amd import io, synthetic


def test_model_and_samples_round_trip(tmp_path):
    model = synthetic.make_model(20)
    savemat(tmp_path / "m.mat", {k: (np.asarray(v).reshape(-1, 1) if np.ndim(v) == 1 else v)
                                 for k, v in model.items()})
    got = io.load_learned_model(str(tmp_path / "m.mat"))
    for k in ("rest_wavelengths", "mu", "log_omega"):
        np.testing.assert_array_equal(got[k], model[k])
    np.testing.assert_array_equal(got["M"], model["M"])
    assert got["log_beta"] == model["log_beta"]
    s = synthetic.make_samples(50)
    savemat(tmp_path / "s.mat", {k: v.reshape(1, -1) for k, v in s.items()})  # row vectors, as MATLAB
    gs = io.load_dla_samples(str(tmp_path / "s.mat"))
    np.testing.assert_array_equal(gs["nhi_samples"], s["nhi_samples"])
    np.testing.assert_array_equal(gs["lls_nhi_samples"], s["lls_nhi_samples"])


def test_preloaded_cells_and_output(tmp_path):
    model = synthetic.make_model(20)
    spectra = [synthetic.make_spectrum(i, 50 + 7 * i, model, mask_fraction=0.1) for i in range(4)]
    cells = {}
    for key, src in (("all_wavelengths", "wavelengths"), ("all_flux", "flux"),
                     ("all_noise_variance", "noise_variance"), ("all_pixel_mask", "pixel_mask")):
        c = np.empty((4, 1), dtype=object)
        for i, sp in enumerate(spectra):
            c[i, 0] = np.asarray(sp[src]).reshape(-1, 1)
        cells[key] = c
    savemat(tmp_path / "p.mat", cells)
    z = [sp["z_qso"] for sp in spectra]
    got = io.load_preloaded_qsos(str(tmp_path / "p.mat"), z, test_ind=np.array([True, False, True, True]))
    assert len(got) == 3 and got[1]["z_qso"] == z[2]
    np.testing.assert_array_equal(got[1]["wavelengths"], spectra[2]["wavelengths"])
    np.testing.assert_array_equal(got[2]["pixel_mask"], spectra[3]["pixel_mask"])
    assert np.array_equal(np.isnan(got[0]["flux"]), np.isnan(spectra[0]["flux"]))
    res = {k: np.arange(3.0) for k in io.SAVED_VARIABLES}
    res["sample_log_likelihoods_dla"] = np.arange(15.0).reshape(3, 5)
    res["model_posteriors"] = np.ones((3, 2)) / 2
    res["num_lines"] = 3
    io.save_processed_qsos(str(tmp_path / "o.mat"), res, test_set_name="dr12q")
    back = loadmat(tmp_path / "o.mat")
    assert back["sample_log_likelihoods_dla"].shape == (3, 5) and back["p_dlas"].shape == (3, 1)
    assert back["test_set_name"][0] == "dr12q" and int(back["num_lines"].ravel()[0]) == 3


def test_v73_needs_h5py(tmp_path):
    p = tmp_path / "v73.mat"
    p.write_bytes(b"MATLAB 7.3 MAT-file, Platform: GLNXA64" + b" " * 90 + b"\x89HDF\r\n\x1a\n")
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py present")
    except ImportError:
        with pytest.raises(ImportError):
            io.load_dla_samples(str(p))
