"""The hot path between the reference's own file formats, end to end on the GPU: the three -v7.3
input files of process_qsos.m:30-61 (written here with MATLAB's conventions: cell arrays, logical
masks, row vectors), the sweep, the -v7.3 output record (process_qsos.m:236-250; multi :498-523)
opened the way CDDF_analysis opens it (qso_loader.py:84-112, calc_cddf.py:217-266), and the
ASCII / JSON catalogues (generate_ascii_catalog.m; qso_loader.py:1927-2031)."""
import json

import numpy as np
import pytest

import gp_dla_detection_amd as gp
from gp_dla_detection_amd import catalog, hdf5, io, synthetic
from gp_dla_detection_amd.parameters import MultiParameters

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipeline")
    model = synthetic.make_model(20)
    samples = synthetic.make_samples(400)
    all_spectra = [synthetic.make_spectrum(900 + i, 260 + 31 * i, model, mask_fraction=0.05) for i in range(9)]
    test_ind = np.array([1, 0, 1, 1, 0, 1, 1, 0, 1], dtype=bool)
    io.savemat73(str(d / "learned_qso_model_dr9q_minus_concordance.mat"),
                 {k: (np.asarray(v).reshape(1, -1) if np.ndim(v) == 1 else v) for k, v in model.items()},  # row vectors
                 compress=True)                                        # learn_qso_model.m:113-123
    io.savemat73(str(d / "dla_samples.mat"), {k: v.reshape(1, -1) for k, v in samples.items()})  # generate_dla_samples.m:59-63
    cells = {}
    for key, src in (("all_wavelengths", "wavelengths"), ("all_flux", "flux"),
                     ("all_noise_variance", "noise_variance"), ("all_pixel_mask", "pixel_mask")):
        cells[key] = [np.asarray(s[src]).astype(bool if src == "pixel_mask" else np.float64).reshape(-1, 1)
                      for s in all_spectra]
    io.savemat73(str(d / "preloaded_qsos.mat"), cells, compress=True)  # preload_qsos.m:64-79
    z_qsos = np.array([s["z_qso"] for s in all_spectra])
    return d, model, samples, all_spectra, test_ind, z_qsos


def test_single_dla_run_between_v73_files(files):
    d, model, samples, all_spectra, test_ind, z_qsos = files
    m = io.load_learned_model(str(d / "learned_qso_model_dr9q_minus_concordance.mat"))
    s = io.load_dla_samples(str(d / "dla_samples.mat"))
    spectra = io.load_preloaded_qsos(str(d / "preloaded_qsos.mat"), z_qsos, test_ind)
    assert len(spectra) == 6
    cat = synthetic.make_prior_catalog()
    out = gp.process_qsos(m, s, spectra, prior_catalog=cat)
    # the same run from the in-memory inputs: the files round-trip every bit
    direct = gp.process_qsos(model, samples, [sp for sp, t in zip(all_spectra, test_ind) if t], prior_catalog=cat)
    for key in ("sample_log_likelihoods_dla", "log_likelihoods_no_dla", "model_posteriors", "MAP_inds"):
        np.testing.assert_array_equal(out[key], direct[key], err_msg=key)
    p = str(d / "processed_qsos_dr12q.mat")
    io.save_processed_qsos(p, out, test_ind=test_ind, training_release="dr12q", release="dr12q",
                           training_set_name="dr9q_minus_concordance", dla_catalog_name="dr9q_concordance",
                           prior_ind="prior_catalog.in_dr9 & prior_catalog.los_inds(dla_catalog_name)",
                           test_set_name="dr12q")
    with hdf5.File(p) as f:  # as qso_loader.py:84-112 / calc_cddf.py:217-220 index it
        ti = f["test_ind"][0, :].astype(bool)
        assert np.array_equal(ti, test_ind)
        np.testing.assert_array_equal(f["p_dlas"][0, :], out["p_dlas"])
        np.testing.assert_array_equal(f["p_no_dlas"][0, :], out["p_no_dlas"])
        np.testing.assert_array_equal(f["log_priors_dla"][0, :], out["log_priors_dla"])
        np.testing.assert_array_equal(f["model_posteriors"][()].T, out["model_posteriors"])
        sll = f["sample_log_likelihoods_dla"]
        assert sll.shape == (400, 6)
        np.testing.assert_array_equal(sll[()].T, out["sample_log_likelihoods_dla"])
    thing_ids = np.arange(100000, 100009)[test_ind]
    catalog.write_results(str(d / "dr12q_results.dat"), thing_ids, out, s)
    lines = open(d / "dr12q_results.dat").read().splitlines()
    assert len(lines) == 6
    last = lines[0].split()
    z_map, n_map, _ = catalog.map_estimates_host(out, s)
    assert abs(float(last[-2]) - z_map[0]) < 5e-5 and abs(float(last[-1]) - n_map[0]) < 5e-5


def test_multi_dla_run_to_v73_and_json(files):
    d, model, samples, all_spectra, test_ind, z_qsos = files
    p = MultiParameters(max_dlas=3)
    spectra = io.load_preloaded_qsos(str(d / "preloaded_qsos.mat"), z_qsos, test_ind)
    cat = synthetic.make_prior_catalog()
    lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z_qsos[test_ind], 0.31, 0.69, p)
    out = gp.process_qsos_multiple_dlas_meanflux(io.load_learned_model(str(d / "learned_qso_model_dr9q_minus_concordance.mat")),
                                                 io.load_dla_samples(str(d / "dla_samples.mat")), spectra, lp, params=p)
    path = str(d / "processed_qsos_multi_meanflux_dr12q.mat")
    io.save_processed_qsos_multi(path, out, test_ind=test_ind, k=20, num_dla_samples=400, test_set_name="dr12q")
    with hdf5.File(path) as f:
        assert f["sample_log_likelihoods_dla"].shape == (3, 400, 6)              # calc_cddf.py:266
        np.testing.assert_array_equal(f["sample_log_likelihoods_dla"][2, :, 4], out["sample_log_likelihoods_dla"][4, 2])
        np.testing.assert_array_equal(f["MAP_z_dlas"][()].T, out["MAP_z_dlas"])    # qso_loader.py:107-109
        np.testing.assert_array_equal(f["model_posteriors"][()].T, out["model_posteriors"])
        assert f["base_sample_inds"].dtype == np.dtype("<u4") and f["base_sample_inds"].shape == (2, 400, 6)
    back = io.load_processed_qsos(path)
    np.testing.assert_array_equal(back["base_sample_inds"], out["base_sample_inds"])
    # replaying the saved indices reproduces the run (what "replaying a reference file" means)
    again = gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p,
                                                   base_sample_inds=back["base_sample_inds"])
    np.testing.assert_array_equal(again["sample_log_likelihoods_dla"], out["sample_log_likelihoods_dla"])
    info = dict(ras=np.linspace(0, 50, 6), decs=np.linspace(-5, 5, 6), plates=np.arange(3586, 3592),
                mjds=np.full(6, 55181), fiber_ids=np.arange(16, 22), thing_ids=np.arange(100000, 100006),
                z_qsos=z_qsos[test_ind], snrs=np.linspace(1, 6, 6))
    recs = catalog.generate_json_catalogue(out, info, str(d / "predictions_multi_DLAs.json"))
    assert json.load(open(d / "predictions_multi_DLAs.json")) == json.loads(json.dumps(recs))
    for r, mp in zip(recs, out["model_posteriors"]):
        assert 0 <= r["num_dlas"] <= 3 and len(r["dlas"]) == r["num_dlas"]
        occ = catalog.occams_model_posteriors(mp[None, :])[0]  # the loader's Occam factor (qso_loader.py:136)
        assert abs(r["p_no_dla"] - (occ[0] + occ[1])) < 1e-12
        for dla in r["dlas"]:
            assert r["min_z_dla"] <= dla["z_dla"] <= r["max_z_dla"] and 20.0 <= dla["log_nhi"] <= 23.0
