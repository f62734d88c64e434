"""Host-side catalogue export (generate_ascii_catalog.m) on synthetic result tables."""
import numpy as np

from gp_dla_detection_amd import catalog


def fake_results(nq=3, S=5):
    rng = np.random.default_rng(1)
    sll = rng.normal(size=(nq, S))
    sll[1, 2] = np.nan
    sll[2] = np.nan
    return dict(sample_log_likelihoods_dla=sll, min_z_dlas=np.array([2.0, 2.1, 2.2]),
                max_z_dlas=np.array([3.0, 3.1, 3.2]), log_priors_no_dla=np.log([0.9, 0.8, 0.7]),
                log_priors_dla=np.log([0.1, 0.2, 0.3]), log_likelihoods_no_dla=np.array([-1500.25, 3.5e4, np.nan]),
                log_likelihoods_dla=np.array([-1490.5, 3.6e4, np.nan]),
                model_posteriors=np.array([[0.25, 0.75], [1e-12, 1.0], [np.nan, np.nan]]))


def test_map_estimates():
    r = fake_results()
    s = dict(offset_samples=np.linspace(0.1, 0.9, 5), log_nhi_samples=np.linspace(20, 22, 5))
    z, n, ind = catalog.map_estimates(r, s)
    assert ind[0] == np.argmax(r["sample_log_likelihoods_dla"][0])
    assert ind[1] == np.nanargmax(r["sample_log_likelihoods_dla"][1])
    assert ind[2] == 0  # all-NaN row: first sample, as MATLAB's nanmax
    assert z[0] == 2.0 + 1.0 * s["offset_samples"][ind[0]] and n[1] == s["log_nhi_samples"][ind[1]]


def test_ascii_formats(tmp_path):
    r = fake_results()
    s = dict(offset_samples=np.linspace(0.1, 0.9, 5), log_nhi_samples=np.linspace(20, 22, 5))
    catalog.write_dla_samples(tmp_path / "s.dat", s)
    assert open(tmp_path / "s.dat").readline() == "0.100000 20.000000\n"
    catalog.write_results(tmp_path / "r.dat", [12345, 7, 999999999], r, s)
    lines = open(tmp_path / "r.dat").read().splitlines()
    f = lines[0].split()
    assert f[0] == "000012345" and f[1] == "2.0000" and f[2] == "3.0000"
    assert f[5] == "-1.50025e+03" and f[7] == "2.50000e-001" and f[8] == "7.50000e-001"
    assert lines[1].split()[7] == "1.00000e-012"  # three-digit exponent (:68-71)
    assert len(lines) == 3
