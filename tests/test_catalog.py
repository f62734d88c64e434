"""Host-side catalogue export (generate_ascii_catalog.m) on synthetic result tables."""
import numpy as np
import pytest

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


def test_json_catalogues(tmp_path):
    """qso_loader.py:1927-2087 on a hand-made multi-DLA result table: model order (no DLA, sub-DLA,
    1..max_dlas DLAs); with sub_dla the sub-DLA mass counts as "no DLA"."""
    import json
    md = 3
    mp = np.array([[0.7, 0.2, 0.1, 0.0, 0.0],     # no DLA
                   [0.1, 0.6, 0.2, 0.1, 0.0],     # sub-DLA most probable -> num_dlas 0
                   [0.05, 0.05, 0.2, 0.6, 0.1],   # two DLAs
                   [0.0, 0.1, 0.8, 0.1, 0.0]])    # one DLA
    nq = mp.shape[0]
    map_z = np.full((nq, md, md), np.nan)
    map_n = np.full((nq, md, md), np.nan)
    map_z[2, 1, :2], map_n[2, 1, :2] = [2.5, 2.9], [20.5, 21.25]
    map_z[3, 0, :1], map_n[3, 0, :1] = [3.1], [20.9]
    res = dict(model_posteriors=mp, p_dlas=1 - mp[:, 0] - mp[:, 1], p_no_dlas=mp[:, 0].copy(),
               min_z_dlas=np.full(nq, 2.0), max_z_dlas=np.full(nq, 3.5), MAP_z_dlas=map_z, MAP_log_nhis=map_n)
    info = dict(ras=np.arange(nq) * 1.5, decs=-np.arange(nq) * 0.5, plates=np.arange(3586, 3586 + nq),
                mjds=np.full(nq, 55181), fiber_ids=np.arange(16, 16 + nq), thing_ids=np.arange(1000, 1000 + nq),
                z_qsos=np.full(nq, 3.6), snrs=np.linspace(1, 4, nq))
    # occams_razor = 1: the posteriors as saved (the reference's factor is exercised below)
    cat = catalog.generate_json_catalogue(res, info, tmp_path / "predictions_multi_DLAs.json", occams_razor=1)
    assert [c["num_dlas"] for c in cat] == [0, 0, 2, 1]
    approx = lambda v: pytest.approx(v, rel=1e-15, abs=0)
    assert cat[0]["p_no_dla"] == approx(0.7 + 0.2) and cat[0]["max_model_posterior"] == cat[0]["p_no_dla"]
    assert cat[1]["max_model_posterior"] == cat[1]["p_no_dla"] == approx(0.1 + 0.6)
    assert cat[2]["max_model_posterior"] == approx(0.6)
    assert cat[2]["dlas"] == [{"log_nhi": 20.5, "z_dla": 2.5}, {"log_nhi": 21.25, "z_dla": 2.9}]
    assert cat[3]["dlas"] == [{"log_nhi": 20.9, "z_dla": 3.1}] and cat[0]["dlas"] == []
    assert set(cat[0]) == {"p_dla", "p_no_dla", "max_model_posterior", "num_dlas", "dlas", "min_z_dla",
                           "max_z_dla", "ra", "snr", "dec", "plate", "mjd", "fiber_id", "thing_id", "z_qso"}
    back = json.load(open(tmp_path / "predictions_multi_DLAs.json"))
    assert back == cat and isinstance(back[2]["plate"], int)
    sub = catalog.generate_sub_dla_catalogue(res, info, tmp_path / "sub.json", occams_razor=1)
    assert len(sub) == 1 and sub[0]["p_sub_dla"] == approx(0.6) and sub[0]["thing_id"] == 1001
    assert set(sub[0]) == {"p_sub_dla", "ra", "snr", "dec", "plate", "mjd", "fiber_id", "thing_id", "z_qso"}
    # without the sub-DLA model the model index IS the number of absorbers minus... (:1973 only)
    plain = catalog.generate_json_catalogue(res, info, sub_dla=False, occams_razor=1)
    assert plain[0]["p_no_dla"] == approx(0.7) and [c["num_dlas"] for c in plain] == [0, 1, 3, 2]


def test_json_catalogue_applies_the_loaders_occam_factor_and_nan_drop(tmp_path):
    """the reference's catalogue methods run on QSOLoader's view of the file --
    _occams_model_posteriors with occams_razor = 10000 (qso_loader.py:136-138, 235-257) and the
    all-NaN posterior rows dropped (:147-170).  The arithmetic of those lines is restated here
    independently (in-place division, tiled normalisation, boolean masks) and compared record by
    record."""
    import json
    md, sub = 3, 1
    mp = np.array([[0.7, 0.2, 0.1, 0.0, 0.0],
                   [1e-4, 0.3, 0.5, 0.2 - 1e-4, 0.0],
                   [np.nan] * 5,                         # a quasar the sweep skipped
                   [1e-7, 0.9, 0.05, 0.05 - 1e-7, 0.0],  # sub-DLA wins even after the penalty
                   [2e-5, 0.0, 0.2, 0.7, 0.1 - 2e-5]])
    nq = mp.shape[0]
    rng = np.random.default_rng(3)
    map_z, map_n = rng.uniform(2, 3, (nq, md, md)), rng.uniform(20, 22, (nq, md, md))
    res = dict(model_posteriors=mp, p_dlas=mp[:, 2:].sum(1), p_no_dlas=mp[:, 0].copy(),
               min_z_dlas=np.linspace(2.0, 2.4, nq), max_z_dlas=np.linspace(3.0, 3.4, nq),
               MAP_z_dlas=map_z, MAP_log_nhis=map_n)
    info = dict(ras=np.arange(nq) * 1.5, decs=-np.arange(nq) * 0.5, plates=np.arange(3586, 3586 + nq),
                mjds=np.full(nq, 55181), fiber_ids=np.arange(16, 16 + nq), thing_ids=np.arange(1000, 1000 + nq),
                z_qsos=np.full(nq, 3.6), snrs=np.linspace(1, 4, nq))
    # --- the loader's arithmetic, restated: qso_loader.py:136-170
    ref = mp.copy()
    ref[:, 1:] = ref[:, 1:] / 10000                                             # :247
    norm = np.sum(ref, axis=1)[:, None] * np.ones(ref.shape[1])[None, :]        # :250
    ref = ref / norm                                                            # :252
    p_dlas, p_no = ref[:, 1 + sub:].sum(axis=1), ref[:, :1 + sub].sum(axis=1)   # :137-138
    idx = np.argmax(ref, axis=1)                                                # :143
    nan = np.isnan(ref[np.arange(nq), idx])                                     # :144-147
    assert nan.tolist() == [False, False, True, False, False]
    ref, p_dlas, p_no, idx = ref[~nan], p_dlas[~nan], p_no[~nan], idx[~nan]
    kept = np.flatnonzero(~nan)
    num = idx - sub
    num[num < 0] = 0
    cat = catalog.generate_json_catalogue(res, info, tmp_path / "c.json")
    assert len(cat) == 4 and [c["thing_id"] for c in cat] == [1000, 1001, 1003, 1004]
    for r, c in enumerate(cat):
        q = kept[r]
        assert c["p_dla"] == p_dlas[r] and c["p_no_dla"] == p_no[r] and c["num_dlas"] == num[r]
        top = p_no[r] if idx[r] < 1 + sub else ref[r].max()                      # :1979-1980
        assert c["max_model_posterior"] == top
        assert c["min_z_dla"] == res["min_z_dlas"][q] and c["plate"] == 3586 + q
        assert c["dlas"] == [{"log_nhi": map_n[q, num[r] - 1, j], "z_dla": map_z[q, num[r] - 1, j]}
                             for j in range(num[r])]
    assert [c["num_dlas"] for c in cat] == [0, 0, 0, 2]   # 1e-4 vs 0.5e-4: the penalty flips quasar 1 to "no DLA"
    json.load(open(tmp_path / "c.json"))                  # strict JSON: no NaN record is emitted
    assert "NaN" not in open(tmp_path / "c.json").read()
    sub_cat = catalog.generate_sub_dla_catalogue(res, info)
    assert [s["thing_id"] for s in sub_cat] == [1003] and sub_cat[0]["p_sub_dla"] == ref[2, 1]
    # the input table is not modified (the reference's helper divides in place)
    assert mp[0, 1] == 0.2
