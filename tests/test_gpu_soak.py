"""The one-shot C entries own everything they create for the length of one call -- a context (three
streams, model and sample tables, scratch), batch slots with their events, two stage threads, staging
vectors.  Called in a loop, the way a MEX gateway or a file-by-file driver calls them
(process_qsos.m:88-233 once per catalogue chunk), nothing may accumulate: device memory, host
memory, threads and file descriptors are read before and after a few dozen calls of every entry,
including calls that fail half way."""
import os
import threading

import numpy as np
import pytest

import gp_dla_detection_amd as gp
from gp_dla_detection_amd import synthetic
from gp_dla_detection_amd.parameters import MultiParameters

pytestmark = pytest.mark.gpu


def rss_bytes() -> int:
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def device_free_bytes() -> int:
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def open_fds() -> int:
    return len(os.listdir("/proc/self/fd"))


def test_repeated_one_shot_calls_leave_nothing_behind():
    model = synthetic.make_model(20)
    samples = synthetic.make_samples(64)
    rng = np.random.default_rng(5)
    spectra = [synthetic.make_spectrum(7000 + i, int(rng.integers(40, 300)), model, mask_fraction=0.03) for i in range(24)]
    n = len(spectra)
    lp = (np.full(n, -1.0), np.full(n, -1.0))
    csr = gp.spectra_to_csr(spectra)
    p = MultiParameters(max_dlas=2)
    cat = synthetic.make_prior_catalog()
    z = np.array([s["z_qso"] for s in spectra])
    mlp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.3, 0.7, p)
    bad = np.ones((n, 1, 64), dtype=np.uint32)
    bad[n // 2, 0, 5] = 1000  # rejected when that block is launched: the pipeline unwinds mid-run

    def round_of_calls():
        a = gp.process_qsos(model, samples, spectra, log_priors=lp, max_quasars_per_batch=5, pipeline_slots=3)
        b = gp.process_qsos(model, samples, csr, log_priors=lp, max_quasars_per_batch=7, pipeline_slots=2)
        np.testing.assert_array_equal(a["sample_log_likelihoods_dla"], b["sample_log_likelihoods_dla"])
        gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, mlp, params=p, max_quasars_per_batch=6)
        with pytest.raises(Exception):
            gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, mlp, params=p, base_sample_inds=bad,
                                                   max_quasars_per_batch=4, pipeline_slots=3)
        return a

    first = round_of_calls()          # warm-up: code objects loaded, allocator pools and the line tables made
    round_of_calls()
    free0, rss0, fds0, thr0 = device_free_bytes(), rss_bytes(), open_fds(), threading.active_count()
    tasks0 = len(os.listdir("/proc/self/task"))
    rounds = 12
    for _ in range(rounds):
        out = round_of_calls()
    np.testing.assert_array_equal(out["sample_log_likelihoods_dla"], first["sample_log_likelihoods_dla"])
    free1, rss1, fds1 = device_free_bytes(), rss_bytes(), open_fds()
    tasks1 = len(os.listdir("/proc/self/task"))
    # 48 calls (12 of them failing): a context or a slot left behind is >= tens of MB on the device
    # (scratch, sample tables, 3 streams), a stage thread left behind is a task
    assert free0 - free1 < 32 << 20, f"device memory: {(free0 - free1) / 2**20:.1f} MiB fewer free after {rounds} rounds"
    assert rss1 - rss0 < 64 << 20, f"host RSS grew by {(rss1 - rss0) / 2**20:.1f} MiB over {rounds} rounds"
    assert fds1 - fds0 <= 4, f"file descriptors: {fds0} -> {fds1}"
    assert tasks1 - tasks0 <= 2 and threading.active_count() == thr0, f"threads: {tasks0} -> {tasks1}"


def test_concurrent_one_shot_calls_from_several_threads():
    """no global state, functions thread-safe.  Three host threads call the one-shot
    entries at the same time on the same device (each call owns its context, streams and slots; the
    per-device line tables are made once under a lock; the last error is per thread): every result is
    bit-equal to the same call made alone, and a failing call in one thread leaves the others alone."""
    model = synthetic.make_model(20)
    samples = synthetic.make_samples(96)
    rng = np.random.default_rng(9)
    sets = [[synthetic.make_spectrum(8000 + 100 * t + i, int(rng.integers(60, 400)), model, mask_fraction=0.02)
             for i in range(18)] for t in range(3)]
    p = MultiParameters(max_dlas=2)
    cat = synthetic.make_prior_catalog()

    def priors(spectra):
        n = len(spectra)
        return (np.full(n, -1.0), np.full(n, -1.0))

    def multi_priors(spectra):
        z = np.array([s["z_qso"] for s in spectra])
        return gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.3, 0.7, p)

    def job(t):
        sp = sets[t]
        if t == 0:
            return gp.process_qsos(model, samples, sp, log_priors=priors(sp), max_quasars_per_batch=4)
        if t == 1:
            return gp.process_qsos(model, samples, gp.spectra_to_csr(sp), log_priors=priors(sp), max_quasars_per_batch=5)
        return gp.process_qsos_multiple_dlas_meanflux(model, samples, sp, multi_priors(sp), params=p, max_quasars_per_batch=6)

    alone = [job(t) for t in range(3)]
    bad = np.ones((18, 1, 96), dtype=np.uint32)
    bad[9, 0, 0] = 5000
    for _ in range(4):
        got, errors = [None] * 4, [None] * 4

        def run(t):
            try:
                if t < 3:
                    got[t] = job(t)
                else:  # a fourth caller whose call fails in the middle of its pipeline
                    gp.process_qsos_multiple_dlas_meanflux(model, samples, sets[2], multi_priors(sets[2]), params=p,
                                                           base_sample_inds=bad, max_quasars_per_batch=3)
            except Exception as e:  # noqa: BLE001
                errors[t] = e
        threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(120)
            assert not th.is_alive()
        assert errors[:3] == [None] * 3, errors
        assert errors[3] is not None and "exceeds num_dla_samples" in str(errors[3])
        for t in range(3):
            for key, want in alone[t].items():
                if isinstance(want, np.ndarray):
                    np.testing.assert_array_equal(got[t][key], want, err_msg=f"thread {t}: {key}")
