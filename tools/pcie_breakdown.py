#!/usr/bin/env python3
"""Where the PCIe-inclusive run of process_qsos spends what the resident sweep does not (GPU box):
times the stages of api.run_pipeline for 2048 quasars of the headline shape and prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_dla_detection_amd as gp  # noqa: E402
from gp_dla_detection_amd import api, synthetic  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
per_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = synthetic.make_model(20)
samples = synthetic.make_samples(10000)
base = synthetic.make_spectra(256, 1500, model)
spectra = [base[i % 256] for i in range(nq)]
lp = (np.full(nq, np.log(0.9)), np.full(nq, np.log(0.1)))
gp.process_qsos(model, samples, spectra[:64], log_priors=(lp[0][:64], lp[1][:64]))  # warm-up

marks = {}
t0 = time.perf_counter()
ctx = gp.Context(0)
ctx.set_model(model)
ctx.set_samples(samples)
marks["context"] = time.perf_counter() - t0
blocks = [(lo, min(lo + per_batch, nq)) for lo in range(0, nq, per_batch)]
out = gp.Batch.empty_results(nq, 10000)
marks["alloc_out"] = time.perf_counter() - t0 - marks["context"]
ev = []


def inputs(i):
    a = time.perf_counter()
    lo, hi = blocks[i]
    csr = api.spectra_to_csr(spectra[lo:hi])
    ev.append(("pack", i, a - t0, time.perf_counter() - t0))
    return csr, lp[0][lo:hi], lp[1][lo:hi]


def process(i, batch):
    a = time.perf_counter()
    batch.process()
    ev.append(("launch", i, a - t0, time.perf_counter() - t0))


def download(i, batch):
    a = time.perf_counter()
    batch.download(True, out, blocks[i][0])
    ev.append(("download", i, a - t0, time.perf_counter() - t0))


orig_upload, orig_reload = gp.Context.upload, gp.Batch.reload


def upload(self, *a, **k):
    s = time.perf_counter()
    r = orig_upload(self, *a, **k)
    ev.append(("upload", -1, s - t0, time.perf_counter() - t0))
    return r


def reload(self, *a, **k):
    s = time.perf_counter()
    r = orig_reload(self, *a, **k)
    ev.append(("reload", -1, s - t0, time.perf_counter() - t0))
    return r


gp.Context.upload, gp.Batch.reload = upload, reload
api.run_pipeline(ctx, len(blocks), inputs, process, download, 3)
marks["pipeline_end"] = time.perf_counter() - t0
ctx.close()
marks["total"] = time.perf_counter() - t0
first_launch = min(e[2] for e in ev if e[0] == "launch")
last_dl = [e for e in ev if e[0] == "download"][-1]
print(json.dumps(dict(nq=nq, per_batch=per_batch, marks=marks, first_launch_at=first_launch,
                      last_download=(last_dl[2], last_dl[3]),
                      rate=nq * 10000 / marks["total"],
                      events=[(k, i, round(a, 4), round(b, 4)) for k, i, a, b in sorted(ev, key=lambda e: e[2])][:40])))
