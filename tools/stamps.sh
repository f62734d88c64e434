#!/bin/bash
# Diagnostic: build the sweep kernel with s_memtime stamps (-DGPDLA_STAMP) and print where a wave's
# cycles go, segment by segment (run on the GPU box).  Shares only: the stamps forbid overlaps.
set -e
set -o pipefail
cd "$(dirname "$0")/.."
SRC=gp_dla_detection_amd/csrc/gpdla.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -no-hip-rt -Wno-inline-asm"
mkdir -p gpurun_out /tmp/ablate
hipcc $FLAGS -DGPDLA_STAMP $SRC -o /tmp/ablate/lib_stamp.so
GPDLA_EXPANDED_RECORDS=1 GPDLA_LIB_PATH=/tmp/ablate/lib_stamp.so python3 - "$@" <<'PY' | tee gpurun_out/stamps.txt
import ctypes, json, subprocess, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gp_dla_detection_amd as gp
from gp_dla_detection_amd import _lib, synthetic
lib = _lib.load()
nq = 256
model = synthetic.make_model(k=20)
samples = synthetic.make_samples(10000)
spectra = synthetic.make_spectra(nq, 1500, model)
out = gp.process_qsos(model, samples, spectra, log_priors=(np.full(nq, -1.0), np.full(nq, -1.0)))
buf = (ctypes.c_ulonglong * 8)()
lib.gpdla_debug_stamps.restype = ctypes.c_int
assert lib.gpdla_debug_stamps(buf) == 0
v = np.array(list(buf), dtype=np.float64)
names = ["raw profile (requests + wing tier)", "accurate tier", "fragments/ring/broadening/weights",
         "MFMA burst", "chunk drain + barrier", "prefetch issue", "epilogue", "-"]
waves = nq * 626
steps = 375
tot = v.sum()
for n, x in zip(names, v):
    print(f"{n:40s} {x / tot * 100:6.2f} %   {x / waves / steps:8.1f} wave-cycles per K-step")
print(f"total {tot / waves / steps:.1f} wave-cycles per K-step (stamped build)")
PY
