#!/bin/bash
# Diagnostic: time the sweep kernel built with different -D flag sets (run on the GPU box).
# GPDLA_PROBE_TARGET=training times tools/bench_training.py instead of the sweep.
# usage: tools/flag_probe.sh "<flags A>" "<flags B>" ...   (an empty string = the shipped build)
set -e
set -o pipefail
cd "$(dirname "$0")/.."
SRC=gp_dla_detection_amd/csrc/gpdla.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -no-hip-rt -Wno-inline-asm"
mkdir -p /tmp/ablate gpurun_out
for v in "$@"; do
  if ! hipcc $FLAGS $v $SRC -o /tmp/ablate/lib_flag.so 2>/tmp/ablate/flag_err.txt; then echo "flags '$v' did not compile: $(tail -1 /tmp/ablate/flag_err.txt)" | tee -a gpurun_out/flag_probe.txt; continue; fi
  if [ "${GPDLA_PROBE_TARGET:-sweep}" = training ]; then
    ms=$(GPDLA_LIB_PATH=/tmp/ablate/lib_flag.so python3 tools/bench_training.py 2>/dev/null | python3 -c "import sys,json; print(1e3*json.loads(sys.stdin.readline())['gpu_seconds_per_eval'])")
  else
    ms=$(GPDLA_LIB_PATH=/tmp/ablate/lib_flag.so python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --spectra ${GPDLA_PROBE_SPECTRA:-512} | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['roofline']['kernel_ms'])")
  fi
  echo "flags '$v' ${GPDLA_PROBE_TARGET:-sweep}_ms=$ms" | tee -a gpurun_out/flag_probe.txt
done
