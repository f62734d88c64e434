#!/bin/bash
# Diagnostic: build ablated variants of the sweep kernel and time each (run on the GPU box).
# The GPDLA_ABLATE_* switches live in k_sweep (pre-expanded records), so the bench is pointed at it
# (GPDLA_EXPANDED_RECORDS=1); k_sweep_slim, the production kernel for k <= 20, shares its K-step.
# Usage: tools/ablate.sh [bench args]   -> gpurun_out/ablate.txt
set -e
set -o pipefail
cd "$(dirname "$0")/.."
SRC=gp_dla_detection_amd/csrc/gpdla.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -no-hip-rt -Wno-inline-asm -DGPDLA_WITH_LEGACY"
mkdir -p gpurun_out /tmp/ablate
: > gpurun_out/ablate.txt
for v in ${GPDLA_ABLATE_SET:-BASE NOBARRIER NOSLOW "NOSLOW -DGPDLA_ABLATE_NOBARRIER" NOEPI NOVOIGT NOMFMA "NOSLOW -DGPDLA_ABLATE_NOEPI" "NOSLOW -DGPDLA_ABLATE_NOEPI -DGPDLA_ABLATE_NOVOIGT" "NOSLOW -DGPDLA_ABLATE_NOEPI -DGPDLA_ABLATE_NOMFMA"}; do
  name=$(echo "$v" | tr -d ' ' | sed 's/-DGPDLA_ABLATE_/+/g')
  hipcc $FLAGS -DGPDLA_ABLATE_$v $SRC -o /tmp/ablate/lib_$name.so
  ms=$(GPDLA_EXPANDED_RECORDS=1 GPDLA_LIB_PATH=/tmp/ablate/lib_$name.so python3 bench.py --no-cpu-baseline --no-mix-rider --steps 2 --warmup 1 "$@" | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['roofline']['kernel_ms'])")
  echo "$name kernel_ms=$ms" | tee -a gpurun_out/ablate.txt
done
