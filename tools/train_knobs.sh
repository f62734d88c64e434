#!/bin/bash
# Diagnostic: time of one training evaluation for several split settings (GPDLA_TRAIN_SPLITS="H,H2,GS";
# the switch exists in libgpdla_legacy.so only -- python -c "from gp_dla_detection_amd import _lib; _lib.build_legacy()").
#   tools/train_knobs.sh [k] ["H,H2,GS" ...]
export GPDLA_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/gp_dla_detection_amd/csrc/libgpdla_legacy.so
cd "$(dirname "$0")/.."
K=${1:-20}
shift || true
SETS=${@:-"6,24,24 5,24,24 7,24,24 6,16,24 6,32,24 6,48,24 6,24,16 6,24,32 6,24,48 6,12,12"}
for round in 1 2; do
for s in $SETS; do
  ms=$(GPDLA_TRAIN_SPLITS=$s python3 tools/bench_training.py --k $K --no-cpu --reps 20 2>/dev/null | python3 -c "import sys,json; print(round(1e3*json.loads(sys.stdin.readline())['gpu_seconds_per_eval'],4))")
  echo "k=$K splits $s: $ms ms"
done
done
