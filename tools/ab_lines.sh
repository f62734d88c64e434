#!/bin/bash
# ON THE GPU BOX: the sweep at a run-time line count, slim records (k_sweep_slim<0> / k_sweep_split_slim<0,0>,
# product library) against the pre-expanded records (k_sweep<...,0> / k_sweep_split<0>, libgpdla_legacy.so),
# same box, alternating:   bash tools/ab_lines.sh [K] [line counts ...]
cd "$(dirname "$0")/.."
LEG=$PWD/gp_dla_detection_amd/csrc/libgpdla_legacy.so
K=${1:-20}
shift || true
N=$([ $K -gt 20 ] && echo 64 || echo 200)
for L in ${@:-31 5 1}; do
  for rep in 1 2; do
    python bench.py --k $K --num-lines $L --spectra $N --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k $K lines $L slim    ', round(d['roofline']['kernel_ms'],2), 'ms')"
    GPDLA_LIB_PATH=$LEG GPDLA_EXPANDED_RECORDS=1 python bench.py --k $K --num-lines $L --spectra $N --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k $K lines $L expanded', round(d['roofline']['kernel_ms'],2), 'ms')"
  done
done
