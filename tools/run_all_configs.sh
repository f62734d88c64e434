#!/bin/bash
# Reproduces, in one go on the GPU box, every throughput figure DESIGN.md quotes (BASELINE configs
# 2, 4, 5 and the training objective) and collects the JSON lines in gpurun_out/r01_configs.jsonl:
#   gpurun --timeout 1100 -- 'bash tools/run_all_configs.sh'
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r01_configs.jsonl
mkdir -p gpurun_out
: > $OUT
run() { echo "== $*"; "$@" 2>/dev/null | tail -1 | tee -a $OUT | cut -c1-160; }
run python3 bench.py --no-cpu-baseline                                          # config 2 shape, fp64, k = 20
run python3 bench.py --no-cpu-baseline --k 40 --spectra 256                     # fp64, k = 40
run python3 bench.py --no-cpu-baseline --contraction f32                        # config 5 variant, k = 20
run python3 bench.py --no-cpu-baseline --contraction f32 --k 40 --spectra 256   # config 5: fp32 contraction, k = 40
run python3 tools/bench_multi.py --spectra 64                                   # config 4: multi-DLA driver
run python3 tools/bench_training.py                                             # N3: training objective
echo "wrote $OUT"
