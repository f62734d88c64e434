#!/bin/bash
# Reproduces, in one go on the GPU box, every throughput figure DESIGN.md quotes (BASELINE configs
# 2, 4, 5, the DR12Q length mix, k = 40 and the training objective) and collects the JSON lines:
#   gpurun --timeout 1100 -- 'bash tools/run_all_configs.sh r02'   -> gpurun_out/<tag>_configs.jsonl
set -e
set -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_configs.jsonl
mkdir -p gpurun_out
: > $OUT
run() { echo "== $*"; "$@" 2>gpurun_out/${TAG}_configs.err | tail -1 | tee -a $OUT | cut -c1-200; }
run python3 bench.py --no-cpu-baseline --no-mix-rider                          # config 2 shape, fp64, k = 20
run python3 bench.py --no-cpu-baseline --workload dr12q-mix                     # DR12Q length mix, 5 % masked
run python3 bench.py --no-cpu-baseline --k 40 --spectra 256                     # fp64, k = 40 (k_sweep_split_slim)
GPDLA_LIB_PATH=$PWD/gp_dla_detection_amd/csrc/libgpdla_legacy.so GPDLA_EXPANDED_RECORDS=1 run python3 bench.py --no-cpu-baseline --k 40 --spectra 256   # the same on pre-expanded records (k_sweep_split)
run python3 bench.py --no-cpu-baseline --contraction f32                        # config 5 variant, k = 20
run python3 bench.py --no-cpu-baseline --contraction f32 --k 40 --spectra 256   # config 5: fp32 contraction, k = 40
run python3 tools/bench_multi.py --spectra 64                                   # config 4: multi-DLA driver, resident
run python3 tools/bench_multi.py --spectra 64 --k 40 --max-dlas 3               # config 4 shape at k = 40
run python3 tools/bench_training.py                                             # N3: training objective, k = 20
run python3 tools/bench_training.py --k 40                                      # N3: training objective, k = 40
echo "wrote $OUT"
