#!/bin/bash
# Kernel trace of the training objective (tools/bench_training.py) at one rank:
#   bash tools/profile_training.sh <tag> <k>   -> gpurun_out/profiles/<tag>_training_k<k>_kernel_stats.csv
set -e
cd "$(dirname "$0")/.."
TAG=${1:-dev}
K=${2:-20}
OUT=gpurun_out/profiles
mkdir -p $OUT gpurun_out/prof_train_$K
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$K -o tr -- python3 tools/bench_training.py --k $K > gpurun_out/prof_train_$K/bench.log 2>&1
f=$(find gpurun_out/prof_train_$K -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_training_k${K}_kernel_stats.csv
cut -c1-140 $OUT/${TAG}_training_k${K}_kernel_stats.csv
