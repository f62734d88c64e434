#!/bin/bash
# Kernel trace of the multi-DLA driver (tools/bench_multi.py, BASELINE config 4):
#   bash tools/profile_multi.sh <tag> [bench_multi args]  -> gpurun_out/profiles/<tag>_multi_kernel_stats.csv
set -e
cd "$(dirname "$0")/.."
TAG=${1:-dev}
shift || true
OUT=gpurun_out/profiles
D=gpurun_out/prof_multi_$TAG
rm -rf $D && mkdir -p $OUT $D
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o mu -- python3 tools/bench_multi.py "$@" > $D/bench.log 2>&1
f=$(find $D -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_multi_kernel_stats.csv
tail -1 $D/bench.log
cut -c1-150 $OUT/${TAG}_multi_kernel_stats.csv | head -8
