#!/bin/bash
# Per-kernel times of the training objective for every library variant in build/ab/*.so (made by
# tools/ab_build.sh), on ONE GPU box: tools/ab_training.sh [k]
cd "$(dirname "$0")/.."
K=${1:-20}
export TMPDIR=/tmp
for so in build/ab/*.so; do
  name=$(basename $so .so)
  d=gpurun_out/abtr_$name
  rm -rf $d && mkdir -p $d
  GPDLA_LIB_PATH=$PWD/$so rocprofv3 --kernel-trace --stats --output-format csv -d $d -o tr -- python3 tools/bench_training.py --k $K > $d/bench.log 2>&1 || { echo "$name FAILED"; tail -3 $d/bench.log; continue; }
  echo "== $name: $(grep '^{"metric"' $d/bench.log | tail -1 | python3 -c "import json,sys; print(round(1e3*json.loads(sys.stdin.readline())['gpu_seconds_per_eval'],4))") ms/eval (under the profiler)"
  python3 - "$d" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_train" in r["Name"]:
        print(f"   {r['Name'][:48]:50s} {float(r['AverageNs'])/1e3:8.1f} us x {int(r['Calls'])//6}")
PY
done
