#!/bin/bash
# Per-kernel times of the multi-DLA driver for every library variant in build/ab/*.so on ONE box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for so in build/ab/*.so; do
  name=$(basename $so .so)
  d=gpurun_out/abmu_$name
  rm -rf $d && mkdir -p $d
  GPDLA_LIB_PATH=$PWD/$so rocprofv3 --kernel-trace --stats --output-format csv -d $d -o mu -- python3 tools/bench_multi.py "$@" > $d/bench.log 2>&1 || { echo "$name FAILED"; tail -3 $d/bench.log; continue; }
  echo "== $name"
  python3 - "$d" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gpdla" in r["Name"] and float(r["Percentage"]) > 0.5:
        print(f"   {r['Name'][:60]:62s} avg {float(r['AverageNs'])/1e6:8.3f} ms  min {float(r['MinNs'])/1e6:8.3f}  x{r['Calls']}")
PY
done
