#!/bin/bash
# PMC passes over the multi-DLA driver (tools/bench_multi.py): per-kernel counter sums ->
# gpurun_out/pmc_multi_<tag>.txt.  Run on the GPU box from the repo root.
set -e
set -o pipefail
TAG=${1:-r02}
shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (FETCH_SIZE and WRITE_SIZE each in a pass of their own: derived TCC counters, in KiB per launch;
# FETCH_SIZE is doubled for gfx950 by whoever reads the sums, as in tools/pmc_to_json.py)
for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  name=$(echo $group | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/pmcm_${TAG}_$name -- \
    python3 $ROOT/tools/bench_multi.py --steps 1 "$@" > $OUT/pmcm_${TAG}_$name.log 2>&1
done
python3 - $OUT $TAG <<'PY' | tee $OUT/pmc_multi_$TAG.txt
import csv, glob, os, sys
out, tag = sys.argv[1:3]
acc = {}
for path in glob.glob(os.path.join(out, f"pmcm_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:30s} launches {len(v):3d}  sum {sum(v):.4g}")
PY
