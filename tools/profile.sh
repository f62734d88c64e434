#!/bin/bash
# Collects the profiles the bench line is judged against.  Run ON THE GPU BOX, from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/profile.sh r02 [configs1|dr12q-mix] [extra bench.py flags]'
# Writes under gpurun_out/ (scratch) and copies the summaries into gpurun_out/profiles/ so they
# travel back:  <tag>_kernel_stats.csv, <tag>_pmc.json (copy them into profiles/ to commit).
# Counters are collected in separate --pmc passes with nothing but --kernel-trace beside them.
set -e
set -o pipefail
TAG=${1:-r02}
WORKLOAD=${2:-configs1}
shift || true
shift || true
EXTRA="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- \
  python3 $ROOT/bench.py --workload $WORKLOAD --steps 3 --warmup 1 --no-cpu-baseline --no-mix-rider $EXTRA > $OUT/prof_$TAG.log 2>&1
cp $(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/profiles/${TAG}_kernel_stats.csv
grep "^{\"metric\"" $OUT/prof_$TAG.log | tail -1 > $OUT/profiles/${TAG}_bench_under_rocprof.json
# one derived TCC counter per pass (FETCH_SIZE + WRITE_SIZE together exceed the hardware's counters)
for group in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $group | tr ' ' '_' | cut -c1-40)
  # a pass that fails or hangs ends the script: no further GPU step after a killed one
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/pmc_${TAG}_$name -- \
    python3 $ROOT/bench.py --workload $WORKLOAD --steps 1 --warmup 0 --no-cpu-baseline --no-mix-rider $EXTRA > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc pass done: $group"
done
# where the cycles that are neither MFMA nor VALU go, and the direct check that the
# fp64 MFMA never co-executes with VALU work (SQ_VALU_MFMA_COEXEC_CYCLES)
for group in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_IFETCH SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $group | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/pmc_${TAG}_$name -- \
    python3 $ROOT/bench.py --workload $WORKLOAD --steps 1 --warmup 0 --no-cpu-baseline --no-mix-rider $EXTRA > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc pass done: $group"
done
python3 $ROOT/tools/pmc_to_json.py $OUT $TAG $WORKLOAD > $OUT/profiles/${TAG}_pmc.json
cat $OUT/profiles/${TAG}_pmc.json
