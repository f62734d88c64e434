#!/bin/bash
# PMC passes over the training objective (tools/bench_training.py), one counter group per pass with
# nothing but --kernel-trace beside it:
#   bash tools/pmc_training.sh <tag> <k>  ->  gpurun_out/profiles/<tag>_training_k<k>_pmc.json
# (per kernel: launches and per-launch means of every counter; FETCH_SIZE doubled for gfx950 as
# MI355X_MICROARCH.md's HBM section prescribes, WRITE_SIZE as read; both in KB per launch).
set -e
set -o pipefail
TAG=${1:-dev}
K=${2:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT/profiles
cd /tmp && export TMPDIR=/tmp
for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $group | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/pmct_${TAG}_k${K}_$name
  # a pass that fails or hangs ends the script: no further GPU step after a killed one
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/pmct_${TAG}_k${K}_$name -- \
    python3 $ROOT/tools/bench_training.py --k $K --reps 2 --no-cpu > $OUT/pmct_${TAG}_k${K}_$name.log 2>&1
  echo "pmc pass done: $group"
done
python3 $ROOT/tools/pmc_kernels_to_json.py $OUT "pmct_${TAG}_k${K}_" k_train > $OUT/profiles/${TAG}_training_k${K}_pmc.json
head -c 3000 $OUT/profiles/${TAG}_training_k${K}_pmc.json
