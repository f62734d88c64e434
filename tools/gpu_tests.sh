#!/bin/bash
# The GPU suite, once, in ONE process, with everything the process AND the HIP runtime wrote kept:
#   gpurun --timeout 1100 -- 'bash tools/gpu_tests.sh <tag> [pytest args]'
# -> gpurun_out/<tag>_tests.log (pytest's report), <tag>_tests.err (file descriptor 2, uncaptured: the
# runtime's own "Memory access fault by GPU node ... on address ..." line of a run that dies lands
# here -- pytest's default fd-level capture swallows it when the process aborts), <tag>_dmesg.txt.
# A failing or dying run is NOT repeated: read what it left, fix the cause, test once.
cd "$(dirname "$0")/.."
TAG=${1:-dev}
shift || true
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1050 python -m pytest tests -m gpu -x -q --capture=sys -p no:cacheprovider "$@" \
  > gpurun_out/${TAG}_tests.log 2> gpurun_out/${TAG}_tests.err
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then (dmesg 2>/dev/null || true) | tail -60 > gpurun_out/${TAG}_dmesg.txt; fi
tail -6 gpurun_out/${TAG}_tests.log
tail -25 gpurun_out/${TAG}_tests.err
exit $rc
