#!/bin/bash
# Diagnostic: run the GPU suite up to N times under tools/segv_trace.c, stopping at the first failure.
# usage (GPU box): tools/repeat_suite.sh N
set -u
gcc -shared -fPIC -O1 -o /tmp/segv_trace.so tools/segv_trace.c || exit 1
mkdir -p gpurun_out
for i in $(seq 1 "${1:-5}"); do
  LD_PRELOAD=/tmp/segv_trace.so python -m pytest tests -m gpu -x -q -s -p no:faulthandler > gpurun_out/repeat_suite_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/repeat_suite_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then exit $rc; fi
  rm -f gpurun_out/repeat_suite_$i.log
done
