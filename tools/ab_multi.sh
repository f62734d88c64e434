#!/bin/bash
# Same-box A/B of library variants (build/ab/*.so) on the multi-DLA driver: see tools/ab.sh.
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for so in build/ab/*.so; do
    GPDLA_LIB_PATH=$PWD/$so python tools/bench_multi.py "$@" 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$so', round(d['gpu_ms_per_call'],2), round(d['value']/1e6,2))"
  done
done
