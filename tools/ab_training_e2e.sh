#!/bin/bash
# Same-box A/B of library variants (build/ab/*.so) on one training-objective evaluation, wall clock
# through the Python call (tools/bench_training.py): tools/ab_training_e2e.sh [k]
cd "$(dirname "$0")/.."
K=${1:-40}
for r in 1 2 3; do for so in build/ab/*.so; do GPDLA_LIB_PATH=$PWD/$so python tools/bench_training.py --k $K --no-cpu --reps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$so', 'k=$K', round(d['gpu_seconds_per_eval']*1e3,4), 'ms')"; done; done
