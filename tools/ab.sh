#!/bin/bash
# A/B timing of library variants on ONE GPU box (box-to-box spread is ~0.5 %, more than most
# micro-changes): tools/ab.sh [bench args] -- runs bench.py alternately on every build/ab/*.so,
# three rounds, and prints the sweep kernel's average launch time per variant.
# Make variants with tools/ab_build.sh <name> (compiles the working tree into build/ab/<name>.so).
cd "$(dirname "$0")/.."
for round in ${AB_ROUNDS:-1 2 3}; do
  for so in build/ab/*.so; do
    GPDLA_LIB_PATH=$PWD/$so python bench.py --no-cpu-baseline --no-mix-rider "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$so', round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4))"
  done
done
