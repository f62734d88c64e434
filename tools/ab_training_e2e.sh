cd "$(dirname "$0")/.."
for r in 1 2 3; do for so in build/ab/*.so; do GPDLA_LIB_PATH=$PWD/$so python tools/bench_training.py --k 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$so', round(d['gpu_seconds_per_eval']*1e3,4), round(d['gpu_quasars_per_s']/1e6,3))"; done; done
