#!/bin/bash
# Round-5 measurements, ON THE GPU BOX from the repo root, in parts (a gpurun call is at most 20 minutes):
#   bash tools/r05_measure.sh headline | k40mix | multi | training | configs
# Every summary lands under gpurun_out/profiles/ (copy into profiles/ to commit).  The last step of
# every part is the hash guard: a PMC summary whose lib_sha256 is not the library's in this tree is
# renamed *.STALE (traffic is reported only on a match).
set -o pipefail
cd "$(dirname "$0")/.."
PART=${1:-headline}
mkdir -p gpurun_out/profiles
case $PART in
  headline)
    bash tools/profile.sh r05 configs1 > gpurun_out/r05_profile.log 2>&1 && echo "headline profile ok"
    # bench.py reports roofline.traffic only from a PMC summary of the library it loads: hand it this one
    cp gpurun_out/profiles/r05_pmc.json profiles/pmc_latest.json
    python bench.py --steps 20 --warmup 5 > gpurun_out/profiles/r05_bench_final.json 2> gpurun_out/r05_bench_final.err && echo "bench ok"
    python bench.py --pcie --pcie-c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/profiles/r05_bench_pcie.json 2> gpurun_out/r05_bench_pcie.err && echo "pcie ok"
    ;;
  k40mix)
    bash tools/profile.sh r05_k40 configs1 --k 40 --spectra 256 > gpurun_out/r05_k40_profile.log 2>&1 && echo "k40 profile ok"
    bash tools/profile.sh r05_mix dr12q-mix > gpurun_out/r05_mix_profile.log 2>&1 && echo "mix profile ok"
    ;;
  multi)
    bash tools/profile_multi.sh r05 > gpurun_out/r05_profile_multi.log 2>&1 && echo "multi profile ok"
    bash tools/profile_multi.sh r05_k40 --k 40 --max-dlas 3 > gpurun_out/r05_profile_multi_k40.log 2>&1 && echo "multi k40 profile ok"
    bash tools/pmc_multi.sh r05 > gpurun_out/r05_pmc_multi.log 2>&1 && cp gpurun_out/pmc_multi_r05.txt gpurun_out/profiles/r05_pmc_multi.txt && echo "multi pmc ok"
    ;;
  training)
    for K in 20 40; do
      bash tools/profile_training.sh r05 $K > gpurun_out/r05_proft$K.log 2>&1 && echo "training trace k=$K ok"
      bash tools/pmc_training.sh r05 $K > gpurun_out/r05_pmct$K.log 2>&1 && echo "training pmc k=$K ok"
      python tools/bench_training.py --k $K 2>/dev/null | tail -1 >> gpurun_out/profiles/r05_training.jsonl
    done
    ;;
  configs)
    bash tools/run_all_configs.sh r05 > gpurun_out/r05_run_all.log 2>&1 && echo "configs ok"
    python tools/bench_run_files.py 20358 10000 > gpurun_out/profiles/r05_run_files_shard.json 2> gpurun_out/r05_run_files.err && echo "run files shard ok"
    python tools/bench_run_files.py 768 10000 0 multi > gpurun_out/profiles/r05_run_files_multi.json 2>> gpurun_out/r05_run_files.err && echo "run files multi ok"
    python bench.py --workload dr12q-shard --total-spectra 20358 --steps 3 > gpurun_out/profiles/r05_bench_shard_20358.json 2> gpurun_out/r05_bench_shard.err && echo "shard bench ok"
    ;;
esac
python3 - <<'PY'
import glob, hashlib, json, os
sha = hashlib.sha256(open("gp_dla_detection_amd/csrc/libgpdla.so", "rb").read()).hexdigest()
for p in glob.glob("gpurun_out/profiles/r05*pmc*.json"):
    try:
        got = json.load(open(p)).get("lib_sha256")
    except ValueError:
        continue
    if got and got != sha:
        os.rename(p, p + ".STALE")
        print("STALE (measured on another library):", p)
print("library", sha[:16])
PY
echo "part $PART done"
