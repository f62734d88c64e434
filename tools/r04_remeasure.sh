# Re-measures what the round-4 epilogue change (factor_paired) moved: run ON THE GPU BOX from the repo root.
set -o pipefail
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 tools/dpp_f64_probe.hip -o /tmp/dpp_probe 2>/dev/null && timeout -k 10 60 /tmp/dpp_probe > gpurun_out/dpp_f64_probe.txt && echo "probe ok"
bash tools/profile.sh r04 configs1 > gpurun_out/r04_profile.log 2>&1 && echo "headline profile ok"
bash tools/profile.sh r04_k40 configs1 --k 40 --spectra 256 > gpurun_out/r04_k40_profile.log 2>&1 && echo "k40 profile ok"
bash tools/profile_multi.sh r04_k40 --k 40 --max-dlas 3 > gpurun_out/r04_profile_multi_k40.log 2>&1 && echo "multi k40 profile ok"
bash tools/run_all_configs.sh r04 > gpurun_out/r04_run_all.log 2>&1 && echo "configs ok"
python bench.py --workload dr12q-mix --spectra 20358 --k 40 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_dr12q_shard_k40.json 2> gpurun_out/r04_dr12q_shard_k40.err && echo "shard k40 ok"
echo done
