set -o pipefail
cd "$(dirname "$0")/.."
bash tools/profile.sh r04_k40 configs1 --k 40 --spectra 256 > gpurun_out/r04_k40_profile.log 2>&1 && echo "k40 profile ok"
bash tools/profile.sh r04_mix dr12q-mix > gpurun_out/r04_mix_profile.log 2>&1 && echo "mix profile ok"
bash tools/profile_multi.sh r04 > gpurun_out/r04_profile_multi.log 2>&1 && echo "multi profile ok"
bash tools/profile_multi.sh r04_k40 --k 40 --max-dlas 3 > gpurun_out/r04_profile_multi_k40.log 2>&1 && echo "multi k40 profile ok"
python tools/bench_run_files.py 20358 10000 > gpurun_out/r04_run_files_shard.json 2> gpurun_out/r04_run_files_shard.err && echo "run files shard ok"
python tools/bench_run_files.py 4000 10000 > gpurun_out/r04_run_files_4000.json 2>> gpurun_out/r04_run_files_shard.err
python tools/bench_run_files.py 768 10000 0 multi > gpurun_out/r04_run_files_multi.json 2> gpurun_out/r04_run_files_multi.err && echo "run files multi ok"
python tools/bench_run_files.py 20358 10000 0 multi > gpurun_out/r04_run_files_multi_shard.json 2>> gpurun_out/r04_run_files_multi.err && echo "run files multi shard ok"
for i in 1 2; do python tools/bench_multi.py >> gpurun_out/r04_pretouch_ab.txt 2>/dev/null; GPDLA_NO_PROFILE_PRETOUCH=1 python tools/bench_multi.py >> gpurun_out/r04_pretouch_ab.txt 2>/dev/null; done
GPDLA_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 3 > gpurun_out/r04_bench_rehearsal.json 2> gpurun_out/r04_bench_rehearsal.err
python bench.py --workload dr12q-shard --total-spectra 20358 --steps 3 > gpurun_out/r04_bench_shard_20358.json 2> gpurun_out/r04_bench_shard.err
python bench.py --pcie --steps 5 --warmup 2 > gpurun_out/r04_bench_pcie.json 2> gpurun_out/r04_bench_pcie.err
echo done
