mkdir -p gpurun_out/final_k
python bench.py > gpurun_out/final_k/bench.json 2> gpurun_out/final_k/bench.err
python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/final_k/bench_untimed.json 2> /dev/null
bash scripts/profile_gpu.sh r01k --time-all-kernels > gpurun_out/final_k/profile.log 2>&1
cut -c1-300 gpurun_out/final_k/bench.json
