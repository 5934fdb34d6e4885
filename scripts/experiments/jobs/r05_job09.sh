#!/bin/bash
# round 5 job 9: after the per-peer wait / one-wave-per-peer hop measurement: the multi-rank tests and the bench contract tests; 3- and 8-rank shared lines
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r05_job09_multirank.log 2>&1; tail -4 gpurun_out/r05_job09_multirank.log
SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus 3 --workload C3 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job09_bench_3ranks.json 2> gpurun_out/r05_job09_bench_3ranks.err; grep -E "exchange transport|flag hop" gpurun_out/r05_job09_bench_3ranks.err | cut -c1-260
