#!/bin/bash
# round 6 job 12: the resident loop's fall-back (a launch that gives up -> the step is taken again with separate sweeps, same bits), then the whole GPU suite
# the way the driver runs it (-x), smoke(), and the stored values an N > 1 bench line is held to (bench.py --write-episode-checks)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_resident.py -q > gpurun_out/r06_job12_resident.txt 2>&1; tail -5 gpurun_out/r06_job12_resident.txt
( time timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/r06_job12_suite.txt 2>&1; tail -24 gpurun_out/r06_job12_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 1 --write-episode-checks gpurun_out/c4_f20_episode_checks.json > gpurun_out/r06_job12_bench_checks.json 2> gpurun_out/r06_job12_bench_checks.err; tail -2 gpurun_out/r06_job12_bench_checks.err; cat gpurun_out/c4_f20_episode_checks.json | head -30
