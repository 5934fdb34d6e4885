#!/bin/bash
# round 5 job 22: the round's rocprofv3 records with the gather-free paired sweeps: kernel trace + PMC (FETCH_SIZE / WRITE_SIZE in their own
# passes) of the driver's command and of --workload C5
mkdir -p gpurun_out
bash scripts/profile_gpu.sh r05_q --steps 20 --warmup 5 > gpurun_out/r05_job22_profile_C4.txt 2>&1; tail -15 gpurun_out/r05_job22_profile_C4.txt
PMC_STEPS=1 bash scripts/profile_gpu.sh r05_q_C5 --workload C5 --steps 1 --warmup 0 --reps 1 > gpurun_out/r05_job22_profile_C5.txt 2>&1; tail -12 gpurun_out/r05_job22_profile_C5.txt
find gpurun_out/r05_q gpurun_out/r05_q_C5 -name "*.csv" -size +8M -delete
