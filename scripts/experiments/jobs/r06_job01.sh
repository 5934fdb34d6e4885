#!/bin/bash
# round 6 job 1: what round 5's review found missing - rocprofv3 kernel trace + PMC (FETCH_SIZE / WRITE_SIZE in their own passes) of the
# STRIP regime on one GPU (one strip of C4 run as a grid of its own: a half, a quarter, an eighth) and of --workload C5 on the final kernels
mkdir -p gpurun_out
for w in C4E C4Q C4H; do
  bash scripts/profile_gpu.sh r06_a_$w --workload $w --steps 6 --warmup 1 > gpurun_out/r06_job01_profile_$w.txt 2>&1; tail -14 gpurun_out/r06_job01_profile_$w.txt
done
PMC_STEPS=1 bash scripts/profile_gpu.sh r06_a_C5 --workload C5 --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job01_profile_C5.txt 2>&1; tail -14 gpurun_out/r06_job01_profile_C5.txt
find gpurun_out/r06_a_* -name "*.csv" -size +8M -delete
