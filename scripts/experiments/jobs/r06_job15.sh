#!/bin/bash
# round 6 job 15: the round's rocprofv3 records on the final code - kernel trace + PMC (FETCH_SIZE / WRITE_SIZE in their own passes) of the driver's command,
# of one strip of C4 with the resident sweep loop (C4E) and of --workload C5; every summary stamped with the fingerprint of the kernel sources
mkdir -p gpurun_out
bash scripts/profile_gpu.sh r06_z --steps 20 --warmup 5 --no-extra-legs > gpurun_out/r06_job15_profile_C4.txt 2>&1; tail -14 gpurun_out/r06_job15_profile_C4.txt
bash scripts/profile_gpu.sh r06_z_C4E --workload C4E --steps 6 --warmup 1 > gpurun_out/r06_job15_profile_C4E.txt 2>&1; tail -12 gpurun_out/r06_job15_profile_C4E.txt
PMC_STEPS=1 bash scripts/profile_gpu.sh r06_z_C5 --workload C5 --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job15_profile_C5.txt 2>&1; tail -12 gpurun_out/r06_job15_profile_C5.txt
find gpurun_out/r06_z gpurun_out/r06_z_C4E gpurun_out/r06_z_C5 -name "*.csv" -size +8M -delete
