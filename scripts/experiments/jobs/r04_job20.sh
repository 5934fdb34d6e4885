#!/bin/bash
# round 4, GPU job 20: profiles of the final code - C4 default (trace + PMC, 6 hours), C5 + heat (trace + PMC, hour 0), N ranks sharing the GPU (functional)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job20
mkdir -p $OUT
cd $ROOT
bash scripts/profile_gpu.sh r04_job20/prof > $OUT/profile_gpu.log 2>&1
PMC_STEPS=1 bash scripts/profile_gpu.sh r04_job20/prof_C5_heat --workload C5 --heat --steps 1 --warmup 0 --reps 1 > $OUT/profile_gpu_C5_heat.log 2>&1
for n in 2 8; do
  SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus $n --steps 6 --warmup 0 --reps 2 --no-cpu-baseline > $OUT/bench_${n}_ranks.json 2> $OUT/bench_${n}_ranks.err
done
head -n 16 $OUT/profile_gpu.log
head -n 22 $OUT/profile_gpu_C5_heat.log
for n in 2 8; do grep -E "sf3d: rank 0|\[bench\]" $OUT/bench_${n}_ranks.err | tail -2; tail -c 400 $OUT/bench_${n}_ranks.json; echo; done
