#!/bin/bash
# round 6 job 32: -DSF3D_RES_PROFILE=1 (build_variants/libresprof.so): the phases of a resident-loop iteration on one GPU (C4E) and on a strip (C4E in two
# strips, SF3D_RESIDENT_PR=2, both ranks on the one GPU) - block 0 of each rank prints them at release
mkdir -p gpurun_out
export SF3D_PRODUCT_LIB=$PWD/build_variants/libresprof.so
timeout 300 python bench.py --workload C4E --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 3 > gpurun_out/r06_job32_one.json 2> gpurun_out/r06_job32_one.err; grep "sf3d\]" gpurun_out/r06_job32_one.err | tail -9
SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 3 > gpurun_out/r06_job32_two.json 2> gpurun_out/r06_job32_two.err; grep "sf3d\]" gpurun_out/r06_job32_two.err | tail -18
