#!/bin/bash
# round 5 job 28: SF3D_SLAB_OVERLAP - an approximation's rows (k_assemble) of one slab of the chunk list beside the node properties (k_props) of
# the next on a second stream: bit-identity in the launch-mode tests, then the headline with S = 0 / 2 / 3 / 4 and grid sizes of the overlapped launches
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "launch_modes_are" 2>&1 | tail -4 | tee gpurun_out/r05_job28_tests.log
O=gpurun_out/r05_job28_slab_overlap.txt; : > $O
run() { python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" >> $O; }
for rep in 1 2; do
  SF3D_SLAB_OVERLAP=0 run "S=0 rep $rep"
  SF3D_SLAB_OVERLAP=2 run "S=2 P512 A512 rep $rep"
  SF3D_SLAB_OVERLAP=3 run "S=3 P512 A512 rep $rep"
done
SF3D_SLAB_OVERLAP=4 run "S=4 P512 A512"
SF3D_SLAB_OVERLAP=2 SF3D_SLAB_FIRST=0.35 run "S=2 first 0.35"
SF3D_SLAB_OVERLAP=2 SF3D_SLAB_PROPS_BLOCKS=256 SF3D_SLAB_ASM_BLOCKS=768 run "S=2 P256 A768"
SF3D_SLAB_OVERLAP=2 SF3D_SLAB_PROPS_BLOCKS=768 SF3D_SLAB_ASM_BLOCKS=512 run "S=2 P768 A512"
SF3D_SLAB_OVERLAP=3 SF3D_SLAB_PROPS_BLOCKS=256 SF3D_SLAB_ASM_BLOCKS=768 run "S=3 P256 A768"
SF3D_SLAB_OVERLAP=3 SF3D_SLAB_PROPS_BLOCKS=1280 SF3D_SLAB_ASM_BLOCKS=1024 run "S=3 P1280 A1024 (full grids)"
cat $O
