#!/bin/bash
# round 5 job 15: 4 x the headline grid (1 024 x 1 024 x 20 = 21 M nodes) on the final code - size-independence check of DESIGN.md 3
mkdir -p gpurun_out
timeout 900 python scripts/experiments/big_grid.py 1024 > gpurun_out/r05_job15_big_grid.txt 2>&1; tail -4 gpurun_out/r05_job15_big_grid.txt
