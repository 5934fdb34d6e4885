#!/bin/bash
# round 5 job 29: is the link-flow-sum kernel beside the next step (SF3D_OVERLAP_ACCEPT, round 1-2: 31.0 -> 33.2 sim-h/s) still worth it with
# today's kernels?  And the grid of k_accept_links (SF3D_LINKS_BLOCKS)
mkdir -p gpurun_out
O=gpurun_out/r05_job29_overlap_accept.txt; : > $O
run() { python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" >> $O; }
for rep in 1 2; do
  SF3D_OVERLAP_ACCEPT=1 run "overlap on rep $rep"
  SF3D_OVERLAP_ACCEPT=0 run "overlap off rep $rep"
done
for nb in 256 512 1024; do SF3D_LINKS_BLOCKS=$nb run "overlap on, k_accept_links on $nb blocks"; done
cat $O
