#!/bin/bash
out=gpurun_out/${1:-div}; mkdir -p $out
W="72 200 300 428"
python scripts/experiments/c5_window_diverge.py 50 16 $W 760 590 > $out/A_default.log 2>&1
SF3D_SLOT_ALIGN=0 python scripts/experiments/c5_window_diverge.py 50 16 $W 760 590 > $out/A_noalign.log 2>&1
SF3D_FUSED_DECIDE=0 SF3D_GRAPHS=0 SF3D_OVERLAP_ACCEPT=0 SF3D_PAIR_SWEEP=0 SF3D_APPROX_PATCH=0 python scripts/experiments/c5_window_diverge.py 50 16 $W 760 590 > $out/A_plain.log 2>&1
python scripts/experiments/c5_window_diverge.py 50 16 380 508 200 328 1500 100000 > $out/B_nohole.log 2>&1
for f in A_default A_noalign A_plain B_nohole; do echo "== $f"; grep -E "!!|>>" $out/$f.log | cut -c1-400 | head -8; tail -2 $out/$f.log | cut -c1-260; done
