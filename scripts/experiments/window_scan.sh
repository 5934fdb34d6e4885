#!/bin/bash
# which 128 x 128 windows of the Ravone project stay within 1e-6 of the oracle over the full 25 mm hour + dry hour?
out=gpurun_out/${1:-scan}; mkdir -p $out
for w in "980 1108 300 428" "600 728 150 278" "200 328 330 458" "380 508 200 328"; do
  tag=$(echo $w | tr ' ' '_')
  python scripts/experiments/c5_window_diverge.py 500 16 $w > $out/win_$tag.log 2>&1
  echo "== $w"; grep -E "!!" $out/win_$tag.log | cut -c1-300 | head -2; grep "^step" $out/win_$tag.log | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' | tail -4; tail -1 $out/win_$tag.log | grep -o "counters_equal.*"
done
