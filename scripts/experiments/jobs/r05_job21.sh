#!/bin/bash
# round 5 job 21: patch height W of the paired sweep re-measured with the gather-free kernels (the cost model of sf3d_host_build.inc was
# fitted to round 4's kernel): C4 and one of two / four / eight strips of it as grids of their own, the Ravone project
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f60 --steps 6 --warmup 1"
for w in C4 C4H C4Q C4E; do
  for W in 6 10 14; do SF3D_PAIR_SWEEP=1 SF3D_PAIR_W=$W python bench.py --workload $w $B > gpurun_out/r05_job21_${w}_W$W.json 2> gpurun_out/r05_job21_${w}_W$W.err; done
done
for W in 6 10 14; do SF3D_PAIR_W=$W python bench.py --workload C5 --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job21_C5_W$W.json 2> gpurun_out/r05_job21_C5_W$W.err; done
python - <<'PY' | tee gpurun_out/r05_job21_pair_W_sweep.txt
import json
print("paired sweep (gather-free kernels, round 5), patch height W (rows per patch incl. the two halo rows): us per pass of two iterations, sim-h/s")
for w in ("C4","C4H","C4Q","C4E","C5"):
    for W in (6,10,14):
        try:
            d=json.loads(open(f"gpurun_out/r05_job21_{w}_W{W}.json").read().strip().splitlines()[-1])
            r=d["roofline"]; print(f"{w:4s} W={W:2d}  {r['kernel']:14s} {r['avg_us']:8.2f} us  frac {r['frac']:.4f}  value {d['value']:.4f} sim-h/s")
        except Exception as e: print(w, W, "-", e)
PY
rm -f gpurun_out/r05_job21_*.err
