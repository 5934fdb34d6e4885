#!/bin/bash
# round 5 job 4: the occupancy point W = 8 of the paired sweep (three blocks of nine waves per CU) at C4, C4H and the Ravone project;
# roctx ranges smoke; the round's rocprofv3 records: kernel trace + PMC of the driver's command and of --workload C5
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f60 --steps 6 --warmup 1"
for w in C4 C4H; do
  for W in 10 8 6; do SF3D_PAIR_W=$W python bench.py --workload $w $B > gpurun_out/r05_job04_${w}_W$W.json 2> gpurun_out/r05_job04_${w}_W$W.err; done
done
for W in 10 8; do SF3D_PAIR_W=$W python bench.py --workload C5 --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job04_C5_W$W.json 2> gpurun_out/r05_job04_C5_W$W.err; done
python - <<'PY' | tee gpurun_out/r05_job04_pair_W8.txt
import json
print("paired sweep, patch height W (rows per patch incl. the two halo rows): us per pass of two iterations, sim-h/s")
for w in ("C4","C4H","C5"):
    for W in (10,8,6):
        try:
            d=json.loads(open(f"gpurun_out/r05_job04_{w}_W{W}.json").read().strip().splitlines()[-1])
            r=d["roofline"]; print(f"{w:4s} W={W:2d}  {r['kernel']:14s} {r['avg_us']:8.2f} us  frac {r['frac']:.4f}  value {d['value']:.4f} sim-h/s")
        except Exception as e: print(w, W, "-", e)
PY
cd /tmp && export TMPDIR=/tmp
SF3D_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_job04_roctx -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 2 --warmup 0 --reps 1 --no-cpu-baseline --no-f60 > $GRAFT_REPO_ROOT/gpurun_out/r05_job04_roctx.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_job04_roctx.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r05_job04_roctx -name "*marker*stats*" | head; head -8 $(find gpurun_out/r05_job04_roctx -name "*marker*stats*" | head -1) 2>/dev/null
find gpurun_out/r05_job04_roctx -name "*_trace.csv" -size +5M -delete
bash scripts/profile_gpu.sh r05_d --steps 20 --warmup 5 > gpurun_out/r05_job04_profile_C4.txt 2>&1; tail -15 gpurun_out/r05_job04_profile_C4.txt
PMC_STEPS=1 bash scripts/profile_gpu.sh r05_d_C5 --workload C5 --steps 1 --warmup 0 --reps 1 > gpurun_out/r05_job04_profile_C5.txt 2>&1; tail -12 gpurun_out/r05_job04_profile_C5.txt
find gpurun_out/r05_d gpurun_out/r05_d_C5 -name "*.csv" -size +8M -delete
