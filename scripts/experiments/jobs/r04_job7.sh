#!/bin/bash
# round 4, GPU job 7: the host-memory fall-back transport; multirank + bench contract tests on the final code
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job7
mkdir -p $OUT
cd $ROOT
df -h /dev/shm | tail -1 > $OUT/shm.txt
python -m pytest tests/test_gpu_multirank.py tests/test_bench_contract.py -m gpu -q --durations=8 > $OUT/multirank.log 2>&1; echo "rc=$?" >> $OUT/multirank.log
SF3D_BENCH_SHARE_GPU=1 SF3D_EXCHANGE=host python bench.py --gpus 2 --steps 6 --warmup 0 --reps 2 --no-cpu-baseline > $OUT/bench_2_ranks_host_windows.json 2> $OUT/bench_2_ranks_host_windows.err
cat $OUT/shm.txt
tail -n 14 $OUT/multirank.log
grep -E "sf3d: rank|\[bench\]" $OUT/bench_2_ranks_host_windows.err | tail -6
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job7"
try:
    d = json.loads(open(out + "/bench_2_ranks_host_windows.json").read().strip().splitlines()[-1])
    print("2 ranks, host windows:", d["value"], d["config"]["work"], d["roofline"]["kernels"]["k_sweep_pair"])
except Exception as e:
    print("failed", e)
PY
