#!/bin/bash
# round 5 job 3: the new tests (RCCL sequencing over the mock, state directory on the device, heat sweep fall-back, edit after connect,
# exchange statistics on the 2-rank bench line), then the kink window for its WHOLE two hours against the pin, then one strip of C4 as a
# grid of its own in 2 / 4 / 8 (the inputs of profiles/r05_scale_model.json)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py -x -q -k "rccl or topology_edit" > gpurun_out/r05_job03_new_multirank.log 2>&1; tail -5 gpurun_out/r05_job03_new_multirank.log
python -m pytest tests/test_checkpoint.py tests/test_gpu_heat.py tests/test_bench_contract.py -x -q -s -m gpu -k "state_directory or coarser or layer_parity or two_ranks" > gpurun_out/r05_job03_new_tests.log 2>&1; tail -8 gpurun_out/r05_job03_new_tests.log
SF3D_LONG_TESTS=1 python -m pytest tests/test_gpu_sensitivity.py -x -q -s -k kink > gpurun_out/r05_job03_kink_whole_two_hours.log 2>&1; grep -E "kink window|passed|failed" gpurun_out/r05_job03_kink_whole_two_hours.log
for w in C4H C4Q C4E; do python bench.py --workload $w --steps 6 --warmup 1 --no-cpu-baseline --no-f60 > gpurun_out/r05_job03_bench_$w.json 2> gpurun_out/r05_job03_bench_$w.err; done
SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job03_bench_2ranks_shared.json 2> gpurun_out/r05_job03_bench_2ranks_shared.err; grep "exchange transport" gpurun_out/r05_job03_bench_2ranks_shared.err
python - <<'PY'
import json
for n in ("C4H","C4Q","C4E","2ranks_shared"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job03_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_us"], d.get("exchange"))
    except Exception as e: print(n, "ERR", e)
PY
