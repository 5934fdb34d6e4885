#!/bin/bash
# round 5 job 1: the library-faithful elementary functions on the device (device == libm), the kink window against the pin, cost on the headline
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_sensitivity.py -x -q -s > gpurun_out/r05_job01_tests.log 2>&1
tail -25 gpurun_out/r05_job01_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_job01_bench_glibc.json 2> gpurun_out/r05_job01_bench_glibc.err
SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_fm.so python bench.py --steps 20 --warmup 5 > gpurun_out/r05_job01_bench_fm.json 2> gpurun_out/r05_job01_bench_fm.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_job01_bench_glibc2.json 2> gpurun_out/r05_job01_bench_glibc2.err
python - <<'PY'
import json
for n in ("glibc","fm","glibc2"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job01_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("f60_hour0"))
    except Exception as e: print(n, "ERR", e)
PY
