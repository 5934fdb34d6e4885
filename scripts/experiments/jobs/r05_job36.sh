#!/bin/bash
# round 5 job 36: the driver's sequence on the final code (list-range fields in DevView, slab experiment behind its switch) - pytest -x -q -m gpu, smoke(), python bench.py
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r05_job36_suite.log 2>&1; tail -14 gpurun_out/r05_job36_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r05_job36_bench.json 2> gpurun_out/r05_job36_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_job36_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_us"], d["cpu_baseline"]["value"], d["cpu_baseline"]["tuned"], d["f60_hour0"]["value"])
PY
