#!/bin/bash
# round 5 job 13: the heat vectors of the unmodified reference in the reference's sweep order: bit for bit?
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_heat.py -q -s -k "reference_sweep_order or reference_order" > gpurun_out/r05_job13_heat_reference_order.log 2>&1; grep -E "bit-identical|GS vs|passed|failed|Error|assert" gpurun_out/r05_job13_heat_reference_order.log | cut -c1-400 | tail -30
