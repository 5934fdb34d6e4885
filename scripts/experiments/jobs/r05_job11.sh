#!/bin/bash
# round 5 job 11: the suite with H and Se of the water path asserted BIT-IDENTICAL to the checker (tests/tolerances.py: assert_water_nodes)
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=6 > gpurun_out/r05_job11_suite_bit_identity.log 2>&1; tail -30 gpurun_out/r05_job11_suite_bit_identity.log | cut -c1-300
