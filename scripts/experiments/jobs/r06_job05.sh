#!/bin/bash
# round 6 job 5: the whole GPU suite the way the driver runs it, with the resident sweep loop the default wherever a grid fits
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests/ -q -m gpu --durations=25 ) > gpurun_out/r06_job05_suite.txt 2>&1; tail -45 gpurun_out/r06_job05_suite.txt
