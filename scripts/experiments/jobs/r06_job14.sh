#!/bin/bash
# round 6 job 14: the whole GPU suite again, with the durations of every test that takes more than a second and the load of the box next to it
mkdir -p gpurun_out
( while true; do echo "$(date +%s) load $(cut -d' ' -f1-3 /proc/loadavg) procs $(ps -e --no-headers | wc -l) py $(pgrep -c python)"; sleep 20; done ) > gpurun_out/r06_job14_load.txt 2>&1 &
LP=$!
( time timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=60 --durations-min=2 ) > gpurun_out/r06_job14_suite.txt 2>&1; tail -75 gpurun_out/r06_job14_suite.txt
kill $LP
