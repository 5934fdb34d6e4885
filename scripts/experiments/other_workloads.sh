#!/bin/bash
# the workloads profiles/README.md quotes next to the headline (no event timing, one repetition)
run() { timeout 900 python bench.py --no-cpu-baseline --no-kernel-timing --reps 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d['config']['work']; print('$*', '->', round(d['value'],4), 'sim-h/s', round(d['ms_per_step']*d['steps']/max(w['accepted'],1),4), 'ms/step', w)"; }
run --workload C3 --heat --steps 3 --warmup 0
run --workload C4 --forcing F60 --steps 1 --warmup 0
run --workload C3 --forcing F60 --steps 2 --warmup 0
run --workload C2 --forcing F60 --steps 2 --warmup 0
run --workload C5 --steps 2 --warmup 0
run --workload C5 --steps 2 --warmup 0 --lineal
