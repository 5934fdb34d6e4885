#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py, then
# summarise into gpurun_out/<tag>/summary.json (copied by hand into profiles/).
# usage: bash scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# PMC passes: the timed region exactly once (no warm-up hour, one repetition, no F60 leg), so that the sum over all dispatches is the
# HBM traffic of the K simulated hours (PMC_STEPS, default 6: roofline.step.traffic)
PMC_STEPS=${PMC_STEPS:-6}
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $OUT/pmc_$cnt -- python3 $ROOT/bench.py --no-cpu-baseline --no-f60 --no-kernel-timing "$@" --steps $PMC_STEPS --warmup 0 --reps 1 > /dev/null 2>&1
done
cd $ROOT
python3 scripts/profile_summary.py $OUT $PMC_STEPS
