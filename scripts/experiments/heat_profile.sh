mkdir -p gpurun_out/heatk
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/heatk/trace -- python3 $ROOT/bench.py --workload C3 --heat --steps 3 --warmup 0 --no-cpu-baseline --no-kernel-timing > $ROOT/gpurun_out/heatk/bench.json 2> $ROOT/gpurun_out/heatk/bench.err
cd $ROOT
python3 bench.py --workload C3 --heat --steps 3 --warmup 0 --no-cpu-baseline --no-kernel-timing 2>/dev/null > gpurun_out/heatk/bench_unprofiled.json
cut -c1-100 gpurun_out/heatk/bench_unprofiled.json
head -25 gpurun_out/heatk/trace/*/*kernel_stats.csv | cut -c1-150
