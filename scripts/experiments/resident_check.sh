mkdir -p gpurun_out/rc
run() { timeout 300 python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/rc/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items()})"; }
run c4_timed --time-all-kernels
SF3D_RESIDENT_GRIDS=0 run c4_timed_off --time-all-kernels
run c3heat --workload C3 --heat --steps 3 --warmup 0 --no-kernel-timing
SF3D_RESIDENT_GRIDS=0 run c3heat_off --workload C3 --heat --steps 3 --warmup 0 --no-kernel-timing
run c3f60 --workload C3 --forcing F60 --steps 1 --warmup 0 --no-kernel-timing
run c2f60 --workload C2 --forcing F60 --steps 2 --warmup 0 --no-kernel-timing
python -m pytest tests/test_gpu_heat.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
