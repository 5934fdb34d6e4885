mkdir -p gpurun_out/ex
python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_heat.py -x -q 2>&1 | tail -4
run() { timeout 300 python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/ex/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:(v['launches'], round(v['total_ms']/max(v['launches'],1)*1e3,1)) for n,v in k.items()})"; }
run c4_timed --time-all-kernels
run c4 --no-kernel-timing
run c3heat --workload C3 --heat --steps 3 --warmup 0 --no-kernel-timing
