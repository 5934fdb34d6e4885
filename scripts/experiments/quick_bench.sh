mkdir -p gpurun_out/qb
run() { timeout 300 python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/qb/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:(v['launches'], round(v['total_ms']/max(v['launches'],1)*1e3,1)) for n,v in k.items()}, d['config']['work'])"; }
run timed --time-all-kernels
run untimed --no-kernel-timing
python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
