mkdir -p gpurun_out/f0
run() { timeout 300 python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/f0/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:(v['launches'], round(v['total_ms']/max(v['launches'],1)*1e3,1)) for n,v in k.items()}, d['config']['work'])"; }
run fused --time-all-kernels
SF3D_FUSE_FIRST_SWEEP=0 run unfused --time-all-kernels
run fused_untimed --no-kernel-timing
SF3D_FUSE_FIRST_SWEEP=0 run unfused_untimed --no-kernel-timing
python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
