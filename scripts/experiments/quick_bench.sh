mkdir -p gpurun_out/qb
run() { timeout 300 python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/qb/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:(v['launches'], round(v['total_ms']/max(v['launches'],1)*1e3,1)) for n,v in k.items()})"; }
run timed --time-all-kernels
run untimed --no-kernel-timing
