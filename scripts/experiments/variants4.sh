mkdir -p gpurun_out/v4
run() { timeout 300 python bench.py --no-cpu-baseline --time-all-kernels 2>gpurun_out/v4/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items()})"; }
for L in V0 V1 V0 V1; do export SF3D_PRODUCT_LIB=$PWD/build_variants/lib$L.so; run $L; done
