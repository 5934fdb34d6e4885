mkdir -p gpurun_out/fl
run() { timeout 300 python bench.py --no-cpu-baseline --time-all-kernels 2>gpurun_out/fl/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$1', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items()})"; }
unset SF3D_PRODUCT_LIB; run default
SF3D_PROPS_BLOCKS=2048 run default_2048
export SF3D_PRODUCT_LIB=$PWD/build_variants/libG.so; run G5
SF3D_PROPS_BLOCKS=2048 run G5_2048
export SF3D_PRODUCT_LIB=$PWD/build_variants/libH.so; run H_post7
