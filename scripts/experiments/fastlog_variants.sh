mkdir -p gpurun_out/fl
python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
for L in default F default; do
  if [ $L = default ]; then unset SF3D_PRODUCT_LIB; else export SF3D_PRODUCT_LIB=$PWD/build_variants/lib$L.so; fi
  timeout 300 python bench.py --no-cpu-baseline --time-all-kernels 2>gpurun_out/fl/$L.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$L', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items()}, d['config']['work'])"
done
