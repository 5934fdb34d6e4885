mkdir -p gpurun_out/hv
for L in A B C D A; do
  SF3D_PRODUCT_LIB=$PWD/build_variants/lib$L.so timeout 300 python bench.py --workload C3 --heat --steps 3 --warmup 0 --no-cpu-baseline --no-kernel-timing 2>gpurun_out/hv/$L.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],3), d['config']['work'])"
done
