#!/bin/bash
# per-kernel times of k_approx_patch build / run-time variants at C4 (build_variants/lib<NAME>.so built in the container first)
out=gpurun_out/${1:-patchvar}; mkdir -p $out
run() {  # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" SF3D_PRODUCT_LIB=$PWD/build_variants/lib$lib.so timeout 300 python bench.py --no-cpu-baseline --no-f60 --time-all-kernels --reps 1 --steps 2 --warmup 0 2>$out/$name.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$name', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items() if v['launches']}, flush=True)"
}
run base PV0 SF3D_APPROX_PATCH=0
for lib in PV0 PV1 PV2 PV3; do
  for w in 6 10; do
    run ${lib}_fused_w$w $lib SF3D_APPROX_PATCH=1 SF3D_PATCH_W=$w SF3D_PATCH_FUSED=1
    run ${lib}_split_w$w $lib SF3D_APPROX_PATCH=1 SF3D_PATCH_W=$w SF3D_PATCH_FUSED=0
  done
done
