for L in default F; do
  if [ $L = default ]; then unset SF3D_PRODUCT_LIB; else export SF3D_PRODUCT_LIB=$PWD/build_variants/lib$L.so; fi
  echo "=== $L"
  bash scripts/pmc_probe.sh flpmc_$L "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" | grep k_assemble
  bash scripts/pmc_probe.sh flpmc2_$L "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" | grep k_assemble
done
