#!/bin/bash
# registers / LDS / scratch of the kernels of the product library whose (mangled) name matches the regular expression $1
# (read from the gfx950 code object's metadata); usage: scripts/kernel_resources.sh 'k_sweep_resident' [library]
LIB=${2:-criteria3d_amd/csrc/libsf3d_hip.so}
T=$(mktemp -d)
LLVM=/opt/rocm/lib/llvm/bin
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB && \
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co && \
$LLVM/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read(); pat=sys.argv[1]
for blk in txt.split('  - .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk)
    if not name or not re.search(pat,name.group(1)): continue
    g=lambda k:(re.search(r'\.'+k+r':\s+(\d+)',blk) or [None,'?'])[1]
    print(f\"{name.group(1)[:80]:80s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} (spilled {g('sgpr_spill_count')}) lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s} vgpr-spill {g('vgpr_spill_count')}\")
" "$1"
rm -rf $T
