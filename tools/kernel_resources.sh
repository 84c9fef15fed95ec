#!/bin/bash
# registers / spills / scratch of every kernel in a host object:  tools/kernel_resources.sh <file.o>
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin "$1" $T/copy.o && $L/clang-offload-bundler --unbundle --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co &&
$L/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|private_segment_fixed|vgpr_spill" | paste - - - - | sed 's/  */ /g'
rm -rf $T
