#!/bin/bash
# Register and spill counts of the kernels in one object file (hipcc -c output).   tools/kernel_regs.sh wn_detect.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $tmp/fat.bin
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.o
$LLVM/llvm-readelf --notes $tmp/dev.o | grep -E "^ +\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" | paste - - - - - - -
rm -rf $tmp
