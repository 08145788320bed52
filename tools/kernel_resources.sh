#!/bin/bash
# registers / spills / LDS of every kernel in a built object: tools/kernel_resources.sh object_nerf_amd/csrc/build/mlp_fused.o
# (no GPU needed: reads the code object's metadata notes)
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
tgt=$($L/clang-offload-bundler --type=o --input=$T/fat.bin --list | grep gfx950)
$L/clang-offload-bundler --type=o --targets=$tgt --input=$T/fat.bin --output=$T/k.co --unbundle
$L/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|\.agpr_count|private_segment_fixed|group_segment_fixed" \
  | paste - - - - - - | sed 's/  */ /g; s/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.vgpr_spill_count/spill/'
rm -rf $T
