#!/bin/bash
# codeobj.sh OBJ.o OUT.co — the gfx950 code object of a `hipcc -c` object (objcopy the .hip_fatbin section, unbundle); prints the kernels' LDS / VGPR / scratch
set -e
T=/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin "$1" "$2.fatbin"
$T/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$2.fatbin" --output="$2" --unbundle
rm -f "$2.fatbin"
$T/llvm-readelf --notes "$2" | grep -E "^\s+\.name:|group_segment_fixed_size|\.vgpr_count|private_segment_fixed_size|\.sgpr_count" | paste - - - - - | sed 's/  */ /g'
