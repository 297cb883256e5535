#!/bin/bash
# kernel_resources.sh <csrc file> [extra hipcc flags]: registers / spills / occupancy / LDS of every kernel in one source file
# (cross-compiles for gfx950 without a GPU; nothing is written into the tree)
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; shift
flags=""
case "$(basename "$src")" in mlp_fused.hip) flags="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -Wno-pass-failed -Wno-unused-value -I "$root/include" "$@" \
    -c "$src" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ *\[-Rpass.*//' | paste - - - - - - - \
  | sed 's/Function Name: //; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/LDS/' | (command -v c++filt >/dev/null && c++filt || cat)
rm -f /tmp/kres_$$.o
