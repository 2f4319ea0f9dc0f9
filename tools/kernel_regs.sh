#!/bin/bash
# usage: tools/kernel_regs.sh dashing2_amd/csrc/d2g_k2_bitslice.hip  -> VGPR/SGPR/LDS/spill per kernel (device-only compile)
set -e
EXTRA="${@:2}"
src=$1; out=/tmp/$(basename $src .hip).co
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only $EXTRA -c $src -o $out
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$out --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out.elf | grep -E "\.name:|\.vgpr_count|\.sgpr_count|group_segment_fixed_size|spill_count|agpr_count" | \
  awk '/\.name:/{n=$2} /agpr_count/{a=$2} /group_segment/{l=$2} /sgpr_count/{s=$2} /sgpr_spill/{ss=$2} /\.vgpr_count/{v=$2} /vgpr_spill/{printf "%-90s vgpr=%s agpr=%s sgpr=%s lds=%s spill(s/v)=%s/%s\n", n, v, a, s, l, ss, $2}'
