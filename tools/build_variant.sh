#!/bin/bash
# usage: tools/build_variant.sh NAME "-DD2G_BS_EXP=1 ..."  -> dashing2_amd/libd2g_NAME.so (experiments; select with D2G_LIB=...)
set -e
cd "$(dirname "$0")/../dashing2_amd/csrc"
name=$1; shift
tmp=/tmp/d2g_variant_$name; mkdir -p $tmp
for f in d2g_runtime d2g_k0 d2g_k1 d2g_k2 d2g_k2_bitslice d2g_k3_bmh d2g_mgpu; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $f.hip -o $tmp/$f.o &
done
g++ -O2 -std=c++17 -fPIC -fopenmp -c d2g_host.cpp -o $tmp/d2g_host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libd2g_$name.so $tmp/*.o -lz -lgomp -ldl
echo built dashing2_amd/libd2g_$name.so
