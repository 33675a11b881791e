#!/bin/bash
# build castro_amd/libvariant_<name>.so from the kernel sources with extra compiler flags (A/B and timing diagnostics)
# usage: tools/build_variant.sh name "-DFLAG ..."
set -e
cd "$(dirname "$0")/../castro_amd/csrc"
D=/tmp/variant_$1; mkdir -p $D
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $2"
for f in ctu_kernels aux_kernels unit_kernels capi halo_rccl; do /opt/rocm/bin/hipcc $F -c $f.hip -o $D/$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libvariant_$1.so $D/*.o
echo built castro_amd/libvariant_$1.so
