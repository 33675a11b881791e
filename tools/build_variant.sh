#!/bin/bash
# build castro_amd/libvariant_<name>.so from the kernel sources with extra compiler flags (A/B and timing diagnostics)
# usage: [NUMERICS=contract] tools/build_variant.sh name "-DFLAG ..."     (NUMERICS: the flag set of castro_amd/csrc/Makefile, default exact)
set -e
cd "$(dirname "$0")/../castro_amd/csrc"
D=/tmp/variant_$1; mkdir -p $D
if [ "${NUMERICS:-exact}" = contract ]; then NF="-fno-fast-math -ffp-contract=fast -fno-signed-zeros -fno-trapping-math -DCAD_NUMERICS_CONTRACT -DHW_MINMAX_ON"
else NF="-ffp-contract=off -fno-fast-math"; fi
F="-O3 -std=c++17 -fPIC $NF --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $2"
for f in ctu_kernels aux_kernels unit_kernels capi halo_rccl cluster_host; do /opt/rocm/bin/hipcc $F -c $f.hip -o $D/$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libvariant_$1.so $D/*.o
echo built castro_amd/libvariant_$1.so
