#!/bin/bash
# Build container: tools/bin/libsmm_hip_lab${LABTAG}.so = the library with the measurement hooks of smm_resident.hip compiled in (-DSMM_RESIDENT_LAB)
set -e
cd "$(dirname "$0")/.."
C=sparse_matrix_math_amd/csrc
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DSMM_RESIDENT_LAB $LABFLAGS -c $C/smm_resident.hip -o tools/bin/smm_resident_lab.o
OBJS=$(ls sparse_matrix_math_amd/lib/obj/*.o | grep -v "\.fma\.o" | grep -v smm_resident.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsmm_hip_lab${LABTAG}.so $OBJS tools/bin/smm_resident_lab.o -ldl
echo built tools/bin/libsmm_hip_lab${LABTAG}.so
