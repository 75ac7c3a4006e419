#!/bin/bash
# Builds library variants of the staged (default-off) kernel changes into variants/ on the CPU box, so that one short gpurun
# call can test and time them all:   bash scripts/build_variants.sh && gpurun -- 'bash scripts/variants.sh sidediag panel2 panel2sd'
# (scripts/variants.sh runs bench.py with each variants/lib_<name>.so in place of the product library and restores it; add
#  `cp variants/lib_<name>.so mcptam_amd/libmcptam_hip.so; python -m pytest tests/test_ba_gpu.py -m gpu -x -q` for parity.)
set -e
cd "$(dirname "$0")/../mcptam_amd/csrc"
mkdir -p ../../variants
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result"
[ -f img_api.o ] || make img_api.o
build() { name=$1; shift; /opt/rocm/bin/hipcc $FL "$@" -c -o /tmp/ba_$name.o ba_solver.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so /tmp/ba_$name.o img_api.o -ldl && echo "built variants/lib_$name.so ($*)"; }
build sidediag -DMCP_CHOL_SIDE_DIAG=1 &        # factored diagonal tiles in a side array (removes the same-launch hazard)
build panel2   -DMCP_CHOL_PANEL2=1 &           # panel split over two wavefronts by column halves
build panel2sd -DMCP_CHOL_PANEL2=1 -DMCP_CHOL_SIDE_DIAG=1 &
wait
