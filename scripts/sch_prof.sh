#!/bin/bash
# phase stamps of k_schur_group and k_linearize_group at the metric size; restores the normal build
mkdir -p gpurun_out
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_SCH_PROF -DMCP_LIN_PROF $*" > gpurun_out/sch_prof_build.log 2>&1
timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "sch prof\|lin prof" | sort | uniq -c | sort -rn | head -6
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc >> gpurun_out/sch_prof_build.log 2>&1
