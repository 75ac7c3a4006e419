#!/bin/bash
# phase stamps of k_linearize_group at the metric size (extra -D flags as arguments, e.g. -DLIN_ABL=1); restores the normal build
mkdir -p gpurun_out
for extra in "" "$@"; do
  make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_LIN_PROF $extra" > gpurun_out/lin_prof_build.log 2>&1
  echo "== flags: $extra"
  MCP_BA_SPECULATE=0 timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "lin prof" | head -2
done
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc >> gpurun_out/lin_prof_build.log 2>&1
