#!/bin/bash
# build the library with in-kernel phase stamps for the Cholesky step, print the phase split, restore the normal build
# usage: chol_prof.sh [extra -D flags ...]
mkdir -p gpurun_out
for extra in "" "$@"; do
  make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_CHOL_PROF $extra" > gpurun_out/chol_prof_build.log 2>&1
  echo "== flags: $extra"
  timeout 120 python scripts/chol_time.py 2>&1 | tee gpurun_out/chol_prof.log | grep -E -A3 "chol prof|back prof" | tail -8
done
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc >> gpurun_out/chol_prof_build.log 2>&1
