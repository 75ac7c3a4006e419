#!/bin/bash
# phase stamps of k_pyr_fast (one level-0 tile in the image interior) on the c3 and c5 frames
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_PYR_PROF" > gpurun_out/pyr_prof_build.log 2>&1 || tail -5 gpurun_out/pyr_prof_build.log
timeout 200 python scripts/bench_tracker.py 2>&1 | grep "pyr prof" | tail -3
timeout 200 python scripts/bench_tracker.py c5 2>&1 | grep "pyr prof" | tail -3
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
