#!/bin/bash
# in-kernel stamps of k_pose_refine_regs on the c3 frame (build with -DMCP_PRR_PROF, restore afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp mcptam_amd/libmcptam_hip.so /tmp/lib_keep.so
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_PRR_PROF" > gpurun_out/prr_build.log 2>&1 || tail -5 gpurun_out/prr_build.log
timeout 200 python scripts/bench_tracker.py 2>&1 | grep "prr prof" | tail -10
cp /tmp/lib_keep.so mcptam_amd/libmcptam_hip.so
