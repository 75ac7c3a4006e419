#!/bin/bash
# phase stamps of the tracker's pose iterations (k_pose_refine_regs) at c3: build the variant on the CPU box first
#   bash scripts/build_variants.sh prrprof && gpurun -- 'bash scripts/prr_prof.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
cp mcptam_amd/libmcptam_hip.so /tmp/lib_base.so
cp variants/lib_prrprof.so mcptam_amd/libmcptam_hip.so
timeout 120 python scripts/bench_tracker.py 2>&1 | grep "prr prof" | sed -n 14,26p
cp /tmp/lib_base.so mcptam_amd/libmcptam_hip.so
