#!/bin/bash
# round 2, first GPU pass: full GPU suite on the fixed-order / race-free build, the bench line, then the PANEL2 variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread 2>&1 | tail -40 > gpurun_out/r02a_tests.log; tail -15 gpurun_out/r02a_tests.log
timeout 400 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r02a_bench.json; tail -3 gpurun_out/r02a_bench.err
if [ -f variants/lib_panel2.so ]; then
  cp mcptam_amd/libmcptam_hip.so /tmp/lib_base.so
  cp variants/lib_panel2.so mcptam_amd/libmcptam_hip.so
  timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 120 --timeout-method=thread -k "cholesky or compute_matches or reduced_system" 2>&1 | tail -8 > gpurun_out/r02a_panel2_tests.log; tail -4 gpurun_out/r02a_panel2_tests.log
  timeout 120 python bench.py --cpu-iters 0 > gpurun_out/r02a_panel2_bench.json 2>/dev/null; cut -c1-400 gpurun_out/r02a_panel2_bench.json
  cp /tmp/lib_base.so mcptam_amd/libmcptam_hip.so
fi
