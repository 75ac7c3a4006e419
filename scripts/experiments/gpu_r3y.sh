#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -3
prof() {
  cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pyr_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr_$1 -- python $R/scripts/bench_tracker.py $2 > $R/gpurun_out/pyr_$1_$2.json 2>/dev/null
  f=$(find $R/gpurun_out/pyr_$1 -name '*kernel_stats.csv' | head -1); grep "k_pyr_fast" $f | cut -d, -f1-4 | sed "s/^/$1 $2 /"
  find $R/gpurun_out/pyr_$1 -name '*kernel_trace.csv' -delete; cd $R
}
prof new c3; prof new c5
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMCP_PYR_PROF" > gpurun_out/pyr_prof_build.log 2>&1 || tail -5 gpurun_out/pyr_prof_build.log
timeout 200 python scripts/bench_tracker.py 2>&1 | grep "pyr prof" | tail -2
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
