#!/bin/bash
# k_pyr_fast with 256 / 512 / 1024 threads per tile: parity with 1024, kernel times for all
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so
prof() {
  cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pyr_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr_$1 -- python $R/scripts/bench_tracker.py $2 > $R/gpurun_out/pyr_$1_$2.json 2>/dev/null
  f=$(find $R/gpurun_out/pyr_$1 -name '*kernel_stats.csv' | head -1); grep "k_pyr_fast" $f | cut -d, -f1-4 | sed "s/^/$1 $2 /"
  find $R/gpurun_out/pyr_$1 -name '*kernel_trace.csv' -delete; cd $R
}
prof nt256 c3; prof nt256 c5
for v in nt512 nt1024; do
  cp variants/lib_$v.so mcptam_amd/libmcptam_hip.so
  timeout 600 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 -k "lite or frame or c5 or odd" 2>&1 | tail -1
  prof $v c3; prof $v c5
done
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
