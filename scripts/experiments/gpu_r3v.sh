#!/bin/bash
# k_pyr_fast with 4-byte stores: image-path parity, then kernel time against the byte-store build (variants/lib_narrow.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -3
prof() {
  cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pyr_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr_$1 -- python $R/scripts/bench_tracker.py $2 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pyr_$1 -name '*kernel_stats.csv' | head -1); grep "k_pyr_fast\|k_row_c" $f | cut -d, -f1-4 | sed "s/^/$1 $2 /"
  find $R/gpurun_out/pyr_$1 -name '*kernel_trace.csv' -delete; cd $R
}
prof wide c3; prof wide c5
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so; cp variants/lib_narrow.so mcptam_amd/libmcptam_hip.so
prof narrow c3; prof narrow c5
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
