#!/bin/bash
# k_pose_refine_regs with 256 x 4 / 512 x 2 / 1024 x 1 (threads x points per thread): parity subset + c3 frame + kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so
prof() {
  cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prr_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prr_$1 -- python $R/scripts/bench_tracker.py c3 > $R/gpurun_out/prr_$1.json 2>/dev/null
  f=$(find $R/gpurun_out/prr_$1 -name '*kernel_stats.csv' | head -1); grep "k_pose_refine_regs" $f | cut -d, -f1-4 | cut -c1-30,200- | sed "s/^/$1 /"
  find $R/gpurun_out/prr_$1 -name '*kernel_trace.csv' -delete; cd $R
}
prof base
for v in prr512 prr1024; do
  cp variants/lib_$v.so mcptam_amd/libmcptam_hip.so
  timeout 600 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 -k "refine or frame or estimators" 2>&1 | tail -1
  prof $v
done
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
