#!/bin/bash
# k_pyr_fast with the compass-point pre-test: image-path parity, then kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -3
prof() {
  cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pyr_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr_$1 -- python $R/scripts/bench_tracker.py $2 > $R/gpurun_out/pyr_$1_$2.json 2>/dev/null
  f=$(find $R/gpurun_out/pyr_$1 -name '*kernel_stats.csv' | head -1); grep "k_pyr_fast" $f | cut -d, -f1-4 | sed "s/^/$1 $2 /"
  find $R/gpurun_out/pyr_$1 -name '*kernel_trace.csv' -delete; cd $R
}
prof pre c3; prof pre c5
trk() { timeout 300 python scripts/bench_tracker.py $1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_ms')})"; }
trk c3 c3; trk c5 c5
