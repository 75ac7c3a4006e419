#!/bin/bash
# k_pose_refine_multi with 1024 / 512 / 256 threads per workgroup (x points per workgroup): parity subset + c5 frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so
trk() { timeout 300 python scripts/bench_tracker.py c5 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('gpu_ms_per_frame','gpu_ms_per_frame_in_library')})"; }
trk nt1024-ppw512
MCP_TRACK_REFINE_PPW=1024 trk nt1024-ppw1024
for v in prm512 prm256; do
  cp variants/lib_$v.so mcptam_amd/libmcptam_hip.so
  timeout 600 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 -k "many_workgroups or c5" 2>&1 | tail -1
  trk $v-ppw512
  MCP_TRACK_REFINE_PPW=256 trk $v-ppw256
  MCP_TRACK_REFINE_PPW=1024 trk $v-ppw1024
done
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
