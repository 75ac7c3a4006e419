#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -3
trk() { timeout 300 python scripts/bench_tracker.py $1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_ms')})"; }
trk c3 io1-c3; trk c5 io1-c5
export MCP_TRACK_IO_STREAM=0
trk c3 io0-c3; trk c5 io0-c5
unset MCP_TRACK_IO_STREAM
trk c3 io1-c3; trk c5 io1-c5
