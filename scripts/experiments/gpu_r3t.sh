#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
trk() { timeout 300 python scripts/bench_tracker.py $1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_ms')})"; }
trk c3 c3; trk c5 c5
