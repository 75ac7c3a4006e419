#!/bin/bash
# Ziv fast path in atan_cr: parity (image + BA suites), then A/B against the previous library on the BA bench and the tracker bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py tests/test_ba_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -4
bash scripts/variants.sh prevW
bash scripts/variants.sh prevW
trk() { timeout 300 python scripts/bench_tracker.py $1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_ms')})"; }
trk c3 new-c3; trk c5 new-c5
