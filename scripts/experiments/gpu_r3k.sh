#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_img_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -5
timeout 300 python scripts/bench_tracker.py | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_') or k=='found_per_frame'})"
timeout 300 python scripts/bench_tracker.py c5 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('gpu_') or k=='found_per_frame'})"
