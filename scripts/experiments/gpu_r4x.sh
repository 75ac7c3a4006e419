#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests -q -m gpu 2>&1 | tail -6
timeout -k 5 300 python bench.py 2>/dev/null > gpurun_out/bench_full.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
print(round(d['value'],1), 'it/s'); print(d.get('recent_window')); t=d.get('tracker_c3'); print({k:t[k] for k in t if not isinstance(t[k],(dict,list))} if isinstance(t,dict) else t)
PY
