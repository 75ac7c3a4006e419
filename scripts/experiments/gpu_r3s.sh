#!/bin/bash
# linearisation outputs double-buffered (no join with the speculative stream before linearize): BA suite + A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -3
bash scripts/variants.sh prevL
bash scripts/variants.sh prevL
MCP_BA_EVT=1 timeout 300 python bench.py --steps 8 --warmup 4 --cpu-iters 0 --no-roofline 2>&1 | grep "\[evt\] iter" | tail -6
