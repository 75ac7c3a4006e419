#!/bin/bash
# round 3: chunk-major W layout -- BA parity tests, then A/B against the previous commit's library (variants/lib_prevW.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 2>&1 | tail -5
bash scripts/variants.sh prevW
bash scripts/variants.sh prevW
