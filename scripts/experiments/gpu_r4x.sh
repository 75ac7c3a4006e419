#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 50 python -m pytest tests/test_ba_gpu.py -q -m gpu -k "small_bundle and window" 2>&1 | tail -1
