#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 300 2>&1 | tail -15
