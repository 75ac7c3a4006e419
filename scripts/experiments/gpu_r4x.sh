#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread 2>&1 | tail -6 > gpurun_out/final_tests.log; cat gpurun_out/final_tests.log
