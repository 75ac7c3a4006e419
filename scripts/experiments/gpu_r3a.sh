#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --timeout-method=thread 2>&1 | tail -25 > gpurun_out/r3a_tests.log; tail -8 gpurun_out/r3a_tests.log
bash scripts/variants.sh ld16 st16 pst16 ldst16 2>&1 | tee gpurun_out/r3a_variants.log
