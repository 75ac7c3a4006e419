#!/bin/bash
# k_assemble with 16 loads per round: parity subset + A/B against the previous library (variants/lib_prevA.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 -k "golden or identical or reduced or parity or scheduling" 2>&1 | tail -3
bash scripts/variants.sh prevA
bash scripts/variants.sh prevA
