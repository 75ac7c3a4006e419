#!/bin/bash
# panel recurrence on unscaled columns (CH_LDL=1, variants/lib_ldl.so): Cholesky + parity tests with the variant, then A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp mcptam_amd/libmcptam_hip.so /tmp/keep.so; cp variants/lib_ldl.so mcptam_amd/libmcptam_hip.so
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -x -q --timeout 400 -k "cholesky or golden or parity or matches or c4 or reproducible or recovers" 2>&1 | tail -3
cp /tmp/keep.so mcptam_amd/libmcptam_hip.so
bash scripts/variants.sh ldl
bash scripts/variants.sh ldl
