#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 5 400 python -m pytest tests/test_img_gpu.py -q -m gpu 2>&1 | tail -4
bash scripts/gpu_trk_prof.sh 2>&1 | cut -c1-110 | head -4
