python -m pytest tests/test_ba_gpu.py -m gpu -x -q -k "dense or solution or compute_matches" 2>&1 | tail -8
python scripts/gpu_quick.py c2 metric 2>&1 | grep -E "==|gpu rc|parity|PARITY"
