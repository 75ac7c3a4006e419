timeout 200 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 100 --timeout-method=thread -x -k "dense or c2_full or compute_matches" 2>&1 | tail -2
timeout 120 python scripts/gpu_quick.py metric 2>&1 | grep -E "gpu rc|parity"
