timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 100 --timeout-method=thread -x 2>&1 | tail -3
timeout 120 python scripts/gpu_quick.py c2 metric 2>&1 | grep -E "gpu rc|parity"
