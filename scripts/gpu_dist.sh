timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 200 --timeout-method=thread -x -k "two_rank" 2>&1 | tail -15 > gpurun_out/dist_test.log; cat gpurun_out/dist_test.log
