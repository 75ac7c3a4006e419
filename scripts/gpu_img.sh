timeout 400 python -m pytest tests/test_img_gpu.py -m gpu -q --timeout 120 --timeout-method=thread -x 2>&1 | tail -25 > gpurun_out/img_tests.log; cat gpurun_out/img_tests.log
