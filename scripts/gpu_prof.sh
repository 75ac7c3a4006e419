timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 60 --timeout-method=thread -x 2>&1 | tail -4 > gpurun_out/tests.log; cat gpurun_out/tests.log
timeout 120 python scripts/gpu_quick.py c2 metric 2>&1 | grep -E "==|gpu rc|parity|PARITY" | tee gpurun_out/quick2.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof2.log | cut -c1-600
