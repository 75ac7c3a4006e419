timeout 400 python -m pytest tests/test_ba_gpu.py -m gpu -q --timeout 300 --timeout-method=thread -x -k "c4_size" 2>&1 | tail -3
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-iters 0 --no-roofline 2>&1 | tail -2 | cut -c1-300
