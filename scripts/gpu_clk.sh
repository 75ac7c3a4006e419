rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -5
rocm-smi --showperflevel 2>&1 | grep -i perf
timeout 120 python scripts/gpu_quick.py metric 2>&1 | grep -E "gpu rc"
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -3
timeout 120 python scripts/gpu_quick.py metric 2>&1 | grep -E "gpu rc"
rocm-smi --setperflevel auto 2>&1 | tail -1
