run() { echo "== $*" >> gpurun_out/kn5.txt; env "$@" timeout 300 python bench.py --cpu-iters 0 --no-roofline --no-tracker 2>>gpurun_out/kn5.err | python scripts/show_bench.py /dev/stdin | head -1 >> gpurun_out/kn5.txt; }
rm -f gpurun_out/kn5.txt gpurun_out/kn5.err
for g in 1024 256 128 64 32 1024 128 64; do run MCP_BA_SELECT_GRID=$g; done
