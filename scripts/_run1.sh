run() { echo "== $*" >> gpurun_out/kn6.txt; env "$@" timeout 300 python bench.py --cpu-iters 0 --no-roofline --no-tracker 2>>gpurun_out/kn6.err | python scripts/show_bench.py /dev/stdin | head -1 >> gpurun_out/kn6.txt; }
rm -f gpurun_out/kn6.txt gpurun_out/kn6.err
for g in 2 1 2 1 2 1; do run MCP_BA_SPEC_TRIALS=$g; done
timeout 900 python -m pytest tests -m gpu -q -x -k "scheduling or small_bundle or two_handles or concurrent or shared" 2>&1 | tail -3 >> gpurun_out/kn6.txt
python scripts/bench_window.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window', d['ms_median'])" >> gpurun_out/kn6.txt
python scripts/bench_secondary.py c2 c4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): print(k, round(v['value'],1))" >> gpurun_out/kn6.txt
