for c in tiny c1; do timeout 60 python scripts/gpu_dbg.py $c > gpurun_out/dbg_$c.log 2>&1; echo "rc=$?" >> gpurun_out/dbg_$c.log; tail -12 gpurun_out/dbg_$c.log; done
