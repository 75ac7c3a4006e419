#!/bin/bash
# times the prebuilt library variants under variants/ (built on the CPU box) at the metric size; restores the product library
cp mcptam_amd/libmcptam_hip.so /tmp/lib_base.so
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), 'it/s', {k: round(v,2) for k,v in d.get('stages',{}).get('ms_total',{}).items()})
"; }
for v in "$@"; do
  cp variants/lib_$v.so mcptam_amd/libmcptam_hip.so
  case $v in
    prof*) echo "== $v"; MCP_BA_SPECULATE=0 timeout 100 python scripts/gpu_quick.py metric 2>&1 | grep "prof\]" | sed -n 2,3p ;;
    *) timeout 100 python bench.py --cpu-iters 0 2>/dev/null | show $v ;;
  esac
done
cp /tmp/lib_base.so mcptam_amd/libmcptam_hip.so
timeout 100 python bench.py --cpu-iters 0 2>/dev/null | show base
