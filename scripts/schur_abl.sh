#!/bin/bash
# ablation of k_schur_group (results are wrong with SCHUR_ABL set; timing only)
mkdir -p gpurun_out
for extra in "" "-DSCHUR_ABL=1" "-DSCHUR_ABL=2" "-DSCHUR_ABL=3"; do
  make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $extra" > gpurun_out/abl_build.log 2>&1
  echo "== flags: $extra"
  timeout 200 python bench.py --cpu-iters 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config'].get('reduced_system_solves'), d['stages']['ms_total'])"
done
make -C mcptam_amd/csrc clean >/dev/null; make -C mcptam_amd/csrc >> gpurun_out/abl_build.log 2>&1
