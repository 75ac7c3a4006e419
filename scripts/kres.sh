#!/bin/bash
# kernel resource usage + per-kernel assembly of the BA library (CPU box; hipcc cross-compiles gfx950)
# usage: scripts/kres.sh <kernel name substring> [extra hipcc flags...]   -> build/scratch/<name>.s and a summary on stdout
R=$(cd "$(dirname "$0")/.." && pwd); K=$1; shift
mkdir -p $R/build/scratch && cd $R/build/scratch || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics "$@" -save-temps -Rpass-analysis=kernel-resource-usage \
  -c -o ba.o $R/mcptam_amd/csrc/ba_solver.hip 2> res.log
grep -A11 "Function Name: .*$K" res.log | grep "Name\|GPRs\|Scratch\|Occupancy\|LDS"
for sym in $(grep -o "^_Z[A-Za-z0-9_]*$K[A-Za-z0-9_]*:" ba_solver-hip-amdgcn-amd-amdhsa-gfx950.s | tr -d ':' | sort -u); do
  awk -v s="$sym:" '$1==s{p=1} p{print} p&&/\.end_amdhsa_kernel/{exit}' ba_solver-hip-amdgcn-amd-amdhsa-gfx950.s > $sym.s
  echo "$sym: $(wc -l < $sym.s) lines, scratch ops $(grep -c 'scratch_' $sym.s), calls $(grep -c s_swappc $sym.s)"
done
