#!/bin/bash
# Builds library variants of staged (default-off) kernel changes into variants/ on the CPU box, so that one short gpurun
# call can test and time them:   bash scripts/build_variants.sh && gpurun -- 'bash scripts/variants.sh panel2'
# (scripts/variants.sh runs bench.py with each variants/lib_<name>.so in place of the product library and restores it.)
set -e
cd "$(dirname "$0")/../mcptam_amd/csrc"
mkdir -p ../../variants
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result"
[ -f img_api.o ] || make img_api.o
build() { name=$1; shift; /opt/rocm/bin/hipcc $FL "$@" -c -o /tmp/ba_$name.o ba_solver.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so /tmp/ba_$name.o img_api.o -ldl && echo "built variants/lib_$name.so ($*)"; }
# variants of the image library (tracker kernels): img_api.hip with extra flags, linked with the product's ba_solver.o
[ -f ba_solver.o ] || make ba_solver.o
build_img() { name=$1; shift; /opt/rocm/bin/hipcc $FL -ffp-contract=off "$@" -c -o /tmp/img_$name.o img_api.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so ba_solver.o /tmp/img_$name.o -ldl && echo "built variants/lib_$name.so ($*)"; }
for v in "$@"; do
  case $v in
    prrprof) build_img prrprof -DMCP_PRR_PROF & ;;          # phase stamps of k_pose_refine_regs and of its last vote select (scripts/prr_prof.sh prints them)
    prrnovote) build_img prrnovote -DPRR_NO_VOTE & ;;        # the Tukey median by histogram passes only (the fallback of the vote select, always)
    panel2) build panel2 -DMCP_CHOL_PANEL2=1 & ;;          # panel split over two wavefronts by column halves
    rsq2) build rsq2 -DCH_RSQ2=1 & ;;
    ld16) build ld16 -DCH_LOAD16=1 & ;;                  # k_chol_step: 16-byte tile loads
    st16) build st16 -DCH_STORE16=1 & ;;                 # k_chol_step: 16-byte stores of the trailing tiles
    pst16) build pst16 -DCH_PSTORE16=1 & ;;              # k_chol_step: panel rows through LDS, 16-byte stores
    ldst16) build ldst16 -DCH_LOAD16=1 -DCH_STORE16=1 & ;;
    lin2) build lin2 -DLIN_WAVES=2 & ;;
    sch4) build sch4 -DSCH_WAVES=4 & ;;                    # k_schur_group held to 128 registers: four workgroups per CU (1024 slots >= 785 groups)
    sch2) build sch2 -DSCH_WAVES=2 & ;;
    linabl3) build linabl3 -DLIN_ABL=3 & ;;                # timing ablation: no W block stores in k_linearize_group (results wrong)
    schprof) build schprof -DMCP_SCH_PROF & ;;             # phase stamps of k_schur4 / k_schur_group (wavefront 0 of sampled groups)
    s4d1) build s4d1 -DS4_DEPHASE=1 & ;;                  # k_schur4: odd-slot workgroup starts 2 k cycles late
    s4d2) build s4d2 -DS4_DEPHASE=2 & ;;
    s4d3) build s4d3 -DS4_DEPHASE=3 & ;;
    s4d2p) build s4d2p -DS4_DEPHASE=2 -DMCP_SCH_PROF & ;;
    cpabl1) build cpabl1 -DMCP_CP_PROF=4 -DCH_ABL=1 & ;;    # persistent factorisation, wave 0 stamps, panel without bulk updates (timing only)
    s4w1) build s4w1 -DS4_WAVES=1 & ;;                     # k_schur4 with 512 registers: one workgroup per compute unit
    linabl1p) build linabl1p -DMCP_LIN_PROF -DLIN_ABL=1 & ;;   # timing ablations of the linearisation with stamps (results wrong): no LDS adds
    linabl2p) build linabl2p -DMCP_LIN_PROF -DLIN_ABL=2 & ;;   # no pose slots
    linabl3p) build linabl3p -DMCP_LIN_PROF -DLIN_ABL=3 & ;;   # no W blocks
    proflin) build proflin -DMCP_LIN_PROF & ;;             # phase stamps of k_linearize_group
    proflin2) build proflin2 -DMCP_LIN_PROF -DLIN_WAVES=2 & ;;                    # k_linearize_group held to 256 registers (two wavefronts per SIMD)                      # second-order rsqrt correction in the panel chain
    hsprof) build hsprof -DMCP_HS_PROF=1 & ;;                # phase stamps of k_head_small (ba_small.h)
    cpprof) build cpprof -DMCP_CP_PROF=1 & ;;                # phase stamps of the one-launch factorisation (ba_chol2.h)
    cpprof2) build cpprof2 -DMCP_CP_PROF=2 & ;;
    cpprof3) build cpprof3 -DMCP_CP_PROF=3 & ;;
    cpprof4) build cpprof4 -DMCP_CP_PROF=4 & ;;
    cpprof5) build cpprof5 -DMCP_CP_PROF=5 & ;;
    hl32) build hl32 -DHL_GRID_N=32 & ;;
    hl128) build hl128 -DHL_GRID_N=128 & ;;
    hlprof) build hlprof -DMCP_HL_PROF & ;;                 # phase stamps of k_head_large (workgroup 0), printed with MCP_BA_EVT=1            # only each wavefront's arrival at the step's last barrier
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
