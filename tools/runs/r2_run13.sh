#!/bin/bash
# scratch GPU-box script of round 2: SQ counters of the alignment kernel, one workgroup per CU vs three
R=$GRAFT_REPO_ROOT
echo "== g1 mode0 (1 WG/CU)"; SVO_GROUPS=1 bash tools/pmc_sia.sh r2_pmc_a
echo "== g1 mode1 stg16 (3 WG/CU)"; SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so SVO_GROUPS=1 bash tools/pmc_sia.sh r2_pmc_b
echo "== 256 seqs g1 mode1 stg16 (1 WG/CU, same code as b)"; SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so SVO_GROUPS=1 bash tools/pmc_sia.sh r2_pmc_c --seqs 256
