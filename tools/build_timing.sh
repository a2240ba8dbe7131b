#!/bin/bash
# tools only: tools/build/t/libmuse_hip.so = the library with tools/experiments/gemm_pp.hip linked in (-DMM_TOOLS_PP: the MM_PP / MM_PP_ABL / MM_PP_DELAY environment
# switches exist ONLY in this build) and gemm_wide.hip compiled with -DMM_GEMM_TIMING (in-kernel cycle stamps), and
# tools/build/gemm_harness_t linked against it.  usage: tools/build_timing.sh ; MM_PP=230 tools/build/gemm_harness_t stamps
set -e
P=muse_maskgit_pytorch_amd
mkdir -p tools/build/t
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -DMM_GEMM_TIMING -DMM_TOOLS_PP -I$P/csrc"
/opt/rocm/bin/hipcc $FL -c tools/experiments/gemm_pp.hip -o tools/build/t/gemm_pp.o &
/opt/rocm/bin/hipcc $FL -c $P/csrc/gemm_wide.hip -o tools/build/t/gemm_wide.o &
wait
objs=$(ls $P/build/*.o | grep -v "/gemm_pp.o\|/gemm_wide.o\|_exp")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/build/t/libmuse_hip.so $objs tools/build/t/gemm_pp.o tools/build/t/gemm_wide.o -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/build/gemm_harness_t tools/gemm_harness.cpp -Ltools/build/t -lmuse_hip -ldl -Wl,-rpath,'$ORIGIN/t'
ls -la tools/build/t/libmuse_hip.so tools/build/gemm_harness_t
