#!/bin/bash
# tools only: build libmuse_exp<N>.so = the library with csrc/<file>.hip compiled with -DMM_EXP=<N>  (A/B experiments, selected with MM_LIB)
set -e
N=$1; F=${2:-gemm}; EXTRA=$3
P=/root/repo/muse_maskgit_pytorch_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -DMM_EXP=$N $EXTRA -c $P/csrc/$F.hip -o $P/build/${F}_exp$N.o 2>&1 | grep -i "error" || true
objs=$(ls $P/build/*.o | grep -v "/$F.o\|_exp")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmuse_exp$N.so $objs $P/build/${F}_exp$N.o
ls -la $P/libmuse_exp$N.so
