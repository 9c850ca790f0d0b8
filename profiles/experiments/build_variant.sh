#!/bin/bash
# build_variant.sh NAME -DFLAG...: a copy of the library with conv_igemm.hip compiled with extra flags -> profiles/experiments/variants/libdiga_hip_NAME.so
set -e
cd "$(dirname "$0")"
SRC=../../diga_amd/csrc
NAME=$1; shift
mkdir -p variants
make -C $SRC -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -I$SRC -c $SRC/conv_igemm.hip -o variants/conv_igemm_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libdiga_hip_$NAME.so variants/conv_igemm_$NAME.o $SRC/bn.o $SRC/stem.o $SRC/sam.o $SRC/tail.o $SRC/input.o $SRC/ccl.o $SRC/xgmi.o
rm variants/conv_igemm_$NAME.o
