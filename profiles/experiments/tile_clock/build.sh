#!/bin/bash
# Builds a copy of the library whose conv kernel records per-tile phase clocks (-DDIGA_TILE_CLOCK); the product build has none.
set -e
cd "$(dirname "$0")"
SRC=../../../diga_amd/csrc
make -C $SRC -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DDIGA_TILE_CLOCK -I$SRC -c $SRC/conv_igemm.hip -o conv_igemm_clock.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdiga_hip_clock.so conv_igemm_clock.o $SRC/bn.o $SRC/stem.o $SRC/sam.o $SRC/tail.o $SRC/input.o $SRC/ccl.o $SRC/xgmi.o
