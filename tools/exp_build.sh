#!/bin/bash
# Measurement build of the tile kernel:  tools/exp_build.sh <name> [-DSRMAP_EXP_...=v ...]
# Compiles csrc/kernels_ztile.hip with SRMAP_ZT_ONLY_CFG2 (the cfg2 instance only) plus the given switches and links it
# with the product's other objects into gpurun_ab/<name>/libsrmap.so (git-ignored; travels to the GPU box).  Select it
# with SRMAP_LIB=gpurun_ab/<name>/libsrmap.so (Python binding).  The product library is untouched.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/gpurun_ab/$name; mkdir -p $out
CS=$ROOT/super-resolution_amd/csrc; LD=$ROOT/super-resolution_amd/lib
src=${SRMAP_EXP_SRC:-$CS/kernels_ztile.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false \
  -Wno-invalid-offsetof -DSRMAP_MEASUREMENT_BUILD -I$ROOT/include -I$CS -DSRMAP_ZT_ONLY_CFG2 "$@" -c $src -o $out/kernels_ztile.hip.o
objs=$(ls $LD/*.hip.o | grep -v kernels_ztile)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsrmap.so $out/kernels_ztile.hip.o $objs \
  -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo built $out/libsrmap.so
