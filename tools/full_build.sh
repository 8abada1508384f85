#!/bin/bash
# Full (all instances) measurement build of the tile kernels:  tools/full_build.sh <name> [-D...]  -> gpurun_ab/<name>/libsrmap.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/gpurun_ab/$name; mkdir -p $out
CS=$ROOT/super-resolution_amd/csrc; LD=$ROOT/super-resolution_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false \
  -Wno-invalid-offsetof -DSRMAP_MEASUREMENT_BUILD -I$ROOT/include -I$CS "$@" -c $CS/kernels_ztile.hip -o $out/kernels_ztile.hip.o
objs=$(ls $LD/*.hip.o | grep -v kernels_ztile)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsrmap.so $out/kernels_ztile.hip.o $objs \
  -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo built $out/libsrmap.so
