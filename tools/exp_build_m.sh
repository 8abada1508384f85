#!/bin/bash
# Measurement build of the marching kernel:  tools/exp_build_m.sh <name> [-DSRMAP_EXP_...=v ...]
# Compiles csrc/kernels_zmarch.hip with the given switches and links it with the product's other objects into
# gpurun_ab/<name>/libsrmap.so (git-ignored; travels to the GPU box).  Select it with SRMAP_LIB=... (Python binding).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/gpurun_ab/$name; mkdir -p $out
CS=$ROOT/super-resolution_amd/csrc; LD=$ROOT/super-resolution_amd/lib
mkdir -p /tmp/isa/$name
(cd /tmp/isa/$name && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false \
  -Wno-invalid-offsetof -I$ROOT/include -I$CS "$@" -c $CS/kernels_zmarch.hip -save-temps -o $out/kernels_zmarch.hip.o)
objs=$(ls $LD/*.hip.o | grep -v kernels_zmarch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsrmap.so $out/kernels_zmarch.hip.o $objs \
  -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
python3 $ROOT/tools/kres.py /tmp/isa/$name/kernels_zmarch-hip-amdgcn-amd-amdhsa-gfx950.s k_eval_m
echo built $out/libsrmap.so
