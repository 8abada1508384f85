#!/bin/bash
# Measurement build of the WHOLE library (for variants that change a shared header: every object is recompiled):
#   tools/exp_build_all.sh <name> [-DSRMAP_EXP_...=v ...]   -> gpurun_ab/<name>/libsrmap.so
# kernels_ztile.hip is compiled with SRMAP_ZT_ONLY_CFG2 (the cfg2 instance only) unless SRMAP_EXP_FULL=1.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/gpurun_ab/$name; mkdir -p $out
CS=$ROOT/super-resolution_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false -Wno-invalid-offsetof -I$ROOT/include -I$CS -DSRMAP_MEASUREMENT_BUILD"
pids=""
for src in $CS/*.hip; do
  b=$(basename $src)
  extra=""
  if [ "$b" = "kernels_ztile.hip" ] && [ "${SRMAP_EXP_FULL:-0}" != "1" ]; then extra="-DSRMAP_ZT_ONLY_CFG2"; fi
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $src -o $out/$b.o 2> $out/$b.log &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsrmap.so $out/*.hip.o \
  -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo built $out/libsrmap.so
