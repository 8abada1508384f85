#!/bin/bash
# tools/phase_clock/build.sh: TIMING-ONLY build of the tile kernel with s_memtime stamps at the phase boundaries
# (patch_kernel.py instruments a COPY of kernels_ztile.hip under /tmp; the product sources and libsrmap.so are not
# touched).  Output: tools/phase_clock/libsrmap_time.so (git-ignored), run with tools/phase_clock/run.py on the GPU box.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/spx
python "$root/tools/phase_clock/patch_kernel.py" "$root"
cd /tmp/spx && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false -DSRMAP_MEASUREMENT_BUILD -I"$root/include" -I"$root/super-resolution_amd/csrc" -c kz_time.hip -o kz_time.o
cd "$root/super-resolution_amd/lib"
objs=$(ls *.hip.o | grep -v kernels_ztile)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/phase_clock/libsrmap_time.so" $objs /tmp/spx/kz_time.o -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo built "$root/tools/phase_clock/libsrmap_time.so"
