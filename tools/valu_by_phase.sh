#!/bin/bash
# VALU/SALU/LDS instruction counts of the fused kernel by phase, via the ablation switches (profiling aid).
root=$(pwd); out=$root/gpurun_out/${1:-phase}; mkdir -p $out
export SRMAP_LIB=$root/super-resolution_amd/lib/libsrmap_prof.so  # python -c 'import __graft_entry__ as g; g.build_lib(profiling=True)'
cd /tmp && export TMPDIR=/tmp
for d in f64 f32; do
 for v in "all 0" "data 0" "reg 0" "all 1" "all 2" "all 3" "data 3"; do
  set -- $v
  SRMAP_DEBUG_SKIP=$2 timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/${d}_$1_$2 -o pmc -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype $d --terms $1 > /dev/null 2>&1
 done
done
