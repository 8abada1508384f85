#!/bin/bash
# Collects the per-round profile set on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# kernel trace + stats, then PMC passes kept separate (FETCH_SIZE / WRITE_SIZE do not fit one pass).
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for d in f64 f32; do
  B="python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-fed --no-cfg3 --no-other-precision --dtype $d"
  BK="python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hbm-fed --no-cfg3 --no-other-precision --dtype $d"
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$d -o kt -- $BK > $out/kt_$d.log 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$d -o pmc -- $B > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$d -o pmc -- $B > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sqa_$d -o pmc -- $B > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sqb_$d -o pmc -- $B > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm_$d -o pmc -- $B > /dev/null 2>&1
  cd $root && timeout 400 python bench.py --dtype $d > $out/bench_$d.json 2> $out/bench_$d.err; cd /tmp
done
cd $root
