#!/bin/bash
# tools/quick_prof.sh <tag> [dtype ...]: bench line + kernel-trace stats + instruction-mix / wait PMC passes of bench.py
# (run through gpurun from the repo root) -> gpurun_out/<tag>/
tag=${1:-qp}; shift
dts=${@:-f64}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for d in $dts; do
  B="python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype $d"
  (cd $root && timeout 200 python bench.py --steps 200 --warmup 20 --dtype $d --no-cpu-baseline > $out/bench_$d.json 2> $out/bench_$d.err)
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$d -o kt -- python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --dtype $d > $out/kt_$d.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sqa_$d -o pmc -- $B > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sqb_$d -o pmc -- $B > /dev/null 2>&1
  cat $out/bench_$d.json
  python - <<PY
import csv,glob
from collections import defaultdict
for f in glob.glob('$out/kt_$d/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob('$out/sq?_$d/**/*counter_collection.csv',recursive=True):
    per=defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'][:40],r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    if 'k_eval' in k or 'k_ring' in k or 'reduce' in k:
        print(k, {c[3:]:round(sum(v)/len(v)/1e6,3) for c,v in sorted(cs.items())})
PY
done
