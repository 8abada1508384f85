#!/bin/bash
# tools/march_pmc.sh <tag>: kernel-trace + PMC passes of bench.py (AUTO -> marching kernel) -> gpurun_out/<tag>/
tag=${1:-mp}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --dtype f64"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d $out/p1 -o pmc -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p2 -o pmc -- $B > /dev/null 2>&1
python - <<PY
import csv,glob
from collections import defaultdict
for f in glob.glob('$out/kt/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob('$out/p?/**/*counter_collection.csv',recursive=True):
    per=defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'][:40],r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    if 'k_eval' in k or 'k_march' in k:
        print(k, {c:round(sum(v)/len(v)/1e6,3) for c,v in sorted(cs.items())})
PY
