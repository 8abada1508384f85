#!/bin/bash
# tools/pmc_quick.sh <lib.so> <dtype> <tag>: instruction mix + wait breakdown of the fused kernel for one library build
root=$(pwd); out=$root/gpurun_out/$3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SRMAP_LIB=$root/$1 timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/a -o pmc -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype $2 > /dev/null 2>&1
SRMAP_LIB=$root/$1 timeout 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/b -o pmc -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype $2 > /dev/null 2>&1
cd $root
python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob('$out/*/**/*counter_collection.csv',recursive=True):
    per=defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_eval_fused' in r['Kernel_Name']: per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
    for (d,c),v in per.items(): acc[c].append(v)
print('$1 $2', {k[3:]:round(sum(v)/len(v)/1e6,2) for k,v in sorted(acc.items())})
PY
