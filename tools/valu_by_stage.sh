#!/bin/bash
# Cumulative instruction counts of the fused kernel when it stops after stage n (SRMAP_DEBUG_STOP): differences = per-stage cost.
root=$(pwd); out=$root/gpurun_out/${1:-stage}; mkdir -p $out; d=${2:-f64}
export SRMAP_LIB=$root/super-resolution_amd/lib/libsrmap_prof.so  # python -c 'import __graft_entry__ as g; g.build_lib(profiling=True)'
cd /tmp && export TMPDIR=/tmp
for st in 0 1 3 4 5 6 7 8 9 10 11 12 13 99; do
  SRMAP_DEBUG_STOP=$st timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/s$st -o pmc -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype $d > /dev/null 2>&1
done
cd $root
python - <<PY
import csv,glob
from collections import defaultdict
prev=None
for st in [0,1,3,4,5,6,7,8,9,10,11,12,13,99]:
    acc=defaultdict(list)
    for f in glob.glob('$out/s%d/**/*counter_collection.csv'%st,recursive=True):
        per=defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_eval_fused' in r['Kernel_Name']: per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
        for (dd,c),v in per.items(): acc[c].append(v)
    cur={k[3:]:sum(v)/len(v)/1e6 for k,v in acc.items()}
    keys=['INSTS_VALU','INSTS_SALU','INSTS_LDS','INSTS_VMEM','INSTS_SMEM','WAVE_CYCLES','BUSY_CYCLES']
    print('stop %2d '%st+' '.join('%s=%7.2f'%(k[6:] if k.startswith('INSTS') else k[:4],cur.get(k,0)) for k in keys)+('   dVALU=%6.2f'%(cur.get('INSTS_VALU',0)-prev) if prev is not None else ''))
    prev=cur.get('INSTS_VALU',0)
PY
