#!/bin/bash
tag=${1:-mq}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype f64"
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o pmc -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/w -o pmc -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $out/t -o pmc -- $B > /dev/null 2>&1
python - <<PY
import csv
from collections import defaultdict
for sub in ('f','w','t'):
    try:
        acc=defaultdict(list); per=defaultdict(float)
        for r in csv.DictReader(open('$out/'+sub+'/pmc_counter_collection.csv')):
            if 'k_eval_march' not in r['Kernel_Name'] and 'k_eval_z' not in r['Kernel_Name']: continue
            per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
        for (d,c),v in per.items(): acc[c].append(v)
        print(sub, {c:round(sum(v)/len(v)/1e6,3) for c,v in sorted(acc.items())})
    except Exception as e: print(sub, 'ERR', e)
PY
