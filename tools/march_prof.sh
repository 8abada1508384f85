#!/bin/bash
# tools/march_prof.sh [tag] [dtype] : per-kernel times (rocprofv3 --kernel-trace --stats) and SQ counters (two --pmc passes) of
# the cfg2 evaluation through the marching kernel (SRMAP_IMPL_MARCH) and the tile kernel (SRMAP_IMPL_TILED).
# Run through gpurun from the repo root; output: gpurun_out/<tag>/{march,tiled}_{stats.csv,pmc.txt}
root=$(pwd); tag=${1:-march_prof}; dt=${2:-f64}; out=$root/gpurun_out/$tag; mkdir -p $out
cat > /tmp/march_child.py <<PY
import os, sys
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.join("$root", "super-resolution_amd", "python"))
import srmap
W, s, K = 2048, 4, 16
f32 = "$dt" == "f32"
td = torch.float32 if f32 else torch.float64
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F32 if f32 else srmap.F64)
p.set_impl(int(os.environ["MARCH_IMPL"]))
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
y = torch.rand((K, 1, W // s, W // s), dtype=td, device="cuda", generator=gen)
x = torch.rand((1, W, W), dtype=td, device="cuda", generator=gen); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(int(os.environ.get("MARCH_N", "20"))): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for n in march tiled; do
  if [ $n = march ]; then export MARCH_IMPL=3; else export MARCH_IMPL=2; fi
  rm -rf /tmp/mp_$n
  MARCH_N=1500 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp_$n/t -o kt -- python /tmp/march_child.py > /dev/null 2>&1
  cp $(find /tmp/mp_$n/t -name '*kernel_stats.csv' | head -1) $out/${n}_stats.csv 2>/dev/null
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/mp_$n/a -o pmc -- python /tmp/march_child.py > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/mp_$n/b -o pmc -- python /tmp/march_child.py > /dev/null 2>&1
  python - <<PY | tee $out/${n}_pmc.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/mp_$n/[ab]/**/*counter_collection.csv', recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_eval' in r['Kernel_Name']:
            per[(r['Kernel_Name'].split('(')[0].split('<')[0][-10:] + '/' + r['Grid_Size'], r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (k, d, c), v in per.items(): acc[k][c].append(v)
for k, cs in acc.items():
    print('%-8s %s' % ('$n', k), {c[3:]: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(cs.items())})
PY
  echo "--- $n kernel stats"; cut -d, -f1-8 $out/${n}_stats.csv | sed 's/void srmap::(anonymous namespace):://' | cut -c1-200 | head -6
done
