#!/bin/bash
# tools/exp_pmc.sh name1 [name2 ...]: SQ instruction-mix / wait counters of the cfg2 evaluation through measurement builds
# (gpurun_ab/<name>/libsrmap.so; `product` = the product library), two rocprofv3 --pmc passes each, means per launch in
# millions -> stdout and gpurun_out/exp_pmc/<name>.txt.  Run through gpurun from the repo root.
root=$(pwd); mkdir -p $root/gpurun_out/exp_pmc
cat > /tmp/exp_pmc_child.py <<PY
import os, sys
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.join("$root", "super-resolution_amd", "python"))
import srmap
W, s, K = 2048, 4, 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda", generator=gen)
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda", generator=gen); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(20): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" != product ]; then export SRMAP_LIB=$root/gpurun_ab/$n/libsrmap.so; else unset SRMAP_LIB; fi
  rm -rf /tmp/pmc_$n
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_$n/a -o pmc -- python /tmp/exp_pmc_child.py > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_$n/b -o pmc -- python /tmp/exp_pmc_child.py > /dev/null 2>&1
  python - <<PY | tee $root/gpurun_out/exp_pmc/$n.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/pmc_$n/*/**/*counter_collection.csv', recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_eval' in r['Kernel_Name']:
            per[(r['Kernel_Name'][:60].split('<')[0].split('(')[0][-12:], r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (k, d, c), v in per.items(): acc[k][c].append(v)
for k, cs in acc.items():
    print('%-14s %s' % ('$n', k), {c[3:]: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(cs.items())})
PY
done
