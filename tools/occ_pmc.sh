#!/bin/bash
# tools/occ_pmc.sh: SQ instruction / wait counters of the tile kernel for the S = 4 (cfg2) and S = 2 instances of the occupancy
# probe (tools/occ_probe.py geometries: 2048^2, blur 3, BTV 3, K = S^2), two rocprofv3 --pmc passes each; means per launch in
# millions.  Run through gpurun from the repo root; stdout -> profiles/r06_occupancy.txt (1b).
root=$(pwd)
cat > /tmp/occ_pmc_child.py <<PY
import os, sys
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.join("$root", "super-resolution_amd", "python"))
import srmap
W = 2048; s = int(sys.argv[1]); K = s * s
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
for s in 4 2; do
  rm -rf /tmp/occpmc_$s
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/occpmc_$s/a -o pmc -- python /tmp/occ_pmc_child.py $s > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/occpmc_$s/b -o pmc -- python /tmp/occ_pmc_child.py $s > /dev/null 2>&1
  python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob('/tmp/occpmc_$s/*/**/*counter_collection.csv', recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_eval_z' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items(): acc[c].append(v)
m = {c[3:]: sum(v) / len(v) / 1e6 for c, v in acc.items()}
print('S = $s:', {k: round(v, 3) for k, v in sorted(m.items())})
print('       VALU per wave %.0f, SALU per wave %.0f, waves parked %.1f %% of their life (WAIT_ANY / WAVE_CYCLES)' % (
    m['INSTS_VALU'] / m['WAVES'], m['INSTS_SALU'] / m['WAVES'], 100 * m['WAIT_ANY'] / m['WAVE_CYCLES']))
PY
done
