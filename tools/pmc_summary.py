"""tools/pmc_summary.py <gpurun_out/tag> [dtype]: per-kernel means (millions per launch) of the SQ counters collected by
tools/quick_prof.sh."""
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/sq?_%s/**/*counter_collection.csv" % dt, recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("k_eval_z", "k_border", "k_eval_fused", "k_reduce_partials", "k_finish") if k in n), None)
        if key:
            per[(key, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    print(k, {c[3:]: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(cs.items())})
