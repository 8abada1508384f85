"""tools/collect_profiles.py <gpurun_out/tag> <round-prefix, e.g. r02>: copies the summaries of a tools/profile_round.sh
run into profiles/ under the names bench.py and the docs cite:
  <r>_bench_<dtype>.json, <r>_bench_<dtype>_kernel_stats.csv, <r>_bench_hbm_pmc.json, <r>_evalz_<dtype>_pmc.json"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
KERNEL = "k_eval_z"


def counters(pattern):
    acc = defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, c), v in per.items():
            acc[c].append(v)
    return {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in acc.items()}


hbm = {}
for d in ("f64", "f32"):
    b = os.path.join(src, "bench_%s.json" % d)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(dst, "%s_bench_%s.json" % (rnd, d)))
    ks = glob.glob(os.path.join(src, "kt_%s" % d, "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "%s_bench_%s_kernel_stats.csv" % (rnd, d)))
    fe = counters(os.path.join(src, "fetch_%s" % d, "**", "*counter_collection.csv"))
    wr = counters(os.path.join(src, "write_%s" % d, "**", "*counter_collection.csv"))
    if fe and wr:
        hbm[d] = {"kernel": KERNEL, "FETCH_SIZE": fe["FETCH_SIZE"]["mean"], "fetch_n": fe["FETCH_SIZE"]["dispatches"],
                  "WRITE_SIZE": wr["WRITE_SIZE"]["mean"], "write_n": wr["WRITE_SIZE"]["dispatches"]}
    sq = {}
    for part in ("sqa", "sqb", "grbm"):
        sq.update(counters(os.path.join(src, "%s_%s" % (part, d), "**", "*counter_collection.csv")))
    if sq:
        json.dump({"kernel": KERNEL, "dtype": d, "counters_mean_per_launch": {k: v["mean"] for k, v in sorted(sq.items())}},
                  open(os.path.join(dst, "%s_evalz_%s_pmc.json" % (rnd, d)), "w"), indent=1)
if hbm:
    json.dump(hbm, open(os.path.join(dst, "%s_bench_hbm_pmc.json" % rnd), "w"), indent=1)
print("profiles written:", sorted(f for f in os.listdir(dst) if f.startswith(rnd)))
