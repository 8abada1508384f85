"""Averages rocprofv3 counter_collection CSVs per kernel: python tools/summarize_pmc.py gpurun_out/<tag>"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def summarize(root):
    res = {}
    for d in sorted(glob.glob(os.path.join(root, "*_f64")) + glob.glob(os.path.join(root, "*_f32"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            per_dispatch = defaultdict(lambda: defaultdict(float))
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
                per_dispatch[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            for (k, _), cs in per_dispatch.items():
                for c, v in cs.items():
                    acc[k][c].append(v)
            for k, cs in acc.items():
                for c, v in cs.items():
                    res.setdefault(os.path.basename(d).split("_")[-1], {}).setdefault(k, {})[c] = {
                        "mean": sum(v) / len(v), "dispatches": len(v)}
    return res


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1]), indent=1))
