"""profiles/rNN_parity_errors.txt from one logged GPU test run:
   SRMAP_PARITY_LOG=$PWD/gpurun_out/rNN/parity_log.txt python -m pytest tests -m gpu -q
   python tools/parity_errors.py gpurun_out/rNN/parity_log.txt > profiles/rNN_parity_errors.txt
One line per test function and precision: how many error figures the run measured and the largest (max |a - ref| /
max(1, |ref|) per element, tests/parity_log.py), next to the bar the tests assert."""
import collections, re, sys
rows = collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    key, e = line.rstrip("\n").split("\t"); e = float(e)
    node = key.split(" ")[0]; what = key[len(node):].strip()
    base = node.split("[")[0]; par = node[len(base):]
    # precision of the case: the parametrised dtype (0 = f64, 1 = f32) is the first or last id component
    ids = par.strip("[]").split("-") if par else []
    f32 = any(i in ("1", "f32", "dtype1") for i in (ids[:1] + ids[-1:])) and e > 5e-10
    k = (base.replace("tests/", ""), what, "f32" if (f32 or e > 5e-10) else "f64")
    rows[k][0] += 1; rows[k][1] = max(rows[k][1], e)
print("measured parity errors of one `pytest tests -m gpu` run on one MI355X (max |a - ref| / max(1, |ref|) per element over")
print("every comparison the tests make; bars asserted: f64 1e-12, f32 2e-5 -- test_gpu_sharded: 1e-11 (two ranks, reduction order))")
print("%-78s %-22s %-4s %6s %10s" % ("test", "quantity", "type", "n", "max error"))
for (b, w, c), (n, mx) in sorted(rows.items()):
    print("%-78s %-22s %-4s %6d %10.2e" % (b, w or "gradient / operator", c, n, mx))
