#!/bin/bash
# bench.py at N = 2 / 4 / 8 with all ranks on GPU 0 over the real RCCL communicator (loopback sockets), the variants the
# driver's scaling run can take; every run under its own watchdog.  ->  gpurun_out/rccl_bench_stress.txt
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/rccl_bench_stress.txt; : > $out
run() {
  tag=$1; shift
  timeout 240 python bench.py --test-rccl-loopback --watchdog-s 100 --steps 20 --warmup 5 "$@" > /tmp/b.json 2> /tmp/b.err
  rc=$?
  python - "$tag" $rc >> $out <<'PY'
import json, sys
tag, rc = sys.argv[1], sys.argv[2]
try:
    j = json.load(open("/tmp/b.json"))
    c3, fv = j.get("cfg3") or {}, j.get("frames_variant") or {}
    print("%-22s rc %s n %d shard %-8s value %9.1f it/s  %.4f ms/step  cfg3 %s  frames_variant %s  [%s]" % (
        tag, rc, j["n_gpus"], j["config"]["shard"], j["value"], j["ms_per_step"],
        ("%.1f it/s (x%.2f vs N=1 in this run)" % (c3["value"], c3["speedup_vs_n1_in_this_run"])) if c3 else "-",
        ("%.1f" % fv["value"]) if fv else "-", j["config"]["comm_backend"]))
except Exception as e:
    print("%-22s rc %s FAILED %s" % (tag, rc, e))
    print(open("/tmp/b.err").read()[-3000:])
PY
}
for rep in 1 2; do
  for n in 2 4 8; do
    run "rows n$n #$rep" --gpus $n
    run "rows+overlap n$n #$rep" --gpus $n --overlap
    run "frames n$n #$rep" --gpus $n --shard frames
  done
done
run "channels n4" --gpus 4 --shard channels
cat $out
