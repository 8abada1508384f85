#!/bin/bash
# Kernel trace of a 2-rank sharded bench over the RCCL communicator (loopback, both ranks on GPU 0): which kernels a
# sharded step consists of -- the library's own and RCCL's.  tools/rccl_trace.sh  ->  gpurun_out/rccl_trace/summary.txt
root=$(pwd)
out=$root/gpurun_out/rccl_trace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shard in rows frames; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$shard -o kt -- \
    python $root/bench.py --gpus 2 --test-rccl-loopback --watchdog-s 120 --shard $shard --steps 200 --warmup 20 --clock-ramp-ms 0 \
      --min-timed-ms 0 --no-cfg3 --no-cpu-baseline > $out/$shard.json 2> $out/$shard.err
  echo "rc $? ($shard)"
done
cd $root
python - <<'PY' > $out/summary.txt
import csv, glob, os, json
for shard in ("rows", "frames"):
    print("== bench.py --gpus 2 --test-rccl-loopback --shard %s --steps 200 --warmup 20 (clock ramp off) ==" % shard)
    try:
        j = json.load(open("gpurun_out/rccl_trace/%s.json" % shard))
        print("bench line: %.4f ms/step (loopback sockets, both ranks on one GPU: not a performance figure); collective_per_step: %s"
              % (j["ms_per_step"], j["config"]["collective_per_step"]))
    except Exception as e:
        print("no bench line:", e)
    for f in sorted(glob.glob("gpurun_out/rccl_trace/%s/**/*kernel_stats.csv" % shard, recursive=True)):
        rows = list(csv.DictReader(open(f)))
        print("-- %s" % os.path.relpath(f, "gpurun_out/rccl_trace"))
        for r in rows[:12]:
            print("   %-110s calls %6s  avg %10.1f ns" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])))
PY
cat $out/summary.txt
