#!/bin/bash
# Two (grid: four) ranks on ONE GPU over the RCCL communicator: every rank poses as its own host (NCCL_HOSTID), RCCL runs
# its socket transport over loopback.  tools/rccl_loopback.sh [modes...]  ->  gpurun_out/rccl_loopback.txt
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
modes=${*:-frames rows channels grid}
out=gpurun_out/rccl_loopback.txt; : > $out
for m in $modes; do
  world=2; [ "$m" = grid ] && world=4
  port=$((20000 + RANDOM % 20000))
  res=/tmp/rccl_$m.json; rm -f $res
  pids=()
  for ((r = 0; r < world; ++r)); do
    NCCL_DEBUG=${NCCL_DEBUG:-WARN} timeout 300 python tests/dist_gpu_worker.py $r $world $port $m $res rccl > /tmp/rccl_${m}_$r.log 2>&1 &
    pids+=($!)
  done
  rc=0
  for p in "${pids[@]}"; do wait $p || rc=$?; done
  echo "== $m: world $world rc $rc" >> $out
  [ -f $res ] && cat $res >> $out && echo >> $out
  if [ $rc != 0 ]; then for ((r = 0; r < world; ++r)); do echo "-- rank $r"; tail -15 /tmp/rccl_${m}_$r.log; done >> $out; fi
done
cat $out
