#!/bin/bash
# Compiler-flag sweep of the cfg2 f64 tile kernel (measurement builds, tools/exp_build.sh): one library per flag set under
# gpurun_ab/fs_<n>/, the list in gpurun_ab/fs_list.txt.  tools/flag_sweep.sh build | run
cd "$(dirname "$0")/.." || exit 1
FLAGS=(
  ""
  "-mllvm -amdgpu-sched-strategy=max-ilp"
  "-mllvm -amdgpu-sched-strategy=max-memory-clause"
  "-mllvm -amdgpu-sched-strategy=iterative-ilp"
  "-mllvm -amdgpu-sched-strategy=iterative-minreg"
  "-mllvm -amdgpu-sched-strategy=iterative-maxocc"
  "-mllvm -amdgpu-schedule-metric-bias=0"
  "-mllvm -amdgpu-schedule-metric-bias=100"
  "-mllvm -amdgpu-set-wave-priority"
  "-mllvm -amdgpu-use-amdgpu-trackers"
  "-mllvm -enable-post-misched=false"
  "-mllvm -amdgpu-kernarg-preload-count=16"
  "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule"
  "-mllvm -amdgpu-disable-clustered-low-occupancy-reschedule"
  "-mllvm -amdgpu-early-ifcvt"
  "-mllvm -misched-cluster=false"
  "-mllvm -amdgpu-max-memory-clause=4"
  "-mllvm -amdgpu-max-memory-clause=32"
  "-mllvm -amdgpu-load-store-vectorizer=false"
  "-mllvm -amdgpu-use-aa-in-codegen=false"
  "-mllvm -amdgpu-dpp-combine=false"
  "-mllvm -amdgpu-opt-vgpr-liverange=false"
  "-mllvm -amdgpu-enable-pre-ra-optimizations=false"
  "-mllvm -amdgpu-scalarize-global-loads=false"
  "-O2"
  "-mllvm -misched-prera-direction=topdown"
  "-mllvm -misched-prera-direction=bottomup"
  "-mllvm -sched-high-latency-cycles=40"
)
if [ "$1" = build ]; then
  mkdir -p gpurun_ab; : > gpurun_ab/fs_list.txt
  n=0
  for f in "${FLAGS[@]}"; do
    ( if tools/exp_build.sh fs_$n $f -Rpass-analysis=kernel-resource-usage > gpurun_ab/fs_$n.log 2>&1; then
        sp=$(grep -A8 "k_eval_zIdLi4ELi3ELi2ELi3ELb0ELb0" gpurun_ab/fs_$n.log | grep -E "VGPRs:|VGPRs Spill|SGPRs Spill" | sed 's/.*remark: *//; s/ \[.*//' | tr '\n' ' ')
        echo "fs_$n | $f | $sp"
      else echo "fs_$n | $f | BUILD FAILED"; rm -rf gpurun_ab/fs_$n; fi ) >> gpurun_ab/fs_list.txt &
    n=$((n + 1))
    if (( n % 8 == 0 )); then wait; fi
  done
  wait
  sort -t_ -k2 -n gpurun_ab/fs_list.txt
else
  out=gpurun_out/flag_sweep.txt; : > $out
  for rep in 1 2; do
    for d in $(ls -d gpurun_ab/fs_* | grep -v '\.log' | sort -t_ -k3 -n); do
      [ -f $d/libsrmap.so ] || continue
      ms=$(SRMAP_LIB=/root/repo/$d/libsrmap.so python bench.py --no-other-precision --no-cpu-baseline --no-hbm-fed --no-cfg3 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % (1e3*j['config']['device_ms_per_step']))")
      echo "$rep $(basename $d) $ms" >> $out
    done
  done
  cat $out
fi
