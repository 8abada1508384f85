root=$(pwd); out=$root/gpurun_out/r02_abl; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in data reg all; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$t -o kt -- python $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline --terms $t > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sqa_$t -o pmc -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --terms $t > /dev/null 2>&1
  echo "== $t"; python - <<PY
import csv,glob
for f in glob.glob('$out/kt_$t/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
done
cd $root; for t in data reg all; do mkdir -p $out/x; rm -rf $out/x/*; cp -r $out/sqa_$t $out/x/sqa_f64; echo $t; python tools/pmc_summary.py $out/x; done
