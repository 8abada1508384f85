root=$(pwd); out=$root/gpurun_out/sp1; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k subpixel 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python $root/tools/subpixel_timing.py > $out/kt.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$out/kt/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
