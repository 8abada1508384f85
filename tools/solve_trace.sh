# tools/solve_trace.sh <tag> [solve_profile.py options, e.g. --subpixel]: the solve's wall figures + its per-kernel times
root=$(pwd); tag=${1:-r06_solve}; shift
out=$root/gpurun_out/$tag; mkdir -p $out
python tools/solve_profile.py --irls 3 --cg 50 "$@" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python $root/tools/solve_profile.py --irls 3 --cg 50 "$@" > $out/kt.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$out/kt/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
