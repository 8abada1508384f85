#!/bin/bash
# tools/cfg_prof.sh <cfg> <channels> [tag]: per-kernel times of one BASELINE configuration at a reduced channel count
cfg=${1:-5}; ch=${2:-16}; tag=${3:-cfgprof}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python $root/tools/config_timing.py $cfg --channels $ch > $out/kt.log 2>&1
tail -2 $out/kt.log
python - <<PY
import csv,glob
for f in glob.glob('$out/kt/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:100], r['Calls'], r['AverageNs'], r['Percentage'])
PY
