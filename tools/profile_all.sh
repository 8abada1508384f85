#!/bin/bash
# tools/profile_all.sh [tag] (run through gpurun from the repo root): the whole per-round measurement set into
# gpurun_out/<tag>/ -- profile_round.sh (bench, kernel stats, PMC passes), all configurations, sub-pixel path, solver
# trace, cfg5 per kernel, the Infinity Cache share of the bench number.
# Then:  python tools/collect_profiles.py gpurun_out/<tag> rNN ; copy the .txt summaries to profiles/rNN_*.txt
tag=${1:-r04}
set -x
bash tools/profile_round.sh $tag > /dev/null 2>&1
python tools/config_timing.py 1 2 3 4 5 2>&1 | grep cfg > gpurun_out/$tag/config_timing.txt
python tools/config_timing.py 3 4 5 --blur 2>&1 | grep cfg >> gpurun_out/$tag/config_timing.txt
python tools/subpixel_timing.py 2>&1 | tail -4 > gpurun_out/$tag/subpixel_timing.txt
bash tools/sp_prof.sh 2>&1 | tail -12 > gpurun_out/$tag/subpixel.txt
bash tools/solve_trace.sh ${tag}_solve 2>&1 | tail -16 > gpurun_out/$tag/solve_trace.txt
bash tools/cfg_prof.sh 5 16 ${tag}_cfg5 2>&1 | tail -12 > gpurun_out/$tag/cfg5_16ch.txt
python tools/hbm_fed_timing.py 2>&1 | tail -12 > gpurun_out/$tag/hbm_fed.txt
ls gpurun_out/$tag
