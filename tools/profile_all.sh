#!/bin/bash
# tools/profile_all.sh (run through gpurun from the repo root): the whole per-round measurement set into gpurun_out/r02b/ --
# profile_round.sh (bench, kernel stats, PMC passes), all configurations, sub-pixel path, solver trace, cfg5 per kernel,
# phase clock and chip-wide timeline of the timing-only build (tools/phase_clock/build.sh first), instruction issue rates.
set -x
bash tools/profile_round.sh r02b > /dev/null 2>&1
python tools/config_timing.py 1 2 3 4 5 2>&1 | grep cfg > gpurun_out/r02b/config_timing.txt
python tools/config_timing.py 3 4 5 --blur 2>&1 | grep cfg >> gpurun_out/r02b/config_timing.txt
python tools/subpixel_timing.py 2>&1 | tail -4 > gpurun_out/r02b/subpixel_timing.txt
bash tools/sp_prof.sh 2>&1 | tail -12 > gpurun_out/r02b/subpixel.txt
bash tools/solve_trace.sh 2>&1 | tail -16 > gpurun_out/r02b/solve_trace.txt
bash tools/cfg_prof.sh 5 16 r02b_cfg5 2>&1 | tail -12 > gpurun_out/r02b/cfg5_16ch.txt
python tools/phase_clock/run.py 2>&1 | tail -23 > gpurun_out/r02b/phase_clock.txt
python tools/phase_clock/timeline.py 2>&1 | tail -19 > gpurun_out/r02b/timeline.txt
tools/ubench/bin/valu_asm > gpurun_out/r02b/valu_rate.txt 2>&1
ls gpurun_out/r02b
