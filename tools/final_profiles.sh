bash tools/profile_all.sh r03f > gpurun_out/r03f_all.log 2>&1
python tools/phase_clock/run.py > gpurun_out/r03f/phase_clock.txt 2>&1
python tools/phase_clock/timeline.py > gpurun_out/r03f/timeline.txt 2>&1
ls gpurun_out/r03f
