tag=${1:-r04}
bash tools/profile_all.sh $tag > gpurun_out/${tag}_all.log 2>&1
python tools/phase_clock/run.py > gpurun_out/$tag/phase_clock.txt 2>&1
python tools/phase_clock/timeline.py > gpurun_out/$tag/timeline.txt 2>&1
ls gpurun_out/$tag
