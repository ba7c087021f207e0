mkdir -p gpurun_out/r02i; O=gpurun_out/r02i
(time python -m pytest tests -m gpu -x -q --durations=6) > $O/gputest.log 2>&1; tail -8 $O/gputest.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/measure_all.sh r02i_m > $O/measure.log 2>&1; cat $O/measure.log | cut -c1-400
