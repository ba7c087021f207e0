#!/bin/bash
# GPU box: round 6 baseline of a build — bench line, C2 / C3 timelines (usage: r06_base.sh TAG)
V=${1:-r06base}
O=gpurun_out/$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
python tools/bsum.py C2-full < $O/bench.json
bash tools/timeline.sh $V/tl_c2 C2 1.0 0.6
bash tools/timeline.sh $V/tl_c3 C3 3.0 2.0
