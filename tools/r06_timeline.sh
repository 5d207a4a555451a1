#!/bin/bash
# The alternating runner's timeline: rocprofv3 kernel trace of the headline command, then per queue the long kernels and the gaps between them (tools/stream_gaps.py).
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r06_timeline
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --no-ba --steps 12 --warmup 5 > $out/run.log 2>&1
db=$(find /tmp/prof_tl -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/stream_gaps.py $db 1.0 @orb_resize:70 60 > $out/gaps.txt 2>&1
tail -1 $out/run.log | cut -c1-300
wc -l $out/gaps.txt
