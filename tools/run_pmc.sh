#!/bin/bash
# usage: tools/run_pmc.sh tag "COUNTER1 COUNTER2 ..." [ENV=VAL ...] -- tools/score_bench.py under rocprofv3 --pmc (no tracing), averages per kernel
tag=$1; ctrs=$2; shift 2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
env "$@" rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -o res -- python $GRAFT_REPO_ROOT/tools/score_bench.py 128 default > $out/sb.log 2>&1
echo "== $tag: $ctrs"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_$tag cuboid_sweep_score | tee $out/pmc.txt
