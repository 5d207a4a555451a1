#!/bin/bash
# usage: tools/run_pmc_cmd.sh tag "COUNTER1 COUNTER2 ..." kernel_filter -- command...   (rocprofv3 --pmc pass of any command, no tracing; per-kernel averages)
tag=$1; ctrs=$2; filt=$3; shift 4
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -o res -- "$@" > $out/run.log 2>&1
echo "== $tag: $ctrs"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_$tag $filt | tee -a $out/pmc.txt
