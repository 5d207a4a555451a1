#!/bin/bash
# usage: tools/run_prof_cmd.sh tag filter -- command...   (rocprofv3 --kernel-trace --stats of any command; per-kernel averages into gpurun_out/<tag>/kernel_stats.csv)
tag=$1; filter=$2; shift 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- "$@" > $out/run.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db > $out/kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open('$out/kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if '$filter' in n: print('%-34s calls %5s avg %8.1f us min %8.1f' % (n[:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
