#!/bin/bash
# usage: tools/run_prof.sh tag [ENV=VAL ...] -- tools/score_bench.py under rocprofv3 --kernel-trace; per-kernel averages into gpurun_out/<tag>/
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- python $GRAFT_REPO_ROOT/tools/score_bench.py 128 default > $out/sb.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db > $out/kernel_stats.csv
echo "== $tag $@"
python - <<PY
import csv
for r in csv.DictReader(open('$out/kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if 'cuboid' in n: print('%-28s calls %3s avg %8.1f us min %8.1f' % (n[:28], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
grep segments $out/sb.log | sed 's/.*alg/alg/'
