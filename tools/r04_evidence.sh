#!/bin/bash
# usage (on the GPU box): tools/r04_evidence.sh tag -- round 4's profiles: (1) rocprofv3 kernel trace + stats of the driver's bench command with every block, (2) of every path alone
# (tools/iso_paths.py), (3) the driver's bench command itself, untraced (its own PMC passes inside), (4) the -m gpu suite's log
tag=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
CUBESLAM_BENCH_NO_TRAFFIC=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_full -o res -- python $R/bench.py --steps 8 --warmup 2 --no-cpu > $out/bench_traced.json 2> $out/bench_traced.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_full -name "*.db" | head -1) > $out/full_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_iso -o res -- python $R/tools/iso_paths.py 1024 2 > $out/iso_paths.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_iso -name "*.db" | head -1) > $out/iso_paths_kernel_stats.csv
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench_full.json
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
tail -2 $out/gpu_tests.log
head -c 600 $out/bench_full.json; echo
head -8 $out/full_kernel_stats.csv | cut -c1-150
