#!/bin/bash
# usage (on the GPU box): tools/r03_evidence.sh tag  -- the profiles VERDICT r2 asked for: kernel trace of bench.py WITH the c3 / c4 / pcie / ba blocks,
# kernel trace of the matcher + PoseOptimization tests, PMC traffic of ba_schur_slots (two separate --pmc passes, no tracing)
tag=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
# 1. the bench command with every block (nested rocprofv3 --pmc passes of the bench are switched off under the tracer)
CUBESLAM_BENCH_NO_TRAFFIC=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_full -o res -- python $R/bench.py --steps 8 --warmup 2 --no-cpu > $out/bench_full.json 2> $out/bench_full.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_full -name "*.db" | head -1) > $out/full_kernel_stats.csv
# 2. matcher + pose-only optimisation at config-3 size (tests/test_match_gpu.py, test_pose_gpu.py are the only callers at that size)
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_mp -o res -- python -m pytest -q -x -m gpu $R/tests/test_match_gpu.py $R/tests/test_pose_gpu.py > $out/match_pose.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_mp -name "*.db" | head -1) > $out/match_pose_kernel_stats.csv
# 3. HBM-side traffic of the BA's Schur kernel
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  t=$(echo $pass | cut -c9-10)
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_${tag}_ba_$t -o res -- python $R/tools/pmc_ba.py > $out/pmc_ba_$t.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_${tag}_ba_$t ba_schur >> $out/pmc_ba.txt
done
cat $out/pmc_ba.txt
head -12 $out/full_kernel_stats.csv | cut -c1-150
