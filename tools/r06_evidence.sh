#!/bin/bash
# usage (on the GPU box): tools/r06_evidence.sh tag -- round 6's profiles, all from ONE tree: (1) rocprofv3 kernel trace + stats of the driver's bench command with every block,
# (2) of every path alone (tools/iso_paths.py), (3) of the window search alone (tools/pmc_c3_match.py) and of the object BA alone (tools/ba_only.py), (4) the driver's bench
# command itself, untraced (its own PMC passes inside), (5) the -m gpu suite's log, (6) PMC passes (their own runs, no tracing): the score kernel on the bench batch, the
# region walks on 1 024 distinct frames, the BA's solver kernels (ba_schur_slots, ba_cr_eliminate: instruction mix, LDS, busy / wait cycles)
tag=${1:-r06}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
CUBESLAM_BENCH_NO_TRAFFIC=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_full -o res -- python $R/bench.py --steps 8 --warmup 2 --no-cpu > $out/bench_traced.json 2> $out/bench_traced.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_full -name "*.db" | head -1) > $out/full_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_iso -o res -- python $R/tools/iso_paths.py 1024 2 > $out/iso_paths.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_iso -name "*.db" | head -1) > $out/iso_paths_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_match -o res -- python $R/tools/pmc_c3_match.py 512 > $out/match_window.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_match -name "*.db" | head -1) > $out/match_window_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_ba -o res -- python $R/tools/ba_only.py 10 2 > $out/ba_only.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_${tag}_ba -name "*.db" | head -1) > $out/ba_kernel_stats.csv
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench_full.json
python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > $out/gpu_tests.log
grep -E "passed|failed" $out/gpu_tests.log | tail -2
# ---- PMC (separate passes, no tracing)
{
echo "# cuboid_sweep_score<512> on the bench batch (1024 frames x 3 boxes), rocprofv3 --pmc passes of tools/score_bench.py 1024 default (tools/run_pmc_cmd.sh)"
timeout 300 $R/tools/run_pmc_cmd.sh ${tag}_pmc_s1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" cuboid_sweep_score -- python $R/tools/score_bench.py 1024 default
} > $out/pmc_sweep_score.txt 2>&1
{
echo "# the two device region stages on 1 024 distinct frames (tools/lsd_wlk_check.py 1024 1024 seq,wlk; WLK_SHAPES=1,8,1), rocprofv3 --pmc, per launch"
WLK_SHAPES="1,8,1" CHECK_ORACLE=0 timeout 300 $R/tools/run_pmc_cmd.sh ${tag}_pmc_w1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" lsd_rg_ -- python $R/tools/lsd_wlk_check.py 1024 1024 seq,wlk
} > $out/pmc_lsd_walks.txt 2>&1
{
echo "# the object BA's solver kernels at config 5's bench graph (tools/pmc_ba.py: 1 000 key frames, 100 k points, 500 cuboids, 3 LM iterations), rocprofv3 --pmc, per launch"
timeout 300 $R/tools/run_pmc_cmd.sh ${tag}_pmc_b1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" ba_ -- python $R/tools/pmc_ba.py
timeout 300 $R/tools/run_pmc_cmd.sh ${tag}_pmc_b2 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" ba_ -- python $R/tools/pmc_ba.py
} > $out/pmc_ba.txt 2>&1
head -c 400 $out/bench_full.json; echo
head -6 $out/full_kernel_stats.csv | cut -c1-150
tail -12 $out/pmc_sweep_score.txt | cut -c1-120
tail -12 $out/pmc_lsd_walks.txt | cut -c1-120
tail -30 $out/pmc_ba.txt | cut -c1-140
