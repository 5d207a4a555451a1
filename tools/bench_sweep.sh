#!/bin/bash
# (development) the headline under runner / kernel-shape variants: tools/bench_sweep.sh   (run on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "== LW=$LW $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu --no-ba --line-workers $LW 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_us']
print('frames/s %.0f  ms/step %.2f  score %.0f us  wlk %.0f us seq %.0f improve %.0f  emit %.0f  fast %.0f  hbm %.1f GB' % (d['value'], d['ms_per_step'], k['cuboid_sweep_score'], k.get('lsd_rg_wlk', 0), k.get('lsd_rg_seq', 0), k['lsd_rg_improve'], k['lsd_emit'], k['orb_fast_score'], d['hbm_in_use_gb']))"; }
for LW in 5 6 8; do run A=1; done
LW=6
run CUBESLAM_LSD_SEQ_WPB=16
run CUBESLAM_SCORE_THREADS=512
