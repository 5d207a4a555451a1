#!/bin/bash
# (development) the headline under runner settings: name, then VAR=value ... for the environment, EXTRA for bench.py's flags
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu --no-ba $EXTRA > gpurun_out/s_$name.json 2> gpurun_out/s_$name.err; python -c "
import json,sys
d=json.load(open('gpurun_out/s_$name.json'))
k=d['kernels_us']
print('$name', round(d['value']), round(d['ms_per_step'],2), 'score frac', round(d['roofline']['frac'],3), 'seq', round(k['lsd_rg_seq']/1e3,1), 'improve', round(k['lsd_rg_improve']/1e3,1), 'emit', round(k['lsd_emit']/1e3,1), 'resize', round(k['lsd_resize']/1e3,1), 'fast', round(k['orb_fast_score']/1e3,1), 'qt', round(k['orb_quadtree']/1e3,1), 'select', round(k['cuboid_select']/1e3,1), 'cc', round(k['cuboid_canny_cc_local']/1e3,1), 'hbm', d['hbm_in_use_gb'])
" || tail -3 gpurun_out/s_$name.err; }
EXTRA="" run base A=1
EXTRA="" run rgs6 CUBESLAM_LIB=$PWD/devlib/libcs_rgs6.so
EXTRA="" run rgs8 CUBESLAM_LIB=$PWD/devlib/libcs_rgs8.so
EXTRA="--cuboid-stream 1" run rgs6cub CUBESLAM_LIB=$PWD/devlib/libcs_rgs6.so
