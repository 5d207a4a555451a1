#!/bin/bash
# (development) the headline under runner settings: name, then VAR=value ... for the environment, EXTRA for bench.py's flags
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu --no-ba $EXTRA > gpurun_out/s_$name.json 2> gpurun_out/s_$name.err; python -c "
import json,sys
d=json.load(open('gpurun_out/s_$name.json'))
k=d['kernels_us']
print('$name', round(d['value']), round(d['ms_per_step'],2), 'score frac', round(d['roofline']['frac'],3), 'seq', round(k['lsd_rg_seq']/1e3,1), 'improve', round(k['lsd_rg_improve']/1e3,1), 'emit', round(k['lsd_emit']/1e3,1), 'resize', round(k['lsd_resize']/1e3,1), 'fast', round(k['orb_fast_score']/1e3,1), 'select', round(k['cuboid_select']/1e3,1), 'hbm', d['hbm_in_use_gb'])
" || tail -3 gpurun_out/s_$name.err; }
export GPU_MAX_HW_QUEUES=16 BENCH_PRIO_LINES=1
EXTRA="" run base A=1
EXTRA="" run imp4 CUBESLAM_LIB=$PWD/devlib/libcs_imp4.so
EXTRA="" run imp5 CUBESLAM_LIB=$PWD/devlib/libcs_imp5.so
EXTRA="--cuboid-stream 1" run cubstream A=1
EXTRA="--line-workers 5" run w5 A=1
