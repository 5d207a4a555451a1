#!/bin/bash
# (development, on the GPU box) ORB + cuboid of the headline (no line path) alone, and beside tools/ubench/walk_thrash: the walks' memory traffic without their instructions
run() { timeout 150 python bench.py --steps 20 --warmup 5 --no-lines --no-extra --no-cpu --no-ba 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_us']
print('$1', round(d['value']), round(d['ms_per_step'],2), *['%s %.1f' % (s, k[f]/1e3) for s,f in (('fast','orb_fast_score'),('qt','orb_quadtree'),('blur','orb_blur'),('cc','cuboid_canny_cc_local'),('score','cuboid_sweep_score'),('select','cuboid_select'))])"; }
run alone
tools/ubench/walk_thrash 2660 4096 45 7 & T=$!
sleep 8
run beside_thrash
kill $T 2>/dev/null; wait $T 2>/dev/null
