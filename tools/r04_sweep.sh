#!/bin/bash
# (development, on the GPU box) the headline under runner settings, one line per variant.  Every argument is "name|VAR=value VAR=value|bench.py flags", e.g.
#   tools/r04_sweep.sh "default||" "five|BENCH_PRIO_LINES=0|--line-workers 5" "q4|GPU_MAX_HW_QUEUES=4 CUBESLAM_LSD_WALK_BG=0|"
# prints frames/s, ms per step, cuboid_sweep_score's fraction in the timed region and the in-run ms per launch of a few kernels (profiles/r04_c_runner_sweeps.txt was made with it).
mkdir -p gpurun_out
for v in "$@"; do
    IFS='|' read -r name envs flags <<< "$v"
    env $envs timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu --no-ba $flags > gpurun_out/s_$name.json 2> gpurun_out/s_$name.err
    python - "$name" <<'PY' || tail -3 gpurun_out/s_$name.err
import json, sys
n = sys.argv[1]
d = json.load(open("gpurun_out/s_%s.json" % n)); k = d["kernels_us"]
print(n, round(d["value"]), round(d["ms_per_step"], 2), "score frac", round(d["roofline"]["frac"], 3), *["%s %.1f" % (s, k[f] / 1e3) for s, f in (("seq", "lsd_rg_seq"), ("improve", "lsd_rg_improve"), ("emit", "lsd_emit"), ("resize", "lsd_resize"), ("fast", "orb_fast_score"), ("qt", "orb_quadtree"), ("select", "cuboid_select"), ("cc", "cuboid_canny_cc_local"))], "hbm", d["hbm_in_use_gb"])
PY
done
