import sys, json, time
sys.path.insert(0, '/root/repo')
import bench
from cube_slam_amd import _lib
ctx = _lib.Context(0)
r = bench.c3_bench(ctx, int(sys.argv[1]), int(sys.argv[2]), with_cpu=False, with_traffic=False, with_small_window=0)
print(json.dumps({k: v for k, v in r.items() if k not in ("local_map",)}))
