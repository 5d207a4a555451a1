#!/usr/bin/env python3
"""Per kernel of a rocprofv3 kernel trace: launches, average duration, waves per launch, nanoseconds of the whole chip per wave, registers -- a kernel whose time follows
its wave count rather than its bytes is bound by the instructions it issues.  usage: tools/kernel_waves.py results.db [min_us]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
agg = collections.OrderedDict()
for n, s, e, gx, gy, gz, wx, wy, wz, vg, lds in c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, vgpr_count, lds_size from kernels order by start"):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    a = agg.setdefault(n, [0, 0.0, 0, vg, lds, wx * wy * wz])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (gx * gy * gz + 63) // 64
print("%-28s %6s %10s %10s %9s %5s %7s %5s" % ("kernel", "calls", "avg us", "waves", "ns/wave", "vgpr", "lds", "wg"))
for n, (k, us, waves, vg, lds, wg) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if us / k >= min_us:
        print("%-28s %6d %10.1f %10d %9.2f %5d %7d %5d" % (n[:28], k, us / k, waves // k, 1e3 * us / max(waves, 1), vg, lds, wg))
