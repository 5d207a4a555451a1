#!/usr/bin/env python3
"""Every dispatch of one kernel in a rocprofv3 kernel trace (rocpd database): duration, grid, workgroup -- e.g. the seven levels of orb_resize.
usage: tools/dispatch_list.py results.db kernel_substring [max_rows]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)).fetchall()
for n, s, e, gx, gy, gz, wx, vg, lds in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-40s %9.1f us  grid %6d x %4d x %5d  wg %4d  vgpr %3d lds %6d" % (n.replace("(anonymous namespace)::", "").split("(")[0][:40], (e - s) / 1e3, gx, gy, gz, wx, vg, lds))
