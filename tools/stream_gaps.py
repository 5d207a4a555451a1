#!/usr/bin/env python3
"""Per-stream timeline of a rocprofv3 kernel trace (rocpd database): for every queue the kernels in time order, runs of short kernels folded, and the idle
gaps between them -- where a line detector's cycle goes besides its kernels.  usage: tools/stream_gaps.py results.db [min_gap_ms]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [x for x in cols if "name" in x][0]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
print("columns:", cols)
rows = c.execute("select %s, %s, start, end from kernels order by start" % (q, name)).fetchall()
by = collections.defaultdict(list)
for qq, n, s, e in rows:
    by[qq].append((n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], s, e))
t_all0 = min(r[2] for r in rows)
for qq, ks in by.items():
    busy = sum(e - s for _, s, e in ks); span = ks[-1][2] - ks[0][1]
    print("== queue %s: %d kernels, busy %.1f ms of span %.1f ms" % (qq, len(ks), busy / 1e6, span / 1e6))
    last_end = None
    for n, s, e in ks[: 4000]:
        if last_end is not None and (s - last_end) / 1e6 >= min_gap:
            print("   gap %7.2f ms before %-22s at %8.1f ms" % ((s - last_end) / 1e6, n, (s - t_all0) / 1e6))
        if (e - s) / 1e6 >= 5.0:
            print("   %-24s %7.2f ms at %8.1f" % (n, (e - s) / 1e6, (s - t_all0) / 1e6))
        last_end = e if last_end is None else max(last_end, e)
if len(sys.argv) > 4:
    if sys.argv[3].startswith("@"):  # "@kernel:n": the window starts at the n-th launch of that kernel, argv[4] is its length in ms
        kn, nth = sys.argv[3][1:].split(":")
        hits = [(s - t_all0) / 1e6 for _, n, s, e in rows if kn in n]
        lo = hits[int(nth)]; hi = lo + float(sys.argv[4])
    else:
        lo, hi = float(sys.argv[3]), float(sys.argv[4])
    print("== window %.0f..%.0f ms: kernels >= 0.5 ms, and per queue/name folded counts of the shorter ones" % (lo, hi))
    small = collections.Counter()
    for qq, n, s, e in rows:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        ts = (s - t_all0) / 1e6
        if ts < lo or ts > hi: continue
        if (e - s) / 1e6 >= 0.5: print("   q%-3s %-26s %8.2f .. %8.2f  (%6.2f ms)" % (qq, n, ts, (e - t_all0) / 1e6, (e - s) / 1e6))
        else: small[(qq, n)] += 1
    for k, v in sorted(small.items(), key=lambda x: str(x)): print("   short:", k, v)
