#!/usr/bin/env python3
"""Distinct 128-byte lines touched per wave gather of the edge-scoring kernel under different lane orders and distance-map layouts
(numpy simulation on the benchmark scenes via the CPU oracle debug rows).  Result: row-major f32 20.2 lines per
gather, 4x8-pixel tiles 9.0 -- and the measured kernel time did not move, i.e. the texture-address path is bound per lane, not per line."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from cube_slam_amd import synth
import pyoracle as po
VIS1 = [(0,1),(1,2),(2,3),(3,0),(1,5),(2,4),(3,7),(4,7),(4,5)]
VIS2 = [(0,1),(1,2),(2,3),(3,0),(1,5),(2,4),(4,5)]
def samples(rows, cfg):
    vis = VIS1 if cfg == 1 else VIS2
    X = rows[:, 9:17]; Y = rows[:, 17:25]
    pts = []
    for k in range(8 if cfg == 1 else 6):
        pts.append((X[:, k].astype(int), Y[:, k].astype(int)))
    for a, b in vis:
        for s in range(1, 10):
            px = s / 10.0 * X[:, a] + (1 - s / 10.0) * X[:, b]; py = s / 10.0 * Y[:, a] + (1 - s / 10.0) * Y[:, b]
            pts.append((px.astype(int), py.astype(int)))
    return pts
def lines_of(px, py, layout, x0, y0, w):
    x = np.clip(px - x0, 0, w - 1); y = np.maximum(py - y0, 0)
    if layout == 'row_f32': return (y * w + x) * 4 // 128
    if layout == 'row_u16': return (y * w + x) * 2 // 128
    if layout == 'tile_f32_8x4': return (y // 4) * 4096 + x // 8
    if layout == 'tile_u16_8x8': return (y // 8) * 4096 + x // 8
    if layout == 'tile_u16_16x4': return (y // 4) * 4096 + x // 16
    if layout == 'tile_f32_4x8': return (y // 8) * 4096 + x // 4
tot = {}
for seed in range(1000, 1006):
    s = synth.cuboid_scene(seed)
    o = po.cuboid_opts(yaw_step_deg=0.5)
    res, dbg = po.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=o, debug=True)
    rows = dbg["rows"]; rc = dbg["row_count"]
    off = 0
    for bi in range(len(s["boxes"])):
        n = int(rc[bi * 3:(bi + 1) * 3].sum()) if len(rc) >= (bi + 1) * 3 else 0
        R = rows[off:off + n]; off += n
        bx = s["boxes"][bi]; x0, y0, w = int(bx[0]), int(bx[1]), int(bx[2])
        for cfg in (1, 2):
            Rc = R[R[:, 0] == cfg]
            if len(Rc) == 0: continue
            for order in ('ref', 'top_major'):
                if order == 'top_major':
                    idx = np.lexsort((Rc[:, 2], Rc[:, 3]))  # top id major, yaw fastest
                    Ro = Rc[idx]
                else: Ro = Rc
                pts = samples(Ro, cfg)
                for layout in ('row_f32', 'tile_f32_8x4', 'tile_f32_4x8', 'tile_u16_8x8', 'tile_u16_16x4'):
                    cnt = 0; ng = 0
                    for px, py in pts:
                        L = lines_of(px, py, layout, x0, y0, w + 40)
                        for wv in range(0, len(L), 64):
                            cnt += len(np.unique(L[wv:wv + 64])); ng += 1
                    k = (order, layout); a = tot.get(k, [0, 0]); a[0] += cnt; a[1] += ng; tot[k] = a
for k, (c, n) in sorted(tot.items()): print(k, "lines/gather %.1f" % (c / n), n)
