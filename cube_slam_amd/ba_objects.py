"""Python host-side mirror of Optimizer::LocalBACameraPointObjects (reference orb_object_slam/include/Optimizer.h:48,
src/Optimizer.cc:826-1534): the graph-level flow LocalMapping runs on every new key frame, over the C-ABI bundle adjuster (cs_ba_*).

The caller hands over the local window as arrays (what adapters/Optimizer_hip.cc gathers from KeyFrame* / MapPoint* / MapObject*, in the
reference's iteration order; the key-frame selection of :829-913 walks pointer containers and stays on the caller's side):

  kf_id, kf_pose (n,7 world-to-camera [t q]), n_local (the first n_local are lLocalKeyFrames, the rest lFixedCameras), cur_cam_center (3)
  mp_id, mp_pos (m,3), mp_nobs (MapPoint::Observations(): nObs, a stereo observation counts twice, MapPoint.cc:78-81)
  obs_mp, obs_kf, obs_uv (o,2), obs_ur (o; < 0 = monocular), obs_inv_sigma2 (o)                 observations by key frames that are not bad
  mo_id, mo_pose (c,7 object-to-world), mo_scale (c,3), mo_meas_quality (c), mo_largest_point_observations (c)
  up_mo, up_pos (u,3), up_count (u)              GetUniqueMapPoints() of each object (not bad) with MapObjObservations[object]
  det_mo, det_kf, det_bbox_vec (d,4 cx cy w h), det_bbox_2d (d,4 x y w h), det_left_right_to_car (d)  the objects' observations

and params: K (3x3), img_width, img_height, bf, camera_object_BA_weight, kitti (scene_unique_id == kitti), build_worldframe_on_ground.

Steps (Optimizer.cc lines): object vertices with the KITTI half size and the height reset :983-1026; points with one observation skipped
:1052; reprojection edges :1068-1137; point-object association -- count threshold max(int(0.4 largest), 2), 4 m / 3 m outlier filter,
centroid reset above 5 points, unary edge above 10 :1141-1266; camera-object edges -- information (w [/2 above 5 objects])^2 q^2, 10 px margin,
level 1 for an object seen once, left / right balancing :1268-1382; optimize(5), outliers to level 1 (chi2 5.991 / 7.815, depth, |bbox
error| > 80), point kernels off, optimize(10) :1389-1438; erase list and write-back :1440-1533.

A level-1 edge is simply absent from the arrays a stage hands to the solver, and so is a vertex no active edge touches
(SparseOptimizer::initializeOptimization builds the active set from the level-0 edges); a free camera left without edges is held fixed.
The outlier tests read the residuals at the accepted estimates (pin D4 of DESIGN.md)."""
import math

import numpy as np

from .ba import BundleAdjuster

KITTI_OBJECT_HALF_SIZE = (1.9420, 0.8143, 0.7631)  # Optimizer.cc:994


def _rot_rows(p7):
    x, y, z, w = p7[..., 3], p7[..., 4], p7[..., 5], p7[..., 6]
    return np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1), p7[..., 2]  # third row of R, t_z


def build_graph(w, params, fixCamera=False):
    """The g2o graph of :983-1382 as cs_ba_problem arrays plus the level of every camera-object edge and the bookkeeping to map the
    result back: dict(problem, cobs_level, point_rows (index into mp_*), obs_rows (index into obs_*), det_rows)."""
    kitti = params.get("kitti", True)
    n_kf, n_local = len(w["kf_id"]), int(w["n_local"])
    cam_fixed = np.ones(n_kf, np.uint8)
    cam_fixed[:n_local] = (np.asarray(w["kf_id"][:n_local]) == 0) | bool(fixCamera)
    cub_pose = np.array(w["mo_pose"], float).reshape(-1, 7).copy(); cub_scale = np.array(w["mo_scale"], float).reshape(-1, 3).copy()
    n_obj = len(cub_pose)
    if kitti and n_obj:
        if not params.get("build_worldframe_on_ground", False):
            cub_pose[:, 1] = np.float32(w["cur_cam_center"][1]) + 1.0
        else:
            cub_pose[:, 2] = np.float32(w["cur_cam_center"][2]) - 1.0
        cub_scale[:] = KITTI_OBJECT_HALF_SIZE
    # points: Observations() == 1 is skipped (:1052)
    point_rows = np.nonzero(np.asarray(w["mp_nobs"]) != 1)[0]
    prow = -np.ones(len(w["mp_id"]), int); prow[point_rows] = np.arange(len(point_rows))
    obs_rows = np.nonzero(prow[np.asarray(w["obs_mp"], int)] >= 0)[0]
    # point-object association (:1141-1266)
    up_mo, up_pos, up_count = np.asarray(w["up_mo"], int), np.asarray(w["up_pos"], float).reshape(-1, 3), np.asarray(w["up_count"], int)
    pc_cub, pc_off, pc_pts = [], [0], []
    for i in range(n_obj):
        thr = max(int(int(w["mo_largest_point_observations"][i]) * 0.4), 2)
        P = up_pos[(up_mo == i) & (up_count > thr)]
        if len(P) == 0:
            continue
        mean = np.zeros(3)
        for p in P:
            mean = mean + p
        mean = mean / float(len(P))
        near = np.linalg.norm(mean - P, axis=1) < 4.0
        if not near.any():
            continue
        mean2 = np.zeros(3)
        for p in P[near]:
            mean2 = mean2 + p
        mean2 = mean2 / float(near.sum())
        good = P[np.linalg.norm(mean2 - P, axis=1) < 3.0]
        if len(good) > 5:
            acc = np.zeros(3)
            for p in good:
                acc = acc + p
            cub_pose[i, :3] = acc / float(len(good))
        if len(good) > 10:
            pc_cub.append(i); pc_pts.append(good); pc_off.append(pc_off[-1] + len(good))
    # camera-object edges (:1268-1382)
    inv_sigma = 1.0 * params.get("camera_object_BA_weight", 1.0)
    if n_obj > 5:
        inv_sigma = inv_sigma / 2
    rect = np.asarray(w["det_bbox_2d"], int).reshape(-1, 4)
    m = 10
    in_fov = (rect[:, 0] > m) & (rect[:, 1] > m) & (rect[:, 0] + rect[:, 2] < params["img_width"] - m) & (rect[:, 1] + rect[:, 3] < params["img_height"] - m)
    det_rows = np.nonzero(in_fov)[0]
    det_mo = np.asarray(w["det_mo"], int)[det_rows]
    q = np.asarray(w["mo_meas_quality"], float)[det_mo]
    cobs_info = np.full((len(det_rows), 4), inv_sigma * inv_sigma) * q[:, None] * q[:, None]
    cobs_level = np.zeros(len(det_rows), int)
    for i in range(n_obj):
        mine = np.nonzero(det_mo == i)[0]
        if len(mine) == 1:
            cobs_level[mine[0]] = 1
    if kitti and len(det_rows):
        lr = np.asarray(w["det_left_right_to_car"], int)[det_rows]
        tl, tr, tm = int((lr == 1).sum()), int((lr == 2).sum()), int((lr == 0).sum())
        if tl > 2 * (tr + tm):
            cobs_info[lr == 1] = cobs_info[lr == 1] / 2.0
        if tr > 2 * (tl + tm):
            cobs_info[lr == 2] = cobs_info[lr == 2] / 2.0
    K = np.asarray(params["K"], float)
    d = {"cam_pose": np.array(w["kf_pose"], float).reshape(-1, 7), "cam_fixed": cam_fixed, "points": np.asarray(w["mp_pos"], float).reshape(-1, 3)[point_rows],
         "cuboid_pose": cub_pose, "cuboid_scale": cub_scale, "cuboid_flags": np.full(n_obj, 1 | 8, np.uint8),
         "obs_cam": np.asarray(w["obs_kf"], np.int32)[obs_rows], "obs_point": prow[np.asarray(w["obs_mp"], int)[obs_rows]].astype(np.int32),
         "obs_uv": np.asarray(w["obs_uv"], float).reshape(-1, 2)[obs_rows], "obs_inv_sigma2": np.asarray(w["obs_inv_sigma2"], float)[obs_rows],
         "obs_ur": np.asarray(w["obs_ur"], float)[obs_rows], "fx": K[0, 0], "fy": K[1, 1], "cx": K[0, 2], "cy": K[1, 2],
         "huber_mono": float(np.float32(math.sqrt(5.991))), "huber_stereo": float(np.float32(math.sqrt(7.815))), "bf": params.get("bf", 0.0),
         "cobs_cam": np.asarray(w["det_kf"], np.int32)[det_rows], "cobs_cuboid": det_mo.astype(np.int32),
         "cobs_bbox": np.asarray(w["det_bbox_vec"], float).reshape(-1, 4)[det_rows], "cobs_info": cobs_info, "K": K, "huber_obj": float(np.float32(math.sqrt(900.0))),  # the widths are `const float` in the reference (:1043-1044, :1292): float-rounded roots
         "pc_cuboid": np.array(pc_cub, np.int32), "pc_offsets": np.array(pc_off, np.int32),
         "pc_points": np.concatenate(pc_pts).reshape(-1, 3) if pc_pts else np.zeros((0, 3)), "max_outside_margin_ratio": 2.0 if kitti else 1.0}
    return {"problem": d, "cobs_level": cobs_level, "point_rows": point_rows, "obs_rows": obs_rows, "det_rows": det_rows}


def active_subgraph(d, keep_obs, keep_cobs):
    """Level-0 edges and the vertices they touch, as a problem the solver takes; -> (problem, point index, cuboid index into d)."""
    keep_obs, keep_cobs = np.asarray(keep_obs, bool), np.asarray(keep_cobs, bool)
    oc, op = d["obs_cam"][keep_obs], d["obs_point"][keep_obs]
    cc, cu = d["cobs_cam"][keep_cobs], d["cobs_cuboid"][keep_cobs]
    pts_used = np.unique(op)
    cub_used = np.unique(np.concatenate([cu, d["pc_cuboid"]]).astype(int))
    pmap = -np.ones(len(d["points"]), int); pmap[pts_used] = np.arange(len(pts_used))
    cmap = -np.ones(len(d["cuboid_pose"]), int); cmap[cub_used] = np.arange(len(cub_used))
    used = np.zeros(len(d["cam_pose"]), bool); used[oc] = True; used[cc] = True
    s = dict(d)
    s.update(cam_fixed=(d["cam_fixed"].astype(bool) | ~used).astype(np.uint8), points=d["points"][pts_used], cuboid_pose=d["cuboid_pose"][cub_used],
             cuboid_scale=d["cuboid_scale"][cub_used], cuboid_flags=d["cuboid_flags"][cub_used], obs_cam=oc, obs_point=pmap[op].astype(np.int32),
             obs_uv=d["obs_uv"][keep_obs], obs_inv_sigma2=d["obs_inv_sigma2"][keep_obs], obs_ur=d["obs_ur"][keep_obs],
             cobs_cam=cc, cobs_cuboid=cmap[cu].astype(np.int32), cobs_bbox=d["cobs_bbox"][keep_cobs], cobs_info=d["cobs_info"][keep_cobs],
             pc_cuboid=cmap[d["pc_cuboid"]].astype(np.int32))
    return s, pts_used, cub_used


def _solve(d, keep_obs, keep_cobs, iterations, ctx, stop_flag=None):
    s, pu, cu = active_subgraph(d, keep_obs, keep_cobs)
    out = dict(d)
    if len(s["obs_cam"]) + len(s["cobs_cam"]) + len(s["pc_cuboid"]) == 0:
        return out, None
    ba = BundleAdjuster(s, ctx=ctx)
    st = ba.optimize(iterations, stop_flag)
    cam, pts, cub = ba.read()
    ba.close()
    out["cam_pose"] = cam
    out["points"] = d["points"].copy(); out["points"][pu] = pts
    out["cuboid_pose"] = d["cuboid_pose"].copy()
    if len(cu):
        out["cuboid_pose"][cu] = cub
    return out, st


def _residuals(d, ctx):
    """Reprojection chi2 / depth of every observation and the bbox error norm of every camera-object edge at the estimates in d."""
    s, _, _ = active_subgraph(d, np.ones(len(d["obs_cam"]), bool), np.ones(len(d["cobs_cam"]), bool))  # every edge, without the vertices none touches
    ba = BundleAdjuster(s, ctx=ctx)
    _, eo, ec, _ = ba.errors()
    ba.close()
    st = d["obs_ur"] >= 0
    chi = np.where(st, (eo ** 2).sum(1), (eo[:, :2] ** 2).sum(1)) * d["obs_inv_sigma2"]
    r3, tz = _rot_rows(d["cam_pose"][d["obs_cam"]])
    z = np.einsum("nj,nj->n", r3, d["points"][d["obs_point"]]) + tz  # EdgeSE3ProjectXYZ::isDepthPositive
    return chi, z, np.sqrt((ec ** 2).sum(1)) if len(ec) else np.zeros(0), st


def LocalBACameraPointObjects(window, params, ctx=None, fixCamera=False, stop_flag=None):
    """-> dict(kf_pose (n_local,7), point_pos {row of mp_*: xyz}, point_unwritten [rows of mp_* the caller leaves as they are], object_pose (c,7), object_scale (c,3), erase [(kf row, mp row)],
    obs_level, cobs_level (after stage 1), stats).  The caller writes the poses back (SetPose / SetWorldPos / UpdateNormalAndDepth) and
    erases the listed observations (:1477-1533)."""
    g = build_graph(window, params, fixCamera)
    d, cobs_level = g["problem"], g["cobs_level"]
    n_obs = len(d["obs_cam"])
    est, st1 = _solve(d, np.ones(n_obs, bool), cobs_level == 0, 5, ctx, stop_flag)                      # :1389-1390
    chi, z, cnorm, stereo = _residuals(est, ctx)
    obs_level = ((chi > np.where(stereo, 7.815, 5.991)) | ~(z > 0)).astype(int)                        # :1399-1428
    cobs_level2 = cobs_level.copy()
    cobs_level2[(cobs_level == 0) & (cnorm > 80)] = 1                                                  # :1430-1437 (an inactive edge holds a zero error)
    est2 = dict(est); est2["huber_mono"] = 0.0; est2["huber_stereo"] = 0.0                             # setRobustKernel(0) on the point edges
    fin, st2 = _solve(est2, obs_level == 0, cobs_level2 == 0, 10, ctx, stop_flag)                       # :1439-1440
    chi2, z2, _, _ = _residuals(fin, ctx)
    chi_used = np.where(obs_level == 0, chi2, chi)  # a level-1 edge keeps the error of stage 1
    bad = (chi_used > np.where(stereo, 7.815, 5.991)) | ~(z2 > 0)                                       # :1445-1475
    order = [k for k in range(n_obs) if not stereo[k]] + [k for k in range(n_obs) if stereo[k]]         # vpEdgesMono, then vpEdgesStereo
    rows = g["obs_rows"]
    erase = [(int(window["obs_kf"][rows[k]]), int(window["obs_mp"][rows[k]])) for k in order if bad[k]]
    erase_stereo = [bool(stereo[k]) for k in order if bad[k]]
    # the reference's write-back re-reads MapPoint::Observations() after the erasures (:1486-1496 before :1509-1516): a point they leave with exactly one
    # observation is NOT written back -- `point_unwritten` lists those rows of mp_*
    left = np.asarray(window["mp_nobs"], int).copy()
    for (_, r), st_ in zip(erase, erase_stereo):
        left[r] -= 2 if st_ else 1  # mp_nobs is MapPoint::Observations() = nObs: a stereo observation counts twice (MapPoint.cc:78-81, 186-189)
    unwritten = [int(r) for r in g["point_rows"] if left[r] == 1]
    return {"kf_pose": fin["cam_pose"][:int(window["n_local"])], "point_pos": {int(r): fin["points"][j] for j, r in enumerate(g["point_rows"])},
            "object_pose": fin["cuboid_pose"], "object_scale": d["cuboid_scale"], "erase": erase, "erase_stereo": erase_stereo, "point_unwritten": unwritten, "obs_level": obs_level, "cobs_level": cobs_level,
            "cobs_level2": cobs_level2, "stats": (st1, st2), "graph": g}
