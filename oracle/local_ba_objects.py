"""TEST INFRASTRUCTURE (oracle): Optimizer::LocalBACameraPointObjects restated over a small pointer graph of KeyFrame / MapPoint / MapObject
objects (reference orb_object_slam/src/Optimizer.cc:826-1534), with the CPU restatement of the g2o machinery (oracle/ba_oracle.cpp through
pyoracle.ba_optimize / ba_errors) as the solver.  Only tests/ may import this module; the product counterpart is
cube_slam_amd/ba_objects.py, which works on flattened arrays through the C-ABI.

PINNED: tests/test_ref_graph_pins.py runs the reference's own function text on the reference's own g2o (oracle/_ref/libref_graph.so) over the same windows and
holds this restatement to what it leaves in the map (erase list, written and unwritten points, poses, objects, counters).

The graph-level steps and the lines they follow:
  gather_local_window          :829-913   local key frames (current + covisible), their points and objects, the fixed key frames that see them
  object vertices              :983-1026  KITTI fixed half size (1.9420, 0.8143, 0.7631), height reset from the current camera, roll/pitch fixed
  point vertices and edges     :1046-1138 points with one observation skipped; mono / stereo reprojection edges, Huber sqrt(5.991) / sqrt(7.815)
  point-object association     :1141-1266 adaptive count threshold, 4 m / 3 m outlier filter, centroid reset (> 5 points), unary edge (> 10 points)
  camera-object edges          :1268-1382 information (w [/2 if > 5 objects])^2 q^2, 10 px field-of-view margin, single-observation objects at
                                          level 1, left / right balancing
  two stages                   :1389-1438 optimize(5); reprojection outliers (chi2, depth) and |bbox error| > 80 to level 1, point kernels
                                          off; optimize(10)
  outcome                      :1440-1533 observations to erase, poses / points / objects written back

Pinned interpretation (D4): the outlier tests read the residuals at the accepted estimate.  g2o's e->chi2() reads the edge's stored
error, which differs only when the last LM trial of a stage was rejected (the stored error is then the rejected trial's)."""
import math

import numpy as np

from . import pyoracle as po

KITTI_HALF = (1.9420, 0.8143, 0.7631)


class KeyFrame:
    def __init__(self, mnId, Tcw, keys_un, u_right, octave, inv_level_sigma2, bad=False):
        self.mnId, self.Tcw, self.bad = mnId, np.asarray(Tcw, float), bad
        self.mvKeysUn, self.mvuRight, self.octave, self.mvInvLevelSigma2 = keys_un, u_right, octave, inv_level_sigma2
        self.local_cuboids = []      # per-frame detections: dict(bbox_vec, bbox_2d, left_right_to_car)
        self.cuboids_landmark = []   # MapObject or None per detection
        self.map_point_matches = []  # MapPoint or None per key point
        self.covisible = []          # GetVectorCovisibleKeyFrames()

    def camera_center(self):  # GetCameraCenter(): -R^T t; `Ow` (3 floats), when a test sets it, is the value KeyFrame::SetPose stored (float arithmetic on the float pose)
        if getattr(self, "Ow", None) is not None:
            return np.asarray(self.Ow, float)
        t, q = self.Tcw[:3], self.Tcw[3:]
        return -_rot(q).T @ t


class MapPoint:
    def __init__(self, mnId, pos, bad=False):
        self.mnId, self.pos, self.bad = mnId, np.asarray(pos, float), bad
        self.observations = {}        # KeyFrame -> key point index (insertion order stands in for the std::map order)
        self.MapObjObservations = {}  # MapObject -> count

    def Observations(self):  # nObs: MapPoint::AddObservation counts a stereo observation twice (MapPoint.cc:78-81), EraseObservation takes it back the same way (:186-189)
        return sum(2 if kf.mvuRight[idx] >= 0 else 1 for kf, idx in self.observations.items())


class MapObject:
    def __init__(self, mnId, pose, scale, meas_quality, bad=False):
        self.mnId, self.pose, self.scale, self.meas_quality, self.bad = mnId, np.asarray(pose, float), np.asarray(scale, float), meas_quality, bad
        self.observations = {}   # KeyFrame -> index into kf.local_cuboids
        self.unique_points = []  # GetUniqueMapPoints()
        self.largest_point_observations = 0


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gather_local_window(pKF):
    """:829-913 -> (local key frames, local map points, local map objects, fixed key frames)."""
    local_kfs = [pKF]
    marked_local = {id(pKF)}
    for kf in pKF.covisible:
        marked_local.add(id(kf))
        if not kf.bad:
            local_kfs.append(kf)
    points, seen = [], set()
    for kf in local_kfs:
        for mp in kf.map_point_matches:
            if mp is not None and not mp.bad and id(mp) not in seen:
                seen.add(id(mp)); points.append(mp)
    objects, seen_o = [], set()
    for kf in local_kfs:
        for mo in kf.cuboids_landmark:
            if mo is not None and not mo.bad and id(mo) not in seen_o:
                seen_o.add(id(mo)); objects.append(mo)
    fixed, marked_fixed = [], set()
    for group in (points, objects):
        for it in group:
            for kf in it.observations:
                if id(kf) not in marked_local and id(kf) not in marked_fixed:
                    marked_fixed.add(id(kf))
                    if not kf.bad:
                        fixed.append(kf)
    return local_kfs, points, objects, fixed


def _compact(d, keep_obs, keep_cobs, keep_pc):
    """The active part of the graph (SparseOptimizer::initializeOptimization(level 0)): edges at level 0 and the vertices they touch.
    Returns the compact problem and the index maps (points, cuboids) back into d; a free camera without an active edge is held fixed."""
    oc, op = np.asarray(d["obs_cam"])[keep_obs], np.asarray(d["obs_point"])[keep_obs]
    cc, cu = np.asarray(d["cobs_cam"])[keep_cobs], np.asarray(d["cobs_cuboid"])[keep_cobs]
    pcs = [k for k in range(len(d["pc_cuboid"])) if keep_pc[k]]
    pts_used = np.unique(op)
    cub_used = np.unique(np.concatenate([cu, np.asarray(d["pc_cuboid"], int)[pcs]]).astype(int)) if (len(cu) or pcs) else np.zeros(0, int)
    pmap = -np.ones(len(d["points"]), int); pmap[pts_used] = np.arange(len(pts_used))
    cmap = -np.ones(len(d["cuboid_pose"]), int); cmap[cub_used] = np.arange(len(cub_used))
    cam_used = np.zeros(len(d["cam_pose"]), bool); cam_used[oc] = True; cam_used[cc] = True
    out = dict(d)
    out["cam_fixed"] = (np.asarray(d["cam_fixed"], bool) | ~cam_used).astype(np.uint8)
    out["points"] = np.asarray(d["points"])[pts_used]
    out["cuboid_pose"] = np.asarray(d["cuboid_pose"])[cub_used]; out["cuboid_scale"] = np.asarray(d["cuboid_scale"])[cub_used]
    out["cuboid_flags"] = np.asarray(d["cuboid_flags"])[cub_used]
    out["obs_cam"] = oc.astype(np.int32); out["obs_point"] = pmap[op].astype(np.int32)
    for k in ("obs_uv", "obs_inv_sigma2", "obs_ur"):
        out[k] = np.asarray(d[k])[keep_obs]
    out["cobs_cam"] = cc.astype(np.int32); out["cobs_cuboid"] = cmap[cu].astype(np.int32)
    out["cobs_bbox"] = np.asarray(d["cobs_bbox"])[keep_cobs]; out["cobs_info"] = np.asarray(d["cobs_info"])[keep_cobs]
    off, pp, pcc = [0], [], []
    for k in pcs:
        a, b = d["pc_offsets"][k], d["pc_offsets"][k + 1]
        pp.append(np.asarray(d["pc_points"])[a:b]); off.append(off[-1] + b - a); pcc.append(cmap[d["pc_cuboid"][k]])
    out["pc_cuboid"] = np.array(pcc, np.int32); out["pc_offsets"] = np.array(off, np.int32)
    out["pc_points"] = np.concatenate(pp).reshape(-1, 3) if pp else np.zeros((0, 3))
    return out, pts_used, cub_used


def local_ba_camera_point_objects(pKF, params, fixCamera=False):
    """-> dict(kf_pose {mnId: Tcw}, point_pos {mnId: xyz}, object_pose {mnId: pose7}, erase [(kf mnId, mp mnId)], stats, problem, levels)."""
    kitti = params.get("kitti", True)
    local_kfs, points, objects, fixed_kfs = gather_local_window(pKF)
    kfs = local_kfs + fixed_kfs
    cam_index = {id(k): i for i, k in enumerate(kfs)}
    cam_pose = np.stack([k.Tcw for k in kfs])
    cam_fixed = np.array([(k.mnId == 0 or fixCamera) if i < len(local_kfs) else True for i, k in enumerate(kfs)], np.uint8)
    # object vertices :983-1026
    cub_pose, cub_scale = [], []
    for mo in objects:
        pose = mo.pose.copy(); scale = mo.scale.copy()
        if kitti:
            if not params.get("build_worldframe_on_ground", False):
                pose[1] = np.float32(pKF.camera_center()[1]) + 1.0  # float cam_height (cv::Mat of floats)
            else:
                pose[2] = np.float32(pKF.camera_center()[2]) - 1.0
            scale = np.array(KITTI_HALF)
        cub_pose.append(pose); cub_scale.append(scale)
    obj_index = {id(mo): i for i, mo in enumerate(objects)}
    # point vertices and reprojection edges :1046-1138
    pts, pt_mp, obs_cam, obs_pt, obs_uv, obs_w, obs_ur, obs_kf, obs_mp = [], [], [], [], [], [], [], [], []
    for mp in points:
        if mp.Observations() == 1:
            continue
        j = len(pts); pts.append(mp.pos.copy()); pt_mp.append(mp)
        for kf, idx in mp.observations.items():
            if kf.bad:
                continue
            obs_cam.append(cam_index[id(kf)]); obs_pt.append(j); obs_uv.append(kf.mvKeysUn[idx]); obs_w.append(float(kf.mvInvLevelSigma2[kf.octave[idx]]))
            obs_ur.append(float(kf.mvuRight[idx]) if kf.mvuRight[idx] >= 0 else -1.0); obs_kf.append(kf); obs_mp.append(mp)
    # point-object association :1141-1266
    pc_cub, pc_off, pc_pts = [], [0], []
    for i, mo in enumerate(objects):
        thr = max(int(mo.largest_point_observations * 0.4), 2)
        cand = [mp.pos for mp in mo.unique_points if mp is not None and not mp.bad and mp.MapObjObservations.get(mo, 0) > thr]
        good = []
        if cand:
            P = np.stack(cand)
            mean = np.zeros(3)
            for p in P:
                mean = mean + p
            mean = mean / float(len(P))
            mean2, n2 = np.zeros(3), 0
            for p in P:
                if np.linalg.norm(mean - p) < 4.0:
                    mean2 = mean2 + p; n2 += 1
            mean2 = mean2 / float(n2) if n2 else np.full(3, np.nan)
            mean_final = np.zeros(3)
            for p in P:
                if np.linalg.norm(mean2 - p) < 3.0:
                    mean_final = mean_final + p; good.append(p)
            if len(good) > 5:
                cub_pose[i][:3] = mean_final / float(len(good))
        if len(good) > 10:
            pc_cub.append(i); pc_pts.append(np.stack(good)); pc_off.append(pc_off[-1] + len(good))
    # camera-object edges :1268-1382
    inv_sigma = 1.0 * params.get("camera_object_BA_weight", 1.0)
    if len(objects) > 5:
        inv_sigma = inv_sigma / 2
    margin, Wimg, Himg = 10, params["img_width"], params["img_height"]
    cobs_cam, cobs_cub, cobs_bbox, cobs_info, cobs_level, cobs_lr = [], [], [], [], [], []
    for i, mo in enumerate(objects):
        mine = []
        for kf, idx in mo.observations.items():
            if kf.bad:
                continue
            det = kf.local_cuboids[idx]
            x, y, w, h = det["bbox_2d"]
            if x > margin and y > margin and x + w < Wimg - margin and y + h < Himg - margin:
                mine.append(len(cobs_cam))
                cobs_cam.append(cam_index[id(kf)]); cobs_cub.append(i); cobs_bbox.append(np.asarray(det["bbox_vec"], float))
                cobs_info.append(np.full(4, inv_sigma * inv_sigma) * mo.meas_quality * mo.meas_quality); cobs_level.append(0)
                cobs_lr.append(det["left_right_to_car"] if kitti else -1)
        if len(mine) == 1:
            cobs_level[mine[0]] = 1
    if kitti:
        lr = np.array(cobs_lr, int)
        tl, tr, tm = int((lr == 1).sum()), int((lr == 2).sum()), int((lr == 0).sum())
        if tl > 2 * (tr + tm):
            for k in np.nonzero(lr == 1)[0]:
                cobs_info[k] = cobs_info[k] / 2.0
        if tr > 2 * (tl + tm):
            for k in np.nonzero(lr == 2)[0]:
                cobs_info[k] = cobs_info[k] / 2.0
    K = np.asarray(params["K"], float)
    d = {"cam_pose": cam_pose, "cam_fixed": cam_fixed, "points": np.array(pts, float).reshape(-1, 3),
         "cuboid_pose": np.array(cub_pose, float).reshape(-1, 7), "cuboid_scale": np.array(cub_scale, float).reshape(-1, 3),
         "cuboid_flags": np.full(len(objects), 1 | 8, np.uint8),
         "obs_cam": np.array(obs_cam, np.int32), "obs_point": np.array(obs_pt, np.int32), "obs_uv": np.array(obs_uv, float).reshape(-1, 2),
         "obs_inv_sigma2": np.array(obs_w, float), "obs_ur": np.array(obs_ur, float), "fx": K[0, 0], "fy": K[1, 1], "cx": K[0, 2], "cy": K[1, 2],
         "huber_mono": float(np.float32(math.sqrt(5.991))), "huber_stereo": float(np.float32(math.sqrt(7.815))), "bf": params.get("bf", 0.0),
         "cobs_cam": np.array(cobs_cam, np.int32), "cobs_cuboid": np.array(cobs_cub, np.int32), "cobs_bbox": np.array(cobs_bbox, float).reshape(-1, 4),
         "cobs_info": np.array(cobs_info, float).reshape(-1, 4), "K": K, "huber_obj": float(np.float32(math.sqrt(900.0))),  # the widths are `const float` in the reference (:1043-1044, :1292): float-rounded roots
         "pc_cuboid": np.array(pc_cub, np.int32), "pc_offsets": np.array(pc_off, np.int32),
         "pc_points": np.concatenate(pc_pts).reshape(-1, 3) if pc_pts else np.zeros((0, 3)), "max_outside_margin_ratio": 2.0 if kitti else 1.0}
    cobs_level = np.array(cobs_level, int)
    # stage 1 :1389-1390
    n_obs, n_pc = len(obs_cam), len(pc_cub)
    s1, pu1, cu1 = _compact(d, np.ones(n_obs, bool), cobs_level == 0, np.ones(n_pc, bool))
    cam1, p1, c1, st1 = po.ba_optimize(s1, 5)
    est = dict(d); est["cam_pose"] = cam1
    est["points"] = d["points"].copy(); est["points"][pu1] = p1
    est["cuboid_pose"] = d["cuboid_pose"].copy()
    if len(cu1):
        est["cuboid_pose"][cu1] = c1
    # outlier classification :1399-1437 from the residuals at these estimates (pin D4)
    _, eo, ec, _ = po.ba_errors(est)
    st = est["obs_ur"] >= 0
    chi = np.where(st, (eo ** 2).sum(1), (eo[:, :2] ** 2).sum(1)) * est["obs_inv_sigma2"]
    z = np.array([(_rot(est["cam_pose"][c][3:]) @ est["points"][p] + est["cam_pose"][c][:3])[2] for c, p in zip(obs_cam, obs_pt)]) if n_obs else np.zeros(0)
    obs_level = ((chi > np.where(st, 7.815, 5.991)) | ~(z > 0)).astype(int)
    active_c = cobs_level == 0
    cobs_level2 = cobs_level.copy()
    if len(ec):
        cobs_level2[active_c & (np.sqrt((ec ** 2).sum(1)) > 80)] = 1   # inactive edges hold a zero error in g2o: they stay where they are
    # stage 2 :1439-1440: no kernel on the point edges, the camera-object edges keep theirs
    est2 = dict(est); est2["huber_mono"] = 0.0; est2["huber_stereo"] = 0.0
    s2, pu2, cu2 = _compact(est2, obs_level == 0, cobs_level2 == 0, np.ones(n_pc, bool))
    cam2, p2, c2, st2 = po.ba_optimize(s2, 10)
    fin = dict(est2); fin["cam_pose"] = cam2
    fin["points"] = est["points"].copy(); fin["points"][pu2] = p2
    fin["cuboid_pose"] = est["cuboid_pose"].copy()
    if len(cu2):
        fin["cuboid_pose"][cu2] = c2
    # observations to erase :1445-1475: active edges by their final error, level-1 edges by the error they kept from stage 1; depth at the final estimates
    _, eo2, _, _ = po.ba_errors(fin)
    chi2 = np.where(st, (eo2 ** 2).sum(1), (eo2[:, :2] ** 2).sum(1)) * fin["obs_inv_sigma2"]
    chi_used = np.where(obs_level == 0, chi2, chi)
    z2 = np.array([(_rot(fin["cam_pose"][c][3:]) @ fin["points"][p] + fin["cam_pose"][c][:3])[2] for c, p in zip(obs_cam, obs_pt)]) if n_obs else np.zeros(0)
    bad = (chi_used > np.where(st, 7.815, 5.991)) | ~(z2 > 0)
    order = [k for k in range(n_obs) if not st[k]] + [k for k in range(n_obs) if st[k]]  # vpEdgesMono first, then vpEdgesStereo
    erase = [(obs_kf[k].mnId, obs_mp[k].mnId) for k in order if bad[k] and not obs_mp[k].bad]
    # write-back :1509-1516 re-reads Observations() AFTER the erasures (:1486-1496): a point they leave with exactly one observation keeps its old position
    n_erased = {}
    for k in order:
        if bad[k] and not obs_mp[k].bad:
            n_erased[obs_mp[k].mnId] = n_erased.get(obs_mp[k].mnId, 0) + (2 if st[k] else 1)  # MapPoint.cc:186-189
    unwritten = [mp.mnId for mp in pt_mp if mp.Observations() - n_erased.get(mp.mnId, 0) == 1]
    return {"kf_pose": {k.mnId: fin["cam_pose"][i] for i, k in enumerate(local_kfs)},
            "point_pos": {mp.mnId: fin["points"][j] for j, mp in enumerate(pt_mp)},
            "object_pose": {mo.mnId: fin["cuboid_pose"][i] for i, mo in enumerate(objects)}, "object_scale": {mo.mnId: d["cuboid_scale"][i] for i, mo in enumerate(objects)},
            "erase": erase, "point_unwritten": unwritten, "stats": (st1, st2), "problem": d, "obs_level": obs_level, "cobs_level": cobs_level, "cobs_level2": cobs_level2,
            "order": {"kfs": [k.mnId for k in kfs], "n_local": len(local_kfs), "points": [mp.mnId for mp in pt_mp], "objects": [mo.mnId for mo in objects]}}


def flatten_window(pKF):
    """The arrays cube_slam_amd.ba_objects.LocalBACameraPointObjects takes (what an adapter gathers from the reference's map), in the
    reference's iteration order."""
    local_kfs, points, objects, fixed_kfs = gather_local_window(pKF)
    kfs = local_kfs + fixed_kfs
    ki = {id(k): i for i, k in enumerate(kfs)}
    w = {"kf_id": np.array([k.mnId for k in kfs]), "kf_pose": np.stack([k.Tcw for k in kfs]), "n_local": len(local_kfs),
         "cur_cam_center": pKF.camera_center(),
         "mp_id": np.array([m.mnId for m in points]), "mp_pos": np.array([m.pos for m in points]).reshape(-1, 3),
         "mp_nobs": np.array([m.Observations() for m in points])}
    om, ok, ouv, our, ow = [], [], [], [], []
    for j, mp in enumerate(points):
        for kf, idx in mp.observations.items():
            if kf.bad:
                continue
            om.append(j); ok.append(ki[id(kf)]); ouv.append(kf.mvKeysUn[idx]); our.append(float(kf.mvuRight[idx]) if kf.mvuRight[idx] >= 0 else -1.0)
            ow.append(float(kf.mvInvLevelSigma2[kf.octave[idx]]))
    w.update(obs_mp=np.array(om, int), obs_kf=np.array(ok, int), obs_uv=np.array(ouv, float).reshape(-1, 2), obs_ur=np.array(our, float), obs_inv_sigma2=np.array(ow, float))
    w.update(mo_id=np.array([m.mnId for m in objects]), mo_pose=np.array([m.pose for m in objects]).reshape(-1, 7), mo_scale=np.array([m.scale for m in objects]).reshape(-1, 3),
             mo_meas_quality=np.array([m.meas_quality for m in objects], float), mo_largest_point_observations=np.array([m.largest_point_observations for m in objects], int))
    um, up, uc, dm, dk, dv, dr, dl = [], [], [], [], [], [], [], []
    for i, mo in enumerate(objects):
        for mp in mo.unique_points:
            if mp is not None and not mp.bad:
                um.append(i); up.append(mp.pos); uc.append(mp.MapObjObservations.get(mo, 0))
        for kf, idx in mo.observations.items():
            if kf.bad:
                continue
            det = kf.local_cuboids[idx]
            dm.append(i); dk.append(ki[id(kf)]); dv.append(det["bbox_vec"]); dr.append(det["bbox_2d"]); dl.append(det["left_right_to_car"])
    w.update(up_mo=np.array(um, int), up_pos=np.array(up, float).reshape(-1, 3), up_count=np.array(uc, int),
             det_mo=np.array(dm, int), det_kf=np.array(dk, int), det_bbox_vec=np.array(dv, float).reshape(-1, 4), det_bbox_2d=np.array(dr, int).reshape(-1, 4),
             det_left_right_to_car=np.array(dl, int))
    return w
