"""A synthetic local map in the shape LocalMapping hands to Optimizer::LocalBACameraPointObjectsDynamic: the window of synth.ba_dyn_problem (a KITTI-like drive
with moving cars) turned back into the pointer graph the reference works on -- key frames with time stamps, key points and per-frame detections, static map
points, dynamic map points that live on a car (PosToObj, best object), map objects with one pose per observing key frame (allDynamicPoses) and a planar
velocity -- and decorated with the cases the function branches on: fixed key frames outside the covisibility list, a bad key frame, key frames more than 5 s
old, static points with one observation and gross outliers, dynamic points with fewer than four observations / owned by an object outside the window / seen
once by another key frame (set bad on the way), a car with fewer than four vertices (no velocity), a car whose velocity is still zero (initialised and written),
a car seen once inside the image margin, a detection at the image border."""
import numpy as np

from cube_slam_amd import synth
from oracle import local_ba_dynamic as ld


def build(seed, n_kf=12, n_points=300, n_objects=4, pts_per_obj=24):
    rng = np.random.default_rng(seed + 2000)
    d = synth.ba_dyn_problem(seed, n_kf=n_kf, n_points=n_points, n_objects=n_objects, pts_per_obj=pts_per_obj)
    W, H = 1241, 376
    sig = (np.float32(1.0) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float64)
    ids = [0] + list(np.cumsum(rng.integers(1, 4, n_kf - 1)))
    stamps = np.arange(n_kf) * 0.1
    stamps[:3] -= 6.0                      # the first three key frames are more than 5 s older than the current one
    # key points per key frame: static observations first, then the observations of dynamic points
    keys = [[] for _ in range(n_kf)]

    def add_key(i, uv, ur, w):
        keys[i].append((uv, ur, int(np.argmin(np.abs(sig - w)))))
        return len(keys[i]) - 1
    obs_key = [add_key(int(c), d["obs_uv"][k], d["obs_ur"][k], d["obs_inv_sigma2"][k]) for k, c in enumerate(d["obs_cam"])]
    dobs_key = [add_key(int(c), d["dobs_uv"][k], -1.0, d["dobs_inv_sigma2"][k]) for k, c in enumerate(d["dobs_cam"])]
    kfs = []
    for i in range(n_kf):
        uv = np.array([k[0] for k in keys[i]], float).reshape(-1, 2); ur = np.array([k[1] for k in keys[i]], float); octv = np.array([k[2] for k in keys[i]], int)
        kf = ld.KeyFrame(int(ids[i]), d["cam_pose"][i], uv, ur, octv, sig, stamps[i])
        kf.map_point_matches = [None] * len(uv)
        kfs.append(kf)
    mps = [ld.MapPoint(100 + j, d["points"][j]) for j in range(len(d["points"]))]
    for k, (c, p) in enumerate(zip(d["obs_cam"], d["obs_point"])):
        mps[p].observations[kfs[c]] = obs_key[k]; kfs[c].map_point_matches[obs_key[k]] = mps[p]
    # 5 % of the static points keep one observation; 3 % of the static observations are gross outliers
    for mp in mps:
        if len(mp.observations) > 1 and rng.uniform() < 0.05:
            last = list(mp.observations)[-1]
            for kf, k in list(mp.observations.items()):
                if kf is not last:
                    kf.map_point_matches[k] = None; del mp.observations[kf]
    for kf in kfs:
        n_static = sum(1 for m in kf.map_point_matches if m is not None)
        bad = rng.uniform(size=len(kf.mvKeysUn)) < 0.03
        kf.mvKeysUn[bad] += rng.choice([-25.0, 25.0], (int(bad.sum()), 2))   # (key points of dynamic points too: three-vertex edges with chi2 > 8)
        del n_static
    # cars: one map object per car, one pose per key frame that has an object vertex in the synthetic window
    mos = [ld.MapObject(c, np.zeros(7), np.array([2.0, 0.9, 0.8]), 1.0) for c in range(n_objects)]
    vertex = {}
    for oi, (c, i) in enumerate(d["obj_key"]):
        mo, kf = mos[c], kfs[i]
        mo.allDynamicPoses[kf] = d["obj_pose"][oi].copy()
        vertex[oi] = (mo, kf)
    for k, (c, oi) in enumerate(zip(d["cobs_cam"], d["cobs_obj"])):
        mo, kf = vertex[int(oi)]
        bb = d["cobs_bbox"][k]
        x, y = int(bb[0] - bb[2] / 2), int(bb[1] - bb[3] / 2)
        x, y = max(x, 11), max(y, 11)
        wd, hd = min(int(bb[2]), W - 11 - x - 1), min(int(bb[3]), H - 11 - y - 1)
        kf.local_cuboids.append({"bbox_vec": bb.copy(), "bbox_2d": (x, y, wd, hd), "left_right_to_car": 1 if bb[0] < W / 3 else (2 if bb[0] > 2 * W / 3 else 0)})
        kf.cuboids_landmark.append(mo)
        mo.observations[kf] = len(kf.local_cuboids) - 1
        mo.observed_frames.append(kf)
        mo.meas_quality = float(np.sqrt(d["cobs_info"][k][0]) / 2.0)
        mo.pose = d["obj_pose"][int(oi)].copy()
    for c, mo in enumerate(mos):
        mo.velocityPlanar = d["vel"][c].copy()
    # dynamic points: PosToObj on their car, observations in the key frames where the car has a vertex
    dmps = []
    first_dp = 0
    for c in range(n_objects):
        for k in range(pts_per_obj):
            j = first_dp + k
            dmps.append(ld.MapPoint(3000 + j, np.zeros(3), is_dynamic=True, PosToObj=d["dpoints"][j], best_object=mos[c]))
        first_dp += pts_per_obj
    for k, (c, p) in enumerate(zip(d["dobs_cam"], d["dobs_point"])):
        dmps[p].observations[kfs[c]] = dobs_key[k]; kfs[c].map_point_matches[dobs_key[k]] = dmps[p]
    for mp in dmps:   # a world position of some kind (the function does not read it for a dynamic point)
        mo = mp.best_object
        T = mo.pose
        mp.pos = ld.lo._rot(T[3:]) @ mp.PosToObj + T[:3] if len(mo.observations) else np.zeros(3)
    # the window: the last key frame is current, six others covisible (one of them bad), the rest enter as fixed key frames through the points they see
    cur = kfs[-1]
    cur.covisible = [kfs[i] for i in (n_kf - 2, n_kf - 3, n_kf - 4, n_kf - 5, n_kf - 6, n_kf - 8)]
    kfs[n_kf - 5].bad = True
    # the branches on the dynamic side
    cars = sorted(mos, key=lambda m: -len(m.observations))
    if len(cars) > 1:   # a car with fewer than four vertices: no velocity vertex
        few = cars[-1]
        for kf in list(few.observations)[:-3]:
            del few.observations[kf]; few.observed_frames.remove(kf)
            kf.cuboids_landmark[kf.cuboids_landmark.index(few)] = None
    cars[0].velocityPlanar = np.zeros(2)             # still zero: initialised from the first and last pose and written back (:2223-2233)
    if len(cars) > 2:   # a detection at the image border: no camera-object edge for it
        kf = list(cars[1].observations)[-1]
        det = kf.local_cuboids[cars[1].observations[kf]]
        det["bbox_2d"] = (3, det["bbox_2d"][1], det["bbox_2d"][2], det["bbox_2d"][3])
    outside = ld.MapObject(77, np.array([50.0, 0, 0.8, 0, 0, 0, 1]), np.array([2.0, 0.9, 0.8]), 1.0)   # an object no local key frame holds
    loose = [mp for mp in dmps if len(mp.observations) >= 4][:3]
    if loose:
        loose[0].best_object = outside
        loose[1].best_object = None
    for mp in [m for m in dmps if len(m.observations) >= 5][5:8]:   # fewer than four observations
        for kf in list(mp.observations)[3:]:
            kf.map_point_matches[mp.observations[kf]] = None; del mp.observations[kf]
    for mp in [m for m in dmps if len(m.observations) >= 5][10:12]:  # seen once, by a covisible key frame: set bad while the window is gathered
        keep = cur.covisible[0] if cur.covisible[0] in mp.observations else None
        for kf in list(mp.observations):
            if kf is not keep and (keep is not None or kf is not list(mp.observations)[0]):
                kf.map_point_matches[mp.observations[kf]] = None; del mp.observations[kf]
    for mo in mos:
        mo.unique_points = []
    params = {"K": synth.K_KITTI, "img_width": W, "img_height": H, "bf": d["bf"], "camera_object_BA_weight": 2.0, "object_velocity_BA_weight": 0.5, "kitti": True,
              "build_worldframe_on_ground": True, "ba_dyna_pt_obj_cam": True, "ba_dyna_obj_velo": True, "ba_dyna_obj_cam": True}
    return cur, params, {"kfs": kfs, "mps": mps + dmps, "mos": mos + [outside], "truth": d}


def flatten_window(pKF):
    """The window as the flat arrays cube_slam_amd.ba_dynamic.LocalBACameraPointObjectsDynamic takes -- the pointer walk of Optimizer.cc:1540-1665 that
    adapters/Optimizer_hip.cc does over KeyFrame* / MapPoint* / MapObject* -- plus the objects behind the rows."""
    local_kfs, marked = [pKF], {id(pKF)}
    for kf in pKF.covisible:
        marked.add(id(kf))
        if not kf.bad:
            local_kfs.append(kf)
    points, seen, set_bad = [], set(), []
    for kf in local_kfs:
        for mp in kf.map_point_matches:
            if mp is None or mp.bad:
                continue
            if kf is not pKF and mp.is_dynamic and mp.Observations() == 1:
                mp.bad = True; set_bad.append(mp)
            if id(mp) not in seen:
                seen.add(id(mp)); points.append(mp)
    objects = []
    for kf in local_kfs:
        for mo in kf.cuboids_landmark:
            if mo is not None and not mo.bad and all(mo is not o for o in objects):
                objects.append(mo)
    fixed = []
    for mp in points:
        for kf in mp.observations:
            if id(kf) not in marked:
                marked.add(id(kf))
                if not kf.bad:
                    fixed.append(kf)
    for mo in objects:
        for kf in mo.observations:
            if (kf.mTimeStamp - pKF.mTimeStamp) > 8.0 and id(kf) not in marked:
                marked.add(id(kf))
                if not kf.bad:
                    fixed.append(kf)
    kfs = local_kfs + fixed
    row = {id(k): i for i, k in enumerate(kfs)}
    mo_row = {id(m): i for i, m in enumerate(objects)}
    w = {"kf_id": np.array([k.mnId for k in kfs]), "kf_pose": np.stack([k.Tcw for k in kfs]), "kf_stamp": np.array([k.mTimeStamp for k in kfs]),
         "kf_cam_center": np.stack([k.camera_center() for k in kfs]), "n_local": len(local_kfs),
         "mp_pos": np.array([m.pos for m in points]).reshape(-1, 3), "mp_nobs": np.array([m.Observations() for m in points]), "mp_dynamic": np.array([bool(m.is_dynamic) for m in points]),
         "mp_pos_to_obj": np.array([m.PosToObj if m.PosToObj is not None else np.zeros(3) for m in points]).reshape(-1, 3),
         "mp_best_mo": np.array([mo_row.get(id(m.best_object), -1) if m.best_object is not None else -1 for m in points])}
    om, ok, ouv, our, ow = [], [], [], [], []
    for j, mp in enumerate(points):
        for kf, idx in mp.observations.items():
            if not kf.bad:
                om.append(j); ok.append(row[id(kf)]); ouv.append(kf.mvKeysUn[idx]); our.append(float(kf.mvuRight[idx]) if kf.mvuRight[idx] >= 0 else -1.0)
                ow.append(float(kf.mvInvLevelSigma2[kf.octave[idx]]))
    w.update(obs_mp=np.array(om, int), obs_kf=np.array(ok, int), obs_uv=np.array(ouv, float).reshape(-1, 2), obs_ur=np.array(our, float), obs_inv_sigma2=np.array(ow, float))
    vm, vk, vp, vb, vr, vl, sm, sk, um, up, uc, ov_key = [], [], [], [], [], [], [], [], [], [], [], []
    for i, mo in enumerate(objects):
        for kf, idx in mo.observations.items():
            if kf.bad or id(kf) not in row:
                continue
            det = kf.local_cuboids[idx]
            vm.append(i); vk.append(row[id(kf)]); vp.append(np.asarray(mo.allDynamicPoses[kf], float)); vb.append(det["bbox_vec"]); vr.append(det["bbox_2d"]); vl.append(det["left_right_to_car"])
            ov_key.append((mo, kf))
        for kf in mo.observed_frames:
            if not kf.bad and id(kf) in row:
                sm.append(i); sk.append(row[id(kf)])
        for mp in mo.unique_points:
            if mp is not None and not mp.bad:
                um.append(i); up.append(mp.pos); uc.append(mp.MapObjObservations.get(mo, 0))
    w.update(mo_id=np.array([m.mnId for m in objects]), mo_meas_quality=np.array([m.meas_quality for m in objects], float),
             mo_largest_point_observations=np.array([m.largest_point_observations for m in objects], int), mo_velocity=np.array([m.velocityPlanar for m in objects], float).reshape(-1, 2),
             ov_mo=np.array(vm, int), ov_kf=np.array(vk, int), ov_pose=np.array(vp, float).reshape(-1, 7), ov_bbox_vec=np.array(vb, float).reshape(-1, 4),
             ov_bbox_2d=np.array(vr, int).reshape(-1, 4), ov_left_right_to_car=np.array(vl, int), seq_mo=np.array(sm, int), seq_kf=np.array(sk, int),
             up_mo=np.array(um, int), up_pos=np.array(up, float).reshape(-1, 3), up_count=np.array(uc, int))
    return w, {"kfs": kfs, "points": points, "objects": objects, "ov_key": ov_key, "set_bad": set_bad}
