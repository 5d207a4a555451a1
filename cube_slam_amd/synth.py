"""Seeded synthetic workloads.  Plumbing for tests and bench.py.

No network / datasets: every input of the hot path is generated here, deterministically from a seed.
"""
import math

import numpy as np

SEED = 20260923

K_TUM = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1]], np.float64)  # object_slam/src/main_obj.cpp:347-349
K_KITTI = np.array([[721.5377, 0, 609.5593], [0, 721.5377, 172.854], [0, 0, 1]], np.float64)


def camera_pose(height=1.1, pitch_deg=25.0, yaw_deg=0.0, roll_deg=0.0, xy=(0.0, 0.0)):
    """T_wc (4x4): camera `height` above the ground plane z=0, looking along world +y rotated by yaw, pitched down."""
    th = math.radians(pitch_deg)
    # camera axes in world: x right, y down, z forward
    R0 = np.array([[1, 0, 0],
                   [0, -math.sin(th), math.cos(th)],
                   [0, -math.cos(th), -math.sin(th)]], np.float64)
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], np.float64)
    cr, sr = math.cos(math.radians(roll_deg)), math.sin(math.radians(roll_deg))
    Rroll = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]], np.float64)  # about the optical axis
    T = np.eye(4)
    T[:3, :3] = Rz @ R0 @ Rroll
    T[:3, 3] = [xy[0], xy[1], height]
    return T


_BODY = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], np.float64)
_FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
_EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def _project(K, Twc, pts_w):
    Tcw = np.linalg.inv(Twc)
    pc = Tcw[:3, :3] @ pts_w + Tcw[:3, 3:4]
    uv = K @ pc
    return uv[:2] / uv[2], pc[2]


def cuboid_scene(seed, W=640, H=480, n_boxes=3, K=K_TUM, n_clutter=40, noise_sigma=2.0, bg_texture=0.0):
    """One synthetic frame: gray u8 image with `n_boxes` drawn cuboids standing on the ground,
    their tight 2-D boxes [x y w h prob], the line segments (cuboid edges + clutter) and T_wc.
    bg_texture > 0 lays a band-limited 1/f texture (texture_image, scaled by bg_texture) under the cuboids, so that the data-dependent
    stages see real content: ~1000 ORB key points and a few hundred LSD segments per frame at 0.5 instead of ~150 / ~25 on the flat floor."""
    from PIL import Image, ImageDraw

    rng = np.random.default_rng(seed)
    Twc = camera_pose(height=1.1 + 0.1 * rng.uniform(-1, 1), pitch_deg=25 + 3 * rng.uniform(-1, 1),
                      yaw_deg=rng.uniform(-30, 30))
    img = Image.new("L", (W, H), 128)
    drw = ImageDraw.Draw(img)
    # mild background texture (floor gradient) so that the ROI is not perfectly flat
    bg = np.tile(np.linspace(118, 138, H)[:, None], (1, W))
    img = Image.fromarray(bg.astype(np.uint8), "L")
    drw = ImageDraw.Draw(img)
    boxes, lines = [], []
    cam_xy = Twc[:2, 3]
    fwd = Twc[:3, 2].copy(); fwd[2] = 0; fwd /= np.linalg.norm(fwd)
    right = np.array([fwd[1], -fwd[0], 0.0])
    placed = []
    tries = 0
    while len(placed) < n_boxes and tries < 400:
        tries += 1
        slot = len(placed)
        lateral = (slot - (n_boxes - 1) / 2.0) * (1.7 / max(1, n_boxes - 1)) + rng.uniform(-0.15, 0.15)
        depth = rng.uniform(2.3, 3.1)
        centre = np.array([cam_xy[0], cam_xy[1], 0.0]) + fwd * depth + right * lateral
        half = np.array([rng.uniform(0.28, 0.45), rng.uniform(0.28, 0.45), rng.uniform(0.4, 0.62)])
        yaw = rng.uniform(-math.pi, math.pi)
        c, s = math.cos(yaw), math.sin(yaw)
        Rm = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        pts = Rm @ (_BODY * half[:, None]) + np.array([centre[0], centre[1], half[2]])[:, None]
        uv, z = _project(K, Twc, pts)
        if z.min() < 0.5:
            continue
        x0, y0, x1, y1 = uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()
        if x0 < 12 or y0 < 12 or x1 > W - 13 or y1 > H - 13:
            continue
        if (x1 - x0) < 110 or (y1 - y0) < 110:
            continue
        placed.append((pts, uv, z))
    # painter's algorithm over all faces of all cuboids
    faces = []
    for ci, (pts, uv, z) in enumerate(placed):
        for fi, f in enumerate(_FACES):
            depth = float(np.mean(np.linalg.norm(pts[:, list(f)] - Twc[:3, 3:4], axis=0)))
            shade = int(40 + 35 * fi + 11 * ci) % 200 + 30
            faces.append((depth, [(float(uv[0, k]), float(uv[1, k])) for k in f], shade))
    faces.sort(key=lambda t: -t[0])
    mask = Image.new("L", (W, H), 0)
    mdrw = ImageDraw.Draw(mask)
    for _, poly, shade in faces:
        drw.polygon(poly, fill=shade)
        mdrw.polygon(poly, fill=255)
    for pts, uv, z in placed:
        x0, y0, x1, y1 = uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()
        bx, by = math.floor(x0), math.floor(y0)
        boxes.append([bx, by, math.ceil(x1) - bx, math.ceil(y1) - by, 0.9])
        for a, b in _EDGES:
            p = np.array([uv[0, a], uv[1, a], uv[0, b], uv[1, b]]) + rng.normal(0, 0.4, 4)
            lines.append(p)
    for _ in range(n_clutter):
        ln = rng.uniform(30, 200)
        a = rng.uniform(0, math.pi)
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        p = np.array([cx - ln / 2 * math.cos(a), cy - ln / 2 * math.sin(a), cx + ln / 2 * math.cos(a), cy + ln / 2 * math.sin(a)])
        p[[0, 2]] = np.clip(p[[0, 2]], 0, W - 1)
        p[[1, 3]] = np.clip(p[[1, 3]], 0, H - 1)
        lines.append(p)
    g = np.asarray(img, np.float64) + rng.normal(0, noise_sigma, (H, W))
    if bg_texture > 0:
        g = g + (np.asarray(mask) == 0) * bg_texture * (texture_image(seed, W, H).astype(np.float64) - 128.0)
    gray = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    return {
        "gray": gray, "K": K.copy(), "Twc": Twc,
        "boxes": np.array(boxes, np.float64).reshape(-1, 5),
        "lines": np.array(lines, np.float64).reshape(-1, 4),
    }


def _texture_base(seed, W, H):
    rng = np.random.default_rng(seed)
    big_w = W + 1024
    fy = np.fft.fftfreq(H)[:, None]
    fx = np.fft.rfftfreq(big_w)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    spec = (rng.normal(size=f.shape) + 1j * rng.normal(size=f.shape)) / f ** 1.1
    spec[0, 0] = 0
    im = np.fft.irfft2(spec, s=(H, big_w))
    im = (im - im.mean()) / im.std()
    im = np.clip(128 + 48 * im, 0, 255)
    return np.rint(im).astype(np.uint8)


def texture_image(seed, W, H, shift=0):
    """Band-limited (1/f) noise texture, translated by `shift` px -- FAST fires everywhere."""
    s = int(shift) % 1024
    return np.ascontiguousarray(_texture_base(seed, W, H)[:, s:s + W])


def texture_stream(seed, W, H, n, step=3):
    """texture_image(seed, W, H, shift=step * i) for i < n as one array (the base texture is made once)."""
    base = _texture_base(seed, W, H)
    return np.stack([base[:, (step * i) % 1024:(step * i) % 1024 + W] for i in range(n)])


# ----------------------------------------------------------------------------------------------- object BA
KITTI_OBJ_HALF = np.array([1.9420, 0.8143, 0.7631])  # orb_object_slam/src/Optimizer.cc:994


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s; q[3] = (R[k, j] - R[j, k]) / s; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _pose7(R, t):
    return np.concatenate([t, _quat_from_R(R)])


def ba_problem(seed, n_kf=1000, n_points=100000, n_cuboids=500, k_obs=5, W=1241, H=376, noise_px=1.0, stereo_frac=0.0):
    """Chain trajectory in a KITTI-like camera world (x right, y down, z forward; ground at y = +1.65):
    keyframes 0.5 m apart on a gentle curve, points in a corridor each seen by ~k_obs consecutive keyframes, cuboids
    (cars) along the road seen by ~10 keyframes through 2-D boxes and owning ~40 fixed surface points.
    stereo_frac > 0 turns that share of the observations into stereo ones (u_right = u - bf/z, close points only, as
    Frame::ComputeStereoMatches yields them); the other arrays do not depend on it.
    Returns a dict of numpy arrays (the SoA layout of cs_ba_problem)."""
    rng = np.random.default_rng(seed)
    K = K_KITTI
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    step, cam_h = 0.5, 1.65
    yaw = 0.15 * np.sin(np.arange(n_kf) * step / 40.0)  # heading about the y axis
    cz = np.cumsum(np.cos(yaw)) * step
    cxw = np.cumsum(np.sin(yaw)) * step
    centers = np.stack([cxw, np.zeros(n_kf), cz], axis=1)
    Rwc = [_rot(1, a) for a in yaw]
    cam_true = np.stack([_pose7(R.T, -R.T @ c) for R, c in zip(Rwc, centers)])

    def project(i, Xw):
        Xc = (Rwc[i].T @ (Xw - centers[i]).T).T
        z = Xc[:, 2]
        return np.stack([fx * Xc[:, 0] / z + cx, fy * Xc[:, 1] / z + cy], axis=1), z

    # points: uniformly in the corridor around the path
    s = rng.uniform(3, n_kf * step + 25, n_points)
    ki = np.clip((s / step).astype(int), 0, n_kf - 1)
    lat = rng.uniform(-15, 15, n_points)
    pts = np.stack([cxw[ki] + lat, rng.uniform(cam_h - 3.0, cam_h, n_points), s], axis=1)
    # observations: for each point the first k_obs consecutive keyframes (from ~30 m before it) that see it
    Rwc_all = np.stack(Rwc)
    first = np.maximum(0, ((pts[:, 2] - 30) / step).astype(int))
    got = np.zeros(n_points, int)
    oc, op, ou, ow, oz = [], [], [], [], []
    for j in range(80):
        ki_j = first + j
        ok = (ki_j < n_kf) & (got < k_obs)
        kk = np.minimum(ki_j, n_kf - 1)
        Xc = np.einsum("nji,nj->ni", Rwc_all[kk], pts - centers[kk])
        z = Xc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * Xc[:, 0] / z + cx
            v = fy * Xc[:, 1] / z + cy
        ok &= (z > 2.0) & (z < 40.0) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
        idx = np.nonzero(ok)[0]
        if len(idx) == 0:
            continue
        octave = rng.integers(0, 8, len(idx))
        oc.append(kk[idx]); op.append(idx); oz.append(z[idx])
        ou.append(np.stack([u[idx], v[idx]], axis=1) + rng.normal(0, noise_px, (len(idx), 2)))
        ow.append((np.float32(1.0) / (np.float32(1.2) ** octave.astype(np.float32)) ** 2).astype(np.float64))
        got[idx] += 1
    obs_cam = np.concatenate(oc) if oc else np.zeros(0, int)
    obs_pt = np.concatenate(op) if op else np.zeros(0, int)
    obs_uv = np.concatenate(ou) if ou else np.zeros((0, 2))
    obs_w = np.concatenate(ow) if ow else np.zeros(0)
    order = np.lexsort((obs_cam, obs_pt))  # landmark-major like the reference's loop over map points
    obs_cam, obs_pt, obs_uv, obs_w = obs_cam[order], obs_pt[order], obs_uv[order], obs_w[order]
    stereo = {}
    if stereo_frac > 0:
        rs = np.random.default_rng(seed + 7919)  # own stream: the monocular arrays stay what they are without stereo
        bf = 386.1448
        obs_z = (np.concatenate(oz) if oz else np.zeros(0))[order]
        is_st = (rs.uniform(size=len(obs_z)) < stereo_frac) & (obs_z < 35.0)
        ur = obs_uv[:, 0] - bf / np.maximum(obs_z, 1e-3) + rs.normal(0, noise_px, len(obs_z))
        stereo = {"obs_ur": np.where(is_st & (ur >= 0), ur, -1.0), "bf": bf, "huber_stereo": math.sqrt(7.815)}
    # cuboids: object z axis = world up (-y): R_align maps object (x fwd, y left, z up) into the camera world
    R_align = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float)
    cub_pose, cobs_cam, cobs_cub, cobs_bbox, cobs_info, pc_cub, pc_off, pc_pts = [], [], [], [], [], [], [0], []
    body = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], float)
    for c in range(n_cuboids):
        sc = rng.uniform(min(12.0, n_kf * step), max(13.0, n_kf * step + 5))
        k0 = int(np.clip(sc / step, 0, n_kf - 1))
        side = rng.choice([-1.0, 1.0]) * rng.uniform(3.0, 6.0)
        centre = np.array([cxw[k0] + side, cam_h - KITTI_OBJ_HALF[2], sc])
        Ro = R_align @ _rot(2, rng.uniform(-0.3, 0.3) + (math.pi if rng.uniform() < 0.5 else 0.0))
        cub_pose.append(_pose7(Ro, centre))
        corners = (Ro @ (body * KITTI_OBJ_HALF[:, None])).T + centre
        n_seen = 0
        for i in range(max(0, k0 - 70), k0):
            uv, z = project(i, corners)
            if z.min() < 4 or z.max() > 40:
                continue
            x0, y0, x1, y1 = uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()
            if x0 > 10 and y0 > 10 and x1 < W - 10 and y1 < H - 10:
                bb = np.array([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0]) + rng.normal(0, 3.0, 4)
                q = rng.uniform(0.5, 1.0)
                cobs_cam.append(i); cobs_cub.append(c); cobs_bbox.append(bb); cobs_info.append(np.full(4, (0.5 * q) ** 2))
                n_seen += 1
                if n_seen >= 10:
                    break
        local = rng.uniform(-1, 1, (40, 3)) * KITTI_OBJ_HALF
        face = rng.integers(0, 3, 40)
        local[np.arange(40), face] = np.sign(local[np.arange(40), face]) * KITTI_OBJ_HALF[face]
        pc_cub.append(c); pc_pts.append((Ro @ local.T).T + centre + rng.normal(0, 0.05, (40, 3))); pc_off.append(pc_off[-1] + 40)
    # perturbed initial estimates
    cam_init = cam_true.copy()
    for i in range(1, n_kf):
        dR = _rot(0, rng.normal(0, math.radians(0.5))) @ _rot(1, rng.normal(0, math.radians(0.5))) @ _rot(2, rng.normal(0, math.radians(0.5)))
        c_i = centers[i] + rng.normal(0, 0.05, 3)
        Rn = Rwc[i] @ dR
        cam_init[i] = _pose7(Rn.T, -Rn.T @ c_i)
    cub_init = np.array(cub_pose).reshape(-1, 7).copy()
    for c in range(len(cub_init)):
        cub_init[c, :3] += rng.normal(0, 0.2, 3) * np.array([1, 0, 1])
    cam_fixed = np.zeros(n_kf, np.uint8); cam_fixed[0] = 1
    n_c = len(cub_init)
    return {
        "cam_pose": cam_init, "cam_fixed": cam_fixed, "points": pts + rng.normal(0, 0.05, pts.shape),
        "cuboid_pose": cub_init, "cuboid_scale": np.tile(KITTI_OBJ_HALF, (n_c, 1)), "cuboid_flags": np.full(n_c, 1 | 8, np.uint8),
        "obs_cam": np.array(obs_cam, np.int32), "obs_point": np.array(obs_pt, np.int32), "obs_uv": np.array(obs_uv, np.float64).reshape(-1, 2),
        "obs_inv_sigma2": np.array(obs_w, np.float64), "fx": fx, "fy": fy, "cx": cx, "cy": cy, "huber_mono": math.sqrt(5.991),
        "cobs_cam": np.array(cobs_cam, np.int32), "cobs_cuboid": np.array(cobs_cub, np.int32), "cobs_bbox": np.array(cobs_bbox, np.float64).reshape(-1, 4),
        "cobs_info": np.array(cobs_info, np.float64).reshape(-1, 4), "K": K.copy(), "huber_obj": math.sqrt(900.0),
        "pc_cuboid": np.array(pc_cub, np.int32), "pc_offsets": np.array(pc_off, np.int32),
        "pc_points": np.concatenate(pc_pts).reshape(-1, 3) if pc_pts else np.zeros((0, 3)), "max_outside_margin_ratio": 2.0,
        "cam_true": cam_true, "points_true": pts, "cuboid_true": np.array(cub_pose).reshape(-1, 7), **stereo,
    }


def pose_frame(seed, n=800, outlier_frac=0.15, stereo_frac=0.0, noise_px=1.0, W=1241, H=376):
    """One tracking frame for Optimizer::PoseOptimization: n map points in front of a KITTI-like camera, pixel noise by octave,
    a fraction of gross outliers (wrong matches), the initial pose perturbed like a constant-velocity prediction."""
    rng = np.random.default_rng(seed)
    K = K_KITTI
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    bf = 386.1448
    R = _rot(1, rng.uniform(-0.2, 0.2)) @ _rot(0, rng.uniform(-0.05, 0.05))
    t = rng.uniform(-1, 1, 3)
    u = rng.uniform(20, W - 20, n); v = rng.uniform(20, H - 20, n); z = rng.uniform(4, 60, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=1)
    Xw = (R.T @ (Xc - t).T).T  # Tcw = [R t]
    octave = rng.integers(0, 8, n)
    sig = 1.2 ** octave
    obs = np.stack([u + rng.normal(0, noise_px, n) * sig, v + rng.normal(0, noise_px, n) * sig, np.full(n, -1.0)], axis=1)
    st = rng.random(n) < stereo_frac
    obs[st, 2] = obs[st, 0] - bf / z[st] + rng.normal(0, noise_px, st.sum()) * sig[st]
    bad = rng.random(n) < outlier_frac
    obs[bad, 0] += rng.uniform(15, 80, bad.sum()) * rng.choice([-1, 1], bad.sum())
    obs[bad, 1] += rng.uniform(15, 60, bad.sum()) * rng.choice([-1, 1], bad.sum())
    Xw = Xw.astype(np.float32).astype(np.float64); obs = obs.astype(np.float32).astype(np.float64)  # the reference reads float Mats
    dR = _rot(1, rng.normal(0, 0.01)) @ _rot(0, rng.normal(0, 0.005)) @ _rot(2, rng.normal(0, 0.005))
    pose0 = _pose7(dR @ R, dR @ t + rng.normal(0, 0.05, 3))
    return {"Xw": Xw, "obs": obs, "inv_sigma2": (1.0 / (sig * sig)).astype(np.float32).astype(np.float64), "intr": np.array([fx, fy, cx, cy, bf]),
            "pose": pose0, "pose_true": _pose7(R, t), "is_outlier": bad}


def ba_dyn_problem(seed, n_kf=12, n_points=400, n_objects=3, pts_per_obj=30, fix_points=False, fix_cams=False, W=1241, H=376, noise_px=1.0, dt=0.1,
                   stereo_frac=0.3):
    """A dynamic-object local BA window in the layout of cs_ba_dyn_problem (Optimizer::LocalBACameraPointObjectsDynamic,
    orb_object_slam/src/Optimizer.cc:1537-2573).  Ground-based world (z up, build_worldframe_on_ground): the camera drives along +x at 1.65 m,
    cars move on the plane with a planar velocity (EdgeObjectMotion's bicycle model), every car has one VertexCuboidFixScale per key frame
    that sees it (whether_fixrotation, KITTI fixed scale), dynamic points live in the car frame, static points in the world."""
    rng = np.random.default_rng(seed)
    K = K_KITTI
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    bf = 386.1448
    Rwc0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float)  # camera axes (x right, y down, z forward) in the world (x forward, y left, z up)
    cam_speed = 8.0
    centers = np.stack([np.arange(n_kf) * cam_speed * dt, 0.05 * np.sin(np.arange(n_kf) * 0.7), np.full(n_kf, 1.65)], axis=1)
    Rwc = [Rwc0 @ _rot(1, 0.01 * math.sin(0.5 * i)) for i in range(n_kf)]
    cam_true = np.stack([_pose7(R.T, -R.T @ c) for R, c in zip(Rwc, centers)])

    def project(i, Xw):
        Xc = (Rwc[i].T @ (np.atleast_2d(Xw) - centers[i]).T).T
        z = Xc[:, 2]
        return np.stack([fx * Xc[:, 0] / z + cx, fy * Xc[:, 1] / z + cy], axis=1), z

    def inv_sigma2(n):
        octave = rng.integers(0, 8, n)
        return (np.float32(1.0) / (np.float32(1.2) ** octave.astype(np.float32)) ** 2).astype(np.float64)

    # static points and their (mono / stereo) observations, landmark-major
    pts = np.stack([rng.uniform(5, 45, n_points) + centers[-1, 0] * rng.uniform(0, 1, n_points), rng.uniform(-12, 12, n_points), rng.uniform(0, 4, n_points)], axis=1)
    oc, op, ou, our, ow = [], [], [], [], []
    for j in range(n_points):
        seen = 0
        for i in range(n_kf):
            uv, z = project(i, pts[j])
            if z[0] < 2 or z[0] > 45 or not (0 <= uv[0, 0] < W and 0 <= uv[0, 1] < H) or rng.uniform() < 0.3:
                continue
            m = uv[0] + rng.normal(0, noise_px, 2)
            ur = m[0] - bf / z[0] + rng.normal(0, noise_px)
            st = rng.uniform() < stereo_frac and z[0] < 35 and ur >= 0
            oc.append(i); op.append(j); ou.append(m); our.append(ur if st else -1.0); seen += 1
        ow.append(inv_sigma2(seen))
    # cars: pose per key frame from the motion model the edge uses, object vertices only where the box is inside the image
    body = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], float)
    obj_pose, obj_key, vel_true = [], [], []
    cobs_cam, cobs_obj, cobs_bbox, cobs_info = [], [], [], []
    mot_from, mot_to, mot_vel, mot_dt = [], [], [], []
    dpts, dobs_cam, dobs_obj, dobs_pt, dobs_uv, dobs_w = [], [], [], [], [], []
    pc_obj, pc_off, pc_pts = [], [0], []
    veh_len = 2.71
    for c in range(n_objects):
        v, steer = rng.uniform(5, 11), rng.uniform(-0.05, 0.05)
        yaw = rng.uniform(-0.15, 0.15)
        pos = np.array([rng.uniform(12, 22), rng.choice([-1.0, 1.0]) * rng.uniform(2.5, 5.0), KITTI_OBJ_HALF[2]])
        quality = rng.uniform(0.6, 1.0)
        vel_true.append([v, steer])
        local = rng.uniform(-1, 1, (pts_per_obj, 3)) * KITTI_OBJ_HALF * 1.15  # some a little outside the box: UnaryLocalPoint is non-zero there
        first_dp = len(dpts)
        dpts.extend(local)
        dp_seen = np.zeros(pts_per_obj, int)
        prev = None
        for i in range(n_kf):
            Ro = _rot(2, yaw)
            corners = (Ro @ (body * KITTI_OBJ_HALF[:, None])).T + pos
            uv, z = project(i, corners)
            x0, y0, x1, y1 = uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()
            if z.min() > 3 and x0 > 10 and y0 > 10 and x1 < W - 10 and y1 < H - 10:
                oi = len(obj_pose)
                obj_pose.append(_pose7(Ro, pos.copy())); obj_key.append((c, i))
                cobs_cam.append(i); cobs_obj.append(oi)
                cobs_bbox.append(np.array([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0]) + rng.normal(0, 3.0, 4))
                cobs_info.append(np.full(4, (2.0 * quality) ** 2))  # camera_object_BA_weight^2 * meas_quality^2
                if prev is not None:
                    mot_from.append(prev[0]); mot_to.append(oi); mot_vel.append(c); mot_dt.append((i - prev[1]) * dt)
                prev = (oi, i)
                wpts = (Ro @ local.T).T + pos
                puv, pz = project(i, wpts)
                for k in range(pts_per_obj):
                    if pz[k] > 2 and 0 <= puv[k, 0] < W and 0 <= puv[k, 1] < H and rng.uniform() < 0.8:
                        dobs_cam.append(i); dobs_obj.append(oi); dobs_pt.append(first_dp + k); dobs_uv.append(puv[k] + rng.normal(0, noise_px, 2)); dp_seen[k] += 1
            # bicycle model of EdgeObjectMotion applied to the back-wheel centre (g2o_Object.cpp:255-263)
            back = pos[:2] - veh_len * 0.5 * np.array([math.cos(yaw), math.sin(yaw)]) + v * dt * np.array([math.cos(yaw), math.sin(yaw)])
            yaw = yaw + math.tan(steer) * dt / veh_len * v
            pos = np.array([back[0] + veh_len * 0.5 * math.cos(yaw), back[1] + veh_len * 0.5 * math.sin(yaw), pos[2]])
        if prev is not None:  # EdgePointCuboidOnlyObjectFixScale on the car's last object vertex: world points on its surface
            Tl = np.array(obj_pose[prev[0]])
            n_pc_pts = 15
            surf = rng.uniform(-1, 1, (n_pc_pts, 3)) * KITTI_OBJ_HALF
            face = rng.integers(0, 3, n_pc_pts)
            surf[np.arange(n_pc_pts), face] = np.sign(surf[np.arange(n_pc_pts), face]) * KITTI_OBJ_HALF[face]
            Rl = _rot(2, 2 * math.atan2(Tl[5], Tl[6]))
            pc_obj.append(prev[0]); pc_pts.append((Rl @ surf.T).T + Tl[:3] + rng.normal(0, 0.1, (n_pc_pts, 3))); pc_off.append(pc_off[-1] + n_pc_pts)
    dobs_w = inv_sigma2(len(dobs_cam))
    # perturbed initial estimates
    cam_init = cam_true.copy()
    for i in range(2, n_kf):
        dR = _rot(0, rng.normal(0, math.radians(0.3))) @ _rot(1, rng.normal(0, math.radians(0.3))) @ _rot(2, rng.normal(0, math.radians(0.3)))
        c_i = centers[i] + rng.normal(0, 0.04, 3)
        Rn = Rwc[i] @ dR
        cam_init[i] = _pose7(Rn.T, -Rn.T @ c_i)
    cam_fixed = np.zeros(n_kf, np.uint8); cam_fixed[:2] = 1
    if fix_cams:
        cam_fixed[:] = 1
        cam_init = cam_true.copy()
    obj_true = np.array(obj_pose).reshape(-1, 7)
    obj_init = obj_true.copy()
    obj_init[:, :3] += rng.normal(0, 0.25, (len(obj_init), 3)) * np.array([1, 1, 0.2])
    n_o = len(obj_init)
    vel_true = np.array(vel_true).reshape(-1, 2)
    return {
        "cam_pose": cam_init, "cam_fixed": cam_fixed,
        "obj_pose": obj_init, "obj_scale": np.tile(KITTI_OBJ_HALF, (n_o, 1)), "obj_flags": np.full(n_o, 2 | 8, np.uint8), "obj_key": np.array(obj_key, np.int32).reshape(-1, 2),
        "vel": vel_true * rng.uniform(0.7, 1.2, vel_true.shape) * np.array([1.0, 0.0]),  # (linear_esti, 0), Optimizer.cc:2231
        "points": pts + rng.normal(0, 0.08, pts.shape), "dpoints": np.array(dpts).reshape(-1, 3) + rng.normal(0, 0.08, (len(dpts), 3)), "fix_points": int(fix_points),
        "obs_cam": np.array(oc, np.int32), "obs_point": np.array(op, np.int32), "obs_uv": np.array(ou, np.float64).reshape(-1, 2), "obs_ur": np.array(our, np.float64),
        "obs_inv_sigma2": np.concatenate(ow) if ow else np.zeros(0), "obs_level": np.zeros(len(oc), np.uint8),
        "fx": fx, "fy": fy, "cx": cx, "cy": cy, "bf": bf, "huber_mono": math.sqrt(5.991), "huber_stereo": math.sqrt(7.815),
        "ulp_info": 10.0, "ulp_scale": KITTI_OBJ_HALF.copy(), "ulp_ratio": 2.0,
        "dobs_cam": np.array(dobs_cam, np.int32), "dobs_obj": np.array(dobs_obj, np.int32), "dobs_point": np.array(dobs_pt, np.int32),
        "dobs_uv": np.array(dobs_uv, np.float64).reshape(-1, 2), "dobs_inv_sigma2": np.asarray(dobs_w, np.float64), "dobs_level": np.zeros(len(dobs_cam), np.uint8),
        "K": K.copy(), "huber_dyn": math.sqrt(5.991),
        "mot_from": np.array(mot_from, np.int32), "mot_to": np.array(mot_to, np.int32), "mot_vel": np.array(mot_vel, np.int32), "mot_dt": np.array(mot_dt, np.float64),
        "mot_info": np.array([1.0, 1.0, 5.0]) ** 2 * 0.5 ** 2,  # (inv_sigma .* inv_sigma), inv_sigma = (1, 1, 5) * object_velocity_BA_weight
        "cobs_cam": np.array(cobs_cam, np.int32), "cobs_obj": np.array(cobs_obj, np.int32), "cobs_bbox": np.array(cobs_bbox, np.float64).reshape(-1, 4),
        "cobs_info": np.array(cobs_info, np.float64).reshape(-1, 4), "cobs_level": np.zeros(len(cobs_cam), np.uint8), "huber_obj": math.sqrt(900.0),
        "pc_obj": np.array(pc_obj, np.int32), "pc_offsets": np.array(pc_off, np.int32), "pc_points": np.concatenate(pc_pts).reshape(-1, 3) if pc_pts else np.zeros((0, 3)),
        "pc_ratio": 2.0,
        "cam_true": cam_true, "obj_true": obj_true, "vel_true": vel_true, "points_true": pts, "dpoints_true": np.array(dpts).reshape(-1, 3),
    }


def ba_dyn_strip(d, objects=True, dynamic=True, static=True):
    """The window without its cars / dynamic points / static points (ragged and empty inputs)."""
    d = dict(d)
    if not static:
        d["points"] = np.zeros((0, 3)); d["obs_cam"] = np.zeros(0, np.int32); d["obs_point"] = np.zeros(0, np.int32); d["obs_uv"] = np.zeros((0, 2))
        d["obs_ur"] = np.zeros(0); d["obs_inv_sigma2"] = np.zeros(0); d["obs_level"] = np.zeros(0, np.uint8)
    if not dynamic or not objects:
        d["dpoints"] = np.zeros((0, 3)); d["dobs_cam"] = np.zeros(0, np.int32); d["dobs_obj"] = np.zeros(0, np.int32); d["dobs_point"] = np.zeros(0, np.int32)
        d["dobs_uv"] = np.zeros((0, 2)); d["dobs_inv_sigma2"] = np.zeros(0); d["dobs_level"] = np.zeros(0, np.uint8)
    if not objects:
        d["obj_pose"] = np.zeros((0, 7)); d["obj_scale"] = np.zeros((0, 3)); d["obj_flags"] = np.zeros(0, np.uint8); d["vel"] = np.zeros((0, 2))
        for k in ("mot_from", "mot_to", "mot_vel", "cobs_cam", "cobs_obj", "pc_obj"):
            d[k] = np.zeros(0, np.int32)
        d["mot_dt"] = np.zeros(0); d["cobs_bbox"] = np.zeros((0, 4)); d["cobs_info"] = np.zeros((0, 4)); d["cobs_level"] = np.zeros(0, np.uint8)
        d["pc_offsets"] = np.zeros(1, np.int32); d["pc_points"] = np.zeros((0, 3))
    return d
