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


def cuboid_scene(seed, W=640, H=480, n_boxes=3, K=K_TUM, n_clutter=40, noise_sigma=2.0):
    """One synthetic frame: gray u8 image with `n_boxes` drawn cuboids standing on the ground,
    their tight 2-D boxes [x y w h prob], the line segments (cuboid edges + clutter) and T_wc."""
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
    for _, poly, shade in faces:
        drw.polygon(poly, fill=shade)
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
    gray = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    return {
        "gray": gray, "K": K.copy(), "Twc": Twc,
        "boxes": np.array(boxes, np.float64).reshape(-1, 5),
        "lines": np.array(lines, np.float64).reshape(-1, 4),
    }


def texture_image(seed, W, H, shift=0):
    """Band-limited (1/f) noise texture, translated by `shift` px -- FAST fires everywhere."""
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
    s = int(shift) % 1024
    return np.ascontiguousarray(np.rint(im[:, s:s + W]).astype(np.uint8))
