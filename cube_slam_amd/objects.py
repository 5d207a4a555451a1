"""Host-side mirror of the object association step between detect_cuboid and the object BA (reference
orb_object_slam/src/Tracking.cc: DetectCuboid :1717-1775, AssociateCuboids :1848-1990) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def _ip(a):
    return a.ctypes.data_as(C.c_void_p)


def associate_keypoints(ctx, keypoints, boxes, enable_ground_height_scale=False):
    """keypoints: list (one per keyframe) of (n_i, 2) float arrays (mvKeys[i].pt); boxes: list of (m_i, 4) int arrays (cv::Rect of
    pKF->local_cuboids, in order).  Returns per keyframe (keypoint_associate_objectID, keypoint_inany_object, overlapped)."""
    F = len(keypoints)
    kp_off = np.zeros(F + 1, np.int32); box_off = np.zeros(F + 1, np.int32)
    for f in range(F):
        kp_off[f + 1] = kp_off[f] + len(keypoints[f]); box_off[f + 1] = box_off[f] + len(boxes[f])
    xy = np.ascontiguousarray(np.concatenate([np.asarray(k, np.float32).reshape(-1, 2) for k in keypoints]) if F else np.zeros((0, 2)), np.float32)
    bx = np.ascontiguousarray(np.concatenate([np.asarray(b, np.int32).reshape(-1, 4) for b in boxes]) if F else np.zeros((0, 4)), np.int32)
    assoc = np.full(max(len(xy), 1), -1, np.int32); inany = np.zeros(max(len(xy), 1), np.uint8); ov = np.zeros(max(len(bx), 1), np.uint8)
    check(ctx.ptr, lib().cs_associate_keypoints(ctx.ptr, F, _ip(kp_off), _ip(xy), _ip(box_off), _ip(bx), int(enable_ground_height_scale), _ip(assoc), _ip(inany), _ip(ov)),
          "cs_associate_keypoints")
    return [(assoc[kp_off[f]:kp_off[f + 1]].copy(), inany[kp_off[f]:kp_off[f + 1]].copy(), ov[box_off[f]:box_off[f + 1]].copy()) for f in range(F)]


def associate_cuboids(cand_id, cand_pts, landmark_id, landmark_bad, point_votes, thres, best_object=None, max_vote=None):
    """Tracking::AssociateCuboids on ids (see include/cubeslam_hip.h).  point_votes: list of dicts object id -> count, updated in place.
    Returns (assoc, created)."""
    n = len(cand_id)
    cand_off = np.zeros(n + 1, np.int32)
    for i, p in enumerate(cand_pts):
        cand_off[i + 1] = cand_off[i] + len(p)
    pts = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32) for p in cand_pts]) if n and cand_off[-1] else np.zeros(0), np.int32)
    P = len(point_votes)
    pobs_off = np.zeros(P + 1, np.int32)
    objs, cnts = [], []
    for p, d in enumerate(point_votes):
        pobs_off[p + 1] = pobs_off[p] + len(d)
        objs += list(d.keys()); cnts += list(d.values())
    pobs_obj = np.array(objs, np.int32); pobs_cnt = np.array(cnts, np.int32)
    cid = np.ascontiguousarray(cand_id, np.int32); lid = np.ascontiguousarray(landmark_id, np.int32); lbad = np.ascontiguousarray(landmark_bad, np.uint8)
    assoc = np.zeros(max(n, 1), np.int32); created = np.zeros(max(n, 1), np.uint8)
    cap = int(cand_off[-1]) + 1
    up, uo, uc = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    nu = C.c_int()
    bo = np.ascontiguousarray(best_object, np.int32) if best_object is not None else None
    mv = np.ascontiguousarray(max_vote, np.int32) if max_vote is not None else None
    r = lib().cs_associate_cuboids(n, _ip(cid), _ip(cand_off), _ip(pts), len(lid), _ip(lid), _ip(lbad), P, _ip(pobs_off), _ip(pobs_obj), _ip(pobs_cnt),
                                   _ip(bo) if bo is not None else None, _ip(mv) if mv is not None else None, int(thres), _ip(assoc), _ip(created), cap, _ip(up), _ip(uo), _ip(uc),
                                   C.byref(nu))
    if r != 0:
        raise RuntimeError("cs_associate_cuboids failed: %d" % r)
    for k in range(nu.value):
        point_votes[int(up[k])][int(uo[k])] = int(uc[k])
    if best_object is not None:
        best_object[:] = bo; max_vote[:] = mv
    return assoc[:n].copy(), created[:n].copy()
