"""Python host-side mirror of ORB_SLAM2::ORBmatcher's Hamming searches (reference orb_object_slam/include/ORBmatcher.h:43-89)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .orb import KEYPOINT_DTYPE


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, ctx=None, device=0, max_keypoints=8192, max_queries=16384, max_candidates=4000000):
        self.ctx = ctx or _lib.Context(device)
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri
        self._m = C.c_void_p()
        check(self.ctx.ptr, lib().cs_matcher_create(self.ctx.ptr, max_keypoints, max_queries, C.c_long(max_candidates), C.byref(self._m)), "cs_matcher_create")
        self.N = 0

    def set_frame(self, keysUn, descriptors, bounds):
        """The frame that is searched (CurrentFrame / F / F2): mvKeysUn, mDescriptors, (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        k = np.ascontiguousarray(keysUn, KEYPOINT_DTYPE); d = np.ascontiguousarray(descriptors, np.uint8)
        self.N = len(k)
        check(self.ctx.ptr, lib().cs_matcher_set_frame(self.ctx.ptr, self._m, k.ctypes.data_as(C.c_void_p), _p(d, C.c_uint8), self.N,
                                                       *[C.c_float(b) for b in bounds]), "cs_matcher_set_frame")

    def set_frame_from_orb(self, orb, frame, K4, dist5=None, bounds=None, width=None, height=None, read_keys=True):
        """Frame post-processing on the device: keypoints / descriptors of frame `frame` of the extractor's last run are undistorted
        (Frame::UndistortKeyPoints) and binned (AssignFeaturesToGrid) without leaving HBM.  Returns (mvKeysUn, bounds); read_keys=False: (None, bounds) and the
        frame's key points stay on the device (every search but SearchForInitialization's vbPrevMatched update works from the device copy)."""
        k4 = np.ascontiguousarray(K4, np.float32)
        d5 = None if dist5 is None else np.ascontiguousarray(dist5, np.float32)
        if bounds is None:
            bounds = frame_image_bounds(width, height, k4, d5)
        out = np.zeros(max(orb.cap, 1), KEYPOINT_DTYPE); n = C.c_int()
        check(self.ctx.ptr, lib().cs_matcher_set_frame_from_orb(self.ctx.ptr, self._m, orb._e, int(frame), _p(k4, C.c_float), None if d5 is None else _p(d5, C.c_float),
                                                                *[C.c_float(float(b)) for b in bounds], out.ctypes.data_as(C.c_void_p) if read_keys else None, C.byref(n)),
              "cs_matcher_set_frame_from_orb")
        self.N = n.value
        return (out[:n.value].copy() if read_keys else None), tuple(float(b) for b in bounds)

    def last_candidate_stats(self):
        q, c = C.c_int(), C.c_long()
        if lib().cs_matcher_last_counts(self._m, C.byref(q), C.byref(c)) != 0:
            return None
        return {"queries": q.value, "candidates": c.value}

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        out = np.zeros(max(self.N, 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_matcher_features_in_area(self.ctx.ptr, self._m, C.c_float(x), C.c_float(y), C.c_float(r), minLevel, maxLevel,
                                                              _p(out, C.c_int), len(out), C.byref(n)), "cs_matcher_features_in_area")
        return out[:n.value].copy()

    def SearchByProjectionFrame(self, world_pos, valid, blocks, mp_desc, last_octave, last_angle, Tcw, fx, fy, cx, cy, scale_factors, th, train_blocked=None):
        wp = np.ascontiguousarray(world_pos, np.float32); va = np.ascontiguousarray(valid, np.uint8); bl = np.ascontiguousarray(blocks, np.uint8)
        md = np.ascontiguousarray(mp_desc, np.uint8); lo = np.ascontiguousarray(last_octave, np.int32); la = np.ascontiguousarray(last_angle, np.float32)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy(); sf = np.ascontiguousarray(scale_factors, np.float32)
        tm = np.zeros(max(self.N, 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_by_projection_frame(self.ctx.ptr, self._m, len(va), _p(wp, C.c_float), _p(va, C.c_uint8), _p(bl, C.c_uint8),
                                                               _p(md, C.c_uint8), _p(lo, C.c_int), _p(la, C.c_float), _p(T, C.c_float), C.c_float(fx),
                                                               C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(sf, C.c_float), len(sf), C.c_float(th),
                                                               int(self.mbCheckOrientation), None if train_blocked is None else _p(np.ascontiguousarray(train_blocked, np.uint8), C.c_uint8),
                                                               _p(tm, C.c_int), C.byref(n)), "cs_match_by_projection_frame")
        return tm[:self.N].copy(), n.value

    def SearchByProjectionLocalMap(self, proj_xy, view_cos, pred_level, in_view, blocks, mp_desc, scale_factors, th, train_blocked=None):
        pxy = np.ascontiguousarray(proj_xy, np.float32); vc = np.ascontiguousarray(view_cos, np.float32); pl = np.ascontiguousarray(pred_level, np.int32)
        iv = np.ascontiguousarray(in_view, np.uint8); bl = np.ascontiguousarray(blocks, np.uint8); md = np.ascontiguousarray(mp_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        tb = None if train_blocked is None else np.ascontiguousarray(train_blocked, np.uint8)
        tm = np.zeros(max(self.N, 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_local_map(self.ctx.ptr, self._m, len(iv), _p(pxy, C.c_float), _p(vc, C.c_float), _p(pl, C.c_int), _p(iv, C.c_uint8),
                                                     _p(bl, C.c_uint8), _p(md, C.c_uint8), _p(sf, C.c_float), len(sf), C.c_float(th), C.c_float(self.mfNNratio),
                                                     None if tb is None else _p(tb, C.c_uint8), _p(tm, C.c_int), C.byref(n)), "cs_match_local_map")
        return tm[:self.N].copy(), n.value

    def SearchForInitialization(self, keys1Un, desc1, vbPrevMatched, windowSize=100):
        k1 = np.ascontiguousarray(keys1Un, KEYPOINT_DTYPE); d1 = np.ascontiguousarray(desc1, np.uint8)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.zeros(max(len(k1), 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_for_initialization(self.ctx.ptr, self._m, k1.ctypes.data_as(C.c_void_p), _p(d1, C.c_uint8), len(k1), _p(prev, C.c_float),
                                                              int(windowSize), C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m12, C.c_int),
                                                              C.byref(n)), "cs_match_for_initialization")
        return m12[:len(k1)].copy(), prev, n.value

    def Fuse(self, u_right, inv_level_sigma2, uv, ur, pred_level, valid, mp_desc, scale_factors, th=3.0, keys_static=None):
        """ORBmatcher::Fuse(pKF, vpMapPoints, th), the search (ORBmatcher.cc:921-983) against the key frame given to set_frame: per map
        point (bestIdx, bestDist); the third value is nFused."""
        uvv = np.ascontiguousarray(uv, np.float32); urr = np.ascontiguousarray(ur, np.float32); pl = np.ascontiguousarray(pred_level, np.int32)
        va = np.ascontiguousarray(valid, np.uint8); md = np.ascontiguousarray(mp_desc, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
        kr = np.ascontiguousarray(u_right, np.float32); isg = np.ascontiguousarray(inv_level_sigma2, np.float32)
        ks = None if keys_static is None else np.ascontiguousarray(keys_static, np.uint8)
        bi = np.zeros(max(len(va), 1), np.int32); bd = np.zeros(max(len(va), 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_fuse(self.ctx.ptr, self._m, _p(kr, C.c_float), _p(isg, C.c_float), len(isg), None if ks is None else _p(ks, C.c_uint8), len(va),
                                                _p(uvv, C.c_float), _p(urr, C.c_float), _p(pl, C.c_int), _p(va, C.c_uint8), _p(md, C.c_uint8), _p(sf, C.c_float), C.c_float(th),
                                                _p(bi, C.c_int), _p(bd, C.c_int), C.byref(n)), "cs_match_fuse")
        return bi[:len(va)].copy(), bd[:len(va)].copy(), n.value

    def SearchForTriangulation(self, keys1Un, desc1, node1, skip1, ur1, keys2Un, desc2, node2, skip2, ur2, F12, ex, ey, scale_factors2, level_sigma2_2, bOnlyStereo=False):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.cc:679-850): matches12 and nmatches."""
        k1 = np.ascontiguousarray(keys1Un, KEYPOINT_DTYPE); d1 = np.ascontiguousarray(desc1, np.uint8); k2 = np.ascontiguousarray(keys2Un, KEYPOINT_DTYPE)
        d2 = np.ascontiguousarray(desc2, np.uint8)
        n1 = np.ascontiguousarray(node1, np.int32); s1 = np.ascontiguousarray(skip1, np.uint8); u1 = np.ascontiguousarray(ur1, np.float32)
        n2 = np.ascontiguousarray(node2, np.int32); s2 = np.ascontiguousarray(skip2, np.uint8); u2 = np.ascontiguousarray(ur2, np.float32)
        Fm = np.ascontiguousarray(F12, np.float32).reshape(-1); sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
        m12 = np.zeros(max(len(k1), 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_for_triangulation(self.ctx.ptr, k1.ctypes.data_as(C.c_void_p), _p(d1, C.c_uint8), len(k1), _p(n1, C.c_int), _p(s1, C.c_uint8),
                                                             _p(u1, C.c_float), k2.ctypes.data_as(C.c_void_p), _p(d2, C.c_uint8), len(k2), _p(n2, C.c_int), _p(s2, C.c_uint8),
                                                             _p(u2, C.c_float), _p(Fm, C.c_float), C.c_float(ex), C.c_float(ey), _p(sf, C.c_float), _p(sg, C.c_float), len(sf),
                                                             int(bOnlyStereo), int(self.mbCheckOrientation), _p(m12, C.c_int), C.byref(n)), "cs_match_for_triangulation")
        return m12[:len(k1)].copy(), n.value

    def SearchByBoW(self, keysKF, descKF, nodeKF, skipKF, keysF, descF, nodeF, skipF=None):
        """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:171-310): per frame feature the key-frame feature whose map
        point it receives (-1 none), and nmatches."""
        kk = np.ascontiguousarray(keysKF, KEYPOINT_DTYPE); dk = np.ascontiguousarray(descKF, np.uint8); kf = np.ascontiguousarray(keysF, KEYPOINT_DTYPE)
        df = np.ascontiguousarray(descF, np.uint8)
        nk = np.ascontiguousarray(nodeKF, np.int32); sk = np.ascontiguousarray(skipKF, np.uint8); nf = np.ascontiguousarray(nodeF, np.int32)
        sf = None if skipF is None else np.ascontiguousarray(skipF, np.uint8)
        mf = np.zeros(max(len(kf), 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_by_bow(self.ctx.ptr, kk.ctypes.data_as(C.c_void_p), _p(dk, C.c_uint8), len(kk), _p(nk, C.c_int), _p(sk, C.c_uint8),
                                                  kf.ctypes.data_as(C.c_void_p), _p(df, C.c_uint8), len(kf), _p(nf, C.c_int), None if sf is None else _p(sf, C.c_uint8),
                                                  C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(mf, C.c_int), C.byref(n)), "cs_match_by_bow")
        return mf[:len(kf)].copy(), n.value

    def SearchByBoWKeyFrames(self, keys1, desc1, node1, skip1, keys2, desc2, node2, skip2):
        """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (ORBmatcher.cc:544-677): per KF1 feature the KF2 feature whose map point it is matched
        to (-1 none), and nmatches.  skip = the feature has no usable map point."""
        k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE); d1 = np.ascontiguousarray(desc1, np.uint8); k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
        d2 = np.ascontiguousarray(desc2, np.uint8)
        n1 = np.ascontiguousarray(node1, np.int32); s1 = np.ascontiguousarray(skip1, np.uint8); n2 = np.ascontiguousarray(node2, np.int32)
        s2 = np.ascontiguousarray(skip2, np.uint8)
        m12 = np.zeros(max(len(k1), 1), np.int32); n = C.c_int()
        check(self.ctx.ptr, lib().cs_match_by_bow_kf(self.ctx.ptr, k1.ctypes.data_as(C.c_void_p), _p(d1, C.c_uint8), len(k1), _p(n1, C.c_int), _p(s1, C.c_uint8),
                                                     k2.ctypes.data_as(C.c_void_p), _p(d2, C.c_uint8), len(k2), _p(n2, C.c_int), _p(s2, C.c_uint8),
                                                     C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m12, C.c_int), C.byref(n)), "cs_match_by_bow_kf")
        return m12[:len(k1)].copy(), n.value

    def close(self):
        if self._m:
            lib().cs_matcher_destroy(self.ctx.ptr, self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBmatcherStream:
    """SearchByProjection(CurrentFrame, LastFrame) for a whole window of a stream that an ORBextractor holds in HBM (cs_match_by_projection_stream)."""

    def __init__(self, mbCheckOrientation=True, ctx=None):
        self.ctx = ctx
        self.mbCheckOrientation = bool(mbCheckOrientation)
        self._m = C.c_void_p()
        check(ctx.ptr, lib().cs_match_stream_create(C.byref(self._m)), "cs_match_stream_create")

    def search(self, orb, f0, n_pairs, K4, dist5, bounds, world_pos, valid, blocks, Tcw, fx, fy, cx, cy, scale_factors, th, n_train, mp_desc=None):
        """world_pos / valid / blocks: concatenated over the last frames f0 .. f0 + n_pairs - 1; Tcw: (n_pairs, 3, 4); n_train: key points of frames f0 + 1 .. f0 + n_pairs.
        Returns (train_match concatenated over the current frames, nmatches per pair)."""
        k4 = np.ascontiguousarray(K4, np.float32); d5 = None if dist5 is None else np.ascontiguousarray(dist5, np.float32)
        wp = np.ascontiguousarray(world_pos, np.float32); va = np.ascontiguousarray(valid, np.uint8); bl = np.ascontiguousarray(blocks, np.uint8)
        T = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(n_pairs, -1)[:, :12]); sf = np.ascontiguousarray(scale_factors, np.float32)
        md = None if mp_desc is None else np.ascontiguousarray(mp_desc, np.uint8)
        tm = np.zeros(max(int(n_train), 1), np.int32); nm = np.zeros(n_pairs, np.int32)
        check(self.ctx.ptr, lib().cs_match_by_projection_stream(self.ctx.ptr, self._m, orb._e, int(f0), int(n_pairs), _p(k4, C.c_float), None if d5 is None else _p(d5, C.c_float),
                                                                C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), len(va), int(n_train), _p(wp, C.c_float), _p(va, C.c_uint8),
                                                                _p(bl, C.c_uint8), None if md is None else _p(md, C.c_uint8), _p(T, C.c_float), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                                                C.c_float(cy), _p(sf, C.c_float), len(sf), C.c_float(th), int(self.mbCheckOrientation), _p(tm, C.c_int), _p(nm, C.c_int)),
              "cs_match_by_projection_stream")
        return tm[:int(n_train)], nm

    def last_counts(self):
        q, c = C.c_long(), C.c_long()
        check(self.ctx.ptr, lib().cs_match_stream_last_counts(self._m, C.byref(q), C.byref(c)), "cs_match_stream_last_counts")
        return {"queries": q.value, "candidates": c.value}

    def close(self):
        if self._m:
            lib().cs_match_stream_destroy(self.ctx.ptr, self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hamming_knn2(ctx, q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    bi = np.zeros(max(len(q), 1), np.int32); bd = np.zeros(max(len(q), 1), np.int32); sd = np.zeros(max(len(q), 1), np.int32)
    check(ctx.ptr, lib().cs_hamming_knn2(ctx.ptr, _p(q, C.c_uint8), len(q), _p(t, C.c_uint8), len(t), _p(bi, C.c_int), _p(bd, C.c_int), _p(sd, C.c_int)), "cs_hamming_knn2")
    return bi[:len(q)], bd[:len(q)], sd[:len(q)]


def frame_image_bounds(cols, rows, K4, dist5=None):
    """Frame::ComputeImageBounds -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    k4 = np.ascontiguousarray(K4, np.float32)
    d5 = None if dist5 is None else np.ascontiguousarray(dist5, np.float32)
    b = np.zeros(4, np.float32)
    r = lib().cs_frame_image_bounds(int(cols), int(rows), _p(k4, C.c_float), None if d5 is None else _p(d5, C.c_float), _p(b, C.c_float))
    if r != 0:
        raise RuntimeError("cs_frame_image_bounds failed: %d" % r)
    return b
