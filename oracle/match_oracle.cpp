/*
 * oracle/match_oracle.cpp -- CPU oracle for ORB_SLAM2::ORBmatcher's Hamming searches.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PINNED: the three window searches, the Frame grid, DescriptorDistance, ComputeThreeMaxima, both SearchByBoW
 * overloads, SearchForTriangulation (with CheckDistEpipolarLine) and Fuse -- tests/test_ref_pins.py runs the reference's own text (oracle/_ref, cut out at
 * build time, against stand-ins for Frame / KeyFrame / MapPoint) on the same inputs: identical match lists and counts.  Restated from
 * /root/reference/orb_object_slam/src/ORBmatcher.cc and src/Frame.cc (grid).  Monocular paths only (mvuRight < 0,
 * bForward = bBackward = false); cv::Mat float products Rcw*x+tcw follow cv::gemm (double accumulation, one rounding).
 */
#include "oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <map>
#include <vector>

namespace {
const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30; // ORBmatcher.cc:42-44
const int GRID_ROWS = 48, GRID_COLS = 64;               // Frame.h:32-33

static int descriptor_distance(const uint8_t *a, const uint8_t *b) { // :1905-1921
    const uint32_t *pa = (const uint32_t *)a, *pb = (const uint32_t *)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct Grid {
    std::vector<int> cell[GRID_COLS][GRID_ROWS];
    float wInv, hInv;
    const orc_frame *F;
    explicit Grid(const orc_frame *f) : F(f) { // Frame.cc:285-286, AssignFeaturesToGrid :303-318, PosInGrid :525-535
        wInv = static_cast<float>(GRID_COLS) / static_cast<float>(f->maxX - f->minX);
        hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(f->maxY - f->minY);
        for (int i = 0; i < f->N; i++) {
            int px = (int)std::round((f->keysUn[i].x - f->minX) * wInv);
            int py = (int)std::round((f->keysUn[i].y - f->minY) * hInv);
            if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
            cell[px][py].push_back(i);
        }
    }
    void area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int> &out) const { // :404-459
        out.clear();
        const int nMinCellX = std::max(0, (int)std::floor((x - F->minX - r) * wInv));
        if (nMinCellX >= GRID_COLS) return;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - F->minX + r) * wInv));
        if (nMaxCellX < 0) return;
        const int nMinCellY = std::max(0, (int)std::floor((y - F->minY - r) * hInv));
        if (nMinCellY >= GRID_ROWS) return;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - F->minY + r) * hInv));
        if (nMaxCellY < 0) return;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int id : cell[ix][iy]) {
                    const orc_keypoint &kp = F->keysUn[id];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(id);
                }
    }
};

static void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3) { // :1860-1901
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}
static inline int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
} // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) { return descriptor_distance(a, b); }

int orc_get_features_in_area(const orc_frame *F, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap) {
    Grid g(F);
    std::vector<int> v;
    g.area(x, y, r, minLevel, maxLevel, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

int orc_search_by_projection_frame(const orc_frame *cur, int n_last, const float *world_pos, const uint8_t *valid, const uint8_t *blocks,
                                   const uint8_t *mp_desc, const int *last_octave, const float *last_angle, const float *T, float fx, float fy,
                                   float cx, float cy, const float *scale_factors, float th, int check_orientation, const uint8_t *train_blocked, int *train_match) {
    Grid g(cur);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < cur->N; i++) train_match[i] = -1;
    std::vector<int> vIndices2;
    for (int i = 0; i < n_last; i++) {
        if (!valid[i]) continue;
        float x3Dc[3];
        for (int r = 0; r < 3; r++) { // cv::gemm: double accumulate, alpha*sum + beta*c, one rounding to float
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)T[r * 4 + k] * (double)world_pos[i * 3 + k];
            x3Dc[r] = (float)(s * 1.0 + (double)T[r * 4 + 3] * 1.0);
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < cur->minX || u > cur->maxX) continue;
        if (v < cur->minY || v > cur->maxY) continue;
        int nLastOctave = last_octave[i];
        float radius = th * scale_factors[nLastOctave];
        g.area(u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (train_match[i2] >= 0 && blocks[train_match[i2]]) continue; // mvpMapPoints[i2] && Observations() > 0 (set during this call)
            if (train_blocked && train_blocked[i2]) continue;             // ... set before the call, or KeysStatic[i2] == false (:1451-1457)
            const int dist = descriptor_distance(dMP, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            train_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) rotHist[rot_bin(last_angle[i], cur->keysUn[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int id : rotHist[i]) { train_match[id] = -1; nmatches--; }
    }
    return nmatches;
}

int orc_search_local_map(const orc_frame *F, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level, const uint8_t *in_view,
                         const uint8_t *blocks, const uint8_t *mp_desc, const float *scale_factors, float th, float nnratio,
                         const uint8_t *train_blocked, int *train_match) {
    Grid g(F);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int i = 0; i < F->N; i++) train_match[i] = -1;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < n_mp; iMP++) {
        if (!in_view[iMP]) continue;
        const int nPredictedLevel = pred_level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f; // RadiusByViewingCos :144-150
        if (bFactor) r *= th;
        g.area(proj_xy[iMP * 2], proj_xy[iMP * 2 + 1], r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *d0 = mp_desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (train_blocked && train_blocked[idx]) continue;
            if (train_match[idx] >= 0 && blocks[train_match[idx]]) continue;
            const int dist = descriptor_distance(d0, F->desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->keysUn[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = F->keysUn[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            train_match[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

int orc_search_for_initialization(const orc_frame *F1, const orc_frame *F2, float *prev, int windowSize, float nnratio, int check_orientation,
                                  int *vnMatches12) {
    Grid g(F2);
    int nmatches = 0;
    for (int i = 0; i < F1->N; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(F2->N, INT_MAX), vnMatches21(F2->N, -1);
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < F1->N; i1++) {
        const orc_keypoint &kp1 = F1->keysUn[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        g.area(prev[i1 * 2], prev[i1 * 2 + 1], (float)windowSize, level1, level1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *d1 = F1->desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            int dist = descriptor_distance(d1, F2->desc + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) rotHist[rot_bin(F1->keysUn[i1].angle, F2->keysUn[bestIdx2].angle)].push_back(i1);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < F1->N; i1++)
        if (vnMatches12[i1] >= 0) { prev[i1 * 2] = F2->keysUn[vnMatches12[i1]].x; prev[i1 * 2 + 1] = F2->keysUn[vnMatches12[i1]].y; }
    return nmatches;
}

void orc_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx, int *best_dist, int *second_dist) {
    for (int i = 0; i < nq; i++) {
        int b = INT_MAX, b2 = INT_MAX, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b) { b2 = b; b = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi; best_dist[i] = b; second_dist[i] = b2;
    }
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- Fuse / SearchForTriangulation
// ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (:852-1003), the search part: the map-point preamble (projection, image bounds,
// distance / viewing-angle tests, PredictScale) is the caller's and arrives as valid / u / v / ur / pred_level; what Replace /
// AddObservation do with (bestIdx, bestDist) is map bookkeeping outside this function.  Returns the number of map points with
// bestDist <= TH_LOW (nFused).
int orc_fuse(const orc_frame *F, const float *u_right, const float *inv_level_sigma2, const uint8_t *keys_static, int n_mp, const float *uv, const float *ur,
             const int *pred_level, const uint8_t *valid, const uint8_t *mp_desc, const float *scale_factors, float th, int *best_idx, int *best_dist) {
    Grid g(F);
    int nFused = 0;
    std::vector<int> vIndices;
    for (int i = 0; i < n_mp; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (!valid[i]) continue;
        const int nPredictedLevel = pred_level[i];
        const float u = uv[i * 2], v = uv[i * 2 + 1];
        const float radius = th * scale_factors[nPredictedLevel];
        g.area(u, v, radius, -1, -1, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            const orc_keypoint &kp = F->keysUn[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (keys_static && !keys_static[idx]) continue;
            if (u_right[idx] >= 0) { // stereo reprojection test :954-966
                const float ex = u - kp.x, ey = v - kp.y, er = ur[i] - u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = descriptor_distance(mp_desc + (size_t)i * 32, F->desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
        if (bestDist <= TH_LOW) nFused++;
    }
    return nFused;
}

// ORBmatcher::SearchForTriangulation (:679-850).  node1 / node2: the DBoW2 FeatureVector node of every feature (-1: none); inside a
// node the features are visited in ascending index (FeatureVector::addFeature appends while features are transformed in index
// order).  skip = has a map point; vbMatched2 is never set in the reference, so the keypoints of KF1 are independent.
int orc_search_for_triangulation(const orc_frame *F1, const int *node1, const uint8_t *skip1, const float *u_right1, const uint8_t *static1, const orc_frame *F2,
                                 const int *node2, const uint8_t *skip2, const float *u_right2, const uint8_t *static2, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation, int *matches12) {
    std::map<int, std::vector<int>> fv1, fv2;
    for (int i = 0; i < F1->N; i++) if (node1[i] >= 0) fv1[node1[i]].push_back(i);
    for (int i = 0; i < F2->N; i++) if (node2[i] >= 0) fv2[node2[i]].push_back(i);
    int nmatches = 0;
    for (int i = 0; i < F1->N; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto f1it = fv1.begin(), f1end = fv1.end();
    auto f2it = fv2.begin(), f2end = fv2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (int idx1 : f1it->second) {
                if (skip1[idx1]) continue;
                if (static1 && !static1[idx1]) continue;
                const bool bStereo1 = u_right1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const orc_keypoint &kp1 = F1->keysUn[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int idx2 : f2it->second) {
                    if (skip2[idx2]) continue;
                    const bool bStereo2 = u_right2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    if (static2 && !static2[idx2]) continue;
                    const int dist = descriptor_distance(F1->desc + (size_t)idx1 * 32, F2->desc + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orc_keypoint &kp2 = F2->keysUn[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2.x, distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2.octave]) continue;
                    }
                    // CheckDistEpipolarLine :152-169
                    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
                    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
                    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
                    const float num = a * kp2.x + b * kp2.y + c;
                    const float den = a * a + b * b;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < 3.84 * level_sigma2_2[kp2.octave]) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1.angle - F2->keysUn[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            ++f1it; ++f2it;
        } else if (f1it->first < f2it->first) f1it = fv1.lower_bound(f2it->first);
        else f2it = fv2.lower_bound(f1it->first);
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:171-310).  nodeKF / nodeF: FeatureVector node of every feature;
// skipKF: no usable map point (NULL / bad / dynamic / not static); skipF: KeysStatic says no.  matchesF[idxF] = index of the KF feature
// whose map point the frame feature receives, -1 none.  F features claimed by an earlier KF feature of the node are skipped.
int orc_search_by_bow(const orc_frame *KF, const int *nodeKF, const uint8_t *skipKF, const orc_frame *F, const int *nodeF, const uint8_t *skipF, float nnratio,
                      int check_orientation, int *matchesF) {
    std::map<int, std::vector<int>> fvK, fvF;
    for (int i = 0; i < KF->N; i++) if (nodeKF[i] >= 0) fvK[nodeKF[i]].push_back(i);
    for (int i = 0; i < F->N; i++) if (nodeF[i] >= 0) fvF[nodeF[i]].push_back(i);
    for (int i = 0; i < F->N; i++) matchesF[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto KFit = fvK.begin(), KFend = fvK.end();
    auto Fit = fvF.begin(), Fend = fvF.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            for (int realIdxKF : KFit->second) {
                if (skipKF[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int realIdxF : Fit->second) {
                    if (matchesF[realIdxF] >= 0) continue;
                    if (skipF && skipF[realIdxF]) continue;
                    const int dist = descriptor_distance(KF->desc + (size_t)realIdxKF * 32, F->desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matchesF[bestIdxF] = realIdxKF;
                        if (check_orientation) {
                            float rot = KF->keysUn[realIdxKF].angle - F->keysUn[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            ++KFit; ++Fit;
        } else if (KFit->first < Fit->first) KFit = fvK.lower_bound(Fit->first);
        else Fit = fvF.lower_bound(KFit->first);
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matchesF[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:544-677).  skip = feature without a usable map point.
int orc_search_by_bow_kf(const orc_frame *K1, const int *node1, const uint8_t *skip1, const orc_frame *K2, const int *node2, const uint8_t *skip2, float nnratio,
                         int check_orientation, int *matches12) {
    std::map<int, std::vector<int>> fv1, fv2;
    for (int i = 0; i < K1->N; i++) if (node1[i] >= 0) fv1[node1[i]].push_back(i);
    for (int i = 0; i < K2->N; i++) if (node2[i] >= 0) fv2[node2[i]].push_back(i);
    for (int i = 0; i < K1->N; i++) matches12[i] = -1;
    std::vector<bool> vbMatched2((size_t)K2->N, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    auto f1it = fv1.begin(), f1end = fv1.end();
    auto f2it = fv2.begin(), f2end = fv2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (int idx1 : f1it->second) {
                if (skip1[idx1]) continue;
                const uint8_t *d1 = K1->desc + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int idx2 : f2it->second) {
                    if (vbMatched2[idx2] || skip2[idx2]) continue;
                    const int dist = descriptor_distance(d1, K2->desc + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (check_orientation) {
                            float rot = K1->keysUn[idx1].angle - K2->keysUn[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            ++f1it; ++f2it;
        } else if (f1it->first < f2it->first) f1it = fv1.lower_bound(f2it->first);
        else f2it = fv2.lower_bound(f1it->first);
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

extern "C" void orc_three_maxima(const int *counts, int L, int *ind) { /* ComputeThreeMaxima over histogram bins holding counts[i] entries */
    std::vector<std::vector<int>> histo(L);
    for (int i = 0; i < L; i++) histo[i].resize(counts[i]);
    ind[0] = ind[1] = ind[2] = -1;
    three_maxima(histo.data(), L, ind[0], ind[1], ind[2]);
}
