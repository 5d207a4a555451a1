// Host-side index structures of the dynamic-object BA (badyn.hip): the pose-landmark slots of every landmark, the target blocks of the Schur
// complement with the slot pairs that feed them, and the slots of every pose vertex.  Plain C++ (also compiled into the CPU item harness
// tests/cpp/badyn_items.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/cubeslam_hip.h"

struct DynLists {
    int NP = 0, L = 0, n_slots = 0;
    std::vector<int> cam_off, obj_off, vel_off; // pose-system offsets (-1: fixed camera)
    std::vector<int> slot_off, slot_lm;         // slot -> offset of its six pose rows (-1: unused), landmark
    std::vector<int> lm_start, lm_slots;        // landmark -> its used slots
    std::vector<int> blk_ou, blk_ot, blk_start; // Schur target block (rows ou, columns ot, ou >= ot) -> its slot pairs
    std::vector<int> pair_u, pair_t;
    std::vector<int> vtx_off, vtx_start, vtx_slots; // pose vertex that owns slots -> those slots
    // deterministic build: targets of the per-edge pieces (see DynG::stage), every list in edge order
    std::vector<int> pb_lo, pb_hi, pb_dim, pb_start, pb_src;   // pose x pose block (rows lo, columns hi, lo <= hi; dim = rows | columns << 8) <- (edge * 6 + pair) * 2 + transposed
    std::vector<int> pg_off, pg_dim, pg_start, pg_src;        // pose vertex gradient <- edge * 3 + vertex
    std::vector<int> lg_start, lg_src;                         // landmark (3x3 block + gradient) <- edge * 3 + vertex
};


// vertices of edge e as badyn_math.h's dyn_lin_item sets them: pose-system offset (or -1), dimension, landmark (or -1); false = the edge is not linearised
inline bool dyn_edge_vertices(const cs_ba_dyn_problem *p, const DynLists &X, int e, int &nv, int *off, int *dim, int *lm) {
    auto lvl = [](const uint8_t *a, int o) { return a && a[o]; };
    for (int v = 0; v < 3; v++) { off[v] = -1; dim[v] = 0; lm[v] = -1; }
    int o = e;
    if (o < p->n_obs) {
        if (lvl(p->obs_level, o)) return false;
        nv = 2; off[0] = X.cam_off[p->obs_cam[o]]; dim[0] = 6; lm[1] = p->fix_points ? -1 : p->obs_point[o]; dim[1] = 3;
        return !(off[0] < 0 && lm[1] < 0);
    }
    o -= p->n_obs;
    if (o < p->n_dobs) {
        if (lvl(p->dobs_level, o)) return false;
        nv = 3; off[0] = X.cam_off[p->dobs_cam[o]]; dim[0] = 6; off[1] = X.obj_off[p->dobs_obj[o]]; dim[1] = 6; lm[2] = p->fix_points ? -1 : p->n_points + p->dobs_point[o]; dim[2] = 3;
        return true;
    }
    o -= p->n_dobs;
    if (o < p->n_mot) { nv = 3; off[0] = X.obj_off[p->mot_from[o]]; dim[0] = 6; off[1] = X.obj_off[p->mot_to[o]]; dim[1] = 6; off[2] = X.vel_off[p->mot_vel[o]]; dim[2] = 2; return true; }
    o -= p->n_mot;
    if (o < p->n_cobs) { if (lvl(p->cobs_level, o)) return false; nv = 2; off[0] = X.cam_off[p->cobs_cam[o]]; dim[0] = 6; off[1] = X.obj_off[p->cobs_obj[o]]; dim[1] = 6; return true; }
    o -= p->n_cobs;
    if (o < p->n_pc) { nv = 1; off[0] = X.obj_off[p->pc_obj[o]]; dim[0] = 6; return true; }
    o -= p->n_pc;
    if (p->fix_points) return false;
    nv = 1; lm[0] = p->n_points + o; dim[0] = 3;
    return true;
}
inline void dyn_build_gather_lists(const cs_ba_dyn_problem *p, DynLists &X) {
    const int n_edges = p->n_obs + p->n_dobs + p->n_mot + p->n_cobs + p->n_pc + p->n_dpoints;
    struct Ent { long long key; int src, dim; };
    std::vector<Ent> pb, pg, lg;
    for (int e = 0; e < n_edges; e++) {
        int nv = 0, off[3], dim[3], lm[3];
        if (!dyn_edge_vertices(p, X, e, nv, off, dim, lm)) continue;
        for (int i = 0; i < nv; i++) {
            if (lm[i] >= 0) lg.push_back(Ent{lm[i], e * 3 + i, 3});
            else if (off[i] >= 0) pg.push_back(Ent{off[i], e * 3 + i, dim[i]});
            for (int j = i; j < nv; j++) {
                if (off[i] < 0 || off[j] < 0) continue;
                const int pair = i * 3 + j - i * (i + 1) / 2;
                if (off[i] <= off[j]) pb.push_back(Ent{(long long)off[i] * (X.NP + 1) + off[j], (e * 6 + pair) * 2, dim[i] | dim[j] << 8});
                else pb.push_back(Ent{(long long)off[j] * (X.NP + 1) + off[i], (e * 6 + pair) * 2 + 1, dim[j] | dim[i] << 8});
            }
        }
    }
    auto by_key = [](const Ent &a, const Ent &b) { return a.key < b.key; };
    std::stable_sort(pb.begin(), pb.end(), by_key); std::stable_sort(pg.begin(), pg.end(), by_key); std::stable_sort(lg.begin(), lg.end(), by_key);
    X.pb_lo.clear(); X.pb_hi.clear(); X.pb_dim.clear(); X.pb_start.assign(1, 0); X.pb_src.clear();
    for (size_t i = 0; i < pb.size(); i++) {
        if (i == 0 || pb[i].key != pb[i - 1].key) { if (i) X.pb_start.push_back((int)i); X.pb_lo.push_back((int)(pb[i].key / (X.NP + 1))); X.pb_hi.push_back((int)(pb[i].key % (X.NP + 1))); X.pb_dim.push_back(pb[i].dim); }
        X.pb_src.push_back(pb[i].src);
    }
    if (!pb.empty()) X.pb_start.push_back((int)pb.size());
    X.pg_off.clear(); X.pg_dim.clear(); X.pg_start.assign(1, 0); X.pg_src.clear();
    for (size_t i = 0; i < pg.size(); i++) {
        if (i == 0 || pg[i].key != pg[i - 1].key) { if (i) X.pg_start.push_back((int)i); X.pg_off.push_back((int)pg[i].key); X.pg_dim.push_back(pg[i].dim); }
        X.pg_src.push_back(pg[i].src);
    }
    if (!pg.empty()) X.pg_start.push_back((int)pg.size());
    const int L = p->fix_points ? 0 : p->n_points + p->n_dpoints;
    X.lg_start.assign((size_t)L + 1, 0); X.lg_src.resize(lg.size());
    for (const Ent &t : lg) X.lg_start[(size_t)t.key + 1]++;
    for (int l = 0; l < L; l++) X.lg_start[l + 1] += X.lg_start[l];
    for (size_t i = 0; i < lg.size(); i++) X.lg_src[i] = lg[i].src; // sorted by landmark, stable: already in CSR order
}

inline void dyn_build_lists(const cs_ba_dyn_problem *p, DynLists &X) {
    X.cam_off.assign(std::max(p->n_cams, 1), -1); X.obj_off.assign(std::max(p->n_objs, 1), -1); X.vel_off.assign(std::max(p->n_vels, 1), -1);
    int NP = 0;
    for (int i = 0; i < p->n_cams; i++) if (!p->cam_fixed[i]) { X.cam_off[i] = NP; NP += 6; }
    for (int i = 0; i < p->n_objs; i++) { X.obj_off[i] = NP; NP += 6; }
    for (int i = 0; i < p->n_vels; i++) { X.vel_off[i] = NP; NP += 2; }
    X.NP = NP; X.L = p->fix_points ? 0 : p->n_points + p->n_dpoints;
    dyn_build_gather_lists(p, X);
    const int n_slots = X.n_slots = p->n_obs + 2 * p->n_dobs;
    X.slot_off.assign(std::max(n_slots, 1), -1); X.slot_lm.assign(std::max(n_slots, 1), -1);
    X.lm_start.assign((size_t)X.L + 1, 0); X.lm_slots.assign(std::max(n_slots, 1), 0);
    X.blk_ou.clear(); X.blk_ot.clear(); X.blk_start.assign(1, 0); X.pair_u.clear(); X.pair_t.clear();
    X.vtx_off.clear(); X.vtx_start.assign(1, 0); X.vtx_slots.clear();
    if (X.L == 0) return;
    auto lvl = [](const uint8_t *a, int o) { return a && a[o]; };
    for (int o = 0; o < p->n_obs; o++) if (!lvl(p->obs_level, o)) { X.slot_off[o] = X.cam_off[p->obs_cam[o]]; X.slot_lm[o] = p->obs_point[o]; }
    for (int o = 0; o < p->n_dobs; o++) if (!lvl(p->dobs_level, o)) {
        const int s = p->n_obs + 2 * o;
        X.slot_off[s] = X.cam_off[p->dobs_cam[o]]; X.slot_off[s + 1] = X.obj_off[p->dobs_obj[o]];
        X.slot_lm[s] = X.slot_lm[s + 1] = p->n_points + p->dobs_point[o];
    }
    for (int s = 0; s < n_slots; s++) if (X.slot_off[s] >= 0) X.lm_start[X.slot_lm[s] + 1]++;
    for (int l = 0; l < X.L; l++) X.lm_start[l + 1] += X.lm_start[l];
    { std::vector<int> pos(X.lm_start.begin(), X.lm_start.end() - 1); for (int s = 0; s < n_slots; s++) if (X.slot_off[s] >= 0) X.lm_slots[pos[X.slot_lm[s]]++] = s; }
    // slot pairs of every landmark, keyed by their target block
    struct PairKey { long long key; int u, t; };
    std::vector<PairKey> pairs;
    for (int l = 0; l < X.L; l++)
        for (int a = X.lm_start[l]; a < X.lm_start[l + 1]; a++)
            for (int b = X.lm_start[l]; b < X.lm_start[l + 1]; b++) {
                const int su = X.lm_slots[a], st = X.lm_slots[b], ou = X.slot_off[su], ot = X.slot_off[st];
                if (ou < ot) continue; // lower triangle; ou == ot keeps every ordered pair (the diagonal block is written in full)
                pairs.push_back(PairKey{(long long)ou * (NP + 1) + ot, su, st});
            }
    std::stable_sort(pairs.begin(), pairs.end(), [](const PairKey &a, const PairKey &b) { return a.key < b.key; });
    X.pair_u.resize(pairs.size()); X.pair_t.resize(pairs.size());
    for (size_t i = 0; i < pairs.size(); i++) {
        if (i == 0 || pairs[i].key != pairs[i - 1].key) {
            if (i) X.blk_start.push_back((int)i);
            X.blk_ou.push_back((int)(pairs[i].key / (NP + 1))); X.blk_ot.push_back((int)(pairs[i].key % (NP + 1)));
        }
        X.pair_u[i] = pairs[i].u; X.pair_t[i] = pairs[i].t;
    }
    if (!pairs.empty()) X.blk_start.push_back((int)pairs.size());
    // slots of every pose vertex (for the reduced right-hand side)
    std::vector<std::pair<int, int>> vs;
    for (int s = 0; s < n_slots; s++) if (X.slot_off[s] >= 0) vs.push_back(std::make_pair(X.slot_off[s], s));
    std::sort(vs.begin(), vs.end());
    for (size_t i = 0; i < vs.size(); i++) {
        if (i == 0 || vs[i].first != vs[i - 1].first) { if (i) X.vtx_start.push_back((int)i); X.vtx_off.push_back(vs[i].first); }
        X.vtx_slots.push_back(vs[i].second);
    }
    if (!vs.empty()) X.vtx_start.push_back((int)vs.size());
}
