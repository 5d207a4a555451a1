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
};

inline void dyn_build_lists(const cs_ba_dyn_problem *p, DynLists &X) {
    X.cam_off.assign(std::max(p->n_cams, 1), -1); X.obj_off.assign(std::max(p->n_objs, 1), -1); X.vel_off.assign(std::max(p->n_vels, 1), -1);
    int NP = 0;
    for (int i = 0; i < p->n_cams; i++) if (!p->cam_fixed[i]) { X.cam_off[i] = NP; NP += 6; }
    for (int i = 0; i < p->n_objs; i++) { X.obj_off[i] = NP; NP += 6; }
    for (int i = 0; i < p->n_vels; i++) { X.vel_off[i] = NP; NP += 2; }
    X.NP = NP; X.L = p->fix_points ? 0 : p->n_points + p->n_dpoints;
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
