// Test harness (not part of the product): runs the per-item bodies of cube_slam_amd/csrc/badyn_math.h -- the code the badyn.hip kernels
// wrap -- serially on the CPU, so that the residuals, the quadratic form, the Schur complement, the back substitution and the vertex update
// can be compared with the oracle without a GPU.  g++ -shared -fPIC -ffp-contract=off tests/cpp/badyn_items.cpp
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "../../cube_slam_amd/csrc/badyn_math.h"
#include "../../cube_slam_amd/csrc/badyn_lists.h"

namespace {
struct Host {
    DynG G;
    std::vector<double> state, obj_scale, e[6], Hpp, bp, Hll, bl, Bslot, S, bs, Dinv, xp, xl;
    DynLists X;
    std::vector<int> pc_off;
    std::vector<double> BD, bsub;
    int n_edges = 0, n_vertices = 0;
    explicit Host(const cs_ba_dyn_problem *p) {
        std::memset(&G, 0, sizeof(G));
        G.n_cams = p->n_cams; G.n_objs = p->n_objs; G.n_vels = p->n_vels; G.n_pts = p->n_points; G.n_dpts = p->n_dpoints; G.fix_points = p->fix_points ? 1 : 0;
        G.n_obs = p->n_obs; G.n_dobs = p->n_dobs; G.n_mot = p->n_mot; G.n_cobs = p->n_cobs; G.n_pc = p->n_pc;
        G.fx = p->fx; G.fy = p->fy; G.cx = p->cx; G.cy = p->cy; G.bf = p->bf; G.huber_mono = p->huber_mono; G.huber_stereo = p->huber_stereo; G.huber_dyn = p->huber_dyn;
        G.huber_obj = p->huber_obj; G.ulp_info = p->ulp_info; G.ulp_ratio = p->ulp_ratio; G.pc_ratio = p->pc_ratio;
        for (int k = 0; k < 9; k++) G.K[k] = p->K[k];
        for (int k = 0; k < 3; k++) { G.ulp_scale[k] = p->ulp_scale[k]; G.mot_info[k] = p->mot_info[k]; }
        dyn_build_lists(p, X);
        const int NP = X.NP;
        G.NP = NP; G.L = X.L;
        const int n_slots = X.n_slots;
        G.n_blocks = (int)X.blk_ou.size(); G.n_vtx = (int)X.vtx_off.size();
        for (std::vector<int> *v : {&X.blk_ou, &X.blk_ot, &X.pair_u, &X.pair_t, &X.vtx_off, &X.vtx_slots}) v->push_back(0); // never empty
        const size_t o_obj = (size_t)p->n_cams * 7, o_vel = o_obj + (size_t)p->n_objs * 7, o_pts = o_vel + (size_t)p->n_vels * 2, o_dp = o_pts + (size_t)p->n_points * 3;
        state.assign(o_dp + (size_t)p->n_dpoints * 3 + 1, 0.0);
        for (int i = 0; i < p->n_cams; i++) { SE3 T = se3_load(p->cam_pose + (size_t)i * 7); normalize_rotation(T); se3_store(T, &state[(size_t)i * 7]); }
        for (int i = 0; i < p->n_objs; i++) { SE3 T = se3_load(p->obj_pose + (size_t)i * 7); normalize_rotation(T); se3_store(T, &state[o_obj + (size_t)i * 7]); }
        for (int i = 0; i < p->n_vels * 2; i++) state[o_vel + i] = p->vel[i];
        for (int i = 0; i < p->n_points * 3; i++) state[o_pts + i] = p->points[i];
        for (int i = 0; i < p->n_dpoints * 3; i++) state[o_dp + i] = p->dpoints[i];
        G.cam = state.data(); G.obj = state.data() + o_obj; G.vel = state.data() + o_vel; G.pts = state.data() + o_pts; G.dpts = state.data() + o_dp;
        G.obj_scale = p->obj_scale; G.obj_flags = p->obj_flags; G.cam_off = X.cam_off.data(); G.obj_off = X.obj_off.data(); G.vel_off = X.vel_off.data();
        G.o_cam = p->obs_cam; G.o_pt = p->obs_point; G.o_uv = p->obs_uv; G.o_ur = p->obs_ur; G.o_w = p->obs_inv_sigma2; G.o_lvl = p->obs_level;
        G.d_cam = p->dobs_cam; G.d_obj = p->dobs_obj; G.d_pt = p->dobs_point; G.d_uv = p->dobs_uv; G.d_w = p->dobs_inv_sigma2; G.d_lvl = p->dobs_level;
        G.m_from = p->mot_from; G.m_to = p->mot_to; G.m_vel = p->mot_vel; G.m_dt = p->mot_dt;
        G.c_cam = p->cobs_cam; G.c_obj = p->cobs_obj; G.c_bbox = p->cobs_bbox; G.c_info = p->cobs_info; G.c_lvl = p->cobs_level;
        pc_off.assign(p->n_pc + 1, 0);
        for (int i = 0; i <= p->n_pc && p->n_pc; i++) pc_off[i] = p->pc_offsets[i];
        G.pc_obj = p->pc_obj; G.pc_off = pc_off.data(); G.pc_pts = p->pc_points;
        const int sizes[6] = {p->n_obs * 3, p->n_dobs * 2, p->n_mot * 3, p->n_cobs * 4, p->n_pc * 3, p->n_dpoints * 3};
        for (int k = 0; k < 6; k++) e[k].assign(sizes[k] + 1, 0.0);
        G.e_obs = e[0].data(); G.e_dobs = e[1].data(); G.e_mot = e[2].data(); G.e_cobs = e[3].data(); G.e_pc = e[4].data(); G.e_ulp = e[5].data();
        Hpp.assign((size_t)NP * NP + 1, 0.0); S = Hpp; bp.assign(NP + 1, 0.0); bs = bp; xp = bp;
        Hll.assign((size_t)G.L * 9 + 1, 0.0); Dinv = Hll; bl.assign((size_t)G.L * 3 + 1, 0.0); xl = bl; Bslot.assign((size_t)n_slots * 18 + 1, 0.0);
        BD.assign((size_t)n_slots * 18 + 1, 0.0); bsub.assign((size_t)n_slots * 6 + 1, 0.0);
        G.Hpp = Hpp.data(); G.bp = bp.data(); G.Hll = Hll.data(); G.bl = bl.data(); G.Bslot = Bslot.data(); G.slot_off = X.slot_off.data();
        G.BD = BD.data(); G.bsub = bsub.data(); G.slot_lm = X.slot_lm.data(); G.blk_ou = X.blk_ou.data(); G.blk_ot = X.blk_ot.data(); G.blk_start = X.blk_start.data();
        G.pair_u = X.pair_u.data(); G.pair_t = X.pair_t.data(); G.vtx_off = X.vtx_off.data(); G.vtx_start = X.vtx_start.data(); G.vtx_slots = X.vtx_slots.data();
        G.lm_start = X.lm_start.data(); G.lm_slots = X.lm_slots.data(); G.S = S.data(); G.bs = bs.data(); G.Dinv = Dinv.data(); G.xp = xp.data(); G.xl = xl.data();
        n_edges = dyn_n_edges(G); n_vertices = p->n_cams + p->n_objs + p->n_vels + p->n_points + p->n_dpoints;
    }
    double errors() { double chi = 0; for (int e2 = 0; e2 < n_edges; e2++) chi += dyn_error_item(G, e2); return chi; }
    void reduce(double lambda) {
        for (int e2 = 0; e2 < n_edges; e2++) dyn_lin_item<-1>(G, e2);
        for (int i = 0; i < G.NP; i++) { for (int j = 0; j < G.NP; j++) S[(size_t)i * G.NP + j] = Hpp[(size_t)i * G.NP + j] + (i == j ? lambda : 0.0); bs[i] = bp[i]; }
        for (int l = 0; l < G.L; l++) dyn_dinv_item(G, l, lambda);
        for (int sl = 0; sl < X.n_slots; sl++) dyn_bd_item(G, sl);
        for (int k = 0; k < G.n_blocks; k++) for (int e2 = 0; e2 < 36; e2++) { double acc = 0; for (int g = 0; g < 7; g++) acc += dyn_schur_block_partial(G, k, e2, g, 7); dyn_schur_block_store(G, k, e2, acc); }
        for (int v = 0; v < G.n_vtx; v++) for (int a = 0; a < 6; a++) { double acc = 0; for (int g = 0; g < 10; g++) acc += dyn_rhs_partial(G, v, a, g, 10); dyn_rhs_store(G, v, a, acc); }
    }
};
} // namespace

extern "C" {
// chi2, the residual arrays (concatenated: obs x3, dobs x2, mot x3, cobs x4, pc x3, ulp x3) and the reduced system; returns NP
int badyn_items_reduced(const cs_ba_dyn_problem *p, double lambda, double *chi, double *errs, double *S, double *bs) {
    Host h(p);
    *chi = h.errors();
    const int sizes[6] = {p->n_obs * 3, p->n_dobs * 2, p->n_mot * 3, p->n_cobs * 4, p->n_pc * 3, p->n_dpoints * 3};
    for (int k = 0, o = 0; k < 6; o += sizes[k], k++) if (errs) std::memcpy(errs + o, h.e[k].data(), sizeof(double) * sizes[k]);
    if (!S) return h.G.NP;
    h.reduce(lambda);
    std::memcpy(S, h.S.data(), sizeof(double) * (size_t)h.G.NP * h.G.NP);
    std::memcpy(bs, h.bs.data(), sizeof(double) * h.G.NP);
    return h.G.NP;
}
// one linear step: reduce, dense solve (plain Cholesky here; the product uses badyn_chol_solve), back substitution items, update items
int badyn_items_step(const cs_ba_dyn_problem *p, double lambda, double *state_out) {
    Host h(p);
    h.errors();
    h.reduce(lambda);
    const int n = h.G.NP;
    std::vector<double> A(h.S.begin(), h.S.begin() + (size_t)n * n), x(h.bs.begin(), h.bs.begin() + n);
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0)) return 1;
        d = std::sqrt(d); A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double v = A[(size_t)i * n + j]; for (int k = 0; k < j; k++) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k]; A[(size_t)i * n + j] = v / d; }
    }
    for (int i = 0; i < n; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= A[(size_t)i * n + k] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    for (int i = 0; i < n; i++) h.xp[i] = x[i];
    for (int l = 0; l < h.G.L; l++) dyn_backsub_item(h.G, l);
    for (int v = 0; v < h.n_vertices; v++) dyn_update_item(h.G, v);
    std::memcpy(state_out, h.state.data(), sizeof(double) * (h.state.size() - 1));
    return 0;
}
}
