// Per-item bodies of the dynamic-object bundle adjustment (badyn.hip): one call = one edge, one landmark or one vertex.  The kernels in
// badyn.hip are thin index wrappers around them.  Replaces the g2o machinery behind Optimizer::LocalBACameraPointObjectsDynamic
// (orb_object_slam/src/Optimizer.cc:1537-2573): VertexSE3Expmap / VertexCuboidFixScale / VelocityPlanarVelocity / VertexSBAPointXYZ,
// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (types_six_dof_expmap.cpp), EdgeDynamicPointCuboidCamera (g2o_Object.cpp:154-233, analytic
// Jacobians), EdgeObjectMotion (:241-272), EdgeSE3CuboidFixScaleProj (:118-128), EdgePointCuboidOnlyObjectFixScale (:336-354),
// UnaryLocalPoint (:378-398) -- the last four with g2o's central differences (delta 1e-9, base_multi_edge.hpp:72-120) -- the robustified
// quadratic form (base_multi_edge.hpp:36-48) and BlockSolverX's Schur complement over the marginalised points (block_solver.hpp:378-486).
//
// Layout in HBM: estimates in one state buffer (cameras 7, object poses 7, velocities 2, static points 3, dynamic points 3 doubles);
// the pose system is dense (NP scalars: non-fixed cameras 6, object poses 6, velocities 2): Hpp NP x NP, bp; landmarks (static points, then
// dynamic points) own Hll 3x3, bl; every (edge, pose vertex, landmark) triple owns one 6x3 slot of Hpl (a reprojection edge one, a
// dynamic-point edge two), listed per landmark in CSR form by the host.
#pragma once
#include "se3_math.h"

#include <cstdint>

#if defined(__HIPCC__)
#define BD_ATOMIC_ADD(ptr, v) unsafeAtomicAdd((ptr), (v))
#else
#define BD_ATOMIC_ADD(ptr, v) (*(ptr) += (v))
#endif

namespace {

struct DynG {
    int n_cams, n_objs, n_vels, n_pts, n_dpts, fix_points, NP, L;
    int n_obs, n_dobs, n_mot, n_cobs, n_pc;
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo, huber_dyn, huber_obj, K[9], ulp_info, ulp_scale[3], ulp_ratio, mot_info[3], pc_ratio;
    double *cam, *obj, *vel, *pts, *dpts;
    const double *obj_scale; const uint8_t *obj_flags;
    const int *cam_off, *obj_off, *vel_off;
    const int *o_cam, *o_pt; const double *o_uv, *o_ur, *o_w; const uint8_t *o_lvl;
    const int *d_cam, *d_obj, *d_pt; const double *d_uv, *d_w; const uint8_t *d_lvl;
    const int *m_from, *m_to, *m_vel; const double *m_dt;
    const int *c_cam, *c_obj; const double *c_bbox, *c_info; const uint8_t *c_lvl;
    const int *pc_obj, *pc_off; const double *pc_pts;
    double *e_obs, *e_dobs, *e_mot, *e_cobs, *e_pc, *e_ulp;
    double *Hpp, *bp, *Hll, *bl, *Bslot;
    // Deterministic build (device): every edge writes its J^T W J blocks and J^T W e vectors into its own DYN_STAGE doubles of `stage`; the gather
    // kernels of badyn.hip then add them per target in edge order -- the order a sequential loop over the edges (g2o, the oracle) adds them in.
    // stage == nullptr (host harness, oracle-style use): accumulate in place.
    double *stage;
    int n_pb, n_pg; const int *pb_lo, *pb_hi, *pb_dim, *pb_start, *pb_src, *pg_off, *pg_dim, *pg_start, *pg_src, *lg_start, *lg_src;
    const int *slot_off;               // pose-system offset of a slot's six rows, -1 = unused slot (fixed camera / inactive edge)
    const int *lm_start, *lm_slots;    // CSR: the slots of every landmark
    double *S, *bs, *Dinv, *xp, *xl;
    // Schur complement without atomics: B D^-1 and B D^-1 b_l per slot, then one wave per target block / one thread per right-hand-side row
    double *BD, *bsub; const int *slot_lm;
    int n_blocks, n_vtx;
    const int *blk_ou, *blk_ot, *blk_start, *pair_u, *pair_t, *vtx_off, *vtx_start, *vtx_slots;
};

HD int dyn_n_edges(const DynG &G) { return G.n_obs + G.n_dobs + G.n_mot + G.n_cobs + G.n_pc + G.n_dpts; }
HD bool dyn_stereo(const DynG &G, int o) { return G.o_ur && G.o_ur[o] >= 0; }
HD bool dyn_lvl(const uint8_t *a, int o) { return a && a[o]; }
HD Cuboid dyn_obj(const DynG &G, int i) {
    Cuboid c; c.pose = se3_load(G.obj + (long)i * 7);
    for (int k = 0; k < 3; k++) c.scale[k] = G.obj_scale[i * 3 + k];
    return c;
}
HD Cuboid dyn_obj_plus(const DynG &G, int i, const Cuboid &c, const double *add) { return cuboid_oplus(c, add, G.obj_flags[i], G.obj_scale + (long)i * 3); }

// ---------------------------------------------------------------------------------------------------- residuals
HD void dyn_err_obs(const DynG &G, int o, const SE3 &T, const double *X, double *e) {
    double pc[3];
    se3_map(T, X, pc);
    if (dyn_stereo(G, o)) { // EdgeStereoSE3ProjectXYZ::cam_project rounds 1/z and bf to float (types_six_dof_expmap.cpp:182-189)
        const float invz = (float)(1.0 / pc[2]);
        const double u = pc[0] * invz * G.fx + G.cx;
        e[0] = G.o_uv[o * 2] - u;
        e[1] = G.o_uv[o * 2 + 1] - (pc[1] * invz * G.fy + G.cy);
        e[2] = G.o_ur[o] - (u - (double)((float)G.bf * invz)); // (bf arrives as const float&: the product is a float product)
        return;
    }
    e[0] = G.o_uv[o * 2] - (pc[0] / pc[2] * G.fx + G.cx);
    e[1] = G.o_uv[o * 2 + 1] - (pc[1] / pc[2] * G.fy + G.cy);
    e[2] = 0.0;
}
HD void dyn_err_dobs(const DynG &G, int o, const SE3 &T, const SE3 &Two, const double *X, double *e) { // g2o_Object.cpp:154-165
    double pw[3], lp[3];
    se3_map(Two, X, pw);
    se3_map(T, pw, lp);
    e[0] = G.d_uv[o * 2] - (G.K[2] + G.K[0] * lp[0] / lp[2]);
    e[1] = G.d_uv[o * 2 + 1] - (G.K[5] + G.K[4] * lp[1] / lp[2]);
}
HD double dyn_yaw(const Quat &q) { return atan2(2 * (q.w * q.z + q.x * q.y), 1 - 2 * (q.y * q.y + q.z * q.z)); } // SE3Quat::toXYZPRYVector se3quat.h:184-207
HD void dyn_err_mot(const DynG &G, int o, const SE3 &from, const SE3 &to, const double *v, double *e) { // g2o_Object.cpp:241-272
    const double yaw_from = dyn_yaw(from.r), yaw_to = dyn_yaw(to.r), dt = G.m_dt[o];
    const double vehicle_length = 2.71;
    const double k1 = v[0] * dt - vehicle_length * 0.5;
    const double bx = from.t[0] + k1 * cos(yaw_from), by = from.t[1] + k1 * sin(yaw_from);
    const double yaw_pred = yaw_from + tan(v[1]) * dt / vehicle_length * v[0];
    const double k2 = vehicle_length * 0.5;
    e[0] = to.t[0] - (bx + k2 * cos(yaw_pred));
    e[1] = to.t[1] - (by + k2 * sin(yaw_pred));
    e[2] = yaw_to - yaw_pred;
    const double two_pi = 2.0 * 3.14159265358979323846;
    if (e[2] > two_pi) e[2] -= two_pi;
    if (e[2] < -two_pi) e[2] += two_pi;
}
HD void dyn_err_cobs(const DynG &G, int o, const SE3 &T, const Cuboid &c, double *e) { // g2o_Object.cpp:118-128
    double bb[4];
    project_bbox(c, T, G.K, bb);
    for (int k = 0; k < 4; k++) e[k] = bb[k] - G.c_bbox[o * 4 + k];
}
HD double dyn_margin_err(double a, double sc, double ratio) { // cuboid::point_boundary_error g2o_Object.cpp:280-298 / UnaryLocalPoint :386-394
    if (a < sc) return 0;
    if (a < (ratio + 1) * sc) return a - sc;
    return ratio * sc;
}
HD void dyn_err_pc(const DynG &G, int o, const Cuboid &c, double *e) { // g2o_Object.cpp:336-354
    double acc[3] = {0, 0, 0};
    const int b0 = G.pc_off[o], b1 = G.pc_off[o + 1];
    const SE3 inv = se3_inv(c.pose);
    for (int i = b0; i < b1; i++) {
        double lp[3];
        se3_map(inv, G.pc_pts + (long)i * 3, lp);
        for (int k = 0; k < 3; k++) acc[k] += fabs(dyn_margin_err(fabs(lp[k]) * 1.0, c.scale[k], G.pc_ratio));
    }
    if (b1 > b0) for (int k = 0; k < 3; k++) acc[k] = acc[k] / (double)(b1 - b0);
    for (int k = 0; k < 3; k++) e[k] = 1.0 * (acc[k] / c.scale[k]);
}
HD void dyn_err_ulp(const DynG &G, const double *X, double *e) { // g2o_Object.cpp:378-398
    for (int k = 0; k < 3; k++) e[k] = dyn_margin_err(fabs(X[k]), G.ulp_scale[k], G.ulp_ratio) / G.ulp_scale[k];
}

// edge e of the concatenated list [obs | dobs | mot | cobs | pc | ulp]: class and local index
HD int dyn_edge_class(const DynG &G, int e, int &o) {
    o = e;
    if (o < G.n_obs) return 0;
    o -= G.n_obs; if (o < G.n_dobs) return 1;
    o -= G.n_dobs; if (o < G.n_mot) return 2;
    o -= G.n_mot; if (o < G.n_cobs) return 3;
    o -= G.n_cobs; if (o < G.n_pc) return 4;
    o -= G.n_pc; return 5;
}
HD double dyn_chi2(const double *e, const double *w, int D) { double c = 0; for (int k = 0; k < D; k++) c += e[k] * w[k] * e[k]; return c; }
HD double dyn_robust(double c, double delta) { if (!(delta > 0)) return c; double rho[3]; huber(c, delta, rho); return rho[0]; }

// computeError of one edge (stored for every edge) and its share of activeRobustChi2 (sparse_optimizer.cpp:100-114; 0 for an edge that is
// not active: level != 0, or every vertex fixed)
HD double dyn_error_item(const DynG &G, int e) {
    int o;
    const int cls = dyn_edge_class(G, e, o);
    if (cls == 0) {
        double *r = G.e_obs + (long)o * 3;
        const int ci = G.o_cam[o];
        dyn_err_obs(G, o, se3_load(G.cam + (long)ci * 7), G.pts + (long)G.o_pt[o] * 3, r);
        if (dyn_lvl(G.o_lvl, o) || (G.fix_points && G.cam_off[ci] < 0)) return 0;
        const double w = G.o_w[o], ww[3] = {w, w, w};
        const bool st = dyn_stereo(G, o);
        return dyn_robust(dyn_chi2(r, ww, st ? 3 : 2), st ? G.huber_stereo : G.huber_mono);
    }
    if (cls == 1) {
        double *r = G.e_dobs + (long)o * 2;
        dyn_err_dobs(G, o, se3_load(G.cam + (long)G.d_cam[o] * 7), se3_load(G.obj + (long)G.d_obj[o] * 7), G.dpts + (long)G.d_pt[o] * 3, r);
        if (dyn_lvl(G.d_lvl, o)) return 0;
        const double w = G.d_w[o], ww[2] = {w, w};
        return dyn_robust(dyn_chi2(r, ww, 2), G.huber_dyn);
    }
    if (cls == 2) {
        double *r = G.e_mot + (long)o * 3;
        dyn_err_mot(G, o, se3_load(G.obj + (long)G.m_from[o] * 7), se3_load(G.obj + (long)G.m_to[o] * 7), G.vel + (long)G.m_vel[o] * 2, r);
        return dyn_chi2(r, G.mot_info, 3);
    }
    if (cls == 3) {
        double *r = G.e_cobs + (long)o * 4;
        dyn_err_cobs(G, o, se3_load(G.cam + (long)G.c_cam[o] * 7), dyn_obj(G, G.c_obj[o]), r);
        if (dyn_lvl(G.c_lvl, o)) return 0;
        return dyn_robust(dyn_chi2(r, G.c_info + (long)o * 4, 4), G.huber_obj);
    }
    const double one[3] = {1, 1, 1};
    if (cls == 4) {
        double *r = G.e_pc + (long)o * 3;
        dyn_err_pc(G, o, dyn_obj(G, G.pc_obj[o]), r);
        return dyn_chi2(r, one, 3);
    }
    double *r = G.e_ulp + (long)o * 3;
    dyn_err_ulp(G, G.dpts + (long)o * 3, r);
    if (G.fix_points) return 0;
    const double ul[3] = {G.ulp_info, G.ulp_info, G.ulp_info};
    return dyn_chi2(r, ul, 3);
}

// ---------------------------------------------------------------------------------------------------- quadratic form
constexpr int DYN_STAGE = 18 + 6 * 36; // per edge: three gradients of six, then the blocks of the vertex pairs (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
HD int dyn_pair_index(int i, int j) { return i * 3 + j - i * (i + 1) / 2; }
struct DynLin { // one linearised edge; vertex v: pose-system offset off[v] (dim 6 or 2), landmark lm[v] (dim 3, always the last vertex), or fixed
    int nv, D, off[3], dim[3], lm[3], slot[3], eid;
    double e[4], w[4], J[3][24], delta;
};
HD void dyn_lin_init(DynLin &E, int nv, int D) {
    E.nv = nv; E.D = D; E.delta = 0; E.eid = -1;
    for (int v = 0; v < 3; v++) { E.off[v] = -1; E.dim[v] = 0; E.lm[v] = -1; E.slot[v] = -1; }
}
template <int NV, int D> HD void dyn_add_edge(const DynG &G, const DynLin &E) { // vertex count and residual dimension at compile time: everything unrolls, E stays in registers
    double rw = 1.0;
    if (E.delta > 0) { double rho[3]; huber(dyn_chi2(E.e, E.w, D), E.delta, rho); rw = rho[1]; }
    double omr[D], W[D];
#pragma unroll
    for (int k = 0; k < D; k++) { omr[k] = -E.w[k] * E.e[k] * rw; W[k] = rw * E.w[k]; }
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const bool free_i = E.off[i] >= 0 || E.lm[i] >= 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            if (free_i && a < E.dim[i]) {
                double g = 0;
#pragma unroll
                for (int k = 0; k < D; k++) g += E.J[i][k * 6 + a] * omr[k];
                if (G.stage) G.stage[(long)E.eid * DYN_STAGE + i * 6 + a] = g;
                else if (E.lm[i] >= 0) BD_ATOMIC_ADD(G.bl + (long)E.lm[i] * 3 + a, g); else BD_ATOMIC_ADD(G.bp + E.off[i] + a, g);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const bool free_j = j >= i && (E.off[j] >= 0 || E.lm[j] >= 0); // upper triangle of the edge's vertex pairs
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    if (free_i && free_j && a < E.dim[i] && c < E.dim[j]) {
                        double h = 0;
#pragma unroll
                        for (int k = 0; k < D; k++) h += (E.J[i][k * 6 + a] * W[k]) * E.J[j][k * 6 + c];
                        if (E.lm[i] < 0 && E.lm[j] >= 0) G.Bslot[(long)E.slot[i] * 18 + a * 3 + c] = h;                // pose rows x landmark columns: own slot
                        else if (G.stage) G.stage[(long)E.eid * DYN_STAGE + 18 + dyn_pair_index(i, j) * 36 + a * 6 + c] = h;
                        else if (E.lm[i] >= 0) BD_ATOMIC_ADD(G.Hll + (long)E.lm[i] * 9 + a * 3 + c, h);         // landmark x landmark (i == j)
                        else {
                            BD_ATOMIC_ADD(G.Hpp + (long)(E.off[i] + a) * G.NP + E.off[j] + c, h);
                            if (i != j) BD_ATOMIC_ADD(G.Hpp + (long)(E.off[j] + c) * G.NP + E.off[i] + a, h);
                        }
                    }
                }
            }
        }
    }
}
HD void dyn_reproj_cam_jac(double X, double Y, double Z, double fx, double fy, double *J) { // rows of 6: the camera block both reprojection edges share
    const double Z2 = Z * Z;
    J[0] = X * Y / Z2 * fx; J[1] = -(1 + (X * X / Z2)) * fx; J[2] = Y / Z * fx; J[3] = -1. / Z * fx; J[4] = 0; J[5] = X / Z2 * fx;
    J[6] = (1 + Y * Y / Z2) * fy; J[7] = -X * Y / Z2 * fy; J[8] = -X / Z * fy; J[9] = 0; J[10] = -1. / Z * fy; J[11] = Y / Z2 * fy;
}

// linearizeOplus + constructQuadraticForm of one active edge (BlockSolver::buildSystem block_solver.hpp:502-560).  ONLY >= 0 compiles the
// body of that edge class alone (the kernels launch the two big classes separately: the generic body needs 256 VGPRs plus scratch).
template <int ONLY> HD void dyn_lin_item(const DynG &G, int e) {
    int o;
    const int cls = ONLY >= 0 ? (dyn_edge_class(G, e, o), ONLY) : dyn_edge_class(G, e, o);
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    DynLin E;
    if (cls == 0) {
        if (dyn_lvl(G.o_lvl, o)) return;
        const int ci = G.o_cam[o], li = G.o_pt[o];
        const bool st = dyn_stereo(G, o);
        dyn_lin_init(E, 2, st ? 3 : 2); E.eid = e;
        E.off[0] = G.cam_off[ci]; E.dim[0] = 6; E.slot[0] = o; E.lm[1] = G.fix_points ? -1 : li; E.dim[1] = 3;
        if (E.off[0] < 0 && E.lm[1] < 0) return;
        const SE3 T = se3_load(G.cam + (long)ci * 7);
        double pc[3], R[3][3];
        se3_map(T, G.pts + (long)li * 3, pc);
        qtoR(T.r, R);
        const double X = pc[0], Y = pc[1], Z = pc[2], Z2 = Z * Z, fx = G.fx, fy = G.fy;
        double *Jc = E.J[0], *Jp = E.J[1];
        if (!st) { // EdgeSE3ProjectXYZ::linearizeOplus types_six_dof_expmap.cpp:135-171
            const double tmp[2][3] = {{fx, 0, -X / Z * fx}, {0, fy, -Y / Z * fy}};
            for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Jp[r * 6 + c] = ((-1. / Z * tmp[r][0]) * R[0][c] + (-1. / Z * tmp[r][1]) * R[1][c]) + (-1. / Z * tmp[r][2]) * R[2][c];
        } else { // EdgeStereoSE3ProjectXYZ::linearizeOplus :220-266
            for (int c = 0; c < 3; c++) {
                Jp[c] = -fx * R[0][c] / Z + fx * X * R[2][c] / Z2;
                Jp[6 + c] = -fy * R[1][c] / Z + fy * Y * R[2][c] / Z2;
                Jp[12 + c] = Jp[c] - G.bf * R[2][c] / Z2;
            }
        }
        dyn_reproj_cam_jac(X, Y, Z, fx, fy, Jc);
        if (st) { Jc[12] = Jc[0] - G.bf * Y / Z2; Jc[13] = Jc[1] + G.bf * X / Z2; Jc[14] = Jc[2]; Jc[15] = Jc[3]; Jc[16] = 0; Jc[17] = Jc[5] - G.bf / Z2; }
        for (int k = 0; k < 3; k++) { E.e[k] = G.e_obs[(long)o * 3 + k]; E.w[k] = G.o_w[o]; } // the third residual of a monocular edge is stored as 0
        E.delta = st ? G.huber_stereo : G.huber_mono;
        if (!st) { for (int c = 0; c < 6; c++) Jc[12 + c] = 0; for (int c = 0; c < 3; c++) Jp[12 + c] = 0; } // a zero third row adds exact zeros
        dyn_add_edge<2, 3>(G, E);
    } else if (cls == 1) { // EdgeDynamicPointCuboidCamera::linearizeOplus g2o_Object.cpp:167-233
        if (dyn_lvl(G.d_lvl, o)) return;
        const int ci = G.d_cam[o], oi = G.d_obj[o], li = G.d_pt[o];
        dyn_lin_init(E, 3, 2); E.eid = e;
        E.off[0] = G.cam_off[ci]; E.dim[0] = 6; E.slot[0] = G.n_obs + 2 * o;
        E.off[1] = G.obj_off[oi]; E.dim[1] = 6; E.slot[1] = G.n_obs + 2 * o + 1;
        E.lm[2] = G.fix_points ? -1 : G.n_pts + li; E.dim[2] = 3;
        const double *op = G.dpts + (long)li * 3;
        const SE3 combinedT = se3_mul(se3_load(G.cam + (long)ci * 7), se3_load(G.obj + (long)oi * 7));
        double cp[3], R[3][3];
        se3_map(combinedT, op, cp);
        qtoR(combinedT.r, R);
        const double fx = G.K[0], fy = G.K[4], x = cp[0], y = cp[1], z = cp[2], z_2 = z * z;
        const double P[2][3] = {{fx / z, 0, -x * fx / z_2}, {0, fy / z, -y * fy / z_2}};
        double *Jc = E.J[0], *Jo = E.J[1], *Jp = E.J[2];
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Jp[r * 6 + c] = ((-P[r][0]) * R[0][c] + (-P[r][1]) * R[1][c]) + (-P[r][2]) * R[2][c];
        dyn_reproj_cam_jac(x, y, z, fx, fy, Jc);
        const double Sk[3][6] = {{-0.0, op[2], -op[1], 1, 0, 0}, {-op[2], -0.0, op[0], 0, 1, 0}, {op[1], -op[0], -0.0, 0, 0, 1}}; // [-skew(p) | I]
        for (int r = 0; r < 2; r++) for (int c = 0; c < 6; c++) Jo[r * 6 + c] = (Jp[r * 6] * Sk[0][c] + Jp[r * 6 + 1] * Sk[1][c]) + Jp[r * 6 + 2] * Sk[2][c];
        const int fl = G.obj_flags[oi];
        if (fl & 1) { Jo[0] = 0; Jo[1] = 0; Jo[6] = 0; Jo[7] = 0; }                          // whether_fixrollpitch
        if (fl & 2) { Jo[0] = 0; Jo[1] = 0; Jo[6] = 0; Jo[7] = 0; Jo[2] = 0; Jo[8] = 0; }    // whether_fixrotation
        for (int k = 0; k < 2; k++) { E.e[k] = G.e_dobs[(long)o * 2 + k]; E.w[k] = G.d_w[o]; }
        E.delta = G.huber_dyn;
        dyn_add_edge<3, 2>(G, E);
    } else if (cls == 2) { // EdgeObjectMotion: numeric, three vertices
        const int a = G.m_from[o], b = G.m_to[o], vi = G.m_vel[o];
        dyn_lin_init(E, 3, 3); E.eid = e;
        E.off[0] = G.obj_off[a]; E.dim[0] = 6; E.off[1] = G.obj_off[b]; E.dim[1] = 6; E.off[2] = G.vel_off[vi]; E.dim[2] = 2;
        const Cuboid ca = dyn_obj(G, a), cb = dyn_obj(G, b);
        const double v[2] = {G.vel[(long)vi * 2], G.vel[(long)vi * 2 + 1]};
        for (int d = 0; d < 6; d++) {
            double add[6] = {0, 0, 0, 0, 0, 0}, e1[3], e2[3];
            add[d] = delta; dyn_err_mot(G, o, dyn_obj_plus(G, a, ca, add).pose, cb.pose, v, e1);
            add[d] = -delta; dyn_err_mot(G, o, dyn_obj_plus(G, a, ca, add).pose, cb.pose, v, e2);
            for (int k = 0; k < 3; k++) E.J[0][k * 6 + d] = scalar * (e1[k] - e2[k]);
            add[d] = delta; dyn_err_mot(G, o, ca.pose, dyn_obj_plus(G, b, cb, add).pose, v, e1);
            add[d] = -delta; dyn_err_mot(G, o, ca.pose, dyn_obj_plus(G, b, cb, add).pose, v, e2);
            for (int k = 0; k < 3; k++) E.J[1][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int d = 0; d < 2; d++) {
            double vp[2] = {v[0], v[1]}, vm[2] = {v[0], v[1]}, e1[3], e2[3];
            vp[d] += delta; vm[d] += -delta;
            dyn_err_mot(G, o, ca.pose, cb.pose, vp, e1);
            dyn_err_mot(G, o, ca.pose, cb.pose, vm, e2);
            for (int k = 0; k < 3; k++) E.J[2][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int k = 0; k < 3; k++) { E.e[k] = G.e_mot[(long)o * 3 + k]; E.w[k] = G.mot_info[k]; }
        dyn_add_edge<3, 3>(G, E);
    } else if (cls == 3) { // EdgeSE3CuboidFixScaleProj: numeric, camera and object
        if (dyn_lvl(G.c_lvl, o)) return;
        const int ci = G.c_cam[o], oi = G.c_obj[o];
        dyn_lin_init(E, 2, 4); E.eid = e;
        E.off[0] = G.cam_off[ci]; E.dim[0] = 6; E.off[1] = G.obj_off[oi]; E.dim[1] = 6;
        const SE3 T = se3_load(G.cam + (long)ci * 7);
        const Cuboid c0 = dyn_obj(G, oi);
        for (int d = 0; d < 6; d++) {
            double add[6] = {0, 0, 0, 0, 0, 0}, e1[4], e2[4];
            if (E.off[0] >= 0) {
                add[d] = delta; dyn_err_cobs(G, o, se3_mul(se3_exp(add), T), c0, e1);
                add[d] = -delta; dyn_err_cobs(G, o, se3_mul(se3_exp(add), T), c0, e2);
                for (int k = 0; k < 4; k++) E.J[0][k * 6 + d] = scalar * (e1[k] - e2[k]);
            }
            add[d] = delta; dyn_err_cobs(G, o, T, dyn_obj_plus(G, oi, c0, add), e1);
            add[d] = -delta; dyn_err_cobs(G, o, T, dyn_obj_plus(G, oi, c0, add), e2);
            for (int k = 0; k < 4; k++) E.J[1][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int k = 0; k < 4; k++) { E.e[k] = G.e_cobs[(long)o * 4 + k]; E.w[k] = G.c_info[(long)o * 4 + k]; }
        E.delta = G.huber_obj;
        dyn_add_edge<2, 4>(G, E);
    } else if (cls == 4) { // EdgePointCuboidOnlyObjectFixScale: numeric unary
        const int oi = G.pc_obj[o];
        dyn_lin_init(E, 1, 3); E.eid = e;
        E.off[0] = G.obj_off[oi]; E.dim[0] = 6;
        const Cuboid c0 = dyn_obj(G, oi);
        for (int d = 0; d < 6; d++) {
            double add[6] = {0, 0, 0, 0, 0, 0}, e1[3], e2[3];
            add[d] = delta; dyn_err_pc(G, o, dyn_obj_plus(G, oi, c0, add), e1);
            add[d] = -delta; dyn_err_pc(G, o, dyn_obj_plus(G, oi, c0, add), e2);
            for (int k = 0; k < 3; k++) E.J[0][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int k = 0; k < 3; k++) { E.e[k] = G.e_pc[(long)o * 3 + k]; E.w[k] = 1.0; }
        dyn_add_edge<1, 3>(G, E);
    } else { // UnaryLocalPoint: numeric unary on the dynamic point
        if (G.fix_points) return;
        dyn_lin_init(E, 1, 3); E.eid = e;
        E.lm[0] = G.n_pts + o; E.dim[0] = 3;
        const double *X = G.dpts + (long)o * 3;
        for (int d = 0; d < 3; d++) {
            double xp[3] = {X[0], X[1], X[2]}, xm[3] = {X[0], X[1], X[2]}, e1[3], e2[3];
            xp[d] += delta; xm[d] += -delta;
            dyn_err_ulp(G, xp, e1);
            dyn_err_ulp(G, xm, e2);
            for (int k = 0; k < 3; k++) E.J[0][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int k = 0; k < 3; k++) { E.e[k] = G.e_ulp[(long)o * 3 + k]; E.w[k] = G.ulp_info; }
        dyn_add_edge<1, 3>(G, E);
    }
}

// ---------------------------------------------------------------------------------------------------- Schur complement, back substitution, update
HD void dyn_inv3(const double *D, double *Di) { // Eigen's fixed-size 3x3 inverse (cofactors)
    const double c00 = D[4] * D[8] - D[5] * D[7], c10 = D[7] * D[2] - D[8] * D[1], c20 = D[1] * D[5] - D[2] * D[4];
    const double det = (c00 * D[0] + c10 * D[3]) + c20 * D[6], inv = 1.0 / det;
    Di[0] = c00 * inv; Di[1] = c10 * inv; Di[2] = c20 * inv;
    Di[3] = (D[5] * D[6] - D[3] * D[8]) * inv; Di[4] = (D[8] * D[0] - D[6] * D[2]) * inv; Di[5] = (D[2] * D[3] - D[0] * D[5]) * inv;
    Di[6] = (D[3] * D[7] - D[4] * D[6]) * inv; Di[7] = (D[6] * D[1] - D[7] * D[0]) * inv; Di[8] = (D[0] * D[4] - D[1] * D[3]) * inv;
}
// The Schur complement S = Hpp + lambda I - sum_l B_l (Hll_l + lambda I)^-1 B_l^T (block_solver.hpp:378-432) in three gather steps, so that no
// two work items write the same word: (1) per landmark the 3x3 inverse, (2) per slot B D^-1 (6x3) and B D^-1 b_l (6), (3) per element of a
// target block the sum over the slot pairs the host listed for it, per right-hand-side row the sum over the vertex' slots.
HD void dyn_dinv_item(const DynG &G, int li, double lambda) {
    double D[9], Di[9];
    for (int k = 0; k < 9; k++) D[k] = G.Hll[(long)li * 9 + k];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    dyn_inv3(D, Di);
    for (int k = 0; k < 9; k++) G.Dinv[(long)li * 9 + k] = Di[k];
}
HD void dyn_bd_item(const DynG &G, int s) {
    if (G.slot_off[s] < 0) return;
    const int li = G.slot_lm[s];
    const double *B = G.Bslot + (long)s * 18, *Di = G.Dinv + (long)li * 9, *b3 = G.bl + (long)li * 3;
    for (int a = 0; a < 6; a++) {
        double r[3];
        for (int c = 0; c < 3; c++) r[c] = (B[a * 3] * Di[c] + B[a * 3 + 1] * Di[3 + c]) + B[a * 3 + 2] * Di[6 + c];
        for (int c = 0; c < 3; c++) G.BD[(long)s * 18 + a * 3 + c] = r[c];
        G.bsub[(long)s * 6 + a] = (r[0] * b3[0] + r[1] * b3[1]) + r[2] * b3[2];
    }
}
// partial sums over every ng-th pair / slot starting at g, so that several lanes can share a long list (the diagonal block of a camera
// collects one pair per point it sees); the caller adds the partials in the order g = 0 .. ng-1 and stores
HD double dyn_schur_block_partial(const DynG &G, int blk, int e, int g, int ng) { // element e = a * 6 + c of target block blk
    const int a = e / 6, c = e % 6;
    double acc = 0;
    for (int q = G.blk_start[blk] + g; q < G.blk_start[blk + 1]; q += ng) {
        const double *BD = G.BD + (long)G.pair_u[q] * 18 + a * 3, *Bt = G.Bslot + (long)G.pair_t[q] * 18 + c * 3;
        acc += (BD[0] * Bt[0] + BD[1] * Bt[1]) + BD[2] * Bt[2];
    }
    return acc;
}
HD void dyn_schur_block_store(const DynG &G, int blk, int e, double acc) {
    const int a = e / 6, c = e % 6, ou = G.blk_ou[blk], ot = G.blk_ot[blk];
    G.S[(long)(ou + a) * G.NP + ot + c] -= acc;
    if (ou != ot) G.S[(long)(ot + c) * G.NP + ou + a] -= acc;
}
HD double dyn_rhs_partial(const DynG &G, int v, int a, int g, int ng) { // row a of pose vertex v (of those that own slots)
    double acc = 0;
    for (int q = G.vtx_start[v] + g; q < G.vtx_start[v + 1]; q += ng) acc += G.bsub[(long)G.vtx_slots[q] * 6 + a];
    return acc;
}
HD void dyn_rhs_store(const DynG &G, int v, int a, double acc) { G.bs[G.vtx_off[v] + a] -= acc; }
HD void dyn_backsub_item(const DynG &G, int li) { // x_l = (Hll + lambda I)^-1 (b_l - B^T x_p), block_solver.hpp:459-485
    double cl[3] = {G.bl[(long)li * 3], G.bl[(long)li * 3 + 1], G.bl[(long)li * 3 + 2]};
    for (int u = G.lm_start[li]; u < G.lm_start[li + 1]; u++) {
        const int su = G.lm_slots[u], ou = G.slot_off[su];
        if (ou < 0) continue;
        const double *Bu = G.Bslot + (long)su * 18;
        for (int c = 0; c < 3; c++) { double s = 0; for (int a = 0; a < 6; a++) s += Bu[a * 3 + c] * G.xp[ou + a]; cl[c] -= s; }
    }
    const double *Di = G.Dinv + (long)li * 9;
    for (int a = 0; a < 3; a++) G.xl[(long)li * 3 + a] = (Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1]) + Di[a * 3 + 2] * cl[2];
}
// oplus of vertex v of [cameras | object poses | velocities | static points | dynamic points] (SparseOptimizer::update)
HD void dyn_update_item(const DynG &G, int v) {
    if (v < G.n_cams) {
        if (G.cam_off[v] < 0) return;
        const SE3 T = se3_mul(se3_exp(G.xp + G.cam_off[v]), se3_load(G.cam + (long)v * 7)); // VertexSE3Expmap::oplusImpl
        se3_store(T, G.cam + (long)v * 7);
        return;
    }
    v -= G.n_cams;
    if (v < G.n_objs) {
        const Cuboid c = dyn_obj_plus(G, v, dyn_obj(G, v), G.xp + G.obj_off[v]);
        se3_store(c.pose, G.obj + (long)v * 7);
        return;
    }
    v -= G.n_objs;
    if (v < G.n_vels) { G.vel[(long)v * 2] += G.xp[G.vel_off[v]]; G.vel[(long)v * 2 + 1] += G.xp[G.vel_off[v] + 1]; return; }
    v -= G.n_vels;
    if (G.fix_points) return;
    if (v < G.n_pts) { for (int k = 0; k < 3; k++) G.pts[(long)v * 3 + k] += G.xl[(long)v * 3 + k]; return; }
    v -= G.n_pts;
    for (int k = 0; k < 3; k++) G.dpts[(long)v * 3 + k] += G.xl[(long)(G.n_pts + v) * 3 + k];
}

} // namespace
