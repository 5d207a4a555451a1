/*
 * oracle/badyn_oracle.cpp -- CPU oracle for the dynamic-object bundle adjustment (Optimizer::LocalBACameraPointObjectsDynamic).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  EDGES PINNED (tests/test_ref_pins.py::test_dynamic_ba_edges_equal_reference, oracle/_ref): computeError of every
 * edge type and the Jacobians of the two three-vertex types equal the reference's own classes (cut out whole, g2o's numeric differentiation under them) bit for
 * bit, and a run driven by the reference's own OptimizationAlgorithmLevenberg::solve + SparseOptimizer::optimize over this file's pieces equals
 * orc_badyn_optimize bit for bit (::test_dynamic_ba_schedule_equals_reference); the graph construction (Optimizer.cc:1537-2350) is restated in oracle/local_ba_dynamic.py and held, with whole runs of orc_badyn_optimize under it, to the reference's own function text running on the reference's g2o (tests/test_ref_graph_pins.py).  Restated from /root/reference/orb_object_slam/src/Optimizer.cc:1537-2573
 * (the graph), orb_object_slam/{include/g2o_Object.h, src/g2o_Object.cpp} (VertexCuboidFixScale :88-116, VelocityPlanarVelocity
 * g2o_Object.h:288-308, EdgeDynamicPointCuboidCamera :154-233, EdgeObjectMotion :241-272, UnaryLocalPoint :378-398,
 * EdgeSE3CuboidFixScaleProj :118-128, EdgePointCuboidOnlyObjectFixScale :336-354) and the vendored g2o under
 * orb_object_slam/Thirdparty/g2o/g2o: core/base_multi_edge.hpp:36-200 (robustified quadratic form, central differences with delta 1e-9
 * per non-fixed vertex), core/base_binary_edge.hpp, core/base_unary_edge.hpp, core/block_solver.hpp:354-486 (BlockSolverX: lambda on
 * both diagonals, Schur complement over the marginalised points, back substitution), core/optimization_algorithm_levenberg.cpp:61-189,
 * types/se3quat.h:184-207 (toXYZPRYVector), types/types_six_dof_expmap.cpp (reprojection edges).
 * LinearSolverDense is replaced by a plain dense Cholesky: any exact factorisation gives the same step up to round-off.
 * Edge order inside the sums: point observations, dynamic-point observations, motion, camera-object, point-object, local-point.
 */
#include "oracle.h"

#include <cstring>
#include <limits>
#include <vector>

#include "se3_util.h"
#include "ba_iface.h"

namespace {

struct DynState { std::vector<SE3> cams; std::vector<Cuboid> objs; std::vector<double> vels, pts, dpts; };

// one linearised edge: up to three vertices; a vertex is a pose-system entry (off >= 0, dim 6 or 2), a landmark (lm >= 0, dim 3) or fixed
struct Lin {
    int nv = 0, D = 0;
    int off[3] = {-1, -1, -1}, dim[3] = {0, 0, 0}, lm[3] = {-1, -1, -1};
    double e[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0}, J[3][24], delta = 0;
    bool fixed(int v) const { return off[v] < 0 && lm[v] < 0; }
};

struct DynBA {
    const orc_badyn_problem *p;
    DynState s;
    std::vector<DynState> stack;
    std::vector<int> cam_off, obj_off, vel_off;
    int NP = 0, L = 0;
    std::vector<double> e_obs, e_dobs, e_mot, e_cobs, e_pc, e_ulp;
    std::vector<double> Hpp, bp, Hll, bl, xp, xl;
    struct PL { int off; double B[18]; }; // rows = the 6 scalars of a pose vertex, columns = the landmark
    std::vector<std::vector<PL>> Hpl;

    explicit DynBA(const orc_badyn_problem *pp) : p(pp) {
        s.cams.resize(p->n_cams); s.objs.resize(p->n_objs);
        for (int i = 0; i < p->n_cams; i++) s.cams[i] = se3_from7(p->cam_pose + (size_t)i * 7);
        for (int i = 0; i < p->n_objs; i++) { s.objs[i].pose = se3_from7(p->obj_pose + (size_t)i * 7); for (int k = 0; k < 3; k++) s.objs[i].scale[k] = p->obj_scale[i * 3 + k]; }
        s.vels.assign(p->vel, p->vel + (size_t)p->n_vels * 2);
        s.pts.assign(p->points, p->points + (size_t)p->n_points * 3);
        s.dpts.assign(p->dpoints, p->dpoints + (size_t)p->n_dpoints * 3);
        cam_off.assign(p->n_cams, -1); obj_off.assign(p->n_objs, -1); vel_off.assign(p->n_vels, -1);
        for (int i = 0; i < p->n_cams; i++) if (!p->cam_fixed[i]) { cam_off[i] = NP; NP += 6; }
        for (int i = 0; i < p->n_objs; i++) { obj_off[i] = NP; NP += 6; }
        for (int i = 0; i < p->n_vels; i++) { vel_off[i] = NP; NP += 2; }
        L = p->fix_points ? 0 : p->n_points + p->n_dpoints;
        e_obs.assign((size_t)p->n_obs * 3, 0.0); e_dobs.assign((size_t)p->n_dobs * 2, 0.0); e_mot.assign((size_t)p->n_mot * 3, 0.0);
        e_cobs.assign((size_t)p->n_cobs * 4, 0.0); e_pc.assign((size_t)p->n_pc * 3, 0.0); e_ulp.assign((size_t)p->n_dpoints * 3, 0.0);
    }
    int lm_static(int i) const { return p->fix_points ? -1 : i; }
    int lm_dynamic(int i) const { return p->fix_points ? -1 : p->n_points + i; }
    static bool lvl(const uint8_t *a, int o) { return a && a[o]; }
    bool stereo(int o) const { return p->obs_ur && p->obs_ur[o] >= 0; }

    // ------------------------------------------------------------------------------------------------ computeError
    void err_obs(int o, const SE3 &T, const double *X, double *e) const { // EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ::computeError
        double pc[3];
        se3_map(T, X, pc);
        if (stereo(o)) { // cam_project types_six_dof_expmap.cpp:182-189: invz and bf are floats there
            const float invz = (float)(1.0 / pc[2]);
            const double u = pc[0] * invz * p->fx + p->cx;
            e[0] = p->obs_uv[o * 2] - u;
            e[1] = p->obs_uv[o * 2 + 1] - (pc[1] * invz * p->fy + p->cy);
            e[2] = p->obs_ur[o] - (u - (double)((float)p->bf * invz)); // `bf*invz` with bf a const float& parameter: a float product
            return;
        }
        e[0] = p->obs_uv[o * 2] - (pc[0] / pc[2] * p->fx + p->cx);
        e[1] = p->obs_uv[o * 2 + 1] - (pc[1] / pc[2] * p->fy + p->cy);
        e[2] = 0.0;
    }
    void err_dobs(int o, const SE3 &T, const Cuboid &c, const double *X, double *e) const { // EdgeDynamicPointCuboidCamera::computeError :154-165
        double pw[3], lp[3];
        se3_map(c.pose, X, pw);
        se3_map(T, pw, lp);
        const double *K = p->K;
        e[0] = p->dobs_uv[o * 2] - (K[2] + K[0] * lp[0] / lp[2]);
        e[1] = p->dobs_uv[o * 2 + 1] - (K[5] + K[4] * lp[1] / lp[2]);
    }
    static double yaw_of(const Quat &q) { return std::atan2(2 * (q.w * q.z + q.x * q.y), 1 - 2 * (q.y * q.y + q.z * q.z)); } // se3quat.h:184-194
    void err_mot(int o, const Cuboid &from, const Cuboid &to, const double *v, double *e) const { // EdgeObjectMotion::computeError :241-272
        const double yaw_from = yaw_of(from.pose.r), yaw_to = yaw_of(to.pose.r), dt = p->mot_dt[o];
        const double vehicle_length = 2.71;
        const double k1 = v[0] * dt - vehicle_length * 0.5;
        const double tb[3] = {from.pose.t[0] + k1 * std::cos(yaw_from), from.pose.t[1] + k1 * std::sin(yaw_from), from.pose.t[2] + k1 * 0.0};
        const double yaw_pred = yaw_from + std::tan(v[1]) * dt / vehicle_length * v[0];
        const double k2 = vehicle_length * 0.5;
        const double tp[2] = {tb[0] + k2 * std::cos(yaw_pred), tb[1] + k2 * std::sin(yaw_pred)};
        e[0] = to.pose.t[0] - tp[0]; e[1] = to.pose.t[1] - tp[1]; e[2] = yaw_to - yaw_pred;
        if (e[2] > 2.0 * M_PI) e[2] -= 2.0 * M_PI;
        if (e[2] < -2.0 * M_PI) e[2] += 2.0 * M_PI;
    }
    void err_cobs(int o, const SE3 &T, const Cuboid &c, double *e) const { // EdgeSE3CuboidFixScaleProj::computeError :118-128
        double bb[4];
        project_bbox(c, T, p->K, bb);
        for (int k = 0; k < 4; k++) e[k] = bb[k] - p->cobs_bbox[o * 4 + k];
    }
    void err_pc(int o, const Cuboid &c, double *e) const { // EdgePointCuboidOnlyObjectFixScale::computeError :336-354 + point_boundary_error :280-298
        double acc[3] = {0, 0, 0};
        const int b0 = p->pc_offsets[o], b1 = p->pc_offsets[o + 1];
        const SE3 inv = se3_inv(c.pose);
        for (int i = b0; i < b1; i++) {
            double lp[3];
            se3_map(inv, p->pc_points + (size_t)i * 3, lp);
            for (int k = 0; k < 3; k++) {
                const double a = std::fabs(lp[k]) * 1.0;
                double er;
                if (a < c.scale[k]) er = 0;
                else if (a < (p->pc_ratio + 1) * c.scale[k]) er = a - c.scale[k];
                else er = p->pc_ratio * c.scale[k];
                acc[k] += std::fabs(er);
            }
        }
        if (b1 > b0) for (int k = 0; k < 3; k++) acc[k] = acc[k] / (double)(b1 - b0);
        for (int k = 0; k < 3; k++) e[k] = 1.0 * (acc[k] / c.scale[k]);
    }
    void err_ulp(const double *X, double *e) const { // UnaryLocalPoint::computeError :378-398
        for (int k = 0; k < 3; k++) {
            const double a = std::fabs(X[k]), sc = p->ulp_scale[k];
            double er;
            if (a < sc) er = 0;
            else if (a < (p->ulp_ratio + 1) * sc) er = a - sc;
            else er = p->ulp_ratio * sc;
            e[k] = er / sc;
        }
    }
    void compute_errors() {
        for (int o = 0; o < p->n_obs; o++) err_obs(o, s.cams[p->obs_cam[o]], &s.pts[(size_t)p->obs_point[o] * 3], &e_obs[(size_t)o * 3]);
        for (int o = 0; o < p->n_dobs; o++) err_dobs(o, s.cams[p->dobs_cam[o]], s.objs[p->dobs_obj[o]], &s.dpts[(size_t)p->dobs_point[o] * 3], &e_dobs[(size_t)o * 2]);
        for (int o = 0; o < p->n_mot; o++) err_mot(o, s.objs[p->mot_from[o]], s.objs[p->mot_to[o]], &s.vels[(size_t)p->mot_vel[o] * 2], &e_mot[(size_t)o * 3]);
        for (int o = 0; o < p->n_cobs; o++) err_cobs(o, s.cams[p->cobs_cam[o]], s.objs[p->cobs_obj[o]], &e_cobs[(size_t)o * 4]);
        for (int o = 0; o < p->n_pc; o++) err_pc(o, s.objs[p->pc_obj[o]], &e_pc[(size_t)o * 3]);
        for (int i = 0; i < p->n_dpoints; i++) err_ulp(&s.dpts[(size_t)i * 3], &e_ulp[(size_t)i * 3]);
    }
    static double chi2_of(const double *e, const double *w, int D) { double c = 0; for (int k = 0; k < D; k++) c += e[k] * w[k] * e[k]; return c; }
    double robust_chi2() const { // SparseOptimizer::activeRobustChi2 sparse_optimizer.cpp:100-114
        double chi = 0, rho[3];
        auto add = [&](double c, double delta) { if (delta > 0) { huber_rho(c, delta, rho); chi += rho[0]; } else chi += c; };
        for (int o = 0; o < p->n_obs; o++) if (!lvl(p->obs_level, o) && !(p->fix_points && cam_off[p->obs_cam[o]] < 0)) { // allVerticesFixed edges are not active
            const double w = p->obs_inv_sigma2[o], ww[3] = {w, w, w};
            add(chi2_of(&e_obs[(size_t)o * 3], ww, stereo(o) ? 3 : 2), stereo(o) ? p->huber_stereo : p->huber_mono);
        }
        for (int o = 0; o < p->n_dobs; o++) if (!lvl(p->dobs_level, o)) { const double w = p->dobs_inv_sigma2[o], ww[2] = {w, w}; add(chi2_of(&e_dobs[(size_t)o * 2], ww, 2), p->huber_dyn); }
        for (int o = 0; o < p->n_mot; o++) add(chi2_of(&e_mot[(size_t)o * 3], p->mot_info, 3), 0);
        for (int o = 0; o < p->n_cobs; o++) if (!lvl(p->cobs_level, o)) add(chi2_of(&e_cobs[(size_t)o * 4], p->cobs_info + (size_t)o * 4, 4), p->huber_obj);
        const double one[3] = {1, 1, 1}, ul[3] = {p->ulp_info, p->ulp_info, p->ulp_info};
        for (int o = 0; o < p->n_pc; o++) add(chi2_of(&e_pc[(size_t)o * 3], one, 3), 0);
        if (!p->fix_points) for (int i = 0; i < p->n_dpoints; i++) add(chi2_of(&e_ulp[(size_t)i * 3], ul, 3), 0); // a unary edge on a fixed vertex is not active
        return chi;
    }

    // ------------------------------------------------------------------------------------------------ quadratic form
    void add_edge(const Lin &E) { // BaseMultiEdge::constructQuadraticForm base_multi_edge.hpp:36-48 + computeQuadraticForm (J^T Omega J per vertex pair)
        double rw = 1.0, rho[3];
        if (E.delta > 0) { huber_rho(chi2_of(E.e, E.w, E.D), E.delta, rho); rw = rho[1]; }
        double omr[4], W[4];
        for (int k = 0; k < E.D; k++) { omr[k] = -E.w[k] * E.e[k] * rw; W[k] = rw * E.w[k]; }
        for (int i = 0; i < E.nv; i++) {
            if (E.fixed(i)) continue;
            for (int a = 0; a < E.dim[i]; a++) {
                double g = 0;
                for (int k = 0; k < E.D; k++) g += E.J[i][k * 6 + a] * omr[k];
                if (E.lm[i] >= 0) bl[(size_t)E.lm[i] * 3 + a] += g; else bp[E.off[i] + a] += g;
            }
            for (int j = i; j < E.nv; j++) {
                if (E.fixed(j)) continue;
                double h[36];
                for (int a = 0; a < E.dim[i]; a++) for (int c = 0; c < E.dim[j]; c++) {
                    double sacc = 0;
                    for (int k = 0; k < E.D; k++) sacc += (E.J[i][k * 6 + a] * W[k]) * E.J[j][k * 6 + c];
                    h[a * 6 + c] = sacc;
                }
                if (E.lm[i] >= 0 && E.lm[j] >= 0) { // an edge has one landmark at most: i == j
                    for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) Hll[(size_t)E.lm[i] * 9 + a * 3 + c] += h[a * 6 + c];
                } else if (E.lm[i] < 0 && E.lm[j] < 0) {
                    for (int a = 0; a < E.dim[i]; a++) for (int c = 0; c < E.dim[j]; c++) {
                        Hpp[(size_t)(E.off[i] + a) * NP + E.off[j] + c] += h[a * 6 + c];
                        if (i != j) Hpp[(size_t)(E.off[j] + c) * NP + E.off[i] + a] += h[a * 6 + c];
                    }
                } else { // the landmark is the last vertex of every Lin built below
                    PL e; e.off = E.off[i];
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) e.B[a * 3 + c] = h[a * 6 + c];
                    Hpl[E.lm[j]].push_back(e);
                }
            }
        }
    }
    template <class F> void numeric(Lin &E, int v, int D, F eval) { // central differences, delta 1e-9 (base_multi_edge.hpp:72-120, base_binary_edge.hpp, base_unary_edge.hpp)
        const double delta = 1e-9, scalar = 1.0 / (2 * delta);
        for (int d = 0; d < E.dim[v]; d++) {
            double add[6] = {0, 0, 0, 0, 0, 0}, e1[4], e2[4];
            add[d] = delta; eval(add, e1);
            add[d] = -delta; eval(add, e2);
            for (int k = 0; k < D; k++) E.J[v][k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
    }
    Cuboid obj_plus(int i, const double *add) const { return cuboid_oplus(s.objs[i], add, p->obj_flags[i], p->obj_scale + (size_t)i * 3); }
    SE3 cam_plus(int i, const double *add) const { return se3_mul(se3_exp(add), s.cams[i]); }

    Lin lin_dobs(int o) { // EdgeDynamicPointCuboidCamera::linearizeOplus :167-233
            const int ci = p->dobs_cam[o], oi = p->dobs_obj[o], li = p->dobs_point[o];
            Lin E; E.nv = 3; E.D = 2;
            E.off[0] = cam_off[ci]; E.dim[0] = 6; E.off[1] = obj_off[oi]; E.dim[1] = 6; E.lm[2] = lm_dynamic(li); E.dim[2] = 3;
            const double *objectpt = &s.dpts[(size_t)li * 3];
            const SE3 combinedT = se3_mul(s.cams[ci], s.objs[oi].pose);
            double cp[3];
            se3_map(combinedT, objectpt, cp);
            const double fx = p->K[0], fy = p->K[4], x = cp[0], y = cp[1], z = cp[2], z_2 = z * z;
            const double P[2][3] = {{fx / z, 0, -x * fx / z_2}, {0, fy / z, -y * fy / z_2}};
            M3 R; qtoR(combinedT.r, R);
            double (*Jc)[6] = reinterpret_cast<double (*)[6]>(E.J[0]), (*Jo)[6] = reinterpret_cast<double (*)[6]>(E.J[1]), (*Jp)[6] = reinterpret_cast<double (*)[6]>(E.J[2]);
            for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Jp[r][c] = ((-P[r][0]) * R[0][c] + (-P[r][1]) * R[1][c]) + (-P[r][2]) * R[2][c];
            Jc[0][0] = x * y / z_2 * fx; Jc[0][1] = -(1 + (x * x / z_2)) * fx; Jc[0][2] = y / z * fx; Jc[0][3] = -1. / z * fx; Jc[0][4] = 0; Jc[0][5] = x / z_2 * fx;
            Jc[1][0] = (1 + y * y / z_2) * fy; Jc[1][1] = -x * y / z_2 * fy; Jc[1][2] = -x / z * fy; Jc[1][3] = 0; Jc[1][4] = -1. / z * fy; Jc[1][5] = y / z_2 * fy;
            const double S[3][6] = {{-0.0, objectpt[2], -objectpt[1], 1, 0, 0}, {-objectpt[2], -0.0, objectpt[0], 0, 1, 0}, {objectpt[1], -objectpt[0], -0.0, 0, 0, 1}}; // [-skew(p) | I]
            for (int r = 0; r < 2; r++) for (int c = 0; c < 6; c++) Jo[r][c] = (Jp[r][0] * S[0][c] + Jp[r][1] * S[1][c]) + Jp[r][2] * S[2][c];
            const int fl = p->obj_flags[oi];
            if (fl & 1) { Jo[0][0] = 0; Jo[0][1] = 0; Jo[1][0] = 0; Jo[1][1] = 0; }
            if (fl & 2) { Jo[0][0] = 0; Jo[0][1] = 0; Jo[1][0] = 0; Jo[1][1] = 0; Jo[0][2] = 0; Jo[1][2] = 0; }
            for (int k = 0; k < 2; k++) { E.e[k] = e_dobs[(size_t)o * 2 + k]; E.w[k] = p->dobs_inv_sigma2[o]; }
            E.delta = p->huber_dyn;
            return E;
    }
    Lin lin_mot(int o) { // EdgeObjectMotion: numeric Jacobians for the three vertices (base_multi_edge.hpp:62-133)
            const int a = p->mot_from[o], b2 = p->mot_to[o], vi = p->mot_vel[o];
            Lin E; E.nv = 3; E.D = 3;
            E.off[0] = obj_off[a]; E.dim[0] = 6; E.off[1] = obj_off[b2]; E.dim[1] = 6; E.off[2] = vel_off[vi]; E.dim[2] = 2;
            const double *v = &s.vels[(size_t)vi * 2];
            numeric(E, 0, 3, [&](const double *add, double *e) { err_mot(o, obj_plus(a, add), s.objs[b2], v, e); });
            numeric(E, 1, 3, [&](const double *add, double *e) { err_mot(o, s.objs[a], obj_plus(b2, add), v, e); });
            numeric(E, 2, 3, [&](const double *add, double *e) { const double v2[2] = {v[0] + add[0], v[1] + add[1]}; err_mot(o, s.objs[a], s.objs[b2], v2, e); });
            for (int k = 0; k < 3; k++) { E.e[k] = e_mot[(size_t)o * 3 + k]; E.w[k] = p->mot_info[k]; }
            return E;
    }

    void build_system() { // BlockSolver::buildSystem block_solver.hpp:502-560
        Hpp.assign((size_t)NP * NP, 0.0); bp.assign(NP, 0.0); Hll.assign((size_t)L * 9, 0.0); bl.assign((size_t)L * 3, 0.0);
        Hpl.assign(L, std::vector<PL>());
        for (int o = 0; o < p->n_obs; o++) { // (camera, point)
            if (lvl(p->obs_level, o)) continue;
            const int ci = p->obs_cam[o], li = p->obs_point[o];
            Lin E; E.nv = 2; E.D = stereo(o) ? 3 : 2;
            E.off[0] = cam_off[ci]; E.dim[0] = 6; E.lm[1] = lm_static(li); E.dim[1] = 3;
            const SE3 &T = s.cams[ci];
            double pc[3];
            se3_map(T, &s.pts[(size_t)li * 3], pc);
            const double X = pc[0], Y = pc[1], Z = pc[2], Z2 = Z * Z, fx = p->fx, fy = p->fy;
            M3 R; qtoR(T.r, R);
            double (*Jc)[6] = reinterpret_cast<double (*)[6]>(E.J[0]), (*Jp)[6] = reinterpret_cast<double (*)[6]>(E.J[1]);
            if (!stereo(o)) { // EdgeSE3ProjectXYZ::linearizeOplus types_six_dof_expmap.cpp:135-171
                const double tmp[2][3] = {{fx, 0, -X / Z * fx}, {0, fy, -Y / Z * fy}};
                for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Jp[r][c] = ((-1. / Z * tmp[r][0]) * R[0][c] + (-1. / Z * tmp[r][1]) * R[1][c]) + (-1. / Z * tmp[r][2]) * R[2][c];
            } else { // EdgeStereoSE3ProjectXYZ::linearizeOplus :220-266
                for (int c = 0; c < 3; c++) {
                    Jp[0][c] = -fx * R[0][c] / Z + fx * X * R[2][c] / Z2;
                    Jp[1][c] = -fy * R[1][c] / Z + fy * Y * R[2][c] / Z2;
                    Jp[2][c] = Jp[0][c] - p->bf * R[2][c] / Z2;
                }
            }
            Jc[0][0] = X * Y / Z2 * fx; Jc[0][1] = -(1 + (X * X / Z2)) * fx; Jc[0][2] = Y / Z * fx; Jc[0][3] = -1. / Z * fx; Jc[0][4] = 0; Jc[0][5] = X / Z2 * fx;
            Jc[1][0] = (1 + Y * Y / Z2) * fy; Jc[1][1] = -X * Y / Z2 * fy; Jc[1][2] = -X / Z * fy; Jc[1][3] = 0; Jc[1][4] = -1. / Z * fy; Jc[1][5] = Y / Z2 * fy;
            if (stereo(o)) { Jc[2][0] = Jc[0][0] - p->bf * Y / Z2; Jc[2][1] = Jc[0][1] + p->bf * X / Z2; Jc[2][2] = Jc[0][2]; Jc[2][3] = Jc[0][3]; Jc[2][4] = 0; Jc[2][5] = Jc[0][5] - p->bf / Z2; }
            for (int k = 0; k < E.D; k++) { E.e[k] = e_obs[(size_t)o * 3 + k]; E.w[k] = p->obs_inv_sigma2[o]; }
            E.delta = stereo(o) ? p->huber_stereo : p->huber_mono;
            add_edge(E);
        }
        for (int o = 0; o < p->n_dobs; o++) if (!lvl(p->dobs_level, o)) add_edge(lin_dobs(o));
        for (int o = 0; o < p->n_mot; o++) add_edge(lin_mot(o));
        for (int o = 0; o < p->n_cobs; o++) { // EdgeSE3CuboidFixScaleProj: numeric (base_binary_edge.hpp:216-320)
            if (lvl(p->cobs_level, o)) continue;
            const int ci = p->cobs_cam[o], oi = p->cobs_obj[o];
            Lin E; E.nv = 2; E.D = 4;
            E.off[0] = cam_off[ci]; E.dim[0] = 6; E.off[1] = obj_off[oi]; E.dim[1] = 6;
            if (E.off[0] >= 0) numeric(E, 0, 4, [&](const double *add, double *e) { err_cobs(o, cam_plus(ci, add), s.objs[oi], e); });
            numeric(E, 1, 4, [&](const double *add, double *e) { err_cobs(o, s.cams[ci], obj_plus(oi, add), e); });
            for (int k = 0; k < 4; k++) { E.e[k] = e_cobs[(size_t)o * 4 + k]; E.w[k] = p->cobs_info[(size_t)o * 4 + k]; }
            E.delta = p->huber_obj;
            add_edge(E);
        }
        for (int o = 0; o < p->n_pc; o++) { // EdgePointCuboidOnlyObjectFixScale: numeric unary
            const int oi = p->pc_obj[o];
            Lin E; E.nv = 1; E.D = 3; E.off[0] = obj_off[oi]; E.dim[0] = 6;
            numeric(E, 0, 3, [&](const double *add, double *e) { err_pc(o, obj_plus(oi, add), e); });
            for (int k = 0; k < 3; k++) { E.e[k] = e_pc[(size_t)o * 3 + k]; E.w[k] = 1.0; }
            add_edge(E);
        }
        if (!p->fix_points) for (int i = 0; i < p->n_dpoints; i++) { // UnaryLocalPoint: numeric unary on the dynamic point
            Lin E; E.nv = 1; E.D = 3; E.lm[0] = lm_dynamic(i); E.dim[0] = 3;
            const double *X = &s.dpts[(size_t)i * 3];
            numeric(E, 0, 3, [&](const double *add, double *e) { const double X2[3] = {X[0] + add[0], X[1] + add[1], X[2] + add[2]}; err_ulp(X2, e); });
            for (int k = 0; k < 3; k++) { E.e[k] = e_ulp[(size_t)i * 3 + k]; E.w[k] = p->ulp_info; }
            add_edge(E);
        }
    }

    // ------------------------------------------------------------------------------------------------ solve
    static void inv3(const double *D, double *Di) { // Eigen fixed 3x3 inverse (cofactors)
        auto cf = [&](int i, int j) { int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return D[i1 * 3 + j1] * D[i2 * 3 + j2] - D[i1 * 3 + j2] * D[i2 * 3 + j1]; };
        const double c00 = cf(0, 0), c10 = cf(1, 0), c20 = cf(2, 0);
        const double det = (c00 * D[0] + c10 * D[3]) + c20 * D[6], inv = 1.0 / det;
        Di[0] = c00 * inv; Di[1] = c10 * inv; Di[2] = c20 * inv;
        Di[3] = cf(0, 1) * inv; Di[4] = cf(1, 1) * inv; Di[5] = cf(2, 1) * inv;
        Di[6] = cf(0, 2) * inv; Di[7] = cf(1, 2) * inv; Di[8] = cf(2, 2) * inv;
    }
    void reduced(double lambda, std::vector<double> &S, std::vector<double> &bs, std::vector<double> *Dinv_out) const { // block_solver.hpp:378-432
        S = Hpp; bs = bp;
        for (int i = 0; i < NP; i++) S[(size_t)i * NP + i] += lambda;
        if (Dinv_out) Dinv_out->assign((size_t)L * 9, 0.0);
        for (int li = 0; li < L; li++) {
            double D[9], Di[9];
            for (int k = 0; k < 9; k++) D[k] = Hll[(size_t)li * 9 + k];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            inv3(D, Di);
            if (Dinv_out) for (int k = 0; k < 9; k++) (*Dinv_out)[(size_t)li * 9 + k] = Di[k];
            const double *b3 = &bl[(size_t)li * 3];
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = (Di[a * 3] * b3[0] + Di[a * 3 + 1] * b3[1]) + Di[a * 3 + 2] * b3[2];
            const std::vector<PL> &v = Hpl[li];
            for (size_t u = 0; u < v.size(); u++) {
                double BD[18];
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[a * 3 + c] = (v[u].B[a * 3] * Di[c] + v[u].B[a * 3 + 1] * Di[3 + c]) + v[u].B[a * 3 + 2] * Di[6 + c];
                for (int a = 0; a < 6; a++) bs[v[u].off + a] -= (v[u].B[a * 3] * db[0] + v[u].B[a * 3 + 1] * db[1]) + v[u].B[a * 3 + 2] * db[2];
                for (size_t t = 0; t < v.size(); t++)
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                        S[(size_t)(v[u].off + a) * NP + v[t].off + c] -= (BD[a * 3] * v[t].B[c * 3] + BD[a * 3 + 1] * v[t].B[c * 3 + 1]) + BD[a * 3 + 2] * v[t].B[c * 3 + 2];
            }
        }
    }
    static bool dense_chol_solve(std::vector<double> &A, int n, std::vector<double> &x) { // in place: A -> L (lower), x: rhs -> solution
        for (int j = 0; j < n; j++) {
            double d = A[(size_t)j * n + j];
            for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
            if (!(d > 0)) return false;
            d = std::sqrt(d);
            A[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; i++) {
                double v = A[(size_t)i * n + j];
                for (int k = 0; k < j; k++) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] = v / d;
            }
        }
        for (int i = 0; i < n; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= A[(size_t)i * n + k] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
        return true;
    }
    bool solve(double lambda) { // BlockSolver::solve block_solver.hpp:354-486
        std::vector<double> S, Dinv;
        reduced(lambda, S, xp, &Dinv);
        xl.assign((size_t)L * 3, 0.0);
        if (NP > 0 && !dense_chol_solve(S, NP, xp)) { xp.assign(NP, 0.0); return false; }
        for (int li = 0; li < L; li++) {
            double cl[3] = {bl[(size_t)li * 3], bl[(size_t)li * 3 + 1], bl[(size_t)li * 3 + 2]};
            for (const PL &e : Hpl[li]) for (int c = 0; c < 3; c++) { double sacc = 0; for (int a = 0; a < 6; a++) sacc += e.B[a * 3 + c] * xp[e.off + a]; cl[c] -= sacc; }
            const double *Di = &Dinv[(size_t)li * 9];
            for (int a = 0; a < 3; a++) xl[(size_t)li * 3 + a] = (Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1]) + Di[a * 3 + 2] * cl[2];
        }
        return true;
    }
    void update() { // SparseOptimizer::update -> oplus per vertex
        for (int i = 0; i < p->n_cams; i++) if (cam_off[i] >= 0) s.cams[i] = cam_plus(i, &xp[cam_off[i]]);
        for (int i = 0; i < p->n_objs; i++) s.objs[i] = obj_plus(i, &xp[obj_off[i]]);
        for (int i = 0; i < p->n_vels; i++) { s.vels[(size_t)i * 2] += xp[vel_off[i]]; s.vels[(size_t)i * 2 + 1] += xp[vel_off[i] + 1]; }
        if (!p->fix_points) {
            for (int i = 0; i < p->n_points; i++) for (int k = 0; k < 3; k++) s.pts[(size_t)i * 3 + k] += xl[(size_t)i * 3 + k];
            for (int i = 0; i < p->n_dpoints; i++) for (int k = 0; k < 3; k++) s.dpts[(size_t)i * 3 + k] += xl[(size_t)(p->n_points + i) * 3 + k];
        }
    }
    double lambda_init() const { // optimization_algorithm_levenberg.cpp:166-180
        double mx = 0;
        for (int i = 0; i < NP; i++) mx = std::max(std::fabs(Hpp[(size_t)i * NP + i]), mx);
        for (int i = 0; i < L; i++) for (int k = 0; k < 3; k++) mx = std::max(std::fabs(Hll[(size_t)i * 9 + k * 4]), mx);
        return 1e-5 * mx;
    }
};

struct DynIface : OrcBAIface { // the pieces of DynBA one at a time (ba_iface.h)
    DynBA ba; std::vector<double> xc, bc; std::vector<int> blk_off, blk_dim;
    explicit DynIface(const orc_badyn_problem *p) : ba(p) {
        for (int i = 0; i < p->n_cams; i++) if (ba.cam_off[i] >= 0) { blk_off.push_back(ba.cam_off[i]); blk_dim.push_back(6); }
        for (int i = 0; i < p->n_objs; i++) { blk_off.push_back(ba.obj_off[i]); blk_dim.push_back(6); }
        for (int i = 0; i < p->n_vels; i++) { blk_off.push_back(ba.vel_off[i]); blk_dim.push_back(2); }
    }
    void compute_errors() override { ba.compute_errors(); }
    double robust_chi2() override { return ba.robust_chi2(); }
    void build_system() override { ba.build_system(); }
    int n_blocks() override { return (int)blk_off.size() + ba.L; }
    int n_pose_blocks() override { return (int)blk_off.size(); }
    int block_dim(int k) override { return k < (int)blk_off.size() ? blk_dim[k] : 3; }
    double hessian_diag(int k, int j) override {
        if (k < (int)blk_off.size()) { const size_t r = (size_t)blk_off[k] + j; return ba.Hpp[r * ba.NP + r]; }
        return ba.Hll[(size_t)(k - (int)blk_off.size()) * 9 + j * 4];
    }
    bool solve(double lambda) override { return ba.solve(lambda); }
    void update() override { ba.update(); }
    void push() override { ba.stack.push_back(ba.s); }
    void pop() override { ba.s = ba.stack.back(); ba.stack.pop_back(); }
    void discard_top() override { ba.stack.pop_back(); }
    const double *x(long *n) override { xc = ba.xp; xc.insert(xc.end(), ba.xl.begin(), ba.xl.end()); xc.resize((size_t)ba.NP + (size_t)ba.L * 3, 0.0); if (n) *n = (long)xc.size(); return xc.data(); }
    const double *b() override { bc = ba.bp; bc.insert(bc.end(), ba.bl.begin(), ba.bl.end()); return bc.data(); }
    void read(double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints) override {
        const orc_badyn_problem *p = ba.p;
        for (int i = 0; i < p->n_cams; i++) se3_to7(ba.s.cams[i], cam_pose + (size_t)i * 7);
        for (int i = 0; i < p->n_objs; i++) se3_to7(ba.s.objs[i].pose, obj_pose + (size_t)i * 7);
        if (p->n_vels) std::memcpy(vel, ba.s.vels.data(), ba.s.vels.size() * sizeof(double));
        if (p->n_points) std::memcpy(points, ba.s.pts.data(), ba.s.pts.size() * sizeof(double));
        if (p->n_dpoints) std::memcpy(dpoints, ba.s.dpts.data(), ba.s.dpts.size() * sizeof(double));
    }
};

} // namespace

OrcBAIface *orc_badyn_make_iface(const orc_badyn_problem *p) { return new DynIface(p); }

extern "C" {

double orc_badyn_errors(const orc_badyn_problem *p, double *e_obs, double *e_dobs, double *e_mot, double *e_cobs, double *e_pc, double *e_ulp) {
    DynBA ba(p);
    ba.compute_errors();
    auto out = [](double *dst, const std::vector<double> &v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
    out(e_obs, ba.e_obs); out(e_dobs, ba.e_dobs); out(e_mot, ba.e_mot); out(e_cobs, ba.e_cobs); out(e_pc, ba.e_pc); out(e_ulp, ba.e_ulp);
    return ba.robust_chi2();
}

// Test hook (tests/test_ref_pins.py): the Jacobians of the two three-vertex edge types at the problem's estimates, to be held against the reference's own
// EdgeDynamicPointCuboidCamera::linearizeOplus and BaseMultiEdge::linearizeOplus over EdgeObjectMotion (oracle/_ref).  J_dobs: n_dobs x 3 vertices (camera, object,
// point) x 2 rows x 6 (row stride 6, unused columns zero); J_mot: n_mot x 3 vertices (from, to, velocity) x 3 rows x 6.
void orc_badyn_edge_jacobians(const orc_badyn_problem *p, double *J_dobs, double *J_mot) {
    DynBA ba(p);
    ba.compute_errors();
    for (int o = 0; o < p->n_dobs; o++) { const Lin E = ba.lin_dobs(o); for (int v = 0; v < 3; v++) for (int k = 0; k < 12; k++) J_dobs[(size_t)o * 36 + v * 12 + k] = (k % 6) < E.dim[v] ? E.J[v][k] : 0.0; }
    for (int o = 0; o < p->n_mot; o++) { const Lin E = ba.lin_mot(o); for (int v = 0; v < 3; v++) for (int k = 0; k < 18; k++) J_mot[(size_t)o * 54 + v * 18 + k] = (k % 6) < E.dim[v] ? E.J[v][k] : 0.0; }
}
int orc_badyn_reduced_dense(const orc_badyn_problem *p, double lambda, double *H, double *bvec) {
    DynBA ba(p);
    if (!H) return ba.NP;
    ba.compute_errors();
    ba.build_system();
    std::vector<double> S, bs;
    ba.reduced(lambda, S, bs, nullptr);
    if (ba.NP) { std::memcpy(H, S.data(), S.size() * sizeof(double)); std::memcpy(bvec, bs.data(), bs.size() * sizeof(double)); }
    return ba.NP;
}

int orc_badyn_step(const orc_badyn_problem *p, double lambda, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints) {
    DynBA ba(p); // one linear step with the given damping: computeActiveErrors, buildSystem, solve(lambda), update
    ba.compute_errors();
    ba.build_system();
    const bool ok = ba.solve(lambda);
    ba.update();
    for (int i = 0; i < p->n_cams; i++) se3_to7(ba.s.cams[i], cam_pose + (size_t)i * 7);
    for (int i = 0; i < p->n_objs; i++) se3_to7(ba.s.objs[i].pose, obj_pose + (size_t)i * 7);
    if (p->n_vels) std::memcpy(vel, ba.s.vels.data(), ba.s.vels.size() * sizeof(double));
    if (p->n_points) std::memcpy(points, ba.s.pts.data(), ba.s.pts.size() * sizeof(double));
    if (p->n_dpoints) std::memcpy(dpoints, ba.s.dpts.data(), ba.s.dpts.size() * sizeof(double));
    return ok ? 0 : 1;
}

int orc_badyn_optimize(const orc_badyn_problem *p, int iterations, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints, orc_ba_stats *st) {
    DynBA ba(p);
    orc_ba_stats S;
    std::memset(&S, 0, sizeof(S));
    double lambda = 0, ni = 2;
    int nBad = 0;
    for (int it = 0; it < iterations; it++) { // OptimizationAlgorithmLevenberg::solve :61-164
        ba.compute_errors();
        double currentChi = ba.robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) S.chi2_init = currentChi;
        ba.build_system();
        if (it == 0) { lambda = ba.lambda_init(); ni = 2; nBad = 0; }
        double rho = 0;
        int qmax = 0;
        do {
            ba.stack.push_back(ba.s);
            const bool ok2 = ba.solve(lambda);
            ba.update();
            ba.compute_errors();
            tempChi = ba.robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = (currentChi - tempChi);
            double scale = 0;
            for (size_t j = 0; j < ba.xp.size(); j++) scale += ba.xp[j] * (lambda * ba.xp[j] + ba.bp[j]);
            for (size_t j = 0; j < ba.xl.size(); j++) scale += ba.xl[j] * (lambda * ba.xl[j] + ba.bl[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = (std::min)(alpha, 2. / 3.);
                const double scaleFactor = (std::max)(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
                ba.stack.pop_back();
            } else {
                lambda *= ni;
                ni *= 2;
                ba.s = ba.stack.back(); ba.stack.pop_back();
            }
            qmax++;
            S.lm_trials++;
        } while (rho < 0 && qmax < 10);
        S.iterations = it + 1;
        if (it < 64) S.chi2_trace[it] = currentChi;
        S.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    S.lambda_final = lambda;
    if (st) *st = S;
    for (int i = 0; i < p->n_cams; i++) se3_to7(ba.s.cams[i], cam_pose + (size_t)i * 7);
    for (int i = 0; i < p->n_objs; i++) se3_to7(ba.s.objs[i].pose, obj_pose + (size_t)i * 7);
    if (p->n_vels) std::memcpy(vel, ba.s.vels.data(), ba.s.vels.size() * sizeof(double));
    if (p->n_points) std::memcpy(points, ba.s.pts.data(), ba.s.pts.size() * sizeof(double));
    if (p->n_dpoints) std::memcpy(dpoints, ba.s.dpts.data(), ba.s.dpts.size() * sizeof(double));
    return 0;
}

} // extern "C"
