/*
 * oracle/ba_oracle.cpp -- CPU oracle for the g2o object bundle adjustment (BlockSolver_6_3 + Levenberg + Schur).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PINNED (tests/test_ref_pins.py, tests/test_ref_graph_pins.py, oracle/_ref): the SE3 / cuboid vertex and edge math
 * equals the reference's se3quat.h / g2o_Object functions bit for bit; a run driven by the reference's own OptimizationAlgorithmLevenberg::solve +
 * SparseOptimizer::optimize + RobustKernelHuber over this file's pieces equals orc_ba_optimize bit for bit; build_system and PoseOpt::eval / build equal
 * g2o's own linearizeOplus / constructQuadraticForm under the reference's vertex / edge classes block for block, bit for bit.  The Schur solve, the landmark
 * back-substitution and whole runs of orc_ba_optimize / orc_pose_optimization are held (to round-off: another elimination order) to the reference's own
 * Optimizer::BundleAdjustment / LocalBACameraPointObjects / PoseOptimization running on the reference's vendored g2o compiled whole (oracle/_ref/libref_graph.so);
 * Eigen's sparse Cholesky itself is absent from the image and stays replaced.  Restated from the vendored g2o under
 * /root/reference/orb_object_slam/Thirdparty/g2o/g2o (core/optimization_algorithm_levenberg.cpp:61-189,
 * core/block_solver.hpp:354-604, core/base_binary_edge.hpp:55-320, core/base_unary_edge.hpp:43-123,
 * core/sparse_optimizer.cpp:61-114, core/robust_kernel_impl.cpp:78-91, types/se3quat.h, types/types_six_dof_expmap.*)
 * and the CubeSLAM types of orb_object_slam/{include/g2o_Object.h, src/g2o_Object.cpp}.
 * The linear solve replaces Eigen::SimplicialLDLT (un-vendored) by an exact block-envelope Cholesky after a reverse
 * Cuthill-McKee ordering: any exact factorisation gives the same step up to round-off.
 * Edges are processed in the order: point observations, camera-cuboid, point-cuboid (the reference's insertion order).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <vector>

#include "se3_util.h"
#include "ba_iface.h"

namespace {

struct State {
    std::vector<SE3> cams; std::vector<double> pts; std::vector<Cuboid> cubs;
};

struct BA {
    const orc_ba_problem *p;
    State s;
    std::vector<State> stack;
    int P = 0, L = 0;                  // non-fixed pose blocks (cams then cuboids), landmarks
    std::vector<int> cam_idx, cub_idx; // hessian block index or -1
    // errors
    std::vector<double> e_obs, e_cobs, e_pc;
    // linear system
    std::vector<double> Hpp_diag;      // P x 36
    std::map<std::pair<int, int>, std::vector<double>> Hpp_off; // (i<j) -> 36, row-major block of rows i cols j
    std::vector<double> Hll;           // L x 9
    std::vector<double> Hpl;           // n_obs x 18 (6x3 row-major), valid if cam not fixed
    std::vector<double> b;             // 6P + 3L
    std::vector<double> x;             // 6P + 3L
    std::vector<std::vector<int>> lm_obs; // per landmark: observation ids

    explicit BA(const orc_ba_problem *pp) : p(pp) {
        s.cams.resize(p->n_cams); s.cubs.resize(p->n_cuboids); s.pts.assign(p->points, p->points + (size_t)p->n_points * 3);
        for (int i = 0; i < p->n_cams; i++) s.cams[i] = se3_from7(p->cam_pose + (size_t)i * 7);
        for (int i = 0; i < p->n_cuboids; i++) { s.cubs[i].pose = se3_from7(p->cuboid_pose + (size_t)i * 7); for (int k = 0; k < 3; k++) s.cubs[i].scale[k] = p->cuboid_scale[i * 3 + k]; }
        cam_idx.assign(p->n_cams, -1); cub_idx.assign(p->n_cuboids, -1);
        for (int i = 0; i < p->n_cams; i++) if (!p->cam_fixed[i]) cam_idx[i] = P++;
        for (int i = 0; i < p->n_cuboids; i++) cub_idx[i] = P++;
        L = p->n_points;
        lm_obs.resize(L);
        for (int o = 0; o < p->n_obs; o++) lm_obs[p->obs_point[o]].push_back(o);
        e_obs.resize((size_t)p->n_obs * 3); e_cobs.resize((size_t)p->n_cobs * 4); e_pc.resize((size_t)p->n_pc * 3);
    }

    void err_obs(int o, const SE3 &T, const double *X, double *e) const { // EdgeSE3ProjectXYZ::computeError
        double pc[3];
        se3_map(T, X, pc);
        if (stereo(o)) { // EdgeStereoSE3ProjectXYZ::cam_project (types_six_dof_expmap.cpp:182-189): invz and bf are rounded to float there
            const float invz = (float)(1.0 / pc[2]);
            const double u = pc[0] * invz * p->fx + p->cx;
            e[0] = p->obs_uv[o * 2] - u;
            e[1] = p->obs_uv[o * 2 + 1] - (pc[1] * invz * p->fy + p->cy);
            e[2] = p->obs_ur[o] - (u - (double)((float)p->bf * invz)); // `bf*invz` with bf a const float& parameter: a float product
            return;
        }
        const double px = pc[0] / pc[2], py = pc[1] / pc[2];
        e[0] = p->obs_uv[o * 2] - (px * p->fx + p->cx);
        e[1] = p->obs_uv[o * 2 + 1] - (py * p->fy + p->cy);
        e[2] = 0.0;
    }
    bool stereo(int o) const { return p->obs_ur && p->obs_ur[o] >= 0; }
    double obs_delta(int o) const { return stereo(o) ? p->huber_stereo : p->huber_mono; }
    void err_cobs(int o, const SE3 &T, const Cuboid &c, double *e) const { // EdgeSE3CuboidFixScaleProj::computeError
        double bb[4];
        project_bbox(c, T, p->K, bb);
        for (int k = 0; k < 4; k++) e[k] = bb[k] - p->cobs_bbox[o * 4 + k];
    }
    void err_pc(int o, const Cuboid &c, double *e) const { // EdgePointCuboidOnlyObjectFixScale::computeError g2o_Object.cpp:336-354
        double acc[3] = {0, 0, 0};
        const int b0 = p->pc_offsets[o], b1 = p->pc_offsets[o + 1];
        const SE3 inv = se3_inv(c.pose);
        const double ratio = p->max_outside_margin_ratio;
        for (int i = b0; i < b1; i++) {
            double lp[3];
            se3_map(inv, p->pc_points + (size_t)i * 3, lp);
            for (int k = 0; k < 3; k++) {
                const double a = std::fabs(lp[k]) * 1.0;
                double er;
                if (a < c.scale[k]) er = 0;
                else if (a < (ratio + 1) * c.scale[k]) er = a - c.scale[k];
                else er = ratio * c.scale[k];
                acc[k] += std::fabs(er);
            }
        }
        if (b1 > b0) for (int k = 0; k < 3; k++) acc[k] = acc[k] / (double)(b1 - b0);
        for (int k = 0; k < 3; k++) e[k] = 1.0 * (acc[k] / c.scale[k]);
    }
    void compute_errors() {
        for (int o = 0; o < p->n_obs; o++) err_obs(o, s.cams[p->obs_cam[o]], &s.pts[(size_t)p->obs_point[o] * 3], &e_obs[(size_t)o * 3]);
        for (int o = 0; o < p->n_cobs; o++) err_cobs(o, s.cams[p->cobs_cam[o]], s.cubs[p->cobs_cuboid[o]], &e_cobs[(size_t)o * 4]);
        for (int o = 0; o < p->n_pc; o++) err_pc(o, s.cubs[p->pc_cuboid[o]], &e_pc[(size_t)o * 3]);
    }
    static void huber(double e, double delta, double *rho) { // robust_kernel_impl.cpp:78-91
        const double dsqr = delta * delta;
        if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
        else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
    }
    // BaseEdge::chi2 = _error.dot(information() * _error) (base_edge.h:58-61): each term is e_k * (w * e_k), summed left to right
    double chi2_obs(int o) const { const double *e = &e_obs[(size_t)o * 3]; const double w = p->obs_inv_sigma2[o]; return stereo(o) ? (e[0] * (w * e[0]) + e[1] * (w * e[1])) + e[2] * (w * e[2]) : e[0] * (w * e[0]) + e[1] * (w * e[1]); }
    double chi2_cobs(int o) const { const double *e = &e_cobs[(size_t)o * 4]; const double *w = p->cobs_info + (size_t)o * 4; return ((e[0] * w[0] * e[0] + e[1] * w[1] * e[1]) + e[2] * w[2] * e[2]) + e[3] * w[3] * e[3]; }
    double chi2_pc(int o) const { const double *e = &e_pc[(size_t)o * 3]; return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]; }
    double robust_chi2() const { // sparse_optimizer.cpp:100-114
        double chi = 0, rho[3];
        for (int o = 0; o < p->n_obs; o++) { double c = chi2_obs(o); if (obs_delta(o) > 0) { huber(c, obs_delta(o), rho); chi += rho[0]; } else chi += c; }
        for (int o = 0; o < p->n_cobs; o++) { double c = chi2_cobs(o); if (p->huber_obj > 0) { huber(c, p->huber_obj, rho); chi += rho[0]; } else chi += c; }
        for (int o = 0; o < p->n_pc; o++) chi += chi2_pc(o);
        return chi;
    }

    double *hpp_block(int i, int j) { // i <= j
        if (i == j) return &Hpp_diag[(size_t)i * 36];
        auto &v = Hpp_off[std::make_pair(i, j)];
        if (v.empty()) v.assign(36, 0.0);
        return v.data();
    }
    // BlockSolver::buildSystem (block_solver.hpp:502-560): linearizeOplus + constructQuadraticForm per edge
    void build_system(int lm_begin, int lm_end, bool with_pose_edges) {
        Hpp_diag.assign((size_t)P * 36, 0.0); Hpp_off.clear(); Hll.assign((size_t)L * 9, 0.0); Hpl.assign((size_t)p->n_obs * 18, 0.0);
        b.assign((size_t)P * 6 + (size_t)L * 3, 0.0);
        double rho[3];
        for (int o = 0; o < p->n_obs; o++) {
            const int li = p->obs_point[o];
            if (li < lm_begin || li >= lm_end) continue;
            const int ci = p->obs_cam[o], pi = cam_idx[ci];
            const SE3 &T = s.cams[ci];
            double pc[3];
            se3_map(T, &s.pts[(size_t)li * 3], pc);
            const double X = pc[0], Y = pc[1], Z = pc[2], Z2 = Z * Z, fx = p->fx, fy = p->fy;
            M3 R; qtoR(T.r, R);
            // EdgeSE3ProjectXYZ::linearizeOplus types_six_dof_expmap.cpp:135-171
            double Ji[3][3], Jj[3][6];
            const bool st = stereo(o);
            if (!st) { // EdgeSE3ProjectXYZ::linearizeOplus types_six_dof_expmap.cpp:135-171
                const double tmp[2][3] = {{fx, 0, -X / Z * fx}, {0, fy, -Y / Z * fy}};
                for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Ji[r][c] = ((-1. / Z * tmp[r][0]) * R[0][c] + (-1. / Z * tmp[r][1]) * R[1][c]) + (-1. / Z * tmp[r][2]) * R[2][c];
                for (int c = 0; c < 3; c++) Ji[2][c] = 0;
            } else { // EdgeStereoSE3ProjectXYZ::linearizeOplus :220-266
                const double bf = p->bf;
                for (int c = 0; c < 3; c++) {
                    Ji[0][c] = -fx * R[0][c] / Z + fx * X * R[2][c] / Z2;
                    Ji[1][c] = -fy * R[1][c] / Z + fy * Y * R[2][c] / Z2;
                    Ji[2][c] = Ji[0][c] - bf * R[2][c] / Z2;
                }
            }
            Jj[0][0] = X * Y / Z2 * fx; Jj[0][1] = -(1 + (X * X / Z2)) * fx; Jj[0][2] = Y / Z * fx; Jj[0][3] = -1. / Z * fx; Jj[0][4] = 0; Jj[0][5] = X / Z2 * fx;
            Jj[1][0] = (1 + Y * Y / Z2) * fy; Jj[1][1] = -X * Y / Z2 * fy; Jj[1][2] = -X / Z * fy; Jj[1][3] = 0; Jj[1][4] = -1. / Z * fy; Jj[1][5] = Y / Z2 * fy;
            if (st) { Jj[2][0] = Jj[0][0] - p->bf * Y / Z2; Jj[2][1] = Jj[0][1] + p->bf * X / Z2; Jj[2][2] = Jj[0][2]; Jj[2][3] = Jj[0][3]; Jj[2][4] = 0; Jj[2][5] = Jj[0][5] - p->bf / Z2; }
            else for (int c = 0; c < 6; c++) Jj[2][c] = 0;
            const double *e = &e_obs[(size_t)o * 3];
            double w = p->obs_inv_sigma2[o], rw = 1.0;
            if (obs_delta(o) > 0) { huber(chi2_obs(o), obs_delta(o), rho); rw = rho[1]; }
            const double omr[3] = {-w * e[0] * rw, -w * e[1] * rw, -w * e[2] * rw}; // omega_r = -omega*e, *= rho[1]
            const double W = rw * w;                                                  // weightedOmega = rho[1]*information (diag)
            double *bl = &b[(size_t)P * 6 + (size_t)li * 3], *hl = &Hll[(size_t)li * 9];
            for (int a = 0; a < 3; a++) { // the third row is exactly zero for monocular edges: adding it leaves the two-row sums unchanged
                bl[a] += (Ji[0][a] * omr[0] + Ji[1][a] * omr[1]) + Ji[2][a] * omr[2];
                for (int c = 0; c < 3; c++) hl[a * 3 + c] += ((Ji[0][a] * W) * Ji[0][c] + (Ji[1][a] * W) * Ji[1][c]) + (Ji[2][a] * W) * Ji[2][c];
            }
            if (pi >= 0) {
                double *bp = &b[(size_t)pi * 6], *hp = &Hpp_diag[(size_t)pi * 36], *hx = &Hpl[(size_t)o * 18];
                for (int a = 0; a < 6; a++) {
                    bp[a] += (Jj[0][a] * omr[0] + Jj[1][a] * omr[1]) + Jj[2][a] * omr[2];
                    for (int c = 0; c < 6; c++) hp[a * 6 + c] += ((Jj[0][a] * W) * Jj[0][c] + (Jj[1][a] * W) * Jj[1][c]) + (Jj[2][a] * W) * Jj[2][c];
                    // the transposed block: B^T * weightedOmega * A under a kernel, B^T * (A^T * omega)^T without one (base_binary_edge.hpp:84-87, 108-111)
                    if (obs_delta(o) > 0) for (int c = 0; c < 3; c++) hx[a * 3 + c] += ((Jj[0][a] * W) * Ji[0][c] + (Jj[1][a] * W) * Ji[1][c]) + (Jj[2][a] * W) * Ji[2][c];
                    else for (int c = 0; c < 3; c++) hx[a * 3 + c] += (Jj[0][a] * (Ji[0][c] * W) + Jj[1][a] * (Ji[1][c] * W)) + Jj[2][a] * (Ji[2][c] * W);
                }
            }
        }
        if (!with_pose_edges) return;
        const double delta = 1e-9, scalar = 1.0 / (2 * delta);
        for (int o = 0; o < p->n_cobs; o++) { // EdgeSE3CuboidFixScaleProj: numeric Jacobians base_binary_edge.hpp:216-320
            const int ci = p->cobs_cam[o], oi = p->cobs_cuboid[o], pi = cam_idx[ci], pj = cub_idx[oi];
            double Ja[4][6], Jb[4][6];
            for (int d = 0; d < 6; d++) {
                double add[6] = {0, 0, 0, 0, 0, 0}, e1[4], e2[4];
                if (pi >= 0) {
                    add[d] = delta; err_cobs(o, se3_mul(se3_exp(add), s.cams[ci]), s.cubs[oi], e1);
                    add[d] = -delta; err_cobs(o, se3_mul(se3_exp(add), s.cams[ci]), s.cubs[oi], e2);
                    for (int k = 0; k < 4; k++) Ja[k][d] = scalar * (e1[k] - e2[k]);
                }
                add[d] = delta; err_cobs(o, s.cams[ci], cuboid_oplus(s.cubs[oi], add, p->cuboid_flags[oi], p->cuboid_scale + (size_t)oi * 3), e1);
                add[d] = -delta; err_cobs(o, s.cams[ci], cuboid_oplus(s.cubs[oi], add, p->cuboid_flags[oi], p->cuboid_scale + (size_t)oi * 3), e2);
                for (int k = 0; k < 4; k++) Jb[k][d] = scalar * (e1[k] - e2[k]);
            }
            const double *e = &e_cobs[(size_t)o * 4], *w = p->cobs_info + (size_t)o * 4;
            double rw = 1.0;
            if (p->huber_obj > 0) { huber(chi2_cobs(o), p->huber_obj, rho); rw = rho[1]; }
            double omr[4], W[4];
            for (int k = 0; k < 4; k++) { omr[k] = -w[k] * e[k] * rw; W[k] = rw * w[k]; }
            auto accum = [&](const double J1[4][6], const double J2[4][6], double *blk) {
                for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) {
                    double sacc = 0;
                    for (int k = 0; k < 4; k++) sacc += (J1[k][a] * W[k]) * J2[k][c];
                    blk[a * 6 + c] += sacc;
                }
            };
            if (pi >= 0) {
                for (int a = 0; a < 6; a++) { double sacc = 0; for (int k = 0; k < 4; k++) sacc += Ja[k][a] * omr[k]; b[(size_t)pi * 6 + a] += sacc; }
                accum(Ja, Ja, hpp_block(pi, pi));
                accum(Ja, Jb, hpp_block(pi, pj)); // cameras precede cuboids: pi < pj
            }
            for (int a = 0; a < 6; a++) { double sacc = 0; for (int k = 0; k < 4; k++) sacc += Jb[k][a] * omr[k]; b[(size_t)pj * 6 + a] += sacc; }
            accum(Jb, Jb, hpp_block(pj, pj));
        }
        for (int o = 0; o < p->n_pc; o++) { // EdgePointCuboidOnlyObjectFixScale: base_unary_edge.hpp:82-123, no robust kernel, information = I
            const int oi = p->pc_cuboid[o], pj = cub_idx[oi];
            double J[3][6];
            for (int d = 0; d < 6; d++) {
                double add[6] = {0, 0, 0, 0, 0, 0}, e1[3], e2[3];
                add[d] = delta; err_pc(o, cuboid_oplus(s.cubs[oi], add, p->cuboid_flags[oi], p->cuboid_scale + (size_t)oi * 3), e1);
                add[d] = -delta; err_pc(o, cuboid_oplus(s.cubs[oi], add, p->cuboid_flags[oi], p->cuboid_scale + (size_t)oi * 3), e2);
                for (int k = 0; k < 3; k++) J[k][d] = scalar * (e1[k] - e2[k]);
            }
            const double *e = &e_pc[(size_t)o * 3];
            double *blk = hpp_block(pj, pj);
            for (int a = 0; a < 6; a++) {
                b[(size_t)pj * 6 + a] += ((J[0][a] * -e[0]) + (J[1][a] * -e[1])) + (J[2][a] * -e[2]);
                for (int c = 0; c < 6; c++) blk[a * 6 + c] += ((J[0][a] * J[0][c]) + (J[1][a] * J[1][c])) + (J[2][a] * J[2][c]);
            }
        }
    }

    // reduced camera system for landmarks [lm_begin, lm_end): Hs (block map upper incl. diag), bs
    void schur(double lambda, bool lambda_on_poses, int lm_begin, int lm_end, std::map<std::pair<int, int>, std::vector<double>> &Hs, std::vector<double> &bs,
               std::vector<double> *Dinv_out) {
        Hs.clear();
        for (int i = 0; i < P; i++) {
            std::vector<double> d(&Hpp_diag[(size_t)i * 36], &Hpp_diag[(size_t)i * 36] + 36);
            if (lambda_on_poses) for (int k = 0; k < 6; k++) d[k * 7] += lambda;
            Hs[std::make_pair(i, i)] = d;
        }
        for (auto &kv : Hpp_off) Hs[kv.first] = kv.second;
        bs.assign(b.begin(), b.begin() + (size_t)P * 6);
        if (Dinv_out) Dinv_out->assign((size_t)L * 9, 0.0);
        for (int li = lm_begin; li < lm_end; li++) { // block_solver.hpp:378-432
            double D[9], Di[9];
            for (int k = 0; k < 9; k++) D[k] = Hll[(size_t)li * 9 + k];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            { // Eigen fixed 3x3 inverse
                auto cf = [&](int i, int j) { int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return D[i1 * 3 + j1] * D[i2 * 3 + j2] - D[i1 * 3 + j2] * D[i2 * 3 + j1]; };
                double c00 = cf(0, 0), c10 = cf(1, 0), c20 = cf(2, 0);
                double det = (c00 * D[0] + c10 * D[3]) + c20 * D[6], inv = 1.0 / det;
                Di[0] = c00 * inv; Di[1] = c10 * inv; Di[2] = c20 * inv;
                Di[3] = cf(0, 1) * inv; Di[4] = cf(1, 1) * inv; Di[5] = cf(2, 1) * inv;
                Di[6] = cf(0, 2) * inv; Di[7] = cf(1, 2) * inv; Di[8] = cf(2, 2) * inv;
            }
            if (Dinv_out) for (int k = 0; k < 9; k++) (*Dinv_out)[(size_t)li * 9 + k] = Di[k];
            const double *bl = &b[(size_t)P * 6 + (size_t)li * 3];
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = (Di[a * 3] * bl[0] + Di[a * 3 + 1] * bl[1]) + Di[a * 3 + 2] * bl[2];
            const std::vector<int> &obs = lm_obs[li];
            for (size_t u = 0; u < obs.size(); u++) {
                const int i1 = cam_idx[p->obs_cam[obs[u]]];
                if (i1 < 0) continue;
                const double *Bi = &Hpl[(size_t)obs[u] * 18];
                double BD[18];
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[a * 3 + c] = (Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c]) + Bi[a * 3 + 2] * Di[6 + c];
                for (int a = 0; a < 6; a++) bs[(size_t)i1 * 6 + a] -= (Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1]) + Bi[a * 3 + 2] * db[2];
                for (size_t v = 0; v < obs.size(); v++) {
                    const int i2 = cam_idx[p->obs_cam[obs[v]]];
                    if (i2 < i1 || (i2 == i1 && v != u)) continue; // a landmark is seen at most once per keyframe (map<KeyFrame*, size_t>)
                    const double *Bj = &Hpl[(size_t)obs[v] * 18];
                    auto &blk = Hs[std::make_pair(i1, i2)];
                    if (blk.empty()) blk.assign(36, 0.0);
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                        blk[a * 6 + c] -= (BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1]) + BD[a * 3 + 2] * Bj[c * 3 + 2];
                }
            }
        }
    }

    // exact solve of the block-sparse SPD system (upper blocks given): RCM ordering + block-envelope Cholesky
    bool solve_reduced(const std::map<std::pair<int, int>, std::vector<double>> &Hs, const std::vector<double> &bs, double *xp) const {
        const int n = P;
        if (n == 0) return true;
        std::vector<std::vector<int>> adj(n);
        for (auto &kv : Hs) if (kv.first.first != kv.first.second) { adj[kv.first.first].push_back(kv.first.second); adj[kv.first.second].push_back(kv.first.first); }
        std::vector<int> order; order.reserve(n);
        std::vector<char> seen(n, 0);
        for (int comp = 0; comp < n; comp++) {
            if (seen[comp]) continue;
            int start = comp; // pseudo-peripheral-ish: lowest degree in the component reached from comp
            {
                std::vector<int> comp_nodes; std::queue<int> q; std::vector<char> s2(n, 0);
                q.push(comp); s2[comp] = 1;
                while (!q.empty()) { int u = q.front(); q.pop(); comp_nodes.push_back(u); for (int v : adj[u]) if (!s2[v] && !seen[v]) { s2[v] = 1; q.push(v); } }
                for (int u : comp_nodes) if (adj[u].size() < adj[start].size()) start = u;
            }
            std::queue<int> q; q.push(start); seen[start] = 1;
            while (!q.empty()) {
                int u = q.front(); q.pop(); order.push_back(u);
                std::vector<int> nb;
                for (int v : adj[u]) if (!seen[v]) { seen[v] = 1; nb.push_back(v); }
                std::sort(nb.begin(), nb.end(), [&](int a, int c) { return adj[a].size() != adj[c].size() ? adj[a].size() < adj[c].size() : a < c; });
                for (int v : nb) q.push(v);
            }
        }
        std::reverse(order.begin(), order.end());
        std::vector<int> pos(n);
        for (int i = 0; i < n; i++) pos[order[i]] = i;
        std::vector<int> first(n);
        for (int i = 0; i < n; i++) first[i] = i;
        for (auto &kv : Hs) { int a = pos[kv.first.first], c = pos[kv.first.second]; if (a > c) std::swap(a, c); first[c] = std::min(first[c], a); }
        std::vector<size_t> rowoff(n + 1, 0);
        for (int i = 0; i < n; i++) rowoff[i + 1] = rowoff[i] + (size_t)(i - first[i] + 1);
        std::vector<double> Lm(rowoff[n] * 36, 0.0); // row i holds blocks (i, first[i]..i), lower triangle
        auto blk = [&](int i, int j) { return &Lm[(rowoff[i] + (size_t)(j - first[i])) * 36]; };
        for (auto &kv : Hs) {
            int a = pos[kv.first.first], c = pos[kv.first.second];
            const double *src = kv.second.data();
            if (a == c) { double *d = blk(a, a); for (int k = 0; k < 36; k++) d[k] = src[k]; }
            else if (a > c) { double *d = blk(a, c); for (int k = 0; k < 36; k++) d[k] = src[k]; }                                   // block (first,second) sits at rows a cols c
            else { double *d = blk(c, a); for (int r = 0; r < 6; r++) for (int q2 = 0; q2 < 6; q2++) d[r * 6 + q2] = src[q2 * 6 + r]; } // transpose
        }
        for (int i = 0; i < n; i++) {
            for (int j = first[i]; j <= i; j++) {
                double *Aij = blk(i, j);
                const int k0 = std::max(first[i], first[j]);
                for (int k = k0; k < j; k++) {
                    const double *Lik = blk(i, k), *Ljk = blk(j, k);
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
                        double sacc = 0;
                        for (int t = 0; t < 6; t++) sacc += Lik[r * 6 + t] * Ljk[c * 6 + t];
                        Aij[r * 6 + c] -= sacc;
                    }
                }
                if (j < i) { // Aij <- Aij * Ljj^-T
                    const double *Ljj = blk(j, j);
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++) {
                            double v = Aij[r * 6 + c];
                            for (int t = 0; t < c; t++) v -= Aij[r * 6 + t] * Ljj[c * 6 + t];
                            Aij[r * 6 + c] = v / Ljj[c * 6 + c];
                        }
                } else { // dense Cholesky of the diagonal block
                    for (int c = 0; c < 6; c++) {
                        double d = Aij[c * 6 + c];
                        for (int t = 0; t < c; t++) d -= Aij[c * 6 + t] * Aij[c * 6 + t];
                        if (!(d > 0)) return false;
                        d = std::sqrt(d);
                        Aij[c * 6 + c] = d;
                        for (int r = c + 1; r < 6; r++) {
                            double v = Aij[r * 6 + c];
                            for (int t = 0; t < c; t++) v -= Aij[r * 6 + t] * Aij[c * 6 + t];
                            Aij[r * 6 + c] = v / d;
                        }
                        for (int r = 0; r < c; r++) Aij[r * 6 + c] = 0;
                    }
                }
            }
        }
        std::vector<double> y((size_t)n * 6);
        for (int i = 0; i < n; i++) for (int k = 0; k < 6; k++) y[(size_t)i * 6 + k] = bs[(size_t)order[i] * 6 + k];
        for (int i = 0; i < n; i++) { // forward
            for (int j = first[i]; j < i; j++) { const double *Lij = blk(i, j); for (int r = 0; r < 6; r++) for (int t = 0; t < 6; t++) y[(size_t)i * 6 + r] -= Lij[r * 6 + t] * y[(size_t)j * 6 + t]; }
            const double *Lii = blk(i, i);
            for (int r = 0; r < 6; r++) { double v = y[(size_t)i * 6 + r]; for (int t = 0; t < r; t++) v -= Lii[r * 6 + t] * y[(size_t)i * 6 + t]; y[(size_t)i * 6 + r] = v / Lii[r * 6 + r]; }
        }
        for (int i = n - 1; i >= 0; i--) { // backward
            const double *Lii = blk(i, i);
            for (int r = 5; r >= 0; r--) { double v = y[(size_t)i * 6 + r]; for (int t = r + 1; t < 6; t++) v -= Lii[t * 6 + r] * y[(size_t)i * 6 + t]; y[(size_t)i * 6 + r] = v / Lii[r * 6 + r]; }
            for (int j = first[i]; j < i; j++) { const double *Lij = blk(i, j); for (int r = 0; r < 6; r++) for (int t = 0; t < 6; t++) y[(size_t)j * 6 + t] -= Lij[r * 6 + t] * y[(size_t)i * 6 + r]; }
        }
        for (int i = 0; i < n; i++) for (int k = 0; k < 6; k++) xp[(size_t)order[i] * 6 + k] = y[(size_t)i * 6 + k];
        return true;
    }

    bool solve(double lambda) { // BlockSolver::solve with Schur (block_solver.hpp:354-486)
        std::map<std::pair<int, int>, std::vector<double>> Hs;
        std::vector<double> bs, Dinv;
        schur(lambda, true, 0, L, Hs, bs, &Dinv);
        x.assign((size_t)P * 6 + (size_t)L * 3, 0.0);
        if (!solve_reduced(Hs, bs, x.data())) return false;
        for (int li = 0; li < L; li++) { // xl = Dinv (bl - Bt xp)
            double cl[3] = {b[(size_t)P * 6 + (size_t)li * 3], b[(size_t)P * 6 + (size_t)li * 3 + 1], b[(size_t)P * 6 + (size_t)li * 3 + 2]};
            for (int o : lm_obs[li]) {
                const int i1 = cam_idx[p->obs_cam[o]];
                if (i1 < 0) continue;
                const double *Bi = &Hpl[(size_t)o * 18], *xp = &x[(size_t)i1 * 6];
                for (int c = 0; c < 3; c++) { double sacc = 0; for (int a = 0; a < 6; a++) sacc += Bi[a * 3 + c] * xp[a]; cl[c] -= sacc; }
            }
            const double *Di = &Dinv[(size_t)li * 9];
            for (int a = 0; a < 3; a++) x[(size_t)P * 6 + (size_t)li * 3 + a] = (Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1]) + Di[a * 3 + 2] * cl[2];
        }
        return true;
    }
    void update() { // SparseOptimizer::update: oplus per active vertex
        for (int i = 0; i < p->n_cams; i++) if (cam_idx[i] >= 0) s.cams[i] = se3_mul(se3_exp(&x[(size_t)cam_idx[i] * 6]), s.cams[i]); // VertexSE3Expmap::oplusImpl
        for (int i = 0; i < p->n_cuboids; i++) s.cubs[i] = cuboid_oplus(s.cubs[i], &x[(size_t)cub_idx[i] * 6], p->cuboid_flags[i], p->cuboid_scale + (size_t)i * 3);
        for (int i = 0; i < L; i++) for (int k = 0; k < 3; k++) s.pts[(size_t)i * 3 + k] += x[(size_t)P * 6 + (size_t)i * 3 + k];
    }
    double lambda_init() const { // optimization_algorithm_levenberg.cpp:166-180
        double mx = 0;
        for (int i = 0; i < P; i++) for (int k = 0; k < 6; k++) mx = std::max(std::fabs(Hpp_diag[(size_t)i * 36 + k * 7]), mx);
        for (int i = 0; i < L; i++) for (int k = 0; k < 3; k++) mx = std::max(std::fabs(Hll[(size_t)i * 9 + k * 4]), mx);
        return 1e-5 * mx;
    }
};

} // namespace

extern "C" {

double orc_ba_errors(const orc_ba_problem *p, double *err_obs, double *err_cobs, double *err_pc) {
    BA ba(p);
    ba.compute_errors();
    if (err_obs) std::memcpy(err_obs, ba.e_obs.data(), ba.e_obs.size() * sizeof(double));
    if (err_cobs) std::memcpy(err_cobs, ba.e_cobs.data(), ba.e_cobs.size() * sizeof(double));
    if (err_pc) std::memcpy(err_pc, ba.e_pc.data(), ba.e_pc.size() * sizeof(double));
    return ba.robust_chi2();
}

int orc_ba_reduced_dense(const orc_ba_problem *p, int lm_begin, int lm_end, int with_pose_edges, double lambda, double *H, double *bvec) {
    BA ba(p);
    ba.compute_errors();
    ba.build_system(lm_begin, lm_end, with_pose_edges != 0);
    std::map<std::pair<int, int>, std::vector<double>> Hs;
    std::vector<double> bs;
    ba.schur(lambda, with_pose_edges != 0, lm_begin, lm_end, Hs, bs, nullptr);
    const int n = ba.P * 6;
    std::fill(H, H + (size_t)n * n, 0.0);
    for (auto &kv : Hs) {
        const int i = kv.first.first, j = kv.first.second;
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
            H[(size_t)(i * 6 + r) * n + j * 6 + c] = kv.second[r * 6 + c];
            if (i != j) H[(size_t)(j * 6 + c) * n + i * 6 + r] = kv.second[r * 6 + c];
        }
    }
    for (int i = 0; i < n; i++) bvec[i] = bs[i];
    return ba.P;
}

int orc_ba_optimize(const orc_ba_problem *p, int iterations, double *cam_pose_out, double *points_out, double *cuboid_pose_out, orc_ba_stats *st) {
    BA ba(p);
    orc_ba_stats S;
    std::memset(&S, 0, sizeof(S));
    double lambda = 0, ni = 2;
    int nBad = 0;
    for (int it = 0; it < iterations; it++) { // OptimizationAlgorithmLevenberg::solve :61-164
        ba.compute_errors();
        double currentChi = ba.robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) S.chi2_init = currentChi;
        ba.build_system(0, ba.L, true);
        if (it == 0) { lambda = ba.lambda_init(); ni = 2; nBad = 0; }
        double rho = 0;
        int qmax = 0;
        do {
            ba.stack.push_back(ba.s);
            bool ok2 = ba.solve(lambda);
            ba.update();
            ba.compute_errors();
            tempChi = ba.robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = (currentChi - tempChi);
            double scale = 0;
            for (size_t j = 0; j < ba.x.size(); j++) scale += ba.x[j] * (lambda * ba.x[j] + ba.b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = (std::min)(alpha, 2. / 3.);
                double scaleFactor = (std::max)(1. / 3., alpha);
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
    for (int i = 0; i < p->n_cams; i++) se3_to7(ba.s.cams[i], cam_pose_out + (size_t)i * 7);
    for (int i = 0; i < p->n_cuboids; i++) se3_to7(ba.s.cubs[i].pose, cuboid_pose_out + (size_t)i * 7);
    std::memcpy(points_out, ba.s.pts.data(), ba.s.pts.size() * sizeof(double));
    return 0;
}

// ---- the pieces of the LM loop one at a time (test hook): tests/test_ref_pins.py drives them from the REFERENCE'S OWN OptimizationAlgorithmLevenberg::solve /
// SparseOptimizer::optimize (oracle/_ref, cut out of the vendored g2o) and compares the run with orc_ba_optimize
struct orc_ba_handle { // the static BA's own pieces, or (dyn) another oracle's behind the same calls
    BA ba; OrcBAIface *dyn = nullptr;
    explicit orc_ba_handle(const orc_ba_problem *p) : ba(p) {}
    ~orc_ba_handle() { delete dyn; }
};
static const orc_ba_problem *empty_problem() { static orc_ba_problem z; static bool init = false; if (!init) { std::memset(&z, 0, sizeof(z)); init = true; } return &z; }
void orc_huber(double e, double delta, double *rho3) { BA::huber(e, delta, rho3); }
orc_ba_handle *orc_ba_open(const orc_ba_problem *p) { return new orc_ba_handle(p); }
orc_ba_handle *orc_badyn_open(const orc_badyn_problem *p) { orc_ba_handle *h = new orc_ba_handle(empty_problem()); h->dyn = orc_badyn_make_iface(p); return h; }
void orc_ba_close(orc_ba_handle *h) { delete h; }
void orc_ba_compute_errors(orc_ba_handle *h) { if (h->dyn) return h->dyn->compute_errors(); h->ba.compute_errors(); }
double orc_ba_robust_chi2(orc_ba_handle *h) { if (h->dyn) return h->dyn->robust_chi2(); return h->ba.robust_chi2(); }
void orc_ba_build_system(orc_ba_handle *h) { if (h->dyn) return h->dyn->build_system(); h->ba.build_system(0, h->ba.L, true); }
void orc_ba_sizes(orc_ba_handle *h, int *P, int *L) { if (h->dyn) { *P = h->dyn->n_pose_blocks(); *L = h->dyn->n_blocks() - *P; return; } *P = h->ba.P; *L = h->ba.L; }
int orc_ba_block_dim(orc_ba_handle *h, int block) { if (h->dyn) return h->dyn->block_dim(block); return block < h->ba.P ? 6 : 3; }
double orc_ba_hessian_diag(orc_ba_handle *h, int block, int j) { // block < P: pose block (6), else landmark block - P (3)
    if (h->dyn) return h->dyn->hessian_diag(block, j);
    return block < h->ba.P ? h->ba.Hpp_diag[(size_t)block * 36 + j * 7] : h->ba.Hll[(size_t)(block - h->ba.P) * 9 + j * 4];
}
int orc_ba_solve(orc_ba_handle *h, double lambda) { if (h->dyn) return h->dyn->solve(lambda) ? 1 : 0; return h->ba.solve(lambda) ? 1 : 0; }
void orc_ba_update(orc_ba_handle *h) { if (h->dyn) return h->dyn->update(); h->ba.update(); }
void orc_ba_push(orc_ba_handle *h) { if (h->dyn) return h->dyn->push(); h->ba.stack.push_back(h->ba.s); }
void orc_ba_pop(orc_ba_handle *h) { if (h->dyn) return h->dyn->pop(); h->ba.s = h->ba.stack.back(); h->ba.stack.pop_back(); }
void orc_ba_discard_top(orc_ba_handle *h) { if (h->dyn) return h->dyn->discard_top(); h->ba.stack.pop_back(); }
const double *orc_ba_x(orc_ba_handle *h, long *n) { if (h->dyn) return h->dyn->x(n); if (n) *n = (long)h->ba.x.size(); return h->ba.x.data(); }
const double *orc_ba_b(orc_ba_handle *h) { if (h->dyn) return h->dyn->b(); return h->ba.b.data(); }
void orc_badyn_read(orc_ba_handle *h, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints) { if (h->dyn) h->dyn->read(cam_pose, obj_pose, vel, points, dpoints); }
int orc_ba_block(orc_ba_handle *h, int kind, int i, int j, double *out) { // after orc_ba_build_system.  kind 0: Hpp(i, i) 36; 1: Hpp(i, j), i < j, 36; 2: Hll(i) 9; 3: Hpl of observation i, 18 (6 x 3)
    const BA &ba = h->ba;
    if (kind == 0) { if (i < 0 || i >= ba.P) return 0; std::copy(&ba.Hpp_diag[(size_t)i * 36], &ba.Hpp_diag[(size_t)i * 36] + 36, out); return 36; }
    if (kind == 1) { auto it = ba.Hpp_off.find(std::make_pair(i, j)); if (it == ba.Hpp_off.end()) return 0; std::copy(it->second.begin(), it->second.end(), out); return 36; }
    if (kind == 2) { if (i < 0 || i >= ba.L) return 0; std::copy(&ba.Hll[(size_t)i * 9], &ba.Hll[(size_t)i * 9] + 9, out); return 9; }
    if (kind == 3) { if (i < 0 || i >= ba.p->n_obs) return 0; std::copy(&ba.Hpl[(size_t)i * 18], &ba.Hpl[(size_t)i * 18] + 18, out); return 18; }
    return 0;
}
int orc_ba_pose_index(orc_ba_handle *h, int is_cuboid, int i) { return is_cuboid ? h->ba.cub_idx[i] : h->ba.cam_idx[i]; } // Hessian block of a camera / cuboid, -1 if fixed
double orc_ba_edge_chi2(orc_ba_handle *h, int kind, int o) { return kind == 0 ? h->ba.chi2_obs(o) : kind == 1 ? h->ba.chi2_cobs(o) : h->ba.chi2_pc(o); } // after orc_ba_compute_errors
void orc_ba_read(orc_ba_handle *h, double *cam_pose_out, double *points_out, double *cuboid_pose_out) {
    const orc_ba_problem *p = h->ba.p;
    for (int i = 0; i < p->n_cams; i++) se3_to7(h->ba.s.cams[i], cam_pose_out + (size_t)i * 7);
    for (int i = 0; i < p->n_cuboids; i++) se3_to7(h->ba.s.cubs[i].pose, cuboid_pose_out + (size_t)i * 7);
    std::memcpy(points_out, h->ba.s.pts.data(), h->ba.s.pts.size() * sizeof(double));
}

} // extern "C"

// ------------------------------------------------------------------------------------------------ Optimizer::PoseOptimization
// Restated from /root/reference/orb_object_slam/src/Optimizer.cc:253-472 and the g2o pieces it reaches: EdgeSE3ProjectXYZOnlyPose /
// EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.h:104-162, .cpp:298-392), BaseUnaryEdge::constructQuadraticForm
// (base_unary_edge.hpp:43-76), RobustKernelHuber, OptimizationAlgorithmLevenberg::solve, VertexSE3Expmap::oplusImpl.
// One 6-dof vertex, N unary edges.  Quirks kept: the estimate is reset to the frame's pose before each of the 4 rounds; inlier
// edges are classified with the error left by the last computeActiveErrors() (the state of the last LM trial, accepted or not),
// outlier (level-1) edges are re-evaluated at the final estimate; chi2 is compared as float; the Huber kernel is dropped after
// the third round; `optimizer.edges().size() < 10` ends the rounds early.
namespace {
struct PoseOpt {
    int n; const double *Xw, *obs, *w; double fx, fy, cx, cy, bf;
    SE3 T;
    std::vector<double> err;      // n x 3, as left by the last evaluation of each edge
    std::vector<uint8_t> level1;  // edge excluded from the optimisation
    bool robust = true;
    bool stereo(int i) const { return obs[(size_t)i * 3 + 2] >= 0; }
    void eval(int i) { // computeError
        double pc[3];
        se3_map(T, Xw + (size_t)i * 3, pc);
        double *e = &err[(size_t)i * 3];
        if (stereo(i)) { // EdgeStereoSE3ProjectXYZOnlyPose::cam_project (types_six_dof_expmap.cpp:331-338): invz is a float there, bf the edge's double
            const float invz = (float)(1.0 / pc[2]);
            const double u = pc[0] * invz * fx + cx;
            e[0] = obs[(size_t)i * 3] - u;
            e[1] = obs[(size_t)i * 3 + 1] - (pc[1] * invz * fy + cy);
            e[2] = obs[(size_t)i * 3 + 2] - (u - bf * invz);
        } else { // EdgeSE3ProjectXYZOnlyPose::cam_project (:322-328) over project2d (:37-42): a division per coordinate
            e[0] = obs[(size_t)i * 3] - (pc[0] / pc[2] * fx + cx);
            e[1] = obs[(size_t)i * 3 + 1] - (pc[1] / pc[2] * fy + cy);
            e[2] = 0.0;
        }
    }
    double chi2(int i) const { const double *e = &err[(size_t)i * 3]; return stereo(i) ? ((e[0] * w[i] * e[0] + e[1] * w[i] * e[1]) + e[2] * w[i] * e[2]) : (e[0] * w[i] * e[0] + e[1] * w[i] * e[1]); }
    double delta(int i) const { return stereo(i) ? (double)(float)std::sqrt(7.815) : (double)(float)std::sqrt(5.991); } // const float deltaMono = sqrt(5.991)
    void compute_active_errors() { for (int i = 0; i < n; i++) if (!level1[i]) eval(i); }
    double active_robust_chi2() const {
        double chi = 0, rho[3];
        for (int i = 0; i < n; i++) if (!level1[i]) { const double c = chi2(i); if (robust) { BA::huber(c, delta(i), rho); chi += rho[0]; } else chi += c; }
        return chi;
    }
    void build(double *H, double *b) const { // linearizeOplus + constructQuadraticForm over the active edges, in edge order
        for (int k = 0; k < 36; k++) H[k] = 0;
        for (int k = 0; k < 6; k++) b[k] = 0;
        for (int i = 0; i < n; i++) {
            if (level1[i]) continue;
            double pc[3], J[18];
            se3_map(T, Xw + (size_t)i * 3, pc);
            const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
            J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
            J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
            const int dim = stereo(i) ? 3 : 2;
            if (dim == 3) { J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2; }
            double rw = 1.0, rho[3];
            if (robust) { BA::huber(chi2(i), delta(i), rho); rw = rho[1]; }
            const double *e = &err[(size_t)i * 3];
            for (int a = 0; a < 6; a++) {
                double acc = 0; // b -= rho[1] * A^T * omega * e (base_unary_edge.hpp:43-76), products left to right
                for (int d = 0; d < dim; d++) acc += ((rw * J[d * 6 + a]) * w[i]) * e[d];
                b[a] -= acc;
                for (int c = 0; c < 6; c++) { double h = 0; for (int d = 0; d < dim; d++) h += J[d * 6 + a] * (rw * w[i]) * J[d * 6 + c]; H[a * 6 + c] += h; }
            }
        }
    }
    static bool solve6(const double *H, const double *b, double lambda, double *x) { // dense SPD solve (LinearSolverDense -> Eigen LDLT)
        double L[36];
        for (int k = 0; k < 36; k++) L[k] = H[k];
        for (int k = 0; k < 6; k++) L[k * 6 + k] += lambda;
        for (int c = 0; c < 6; c++) {
            double d = L[c * 6 + c];
            for (int t = 0; t < c; t++) d -= L[c * 6 + t] * L[c * 6 + t];
            if (!(d > 0)) return false;
            d = std::sqrt(d); L[c * 6 + c] = d;
            for (int r = c + 1; r < 6; r++) { double v = L[r * 6 + c]; for (int t = 0; t < c; t++) v -= L[r * 6 + t] * L[c * 6 + t]; L[r * 6 + c] = v / d; }
        }
        double y[6];
        for (int r = 0; r < 6; r++) { double v = b[r]; for (int t = 0; t < r; t++) v -= L[r * 6 + t] * y[t]; y[r] = v / L[r * 6 + r]; }
        for (int r = 5; r >= 0; r--) { double v = y[r]; for (int t = r + 1; t < 6; t++) v -= L[t * 6 + r] * x[t]; x[r] = v / L[r * 6 + r]; }
        return true;
    }
    int optimize(int iterations) { // SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
        double lambda = 0, ni = 2;
        int nBad = 0, done = 0;
        for (int it = 0; it < iterations; it++) {
            compute_active_errors();
            double currentChi = active_robust_chi2(), tempChi = currentChi;
            const double iniChi = currentChi;
            double H[36], b[6], x[6] = {0, 0, 0, 0, 0, 0};
            build(H, b);
            if (it == 0) { double mx = 0; for (int k = 0; k < 6; k++) mx = std::max(std::fabs(H[k * 6 + k]), mx); lambda = 1e-5 * mx; ni = 2; nBad = 0; }
            double rho = 0;
            int qmax = 0;
            do {
                const SE3 backup = T;
                const bool ok2 = solve6(H, b, lambda, x);
                if (ok2) T = se3_mul(se3_exp(x), T); // oplus; a failed solve leaves x = 0 (block_solver.hpp: _x is zeroed on failure paths)
                compute_active_errors();
                tempChi = active_robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = (std::min)(alpha, 2. / 3.);
                    lambda *= (std::max)(1. / 3., alpha);
                    ni = 2; currentChi = tempChi;
                } else { lambda *= ni; ni *= 2; T = backup; }
                qmax++;
            } while (rho < 0 && qmax < 10);
            done = it + 1;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        return done;
    }
};
} // namespace

extern "C" int orc_pose_optimization(int n, const double *Xw, const double *obs, const double *inv_sigma2, double fx, double fy, double cx, double cy, double bf,
                                     const double *pose_in, double *pose_out, uint8_t *outlier) {
    PoseOpt P;
    P.n = n; P.Xw = Xw; P.obs = obs; P.w = inv_sigma2; P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
    P.err.assign((size_t)n * 3, 0.0); P.level1.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) outlier[i] = 0;
    const SE3 T0 = se3_from7(pose_in);
    P.T = T0;
    if (n < 3) { se3_to7(T0, pose_out); return 0; } // Optimizer.cc:385-386 (pose untouched)
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        P.T = T0;
        P.optimize(10);
        nBad = 0;
        for (int i = 0; i < n; i++) {
            if (outlier[i]) P.eval(i);
            const float chi2 = (float)P.chi2(i);
            if (chi2 > (P.stereo(i) ? chi2Stereo : chi2Mono)) { outlier[i] = 1; P.level1[i] = 1; nBad++; }
            else { outlier[i] = 0; P.level1[i] = 0; }
        }
        if (it == 2) P.robust = false;
        if (n < 10) break;
    }
    se3_to7(P.T, pose_out);
    return n - nBad;
}

// Test hook (tests/test_ref_pins.py): one computeActiveErrors + linearisation of PoseOptimization's graph at pose_in, to be held against the reference's own
// pose-only edges (oracle/_ref).  H 6 x 6, b 6, chi2 n.
extern "C" void orc_pose_linearize(int n, const double *Xw, const double *obs, const double *inv_sigma2, double fx, double fy, double cx, double cy, double bf, const double *pose_in,
                                   int robust, double *H, double *b, double *chi2) {
    PoseOpt P;
    P.n = n; P.Xw = Xw; P.obs = obs; P.w = inv_sigma2; P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
    P.err.assign((size_t)n * 3, 0.0); P.level1.assign((size_t)n, 0);
    P.T = se3_from7(pose_in); P.robust = robust != 0;
    P.compute_active_errors();
    for (int i = 0; i < n; i++) chi2[i] = P.chi2(i);
    P.build(H, b);
}

// ------------------------------------------------------------------------------------------------ 9-dof g2o::cuboid (object_slam's graph)
// Restated from /root/reference/object_slam/include/object_slam/g2o_Object.h:23-191 (class cuboid: exp_update, cube_log_error,
// min_log_error, rotate_cuboid, transform_from), :193-224 (VertexCuboid::oplusImpl) and :227-252 (EdgeSE3Cuboid::computeError);
// SE3Quat::log from Thirdparty/g2o/g2o/types/se3quat.h:229-266.  A cuboid is 10 doubles: [t, qx qy qz qw, half scale].
namespace {
static void se3_log(const SE3 &T, double *res) { // se3quat.h:229-266
    M3 R; qtoR(T.r, R);
    const double d = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1);
    const double dR[3] = {R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]}; // deltaR, se3_ops.hpp
    double om[3];
    M3 Vinv;
    auto build = [&](double coef) { // V_inv = I - 0.5 Omega + coef Omega^2
        M3 O = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}}, O2;
        mat3mul(O, O, O2);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Vinv[i][j] = ((i == j ? 1.0 : 0.0) - 0.5 * O[i][j]) + coef * O2[i][j];
    };
    if (d > 0.99999) { for (int i = 0; i < 3; i++) om[i] = 0.5 * dR[i]; build(1. / 12.); }
    else {
        const double theta = std::acos(d);
        for (int i = 0; i < 3; i++) om[i] = theta / (2 * std::sqrt(1 - d * d)) * dR[i];
        build((1 - theta / (2 * std::tan(theta / 2))) / (theta * theta));
    }
    for (int i = 0; i < 3; i++) { res[i] = om[i]; res[i + 3] = (Vinv[i][0] * T.t[0] + Vinv[i][1] * T.t[1]) + Vinv[i][2] * T.t[2]; }
}
struct Cub9 { SE3 pose; double scale[3]; };
static Cub9 cub9_load(const double *v) { Cub9 c; c.pose.t[0] = v[0]; c.pose.t[1] = v[1]; c.pose.t[2] = v[2]; c.pose.r = Quat{v[3], v[4], v[5], v[6]}; c.scale[0] = v[7]; c.scale[1] = v[8]; c.scale[2] = v[9]; return c; }
static void cub9_store(const Cub9 &c, double *v) { se3_to7(c.pose, v); v[7] = c.scale[0]; v[8] = c.scale[1]; v[9] = c.scale[2]; }
static Cub9 cub9_rotate(const Cub9 &c, double yaw_angle) { // rotate_cuboid :105-116
    Cub9 r;
    SE3 rot; rot.r = Quat{0, 0, std::sin(yaw_angle * 0.5), std::cos(yaw_angle * 0.5)}; rot.t[0] = rot.t[1] = rot.t[2] = 0;
    normalize_rotation(rot); // SE3Quat(q, t) constructor
    r.pose = se3_mul(c.pose, rot);
    r.scale[0] = c.scale[0]; r.scale[1] = c.scale[1]; r.scale[2] = c.scale[2];
    if ((yaw_angle == M_PI / 2.0) || (yaw_angle == -M_PI / 2.0) || (yaw_angle == 3 * M_PI / 2.0)) std::swap(r.scale[0], r.scale[1]);
    return r;
}
static void cub9_log_error(const Cub9 &self, const Cub9 &newone, double *res) { // cube_log_error :66-74
    const SE3 diff = se3_mul(se3_inv(newone.pose), self.pose);
    se3_log(diff, res);
    for (int i = 0; i < 3; i++) res[6 + i] = self.scale[i] - newone.scale[i];
}
static void cub9_min_log_error(const Cub9 &self, const Cub9 &newone, double *res) { // min_log_error :77-101
    const double ang[4] = {-1, 0, 1, 2};
    double best = 0; int lbl = -1; double errs[4][9];
    for (int i = 0; i < 4; i++) {
        cub9_log_error(self, cub9_rotate(newone, ang[i] * M_PI / 2.0), errs[i]);
        double nn = 0; for (int k = 0; k < 9; k++) nn += errs[i][k] * errs[i][k];
        nn = std::sqrt(nn);
        if (lbl < 0 || nn < best) { best = nn; lbl = i; } // minCoeff: first minimum
    }
    for (int k = 0; k < 9; k++) res[k] = errs[lbl][k];
}
} // namespace

extern "C" int orc_cuboid9_oplus(int n, const double *cub, const double *upd, double *out) { // VertexCuboid::oplusImpl -> exp_update :58-64
    for (int i = 0; i < n; i++) {
        Cub9 c = cub9_load(cub + (size_t)i * 10), r;
        normalize_rotation(c.pose);
        r.pose = se3_mul(c.pose, se3_exp(upd + (size_t)i * 9));
        for (int k = 0; k < 3; k++) r.scale[k] = c.scale[k] + upd[(size_t)i * 9 + 6 + k];
        cub9_store(r, out + (size_t)i * 10);
    }
    return 0;
}
extern "C" int orc_cuboid9_edge_error(int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err) { // EdgeSE3Cuboid::computeError
    for (int i = 0; i < n; i++) {
        SE3 Tcw = se3_from7(cam_Tcw + (size_t)i * 7);
        const SE3 Twc = se3_inv(Tcw);
        Cub9 g = cub9_load(cub_global + (size_t)i * 10), m = cub9_load(cub_meas_local + (size_t)i * 10), e;
        normalize_rotation(g.pose); normalize_rotation(m.pose);
        e.pose = se3_mul(Twc, m.pose); // transform_from :119-125
        for (int k = 0; k < 3; k++) e.scale[k] = m.scale[k];
        cub9_min_log_error(g, e, err + (size_t)i * 9);
    }
    return 0;
}
// Test hook for tests/test_ref_pins.py: the pose helpers of this file and se3_util.h one at a time, to be held against the reference's own
// se3quat.h / g2o_Object code (oracle/_ref).  op: 0 SE3Quat::exp(a[6]) 1 log(a[7]) 2 a*b 3 inverse(a) 4 a*point b[3] 5 exptwist_norollpitch(a[6])
// 6 cuboid(a[10]).exp_update(b[9]) 7 min_log_error(self a[10], new b[10]) 8 cube_log_error 9 rotate_cuboid(a[10], s) 10 transform_from(a[10], Twc b[7])
// 11 transform_to 12 point_boundary_error(a[10], point b[3], ratio s).  Poses come in as [t, qx qy qz qw] and are normalised like SE3Quat(Vector7d).
extern "C" int orc_se3_op(int op, const double *a, const double *b, double s, double *out) {
    auto load10 = [](const double *v) { Cub9 c = cub9_load(v); normalize_rotation(c.pose); return c; };
    switch (op) {
    case 0: se3_to7(se3_exp(a), out); return 0;
    case 1: se3_log(se3_from7(a), out); return 0;
    case 2: se3_to7(se3_mul(se3_from7(a), se3_from7(b)), out); return 0;
    case 3: se3_to7(se3_inv(se3_from7(a)), out); return 0;
    case 4: se3_map(se3_from7(a), b, out); return 0;
    case 5: se3_to7(exptwist_norollpitch(a), out); return 0;
    case 6: return orc_cuboid9_oplus(1, a, b, out);
    case 7: cub9_min_log_error(load10(a), load10(b), out); return 0;
    case 8: cub9_log_error(load10(a), load10(b), out); return 0;
    case 9: cub9_store(cub9_rotate(load10(a), s), out); return 0;
    case 10: case 11: {
        const Cub9 c = load10(a); Cub9 r;
        const SE3 T = se3_from7(b);
        r.pose = se3_mul(op == 10 ? T : se3_inv(T), c.pose);
        for (int k = 0; k < 3; k++) r.scale[k] = c.scale[k];
        cub9_store(r, out); return 0;
    }
    case 12: { // cuboid::point_boundary_error g2o_Object.cpp:280-298 (the per-point term of err_pc above, before its fabs / mean / scale division)
        const Cub9 c = load10(a);
        double lp[3];
        se3_map(se3_inv(c.pose), b, lp);
        for (int k = 0; k < 3; k++) {
            const double v = 1.0 * std::fabs(lp[k]);
            out[k] = v < c.scale[k] ? 0.0 : (v < (s + 1) * c.scale[k] ? v - c.scale[k] : s * c.scale[k]);
        }
        return 0;
    }
    case 13: { // cuboid::projectOntoImageBbox g2o_Object.h:197-205: b = Tcw (7) then K (9, row-major)
        const Cub9 c9 = load10(a);
        Cuboid c; c.pose = c9.pose;
        for (int k = 0; k < 3; k++) c.scale[k] = c9.scale[k];
        project_bbox(c, se3_from7(b), b + 7, out);
        return 0;
    }
    }
    return -1;
}
// numeric Jacobians of EdgeSE3Cuboid the way BaseBinaryEdge::linearizeOplus does (base_binary_edge.hpp:55-120): central differences, delta 1e-9,
// column d of vertex i: (e(+delta) - e(-delta)) / (2 delta).  Jcam: 9 x 6 (row-major), Jcub: 9 x 9.
extern "C" int orc_cuboid9_edge_linearize(int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err, double *Jcam, double *Jcub) {
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    for (int i = 0; i < n; i++) {
        const double *T7 = cam_Tcw + (size_t)i * 7, *G = cub_global + (size_t)i * 10, *M = cub_meas_local + (size_t)i * 10;
        orc_cuboid9_edge_error(1, T7, G, M, err + (size_t)i * 9);
        for (int d = 0; d < 6; d++) {
            double e1[9], e2[9], add[6] = {0, 0, 0, 0, 0, 0}, Tp[7];
            for (int sgn = 0; sgn < 2; sgn++) {
                add[d] = sgn == 0 ? delta : -delta;
                SE3 T = se3_mul(se3_exp(add), se3_from7(T7)); // VertexSE3Expmap::oplusImpl
                se3_to7(T, Tp);
                orc_cuboid9_edge_error(1, Tp, G, M, sgn == 0 ? e1 : e2);
            }
            for (int k = 0; k < 9; k++) Jcam[(size_t)i * 54 + k * 6 + d] = scalar * (e1[k] - e2[k]);
        }
        for (int d = 0; d < 9; d++) {
            double e1[9], e2[9], add[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gp[10];
            for (int sgn = 0; sgn < 2; sgn++) {
                add[d] = sgn == 0 ? delta : -delta;
                orc_cuboid9_oplus(1, G, add, Gp);
                orc_cuboid9_edge_error(1, T7, Gp, M, sgn == 0 ? e1 : e2);
            }
            for (int k = 0; k < 9; k++) Jcub[(size_t)i * 81 + k * 9 + d] = scalar * (e1[k] - e2[k]);
        }
    }
    return 0;
}
