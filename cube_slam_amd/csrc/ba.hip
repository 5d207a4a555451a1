// ba.hip -- g2o object bundle adjustment (Levenberg-Marquardt, BlockSolver_6_3 with Schur) on MI355X (gfx950).
//
// Replaces SparseOptimizer::optimize as driven by Optimizer::BundleAdjustment / LocalBACameraPointObjects (reference
// orb_object_slam/src/Optimizer.cc:64-251, :826-1534; vendored g2o core/optimization_algorithm_levenberg.cpp:61-189,
// core/block_solver.hpp:354-604, core/base_binary_edge.hpp:55-320, core/base_unary_edge.hpp:82-123; CubeSLAM types
// orb_object_slam/src/g2o_Object.cpp).  fp64 throughout, like g2o.  Deterministic: every reduction has a fixed order
// (thread per landmark, wave per pose block / Schur block with shuffle trees), no floating-point atomics.
//
//   ba_err_*          computeActiveErrors: reprojection / cuboid bbox / point-in-cuboid residuals + robust chi2 partials
//   ba_lin_lm         thread per landmark: analytic 2x3 / 2x6 Jacobians of its observations, Hll, bl, Hpl blocks
//   ba_lin_pose       wave per pose block: sum_obs Jj^T W Jj and Jj^T W r (lane = observation, shuffle-tree reduce)
//   ba_num_cols       thread per (cuboid edge, perturbed dimension): central difference with delta 1e-9 through the same
//                     oplus (exp map, exptwist_norollpitch) as g2o's numeric linearizeOplus
//   ba_lin_pose_edges lane per element of a pose block: camera-cuboid and point-cuboid terms into Hpp / b
//   ba_lm_dinv        thread per landmark: (Hll + lambda I)^-1 and D^-1 b_l
//   ba_schur_slots    wave per block of the reduced camera system: Hpp - sum_l B D^-1 B^T (lane = contributing landmark)
//   ba_schur_b        wave per pose block: b_p - sum B D^-1 b_l
//   (all-reduce)      one sum over [Schur blocks | b | scalars] across ranks when landmarks are sharded (RCCL via callback)
//   ba_band_*         block-band Cholesky (reverse Cuthill-McKee order from the host) + two triangular sweeps, one workgroup
//   ba_backsub        thread per landmark: x_l = D^-1 (b_l - B^T x_p)
//   ba_update         oplus per vertex (cameras: exp(dx) * T; cuboids: T * exp(dx) with the fix-roll-pitch / height / scale flags)
#include "common.h"
#include "ba_cr.h"
#include "se3_math.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <map>
#include <queue>
#include <vector>

namespace {

struct Params { // device-side view of the graph
    int n_cams, L, n_cub, P, n_obs, n_cobs, n_pc, lm_b, lm_e, o_b, o_e, pose_edges; // this rank: landmarks [lm_b, lm_e), obs [o_b, o_e)
    double fx, fy, cx, cy, huber_mono, huber_obj, margin_ratio, K[9], bf, huber_stereo;
    double *cam, *pts, *cub, *cub_scale;           // estimates: 7 / 3 / 7 (+3)
    const int *cam_idx; const uint8_t *cub_flags;
    const int *o_cam, *o_pt; const double *o_uv, *o_w, *o_ur; const int *lm_off; // o_ur: right-image u of a stereo edge (< 0 or NULL array: monocular)
    const int *c_cam, *c_cub; const double *c_bbox, *c_info;
    const int *pc_cub, *pc_off; const double *pc_pts;
    double *e_obs, *e_cobs, *e_pc;
    double *Hpl, *HplD, *Hll, *bl, *Dinv, *db, *Hpp, *bp, *Hoff; // HplD: B D^-1 per observation (6x3), refreshed with every damping value; // Hoff: one 6x6 block per camera-cuboid edge (camera rows, cuboid cols)
    double *Jc, *Jp;                                       // numeric Jacobian columns: cobs x 12 x 4, pc x 6 x 3
    double *x;                                             // 6P + 3L
};

__device__ inline Cuboid load_cuboid(const Params &G, int i) {
    Cuboid c; c.pose = se3_load(G.cub + (long)i * 7);
    for (int k = 0; k < 3; k++) c.scale[k] = G.cub_scale[i * 3 + k];
    return c;
}
__device__ inline void err_cobs_eval(const Params &G, int o, const SE3 &T, const Cuboid &c, double *e) { // EdgeSE3CuboidFixScaleProj::computeError
    double bb[4];
    project_bbox(c, T, G.K, bb);
    for (int k = 0; k < 4; k++) e[k] = bb[k] - G.c_bbox[o * 4 + k];
}
__device__ inline void err_pc_eval(const Params &G, int o, const Cuboid &c, double *e) { // EdgePointCuboidOnlyObjectFixScale::computeError
    double acc[3] = {0, 0, 0};
    const int b0 = G.pc_off[o], b1 = G.pc_off[o + 1];
    const SE3 inv = se3_inv(c.pose);
    const double ratio = G.margin_ratio;
    for (int i = b0; i < b1; i++) {
        double lp[3];
        se3_map(inv, G.pc_pts + (long)i * 3, lp);
        for (int k = 0; k < 3; k++) { // cuboid::point_boundary_error g2o_Object.cpp:280-298
            const double a = fabs(lp[k]) * 1.0;
            double er;
            if (a < c.scale[k]) er = 0;
            else if (a < (ratio + 1) * c.scale[k]) er = a - c.scale[k];
            else er = ratio * c.scale[k];
            acc[k] += fabs(er);
        }
    }
    if (b1 > b0) for (int k = 0; k < 3; k++) acc[k] = acc[k] / (double)(b1 - b0);
    for (int k = 0; k < 3; k++) e[k] = 1.0 * (acc[k] / c.scale[k]);
}

// block partial sums, then a fixed-order final sum on the host side of the tiny partial array
__device__ inline void block_sum_store(double v, double *partials, int bid) {
    __shared__ double s[4];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[bid] = (s[0] + s[1]) + (s[2] + s[3]);
}
__device__ inline void block_sum_store(double v, double *partials) { block_sum_store(v, partials, (int)blockIdx.x); }

__device__ inline bool obs_stereo(const Params &G, int o) { return G.o_ur && G.o_ur[o] >= 0; }
__device__ __forceinline__ void ba_err_obs_body(const Params &G, double *partials, const int bid) {
    const int o = G.o_b + bid * 256 + threadIdx.x;
    double chi = 0;
    if (o < G.o_e) { // EdgeSE3ProjectXYZ::computeError
        const SE3 T = se3_load(G.cam + (long)G.o_cam[o] * 7);
        double pc[3];
        se3_map(T, G.pts + (long)G.o_pt[o] * 3, pc);
        double e0, e1, e2 = 0.0, delta = G.huber_mono;
        if (obs_stereo(G, o)) { // EdgeStereoSE3ProjectXYZ::cam_project (types_six_dof_expmap.cpp:182-189): invz and bf are floats there
            const float invz = (float)(1.0 / pc[2]);
            const double u = pc[0] * invz * G.fx + G.cx;
            e0 = G.o_uv[o * 2] - u; e1 = G.o_uv[o * 2 + 1] - (pc[1] * invz * G.fy + G.cy); e2 = G.o_ur[o] - (u - (double)((float)G.bf * invz)); // (bf arrives as const float&: the product is a float product)
            chi = ((e0 * e0 + e1 * e1) + e2 * e2) * G.o_w[o];
            delta = G.huber_stereo;
        } else {
            e0 = G.o_uv[o * 2] - (pc[0] / pc[2] * G.fx + G.cx); e1 = G.o_uv[o * 2 + 1] - (pc[1] / pc[2] * G.fy + G.cy);
            chi = (e0 * e0 + e1 * e1) * G.o_w[o];
        }
        G.e_obs[(long)o * 3] = e0; G.e_obs[(long)o * 3 + 1] = e1; G.e_obs[(long)o * 3 + 2] = e2;
        if (delta > 0) { double rho[3]; huber(chi, delta, rho); chi = rho[0]; }
    }
    block_sum_store(chi, partials, bid);
}
__global__ void __launch_bounds__(256) ba_err_obs(Params G, double *partials) { ba_err_obs_body(G, partials, (int)blockIdx.x); }
__device__ __forceinline__ void ba_err_pose_edges_body(const Params &G, double *partials, const int bid) {
    const int t = bid * 256 + threadIdx.x;
    double chi = 0;
    if (t < G.n_cobs) {
        double e[4];
        err_cobs_eval(G, t, se3_load(G.cam + (long)G.c_cam[t] * 7), load_cuboid(G, G.c_cub[t]), e);
        const double *w = G.c_info + (long)t * 4;
        for (int k = 0; k < 4; k++) G.e_cobs[(long)t * 4 + k] = e[k];
        chi = ((e[0] * w[0] * e[0] + e[1] * w[1] * e[1]) + e[2] * w[2] * e[2]) + e[3] * w[3] * e[3];
        if (G.huber_obj > 0) { double rho[3]; huber(chi, G.huber_obj, rho); chi = rho[0]; }
    } else if (t < G.n_cobs + G.n_pc) {
        const int o = t - G.n_cobs;
        double e[3];
        err_pc_eval(G, o, load_cuboid(G, G.pc_cub[o]), e);
        for (int k = 0; k < 3; k++) G.e_pc[(long)o * 3 + k] = e[k];
        chi = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
    }
    block_sum_store(chi, partials, bid);
}
__global__ void __launch_bounds__(256) ba_err_pose_edges(Params G, double *partials) { ba_err_pose_edges_body(G, partials, (int)blockIdx.x); }

// analytic Jacobians of one reprojection edge + weights.  Monocular: EdgeSE3ProjectXYZ::linearizeOplus (types_six_dof_expmap.cpp:135-171),
// third row exactly zero.  Stereo: EdgeStereoSE3ProjectXYZ::linearizeOplus (:220-266).
__device__ inline void obs_jac(const Params &G, int o, double Ji[3][3], double Jj[3][6], double omr[3], double &W) {
    const SE3 T = se3_load(G.cam + (long)G.o_cam[o] * 7);
    double pc[3], R[3][3];
    se3_map(T, G.pts + (long)G.o_pt[o] * 3, pc);
    qtoR(T.r, R);
    const double X = pc[0], Y = pc[1], Z = pc[2], Z2 = Z * Z, fx = G.fx, fy = G.fy;
    const bool st = obs_stereo(G, o);
    if (!st) {
        const double tmp[2][3] = {{fx, 0, -X / Z * fx}, {0, fy, -Y / Z * fy}};
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) Ji[r][c] = ((-1. / Z * tmp[r][0]) * R[0][c] + (-1. / Z * tmp[r][1]) * R[1][c]) + (-1. / Z * tmp[r][2]) * R[2][c];
        for (int c = 0; c < 3; c++) Ji[2][c] = 0;
    } else {
        for (int c = 0; c < 3; c++) {
            Ji[0][c] = -fx * R[0][c] / Z + fx * X * R[2][c] / Z2;
            Ji[1][c] = -fy * R[1][c] / Z + fy * Y * R[2][c] / Z2;
            Ji[2][c] = Ji[0][c] - G.bf * R[2][c] / Z2;
        }
    }
    Jj[0][0] = X * Y / Z2 * fx; Jj[0][1] = -(1 + (X * X / Z2)) * fx; Jj[0][2] = Y / Z * fx; Jj[0][3] = -1. / Z * fx; Jj[0][4] = 0; Jj[0][5] = X / Z2 * fx;
    Jj[1][0] = (1 + Y * Y / Z2) * fy; Jj[1][1] = -X * Y / Z2 * fy; Jj[1][2] = -X / Z * fy; Jj[1][3] = 0; Jj[1][4] = -1. / Z * fy; Jj[1][5] = Y / Z2 * fy;
    if (st) { Jj[2][0] = Jj[0][0] - G.bf * Y / Z2; Jj[2][1] = Jj[0][1] + G.bf * X / Z2; Jj[2][2] = Jj[0][2]; Jj[2][3] = Jj[0][3]; Jj[2][4] = 0; Jj[2][5] = Jj[0][5] - G.bf / Z2; }
    else for (int c = 0; c < 6; c++) Jj[2][c] = 0;
    const double e0 = G.e_obs[(long)o * 3], e1 = G.e_obs[(long)o * 3 + 1], e2 = G.e_obs[(long)o * 3 + 2], w = G.o_w[o];
    const double delta = st ? G.huber_stereo : G.huber_mono;
    double rw = 1.0;
    if (delta > 0) { double rho[3]; huber(st ? ((e0 * e0 + e1 * e1) + e2 * e2) * w : (e0 * e0 + e1 * e1) * w, delta, rho); rw = rho[1]; }
    omr[0] = -w * e0 * rw; omr[1] = -w * e1 * rw; omr[2] = -w * e2 * rw; // omega_r = -Omega e, scaled by rho' (base_binary_edge.hpp:77,96)
    W = rw * w;                                                          // robustInformation = rho' * Omega (base_edge.h:96-102)
}

__device__ __forceinline__ void ba_lin_lm_body(const Params &G, const int bid) { // thread per landmark of this rank
    const int li = G.lm_b + bid * 256 + threadIdx.x;
    if (li >= G.lm_e) return;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int o = G.lm_off[li]; o < G.lm_off[li + 1]; o++) {
        double Ji[3][3], Jj[3][6], omr[3], W;
        obs_jac(G, o, Ji, Jj, omr, W);
        for (int a = 0; a < 3; a++) { // third row: zero for monocular edges, so their sums are unchanged by it
            b[a] += (Ji[0][a] * omr[0] + Ji[1][a] * omr[1]) + Ji[2][a] * omr[2];
            for (int c = 0; c < 3; c++) H[a * 3 + c] += ((Ji[0][a] * W) * Ji[0][c] + (Ji[1][a] * W) * Ji[1][c]) + (Ji[2][a] * W) * Ji[2][c];
        }
        double *hx = G.Hpl + (long)o * 18; // Hpl block (pose rows, landmark cols) = Jj^T W Ji
        for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) hx[a * 3 + c] = ((Jj[0][a] * W) * Ji[0][c] + (Jj[1][a] * W) * Ji[1][c]) + (Jj[2][a] * W) * Ji[2][c];
    }
    for (int k = 0; k < 9; k++) G.Hll[(long)li * 9 + k] = H[k];
    for (int k = 0; k < 3; k++) G.bl[(long)li * 3 + k] = b[k];
}
__global__ void __launch_bounds__(256) ba_lin_lm(Params G) { ba_lin_lm_body(G, (int)blockIdx.x); }

// workgroup per pose block (four waves share the camera's observation list: a thousand single waves leave the SIMDs one wave deep and the
// Jacobian chain exposed): observations of this rank that involve the camera (CSR pose_off / pose_obs); wave sums by shuffles, then in wave order
__device__ __forceinline__ void ba_lin_pose_body(const Params &G, const int *pose_off, const int *pose_obs, const int bid) {
    __shared__ double part[4][42];
    const int pi = bid, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[42];
#pragma unroll
    for (int k = 0; k < 42; k++) acc[k] = 0;
    for (int q = pose_off[pi] + (int)threadIdx.x; q < pose_off[pi + 1]; q += 256) {
        double Ji[3][3], Jj[3][6], omr[3], W;
        obs_jac(G, pose_obs[q], Ji, Jj, omr, W);
#pragma unroll
        for (int a = 0; a < 6; a++) {
            acc[36 + a] += (Jj[0][a] * omr[0] + Jj[1][a] * omr[1]) + Jj[2][a] * omr[2];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[a * 6 + c] += ((Jj[0][a] * W) * Jj[0][c] + (Jj[1][a] * W) * Jj[1][c]) + (Jj[2][a] * W) * Jj[2][c];
        }
    }
#pragma unroll
    for (int k = 0; k < 42; k++) for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0) for (int k = 0; k < 42; k++) part[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 42) {
        const int k = threadIdx.x;
        const double v = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
        if (k < 36) G.Hpp[(long)pi * 36 + k] = v; else G.bp[(long)pi * 6 + k - 36] = v;
    }
}
__global__ void __launch_bounds__(256) ba_lin_pose(Params G, const int *pose_off, const int *pose_obs) { ba_lin_pose_body(G, pose_off, pose_obs, (int)blockIdx.x); }

// numeric Jacobian columns (base_binary_edge.hpp:216-320, base_unary_edge.hpp:82-123): delta = 1e-9, central difference.  A lane per (edge, column,
// SIGN): the even lane evaluates the error at +delta, its odd neighbour at -delta (the two evaluations are the whole cost: se3 exponential, cuboid
// retraction, projection), the difference is taken by the even lane -- the same two values, the same subtraction.
__device__ __forceinline__ void ba_num_cols_body(const Params &G, const int bid) {
    const int tt = bid * 256 + threadIdx.x, t = tt >> 1;
    const bool minus = tt & 1;
    const double delta = 1e-9, scalar = 1.0 / (2 * delta), dd = minus ? -delta : delta;
    double ev[4] = {0, 0, 0, 0};
    int kind = 0, o = 0, d = 0; // 1: camera-cuboid column, 2: point-cuboid column
    if (t < G.n_cobs * 12) {
        o = t / 12; d = t % 12;
        const int ci = G.c_cam[o], oi = G.c_cub[o];
        if (d >= 6 || G.cam_idx[ci] >= 0) {
            kind = 1;
            const SE3 T = se3_load(G.cam + (long)ci * 7);
            const Cuboid C = load_cuboid(G, oi);
            double add[6] = {0, 0, 0, 0, 0, 0};
            if (d < 6) { add[d] = dd; err_cobs_eval(G, o, se3_mul(se3_exp(add), T), C, ev); } // camera: VertexSE3Expmap::oplusImpl = exp(update) * estimate
            else { add[d - 6] = dd; err_cobs_eval(G, o, T, cuboid_oplus(C, add, G.cub_flags[oi], G.cub_scale + (long)oi * 3), ev); }
        }
    } else if (t < G.n_cobs * 12 + G.n_pc * 6) {
        kind = 2;
        const int u = t - G.n_cobs * 12;
        o = u / 6; d = u % 6;
        const int oi = G.pc_cub[o];
        const Cuboid C = load_cuboid(G, oi);
        double add[6] = {0, 0, 0, 0, 0, 0};
        add[d] = dd; err_pc_eval(G, o, cuboid_oplus(C, add, G.cub_flags[oi], G.cub_scale + (long)oi * 3), ev);
    }
    double em[4];
#pragma unroll
    for (int k = 0; k < 4; k++) em[k] = __shfl_xor(ev[k], 1); // the other sign's errors (both lanes of a pair take the same branch)
    if (minus) return;
    if (kind == 1) for (int k = 0; k < 4; k++) G.Jc[((long)o * 12 + d) * 4 + k] = scalar * (ev[k] - em[k]);
    else if (kind == 2) for (int k = 0; k < 3; k++) G.Jp[((long)o * 6 + d) * 3 + k] = scalar * (ev[k] - em[k]);
}
__global__ void __launch_bounds__(256) ba_num_cols(Params G) { ba_num_cols_body(G, (int)blockIdx.x); }
// The three launches of buildSystem that read nothing of each other, as ONE grid: workgroups [0, n_pose) are ba_lin_pose's, the next n_cols ba_num_cols', the rest
// ba_lin_lm's (the longest first).  At 1 000 key frames each of them fills part of the chip for 30 - 65 us; in a row they were 141 us, side by side they are the longest
// of them.  (Two more streams with event joins were measured first: a join costs more than the kernels it orders, 1 040 -> 544 it/s.)
__global__ void __launch_bounds__(256) ba_build_abc(Params G, const int *pose_off, const int *pose_obs, int n_pose, int n_cols) {
    const int bid = (int)blockIdx.x;
    if (bid < n_pose) ba_lin_pose_body(G, pose_off, pose_obs, bid);
    else if (bid < n_pose + n_cols) ba_num_cols_body(G, bid - n_pose);
    else ba_lin_lm_body(G, bid - n_pose - n_cols);
}

// A lane per ELEMENT of a pose block (36 of H, 6 of b; a wave per pose): adds the camera-cuboid / point-cuboid terms (constructQuadraticForm) in edge
// order -- every element is the same sum in the same order as with a thread per block, 42 lanes wide instead of a chain of 42 accumulators per thread
// (1500 threads had the six workgroups of the launch walk their edge lists alone: 71 us).  The same for the off-diagonal Hpp block of a
// camera-cuboid edge: a lane per element.  pe_off/pe_list: per pose block, incident edges encoded as (edge << 2) | kind, kind 0 = cobs seen from
// the camera, 1 = cobs seen from the cuboid, 2 = point-cuboid edge
__global__ void __launch_bounds__(256) ba_lin_pose_edges(Params G, const int *pe_off, const int *pe_list) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave < G.P) {
        const int t = wave;
        if (lane >= 42) return;
        const bool isb = lane >= 36;
        const int a = isb ? lane - 36 : lane / 6, c = isb ? 0 : lane % 6;
        double acc = isb ? G.bp[(long)t * 6 + a] : G.Hpp[(long)t * 36 + lane];
        for (int q = pe_off[t]; q < pe_off[t + 1]; q++) {
            const int kind = pe_list[q] & 3, o = pe_list[q] >> 2;
            if (kind < 2) {
                const double *J = G.Jc + ((long)o * 12 + (kind ? 6 : 0)) * 4; // column-major: J[d*4 + k]
                const double *e = G.e_cobs + (long)o * 4, *w = G.c_info + (long)o * 4;
                double rw = 1.0;
                if (G.huber_obj > 0) { double rho[3]; huber(((e[0] * w[0] * e[0] + e[1] * w[1] * e[1]) + e[2] * w[2] * e[2]) + e[3] * w[3] * e[3], G.huber_obj, rho); rw = rho[1]; }
                if (isb) { double sb = 0; for (int k = 0; k < 4; k++) sb += J[a * 4 + k] * (-w[k] * e[k] * rw); acc += sb; }
                else { double sh = 0; for (int k = 0; k < 4; k++) sh += (J[a * 4 + k] * (rw * w[k])) * J[c * 4 + k]; acc += sh; }
            } else {
                const double *J = G.Jp + (long)o * 18, *e = G.e_pc + (long)o * 3; // information = I, no kernel
                if (isb) acc += ((J[a * 3] * -e[0]) + (J[a * 3 + 1] * -e[1])) + (J[a * 3 + 2] * -e[2]);
                else acc += ((J[a * 3] * J[c * 3]) + (J[a * 3 + 1] * J[c * 3 + 1])) + (J[a * 3 + 2] * J[c * 3 + 2]);
            }
        }
        if (isb) G.bp[(long)t * 6 + a] = acc; else G.Hpp[(long)t * 36 + lane] = acc;
    } else if (wave < G.P + G.n_cobs) {
        const int o = wave - G.P;
        if (lane >= 36 || G.cam_idx[G.c_cam[o]] < 0) return;
        const int a = lane / 6, c = lane % 6;
        const double *Ja = G.Jc + (long)o * 48, *Jb = Ja + 24, *e = G.e_cobs + (long)o * 4, *w = G.c_info + (long)o * 4;
        double rw = 1.0;
        if (G.huber_obj > 0) { double rho[3]; huber(((e[0] * w[0] * e[0] + e[1] * w[1] * e[1]) + e[2] * w[2] * e[2]) + e[3] * w[3] * e[3], G.huber_obj, rho); rw = rho[1]; }
        double sh = 0;
        for (int k = 0; k < 4; k++) sh += (Ja[a * 4 + k] * (rw * w[k])) * Jb[c * 4 + k];
        G.Hoff[(long)o * 36 + lane] = sh;
    }
}

__global__ void __launch_bounds__(256) ba_lm_dinv(Params G, double lambda) { // block_solver.hpp:383-395
    const int li = G.lm_b + blockIdx.x * 256 + threadIdx.x;
    if (li >= G.lm_e) return;
    double D[9], Di[9];
    for (int k = 0; k < 9; k++) D[k] = G.Hll[(long)li * 9 + k];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    auto cf = [&](int i, int j) { int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return D[i1 * 3 + j1] * D[i2 * 3 + j2] - D[i1 * 3 + j2] * D[i2 * 3 + j1]; };
    const double c00 = cf(0, 0), c10 = cf(1, 0), c20 = cf(2, 0);
    const double det = (c00 * D[0] + c10 * D[3]) + c20 * D[6], inv = 1.0 / det;
    Di[0] = c00 * inv; Di[1] = c10 * inv; Di[2] = c20 * inv;
    Di[3] = cf(0, 1) * inv; Di[4] = cf(1, 1) * inv; Di[5] = cf(2, 1) * inv;
    Di[6] = cf(0, 2) * inv; Di[7] = cf(1, 2) * inv; Di[8] = cf(2, 2) * inv;
    const double *b = G.bl + (long)li * 3;
    for (int k = 0; k < 9; k++) G.Dinv[(long)li * 9 + k] = Di[k];
    for (int a = 0; a < 3; a++) G.db[(long)li * 3 + a] = (Di[a * 3] * b[0] + Di[a * 3 + 1] * b[1]) + Di[a * 3 + 2] * b[2];
}

// wave per block of the reduced system.  slot s < P: diagonal block of pose s; P <= s < P + n_cobs: camera-cuboid block;
// the rest: camera-camera blocks created by shared landmarks.  trip_*: contributing (obs_u, obs_v) pairs, this rank only.
// B D^-1 of every observation of this rank (6x3 per observation): each is used by all the pairs its observation takes part in
// (k(k+1)/2 pairs for a landmark seen k times), so it is formed once here instead of once per pair
__global__ void __launch_bounds__(256) ba_schur_bd(Params G, double *w) { // thread per row of a block: neighbouring lanes touch neighbouring 24-byte rows
    const long t = (long)G.o_b * 6 + (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)G.o_e * 6) return;
    const int u = (int)(t / 6);
    const long pt = G.o_pt[u];
    const double *Bi = G.Hpl + t * 3, *Di = G.Dinv + pt * 9, *d = G.db + pt * 3;
    const double b0 = Bi[0], b1 = Bi[1], b2 = Bi[2];
    double *o = G.HplD + t * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (b0 * Di[c] + b1 * Di[3 + c]) + b2 * Di[6 + c];
    w[t] = (b0 * d[0] + b1 * d[1]) + b2 * d[2]; // row of B (D^-1 b_l): summed per pose by ba_schur_b
}
// 18 doubles of a 6x3 block with nine 16-byte loads (blocks are 144 bytes apart in 256-byte-aligned arenas): the pair loop is bound by
// the number of gathered lanes per load instruction, not by their width
__device__ __forceinline__ void load_block18(const double *p, double (&v)[18]) {
    const double2 *q = reinterpret_cast<const double2 *>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int k = 0; k < 9; k++) { const double2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}
// The 64 pairs of a wave iteration need 128 blocks of 144 bytes from arbitrary places.  A lane fetching its own two blocks makes
// every load instruction touch 64 different cache lines (the gather cost is per line touched per instruction: 18 instructions x 64
// lines); instead the wave fetches the 128 blocks cooperatively -- 16 bytes per lane, nine consecutive lanes on one block, so an
// instruction touches ~11 lines -- into LDS, and every lane then reads its pair from there.  Same arithmetic, same lane -> pair
// assignment and the same reduction as a per-lane gather, so the sums are bit-identical to it.
__global__ void __launch_bounds__(256) ba_schur_slots(Params G, int n_slots, const int *slot_perm, const int *slot_off, const int2 *trips, double lambda, double *S) {
    __shared__ double2 s_blk[4][64 * 9]; // one operand at a time (B D^-1 of the 64 pairs, then B): 9 KB per wave keeps 16 waves per CU
    __shared__ int s_idx[4][128];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (blockIdx.x * 4 + wv >= n_slots) return; // whole wave; no workgroup barrier below
    // slot_perm: workgroup b runs on XCD b % 8 (observed dispatch rule), and the permutation gives every XCD a contiguous range of
    // the trajectory, so the B / B D^-1 blocks of a camera's observations -- shared by the ~Bc slots of that camera -- are fetched
    // into one XCD's L2 instead of all eight (fabric reads 557 MB -> see DESIGN.md 7.5); speed only, the sums do not change
    const int s = slot_perm[blockIdx.x * 4 + wv];
    double2 *sb = s_blk[wv];
    int *si = s_idx[wv];
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0;
    const int qe = slot_off[s + 1];
    int2 tn = (slot_off[s] + lane < qe) ? trips[slot_off[s] + lane] : make_int2(-1, -1); // the pair list is read one iteration ahead
    for (int q0 = slot_off[s]; q0 < qe; q0 += 64) {
        const int2 t = tn;
        const bool valid = t.x >= 0;
        if (q0 + 64 < qe) tn = (q0 + 64 + lane < qe) ? trips[q0 + 64 + lane] : make_int2(-1, -1);
        si[lane] = t.x; si[64 + lane] = t.y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double BD[18], Bj[18];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            double2 pv[9]; // all requests first, then the LDS stores: a store right after its load would expose every round trip
#pragma unroll
            for (int it = 0; it < 9; it++) {
                const int p = it * 64 + lane, blk = (p * 7282) >> 16, piece = p - blk * 9; // p / 9 for p < 1152
                const int id = max(si[half * 64 + blk], 0); // pairs past the end of the list read block 0; their lanes do not accumulate
                pv[it] = reinterpret_cast<const double2 *>(__builtin_assume_aligned((half == 0 ? G.HplD : G.Hpl) + (long)id * 18, 16))[piece];
            }
#pragma unroll
            for (int it = 0; it < 9; it++) sb[it * 64 + lane] = pv[it];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 9; k++) { const double2 a = sb[lane * 9 + k]; if (half == 0) { BD[2 * k] = a.x; BD[2 * k + 1] = a.y; } else { Bj[2 * k] = a.x; Bj[2 * k + 1] = a.y; } }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (valid) {
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a * 6 + c] -= (BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1]) + BD[a * 3 + 2] * Bj[c * 3 + 2];
        }
    }
#pragma unroll
    for (int k = 0; k < 36; k++) for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0) {
        if (s < G.P) for (int k = 0; k < 36; k++) acc[k] += G.Hpp[(long)s * 36 + k] + ((G.pose_edges && (k % 7) == 0) ? lambda : 0.0); // setLambda on rank 0 only
        else if (s < G.P + G.n_cobs) { if (G.pose_edges && G.cam_idx[G.c_cam[s - G.P]] >= 0) for (int k = 0; k < 36; k++) acc[k] += G.Hoff[(long)(s - G.P) * 36 + k]; }
        for (int k = 0; k < 36; k++) S[(long)s * 36 + k] = acc[k];
    }
}
// per-pose sums of the rows w = B (D^-1 b_l) that ba_schur_bd wrote (48 bytes per observation instead of a 168-byte gather)
__global__ void __launch_bounds__(256) ba_schur_b(Params G, const int *pose_off, const int *pose_obs, const double *w, double *bs) { // workgroup per pose, as ba_lin_pose
    __shared__ double part[4][6];
    const int pi = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int q = pose_off[pi] + (int)threadIdx.x; q < pose_off[pi + 1]; q += 256) {
        const double2 *wo = reinterpret_cast<const double2 *>(__builtin_assume_aligned(w + (long)pose_obs[q] * 6, 16));
        const double2 w0 = wo[0], w1 = wo[1], w2 = wo[2];
        acc[0] -= w0.x; acc[1] -= w0.y; acc[2] -= w1.x; acc[3] -= w1.y; acc[4] -= w2.x; acc[5] -= w2.y;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0) for (int k = 0; k < 6; k++) part[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 6) { const int k = threadIdx.x; bs[(long)pi * 6 + k] = G.bp[(long)pi * 6 + k] + (((part[0][k] + part[1][k]) + part[2][k]) + part[3][k]); }
}

// ------------------------------------------------------------------------------------------------ band path of the reduced solve
// When no two cuboids share a block of the reduced system (always true for the graphs Optimizer.cc builds: cuboids only meet
// cameras) the cuboids are eliminated like landmarks -- 6x6 blocks, independent, in parallel -- and what remains is a system
// over the cameras whose blocks couple cameras that share a landmark or a cuboid.  For a trajectory that is block-banded with a
// small half-width Bc (9 for the 1k-keyframe benchmark graph), and it is factored by ONE workgroup that keeps the active
// (Bc+1) x (Bc+1) window of blocks in LDS: no global-memory round trip on the critical path of a column (the sparse kernel
// below pays three per column).  Exact same system as BlockSolver::solve, different elimination order (round-off only).
struct CubInv { double D[36]; double g[6]; };

// D_q = (H_qq)^-1 (Cholesky inverse), g_q = D_q b_q.  S: slots, bs: right-hand side of the reduced system (6P)
__global__ void __launch_bounds__(64) ba_cub_inv(int C, int Q, const double *S, const double *bs, double *cubD, double *cubg, int *status) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Q) return;
    double A[36], Li[36];
    const double *H = S + (long)(C + q) * 36;
    for (int k = 0; k < 36; k++) A[k] = H[k];
    bool fail = false;
    for (int c = 0; c < 6; c++) { // A = L L^T, in place (lower)
        double d = A[c * 6 + c];
        for (int t = 0; t < c; t++) d -= A[c * 6 + t] * A[c * 6 + t];
        if (!(d > 0)) { fail = true; d = 1; }
        d = sqrt(d);
        A[c * 6 + c] = d;
        for (int r = c + 1; r < 6; r++) { double v = A[r * 6 + c]; for (int t = 0; t < c; t++) v -= A[r * 6 + t] * A[c * 6 + t]; A[r * 6 + c] = v / d; }
    }
    for (int c = 0; c < 6; c++) // Li = L^-1 (lower)
        for (int r = 0; r < 6; r++) {
            if (r < c) { Li[r * 6 + c] = 0; continue; }
            double v = r == c ? 1.0 : 0.0;
            for (int t = c; t < r; t++) v -= A[r * 6 + t] * Li[t * 6 + c];
            Li[r * 6 + c] = v / A[r * 6 + r];
        }
    double *D = cubD + (long)q * 36, *g = cubg + (long)q * 6;
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) { double v = 0; for (int t = (r > c ? r : c); t < 6; t++) v += Li[t * 6 + r] * Li[t * 6 + c]; D[r * 6 + c] = v; }
    const double *bq = bs + (long)(C + q) * 6;
    for (int r = 0; r < 6; r++) { double v = 0; for (int c = 0; c < 6; c++) v += D[r * 6 + c] * bq[c]; g[r] = v; }
    if (fail) *status = 1;
}
// band[target] = camera block of the reduced system - sum over the cuboids seen by both cameras of H_iq D_q H_kq^T.
// One thread per (target block, element); fixed summation order.  ct_list: (cuboid, slot of edge (i,q), slot of edge (k,q)).
__device__ __forceinline__ void ba_band_assemble_body(int n_targets, const int *tgt_slot, const uint8_t *tgt_tr, const int *ct_off, const int *ct_list, const double *S,
                                                        const double *cubD, double *band, const int bid) {
    const int t = bid * 256 + threadIdx.x;
    if (t >= n_targets * 36) return;
    const int tg = t / 36, k = t % 36, r = k / 6, c = k % 6;
    double acc = 0;
    const int sl = tgt_slot[tg];
    if (sl >= 0) acc = tgt_tr[tg] ? S[(long)sl * 36 + c * 6 + r] : S[(long)sl * 36 + k];
    for (int e = ct_off[tg]; e < ct_off[tg + 1]; e++) {
        const double *D = cubD + (long)ct_list[e * 3] * 36, *Hi = S + (long)ct_list[e * 3 + 1] * 36 + r * 6, *Hk = S + (long)ct_list[e * 3 + 2] * 36 + c * 6;
        double v = 0;
        for (int a = 0; a < 6; a++) { double w = 0; for (int bb = 0; bb < 6; bb++) w += D[a * 6 + bb] * Hk[bb]; v += Hi[a] * w; }
        acc -= v;
    }
    band[t] = acc;
}
__global__ void __launch_bounds__(256) ba_band_assemble(int n_targets, const int *tgt_slot, const uint8_t *tgt_tr, const int *ct_off, const int *ct_list, const double *S,
                                                        const double *cubD, double *band) { ba_band_assemble_body(n_targets, tgt_slot, tgt_tr, ct_off, ct_list, S, cubD, band, (int)blockIdx.x); }
// rhs_i = b_i - sum over the cuboids seen by camera i of H_iq g_q.  cr_list: (cuboid, slot)
__device__ __forceinline__ void ba_band_rhs_body(int C, const int *cr_off, const int *cr_list, const double *S, const double *bs, const double *cubg, double *rhs, const int bid) {
    const int t = bid * 256 + threadIdx.x;
    if (t >= C * 6) return;
    const int i = t / 6, r = t % 6;
    double acc = bs[t];
    for (int e = cr_off[i]; e < cr_off[i + 1]; e++) {
        const double *g = cubg + (long)cr_list[e * 2] * 6, *H = S + (long)cr_list[e * 2 + 1] * 36 + r * 6;
        double v = 0;
        for (int a = 0; a < 6; a++) v += H[a] * g[a];
        acc -= v;
    }
    rhs[t] = acc;
}
__global__ void __launch_bounds__(256) ba_band_rhs(int C, const int *cr_off, const int *cr_list, const double *S, const double *bs, const double *cubg, double *rhs) { ba_band_rhs_body(C, cr_off, cr_list, S, bs, cubg, rhs, (int)blockIdx.x); }
// x_q = g_q - D_q sum over the cameras that see cuboid q of H_iq^T x_i.  cq_list: (slot, camera)
__device__ __forceinline__ void ba_cub_back_body(int C, int Q, const int *cq_off, const int *cq_list, const double *S, const double *cubD, const double *cubg, double *x, const int q) {
    if (q >= Q) return;
    double w[6] = {0, 0, 0, 0, 0, 0};
    for (int e = cq_off[q]; e < cq_off[q + 1]; e++) {
        const double *H = S + (long)cq_list[e * 2] * 36, *xi = x + (long)cq_list[e * 2 + 1] * 6;
        for (int c = 0; c < 6; c++) { double v = 0; for (int r = 0; r < 6; r++) v += H[r * 6 + c] * xi[r]; w[c] += v; }
    }
    const double *D = cubD + (long)q * 36, *g = cubg + (long)q * 6;
    for (int r = 0; r < 6; r++) { double v = 0; for (int c = 0; c < 6; c++) v += D[r * 6 + c] * w[c]; x[(long)(C + q) * 6 + r] = g[r] - v; }
}
__global__ void __launch_bounds__(64) ba_cub_back(int C, int Q, const int *cq_off, const int *cq_list, const double *S, const double *cubD, const double *cubg, double *x) { ba_cub_back_body(C, Q, cq_off, cq_list, S, cubD, cubg, x, (int)(blockIdx.x * 64 + threadIdx.x)); }
// Pairs of small launches that read nothing of each other, as one grid each (every one of them is 10 - 40 us of mostly launch and tail at 1 000 key frames):
// the band's blocks and its right-hand side; the cuboids' and the landmarks' back-substitution; the scale of the step and the update of the estimates.
__global__ void __launch_bounds__(256) ba_band_both(int nb_a, int n_targets, const int *tgt_slot, const uint8_t *tgt_tr, const int *ct_off, const int *ct_list, const double *S, const double *cubD, double *band,
                                                    int C, const int *cr_off, const int *cr_list, const double *bs, const double *cubg, double *rhs) {
    const int bid = (int)blockIdx.x;
    if (bid < nb_a) ba_band_assemble_body(n_targets, tgt_slot, tgt_tr, ct_off, ct_list, S, cubD, band, bid);
    else ba_band_rhs_body(C, cr_off, cr_list, S, bs, cubg, rhs, bid - nb_a);
}
// ---- block-band Cholesky of the camera system.  A: C columns x (Bc+1) blocks (block d of column j = block (j+d, j), row-major 6x6).
// The elimination is a serial chain over the columns (each step: 6x6 pivot, Bc panel blocks, Bc(Bc+1)/2 trailing blocks, all in
// LDS), so it is run from BOTH ends at once ("twisted" factorisation): workgroup 0 eliminates columns 0, 1, 2, ... of A,
// workgroup 1 eliminates C-1, C-2, ... (the same band algorithm on the index-reversed matrix, whose columns are gathered from A
// with a transpose), each stopping before a middle group of R = Bc cameras.  Neither side creates fill outside the band, the
// middle group receives the Schur updates of both sides (ba_band_mid adds them, factors the dense 6R x 6R block and solves it),
// and the two back substitutions again run concurrently.  Half the chain length, three launches instead of one.
// A "side" sees the principal submatrix of its ne eliminated columns plus the R tail (middle) columns, in its own (local) order.
struct BandView {
    const double *A, *rhs; // the assembled band and right-hand side (original order)
    int C, Bc, n, ne, rev; // n = ne + tail columns; rev: local column j is original column C-1-j
    int lf_base;           // first column of this side in the factor storage Lf / ybuf
};
__device__ __forceinline__ double band_a(const BandView &V, int cn, int i) { // element i (block d = i/36) of local column cn
    const int d = i / 36;
    if (cn + d >= V.n) return 0.0; // rows outside the side's submatrix
    if (!V.rev) return V.A[(long)cn * ((V.Bc + 1) * 36) + i];
    if (cn >= V.ne) return 0.0;    // middle x middle blocks (and the middle right-hand side) are counted once, by the forward side
    const int e = i - d * 36, r = e / 6, c = e - r * 6, o = V.C - 1 - cn;
    return V.A[(long)(o - d) * ((V.Bc + 1) * 36) + d * 36 + c * 6 + r]; // reversed block (cn+d, cn) = A(o, o-d)^T ... stored in column o-d
}
__device__ __forceinline__ double band_b(const BandView &V, int cn, int t) {
    if (cn >= V.n) return 0.0;
    if (!V.rev) return V.rhs[(long)cn * 6 + t];
    return cn < V.ne ? V.rhs[(long)(V.C - 1 - cn) * 6 + t] : 0.0;
}
// value of lane - n (DPP row_shr:n inside each row of 16 lanes; 0.0 where the row has no such lane)
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int BAND_YB = 28; // doubles per column in ybuf: y_j (6), lower triangle of L_jj^-1 (21), pad
struct BandLds { double *W, *bw, *yj, *Linv, *part; int *pair_d; };
__device__ __forceinline__ BandLds band_lds(double *sh, int Bc) {
    const int NB = Bc + 1, NS = Bc + 2, CS = NB * 36;
    BandLds L;
    L.W = sh; L.bw = L.W + (long)NS * CS; L.yj = L.bw + NS * 6; L.Linv = L.yj + 6; L.part = L.Linv + 6; // part: 64 doubles; the factor of the current diagonal block (36) and the lower triangle of its inverse (21)
    L.pair_d = (int *)(L.part + 64); // pair -> (di << 8 | dk)
    return L;
}
static size_t band_lds_bytes(int Bc) { return sizeof(double) * ((size_t)(Bc + 2) * (Bc + 1) * 36 + (size_t)(Bc + 2) * 6 + 6 + 6 + 64) + sizeof(int) * (size_t)std::max(1, Bc * (Bc + 1) / 2); }

// Eliminates local columns 0..ne-1 (L to Lf, y_j and 1/diag to ybuf), forward solve fused.  LDS: a ring of Bc+2 columns (the extra
// slot receives column j+Bc+1 while column j is processed) and the matching right-hand-side window.  On return the window holds
// the updated tail columns ne..n-1.
typedef double band_v4d __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ bool band_factor(const BandView &V, const BandLds &S, double *Lf, double *ybuf) {
    const int Bc = V.Bc, NB = Bc + 1, NS = Bc + 2, CS = NB * 36;
    double *W = S.W, *bw = S.bw, *yj = S.yj;
    int *pair_d = S.pair_d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int pr = tid; pr < Bc * (Bc + 1) / 2; pr += NT) {
        int di = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
        while (di * (di + 1) / 2 > pr) di--;
        while ((di + 1) * (di + 2) / 2 <= pr) di++;
        pair_d[pr] = ((di + 1) << 8) | (pr - di * (di + 1) / 2 + 1);
    }
    const int n0 = V.n < NB ? V.n : NB;
    for (int i = tid; i < n0 * CS; i += NT) W[i] = band_a(V, i / CS, i % CS); // columns 0..n0-1 sit in slots 0..n0-1
    for (int i = tid; i < n0 * 6; i += NT) bw[i] = band_b(V, i / 6, i % 6);
    // prefetch addressing of this thread's (up to four) elements of a column, hoisted out of the chain: element i of local column cn
    // lives at pk[u] + cn * cstep and exists while cn < plim[u]
    long pk[4]; int plim[4];
    const long cstep = V.rev ? -(long)CS : (long)CS;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = tid + u * NT, d = i / 36, e = i - d * 36, r = e / 6, c = e - r * 6;
        plim[u] = i < CS ? (V.rev ? min(V.n - d, V.ne) : V.n - d) : 0;
        pk[u] = V.rev ? (long)(V.C - 1 - d) * CS + d * 36 + c * 6 + r : (long)i;
    }
    // trailing update on the matrix cores (Bc <= 10): T -= P P^T with P = the nd*6 x 6 panel, as 16x16 tiles of v_mfma_f64_16x16x4_f64
    // (lower tiles only, K = 6 padded to 8).  Every wave owns up to two tiles; where its operands and results live inside a column
    // of the window does not depend on the column, so it is worked out once: offsets inside the column for the two A and two B
    // operand values, and for the four results (block distance dk of the target column, offset inside it, first panel block di that
    // must exist).  C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.
    const bool use_mfma = Bc <= 10;
    int ta_off[2], tb_off[2], ta_di[2], tb_di[2], t_dk[2][4], t_in[2][4], t_di[2][4];
    const int k0 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int tl = wave >= 1 ? (wave - 1) + q * (NT / 64 - 1) : 99; // wave 0 owns the pivot and its bookkeeping, the other five the ten tiles
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tl) ti++;
        const int tj = tl - ti * (ti + 1) / 2;
        const bool on = use_mfma && tl < 10;
        const int arow = ti * 16 + (lane & 15), bcol = tj * 16 + (lane & 15);
        ta_off[q] = (1 + arow / 6) * 36 + (arow % 6) * 6 + k0; ta_di[q] = on ? arow / 6 + 1 : 99;
        tb_off[q] = (1 + bcol / 6) * 36 + (bcol % 6) * 6 + k0; tb_di[q] = on ? bcol / 6 + 1 : 99;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = ti * 16 + k0 + 4 * g, col = bcol, di = row / 6 + 1, dk = col / 6 + 1;
            t_dk[q][g] = dk * CS; t_in[q][g] = (di - dk) * 36 + (row % 6) * 6 + col % 6; // slot distance in doubles, offset inside the column
            t_di[q][g] = (on && col <= row) ? di : 99; // needs panel block di (and dk <= di)
        }
    }
    const int NSCS = NS * CS;
    const int p_off = (1 + lane / 6) * 36 + (lane % 6) * 6, p_dd = 1 + lane / 6, p_r = lane % 6; // panel row / right-hand-side row of this lane (wave-local index)
    __syncthreads();
#ifdef BAND_PROF
    unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p0_ = 0, p1_;
#define BP(k) do { p1_ = __builtin_readcyclecounter(); pt_[k] += p1_ - p0_; p0_ = p1_; } while (0)
    p0_ = __builtin_readcyclecounter();
#else
#define BP(k) do { } while (0)
#endif
    bool fail = false;
    int sj = 0, sjcs = 0; // j % NS, and times CS
    for (int j = 0; j < V.ne; j++) {
        double *Wc = W + (long)sj * CS;
        const int nd = (V.n - 1 - j) < Bc ? (V.n - 1 - j) : Bc;
        const int cn = j + Bc + 1, scn = sj == 0 ? NS - 1 : sj - 1; // slot of column cn = (j + NS - 1) % NS
        // next column: loads issued now, stored to the free LDS slot at the end of the step (global latency off the critical path)
        double pf[4] = {0, 0, 0, 0}, pfb = 0;
        if (cn < V.n) {
#pragma unroll
            for (int u = 0; u < 4; u++) if (cn < plim[u]) pf[u] = V.A[pk[u] + cn * cstep];
            if (tid < 6) pfb = band_b(V, cn, tid);
        }
        // L_jj = chol(A_jj), its inverse and y_j = L_jj^-1 b_j: every lane of WAVE 0 computes them in its own registers from LDS
        // broadcasts (six v_rsq_f64 + two Newton steps, ~150 fused multiply-adds) and goes straight on to its panel rows, which
        // with the explicit inverse are independent dot products instead of a 16-deep dependent chain -- no cross-lane traffic and no
        // barrier between the pivot and the panel (a shuffle-based version on six lanes cost 0.86 us per column).  The other waves wait.
        BP(0);
        double Lr[6][6], Mi[6][6], invs[6], yv[6];
        if (tid < 64) {
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c <= r; c++) Lr[r][c] = Wc[r * 6 + c];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double d = Lr[c][c];
#pragma unroll
                for (int t = 0; t < c; t++) d = __builtin_fma(-Lr[c][t], Lr[c][t], d);
                if (!(d > 0)) { fail = true; d = 1; }
                double inv = __builtin_amdgcn_rsq(d);
                inv = inv * (1.5 - (0.5 * d) * (inv * inv));
                inv = inv * (1.5 - (0.5 * d) * (inv * inv));
                invs[c] = inv;
                Lr[c][c] = d * inv;
#pragma unroll
                for (int r = c + 1; r < 6; r++) {
                    double v = Lr[r][c];
#pragma unroll
                    for (int t = 0; t < c; t++) v = __builtin_fma(-Lr[r][t], Lr[c][t], v);
                    Lr[r][c] = v * inv;
                }
            }
#pragma unroll
            for (int c = 0; c < 6; c++) { // Mi = L_jj^-1 (lower triangular), column by column
                Mi[c][c] = invs[c];
#pragma unroll
                for (int r = c + 1; r < 6; r++) {
                    double v = 0;
#pragma unroll
                    for (int t = c; t < r; t++) v = __builtin_fma(Lr[r][t], Mi[t][c], v);
                    Mi[r][c] = -invs[r] * v;
                }
            }
#pragma unroll
            for (int t = 0; t < 6; t++) { // y = L_jj^-1 b
                double sacc = 0;
#pragma unroll
                for (int u = 0; u <= t; u++) sacc = __builtin_fma(Mi[t][u], bw[sj * 6 + u], sacc);
                yv[t] = sacc;
            }
            BP(1);
            for (int t = tid; t < nd * 6; t += 64) { // L_d = A_d L_jj^-T = A_d Mi^T, one block row per lane
                double *blk = Wc + (t == tid ? p_off : (1 + t / 6) * 36 + (t % 6) * 6);
                double a[6], row[6];
#pragma unroll
                for (int c = 0; c < 6; c++) a[c] = blk[c];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = 0;
#pragma unroll
                    for (int q = 0; q <= c; q++) v = __builtin_fma(a[q], Mi[c][q], v);
                    row[c] = v;
                }
#pragma unroll
                for (int c = 0; c < 6; c++) blk[c] = row[c];
            }
            if (tid == 0) { // the other waves only need y_j before the barrier
#pragma unroll
                for (int r = 0; r < 6; r++) yj[r] = yv[r];
            }
        }
        BP(2);
        lds_barrier();
        BP(3);
        if (use_mfma) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (ta_di[q] > 20) continue; // this wave has no such tile (uniform over the wave)
                double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                if (ta_di[q] <= nd) { a0 = -Wc[ta_off[q]]; if (k0 < 2) a1 = -Wc[ta_off[q] + 4]; } // K = 6: k0, k0 + 4 < 6
                if (tb_di[q] <= nd) { b0 = Wc[tb_off[q]]; if (k0 < 2) b1 = Wc[tb_off[q] + 4]; }
                band_v4d acc;
                double *tp[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const unsigned x = (unsigned)(sjcs + t_dk[q][g]);
                    tp[g] = W + (int)(min(x, x - (unsigned)NSCS) + (unsigned)t_in[q][g]); // ((sj + dk) mod NS) * CS + offset, no multiply
                    acc[g] = t_di[q][g] <= nd ? *tp[g] : 0.0;
                }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; g++) if (t_di[q][g] <= nd) *tp[g] = acc[g];
            }
            if (wave == NT / 64 - 1 && lane < nd * 6) { // b_{j+d} -= L_d y_j
                const double *Ld = Wc + p_off;
                double v = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) v += Ld[q] * yj[q];
                int sl = sj + p_dd; if (sl >= NS) sl -= NS;
                bw[sl * 6 + p_r] -= v;
            }
        } else {
        const int npair = nd * (nd + 1) / 2;
        for (int t = tid; t < npair * 6 + nd * 6; t += NT) {
            if (t < npair * 6) { // row r of block (j+di, j+dk) -= (row r of L_di) L_dk^T: the row of L_di stays in registers
                const int pr = t / 6, r = t % 6, di = pair_d[pr] >> 8, dk = pair_d[pr] & 255;
                const double *Li = Wc + di * 36 + r * 6, *Lk = Wc + dk * 36;
                double a[6];
#pragma unroll
                for (int q = 0; q < 6; q++) a[q] = Li[q];
                int sl = sj + dk; if (sl >= NS) sl -= NS;
                double *T = W + (long)sl * CS + (di - dk) * 36 + r * 6;
                double tv[6];
#pragma unroll
                for (int c = 0; c < 6; c++) tv[c] = T[c];
#pragma unroll
                for (int c = 0; c < 6; c++) { // fused multiply-adds: the solver's round-off is not part of the parity contract
#pragma unroll
                    for (int q = 0; q < 6; q++) tv[c] = __builtin_fma(-a[q], Lk[c * 6 + q], tv[c]);
                }
#pragma unroll
                for (int c = 0; c < 6; c++) T[c] = tv[c];
            } else { // b_{j+d} -= L_d y_j
                const int u = t - npair * 6, d = 1 + u / 6, r = u % 6;
                const double *Ld = Wc + d * 36 + r * 6;
                double v = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) v += Ld[q] * yj[q];
                int sl = sj + d; if (sl >= NS) sl -= NS;
                bw[sl * 6 + r] -= v;
            }
        }
        }
        BP(6);
        for (int i = 36 + tid; i < (nd + 1) * 36; i += NT) Lf[(long)(V.lf_base + j) * CS + i] = Wc[i]; // the panel; the diagonal block comes from wave 0's registers
        if (tid == 0) { // L_jj, y_j and L_jj^-1 (lower, packed) straight from registers to global memory, while the other waves update
            double *Lo = Lf + (long)(V.lf_base + j) * CS, *yo = ybuf + (long)(V.lf_base + j) * BAND_YB; // buffers this kernel has not read before: no stale L1 line possible
#pragma unroll
            for (int r = 0; r < 6; r++) {
                yo[r] = yv[r];
#pragma unroll
                for (int c = 0; c < 6; c++) { Lo[r * 6 + c] = c <= r ? Lr[r][c] : 0.0; if (c <= r) yo[6 + r * (r + 1) / 2 + c] = Mi[r][c]; }
            }
        }
        BP(7);
        if (cn < V.n) {
            double *dst = W + (long)scn * CS;
#pragma unroll
            for (int u = 0; u < 4; u++) if (tid + u * NT < CS) dst[tid + u * NT] = pf[u];
            if (tid < 6) bw[scn * 6 + tid] = pfb;
        }
        BP(4);
        lds_barrier();
        BP(5);
        if (++sj == NS) sj = 0;
        sjcs = sj * CS;
    }
#ifdef BAND_PROF
    if (tid == 0) printf("band_factor side %d cols %d cycles: top %llu pivot %llu panel %llu bar1 %llu mfma %llu store %llu commit %llu bar2 %llu\n", V.rev, V.ne, pt_[0], pt_[1], pt_[2], pt_[3], pt_[6], pt_[7], pt_[4], pt_[5]);
#endif
    return fail;
}

// L^T x = y for the local columns ne-1 .. 0, last first; xtail (may be NULL when n == ne): solved tail blocks in local order.
// The columns of L come back from global memory in chunks of KC columns through two LDS buffers (the whole workgroup loads chunk
// k+1 into registers while chunk k is used, one barrier per chunk); inside a chunk wave 0 works alone, without barriers: lane =
// part * 6 + c sums L_d^T x_{j+d} over the blocks d = 1 + part, 9 + part, 17 + part, shuffles reduce the 8 parts, lanes 0..5
// finish with L_jj^T by lane broadcasts.  Solved blocks: LDS ring xw; xout in original order.
template <int NT> __device__ __forceinline__ void band_back(const BandView &V, const BandLds &S, const double *Lf, const double *ybuf, const double *xtail, int xtail_rev, double *xout) {
    const int Bc = V.Bc, NB = Bc + 1, NS = Bc + 2, CS = NB * 36, C = V.ne;
    double *W = S.W;
    const int tid = threadIdx.x;
    double *xw = S.bw;
    const int R = V.n - V.ne;
    for (int i = tid; i < R * 6; i += NT) { const int t = i / 6; xw[((V.ne + t) % NS) * 6 + i % 6] = xtail[(long)(xtail_rev ? R - 1 - t : t) * 6 + i % 6]; }
    const int KC = (NS - 1) / 2, CB = KC * CS;           // chunk: KC columns of CS doubles; buffers W[0..CB) and W[CB..2CB), >= CS doubles left for ych
    double *ych = W + 2 * (long)CB;                      // 2 x KC x BAND_YB (y_j, L_jj^-1) -- fits: (NS - 2 KC) * CS >= CS >= 56 * KC for Bc >= 1
    const int nchunk = (C + KC - 1) / KC;
    auto chunk_lo = [&](int ch) { return C - (ch + 1) * KC < 0 ? 0 : C - (ch + 1) * KC; }; // chunk ch covers columns [lo, hi)
    auto chunk_hi = [&](int ch) { return C - ch * KC; };
    constexpr int UPF = (9216 + NT - 1) / NT; // doubles per thread per chunk: Bc <= 20 -> CB <= 8316 (checked by the host)
    double pf[UPF], pfy = 0;
    auto issue = [&](int ch) {
        const int lo = chunk_lo(ch), n = (chunk_hi(ch) - lo) * CS;
#pragma unroll
        for (int u = 0; u < UPF; u++) { const int i = tid + u * NT; pf[u] = i < n ? __builtin_nontemporal_load(Lf + (long)(V.lf_base + lo) * CS + i) : 0.0; }
        const int ny = (chunk_hi(ch) - lo) * BAND_YB;
        pfy = tid < ny ? __builtin_nontemporal_load(ybuf + (long)(V.lf_base + lo) * BAND_YB + tid) : 0.0;
    };
    auto commit = [&](int ch) {
        const int lo = chunk_lo(ch), n = (chunk_hi(ch) - lo) * CS;
        double *dst = W + (long)(ch & 1) * CB;
#pragma unroll
        for (int u = 0; u < UPF; u++) { const int i = tid + u * NT; if (i < n) dst[i] = pf[u]; }
        const int ny = (chunk_hi(ch) - lo) * BAND_YB;
        if (tid < ny) ych[(ch & 1) * KC * BAND_YB + tid] = pfy;
    };
    if (nchunk > 0) { issue(0); commit(0); }
    lds_barrier();
    const int c = (tid >> 3) < 6 ? (tid >> 3) : 0, pt = tid & 7; // wave 0: lane = c * 8 + part, lanes 48..63 idle
    const bool work = tid < 48, last = work && pt == 7;           // lane c*8+7 ends up with the sum over the 8 parts
    for (int ch = 0; ch < nchunk; ch++) {
        if (ch + 1 < nchunk) issue(ch + 1);
        if (tid < 64) {
            const int lo = chunk_lo(ch);
            const double *Lb0 = W + (long)(ch & 1) * CB, *yb0 = ych + (ch & 1) * KC * BAND_YB;
            int sj = (chunk_hi(ch) - 1) % NS; // slot of column j in the ring of solved blocks, kept incrementally
            for (int j = chunk_hi(ch) - 1; j >= lo; j--) {
                const int nd = (V.n - 1 - j) < Bc ? (V.n - 1 - j) : Bc;
                const double *Lc = Lb0 + (long)(j - lo) * CS, *yl = yb0 + (j - lo) * BAND_YB;
                // v_c = sum_d (L_d^T x_{j+d})_c: this lane's blocks d = 1 + pt, 9 + pt, 17 + pt, an independent accumulator each
                // (a dependent v_fma_f64 costs 40 cycles), then DPP row shifts over the 8 parts
                double va[3] = {0, 0, 0};
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int d = 1 + pt + 8 * u;
                    if (work && d <= nd) {
                        int sl = sj + d; if (sl >= NS) sl -= NS; if (sl >= NS) sl -= NS;
                        const double *Ld = Lc + d * 36 + c, *xd = xw + sl * 6;
#pragma unroll
                        for (int q = 0; q < 6; q++) va[u] = __builtin_fma(Ld[q * 6], xd[q], va[u]);
                    }
                }
                double v = (va[0] + va[1]) + va[2];
                v += dpp_f64<0x114>(v); v += dpp_f64<0x112>(v); v += dpp_f64<0x111>(v);
                // x_j = L_jj^-T (y_j - v) with the explicit inverse from the factorisation: six independent broadcasts and a 3 + 3
                // multiply-add tree per lane instead of a six-step substitution chain
                const double sv = yl[c] - v;
                double s6[6];
#pragma unroll
                for (int k = 0; k < 6; k++) s6[k] = __shfl(sv, k * 8 + 7);
                double e = 0, o = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double m = k >= c ? yl[6 + k * (k + 1) / 2 + c] : 0.0; // (L^-1)[k][c]
                    if (k & 1) o = __builtin_fma(m, s6[k], o); else e = __builtin_fma(m, s6[k], e);
                }
                const double xv = e + o;
                if (last) { xw[sj * 6 + c] = xv; xout[(long)(V.rev ? V.C - 1 - j : j) * 6 + c] = xv; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sj = sj == 0 ? NS - 1 : sj - 1;
            }
        }
        if (ch + 1 < nchunk) commit(ch + 1);
        lds_barrier();
    }
}

constexpr int BAND_NT = 384; // 6 waves: the nd(nd+1)/2 * 6 + nd * 6 = 324 work items of a trailing update (Bc = 9) fit one pass
// one workgroup, whole chain (short systems, or CUBESLAM_BA_SOLVER=band1).  rhs: in b (6C), out x; ybuf: 28C scratch
__global__ void __launch_bounds__(BAND_NT) ba_band_chol(int C, int Bc, const double *A, double *Lf, double *rhs, double *ybuf, int *status) {
    extern __shared__ double sh[];
    const BandLds S = band_lds(sh, Bc);
    const BandView V{A, rhs, C, Bc, C, C, 0, 0};
    const bool fail = band_factor<BAND_NT>(V, S, Lf, ybuf);
    __syncthreads(); // every store to Lf / ybuf has completed
    band_back<BAND_NT>(V, S, Lf, ybuf, nullptr, 0, rhs);
    if (threadIdx.x == 0 && fail) *status = 1;
}
__device__ __forceinline__ BandView band_side(int side, int C, int Bc, const double *A, const double *rhs) { // side 0: columns [0, neF); side 1: the other end
    const int R = Bc, neF = (C - R) / 2, neB = C - R - neF;
    return side == 0 ? BandView{A, rhs, C, Bc, neF + R, neF, 0, 0} : BandView{A, rhs, C, Bc, neB + R, neB, 1, neF};
}
// mid: per side R columns of CS doubles (the updated tail columns, local order) followed by 6R right-hand-side entries
__global__ void __launch_bounds__(BAND_NT) ba_band_twist_factor(int C, int Bc, const double *A, const double *rhs, double *Lf, double *ybuf, double *mid, int *status) {
    extern __shared__ double sh[];
    const BandLds S = band_lds(sh, Bc);
    const BandView V = band_side(blockIdx.x, C, Bc, A, rhs);
    const bool fail = band_factor<BAND_NT>(V, S, Lf, ybuf);
    const int NS = Bc + 2, CS = (Bc + 1) * 36, R = V.n - V.ne;
    double *m = mid + (long)blockIdx.x * ((long)R * CS + R * 6);
    for (int i = threadIdx.x; i < R * CS; i += BAND_NT) m[i] = S.W[(long)((V.ne + i / CS) % NS) * CS + i % CS];
    for (int i = threadIdx.x; i < R * 6; i += BAND_NT) m[(long)R * CS + i] = S.bw[((V.ne + i / 6) % NS) * 6 + i % 6];
    if (threadIdx.x == 0 && fail) *status = 1;
}
// middle group: M = tail(forward) + reversed tail(backward), dense Cholesky in LDS (N = 6R <= 120), two triangular solves.  xm: 6R
__global__ void __launch_bounds__(256) ba_band_mid(int C, int Bc, const double *mid, double *xm, double *xout, int *status) {
    extern __shared__ double sh[];
    const int R = Bc, N = 6 * R, CS = (Bc + 1) * 36, tid = threadIdx.x;
    double *M = sh, *bv = M + (long)N * N;
    const double *m0 = mid, *m1 = mid + ((long)R * CS + R * 6);
    for (int i = tid; i < N * N; i += 256) M[i] = 0;
    __syncthreads();
    for (int i = tid; i < R * R * 36; i += 256) { // forward side: block (t+d, t) as stored
        const int t = i / (R * 36), d = (i / 36) % R, e = i % 36, r = e / 6, c = e % 6;
        if (t + d < R) M[(long)((t + d) * 6 + r) * N + t * 6 + c] = m0[(long)t * CS + d * 36 + e];
    }
    __syncthreads();
    for (int i = tid; i < R * R * 36; i += 256) { // backward side: its block (t+d, t) is the transposed original block (R-1-t, R-1-t-d)
        const int t = i / (R * 36), d = (i / 36) % R, e = i % 36, r = e / 6, c = e % 6;
        if (t + d < R && (d > 0 || r >= c)) { // diagonal blocks: only their lower triangle is maintained by the factor kernels
            const int bb = R - 1 - t, aa = bb - d;
            if (d > 0) M[(long)(bb * 6 + c) * N + aa * 6 + r] += m1[(long)t * CS + d * 36 + e];
            else M[(long)(bb * 6 + r) * N + bb * 6 + c] += m1[(long)t * CS + e];
        }
    }
    for (int i = tid; i < N; i += 256) bv[i] = m0[(long)R * CS + i] + m1[(long)R * CS + (R - 1 - i / 6) * 6 + i % 6];
    __syncthreads();
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    for (int k = 0; k < N; k++) { // right-looking Cholesky, lower triangle
        if (tid == 0) { double d = M[(long)k * N + k]; if (!(d > 0)) { s_fail = 1; d = 1; } M[(long)k * N + k] = sqrt(d); }
        __syncthreads();
        const double dk = M[(long)k * N + k];
        for (int i = k + 1 + tid; i < N; i += 256) M[(long)i * N + k] /= dk;
        __syncthreads();
        const int m = N - 1 - k; // trailing rows k+1..N-1: element (i, j), j <= i
        for (int e = tid; e < m * m; e += 256) { const int i = k + 1 + e / m, j = k + 1 + e % m; if (j <= i) M[(long)i * N + j] = __builtin_fma(-M[(long)i * N + k], M[(long)j * N + k], M[(long)i * N + j]); }
        __syncthreads();
    }
    for (int k = 0; k < N; k++) { // L y = b
        if (tid == 0) bv[k] /= M[(long)k * N + k];
        __syncthreads();
        const double yk = bv[k];
        for (int i = k + 1 + tid; i < N; i += 256) bv[i] = __builtin_fma(-M[(long)i * N + k], yk, bv[i]);
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; k--) { // L^T x = y
        if (tid == 0) bv[k] /= M[(long)k * N + k];
        __syncthreads();
        const double xk = bv[k];
        for (int i = tid; i < k; i += 256) bv[i] = __builtin_fma(-M[(long)k * N + i], xk, bv[i]);
        __syncthreads();
    }
    const int neF = (C - R) / 2;
    for (int i = tid; i < N; i += 256) { xm[i] = bv[i]; xout[(long)neF * 6 + i] = bv[i]; }
    if (tid == 0 && s_fail) *status = 1;
}
__global__ void __launch_bounds__(BAND_NT) ba_band_twist_back(int C, int Bc, const double *Lf, const double *ybuf, const double *xm, double *xout) {
    extern __shared__ double sh[];
    const BandLds S = band_lds(sh, Bc);
    const BandView V = band_side(blockIdx.x, C, Bc, nullptr, nullptr);
    band_back<BAND_NT>(V, S, Lf, ybuf, xm, blockIdx.x, xout);
}

// scatter the (all-reduced) blocks of the reduced system into the factor storage Lb = [P diagonal blocks | off-diagonal
// blocks column by column in elimination order]; slot_dst[s] = destination block (-1: dead), slot_tr[s] = transpose
__global__ void ba_chol_fill(int n_slots, const int *slot_dst, const uint8_t *slot_tr, const double *S, double *Lb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots * 36) return;
    const int s = t / 36, k = t % 36, r = k / 6, c = k % 6;
    if (slot_dst[s] < 0) return;
    Lb[(long)slot_dst[s] * 36 + k] = slot_tr[s] ? S[(long)s * 36 + c * 6 + r] : S[t];
}

// Right-looking sparse block Cholesky in a minimum-degree elimination order (symbolic factorisation on the host:
// col_off/rows = structure of every column of L, upd_tgt = destination block of every update pair).  One workgroup.
__global__ void __launch_bounds__(512) ba_chol_factor(int n, const int *col_off, const int *pair_off, const int *upd_tgt, double *Lb, int *status) {
    extern __shared__ double sh[];
    double *Ljj = sh, *Lcol = sh + 36;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int j = 0; j < n; j++) {
        double *Ajj = Lb + (long)j * 36;
        if (tid == 0) {
            double A[36];
            for (int k = 0; k < 36; k++) A[k] = Ajj[k];
            for (int c = 0; c < 6; c++) {
                double d = A[c * 6 + c];
                for (int t = 0; t < c; t++) d -= A[c * 6 + t] * A[c * 6 + t];
                if (!(d > 0)) { s_fail = 1; d = 1; }
                d = sqrt(d);
                A[c * 6 + c] = d;
                for (int r = c + 1; r < 6; r++) { double v = A[r * 6 + c]; for (int t = 0; t < c; t++) v -= A[r * 6 + t] * A[c * 6 + t]; A[r * 6 + c] = v / d; }
                for (int r = 0; r < c; r++) A[r * 6 + c] = 0;
            }
            for (int k = 0; k < 36; k++) { Ajj[k] = A[k]; Ljj[k] = A[k]; }
        }
        __syncthreads();
        const int c0 = col_off[j], m = col_off[j + 1] - c0;
        for (int t = tid; t < m * 6; t += 512) { // L_ij = A_ij Ljj^-T
            const int bi = t / 6, r = t % 6;
            double *Aij = Lb + ((long)n + c0 + bi) * 36;
            double row[6];
            for (int c = 0; c < 6; c++) { double v = Aij[r * 6 + c]; for (int q = 0; q < c; q++) v -= row[q] * Ljj[c * 6 + q]; row[c] = v / Ljj[c * 6 + c]; }
            for (int c = 0; c < 6; c++) { Aij[r * 6 + c] = row[c]; Lcol[bi * 36 + r * 6 + c] = row[c]; }
        }
        __syncthreads();
        const int npair = m * (m + 1) / 2, p0 = pair_off[j];
        for (int t = tid; t < npair * 36; t += 512) { // A(rows[bi], rows[bk]) -= L_bi L_bk^T, bk <= bi
            const int pr = t / 36, k = t % 36, r = k / 6, c = k % 6;
            int bi = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
            while (bi * (bi + 1) / 2 > pr) bi--;
            while ((bi + 1) * (bi + 2) / 2 <= pr) bi++;
            const int bk = pr - bi * (bi + 1) / 2;
            const double *Li = Lcol + bi * 36 + r * 6, *Lk = Lcol + bk * 36 + c * 6;
            double v = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) v += Li[q] * Lk[q];
            Lb[(long)upd_tgt[p0 + pr] * 36 + k] -= v;
        }
        __syncthreads();
    }
    if (tid == 0) *status = s_fail;
}
// L y = b (column sweep), then L^T x = y (gather per column); xp is the right-hand side in elimination order, in place
__global__ void __launch_bounds__(256) ba_chol_solve(int n, const int *col_off, const int *rows, const double *Lb, double *xp) {
    __shared__ double y[6];
    const int tid = threadIdx.x;
    for (int j = 0; j < n; j++) {
        const double *Ljj = Lb + (long)j * 36;
        if (tid == 0) { for (int r = 0; r < 6; r++) { double v = xp[(long)j * 6 + r]; for (int t = 0; t < r; t++) v -= Ljj[r * 6 + t] * y[t]; y[r] = v / Ljj[r * 6 + r]; } for (int r = 0; r < 6; r++) xp[(long)j * 6 + r] = y[r]; }
        __syncthreads();
        const int c0 = col_off[j], m = col_off[j + 1] - c0;
        for (int t = tid; t < m * 6; t += 256) {
            const int bi = t / 6, r = t % 6;
            const double *Lij = Lb + ((long)n + c0 + bi) * 36 + r * 6;
            double v = 0;
            for (int q = 0; q < 6; q++) v += Lij[q] * y[q];
            xp[(long)rows[c0 + bi] * 6 + r] -= v;
        }
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; j--) {
        const int c0 = col_off[j], m = col_off[j + 1] - c0;
        if (tid < 6) { // y_c = x_j[c] - sum_i (L_ij^T x_i)[c]
            double v = xp[(long)j * 6 + tid];
            for (int bi = 0; bi < m; bi++) {
                const double *Lij = Lb + ((long)n + c0 + bi) * 36, *xi = xp + (long)rows[c0 + bi] * 6;
                for (int q = 0; q < 6; q++) v -= Lij[q * 6 + tid] * xi[q];
            }
            y[tid] = v;
        }
        __syncthreads();
        if (tid == 0) {
            const double *Ljj = Lb + (long)j * 36;
            double x[6];
            for (int r = 5; r >= 0; r--) { double v = y[r]; for (int t = r + 1; t < 6; t++) v -= Ljj[t * 6 + r] * x[t]; x[r] = v / Ljj[r * 6 + r]; }
            for (int r = 0; r < 6; r++) xp[(long)j * 6 + r] = x[r];
        }
        __syncthreads();
    }
}
__global__ void ba_permute(int P, const int *pos, const double *src, double *dst, int forward) { // forward: dst[pos[i]] = src[i]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P * 6) return;
    const int i = t / 6, k = t % 6;
    if (forward) dst[(long)pos[i] * 6 + k] = src[t]; else dst[t] = src[(long)pos[i] * 6 + k];
}

__device__ __forceinline__ void ba_backsub_body(const Params &G, const int bid) { // x_l = Dinv (b_l - B^T x_p), block_solver.hpp:459-485
    const int li = G.lm_b + bid * 256 + threadIdx.x;
    if (li >= G.lm_e) return;
    double cl[3] = {G.bl[(long)li * 3], G.bl[(long)li * 3 + 1], G.bl[(long)li * 3 + 2]};
    for (int o = G.lm_off[li]; o < G.lm_off[li + 1]; o++) {
        const int pi = G.cam_idx[G.o_cam[o]];
        if (pi < 0) continue;
        const double *Bi = G.Hpl + (long)o * 18, *xpp = G.x + (long)pi * 6;
        for (int c = 0; c < 3; c++) { double s = 0; for (int a = 0; a < 6; a++) s += Bi[a * 3 + c] * xpp[a]; cl[c] -= s; }
    }
    const double *Di = G.Dinv + (long)li * 9;
    for (int a = 0; a < 3; a++) G.x[(long)G.P * 6 + (long)li * 3 + a] = (Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1]) + Di[a * 3 + 2] * cl[2];
}
__global__ void __launch_bounds__(256) ba_backsub(Params G) { ba_backsub_body(G, (int)blockIdx.x); }

// sum_j x_j (lambda x_j + b_j) (computeScale :182-189): landmark part of this rank + pose part (b_p partial; lambda term once)
__device__ __forceinline__ void ba_scale_body(const Params &G, double lambda, double *partials, const int bid) {
    const long t = (long)bid * 256 + threadIdx.x, nP = (long)G.P * 6, nL = (long)(G.lm_e - G.lm_b) * 3;
    double v = 0;
    if (t < nP) v = G.x[t] * ((G.pose_edges ? lambda * G.x[t] : 0.0) + G.bp[t]);
    else if (t < nP + nL) { const long k = (long)G.lm_b * 3 + (t - nP); v = G.x[nP + k] * (lambda * G.x[nP + k] + G.bl[k]); }
    block_sum_store(v, partials, bid);
}
__global__ void __launch_bounds__(256) ba_scale(Params G, double lambda, double *partials) { ba_scale_body(G, lambda, partials, (int)blockIdx.x); }
__device__ __forceinline__ void ba_update_body(const Params &G, const int bid) { // SparseOptimizer::update -> oplus per vertex
    const int t = bid * 256 + threadIdx.x;
    if (t < G.n_cams) {
        const int pi = G.cam_idx[t];
        if (pi < 0) return;
        SE3 T = se3_mul(se3_exp(G.x + (long)pi * 6), se3_load(G.cam + (long)t * 7)); // VertexSE3Expmap::oplusImpl types_six_dof_expmap.h:73-91
        se3_store(T, G.cam + (long)t * 7);
    } else if (t < G.n_cams + G.n_cub) {
        const int i = t - G.n_cams, pi = G.P - G.n_cub + i;
        Cuboid c = cuboid_oplus(load_cuboid(G, i), G.x + (long)pi * 6, G.cub_flags[i], G.cub_scale + (long)i * 3);
        se3_store(c.pose, G.cub + (long)i * 7);
    } else {
        const long k = (long)(t - G.n_cams - G.n_cub);
        if (k < (long)(G.lm_e - G.lm_b) * 3) { const long q = (long)G.lm_b * 3 + k; G.pts[q] += G.x[(long)G.P * 6 + q]; } // VertexSBAPointXYZ::oplusImpl
    }
}
__global__ void __launch_bounds__(256) ba_update(Params G) { ba_update_body(G, (int)blockIdx.x); }
__global__ void __launch_bounds__(256) ba_back_both(Params G, int nb_l, int C, int Q, const int *cq_off, const int *cq_list, const double *S, const double *cubD, const double *cubg) {
    const int bid = (int)blockIdx.x, nb_q = (Q + 255) / 256; // the cuboids' few workgroups first: theirs is the longer chain
    if (bid < nb_q) ba_cub_back_body(C, Q, cq_off, cq_list, S, cubD, cubg, G.x, bid * 256 + (int)threadIdx.x);
    else ba_backsub_body(G, bid - nb_q);
}
__global__ void __launch_bounds__(256) ba_scale_update(Params G, double lambda, double *partials, int nb_s) {
    const int bid = (int)blockIdx.x;
    if (bid < nb_s) ba_scale_body(G, lambda, partials, bid);
    else ba_update_body(G, bid - nb_s);
}
__global__ void __launch_bounds__(256) ba_maxdiag(Params G, double *partials) { // max |H_jj| of this rank's landmark blocks (computeLambdaInit)
    const int li = G.lm_b + blockIdx.x * 256 + threadIdx.x;
    double v = 0;
    if (li < G.lm_e) v = fmax(fmax(fabs(G.Hll[(long)li * 9]), fabs(G.Hll[(long)li * 9 + 4])), fabs(G.Hll[(long)li * 9 + 8]));
    __shared__ double s[4];
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}

} // namespace

// ------------------------------------------------------------------------------------------------ 9-dof g2o::cuboid (object_slam's graph)
// class cuboid / VertexCuboid / EdgeSE3Cuboid of object_slam/include/object_slam/g2o_Object.h:23-252, SE3Quat::log of
// Thirdparty/g2o/g2o/types/se3quat.h:229-266.  A cuboid is 10 doubles: [t, qx qy qz qw, half scale].
HD void se3_log(const SE3 &T, double *res) {
    double R[3][3]; qtoR(T.r, R);
    const double d = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1);
    const double dR[3] = {R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]};
    double om[3], coef;
    if (d > 0.99999) { for (int i = 0; i < 3; i++) om[i] = 0.5 * dR[i]; coef = 1. / 12.; }
    else {
        const double theta = acos(d);
        for (int i = 0; i < 3; i++) om[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
        coef = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    }
    const double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double O2[3][3];
    mat3mul(O, O, O2);
    for (int i = 0; i < 3; i++) {
        double v[3];
        for (int j = 0; j < 3; j++) v[j] = ((i == j ? 1.0 : 0.0) - 0.5 * O[i][j]) + coef * O2[i][j];
        res[i] = om[i];
        res[i + 3] = (v[0] * T.t[0] + v[1] * T.t[1]) + v[2] * T.t[2];
    }
}
struct Cub9 { SE3 pose; double scale[3]; };
HD Cub9 cub9_load(const double *v) { Cub9 c; c.pose = se3_load(v); c.scale[0] = v[7]; c.scale[1] = v[8]; c.scale[2] = v[9]; return c; }
HD void cub9_store(const Cub9 &c, double *v) { se3_store(c.pose, v); v[7] = c.scale[0]; v[8] = c.scale[1]; v[9] = c.scale[2]; }
HD Cub9 cub9_oplus(Cub9 c, const double *upd) { // VertexCuboid::oplusImpl -> cuboid::exp_update
    Cub9 r;
    normalize_rotation(c.pose);
    r.pose = se3_mul(c.pose, se3_exp(upd));
    for (int k = 0; k < 3; k++) r.scale[k] = c.scale[k] + upd[6 + k];
    return r;
}
HD void cub9_edge_error(const SE3 &Tcw, Cub9 g, Cub9 m, double *err) { // EdgeSE3Cuboid::computeError
    const double PI_ = 3.14159265358979323846;
    normalize_rotation(g.pose); normalize_rotation(m.pose);
    Cub9 e;
    e.pose = se3_mul(se3_inv(Tcw), m.pose); // transform_from
    double best = 0; int lbl = -1;
    for (int i = 0; i < 4; i++) { // min_log_error over rotate_cuboid(-90, 0, 90, 180 degrees)
        const double yaw_angle = (double)(i - 1) * PI_ / 2.0;
        SE3 rot; rot.r = Quat{0, 0, sin(yaw_angle * 0.5), cos(yaw_angle * 0.5)}; rot.t[0] = rot.t[1] = rot.t[2] = 0;
        normalize_rotation(rot);
        Cub9 rc; rc.pose = se3_mul(e.pose, rot);
        const bool swp = (yaw_angle == PI_ / 2.0) || (yaw_angle == -PI_ / 2.0) || (yaw_angle == 3 * PI_ / 2.0);
        rc.scale[0] = swp ? m.scale[1] : m.scale[0]; rc.scale[1] = swp ? m.scale[0] : m.scale[1]; rc.scale[2] = m.scale[2];
        double ei[9];
        se3_log(se3_mul(se3_inv(rc.pose), g.pose), ei); // cube_log_error
        for (int k = 0; k < 3; k++) ei[6 + k] = g.scale[k] - rc.scale[k];
        double nn = 0; for (int k = 0; k < 9; k++) nn += ei[k] * ei[k];
        nn = sqrt(nn);
        if (lbl < 0 || nn < best) { best = nn; lbl = i; for (int k = 0; k < 9; k++) err[k] = ei[k]; }
    }
}
__global__ void __launch_bounds__(64) cub9_oplus_kernel(int n, const double *cub, const double *upd, double *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    cub9_store(cub9_oplus(cub9_load(cub + (long)i * 10), upd + (long)i * 9), out + (long)i * 10);
}
// thread per (edge, column): column 0 = the error itself, 1..6 camera dofs, 7..15 cuboid dofs (central differences, delta 1e-9)
__global__ void __launch_bounds__(64) cub9_edge_kernel(int n, int with_jac, const double *cam_Tcw, const double *cub_global, const double *cub_meas, double *err, double *Jcam,
                                                       double *Jcub) {
    const int ncol = with_jac ? 16 : 1;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n * ncol) return;
    const int i = t / ncol, col = t % ncol;
    SE3 T = se3_load(cam_Tcw + (long)i * 7);
    normalize_rotation(T);
    const Cub9 G = cub9_load(cub_global + (long)i * 10), M = cub9_load(cub_meas + (long)i * 10);
    if (col == 0) { cub9_edge_error(T, G, M, err + (long)i * 9); return; }
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    double e1[9], e2[9];
    if (col <= 6) {
        double add[6] = {0, 0, 0, 0, 0, 0};
        add[col - 1] = delta; cub9_edge_error(se3_mul(se3_exp(add), T), G, M, e1);
        add[col - 1] = -delta; cub9_edge_error(se3_mul(se3_exp(add), T), G, M, e2);
        for (int k = 0; k < 9; k++) Jcam[(long)i * 54 + k * 6 + (col - 1)] = scalar * (e1[k] - e2[k]);
    } else {
        double add[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        add[col - 7] = delta; cub9_edge_error(T, cub9_oplus(G, add), M, e1);
        add[col - 7] = -delta; cub9_edge_error(T, cub9_oplus(G, add), M, e2);
        for (int k = 0; k < 9; k++) Jcub[(long)i * 81 + k * 9 + (col - 7)] = scalar * (e1[k] - e2[k]);
    }
}

// ------------------------------------------------------------------------------------------------ Optimizer::PoseOptimization
// One workgroup per frame runs the whole routine (Optimizer.cc:253-472): 4 rounds x up to 10 Levenberg-Marquardt iterations of a
// single 6-dof pose over the frame's map-point matches, inlier / outlier re-classification after every round.  Edges are strided
// over the 256 threads; the 6x6 normal equations and chi2 are reduced in a fixed order (thread partials in edge order, then a
// shuffle tree, then the four wave results); thread 0 solves the damped system and every LM decision is broadcast through LDS.
struct PoseFrame { int e0, e1; double fx, fy, cx, cy, bf; };
__device__ __forceinline__ void pose_edge_eval(const SE3 &T, const double *Xw, const double *ob, const PoseFrame &F, double *e) {
    double pc[3];
    se3_map(T, Xw, pc);
    if (ob[2] >= 0) { // EdgeStereoSE3ProjectXYZOnlyPose::cam_project (types_six_dof_expmap.cpp:331-338): invz is a float there
        const float invz = (float)(1.0 / pc[2]);
        const double u = pc[0] * invz * F.fx + F.cx;
        e[0] = ob[0] - u;
        e[1] = ob[1] - (pc[1] * invz * F.fy + F.cy);
        e[2] = ob[2] - (u - F.bf * invz);
    } else { // EdgeSE3ProjectXYZOnlyPose::cam_project over project2d (:37-42, 322-328): a division per coordinate
        e[0] = ob[0] - (pc[0] / pc[2] * F.fx + F.cx);
        e[1] = ob[1] - (pc[1] / pc[2] * F.fy + F.cy);
        e[2] = 0.0;
    }
}
__device__ __forceinline__ double pose_edge_chi2(const double *e, double w, bool stereo) {
    return stereo ? ((e[0] * w * e[0] + e[1] * w * e[1]) + e[2] * w * e[2]) : (e[0] * w * e[0] + e[1] * w * e[1]);
}
template <int N> __device__ __forceinline__ void pose_block_reduce(double (&v)[N], double *s_red /* 4 x N */, double *s_out /* N */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    if (lane == 0) for (int k = 0; k < N; k++) s_red[wave * N + k] = v[k];
    __syncthreads();
    if (threadIdx.x < N) s_out[threadIdx.x] = (s_red[threadIdx.x] + s_red[N + threadIdx.x]) + (s_red[2 * N + threadIdx.x] + s_red[3 * N + threadIdx.x]);
    __syncthreads();
}
__global__ void __launch_bounds__(256) pose_opt_kernel(const PoseFrame *frames, const double *Xw, const double *obs, const double *wgt, const double *pose_in, double *pose_out,
                                                       uint8_t *outlier, int *n_inliers, double *err /* per edge x 3 */) {
    __shared__ double s_red[4 * 28], s_sys[28]; // 21 upper-triangle entries of H, 6 of b, 1 chi2
    __shared__ double s_x[6], s_T[7];
    __shared__ int s_flag;
    const PoseFrame F = frames[blockIdx.x];
    const int n = F.e1 - F.e0, tid = threadIdx.x;
    const double *X = Xw + (long)F.e0 * 3, *O = obs + (long)F.e0 * 3, *W = wgt + F.e0;
    double *E = err + (long)F.e0 * 3;
    uint8_t *out = outlier + F.e0;
    for (int i = tid; i < n; i += 256) out[i] = 0;
    SE3 T0 = se3_load(pose_in + (long)blockIdx.x * 7);
    normalize_rotation(T0);
    if (n < 3) { if (tid == 0) { se3_store(T0, pose_out + (long)blockIdx.x * 7); n_inliers[blockIdx.x] = 0; } return; }
    const double dMono = (double)(float)sqrt(5.991), dStereo = (double)(float)sqrt(7.815);
    bool robust = true;
    SE3 T = T0;
    int nBadEdges = 0;
    __syncthreads();
    auto chi2_sum = [&](const SE3 &Tc) -> double { // computeActiveErrors + activeRobustChi2
        double v[1] = {0};
        for (int i = tid; i < n; i += 256) {
            if (out[i]) continue;
            double e[3];
            pose_edge_eval(Tc, X + (long)i * 3, O + (long)i * 3, F, e);
            E[(long)i * 3] = e[0]; E[(long)i * 3 + 1] = e[1]; E[(long)i * 3 + 2] = e[2];
            const bool st = O[(long)i * 3 + 2] >= 0;
            double c = pose_edge_chi2(e, W[i], st);
            if (robust) { const double d = st ? dStereo : dMono, dsqr = d * d; if (c > dsqr) c = 2 * sqrt(c) * d - dsqr; }
            v[0] += c;
        }
        pose_block_reduce<1>(v, s_red, s_sys + 27);
        return s_sys[27];
    };
    for (int round = 0; round < 4; round++) {
        T = T0;
        double lambda = 0, ni = 2;
        int nBad = 0;
        for (int it = 0; it < 10; it++) { // OptimizationAlgorithmLevenberg::solve
            double currentChi = chi2_sum(T);
            const double iniChi = currentChi;
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0;
            for (int i = tid; i < n; i += 256) { // linearizeOplus + constructQuadraticForm
                if (out[i]) continue;
                double pc[3], J[18];
                se3_map(T, X + (long)i * 3, pc);
                const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
                J[0] = x * y * invz_2 * F.fx; J[1] = -(1 + (x * x * invz_2)) * F.fx; J[2] = y * invz * F.fx; J[3] = -invz * F.fx; J[4] = 0; J[5] = x * invz_2 * F.fx;
                J[6] = (1 + y * y * invz_2) * F.fy; J[7] = -x * y * invz_2 * F.fy; J[8] = -x * invz * F.fy; J[9] = 0; J[10] = -invz * F.fy; J[11] = y * invz_2 * F.fy;
                const bool st = O[(long)i * 3 + 2] >= 0;
                if (st) { J[12] = J[0] - F.bf * y * invz_2; J[13] = J[1] + F.bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - F.bf * invz_2; }
                else { for (int k = 12; k < 18; k++) J[k] = 0; }
                const double e[3] = {E[(long)i * 3], E[(long)i * 3 + 1], E[(long)i * 3 + 2]}, w = W[i];
                double rw = 1.0;
                if (robust) { const double c = pose_edge_chi2(e, w, st), d = st ? dStereo : dMono; if (c > d * d) rw = d / sqrt(c); }
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int c2 = a; c2 < 6; c2++) { acc[k] += (J[a] * (rw * w) * J[c2] + J[6 + a] * (rw * w) * J[6 + c2]) + J[12 + a] * (rw * w) * J[12 + c2]; k++; }
                }
#pragma unroll
                for (int a = 0; a < 6; a++) acc[21 + a] += rw * ((J[a] * (-(w * e[0])) + J[6 + a] * (-(w * e[1]))) + J[12 + a] * (-(w * e[2])));
            }
            pose_block_reduce<27>(acc, s_red, s_sys);
            if (it == 0) { double mx = 0; int k = 0; for (int a = 0; a < 6; a++) { mx = fmax(fabs(s_sys[k]), mx); k += 6 - a; } lambda = 1e-5 * mx; ni = 2; nBad = 0; }
            double rho = 0;
            int qmax = 0;
            do {
                const SE3 backup = T;
                if (tid == 0) { // (H + lambda I) x = b, Cholesky
                    double L[36], y6[6], x6[6] = {0, 0, 0, 0, 0, 0};
                    int k = 0;
                    for (int a = 0; a < 6; a++) for (int c2 = a; c2 < 6; c2++) { L[c2 * 6 + a] = s_sys[k]; L[a * 6 + c2] = s_sys[k]; k++; }
                    for (int a = 0; a < 6; a++) L[a * 6 + a] += lambda;
                    bool ok = true;
                    for (int c2 = 0; c2 < 6 && ok; c2++) {
                        double d = L[c2 * 6 + c2];
                        for (int t = 0; t < c2; t++) d -= L[c2 * 6 + t] * L[c2 * 6 + t];
                        if (!(d > 0)) { ok = false; break; }
                        d = sqrt(d); L[c2 * 6 + c2] = d;
                        for (int r = c2 + 1; r < 6; r++) { double v = L[r * 6 + c2]; for (int t = 0; t < c2; t++) v -= L[r * 6 + t] * L[c2 * 6 + t]; L[r * 6 + c2] = v / d; }
                    }
                    if (ok) {
                        for (int r = 0; r < 6; r++) { double v = s_sys[21 + r]; for (int t = 0; t < r; t++) v -= L[r * 6 + t] * y6[t]; y6[r] = v / L[r * 6 + r]; }
                        for (int r = 5; r >= 0; r--) { double v = y6[r]; for (int t = r + 1; t < 6; t++) v -= L[t * 6 + r] * x6[t]; x6[r] = v / L[r * 6 + r]; }
                    }
                    for (int r = 0; r < 6; r++) s_x[r] = x6[r];
                    s_flag = ok ? 1 : 0;
                    const SE3 Tn = ok ? se3_mul(se3_exp(x6), T) : T; // VertexSE3Expmap::oplusImpl
                    se3_store(Tn, s_T);
                }
                __syncthreads();
                const bool ok2 = s_flag != 0;
                T = se3_load(s_T);
                double tempChi = chi2_sum(T);
                if (!ok2) tempChi = DBL_MAX;
                rho = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += s_x[j] * (lambda * s_x[j] + s_sys[21 + j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2; currentChi = tempChi;
                } else { lambda *= ni; ni *= 2; T = backup; }
                qmax++;
                __syncthreads(); // s_x / s_sys[21..] are rewritten by the next trial
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        // classification (Optimizer.cc:401-431): outliers are re-evaluated at the final estimate, inliers keep the error of the last trial
        double cnt[1] = {0};
        for (int i = tid; i < n; i += 256) {
            double e[3];
            if (out[i]) { pose_edge_eval(T, X + (long)i * 3, O + (long)i * 3, F, e); E[(long)i * 3] = e[0]; E[(long)i * 3 + 1] = e[1]; E[(long)i * 3 + 2] = e[2]; }
            else { e[0] = E[(long)i * 3]; e[1] = E[(long)i * 3 + 1]; e[2] = E[(long)i * 3 + 2]; }
            const bool st = O[(long)i * 3 + 2] >= 0;
            const float chi2 = (float)pose_edge_chi2(e, W[i], st);
            const bool bad = chi2 > (st ? 7.815f : 5.991f);
            out[i] = bad ? 1 : 0;
            cnt[0] += bad ? 1.0 : 0.0;
        }
        pose_block_reduce<1>(cnt, s_red, s_sys + 27);
        nBadEdges = (int)s_sys[27];
        __syncthreads();
        if (round == 2) robust = false;
        if (n < 10) break;
    }
    if (tid == 0) { se3_store(T, pose_out + (long)blockIdx.x * 7); n_inliers[blockIdx.x] = n - nBadEdges; }
}

struct cs_ba {
    Params G{};
    int rank = 0, world = 1, n_slots = 0, max_col = 0, max_part = 0;
    cs_allreduce_fn allreduce = nullptr; void *ar_user = nullptr;
    const volatile unsigned char *stop8 = nullptr; // the caller's bool (setForceStopFlag)
    std::vector<void *> owned;
    int *d_pose_off = nullptr, *d_pose_obs = nullptr, *d_pe_off = nullptr, *d_pe_list = nullptr, *d_slot_off = nullptr, *d_slot_perm = nullptr, *d_slot_dst = nullptr, *d_col_off = nullptr,
        *d_rows = nullptr, *d_pair_off = nullptr, *d_upd_tgt = nullptr, *d_pos = nullptr, *d_status = nullptr;
    int2 *d_trips = nullptr; uint8_t *d_slot_tr = nullptr;
    double *d_reduce = nullptr, *d_band = nullptr, *d_xperm = nullptr, *d_partials = nullptr, *d_scal = nullptr;
    double *d_bak_cam = nullptr, *d_bak_pts = nullptr, *d_bak_cub = nullptr;
    long reduce_len = 0, band_len = 0; // band_len: doubles of the factor storage
    // band path (see ba_band_chol): cuboids eliminated first, cameras block-banded with half-width band_bc
    bool use_band = false, band_twist = false, band_cr = false; int band_C = 0, band_Q = 0, band_bc = 0, band_targets = 0;
    BaCr *cr = nullptr; // nested-dissection solver of the band system (ba_cr.hip)
    int *d_tgt_slot = nullptr, *d_ct_off = nullptr, *d_ct_list = nullptr, *d_cr_off = nullptr, *d_cr_list = nullptr, *d_cq_off = nullptr, *d_cq_list = nullptr;
    uint8_t *d_tgt_tr = nullptr;
    double *d_bandA = nullptr, *d_bandL = nullptr, *d_cubD = nullptr, *d_cubg = nullptr, *d_brhs = nullptr, *d_ybuf = nullptr, *d_mid = nullptr, *d_xmid = nullptr, *d_bw = nullptr;
    std::vector<int> h_slot_dst, h_pos, h_col_off, h_rows; std::vector<uint8_t> h_slot_tr;
    std::vector<int> obs_perm; // sorted-by-landmark position -> caller's observation index
    std::vector<double> h_partials;
    double *h_pin = nullptr; size_t pin_cap = 0; hipEvent_t ev_trial = nullptr; // pinned [status | partials] of a trial and the event behind their copies (cs_ba_optimize)
};

namespace {
template <class T> int dalloc_copy(cs_ctx *ctx, cs_ba *b, T **d, const T *h, size_t n) {
    int r = cs_dalloc(ctx, d, n); if (r) return r;
    b->owned.push_back(*d);
    if (h && n) { r = cs_h2d(ctx, *d, h, n); if (r) return r; }
    return CS_OK;
}
static double sum_partials(cs_ctx *ctx, cs_ba *b, int n, bool is_max = false) {
    b->h_partials.resize((size_t)std::max(n, 1));
    hipMemcpyAsync(b->h_partials.data(), b->d_partials, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    double s = 0;
    for (int i = 0; i < n; i++) s = is_max ? std::max(s, b->h_partials[i]) : s + b->h_partials[i];
    return s;
}
// all-reduce(sum) of n doubles in device memory across the ranks: the caller's callback when one is registered (it may use another stream or
// library, so the context's stream is drained first), else RCCL on the context's own stream (cs_comm_init), which needs no host round trip
static int ba_allreduce(cs_ctx *ctx, cs_ba *b, double *dbuf, long n) {
    if (b->world <= 1) return CS_OK;
    if (b->allreduce) {
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (b->allreduce(b->ar_user, dbuf, n) != 0) { ctx->err = "all-reduce callback failed"; return CS_ERR_HIP; }
        return CS_OK;
    }
    return cs_comm_allreduce_sum_f64(ctx, dbuf, n);
}
// sum of k host scalars across ranks (k <= the size of d_scal: 6 P + world + 64)
static int allreduce_scalars(cs_ctx *ctx, cs_ba *b, double *v, int k) {
    if (b->world <= 1) return CS_OK;
    int r = cs_h2d(ctx, b->d_scal, v, (size_t)k); if (r) return r;
    r = ba_allreduce(ctx, b, b->d_scal, k); if (r) return r;
    r = cs_d2h(ctx, v, b->d_scal, (size_t)k); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}
static int ba_compute_errors(cs_ctx *ctx, cs_ba *b, double *chi2) { // computeActiveErrors + activeRobustChi2
    const Params &G = b->G;
    double chi = 0;
    const int nb1 = (G.o_e - G.o_b + 255) / 256;
    if (nb1 > 0) { CS_LAUNCH(ctx, "ba_err_obs", ba_err_obs, dim3(nb1), dim3(256), 0, G, b->d_partials); chi += sum_partials(ctx, b, nb1); }
    if (G.pose_edges && G.n_cobs + G.n_pc > 0) {
        const int nb2 = (G.n_cobs + G.n_pc + 255) / 256;
        CS_LAUNCH(ctx, "ba_err_pose_edges", ba_err_pose_edges, dim3(nb2), dim3(256), 0, G, b->d_partials);
        chi += sum_partials(ctx, b, nb2);
    }
    int r = allreduce_scalars(ctx, b, &chi, 1); if (r) return r;
    *chi2 = chi;
    return CS_OK;
}
static int ba_build_system(cs_ctx *ctx, cs_ba *b) { // BlockSolver::buildSystem
    const Params &G = b->G;
    const int nl = G.lm_e - G.lm_b;
    const bool edges = G.pose_edges && G.n_cobs + G.n_pc > 0;
    const int n_pose = G.P, n_cols = edges ? (2 * (G.n_cobs * 12 + G.n_pc * 6) + 255) / 256 : 0, n_lm = (nl + 255) / 256;
    static const bool fused = !(getenv("CUBESLAM_BA_FUSED") && atoi(getenv("CUBESLAM_BA_FUSED")) == 0); // (0: a launch per kernel, the cross-check)
    if (fused && n_pose + n_cols + n_lm > 0) CS_LAUNCH(ctx, "ba_build_abc", ba_build_abc, dim3(n_pose + n_cols + n_lm), dim3(256), 0, G, b->d_pose_off, b->d_pose_obs, n_pose, n_cols);
    else {
        if (nl > 0) CS_LAUNCH(ctx, "ba_lin_lm", ba_lin_lm, dim3(n_lm), dim3(256), 0, G);
        if (G.P > 0) CS_LAUNCH(ctx, "ba_lin_pose", ba_lin_pose, dim3(G.P), dim3(256), 0, G, b->d_pose_off, b->d_pose_obs);
        if (edges) CS_LAUNCH(ctx, "ba_num_cols", ba_num_cols, dim3(n_cols), dim3(256), 0, G); // a lane per (edge, column, sign)
    }
    if (edges) CS_LAUNCH(ctx, "ba_lin_pose_edges", ba_lin_pose_edges, dim3((G.P + G.n_cobs + 3) / 4), dim3(256), 0, G, b->d_pe_off, b->d_pe_list); // a wave per pose block / per camera-cuboid edge: adds to what ba_lin_pose left
    return CS_OK;
}
static int ba_schur(cs_ctx *ctx, cs_ba *b, double lambda) { // Schur part of BlockSolver::solve, result in d_reduce = [slots | bschur]
    const Params &G = b->G;
    const int nl = G.lm_e - G.lm_b;
    if (nl > 0) CS_LAUNCH(ctx, "ba_lm_dinv", ba_lm_dinv, dim3((nl + 255) / 256), dim3(256), 0, G, lambda);
    if (G.o_e > G.o_b) CS_LAUNCH(ctx, "ba_schur_bd", ba_schur_bd, dim3((int)(((long)(G.o_e - G.o_b) * 6 + 255) / 256)), dim3(256), 0, G, b->d_bw);
    CS_LAUNCH(ctx, "ba_schur_slots", ba_schur_slots, dim3((b->n_slots + 3) / 4), dim3(256), 0, G, b->n_slots, b->d_slot_perm, b->d_slot_off, b->d_trips, lambda, b->d_reduce);
    CS_LAUNCH(ctx, "ba_schur_b", ba_schur_b, dim3(G.P), dim3(256), 0, G, b->d_pose_off, b->d_pose_obs, b->d_bw, b->d_reduce + (long)b->n_slots * 36);
    return CS_OK;
}
static int ba_solve(cs_ctx *ctx, cs_ba *b, double lambda, bool *ok, bool defer_status = false) { // BlockSolver::solve; defer_status: the caller reads d_status later
    const Params &G = b->G;
    int r = ba_schur(ctx, b, lambda); if (r) return r;
    if (b->world > 1) { // one fused buffer [36 * slots | 6 * P] per LM trial
        ctx->begin("ba_allreduce");
        const int rc = ba_allreduce(ctx, b, b->d_reduce, b->reduce_len);
        ctx->end();
        if (rc != CS_OK) return rc;
    }
    const int nl = G.lm_e - G.lm_b;
    if (b->use_band) {
        const int C = b->band_C, Q = b->band_Q, Bc = b->band_bc;
        const double *S = b->d_reduce, *bs = b->d_reduce + (long)b->n_slots * 36;
        CS_HIP(ctx, hipMemsetAsync(b->d_status, 0, sizeof(int), ctx->stream));
        if (Q > 0) CS_LAUNCH(ctx, "ba_cub_inv", ba_cub_inv, dim3((Q + 63) / 64), dim3(64), 0, C, Q, S, bs, b->d_cubD, b->d_cubg, b->d_status);
        { const int nb_a = (b->band_targets * 36 + 255) / 256, nb_r = (C * 6 + 255) / 256;
          CS_LAUNCH(ctx, "ba_band_assemble", ba_band_both, dim3(nb_a + nb_r), dim3(256), 0, nb_a, b->band_targets, b->d_tgt_slot, b->d_tgt_tr, b->d_ct_off, b->d_ct_list, S, b->d_cubD, b->d_bandA,
                    C, b->d_cr_off, b->d_cr_list, bs, b->d_cubg, b->d_brhs); }
        const size_t lds = band_lds_bytes(Bc);
        if (b->band_cr) { // nested dissection over super-blocks of Bc cameras: log2(C / Bc) parallel levels (ba_cr.hip)
            r = ba_cr_solve(ctx, &b->cr, C, Bc, b->d_bandA, b->d_brhs, G.x, b->d_status); if (r) return r;
        } else if (b->band_twist) { // both ends at once (see the comment above band_factor)
            const size_t lds_mid = sizeof(double) * ((size_t)36 * Bc * Bc + 6 * (size_t)Bc);
            if (lds > 64 * 1024) {
                CS_HIP(ctx, hipFuncSetAttribute((const void *)ba_band_twist_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CS_HIP(ctx, hipFuncSetAttribute((const void *)ba_band_twist_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (lds_mid > 64 * 1024) CS_HIP(ctx, hipFuncSetAttribute((const void *)ba_band_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid));
            CS_LAUNCH(ctx, "ba_band_twist_factor", ba_band_twist_factor, dim3(2), dim3(BAND_NT), lds, C, Bc, b->d_bandA, b->d_brhs, b->d_bandL, b->d_ybuf, b->d_mid, b->d_status);
            CS_LAUNCH(ctx, "ba_band_mid", ba_band_mid, dim3(1), dim3(256), lds_mid, C, Bc, b->d_mid, b->d_xmid, G.x, b->d_status);
            CS_LAUNCH(ctx, "ba_band_twist_back", ba_band_twist_back, dim3(2), dim3(BAND_NT), lds, C, Bc, b->d_bandL, b->d_ybuf, b->d_xmid, G.x);
        } else {
            if (lds > 64 * 1024) CS_HIP(ctx, hipFuncSetAttribute((const void *)ba_band_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CS_LAUNCH(ctx, "ba_band_chol", ba_band_chol, dim3(1), dim3(BAND_NT), lds, C, Bc, b->d_bandA, b->d_bandL, b->d_brhs, b->d_ybuf, b->d_status);
        }
        if (!b->band_twist && !b->band_cr) CS_HIP(ctx, hipMemcpyAsync(G.x, b->d_brhs, sizeof(double) * (size_t)C * 6, hipMemcpyDeviceToDevice, ctx->stream));
        if (nl > 0 || Q > 0) CS_LAUNCH(ctx, "ba_backsub", ba_back_both, dim3((nl + 255) / 256 + (Q + 255) / 256), dim3(256), 0, G, (nl + 255) / 256, C, Q, b->d_cq_off, b->d_cq_list, S, b->d_cubD, b->d_cubg);
        if (defer_status) return CS_OK;
        int status = 0;
        r = cs_d2h(ctx, &status, b->d_status, 1); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *ok = status == 0;
        return CS_OK;
    }
    CS_HIP(ctx, hipMemsetAsync(b->d_band, 0, sizeof(double) * (size_t)b->band_len, ctx->stream));
    CS_LAUNCH(ctx, "ba_chol_fill", ba_chol_fill, dim3((b->n_slots * 36 + 255) / 256), dim3(256), 0, b->n_slots, b->d_slot_dst, b->d_slot_tr, b->d_reduce, b->d_band);
    CS_LAUNCH(ctx, "ba_chol_factor", ba_chol_factor, dim3(1), dim3(512), sizeof(double) * (36 + (size_t)std::max(b->max_col, 1) * 36), G.P, b->d_col_off, b->d_pair_off,
              b->d_upd_tgt, b->d_band, b->d_status);
    CS_LAUNCH(ctx, "ba_permute", ba_permute, dim3((G.P * 6 + 255) / 256), dim3(256), 0, G.P, b->d_pos, b->d_reduce + (long)b->n_slots * 36, b->d_xperm, 1);
    CS_LAUNCH(ctx, "ba_chol_solve", ba_chol_solve, dim3(1), dim3(256), 0, G.P, b->d_col_off, b->d_rows, b->d_band, b->d_xperm);
    CS_LAUNCH(ctx, "ba_permute", ba_permute, dim3((G.P * 6 + 255) / 256), dim3(256), 0, G.P, b->d_pos, b->d_xperm, G.x, 0);
    if (nl > 0) CS_LAUNCH(ctx, "ba_backsub", ba_backsub, dim3((nl + 255) / 256), dim3(256), 0, G);
    if (defer_status) return CS_OK;
    int status = 0;
    r = cs_d2h(ctx, &status, b->d_status, 1); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *ok = status == 0;
    return CS_OK;
}
} // namespace

extern "C" {

void cs_ba_destroy(cs_ctx *ctx, cs_ba *b) {
    if (!b) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    for (void *p : b->owned) if (p) hipFree(p);
    if (b->h_pin) hipHostFree(b->h_pin);
    if (b->ev_trial) hipEventDestroy(b->ev_trial);
    ba_cr_destroy(b->cr);
    delete b;
}
int cs_ba_set_allreduce(cs_ba *b, cs_allreduce_fn fn, void *user) {
    if (!b) return CS_ERR_BAD_ARG;
    b->allreduce = fn; b->ar_user = user;
    return CS_OK;
}

int cs_ba_create(cs_ctx *ctx, const cs_ba_problem *p, int rank, int world, cs_ba **out) {
    if (!ctx || !p || !out || world < 1 || rank < 0 || rank >= world || p->n_cams < 1 || p->n_points < 0 || p->n_cuboids < 0 || p->n_obs < 0) return CS_ERR_BAD_ARG;
    *out = nullptr;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_ba *b = new (std::nothrow) cs_ba();
    if (!b) return CS_ERR_NOMEM;
    b->rank = rank; b->world = world;
    Params &G = b->G;
    G.n_cams = p->n_cams; G.L = p->n_points; G.n_cub = p->n_cuboids; G.n_obs = p->n_obs; G.n_cobs = p->n_cobs; G.n_pc = p->n_pc;
    G.fx = p->fx; G.fy = p->fy; G.cx = p->cx; G.cy = p->cy; G.huber_mono = p->huber_mono; G.huber_obj = p->huber_obj; G.bf = p->bf; G.huber_stereo = p->huber_stereo; G.margin_ratio = p->max_outside_margin_ratio;
    for (int i = 0; i < 9; i++) G.K[i] = p->K[i];
    G.pose_edges = rank == 0; // camera-cuboid and point-cuboid edges (and setLambda on the pose diagonal) live on rank 0
    // index mapping: non-fixed cameras, then cuboids (sparse_optimizer.cpp:166-190: non-marginalised first, by vertex id)
    std::vector<int> cam_idx(p->n_cams, -1);
    int P = 0;
    for (int i = 0; i < p->n_cams; i++) if (!p->cam_fixed[i]) cam_idx[i] = P++;
    const int first_cub = P;
    P += p->n_cuboids;
    G.P = P;
    if (P < 1) { delete b; return CS_ERR_BAD_ARG; }
    // observations sorted by landmark (stable): CSR by landmark
    for (int o = 0; o < p->n_obs; o++) if (p->obs_point[o] < 0 || p->obs_point[o] >= p->n_points || p->obs_cam[o] < 0 || p->obs_cam[o] >= p->n_cams) { delete b; return CS_ERR_BAD_ARG; }
    b->obs_perm.resize(p->n_obs);
    for (int o = 0; o < p->n_obs; o++) b->obs_perm[o] = o;
    std::stable_sort(b->obs_perm.begin(), b->obs_perm.end(), [&](int a, int c) { return p->obs_point[a] < p->obs_point[c]; });
    std::vector<int> o_cam(p->n_obs), o_pt(p->n_obs), lm_off(p->n_points + 1, 0);
    std::vector<double> o_uv((size_t)p->n_obs * 2), o_w(p->n_obs), o_ur(p->obs_ur ? p->n_obs : 0);
    for (int q = 0; q < p->n_obs; q++) {
        const int o = b->obs_perm[q];
        o_cam[q] = p->obs_cam[o]; o_pt[q] = p->obs_point[o]; o_uv[q * 2] = p->obs_uv[o * 2]; o_uv[q * 2 + 1] = p->obs_uv[o * 2 + 1]; o_w[q] = p->obs_inv_sigma2[o]; if (p->obs_ur) o_ur[q] = p->obs_ur[o];
        lm_off[o_pt[q] + 1]++;
    }
    for (int l = 0; l < p->n_points; l++) lm_off[l + 1] += lm_off[l];
    G.lm_b = (int)((long)p->n_points * rank / world); G.lm_e = (int)((long)p->n_points * (rank + 1) / world);
    G.o_b = lm_off[G.lm_b]; G.o_e = lm_off[G.lm_e];
    // per pose block: this rank's observations (ascending)
    std::vector<int> pose_off(P + 1, 0), pose_obs;
    for (int q = G.o_b; q < G.o_e; q++) if (cam_idx[o_cam[q]] >= 0) pose_off[cam_idx[o_cam[q]] + 1]++;
    for (int i = 0; i < P; i++) pose_off[i + 1] += pose_off[i];
    pose_obs.resize(std::max(pose_off[P], 1));
    { std::vector<int> pos(pose_off.begin(), pose_off.end() - 1); for (int q = G.o_b; q < G.o_e; q++) { int pi = cam_idx[o_cam[q]]; if (pi >= 0) pose_obs[pos[pi]++] = q; } }
    // pose-edge incidence (edge order: camera-cuboid edges, then point-cuboid edges)
    std::vector<std::vector<int>> pe(P);
    for (int o = 0; o < p->n_cobs; o++) {
        if (p->cobs_cam[o] < 0 || p->cobs_cam[o] >= p->n_cams || p->cobs_cuboid[o] < 0 || p->cobs_cuboid[o] >= p->n_cuboids) { delete b; return CS_ERR_BAD_ARG; }
        if (cam_idx[p->cobs_cam[o]] >= 0) pe[cam_idx[p->cobs_cam[o]]].push_back((o << 2) | 0);
        pe[first_cub + p->cobs_cuboid[o]].push_back((o << 2) | 1);
    }
    for (int o = 0; o < p->n_pc; o++) pe[first_cub + p->pc_cuboid[o]].push_back((o << 2) | 2);
    std::vector<int> pe_off(P + 1, 0), pe_list;
    for (int i = 0; i < P; i++) { pe_off[i + 1] = pe_off[i] + (int)pe[i].size(); pe_list.insert(pe_list.end(), pe[i].begin(), pe[i].end()); }
    if (pe_list.empty()) pe_list.push_back(0);
    // Schur pattern (BlockSolver::buildStructure block_solver.hpp:143-295) over ALL landmarks (identical on every rank);
    // the contributing (obs_u, obs_v) pairs only for this rank's landmarks
    std::map<std::pair<int, int>, int> slot_of;
    std::vector<std::pair<int, int>> slot_rc;
    for (int i = 0; i < P; i++) { slot_of[std::make_pair(i, i)] = i; slot_rc.push_back(std::make_pair(i, i)); }
    for (int o = 0; o < p->n_cobs; o++) { // one block per camera-cuboid edge, slot P + o (fixed cameras: unused block)
        const int pi = cam_idx[p->cobs_cam[o]], pj = first_cub + p->cobs_cuboid[o];
        slot_rc.push_back(pi >= 0 ? std::make_pair(pi, pj) : std::make_pair(-1, -1)); // fixed camera: dead block
        if (pi >= 0) slot_of[std::make_pair(pi, pj)] = P + o;
    }
    std::vector<std::vector<int2>> slot_trips;
    slot_trips.resize(slot_rc.size());
    for (int l = 0; l < p->n_points; l++) {
        const bool mine = l >= G.lm_b && l < G.lm_e;
        for (int u = lm_off[l]; u < lm_off[l + 1]; u++) {
            const int i1 = cam_idx[o_cam[u]];
            if (i1 < 0) continue;
            for (int v = lm_off[l]; v < lm_off[l + 1]; v++) {
                const int i2 = cam_idx[o_cam[v]];
                if (i2 < i1 || (i2 == i1 && v != u)) continue;
                auto key = std::make_pair(i1, i2);
                auto it = slot_of.find(key);
                int s;
                if (it == slot_of.end()) { s = (int)slot_rc.size(); slot_of[key] = s; slot_rc.push_back(key); slot_trips.emplace_back(); }
                else s = it->second;
                if (mine) slot_trips[s].push_back(make_int2(u, v));
            }
        }
    }
    b->n_slots = (int)slot_rc.size();
    std::vector<int> slot_off(b->n_slots + 1, 0);
    std::vector<int2> trips;
    for (int s = 0; s < b->n_slots; s++) { slot_off[s + 1] = slot_off[s] + (int)slot_trips[s].size(); trips.insert(trips.end(), slot_trips[s].begin(), slot_trips[s].end()); }
    if (trips.empty()) trips.push_back(make_int2(0, 0));
    // ---- band path plan: cuboids eliminated first, cameras in index order; used when the camera system is narrow-banded
    std::vector<int> tgt_slot, ct_off, ct_list, cr_off, cr_list, cq_off, cq_list;
    std::vector<uint8_t> tgt_tr;
    {
        const int C = first_cub, Q = p->n_cuboids;
        bool okb = C >= 1;
        int Bc = 0;
        for (auto &rc : slot_rc) {
            if (rc.first < 0) continue;
            if (rc.first >= C && rc.second >= C && rc.first != rc.second) okb = false; // two cuboids in one block: not a graph Optimizer.cc builds
            if (rc.first < C && rc.second < C) Bc = std::max(Bc, std::abs(rc.first - rc.second));
        }
        std::vector<std::vector<std::pair<int, int>>> cub_edges(Q); // (camera, slot), alive edges only
        for (int o = 0; o < p->n_cobs; o++) { const int pi = cam_idx[p->cobs_cam[o]]; if (pi >= 0) cub_edges[p->cobs_cuboid[o]].push_back(std::make_pair(pi, P + o)); }
        for (auto &e : cub_edges) { std::stable_sort(e.begin(), e.end()); if (!e.empty()) Bc = std::max(Bc, e.back().first - e.front().first); }
        const char *force = getenv("CUBESLAM_BA_SOLVER"); // "sparse" / "band" / "band1" (band, one workgroup): pick the path (tests run all)
        const int BAND_MAX = 20; // (Bc+2)(Bc+1) blocks of 288 B must fit the 160 KB LDS
        if (force && !strcmp(force, "sparse")) okb = false;
        if (Bc > BAND_MAX) okb = false;
        if (force && (!strcmp(force, "band") || !strcmp(force, "band1")) && !okb) { ctx->err = "CUBESLAM_BA_SOLVER=band: the reduced camera system is not narrow-banded"; delete b; return CS_ERR_CAPACITY; }
        if (okb) {
            b->use_band = true; b->band_C = C; b->band_Q = Q; b->band_bc = Bc; b->band_targets = C * (Bc + 1);
            b->band_twist = Bc >= 1 && C >= 6 * (Bc + 1) && !(force && !strcmp(force, "band1")); // two-sided elimination once the chain is long enough to split
            // nested dissection (ba_cr.hip) once there are enough super-blocks to split; CUBESLAM_BA_SOLVER=band keeps the two-sided chain, =cr forces this
            b->band_cr = ba_cr_supported(C, Bc) && ((force && !strcmp(force, "cr")) || (!force && C >= 16 * Bc));
            tgt_slot.assign((size_t)C * (Bc + 1), -1); tgt_tr.assign((size_t)C * (Bc + 1), 0);
            for (int s2 = 0; s2 < (int)slot_rc.size(); s2++) {
                const int r0 = slot_rc[s2].first, c0 = slot_rc[s2].second;
                if (r0 < 0 || r0 >= C || c0 >= C) continue;
                if (s2 >= P && s2 < P + p->n_cobs) continue;
                // stored block = (rows r0, cols c0); the band keeps block (i, k) with i >= k
                if (r0 >= c0) { tgt_slot[(size_t)c0 * (Bc + 1) + (r0 - c0)] = s2; tgt_tr[(size_t)c0 * (Bc + 1) + (r0 - c0)] = 0; }
                else { tgt_slot[(size_t)r0 * (Bc + 1) + (c0 - r0)] = s2; tgt_tr[(size_t)r0 * (Bc + 1) + (c0 - r0)] = 1; }
            }
            std::vector<std::vector<int>> ct((size_t)C * (Bc + 1));
            std::vector<std::vector<int>> cr(C);
            cq_off.assign(Q + 1, 0);
            for (int q = 0; q < Q; q++) {
                const auto &e = cub_edges[q];
                for (size_t a = 0; a < e.size(); a++) {
                    cr[e[a].first].push_back(q); cr[e[a].first].push_back(e[a].second);
                    cq_list.push_back(e[a].second); cq_list.push_back(e[a].first);
                    for (size_t c2 = 0; c2 < e.size(); c2++) {
                        if (e[a].first < e[c2].first) continue; // target block (i, k) with i >= k; equal cameras: every ordered pair
                        auto &l = ct[(size_t)e[c2].first * (Bc + 1) + (e[a].first - e[c2].first)];
                        l.push_back(q); l.push_back(e[a].second); l.push_back(e[c2].second);
                    }
                }
                cq_off[q + 1] = (int)(cq_list.size() / 2);
            }
            ct_off.assign(ct.size() + 1, 0);
            for (size_t t = 0; t < ct.size(); t++) { ct_off[t + 1] = ct_off[t] + (int)(ct[t].size() / 3); ct_list.insert(ct_list.end(), ct[t].begin(), ct[t].end()); }
            cr_off.assign(C + 1, 0);
            for (int i = 0; i < C; i++) { cr_off[i + 1] = cr_off[i] + (int)(cr[i].size() / 2); cr_list.insert(cr_list.end(), cr[i].begin(), cr[i].end()); }
            if (ct_list.empty()) ct_list.assign(3, 0);
            if (cr_list.empty()) cr_list.assign(2, 0);
            if (cq_list.empty()) cq_list.assign(2, 0);
        }
    }
    // ordering of the pose blocks: minimum degree on the block graph (the reference lets Eigen::SimplicialLDLT order with AMD,
    // linear_solver_eigen.h:60-75); the simulated elimination also yields the structure of every column of L
    std::vector<std::vector<int>> adj(P);
    for (auto &rc : slot_rc) if (rc.first != rc.second && rc.first >= 0) { adj[rc.first].push_back(rc.second); adj[rc.second].push_back(rc.first); }
    for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    std::vector<int> pos(P, -1), order; order.reserve(P);
    std::vector<std::vector<int>> col_struct(P); // by elimination position: neighbour node ids at elimination time
    {
        std::vector<char> gone(P, 0);
        for (int k = 0; k < P; k++) {
            int v = -1;
            for (int i = 0; i < P; i++) if (!gone[i] && (v < 0 || adj[i].size() < adj[v].size())) v = i;
            gone[v] = 1; pos[v] = k; order.push_back(v);
            std::vector<int> nb = adj[v];
            col_struct[k] = nb;
            for (int a : nb) { // remove v, connect the neighbours pairwise (sorted-vector sets)
                std::vector<int> &s = adj[a];
                std::vector<int> merged;
                merged.reserve(s.size() + nb.size());
                std::set_union(s.begin(), s.end(), nb.begin(), nb.end(), std::back_inserter(merged));
                merged.erase(std::remove_if(merged.begin(), merged.end(), [&](int x) { return x == v || x == a; }), merged.end());
                s.swap(merged);
            }
            adj[v].clear();
        }
    }
    std::vector<int> col_off(P + 1, 0), rows, pair_off(P + 1, 0);
    int max_col = 0;
    for (int k = 0; k < P; k++) {
        std::vector<int> r;
        for (int v : col_struct[k]) r.push_back(pos[v]);
        std::sort(r.begin(), r.end());
        rows.insert(rows.end(), r.begin(), r.end());
        col_off[k + 1] = (int)rows.size();
        pair_off[k + 1] = pair_off[k] + (int)(r.size() * (r.size() + 1) / 2);
        max_col = std::max(max_col, (int)r.size());
    }
    if ((size_t)(36 + (size_t)max_col * 36) * sizeof(double) > 150 * 1024) { ctx->err = "reduced camera system too dense for the LDS column buffer"; delete b; return CS_ERR_CAPACITY; }
    auto find_block = [&](int col, int row) { // storage index of L(row, col), row > col
        const int *lo = &rows[col_off[col]], *hi = &rows[col_off[col + 1]];
        const int *it = std::lower_bound(lo, hi, row);
        return (it != hi && *it == row) ? P + (int)(it - &rows[0]) : -1;
    };
    std::vector<int> upd_tgt((size_t)std::max(pair_off[P], 1));
    for (int k = 0; k < P; k++) {
        const int m = col_off[k + 1] - col_off[k];
        for (int bi = 0; bi < m; bi++)
            for (int bk = 0; bk <= bi; bk++) {
                const int ri = rows[col_off[k] + bi], rk = rows[col_off[k] + bk];
                upd_tgt[(size_t)pair_off[k] + bi * (bi + 1) / 2 + bk] = bi == bk ? ri : find_block(rk, ri);
            }
    }
    std::vector<int> slot_dst(b->n_slots);
    std::vector<uint8_t> slot_tr(b->n_slots);
    for (int s = 0; s < b->n_slots; s++) {
        slot_tr[s] = 0;
        if (slot_rc[s].first < 0) { slot_dst[s] = -1; continue; }
        const int a = pos[slot_rc[s].first], c = pos[slot_rc[s].second];
        if (a == c) slot_dst[s] = a;
        else if (a > c) slot_dst[s] = find_block(c, a);            // block (rows first, cols second) sits below the diagonal as is
        else { slot_dst[s] = find_block(a, c); slot_tr[s] = 1; }   // stored transposed
        if (slot_dst[s] < 0) { ctx->err = "internal: Schur block outside the symbolic factor"; delete b; return CS_ERR_BAD_ARG; }
    }
    b->max_col = max_col;
    b->band_len = ((long)P + (long)rows.size()) * 36;
    b->h_slot_dst = slot_dst; b->h_slot_tr = slot_tr; b->h_pos = pos; b->h_col_off = col_off; b->h_rows = rows;
    if (rows.empty()) rows.push_back(0);
    b->reduce_len = (long)b->n_slots * 36 + (long)P * 6;
    const int nl = G.lm_e - G.lm_b;
    b->max_part = std::max(std::max((G.o_e - G.o_b + 255) / 256, (p->n_cobs + p->n_pc + 255) / 256), std::max((int)(((long)P * 6 + (long)nl * 3 + 255) / 256), (nl + 255) / 256)) + 1;

#define A_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_ba_destroy(ctx, b); return r__; } } while (0)
    int *d_cam_idx, *d_o_cam, *d_o_pt, *d_lm_off, *d_c_cam, *d_c_cub, *d_pc_cub, *d_pc_off;
    double *d_o_uv, *d_o_w, *d_o_ur = nullptr, *d_c_bbox, *d_c_info, *d_pc_pts;
    uint8_t *d_flags;
    const double zero4[4] = {0, 0, 0, 0}; const int zero2[2] = {0, 0};
    A_(dalloc_copy(ctx, b, &G.cam, p->cam_pose, (size_t)p->n_cams * 7));
    A_(dalloc_copy(ctx, b, &G.pts, p->points, (size_t)std::max(p->n_points, 1) * 3));
    A_(dalloc_copy(ctx, b, &G.cub, p->cuboid_pose, (size_t)std::max(p->n_cuboids, 1) * 7));
    A_(dalloc_copy(ctx, b, &G.cub_scale, p->cuboid_scale, (size_t)std::max(p->n_cuboids, 1) * 3));
    A_(dalloc_copy(ctx, b, &d_flags, p->cuboid_flags, (size_t)std::max(p->n_cuboids, 1)));
    A_(dalloc_copy(ctx, b, &d_cam_idx, cam_idx.data(), cam_idx.size()));
    A_(dalloc_copy(ctx, b, &d_o_cam, o_cam.data(), (size_t)std::max(p->n_obs, 1)));
    A_(dalloc_copy(ctx, b, &d_o_pt, o_pt.data(), (size_t)std::max(p->n_obs, 1)));
    A_(dalloc_copy(ctx, b, &d_o_uv, o_uv.data(), (size_t)std::max(p->n_obs, 1) * 2));
    if (p->obs_ur && p->n_obs > 0) A_(dalloc_copy(ctx, b, &d_o_ur, o_ur.data(), (size_t)p->n_obs));
    A_(dalloc_copy(ctx, b, &d_o_w, o_w.data(), (size_t)std::max(p->n_obs, 1)));
    A_(dalloc_copy(ctx, b, &d_lm_off, lm_off.data(), lm_off.size()));
    A_(dalloc_copy(ctx, b, &d_c_cam, p->n_cobs ? p->cobs_cam : zero2, (size_t)std::max(p->n_cobs, 1)));
    A_(dalloc_copy(ctx, b, &d_c_cub, p->n_cobs ? p->cobs_cuboid : zero2, (size_t)std::max(p->n_cobs, 1)));
    A_(dalloc_copy(ctx, b, &d_c_bbox, p->n_cobs ? p->cobs_bbox : zero4, (size_t)std::max(p->n_cobs, 1) * 4));
    A_(dalloc_copy(ctx, b, &d_c_info, p->n_cobs ? p->cobs_info : zero4, (size_t)std::max(p->n_cobs, 1) * 4));
    A_(dalloc_copy(ctx, b, &d_pc_cub, p->n_pc ? p->pc_cuboid : zero2, (size_t)std::max(p->n_pc, 1)));
    A_(dalloc_copy(ctx, b, &d_pc_off, p->n_pc ? p->pc_offsets : zero2, (size_t)p->n_pc + 1));
    const int n_pcp = p->n_pc ? p->pc_offsets[p->n_pc] : 0;
    A_(dalloc_copy(ctx, b, &d_pc_pts, n_pcp ? p->pc_points : zero4, (size_t)std::max(n_pcp, 1) * 3));
    G.cam_idx = d_cam_idx; G.cub_flags = d_flags; G.o_cam = d_o_cam; G.o_pt = d_o_pt; G.o_uv = d_o_uv; G.o_w = d_o_w; G.o_ur = d_o_ur; G.lm_off = d_lm_off;
    G.c_cam = d_c_cam; G.c_cub = d_c_cub; G.c_bbox = d_c_bbox; G.c_info = d_c_info; G.pc_cub = d_pc_cub; G.pc_off = d_pc_off; G.pc_pts = d_pc_pts;
    A_(dalloc_copy(ctx, b, &G.e_obs, (const double *)nullptr, (size_t)std::max(p->n_obs, 1) * 3));
    A_(dalloc_copy(ctx, b, &G.e_cobs, (const double *)nullptr, (size_t)std::max(p->n_cobs, 1) * 4));
    A_(dalloc_copy(ctx, b, &G.e_pc, (const double *)nullptr, (size_t)std::max(p->n_pc, 1) * 3));
    A_(dalloc_copy(ctx, b, &G.Hpl, (const double *)nullptr, (size_t)std::max(p->n_obs, 1) * 18));
    A_(dalloc_copy(ctx, b, &G.HplD, (const double *)nullptr, (size_t)std::max(p->n_obs, 1) * 18));
    A_(dalloc_copy(ctx, b, &b->d_bw, (const double *)nullptr, (size_t)std::max(p->n_obs, 1) * 6));
    A_(dalloc_copy(ctx, b, &G.Hll, (const double *)nullptr, (size_t)std::max(p->n_points, 1) * 9));
    A_(dalloc_copy(ctx, b, &G.bl, (const double *)nullptr, (size_t)std::max(p->n_points, 1) * 3));
    A_(dalloc_copy(ctx, b, &G.Dinv, (const double *)nullptr, (size_t)std::max(p->n_points, 1) * 9));
    A_(dalloc_copy(ctx, b, &G.db, (const double *)nullptr, (size_t)std::max(p->n_points, 1) * 3));
    A_(dalloc_copy(ctx, b, &G.Hpp, (const double *)nullptr, (size_t)P * 36));
    A_(dalloc_copy(ctx, b, &G.bp, (const double *)nullptr, (size_t)P * 6));
    A_(dalloc_copy(ctx, b, &G.Hoff, (const double *)nullptr, (size_t)std::max(p->n_cobs, 1) * 36));
    A_(dalloc_copy(ctx, b, &G.Jc, (const double *)nullptr, (size_t)std::max(p->n_cobs, 1) * 48));
    A_(dalloc_copy(ctx, b, &G.Jp, (const double *)nullptr, (size_t)std::max(p->n_pc, 1) * 18));
    A_(dalloc_copy(ctx, b, &G.x, (const double *)nullptr, (size_t)P * 6 + (size_t)std::max(p->n_points, 1) * 3));
    A_(dalloc_copy(ctx, b, &b->d_pose_off, pose_off.data(), pose_off.size()));
    A_(dalloc_copy(ctx, b, &b->d_pose_obs, pose_obs.data(), pose_obs.size()));
    A_(dalloc_copy(ctx, b, &b->d_pe_off, pe_off.data(), pe_off.size()));
    A_(dalloc_copy(ctx, b, &b->d_pe_list, pe_list.data(), pe_list.size()));
    A_(dalloc_copy(ctx, b, &b->d_slot_off, slot_off.data(), slot_off.size()));
    { // XCD-aware slot order for ba_schur_slots: sort by the smaller pose index of the block, cut into 8 contiguous ranges, deal range x to
      // the workgroups with blockIdx % 8 == x (4 slots per workgroup)
        const int ns = b->n_slots;
        std::vector<int> by_pose((size_t)ns), perm((size_t)((ns + 3) / 4) * 4, 0);
        for (int i = 0; i < ns; i++) by_pose[i] = i;
        auto keyf = [&](int q) { const int a = slot_rc[q].first, c = slot_rc[q].second; return a < 0 ? (1 << 30) : std::min(a, c); };
        std::stable_sort(by_pose.begin(), by_pose.end(), [&](int x, int y) { return keyf(x) < keyf(y); });
        const int nblk = (ns + 3) / 4;
        int next = 0;
        for (int x = 0; x < 8; x++) // blocks x, x+8, x+16, ... take consecutive slots of the sorted list
            for (int blk = x; blk < nblk; blk += 8)
                for (int k = 0; k < 4; k++) perm[(size_t)blk * 4 + k] = next < ns ? by_pose[next++] : -1;
        // the tail entries (-1) are never read: the kernel stops at n_slots positions -- make the first n_slots positions a permutation
        std::vector<int> flat; flat.reserve(ns);
        for (int v : perm) if (v >= 0) flat.push_back(v);
        std::vector<int> fin((size_t)ns);
        // positions [0, ns) must all be valid: fill in block order, skipping the holes of the partially filled last blocks
        { size_t w = 0; for (size_t i = 0; i < perm.size() && w < (size_t)ns; i++) if (perm[i] >= 0) fin[w++] = perm[i]; }
        A_(dalloc_copy(ctx, b, &b->d_slot_perm, fin.data(), fin.size()));
    }
    A_(dalloc_copy(ctx, b, &b->d_trips, trips.data(), trips.size()));
    A_(dalloc_copy(ctx, b, &b->d_slot_dst, slot_dst.data(), slot_dst.size()));
    A_(dalloc_copy(ctx, b, &b->d_slot_tr, slot_tr.data(), slot_tr.size()));
    A_(dalloc_copy(ctx, b, &b->d_col_off, col_off.data(), col_off.size()));
    A_(dalloc_copy(ctx, b, &b->d_rows, rows.data(), rows.size()));
    A_(dalloc_copy(ctx, b, &b->d_pair_off, pair_off.data(), pair_off.size()));
    A_(dalloc_copy(ctx, b, &b->d_upd_tgt, upd_tgt.data(), upd_tgt.size()));
    A_(dalloc_copy(ctx, b, &b->d_pos, pos.data(), pos.size()));
    A_(dalloc_copy(ctx, b, &b->d_status, (const int *)nullptr, 1));
    if (b->use_band) {
        const size_t nt = (size_t)b->band_targets;
        A_(dalloc_copy(ctx, b, &b->d_tgt_slot, tgt_slot.data(), tgt_slot.size()));
        A_(dalloc_copy(ctx, b, &b->d_tgt_tr, tgt_tr.data(), tgt_tr.size()));
        A_(dalloc_copy(ctx, b, &b->d_ct_off, ct_off.data(), ct_off.size()));
        A_(dalloc_copy(ctx, b, &b->d_ct_list, ct_list.data(), ct_list.size()));
        A_(dalloc_copy(ctx, b, &b->d_cr_off, cr_off.data(), cr_off.size()));
        A_(dalloc_copy(ctx, b, &b->d_cr_list, cr_list.data(), cr_list.size()));
        A_(dalloc_copy(ctx, b, &b->d_cq_off, cq_off.data(), cq_off.size()));
        A_(dalloc_copy(ctx, b, &b->d_cq_list, cq_list.data(), cq_list.size()));
        A_(dalloc_copy(ctx, b, &b->d_bandA, (const double *)nullptr, nt * 36));
        A_(dalloc_copy(ctx, b, &b->d_bandL, (const double *)nullptr, nt * 36));
        A_(dalloc_copy(ctx, b, &b->d_cubD, (const double *)nullptr, (size_t)std::max(b->band_Q, 1) * 36));
        A_(dalloc_copy(ctx, b, &b->d_cubg, (const double *)nullptr, (size_t)std::max(b->band_Q, 1) * 6));
        A_(dalloc_copy(ctx, b, &b->d_brhs, (const double *)nullptr, (size_t)b->band_C * 6));
        A_(dalloc_copy(ctx, b, &b->d_ybuf, (const double *)nullptr, (size_t)b->band_C * BAND_YB));
        A_(dalloc_copy(ctx, b, &b->d_mid, (const double *)nullptr, 2 * ((size_t)b->band_bc * (b->band_bc + 1) * 36 + (size_t)b->band_bc * 6) + 8));
        A_(dalloc_copy(ctx, b, &b->d_xmid, (const double *)nullptr, (size_t)b->band_bc * 6 + 8));
    }
    A_(dalloc_copy(ctx, b, &b->d_reduce, (const double *)nullptr, (size_t)b->reduce_len));
    A_(dalloc_copy(ctx, b, &b->d_band, (const double *)nullptr, (size_t)b->band_len));
    A_(dalloc_copy(ctx, b, &b->d_xperm, (const double *)nullptr, (size_t)P * 6));
    A_(dalloc_copy(ctx, b, &b->d_partials, (const double *)nullptr, (size_t)b->max_part * 3)); // three regions: scale | chi2 of the observations | chi2 of the pose edges
    A_(dalloc_copy(ctx, b, &b->d_scal, (const double *)nullptr, (size_t)G.P * 6 + (size_t)world + 64));
    A_(dalloc_copy(ctx, b, &b->d_bak_cam, (const double *)nullptr, (size_t)p->n_cams * 7));
    A_(dalloc_copy(ctx, b, &b->d_bak_pts, (const double *)nullptr, (size_t)std::max(p->n_points, 1) * 3));
    A_(dalloc_copy(ctx, b, &b->d_bak_cub, (const double *)nullptr, (size_t)std::max(p->n_cuboids, 1) * 7));
#undef A_
    // estimates enter through SE3Quat(Vector7d): normalizeRotation (se3quat.h:64-67) -- done on the host copy
    {
        std::vector<double> cam(p->cam_pose, p->cam_pose + (size_t)p->n_cams * 7), cub(p->cuboid_pose, p->cuboid_pose + (size_t)p->n_cuboids * 7);
        for (int i = 0; i < p->n_cams; i++) { SE3 T = se3_load(&cam[(size_t)i * 7]); normalize_rotation(T); se3_store(T, &cam[(size_t)i * 7]); }
        for (int i = 0; i < p->n_cuboids; i++) { SE3 T = se3_load(&cub[(size_t)i * 7]); normalize_rotation(T); se3_store(T, &cub[(size_t)i * 7]); }
        int r = cs_h2d(ctx, G.cam, cam.data(), cam.size());
        if (!r && !cub.empty()) r = cs_h2d(ctx, G.cub, cub.data(), cub.size());
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (r || e != hipSuccess) { cs_ba_destroy(ctx, b); return r ? r : CS_ERR_HIP; }
    }
    *out = b;
    return CS_OK;
}

int cs_ba_errors(cs_ctx *ctx, cs_ba *b, double *chi2, double *err_obs, double *err_cobs, double *err_pc) {
    if (!ctx || !b || !chi2) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    int r = ba_compute_errors(ctx, b, chi2); if (r) return r;
    const Params &G = b->G;
    if (err_obs) { // back to the caller's observation order
        std::vector<double> e((size_t)std::max(G.n_obs, 1) * 3);
        r = cs_d2h(ctx, e.data(), G.e_obs, (size_t)G.n_obs * 3); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int q = G.o_b; q < G.o_e; q++) for (int k = 0; k < 3; k++) err_obs[(size_t)b->obs_perm[q] * 3 + k] = e[(size_t)q * 3 + k];
    }
    if (err_cobs && G.pose_edges) { r = cs_d2h(ctx, err_cobs, G.e_cobs, (size_t)G.n_cobs * 4); if (r) return r; }
    if (err_pc && G.pose_edges) { r = cs_d2h(ctx, err_pc, G.e_pc, (size_t)G.n_pc * 3); if (r) return r; }
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_ba_reduced_dense(cs_ctx *ctx, cs_ba *b, double lambda, double *H, double *bvec, int *Pout) {
    if (!ctx || !b || !H || !bvec || !Pout) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    double chi;
    int r = ba_compute_errors(ctx, b, &chi); if (r) return r;
    r = ba_build_system(ctx, b); if (r) return r;
    r = ba_schur(ctx, b, lambda); if (r) return r;
    std::vector<double> red((size_t)b->reduce_len);
    r = cs_d2h(ctx, red.data(), b->d_reduce, red.size()); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int P = b->G.P, n = P * 6;
    std::vector<int> inv(P);
    for (int i = 0; i < P; i++) inv[b->h_pos[i]] = i;
    std::vector<int> blk_col((size_t)b->h_rows.size() + 1, 0); // column (elimination position) of every off-diagonal block
    for (int k = 0; k < P; k++) for (int q = b->h_col_off[k]; q < b->h_col_off[k + 1]; q++) blk_col[q] = k;
    std::fill(H, H + (size_t)n * n, 0.0);
    for (int s = 0; s < b->n_slots; s++) {
        const int dst = b->h_slot_dst[s];
        if (dst < 0) continue;
        int i, j; // the slot holds block (rows i, cols j) in pose numbering
        if (dst < P) i = j = inv[dst];
        else { const int q = dst - P, rowp = b->h_rows[q], colp = blk_col[q]; if (b->h_slot_tr[s]) { i = inv[colp]; j = inv[rowp]; } else { i = inv[rowp]; j = inv[colp]; } }
        for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) {
            const double v = red[(size_t)s * 36 + a * 6 + c];
            H[(size_t)(i * 6 + a) * n + j * 6 + c] += v;
            if (i != j) H[(size_t)(j * 6 + c) * n + i * 6 + a] += v;
        }
    }
    for (int i = 0; i < n; i++) bvec[i] = red[(size_t)b->n_slots * 36 + i];
    *Pout = P;
    return CS_OK;
}

int cs_ba_read(cs_ctx *ctx, cs_ba *b, double *cam_pose, double *points, double *cuboid_pose) {
    if (!ctx || !b) return CS_ERR_BAD_ARG;
    const Params &G = b->G;
    int r;
    if (cam_pose) { r = cs_d2h(ctx, cam_pose, G.cam, (size_t)G.n_cams * 7); if (r) return r; }
    if (points) { r = cs_d2h(ctx, points + (size_t)G.lm_b * 3, G.pts + (size_t)G.lm_b * 3, (size_t)(G.lm_e - G.lm_b) * 3); if (r) return r; }
    if (cuboid_pose) { r = cs_d2h(ctx, cuboid_pose, G.cub, (size_t)G.n_cub * 7); if (r) return r; }
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_ba_set_stop_flag_bool(cs_ba *b, const volatile unsigned char *flag) { if (!b) return CS_ERR_BAD_ARG; b->stop8 = flag; return CS_OK; }

int cs_ba_optimize(cs_ctx *ctx, cs_ba *b, int iterations, const volatile int *stop_flag, cs_ba_stats *st) {
    if (!ctx || !b || iterations < 0) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    if (b->world > 1 && !b->allreduce && !(ctx->comm && ctx->comm_world == b->world)) { ctx->err = "sharded BA needs cs_comm_init (RCCL) or cs_ba_set_allreduce"; return CS_ERR_BAD_ARG; }
    const Params &G = b->G;
    cs_ba_stats S;
    memset(&S, 0, sizeof(S));
    double lambda = 0, ni = 2, currentChi = 0;
    int nBad = 0, r;
    const int nl = G.lm_e - G.lm_b;
    auto terminate = [&]() { return (stop_flag && *stop_flag) || (b->stop8 && *b->stop8); }; // sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149
    // A trial is accepted nearly always, and the next iteration then starts by linearising at the state the trial left.  So that system is BUILT AHEAD: enqueued behind the
    // trial's residual kernels before the host waits for their sums, and the GPU goes from one iteration into the next without the round trip (copy back, decision, launch)
    // in between.  A rejected trial restores the estimates and rebuilds what the build-ahead overwrote -- residuals, then the system -- from the restored state: the same
    // kernels on the same bits, so the retried solve sees the system it would have kept (CUBESLAM_BA_AHEAD=0: the plain order).
    const bool ahead_on = b->world == 1 && !(getenv("CUBESLAM_BA_AHEAD") && atoi(getenv("CUBESLAM_BA_AHEAD")) == 0);
    bool built_ahead = false;
    for (int it = 0; it < iterations && !terminate(); it++) { // OptimizationAlgorithmLevenberg::solve :61-164
        // computeActiveErrors: after an accepted trial the residual arrays and chi2 on the device are those of the current state (every
        // way out of the trial loop with a rejected last trial also leaves this loop), so only the first iteration evaluates them
        double tempChi;
        if (it == 0) { r = ba_compute_errors(ctx, b, &currentChi); if (r) return r; }
        tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) S.chi2_init = currentChi;
        if (!built_ahead) { r = ba_build_system(ctx, b); if (r) return r; }
        built_ahead = false;
        if (it == 0) { // computeLambdaInit :166-180: tau * max |diag(H)| over all active vertices
            std::vector<double> v((size_t)G.P * 6 + (size_t)b->world, 0.0), hpp((size_t)G.P * 36);
            r = cs_d2h(ctx, hpp.data(), G.Hpp, hpp.size()); if (r) return r;
            double mxl = 0;
            if (nl > 0) { CS_LAUNCH(ctx, "ba_maxdiag", ba_maxdiag, dim3((nl + 255) / 256), dim3(256), 0, G, b->d_partials); mxl = sum_partials(ctx, b, (nl + 255) / 256, true); }
            CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            double mx = mxl;
            if (b->world > 1) { // pose diagonals are sums over ranks; landmark maxima travel in per-rank slots of the same sum
                std::vector<double> buf((size_t)G.P * 6 + b->world, 0.0);
                for (int i = 0; i < G.P; i++) for (int k = 0; k < 6; k++) buf[(size_t)i * 6 + k] = hpp[(size_t)i * 36 + k * 7];
                buf[(size_t)G.P * 6 + b->rank] = mxl;
                r = allreduce_scalars(ctx, b, buf.data(), (int)buf.size()); if (r) return r; // one collective (it was one per 64 scalars)
                mx = 0;
                for (double d : buf) mx = std::max(mx, std::fabs(d));
            } else
                for (int i = 0; i < G.P; i++) for (int k = 0; k < 6; k++) mx = std::max(mx, std::fabs(hpp[(size_t)i * 36 + k * 7]));
            lambda = 1e-5 * mx; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            // push(): back up the estimates
            CS_HIP(ctx, hipMemcpyAsync(b->d_bak_cam, G.cam, sizeof(double) * (size_t)G.n_cams * 7, hipMemcpyDeviceToDevice, ctx->stream));
            if (G.L) CS_HIP(ctx, hipMemcpyAsync(b->d_bak_pts, G.pts, sizeof(double) * (size_t)G.L * 3, hipMemcpyDeviceToDevice, ctx->stream));
            if (G.n_cub) CS_HIP(ctx, hipMemcpyAsync(b->d_bak_cub, G.cub, sizeof(double) * (size_t)G.n_cub * 7, hipMemcpyDeviceToDevice, ctx->stream));
            bool ok2 = true;
            const int nbs = (int)(((long)G.P * 6 + (long)nl * 3 + 255) / 256);
            double scale;
            if (b->world == 1) { // one host round trip per trial: solve, scale, update and the new residuals are enqueued back to back
                r = ba_solve(ctx, b, lambda, &ok2, true); if (r) return r;
                const int nb1 = (G.o_e - G.o_b + 255) / 256, nb2 = (G.pose_edges && G.n_cobs + G.n_pc > 0) ? (G.n_cobs + G.n_pc + 255) / 256 : 0, mp = b->max_part;
                CS_LAUNCH(ctx, "ba_update", ba_scale_update, dim3(nbs + (G.n_cams + G.n_cub + nl * 3 + 255) / 256), dim3(256), 0, G, lambda, b->d_partials, nbs);
                if (nb1 > 0) CS_LAUNCH(ctx, "ba_err_obs", ba_err_obs, dim3(nb1), dim3(256), 0, G, b->d_partials + mp);
                if (nb2 > 0) CS_LAUNCH(ctx, "ba_err_pose_edges", ba_err_pose_edges, dim3(nb2), dim3(256), 0, G, b->d_partials + 2 * mp); // (one grid for both measured no faster: 36 us against 14 + 20, the cuboid edges' registers halve the reprojection edges' occupancy)
                const size_t need = (size_t)mp * 3 + 1;
                if (b->pin_cap < need) {
                    if (b->h_pin) hipHostFree(b->h_pin);
                    b->h_pin = nullptr; b->pin_cap = 0;
                    CS_HIP(ctx, hipHostMalloc((void **)&b->h_pin, need * sizeof(double), hipHostMallocDefault));
                    b->pin_cap = need;
                }
                if (!b->ev_trial) CS_HIP(ctx, hipEventCreateWithFlags(&b->ev_trial, hipEventDisableTiming));
                int *h_status = reinterpret_cast<int *>(b->h_pin + (size_t)mp * 3);
                r = cs_d2h(ctx, h_status, b->d_status, 1); if (r) return r;
                r = cs_d2h(ctx, b->h_pin, b->d_partials, (size_t)mp * 3); if (r) return r;
                CS_HIP(ctx, hipEventRecord(b->ev_trial, ctx->stream));
                if (ahead_on && it + 1 < iterations) { r = ba_build_system(ctx, b); if (r) return r; built_ahead = true; } // (the last iteration has no successor to build for)
                CS_HIP(ctx, hipEventSynchronize(b->ev_trial));
                const int status = *h_status;
                const double *hp = b->h_pin;
                ok2 = status == 0;
                scale = 0;
                for (int i = 0; i < nbs; i++) scale += hp[i];
                double chi = 0, c1 = 0, c2 = 0; // same order of additions as ba_compute_errors
                for (int i = 0; i < nb1; i++) c1 += hp[(size_t)mp + i];
                for (int i = 0; i < nb2; i++) c2 += hp[(size_t)2 * mp + i];
                if (nb1 > 0) chi += c1;
                if (nb2 > 0) chi += c2;
                tempChi = chi;
            } else {
                r = ba_solve(ctx, b, lambda, &ok2); if (r) return r;
                CS_LAUNCH(ctx, "ba_scale", ba_scale, dim3(nbs), dim3(256), 0, G, lambda, b->d_partials);
                scale = sum_partials(ctx, b, nbs);
                r = allreduce_scalars(ctx, b, &scale, 1); if (r) return r;
                CS_LAUNCH(ctx, "ba_update", ba_update, dim3((G.n_cams + G.n_cub + nl * 3 + 255) / 256), dim3(256), 0, G);
                r = ba_compute_errors(ctx, b, &tempChi); if (r) return r;
            }
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = (currentChi - tempChi);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = (std::min)(alpha, 2. / 3.);
                const double scaleFactor = (std::max)(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi; // discardTop
            } else {
                lambda *= ni;
                ni *= 2; // pop(): restore
                CS_HIP(ctx, hipMemcpyAsync(G.cam, b->d_bak_cam, sizeof(double) * (size_t)G.n_cams * 7, hipMemcpyDeviceToDevice, ctx->stream));
                if (G.L) CS_HIP(ctx, hipMemcpyAsync(G.pts, b->d_bak_pts, sizeof(double) * (size_t)G.L * 3, hipMemcpyDeviceToDevice, ctx->stream));
                if (G.n_cub) CS_HIP(ctx, hipMemcpyAsync(G.cub, b->d_bak_cub, sizeof(double) * (size_t)G.n_cub * 7, hipMemcpyDeviceToDevice, ctx->stream));
                if (built_ahead) { // the system in the buffers is the rejected state's: residuals and system of the restored one again (the retry solves that)
                    double unused; r = ba_compute_errors(ctx, b, &unused); if (r) return r;
                    r = ba_build_system(ctx, b); if (r) return r;
                    built_ahead = false;
                }
            }
            qmax++;
            S.lm_trials++;
        } while (rho < 0 && qmax < 10 && !terminate());
        S.iterations = it + 1;
        if (it < 64) S.chi2_trace[it] = currentChi;
        S.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; // stop criterion :155-161
        if (nBad >= 3) break;
    }
    S.lambda_final = lambda;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st) *st = S;
    return CS_OK;
}

int cs_pose_optimization(cs_ctx *ctx, int n_frames, const int *edge_off, const double *Xw, const double *obs, const double *inv_sigma2, const double *intrinsics,
                         const double *pose_in, double *pose_out, uint8_t *outlier, int *n_inliers) {
    if (!ctx || n_frames < 0 || !edge_off || (n_frames && (!intrinsics || !pose_in || !pose_out || !n_inliers))) return CS_ERR_BAD_ARG;
    if (n_frames == 0) return CS_OK;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const int ne = edge_off[n_frames];
    for (int f = 0; f < n_frames; f++) if (edge_off[f + 1] < edge_off[f]) return CS_ERR_BAD_ARG;
    if (ne > 0 && (!Xw || !obs || !inv_sigma2 || !outlier)) return CS_ERR_BAD_ARG;
    std::vector<PoseFrame> fr((size_t)n_frames);
    for (int f = 0; f < n_frames; f++) { fr[f].e0 = edge_off[f]; fr[f].e1 = edge_off[f + 1]; fr[f].fx = intrinsics[f * 5]; fr[f].fy = intrinsics[f * 5 + 1]; fr[f].cx = intrinsics[f * 5 + 2]; fr[f].cy = intrinsics[f * 5 + 3]; fr[f].bf = intrinsics[f * 5 + 4]; }
    PoseFrame *d_fr = nullptr; double *d_X = nullptr, *d_o = nullptr, *d_w = nullptr, *d_pi = nullptr, *d_po = nullptr, *d_err = nullptr; uint8_t *d_out = nullptr; int *d_ni = nullptr;
    const size_t ne1 = (size_t)std::max(ne, 1);
    int r = cs_dalloc(ctx, &d_fr, (size_t)n_frames);
    if (!r) r = cs_dalloc(ctx, &d_X, ne1 * 3); if (!r) r = cs_dalloc(ctx, &d_o, ne1 * 3); if (!r) r = cs_dalloc(ctx, &d_w, ne1); if (!r) r = cs_dalloc(ctx, &d_err, ne1 * 3);
    if (!r) r = cs_dalloc(ctx, &d_pi, (size_t)n_frames * 7); if (!r) r = cs_dalloc(ctx, &d_po, (size_t)n_frames * 7); if (!r) r = cs_dalloc(ctx, &d_out, ne1); if (!r) r = cs_dalloc(ctx, &d_ni, (size_t)n_frames);
    if (!r) r = cs_h2d(ctx, d_fr, fr.data(), fr.size());
    if (!r && ne) { r = cs_h2d(ctx, d_X, Xw, (size_t)ne * 3); if (!r) r = cs_h2d(ctx, d_o, obs, (size_t)ne * 3); if (!r) r = cs_h2d(ctx, d_w, inv_sigma2, (size_t)ne); }
    if (!r) r = cs_h2d(ctx, d_pi, pose_in, (size_t)n_frames * 7);
    if (!r) {
        ctx->begin("pose_opt_kernel");
        hipLaunchKernelGGL(pose_opt_kernel, dim3(n_frames), dim3(256), 0, ctx->stream, d_fr, d_X, d_o, d_w, d_pi, d_po, d_out, d_ni, d_err);
        ctx->end();
        r = cs_d2h(ctx, pose_out, d_po, (size_t)n_frames * 7);
        if (!r && ne) r = cs_d2h(ctx, outlier, d_out, (size_t)ne);
        if (!r) r = cs_d2h(ctx, n_inliers, d_ni, (size_t)n_frames);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_fr, d_X, d_o, d_w, d_pi, d_po, d_err, d_out, d_ni};
    for (void *q : ptrs) if (q) hipFree(q);
    return r;
}

int cs_cuboid9_oplus(cs_ctx *ctx, int n, const double *cub, const double *upd, double *out) {
    if (!ctx || n < 0 || (n && (!cub || !upd || !out))) return CS_ERR_BAD_ARG;
    if (n == 0) return CS_OK;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    double *d_c = nullptr, *d_u = nullptr, *d_o = nullptr;
    int r = cs_dalloc(ctx, &d_c, (size_t)n * 10); if (!r) r = cs_dalloc(ctx, &d_u, (size_t)n * 9); if (!r) r = cs_dalloc(ctx, &d_o, (size_t)n * 10);
    if (!r) r = cs_h2d(ctx, d_c, cub, (size_t)n * 10); if (!r) r = cs_h2d(ctx, d_u, upd, (size_t)n * 9);
    if (!r) { hipLaunchKernelGGL(cub9_oplus_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, d_c, d_u, d_o); r = cs_d2h(ctx, out, d_o, (size_t)n * 10); }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    if (d_c) hipFree(d_c); if (d_u) hipFree(d_u); if (d_o) hipFree(d_o);
    return r;
}

int cs_cuboid9_edge_linearize(cs_ctx *ctx, int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err, double *Jcam, double *Jcub) {
    if (!ctx || n < 0 || (n && (!cam_Tcw || !cub_global || !cub_meas_local || !err)) || ((Jcam == nullptr) != (Jcub == nullptr))) return CS_ERR_BAD_ARG;
    if (n == 0) return CS_OK;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const int with_jac = Jcam != nullptr;
    double *d_T = nullptr, *d_g = nullptr, *d_m = nullptr, *d_e = nullptr, *d_jc = nullptr, *d_jq = nullptr;
    int r = cs_dalloc(ctx, &d_T, (size_t)n * 7); if (!r) r = cs_dalloc(ctx, &d_g, (size_t)n * 10); if (!r) r = cs_dalloc(ctx, &d_m, (size_t)n * 10); if (!r) r = cs_dalloc(ctx, &d_e, (size_t)n * 9);
    if (!r && with_jac) { r = cs_dalloc(ctx, &d_jc, (size_t)n * 54); if (!r) r = cs_dalloc(ctx, &d_jq, (size_t)n * 81); }
    if (!r) r = cs_h2d(ctx, d_T, cam_Tcw, (size_t)n * 7); if (!r) r = cs_h2d(ctx, d_g, cub_global, (size_t)n * 10); if (!r) r = cs_h2d(ctx, d_m, cub_meas_local, (size_t)n * 10);
    if (!r) {
        const int nt = n * (with_jac ? 16 : 1);
        hipLaunchKernelGGL(cub9_edge_kernel, dim3((nt + 63) / 64), dim3(64), 0, ctx->stream, n, with_jac, d_T, d_g, d_m, d_e, d_jc, d_jq);
        r = cs_d2h(ctx, err, d_e, (size_t)n * 9);
        if (!r && with_jac) { r = cs_d2h(ctx, Jcam, d_jc, (size_t)n * 54); if (!r) r = cs_d2h(ctx, Jcub, d_jq, (size_t)n * 81); }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_T, d_g, d_m, d_e, d_jc, d_jq};
    for (void *q : ptrs) if (q) hipFree(q);
    return r;
}

} // extern "C"
