/*
 * oracle/se3_util.h -- SE3Quat / cuboid helpers shared by the bundle-adjustment oracles (ba_oracle.cpp, badyn_oracle.cpp).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Restated from the vendored g2o types/se3quat.h and orb_object_slam/{include/g2o_Object.h,
 * src/g2o_Object.cpp}; the Eigen expressions they reach are spelled out in Eigen's evaluation order.
 */
#ifndef ORACLE_SE3_UTIL_H
#define ORACLE_SE3_UTIL_H
#include <algorithm>
#include <cmath>

namespace {

struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };
typedef double M3[3][3];

static inline Quat qmul(const Quat &a, const Quat &b) { // Eigen quaternion product
    return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
static inline void qrot(const Quat &q, const double *v, double *o) { // Eigen QuaternionBase::_transformVector
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
static inline void qtoR(const Quat &q, M3 R) { // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
static inline Quat qfromR(const M3 m) { // Eigen Quaterniond(Matrix3d)
    Quat q;
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0) {
        t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m[2][1] - m[1][2]) * t; q.y = (m[0][2] - m[2][0]) * t; q.z = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m[k][j] - m[j][k]) * t; v[j] = (m[j][i] + m[i][j]) * t; v[k] = (m[k][i] + m[i][k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
static inline void normalize_rotation(SE3 &T) { // se3quat.h:331-336
    if (T.r.w < 0) { T.r.x *= -1; T.r.y *= -1; T.r.z *= -1; T.r.w *= -1; }
    double n = std::sqrt(T.r.x * T.r.x + T.r.y * T.r.y + T.r.z * T.r.z + T.r.w * T.r.w);
    T.r.x /= n; T.r.y /= n; T.r.z /= n; T.r.w /= n;
}
static inline SE3 se3_mul(const SE3 &a, const SE3 &b) { // se3quat.h:110-116
    SE3 r = a;
    double rt[3];
    qrot(a.r, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    r.r = qmul(a.r, b.r);
    normalize_rotation(r);
    return r;
}
static inline SE3 se3_inv(const SE3 &a) { // :129-134
    SE3 r;
    r.r = Quat{-a.r.x, -a.r.y, -a.r.z, a.r.w};
    double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
    qrot(r.r, nt, r.t);
    return r;
}
static inline void se3_map(const SE3 &T, const double *p, double *o) { qrot(T.r, p, o); o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2]; }
static inline SE3 se3_from7(const double *v) { // [t, qx qy qz qw], normalizeRotation
    SE3 T; T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2]; T.r = Quat{v[3], v[4], v[5], v[6]};
    normalize_rotation(T);
    return T;
}
static inline void se3_to7(const SE3 &T, double *v) { v[0] = T.t[0]; v[1] = T.t[1]; v[2] = T.t[2]; v[3] = T.r.x; v[4] = T.r.y; v[5] = T.r.z; v[6] = T.r.w; }

static inline void mat3mul(const M3 a, const M3 b, M3 c) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c[i][j] = (a[i][0] * b[0][j] + a[i][1] * b[1][j]) + a[i][2] * b[2][j];
}
static SE3 se3_exp(const double *u) { // se3quat.h:272-306
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    M3 O = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}}, O2, R, V;
    mat3mul(O, O, O2);
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = ((i == j ? 1.0 : 0.0) + O[i][j]) + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / (std::pow(theta, 3));
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                R[i][j] = ((i == j ? 1.0 : 0.0) + a * O[i][j]) + b * O2[i][j];
                V[i][j] = ((i == j ? 1.0 : 0.0) + b * O[i][j]) + c * O2[i][j];
            }
    }
    SE3 T;
    T.r = qfromR(R);
    for (int i = 0; i < 3; i++) T.t[i] = (V[i][0] * up[0] + V[i][1] * up[1]) + V[i][2] * up[2];
    normalize_rotation(T);
    return T;
}
static SE3 exptwist_norollpitch(const double *u) { // g2o_Object.cpp:24-54
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    M3 O = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}}, O2, V;
    M3 R = {{std::cos(om[2]), -std::sin(om[2]), 0}, {std::sin(om[2]), std::cos(om[2]), 0}, {0, 0, 1}};
    if (theta < 0.00001) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = R[i][j]; }
    else {
        mat3mul(O, O, O2);
        const double b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / (std::pow(theta, 3));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = ((i == j ? 1.0 : 0.0) + b * O[i][j]) + c * O2[i][j];
    }
    SE3 T;
    T.r = qfromR(R);
    for (int i = 0; i < 3; i++) T.t[i] = (V[i][0] * up[0] + V[i][1] * up[1]) + V[i][2] * up[2];
    normalize_rotation(T);
    return T;
}

struct Cuboid { SE3 pose; double scale[3]; };

// VertexCuboidFixScale::oplusImpl g2o_Object.cpp:88-116
static Cuboid cuboid_oplus(const Cuboid &e, const double *upd, int flags, const double *fixedscale) {
    Cuboid n;
    n.pose.r = Quat{0, 0, 0, 1}; n.pose.t[0] = n.pose.t[1] = n.pose.t[2] = 0;
    if (flags & 2) { // whether_fixrotation
        n.pose.r = e.pose.r;
        for (int i = 0; i < 3; i++) n.pose.t[i] = e.pose.t[i] + upd[3 + i];
    } else if (flags & 1) { // whether_fixrollpitch
        double u2[6] = {0, 0, upd[2], upd[3], upd[4], upd[5]};
        n.pose = se3_mul(e.pose, exptwist_norollpitch(u2));
    } else
        n.pose = se3_mul(e.pose, se3_exp(upd));
    if (flags & 4) n.pose.t[1] = e.pose.t[1]; // whether_fixheight keeps y (:107-108)
    for (int i = 0; i < 3; i++) n.scale[i] = (flags & 8) ? fixedscale[i] : e.scale[i];
    return n;
}

// cuboid::projectOntoImageBbox g2o_Object.h:189-220 -> [cx, cy, w, h]
static void project_bbox(const Cuboid &c, const SE3 &Tcw, const double *K, double *out) {
    static const double body[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
    M3 Ro, Rc;
    qtoR(c.pose.r, Ro);
    qtoR(Tcw.r, Rc);
    double rs[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rs[i][j] = Ro[i][j] * c.scale[j];
    double mnx = 0, mny = 0, mxx = 0, mxy = 0;
    for (int k = 0; k < 8; k++) {
        double pw[3], pc[3];
        for (int i = 0; i < 3; i++) pw[i] = ((rs[i][0] * body[0][k] + rs[i][1] * body[1][k]) + rs[i][2] * body[2][k]) + c.pose.t[i];
        for (int i = 0; i < 3; i++) pc[i] = ((Rc[i][0] * pw[0] + Rc[i][1] * pw[1]) + Rc[i][2] * pw[2]) + Tcw.t[i];
        double h[3];
        for (int i = 0; i < 3; i++) h[i] = (K[i * 3] * pc[0] + K[i * 3 + 1] * pc[1]) + K[i * 3 + 2] * pc[2];
        const double u = h[0] / h[2], v = h[1] / h[2];
        if (k == 0) { mnx = mxx = u; mny = mxy = v; }
        else { mnx = std::min(mnx, u); mxx = std::max(mxx, u); mny = std::min(mny, v); mxy = std::max(mxy, v); }
    }
    out[0] = (mxx + mnx) / 2; out[1] = (mxy + mny) / 2; out[2] = mxx - mnx; out[3] = mxy - mny;
}

// RobustKernelHuber::robustify robust_kernel_impl.cpp:78-91
static inline void huber_rho(double e, double delta, double *rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}

} // namespace
#endif
