// ref_g2o_api.cpp -- TEST INFRASTRUCTURE.  C entry points over the reference's own pose code, compiled from /root/reference where it lies:
//   * Thirdparty/g2o/g2o/types/se3quat.h + se3_ops.h(pp) WHOLE (SE3Quat: exp, log, *, inverse, normalizeRotation, toVector ...), against
//     oracle/ref_shim/eigen_mini (a stand-in for Eigen's fixed-size interface; Eigen is absent here);
//   * members of class g2o::cuboid (exp_update, cube_log_error, min_log_error, rotate_cuboid, transform_from / _to) cut out of
//     include/g2o_Object.h, and exptwist_norollpitch / cuboid::point_boundary_error cut out of src/g2o_Object.cpp, by extract_ref.py at build time.
// tests/test_ref_pins.py compares the oracle's restatement (oracle/se3_util.h, ba_oracle.cpp) with these.
#include "ref_g2o_types.hpp"

namespace g2o {
using namespace std;
#include "extracted_g2o_cpp.inc"
} // namespace g2o

namespace {
g2o::SE3Quat se3(const double *v) { g2o::Vector7d x; for (int i = 0; i < 7; i++) x[i] = v[i]; return g2o::SE3Quat(x); } // SE3Quat(Vector7d): fromVector + normalizeRotation
void put7(const g2o::SE3Quat &T, double *o) { const g2o::Vector7d x = T.toVector(); for (int i = 0; i < 7; i++) o[i] = x[i]; }
g2o::cuboid cub(const double *v) { g2o::cuboid c; c.pose = se3(v); for (int i = 0; i < 3; i++) c.scale[i] = v[7 + i]; return c; }
void put10(const g2o::cuboid &c, double *o) { put7(c.pose, o); for (int i = 0; i < 3; i++) o[7 + i] = c.scale[i]; }
} // namespace

extern "C" {
void ref_se3_exp(const double *u6, double *out7) { Vector6d u; for (int i = 0; i < 6; i++) u[i] = u6[i]; put7(g2o::SE3Quat::exp(u), out7); }
void ref_se3_log(const double *p7, double *out6) { const Vector6d l = se3(p7).log(); for (int i = 0; i < 6; i++) out6[i] = l[i]; }
void ref_se3_mul(const double *a7, const double *b7, double *out7) { put7(se3(a7) * se3(b7), out7); }
void ref_se3_inverse(const double *a7, double *out7) { put7(se3(a7).inverse(), out7); }
void ref_se3_map(const double *a7, const double *p3, double *out3) { const Eigen::Vector3d r = se3(a7) * Eigen::Vector3d(p3[0], p3[1], p3[2]); for (int i = 0; i < 3; i++) out3[i] = r[i]; }
void ref_exptwist_norollpitch(const double *u6, double *out7) { Vector6d u; for (int i = 0; i < 6; i++) u[i] = u6[i]; put7(g2o::exptwist_norollpitch(u), out7); }
void ref_cuboid_exp_update(const double *c10, const double *u9, double *out10) { Vector9d u; for (int i = 0; i < 9; i++) u[i] = u9[i]; g2o::cuboid c = cub(c10); put10(c.exp_update(u), out10); }
void ref_cuboid_min_log_error(const double *self10, const double *new10, double *out9) { const Vector9d e = cub(self10).min_log_error(cub(new10)); for (int i = 0; i < 9; i++) out9[i] = e[i]; }
void ref_cuboid_cube_log_error(const double *self10, const double *new10, double *out9) { const Vector9d e = cub(self10).cube_log_error(cub(new10)); for (int i = 0; i < 9; i++) out9[i] = e[i]; }
void ref_cuboid_rotate(const double *c10, double yaw, double *out10) { put10(cub(c10).rotate_cuboid(yaw), out10); }
void ref_cuboid_transform_from(const double *c10, const double *Twc7, double *out10) { put10(cub(c10).transform_from(se3(Twc7)), out10); }
void ref_cuboid_transform_to(const double *c10, const double *Twc7, double *out10) { put10(cub(c10).transform_to(se3(Twc7)), out10); }
void ref_project_bbox(const double *c10, const double *Tcw7, const double *K9, double *out4) { // cuboid::projectOntoImageBbox g2o_Object.h:197-205
    Eigen::Matrix3d K;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K(i, j) = K9[i * 3 + j];
    const Eigen::Vector4d b = cub(c10).projectOntoImageBbox(se3(Tcw7), K);
    for (int i = 0; i < 4; i++) out4[i] = b[i];
}
void ref_point_boundary_error(const double *c10, const double *p3, double ratio, double *out3) {
    const Eigen::Vector3d e = cub(c10).point_boundary_error(Eigen::Vector3d(p3[0], p3[1], p3[2]), ratio);
    for (int i = 0; i < 3; i++) out3[i] = e[i];
}
}
