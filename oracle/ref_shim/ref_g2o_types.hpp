// ref_g2o_types.hpp -- TEST INFRASTRUCTURE.  The reference's pose types as the translation units of oracle/_ref see them: the vendored
// Thirdparty/g2o/g2o/types/se3quat.h WHOLE (against oracle/ref_shim/eigen_mini) and class g2o::cuboid = the two data members and the default
// constructor of include/g2o_Object.h:29-35 around the member functions cut out of that header at build time (extracted_g2o_members.inc).
// exptwist_norollpitch / cuboid::point_boundary_error (src/g2o_Object.cpp, cut out too) are defined once, in ref_g2o_api.cpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <iostream>

#include "Thirdparty/g2o/g2o/types/se3quat.h"

typedef Eigen::Matrix<double, 9, 1> Vector9d;
typedef Eigen::Matrix<double, 10, 1> Vector10d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;

using namespace Eigen;
#include "extracted_g2o_utils.inc"

namespace g2o {
using namespace Eigen;
class cuboid { // g2o_Object.h:29-35: the two data members and the default constructor; the member functions below are the reference's text
  public:
    SE3Quat pose;
    Vector3d scale;
    cuboid() { pose = SE3Quat(); scale.setZero(); }
    inline const Vector3d &translation() const { return pose.translation(); }
    inline void setTranslation(const Vector3d &t_) { pose.setTranslation(t_); }
#include "extracted_g2o_members.inc"
    Vector3d point_boundary_error(const Vector3d &point, const double max_outside_margin_ratio, double point_scale = 1) const;
};
SE3Quat exptwist_norollpitch(const Vector6d &update);
} // namespace g2o
