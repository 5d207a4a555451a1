// cv_prims.h -- TEST INFRASTRUCTURE.  The OpenCV 3.x primitives the reference's hot path calls (cv::resize, GaussianBlur, FAST,
// copyMakeBorder, Canny, distanceTransform, cvtColor), as restated inside oracle/*.cpp, exposed so that oracle/ref_shim/ can stand in for
// OpenCV when the reference's own translation units are compiled into oracle/_ref/ (OpenCV itself is absent here; Appendix B of
// these semantics are recalled, not verified).  Not used by the product.
#pragma once
#include <cstdint>
#include <vector>

namespace orc_cv {
void resize_linear_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);      // orb_oracle.cpp
void gaussian_blur7_u8(const uint8_t *src, int w, int h, uint8_t *dst);                       // orb_oracle.cpp: 7x7, sigma 2, REFLECT_101
int reflect101(int p, int len);
// lsd_oracle.cpp: GaussianBlur on CV_64F (separable, symmetric summation) and resize(INTER_LINEAR) on CV_64F with float coefficients
void gaussian_blur_f64(const double *src, int w, int h, int ksize, double sigma, double *dst);
void resize_linear_f64(const double *src, int sw, int sh, double *dst, int dw, int dh, double scale_x, double scale_y); // scale = 1 / fx when factors were given
}
