// cvshim.cpp -- TEST INFRASTRUCTURE (see cvshim.hpp).  The OpenCV primitives the reference calls, forwarded to the oracle's restatements.
#include "cvshim.hpp"

#include "../cv_prims.h"
#include "../oracle.h"

namespace cv {

float fastAtan2(float y, float x) { return orc_fast_atan2(y, x); }

void Mat::copyTo(const _OutputArray &o) const {
    if (o.m) { copyTo(*o.m); return; }
    if (o.v4f) { o.v4f->resize(rows); for (int r = 0; r < rows; r++) (*o.v4f)[r] = *ptr<Vec4f>(r); return; }
    if (o.v4i) { o.v4i->resize(rows); for (int r = 0; r < rows; r++) (*o.v4i)[r] = *ptr<Vec4i>(r); return; }
    if (o.vd) { o.vd->resize(rows); for (int r = 0; r < rows; r++) (*o.vd)[r] = *ptr<double>(r); return; }
    if (o.vf) { o.vf->resize(rows); for (int r = 0; r < rows; r++) (*o.vf)[r] = *ptr<float>(r); return; }
}
void Mat::convertTo(Mat &m, int rtype) const {
    Mat out(rows, cols, rtype);
    CV_Assert(channels() == 1);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            double v;
            switch (depth()) {
            case CV_8U: v = at<uchar>(r, c); break;
            case CV_16S: v = at<short>(r, c); break;
            case CV_32S: v = at<int>(r, c); break;
            case CV_32F: v = at<float>(r, c); break;
            default: v = at<double>(r, c); break;
            }
            switch (CV_MAT_DEPTH(rtype)) {
            case CV_8U: out.at<uchar>(r, c) = saturate_cast<uchar>(v); break;
            case CV_32F: out.at<float>(r, c) = (float)v; break;
            case CV_32S: out.at<int>(r, c) = cvRound(v); break;
            default: out.at<double>(r, c) = v; break;
            }
        }
    m = out;
}
Mat &Mat::setTo(const Scalar &s) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols * channels(); c++) {
            const double v = s.val[c % channels()];
            switch (depth()) {
            case CV_8U: ptr<uchar>(r)[c] = saturate_cast<uchar>(v); break;
            case CV_32F: ptr<float>(r)[c] = (float)v; break;
            case CV_32S: ptr<int>(r)[c] = (int)v; break;
            default: ptr<double>(r)[c] = v; break;
            }
        }
    return *this;
}

void KeyPointsFilter::retainBest(std::vector<KeyPoint> &kps, int n) { // only reached by ComputeKeyPointsOld, which nothing calls
    if (n >= 0 && (int)kps.size() > n) {
        std::stable_sort(kps.begin(), kps.end(), [](const KeyPoint &a, const KeyPoint &b) { return a.response > b.response; });
        kps.resize(n);
    }
}

static std::vector<uchar> packed_u8(const Mat &m) {
    std::vector<uchar> v((size_t)m.rows * m.cols);
    for (int r = 0; r < m.rows; r++) std::memcpy(v.data() + (size_t)r * m.cols, m.ptr(r), m.cols);
    return v;
}

void resize(InputArray _src, OutputArray _dst, Size dsize, double fx, double fy, int interpolation) {
    const Mat src = _src.getMat();
    CV_Assert(interpolation == INTER_LINEAR && src.channels() == 1);
    double scale_x, scale_y; // imgwarp.cpp resize(): factors are kept when dsize is empty, derived from the sizes otherwise
    if (dsize.width == 0) { dsize = Size(saturate_cast<int>(src.cols * fx), saturate_cast<int>(src.rows * fy)); scale_x = 1. / fx; scale_y = 1. / fy; }
    else { scale_x = 1. / ((double)dsize.width / src.cols); scale_y = 1. / ((double)dsize.height / src.rows); }
    _dst.create(dsize, src.type());
    Mat dst = _dst.getMat();
    if (src.depth() == CV_8U) {
        const std::vector<uchar> s = packed_u8(src);
        std::vector<uchar> d((size_t)dsize.width * dsize.height);
        orc_cv::resize_linear_u8(s.data(), src.cols, src.rows, d.data(), dsize.width, dsize.height);
        for (int r = 0; r < dsize.height; r++) std::memcpy(dst.ptr(r), d.data() + (size_t)r * dsize.width, dsize.width);
    } else {
        CV_Assert(src.depth() == CV_64F && src.isContinuous() && dst.isContinuous());
        orc_cv::resize_linear_f64(src.ptr<double>(), src.cols, src.rows, dst.ptr<double>(), dsize.width, dsize.height, scale_x, scale_y);
    }
}
void GaussianBlur(InputArray _src, OutputArray _dst, Size ksize, double sigmaX, double sigmaY, int borderType) {
    const Mat src = _src.getMat().clone(); // in-place calls
    CV_Assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101 && src.channels() == 1 && (sigmaY == 0 || sigmaY == sigmaX));
    _dst.create(src.size(), src.type());
    Mat dst = _dst.getMat();
    if (src.depth() == CV_8U && ksize.width == 5) { // BinaryDescriptor::computeGaussianPyramid (binary_descriptor.cpp:359): 5 x 5, sigma 1, 8-bit fixed point
        CV_Assert(ksize.height == 5 && sigmaX == 1);
        const std::vector<uchar> s = packed_u8(src);
        std::vector<uchar> d(s.size());
        std::vector<int16_t> gx(s.size()), gy(s.size());
        orc_lbd_maps(s.data(), src.cols, src.rows, d.data(), gx.data(), gy.data());
        for (int r = 0; r < src.rows; r++) std::memcpy(dst.ptr(r), d.data() + (size_t)r * src.cols, src.cols);
    } else if (src.depth() == CV_8U) {
        CV_Assert(ksize.width == 7 && ksize.height == 7 && sigmaX == 2);
        const std::vector<uchar> s = packed_u8(src);
        std::vector<uchar> d(s.size());
        orc_cv::gaussian_blur7_u8(s.data(), src.cols, src.rows, d.data());
        for (int r = 0; r < src.rows; r++) std::memcpy(dst.ptr(r), d.data() + (size_t)r * src.cols, src.cols);
    } else {
        CV_Assert(src.depth() == CV_64F && src.isContinuous() && dst.isContinuous() && ksize.width == ksize.height);
        orc_cv::gaussian_blur_f64(src.ptr<double>(), src.cols, src.rows, ksize.width, sigmaX, dst.ptr<double>());
    }
}
// cv::Sobel as BinaryDescriptor::computeSobel calls it (binary_descriptor.cpp:396-397): 8-bit source, CV_16S result, 3 x 3, first derivative in x or
// y -- the separable kernels [-1 0 1] and [1 2 1], no scaling, BORDER_REFLECT_101.  Written from the definition (the oracle's own maps come from
// lbd_oracle.cpp; tests/test_ref_pins.py has the two agree).
void Sobel(InputArray _src, OutputArray _dst, int ddepth, int dx, int dy, int ksize, double scale, double delta, int borderType) {
    const Mat src = _src.getMat().clone();
    CV_Assert(src.type() == CV_8UC1 && CV_MAT_DEPTH(ddepth) == CV_16S && ksize == 3 && scale == 1 && delta == 0 && dx + dy == 1 && (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    _dst.create(src.rows, src.cols, CV_16SC1);
    Mat dst = _dst.getMat();
    const std::vector<uchar> s = packed_u8(src);
    const int W = src.cols, H = src.rows;
    for (int y = 0; y < H; y++) {
        const int ym = orc_cv::reflect101(y - 1, H), yp = orc_cv::reflect101(y + 1, H);
        for (int x = 0; x < W; x++) {
            const int xm = orc_cv::reflect101(x - 1, W), xp = orc_cv::reflect101(x + 1, W);
            auto px = [&](int yy, int xx) { return (int)s[(size_t)yy * W + xx]; };
            const int v = dx ? (px(ym, xp) - px(ym, xm)) + 2 * (px(y, xp) - px(y, xm)) + (px(yp, xp) - px(yp, xm))
                             : (px(yp, xm) - px(ym, xm)) + 2 * (px(yp, x) - px(ym, x)) + (px(yp, xp) - px(ym, xp));
            dst.at<short>(y, x) = (short)v;
        }
    }
}
void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType, const Scalar &) {
    const Mat src = _src.getMat();
    CV_Assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101 && src.type() == CV_8UC1);
    // BORDER_ISOLATED: the source is treated as a whole image even when it is a view (ORBextractor.cc:1114 fills the frame around a view of dst itself)
    _dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
    Mat dst = _dst.getMat();
    const std::vector<uchar> s = packed_u8(src);
    for (int y = 0; y < dst.rows; y++) {
        const int sy = orc_cv::reflect101(y - top, src.rows);
        for (int x = 0; x < dst.cols; x++) dst.at<uchar>(y, x) = s[(size_t)sy * src.cols + orc_cv::reflect101(x - left, src.cols)];
    }
}
void FAST(InputArray _image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression) {
    const Mat img = _image.getMat();
    CV_Assert(nonmaxSuppression && img.type() == CV_8UC1);
    std::vector<float> xyr((size_t)img.rows * img.cols * 3 + 3);
    const int n = orc_fast(img.data, (int)img.step, img.cols, img.rows, threshold, xyr.data(), img.rows * img.cols + 1);
    keypoints.clear();
    for (int i = 0; i < n; i++) keypoints.push_back(KeyPoint(xyr[3 * i], xyr[3 * i + 1], 7.f, -1, xyr[3 * i + 2]));
}
void cvtColor(InputArray _src, OutputArray _dst, int code, int) {
    const Mat src = _src.getMat();
    CV_Assert(code == CV_BGR2GRAY && src.type() == CV_8UC3 && src.isContinuous());
    _dst.create(src.size(), CV_8UC1);
    Mat dst = _dst.getMat();
    orc_bgr2gray(src.data, src.cols, src.rows, dst.data);
}
void Canny(InputArray _image, OutputArray _edges, double t1, double t2, int apertureSize, bool L2gradient) {
    const Mat img = _image.getMat();
    CV_Assert(apertureSize == 3 && !L2gradient && img.type() == CV_8UC1);
    _edges.create(img.size(), CV_8UC1);
    Mat e = _edges.getMat();
    // cv::Canny on a view reads the parent image around it (Sobel with BORDER_REPLICATE at the PARENT's border only when the view touches
    // it): the oracle's ROI entry takes the parent and the window.
    CV_Assert(img.datastart && img.step == (size_t)img.whole_cols);
    const size_t off = (size_t)(img.data - img.datastart);
    std::vector<uchar> edges((size_t)img.cols * img.rows);
    orc_canny_roi(img.datastart, img.whole_cols, img.whole_rows, (int)(off % img.step), (int)(off / img.step), img.cols, img.rows, (int)t1, (int)t2, edges.data());
    for (int r = 0; r < img.rows; r++) std::memcpy(e.ptr(r), edges.data() + (size_t)r * img.cols, img.cols);
}
void distanceTransform(InputArray _src, OutputArray _dst, int distanceType, int maskSize, int) {
    const Mat src = _src.getMat();
    CV_Assert(distanceType == DIST_L2 && maskSize == 3 && src.type() == CV_8UC1);
    _dst.create(src.size(), CV_32FC1);
    Mat dst = _dst.getMat();
    const std::vector<uchar> s = packed_u8(src);
    orc_dist_transform_3x3(s.data(), src.cols, src.rows, dst.ptr<float>());
}
void pyrDown(InputArray, OutputArray, const Size &, int) { throw std::runtime_error("cvshim: pyrDown is not on the measured path (numOctaves = 1)"); }
void line(InputOutputArray, Point, Point, const Scalar &, int, int, int) {}
void merge(const std::vector<Mat> &, OutputArray) { throw std::runtime_error("cvshim: merge (drawing) is not on the measured path"); }
void bitwise_xor(InputArray, InputArray, OutputArray) { throw std::runtime_error("cvshim: bitwise_xor (drawing) is not on the measured path"); }
int countNonZero(InputArray) { throw std::runtime_error("cvshim: countNonZero (drawing) is not on the measured path"); }

} // namespace cv
