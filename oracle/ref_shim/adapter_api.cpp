// adapter_api.cpp -- TEST INFRASTRUCTURE.  C entry points that drive adapters/ORBextractor_hip.cc and adapters/line_lbd_allclass_hip.cpp
// through the REFERENCE'S class interfaces (ORB_SLAM2::ORBextractor, line_lbd_detect; the class definitions come from the reference's
// headers), so that tests/test_adapters_gpu.py can run the drop-in path next to the reference's own translation units (oracle/_ref/libref.so)
// on the MI355X.  oracle/_ref/libadapters.so = the two adapters + this file + the OpenCV stand-in, linked against libcubeslam_hip.so.
#include <opencv2/core/core.hpp>

#include "ORBextractor.h"
#include "line_lbd/line_lbd_allclass.h"

#include "../oracle.h"

// members of the reference's line library that the adapter's constructor touches but this test never calls (binary_descriptor*.cpp are not
// part of the _ref build)
namespace cv { namespace line_descriptor {
BinaryDescriptor::Params::Params() { numOfOctave_ = 1; widthOfBand_ = 7; reductionRatio = 2; ksize_ = 5; Octave_ratio = 2; }
Ptr<BinaryDescriptor> BinaryDescriptor::createBinaryDescriptor(Params) { return Ptr<BinaryDescriptor>(); }
Ptr<BinaryDescriptorMatcher> BinaryDescriptorMatcher::createBinaryDescriptorMatcher() { return Ptr<BinaryDescriptorMatcher>(); }
void BinaryDescriptor::detect(const Mat &, std::vector<KeyLine> &, const Mat &) { throw std::runtime_error("EDLine path: not part of the _ref build"); }
}} // namespace cv::line_descriptor

extern "C" {
int adp_orb_extract(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, const uint8_t *gray, int W, int H, orc_keypoint *kps, uint8_t *desc, int cap,
                    uint8_t *levels_out, int *level_dims, float *tables /* 4 x nlevels */) {
    ORB_SLAM2::ORBextractor ext(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<cv::KeyPoint> k;
    cv::Mat d;
    ext(img, cv::Mat(), k, d);
    const int n = std::min((int)k.size(), cap);
    for (int i = 0; i < n; i++) {
        kps[i].x = k[i].pt.x; kps[i].y = k[i].pt.y; kps[i].size = k[i].size; kps[i].angle = k[i].angle; kps[i].response = k[i].response;
        kps[i].octave = k[i].octave; kps[i].class_id = k[i].class_id;
        std::memcpy(desc + 32 * (size_t)i, d.ptr(i), 32);
    }
    size_t off = 0;
    for (int l = 0; l < nlevels; l++) {
        const cv::Mat &m = ext.mvImagePyramid[l];
        if (level_dims) { level_dims[2 * l] = m.cols; level_dims[2 * l + 1] = m.rows; }
        if (levels_out) for (int r = 0; r < m.rows; r++) { std::memcpy(levels_out + off, m.ptr(r), m.cols); off += m.cols; }
    }
    if (tables) {
        const std::vector<float> t0 = ext.GetScaleFactors(), t1 = ext.GetInverseScaleFactors(), t2 = ext.GetScaleSigmaSquares(), t3 = ext.GetInverseScaleSigmaSquares();
        for (int l = 0; l < nlevels; l++) { tables[l] = t0[l]; tables[nlevels + l] = t1[l]; tables[2 * nlevels + l] = t2[l]; tables[3 * nlevels + l] = t3[l]; }
    }
    extern void ORBextractor_hip_release(const void *);
    ORBextractor_hip_release(&ext);
    return (int)k.size();
}
// line_lbd_detect with use_LSD = true, line_length_thres as in main_obj.cpp:361-366: raw KeyLines, and detect_filter_lines' n x 4 matrix
int adp_lsd_keylines(const uint8_t *gray, int W, int H, float line_length_thres, orc_keyline *out, int cap, float *filtered /* cap x 4 */, int *n_filtered) {
    line_lbd_detect det;
    det.use_LSD = true;
    det.line_length_thres = line_length_thres;
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<KeyLine> kl;
    det.detect_raw_lines(img, kl);
    const int m = std::min((int)kl.size(), cap);
    for (int i = 0; i < m; i++) {
        const KeyLine &k = kl[i];
        orc_keyline &o = out[i];
        o.angle = k.angle; o.class_id = k.class_id; o.octave = k.octave; o.pt_x = k.pt.x; o.pt_y = k.pt.y; o.response = k.response; o.size = k.size;
        o.startPointX = k.startPointX; o.startPointY = k.startPointY; o.endPointX = k.endPointX; o.endPointY = k.endPointY;
        o.sPointInOctaveX = k.sPointInOctaveX; o.sPointInOctaveY = k.sPointInOctaveY; o.ePointInOctaveX = k.ePointInOctaveX; o.ePointInOctaveY = k.ePointInOctaveY;
        o.lineLength = k.lineLength; o.numOfPixels = k.numOfPixels;
    }
    cv::Mat lines;
    det.detect_filter_lines(img, lines);
    *n_filtered = lines.rows;
    for (int i = 0; i < lines.rows && i < cap; i++) for (int j = 0; j < 4; j++) filtered[4 * i + j] = lines.at<float>(i, j);
    return (int)kl.size();
}
}
