// TEST INFRASTRUCTURE (never linked into the product): the reference's LBD descriptor -- cv::line_descriptor::BinaryDescriptor's compute path
// (line_lbd/libs/binary_descriptor.cpp: the band weights of the constructor :218-260, computeGaussianPyramid :352-370, computeSobel :373-402,
// binaryConversion :405-416, compute :588-593, computeImpl :603-790, computeLBD :1146-1509), cut out of the reference at build time
// (oracle/_ref/extracted_lbd.inc, extract_ref.py) and compiled against the class declaration of the reference's own header.  The rest of that file
// is the EDLine detector, which CubeSLAM does not use (line_lbd_detect runs LSD); its constructor / destructor and the virtual functions of
// BinaryDescriptor that are not on the path are empty here.  cv::GaussianBlur (8-bit, 5 x 5, sigma 1) and cv::Sobel (3 x 3 -> CV_16S) are the
// OpenCV stand-in's (cvshim.cpp): OpenCV itself is absent, so those two primitives stay restated -- everything above them is the reference's.
// tests/test_ref_pins.py compares the oracle's descriptors with these byte for byte.
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <stdexcept>
#include <vector>

#include "cvshim.hpp"
#include "../oracle.h"
#define private public
#define protected public
#include "line_lbd/line_descriptor.hpp"
#undef private
#undef protected

#define NUM_OF_BANDS 9 // binary_descriptor.cpp:57

namespace cv {
namespace line_descriptor {
#include "extracted_lbd.inc"

// off the path: never called
void BinaryDescriptor::read(const cv::FileNode &) {}
void BinaryDescriptor::write(cv::FileStorage &) const {}
void BinaryDescriptor::operator()(InputArray, InputArray, std::vector<KeyLine> &, OutputArray, bool, bool) {}
void BinaryDescriptor::detectImpl(const Mat &, std::vector<KeyLine> &, std::vector<std::vector<KeyLine>> &, const Mat &) const {}
BinaryDescriptor::EDLineDetector::EDLineDetector() {}
BinaryDescriptor::EDLineDetector::~EDLineDetector() {}
} // namespace line_descriptor
} // namespace cv

extern "C" {
// line_lbd_detect::get_line_descriptors (line_lbd_allclass.cpp:192-198): lbd->compute(gray, keylines, descriptors) with default parameters.
// desc: n x 32 bytes.  Returns the number of descriptor rows, -1 when the reference produced another shape.
int ref_lbd_compute(const uint8_t *gray, int W, int H, const orc_keyline *kl, int n, uint8_t *desc) {
    using namespace cv::line_descriptor;
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<KeyLine> lines((size_t)n);
    for (int i = 0; i < n; i++) {
        KeyLine &k = lines[i]; const orc_keyline &o = kl[i];
        k.angle = o.angle; k.class_id = o.class_id; k.octave = o.octave; k.pt.x = o.pt_x; k.pt.y = o.pt_y; k.response = o.response; k.size = o.size;
        k.startPointX = o.startPointX; k.startPointY = o.startPointY; k.endPointX = o.endPointX; k.endPointY = o.endPointY;
        k.sPointInOctaveX = o.sPointInOctaveX; k.sPointInOctaveY = o.sPointInOctaveY; k.ePointInOctaveX = o.ePointInOctaveX; k.ePointInOctaveY = o.ePointInOctaveY;
        k.lineLength = o.lineLength; k.numOfPixels = o.numOfPixels;
    }
    BinaryDescriptor bd{BinaryDescriptor::Params()};
    cv::Mat d;
    bd.compute(img, lines, d, false);
    if (d.rows != n || d.cols != 32 || d.type() != CV_8UC1) return -1;
    for (int i = 0; i < n; i++) std::memcpy(desc + (size_t)i * 32, d.ptr<uchar>(i), 32);
    return d.rows;
}
// the band weights (constructor :233-259): gaussCoefL_ (21 doubles), gaussCoefG_ (63 doubles)
void ref_lbd_weights(double *coefL21, double *coefG63) {
    using namespace cv::line_descriptor;
    BinaryDescriptor bd{BinaryDescriptor::Params()};
    for (int i = 0; i < 21; i++) coefL21[i] = bd.gaussCoefL_[i];
    for (int i = 0; i < 63; i++) coefG63[i] = bd.gaussCoefG_[i];
}
}
