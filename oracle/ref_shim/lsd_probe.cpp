// lsd_probe.cpp -- TEST INFRASTRUCTURE.  The reference's line_lbd/libs/lsd.cpp, compiled where it lies with its private members opened, so
// that tests/test_ref_pins.py can also compare the intermediate maps (scaled image, level-line angles, gradient norms, used map) of
// LineSegmentDetectorImpl with the oracle's.  Replaces a separate lsd.o in libref.so (same translation unit, same code).
#include "precomp.hpp" // every header lsd.cpp pulls in, before the access specifiers are touched
#include <vector>

#define private public
#define protected public
#include "lsd.cpp"
#undef private
#undef protected

extern "C" int ref_lsd_maps(const uint8_t *gray, int W, int H, int *sw, int *sh, double *scaled, double *modgrad, double *angles, uint8_t *used) {
    cv::line_descriptor::LineSegmentDetectorImpl d(cv::line_descriptor::LSD_REFINE_ADV);
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<cv::Vec4f> lines;
    d.detect(img, lines);
    *sw = d.img_width; *sh = d.img_height;
    const size_t n = (size_t)d.img_width * d.img_height;
    for (size_t i = 0; i < n; i++) {
        if (scaled) scaled[i] = d.scaled_image.ptr<double>(0)[i];
        if (modgrad) modgrad[i] = d.modgrad.ptr<double>(0)[i];
        if (angles) angles[i] = d.angles.ptr<double>(0)[i];
        if (used) used[i] = d.used.ptr<uchar>(0)[i];
    }
    return (int)lines.size();
}
