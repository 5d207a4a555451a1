// ref_api.cpp -- TEST INFRASTRUCTURE.  C entry points over the REFERENCE'S OWN classes, compiled (oracle/Makefile.ref) from the sources
// where they lie under /root/reference against oracle/ref_shim/ (a stand-in for the OpenCV headers): ORB_SLAM2::ORBextractor
// (orb_object_slam/src/ORBextractor.cc), cv::line_descriptor::LineSegmentDetector (line_lbd/libs/lsd.cpp) and LSDDetector
// (line_lbd/libs/LSDDetector.cpp).  tests/test_ref_pins.py runs them next to the oracle's restatement on the same inputs.
#include <opencv2/core/core.hpp>

#include "ORBextractor.h"
#include "line_lbd/line_descriptor.hpp"

#include "../oracle.h"

#include <cstdlib>
#include <new>

// ---- allocation order = address order.  DistributeOctTree sorts pair<int, ExtractorNode*> (ORBextractor.cc:685): nodes holding the same
// number of points are ordered by their HEAP ADDRESS, i.e. by whatever the allocator did.  Inside a ref_* call every operator new of this
// library comes from a bump arena that never recycles, so address order is creation order -- the deterministic reading of that line which
// the oracle pins as O1.  libref.so is linked -Bsymbolic: only its own allocations are affected.
namespace {
struct Arena {
    char *base = nullptr; size_t cap = 0, used = 0; bool on = false;
    void begin() { if (!base) { cap = (size_t)1 << 30; base = (char *)std::malloc(cap); } used = 0; on = true; }
    void end() { on = false; }
    bool owns(void *p) const { return base && (char *)p >= base && (char *)p < base + cap; }
} g_arena;
struct ArenaScope { ArenaScope() { g_arena.begin(); } ~ArenaScope() { g_arena.end(); } };
} // namespace
void *operator new(std::size_t n) {
    if (g_arena.on) {
        const size_t a = (g_arena.used + 15) & ~(size_t)15;
        if (a + n <= g_arena.cap) { g_arena.used = a + n; return g_arena.base + a; }
    }
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new[](std::size_t n) { return operator new(n); }
void operator delete(void *p) noexcept { if (p && !g_arena.owns(p)) std::free(p); }
void operator delete[](void *p) noexcept { operator delete(p); }
void operator delete(void *p, std::size_t) noexcept { operator delete(p); }
void operator delete[](void *p, std::size_t) noexcept { operator delete(p); }

extern "C" {

// ORBextractor::operator() (ORBextractor.cc:1036-1099).  kps: cv::KeyPoint records (28 B), desc: 32 B each.  Returns the count.
int ref_orb_extract(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, const uint8_t *gray, int W, int H, orc_keypoint *kps, uint8_t *desc,
                    int cap, uint8_t *levels_out /* optional: all pyramid levels, packed one after another */, int *level_dims /* optional: 2 per level */) {
    ArenaScope arena;
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
    return (int)k.size();
}
int ref_orb_features_per_level(int nfeatures, float scaleFactor, int nlevels, int *out) {
    ORB_SLAM2::ORBextractor ext(nfeatures, scaleFactor, nlevels, 20, 7);
    // mnFeaturesPerLevel is protected: read it through a derived accessor
    struct Peek : ORB_SLAM2::ORBextractor { using ORB_SLAM2::ORBextractor::mnFeaturesPerLevel; };
    const std::vector<int> &v = static_cast<Peek &>(ext).mnFeaturesPerLevel;
    for (int l = 0; l < nlevels; l++) out[l] = v[l];
    return nlevels;
}

// cv::line_descriptor::createLineSegmentDetector(LSD_REFINE_ADV)->detect (lsd.cpp:414-438 and everything under flsd).  7 floats per line:
// x1 y1 x2 y2 width prec nfa.  Returns the count.
int ref_lsd_segments(const uint8_t *gray, int W, int H, double *out, int cap) {
    ArenaScope arena;
    cv::Ptr<cv::line_descriptor::LineSegmentDetector> ls = cv::line_descriptor::createLineSegmentDetector(cv::line_descriptor::LSD_REFINE_ADV);
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<cv::Vec4f> lines;
    std::vector<double> w, p, n;
    ls->detect(img, lines, w, p, n);
    const int m = std::min((int)lines.size(), cap);
    for (int i = 0; i < m; i++) {
        for (int k = 0; k < 4; k++) out[7 * i + k] = lines[i][k];
        out[7 * i + 4] = w[i]; out[7 * i + 5] = p[i]; out[7 * i + 6] = n[i];
    }
    return (int)lines.size();
}
// LSDDetector::detect (LSDDetector.cpp:102-263), one octave, as line_lbd_detect::detect_raw_lines calls it (line_lbd_allclass.cpp:125-141)
int ref_lsd_keylines(const uint8_t *gray, int W, int H, orc_keyline *out, int cap) {
    ArenaScope arena;
    cv::Ptr<cv::line_descriptor::LSDDetector> lsd = cv::line_descriptor::LSDDetector::createLSDDetector();
    cv::line_descriptor::LSDDetector::LSDOptions opts;
    opts.refine = 0; opts.scale = 0; opts.sigma_scale = 0; opts.quant = 0; opts.ang_th = 0; opts.log_eps = 0; opts.density_th = 0; opts.n_bins = 0;
    opts.min_length = 0; // line_lbd_allclass.cpp:130-138: detectImpl ignores them (it builds its own LSD_REFINE_ADV detector, LSDDetector.cpp:173)
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    std::vector<cv::line_descriptor::KeyLine> kl;
    lsd->detect(img, kl, 1, 1, opts); // (int)octaveratio_ = 1, numoctaves_ = 1 (line_lbd_allclass.h:26)
    const int m = std::min((int)kl.size(), cap);
    for (int i = 0; i < m; i++) {
        const cv::line_descriptor::KeyLine &k = kl[i];
        orc_keyline &o = out[i];
        o.angle = k.angle; o.class_id = k.class_id; o.octave = k.octave; o.pt_x = k.pt.x; o.pt_y = k.pt.y; o.response = k.response; o.size = k.size;
        o.startPointX = k.startPointX; o.startPointY = k.startPointY; o.endPointX = k.endPointX; o.endPointY = k.endPointY;
        o.sPointInOctaveX = k.sPointInOctaveX; o.sPointInOctaveY = k.sPointInOctaveY; o.ePointInOctaveX = k.ePointInOctaveX; o.ePointInOctaveY = k.ePointInOctaveY;
        o.lineLength = k.lineLength; o.numOfPixels = k.numOfPixels;
    }
    return (int)kl.size();
}

} // extern "C"
