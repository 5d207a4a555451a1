// ORBextractor_hip.cc -- replaces orb_object_slam/src/ORBextractor.cc in the reference's build: ORB_SLAM2::ORBextractor with the
// reference's own class definition (include/ORBextractor.h:44-112) on top of libcubeslam_hip.so.  Same constructor arguments, same
// operator() (key points in level-major order, 32-byte descriptors, mvImagePyramid filled for Frame.cc / the viewer), same level tables.
//
// The reference class has no member to hang a device handle on, so the extractor's cs_orb lives in a side table keyed by `this`
// (the reference's destructor is inline and empty: handles are released at process exit or by ORBextractor_hip_release()).
#include "ORBextractor.h"

#include <map>
#include <mutex>
#include <stdexcept>

#include "cubeslam_hip.h"

namespace ORB_SLAM2 {
namespace {
struct Dev { cs_ctx *ctx = nullptr; cs_orb *orb = nullptr; int w = 0, h = 0; };
std::mutex g_mu;
std::map<const ORBextractor *, Dev> g_dev;
// One context (device + stream) PER EXTRACTOR: a cs_ctx is not thread-safe, and the stereo Frame constructor runs the left and the right extractor in
// two concurrent threads (Frame.cc:106-107).  An extractor itself is only ever called by one thread at a time.
cs_ctx *new_ctx() {
    cs_ctx *ctx = nullptr;
    if (cs_create(0, &ctx) != CS_OK) throw std::runtime_error("ORBextractor (HIP): no device -- there is no CPU path");
    return ctx;
}
} // namespace

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    // ORBextractor.cc:412-471: the level tables; computed by the library with the reference's float arithmetic (cs_orb_create) and read
    // back, so that GetScaleFactors() etc. return the same numbers the extraction uses.  The device buffers are sized at the first frame.
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    mvImagePyramid.resize(nlevels);
    cs_orb *probe = nullptr;
    cs_ctx *my_ctx = new_ctx();
    { // (an extractor constructed at the address of one that was never released: its stale device side goes first)
        std::lock_guard<std::mutex> lk(g_mu);
        Dev &e = g_dev[this];
        if (e.orb) cs_orb_destroy(e.ctx, e.orb);
        if (e.ctx) cs_destroy(e.ctx);
        e = Dev();
        e.ctx = my_ctx;
    }
    if (cs_orb_create(my_ctx, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, 64, 64, 1, &probe) != CS_OK)
        throw std::runtime_error(std::string("ORBextractor (HIP): ") + cs_last_error(my_ctx));
    cs_orb_get_table(probe, 0, mvScaleFactor.data()); cs_orb_get_table(probe, 1, mvInvScaleFactor.data());
    cs_orb_get_table(probe, 2, mvLevelSigma2.data()); cs_orb_get_table(probe, 3, mvInvLevelSigma2.data());
    cs_orb_get_table(probe, 4, mnFeaturesPerLevel.data());
    cs_orb_destroy(my_ctx, probe);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask: unused by the reference as well*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    Dev d;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Dev &e = g_dev[this];
        if (!e.orb || e.w != image.cols || e.h != image.rows) { // first frame, or the camera changed resolution
            if (e.orb) cs_orb_destroy(e.ctx, e.orb);
            if (!e.ctx) e.ctx = new_ctx(); // (an extractor copied from another one)
            e.w = image.cols; e.h = image.rows; e.orb = nullptr;
            if (cs_orb_create(e.ctx, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, e.w, e.h, 1, &e.orb) != CS_OK)
                throw std::runtime_error(std::string("ORBextractor (HIP): ") + cs_last_error(e.ctx));
        }
        d = e;
    }
    const int cap = 2 * nfeatures + 64; // DistributeOctTree may return a few more than the quota (ORBextractor.cc:640-722)
    std::vector<cs_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    if (cs_orb_extract(d.ctx, d.orb, image.data, 1, (int)image.step, kps.data(), desc.data(), cap, &n) != CS_OK)
        throw std::runtime_error(std::string("ORBextractor (HIP): ") + cs_last_error(d.ctx));
    _keypoints.clear();
    _keypoints.reserve(n);
    for (int i = 0; i < n; i++) {
        cv::KeyPoint k;
        k.pt.x = kps[i].x; k.pt.y = kps[i].y; k.size = kps[i].size; k.angle = kps[i].angle; k.response = kps[i].response; k.octave = kps[i].octave; k.class_id = kps[i].class_id;
        _keypoints.push_back(k);
    }
    if (n == 0) _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat out = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(out.ptr(i), desc.data() + (size_t)i * 32, 32);
    }
    // mvImagePyramid (ORBextractor.h:85; read by the stereo matcher in Frame.cc): the levels the extraction used, without the 19-px frame
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0;
        cs_orb_get_level(d.ctx, d.orb, 0, l, 0, nullptr, &w, &h);
        mvImagePyramid[l].create(h, w, CV_8UC1);
        cs_orb_get_level(d.ctx, d.orb, 0, l, 0, mvImagePyramid[l].data, &w, &h);
    }
}

// The remaining protected members are steps of the reference's CPU implementation; they are kept only so that the class links.
void ORBextractor::ComputePyramid(cv::Mat) {}
void ORBextractor::ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint>> &) {}
void ORBextractor::ComputeKeyPointsOld(std::vector<std::vector<cv::KeyPoint>> &) {}
std::vector<cv::KeyPoint> ORBextractor::DistributeOctTree(const std::vector<cv::KeyPoint> &v, const int &, const int &, const int &, const int &, const int &, const int &) { return v; }
void ExtractorNode::DivideNode(ExtractorNode &, ExtractorNode &, ExtractorNode &, ExtractorNode &) {}

} // namespace ORB_SLAM2

extern "C" void ORBextractor_hip_release(const void *extractor) { // optional: free the device buffers of one extractor
    std::lock_guard<std::mutex> lk(ORB_SLAM2::g_mu);
    auto it = ORB_SLAM2::g_dev.find(static_cast<const ORB_SLAM2::ORBextractor *>(extractor));
    if (it != ORB_SLAM2::g_dev.end()) { if (it->second.orb) cs_orb_destroy(it->second.ctx, it->second.orb); if (it->second.ctx) cs_destroy(it->second.ctx); ORB_SLAM2::g_dev.erase(it); }
}
