// line_lbd_allclass_hip.cpp -- replaces line_lbd/class/line_lbd_allclass.cpp in the reference's build: class line_lbd_detect with the
// reference's own definition (include/line_lbd/line_lbd_allclass.h) on top of libcubeslam_hip.so.  The LSD path (use_LSD = true, what
// object_slam/src/main_obj.cpp:361-366,428-449 uses) and the LBD descriptor / matcher run on the device; the EDLine path (use_LSD = false)
// still calls the reference's BinaryDescriptor::detect, which stays in the reference's library.
#include "line_lbd/line_lbd_allclass.h"

#include <cmath>
#include <map>
#include <mutex>
#include <stdexcept>

#include "cubeslam_hip.h"

using namespace cv;
using namespace cv::line_descriptor;

namespace {
// One context (device + stream) and one detector PER line_lbd_detect OBJECT, and a lock around its use: a cs_ctx is not thread-safe, the object may be
// called from any thread of the SLAM system and outlives the thread that first used it (a context that belonged to a thread would dangle in this table
// once that thread exits).  Released by line_lbd_detect_hip_release() or at process exit (the reference's class has no destructor to hook).
struct Dev { cs_ctx *ctx = nullptr; cs_lsd *lsd = nullptr; int w = 0, h = 0; std::mutex *use = nullptr; };
std::mutex g_mu;
std::map<const line_lbd_detect *, Dev> g_dev;
// The functions that need no detector (LBD descriptors, the matcher) take a context of the CALLING THREAD: created at the thread's first call, destroyed when
// the thread exits, never handed to another thread.
struct ThreadCtx { cs_ctx *c = nullptr; ~ThreadCtx() { if (c) cs_destroy(c); } };
cs_ctx *shared_ctx() {
    thread_local ThreadCtx t;
    if (!t.c && cs_create(0, &t.c) != CS_OK) throw std::runtime_error("line_lbd_detect (HIP): no device -- there is no CPU path");
    return t.c;
}
Dev device_for(const line_lbd_detect *self) { // the object's entry (its context is created at the first call)
    std::lock_guard<std::mutex> lk(g_mu);
    Dev &e = g_dev[self];
    if (!e.ctx) {
        if (cs_create(0, &e.ctx) != CS_OK) { e.ctx = nullptr; throw std::runtime_error("line_lbd_detect (HIP): no device -- there is no CPU path"); }
        e.use = new std::mutex();
    }
    return e;
}
void size_detector(const line_lbd_detect *self, int w, int h, Dev &d) { // called with d.use held: the detector of the frame size
    if (d.lsd && d.w == w && d.h == h) return;
    if (d.lsd) cs_lsd_destroy(d.ctx, d.lsd);
    d.lsd = nullptr; d.w = w; d.h = h;
    const int r = cs_lsd_create(d.ctx, w, h, 1, &d.lsd);
    { std::lock_guard<std::mutex> lk(g_mu); Dev &e = g_dev[self]; e.lsd = d.lsd; e.w = w; e.h = h; }
    if (r != CS_OK) throw std::runtime_error(std::string("line_lbd_detect (HIP): ") + cs_last_error(d.ctx));
}
Mat as_gray(const Mat &img) {
    if (img.channels() == 1) return img;
    Mat g;
    cvtColor(img, g, COLOR_BGR2GRAY);
    return g;
}
KeyLine to_keyline(const cs_keyline &c) {
    KeyLine k;
    k.angle = c.angle; k.class_id = c.class_id; k.octave = c.octave; k.pt = Point2f(c.pt_x, c.pt_y); k.response = c.response; k.size = c.size;
    k.startPointX = c.startPointX; k.startPointY = c.startPointY; k.endPointX = c.endPointX; k.endPointY = c.endPointY;
    k.sPointInOctaveX = c.sPointInOctaveX; k.sPointInOctaveY = c.sPointInOctaveY; k.ePointInOctaveX = c.ePointInOctaveX; k.ePointInOctaveY = c.ePointInOctaveY;
    k.lineLength = c.lineLength; k.numOfPixels = c.numOfPixels;
    return k;
}
cs_keyline from_keyline(const KeyLine &k) {
    cs_keyline c;
    c.angle = k.angle; c.class_id = k.class_id; c.octave = k.octave; c.pt_x = k.pt.x; c.pt_y = k.pt.y; c.response = k.response; c.size = k.size;
    c.startPointX = k.startPointX; c.startPointY = k.startPointY; c.endPointX = k.endPointX; c.endPointY = k.endPointY;
    c.sPointInOctaveX = k.sPointInOctaveX; c.sPointInOctaveY = k.sPointInOctaveY; c.ePointInOctaveX = k.ePointInOctaveX; c.ePointInOctaveY = k.ePointInOctaveY;
    c.lineLength = k.lineLength; c.numOfPixels = k.numOfPixels;
    return c;
}
// BinaryDescriptor::compute for octave-0 lines (binary_descriptor.cpp:588-790): n x 32 CV_8U
void lbd_compute(const Mat &gray_img, const std::vector<KeyLine> &keylines, Mat &descriptors) {
    const Mat g = as_gray(gray_img);
    const int n = (int)keylines.size();
    if (n == 0) { descriptors = Mat(); return; }
    std::vector<cs_keyline> kl(n);
    // class_id / octave of a KeyLine made by mat_to_keylines are whatever KeyLine() left there (see above); the descriptor of a line of the only octave
    // depends on neither, so the library is handed octave 0 and the running index
    for (int i = 0; i < n; i++) { kl[i] = from_keyline(keylines[i]); kl[i].octave = 0; kl[i].class_id = i; }
    descriptors.create(n, 32, CV_8UC1);
    std::vector<uint8_t> d((size_t)n * 32);
    cs_ctx *ctx = shared_ctx();
    if (cs_lbd_compute(ctx, g.data, g.cols, g.rows, (int)g.step, kl.data(), n, d.data(), nullptr) != CS_OK)
        throw std::runtime_error(std::string("line_lbd_detect (HIP): ") + cs_last_error(ctx));
    for (int i = 0; i < n; i++) std::memcpy(descriptors.ptr(i), d.data() + (size_t)i * 32, 32);
}
} // namespace

void keylines_to_mat(const std::vector<KeyLine> &keylines_src, cv::Mat &linesmat_out, float scale) { // n x 4 CV_32F: x1 y1 x2 y2
    linesmat_out.create((int)keylines_src.size(), 4, CV_32FC1);
    for (int j = 0; j < (int)keylines_src.size(); j++) {
        linesmat_out.at<float>(j, 0) = keylines_src[j].startPointX * scale; linesmat_out.at<float>(j, 1) = keylines_src[j].startPointY * scale;
        linesmat_out.at<float>(j, 2) = keylines_src[j].endPointX * scale; linesmat_out.at<float>(j, 3) = keylines_src[j].endPointY * scale;
    }
}

// the KeyLine fields of a segment given in the coordinates of its octave (the reference's fill_line_information + mat_to_keylines)
void mat_to_keylines(const cv::Mat &linesmat_src, std::vector<KeyLine> &keylines_out, int raw_img_width, int raw_img_height, float raw_length_threshold,
                     float close_boundary_threshold, float scaling, int octave_id, float each_octave_scale) {
    keylines_out.clear();
    const float octave_scale = std::pow(each_octave_scale, (float)octave_id);
    const float pre_boundary_thre = close_boundary_threshold / octave_scale, octave_length_thre = raw_length_threshold / octave_scale;
    const int oct_w = (int)(raw_img_width / octave_scale), oct_h = (int)(raw_img_height / octave_scale);
    const Mat temp_img(Size(oct_w, oct_h), CV_8UC1, Scalar(0));
    int line_ind = -1;
    for (int j = 0; j < linesmat_src.rows; j++) {
        KeyLine kl;
        kl.sPointInOctaveX = linesmat_src.at<float>(j, 0) * scaling; kl.sPointInOctaveY = linesmat_src.at<float>(j, 1) * scaling;
        kl.ePointInOctaveX = linesmat_src.at<float>(j, 2) * scaling; kl.ePointInOctaveY = linesmat_src.at<float>(j, 3) * scaling;
        if (((kl.startPointX < pre_boundary_thre) && (kl.endPointX < pre_boundary_thre)) || ((kl.startPointX > oct_w - pre_boundary_thre) && (kl.endPointX > oct_w - pre_boundary_thre)) ||
            ((kl.startPointY < pre_boundary_thre) && (kl.endPointY < pre_boundary_thre)) || ((kl.startPointY > oct_h - pre_boundary_thre) && (kl.endPointY > oct_h - pre_boundary_thre)))
            continue; // (the reference tests the not-yet-filled startPoint fields here as well)
        const float dx = kl.ePointInOctaveX - kl.sPointInOctaveX, dy = kl.ePointInOctaveY - kl.sPointInOctaveY;
        kl.lineLength = std::sqrt(dx * dx + dy * dy);
        if (kl.lineLength < octave_length_thre) continue;
        kl.startPointX = kl.sPointInOctaveX * octave_scale; kl.startPointY = kl.sPointInOctaveY * octave_scale;
        kl.endPointX = kl.ePointInOctaveX * octave_scale; kl.endPointY = kl.ePointInOctaveY * octave_scale;
        kl.pt = Point2f((kl.endPointX + kl.startPointX) / 2, (kl.endPointY + kl.startPointY) / 2);
        kl.angle = std::atan2(dy, dx);
        kl.size = std::fabs(dx * dy) * octave_scale * octave_scale;
        kl.response = kl.lineLength / (float)std::max(oct_w, oct_h);
        LineIterator li(temp_img, Point2f(kl.sPointInOctaveX, kl.sPointInOctaveY), Point2f(kl.ePointInOctaveX, kl.ePointInOctaveY));
        kl.numOfPixels = li.count;
        keylines_out.push_back(kl);
        line_ind++;
        kl.class_id = line_ind; kl.octave = octave_id; // :104-105 as written: on the local, AFTER its copy was stored (the stored KeyLine keeps what KeyLine() left in the two fields)
    }
}

line_lbd_detect::line_lbd_detect(int numoctaves, float octaveratio) : numoctaves_(numoctaves), octaveratio_(octaveratio) {
    BinaryDescriptor::Params line_params;
    line_params.numOfOctave_ = numoctaves_;
    line_params.Octave_ratio = octaveratio_;
    lbd = BinaryDescriptor::createBinaryDescriptor(line_params); // kept for the EDLine path and for callers that reach into the member
    bdm = BinaryDescriptorMatcher::createBinaryDescriptorMatcher();
    lsd = LSDDetector::createLSDDetector();
    use_LSD = false;
    line_length_thres = 50;
}

void line_lbd_detect::detect_raw_lines(const cv::Mat &gray_img, std::vector<KeyLine> &keylines_out) {
    if (use_LSD && numoctaves_ == 1) { // LSDDetector::detect(gray, keylines, (int)octaveratio_, 1, opts): LSD_REFINE_ADV, reference defaults
        const Mat g = as_gray(gray_img);
        Dev d = device_for(this);
        const int cap = 16384;
        std::vector<cs_keyline> kl(cap);
        int n = 0;
        {
            std::lock_guard<std::mutex> use(*d.use); // one caller at a time per object
            { std::lock_guard<std::mutex> lk(g_mu); d = g_dev[this]; }
            size_detector(this, g.cols, g.rows, d);
            if (cs_lsd_detect(d.ctx, d.lsd, g.data, 1, (int)g.step, kl.data(), cap, &n) != CS_OK) throw std::runtime_error(std::string("line_lbd_detect (HIP): ") + cs_last_error(d.ctx));
        }
        keylines_out.clear();
        for (int i = 0; i < n; i++) keylines_out.push_back(to_keyline(kl[i]));
    } else if (use_LSD) { // several octaves: the reference's pyramid loop (pyrDown) stays on the host
        LSDDetector::LSDOptions opts;
        opts.refine = 0; opts.scale = 0; opts.sigma_scale = 0; opts.quant = 0; opts.ang_th = 0; opts.log_eps = 0; opts.density_th = 0; opts.n_bins = 0; opts.min_length = 0;
        lsd->detect(gray_img, keylines_out, (int)octaveratio_, numoctaves_, opts);
    } else {
        cv::Mat mask1 = Mat(gray_img.size(), CV_8UC1, Scalar(1));
        lbd->detect(gray_img, keylines_out, mask1);
    }
}
void line_lbd_detect::detect_raw_lines(const cv::Mat &gray_img, std::vector<std::vector<KeyLine>> &keyline_octaves) {
    std::vector<KeyLine> all;
    detect_raw_lines(gray_img, all);
    keyline_octaves.assign(numoctaves_, std::vector<KeyLine>());
    for (const KeyLine &k : all) if (k.octave >= 0 && k.octave < numoctaves_) keyline_octaves[k.octave].push_back(k);
}
void line_lbd_detect::detect_raw_lines(const Mat &gray_img, cv::Mat &lines_mat, bool downsample_img) {
    cv::Mat gray_img2;
    if (downsample_img) cv::resize(gray_img, gray_img2, cv::Size(), 0.5, 0.5);
    else gray_img2 = gray_img;
    std::vector<KeyLine> lbd_octave;
    detect_raw_lines(gray_img2, lbd_octave);
    keylines_to_mat(lbd_octave, lines_mat, downsample_img ? 2.f : 1.f);
}
void line_lbd_detect::get_line_descriptors(const cv::Mat &gray_img, const cv::Mat &linesmat_src, cv::Mat &line_descrips) {
    std::vector<KeyLine> keylines;
    mat_to_keylines(linesmat_src, keylines, gray_img.cols, gray_img.rows);
    lbd_compute(gray_img, keylines, line_descrips);
}
void line_lbd_detect::filter_lines(std::vector<KeyLine> &keylines_in, std::vector<KeyLine> &keylines_out) {
    keylines_out.clear();
    for (const KeyLine &k : keylines_in) if (k.octave == 0 && k.lineLength > line_length_thres) keylines_out.push_back(k);
}
void line_lbd_detect::detect_filter_lines(const cv::Mat &gray_img, std::vector<KeyLine> &keylines_out) {
    std::vector<KeyLine> keylines_raw;
    detect_raw_lines(gray_img, keylines_raw);
    filter_lines(keylines_raw, keylines_out);
}
void line_lbd_detect::detect_filter_lines(const cv::Mat &gray_img, cv::Mat &linesmat_out) {
    std::vector<KeyLine> keylines;
    detect_filter_lines(gray_img, keylines);
    keylines_to_mat(keylines, linesmat_out, 1);
}
void line_lbd_detect::detect_descrip_lines(const cv::Mat &gray_img, cv::Mat &lines_mat, Mat &line_descrips) {
    std::vector<KeyLine> raw, oct0;
    detect_raw_lines(gray_img, raw);
    for (const KeyLine &k : raw) if (k.octave == 0) oct0.push_back(k);
    lbd_compute(gray_img, oct0, line_descrips);
    keylines_to_mat(oct0, lines_mat);
}
void line_lbd_detect::detect_descrip_lines(const cv::Mat &gray_img, std::vector<KeyLine> &keylines_out, cv::Mat &line_descrips) {
    std::vector<KeyLine> raw;
    detect_raw_lines(gray_img, raw);
    Mat all;
    lbd_compute(gray_img, raw, all); // descriptors of all raw lines first: the reference filters afterwards (:255-268)
    keylines_out.clear();
    std::vector<int> keep;
    for (int i = 0; i < (int)raw.size(); i++) if (raw[i].octave == 0 && raw[i].lineLength > line_length_thres) { keylines_out.push_back(raw[i]); keep.push_back(i); }
    line_descrips.create((int)keep.size(), 32, CV_8UC1);
    for (int i = 0; i < (int)keep.size(); i++) std::memcpy(line_descrips.ptr(i), all.ptr(keep[i]), 32);
}
void line_lbd_detect::match_line_descrip(const cv::Mat &descrips_query, const cv::Mat &descrips_train, std::vector<cv::DMatch> &good_matches, float matching_dist_thres) {
    good_matches.clear();
    const int nq = descrips_query.rows, nt = descrips_train.rows;
    if (nq == 0 || nt == 0) return;
    std::vector<uint8_t> q((size_t)nq * 32), t((size_t)nt * 32);
    for (int i = 0; i < nq; i++) std::memcpy(q.data() + (size_t)i * 32, descrips_query.ptr(i), 32);
    for (int i = 0; i < nt; i++) std::memcpy(t.data() + (size_t)i * 32, descrips_train.ptr(i), 32);
    std::vector<int> qi(nq), ti(nq), di(nq);
    int n = 0;
    cs_ctx *ctx = shared_ctx();
    if (cs_lbd_match(ctx, q.data(), nq, t.data(), nt, matching_dist_thres, qi.data(), ti.data(), di.data(), &n) != CS_OK)
        throw std::runtime_error(std::string("line_lbd_detect (HIP): ") + cs_last_error(ctx));
    for (int i = 0; i < n; i++) { cv::DMatch m; m.queryIdx = qi[i]; m.trainIdx = ti[i]; m.imgIdx = 0; m.distance = (float)di[i]; good_matches.push_back(m); }
}

extern "C" void line_lbd_detect_hip_release(const void *detector) { // optional: free the device side of one line_lbd_detect (no call on it may be in flight)
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dev.find(static_cast<const line_lbd_detect *>(detector));
    if (it == g_dev.end()) return;
    if (it->second.lsd) cs_lsd_destroy(it->second.ctx, it->second.lsd);
    if (it->second.ctx) cs_destroy(it->second.ctx);
    delete it->second.use;
    g_dev.erase(it);
}
