// cvshim.hpp -- TEST INFRASTRUCTURE.  A minimal stand-in for the OpenCV 3.x API surface that the reference's hot-path translation units
// (orb_object_slam/src/ORBextractor.cc, line_lbd/libs/lsd.cpp, line_lbd/libs/LSDDetector.cpp, detect_3d_cuboid/src/*.cpp) use, so that
// those files can be compiled where they lie under /root/reference into oracle/_ref/libref.so (oracle/Makefile.ref) and run against the
// restatement in oracle/*.cpp.  Containers and geometry types are written from the documented OpenCV interface; the image-processing
// primitives (resize, GaussianBlur, FAST, Canny, distanceTransform, cvtColor) forward to the oracle's own restatements (cv_prims.h,
// oracle.h): what the _ref build pins is the REFERENCE'S logic around those calls, not OpenCV's arithmetic.
#pragma once
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_PI 3.1415926535897932384626433832795
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16SC1 CV_MAKETYPE(CV_16S, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error(std::string("CV_Assert failed: ") + #expr); } while (0)
#define CV_Error(code, msg) throw std::runtime_error(msg)
#define CV_DbgAssert(expr)
#define CV_BGR2GRAY 6
#define CV_DIST_L2 2

namespace cv {
typedef std::string String;
using std::min; using std::max; using std::abs; using std::swap; using std::sqrt; using std::exp; using std::pow; using std::log; // core/base.hpp

inline int cvRound(double v) { return (int)std::lrint(v); } // SSE2 cvtsd2si: round half to even
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
float fastAtan2(float y, float x); // cvshim.cpp -> orc_fast_atan2

template <typename T> inline T saturate_cast(double v) { return (T)v; }
template <> inline int saturate_cast<int>(double v) { return cvRound(v); }
template <> inline uchar saturate_cast<uchar>(double v) { int i = cvRound(v); return (uchar)(i < 0 ? 0 : (i > 255 ? 255 : i)); }

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    template <typename U> Point_(const Point_<U> &p) : x((T)p.x), y((T)p.y) {}
    Point_ &operator*=(double s) { x = (T)(x * s); y = (T)(y * s); return *this; }
    Point_ &operator+=(const Point_ &o) { x += o.x; y += o.y; return *this; }
    bool operator==(const Point_ &o) const { return x == o.x && y == o.y; }
};
template <> template <> inline Point_<int>::Point_(const Point_<float> &p) : x(cvRound(p.x)), y(cvRound(p.y)) {} // saturate_cast<int>
template <> template <> inline Point_<int>::Point_(const Point_<double> &p) : x(cvRound(p.x)), y(cvRound(p.y)) {}
template <typename T> inline Point_<T> operator+(const Point_<T> &a, const Point_<T> &b) { return Point_<T>(a.x + b.x, a.y + b.y); }
template <typename T> inline Point_<T> operator-(const Point_<T> &a, const Point_<T> &b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <typename T> inline Point_<T> operator*(const Point_<T> &a, double s) { return Point_<T>((T)(a.x * s), (T)(a.y * s)); }
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

template <typename T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    bool operator==(const Size_ &o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_ &o) const { return !(*this == o); }
    T area() const { return width * height; }
};
typedef Size_<int> Size;
template <typename T> struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T _x, T _y, T w, T h) : x(_x), y(_y), width(w), height(h) {}
};
typedef Rect_<int> Rect;
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; } double operator[](int i) const { return val[i]; } static Scalar all(double v) { return Scalar(v, v, v, v); } };

template <typename T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; i++) val[i] = T(); }
    Vec(T a, T b) { static_assert(N >= 2, ""); val[0] = a; val[1] = b; for (int i = 2; i < N; i++) val[i] = T(); }
    Vec(T a, T b, T c) { static_assert(N >= 3, ""); val[0] = a; val[1] = b; val[2] = c; for (int i = 3; i < N; i++) val[i] = T(); }
    Vec(T a, T b, T c, T d) { static_assert(N >= 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; for (int i = 4; i < N; i++) val[i] = T(); }
    T &operator[](int i) { return val[i]; }
    const T &operator[](int i) const { return val[i]; }
};
typedef Vec<float, 4> Vec4f;
typedef Vec<int, 4> Vec4i;
typedef Vec<uchar, 3> Vec3b;

template <typename T> struct DataType { enum { type = -1 }; };
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };
template <> struct DataType<short> { enum { type = CV_16SC1 }; };
template <> struct DataType<int> { enum { type = CV_32SC1 }; };
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<double> { enum { type = CV_64FC1 }; };
template <> struct DataType<Vec4f> { enum { type = CV_32FC4 }; };
template <> struct DataType<Vec4i> { enum { type = CV_MAKETYPE(CV_32S, 4) }; };

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { COLOR_BGR2GRAY = 6 };
enum { DIST_L2 = 2 };

// Mat::zeros returns a matrix EXPRESSION in OpenCV: assigned to an allocated Mat of the same size and type it fills that Mat's own buffer
// (MatOp_Initializer::assign -> create() is a no-op), which ORBextractor.cc:1030 relies on -- `descriptors = Mat::zeros(...)` writes
// through a rowRange view of the output matrix.
struct MatZeros { int rows, cols, type; };

class Mat {
public:
    int flags_type = 0, rows = 0, cols = 0;
    size_t step = 0; // bytes per row
    uchar *data = nullptr;
    std::shared_ptr<std::vector<uchar>> buf;
    uchar *datastart = nullptr; int whole_rows = 0, whole_cols = 0; // the matrix a view was cut from (cv::Mat::locateROI): Canny on a view reads around it

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size sz, int type) { create(sz.height, sz.width, type); }
    Mat(int r, int c, int type, const Scalar &s) { create(r, c, type); setTo(s); }
    Mat(Size sz, int type, const Scalar &s) { create(sz.height, sz.width, type); setTo(s); }
    Mat(int r, int c, int type, void *d, size_t st = 0) : flags_type(type), rows(r), cols(c), data((uchar *)d), datastart((uchar *)d), whole_rows(r), whole_cols(c) { step = st ? st : (size_t)c * elemSize(); }
    Mat(Size sz, int type, void *d, size_t st = 0) : Mat(sz.height, sz.width, type, d, st) {}
    template <typename T> explicit Mat(const std::vector<T> &v) : flags_type(DataType<T>::type), rows((int)v.size()), cols(1), data((uchar *)v.data()) { step = sizeof(T); }
    Mat(const Mat &m, const Rect &r) : flags_type(m.flags_type), rows(r.height), cols(r.width), step(m.step), data(m.data + (size_t)r.y * m.step + (size_t)r.x * m.elemSize()), buf(m.buf), datastart(m.datastart), whole_rows(m.whole_rows), whole_cols(m.whole_cols) {}

    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && flags_type == type) return;
        flags_type = type; rows = r; cols = c; step = (size_t)c * elemSize();
        buf = std::make_shared<std::vector<uchar>>((size_t)r * step + 64);
        data = buf->data();
        datastart = data; whole_rows = r; whole_cols = c;
    }
    void create(Size sz, int type) { create(sz.height, sz.width, type); }
    void release() { rows = cols = 0; data = nullptr; buf.reset(); step = 0; }
    int type() const { return flags_type; }
    int depth() const { return CV_MAT_DEPTH(flags_type); }
    int channels() const { return CV_MAT_CN(flags_type); }
    size_t elemSize1() const { static const int sz[] = {1, 1, 2, 2, 4, 4, 8}; return sz[depth()]; }
    size_t elemSize() const { return elemSize1() * channels(); }
    size_t step1() const { return step / elemSize1(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize() || rows <= 1; }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    uchar *ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T &at(int r, int c) { return ((T *)(data + (size_t)r * step))[c]; }
    template <typename T> const T &at(int r, int c) const { return ((const T *)(data + (size_t)r * step))[c]; }
    template <typename T> T &at(Point p) { return at<T>(p.y, p.x); }
    template <typename T> const T &at(Point p) const { return at<T>(p.y, p.x); }
    Mat rowRange(int a, int b) const { return Mat(*this, Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return Mat(*this, Rect(a, 0, b - a, rows)); }
    Mat rowRange(const Range &r) const { return rowRange(r.start, r.end); }
    Mat colRange(const Range &r) const { return colRange(r.start, r.end); }
    Mat row(int r) const { return rowRange(r, r + 1); }
    Mat col(int c) const { return colRange(c, c + 1); }
    int checkVector(int elemChannels, int = -1, bool = true) const { return (cols == 1 && channels() == elemChannels) ? rows : ((channels() == 1 && cols == elemChannels) ? rows : -1); }
    double dot(const Mat &m) const { double s = 0; for (int i = 0; i < rows * cols; i++) s += (double)at<float>(i) * (double)m.at<float>(i); return s; } // CV_32F vectors: cv::Mat::dot accumulates in double
    template <typename T> T &at(int i) { return rows == 1 ? ((T *)data)[i] : *(T *)(data + (size_t)i * step); }
    template <typename T> const T &at(int i) const { return rows == 1 ? ((const T *)data)[i] : *(const T *)(data + (size_t)i * step); }
    Mat operator()(const Rect &r) const { return Mat(*this, r); }
    Mat clone() const { Mat m; copyTo(m); return m; }
    Mat t() const { Mat m(cols, rows, flags_type); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) std::memcpy(m.data + (size_t)c * m.step + (size_t)r * elemSize(), data + (size_t)r * step + (size_t)c * elemSize(), elemSize()); return m; } // (a copy: the few callers only read it)
    void copyTo(Mat &m) const {
        m.create(rows, cols, flags_type);
        for (int r = 0; r < rows; r++) std::memmove(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
    }
    void copyTo(const class _OutputArray &o) const;
    void convertTo(Mat &m, int rtype) const;
    Mat &setTo(const Scalar &s);
    static MatZeros zeros(int r, int c, int type) { return MatZeros{r, c, type}; }
    static MatZeros zeros(Size sz, int type) { return MatZeros{sz.height, sz.width, type}; }
    Mat(const MatZeros &z) { *this = z; }
    Mat &operator=(const MatZeros &z) { create(z.rows, z.cols, z.type); for (int r = 0; r < rows; r++) std::memset(data + (size_t)r * step, 0, (size_t)cols * elemSize()); return *this; }
    Mat &operator=(const Scalar &s) { return setTo(s); }
};

template <typename T> class Mat_ : public Mat {
public:
    Mat_() { flags_type = DataType<T>::type; }
    Mat_(int r, int c) : Mat(r, c, DataType<T>::type) {}
    explicit Mat_(Size sz) : Mat(sz.height, sz.width, DataType<T>::type) {}
    static Mat_ zeros(Size sz) { return Mat_(Mat(Mat::zeros(sz, DataType<T>::type))); }
    static Mat_ zeros(int r, int c) { return Mat_(Mat(Mat::zeros(r, c, DataType<T>::type))); }
    Mat_(const Mat &m) { assign(m); }
    Mat_ &operator=(const Mat &m) { assign(m); return *this; }
    T &operator()(int r, int c) { return this->template at<T>(r, c); }
    const T &operator()(int r, int c) const { return this->template at<T>(r, c); }
    T &operator()(Point p) { return this->template at<T>(p.y, p.x); }
    T *operator[](int r) { return this->template ptr<T>(r); }
    const T *operator[](int r) const { return this->template ptr<T>(r); }
private:
    void assign(const Mat &m) { // shares the data when the type matches, converts otherwise (cv::Mat_ semantics)
        if (m.empty()) { Mat::operator=(Mat()); flags_type = DataType<T>::type; return; }
        if (m.type() == DataType<T>::type) Mat::operator=(m);
        else { Mat t; m.convertTo(t, DataType<T>::type); Mat::operator=(t); }
    }
};

class _InputArray {
public:
    const Mat *m = nullptr;
    Mat tmp; // header over a std::vector
    _InputArray() {}
    _InputArray(const Mat &mm) : m(&mm) {}
    template <typename T> _InputArray(const Mat_<T> &mm) : m(&mm) {}
    template <typename T> _InputArray(const std::vector<T> &v) : tmp(v) { m = &tmp; }
    Mat getMat() const { return m ? *m : Mat(); }
    bool empty() const { return !m || m->empty(); }
    Size size() const { return m ? m->size() : Size(); }
    int type() const { return m ? m->type() : 0; }
};
class _OutputArray {
public:
    Mat *m = nullptr;
    std::vector<Vec4f> *v4f = nullptr; std::vector<Vec4i> *v4i = nullptr; std::vector<double> *vd = nullptr; std::vector<float> *vf = nullptr;
    _OutputArray() {}
    _OutputArray(Mat &mm) : m(&mm) {}
    template <typename T> _OutputArray(Mat_<T> &mm) : m(&mm) {}
    _OutputArray(std::vector<Vec4f> &v) : v4f(&v) {}
    _OutputArray(std::vector<Vec4i> &v) : v4i(&v) {}
    _OutputArray(std::vector<double> &v) : vd(&v) {}
    _OutputArray(std::vector<float> &v) : vf(&v) {}
    bool needed() const { return m || v4f || v4i || vd || vf; }
    void create(int r, int c, int type) const { if (m) m->create(r, c, type); }
    void create(Size s, int type) const { create(s.height, s.width, type); }
    void release() const { if (m) m->release(); }
    Mat getMat() const { return m ? *m : Mat(); }
    Mat &getMatRef() const { return *m; }
    bool empty() const { return !m || m->empty(); }
    int channels() const { return m ? m->channels() : 0; }
    Size size() const { return m ? m->size() : Size(); }
    operator _InputArray() const { return m ? _InputArray(*m) : _InputArray(); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
typedef const _OutputArray &InputOutputArray;
inline const _OutputArray &noArray() { static _OutputArray none; return none; }

struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
struct KeyPointsFilter { static void retainBest(std::vector<KeyPoint> &kps, int n); };
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; };

// cv::LineIterator: only .count is read (LSDDetector.cpp:241-242).  8-connected Bresenham between end points that lie inside the image
// (checkLineExtremes clamps them): max(|dx|, |dy|) + 1 pixels; Point2f -> Point rounds (saturate_cast<int> = cvRound)
struct LineIterator {
    int count;
    LineIterator(const Mat &img, Point pt1, Point pt2, int connectivity = 8, bool = false) {
        CV_Assert(connectivity == 8 && pt1.x >= 0 && pt1.y >= 0 && pt2.x >= 0 && pt2.y >= 0 && pt1.x < img.cols && pt2.x < img.cols && pt1.y < img.rows && pt2.y < img.rows);
        count = std::max(std::abs(pt2.x - pt1.x), std::abs(pt2.y - pt1.y)) + 1;
    }
};

template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> inline Ptr<T> makePtr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
class FileNode {}; class FileStorage {};
class Algorithm { public: virtual ~Algorithm() {} virtual void read(const FileNode &) {} virtual void write(FileStorage &) const {} };

// image-processing primitives: cvshim.cpp forwards them to the oracle's restatements
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType, const Scalar &value = Scalar());
void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
void Canny(InputArray image, OutputArray edges, double threshold1, double threshold2, int apertureSize = 3, bool L2gradient = false);
void distanceTransform(InputArray src, OutputArray dst, int distanceType, int maskSize, int dstType = CV_32F);
void pyrDown(InputArray src, OutputArray dst, const Size &dstsize = Size(), int borderType = BORDER_DEFAULT);
void Sobel(InputArray src, OutputArray dst, int ddepth, int dx, int dy, int ksize = 3, double scale = 1, double delta = 0, int borderType = BORDER_DEFAULT);
void merge(const std::vector<Mat> &mv, OutputArray dst);      // drawing helpers of lsd.cpp: never called, declared so the file compiles
void bitwise_xor(InputArray a, InputArray b, OutputArray dst);
int countNonZero(InputArray a);
inline void imshow(const String &, InputArray) {}
inline int waitKey(int = 0) { return -1; }
void line(InputOutputArray img, Point pt1, Point pt2, const Scalar &color, int thickness = 1, int lineType = 8, int shift = 0);
inline Mat operator-(const Scalar &s, const Mat &m) { // only `255 - edges` on CV_8U (box_proposal_detail.cpp)
    Mat r(m.rows, m.cols, m.type());
    for (int y = 0; y < m.rows; y++) for (int x = 0; x < m.cols; x++) r.at<uchar>(y, x) = saturate_cast<uchar>(s.val[0] - m.at<uchar>(y, x));
    return r;
}
} // namespace cv
