// eigshim.hpp -- TEST INFRASTRUCTURE.  The sliver of Eigen that the reference functions extracted by oracle/ref_shim/extract_ref.py use
// (merge_break_lines, box_edge_sum_dists, box_edge_alignment_angle_error, fuse_normalize_scores_v2 and their helpers): dynamic double / int
// matrices with block views, evaluated eagerly.  Eigen evaluates the expressions of those functions coefficient by coefficient in the
// written order (`a * v1 + b * v2` is (a*v1[i]) + (b*v2[i])), so eager temporaries produce the same doubles.  Eigen itself is absent here.
#pragma once
#include <cassert>
#include <cmath>
#include <vector>

namespace Eigen {
enum NoChange_t { NoChange };
template <typename T> class Mat;
template <typename T> class Block { // a rectangular view of a Mat
public:
    Mat<T> *m; int r0, c0, nr, nc;
    Block(Mat<T> *mm, int r, int c, int h, int w) : m(mm), r0(r), c0(c), nr(h), nc(w) {}
    int rows() const { return nr; } int cols() const { return nc; } int size() const { return nr * nc; }
    T &operator()(int i, int j) { return (*m)(r0 + i, c0 + j); }
    const T &operator()(int i, int j) const { return (*m)(r0 + i, c0 + j); }
    T &operator()(int i) { return nc == 1 ? (*this)(i, 0) : (*this)(0, i); }
    const T &operator()(int i) const { return nc == 1 ? (*this)(i, 0) : (*this)(0, i); }
    Block head(int n) const { return nc == 1 ? Block(m, r0, c0, n, 1) : Block(m, r0, c0, 1, n); }
    Block tail(int n) const { return nc == 1 ? Block(m, r0 + nr - n, c0, n, 1) : Block(m, r0, c0 + nc - n, 1, n); }
    Block &operator=(const Mat<T> &o);
    Block &operator=(const Block &o) { Mat<T> t(o); return *this = t; }
    double norm() const { double s = 0; for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) s += (double)(*this)(i, j) * (*this)(i, j); return std::sqrt(s); }
};
template <typename T> class Mat {
public:
    int nr = 0, nc = 0;
    std::vector<T> d; // row-major (the storage order is invisible to the extracted functions)
    Mat() {}
    explicit Mat(int n) : nr(n), nc(1), d(n) {}
    Mat(int r, int c) : nr(r), nc(c), d((size_t)r * c) {}
    Mat(const Block<T> &b) : nr(b.nr), nc(b.nc), d((size_t)b.nr * b.nc) { for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) (*this)(i, j) = b(i, j); }
    int rows() const { return nr; } int cols() const { return nc; } int size() const { return nr * nc; }
    T &operator()(int i, int j) { return d[(size_t)i * nc + j]; }
    const T &operator()(int i, int j) const { return d[(size_t)i * nc + j]; }
    T &operator()(int i) { return d[i]; } // vectors
    const T &operator()(int i) const { return d[i]; }
    T &operator[](int i) { return d[i]; }
    const T &operator[](int i) const { return d[i]; }
    void resize(int r, int c) { nr = r; nc = c; d.assign((size_t)r * c, T()); }
    void resize(int n) { resize(n, 1); }
    void conservativeResize(int r, NoChange_t) { d.resize((size_t)r * nc); nr = r; }
    Mat &operator=(const Block<T> &b) { Mat t(b); nr = t.nr; nc = t.nc; d = t.d; return *this; }
    Block<T> block(int r, int c, int h, int w) const { return Block<T>(const_cast<Mat *>(this), r, c, h, w); }
    Block<T> row(int i) const { return block(i, 0, 1, nc); }
    Block<T> col(int j) const { return block(0, j, nr, 1); }
    Block<T> head(int n) const { return nc == 1 ? block(0, 0, n, 1) : block(0, 0, 1, n); }
    Block<T> tail(int n) const { return nc == 1 ? block(nr - n, 0, n, 1) : block(0, nc - n, 1, n); }
    Block<T> topRows(int n) const { return block(0, 0, n, nc); }
    Block<T> topLeftCorner(int h, int w) const { return block(0, 0, h, w); }
    Block<T> topRightCorner(int h, int w) const { return block(0, nc - w, h, w); }
    double norm() const { double s = 0; for (const T &v : d) s += (double)v * v; return std::sqrt(s); }
    struct Rowwise { const Mat *m; Mat norm() const { Mat r(m->nr, 1); for (int i = 0; i < m->nr; i++) { double s = 0; for (int j = 0; j < m->nc; j++) s += (*m)(i, j) * (*m)(i, j); r(i) = std::sqrt(s); } return r; } };
    Rowwise rowwise() const { return Rowwise{this}; }
    const Mat &array() const { return *this; } // the array view only changes which operators apply; the ones below are coefficient-wise anyway
    T minCoeff() const { T v = d[0]; for (const T &x : d) if (x < v) v = x; return v; }
    T maxCoeff() const { T v = d[0]; for (const T &x : d) if (v < x) v = x; return v; }
};
template <typename T> Block<T> &Block<T>::operator=(const Mat<T> &o) {
    assert(o.size() == size());
    for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) (*this)(i, j) = o.d[(size_t)i * nc + j]; // a 2-vector assigned to a 1x2 block: by linear index
    return *this;
}
template <typename T> Mat<T> operator-(const Block<T> &a, const Block<T> &b) { Mat<T> r(a.nr, a.nc); for (int i = 0; i < a.nr; i++) for (int j = 0; j < a.nc; j++) r(i, j) = a(i, j) - b(i, j); return r; }
template <typename T> Mat<T> operator+(const Mat<T> &a, const Mat<T> &b) { Mat<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <typename T> Mat<T> operator-(const Mat<T> &a, const Mat<T> &b) { Mat<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <typename T> Mat<T> operator-(const Mat<T> &a, double s) { Mat<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] - s; return r; }
template <typename T> Mat<T> operator*(double s, const Mat<T> &a) { Mat<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = s * a.d[i]; return r; }
template <typename T> Mat<T> operator*(double s, const Block<T> &a) { return s * Mat<T>(a); }
template <typename T> Mat<T> operator/(const Mat<T> &a, double s) { Mat<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] / s; return r; }
typedef Mat<double> MatrixXd;
typedef Mat<double> VectorXd;
typedef Mat<int> MatrixXi;
struct Vector2d : Mat<double> { // fixed size 2
    Vector2d() : Mat<double>(2) {}
    Vector2d(const Mat<double> &m) : Mat<double>(m) { assert(m.size() == 2); nr = 2; nc = 1; }
    Vector2d(const Block<double> &b) : Mat<double>(b) { assert(b.size() == 2); nr = 2; nc = 1; }
    Vector2d &operator=(const Block<double> &b) { Mat<double> t(b); d = t.d; nr = 2; nc = 1; return *this; }
    Vector2d &operator=(const Mat<double> &m) { d = m.d; nr = 2; nc = 1; return *this; }
};
} // namespace Eigen
