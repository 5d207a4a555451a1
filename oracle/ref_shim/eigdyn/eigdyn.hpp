// eigdyn.hpp -- TEST INFRASTRUCTURE.  A second sliver of Eigen (Eigen itself is absent here), for the geometry functions of detect_3d_cuboid that
// oracle/ref_shim/extract_ref.py cuts out of object_3d_util.cpp / matrix_utils.cpp (ref_geom_api.cpp): fixed and dynamic double / int matrices with
// views, comma initialisers, coefficient-wise arrays and small products, all on one dynamically sized class and evaluated EAGERLY.  Eigen evaluates
// these expressions coefficient by coefficient in the written order and its small products as sum_k a(i,k) * b(k,j) with k ascending, so eager
// temporaries produce the same doubles (no FMA contraction: -ffp-contract=off).  Vectors assigned from a row expression turn into columns like
// Eigen's do.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <vector>

// (its own namespace, aliased to `Eigen` by the file that includes it: libref.so also holds eigen_mini's Eigen::Matrix, and two different inline
// definitions under one mangled name would be merged by the linker)
namespace EigenDyn {
enum NoChange_t { NoChange };
constexpr int Dynamic = -1;
template <typename T> class M;
template <typename T> class Arr;

template <typename T> class View { // a rectangular window of an M (rows r0.., columns c0..)
  public:
    M<T> *m; int r0, c0, nr, nc;
    View(M<T> *mm, int r, int c, int h, int w) : m(mm), r0(r), c0(c), nr(h), nc(w) {}
    int rows() const { return nr; } int cols() const { return nc; } int size() const { return nr * nc; }
    T &operator()(int i, int j) const { return (*m)(r0 + i, c0 + j); }
    T &operator()(int i) const { return nc == 1 ? (*this)(i, 0) : (*this)(0, i); }
    T &operator[](int i) const { return (*this)(i); }
    View head(int n) const { return nc == 1 ? View(m, r0, c0, n, 1) : View(m, r0, c0, 1, n); }
    View tail(int n) const { return nc == 1 ? View(m, r0 + nr - n, c0, n, 1) : View(m, r0, c0 + nc - n, 1, n); }
    template <int N> View head() const { return head(N); }
    template <int N> View tail() const { return tail(N); }
    View row(int i) const { return View(m, r0 + i, c0, 1, nc); }
    View col(int j) const { return View(m, r0, c0 + j, nr, 1); }
    template <int N> View segment(int i) const { return nc == 1 ? View(m, r0 + i, c0, N, 1) : View(m, r0, c0 + i, 1, N); }
    T maxCoeff() const { return eval().maxCoeff(); }
    T minCoeff() const { return eval().minCoeff(); }
    M<T> eval() const;
    const View &operator=(const M<T> &o) const; // by linear index when the shapes are transposes of each other (vectors)
    const View &operator=(const View &o) const { return *this = o.eval(); }
    const View &operator=(const Arr<T> &a) const;
    double norm() const { return eval().norm(); }
    T mean() const { return eval().mean(); }
    M<T> transpose() const { return eval().transpose(); }
    Arr<T> array() const;
};

template <typename T> class M {
  public:
    int nr = 0, nc = 0;
    bool vec = false; // a column vector type: a row assigned to it is transposed
    std::vector<T> d; // row-major
    M() {}
    M(int r, int c) : nr(r), nc(c), d((size_t)r * c, T()) {}
    explicit M(int n) : nr(n), nc(1), vec(true), d((size_t)n, T()) {}
    M(const View<T> &v) : nr(v.nr), nc(v.nc), d((size_t)v.nr * v.nc) { for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) (*this)(i, j) = v(i, j); }
    M(const Arr<T> &a);
    int rows() const { return nr; } int cols() const { return nc; } int size() const { return nr * nc; }
    T &operator()(int i, int j) { return d[(size_t)i * nc + j]; }
    const T &operator()(int i, int j) const { return d[(size_t)i * nc + j]; }
    T &operator()(int i) { return d[i]; }
    const T &operator()(int i) const { return d[i]; }
    T &operator[](int i) { return d[i]; }
    const T &operator[](int i) const { return d[i]; }
    operator T() const { assert(nr * nc == 1); return d[0]; } // a 1 x 1 product used as a scalar
    void resize(int r, int c) { nr = r; nc = c; d.assign((size_t)r * c, T()); }
    void resize(int n) { resize(n, 1); }
    void conservativeResize(int r, NoChange_t) { d.resize((size_t)r * nc, T()); nr = r; } // (row-major: the old rows stay where they are)
    M &assign(const M &o) { // Eigen's vector assignment: a row expression into a column vector type
        const bool v = vec;
        nr = o.nr; nc = o.nc; d = o.d; vec = v;
        if (vec && nr == 1) { nr = nc; nc = 1; }
        return *this;
    }
    View<T> block(int r, int c, int h, int w) const { return View<T>(const_cast<M *>(this), r, c, h, w); }
    View<T> row(int i) const { return block(i, 0, 1, nc); }
    View<T> col(int j) const { return block(0, j, nr, 1); }
    View<T> head(int n) const { return nc == 1 ? block(0, 0, n, 1) : block(0, 0, 1, n); }
    View<T> tail(int n) const { return nc == 1 ? block(nr - n, 0, n, 1) : block(0, nc - n, 1, n); }
    template <int N> View<T> head() const { return head(N); }
    template <int N> View<T> tail() const { return tail(N); }
    View<T> rightCols(int n) const { return block(0, nc - n, nr, n); }
    View<T> bottomRows(int n) const { return block(nr - n, 0, n, nc); }
    View<T> topRows(int n) const { return block(0, 0, n, nc); }
    template <int H, int W> View<T> topLeftCorner() const { return block(0, 0, H, W); }
    View<T> topLeftCorner(int h, int w) const { return block(0, 0, h, w); }
    View<T> topRightCorner(int h, int w) const { return block(0, nc - w, h, w); }
    M transpose() const { M r(nc, nr); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) r(j, i) = (*this)(i, j); return r; }
    double norm() const { double s = 0; for (const T &v : d) s += (double)v * v; return std::sqrt(s); } // (sum in storage order: a vector's natural order)
    T mean() const { T s = d[0]; for (size_t i = 1; i < d.size(); i++) s += d[i]; return s / (T)d.size(); }
    T maxCoeff() const { T v = d[0]; for (const T &x : d) if (v < x) v = x; return v; }
    T minCoeff() const { T v = d[0]; for (const T &x : d) if (x < v) v = x; return v; }
    template <int N> View<T> topRows() const { return block(0, 0, N, nc); }
    struct Rowwise { const M *m; M norm() const { M r(m->nr); for (int i = 0; i < m->nr; i++) { double q = 0; for (int j = 0; j < m->nc; j++) q += (*m)(i, j) * (*m)(i, j); r(i) = std::sqrt(q); } return r; } };
    Rowwise rowwise() const { return Rowwise{this}; }
    M inverse() const;
    T maxCoeff(int *idx) const { int b = 0; for (int i = 1; i < size(); i++) if (d[i] > d[b]) b = i; *idx = b; return d[b]; }  // first occurrence
    T minCoeff(int *idx) const { int b = 0; for (int i = 1; i < size(); i++) if (d[i] < d[b]) b = i; *idx = b; return d[b]; }
    M cross(const M &o) const { M r(3); r(0) = d[1] * o.d[2] - d[2] * o.d[1]; r(1) = d[2] * o.d[0] - d[0] * o.d[2]; r(2) = d[0] * o.d[1] - d[1] * o.d[0]; return r; }
    M asDiagonal() const { M r(size(), size()); for (int i = 0; i < size(); i++) r(i, i) = d[i]; return r; }
    template <int RR, int CC> M replicate() const { M r(nr * RR, nc * CC); for (int i = 0; i < r.nr; i++) for (int j = 0; j < r.nc; j++) r(i, j) = (*this)(i % nr, j % nc); return r; }
    template <typename U> M<U> cast() const { M<U> r(nr, nc); for (size_t i = 0; i < d.size(); i++) r.d[i] = (U)d[i]; return r; }
    Arr<T> array() const;
    static M Ones(int r, int c) { M m(r, c); for (T &v : m.d) v = T(1); return m; }
    static M Ones(int n) { M m(n); for (T &v : m.d) v = T(1); return m; }
    // comma initialiser: scalars and blocks, left to right, then the next rows
    struct Comma {
        M &m; int row, col, bh;
        void put(const M &b) { if (col == m.nc) { row += bh; col = 0; } for (int i = 0; i < b.nr; i++) for (int j = 0; j < b.nc; j++) m(row + i, col + j) = b(i, j); col += b.nc; bh = b.nr; }
        Comma &operator,(const M &b) { put(b); return *this; }
        Comma &operator,(const View<T> &b) { put(M(b)); return *this; }
        Comma &operator,(T v) { M b(1, 1); b.d[0] = v; put(b); return *this; }
    };
    Comma operator<<(const M &b) { Comma k{*this, 0, 0, 1}; k.put(b); return k; }
    Comma operator<<(T v) { Comma k{*this, 0, 0, 1}; M b(1, 1); b.d[0] = v; k.put(b); return k; }
};
template <typename T> M<T> View<T>::eval() const { return M<T>(*this); }
template <typename T> const View<T> &View<T>::operator=(const M<T> &o) const {
    assert(o.size() == size());
    for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) (*this)(i, j) = o.d[(size_t)i * nc + j];
    return *this;
}

// coefficient-wise view: only changes which operators apply
template <typename T> class Arr {
  public:
    M<T> v;
    explicit Arr(const M<T> &m) : v(m) {}
    Arr operator/(const Arr &o) const { Arr r(v); for (size_t i = 0; i < r.v.d.size(); i++) r.v.d[i] = v.d[i] / o.v.d[i]; return r; }
    Arr operator*(const Arr &o) const { Arr r(v); for (size_t i = 0; i < r.v.d.size(); i++) r.v.d[i] = v.d[i] * o.v.d[i]; return r; }
    Arr operator-(double s) const { Arr r(v); for (T &x : r.v.d) x = (T)(x - s); return r; }
    Arr operator/(double s) const { Arr r(v); for (T &x : r.v.d) x = (T)(x / s); return r; }
    struct Bools { std::vector<bool> b; bool any() const { for (bool x : b) if (x) return true; return false; } };
    Bools operator<(double s) const { Bools r; for (const T &x : v.d) r.b.push_back(x < s); return r; }
};
template <typename T> Arr<T> operator/(T s, const Arr<T> &a) { Arr<T> r(a.v); for (T &x : r.v.d) x = s / x; return r; }
template <typename T> M<T>::M(const Arr<T> &a) : nr(a.v.nr), nc(a.v.nc), d(a.v.d) {}
template <typename T> Arr<T> M<T>::array() const { return Arr<T>(*this); }
template <typename T> Arr<T> View<T>::array() const { return Arr<T>(eval()); }
template <typename T> const View<T> &View<T>::operator=(const Arr<T> &a) const { return *this = a.v; }
// `v.array() /= s` on a named vector
template <typename T> struct ArrRef {
    M<T> &m;
    ArrRef &operator/=(T s) { for (T &x : m.d) x = x / s; return *this; }
    ArrRef &operator-=(T s) { for (T &x : m.d) x = x - s; return *this; }
    Arr<T> operator-(double s) const { return Arr<T>(m) - s; }
    typename Arr<T>::Bools operator<(double s) const { return Arr<T>(m) < s; }
};

template <typename T> M<T> operator+(const M<T> &a, const M<T> &b) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <typename T> M<T> operator-(const M<T> &a, const M<T> &b) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <typename T> M<T> operator-(const M<T> &a) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = -a.d[i]; return r; }
template <typename T> M<T> operator-(const View<T> &a, const View<T> &b) { return a.eval() - b.eval(); }
template <typename T> M<T> operator+(const View<T> &a, const View<T> &b) { return a.eval() + b.eval(); }
template <typename T> M<T> operator*(double s, const M<T> &a) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = s * a.d[i]; return r; }
template <typename T> M<T> operator*(const M<T> &a, double s) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] * s; return r; }
template <typename T> M<T> operator/(const M<T> &a, double s) { M<T> r(a.nr, a.nc); for (size_t i = 0; i < r.d.size(); i++) r.d[i] = a.d[i] / s; return r; }
template <typename T> M<T> operator/(const View<T> &a, double s) { return a.eval() / s; }
template <typename T> M<T> operator*(const M<T> &a, const M<T> &b) { // sum over k ascending, like Eigen's coefficient-based product of small matrices
    assert(a.nc == b.nr);
    M<T> r(a.nr, b.nc);
    for (int i = 0; i < a.nr; i++) for (int j = 0; j < b.nc; j++) { T s = a(i, 0) * b(0, j); for (int k = 1; k < a.nc; k++) s += a(i, k) * b(k, j); r(i, j) = s; }
    return r;
}
template <typename T> M<T> operator*(const M<T> &a, const View<T> &b) { return a * b.eval(); }

// inverse(): Eigen's own algorithms are not here to compile; the 3 x 3 one is its cofactor formula (cofactors * (1 / det), det along the first column), the
// 4 x 4 one a Gauss-Jordan elimination with partial pivoting (it only feeds cam_pose.projectionMatrix, which no output of detect_cuboid reads) -- the
// same two restatements the oracle carries (cuboid_oracle.cpp): what this file pins is everything AROUND them.
template <typename T> M<T> M<T>::inverse() const {
    const M &a = *this;
    if (nr == 3 && nc == 3) {
        auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return a(i1, j1) * a(i2, j2) - a(i1, j2) * a(i2, j1); };
        const T c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
        const T det = (c00 * a(0, 0) + c10 * a(1, 0)) + c20 * a(2, 0);
        const T invdet = T(1) / det;
        M r(3, 3);
        r(0, 0) = c00 * invdet; r(0, 1) = c10 * invdet; r(0, 2) = c20 * invdet;
        r(1, 0) = cof(0, 1) * invdet; r(1, 1) = cof(1, 1) * invdet; r(1, 2) = cof(2, 1) * invdet;
        r(2, 0) = cof(0, 2) * invdet; r(2, 1) = cof(1, 2) * invdet; r(2, 2) = cof(2, 2) * invdet;
        return r;
    }
    assert(nr == 4 && nc == 4);
    T w[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { w[i][j] = a(i, j); w[i][4 + j] = (i == j) ? T(1) : T(0); }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c) for (int j = 0; j < 8; j++) { const T t = w[p][j]; w[p][j] = w[c][j]; w[c][j] = t; }
        const T d = T(1) / w[c][c];
        for (int j = 0; j < 8; j++) w[c][j] *= d;
        for (int r = 0; r < 4; r++) if (r != c) { const T f = w[r][c]; for (int j = 0; j < 8; j++) w[r][j] -= f * w[c][j]; }
    }
    M r(4, 4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r(i, j) = w[i][4 + j];
    return r;
}
// Quaternion from a rotation matrix: Eigen's quaternionbase_assign_impl<Other, 3, 3> as the oracle restates it
template <typename T> class Quaternion {
  public:
    T qw, qx, qy, qz;
    Quaternion(T w, T x, T y, T z) : qw(w), qx(x), qy(y), qz(z) {}
    Quaternion(const M<T> &m) {
        T t = m(0, 0) + m(1, 1) + m(2, 2);
        T q[3];
        if (t > 0) {
            t = std::sqrt(t + T(1));
            qw = T(0.5) * t;
            t = T(0.5) / t;
            qx = (m(2, 1) - m(1, 2)) * t; qy = (m(0, 2) - m(2, 0)) * t; qz = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + T(1));
            q[i] = T(0.5) * t;
            t = T(0.5) / t;
            qw = (m(k, j) - m(j, k)) * t;
            q[j] = (m(j, i) + m(i, j)) * t;
            q[k] = (m(k, i) + m(i, k)) * t;
            qx = q[0]; qy = q[1]; qz = q[2];
        }
    }
    T w() const { return qw; } T x() const { return qx; } T y() const { return qy; } T z() const { return qz; }
};
typedef Quaternion<double> Quaterniond;

// the named types: sized (or not) flavours of M
template <typename T, int R, int C> class Matrix : public M<T> {
  public:
    Matrix() : M<T>(R > 0 ? R : 0, C > 0 ? C : 0) { this->vec = (C == 1); }
    Matrix(int a, int b) : M<T>(R * C == 2 ? R : a, R * C == 2 ? C : b) { this->vec = (C == 1); if (R * C == 2) { this->d[0] = (T)a; this->d[1] = (T)b; } } // sizes -- or, for a fixed 2-vector, its coefficients
    explicit Matrix(int n) : M<T>(C == 1 ? n : (R == 1 ? 1 : n), C == 1 ? 1 : n) { this->vec = (C == 1); }
    Matrix(double x, double y) : M<T>(2, 1) { this->vec = true; this->d[0] = (T)x; this->d[1] = (T)y; }
    Matrix(double x, double y, double z) : M<T>(3, 1) { this->vec = true; this->d[0] = (T)x; this->d[1] = (T)y; this->d[2] = (T)z; }
    Matrix(double x, double y, double z, double w) : M<T>(4, 1) { this->vec = true; this->d[0] = (T)x; this->d[1] = (T)y; this->d[2] = (T)z; this->d[3] = (T)w; }
    Matrix(const M<T> &m) { this->vec = (C == 1); this->assign(m); }
    Matrix(const View<T> &v) { this->vec = (C == 1); this->assign(M<T>(v)); }
    Matrix(const Arr<T> &a) { this->vec = (C == 1); this->assign(a.v); }
    Matrix &operator=(const M<T> &m) { this->assign(m); return *this; }
    Matrix &operator=(const View<T> &v) { this->assign(M<T>(v)); return *this; }
    Matrix &operator=(const Arr<T> &a) { this->assign(a.v); return *this; }
    ArrRef<T> array() { return ArrRef<T>{*this}; }
    Arr<T> array() const { return Arr<T>(*this); }
    static M<T> Identity() { M<T> m(R, C); for (int i = 0; i < R && i < C; i++) m(i, i) = T(1); return m; }
    static M<T> Ones(int r, int c) { return M<T>::Ones(r, c); }
    static M<T> Ones(int n) { return C == 1 ? M<T>::Ones(n) : M<T>::Ones(1, n); }
};
typedef Matrix<double, Dynamic, Dynamic> MatrixXd; typedef Matrix<double, Dynamic, 1> VectorXd; typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d; typedef Matrix<double, 3, Dynamic> Matrix3Xd; typedef Matrix<double, 2, Dynamic> Matrix2Xd;
typedef Matrix<int, 2, Dynamic> Matrix2Xi;
} // namespace EigenDyn
