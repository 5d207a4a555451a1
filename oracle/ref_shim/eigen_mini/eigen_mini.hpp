// eigen_mini.hpp -- TEST INFRASTRUCTURE: the part of Eigen 3's fixed-size interface that the reference's vendored g2o types/se3quat.h, se3_ops.h and the
// pose functions of orb_object_slam/{include/g2o_Object.h, src/g2o_Object.cpp} use, so that those files compile FROM /root/reference where they lie
// (oracle/Makefile.ref) although Eigen is absent here.  Everything is evaluated eagerly, coefficient by coefficient, in the order Eigen's lazy
// expressions evaluate a coefficient (sums over k ascending, chained sums left to right); Quaternion follows Eigen's generic (non-SIMD) code.
#pragma once
#include <cassert>
#include <cmath>
#include <iostream>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

#include <vector>

namespace Eigen {
constexpr int Dynamic = -1;
template <typename T> class DynMat;
template <typename Derived> struct MatrixBase {
    const Derived &derived() const { return *static_cast<const Derived *>(this); }
    Derived &derived() { return *static_cast<Derived *>(this); }
    int size() const { return derived().rows() * derived().cols(); }
    double operator[](int i) const { return derived().coeff(i); }
    double operator()(int i) const { return derived().coeff(i); }
};
template <typename T, int R, int C> class Matrix;
template <typename T, int R, int C> struct CommaInit {
    Matrix<T, R, C> &m; int k;
    CommaInit &operator,(T v) { m.d[(k / C) * C + k % C] = v; k++; return *this; } // row by row
};
template <typename T, int R, int C> class Matrix : public MatrixBase<Matrix<T, R, C>> {
  public:
    T d[R * C]; // row-major
    Matrix() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    Matrix(T x, T y) { static_assert(R * C == 2, ""); d[0] = x; d[1] = y; }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, ""); d[0] = x; d[1] = y; d[2] = z; }
    Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, ""); d[0] = x; d[1] = y; d[2] = z; d[3] = w; }
    Matrix(const DynMat<T> &m); // sizes must agree
    int rows() const { return R; }
    int cols() const { return C; }
    T coeff(int i) const { return d[i]; }
    T &operator()(int i, int j) { return d[i * C + j]; }
    T operator()(int i, int j) const { return d[i * C + j]; }
    T &operator()(int i) { static_assert(R == 1 || C == 1, ""); return d[i]; }
    T operator()(int i) const { return d[i]; }
    T &operator[](int i) { return d[i]; }
    T operator[](int i) const { return d[i]; }
    T *data() { return d; }
    const T *data() const { return d; }
    void setZero() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    void fill(T v) { for (int i = 0; i < R * C; i++) d[i] = v; }
    void setIdentity() { setZero(); for (int i = 0; i < (R < C ? R : C); i++) d[i * C + i] = T(1); }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int, int) { return Matrix(); }
    CommaInit<T, R, C> operator<<(T v) { d[0] = v; return CommaInit<T, R, C>{*this, 1}; }
    T squaredNorm() const { T s = d[0] * d[0]; for (int i = 1; i < R * C; i++) s += d[i] * d[i]; return s; }
    T norm() const { return std::sqrt(squaredNorm()); }
    Matrix operator+(const Matrix &o) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] + o.d[i]; return r; }
    Matrix operator-(const Matrix &o) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] - o.d[i]; return r; }
    Matrix operator-() const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = -d[i]; return r; }
    Matrix operator*(T s) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] * s; return r; }
    Matrix operator/(T s) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] / s; return r; }
    Matrix &operator+=(const Matrix &o) { for (int i = 0; i < R * C; i++) d[i] += o.d[i]; return *this; }
    Matrix &operator-=(const Matrix &o) { for (int i = 0; i < R * C; i++) d[i] -= o.d[i]; return *this; }
    Matrix &operator*=(T s) { for (int i = 0; i < R * C; i++) d[i] *= s; return *this; }
    Matrix &operator/=(T s) { for (int i = 0; i < R * C; i++) d[i] /= s; return *this; }
    template <int K> Matrix<T, R, K> operator*(const Matrix<T, C, K> &o) const {
        Matrix<T, R, K> r;
        for (int i = 0; i < R; i++) for (int j = 0; j < K; j++) { T s = d[i * C] * o.d[j]; for (int k = 1; k < C; k++) s += d[i * C + k] * o.d[k * K + j]; r.d[i * K + j] = s; }
        return r;
    }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.d[j * R + i] = d[i * C + j]; return r; }
    Matrix &noalias() { return *this; }
    T dot(const Matrix &o) const { T s = d[0] * o.d[0]; for (int i = 1; i < R * C; i++) s += d[i] * o.d[i]; return s; }
    struct ArrF { // .array(): coefficient-wise quotient, assignable back to a matrix
        Matrix m;
        ArrF operator/(const ArrF &o) const { ArrF r{m}; for (int i = 0; i < R * C; i++) r.m.d[i] = m.d[i] / o.m.d[i]; return r; }
    };
    ArrF array() const { return ArrF{*this}; }
    Matrix(const ArrF &a) { for (int i = 0; i < R * C; i++) d[i] = a.m.d[i]; }
    Matrix cwiseAbs() const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = std::fabs(d[i]); return r; }
    T minCoeff(int *idx) const { int b = 0; for (int i = 1; i < R * C; i++) if (d[i] < d[b]) b = i; *idx = b; return d[b]; }
    // segments of a vector: assignable views on a non-const object, values on a const one
    template <int N> struct Seg {
        T *p;
        Seg &operator=(const Matrix<T, N, 1> &v) { for (int i = 0; i < N; i++) p[i] = v.d[i]; return *this; }
        operator Matrix<T, N, 1>() const { Matrix<T, N, 1> r; for (int i = 0; i < N; i++) r.d[i] = p[i]; return r; }
        Matrix<T, N, 1> operator+(const Seg &o) const { return Matrix<T, N, 1>(*this) + Matrix<T, N, 1>(o); }
        Matrix<T, N, 1> operator-(const Seg &o) const { return Matrix<T, N, 1>(*this) - Matrix<T, N, 1>(o); }
    };
    template <int N> Seg<N> head() { return Seg<N>{d}; }
    template <int N> Seg<N> tail() { return Seg<N>{d + R * C - N}; }
    template <int N> Seg<N> segment(int i) { return Seg<N>{d + i}; }
    template <int N> Matrix<T, N, 1> head() const { Matrix<T, N, 1> r; for (int i = 0; i < N; i++) r.d[i] = d[i]; return r; }
    template <int N> Matrix<T, N, 1> tail() const { Matrix<T, N, 1> r; for (int i = 0; i < N; i++) r.d[i] = d[R * C - N + i]; return r; }
    template <int N> Matrix<T, N, 1> segment(int s) const { Matrix<T, N, 1> r; for (int i = 0; i < N; i++) r.d[i] = d[s + i]; return r; }
    struct ColView { // a column, or its first n rows
        Matrix &m; int j, n;
        ColView head(int k) { return ColView{m, j, k}; }
        template <int N> ColView &operator=(const Matrix<T, N, 1> &v) { assert(N == n); for (int i = 0; i < N; i++) m.d[i * C + j] = v.d[i]; return *this; }
        operator Matrix<T, R, 1>() const { Matrix<T, R, 1> r; for (int i = 0; i < R; i++) r.d[i] = m.d[i * C + j]; return r; }
    };
    ColView col(int j) { return ColView{*this, j, R}; }
    Matrix<T, R, 1> col(int j) const { Matrix<T, R, 1> r; for (int i = 0; i < R; i++) r.d[i] = d[i * C + j]; return r; }
    struct BlockView {
        Matrix &m; int i0, j0, r, c;
        template <int RR, int CC> BlockView &operator=(const Matrix<T, RR, CC> &v) { assert(RR == r && CC == c); for (int i = 0; i < RR; i++) for (int j = 0; j < CC; j++) m.d[(i0 + i) * C + j0 + j] = v.d[i * CC + j]; return *this; }
    };
    BlockView block(int i0, int j0, int r, int c) { return BlockView{*this, i0, j0, r, c}; }
    template <int RR, int CC> struct Corner { // topLeftCorner<RR, CC>(): readable, assignable
        Matrix &m;
        operator Matrix<T, RR, CC>() const { Matrix<T, RR, CC> r; for (int i = 0; i < RR; i++) for (int j = 0; j < CC; j++) r.d[i * CC + j] = m.d[i * C + j]; return r; }
        template <int K> Matrix<T, RR, K> operator*(const Matrix<T, CC, K> &o) const { return Matrix<T, RR, CC>(*this) * o; }
        Corner &operator=(const Matrix<T, RR, CC> &v) { for (int i = 0; i < RR; i++) for (int j = 0; j < CC; j++) m.d[i * C + j] = v.d[i * CC + j]; return *this; }
    };
    template <int RR, int CC> Corner<RR, CC> topLeftCorner() { return Corner<RR, CC>{*this}; }
    template <int N> struct Cols { // leftCols<N>() / rightCols<N>(): assignable
        Matrix &m; int j0;
        Cols &operator=(const Matrix<T, R, N> &v) { for (int i = 0; i < R; i++) for (int j = 0; j < N; j++) m.d[i * C + j0 + j] = v.d[i * N + j]; return *this; }
    };
    template <int N> Cols<N> leftCols() { return Cols<N>{*this, 0}; }
    template <int N> Cols<N> rightCols() { return Cols<N>{*this, C - N}; }
    Matrix<T, R, R> asDiagonal() const { static_assert(C == 1, ""); Matrix<T, R, R> r; for (int i = 0; i < R; i++) r.d[i * R + i] = d[i]; return r; }
    DynMat<T> operator*(const DynMat<T> &o) const; // fixed x dynamic
};
template <typename T, int R, int C> Matrix<T, R, C> operator*(T s, const Matrix<T, R, C> &m) { Matrix<T, R, C> r; for (int i = 0; i < R * C; i++) r.d[i] = s * m.d[i]; return r; }
template <typename T, int R, int C> std::ostream &operator<<(std::ostream &o, const Matrix<T, R, C> &m) { for (int i = 0; i < R; i++) { for (int j = 0; j < C; j++) o << m(i, j) << " "; o << "\n"; } return o; }

typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;


// ---- the few dynamic-size matrices the cuboid's projection functions use (3 x N corner lists): one implementation behind every Matrix<T, ., Dynamic>
template <typename T> class DynMat {
  public:
    int r = 0, c = 0;
    std::vector<T> d; // row-major
    DynMat() {}
    DynMat(int rr, int cc) : r(rr), c(cc), d((size_t)rr * cc, T(0)) {}
    void resize(int rr, int cc) { r = rr; c = cc; d.assign((size_t)rr * cc, T(0)); }
    int rows() const { return r; }
    int cols() const { return c; }
    T &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    T operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    T operator()(int i) const { return d[i]; }
    static DynMat Ones(int n) { DynMat m(1, n); for (auto &v : m.d) v = T(1); return m; }
    template <int R, int C> DynMat &operator=(const Matrix<T, R, C> &m) { r = R; c = C; d.assign(m.d, m.d + R * C); return *this; }
    template <int R, int C> DynMat operator*(const Matrix<T, R, C> &o) const { // coefficient-based product, k ascending
        assert(c == R);
        DynMat x(r, C);
        for (int i = 0; i < r; i++) for (int j = 0; j < C; j++) { T s = (*this)(i, 0) * o.d[j]; for (int k = 1; k < R; k++) s += (*this)(i, k) * o.d[k * C + j]; x(i, j) = s; }
        return x;
    }
    struct ColRef { DynMat &m; int j; template <int N> ColRef &operator=(const Matrix<T, N, 1> &v) { assert(N == m.r); for (int i = 0; i < N; i++) m(i, j) = v.d[i]; return *this; } };
    ColRef col(int j) { return ColRef{*this, j}; }
    struct Comma { // Eigen's comma initialiser: scalars and blocks, left to right, then the next rows
        DynMat &m; int row, col, bh;
        void put(const DynMat &b) { if (col == m.c) { row += bh; col = 0; } for (int i = 0; i < b.r; i++) for (int j = 0; j < b.c; j++) m(row + i, col + j) = b(i, j); col += b.c; bh = b.r; }
        Comma &operator,(const DynMat &b) { put(b); return *this; }
        Comma &operator,(T v) { DynMat b(1, 1); b(0, 0) = v; put(b); return *this; }
    };
    Comma operator<<(const DynMat &b) { Comma k{*this, 0, 0, 1}; k.put(b); return k; }
    Comma operator<<(T v) { Comma k{*this, 0, 0, 1}; DynMat b(1, 1); b(0, 0) = v; k.put(b); return k; }
    struct Arr { DynMat v; Arr operator/(const Arr &o) const { Arr r{v}; for (size_t i = 0; i < r.v.d.size(); i++) r.v.d[i] = v.d[i] / o.v.d[i]; return r; } };
    Arr array() const { return Arr{*this}; }
    struct RowView { DynMat &m; int i; RowView &operator=(const Arr &a) { for (int j = 0; j < m.c; j++) m(i, j) = a.v.d[j]; return *this; } };
    RowView row(int i) { return RowView{*this, i}; }
    DynMat row(int i) const { DynMat x(1, c); for (int j = 0; j < c; j++) x(0, j) = (*this)(i, j); return x; }
    DynMat bottomRows(int n) const { DynMat x(n, c); for (int i = 0; i < n; i++) for (int j = 0; j < c; j++) x(i, j) = (*this)(r - n + i, j); return x; }
    struct Rowwise {
        const DynMat &m;
        DynMat maxCoeff() const { DynMat x(m.r, 1); for (int i = 0; i < m.r; i++) { T b = m(i, 0); for (int j = 1; j < m.c; j++) if (m(i, j) > b) b = m(i, j); x(i, 0) = b; } return x; }
        DynMat minCoeff() const { DynMat x(m.r, 1); for (int i = 0; i < m.r; i++) { T b = m(i, 0); for (int j = 1; j < m.c; j++) if (m(i, j) < b) b = m(i, j); x(i, 0) = b; } return x; }
    };
    Rowwise rowwise() const { return Rowwise{*this}; }
};
template <typename T, int R, int C> Matrix<T, R, C>::Matrix(const DynMat<T> &m) { assert(m.r * m.c == R * C); for (int i = 0; i < R * C; i++) d[i] = m.d[i]; }
template <typename T, int R, int C> DynMat<T> Matrix<T, R, C>::operator*(const DynMat<T> &o) const { // coefficient-based product, k ascending
    assert(o.r == C);
    DynMat<T> x(R, o.c);
    for (int i = 0; i < R; i++) for (int j = 0; j < o.c; j++) { T s = d[i * C] * o(0, j); for (int k = 1; k < C; k++) s += d[i * C + k] * o(k, j); x(i, j) = s; }
    return x;
}
#define EIGEN_MINI_DYN(RR, CC)                                                                              \
    template <typename T> class Matrix<T, RR, CC> : public DynMat<T> {                                      \
      public:                                                                                               \
        Matrix() {}                                                                                         \
        Matrix(int r, int c) : DynMat<T>(r, c) {}                                                           \
        Matrix(const DynMat<T> &m) : DynMat<T>(m) {}                                                        \
        static DynMat<T> Ones(int n) { return DynMat<T>::Ones(n); }                                         \
    };
EIGEN_MINI_DYN(Dynamic, Dynamic)
EIGEN_MINI_DYN(3, Dynamic)
EIGEN_MINI_DYN(2, Dynamic)
EIGEN_MINI_DYN(1, Dynamic)
typedef Matrix<double, Dynamic, Dynamic> MatrixXd; typedef Matrix<double, 3, Dynamic> Matrix3Xd; typedef Matrix<double, 2, Dynamic> Matrix2Xd;

template <typename M> class Map;
template <typename T, int R, int C> class Map<const Matrix<T, R, C>> : public Matrix<T, R, C> {
  public:
    explicit Map(const T *p) { for (int i = 0; i < R * C; i++) this->d[i] = p[i]; } // (vectors only: storage order does not matter)
};

template <typename T> class Quaternion {
    Matrix<T, 4, 1> c; // x y z w, like Eigen's coeffs()
  public:
    Quaternion() {}
    Quaternion(T w, T x, T y, T z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    explicit Quaternion(const Matrix<T, 3, 3> &mat) { // Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3>
        T t = mat(0, 0) + mat(1, 1) + mat(2, 2);
        if (t > T(0)) {
            t = std::sqrt(t + T(1.0)); w() = T(0.5) * t; t = T(0.5) / t;
            x() = (mat(2, 1) - mat(1, 2)) * t; y() = (mat(0, 2) - mat(2, 0)) * t; z() = (mat(1, 0) - mat(0, 1)) * t;
        } else {
            int i = 0;
            if (mat(1, 1) > mat(0, 0)) i = 1;
            if (mat(2, 2) > mat(i, i)) i = 2;
            int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + T(1.0));
            c[i] = T(0.5) * t; t = T(0.5) / t;
            w() = (mat(k, j) - mat(j, k)) * t; c[j] = (mat(j, i) + mat(i, j)) * t; c[k] = (mat(k, i) + mat(i, k)) * t;
        }
    }
    T &x() { return c[0]; } T &y() { return c[1]; } T &z() { return c[2]; } T &w() { return c[3]; }
    T x() const { return c[0]; } T y() const { return c[1]; } T z() const { return c[2]; } T w() const { return c[3]; }
    Matrix<T, 4, 1> &coeffs() { return c; }
    const Matrix<T, 4, 1> &coeffs() const { return c; }
    void setIdentity() { c[0] = c[1] = c[2] = 0; c[3] = 1; }
    T squaredNorm() const { return c.squaredNorm(); }
    T norm() const { return c.norm(); }
    void normalize() { c /= norm(); }
    Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
    Quaternion operator*(const Quaternion &b) const { // quat_product (generic)
        const Quaternion &a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(), a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(), a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaternion &operator*=(const Quaternion &b) { *this = *this * b; return *this; }
    Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1> &v) const { // _transformVector
        Matrix<T, 3, 1> uv(y() * v[2] - z() * v[1], z() * v[0] - x() * v[2], x() * v[1] - y() * v[0]);
        uv += uv;
        return Matrix<T, 3, 1>(v[0] + w() * uv[0] + (y() * uv[2] - z() * uv[1]), v[1] + w() * uv[1] + (z() * uv[0] - x() * uv[2]), v[2] + w() * uv[2] + (x() * uv[1] - y() * uv[0]));
    }
    Matrix<T, 3, 3> toRotationMatrix() const {
        Matrix<T, 3, 3> res;
        const T tx = T(2) * x(), ty = T(2) * y(), tz = T(2) * z(), twx = tx * w(), twy = ty * w(), twz = tz * w(), txx = tx * x(), txy = ty * x(), txz = tz * x(), tyy = ty * y(), tyz = tz * y(),
                tzz = tz * z();
        res(0, 0) = T(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = T(1) - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = T(1) - (txx + tyy);
        return res;
    }
};
typedef Quaternion<double> Quaterniond;
struct Isometry3d { // only named by SE3Quat's conversion operator, which nothing here calls
    Vector3d t;
    Isometry3d() {}
    explicit Isometry3d(const Quaterniond &) {}
    Vector3d &translation() { return t; }
};
} // namespace Eigen
