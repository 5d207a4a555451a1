// eigen_full.hpp -- TEST INFRASTRUCTURE (never linked into the product): a stand-in for the part of Eigen 3's interface that the reference's vendored g2o
// (orb_object_slam/Thirdparty/g2o: core/, types/, solvers/linear_solver_dense.h), its object types (orb_object_slam/{include/g2o_Object.h, src/g2o_Object.cpp})
// and the graph-building functions of Optimizer.cc / Tracking.cc use, so that those files compile FROM /root/reference where they lie (oracle/Makefile.ref)
// although Eigen itself is not in this image.  One eager matrix class for fixed and dynamic sizes (column-major, like Eigen's default, because g2o maps raw
// Hessian memory), views for blocks / maps, and the few decompositions g2o calls.  Every expression is evaluated at once, coefficient by coefficient, sums
// over k ascending and chained sums left to right -- the order Eigen's coefficient-based products have.  The decompositions (LDLT, LLT, LU) are plain
// textbook ones: graph-level pins built on this header compare at round-off tolerance, not bit for bit (tests/test_ref_graph_pins.py).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW_IF(x)
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#define EIGEN_STRONG_INLINE inline
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 2
#define EIGEN_MINOR_VERSION 0
#define EIGEN_VERSION_AT_LEAST(x, y, z) 1

namespace Eigen {
enum { Dynamic = -1 };
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Lower = 1, Upper = 2 };
enum { Unaligned = 0, Aligned = 1 };
enum { AlignedBit = 0x40 };
enum { Infinity = -1 };
enum TransformTraits { Isometry = 1, Affine = 2, AffineCompact = 0x12, Projective = 0x20 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
typedef std::ptrdiff_t Index;
typedef std::ptrdiff_t DenseIndex;

template <typename T> struct aligned_allocator : public std::allocator<T> {
    template <typename U> struct rebind { typedef aligned_allocator<U> other; };
    aligned_allocator() {}
    template <typename U> aligned_allocator(const aligned_allocator<U> &) {}
};
template <typename T> struct NumTraits {
    static T epsilon() { return std::numeric_limits<T>::epsilon(); }
    static T dummy_precision() { return T(1e-12); }
    static T highest() { return (std::numeric_limits<T>::max)(); }
    static T lowest() { return std::numeric_limits<T>::lowest(); }
};

template <typename T, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
template <typename Xpr, int BR = Dynamic, int BC = Dynamic> class Block;
template <typename Plain, int MapOpt = 0, typename Stride = void> class Map;
template <typename Xpr> class Diagonal;
template <typename Derived> struct traits;
template <typename T> class Quaternion;

template <typename T, int R, int C, int O, int MR, int MC> struct traits<Matrix<T, R, C, O, MR, MC>> { typedef T Scalar; enum { Rows = R, Cols = C }; };
template <typename T, int R, int C, int O, int MR, int MC> struct traits<const Matrix<T, R, C, O, MR, MC>> { typedef T Scalar; enum { Rows = R, Cols = C }; };
template <typename X, int BR, int BC> struct traits<Block<X, BR, BC>> { typedef typename traits<X>::Scalar Scalar; enum { Rows = BR, Cols = BC }; };
template <typename X> struct traits<Diagonal<X>> { typedef typename traits<X>::Scalar Scalar; enum { Rows = (traits<X>::Rows == traits<X>::Cols ? traits<X>::Rows : Dynamic), Cols = 1 }; };
template <typename P, int O, typename S> struct traits<Map<P, O, S>> { typedef typename traits<P>::Scalar Scalar; enum { Rows = traits<P>::Rows, Cols = traits<P>::Cols }; };

namespace internal {
template <int A, int B> struct pick { enum { v = (A != Dynamic ? A : B) }; };
template <typename T> T pabs(T x) { return x < 0 ? -x : x; }
} // namespace internal

template <typename Derived> class ArrayWrap;
template <typename M> class LDLT;
template <typename M> class LLT;
template <typename M> class PartialPivLU;

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// everything a dense expression can do; Derived supplies rows(), cols(), coeff(i, j) and (when writable) coeffRef(i, j)
template <typename Derived> class MatrixBase {
  public:
    typedef typename traits<Derived>::Scalar Scalar;
    enum { RowsAtCompileTime = traits<Derived>::Rows, ColsAtCompileTime = traits<Derived>::Cols,
           SizeAtCompileTime = (RowsAtCompileTime == Dynamic || ColsAtCompileTime == Dynamic) ? Dynamic : RowsAtCompileTime * ColsAtCompileTime,
           IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1), Flags = 0 };
    typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
    typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposedPlain;
    typedef Scalar RealScalar;
    typedef std::ptrdiff_t Index;

    const Derived &derived() const { return *static_cast<const Derived *>(this); }
    Derived &derived() { return *static_cast<Derived *>(this); }
    int rows() const { return derived().rows(); }
    int cols() const { return derived().cols(); }
    int size() const { return rows() * cols(); }
    Scalar coeff(int i, int j) const { return derived().coeff(i, j); }
    Scalar &coeffRef(int i, int j) { return derived().coeffRef(i, j); }
    Scalar coeff(int k) const { return cols() == 1 ? coeff(k, 0) : (rows() == 1 ? coeff(0, k) : coeff(k % rows(), k / rows())); }
    Scalar &coeffRef(int k) { return cols() == 1 ? coeffRef(k, 0) : (rows() == 1 ? coeffRef(0, k) : coeffRef(k % rows(), k / rows())); }
    Scalar operator()(int i, int j) const { return coeff(i, j); }
    Scalar &operator()(int i, int j) { return coeffRef(i, j); }
    Scalar operator()(int k) const { return coeff(k); }
    Scalar &operator()(int k) { return coeffRef(k); }
    Scalar operator[](int k) const { return coeff(k); }
    Scalar &operator[](int k) { return coeffRef(k); }
    Scalar x() const { return coeff(0); } Scalar y() const { return coeff(1); } Scalar z() const { return coeff(2); } Scalar w() const { return coeff(3); }
    Scalar &x() { return coeffRef(0); } Scalar &y() { return coeffRef(1); } Scalar &z() { return coeffRef(2); } Scalar &w() { return coeffRef(3); }

    PlainObject eval() const { return PlainObject(derived()); }
    template <typename U> Matrix<U, RowsAtCompileTime, ColsAtCompileTime> cast() const {
        Matrix<U, RowsAtCompileTime, ColsAtCompileTime> r(rows(), cols());
        for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = (U)coeff(i, j);
        return r;
    }
    // assignment from any expression (sizes must agree; dynamic plain objects are resized by their own operator=)
    template <typename O> Derived &assign(const MatrixBase<O> &o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = o.coeff(i, j);
        return derived();
    }
    template <typename O> Derived &operator+=(const MatrixBase<O> &o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) += o.coeff(i, j);
        return derived();
    }
    template <typename O> Derived &operator-=(const MatrixBase<O> &o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) -= o.coeff(i, j);
        return derived();
    }
    Derived &operator*=(Scalar s) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) *= s; return derived(); }
    Derived &operator/=(Scalar s) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) /= s; return derived(); }
    template <typename O> Derived &operator*=(const MatrixBase<O> &o) { PlainObject t = (*this) * o; return assign(t); }
    Derived &noalias() { return derived(); }
    Derived &setZero() { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = Scalar(0); return derived(); }
    Derived &setOnes() { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = Scalar(1); return derived(); }
    Derived &setConstant(Scalar v) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = v; return derived(); }
    Derived &fill(Scalar v) { return setConstant(v); }
    Derived &setIdentity() { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = i == j ? Scalar(1) : Scalar(0); return derived(); }

    PlainObject operator-() const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = -coeff(i, j); return r; }
    PlainObject operator*(Scalar s) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = coeff(i, j) * s; return r; }
    PlainObject operator/(Scalar s) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = coeff(i, j) / s; return r; }
    TransposedPlain transpose() const { TransposedPlain r(cols(), rows()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(j, i) = coeff(i, j); return r; }
    TransposedPlain adjoint() const { return transpose(); }
    void transposeInPlace() { PlainObject t = eval(); assert(rows() == cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = t(j, i); }

    Scalar squaredNorm() const { Scalar s = Scalar(0); bool first = true; for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) { Scalar v = coeff(i, j) * coeff(i, j); s = first ? v : s + v; first = false; } return s; }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    PlainObject normalized() const { return (*this) / norm(); }
    void normalize() { (*this) /= norm(); }
    Scalar sum() const { Scalar s = Scalar(0); bool first = true; for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) { s = first ? coeff(i, j) : s + coeff(i, j); first = false; } return s; }
    Scalar mean() const { return sum() / Scalar(size()); }
    Scalar trace() const { Scalar s = coeff(0, 0); for (int i = 1; i < rows(); i++) s += coeff(i, i); return s; }
    Scalar maxCoeff() const { Scalar b = coeff(0, 0); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (coeff(i, j) > b) b = coeff(i, j); return b; }
    Scalar minCoeff() const { Scalar b = coeff(0, 0); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (coeff(i, j) < b) b = coeff(i, j); return b; }
    template <typename I> Scalar maxCoeff(I *idx) const { int b = 0; for (int k = 1; k < size(); k++) if (coeff(k) > coeff(b)) b = k; *idx = (I)b; return coeff(b); }
    template <typename I> Scalar minCoeff(I *idx) const { int b = 0; for (int k = 1; k < size(); k++) if (coeff(k) < coeff(b)) b = k; *idx = (I)b; return coeff(b); }
    template <typename I> Scalar maxCoeff(I *ri, I *ci) const { int bi = 0, bj = 0; for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (coeff(i, j) > coeff(bi, bj)) { bi = i; bj = j; } *ri = (I)bi; *ci = (I)bj; return coeff(bi, bj); }
    template <int P> Scalar lpNorm() const { static_assert(P == Infinity, "only the infinity norm"); Scalar b = Scalar(0); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) b = (std::max)(b, internal::pabs(coeff(i, j))); return b; }
    bool allFinite() const { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (!std::isfinite((double)coeff(i, j))) return false; return true; }
    bool hasNaN() const { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (coeff(i, j) != coeff(i, j)) return true; return false; }
    PlainObject cwiseAbs() const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = internal::pabs(coeff(i, j)); return r; }
    PlainObject cwiseSqrt() const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = std::sqrt(coeff(i, j)); return r; }
    PlainObject cwiseInverse() const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = Scalar(1) / coeff(i, j); return r; }
    template <typename O> PlainObject cwiseProduct(const MatrixBase<O> &o) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = coeff(i, j) * o.coeff(i, j); return r; }
    template <typename O> PlainObject cwiseQuotient(const MatrixBase<O> &o) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = coeff(i, j) / o.coeff(i, j); return r; }
    template <typename O> PlainObject cwiseMax(const MatrixBase<O> &o) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = (std::max)(coeff(i, j), o.coeff(i, j)); return r; }
    template <typename O> PlainObject cwiseMin(const MatrixBase<O> &o) const { PlainObject r(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) r(i, j) = (std::min)(coeff(i, j), o.coeff(i, j)); return r; }
    template <typename O> Scalar dot(const MatrixBase<O> &o) const { assert(size() == o.size()); Scalar s = coeff(0) * o.coeff(0); for (int k = 1; k < size(); k++) s += coeff(k) * o.coeff(k); return s; }
    template <typename O> Matrix<Scalar, 3, 1> cross(const MatrixBase<O> &o) const {
        Matrix<Scalar, 3, 1> r;
        r(0) = coeff(1) * o.coeff(2) - coeff(2) * o.coeff(1); r(1) = coeff(2) * o.coeff(0) - coeff(0) * o.coeff(2); r(2) = coeff(0) * o.coeff(1) - coeff(1) * o.coeff(0);
        return r;
    }
    template <typename O> bool isApprox(const MatrixBase<O> &o, Scalar prec = NumTraits<Scalar>::dummy_precision()) const {
        PlainObject d = (*this) - o;
        return d.squaredNorm() <= prec * prec * (std::min)(squaredNorm(), o.squaredNorm());
    }
    template <typename O> bool operator==(const MatrixBase<O> &o) const { if (rows() != o.rows() || cols() != o.cols()) return false; for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) if (coeff(i, j) != o.coeff(i, j)) return false; return true; }
    template <typename O> bool operator!=(const MatrixBase<O> &o) const { return !(*this == o); }
    Matrix<Scalar, Dynamic, Dynamic> asDiagonal() const { int n = size(); Matrix<Scalar, Dynamic, Dynamic> r(n, n); r.setZero(); for (int i = 0; i < n; i++) r(i, i) = coeff(i); return r; }
    Diagonal<Derived> diagonal() { return Diagonal<Derived>(derived()); } // a writable view: b->diagonal().array() += lambda (block_solver.hpp:577)
    Matrix<Scalar, (RowsAtCompileTime == ColsAtCompileTime ? RowsAtCompileTime : Dynamic), 1> diagonal() const {
        int n = (std::min)(rows(), cols()); Matrix<Scalar, (RowsAtCompileTime == ColsAtCompileTime ? RowsAtCompileTime : Dynamic), 1> r(n, 1);
        for (int i = 0; i < n; i++) r(i) = coeff(i, i);
        return r;
    }
    PlainObject inverse() const;
    Scalar determinant() const;
    LDLT<Matrix<Scalar, Dynamic, Dynamic>> ldlt() const;
    LLT<Matrix<Scalar, Dynamic, Dynamic>> llt() const;
    PartialPivLU<Matrix<Scalar, Dynamic, Dynamic>> lu() const;
    PartialPivLU<Matrix<Scalar, Dynamic, Dynamic>> partialPivLu() const;
    Matrix<Scalar, 3, 1> eulerAngles(int a0, int a1, int a2) const;

    // views (writable on a non-const expression; a const expression hands out evaluated copies)
    Block<Derived> block(int i, int j, int r, int c) { return Block<Derived>(derived(), i, j, r, c); }
    template <int R, int C> Block<Derived, R, C> block(int i, int j) { return Block<Derived, R, C>(derived(), i, j, R, C); }
    template <int R, int C> Block<Derived, R, C> block(int i, int j, int, int) { return Block<Derived, R, C>(derived(), i, j, R, C); }
    Matrix<Scalar, Dynamic, Dynamic> block(int i, int j, int r, int c) const { Matrix<Scalar, Dynamic, Dynamic> m(r, c); for (int b = 0; b < c; b++) for (int a = 0; a < r; a++) m(a, b) = coeff(i + a, j + b); return m; }
    template <int R, int C> Matrix<Scalar, R, C> block(int i, int j) const { Matrix<Scalar, R, C> m; for (int b = 0; b < C; b++) for (int a = 0; a < R; a++) m(a, b) = coeff(i + a, j + b); return m; }
    template <int R, int C> Matrix<Scalar, R, C> block(int i, int j, int, int) const { return block<R, C>(i, j); }
    Block<Derived, RowsAtCompileTime, 1> col(int j) { return Block<Derived, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
    Block<Derived, 1, ColsAtCompileTime> row(int i) { return Block<Derived, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
    Matrix<Scalar, RowsAtCompileTime, 1> col(int j) const { Matrix<Scalar, RowsAtCompileTime, 1> m(rows(), 1); for (int i = 0; i < rows(); i++) m(i) = coeff(i, j); return m; }
    Matrix<Scalar, 1, ColsAtCompileTime> row(int i) const { Matrix<Scalar, 1, ColsAtCompileTime> m(1, cols()); for (int j = 0; j < cols(); j++) m(j) = coeff(i, j); return m; }
    // vector segments
    Block<Derived> seg_(int s, int n) { return cols() == 1 ? Block<Derived>(derived(), s, 0, n, 1) : Block<Derived>(derived(), 0, s, 1, n); }
    template <int N> Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> fseg_(int s) {
        typedef Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> B;
        return cols() == 1 ? B(derived(), s, 0, N, 1) : B(derived(), 0, s, 1, N);
    }
    Matrix<Scalar, Dynamic, 1> cseg_(int s, int n) const { Matrix<Scalar, Dynamic, 1> m(n, 1); for (int i = 0; i < n; i++) m(i) = coeff(s + i); return m; }
    template <int N> Matrix<Scalar, N, 1> cfseg_(int s) const { Matrix<Scalar, N, 1> m; for (int i = 0; i < N; i++) m(i) = coeff(s + i); return m; }
    Block<Derived> segment(int s, int n) { return seg_(s, n); }
    Block<Derived> head(int n) { return seg_(0, n); }
    Block<Derived> tail(int n) { return seg_(size() - n, n); }
    template <int N> auto segment(int s) -> decltype(this->template fseg_<N>(0)) { return fseg_<N>(s); }
    template <int N> auto segment(int s, int) -> decltype(this->template fseg_<N>(0)) { return fseg_<N>(s); }
    template <int N> auto head() -> decltype(this->template fseg_<N>(0)) { return fseg_<N>(0); }
    template <int N> auto tail() -> decltype(this->template fseg_<N>(0)) { return fseg_<N>(size() - N); }
    Matrix<Scalar, Dynamic, 1> segment(int s, int n) const { return cseg_(s, n); }
    Matrix<Scalar, Dynamic, 1> head(int n) const { return cseg_(0, n); }
    Matrix<Scalar, Dynamic, 1> tail(int n) const { return cseg_(size() - n, n); }
    template <int N> Matrix<Scalar, N, 1> segment(int s) const { return cfseg_<N>(s); }
    template <int N> Matrix<Scalar, N, 1> segment(int s, int) const { return cfseg_<N>(s); }
    template <int N> Matrix<Scalar, N, 1> head() const { return cfseg_<N>(0); }
    template <int N> Matrix<Scalar, N, 1> tail() const { return cfseg_<N>(size() - N); }
    // corners and row / column ranges
    template <int R, int C> Block<Derived, R, C> topLeftCorner() { return Block<Derived, R, C>(derived(), 0, 0, R, C); }
    template <int R, int C> Block<Derived, R, C> topRightCorner() { return Block<Derived, R, C>(derived(), 0, cols() - C, R, C); }
    template <int R, int C> Block<Derived, R, C> bottomLeftCorner() { return Block<Derived, R, C>(derived(), rows() - R, 0, R, C); }
    template <int R, int C> Block<Derived, R, C> bottomRightCorner() { return Block<Derived, R, C>(derived(), rows() - R, cols() - C, R, C); }
    template <int R, int C> Matrix<Scalar, R, C> topLeftCorner() const { return block<R, C>(0, 0); }
    template <int R, int C> Matrix<Scalar, R, C> topRightCorner() const { return block<R, C>(0, cols() - C); }
    template <int R, int C> Matrix<Scalar, R, C> bottomLeftCorner() const { return block<R, C>(rows() - R, 0); }
    template <int R, int C> Matrix<Scalar, R, C> bottomRightCorner() const { return block<R, C>(rows() - R, cols() - C); }
    Block<Derived> topLeftCorner(int r, int c) { return block(0, 0, r, c); }
    Block<Derived> topRightCorner(int r, int c) { return block(0, cols() - c, r, c); }
    Block<Derived> bottomLeftCorner(int r, int c) { return block(rows() - r, 0, r, c); }
    Block<Derived> bottomRightCorner(int r, int c) { return block(rows() - r, cols() - c, r, c); }
    Matrix<Scalar, Dynamic, Dynamic> topLeftCorner(int r, int c) const { return block(0, 0, r, c); }
    Matrix<Scalar, Dynamic, Dynamic> topRightCorner(int r, int c) const { return block(0, cols() - c, r, c); }
    Block<Derived> topRows(int n) { return block(0, 0, n, cols()); }
    Block<Derived> bottomRows(int n) { return block(rows() - n, 0, n, cols()); }
    Block<Derived> leftCols(int n) { return block(0, 0, rows(), n); }
    Block<Derived> rightCols(int n) { return block(0, cols() - n, rows(), n); }
    Block<Derived> middleRows(int s, int n) { return block(s, 0, n, cols()); }
    Block<Derived> middleCols(int s, int n) { return block(0, s, rows(), n); }
    template <int N> Block<Derived, N, ColsAtCompileTime> topRows() { return Block<Derived, N, ColsAtCompileTime>(derived(), 0, 0, N, cols()); }
    template <int N> Block<Derived, N, ColsAtCompileTime> bottomRows() { return Block<Derived, N, ColsAtCompileTime>(derived(), rows() - N, 0, N, cols()); }
    template <int N> Block<Derived, RowsAtCompileTime, N> leftCols() { return Block<Derived, RowsAtCompileTime, N>(derived(), 0, 0, rows(), N); }
    template <int N> Block<Derived, RowsAtCompileTime, N> rightCols() { return Block<Derived, RowsAtCompileTime, N>(derived(), 0, cols() - N, rows(), N); }
    Matrix<Scalar, Dynamic, Dynamic> topRows(int n) const { return block(0, 0, n, cols()); }
    Matrix<Scalar, Dynamic, Dynamic> bottomRows(int n) const { return block(rows() - n, 0, n, cols()); }
    Matrix<Scalar, Dynamic, Dynamic> leftCols(int n) const { return block(0, 0, rows(), n); }
    Matrix<Scalar, Dynamic, Dynamic> rightCols(int n) const { return block(0, cols() - n, rows(), n); }
    template <int N> Matrix<Scalar, N, ColsAtCompileTime> topRows() const { Matrix<Scalar, N, ColsAtCompileTime> m(N, cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < N; i++) m(i, j) = coeff(i, j); return m; }
    template <int N> Matrix<Scalar, N, ColsAtCompileTime> bottomRows() const { Matrix<Scalar, N, ColsAtCompileTime> m(N, cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < N; i++) m(i, j) = coeff(rows() - N + i, j); return m; }
    template <int N> Matrix<Scalar, RowsAtCompileTime, N> leftCols() const { Matrix<Scalar, RowsAtCompileTime, N> m(rows(), N); for (int j = 0; j < N; j++) for (int i = 0; i < rows(); i++) m(i, j) = coeff(i, j); return m; }
    template <int N> Matrix<Scalar, RowsAtCompileTime, N> rightCols() const { Matrix<Scalar, RowsAtCompileTime, N> m(rows(), N); for (int j = 0; j < N; j++) for (int i = 0; i < rows(); i++) m(i, j) = coeff(i, cols() - N + j); return m; }

    ArrayWrap<Derived> array() { return ArrayWrap<Derived>(derived()); }
    ArrayWrap<const Derived> array() const { return ArrayWrap<const Derived>(derived()); }
    Derived &matrix() { return derived(); }
    const Derived &matrix() const { return derived(); }
    struct RowwiseOp {
        const Derived &m;
        Matrix<Scalar, RowsAtCompileTime, 1> maxCoeff() const { Matrix<Scalar, RowsAtCompileTime, 1> x(m.rows(), 1); for (int i = 0; i < m.rows(); i++) { Scalar b = m.coeff(i, 0); for (int j = 1; j < m.cols(); j++) if (m.coeff(i, j) > b) b = m.coeff(i, j); x(i) = b; } return x; }
        Matrix<Scalar, RowsAtCompileTime, 1> minCoeff() const { Matrix<Scalar, RowsAtCompileTime, 1> x(m.rows(), 1); for (int i = 0; i < m.rows(); i++) { Scalar b = m.coeff(i, 0); for (int j = 1; j < m.cols(); j++) if (m.coeff(i, j) < b) b = m.coeff(i, j); x(i) = b; } return x; }
        Matrix<Scalar, RowsAtCompileTime, 1> sum() const { Matrix<Scalar, RowsAtCompileTime, 1> x(m.rows(), 1); for (int i = 0; i < m.rows(); i++) { Scalar b = m.coeff(i, 0); for (int j = 1; j < m.cols(); j++) b += m.coeff(i, j); x(i) = b; } return x; }
        Matrix<Scalar, RowsAtCompileTime, 1> mean() const { Matrix<Scalar, RowsAtCompileTime, 1> x = sum(); x /= Scalar(m.cols()); return x; }
        Matrix<Scalar, RowsAtCompileTime, 1> norm() const { Matrix<Scalar, RowsAtCompileTime, 1> x(m.rows(), 1); for (int i = 0; i < m.rows(); i++) { Scalar b = m.coeff(i, 0) * m.coeff(i, 0); for (int j = 1; j < m.cols(); j++) b += m.coeff(i, j) * m.coeff(i, j); x(i) = std::sqrt(b); } return x; }
    };
    struct ColwiseOp {
        const Derived &m;
        Matrix<Scalar, 1, ColsAtCompileTime> maxCoeff() const { Matrix<Scalar, 1, ColsAtCompileTime> x(1, m.cols()); for (int j = 0; j < m.cols(); j++) { Scalar b = m.coeff(0, j); for (int i = 1; i < m.rows(); i++) if (m.coeff(i, j) > b) b = m.coeff(i, j); x(j) = b; } return x; }
        Matrix<Scalar, 1, ColsAtCompileTime> minCoeff() const { Matrix<Scalar, 1, ColsAtCompileTime> x(1, m.cols()); for (int j = 0; j < m.cols(); j++) { Scalar b = m.coeff(0, j); for (int i = 1; i < m.rows(); i++) if (m.coeff(i, j) < b) b = m.coeff(i, j); x(j) = b; } return x; }
        Matrix<Scalar, 1, ColsAtCompileTime> sum() const { Matrix<Scalar, 1, ColsAtCompileTime> x(1, m.cols()); for (int j = 0; j < m.cols(); j++) { Scalar b = m.coeff(0, j); for (int i = 1; i < m.rows(); i++) b += m.coeff(i, j); x(j) = b; } return x; }
        Matrix<Scalar, 1, ColsAtCompileTime> mean() const { Matrix<Scalar, 1, ColsAtCompileTime> x = sum(); x /= Scalar(m.rows()); return x; }
        Matrix<Scalar, 1, ColsAtCompileTime> norm() const { Matrix<Scalar, 1, ColsAtCompileTime> x(1, m.cols()); for (int j = 0; j < m.cols(); j++) { Scalar b = m.coeff(0, j) * m.coeff(0, j); for (int i = 1; i < m.rows(); i++) b += m.coeff(i, j) * m.coeff(i, j); x(j) = std::sqrt(b); } return x; }
    };
    RowwiseOp rowwise() const { return RowwiseOp{derived()}; }
    ColwiseOp colwise() const { return ColwiseOp{derived()}; }
};

// sums, differences and products of any two expressions
template <typename A, typename B>
Matrix<typename A::Scalar, internal::pick<A::RowsAtCompileTime, B::RowsAtCompileTime>::v, internal::pick<A::ColsAtCompileTime, B::ColsAtCompileTime>::v> operator+(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    Matrix<typename A::Scalar, internal::pick<A::RowsAtCompileTime, B::RowsAtCompileTime>::v, internal::pick<A::ColsAtCompileTime, B::ColsAtCompileTime>::v> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); j++) for (int i = 0; i < a.rows(); i++) r(i, j) = a.coeff(i, j) + b.coeff(i, j);
    return r;
}
template <typename A, typename B>
Matrix<typename A::Scalar, internal::pick<A::RowsAtCompileTime, B::RowsAtCompileTime>::v, internal::pick<A::ColsAtCompileTime, B::ColsAtCompileTime>::v> operator-(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    Matrix<typename A::Scalar, internal::pick<A::RowsAtCompileTime, B::RowsAtCompileTime>::v, internal::pick<A::ColsAtCompileTime, B::ColsAtCompileTime>::v> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); j++) for (int i = 0; i < a.rows(); i++) r(i, j) = a.coeff(i, j) - b.coeff(i, j);
    return r;
}
template <typename A, typename B> Matrix<typename A::Scalar, A::RowsAtCompileTime, B::ColsAtCompileTime> operator*(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    assert(a.cols() == b.rows());
    Matrix<typename A::Scalar, A::RowsAtCompileTime, B::ColsAtCompileTime> r(a.rows(), b.cols());
    const int K = a.cols();
    for (int j = 0; j < b.cols(); j++)
        for (int i = 0; i < a.rows(); i++) {
            if (K == 0) { r(i, j) = 0; continue; }
            typename A::Scalar s = a.coeff(i, 0) * b.coeff(0, j);
            for (int k = 1; k < K; k++) s += a.coeff(i, k) * b.coeff(k, j);
            r(i, j) = s;
        }
    return r;
}
template <typename A> typename A::PlainObject operator*(typename A::Scalar s, const MatrixBase<A> &a) {
    typename A::PlainObject r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); j++) for (int i = 0; i < a.rows(); i++) r(i, j) = s * a.coeff(i, j);
    return r;
}
template <typename A, typename = typename std::enable_if<!std::is_same<typename A::Scalar, int>::value>::type> typename A::PlainObject operator*(int s, const MatrixBase<A> &a) { return (typename A::Scalar)s * a; }
template <typename A> std::ostream &operator<<(std::ostream &o, const MatrixBase<A> &m) {
    for (int i = 0; i < m.rows(); i++) { for (int j = 0; j < m.cols(); j++) o << (j ? " " : "") << m.coeff(i, j); if (i + 1 < m.rows()) o << "\n"; }
    return o;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
namespace internal {
template <typename T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage {
    T d[R * C > 0 ? R * C : 1];
    Storage() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    int rows() const { return R; }
    int cols() const { return C; }
    void resize(int r, int c) { assert(r == R && c == C); (void)r; (void)c; }
    T *data() { return d; }
    const T *data() const { return d; }
};
template <typename T, int R, int C> struct Storage<T, R, C, true> {
    std::vector<T> d; int r, c;
    Storage() : r(R == Dynamic ? 0 : R), c(C == Dynamic ? 0 : C) {}
    int rows() const { return r; }
    int cols() const { return c; }
    void resize(int rr, int cc) { if (rr != r || cc != c) { r = rr; c = cc; d.assign((size_t)rr * cc, T(0)); } }
    T *data() { return d.data(); }
    const T *data() const { return d.data(); }
};
} // namespace internal

template <typename Derived> struct CommaInitializer {
    Derived &m; int row, col, bh;
    template <typename O> void put(const MatrixBase<O> &b) {
        if (col == m.cols()) { row += bh; col = 0; }
        for (int j = 0; j < b.cols(); j++) for (int i = 0; i < b.rows(); i++) m.coeffRef(row + i, col + j) = b.coeff(i, j);
        col += b.cols(); bh = b.rows();
    }
    void put(typename traits<Derived>::Scalar v) { if (col == m.cols()) { row += bh; col = 0; } m.coeffRef(row, col) = v; col++; bh = 1; }
    template <typename O> CommaInitializer &operator,(const MatrixBase<O> &b) { put(b); return *this; }
    CommaInitializer &operator,(typename traits<Derived>::Scalar v) { put(v); return *this; }
    Derived &finished() { return m; }
};

template <typename T, int R, int C, int Opt, int MR, int MC> class Matrix : public MatrixBase<Matrix<T, R, C, Opt, MR, MC>> {
    internal::Storage<T, R, C> s;
    enum { RowMaj = (Opt & RowMajor) ? 1 : 0 };

  public:
    typedef MatrixBase<Matrix> Base;
    typedef T Scalar;
    typedef Eigen::Map<Matrix> MapType;
    typedef Eigen::Map<const Matrix> ConstMapType;
    typedef Eigen::Map<Matrix, Aligned> AlignedMapType;
    typedef Eigen::Map<const Matrix, Aligned> ConstAlignedMapType;
    using Base::operator+=; using Base::operator-=; using Base::operator*=;
    Matrix() {}
    explicit Matrix(int n) { if (R == Dynamic && C == Dynamic) s.resize(n, 1); else if (R == Dynamic) s.resize(n, C); else if (C == Dynamic) s.resize(R, n); else if (R * C == 1) s.data()[0] = T(n); }
    template <typename U, typename V, typename = typename std::enable_if<std::is_arithmetic<U>::value && std::is_arithmetic<V>::value>::type> Matrix(U x, V y) { // two coefficients or two sizes
        if (R * C == 2 && R != Dynamic && C != Dynamic) { s.data()[0] = T(x); s.data()[1] = T(y); } else s.resize((int)x, (int)y);
    }
    template <typename U, typename = typename std::enable_if<std::is_floating_point<U>::value && R * C == 1>::type> explicit Matrix(U x) { s.data()[0] = T(x); }
    Matrix(T x, T y, T z) { s.resize(R == Dynamic ? 3 : R, C == Dynamic ? 3 : C); s.data()[0] = x; s.data()[1] = y; s.data()[2] = z; }
    Matrix(T x, T y, T z, T w) { s.resize(R == Dynamic ? 4 : R, C == Dynamic ? 4 : C); s.data()[0] = x; s.data()[1] = y; s.data()[2] = z; s.data()[3] = w; }
    explicit Matrix(const T *p) { for (int i = 0; i < R * C; i++) s.data()[i] = p[i]; }
    Matrix(const Matrix &o) : s(o.s) {}
    template <typename O> Matrix(const MatrixBase<O> &o) { s.resize(fit_r(o), fit_c(o)); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = (T)oc(o, i, j); }
    template <typename O> Matrix(const ArrayWrap<O> &a);
    Matrix &operator=(const Matrix &o) { s = o.s; return *this; }
    template <typename O> Matrix &operator=(const MatrixBase<O> &o) {
        if ((const void *)&o == (const void *)this) return *this;
        Matrix t; t.s.resize(fit_r(o), fit_c(o));
        for (int j = 0; j < t.cols(); j++) for (int i = 0; i < t.rows(); i++) t.coeffRef(i, j) = (T)oc(o, i, j);
        s = t.s; return *this;
    }
    template <typename O> Matrix &operator=(const ArrayWrap<O> &a);
    int rows() const { return s.rows(); }
    int cols() const { return s.cols(); }
    void resize(int r, int c) { s.resize(r, c); }
    void resize(int n) { if (C == 1 || (R == Dynamic && C == Dynamic)) s.resize(n, C == Dynamic ? 1 : C); else s.resize(R == Dynamic ? 1 : R, n); }
    void conservativeResize(int r, int c) { Matrix t(r, c); t.setZero(); for (int j = 0; j < (std::min)(c, cols()); j++) for (int i = 0; i < (std::min)(r, rows()); i++) t(i, j) = coeff(i, j); s = t.s; }
    void conservativeResize(int n) { if (cols() == 1 || (R == Dynamic && C == Dynamic && cols() <= 1)) conservativeResize(n, 1); else conservativeResize(1, n); }
    T coeff(int i, int j) const { assert(i >= 0 && i < rows() && j >= 0 && j < cols()); return s.data()[RowMaj ? (size_t)i * cols() + j : (size_t)j * rows() + i]; }
    T &coeffRef(int i, int j) { assert(i >= 0 && i < rows() && j >= 0 && j < cols()); return s.data()[RowMaj ? (size_t)i * cols() + j : (size_t)j * rows() + i]; }
    using Base::coeff; using Base::coeffRef;
    T *data() { return s.data(); }
    const T *data() const { return s.data(); }
    template <int RR = R, int CC = C, typename = typename std::enable_if<RR == 1 && CC == 1>::type> operator T() const { return s.data()[0]; }
    CommaInitializer<Matrix> operator<<(T v) { CommaInitializer<Matrix> c{*this, 0, 0, 1}; c.put(v); return c; }
    template <typename O> CommaInitializer<Matrix> operator<<(const MatrixBase<O> &b) { CommaInitializer<Matrix> c{*this, 0, 0, 1}; c.put(b); return c; }

    static Matrix Zero() { Matrix m; m.setZero(); return m; }
    static Matrix Zero(int r, int c) { Matrix m(r, c); m.setZero(); return m; }
    static Matrix Zero(int n) { Matrix m; m.resize(n); m.setZero(); return m; }
    static Matrix Ones() { Matrix m; m.setOnes(); return m; }
    static Matrix Ones(int r, int c) { Matrix m(r, c); m.setOnes(); return m; }
    static Matrix Ones(int n) { Matrix m; m.resize(n); m.setOnes(); return m; }
    static Matrix Constant(T v) { Matrix m; m.setConstant(v); return m; }
    static Matrix Constant(int r, int c, T v) { Matrix m(r, c); m.setConstant(v); return m; }
    static Matrix Constant(int n, T v) { Matrix m; m.resize(n); m.setConstant(v); return m; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(int r, int c) { Matrix m(r, c); m.setIdentity(); return m; }
    static Matrix UnitX() { Matrix m; m.setZero(); m(0) = T(1); return m; }
    static Matrix UnitY() { Matrix m; m.setZero(); m(1) = T(1); return m; }
    static Matrix UnitZ() { Matrix m; m.setZero(); m(2) = T(1); return m; }
    static MapType Map(T *p) { return MapType(p); }
    static ConstMapType Map(const T *p) { return ConstMapType(p); }
    static MapType Map(T *p, int n) { return MapType(p, n); }
    static ConstMapType Map(const T *p, int n) { return ConstMapType(p, n); }
    static MapType Map(T *p, int r, int c) { return MapType(p, r, c); }
    static ConstMapType Map(const T *p, int r, int c) { return ConstMapType(p, r, c); }
    using Base::setZero; using Base::setOnes; using Base::setIdentity; using Base::setConstant;
    Matrix &setZero(int r, int c) { s.resize(r, c); return Base::setZero(); }
    Matrix &setZero(int n) { resize(n); return Base::setZero(); }
    Matrix &setOnes(int r, int c) { s.resize(r, c); return Base::setOnes(); }
    Matrix &setIdentity(int r, int c) { s.resize(r, c); return Base::setIdentity(); }
    Matrix &setConstant(int n, T v) { resize(n); return Base::setConstant(v); }
    void swap(Matrix &o) { std::swap(s, o.s); }

  private:
    template <typename O> static typename O::Scalar oc(const MatrixBase<O> &o, int i, int j) { return o.coeff(i, j); }
    template <typename O> int fit_r(const MatrixBase<O> &o) const { assert(R == Dynamic || R == o.rows()); return o.rows(); }
    template <typename O> int fit_c(const MatrixBase<O> &o) const { assert(C == Dynamic || C == o.cols()); return o.cols(); }
};

// a rectangular window of another expression (holds a pointer: the parent must outlive it)
template <typename Xpr, int BR, int BC> class Block : public MatrixBase<Block<Xpr, BR, BC>> {
    Xpr *x; int i0, j0, r, c;

  public:
    typedef MatrixBase<Block> Base;
    typedef typename traits<Xpr>::Scalar Scalar;
    using Base::operator+=; using Base::operator-=; using Base::operator*=;
    Block(Xpr &xx, int i, int j, int rr, int cc) : x(&xx), i0(i), j0(j), r(rr), c(cc) { assert(i >= 0 && j >= 0 && i + rr <= xx.rows() && j + cc <= xx.cols()); }
    int rows() const { return r; }
    int cols() const { return c; }
    Scalar coeff(int i, int j) const { return const_cast<const Xpr *>(x)->coeff(i0 + i, j0 + j); }
    Scalar &coeffRef(int i, int j) { return x->coeffRef(i0 + i, j0 + j); }
    using Base::coeff; using Base::coeffRef;
    Block &operator=(const Block &o) { typename Base::PlainObject t(o); return this->assign(t); }
    template <typename O> Block &operator=(const MatrixBase<O> &o) { typename O::PlainObject t(o); return this->assign(t); } // (through a copy: o may alias the parent)
    template <typename O> Block &operator=(const ArrayWrap<O> &a);
    CommaInitializer<Block> operator<<(Scalar v) { CommaInitializer<Block> k{*this, 0, 0, 1}; k.put(v); return k; }
    template <typename O> CommaInitializer<Block> operator<<(const MatrixBase<O> &b) { CommaInitializer<Block> k{*this, 0, 0, 1}; k.put(b); return k; }
};

// the main diagonal of another expression as a column vector (holds a pointer to it)
template <typename Xpr> class Diagonal : public MatrixBase<Diagonal<Xpr>> {
    Xpr *x;

  public:
    typedef MatrixBase<Diagonal> Base;
    typedef typename traits<Xpr>::Scalar Scalar;
    using Base::operator+=; using Base::operator-=; using Base::operator*=;
    explicit Diagonal(Xpr &xx) : x(&xx) {}
    int rows() const { return (std::min)(x->rows(), x->cols()); }
    int cols() const { return 1; }
    Scalar coeff(int i, int) const { return const_cast<const Xpr *>(x)->coeff(i, i); }
    Scalar &coeffRef(int i, int) { return x->coeffRef(i, i); }
    using Base::coeff; using Base::coeffRef;
    Diagonal &operator=(const Diagonal &o) { typename Base::PlainObject t(o); return this->assign(t); }
    template <typename O> Diagonal &operator=(const MatrixBase<O> &o) { typename O::PlainObject t(o); return this->assign(t); }
};

// raw memory seen as a matrix (column-major unless the plain type is row-major)
template <typename P, int MapOpt, typename Stride> class Map : public MatrixBase<Map<P, MapOpt, Stride>> {
    typedef typename std::remove_const<P>::type Plain;
    typedef typename traits<Plain>::Scalar T;
    typedef typename std::conditional<std::is_const<P>::value, const T, T>::type Elem;
    Elem *p; int r, c;
    enum { R = traits<Plain>::Rows, C = traits<Plain>::Cols };

  public:
    typedef MatrixBase<Map> Base;
    typedef T Scalar;
    using Base::operator+=; using Base::operator-=; using Base::operator*=;
    explicit Map(Elem *pp) : p(pp), r(R), c(C) { static_assert(R != Dynamic && C != Dynamic, "sizes needed"); }
    Map(Elem *pp, int n) : p(pp), r(R == Dynamic ? n : R), c(C != Dynamic ? C : (R == Dynamic ? 1 : n)) {} // a vector of n coefficients
    Map(Elem *pp, int rr, int cc) : p(pp), r(rr), c(cc) {}
    int rows() const { return r; }
    int cols() const { return c; }
    T coeff(int i, int j) const { assert(i >= 0 && i < r && j >= 0 && j < c); return p[(size_t)j * r + i]; }
    T &coeffRef(int i, int j) { assert(i >= 0 && i < r && j >= 0 && j < c); return const_cast<T &>(p[(size_t)j * r + i]); }
    using Base::coeff; using Base::coeffRef;
    Elem *data() const { return p; }
    void resize(int rr, int cc) { assert(rr == r && cc == c); (void)rr; (void)cc; } // (a map cannot change its size: Eigen asserts the same)
    Map &operator=(const Map &o) { Plain t(o); return this->assign(t); }
    template <typename O> Map &operator=(const MatrixBase<O> &o) { typename O::PlainObject t(o); return this->assign(t); }
    CommaInitializer<Map> operator<<(T v) { CommaInitializer<Map> k{*this, 0, 0, 1}; k.put(v); return k; }
    // placement-new re-seating, as g2o's mapHessianMemory does: new (&_hessian) HessianBlockType(d, D, D)
};

// coefficient-wise view: m.array() op m2.array(), assignable back to a matrix
template <typename Derived> class ArrayWrap {
    typedef typename std::remove_const<Derived>::type D;
    Derived &m;

  public:
    typedef typename traits<D>::Scalar Scalar;
    typedef Matrix<Scalar, traits<D>::Rows, traits<D>::Cols> Plain;
    typedef ArrayWrap<const Plain> Tmp;
    explicit ArrayWrap(Derived &mm) : m(mm) {}
    int rows() const { return m.rows(); }
    int cols() const { return m.cols(); }
    Scalar coeff(int i, int j) const { return m.coeff(i, j); }
    const D &matrix() const { return m; }
    Derived &matrix() { return m; }
    // results own their storage through a shared plain matrix
    struct Owned : public ArrayWrap<const Plain> {
        std::shared_ptr<Plain> hold;
        explicit Owned(std::shared_ptr<Plain> h) : ArrayWrap<const Plain>(*h), hold(h) {}
    };
    template <typename F> Owned map1(F f) const { auto h = std::make_shared<Plain>(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) (*h)(i, j) = f(coeff(i, j)); return Owned(h); }
    template <typename O, typename F> Owned map2(const ArrayWrap<O> &o, F f) const { assert(rows() == o.rows() && cols() == o.cols()); auto h = std::make_shared<Plain>(rows(), cols()); for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) (*h)(i, j) = f(coeff(i, j), o.coeff(i, j)); return Owned(h); }
    template <typename O> Owned operator*(const ArrayWrap<O> &o) const { return map2(o, [](Scalar a, Scalar b) { return a * b; }); }
    template <typename O> Owned operator/(const ArrayWrap<O> &o) const { return map2(o, [](Scalar a, Scalar b) { return a / b; }); }
    template <typename O> Owned operator+(const ArrayWrap<O> &o) const { return map2(o, [](Scalar a, Scalar b) { return a + b; }); }
    template <typename O> Owned operator-(const ArrayWrap<O> &o) const { return map2(o, [](Scalar a, Scalar b) { return a - b; }); }
    Owned operator*(Scalar s) const { return map1([s](Scalar a) { return a * s; }); }
    Owned operator/(Scalar s) const { return map1([s](Scalar a) { return a / s; }); }
    Owned operator+(Scalar s) const { return map1([s](Scalar a) { return a + s; }); }
    Owned operator-(Scalar s) const { return map1([s](Scalar a) { return a - s; }); }
    Owned abs() const { return map1([](Scalar a) { return internal::pabs(a); }); }
    Owned cwiseAbs() const { return abs(); }
    Owned sqrt() const { return map1([](Scalar a) { return std::sqrt(a); }); }
    Owned square() const { return map1([](Scalar a) { return a * a; }); }
    Owned inverse() const { return map1([](Scalar a) { return Scalar(1) / a; }); }
    Scalar sum() const { return m.sum(); }
    Scalar maxCoeff() const { return m.maxCoeff(); }
    Scalar minCoeff() const { return m.minCoeff(); }
    template <typename O> ArrayWrap &operator=(const ArrayWrap<O> &o) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) m.coeffRef(i, j) = o.coeff(i, j); return *this; }
    template <typename O> ArrayWrap &operator*=(const ArrayWrap<O> &o) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) m.coeffRef(i, j) *= o.coeff(i, j); return *this; }
    template <typename O> ArrayWrap &operator/=(const ArrayWrap<O> &o) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) m.coeffRef(i, j) /= o.coeff(i, j); return *this; }
    ArrayWrap &operator+=(Scalar s) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) m.coeffRef(i, j) += s; return *this; }
    ArrayWrap &operator-=(Scalar s) { for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) m.coeffRef(i, j) -= s; return *this; }
};
template <typename T, int R, int C, int O, int MR, int MC> template <typename A> Matrix<T, R, C, O, MR, MC>::Matrix(const ArrayWrap<A> &a) {
    s.resize(a.rows(), a.cols());
    for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) coeffRef(i, j) = a.coeff(i, j);
}
template <typename T, int R, int C, int O, int MR, int MC> template <typename A> Matrix<T, R, C, O, MR, MC> &Matrix<T, R, C, O, MR, MC>::operator=(const ArrayWrap<A> &a) {
    Matrix t(a); s = t.s; return *this;
}
template <typename X, int BR, int BC> template <typename A> Block<X, BR, BC> &Block<X, BR, BC>::operator=(const ArrayWrap<A> &a) {
    typename Base::PlainObject t(a); return this->assign(t);
}

#define EIGEN_FULL_TYPEDEFS(T, S)                                                                                                       \
    typedef Matrix<T, 2, 1> Vector2##S; typedef Matrix<T, 3, 1> Vector3##S; typedef Matrix<T, 4, 1> Vector4##S;                           \
    typedef Matrix<T, Dynamic, 1> VectorX##S; typedef Matrix<T, 1, 2> RowVector2##S; typedef Matrix<T, 1, 3> RowVector3##S;               \
    typedef Matrix<T, 1, 4> RowVector4##S; typedef Matrix<T, 1, Dynamic> RowVectorX##S;                                                   \
    typedef Matrix<T, 2, 2> Matrix2##S; typedef Matrix<T, 3, 3> Matrix3##S; typedef Matrix<T, 4, 4> Matrix4##S;                           \
    typedef Matrix<T, Dynamic, Dynamic> MatrixX##S; typedef Matrix<T, 2, Dynamic> Matrix2X##S; typedef Matrix<T, 3, Dynamic> Matrix3X##S; \
    typedef Matrix<T, 4, Dynamic> Matrix4X##S; typedef Matrix<T, Dynamic, 2> MatrixX2##S; typedef Matrix<T, Dynamic, 3> MatrixX3##S;
EIGEN_FULL_TYPEDEFS(double, d)
EIGEN_FULL_TYPEDEFS(float, f)
EIGEN_FULL_TYPEDEFS(int, i)

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// decompositions (textbook forms; Eigen's are blocked / pivoted differently: results agree to round-off, not to the bit)
template <typename M> class PartialPivLU {
    typedef typename M::Scalar T;
    Matrix<T, Dynamic, Dynamic> lu; std::vector<int> perm; int sign = 1;

  public:
    PartialPivLU() {}
    template <typename O> explicit PartialPivLU(const MatrixBase<O> &a) { compute(a); }
    template <typename O> PartialPivLU &compute(const MatrixBase<O> &a) {
        const int n = a.rows(); assert(n == a.cols());
        lu = a; perm.resize(n); sign = 1;
        for (int i = 0; i < n; i++) perm[i] = i;
        for (int k = 0; k < n; k++) {
            int p = k;
            for (int i = k + 1; i < n; i++) if (internal::pabs(lu(i, k)) > internal::pabs(lu(p, k))) p = i;
            if (p != k) { for (int j = 0; j < n; j++) std::swap(lu(k, j), lu(p, j)); std::swap(perm[k], perm[p]); sign = -sign; }
            for (int i = k + 1; i < n; i++) { lu(i, k) /= lu(k, k); for (int j = k + 1; j < n; j++) lu(i, j) -= lu(i, k) * lu(k, j); }
        }
        return *this;
    }
    template <typename O> Matrix<T, Dynamic, O::ColsAtCompileTime> solve(const MatrixBase<O> &b) const {
        const int n = lu.rows();
        Matrix<T, Dynamic, O::ColsAtCompileTime> x(n, b.cols());
        for (int c = 0; c < b.cols(); c++) {
            for (int i = 0; i < n; i++) { T s = b.coeff(perm[i], c); for (int k = 0; k < i; k++) s -= lu(i, k) * x(k, c); x(i, c) = s; }
            for (int i = n - 1; i >= 0; i--) { T s = x(i, c); for (int k = i + 1; k < n; k++) s -= lu(i, k) * x(k, c); x(i, c) = s / lu(i, i); }
        }
        return x;
    }
    Matrix<T, Dynamic, Dynamic> inverse() const { const int n = lu.rows(); return solve(Matrix<T, Dynamic, Dynamic>::Identity(n, n)); }
    T determinant() const { T d = T(sign); for (int i = 0; i < lu.rows(); i++) d *= lu(i, i); return d; }
};
template <typename M> class FullPivLU : public PartialPivLU<M> {
  public:
    template <typename O> explicit FullPivLU(const MatrixBase<O> &a) : PartialPivLU<M>(a) {}
};

template <typename M> class LLT {
    typedef typename M::Scalar T;
    Matrix<T, Dynamic, Dynamic> L; bool ok = false;

  public:
    LLT() {}
    template <typename O> explicit LLT(const MatrixBase<O> &a) { compute(a); }
    template <typename O> LLT &compute(const MatrixBase<O> &a) {
        const int n = a.rows();
        L = Matrix<T, Dynamic, Dynamic>::Zero(n, n); ok = true;
        for (int j = 0; j < n; j++) {
            T d = a.coeff(j, j);
            for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
            if (!(d > T(0))) { ok = false; return *this; }
            L(j, j) = std::sqrt(d);
            for (int i = j + 1; i < n; i++) { T s = a.coeff(i, j); for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k); L(i, j) = s / L(j, j); }
        }
        return *this;
    }
    ComputationInfo info() const { return ok ? Success : NumericalIssue; }
    const Matrix<T, Dynamic, Dynamic> &matrixL() const { return L; }
    Matrix<T, Dynamic, Dynamic> matrixU() const { return L.transpose(); }
    template <typename O> Matrix<T, O::RowsAtCompileTime, O::ColsAtCompileTime> solve(const MatrixBase<O> &b) const {
        const int n = L.rows();
        Matrix<T, O::RowsAtCompileTime, O::ColsAtCompileTime> x(n, b.cols());
        for (int c = 0; c < b.cols(); c++) {
            for (int i = 0; i < n; i++) { T s = b.coeff(i, c); for (int k = 0; k < i; k++) s -= L(i, k) * x(k, c); x(i, c) = s / L(i, i); }
            for (int i = n - 1; i >= 0; i--) { T s = x(i, c); for (int k = i + 1; k < n; k++) s -= L(k, i) * x(k, c); x(i, c) = s / L(i, i); }
        }
        return x;
    }
};
template <typename M> class LDLT { // unpivoted L D L^T (Eigen pivots on the largest diagonal entry; for the SPD systems of the BA both are stable)
    typedef typename M::Scalar T;
    Matrix<T, Dynamic, Dynamic> L; Matrix<T, Dynamic, 1> D; bool positive = false, ok = false;

  public:
    LDLT() {}
    template <typename O> explicit LDLT(const MatrixBase<O> &a) { compute(a); }
    template <typename O> LDLT &compute(const MatrixBase<O> &a) {
        const int n = a.rows();
        L = Matrix<T, Dynamic, Dynamic>::Identity(n, n); D = Matrix<T, Dynamic, 1>::Zero(n); positive = true; ok = true;
        for (int j = 0; j < n; j++) {
            T d = a.coeff(j, j);
            for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k) * D(k);
            D(j) = d;
            if (!(d > T(0))) positive = false;
            if (d == T(0)) { ok = false; continue; }
            for (int i = j + 1; i < n; i++) { T s = a.coeff(i, j); for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k) * D(k); L(i, j) = s / d; }
        }
        return *this;
    }
    bool isPositive() const { return positive; }
    bool isNegative() const { for (int i = 0; i < D.size(); i++) if (D(i) > T(0)) return false; return true; }
    ComputationInfo info() const { return ok ? Success : NumericalIssue; }
    const Matrix<T, Dynamic, 1> &vectorD() const { return D; }
    template <typename O> Matrix<T, O::RowsAtCompileTime, O::ColsAtCompileTime> solve(const MatrixBase<O> &b) const {
        const int n = L.rows();
        Matrix<T, O::RowsAtCompileTime, O::ColsAtCompileTime> x(n, b.cols());
        for (int c = 0; c < b.cols(); c++) {
            for (int i = 0; i < n; i++) { T s = b.coeff(i, c); for (int k = 0; k < i; k++) s -= L(i, k) * x(k, c); x(i, c) = s; }
            for (int i = 0; i < n; i++) x(i, c) = x(i, c) / D(i);
            for (int i = n - 1; i >= 0; i--) { T s = x(i, c); for (int k = i + 1; k < n; k++) s -= L(k, i) * x(k, c); x(i, c) = s; }
        }
        return x;
    }
};
// symmetric eigenvalues by cyclic Jacobi rotations (only optimizable_graph.cpp's information-matrix check asks for them)
template <typename M> class SelfAdjointEigenSolver {
    typedef typename M::Scalar T;
    Matrix<T, Dynamic, 1> ev; Matrix<T, Dynamic, Dynamic> V;

  public:
    SelfAdjointEigenSolver() {}
    template <typename O> explicit SelfAdjointEigenSolver(const MatrixBase<O> &a) { compute(a); }
    template <typename O> SelfAdjointEigenSolver &compute(const MatrixBase<O> &a, int = 0) {
        const int n = a.rows();
        Matrix<T, Dynamic, Dynamic> A = a; V = Matrix<T, Dynamic, Dynamic>::Identity(n, n);
        for (int sweep = 0; sweep < 100; sweep++) {
            T off = 0;
            for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A(p, q) * A(p, q);
            if (off < T(1e-300)) break;
            for (int p = 0; p < n; p++)
                for (int q = p + 1; q < n; q++) {
                    if (A(p, q) == T(0)) continue;
                    T th = (A(q, q) - A(p, p)) / (2 * A(p, q)), t = (th >= 0 ? T(1) : T(-1)) / (internal::pabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
                    for (int k = 0; k < n; k++) { T akp = A(k, p), akq = A(k, q); A(k, p) = c * akp - s * akq; A(k, q) = s * akp + c * akq; }
                    for (int k = 0; k < n; k++) { T apk = A(p, k), aqk = A(q, k); A(p, k) = c * apk - s * aqk; A(q, k) = s * apk + c * aqk; }
                    for (int k = 0; k < n; k++) { T vkp = V(k, p), vkq = V(k, q); V(k, p) = c * vkp - s * vkq; V(k, q) = s * vkp + c * vkq; }
                }
        }
        ev.resize(n);
        std::vector<int> idx(n);
        for (int i = 0; i < n; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return A(x, x) < A(y, y); });
        Matrix<T, Dynamic, Dynamic> W(n, n);
        for (int i = 0; i < n; i++) { ev(i) = A(idx[i], idx[i]); for (int k = 0; k < n; k++) W(k, i) = V(k, idx[i]); }
        V = W;
        return *this;
    }
    const Matrix<T, Dynamic, 1> &eigenvalues() const { return ev; }
    const Matrix<T, Dynamic, Dynamic> &eigenvectors() const { return V; }
};
enum { ComputeEigenvectors = 0x80, EigenvaluesOnly = 0x40 };

template <typename D> typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
    const int n = rows(); assert(n == cols());
    PlainObject r(n, n);
    if (RowsAtCompileTime == 1 || (RowsAtCompileTime == Dynamic && n == 1)) { r(0, 0) = Scalar(1) / coeff(0, 0); return r; }
    if (RowsAtCompileTime == 2) { // Eigen: compute_inverse<.., 2>: the determinant's reciprocal times the adjugate
        const Scalar invdet = Scalar(1) / (coeff(0, 0) * coeff(1, 1) - coeff(1, 0) * coeff(0, 1));
        r(0, 0) = coeff(1, 1) * invdet; r(1, 0) = -coeff(1, 0) * invdet; r(0, 1) = -coeff(0, 1) * invdet; r(1, 1) = coeff(0, 0) * invdet;
        return r;
    }
    if (RowsAtCompileTime == 3) { // Eigen: compute_inverse<.., 3>: cofactors, determinant along the first column, one reciprocal
        auto cf = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return coeff(i1, j1) * coeff(i2, j2) - coeff(i1, j2) * coeff(i2, j1); };
        const Scalar c00 = cf(0, 0), c10 = cf(1, 0), c20 = cf(2, 0);
        const Scalar det = (c00 * coeff(0, 0) + c10 * coeff(1, 0)) + c20 * coeff(2, 0), inv = Scalar(1) / det;
        r(0, 0) = c00 * inv; r(0, 1) = c10 * inv; r(0, 2) = c20 * inv;
        r(1, 0) = cf(0, 1) * inv; r(1, 1) = cf(1, 1) * inv; r(1, 2) = cf(2, 1) * inv;
        r(2, 0) = cf(0, 2) * inv; r(2, 1) = cf(1, 2) * inv; r(2, 2) = cf(2, 2) * inv;
        return r;
    }
    Matrix<Scalar, Dynamic, Dynamic> a(derived());
    Matrix<Scalar, Dynamic, Dynamic> x = PartialPivLU<Matrix<Scalar, Dynamic, Dynamic>>(a).inverse();
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) r(i, j) = x(i, j);
    return r;
}
template <typename D> typename MatrixBase<D>::Scalar MatrixBase<D>::determinant() const {
    const int n = rows();
    if (n == 1) return coeff(0, 0);
    if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(1, 0) * coeff(0, 1);
    if (n == 3) return coeff(0, 0) * (coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1)) - coeff(0, 1) * (coeff(1, 0) * coeff(2, 2) - coeff(1, 2) * coeff(2, 0)) + coeff(0, 2) * (coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0));
    Matrix<Scalar, Dynamic, Dynamic> a(derived());
    return PartialPivLU<Matrix<Scalar, Dynamic, Dynamic>>(a).determinant();
}
template <typename D> LDLT<Matrix<typename MatrixBase<D>::Scalar, Dynamic, Dynamic>> MatrixBase<D>::ldlt() const { return LDLT<Matrix<Scalar, Dynamic, Dynamic>>(derived()); }
template <typename D> LLT<Matrix<typename MatrixBase<D>::Scalar, Dynamic, Dynamic>> MatrixBase<D>::llt() const { return LLT<Matrix<Scalar, Dynamic, Dynamic>>(derived()); }
template <typename D> PartialPivLU<Matrix<typename MatrixBase<D>::Scalar, Dynamic, Dynamic>> MatrixBase<D>::lu() const { return PartialPivLU<Matrix<Scalar, Dynamic, Dynamic>>(derived()); }
template <typename D> PartialPivLU<Matrix<typename MatrixBase<D>::Scalar, Dynamic, Dynamic>> MatrixBase<D>::partialPivLu() const { return lu(); }
template <typename D> Matrix<typename MatrixBase<D>::Scalar, 3, 1> MatrixBase<D>::eulerAngles(int a0, int a1, int a2) const { // Eigen/src/Geometry/EulerAngles.h
    Matrix<Scalar, 3, 1> res;
    const int odd = ((a0 + 1) % 3 == a1) ? 0 : 1, i = a0, j = (a0 + 1 + odd) % 3, k = (a0 + 2 - odd) % 3;
    if (a0 == a2) {
        res[0] = std::atan2(coeff(j, i), coeff(k, i));
        if ((odd && res[0] < Scalar(0)) || ((!odd) && res[0] > Scalar(0))) {
            res[0] = (res[0] > Scalar(0)) ? res[0] - Scalar(M_PI) : res[0] + Scalar(M_PI);
            Scalar s2 = std::sqrt(coeff(j, i) * coeff(j, i) + coeff(k, i) * coeff(k, i));
            res[1] = -std::atan2(s2, coeff(i, i));
        } else {
            Scalar s2 = std::sqrt(coeff(j, i) * coeff(j, i) + coeff(k, i) * coeff(k, i));
            res[1] = std::atan2(s2, coeff(i, i));
        }
        Scalar s1 = std::sin(res[0]), c1 = std::cos(res[0]);
        res[2] = std::atan2(c1 * coeff(j, k) - s1 * coeff(k, k), c1 * coeff(j, j) - s1 * coeff(k, j));
    } else {
        res[0] = std::atan2(coeff(j, k), coeff(k, k));
        Scalar c2 = std::sqrt(coeff(i, i) * coeff(i, i) + coeff(i, j) * coeff(i, j));
        if ((odd && res[0] < Scalar(0)) || ((!odd) && res[0] > Scalar(0))) {
            res[0] = (res[0] > Scalar(0)) ? res[0] - Scalar(M_PI) : res[0] + Scalar(M_PI);
            res[1] = std::atan2(-coeff(i, k), -c2);
        } else
            res[1] = std::atan2(-coeff(i, k), c2);
        Scalar s1 = std::sin(res[0]), c1 = std::cos(res[0]);
        res[2] = std::atan2(s1 * coeff(k, i) - c1 * coeff(j, i), c1 * coeff(j, j) - s1 * coeff(k, j));
    }
    if (!odd) res = -res;
    return res;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// geometry: Quaternion (Eigen's generic, non-SIMD code paths), AngleAxis, the isometry / affine transform
template <typename T> class AngleAxis;
template <typename T> class Quaternion {
    Matrix<T, 4, 1> c; // x y z w, like Eigen's coeffs()
  public:
    typedef T Scalar;
    Quaternion() {}
    Quaternion(T w, T x, T y, T z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    explicit Quaternion(const T *p) { for (int i = 0; i < 4; i++) c[i] = p[i]; }
    template <typename O, typename = typename std::enable_if<O::RowsAtCompileTime == 4 && O::ColsAtCompileTime == 1>::type> explicit Quaternion(const MatrixBase<O> &v, int = 0) { for (int i = 0; i < 4; i++) c[i] = v.coeff(i); }
    template <typename O, typename = typename std::enable_if<O::RowsAtCompileTime == 3 && O::ColsAtCompileTime == 3>::type> explicit Quaternion(const MatrixBase<O> &mat) { *this = mat; }
    explicit Quaternion(const AngleAxis<T> &aa);
    template <typename O> typename std::enable_if<O::RowsAtCompileTime == 3 && O::ColsAtCompileTime == 3, Quaternion &>::type operator=(const MatrixBase<O> &mat) { // quaternionbase_assign_impl<Other,3,3>
        T t = mat.coeff(0, 0) + mat.coeff(1, 1) + mat.coeff(2, 2);
        if (t > T(0)) {
            t = std::sqrt(t + T(1.0)); w() = T(0.5) * t; t = T(0.5) / t;
            x() = (mat.coeff(2, 1) - mat.coeff(1, 2)) * t; y() = (mat.coeff(0, 2) - mat.coeff(2, 0)) * t; z() = (mat.coeff(1, 0) - mat.coeff(0, 1)) * t;
        } else {
            int i = 0;
            if (mat.coeff(1, 1) > mat.coeff(0, 0)) i = 1;
            if (mat.coeff(2, 2) > mat.coeff(i, i)) i = 2;
            int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat.coeff(i, i) - mat.coeff(j, j) - mat.coeff(k, k) + T(1.0));
            c[i] = T(0.5) * t; t = T(0.5) / t;
            w() = (mat.coeff(k, j) - mat.coeff(j, k)) * t; c[j] = (mat.coeff(j, i) + mat.coeff(i, j)) * t; c[k] = (mat.coeff(k, i) + mat.coeff(i, k)) * t;
        }
        return *this;
    }
    Quaternion &operator=(const AngleAxis<T> &aa);
    static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
    T &x() { return c[0]; } T &y() { return c[1]; } T &z() { return c[2]; } T &w() { return c[3]; }
    T x() const { return c[0]; } T y() const { return c[1]; } T z() const { return c[2]; } T w() const { return c[3]; }
    Matrix<T, 4, 1> &coeffs() { return c; }
    const Matrix<T, 4, 1> &coeffs() const { return c; }
    Matrix<T, 3, 1> vec() const { return Matrix<T, 3, 1>(c[0], c[1], c[2]); }
    Quaternion &setIdentity() { c[0] = c[1] = c[2] = 0; c[3] = 1; return *this; }
    T squaredNorm() const { return c.squaredNorm(); }
    T norm() const { return c.norm(); }
    void normalize() { c /= norm(); }
    Quaternion normalized() const { Quaternion q = *this; q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
    Quaternion inverse() const { T n2 = squaredNorm(); Quaternion q = conjugate(); q.c /= n2; return q; }
    T dot(const Quaternion &o) const { return c.dot(o.c); }
    T angularDistance(const Quaternion &o) const { Quaternion d = (*this) * o.conjugate(); return T(2) * std::atan2(d.vec().norm(), internal::pabs(d.w())); }
    Quaternion operator*(const Quaternion &b) const { // quat_product (generic)
        const Quaternion &a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(), a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(), a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaternion &operator*=(const Quaternion &b) { *this = *this * b; return *this; }
    template <typename O> Matrix<T, 3, 1> operator*(const MatrixBase<O> &v) const { return _transformVector(Matrix<T, 3, 1>(v)); }
    Matrix<T, 3, 1> _transformVector(const Matrix<T, 3, 1> &v) const {
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
    Matrix<T, 3, 3> matrix() const { return toRotationMatrix(); }
    template <typename U> Quaternion<U> cast() const { return Quaternion<U>((U)w(), (U)x(), (U)y(), (U)z()); }
    Quaternion slerp(T t, const Quaternion &o) const {
        const T one = T(1) - NumTraits<T>::epsilon();
        T d = dot(o), ad = internal::pabs(d), s0, s1;
        if (ad >= one) { s0 = T(1) - t; s1 = t; }
        else { T th = std::acos(ad), st = std::sin(th); s0 = std::sin((T(1) - t) * th) / st; s1 = std::sin(t * th) / st; }
        if (d < T(0)) s1 = -s1;
        Quaternion r; r.c = s0 * c + s1 * o.c; return r;
    }
};
typedef Quaternion<double> Quaterniond; typedef Quaternion<float> Quaternionf;

template <typename T> class AngleAxis {
    Matrix<T, 3, 1> ax; T ang;
  public:
    AngleAxis() : ang(0) {}
    template <typename O> AngleAxis(T a, const MatrixBase<O> &v) : ax(v), ang(a) {}
    explicit AngleAxis(const Quaternion<T> &q) { // AngleAxis::operator=(QuaternionBase) of Eigen 3.2
        T n2 = q.vec().squaredNorm();
        if (n2 < NumTraits<T>::dummy_precision() * NumTraits<T>::dummy_precision()) { ang = 0; ax = Matrix<T, 3, 1>(1, 0, 0); }
        else { ang = T(2) * std::acos((std::min)((std::max)(T(-1), q.w()), T(1))); ax = q.vec() / std::sqrt(n2); }
    }
    template <typename O, typename = typename std::enable_if<O::RowsAtCompileTime == 3 && O::ColsAtCompileTime == 3>::type> explicit AngleAxis(const MatrixBase<O> &m) { *this = AngleAxis(Quaternion<T>(m)); }
    T angle() const { return ang; } T &angle() { return ang; }
    const Matrix<T, 3, 1> &axis() const { return ax; } Matrix<T, 3, 1> &axis() { return ax; }
    Matrix<T, 3, 3> toRotationMatrix() const { // AngleAxis::toRotationMatrix
        Matrix<T, 3, 3> res;
        const T s = std::sin(ang), co = std::cos(ang);
        Matrix<T, 3, 1> sin_axis = s * ax, cos1_axis = (T(1) - co) * ax;
        T tmp;
        tmp = cos1_axis.x() * ax.y(); res(0, 1) = tmp - sin_axis.z(); res(1, 0) = tmp + sin_axis.z();
        tmp = cos1_axis.x() * ax.z(); res(0, 2) = tmp + sin_axis.y(); res(2, 0) = tmp - sin_axis.y();
        tmp = cos1_axis.y() * ax.z(); res(1, 2) = tmp - sin_axis.x(); res(2, 1) = tmp + sin_axis.x();
        res(0, 0) = cos1_axis.x() * ax.x() + co; res(1, 1) = cos1_axis.y() * ax.y() + co; res(2, 2) = cos1_axis.z() * ax.z() + co;
        return res;
    }
    Matrix<T, 3, 3> matrix() const { return toRotationMatrix(); }
    Quaternion<T> operator*(const AngleAxis &o) const { return Quaternion<T>(*this) * Quaternion<T>(o); }
    Quaternion<T> operator*(const Quaternion<T> &o) const { return Quaternion<T>(*this) * o; }
    template <typename O> Matrix<T, 3, 1> operator*(const MatrixBase<O> &v) const { return toRotationMatrix() * v; }
    AngleAxis inverse() const { return AngleAxis(-ang, ax); }
};
typedef AngleAxis<double> AngleAxisd; typedef AngleAxis<float> AngleAxisf;
template <typename T> Quaternion<T>::Quaternion(const AngleAxis<T> &aa) { *this = aa; }
template <typename T> Quaternion<T> &Quaternion<T>::operator=(const AngleAxis<T> &aa) {
    const T ha = T(0.5) * aa.angle(), s = std::sin(ha);
    w() = std::cos(ha); x() = s * aa.axis().x(); y() = s * aa.axis().y(); z() = s * aa.axis().z();
    return *this;
}
template <typename T> Quaternion<T> operator*(const Quaternion<T> &q, const AngleAxis<T> &a) { return q * Quaternion<T>(a); }

template <typename T, int Dim, int Mode, int Opt = 0> class Transform {
    Matrix<T, Dim + 1, Dim + 1> m;
  public:
    Transform() { m.setIdentity(); }
    explicit Transform(const Quaternion<T> &q) { m.setIdentity(); m.template topLeftCorner<3, 3>() = q.toRotationMatrix(); }
    template <typename O> explicit Transform(const MatrixBase<O> &o) { m.setIdentity(); if (o.rows() == Dim) m.template topLeftCorner<Dim, Dim>() = o; else m = o; }
    Transform &operator=(const Quaternion<T> &q) { m.setIdentity(); m.template topLeftCorner<3, 3>() = q.toRotationMatrix(); return *this; }
    template <typename O> Transform &operator=(const MatrixBase<O> &o) { if (o.rows() == Dim) { m.setIdentity(); m.template topLeftCorner<Dim, Dim>() = o; } else m = o; return *this; }
    static Transform Identity() { return Transform(); }
    void setIdentity() { m.setIdentity(); }
    Matrix<T, Dim + 1, Dim + 1> &matrix() { return m; }
    const Matrix<T, Dim + 1, Dim + 1> &matrix() const { return m; }
    Block<Matrix<T, Dim + 1, Dim + 1>, Dim, 1> translation() { return Block<Matrix<T, Dim + 1, Dim + 1>, Dim, 1>(m, 0, Dim, Dim, 1); }
    Matrix<T, Dim, 1> translation() const { return m.template block<Dim, 1>(0, Dim); }
    Block<Matrix<T, Dim + 1, Dim + 1>, Dim, Dim> linear() { return Block<Matrix<T, Dim + 1, Dim + 1>, Dim, Dim>(m, 0, 0, Dim, Dim); }
    Matrix<T, Dim, Dim> linear() const { return m.template block<Dim, Dim>(0, 0); }
    Matrix<T, Dim, Dim> rotation() const { return linear(); }
    T operator()(int i, int j) const { return m(i, j); }
    T &operator()(int i, int j) { return m(i, j); }
    Transform operator*(const Transform &o) const { Transform r; r.m = m * o.m; return r; }
    template <typename O> Matrix<T, Dim, 1> operator*(const MatrixBase<O> &v) const { Matrix<T, Dim, 1> r = linear() * v; r += translation(); return r; }
    Transform inverse() const {
        Transform r;
        if (Mode == Isometry) { Matrix<T, Dim, Dim> Rt = linear().transpose(); r.linear() = Rt; r.translation() = -(Rt * translation()); }
        else { Matrix<T, Dim, Dim> Li = linear().inverse(); r.linear() = Li; r.translation() = -(Li * translation()); }
        return r;
    }
    Transform &translate(const Matrix<T, Dim, 1> &t) { translation() = linear() * t + translation(); return *this; }
    Transform &pretranslate(const Matrix<T, Dim, 1> &t) { translation() = translation() + t; return *this; }
    Transform &rotate(const Quaternion<T> &q) { linear() = linear() * q.toRotationMatrix(); return *this; }
    template <typename O> Transform &rotate(const MatrixBase<O> &R) { linear() = linear() * R; return *this; }
    Transform &prerotate(const Quaternion<T> &q) { Matrix<T, Dim, Dim> R = q.toRotationMatrix(); linear() = R * linear(); translation() = R * translation(); return *this; }
};
typedef Transform<double, 3, Isometry> Isometry3d; typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Affine> Affine3d; typedef Transform<double, 2, Affine> Affine2d;
typedef Transform<float, 3, Isometry> Isometry3f; typedef Transform<float, 3, Affine> Affine3f;
} // namespace Eigen
