// eigen_syntax.hpp -- TEST INFRASTRUCTURE.  A syntax-level stand-in for the Eigen types and calls that adapters/detect_3d_cuboid_hip.cpp and the
// reference headers it includes (detect_3d_cuboid.h, matrix_utils.h, object_3d_util.h) use, so that the adapter can be type-checked with
// `g++ -fsyntax-only` against the reference's own headers where Eigen is absent (tests/test_adapters.py).  It is never executed.
#pragma once
#include <vector>

namespace Eigen {
const int Dynamic = -1;
template <typename T, int R, int C, int O = 0, int MR = R, int MC = C> class Matrix {
public:
    Matrix() {}
    template <typename A, typename B> Matrix(A, B) {} // (rows, cols) or two coefficients
    explicit Matrix(int) {}
    Matrix(T, T, T) {}
    Matrix(T, T, T, T) {}
    template <int R2, int C2> Matrix(const Matrix<T, R2, C2> &) {}
    T &operator()(int, int) { return v_; }
    const T &operator()(int, int) const { return v_; }
    T &operator()(int) { return v_; }
    const T &operator()(int) const { return v_; }
    T &operator[](int) { return v_; }
    const T &operator[](int) const { return v_; }
    long rows() const { return 0; }
    long cols() const { return 0; }
    void resize(int, int) {}
    void resize(int) {}
    Matrix inverse() const { return *this; }
    Matrix<T, C, R> transpose() const { return Matrix<T, C, R>(); }
    template <int A, int B> Matrix<T, A, B> block(int, int) const { return Matrix<T, A, B>(); }
    template <int A, int B> Matrix<T, A, B> topLeftCorner() const { return Matrix<T, A, B>(); }
    template <int A> Matrix<T, A, C> topRows() const { return Matrix<T, A, C>(); }
    Matrix<T, 1, C> row(int) const { return Matrix<T, 1, C>(); }
    Matrix<T, R, 1> col(int) const { return Matrix<T, R, 1>(); }
    static Matrix Identity() { return Matrix(); }
    static Matrix Zero() { return Matrix(); }
    T norm() const { return T(); }
private:
    T v_ = T();
};
template <typename T, int R, int K, int C> Matrix<T, R, C> operator*(const Matrix<T, R, K> &, const Matrix<T, K, C> &) { return Matrix<T, R, C>(); }
template <typename T, int R, int C> Matrix<T, R, C> operator*(T, const Matrix<T, R, C> &m) { return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator*(const Matrix<T, R, C> &m, T) { return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator+(const Matrix<T, R, C> &m, const Matrix<T, R, C> &) { return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator-(const Matrix<T, R, C> &m, const Matrix<T, R, C> &) { return m; }
template <typename T> class Quaternion {
public:
    Quaternion() {}
    Quaternion(T, T, T, T) {}
    explicit Quaternion(const Matrix<T, 3, 3> &) {}
    T w() const { return T(); } T x() const { return T(); } T y() const { return T(); } T z() const { return T(); }
    Matrix<T, 3, 3> toRotationMatrix() const { return Matrix<T, 3, 3>(); }
};
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf;
typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<int, Dynamic, 1> VectorXi;
typedef Matrix<int, 2, Dynamic> Matrix2Xi;
typedef Matrix<double, 3, Dynamic> Matrix3Xd;
typedef Matrix<double, 2, Dynamic> Matrix2Xd;
typedef Quaternion<double> Quaterniond;
} // namespace Eigen
