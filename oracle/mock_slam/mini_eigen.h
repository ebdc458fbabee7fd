/* TEST INFRASTRUCTURE ONLY.  The few Eigen / Sophus names the reference's matcher and camera code spells, as plain float
 * structs: enough for that code to COMPILE against oracle/mock_slam and oracle/mock_frame.  This is not Eigen: operations are
 * naive scalar float arithmetic in the order written here.  The parity tests never rely on its rounding: they drive the
 * geometry with identity poses, or read back the matrix the reference code actually used (last_product()). */
#pragma once
#include <cmath>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {
template <int N>
struct Vec {
    float v[N];
    Vec() { for (int i = 0; i < N; i++) v[i] = 0.f; }
    Vec(float a, float b) { static_assert(N == 2, ""); v[0] = a; v[1] = b; }
    Vec(float a, float b, float c) { static_assert(N == 3, ""); v[0] = a; v[1] = b; v[2] = c; }
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    Vec operator-(const Vec &o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] - o.v[i]; return r; }
    Vec operator+(const Vec &o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] + o.v[i]; return r; }
    Vec operator-() const { Vec r; for (int i = 0; i < N; i++) r.v[i] = -v[i]; return r; }
    Vec operator*(float s) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] * s; return r; }
    Vec operator/(float s) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] / s; return r; }
    float dot(const Vec &o) const { float s = 0.f; for (int i = 0; i < N; i++) s += v[i] * o.v[i]; return s; }
    float squaredNorm() const { return dot(*this); }
    float norm() const { return std::sqrt(dot(*this)); }
    const Vec &transpose() const { return *this; }
    static Vec Zero() { return Vec(); }
    Vec<3> head(int n) const { Vec<3> r; for (int i = 0; i < 3; i++) r.v[i] = v[i]; (void)n; return r; }   /* x3Dh.head(3) */
};
template <int N> inline Vec<N> operator*(float s, const Vec<N> &a) { return a * s; }
typedef Vec<2> Vector2f;
typedef Vec<3> Vector3f;
struct Matrix3f;
inline Matrix3f &last_product(); /* result of the most recent Matrix3f * Matrix3f (how a test learns the F12 the code used) */
struct Matrix3f {
    float m[3][3];
    Matrix3f() { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = (i == j) ? 1.f : 0.f; }
    static Matrix3f Identity() { return Matrix3f(); }
    float &operator()(int i, int j) { return m[i][j]; }
    float operator()(int i, int j) const { return m[i][j]; }
    Vector3f operator*(const Vector3f &x) const {
        Vector3f r;
        for (int i = 0; i < 3; i++) r.v[i] = m[i][0] * x.v[0] + m[i][1] * x.v[1] + m[i][2] * x.v[2];
        return r;
    }
    /* out of line: the caller's arithmetic on the product's entries then starts from one materialised matrix, the same one
     * last_product() reports (inlined, GCC vectorises the product and may round a second scalar copy differently) */
    __attribute__((noinline)) Matrix3f operator*(const Matrix3f &o) const {
        Matrix3f r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r.m[i][j] = m[i][0] * o.m[0][j] + m[i][1] * o.m[1][j] + m[i][2] * o.m[2][j];
        last_product() = r;
        return r;
    }
    Matrix3f inverse() const { /* adjugate / determinant */
        const float (*a)[3] = m;
        const float c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                    c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
        const float det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02, id = 1.0f / det;
        Matrix3f r;
        r.m[0][0] = c00 * id; r.m[1][0] = c01 * id; r.m[2][0] = c02 * id;
        r.m[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
        r.m[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
        r.m[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
        r.m[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
        r.m[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
        r.m[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
        return r;
    }
    Matrix3f operator*(float s) const { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] * s; return r; }
    Matrix3f transpose() const { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i]; return r; }
    Matrix3f operator-() const { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = -m[i][j]; return r; }
    Vector3f row(int i) const { return Vector3f(m[i][0], m[i][1], m[i][2]); }
};
inline Matrix3f &last_product() { static thread_local Matrix3f m; return m; }
}  // namespace Eigen

namespace Sophus {
struct SO3f {
    static Eigen::Matrix3f hat(const Eigen::Vector3f &w) {
        Eigen::Matrix3f r;
        r.m[0][0] = 0.f; r.m[0][1] = -w(2); r.m[0][2] = w(1);
        r.m[1][0] = w(2); r.m[1][1] = 0.f; r.m[1][2] = -w(0);
        r.m[2][0] = -w(1); r.m[2][1] = w(0); r.m[2][2] = 0.f;
        return r;
    }
};
struct SE3f {
    Eigen::Matrix3f R;
    Eigen::Vector3f t;
    SE3f() {}
    SE3f(const Eigen::Matrix3f &R_, const Eigen::Vector3f &t_) : R(R_), t(t_) {}
    Eigen::Matrix3f rotationMatrix() const { return R; }
    Eigen::Vector3f translation() const { return t; }
    SE3f inverse() const { Eigen::Matrix3f Rt = R.transpose(); return SE3f(Rt, -(Rt * t)); }
    Eigen::Vector3f operator*(const Eigen::Vector3f &x) const { return R * x + t; }
    SE3f operator*(const SE3f &o) const { return SE3f(R * o.R, R * o.t + t); }
};
template <class T>
struct Sim3 {
    float s = 1.f;
    Eigen::Matrix3f R;
    Eigen::Vector3f t;
    Sim3() {}
    Sim3(float s_, const Eigen::Matrix3f &R_, const Eigen::Vector3f &t_) : s(s_), R(R_), t(t_) {}
    Eigen::Matrix3f rotationMatrix() const { return R; }
    Eigen::Vector3f translation() const { return t; }
    float scale() const { return s; }
    Sim3 inverse() const { Eigen::Matrix3f Rt = R.transpose(); return Sim3(1.f / s, Rt, -((Rt * t) * (1.f / s))); }
    Eigen::Vector3f operator*(const Eigen::Vector3f &x) const { return (R * x) * s + t; }
};
typedef Sim3<float> Sim3f;
}  // namespace Sophus

