// TEST INFRASTRUCTURE ONLY -- CPU oracle helpers (see oracle/README.md).
// Tiny dense linear-algebra types standing in for the Eigen types the reference uses.
// Nothing in the product path (raptor_amd/, include/) may include this file.
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace oracle {

struct Vec3 {
    double v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(double a, double b, double c) : v{a, b, c} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    Vec3 operator+(const Vec3& o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3 operator-(const Vec3& o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3 operator-() const { return Vec3(-v[0], -v[1], -v[2]); }
    Vec3 operator*(double s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
    Vec3 operator/(double s) const { return Vec3(v[0] / s, v[1] / s, v[2] / s); }
    double dot(const Vec3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    double norm() const { return std::sqrt(dot(*this)); }
    Vec3 cross(const Vec3& o) const {
        return Vec3(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
};

struct Mat3 {
    double m[3][3];
    Mat3() { std::memset(m, 0, sizeof(m)); }
    static Mat3 Identity() {
        Mat3 r;
        r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
        return r;
    }
    double& operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    Mat3 operator*(const Mat3& o) const {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += m[i][k] * o.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
    Vec3 operator*(const Vec3& o) const {
        Vec3 r;
        for (int i = 0; i < 3; i++) r.v[i] = m[i][0] * o.v[0] + m[i][1] * o.v[1] + m[i][2] * o.v[2];
        return r;
    }
    Mat3 operator*(double s) const {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] * s;
        return r;
    }
    Mat3 operator+(const Mat3& o) const {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j] + o.m[i][j];
        return r;
    }
    Mat3 transpose() const {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
        return r;
    }
};

// Utils::skew (Utils/include/Utils.h:151-157)
inline Mat3 skew(const Vec3& v) {
    Mat3 r;
    r(0, 1) = -v(2);
    r(0, 2) = v(1);
    r(1, 0) = v(2);
    r(1, 2) = -v(0);
    r(2, 0) = -v(1);
    r(2, 1) = v(0);
    return r;
}

// dynamic row-major matrix (Eigen::MatrixXd stand-in)
struct MatX {
    int r = 0, c = 0;
    std::vector<double> a;
    MatX() {}
    MatX(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    static MatX Zero(int r_, int c_) { return MatX(r_, c_); }
    static MatX Identity(int n) {
        MatX m(n, n);
        for (int i = 0; i < n; i++) m(i, i) = 1.0;
        return m;
    }
    int rows() const { return r; }
    int cols() const { return c; }
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    void setZero() { std::fill(a.begin(), a.end(), 0.0); }
    MatX operator*(const MatX& o) const {
        if (c != o.r) throw std::invalid_argument("MatX product: shape mismatch");
        MatX out(r, o.c);
        for (int i = 0; i < r; i++)
            for (int k = 0; k < c; k++) {
                const double x = (*this)(i, k);
                if (x == 0.0) continue;
                for (int j = 0; j < o.c; j++) out(i, j) += x * o(k, j);
            }
        return out;
    }
    MatX operator+(const MatX& o) const {
        MatX out(r, c);
        for (size_t i = 0; i < a.size(); i++) out.a[i] = a[i] + o.a[i];
        return out;
    }
    MatX operator-(const MatX& o) const {
        MatX out(r, c);
        for (size_t i = 0; i < a.size(); i++) out.a[i] = a[i] - o.a[i];
        return out;
    }
    MatX operator*(double s) const {
        MatX out(r, c);
        for (size_t i = 0; i < a.size(); i++) out.a[i] = a[i] * s;
        return out;
    }
    MatX transpose() const {
        MatX out(c, r);
        for (int i = 0; i < r; i++)
            for (int j = 0; j < c; j++) out(j, i) = (*this)(i, j);
        return out;
    }
};

using VecX = std::vector<double>;

struct Vec6 {
    double v[6];
    Vec6() : v{0, 0, 0, 0, 0, 0} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    Vec6 operator+(const Vec6& o) const {
        Vec6 r;
        for (int i = 0; i < 6; i++) r.v[i] = v[i] + o.v[i];
        return r;
    }
    Vec6 operator*(double s) const {
        Vec6 r;
        for (int i = 0; i < 6; i++) r.v[i] = v[i] * s;
        return r;
    }
    Vec6 operator-() const {
        Vec6 r;
        for (int i = 0; i < 6; i++) r.v[i] = -v[i];
        return r;
    }
    double dot(const Vec6& o) const {
        double s = 0;
        for (int i = 0; i < 6; i++) s += v[i] * o.v[i];
        return s;
    }
};

struct Mat6 {
    double m[6][6];
    Mat6() { std::memset(m, 0, sizeof(m)); }
    static Mat6 Identity() {
        Mat6 r;
        for (int i = 0; i < 6; i++) r.m[i][i] = 1.0;
        return r;
    }
    double& operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    Mat6 operator*(const Mat6& o) const {
        Mat6 r;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += m[i][k] * o.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
    Vec6 operator*(const Vec6& o) const {
        Vec6 r;
        for (int i = 0; i < 6; i++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += m[i][k] * o.v[k];
            r.v[i] = s;
        }
        return r;
    }
    Mat6 transpose() const {
        Mat6 r;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) r.m[i][j] = m[j][i];
        return r;
    }
    // 6x6 * (6 x W) with the right operand given as 6 row vectors
    MatX operator*(const MatX& o) const {
        if (o.r != 6) throw std::invalid_argument("Mat6*MatX: need 6 rows");
        MatX out(6, o.c);
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 6; k++) {
                const double x = m[i][k];
                if (x == 0.0) continue;
                for (int j = 0; j < o.c; j++) out(i, j) += x * o(k, j);
            }
        return out;
    }
};

}  // namespace oracle
#endif
