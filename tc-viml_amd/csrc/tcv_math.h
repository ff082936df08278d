// Small fixed-size FP64 vector / quaternion / 3x3 helpers shared by the HIP kernels and their
// CPU-sanitizer build.  Semantics follow the Eigen operations the reference relies on
// (SURVEY.md Appendix A): Hamilton product, inverse = conjugate / squaredNorm, q * v without
// normalisation, toRotationMatrix without normalisation.  Quaternion memory order is x y z w.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TCV_HD __host__ __device__ __forceinline__
#define TCV_D __device__ __forceinline__
#else
#define TCV_HD inline
#define TCV_D inline
#endif

namespace tcv {

struct V3 {
    double x, y, z;
    TCV_HD V3() : x(0), y(0), z(0) {}
    TCV_HD V3(double a, double b, double c) : x(a), y(b), z(c) {}
    TCV_HD explicit V3(const double *p) : x(p[0]), y(p[1]), z(p[2]) {}
    TCV_HD double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
TCV_HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
TCV_HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
TCV_HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
TCV_HD V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
TCV_HD V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
TCV_HD V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
TCV_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TCV_HD V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

struct M3 {  // row-major
    double m[9];
    TCV_HD double &operator()(int r, int c) { return m[3 * r + c]; }
    TCV_HD double operator()(int r, int c) const { return m[3 * r + c]; }
};
TCV_HD M3 m3_zero() { M3 r; for (int i = 0; i < 9; i++) r.m[i] = 0; return r; }
TCV_HD M3 m3_identity() { M3 r = m3_zero(); r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
TCV_HD M3 m3_load(const double *p) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
TCV_HD M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
TCV_HD V3 operator*(const M3 &a, V3 v) {
    return V3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
              a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
TCV_HD M3 operator+(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
TCV_HD M3 operator-(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] - b.m[i]; return r; }
TCV_HD M3 operator-(const M3 &a) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = -a.m[i]; return r; }
TCV_HD M3 operator*(double s, const M3 &a) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = s * a.m[i]; return r; }
TCV_HD M3 transpose(const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * j + i];
    return r;
}
TCV_HD M3 skew(V3 v) {  // utility.h:30-38
    M3 r;
    r.m[0] = 0; r.m[1] = -v.z; r.m[2] = v.y;
    r.m[3] = v.z; r.m[4] = 0; r.m[5] = -v.x;
    r.m[6] = -v.y; r.m[7] = v.x; r.m[8] = 0;
    return r;
}
// v^T * M as a vector
TCV_HD V3 vT_mul(V3 v, const M3 &a) {
    return V3(v.x * a.m[0] + v.y * a.m[3] + v.z * a.m[6], v.x * a.m[1] + v.y * a.m[4] + v.z * a.m[7],
              v.x * a.m[2] + v.y * a.m[5] + v.z * a.m[8]);
}

struct Quat {  // x y z w
    double x, y, z, w;
    TCV_HD Quat() : x(0), y(0), z(0), w(1) {}
    TCV_HD Quat(double X, double Y, double Z, double W) : x(X), y(Y), z(Z), w(W) {}
    TCV_HD explicit Quat(const double *p) : x(p[0]), y(p[1]), z(p[2]), w(p[3]) {}
    TCV_HD V3 vec() const { return V3(x, y, z); }
};
TCV_HD Quat operator*(Quat a, Quat b) {
    return Quat(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
TCV_HD Quat inverse(Quat q) {
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return Quat(-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2);
}
TCV_HD Quat normalized(Quat q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quat(q.x / n, q.y / n, q.z / n, q.w / n);
}
TCV_HD V3 rotate(Quat q, V3 v) {  // Eigen: v + 2w(u x v) + 2 u x (u x v)
    V3 u = q.vec();
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
TCV_HD M3 to_matrix(Quat q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0] = 1 - (tyy + tzz); r.m[1] = txy - twz; r.m[2] = txz + twy;
    r.m[3] = txy + twz; r.m[4] = 1 - (txx + tzz); r.m[5] = tyz - twx;
    r.m[6] = txz - twy; r.m[7] = tyz + twx; r.m[8] = 1 - (txx + tyy);
    return r;
}
TCV_HD Quat delta_q(V3 theta) { return Quat(theta.x / 2, theta.y / 2, theta.z / 2, 1.0); }  // utility.h:15-28
TCV_HD M3 qleft33(Quat q) {  // utility.h:50-58, bottom-right 3x3 (positify is the identity, :40-48)
    M3 r = skew(q.vec());
    r.m[0] += q.w; r.m[4] += q.w; r.m[8] += q.w;
    return r;
}
TCV_HD M3 qright33(Quat p) {  // utility.h:60-68
    M3 r = -skew(p.vec());
    r.m[0] += p.w; r.m[4] += p.w; r.m[8] += p.w;
    return r;
}
// PoseLocalParameterization::Plus  pose_local_parameterization.cpp:3-19
TCV_HD void pose_plus(const double *x, const double *d, double *out) {
    out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
    Quat q = normalized(Quat(x + 3) * delta_q(V3(d + 3)));
    out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

}  // namespace tcv
