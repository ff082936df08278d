// Gauge fix maths shared by the stand-alone kernels (tcv_gauge.hip) and the solve kernel's epilogue (tcv_solve.hip, SolveArgs::gauge_fix):
// Estimator::double2vector() (reference vins_estimator/src/estimator.cpp:1537-1581) followed by what the next vector2double() (:1492-1512)
// makes of its result.
#pragma once
#include "tcv_math.h"

namespace tcv {

#define TCV_PI 3.14159265358979323846   // M_PI

// Utility::R2ypr, utility.h:70-85 (degrees)
TCV_HD void r2ypr(const M3 &R, double ypr[3]) {
    const double n0 = R(0, 0), n1 = R(1, 0), n2 = R(2, 0), o0 = R(0, 1), o1 = R(1, 1), a0 = R(0, 2), a1 = R(1, 2);
    const double y = atan2(n1, n0);
    const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
    const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
    ypr[0] = y / TCV_PI * 180.0; ypr[1] = p / TCV_PI * 180.0; ypr[2] = r / TCV_PI * 180.0;
}
// Utility::ypr2R, utility.h:87-112 (degrees): Rz * Ry * Rx
TCV_HD M3 ypr2R(double yd, double pd, double rd) {
    const double y = yd / 180.0 * TCV_PI, p = pd / 180.0 * TCV_PI, r = rd / 180.0 * TCV_PI;
    M3 Rz = m3_zero(), Ry = m3_zero(), Rx = m3_zero();
    Rz(0, 0) = cos(y); Rz(0, 1) = -sin(y); Rz(1, 0) = sin(y); Rz(1, 1) = cos(y); Rz(2, 2) = 1.0;
    Ry(0, 0) = cos(p); Ry(0, 2) = sin(p); Ry(1, 1) = 1.0; Ry(2, 0) = -sin(p); Ry(2, 2) = cos(p);
    Rx(0, 0) = 1.0; Rx(1, 1) = cos(r); Rx(1, 2) = -sin(r); Rx(2, 1) = sin(r); Rx(2, 2) = cos(r);
    return (Rz * Ry) * Rx;
}
// `Quaterniond q{R}` of vector2double (estimator.cpp:1499): Eigen's rotation-matrix -> quaternion conversion
TCV_HD Quat r2q(const M3 &m) {
    double q[4];   // x y z w
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t; q[1] = (m(0, 2) - m(2, 0)) * t; q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
    return Quat(q[0], q[1], q[2], q[3]);
}
// rot_diff of estimator.cpp:1548-1563
TCV_HD M3 gauge_rot_diff(const M3 &R0, const double *pose0) {
    double a[3], b[3];
    r2ypr(R0, a);
    const M3 R00 = to_matrix(Quat(pose0 + 3));
    r2ypr(R00, b);
    const double y_diff = a[0] - b[0];
    M3 rot = ypr2R(y_diff, 0.0, 0.0);
    if (fabs(fabs(a[1]) - 90.0) < 1.0 || fabs(fabs(b[1]) - 90.0) < 1.0) rot = R0 * transpose(R00);   // "euler singular point"
    return rot;
}
// one frame of the loop :1565-1581
TCV_HD void gauge_frame(const M3 &rot, const double *P0, const double *pose0, const double *pose_i, const double *vel_i, M3 &Rs, V3 &Ps, V3 &Vs) {
    Rs = rot * to_matrix(normalized(Quat(pose_i + 3)));
    Ps = rot * V3(pose_i[0] - pose0[0], pose_i[1] - pose0[1], pose_i[2] - pose0[2]) + V3(P0);
    Vs = vel_i ? rot * V3(vel_i) : V3();
}

}  // namespace tcv
