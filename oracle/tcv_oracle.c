/* CPU oracle (plain C99) -- see tcv_oracle.h.  TEST INFRASTRUCTURE ONLY, parity unpinned.
 * Compile with -ffp-contract=off (the sqrt_info routine is mirrored operation-for-operation
 * by the HIP kernel).  Paths cited are relative to /root/reference/vins_estimator/src/. */
#include "tcv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 }; /* parameters.h:66-73 */

/* ------------------------------------------------------------------------------------------
 * utility/utility.h:15-68 + the Eigen behaviours of SURVEY.md Appendix A.  q = [x y z w].
 * ---------------------------------------------------------------------------------------- */
static void q_mul(const double *a, const double *b, double *o) {
    double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
static void q_inv(const double *q, double *o) { /* Eigen inverse(): conjugate / squaredNorm */
    double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2;
}
static void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void q_rot(const double *q, const double *v, double *o) { /* Eigen _transformVector */
    double uv[3], c[3];
    cross3(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(q, uv, c);
    o[0] = v[0] + q[3] * uv[0] + c[0]; o[1] = v[1] + q[3] * uv[1] + c[1]; o[2] = v[2] + q[3] * uv[2] + c[2];
}
static void q_normalized(const double *q, double *o) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n;
}
static void q_toR(const double *q, double *R) { /* Eigen toRotationMatrix, row-major 3x3 */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void delta_q(const double *th, double *o) { o[0] = th[0] / 2; o[1] = th[1] / 2; o[2] = th[2] / 2; o[3] = 1.0; }
static void skew3(const double *v, double *S) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void m33_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void m33_T(const double *A, double *T) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
}
static void m33_v(const double *A, const double *v, double *o) {
    for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void qleft33(const double *q, double *M) { /* utility.h:50-58 bottomRightCorner<3,3> */
    double S[9]; skew3(q, S);
    for (int i = 0; i < 9; i++) M[i] = S[i];
    M[0] += q[3]; M[4] += q[3]; M[8] += q[3];
}
static void qright33(const double *p, double *M) { /* utility.h:60-68 */
    double S[9]; skew3(p, S);
    for (int i = 0; i < 9; i++) M[i] = -S[i];
    M[0] += p[3]; M[4] += p[3]; M[8] += p[3];
}

void orc_pose_plus(const double *x, const double *d, double *out) { /* pose_local_parameterization.cpp:3-19 */
    double dq[4], q[4];
    out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
    delta_q(d + 3, dq);
    q_mul(x + 3, dq, q);
    q_normalized(q, out + 3);
}

/* ------------------------------------------------------------------------------------------
 * I0  factor/integration_base.h:13-158
 * ---------------------------------------------------------------------------------------- */
static void mat_mul(int n, int k, int m, const double *A, const double *B, double *C) { /* C(n x m) = A(n x k) B(k x m) */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
static void set33(double *M, int ld, int r0, int c0, const double *B, double s) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = s * B[3 * i + j];
}
void orc_preintegrate(const double *acc, const double *gyr, int n_samples, double dt, const double *lin_ba,
                      const double *lin_bg, double acc_n, double gyr_n, double acc_w, double gyr_w, double *out,
                      double *jac_out) {
    double noise[18]; /* diagonal of the 18x18 noise matrix :21-27 */
    for (int i = 0; i < 3; i++) {
        noise[i] = acc_n * acc_n; noise[3 + i] = gyr_n * gyr_n; noise[6 + i] = acc_n * acc_n;
        noise[9 + i] = gyr_n * gyr_n; noise[12 + i] = acc_w * acc_w; noise[15 + i] = gyr_w * gyr_w;
    }
    double dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1};
    double jac[225], cov[225], F[225], V[270], T1[225], T2[270], T3[225];
    memset(jac, 0, sizeof jac); memset(cov, 0, sizeof cov);
    for (int i = 0; i < 15; i++) jac[16 * i] = 1.0;
    double sum_dt = 0;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k + 1 < n_samples; k++) {
        const double *a0 = acc + 3 * k, *g0 = gyr + 3 * k, *a1 = acc + 3 * k + 3, *g1 = gyr + 3 * k + 3;
        double a0b[3], a1b[3], ung[3], una0[3], una1[3], una[3], rq[4], st[4], rp[3], rv[3];
        for (int i = 0; i < 3; i++) { a0b[i] = a0[i] - lin_ba[i]; a1b[i] = a1[i] - lin_ba[i]; ung[i] = 0.5 * (g0[i] + g1[i]) - lin_bg[i]; }
        q_rot(dq, a0b, una0);
        st[0] = ung[0] * dt / 2; st[1] = ung[1] * dt / 2; st[2] = ung[2] * dt / 2; st[3] = 1.0;
        q_mul(dq, st, rq);
        q_rot(rq, a1b, una1);
        for (int i = 0; i < 3; i++) {
            una[i] = 0.5 * (una0[i] + una1[i]);
            rp[i] = dp[i] + dv[i] * dt + 0.5 * una[i] * dt * dt;
            rv[i] = dv[i] + una[i] * dt;
        }
        double Rw[9], Ra0[9], Ra1[9], R0[9], R1[9], IRw[9], A[9], Bm[9], C[9];
        skew3(ung, Rw); skew3(a0b, Ra0); skew3(a1b, Ra1); q_toR(dq, R0); q_toR(rq, R1);
        for (int i = 0; i < 9; i++) IRw[i] = I3[i] - Rw[i] * dt;
        memset(F, 0, sizeof F); memset(V, 0, sizeof V);
        m33_mul(R0, Ra0, A);          /* R0 R_a_0_x */
        m33_mul(R1, Ra1, Bm);         /* R1 R_a_1_x */
        m33_mul(Bm, IRw, C);          /* R1 R_a_1_x (I - R_w_x dt) */
        double R01[9];
        for (int i = 0; i < 9; i++) R01[i] = R0[i] + R1[i];
        set33(F, 15, 0, 0, I3, 1.0);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.25 * A[i] * dt * dt + -0.25 * C[i] * dt * dt; set33(F, 15, 0, 3, t, 1.0); }
        set33(F, 15, 0, 6, I3, dt);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.25 * R01[i] * dt * dt; set33(F, 15, 0, 9, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.25 * Bm[i] * dt * dt * -dt; set33(F, 15, 0, 12, t, 1.0); }
        set33(F, 15, 3, 3, IRw, 1.0);
        set33(F, 15, 3, 12, I3, -1.0 * dt);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.5 * A[i] * dt + -0.5 * C[i] * dt; set33(F, 15, 6, 3, t, 1.0); }
        set33(F, 15, 6, 6, I3, 1.0);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.5 * R01[i] * dt; set33(F, 15, 6, 9, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = -0.5 * Bm[i] * dt * -dt; set33(F, 15, 6, 12, t, 1.0); }
        set33(F, 15, 9, 9, I3, 1.0);
        set33(F, 15, 12, 12, I3, 1.0);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.25 * R0[i] * dt * dt; set33(V, 18, 0, 0, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.25 * -Bm[i] * dt * dt * 0.5 * dt; set33(V, 18, 0, 3, t, 1.0); set33(V, 18, 0, 9, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.25 * R1[i] * dt * dt; set33(V, 18, 0, 6, t, 1.0); }
        set33(V, 18, 3, 3, I3, 0.5 * dt);
        set33(V, 18, 3, 9, I3, 0.5 * dt);
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.5 * R0[i] * dt; set33(V, 18, 6, 0, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.5 * -Bm[i] * dt * 0.5 * dt; set33(V, 18, 6, 3, t, 1.0); set33(V, 18, 6, 9, t, 1.0); }
        { double t[9]; for (int i = 0; i < 9; i++) t[i] = 0.5 * R1[i] * dt; set33(V, 18, 6, 6, t, 1.0); }
        set33(V, 18, 9, 12, I3, dt);
        set33(V, 18, 12, 15, I3, dt);
        /* jacobian = F * jacobian ; covariance = F cov F' + V noise V' (:124-125) */
        mat_mul(15, 15, 15, F, jac, T1); memcpy(jac, T1, sizeof jac);
        mat_mul(15, 15, 15, F, cov, T1);
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) {
                double s = 0;
                for (int l = 0; l < 15; l++) s += T1[i * 15 + l] * F[j * 15 + l];
                T3[i * 15 + j] = s;
            }
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 18; j++) T2[i * 18 + j] = V[i * 18 + j] * noise[j];
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) {
                double s = 0;
                for (int l = 0; l < 18; l++) s += T2[i * 18 + l] * V[j * 18 + l];
                cov[i * 15 + j] = T3[i * 15 + j] + s;
            }
        for (int i = 0; i < 3; i++) { dp[i] = rp[i]; dv[i] = rv[i]; }
        q_normalized(rq, dq);
        sum_dt += dt;
    }
    double *o = out;
    memcpy(o, dp, 24); memcpy(o + 3, dq, 32); memcpy(o + 7, dv, 24); memcpy(o + 10, lin_ba, 24); memcpy(o + 13, lin_bg, 24);
    o[16] = sum_dt;
    const int br[5] = {O_P, O_P, O_R, O_V, O_V}, bc[5] = {O_BA, O_BG, O_BG, O_BA, O_BG};
    for (int b = 0; b < 5; b++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) o[17 + 9 * b + 3 * i + j] = jac[(br[b] + i) * 15 + bc[b] + j];
    memcpy(o + 62, cov, sizeof cov);
    if (jac_out) memcpy(jac_out, jac, sizeof jac);
}

/* ------------------------------------------------------------------------------------------
 * imu_factor.h:64  sqrt_info = LLT(cov.inverse()).matrixL().transpose()
 * inverse = partial-pivot LU + triangular solves of the permuted identity; LLT reads the lower
 * triangle.  The HIP kernel mirrors this operation order exactly (no FMA contraction).
 * ---------------------------------------------------------------------------------------- */
int orc_imu_sqrt_info(const double *cov, double *S) {
    double a[15][15], inv[15][15], l[15][15];
    int perm[15];
    for (int i = 0; i < 15; i++) { perm[i] = i; for (int j = 0; j < 15; j++) a[i][j] = cov[i * 15 + j]; }
    for (int k = 0; k < 15; k++) {
        int p = k; double mv = fabs(a[k][k]);
        for (int i = k + 1; i < 15; i++) if (fabs(a[i][k]) > mv) { mv = fabs(a[i][k]); p = i; }
        if (p != k) {
            for (int j = 0; j < 15; j++) { double t = a[k][j]; a[k][j] = a[p][j]; a[p][j] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        for (int i = k + 1; i < 15; i++) a[i][k] = a[i][k] / a[k][k];
        for (int i = k + 1; i < 15; i++)
            for (int j = k + 1; j < 15; j++) a[i][j] = a[i][j] - a[i][k] * a[k][j];
    }
    for (int c = 0; c < 15; c++) {
        double y[15], x[15];
        for (int i = 0; i < 15; i++) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s = s - a[i][k] * y[k];
            y[i] = s;
        }
        for (int i = 14; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 15; k++) s = s - a[i][k] * x[k];
            x[i] = s / a[i][i];
        }
        for (int i = 0; i < 15; i++) inv[i][c] = x[i];
    }
    for (int i = 0; i < 15; i++)
        for (int j = 0; j <= i; j++) {
            double s = inv[i][j];
            for (int k = 0; k < j; k++) s = s - l[i][k] * l[j][k];
            if (i == j) { if (!(s > 0)) return -1; l[i][i] = sqrt(s); }
            else l[i][j] = s / l[j][j];
        }
    for (int r = 0; r < 15; r++)
        for (int c = 0; c < 15; c++) S[r * 15 + c] = (c >= r) ? l[c][r] : 0.0;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * I1  factor/imu_factor.h:19-181 + integration_base.h:160-186
 * ---------------------------------------------------------------------------------------- */
static void left_mul_sqrt(const double *S, int cols, double *J) { /* J(15 x cols) = S * J */
    double T[15 * 9];
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < cols; j++) {
            double s = 0;
            for (int k = 0; k < 15; k++) s += S[i * 15 + k] * J[k * cols + j];
            T[i * cols + j] = s;
        }
    memcpy(J, T, sizeof(double) * 15 * cols);
}
static void blk33(double *J, int ld, int r0, int c0, const double *B, double s) { set33(J, ld, r0, c0, B, s); }

void orc_imu_evaluate(const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                      const double *c, const double *G, const double *sqrt_info, double *r, double **jac) {
    const double *Pi = pose_i, *Qi = pose_i + 3, *Vi = sb_i, *Bai = sb_i + 3, *Bgi = sb_i + 6;
    const double *Pj = pose_j, *Qj = pose_j + 3, *Vj = sb_j, *Baj = sb_j + 3, *Bgj = sb_j + 6;
    const double *dp0 = c, *dq0 = c + 3, *dv0 = c + 7, *lba = c + 10, *lbg = c + 13;
    double sum_dt = c[16];
    const double *dp_dba = c + 17, *dp_dbg = c + 26, *dq_dbg = c + 35, *dv_dba = c + 44, *dv_dbg = c + 53;
    double S_local[225];
    if (!sqrt_info) { orc_imu_sqrt_info(c + 62, S_local); sqrt_info = S_local; }
    double dba[3], dbg[3], th[3], dqc[4], cdq[4], cdv[3], cdp[3], t1[3], t2[3];
    for (int i = 0; i < 3; i++) { dba[i] = Bai[i] - lba[i]; dbg[i] = Bgi[i] - lbg[i]; }
    m33_v(dq_dbg, dbg, th); delta_q(th, dqc); q_mul(dq0, dqc, cdq);          /* integration_base.h:176 */
    m33_v(dv_dba, dba, t1); m33_v(dv_dbg, dbg, t2);
    for (int i = 0; i < 3; i++) cdv[i] = dv0[i] + t1[i] + t2[i];
    m33_v(dp_dba, dba, t1); m33_v(dp_dbg, dbg, t2);
    for (int i = 0; i < 3; i++) cdp[i] = dp0[i] + t1[i] + t2[i];
    double Qi_inv[4], vp[3], vv[3], rp[3], rvv[3], qij[4], cdq_inv[4], qe[4], raw[15];
    q_inv(Qi, Qi_inv);
    for (int i = 0; i < 3; i++) {
        vp[i] = 0.5 * G[i] * sum_dt * sum_dt + Pj[i] - Pi[i] - Vi[i] * sum_dt;
        vv[i] = G[i] * sum_dt + Vj[i] - Vi[i];
    }
    q_rot(Qi_inv, vp, rp); q_rot(Qi_inv, vv, rvv);
    q_mul(Qi_inv, Qj, qij); q_inv(cdq, cdq_inv); q_mul(cdq_inv, qij, qe);
    for (int i = 0; i < 3; i++) {
        raw[O_P + i] = rp[i] - cdp[i];
        raw[O_R + i] = 2 * qe[i];
        raw[O_V + i] = rvv[i] - cdv[i];
        raw[O_BA + i] = Baj[i] - Bai[i];
        raw[O_BG + i] = Bgj[i] - Bgi[i];
    }
    for (int i = 0; i < 15; i++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += sqrt_info[i * 15 + k] * raw[k];
        r[i] = s;
    }
    if (!jac) return;
    double Ri_inv[9], M[9], Sk[9];
    q_toR(Qi_inv, Ri_inv);
    if (jac[0]) { /* :88-113 */
        double *J = jac[0]; memset(J, 0, sizeof(double) * 105);
        blk33(J, 7, O_P, O_P, Ri_inv, -1.0);
        skew3(rp, Sk); blk33(J, 7, O_P, O_R, Sk, 1.0);
        /* -(Qleft(Qj^-1 Qi) Qright(cdq)).bottomRightCorner<3,3>() : corner of the 4x4 product */
        double Qj_inv[4], ql[4], L[9], Rr[9];
        q_inv(Qj, Qj_inv); q_mul(Qj_inv, Qi, ql);
        qleft33(ql, L); qright33(cdq, Rr); m33_mul(L, Rr, M);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M[3 * i + j] += ql[i] * (-cdq[j]);
        blk33(J, 7, O_R, O_R, M, -1.0);
        skew3(rvv, Sk); blk33(J, 7, O_V, O_R, Sk, 1.0);
        left_mul_sqrt(sqrt_info, 7, J);
    }
    if (jac[1]) { /* :114-142 */
        double *J = jac[1]; memset(J, 0, sizeof(double) * 135);
        blk33(J, 9, O_P, 0, Ri_inv, -sum_dt);
        blk33(J, 9, O_P, 3, dp_dba, -1.0);
        blk33(J, 9, O_P, 6, dp_dbg, -1.0);
        double Qj_inv[4], q1[4], q2[4], L[9];
        q_inv(Qj, Qj_inv); q_mul(Qj_inv, Qi, q1); q_mul(q1, dq0, q2);     /* :127 un-corrected delta_q */
        qleft33(q2, L); m33_mul(L, dq_dbg, M);
        blk33(J, 9, O_R, 6, M, -1.0);
        blk33(J, 9, O_V, 0, Ri_inv, -1.0);
        blk33(J, 9, O_V, 3, dv_dba, -1.0);
        blk33(J, 9, O_V, 6, dv_dbg, -1.0);
        for (int i = 0; i < 3; i++) { J[(O_BA + i) * 9 + 3 + i] = -1.0; J[(O_BG + i) * 9 + 6 + i] = -1.0; }
        left_mul_sqrt(sqrt_info, 9, J);
    }
    if (jac[2]) { /* :143-161 */
        double *J = jac[2]; memset(J, 0, sizeof(double) * 105);
        blk33(J, 7, O_P, O_P, Ri_inv, 1.0);
        double q1[4], q2[4], L[9];
        q_mul(cdq_inv, Qi_inv, q1); q_mul(q1, Qj, q2);
        qleft33(q2, L);
        blk33(J, 7, O_R, O_R, L, 1.0);
        left_mul_sqrt(sqrt_info, 7, J);
    }
    if (jac[3]) { /* :162-177 */
        double *J = jac[3]; memset(J, 0, sizeof(double) * 135);
        blk33(J, 9, O_V, 0, Ri_inv, 1.0);
        for (int i = 0; i < 3; i++) { J[(O_BA + i) * 9 + 3 + i] = 1.0; J[(O_BG + i) * 9 + 6 + i] = 1.0; }
        left_mul_sqrt(sqrt_info, 9, J);
    }
}

/* ------------------------------------------------------------------------------------------
 * P1  factor/projection_factor.cpp:21-124
 * ---------------------------------------------------------------------------------------- */
void orc_proj_evaluate(const double *pose_i, const double *pose_j, const double *ex, double lam,
                       const double *pts_i, const double *pts_j, double si, double *r, double **jac) {
    const double *Pi = pose_i, *Qi = pose_i + 3, *Pj = pose_j, *Qj = pose_j + 3, *tic = ex, *qic = ex + 3;
    double pci[3], pii[3], pw[3], pij[3], pcj[3], t[3], Qj_inv[4], qic_inv[4];
    for (int i = 0; i < 3; i++) pci[i] = pts_i[i] / lam;
    q_rot(qic, pci, t); for (int i = 0; i < 3; i++) pii[i] = t[i] + tic[i];
    q_rot(Qi, pii, t); for (int i = 0; i < 3; i++) pw[i] = t[i] + Pi[i];
    q_inv(Qj, Qj_inv); for (int i = 0; i < 3; i++) t[i] = pw[i] - Pj[i];
    q_rot(Qj_inv, t, pij);
    q_inv(qic, qic_inv); for (int i = 0; i < 3; i++) t[i] = pij[i] - tic[i];
    q_rot(qic_inv, t, pcj);
    double dep_j = pcj[2];
    r[0] = si * (pcj[0] / dep_j - pts_j[0]);
    r[1] = si * (pcj[1] / dep_j - pts_j[1]);
    if (!jac) return;
    double Ri[9], Rj[9], ric[9], ricT[9], RjT[9], red[6];
    q_toR(Qi, Ri); q_toR(Qj, Rj); q_toR(qic, ric); m33_T(ric, ricT); m33_T(Rj, RjT);
    red[0] = si * (1. / dep_j); red[1] = 0; red[2] = si * (-pcj[0] / (dep_j * dep_j));
    red[3] = 0; red[4] = si * (1. / dep_j); red[5] = si * (-pcj[1] / (dep_j * dep_j));
    double A[9], Bm[9], C[9], Sk[9], jaco[18];
    m33_mul(ricT, RjT, A);                  /* ric' Rj' */
    if (jac[0]) {
        m33_mul(A, Ri, Bm); skew3(pii, Sk); m33_mul(Bm, Sk, C);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { jaco[6 * i + j] = A[3 * i + j]; jaco[6 * i + 3 + j] = -C[3 * i + j]; }
        for (int i = 0; i < 2; i++) { for (int j = 0; j < 6; j++) jac[0][7 * i + j] = red[3 * i] * jaco[j] + red[3 * i + 1] * jaco[6 + j] + red[3 * i + 2] * jaco[12 + j]; jac[0][7 * i + 6] = 0; }
    }
    if (jac[1]) {
        skew3(pij, Sk); m33_mul(ricT, Sk, C);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { jaco[6 * i + j] = -A[3 * i + j]; jaco[6 * i + 3 + j] = C[3 * i + j]; }
        for (int i = 0; i < 2; i++) { for (int j = 0; j < 6; j++) jac[1][7 * i + j] = red[3 * i] * jaco[j] + red[3 * i + 1] * jaco[6 + j] + red[3 * i + 2] * jaco[12 + j]; jac[1][7 * i + 6] = 0; }
    }
    double tmp_r[9];
    { double T[9]; m33_mul(A, Ri, T); m33_mul(T, ric, tmp_r); }
    if (jac[2]) {
        double RjTRi[9], D[9], left[9], v1[3], v2[3], v3[3], S1[9], S2[9], S3[9], T[9];
        m33_mul(RjT, Ri, RjTRi);
        for (int i = 0; i < 9; i++) D[i] = RjTRi[i];
        D[0] -= 1; D[4] -= 1; D[8] -= 1;
        m33_mul(ricT, D, left);
        skew3(pci, S1); m33_mul(tmp_r, S1, T);
        m33_v(tmp_r, pci, v1); skew3(v1, S2);
        m33_v(Ri, tic, v2); for (int i = 0; i < 3; i++) v2[i] = v2[i] + Pi[i] - Pj[i];
        m33_v(RjT, v2, v3); for (int i = 0; i < 3; i++) v3[i] -= tic[i];
        m33_v(ricT, v3, v1); skew3(v1, S3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { jaco[6 * i + j] = left[3 * i + j]; jaco[6 * i + 3 + j] = -T[3 * i + j] + S2[3 * i + j] + S3[3 * i + j]; }
        for (int i = 0; i < 2; i++) { for (int j = 0; j < 6; j++) jac[2][7 * i + j] = red[3 * i] * jaco[j] + red[3 * i + 1] * jaco[6 + j] + red[3 * i + 2] * jaco[12 + j]; jac[2][7 * i + 6] = 0; }
    }
    if (jac[3]) {
        double v[3];
        m33_v(tmp_r, pts_i, v);
        for (int i = 0; i < 2; i++) jac[3][i] = (red[3 * i] * v[0] + red[3 * i + 1] * v[1] + red[3 * i + 2] * v[2]) * -1.0 / (lam * lam);
    }
}

/* ------------------------------------------------------------------------------------------
 * T1  factor/projection_td_factor.cpp:34-140: P1 on time-offset / rolling-shutter corrected observations + d/dtd
 * ---------------------------------------------------------------------------------------- */
void orc_proj_td_evaluate(const double *pose_i, const double *pose_j, const double *ex, double lam, double td,
                          const double *pts_i, const double *pts_j, const double *aux, double si, double TR, double ROW,
                          double *r, double **jac) {
    const double vi[3] = {aux[0], aux[1], 0.0}, vj[3] = {aux[2], aux[3], 0.0};       /* :11-16 */
    const double row_i = aux[6] - ROW / 2, row_j = aux[7] - ROW / 2;                   /* :17-18 */
    double pi_td[3], pj_td[3];
    for (int i = 0; i < 3; i++) {                                                      /* :50-51 */
        pi_td[i] = pts_i[i] - (td - aux[4] + TR / ROW * row_i) * vi[i];
        pj_td[i] = pts_j[i] - (td - aux[5] + TR / ROW * row_j) * vj[i];
    }
    orc_proj_evaluate(pose_i, pose_j, ex, lam, pi_td, pj_td, si, r, jac);              /* :52-130 = ProjectionFactor on the corrected points */
    if (!jac || !jac[4]) return;
    const double *Pi = pose_i, *Qi = pose_i + 3, *Pj = pose_j, *Qj = pose_j + 3, *tic = ex, *qic = ex + 3;
    double pci[3], pii[3], pw[3], pij[3], pcj[3], t[3], Qj_inv[4], qic_inv[4];
    for (int i = 0; i < 3; i++) pci[i] = pi_td[i] / lam;
    q_rot(qic, pci, t); for (int i = 0; i < 3; i++) pii[i] = t[i] + tic[i];
    q_rot(Qi, pii, t); for (int i = 0; i < 3; i++) pw[i] = t[i] + Pi[i];
    q_inv(Qj, Qj_inv); for (int i = 0; i < 3; i++) t[i] = pw[i] - Pj[i];
    q_rot(Qj_inv, t, pij);
    q_inv(qic, qic_inv); for (int i = 0; i < 3; i++) t[i] = pij[i] - tic[i];
    q_rot(qic_inv, t, pcj);
    const double dep_j = pcj[2];
    const double red[6] = {si * (1. / dep_j), 0, si * (-pcj[0] / (dep_j * dep_j)), 0, si * (1. / dep_j), si * (-pcj[1] / (dep_j * dep_j))};
    double Ri[9], Rj[9], ric[9], ricT[9], RjT[9], A[9], T2[9], tmp_r[9], v[3];
    q_toR(Qi, Ri); q_toR(Qj, Rj); q_toR(qic, ric); m33_T(ric, ricT); m33_T(Rj, RjT);
    m33_mul(ricT, RjT, A); m33_mul(A, Ri, T2); m33_mul(T2, ric, tmp_r);
    m33_v(tmp_r, vi, v);
    for (int i = 0; i < 2; i++)                                                        /* :131-136 */
        jac[4][i] = (red[3 * i] * v[0] + red[3 * i + 1] * v[1] + red[3 * i + 2] * v[2]) / lam * -1.0 + si * vj[i];
}

/* ------------------------------------------------------------------------------------------
 * L1  factor/line_projection_factor.cpp:19-120  (Jacobian copied as written)
 * ---------------------------------------------------------------------------------------- */
void orc_line_evaluate(const double *pose, const double *lc, const double *K, const double *bcR, const double *bcT,
                       double *r, double *jac) {
    double qn[4], Rw[9], RwT[9], bcRT[9], R[9], t[3], t1[3], t2[3], pcs[3], pce[3], si[3], ei[3];
    q_normalized(pose + 3, qn); q_toR(qn, Rw); m33_T(Rw, RwT); m33_T(bcR, bcRT);
    m33_mul(bcRT, RwT, R);
    m33_v(R, pose, t1); m33_v(bcRT, bcT, t2);
    for (int i = 0; i < 3; i++) t[i] = -t1[i] - t2[i];
    m33_v(R, lc, pcs); m33_v(R, lc + 3, pce);
    for (int i = 0; i < 3; i++) { pcs[i] += t[i]; pce[i] += t[i]; }
    m33_v(K, pcs, si); m33_v(K, pce, ei);
    double us = si[0] / si[2], vs = si[1] / si[2], ue = ei[0] / ei[2], ve = ei[1] / ei[2];
    double a = lc[6], b = lc[7], c = lc[8], d = a * a + b * b;
    double mus = (b * b * us - a * b * vs - a * c) / d, mvs = (a * a * vs - a * b * us - b * c) / d;
    double mue = (b * b * ue - a * b * ve - a * c) / d, mve = (a * a * ve - a * b * ue - b * c) / d;
    r[0] = 1.0 * sqrt((mus - us) * (mus - us) + (mvs - vs) * (mvs - vs));
    r[1] = 1.0 * sqrt((mue - ue) * (mue - ue) + (mve - ve) * (mve - ve));
    if (!jac) return;
    double lambda = 1;
    double ep[2] = {-2 / d * ((mus - us) * a * a + a * b * (mvs - vs)) * lambda, -2 / d * ((mus - us) * a * b + b * b * (mvs - vs)) * lambda};
    double ee[2] = {-2 / d * ((mue - ue) * a * a + a * b * (mve - ve)) * lambda, -2 / d * ((mue - ue) * a * b + b * b * (mve - ve)) * lambda};
    double fx = K[0], fy = K[4];
    for (int e = 0; e < 2; e++) {
        const double *p = e ? pce : pcs; const double *w = e ? ee : ep;
        double pp[6] = {fx / p[2], 0, -fx * p[0] / (p[2] * p[2]), 0, fy / p[2], -fy * p[1] / (p[2] * p[2])};
        double ew[3] = {w[0] * pp[0] + w[1] * pp[3], w[0] * pp[1] + w[1] * pp[4], w[0] * pp[2] + w[1] * pp[5]};
        double Sk[9]; skew3(p, Sk);
        for (int j = 0; j < 3; j++) {
            jac[7 * e + j] = ew[j];
            jac[7 * e + 3 + j] = ew[0] * Sk[j] + ew[1] * Sk[3 + j] + ew[2] * Sk[6 + j];
        }
        jac[7 * e + 6] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * C1  marginalization_factor.cpp:37-68 ; ceres::CauchyLoss::Evaluate per upstream Ceres
 * ---------------------------------------------------------------------------------------- */
double orc_loss_correct(int nr, double *r, int nblk, double **Js, const int *cols, double loss_a) {
    double sq = 0;
    for (int i = 0; i < nr; i++) sq += r[i] * r[i];
    if (!(loss_a > 0)) return 0.5 * sq;
    double b = loss_a * loss_a, c = 1.0 / b, sum = 1.0 + sq * c, inv = 1.0 / sum, rho[3];
    rho[0] = b * log(sum); rho[1] = inv > DBL_MIN ? inv : DBL_MIN; rho[2] = -c * (inv * inv);
    double sqrt_rho1 = sqrt(rho[1]), residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else {
        double D = 1.0 + 2.0 * sq * rho[2] / rho[1], alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq;
    }
    for (int k = 0; k < nblk; k++) {
        if (!Js || !Js[k]) continue;
        int nc = cols[k];
        for (int j = 0; j < nc; j++) {
            double rtJ = 0;
            for (int i = 0; i < nr; i++) rtJ += r[i] * Js[k][i * nc + j];
            for (int i = 0; i < nr; i++) Js[k][i * nc + j] = sqrt_rho1 * (Js[k][i * nc + j] - alpha_sq_norm * r[i] * rtJ);
        }
    }
    for (int i = 0; i < nr; i++) r[i] *= residual_scaling;
    return 0.5 * rho[0];
}

/* ------------------------------------------------------------------------------------------
 * M0  marginalization_factor.cpp:335-384
 * ---------------------------------------------------------------------------------------- */
static const double *blk_ptr(const orc_window *w, int kind, int index) {
    if (kind == ORC_BLK_POSE) return w->pose + 7 * index;
    if (kind == ORC_BLK_SB) return w->speedbias + 9 * index;
    return w->ex_pose;
}
void orc_prior_residual(const orc_window *w, double *r) {
    int n = w->prior_n;
    double *dx = (double *)calloc(n > 0 ? n : 1, sizeof(double));
    const double *x0 = w->prior_x0;
    for (int k = 0; k < w->prior_nblk; k++) {
        int size = w->prior_size[k], idx = w->prior_idx[k];
        const double *x = blk_ptr(w, w->prior_kind[k], w->prior_index[k]);
        if (size != 7) { for (int i = 0; i < size; i++) dx[idx + i] = x[i] - x0[i]; }
        else {
            for (int i = 0; i < 3; i++) dx[idx + i] = x[i] - x0[i];
            double qi[4], dq[4];
            q_inv(x0 + 3, qi); q_mul(qi, x + 3, dq);
            for (int i = 0; i < 3; i++) dx[idx + 3 + i] = 2.0 * dq[i];
            if (!(dq[3] >= 0)) for (int i = 0; i < 3; i++) dx[idx + 3 + i] = 2.0 * -dq[i];
        }
        x0 += size;
    }
    for (int i = 0; i < n; i++) {
        double s = w->prior_r0[i];
        for (int j = 0; j < n; j++) s += w->prior_J0[j * n + i] * dx[j];
        r[i] = s;
    }
    free(dx);
}

/* ------------------------------------------------------------------------------------------
 * Problem structure G1 (estimator.cpp:1679-1886): camera-side local offsets pose_i -> 15 i,
 * sb_i -> 15 i + 6, ex -> 15 F (absent if constant); landmark l -> nc + l.
 * ---------------------------------------------------------------------------------------- */
#define MAXB 32
typedef struct {
    int nr, nb;
    int lo[MAXB], ls[MAXB];
    double *J[MAXB]; /* nr x ls row-major, local, loss-corrected, later column-scaled */
    double *r;
    double cost;
} fac_t;

typedef struct {
    int nfac, nrow, nc, nl, F, L;
    fac_t *f;
    double *pool;
} lin_t;

static int loff_of(const orc_window *w, int kind, int index) {
    if (kind == ORC_BLK_POSE) return 15 * index;
    if (kind == ORC_BLK_SB) return 15 * index + 6;
    return w->ex_constant ? -1 : 15 * w->n_frames;
}

static void lin_alloc(const orc_window *w, lin_t *L) {
    int F = w->n_frames;
    L->F = F; L->L = w->n_landmarks;
    L->nc = 15 * F + (w->ex_constant ? 0 : 6);
    L->nl = L->nc + w->n_landmarks;
    int nimu = 0;
    for (int k = 0; k < w->n_imu; k++) if (!(w->imu_c[k * ORC_IMU_STRIDE + 16] > 10.0)) nimu++; /* estimator.cpp:1726 */
    L->nfac = (w->prior_n > 0) + nimu + w->n_proj + w->n_line;
    L->f = (fac_t *)calloc(L->nfac, sizeof(fac_t));
    size_t need = 0;
    if (w->prior_n > 0) need += (size_t)w->prior_n * (w->prior_n + 1);
    need += (size_t)nimu * (15 * 30 + 15) + (size_t)w->n_proj * (2 * 19 + 2) + (size_t)w->n_line * (2 * 6 + 2);
    L->pool = (double *)calloc(need, sizeof(double));
    double *p = L->pool;
    int fi = 0, nrow = 0;
    if (w->prior_n > 0) {
        fac_t *f = &L->f[fi++];
        f->nr = w->prior_n; f->nb = 0;
        for (int k = 0; k < w->prior_nblk; k++) {
            int lo = loff_of(w, w->prior_kind[k], w->prior_index[k]);
            int ls = w->prior_size[k] == 7 ? 6 : w->prior_size[k];
            if (lo < 0) continue;
            f->lo[f->nb] = lo; f->ls[f->nb] = ls; f->J[f->nb] = p; p += (size_t)f->nr * ls; f->nb++;
        }
        f->r = p; p += f->nr; nrow += f->nr;
    }
    for (int k = 0; k < w->n_imu; k++) {
        if (w->imu_c[k * ORC_IMU_STRIDE + 16] > 10.0) continue;
        fac_t *f = &L->f[fi++];
        int i = w->imu_i[k], j = w->imu_j[k];
        f->nr = 15; f->nb = 4;
        int lo[4] = {15 * i, 15 * i + 6, 15 * j, 15 * j + 6}, ls[4] = {6, 9, 6, 9};
        for (int b = 0; b < 4; b++) { f->lo[b] = lo[b]; f->ls[b] = ls[b]; f->J[b] = p; p += 15 * ls[b]; }
        f->r = p; p += 15; nrow += 15;
    }
    for (int k = 0; k < w->n_proj; k++) {
        fac_t *f = &L->f[fi++];
        f->nr = 2; f->nb = 0;
        int lo[4] = {15 * w->proj_i[k], 15 * w->proj_j[k], w->ex_constant ? -1 : 15 * F, L->nc + w->proj_l[k]}, ls[4] = {6, 6, 6, 1};
        for (int b = 0; b < 4; b++) { if (lo[b] < 0) continue; f->lo[f->nb] = lo[b]; f->ls[f->nb] = ls[b]; f->J[f->nb] = p; p += 2 * ls[b]; f->nb++; }
        f->r = p; p += 2; nrow += 2;
    }
    for (int k = 0; k < w->n_line; k++) {
        fac_t *f = &L->f[fi++];
        f->nr = 2; f->nb = 1; f->lo[0] = 15 * w->line_f[k]; f->ls[0] = 6; f->J[0] = p; p += 12;
        f->r = p; p += 2; nrow += 2;
    }
    L->nrow = nrow;
}
static void lin_free(lin_t *L) { free(L->f); free(L->pool); }

static void to_local(const double *Jg, int nr, int gs, int ls, double *Jl) { /* J_local = J_global [I;0] */
    for (int i = 0; i < nr; i++) for (int j = 0; j < ls; j++) Jl[i * ls + j] = Jg[i * gs + j];
}

/* evaluate every factor at the window's current state; want_jac = 0 -> residuals/cost only */
static double lin_eval(const orc_window *w, lin_t *L, const double *imu_sqrt, int want_jac) {
    int fi = 0;
    double cost = 0;
    if (w->prior_n > 0) {
        fac_t *f = &L->f[fi++];
        orc_prior_residual(w, f->r);
        if (want_jac) {
            int n = w->prior_n, b = 0;
            for (int k = 0; k < w->prior_nblk; k++) {
                if (loff_of(w, w->prior_kind[k], w->prior_index[k]) < 0) continue;
                int ls = f->ls[b], idx = w->prior_idx[k];
                for (int i = 0; i < n; i++) for (int j = 0; j < ls; j++) f->J[b][i * ls + j] = w->prior_J0[(idx + j) * n + i];
                b++;
            }
        }
        double s = 0; for (int i = 0; i < f->nr; i++) s += f->r[i] * f->r[i];
        f->cost = 0.5 * s; cost += f->cost;
    }
    for (int k = 0; k < w->n_imu; k++) {
        const double *c = w->imu_c + (size_t)k * ORC_IMU_STRIDE;
        if (c[16] > 10.0) continue;
        fac_t *f = &L->f[fi++];
        int i = w->imu_i[k], j = w->imu_j[k];
        double J0[105], J1[135], J2[105], J3[135]; double *jj[4] = {J0, J1, J2, J3};
        orc_imu_evaluate(w->pose + 7 * i, w->speedbias + 9 * i, w->pose + 7 * j, w->speedbias + 9 * j, c, w->G,
                         imu_sqrt + 225 * k, f->r, want_jac ? jj : NULL);
        if (want_jac) { to_local(J0, 15, 7, 6, f->J[0]); to_local(J1, 15, 9, 9, f->J[1]); to_local(J2, 15, 7, 6, f->J[2]); to_local(J3, 15, 9, 9, f->J[3]); }
        double s = 0; for (int q = 0; q < 15; q++) s += f->r[q] * f->r[q];
        f->cost = 0.5 * s; cost += f->cost;
    }
    for (int k = 0; k < w->n_proj; k++) {
        fac_t *f = &L->f[fi++];
        double J0[14], J1[14], J2[14], J3[2]; double *jj[4] = {J0, J1, J2, J3}; int cols[4] = {7, 7, 7, 1};
        orc_proj_evaluate(w->pose + 7 * w->proj_i[k], w->pose + 7 * w->proj_j[k], w->ex_pose, w->lam[w->proj_l[k]],
                          w->proj_pts + 6 * k, w->proj_pts + 6 * k + 3, w->proj_sqrt_info, f->r, want_jac ? jj : NULL);
        f->cost = orc_loss_correct(2, f->r, 4, want_jac ? jj : NULL, cols, w->proj_loss_a);
        cost += f->cost;
        if (want_jac) {
            int b = 0;
            to_local(J0, 2, 7, 6, f->J[b++]); to_local(J1, 2, 7, 6, f->J[b++]);
            if (!w->ex_constant) to_local(J2, 2, 7, 6, f->J[b++]);
            to_local(J3, 2, 1, 1, f->J[b++]);
        }
    }
    for (int k = 0; k < w->n_line; k++) {
        fac_t *f = &L->f[fi++];
        double J0[14]; double *jj[1] = {J0}; int cols[1] = {7};
        orc_line_evaluate(w->pose + 7 * w->line_f[k], w->line_c + 9 * k, w->K, w->Ric, w->Tic, f->r, want_jac ? J0 : NULL);
        f->cost = orc_loss_correct(2, f->r, 1, want_jac ? jj : NULL, cols, w->line_loss_a);
        cost += f->cost;
        if (want_jac) to_local(J0, 2, 7, 6, f->J[0]);
    }
    return cost;
}

static void lin_colsq(const lin_t *L, double *out) { /* squared column norms */
    for (int j = 0; j < L->nl; j++) out[j] = 0;
    for (int fi = 0; fi < L->nfac; fi++) {
        const fac_t *f = &L->f[fi];
        for (int b = 0; b < f->nb; b++)
            for (int i = 0; i < f->nr; i++) for (int j = 0; j < f->ls[b]; j++) { double v = f->J[b][i * f->ls[b] + j]; out[f->lo[b] + j] += v * v; }
    }
}
static void lin_scale(lin_t *L, const double *s) {
    for (int fi = 0; fi < L->nfac; fi++) {
        fac_t *f = &L->f[fi];
        for (int b = 0; b < f->nb; b++)
            for (int i = 0; i < f->nr; i++) for (int j = 0; j < f->ls[b]; j++) f->J[b][i * f->ls[b] + j] *= s[f->lo[b] + j];
    }
}
static void lin_JtR(const lin_t *L, double *g) { /* g = J' r */
    for (int j = 0; j < L->nl; j++) g[j] = 0;
    for (int fi = 0; fi < L->nfac; fi++) {
        const fac_t *f = &L->f[fi];
        for (int b = 0; b < f->nb; b++)
            for (int i = 0; i < f->nr; i++) for (int j = 0; j < f->ls[b]; j++) g[f->lo[b] + j] += f->J[b][i * f->ls[b] + j] * f->r[i];
    }
}
/* out rows = J v, also returns via pointers sum (Jv)^2 and -(Jv)'(r + Jv/2) */
static void lin_Jv(const lin_t *L, const double *v, double *sq, double *model) {
    double a = 0, m = 0;
    for (int fi = 0; fi < L->nfac; fi++) {
        const fac_t *f = &L->f[fi];
        for (int i = 0; i < f->nr; i++) {
            double s = 0;
            for (int b = 0; b < f->nb; b++) for (int j = 0; j < f->ls[b]; j++) s += f->J[b][i * f->ls[b] + j] * v[f->lo[b] + j];
            a += s * s; m += s * (f->r[i] + s / 2.0);
        }
    }
    if (sq) *sq = a;
    if (model) *model = -m;
}
static void lin_H(const lin_t *L, double *H) { /* H = J'J dense nl x nl */
    int n = L->nl;
    memset(H, 0, sizeof(double) * n * n);
    for (int fi = 0; fi < L->nfac; fi++) {
        const fac_t *f = &L->f[fi];
        for (int a = 0; a < f->nb; a++)
            for (int b = 0; b < f->nb; b++) {
                const double *Ja = f->J[a], *Jb = f->J[b]; int la = f->ls[a], lb = f->ls[b];
                for (int r = 0; r < f->nr; r++)
                    for (int i = 0; i < la; i++) {
                        double v = Ja[r * la + i]; if (v == 0.0) continue;
                        double *row = H + (size_t)(f->lo[a] + i) * n + f->lo[b];
                        for (int j = 0; j < lb; j++) row[j] += v * Jb[r * lb + j];
                    }
            }
    }
}

/* (H + diag(reg)) y = g via landmark Schur + dense Cholesky (SPARSE_SCHUR restated densely) */
static int schur_solve(int nl, int nc, const double *H, const double *reg, const double *g, double *y) {
    int L = nl - nc;
    double *S = (double *)malloc(sizeof(double) * nc * nc), *rhs = (double *)malloc(sizeof(double) * nc);
    int ok = 1;
    for (int i = 0; i < nc; i++) { for (int j = 0; j < nc; j++) S[i * nc + j] = H[(size_t)i * nl + j]; S[i * nc + i] += reg[i]; rhs[i] = g[i]; }
    int *nz = (int *)malloc(sizeof(int) * nc);
    for (int l = 0; l < L; l++) {
        double hll = H[(size_t)(nc + l) * nl + nc + l] + reg[nc + l];
        if (!(hll > 0) || !isfinite(hll)) { ok = 0; break; }
        int cnt = 0;
        for (int i = 0; i < nc; i++) if (H[(size_t)i * nl + nc + l] != 0.0) nz[cnt++] = i;
        for (int a = 0; a < cnt; a++) {
            int i = nz[a]; double wi = H[(size_t)i * nl + nc + l] / hll;
            for (int b = 0; b < cnt; b++) { int j = nz[b]; S[i * nc + j] -= wi * H[(size_t)j * nl + nc + l]; }
            rhs[i] -= wi * g[nc + l];
        }
    }
    if (ok) {
        for (int j = 0; j < nc && ok; j++) { /* Cholesky, lower, in place */
            double d = S[j * nc + j];
            for (int k = 0; k < j; k++) d -= S[j * nc + k] * S[j * nc + k];
            if (!(d > 0) || !isfinite(d)) { ok = 0; break; }
            d = sqrt(d); S[j * nc + j] = d;
            for (int i = j + 1; i < nc; i++) {
                double s = S[i * nc + j];
                for (int k = 0; k < j; k++) s -= S[i * nc + k] * S[j * nc + k];
                S[i * nc + j] = s / d;
            }
        }
    }
    if (ok) {
        for (int i = 0; i < nc; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= S[i * nc + k] * y[k]; y[i] = s / S[i * nc + i]; }
        for (int i = nc - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < nc; k++) s -= S[k * nc + i] * y[k]; y[i] = s / S[i * nc + i]; }
        for (int l = 0; l < L; l++) {
            double hll = H[(size_t)(nc + l) * nl + nc + l] + reg[nc + l], s = g[nc + l];
            for (int i = 0; i < nc; i++) { double h = H[(size_t)i * nl + nc + l]; if (h != 0.0) s -= h * y[i]; }
            y[nc + l] = s / hll;
        }
        for (int i = 0; i < nl; i++) if (!isfinite(y[i])) ok = 0;
    }
    free(S); free(rhs); free(nz);
    return ok;
}

static void apply_plus(const orc_window *w, const double *pose, const double *sb, const double *ex, const double *lam,
                       const double *delta, int nc, double *pose_o, double *sb_o, double *ex_o, double *lam_o) {
    int F = w->n_frames;
    for (int i = 0; i < F; i++) {
        orc_pose_plus(pose + 7 * i, delta + 15 * i, pose_o + 7 * i);
        for (int k = 0; k < 9; k++) sb_o[9 * i + k] = sb[9 * i + k] + delta[15 * i + 6 + k];
    }
    if (w->ex_constant) memcpy(ex_o, ex, 56); else orc_pose_plus(ex, delta + 15 * F, ex_o);
    for (int l = 0; l < w->n_landmarks; l++) lam_o[l] = lam[l] + delta[nc + l];
}
static double ambient_norm(const orc_window *w, const double *pose, const double *sb, const double *ex, const double *lam,
                           const double *pose2, const double *sb2, const double *ex2, const double *lam2) {
    double s = 0; int F = w->n_frames;
    for (int i = 0; i < 7 * F; i++) { double d = pose[i] - (pose2 ? pose2[i] : 0); s += d * d; }
    for (int i = 0; i < 9 * F; i++) { double d = sb[i] - (sb2 ? sb2[i] : 0); s += d * d; }
    if (!w->ex_constant) for (int i = 0; i < 7; i++) { double d = ex[i] - (ex2 ? ex2[i] : 0); s += d * d; }
    for (int i = 0; i < w->n_landmarks; i++) { double d = lam[i] - (lam2 ? lam2[i] : 0); s += d * d; }
    return sqrt(s);
}

static double *make_imu_sqrt(const orc_window *w) {
    double *S = (double *)malloc(sizeof(double) * 225 * (w->n_imu > 0 ? w->n_imu : 1));
    for (int k = 0; k < w->n_imu; k++) {
        if (w->imu_sqrt) memcpy(S + 225 * k, w->imu_sqrt + 225 * k, sizeof(double) * 225);
        else orc_imu_sqrt_info(w->imu_c + (size_t)k * ORC_IMU_STRIDE + 62, S + 225 * k);
    }
    return S;
}

int orc_linearize_dense(const orc_window *w, double *H, double *g, double *cost, int *nlocal, int *nc) {
    lin_t L; lin_alloc(w, &L);
    double *S = make_imu_sqrt(w);
    *cost = lin_eval(w, &L, S, 1);
    lin_H(&L, H); lin_JtR(&L, g);
    *nlocal = L.nl; *nc = L.nc;
    free(S); lin_free(&L);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Trust-region minimiser with traditional dogleg: upstream Ceres 2.x defaults, NOT in
 * /root/reference, unverified here.  Options set by the reference:
 * estimator.cpp:1888-1897 (SPARSE_SCHUR, DOGLEG, max_num_iterations).
 * ---------------------------------------------------------------------------------------- */
int orc_solve(orc_window *w, int max_it, int fixed, orc_summary *out) {
    lin_t L; lin_alloc(w, &L);
    int nl = L.nl, nc = L.nc, F = w->n_frames, NL = w->n_landmarks;
    double *imu_sqrt = make_imu_sqrt(w);
    double *scale = (double *)malloc(sizeof(double) * nl * 10), *diag = scale + nl, *grad = diag + nl, *gn = grad + nl,
           *step = gn + nl, *g = step + nl, *y = g + nl, *tmp = y + nl, *delta = tmp + nl, *reg = delta + nl;
    double *H = (double *)malloc(sizeof(double) * nl * nl);
    double *pose_c = (double *)malloc(sizeof(double) * (16 * F + 7 + NL)), *sb_c = pose_c + 7 * F, *ex_c = sb_c + 9 * F, *lam_c = ex_c + 7;
    double *pose_s = (double *)malloc(sizeof(double) * (16 * F + 7 + NL)), *sb_s = pose_s + 7 * F, *ex_s = sb_s + 9 * F, *lam_s = ex_s + 7;
    memset(out, 0, sizeof *out);
    out->n_local = nl; out->n_cam = nc;
    double cost = lin_eval(w, &L, imu_sqrt, 1);
    lin_colsq(&L, scale);
    for (int j = 0; j < nl; j++) scale[j] = 1.0 / (1.0 + sqrt(scale[j]));
    lin_JtR(&L, g); /* unscaled gradient for the gradient tolerance */
    double gmax = 0; for (int j = 0; j < nl; j++) if (fabs(g[j]) > gmax) gmax = fabs(g[j]);
    lin_scale(&L, scale);
    out->initial_cost = cost; out->cost[0] = cost; out->step_ok[0] = 1; out->num_iterations = 1;
    int term = 0;
    if (!fixed && gmax <= 1e-10) { term = 1; goto done; }
    {
        double radius = 1e4, mu = 1e-8, alpha = 0;
        const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
        int reuse = 0, invalid = 0, it = 0;
        double x_norm = ambient_norm(w, w->pose, w->speedbias, w->ex_pose, w->lam, 0, 0, 0, 0);
        while (it < max_it) {
            it++;
            int rec = out->num_iterations < 128 ? out->num_iterations : 127;
            int ls_ok = 1;
            if (!reuse) {
                reuse = 1;
                lin_colsq(&L, diag);
                for (int j = 0; j < nl; j++) { double d = diag[j]; d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d); diag[j] = sqrt(d); }
                lin_JtR(&L, g);
                for (int j = 0; j < nl; j++) { grad[j] = g[j] / diag[j]; tmp[j] = grad[j] / diag[j]; }
                double g2 = 0, Jg2; for (int j = 0; j < nl; j++) g2 += grad[j] * grad[j];
                lin_Jv(&L, tmp, &Jg2, 0);
                alpha = g2 / Jg2;
                lin_H(&L, H);
                ls_ok = 0;
                while (mu < max_mu) {
                    for (int j = 0; j < nl; j++) reg[j] = mu * diag[j] * diag[j];
                    if (schur_solve(nl, nc, H, reg, g, y)) { ls_ok = 1; break; }
                    mu *= mu_inc;
                }
                if (ls_ok) for (int j = 0; j < nl; j++) gn[j] = -diag[j] * y[j];
            }
            out->mu[rec] = mu;
            int valid = 0; double model = 0, step_norm = 0; int dcase = 0;
            if (ls_ok) {
                double gnorm = 0, gnn = 0, gdot = 0;
                for (int j = 0; j < nl; j++) { gnorm += grad[j] * grad[j]; gnn += gn[j] * gn[j]; gdot += grad[j] * gn[j]; }
                gnorm = sqrt(gnorm); gnn = sqrt(gnn);
                if (gnn <= radius) { for (int j = 0; j < nl; j++) step[j] = gn[j]; step_norm = gnn; dcase = 1; }
                else if (gnorm * alpha >= radius) { for (int j = 0; j < nl; j++) step[j] = -(radius / gnorm) * grad[j]; step_norm = radius; dcase = 2; }
                else {
                    double b_dot_a = -alpha * gdot, a_sq = pow(alpha * gnorm, 2.0), bma = a_sq - 2 * b_dot_a + pow(gnn, 2);
                    double c = b_dot_a - a_sq, d = sqrt(c * c + bma * (pow(radius, 2.0) - a_sq));
                    double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
                    double s2 = 0;
                    for (int j = 0; j < nl; j++) { step[j] = (-alpha * (1.0 - beta)) * grad[j] + beta * gn[j]; s2 += step[j] * step[j]; }
                    step_norm = sqrt(s2); dcase = 3;
                }
                for (int j = 0; j < nl; j++) step[j] /= diag[j];
                lin_Jv(&L, step, 0, &model);
                valid = model > 0.0;
            }
            out->dogleg_case[rec] = dcase; out->radius[rec] = radius; out->model_cost_change[rec] = model; out->step_norm[rec] = step_norm;
            if (!valid) {
                invalid++;
                out->cost[rec] = cost; out->step_ok[rec] = 0; out->num_iterations = rec + 1;
                if (invalid >= 5) { term = 5; break; }
                mu *= mu_inc; reuse = 0;
                continue;
            }
            invalid = 0;
            for (int j = 0; j < nl; j++) delta[j] = step[j] * scale[j];
            if (it == 1) for (int j = 0; j < nl && j < 2048; j++) out->first_delta[j] = delta[j];
            apply_plus(w, w->pose, w->speedbias, w->ex_pose, w->lam, delta, nc, pose_c, sb_c, ex_c, lam_c);
            /* evaluate cost at the candidate (swap states in, evaluate, keep a copy of x) */
            memcpy(pose_s, w->pose, sizeof(double) * 7 * F); memcpy(sb_s, w->speedbias, sizeof(double) * 9 * F);
            memcpy(ex_s, w->ex_pose, 56); memcpy(lam_s, w->lam, sizeof(double) * NL);
            memcpy(w->pose, pose_c, sizeof(double) * 7 * F); memcpy(w->speedbias, sb_c, sizeof(double) * 9 * F);
            memcpy(w->ex_pose, ex_c, 56); memcpy(w->lam, lam_c, sizeof(double) * NL);
            /* residual buffers are overwritten by the cost-only evaluation; keep r for a rejected step */
            double *r_keep = (double *)malloc(sizeof(double) * L.nrow); { int o = 0; for (int fi = 0; fi < L.nfac; fi++) { memcpy(r_keep + o, L.f[fi].r, sizeof(double) * L.f[fi].nr); o += L.f[fi].nr; } }
            double cost_c = lin_eval(w, &L, imu_sqrt, 0);
            out->cost_candidate[rec] = cost_c;
            double dxn = ambient_norm(w, pose_s, sb_s, ex_s, lam_s, pose_c, sb_c, ex_c, lam_c);
            int stop = 0;
            if (!fixed && dxn <= 1e-8 * (x_norm + 1e-8)) { term = 2; stop = 1; }
            double cost_change = cost - cost_c;
            if (!stop && !fixed && fabs(cost_change) <= 1e-6 * cost) { term = 3; stop = 1; }
            double rho = cost_change / model;
            out->rho[rec] = rho;
            if (stop || !(rho > 1e-3)) { /* restore x and its residuals */
                memcpy(w->pose, pose_s, sizeof(double) * 7 * F); memcpy(w->speedbias, sb_s, sizeof(double) * 9 * F);
                memcpy(w->ex_pose, ex_s, 56); memcpy(w->lam, lam_s, sizeof(double) * NL);
                int o = 0; for (int fi = 0; fi < L.nfac; fi++) { memcpy(L.f[fi].r, r_keep + o, sizeof(double) * L.f[fi].nr); o += L.f[fi].nr; }
            }
            free(r_keep);
            if (stop) { out->cost[rec] = cost; out->step_ok[rec] = 0; out->num_iterations = rec + 1; break; }
            if (rho > 1e-3) {
                x_norm = ambient_norm(w, w->pose, w->speedbias, w->ex_pose, w->lam, 0, 0, 0, 0);
                cost = lin_eval(w, &L, imu_sqrt, 1);
                lin_JtR(&L, g);
                gmax = 0; for (int j = 0; j < nl; j++) if (fabs(g[j]) > gmax) gmax = fabs(g[j]);
                lin_scale(&L, scale);
                if (rho < 0.25) radius *= 0.5;
                if (rho > 0.75) radius = radius > 3.0 * step_norm ? radius : 3.0 * step_norm;
                mu = (2.0 * mu / mu_inc) > min_mu ? (2.0 * mu / mu_inc) : min_mu;
                reuse = 0;
                out->cost[rec] = cost; out->step_ok[rec] = 1; out->num_iterations = rec + 1;
                if (!fixed && gmax <= 1e-10) { term = 1; break; }
            } else {
                radius *= 0.5; reuse = 1;
                out->cost[rec] = cost; out->step_ok[rec] = 0; out->num_iterations = rec + 1;
            }
            if (radius < 1e-32) { term = 4; break; }
        }
    }
done:
    out->termination = term; out->final_cost = cost;
    free(scale); free(H); free(pose_c); free(pose_s); free(imu_sqrt); lin_free(&L);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Symmetric eigen-decomposition, cyclic Jacobi (stands in for Eigen::SelfAdjointEigenSolver:
 * ascending eigenvalues, orthonormal eigenvectors, arbitrary sign).
 * ---------------------------------------------------------------------------------------- */
void orc_eig_sym(int n, double *A, double *ev, double *V) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[j * n + i] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, dn = 0;
        for (int i = 0; i < n; i++) { dn += A[i * n + i] * A[i * n + i]; for (int j = 0; j < i; j++) off += 2 * A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-60 * dn || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
                for (int k = 0; k < n; k++) { double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
                A[p * n + q] = 0; A[q * n + p] = 0;
                for (int k = 0; k < n; k++) { double vkp = V[p * n + k], vkq = V[q * n + k]; V[p * n + k] = c * vkp - s * vkq; V[q * n + k] = s * vkp + c * vkq; }
            }
    }
    /* sort ascending (selection) */
    for (int i = 0; i < n; i++) ev[i] = A[i * n + i];
    for (int i = 0; i < n; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++) if (ev[j] < ev[m]) m = j;
        if (m != i) {
            double t = ev[i]; ev[i] = ev[m]; ev[m] = t;
            for (int k = 0; k < n; k++) { double u = V[i * n + k]; V[i * n + k] = V[m * n + k]; V[m * n + k] = u; }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * M1-M4  marginalization_factor.cpp:89-321 with the MARGIN_OLD factor set of
 * estimator.cpp:1911-2046.  Deterministic order: dropped blocks (pose0, sb0, then landmarks
 * anchored in frame 0 by index), then kept blocks in problem order (pose1, sb1, pose2, ..., ex).
 * ---------------------------------------------------------------------------------------- */
int orc_marginalize_old(const orc_window *w, int *m_out, int *n_out, int *nblk_out, int *kind, int *index, int *size,
                        int *idx, double *x0, double *J0, double *r0, double *As_out, double *bs_out) {
    int F = w->n_frames, NL = w->n_landmarks;
    /* which blocks are touched */
    int *pose_t = (int *)calloc(3 * F + 1 + NL, sizeof(int)), *sb_t = pose_t + F, *lm_d = sb_t + F; int ex_t = 0;
    int imu0 = -1;
    if (w->prior_n > 0) for (int k = 0; k < w->prior_nblk; k++) {
        if (w->prior_kind[k] == ORC_BLK_POSE) pose_t[w->prior_index[k]] = 1;
        else if (w->prior_kind[k] == ORC_BLK_SB) sb_t[w->prior_index[k]] = 1; else ex_t = 1;
    }
    for (int k = 0; k < w->n_imu; k++) if (w->imu_i[k] == 0 && w->imu_c[k * ORC_IMU_STRIDE + 16] < 10.0) { imu0 = k; pose_t[0] = sb_t[0] = 1; pose_t[w->imu_j[k]] = sb_t[w->imu_j[k]] = 1; }
    for (int k = 0; k < w->n_proj; k++) if (w->proj_i[k] == 0) { pose_t[0] = 1; pose_t[w->proj_j[k]] = 1; ex_t = 1; lm_d[w->proj_l[k]] = 1; }
    /* local index of every block */
    int *pose_ix = (int *)malloc(sizeof(int) * (2 * F + 1 + NL)), *sb_ix = pose_ix + F, *lm_ix = sb_ix + F + 1; int ex_ix = -1;
    int pos = 0;
    for (int i = 0; i < F; i++) { pose_ix[i] = sb_ix[i] = -1; }
    if (pose_t[0]) { pose_ix[0] = pos; pos += 6; }
    if (sb_t[0]) { sb_ix[0] = pos; pos += 9; }
    for (int l = 0; l < NL; l++) { lm_ix[l] = -1; if (lm_d[l]) { lm_ix[l] = pos; pos += 1; } }
    int m = pos, nb = 0;
    double *xp = x0;
    for (int i = 1; i < F; i++) {
        if (pose_t[i]) { pose_ix[i] = pos; kind[nb] = ORC_BLK_POSE; index[nb] = i - 1; size[nb] = 7; idx[nb] = pos - m; memcpy(xp, w->pose + 7 * i, 56); xp += 7; nb++; pos += 6; }
        if (sb_t[i]) { sb_ix[i] = pos; kind[nb] = ORC_BLK_SB; index[nb] = i - 1; size[nb] = 9; idx[nb] = pos - m; memcpy(xp, w->speedbias + 9 * i, 72); xp += 9; nb++; pos += 9; }
    }
    if (ex_t) { ex_ix = pos; kind[nb] = ORC_BLK_EX; index[nb] = 0; size[nb] = 7; idx[nb] = pos - m; memcpy(xp, w->ex_pose, 56); xp += 7; nb++; pos += 6; }
    int n = pos - m;
    double *A = (double *)calloc((size_t)pos * pos + pos, sizeof(double)), *b = A + (size_t)pos * pos;
#define ACC(NR, Ja, la, ia, Jb, lb, ib)                                                                     \
    for (int r_ = 0; r_ < (NR); r_++) for (int i_ = 0; i_ < (la); i_++) for (int j_ = 0; j_ < (lb); j_++)   \
        A[(size_t)((ia) + i_) * pos + (ib) + j_] += (Ja)[r_ * (la) + i_] * (Jb)[r_ * (lb) + j_];
    /* prior factor (ResidualBlockInfo with loss NULL) */
    if (w->prior_n > 0) {
        int pn = w->prior_n;
        double *r = (double *)malloc(sizeof(double) * pn);
        orc_prior_residual(w, r);
        for (int a = 0; a < w->prior_nblk; a++) {
            int ia = w->prior_kind[a] == ORC_BLK_POSE ? pose_ix[w->prior_index[a]] : (w->prior_kind[a] == ORC_BLK_SB ? sb_ix[w->prior_index[a]] : ex_ix);
            int la = w->prior_size[a] == 7 ? 6 : w->prior_size[a];
            for (int bq = 0; bq < w->prior_nblk; bq++) {
                int ib = w->prior_kind[bq] == ORC_BLK_POSE ? pose_ix[w->prior_index[bq]] : (w->prior_kind[bq] == ORC_BLK_SB ? sb_ix[w->prior_index[bq]] : ex_ix);
                int lb = w->prior_size[bq] == 7 ? 6 : w->prior_size[bq];
                for (int i = 0; i < la; i++) for (int j = 0; j < lb; j++) {
                    double s = 0;
                    const double *ca = w->prior_J0 + (size_t)(w->prior_idx[a] + i) * pn, *cb = w->prior_J0 + (size_t)(w->prior_idx[bq] + j) * pn;
                    for (int q = 0; q < pn; q++) s += ca[q] * cb[q];
                    A[(size_t)(ia + i) * pos + ib + j] += s;
                }
            }
            for (int i = 0; i < la; i++) { double s = 0; const double *ca = w->prior_J0 + (size_t)(w->prior_idx[a] + i) * pn; for (int q = 0; q < pn; q++) s += ca[q] * r[q]; b[ia + i] += s; }
        }
        free(r);
    }
    if (imu0 >= 0) {
        int j = w->imu_j[imu0];
        double S[225], r[15], J0g[105], J1g[135], J2g[105], J3g[135], Jl[4][135]; double *jj[4] = {J0g, J1g, J2g, J3g};
        if (w->imu_sqrt) memcpy(S, w->imu_sqrt + 225 * imu0, sizeof S); else orc_imu_sqrt_info(w->imu_c + (size_t)imu0 * ORC_IMU_STRIDE + 62, S);
        orc_imu_evaluate(w->pose, w->speedbias, w->pose + 7 * j, w->speedbias + 9 * j, w->imu_c + (size_t)imu0 * ORC_IMU_STRIDE, w->G, S, r, jj);
        to_local(J0g, 15, 7, 6, Jl[0]); to_local(J1g, 15, 9, 9, Jl[1]); to_local(J2g, 15, 7, 6, Jl[2]); to_local(J3g, 15, 9, 9, Jl[3]);
        int ix[4] = {pose_ix[0], sb_ix[0], pose_ix[j], sb_ix[j]}, ls[4] = {6, 9, 6, 9};
        for (int a = 0; a < 4; a++) {
            for (int bq = 0; bq < 4; bq++) { ACC(15, Jl[a], ls[a], ix[a], Jl[bq], ls[bq], ix[bq]); }
            for (int i = 0; i < ls[a]; i++) { double s = 0; for (int q = 0; q < 15; q++) s += Jl[a][q * ls[a] + i] * r[q]; b[ix[a] + i] += s; }
        }
    }
    for (int k = 0; k < w->n_proj; k++) {
        if (w->proj_i[k] != 0) continue;
        double r[2], J0g[14], J1g[14], J2g[14], J3g[2], Jl[4][12]; double *jj[4] = {J0g, J1g, J2g, J3g}; int cols[4] = {7, 7, 7, 1};
        orc_proj_evaluate(w->pose, w->pose + 7 * w->proj_j[k], w->ex_pose, w->lam[w->proj_l[k]], w->proj_pts + 6 * k, w->proj_pts + 6 * k + 3, w->proj_sqrt_info, r, jj);
        orc_loss_correct(2, r, 4, jj, cols, w->proj_loss_a);
        to_local(J0g, 2, 7, 6, Jl[0]); to_local(J1g, 2, 7, 6, Jl[1]); to_local(J2g, 2, 7, 6, Jl[2]); Jl[3][0] = J3g[0]; Jl[3][1] = J3g[1];
        int ix[4] = {pose_ix[0], pose_ix[w->proj_j[k]], ex_ix, lm_ix[w->proj_l[k]]}, ls[4] = {6, 6, 6, 1};
        for (int a = 0; a < 4; a++) {
            for (int bq = 0; bq < 4; bq++) { ACC(2, Jl[a], ls[a], ix[a], Jl[bq], ls[bq], ix[bq]); }
            for (int i = 0; i < ls[a]; i++) b[ix[a] + i] += Jl[a][i] * r[0] + Jl[a][ls[a] + i] * r[1];
        }
    }
#undef ACC
    /* Amm = 0.5 (A_mm + A_mm'), eigen pseudo-inverse with eps = 1e-8 (:267-272) */
    const double eps = 1e-8;
    double *Amm = (double *)malloc(sizeof(double) * (3 * (size_t)m * m + m)), *Vm = Amm + (size_t)m * m, *Ainv = Vm + (size_t)m * m, *evm = Ainv + (size_t)m * m;
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Amm[i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
    orc_eig_sym(m, Amm, evm, Vm);
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) {
        double s = 0;
        for (int k = 0; k < m; k++) if (evm[k] > eps) s += Vm[k * m + i] * (1.0 / evm[k]) * Vm[k * m + j];
        Ainv[i * m + j] = s;
    }
    /* Schur (:275-282) */
    double *T = (double *)malloc(sizeof(double) * ((size_t)n * m + 2 * (size_t)n * n + 2 * n)), *A2 = T + (size_t)n * m, *V2 = A2 + (size_t)n * n, *b2 = V2 + (size_t)n * n, *ev2 = b2 + n;
    for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) { double s = 0; for (int k = 0; k < m; k++) s += A[(size_t)(m + i) * pos + k] * Ainv[k * m + j]; T[i * m + j] = s; }
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < m; k++) s += T[i * m + k] * A[(size_t)k * pos + m + j]; A2[i * n + j] = A[(size_t)(m + i) * pos + m + j] - s; }
        double s = 0; for (int k = 0; k < m; k++) s += T[i * m + k] * b[k];
        b2[i] = b[m + i] - s;
    }
    if (As_out) memcpy(As_out, A2, sizeof(double) * n * n);
    if (bs_out) memcpy(bs_out, b2, sizeof(double) * n);
    /* second eigen-decomposition and square-root factors (:284-293) */
    orc_eig_sym(n, A2, ev2, V2);
    for (int k = 0; k < n; k++) {
        double S = ev2[k] > eps ? ev2[k] : 0.0, Si = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
        double ss = sqrt(S), sis = sqrt(Si), vb = 0;
        for (int i = 0; i < n; i++) { J0[(size_t)i * n + k] = ss * V2[k * n + i]; vb += V2[k * n + i] * b2[i]; } /* J0 column-major: (k,i) at i*n+k */
        r0[k] = sis * vb;
    }
    *m_out = m; *n_out = n; *nblk_out = nb;
    free(A); free(Amm); free(T); free(pose_t); free(pose_ix);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * G3: Estimator::double2vector() gauge fix, estimator.cpp:1537-1581
 * ------------------------------------------------------------------------------------------------ */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static void r2ypr_deg(const double *R, double *ypr) { /* utility.h:70-85; R row-major, n/o/a = columns 0/1/2 */
    const double y = atan2(R[3], R[0]);
    const double p = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
    const double r = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
static void mat3_mul(const double *A, const double *B, double *Cm) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Cm[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void ypr2R_deg(double yd, double pd, double rd, double *R) { /* utility.h:87-112 */
    const double y = yd / 180.0 * M_PI, p = pd / 180.0 * M_PI, r = rd / 180.0 * M_PI;
    const double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
    const double Ry[9] = {cos(p), 0, sin(p), 0, 1, 0, -sin(p), 0, cos(p)};
    const double Rx[9] = {1, 0, 0, 0, cos(r), -sin(r), 0, sin(r), cos(r)};
    double T[9];
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, Rx, R);
}
static void mat3_to_quat(const double *m, double *q /* x y z w */) { /* Eigen Quaternion(Matrix3) */
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
void orc_gauge_fix(int n, const double *R0, const double *P0, const double *pose, const double *sb, double *Rs,
                   double *Ps, double *Vs, double *pose_out) {
    double a[3], b[3], R00[9], rot[9];
    r2ypr_deg(R0, a);                                   /* :1539 */
    q_toR(pose + 3, R00);                               /* :1548-1551 */
    r2ypr_deg(R00, b);
    ypr2R_deg(a[0] - b[0], 0.0, 0.0, rot);              /* :1552-1554 */
    if (fabs(fabs(a[1]) - 90.0) < 1.0 || fabs(fabs(b[1]) - 90.0) < 1.0) { /* :1555-1563 */
        double R00t[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R00t[3 * i + j] = R00[3 * j + i];
        mat3_mul(R0, R00t, rot);
    }
    for (int i = 0; i < n; i++) {                       /* :1565-1581 */
        double qn[4], Ri[9];
        q_normalized(pose + 7 * i + 3, qn);
        q_toR(qn, Ri);
        mat3_mul(rot, Ri, Rs + 9 * i);
        const double d[3] = {pose[7 * i] - pose[0], pose[7 * i + 1] - pose[1], pose[7 * i + 2] - pose[2]};
        for (int r = 0; r < 3; r++) {
            Ps[3 * i + r] = rot[3 * r] * d[0] + rot[3 * r + 1] * d[1] + rot[3 * r + 2] * d[2] + P0[r];
            Vs[3 * i + r] = rot[3 * r] * sb[9 * i] + rot[3 * r + 1] * sb[9 * i + 1] + rot[3 * r + 2] * sb[9 * i + 2];
        }
        if (pose_out) {
            for (int r = 0; r < 3; r++) pose_out[7 * i + r] = Ps[3 * i + r];
            mat3_to_quat(Rs + 9 * i, pose_out + 7 * i + 3);
        }
    }
}
