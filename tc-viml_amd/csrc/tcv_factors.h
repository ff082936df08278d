// Residual / Jacobian evaluators of the three TC-VIML factor types plus the marginalisation prior
// and the robust-loss corrector, written for one GPU lane per residual block (point / line) or
// one lane per block for the "raw" IMU part followed by a block-parallel sqrt_info product.
// Jacobians are produced directly in LOCAL (tangent) coordinates: the reference multiplies the
// 7-wide global Jacobian by PoseLocalParameterization::ComputeJacobian = [I6;0]
// (pose_local_parameterization.cpp:20-27), i.e. drops the 7th column.
//
// Reference files followed (relative to /root/reference/vins_estimator/src/):
//   factor/imu_factor.h:19-181, factor/integration_base.h:160-186
//   factor/projection_factor.cpp:21-124
//   factor/line_projection_factor.cpp:19-120   (Jacobian reproduced "as written")
//   factor/marginalization_factor.cpp:37-68 (corrector), :335-384 (prior)
#pragma once
#include "tcv_math.h"

namespace tcv {

// packed IMU constants, 287 doubles per pre-integration (matches SURVEY.md 8(a) I1 byte count)
enum {
    IMU_DP = 0, IMU_DQ = 3, IMU_DV = 7, IMU_BA = 10, IMU_BG = 13, IMU_DT = 16,
    IMU_DP_DBA = 17, IMU_DP_DBG = 26, IMU_DQ_DBG = 35, IMU_DV_DBA = 44, IMU_DV_DBG = 53, IMU_COV = 62,
    IMU_STRIDE = 287
};
// local column layout of an IMU block row: [p_i th_i | v_i ba_i bg_i | p_j th_j | v_j ba_j bg_j]
enum { IMU_COLS = 30, IMU_ROWS = 15, PROJ_COLS = 19, LINE_COLS = 6 };

TCV_HD void put33(double *J, int ld, int r0, int c0, const M3 &B) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[(r0 + i) * ld + c0 + j] = B.m[3 * i + j];
}

// ---- IMU factor, part 1: un-whitened residual and Jacobian (everything except sqrt_info) --------
// raw_r: 15 entries with stride rs; Jraw: 15 x 30 row-major with leading dimension ld (zeros included) or nullptr.
TCV_HD void imu_raw(const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                    const double *c, const double *G3, double *raw_r, int rs, double *Jraw, int ld) {
    const V3 Pi(pose_i), Vi(sb_i), Bai(sb_i + 3), Bgi(sb_i + 6);
    const V3 Pj(pose_j), Vj(sb_j), Baj(sb_j + 3), Bgj(sb_j + 6);
    const Quat Qi(pose_i + 3), Qj(pose_j + 3), dq0(c + IMU_DQ);
    const V3 G(G3);
    const double dt = c[IMU_DT];
    const M3 dp_dba = m3_load(c + IMU_DP_DBA), dp_dbg = m3_load(c + IMU_DP_DBG), dq_dbg = m3_load(c + IMU_DQ_DBG);
    const M3 dv_dba = m3_load(c + IMU_DV_DBA), dv_dbg = m3_load(c + IMU_DV_DBG);
    const V3 dba = Bai - V3(c + IMU_BA), dbg = Bgi - V3(c + IMU_BG);
    // integration_base.h:176-178
    const Quat cdq = dq0 * delta_q(dq_dbg * dbg);
    const V3 cdv = V3(c + IMU_DV) + dv_dba * dba + dv_dbg * dbg;
    const V3 cdp = V3(c + IMU_DP) + dp_dba * dba + dp_dbg * dbg;
    const Quat Qi_inv = inverse(Qi);
    const V3 rp = rotate(Qi_inv, 0.5 * G * dt * dt + Pj - Pi - Vi * dt);
    const V3 rv = rotate(Qi_inv, G * dt + Vj - Vi);
    const Quat cdq_inv = inverse(cdq);
    const Quat qe = cdq_inv * (Qi_inv * Qj);
    // integration_base.h:180-184
    raw_r[0] = rp.x - cdp.x; raw_r[rs] = rp.y - cdp.y; raw_r[2 * rs] = rp.z - cdp.z;
    raw_r[3 * rs] = 2 * qe.x; raw_r[4 * rs] = 2 * qe.y; raw_r[5 * rs] = 2 * qe.z;
    raw_r[6 * rs] = rv.x - cdv.x; raw_r[7 * rs] = rv.y - cdv.y; raw_r[8 * rs] = rv.z - cdv.z;
    raw_r[9 * rs] = Baj.x - Bai.x; raw_r[10 * rs] = Baj.y - Bai.y; raw_r[11 * rs] = Baj.z - Bai.z;
    raw_r[12 * rs] = Bgj.x - Bgi.x; raw_r[13 * rs] = Bgj.y - Bgi.y; raw_r[14 * rs] = Bgj.z - Bgi.z;
    if (!Jraw) return;
    for (int i = 0; i < IMU_ROWS; i++) for (int j = 0; j < IMU_COLS; j++) Jraw[i * ld + j] = 0.0;
    const M3 Ri_inv = to_matrix(Qi_inv);
    const M3 nRi = -Ri_inv;
    // pose_i (cols 0..5)  imu_factor.h:93-103
    put33(Jraw, ld, 0, 0, nRi);
    put33(Jraw, ld, 0, 3, skew(rp));
    {
        const Quat ql = inverse(Qj) * Qi;
        M3 M = qleft33(ql) * qright33(cdq);  // bottom-right corner of the 4x4 product Qleft * Qright
        const V3 a = ql.vec(), b = cdq.vec();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M.m[3 * i + j] -= a[i] * b[j];
        put33(Jraw, ld, 3, 3, -M);
    }
    put33(Jraw, ld, 6, 3, skew(rv));
    // speedbias_i (cols 6..14)  :118-136
    put33(Jraw, ld, 0, 6, dt * nRi);
    put33(Jraw, ld, 0, 9, -dp_dba);
    put33(Jraw, ld, 0, 12, -dp_dbg);
    put33(Jraw, ld, 3, 12, -(qleft33((inverse(Qj) * Qi) * dq0) * dq_dbg));  // :127, un-corrected delta_q
    put33(Jraw, ld, 6, 6, nRi);
    put33(Jraw, ld, 6, 9, -dv_dba);
    put33(Jraw, ld, 6, 12, -dv_dbg);
    for (int i = 0; i < 3; i++) { Jraw[(9 + i) * ld + 9 + i] = -1.0; Jraw[(12 + i) * ld + 12 + i] = -1.0; }
    // pose_j (cols 15..20)  :148-154
    put33(Jraw, ld, 0, 15, Ri_inv);
    put33(Jraw, ld, 3, 18, qleft33((cdq_inv * Qi_inv) * Qj));
    // speedbias_j (cols 21..29)  :167-171
    put33(Jraw, ld, 6, 21, Ri_inv);
    for (int i = 0; i < 3; i++) { Jraw[(9 + i) * ld + 24 + i] = 1.0; Jraw[(12 + i) * ld + 27 + i] = 1.0; }
}

// Same Jacobian / residual as imu_raw, split into four parts so that four wavefronts can work on the same factors
// (the branch on `part` is wave-uniform).  Every part fills (and zero-fills) only its own entries of the 15 x 31 record
// [J_raw | r_raw]:  0: residual + pose_j + speedbias_j columns,  1: pose_i columns,  2: speedbias_i columns, rows
// p and theta,  3: speedbias_i columns, rows v, ba, bg.
TCV_HD void imu_raw_part(int part, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                         const double *c, const double *G3, double *rec, int ld, bool want_jac) {
    const V3 Pi(pose_i), Vi(sb_i), Bai(sb_i + 3), Bgi(sb_i + 6);
    const V3 Pj(pose_j), Vj(sb_j), Baj(sb_j + 3), Bgj(sb_j + 6);
    const Quat Qi(pose_i + 3), Qj(pose_j + 3), dq0(c + IMU_DQ);
    const V3 G(G3);
    const double dt = c[IMU_DT];
    const M3 dq_dbg = m3_load(c + IMU_DQ_DBG);
    const V3 dba = Bai - V3(c + IMU_BA), dbg = Bgi - V3(c + IMU_BG);
    const Quat cdq = dq0 * delta_q(dq_dbg * dbg);
    const Quat Qi_inv = inverse(Qi);
    if (part == 0) {
        const M3 dp_dba = m3_load(c + IMU_DP_DBA), dp_dbg = m3_load(c + IMU_DP_DBG);
        const M3 dv_dba = m3_load(c + IMU_DV_DBA), dv_dbg = m3_load(c + IMU_DV_DBG);
        const V3 cdv = V3(c + IMU_DV) + dv_dba * dba + dv_dbg * dbg;
        const V3 cdp = V3(c + IMU_DP) + dp_dba * dba + dp_dbg * dbg;
        const V3 rp = rotate(Qi_inv, 0.5 * G * dt * dt + Pj - Pi - Vi * dt);
        const V3 rv = rotate(Qi_inv, G * dt + Vj - Vi);
        const Quat cdq_inv = inverse(cdq);
        const Quat qe = cdq_inv * (Qi_inv * Qj);
        double *r = rec + 30;
        r[0] = rp.x - cdp.x; r[ld] = rp.y - cdp.y; r[2 * ld] = rp.z - cdp.z;
        r[3 * ld] = 2 * qe.x; r[4 * ld] = 2 * qe.y; r[5 * ld] = 2 * qe.z;
        r[6 * ld] = rv.x - cdv.x; r[7 * ld] = rv.y - cdv.y; r[8 * ld] = rv.z - cdv.z;
        r[9 * ld] = Baj.x - Bai.x; r[10 * ld] = Baj.y - Bai.y; r[11 * ld] = Baj.z - Bai.z;
        r[12 * ld] = Bgj.x - Bgi.x; r[13 * ld] = Bgj.y - Bgi.y; r[14 * ld] = Bgj.z - Bgi.z;
        if (!want_jac) return;
        for (int i = 0; i < 15; i++) for (int j = 15; j < 30; j++) rec[i * ld + j] = 0.0;
        const M3 Ri_inv = to_matrix(Qi_inv);
        put33(rec, ld, 0, 15, Ri_inv);                                  // pose_j  imu_factor.h:148-154
        put33(rec, ld, 3, 18, qleft33((cdq_inv * Qi_inv) * Qj));
        put33(rec, ld, 6, 21, Ri_inv);                                  // speedbias_j  :167-171
        for (int i = 0; i < 3; i++) { rec[(9 + i) * ld + 24 + i] = 1.0; rec[(12 + i) * ld + 27 + i] = 1.0; }
    } else if (part == 1) {
        if (!want_jac) return;
        for (int i = 0; i < 15; i++) for (int j = 0; j < 6; j++) rec[i * ld + j] = 0.0;
        const V3 rp = rotate(Qi_inv, 0.5 * G * dt * dt + Pj - Pi - Vi * dt);
        const V3 rv = rotate(Qi_inv, G * dt + Vj - Vi);
        const M3 Ri_inv = to_matrix(Qi_inv);
        put33(rec, ld, 0, 0, -Ri_inv);                                  // pose_i  :93-103
        put33(rec, ld, 0, 3, skew(rp));
        const Quat ql = inverse(Qj) * Qi;
        M3 M = qleft33(ql) * qright33(cdq);
        const V3 a = ql.vec(), b = cdq.vec();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M.m[3 * i + j] -= a[i] * b[j];
        put33(rec, ld, 3, 3, -M);
        put33(rec, ld, 6, 3, skew(rv));
    } else if (part == 2) {
        if (!want_jac) return;
        for (int i = 0; i < 6; i++) for (int j = 6; j < 15; j++) rec[i * ld + j] = 0.0;
        const M3 nRi = -to_matrix(Qi_inv);
        put33(rec, ld, 0, 6, dt * nRi);                                 // speedbias_i  :118-136
        put33(rec, ld, 0, 9, -m3_load(c + IMU_DP_DBA));
        put33(rec, ld, 0, 12, -m3_load(c + IMU_DP_DBG));
        put33(rec, ld, 3, 12, -(qleft33((inverse(Qj) * Qi) * dq0) * dq_dbg));   // :127, un-corrected delta_q
    } else {
        if (!want_jac) return;
        for (int i = 6; i < 15; i++) for (int j = 6; j < 15; j++) rec[i * ld + j] = 0.0;
        put33(rec, ld, 6, 6, -to_matrix(Qi_inv));
        put33(rec, ld, 6, 9, -m3_load(c + IMU_DV_DBA));
        put33(rec, ld, 6, 12, -m3_load(c + IMU_DV_DBG));
        for (int i = 0; i < 3; i++) { rec[(9 + i) * ld + 9 + i] = -1.0; rec[(12 + i) * ld + 12 + i] = -1.0; }
    }
}

// ---- imu_factor.h:64  sqrt_info = LLT(cov^-1).matrixL()^T ----------------------------------------
// Partial-pivot LU inverse followed by a lower Cholesky, one IEEE operation at a time with FMA
// contraction disabled so that the result is reproducible against the CPU oracle.
#if defined(__clang__)
#define TCV_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define TCV_NO_CONTRACT
#endif
TCV_HD int imu_sqrt_info(const double *cov, double *S /*225 row-major*/, double *work /*450*/) {
    TCV_NO_CONTRACT
    double *a = work, *inv = work + 225;
    int perm[15];
    for (int i = 0; i < 15; i++) perm[i] = i;
    for (int i = 0; i < 225; i++) a[i] = cov[i];
    for (int k = 0; k < 15; k++) {
        int p = k;
        double mv = fabs(a[k * 15 + k]);
        for (int i = k + 1; i < 15; i++) if (fabs(a[i * 15 + k]) > mv) { mv = fabs(a[i * 15 + k]); p = i; }
        if (p != k) {
            for (int j = 0; j < 15; j++) { double t = a[k * 15 + j]; a[k * 15 + j] = a[p * 15 + j]; a[p * 15 + j] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        for (int i = k + 1; i < 15; i++) a[i * 15 + k] = a[i * 15 + k] / a[k * 15 + k];
        for (int i = k + 1; i < 15; i++)
            for (int j = k + 1; j < 15; j++) a[i * 15 + j] = a[i * 15 + j] - a[i * 15 + k] * a[k * 15 + j];
    }
    for (int c = 0; c < 15; c++) {
        double y[15], x[15];
        for (int i = 0; i < 15; i++) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s = s - a[i * 15 + k] * y[k];
            y[i] = s;
        }
        for (int i = 14; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 15; k++) s = s - a[i * 15 + k] * x[k];
            x[i] = s / a[i * 15 + i];
        }
        for (int i = 0; i < 15; i++) inv[i * 15 + c] = x[i];
    }
    // Cholesky of the lower triangle of inv, L stored over `a`
    for (int i = 0; i < 15; i++)
        for (int j = 0; j <= i; j++) {
            double s = inv[i * 15 + j];
            for (int k = 0; k < j; k++) s = s - a[i * 15 + k] * a[j * 15 + k];
            if (i == j) {
                if (!(s > 0)) return -1;
                a[i * 15 + i] = sqrt(s);
            } else {
                a[i * 15 + j] = s / a[j * 15 + j];
            }
        }
    for (int r = 0; r < 15; r++)
        for (int c = 0; c < 15; c++) S[r * 15 + c] = (c >= r) ? a[c * 15 + r] : 0.0;
    return 0;
}

#if defined(__HIPCC__)
// Same computation as imu_sqrt_info, cooperatively by the 16 lanes of one lane group (lane = 0..15 of a
// 16-aligned group inside one wavefront; every element goes through the identical sequence of IEEE
// operations, so the result is bit-identical to the serial routine and to the CPU oracle).
// a, inv: 225 doubles each in LDS, private to the group.  Returns 0, or -1 if cov^-1 is not positive definite.
// (CP / SP / LP: pointer types of cov, S and of the LDS workspace -- address-space pointers in the kernels: through generic pointers every
// workspace access was a FLAT instruction with both memory counters to wait for)
template <class CP, class SP, class LP>
__device__ __forceinline__ int imu_sqrt_info_group(CP cov, SP S, LP a, LP inv, int lane) {
    TCV_NO_CONTRACT
#define TCV_GFENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
    int perm = lane;   // lane i holds perm[i]
    for (int i = lane; i < 225; i += 16) a[i] = cov[i];
    TCV_GFENCE();
    for (int k = 0; k < 15; k++) {
        int p = k;
        double mv = fabs(a[k * 15 + k]);
        for (int i = k + 1; i < 15; i++) { const double v = fabs(a[i * 15 + k]); if (v > mv) { mv = v; p = i; } }
        if (p != k) {
            if (lane < 15) { const double t = a[k * 15 + lane]; a[k * 15 + lane] = a[p * 15 + lane]; a[p * 15 + lane] = t; }
            const int pk = __shfl(perm, k, 16), pp = __shfl(perm, p, 16);
            if (lane == k) perm = pp; else if (lane == p) perm = pk;
        }
        TCV_GFENCE();
        if (lane > k && lane < 15) a[lane * 15 + k] = a[lane * 15 + k] / a[k * 15 + k];
        TCV_GFENCE();
        if (lane > k && lane < 15) {
            const double m = a[lane * 15 + k];
            for (int j = k + 1; j < 15; j++) a[lane * 15 + j] = a[lane * 15 + j] - m * a[k * 15 + j];
        }
        TCV_GFENCE();
    }
    {   // column `lane` of the inverse
        double y[15], x[15];
        for (int i = 0; i < 15; i++) {
            const int pi = __shfl(perm, i, 16);
            double s = (pi == lane) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s = s - a[i * 15 + k] * y[k];
            y[i] = s;
        }
        for (int i = 14; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 15; k++) s = s - a[i * 15 + k] * x[k];
            x[i] = s / a[i * 15 + i];
        }
        if (lane < 15) for (int i = 0; i < 15; i++) inv[i * 15 + lane] = x[i];
    }
    TCV_GFENCE();
    int bad = 0;
    for (int j = 0; j < 15; j++) {   // Cholesky of the lower triangle of inv, L over a, column by column
        double s = 0;
        if (lane >= j && lane < 15) {
            s = inv[lane * 15 + j];
            for (int k = 0; k < j; k++) s = s - a[lane * 15 + k] * a[j * 15 + k];
        }
        const double dj = __shfl(s, j, 16);
        if (!(dj > 0)) bad = 1;
        const double ljj = sqrt(dj);
        if (lane == j) a[j * 15 + j] = ljj;
        else if (lane > j && lane < 15) a[lane * 15 + j] = s / ljj;
        TCV_GFENCE();
    }
    if (lane < 15) for (int c = 0; c < 15; c++) S[lane * 15 + c] = (c >= lane) ? a[c * 15 + lane] : 0.0;
#undef TCV_GFENCE
    return bad ? -1 : 0;
}
#endif

// ---- point re-projection factor --------------------------------------------------------------------
// pts: pts_i xyz, pts_j xyz.  r[2]; J 2 x 19 row-major (leading dimension ld) local [pose_i 6 | pose_j 6 | ex 6 | inv depth 1]
TCV_HD void proj_eval(const double *pose_i, const double *pose_j, const double *ex, double inv_dep,
                      const double *pts, double sqrt_info, double *r, double *J, int ld) {
    const V3 Pi(pose_i), Pj(pose_j), tic(ex), pts_i(pts), pts_j(pts + 3);
    const Quat Qi(pose_i + 3), Qj(pose_j + 3), qic(ex + 3);
    const V3 pc_i = pts_i / inv_dep;
    const V3 pi_i = rotate(qic, pc_i) + tic;
    const V3 pw = rotate(Qi, pi_i) + Pi;
    const V3 pi_j = rotate(inverse(Qj), pw - Pj);
    const V3 pc_j = rotate(inverse(qic), pi_j - tic);
    const double dep_j = pc_j.z;
    r[0] = sqrt_info * (pc_j.x / dep_j - pts_j.x);
    r[1] = sqrt_info * (pc_j.y / dep_j - pts_j.y);
    if (!J) return;
    const M3 Ri = to_matrix(Qi), Rj = to_matrix(Qj), ric = to_matrix(qic);
    const M3 ricT = transpose(ric), RjT = transpose(Rj);
    // reduce (2x3), projection_factor.cpp:72-75
    const double red[6] = {sqrt_info * (1. / dep_j), 0.0, sqrt_info * (-pc_j.x / (dep_j * dep_j)),
                           0.0, sqrt_info * (1. / dep_j), sqrt_info * (-pc_j.y / (dep_j * dep_j))};
    const M3 A = ricT * RjT;   // ric^T Rj^T
    const M3 ARi = A * Ri;     // ric^T Rj^T Ri
    const M3 Ji_th = -(ARi * skew(pi_i));
    const M3 Jj_th = ricT * skew(pi_j);
    const M3 tmp_r = ARi * ric;
    const M3 Jex_p = ricT * (RjT * Ri - m3_identity());
    const M3 Jex_th = -(tmp_r * skew(pc_i)) + skew(tmp_r * pc_i) + skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
    const V3 jl = (tmp_r * pts_i) * (-1.0 / (inv_dep * inv_dep));
    for (int row = 0; row < 2; row++) {
        const V3 rd(red[3 * row], red[3 * row + 1], red[3 * row + 2]);
        double *o = J + row * ld;
        const V3 a = vT_mul(rd, A), b = vT_mul(rd, Ji_th), c2 = vT_mul(rd, Jj_th), d = vT_mul(rd, Jex_p), e = vT_mul(rd, Jex_th);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = b.x; o[4] = b.y; o[5] = b.z;
        o[6] = -a.x; o[7] = -a.y; o[8] = -a.z; o[9] = c2.x; o[10] = c2.y; o[11] = c2.z;
        o[12] = d.x; o[13] = d.y; o[14] = d.z; o[15] = e.x; o[16] = e.y; o[17] = e.z;
        o[18] = dot(rd, jl);
    }
}

// ---- point re-projection with time offset / rolling shutter (ProjectionTdFactor::Evaluate, projection_td_factor.cpp:34-140)
// aux: velocity_i xy, velocity_j xy, td_i, td_j, row_i, row_j (rows as passed to the constructor; ROW / 2 is subtracted like
// :17-18).  J: 2 x 20 row-major local [pose_i 6 | pose_j 6 | ex 6 | inverse depth 1 | td 1].
TCV_HD void proj_td_eval(const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td, const double *pts,
                         const double *aux, double sqrt_info, double TR, double ROW, double *r, double *J, int ld) {
    const double row_i = aux[6] - ROW / 2, row_j = aux[7] - ROW / 2;
    const double si = td - aux[4] + TR / ROW * row_i, sj = td - aux[5] + TR / ROW * row_j;     // :50-51
    const double pts_td[6] = {pts[0] - si * aux[0], pts[1] - si * aux[1], pts[2] - si * 0.0,
                              pts[3] - sj * aux[2], pts[4] - sj * aux[3], pts[5] - sj * 0.0};
    double Jp[2 * 19];
    proj_eval(pose_i, pose_j, ex, inv_dep, pts_td, sqrt_info, r, J ? Jp : nullptr, 19);
    if (!J) return;
    // jacobian_td = reduce * ric^T Rj^T Ri ric * velocity_i / inv_dep * -1 + sqrt_info * velocity_j.head(2)   (:131-136)
    const V3 Pi(pose_i), Pj(pose_j), tic(ex);
    const Quat Qi(pose_i + 3), Qj(pose_j + 3), qic(ex + 3);
    const V3 pc_i = V3(pts_td) / inv_dep;
    const V3 pc_j = rotate(inverse(qic), rotate(inverse(Qj), rotate(Qi, rotate(qic, pc_i) + tic) + Pi - Pj) - tic);
    const double dep_j = pc_j.z;
    const double red[6] = {sqrt_info * (1. / dep_j), 0.0, sqrt_info * (-pc_j.x / (dep_j * dep_j)),
                           0.0, sqrt_info * (1. / dep_j), sqrt_info * (-pc_j.y / (dep_j * dep_j))};
    const M3 tmp_r = ((transpose(to_matrix(qic)) * transpose(to_matrix(Qj))) * to_matrix(Qi)) * to_matrix(qic);
    const V3 mv = (tmp_r * V3(aux[0], aux[1], 0.0)) * (-1.0 / inv_dep);
    for (int row = 0; row < 2; row++) {
        for (int c = 0; c < 19; c++) J[row * ld + c] = Jp[row * 19 + c];
        J[row * ld + 19] = red[3 * row] * mv.x + red[3 * row + 1] * mv.y + red[3 * row + 2] * mv.z + sqrt_info * aux[2 + row];
    }
}

// ---- 2D-3D line factor (prior 3D map line vs detected 2D line) ---------------------------------------
// lc: pts_start xyz, pts_end xyz, A B C;  K / Ric / Tic row-major constants.  J 2 x 6 row-major local.
// exact Jacobian of the line residual (opt-in, see line_eval); out of line so that the default path keeps its register allocation
TCV_HD __attribute__((noinline)) void line_exact_jacobian(const V3 &T_w, const M3 &R_w, const M3 &bcRT, const V3 &pcs, const V3 &pce, double us, double vs,
                                                          double ue, double ve, const double *lc, double fx, double fy, double *J, int ld) {
    const double a = lc[6], b = lc[7], c = lc[8], d = a * a + b * b;
    const double isd = 1.0 / sqrt(d);
    for (int e = 0; e < 2; e++) {
        const V3 p = e ? pce : pcs;
        const double L = a * (e ? ue : us) + b * (e ? ve : vs) + c;          // r = |L| / sqrt(d)
        const double sg = L > 0.0 ? isd : (L < 0.0 ? -isd : 0.0);
        const double e1 = sg * a, e2 = sg * b;
        const V3 ew(e1 * (fx / p.z), e2 * (fy / p.z), e1 * (-fx * p.x / (p.z * p.z)) + e2 * (-fy * p.y / (p.z * p.z)));      // d r / d p_cam
        // p_cam = Ric' (R_w' (P - T_w) - Tic):  d p_cam / d dp = -Ric' R_w',  d p_cam / d dtheta = Ric' skew(R_w' (P - T_w))
        const V3 y = transpose(R_w) * (V3(lc + 3 * e) - T_w);
        const V3 g = vT_mul(ew, bcRT);                 // ew' Ric'
        const V3 dp = -vT_mul(g, transpose(R_w));
        const V3 th = vT_mul(g, skew(y));
        double *o = J + e * ld;
        o[0] = dp.x; o[1] = dp.y; o[2] = dp.z; o[3] = th.x; o[4] = th.y; o[5] = th.z;
    }
}

// exact = false: the reference's Jacobian as written (the derivative of the SQUARED distance chained through [I | skew(p_cam)], a
// camera-frame perturbation -- not the derivative of r with respect to the world-frame body pose);
// exact = true (opt-in extension, tcv_problem_set_line_jacobian): d r / d(delta p, delta theta) of the same residual under
// PoseLocalParameterization::Plus.
TCV_HD void line_eval(const double *pose, const double *lc, const double *K9, const double *Ric9,
                      const double *Tic3, double *r, double *J, int ld, bool exact = false) {
    const V3 T_w(pose);
    const M3 R_w = to_matrix(normalized(Quat(pose + 3)));  // line_projection_factor.cpp:33
    const M3 K = m3_load(K9), bcRT = transpose(m3_load(Ric9));
    const M3 R = bcRT * transpose(R_w);
    const V3 t = -(R * T_w) - bcRT * V3(Tic3);
    const V3 pcs = R * V3(lc) + t, pce = R * V3(lc + 3) + t;
    const V3 si = K * pcs, ei = K * pce;
    const double us = si.x / si.z, vs = si.y / si.z, ue = ei.x / ei.z, ve = ei.y / ei.z;
    const double a = lc[6], b = lc[7], c = lc[8], d = a * a + b * b;
    const double mus = (b * b * us - a * b * vs - a * c) / d, mvs = (a * a * vs - a * b * us - b * c) / d;
    const double mue = (b * b * ue - a * b * ve - a * c) / d, mve = (a * a * ve - a * b * ue - b * c) / d;
    r[0] = 1.0 * sqrt((mus - us) * (mus - us) + (mvs - vs) * (mvs - vs));
    r[1] = 1.0 * sqrt((mue - ue) * (mue - ue) + (mve - ve) * (mve - ve));
    if (!J) return;
    const double fx = K9[0], fy = K9[4];
    if (exact) {
        line_exact_jacobian(T_w, R_w, bcRT, pcs, pce, us, vs, ue, ve, lc, fx, fy, J, ld);
        return;
    }
    for (int e = 0; e < 2; e++) {
        const V3 p = e ? pce : pcs;
        const double du = e ? (mue - ue) : (mus - us), dv = e ? (mve - ve) : (mvs - vs);
        const double e1 = -2 / d * (du * a * a + a * b * dv) * 1.0, e2 = -2 / d * (du * a * b + b * b * dv) * 1.0;   // :76-80
        // _e_p (1x2) * _p_p (2x3)   :93-100
        const V3 ew(e1 * (fx / p.z), e2 * (fy / p.z), e1 * (-fx * p.x / (p.z * p.z)) + e2 * (-fy * p.y / (p.z * p.z)));
        const V3 th = vT_mul(ew, skew(p));   // * [I | skew(p_cam)]   :104-113
        double *o = J + e * ld;
        o[0] = ew.x; o[1] = ew.y; o[2] = ew.z; o[3] = th.x; o[4] = th.y; o[5] = th.z;
    }
}

// ---- robust loss corrector for a 2-row block ---------------------------------------------------------
// marginalization_factor.cpp:37-68 (same algebra as ceres::Corrector); rho from ceres::CauchyLoss.
// Scales r (2) and J (2 x ncols, may be null) in place, returns the block cost 0.5*rho0.
TCV_HD double loss_correct2(double *r, double *J, int ncols, int ld, double loss_a) {
    const double sq = r[0] * r[0] + r[1] * r[1];
    if (!(loss_a > 0)) return 0.5 * sq;
    const double b = loss_a * loss_a, cc = 1.0 / b, sum = 1.0 + sq * cc, inv = 1.0 / sum;
    const double rho0 = b * log(sum), rho1 = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308;
    const double rho2 = -cc * (inv * inv);
    const double sqrt_rho1 = sqrt(rho1);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho2 <= 0.0) {
        residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0;
    } else {
        const double D = 1.0 + 2.0 * sq * rho2 / rho1, alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq;
    }
    if (J) {
        for (int j = 0; j < ncols; j++) {
            const double j0 = J[j], j1 = J[ld + j], rtJ = r[0] * j0 + r[1] * j1;
            J[j] = sqrt_rho1 * (j0 - alpha_sq_norm * r[0] * rtJ);
            J[ld + j] = sqrt_rho1 * (j1 - alpha_sq_norm * r[1] * rtJ);
        }
    }
    r[0] *= residual_scaling; r[1] *= residual_scaling;
    return 0.5 * rho0;
}

// ---- prior: tangent-space offset of one kept block (marginalization_factor.cpp:348-364) ---------------
TCV_HD void prior_block_dx(const double *x, const double *x0, int size, double *dx /* local size */) {
    if (size != 7) {
        for (int i = 0; i < size; i++) dx[i] = x[i] - x0[i];
    } else {
        dx[0] = x[0] - x0[0]; dx[1] = x[1] - x0[1]; dx[2] = x[2] - x0[2];
        const Quat dq = inverse(Quat(x0 + 3)) * Quat(x + 3);
        dx[3] = 2.0 * dq.x; dx[4] = 2.0 * dq.y; dx[5] = 2.0 * dq.z;
        if (!(dq.w >= 0)) { dx[3] = 2.0 * -dq.x; dx[4] = 2.0 * -dq.y; dx[5] = 2.0 * -dq.z; }
    }
}

}  // namespace tcv
