// Minimal pose solver for the RANSAC kernel of cv::solvePnPRansac(..., SOLVEPNP_P3P) as the
// reference calls it (visualOdometry.cpp:122-127 selects the flag, :206-208 is the call): three
// 3-D/2-D correspondences give up to four camera poses, a fourth correspondence picks one
// (cv::solvePnP's P3P rule).
//
// OpenCV 3's solver for that flag is Gao, Hou, Tang, Cheng, "Complete solution classification for the
// perspective-three-point problem" (PAMI 25(8), 2003), and so is this one: with X, Y, Z the distances
// from the camera centre to the three points and x = X/Z, y = Y/Z, the cosine-law equations reduce to
// one quartic in x with closed-form coefficients, y follows rationally, the quartic is solved in closed
// form (Ferrari split through one root of the resolvent cubic, Cardano or the trigonometric form) and
// the pose is the absolute orientation of the two triangles (Horn's unit quaternion).
// The closed-form roots are part of the reference's RESULT: they are not polished, so near a multiple
// root they carry errors many orders above rounding, and which hypothesis then collects the most
// inliers at a 0.1 px threshold depends on them.  The quartic's coefficients and the root formulas
// therefore keep the operation order of the published closed forms (every product and sum rounds where
// theirs does); `polish` > 0 adds that many Newton steps per root, the refinement later OpenCV releases
// apply.  What is well conditioned (the alignment) is written for one thread per hypothesis: fixed trip
// counts, no recursion, no heap, everything in registers or a few dozen doubles of scratch.
// Everything is double precision and usable from host and device code.
#pragma once

#include <math.h>

#ifdef __HIPCC__
#define KTLVO_HD __host__ __device__ inline
#else
#define KTLVO_HD inline
#endif

namespace ktlvo {
namespace p3p {

struct Pose {
    double R[9]; // row-major, camera = R * world + t
    double t[3];
};

// One root of the monic resolvent cubic y^3 + b y^2 + c y + d: with three real roots the one the
// trigonometric form gives at theta / 3, otherwise the real one by Cardano.  (2 R)^(1/3) of a negative
// R is NaN, as with the library pow the closed form is written on; the quartic then has no roots.
KTLVO_HD double resolvent_root(double b, double c, double d) {
    const double b2 = b * b;
    const double Q = (3 * c - b2) / 9;
    const double R = (9 * b * c - 27 * d - 2 * b * b2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_3 = (1. / 3.) * b;
    if (Q == 0) return R == 0 ? -b_3 : pow(2 * R, 1 / 3.0) - b_3;
    if (D <= 0) {
        const double theta = acos(R / sqrt(-Q3));
        return 2 * sqrt(-Q) * cos(theta / 3.0) - b_3;
    }
    const double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    return AD + BD - b_3;
}

// Real roots of a x^4 + b x^3 + c x^2 + d x + e, a != 0, in closed form (Ferrari); multiple roots are
// listed as often as the two quadratics yield them.
KTLVO_HD int quartic_roots(double a, double b, double c, double d, double e, double x[4]) {
    const double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    const double r0 = resolvent_root(-c, d * b - 4 * e, 4 * c * e - d * d - b2 * e);
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = sqrt(R2);
    const double inv_R = 1. / R;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) {
            D2 = E2 = -1;
        } else {
            const double sq = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sq;
            E2 = D2 - 4 * sq;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    int n = 0;
    if (D2 >= 0) {
        const double Dq = sqrt(D2);
        x[0] = R_2 + 0.5 * Dq - b_4;
        x[1] = x[0] - Dq;
        n = 2;
    }
    if (E2 >= 0) {
        const double Eq = sqrt(E2);
        x[n] = -R_2 + 0.5 * Eq - b_4;
        x[n + 1] = x[n] - Eq;
        n += 2;
    }
    return n;
}

// Gao et al.: the distances (X, Y, Z) from the camera centre to the three points, up to four triples.
// dist = |P1 P2|, |P0 P2|, |P0 P1|; cosv = the cosines between the bearings of the same pairs.
KTLVO_HD int gao_distances(const double dist[3], const double cosv[3], int polish, double len[4][3]) {
    const double p = cosv[0] * 2, q = cosv[1] * 2, r = cosv[2] * 2;
    const double inv_d22 = 1. / (dist[2] * dist[2]);
    const double a = inv_d22 * (dist[0] * dist[0]);
    const double b = inv_d22 * (dist[1] * dist[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0; // the four points (camera centre included) are coplanar
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double xs[4];
    const int n = quartic_roots(A, B, C, D, E, xs);
    for (int i = 0; i < n; i++)
        for (int it = 0; it < polish; it++) {
            const double x = xs[i];
            const double f = (((A * x + B) * x + C) * x + D) * x + E, df = ((4 * A * x + 3 * B) * x + 2 * C) * x + D;
            if (df == 0) break;
            xs[i] = x - f / df;
        }
    // y = b1(x) / b0; the sub-expressions of b1 that do not depend on the root are formed once, and the sums that
    // run across them keep their order
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
    const double k2 = 1 - a - b, k1 = q * a - q;
    const double c3 = r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1);
    const double c2 = r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2);
    const double c1 = r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
                      r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) + pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b);
    const double t1 = 2 * r3q * (a_2 - b - a2 + ab - 1);
    const double t2 = pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2);
    const double t3 = p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1));
    int ns = 0;
    for (int i = 0; i < n; i++) {
        const double x = xs[i];
        if (x <= 0) continue;
        const double x2 = x * x;
        const double b1 = (k2 * x2 + k1 * x + 1 - a + b) * ((c3 * x + c2) * x2 + c1 * x + t1 + t2 + t3);
        if (b1 <= 0) continue;
        const double y = inv_b0 * b1;
        const double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        const double Z = dist[2] / sqrt(v);
        len[ns][0] = x * Z;
        len[ns][1] = y * Z;
        len[ns][2] = Z;
        ns++;
    }
    return ns;
}

// Unit eigenvector of the largest eigenvalue of a symmetric 4x4 matrix (upper triangle of N used): cyclic Jacobi
// rotations, a fixed number of sweeps (convergence is quadratic: after five sweeps the off-diagonal mass of a
// 4x4 matrix is below rounding, further sweeps change nothing).
KTLVO_HD void top_eigenvector4(double N[4][4], double q[4]) {
    double U[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) U[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 8; sweep++)
        for (int pi = 0; pi < 3; pi++)
            for (int qi = pi + 1; qi < 4; qi++) {
                const double apq = N[pi][qi];
                if (apq == 0) continue;
                const double theta = (N[qi][qi] - N[pi][pi]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) {
                    const double akp = N[k][pi], akq = N[k][qi];
                    N[k][pi] = c * akp - s * akq;
                    N[k][qi] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = N[pi][k], aqk = N[qi][k];
                    N[pi][k] = c * apk - s * aqk;
                    N[qi][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double ukp = U[k][pi], ukq = U[k][qi];
                    U[k][pi] = c * ukp - s * ukq;
                    U[k][qi] = s * ukp + c * ukq;
                }
            }
    int im = 0;
    for (int i = 1; i < 4; i++)
        if (N[i][i] > N[im][im]) im = i;
    for (int i = 0; i < 4; i++) q[i] = U[i][im];
}

// determinant of the 3x3 matrix left of rows (r0, r1, r2) and columns (c0, c1, c2) of M
KTLVO_HD double minor3(const double M[4][4], int r0, int r1, int r2, int c0, int c1, int c2) {
    return M[r0][c0] * (M[r1][c1] * M[r2][c2] - M[r1][c2] * M[r2][c1]) -
           M[r0][c1] * (M[r1][c0] * M[r2][c2] - M[r1][c2] * M[r2][c0]) +
           M[r0][c2] * (M[r1][c0] * M[r2][c1] - M[r1][c1] * M[r2][c0]);
}

// The same eigenvector without the rotations, for the well-conditioned case (which is the rule): the largest root of
// the characteristic polynomial x^4 + c2 x^2 + c1 x + c0 of the traceless matrix by Newton steps from an upper bound
// (they descend monotonically onto it), then the eigenvector as the largest column of adj(N - x I), which is a
// multiple of q q^T.  Accepted only with a certificate: the residual |N q - x q| over the gap to the next eigenvalue
// bounds the eigenvector's error, and the gap is at least P'(x) / (2 x)^2 because the other two factors of
// P'(x) = (x - x2)(x - x3)(x - x4) are at most 2 x each.  Returns false where that bound is not far below what the pose
// is used for (thin triangles, where the top two eigenvalues approach each other): the caller then takes the rotations.
// About twenty times cheaper than the Jacobi sweeps, which is what the host path of a 300-hypothesis RANSAC spends
// its time on.
KTLVO_HD bool top_eigenvector4_certified(const double N[4][4], double upper_bound, double q[4]) {
    double f2 = 0; // c2 = -(sum of squares of N) / 2 for a traceless symmetric matrix
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) f2 += N[i][j] * N[i][j];
    const double c2 = -0.5 * f2;
    // c1 = -(sum of the principal 3x3 minors), c0 = det N
    const double c1 = -(minor3(N, 1, 2, 3, 1, 2, 3) + minor3(N, 0, 2, 3, 0, 2, 3) + minor3(N, 0, 1, 3, 0, 1, 3) + minor3(N, 0, 1, 2, 0, 1, 2));
    const double c0 = N[0][0] * minor3(N, 1, 2, 3, 1, 2, 3) - N[0][1] * minor3(N, 1, 2, 3, 0, 2, 3) +
                      N[0][2] * minor3(N, 1, 2, 3, 0, 1, 3) - N[0][3] * minor3(N, 1, 2, 3, 0, 1, 2);
    double x = upper_bound;
    if (!(x > 0)) return false;
    double dP = 0;
    for (int it = 0; it < 32; it++) {
        const double P = ((x * x + c2) * x + c1) * x + c0;
        dP = (4 * x * x + 2 * c2) * x + c1;
        if (!(dP > 0)) return false;
        const double step = P / dP;
        x -= step;
        if (fabs(step) <= 1e-16 * x) break;
    }
    double M[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) M[i][j] = N[i][j] - (i == j ? x : 0.0);
    // diagonal cofactors pick the column; its four cofactors are the eigenvector up to scale
    const int o[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    int col = 0;
    double best = 0;
    for (int k = 0; k < 4; k++) {
        const double d = fabs(minor3(M, o[k][0], o[k][1], o[k][2], o[k][0], o[k][1], o[k][2]));
        if (d > best) { best = d; col = k; }
    }
    if (!(best > 0)) return false;
    double nrm = 0;
    for (int i = 0; i < 4; i++) { // adj(M)[i][col] = (-1)^(i + col) * minor of M without row col and column i
        const double m = minor3(M, o[col][0], o[col][1], o[col][2], o[i][0], o[i][1], o[i][2]);
        q[i] = ((i + col) & 1) ? -m : m;
        nrm += q[i] * q[i];
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0)) return false;
    for (int i = 0; i < 4; i++) q[i] /= nrm;
    double res = 0;
    for (int i = 0; i < 4; i++) {
        const double r = N[i][0] * q[0] + N[i][1] * q[1] + N[i][2] * q[2] + N[i][3] * q[3] - x * q[i];
        res += r * r;
    }
    // error of q <= residual / gap, gap >= P'(x) / (2 x)^2
    return sqrt(res) * 4 * x * x <= 1e-12 * dP;
}

// Horn's absolute orientation of three point pairs: cam_i = R world_i + t in the least-squares sense.
KTLVO_HD void align3(const double cam[3][3], const double world[3][3], Pose &P) {
    double Cc[3], Cw[3];
    for (int j = 0; j < 3; j++) {
        Cc[j] = (cam[0][j] + cam[1][j] + cam[2][j]) / 3;
        Cw[j] = (world[0][j] + world[1][j] + world[2][j]) / 3;
    }
    double s[3][3]; // cross-covariance of world (rows) and camera (columns) coordinates
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            s[i][j] = (world[0][i] * cam[0][j] + world[1][i] * cam[1][j] + world[2][i] * cam[2][j]) / 3 - Cc[j] * Cw[i];
    double N[4][4];
    N[0][0] = s[0][0] + s[1][1] + s[2][2];
    N[1][1] = s[0][0] - s[1][1] - s[2][2];
    N[2][2] = s[1][1] - s[2][2] - s[0][0];
    N[3][3] = s[2][2] - s[0][0] - s[1][1];
    N[1][0] = N[0][1] = s[1][2] - s[2][1];
    N[2][0] = N[0][2] = s[2][0] - s[0][2];
    N[3][0] = N[0][3] = s[0][1] - s[1][0];
    N[2][1] = N[1][2] = s[1][0] + s[0][1];
    N[3][1] = N[1][3] = s[2][0] + s[0][2];
    N[3][2] = N[2][3] = s[2][1] + s[1][2];
    // the largest eigenvalue is at most half the summed spreads of the two point sets (both in S's 1/3 scaling)
    double spread = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double dc = cam[i][j] - Cc[j], dw = world[i][j] - Cw[j];
            spread += dc * dc + dw * dw;
        }
    double q[4];
    if (!top_eigenvector4_certified(N, spread / 6, q)) top_eigenvector4(N, q);
    const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double *R = P.R;
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q[1] * q[2] - q[0] * q[3]); R[2] = 2. * (q[1] * q[3] + q[0] * q[2]);
    R[3] = 2. * (q[1] * q[2] + q[0] * q[3]); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q[2] * q[3] - q[0] * q[1]);
    R[6] = 2. * (q[1] * q[3] - q[0] * q[2]); R[7] = 2. * (q[2] * q[3] + q[0] * q[1]); R[8] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) P.t[i] = Cc[i] - (R[3 * i] * Cw[0] + R[3 * i + 1] * Cw[1] + R[3 * i + 2] * Cw[2]);
}

// X: three world points (9), uv: their pixels (6), K = fx, fy, cx, cy. Returns the pose count.
KTLVO_HD int solve3(const double X[9], const double uv[6], const double K[4], int polish, Pose out[4]) {
    const double inv_fx = 1. / K[0], inv_fy = 1. / K[1], cx_fx = K[2] / K[0], cy_fy = K[3] / K[1];
    double m[3][3], world[3][3]; // unit bearings, world points
    for (int i = 0; i < 3; i++) {
        const double mu = inv_fx * uv[2 * i] - cx_fx, mv = inv_fy * uv[2 * i + 1] - cy_fy;
        const double mk = 1. / sqrt(mu * mu + mv * mv + 1);
        m[i][0] = mu * mk; m[i][1] = mv * mk; m[i][2] = mk;
        world[i][0] = X[3 * i]; world[i][1] = X[3 * i + 1]; world[i][2] = X[3 * i + 2];
    }
    double dist[3], cosv[3]; // pairs (1,2), (0,2), (0,1)
    for (int k = 0; k < 3; k++) {
        const int i = k == 0 ? 1 : 0, j = k == 2 ? 1 : 2;
        const double dx = world[i][0] - world[j][0], dy = world[i][1] - world[j][1], dz = world[i][2] - world[j][2];
        dist[k] = sqrt(dx * dx + dy * dy + dz * dz);
        cosv[k] = m[i][0] * m[j][0] + m[i][1] * m[j][1] + m[i][2] * m[j][2];
    }
    double len[4][3];
    const int n = gao_distances(dist, cosv, polish, len);
    for (int s = 0; s < n; s++) {
        double cam[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cam[i][j] = len[s][i] * m[i][j];
        align3(cam, world, out[s]);
    }
    return n;
}

// Four correspondences: candidates from the first three, the fourth picks the one with the
// smallest reprojection error.
KTLVO_HD bool solve4(const double X[12], const double uv[8], const double K[4], int polish, Pose &best) {
    Pose cand[4];
    const int n = solve3(X, uv, K, polish, cand);
    if (n == 0) return false;
    int ns = 0;
    double min_reproj = 0;
    for (int i = 0; i < n; i++) {
        const double *R = cand[i].R, *t = cand[i].t;
        const double Xp = R[0] * X[9] + R[1] * X[10] + R[2] * X[11] + t[0];
        const double Yp = R[3] * X[9] + R[4] * X[10] + R[5] * X[11] + t[1];
        const double Zp = R[6] * X[9] + R[7] * X[10] + R[8] * X[11] + t[2];
        const double mu = K[2] + K[0] * Xp / Zp, mv = K[3] + K[1] * Yp / Zp;
        const double reproj = (mu - uv[6]) * (mu - uv[6]) + (mv - uv[7]) * (mv - uv[7]);
        if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
    }
    best = cand[ns];
    return true;
}

// cv::Rodrigues, both directions (matrix branch without OpenCV's SVD re-orthonormalisation:
// the matrices passed here are orthonormal to rounding).
KTLVO_HD void rodrigues_v2m(const double r_in[3], double R[9]) {
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < 2.220446049250313e-16) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    rx *= it; ry *= it; rz *= it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * r_x[i];
}

KTLVO_HD void rodrigues_m2v(const double R[9], double r[3]) {
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    const double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            x = y = z = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5; x = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; y = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    r[0] = x; r[1] = y; r[2] = z;
}

} // namespace p3p
} // namespace ktlvo
