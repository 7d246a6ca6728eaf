/*
 * ktlvo_oracle_p3p.cpp -- CPU oracle for the minimal solver behind
 * cv::solvePnPRansac(..., SOLVEPNP_P3P) as the reference calls it
 * (visualOdometry.cpp:122-127 selects the flag, :206-208 is the call).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED: see ktlvo_oracle.h.  Build with -ffp-contract=off.
 *
 * The solver lives in OpenCV 3.x calib3d (modules/calib3d/src/p3p.cpp, polynom_solver.cpp), which
 * is not under /root/reference and not installed here; this file restates the published algorithm
 * it implements, X.-S. Gao, X.-R. Hou, J. Tang, H.-F. Cheng, "Complete solution classification for
 * the perspective-three-point problem", IEEE PAMI 25(8), 2003:
 *   1. unit bearing vectors of the three pixels, pairwise cosines, pairwise world distances;
 *   2. with X, Y, Z the distances camera -> points 0, 1, 2 and x = X/Z, y = Y/Z, the three cosine-law
 *      equations reduce (Wu-Ritt zero decomposition) to one quartic in x whose coefficients are
 *      closed forms in a = |P1P2|^2/|P0P1|^2, b = |P0P2|^2/|P0P1|^2 and p, q, r = twice the cosines;
 *      y follows rationally as b1(x)/b0;
 *   3. the quartic is solved in closed form (Ferrari: resolvent cubic by Cardano / the
 *      trigonometric form, then two quadratics); no iterative polishing;
 *   4. each admissible (x, y) gives the three camera-frame points; the pose is the absolute
 *      orientation of the two triangles by Horn's unit-quaternion method (largest eigenvector of the
 *      4x4 matrix built from the cross-covariance, found with Jacobi rotations).
 * This is deliberately a different formulation, root finder and alignment from the library's own
 * solver (slam_amd/csrc/p3p.hpp: distance ratios relative to the first point, depressed-quartic
 * Ferrari with Newton polishing, triad alignment), so that agreement between the two is evidence
 * and not a tautology.  The closed forms of step 2 were checked numerically against the resultant of
 * the two quadratics they come from (tests/test_oracle_kat.py).
 */
#include "ktlvo_oracle.h"

#include <cmath>
#include <cstring>

namespace {

/* 0: roots exactly as the closed forms deliver them (OpenCV 3.x); n > 0: n Newton steps on the quartic
 * per root (what later OpenCV releases added to this solver for accuracy).  Test-only switch. */
int g_polish_steps = 0;

/* ---- closed-form real roots (multiplicities are not merged; that is harmless to the caller) ---- */
int roots2(double a, double b, double c, double &x1, double &x2) {
    const double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) {
        x1 = -b * inv_2a;
        x2 = x1;
        return 1;
    }
    const double sq = std::sqrt(delta);
    x1 = (-b + sq) * inv_2a;
    x2 = (-b - sq) * inv_2a;
    return 2;
}

int roots3(double a, double b, double c, double d, double &x0, double &x1, double &x2) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return roots2(b, c, d, x0, x1);
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) {
            x0 = x1 = x2 = -b_a_3;
            return 3;
        }
        x0 = std::pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) { /* three real roots: trigonometric form */
        const double theta = std::acos(R / std::sqrt(-Q3));
        const double sqrt_Q = std::sqrt(-Q);
        x0 = 2 * sqrt_Q * std::cos(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * std::cos((theta + 2 * M_PI) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * std::cos((theta + 4 * M_PI) / 3.0) - b_a_3;
        return 3;
    }
    /* one real root: Cardano */
    const double AD = std::pow(std::fabs(R) + std::sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

int roots4(double a, double b, double c, double d, double e, double x[4]) {
    if (a == 0) {
        x[3] = 0;
        return roots3(b, c, d, e, x[0], x[1], x[2]);
    }
    const double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    /* resolvent cubic; its first root r0 splits the quartic into two quadratics */
    double r0, r1, r2;
    if (roots3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2) == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = std::sqrt(R2);
    const double inv_R = 1. / R;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) {
            D2 = E2 = -1;
        } else {
            const double sq = std::sqrt(temp);
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
        const double Dq = std::sqrt(D2);
        x[0] = R_2 + 0.5 * Dq - b_4;
        x[1] = x[0] - Dq;
        n = 2;
    }
    if (E2 >= 0) {
        const double Eq = std::sqrt(E2);
        x[n] = -R_2 + 0.5 * Eq - b_4;
        x[n + 1] = x[n] - Eq;
        n += 2;
    }
    return n;
}

/* ---- Gao et al.: the three camera-to-point distances, up to four triples ----------------------------
 * dist[0] = |P1 P2|, dist[1] = |P0 P2|, dist[2] = |P0 P1|; cosv[i] likewise between the bearings. */
int gao_lengths(double len[4][3], const double dist[3], const double cosv[3]) {
    const double p = cosv[0] * 2, q = cosv[1] * 2, r = cosv[2] * 2;
    const double inv_d22 = 1. / (dist[2] * dist[2]);
    const double a = inv_d22 * (dist[0] * dist[0]);
    const double b = inv_d22 * (dist[1] * dist[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0; /* the camera centre lies on the danger cylinder's limit: no finite solution set */
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
    const int n = roots4(A, B, C, D, E, xs);
    if (n == 0) return 0;
    for (int i = 0; i < n; i++)
        for (int it = 0; it < g_polish_steps; it++) {
            const double x = xs[i];
            const double f = (((A * x + B) * x + C) * x + D) * x + E, df = ((4 * A * x + 3 * B) * x + 2 * C) * x + D;
            if (df == 0) break;
            xs[i] = x - f / df;
        }
    int ns = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        const double x = xs[i];
        if (x <= 0) continue;
        const double x2 = x * x;
        const double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        const double y = inv_b0 * b1;
        const double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        const double Z = dist[2] / std::sqrt(v);
        len[ns][0] = x * Z;
        len[ns][1] = y * Z;
        len[ns][2] = Z;
        ns++;
    }
    return ns;
}

/* ---- symmetric 4x4 eigen decomposition by cyclic Jacobi rotations: A = U diag(ev) U^T ---------------- */
void jacobi4(double A[4][4], double ev[4], double U[4][4]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) U[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0;
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 4; j++) off += std::fabs(A[i][j]);
        if (off == 0) break;
        for (int pi = 0; pi < 3; pi++)
            for (int qi = pi + 1; qi < 4; qi++) {
                const double apq = A[pi][qi];
                if (apq == 0) continue;
                const double theta = (A[qi][qi] - A[pi][pi]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) { /* A <- A G */
                    const double akp = A[k][pi], akq = A[k][qi];
                    A[k][pi] = c * akp - s * akq;
                    A[k][qi] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) { /* A <- G^T A */
                    const double apk = A[pi][k], aqk = A[qi][k];
                    A[pi][k] = c * apk - s * aqk;
                    A[qi][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) { /* U <- U G */
                    const double ukp = U[k][pi], ukq = U[k][qi];
                    U[k][pi] = c * ukp - s * ukq;
                    U[k][qi] = s * ukp + c * ukq;
                }
            }
    }
    for (int i = 0; i < 4; i++) ev[i] = A[i][i];
}

/* ---- Horn's absolute orientation: R, T with cam_i = R world_i + T for three point pairs ------------- */
void horn_align(const double cam[3][3], const double world[3][3], double R[9], double T[3]) {
    double Cc[3], Cw[3];
    for (int j = 0; j < 3; j++) {
        Cc[j] = (cam[0][j] + cam[1][j] + cam[2][j]) / 3;
        Cw[j] = (world[0][j] + world[1][j] + world[2][j]) / 3;
    }
    double s[3][3]; /* s[i][j] = mean(world_i * cam_j) - mean(world_i) mean(cam_j) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            s[i][j] = (world[0][i] * cam[0][j] + world[1][i] * cam[1][j] + world[2][i] * cam[2][j]) / 3 - Cc[j] * Cw[i];
    double N[4][4], ev[4], U[4][4];
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
    jacobi4(N, ev, U);
    int im = 0;
    for (int i = 1; i < 4; i++)
        if (ev[i] > ev[im]) im = i;
    const double q0 = U[0][im], q1 = U[1][im], q2 = U[2][im], q3 = U[3][im];
    const double q02 = q0 * q0, q12 = q1 * q1, q22 = q2 * q2, q32 = q3 * q3;
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1 * q2 - q0 * q3); R[2] = 2. * (q1 * q3 + q0 * q2);
    R[3] = 2. * (q1 * q2 + q0 * q3); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2 * q3 - q0 * q1);
    R[6] = 2. * (q1 * q3 - q0 * q2); R[7] = 2. * (q2 * q3 + q0 * q1); R[8] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = Cc[i] - (R[3 * i] * Cw[0] + R[3 * i + 1] * Cw[1] + R[3 * i + 2] * Cw[2]);
}

}  // namespace

/* Test hook: the quartic's coefficients (A..E, highest power first) for given a, b, p, q, r, so that
 * a test can compare the closed forms with the resultant they are derived from. */
extern "C" void orc_gao_quartic(double a, double b, double p, double q, double r, double coef[5]) {
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, pr = p * r, pqr = q * pr, ab = a * b;
    coef[0] = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - 2 * a;
    coef[1] = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + 4 * a) + pr * (b - b2 + ab);
    coef[2] = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - 2 * a) * (2 + q2) + 2;
    coef[3] = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + 4 * a - 2);
    coef[4] = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
}

extern "C" void orc_p3p_set_polish(int newton_steps) { g_polish_steps = newton_steps < 0 ? 0 : newton_steps; }

extern "C" int orc_roots4(const double coef[5], double roots[4]) {
    return roots4(coef[0], coef[1], coef[2], coef[3], coef[4], roots);
}

extern "C" int orc_p3p(const double X[9], const double uv[6], const double K[4], double Rout[36], double tout[12]) {
    const double inv_fx = 1. / K[0], inv_fy = 1. / K[1], cx_fx = K[2] / K[0], cy_fy = K[3] / K[1];
    double m[3][3]; /* unit bearing vectors */
    for (int i = 0; i < 3; i++) {
        const double mu = inv_fx * uv[2 * i] - cx_fx, mv = inv_fy * uv[2 * i + 1] - cy_fy;
        const double mk = 1. / std::sqrt(mu * mu + mv * mv + 1);
        m[i][0] = mu * mk; m[i][1] = mv * mk; m[i][2] = mk;
    }
    double world[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) world[i][j] = X[3 * i + j];
    auto dist = [&](int i, int j) {
        const double dx = world[i][0] - world[j][0], dy = world[i][1] - world[j][1], dz = world[i][2] - world[j][2];
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    auto cosb = [&](int i, int j) { return m[i][0] * m[j][0] + m[i][1] * m[j][1] + m[i][2] * m[j][2]; };
    const double distances[3] = {dist(1, 2), dist(0, 2), dist(0, 1)};
    const double cosines[3] = {cosb(1, 2), cosb(0, 2), cosb(0, 1)};
    double len[4][3];
    const int n = gao_lengths(len, distances, cosines);
    for (int s = 0; s < n; s++) {
        double cam[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cam[i][j] = len[s][i] * m[i][j];
        horn_align(cam, world, Rout + 9 * s, tout + 3 * s);
    }
    return n;
}

/* cv::solvePnP(..., SOLVEPNP_P3P) on four correspondences: the candidates of the first three, the
 * fourth picks the one that reprojects it best. */
extern "C" int orc_solve_p3p4(const double X[12], const double uv[8], const double K[4], double R[9], double t[3]) {
    double Rs[36], ts[12];
    const int n = orc_p3p(X, uv, K, Rs, ts);
    if (n == 0) return 0;
    int best = 0;
    double min_reproj = 0;
    for (int i = 0; i < n; i++) {
        const double *Ri = Rs + 9 * i, *ti = ts + 3 * i;
        const double Xp = Ri[0] * X[9] + Ri[1] * X[10] + Ri[2] * X[11] + ti[0];
        const double Yp = Ri[3] * X[9] + Ri[4] * X[10] + Ri[5] * X[11] + ti[1];
        const double Zp = Ri[6] * X[9] + Ri[7] * X[10] + Ri[8] * X[11] + ti[2];
        const double mu = K[2] + K[0] * Xp / Zp, mv = K[3] + K[1] * Yp / Zp;
        const double reproj = (mu - uv[6]) * (mu - uv[6]) + (mv - uv[7]) * (mv - uv[7]);
        if (i == 0 || min_reproj > reproj) {
            best = i;
            min_reproj = reproj;
        }
    }
    memcpy(R, Rs + 9 * best, 9 * sizeof(double));
    memcpy(t, ts + 3 * best, 3 * sizeof(double));
    return 1;
}
