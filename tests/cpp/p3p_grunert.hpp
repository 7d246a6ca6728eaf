// An independent P3P solver, test infrastructure only: the cross-check for the library's Gao solver
// (slam_amd/csrc/p3p.hpp) and for the oracle's restatement of it, both of which follow the published closed
// forms term by term.  This one shares nothing with them but the problem: unknowns are the distance RATIOS to
// the first point (Grunert / Fischler-Bolles family: s2 = u s1, s3 = v s1, u = N(v) / D(v) and one quartic in
// v), the quartic is solved through its depressed form with Newton-polished roots, and the pose comes from
// aligning the two triangles' orthonormal triads instead of Horn's quaternion.  Until the product took the
// reference's own solver (cv::solvePnPRansac with SOLVEPNP_P3P, visualOdometry.cpp:122-127) this was the
// product's minimal solver.
#pragma once

#include <math.h>


namespace grunert {

struct Pose {
    double R[9]; // row-major, camera = R * world + t
    double t[3];
};

static inline int quadratic(double a, double b, double c, double x[2]) {
    if (a == 0) {
        if (b == 0) return 0;
        x[0] = -c / b;
        return 1;
    }
    const double d = b * b - 4 * a * c;
    if (d < 0) return 0;
    const double sd = sqrt(d);
    const double q = -0.5 * (b + (b >= 0 ? sd : -sd));
    if (q == 0) { x[0] = x[1] = 0; return 2; }
    x[0] = q / a;
    x[1] = c / q;
    return 2;
}

// largest real root of x^3 + a x^2 + b x + c
static inline double cubic_top_root(double a, double b, double c) {
    const double a3 = a / 3.0;
    const double p = b - a * a3, q = 2 * a3 * a3 * a3 - a3 * b + c;
    const double disc = q * q / 4 + p * p * p / 27;
    double t;
    if (disc > 0) {
        const double sd = sqrt(disc);
        t = cbrt(-q / 2 + sd) + cbrt(-q / 2 - sd);
    } else if (p == 0) {
        t = cbrt(-q);
    } else {
        const double m = 2 * sqrt(-p / 3);
        double arg = 3 * q / (p * m);
        arg = arg < -1.0 ? -1.0 : arg > 1.0 ? 1.0 : arg;
        t = m * cos(acos(arg) / 3);
    }
    double x = t - a3;
    for (int i = 0; i < 3; i++) {
        const double f = ((x + a) * x + b) * x + c, df = (3 * x + 2 * a) * x + b;
        if (df == 0) break;
        x -= f / df;
    }
    return x;
}

// real roots of k[0] + k[1] x + ... + k[4] x^4 (Ferrari), polished by Newton steps
static inline int quartic(const double k[5], double roots[4]) {
    if (fabs(k[4]) < 1e-14 * (fabs(k[3]) + fabs(k[2]) + fabs(k[1]) + fabs(k[0]))) return 0;
    const double a = k[3] / k[4], b = k[2] / k[4], c = k[1] / k[4], d = k[0] / k[4];
    const double a2 = a * a;
    const double p = b - 3 * a2 / 8, q = c - a * b / 2 + a2 * a / 8;
    const double r = d - a * c / 4 + a2 * b / 16 - 3 * a2 * a2 / 256;
    double y[4];
    int n = 0;
    if (fabs(q) < 1e-14 * (1 + fabs(p) + fabs(r))) {
        double t[2];
        const int nt = quadratic(1, p, r, t);
        for (int i = 0; i < nt; i++)
            if (t[i] >= 0) {
                const double s = sqrt(t[i]);
                y[n++] = s;
                y[n++] = -s;
            }
    } else {
        const double m = cubic_top_root(p, (p * p - 4 * r) / 4, -q * q / 8);
        if (m <= 0) return 0;
        const double s2m = sqrt(2 * m);
        double t[2];
        int nt = quadratic(1, -s2m, p / 2 + m + q / (2 * s2m), t);
        for (int i = 0; i < nt; i++) y[n++] = t[i];
        nt = quadratic(1, s2m, p / 2 + m - q / (2 * s2m), t);
        for (int i = 0; i < nt; i++) y[n++] = t[i];
    }
    for (int i = 0; i < n; i++) {
        double x = y[i] - a / 4;
        for (int it = 0; it < 2; it++) {
            const double f = (((k[4] * x + k[3]) * x + k[2]) * x + k[1]) * x + k[0];
            const double df = ((4 * k[4] * x + 3 * k[3]) * x + 2 * k[2]) * x + k[1];
            if (df == 0) break;
            x -= f / df;
        }
        roots[i] = x;
    }
    return n;
}

static inline double dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline bool unit(double a[3]) {
    const double n = sqrt(dot(a, a));
    if (n < 1e-300) return false;
    a[0] /= n; a[1] /= n; a[2] /= n;
    return true;
}
// E = [e1 e2 e3] (columns): e1 along P1P2, e3 normal of the triangle, e2 = e3 x e1
static inline bool triad(const double P1[3], const double P2[3], const double P3[3], double E[9]) {
    double e1[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]};
    const double w[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]};
    double e2[3], e3[3];
    if (!unit(e1)) return false;
    cross(e1, w, e3);
    if (!unit(e3)) return false;
    cross(e3, e1, e2);
    for (int i = 0; i < 3; i++) { E[3 * i] = e1[i]; E[3 * i + 1] = e2[i]; E[3 * i + 2] = e3[i]; }
    return true;
}

// X: three world points (9), uv: their pixels (6), K = fx, fy, cx, cy. Returns the pose count.
static inline int solve3(const double X[9], const double uv[6], const double K[4], Pose out[4]) {
    double f[3][3];
    for (int i = 0; i < 3; i++) {
        f[i][0] = (uv[2 * i] - K[2]) / K[0];
        f[i][1] = (uv[2 * i + 1] - K[3]) / K[1];
        f[i][2] = 1.0;
        unit(f[i]);
    }
    const double *P1 = X, *P2 = X + 3, *P3 = X + 6;
    double d[3];
    d[0] = P2[0] - P3[0]; d[1] = P2[1] - P3[1]; d[2] = P2[2] - P3[2];
    const double a2 = dot(d, d);
    d[0] = P1[0] - P3[0]; d[1] = P1[1] - P3[1]; d[2] = P1[2] - P3[2];
    const double b2 = dot(d, d);
    d[0] = P1[0] - P2[0]; d[1] = P1[1] - P2[1]; d[2] = P1[2] - P2[2];
    const double c2 = dot(d, d);
    if (a2 < 1e-20 || b2 < 1e-20 || c2 < 1e-20) return 0;
    const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
    double Ew[9];
    if (!triad(P1, P2, P3, Ew)) return 0;

    // A(v) = 1 - 2 cb v + v^2; N(v) = (a2 - c2) A + b2 (1 - v^2); D(v) = 2 b2 (cg - ca v)
    // quartic: b2 (D^2 + N^2 - 2 cg N D) - c2 A D^2
    const double A[3] = {1, -2 * cb, 1};
    const double N[3] = {(a2 - c2) + b2, (a2 - c2) * A[1], (a2 - c2) - b2};
    const double D[2] = {2 * b2 * cg, -2 * b2 * ca};
    const double DD[3] = {D[0] * D[0], 2 * D[0] * D[1], D[1] * D[1]};
    double NN[5] = {0, 0, 0, 0, 0}, ND[4] = {0, 0, 0, 0}, ADD[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            NN[i + j] += N[i] * N[j];
            ADD[i + j] += A[i] * DD[j];
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) ND[i + j] += N[i] * D[j];
    double Q[5], qs = 0;
    for (int i = 0; i < 5; i++) {
        const double dd = i < 3 ? DD[i] : 0, nd = i < 4 ? ND[i] : 0;
        Q[i] = b2 * (dd + NN[i] - 2 * cg * nd) - c2 * ADD[i];
        qs = fabs(Q[i]) > qs ? fabs(Q[i]) : qs;
    }
    if (qs == 0) return 0;
    for (int i = 0; i < 5; i++) Q[i] /= qs;

    double roots[4];
    const int nr = quartic(Q, roots);
    int ns = 0;
    for (int ri = 0; ri < nr && ns < 4; ri++) {
        const double v = roots[ri];
        if (!(v > 0)) continue;
        const double Dv = D[0] + D[1] * v;
        if (fabs(Dv) < 1e-12 * b2) continue;
        const double u = (N[0] + (N[1] + N[2] * v) * v) / Dv;
        if (!(u > 0)) continue;
        const double Av = A[0] + (A[1] + A[2] * v) * v;
        if (!(Av > 0)) continue;
        const double s1 = sqrt(b2 / Av), s2 = u * s1, s3 = v * s1;
        const double C1[3] = {s1 * f[0][0], s1 * f[0][1], s1 * f[0][2]};
        const double C2[3] = {s2 * f[1][0], s2 * f[1][1], s2 * f[1][2]};
        const double C3[3] = {s3 * f[2][0], s3 * f[2][1], s3 * f[2][2]};
        double Ec[9];
        if (!triad(C1, C2, C3, Ec)) continue;
        Pose &P = out[ns];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Ec[3 * i + k] * Ew[3 * j + k];
                P.R[3 * i + j] = s;
            }
        for (int i = 0; i < 3; i++)
            P.t[i] = C1[i] - (P.R[3 * i] * P1[0] + P.R[3 * i + 1] * P1[1] + P.R[3 * i + 2] * P1[2]);
        bool dup = false; // a double root gives the same pose twice
        for (int q = 0; q < ns && !dup; q++) {
            double e = 0;
            for (int i = 0; i < 9; i++) e = fmax(e, fabs(out[q].R[i] - P.R[i]));
            for (int i = 0; i < 3; i++) e = fmax(e, fabs(out[q].t[i] - P.t[i]) / (1 + fabs(P.t[i])));
            dup = e < 1e-9;
        }
        if (!dup) ns++;
    }
    return ns;
}

// Four correspondences: candidates from the first three, the fourth picks the one with the
// smallest reprojection error.
static inline bool solve4(const double X[12], const double uv[8], const double K[4], Pose &best) {
    Pose cand[4];
    const int n = solve3(X, uv, K, cand);
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

} // namespace grunert
