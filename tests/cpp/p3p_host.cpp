// Host builds of two minimal pose solvers, exported with a C ABI so that CPU tests can compare them with the
// oracle's restatement of Gao's P3P without a GPU:
//   p3p_host_*     the library's solver (slam_amd/csrc/p3p.hpp compiles as plain C++); ktlvo_solve_pnp_ransac runs
//                  exactly this code for its hypotheses when ktlvo_config.device_p3p == 0;
//   p3p_grunert_*  an independent solver of another family (tests/cpp/p3p_grunert.hpp), the cross-check of both.
#include "../../slam_amd/csrc/p3p.hpp"
#include "p3p_grunert.hpp"

#include <string.h>

extern "C" int p3p_host_solve3(const double X[9], const double uv[6], const double K[4], int polish, double R[36],
                               double t[12]) {
    ktlvo::p3p::Pose out[4];
    const int n = ktlvo::p3p::solve3(X, uv, K, polish, out);
    for (int i = 0; i < n; i++) {
        memcpy(R + 9 * i, out[i].R, sizeof(out[i].R));
        memcpy(t + 3 * i, out[i].t, sizeof(out[i].t));
    }
    return n;
}

// model of one RANSAC hypothesis as the library forms it: P3P on four points, R -> rvec -> R round trip
extern "C" int p3p_host_model(const double X[12], const double uv[8], const double K[4], int polish, double Rt[12],
                              double rvec[3]) {
    ktlvo::p3p::Pose P;
    if (!ktlvo::p3p::solve4(X, uv, K, polish, P)) return 0;
    ktlvo::p3p::rodrigues_m2v(P.R, rvec);
    ktlvo::p3p::rodrigues_v2m(rvec, Rt);
    memcpy(Rt + 9, P.t, sizeof(P.t));
    return 1;
}

extern "C" int p3p_grunert_solve3(const double X[9], const double uv[6], const double K[4], double R[36], double t[12]) {
    grunert::Pose out[4];
    const int n = grunert::solve3(X, uv, K, out);
    for (int i = 0; i < n; i++) {
        memcpy(R + 9 * i, out[i].R, sizeof(out[i].R));
        memcpy(t + 3 * i, out[i].t, sizeof(out[i].t));
    }
    return n;
}

extern "C" int p3p_grunert_model(const double X[12], const double uv[8], const double K[4], double Rt[12], double rvec[3]) {
    grunert::Pose P;
    if (!grunert::solve4(X, uv, K, P)) return 0;
    ktlvo::p3p::rodrigues_m2v(P.R, rvec);
    ktlvo::p3p::rodrigues_v2m(rvec, Rt);
    memcpy(Rt + 9, P.t, sizeof(P.t));
    return 1;
}
