/*
 * srukf_oracle.c — CPU restatement of CV-MonoSLAM's SRUKF predict/update path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cv-monoslam_amd/ (the product) may link, import or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be compiled here (MFC + OpenCV 2.4.3 + GSL 1.8, Windows only),
 * so this restatement cannot be checked against outputs of the reference itself.  It is pinned
 * only against an independent numpy restatement (cv-monoslam_amd/synth.py, tests/) and
 * against numpy/scipy factorizations on convention-independent quantities (R^T R, L D L^T).
 *
 * Every function cites the reference lines it follows ("SLAM.cpp" = /root/reference/MonoSLAM/SLAM.cpp).
 * Third-party arithmetic that is NOT under /root/reference is restated from its published
 * algorithm and named where used:
 *   - GSL (GnuWin32 1.8, README.md:13): gsl_linalg_QR_decomp -> householder_transform /
 *     householder_hm (linalg/qr.c, linalg/householder.c), unblocked Householder,
 *     beta = -sign(alpha)*hypot(alpha, |x|), tau = 0 when the sub-column is already zero.
 *   - OpenCV 2.4.3 (README.md:12): Mat::inv() on 2x2 / 3x3 (closed-form cofactor inverse in
 *     cv::invert), addWeighted (a*alpha + b*beta + gamma), Mat*Mat (plain gemm), minMaxLoc.
 *   - Visual C++ 2010 runtime (MonoSLAM.sln:2-3 "Format Version 11.00 / Visual Studio 2010", MonoSLAM.vcxproj:2 ToolsVersion 4.0): every pow() on the path has an int
 *     literal exponent (SLAM.cpp:1053,1056,3184,3190,3191,3195,3230), which in that compiler's <math.h> selects the
 *     overload pow(double, int) -> _Pow_int: repeated multiplication by squaring, NOT the transcendental pow.
 *     Restated as pow_di below (recalled from the VC10 header; the header is not in /root/reference).  sin / cos
 *     come from the same runtime and are not reproducible bit for bit by any other libm: the filter path uses
 *     this host's libm (results are compared with tolerances), the byte-producing wrapPatch uses correctly
 *     rounded values (libquadmath, rounded once) so that its uchar output is defined independently of a libm.
 *
 * Structure follows the reference, not a textbook SRUKF: deviations from sigma_0 (not the
 * mean) in every QR, no centre-weight term in the QRs, x/y swap in the projection, Mt/Qt used
 * as square roots, per-measurement-column  S <- gmw(S^T S - u u^T)  (SURVEY.md App. A.7).
 *
 * Build:  make -C oracle        (gcc -O2 -ffp-contract=off, no dependencies)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <quadmath.h>
#include "../include/srukf.h"

#define ORC_API __attribute__((visibility("default")))

typedef struct orc_state {
    srukf_params p;
    int N, n;            /* landmarks, state dim 6N+4                        */
    double *X;           /* n            m_X_k   SLAM.h:271                  */
    double *S;           /* n*n upper    m_S_k   SLAM.h:272                  */
    /* per-frame scratch (reallocated by the reference every frame, SLAM.cpp:1442,1621) */
    int Na, L;           /* Na = n+5 (SLAM.cpp:1432), L = 2Na+1              */
    double *sigma;       /* Na*L row-major, column = sigma point  m_sigma   SLAM.h:284 */
    double *Z;           /* 2N*L row-major             m_sigma_allPixel     SLAM.h:285 */
    double *h;           /* 2N                         m_allPredictSet      SLAM.h:281 */
    double *Si;          /* N*4 (2x2 row-major each)   PointsMap::Si        SLAM.h:57  */
    int    *visible;     /* N                          PointsMap::isVisible SLAM.h:50  */
    double Ut[3], Mt[3], Qt[2];     /* SLAM.h:287-289; Mt, Qt diagonal                 */
    double wm0, wc0, wi, wi_sr, gamma, wm0_sr, wc0_sr;   /* SLAM.h:251-257            */
    int newton_early_exit;          /* 1: leave the 100-iteration loop once rd is a fixed point (bit-exact) */
    int frozen_center;              /* test knob, NOT the reference: 1 = the cross covariance centres on the state at the
                                       start of KalmanUpdate instead of the running m_X_k (SLAM.cpp:2030); quantifies
                                       what the running-state term is worth when wc0 != wm0 (tests only)            */
    const double *Xcenter;          /* state the cross covariance centres on (X, or the frozen copy)                 */
    long long clamp_eps, clamp_theta, pivots;            /* GMW statistics             */
} orc_state;

/* pow(double, int) of the Visual C++ 2010 <math.h> (_Pow_int): Z = 1; for each bit of |y| from the lowest:
 * if set Z *= X; X *= X.  pow_di(x, 2) = x*x, (x, 3) = x*(x*x), (x, 4) = (x*x)*(x*x), (x, 5) = x*((x*x)*(x*x)). */
static double pow_di(double x, int y)
{
    unsigned int n = y >= 0 ? (unsigned int)y : (unsigned int)(-y);
    double z = 1.0;
    for (;; x *= x) {
        if (n & 1u) z *= x;
        if ((n >>= 1) == 0) return y < 0 ? 1.0 / z : z;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* calculateSampleParameter, SLAM.cpp:1050-1103                                                */
ORC_API void orc_sample_parameter(int Na, int weight_type, double alpha, double beta, double out[7])
{
    double Kappa  = 0;                                           /* 1052 */
    double Lammda = pow_di(alpha, 2) * (Na + Kappa) - Na;           /* 1053 */
    double Gamma  = sqrt(Na + Lammda);                           /* 1054 */
    double s_wm0  = Lammda / (Na + Lammda);                      /* 1055 */
    double s_wc0  = s_wm0 + (1 - pow_di(alpha, 2) + beta);          /* 1056 */
    double s_wi   = 1.0 / (2 * (Na + Lammda));                   /* 1057 */
    double wm0, wm0_sr, wc0, wc0_sr, wi, wi_sr, gamma;
    switch (weight_type) {
    case 0:                                                      /* 1064-1075 */
        wm0 = 1.0 - Na / 3.0;  wm0_sr = sqrt(fabs(wm0));
        wc0 = 1.0 - Na / 3.0;  wc0_sr = sqrt(fabs(wm0));
        wi  = (1.0 - wc0) / (2 * Na);  wi_sr = sqrt(wi);
        gamma = sqrt(Na / (1.0 - wm0));
        break;
    case 1:                                                      /* 1077-1088 */
        gamma = Gamma; wm0 = s_wm0; wm0_sr = sqrt(fabs(s_wm0));
        wc0 = s_wc0; wc0_sr = sqrt(fabs(s_wc0)); wi = s_wi; wi_sr = sqrt(fabs(s_wi));
        break;
    default:                                                     /* 1090-1101 */
        gamma = sqrt(3.0 * Na / 2.0); wm0 = 1.0 / 3.0; wm0_sr = sqrt(wm0);
        wc0 = 1.0 / 3.0; wc0_sr = sqrt(wc0); wi = 1.0 / (3.0 * Na); wi_sr = sqrt(wi);
        break;
    }
    out[0] = wm0; out[1] = wc0; out[2] = wi; out[3] = wi_sr; out[4] = gamma; out[5] = wm0_sr; out[6] = wc0_sr;
}

static void set_weights(orc_state *st, int Na)
{
    double w[7];
    orc_sample_parameter(Na, st->p.weight_type, st->p.ut_alpha, st->p.ut_beta, w);
    st->wm0 = w[0]; st->wc0 = w[1]; st->wi = w[2]; st->wi_sr = w[3]; st->gamma = w[4];
    st->wm0_sr = w[5]; st->wc0_sr = w[6];
}

/* ------------------------------------------------------------------------------------------ */
/* GSLQrDecomposition, SLAM.cpp:2330-2353: R = first k rows of gsl_linalg_QR_decomp(A), upper
 * triangle only.  A is m x k row-major (copied, 3432-3444).  GSL 1.8 linalg/qr.c +
 * linalg/householder.c restated (un-vendored dependency):
 *   householder_transform(v): n==1 -> tau=0; xnorm=|v[1:]|; xnorm==0 -> tau=0;
 *       alpha=v0; beta=-sign(alpha)*hypot(alpha,xnorm); tau=(beta-alpha)/beta;
 *       v[1:] /= (alpha-beta); v0=beta
 *   householder_hm(tau,v,A): per column j: wj = A0j + sum_i Aij*vi; A0j -= tau*wj; Aij -= tau*vi*wj */
ORC_API void orc_qr_r(const double *A_in, int m, int k, double *R /* k*k */)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)m * k);
    memcpy(A, A_in, sizeof(double) * (size_t)m * k);
    double *wrow = (double *)malloc(sizeof(double) * (size_t)k);
    int steps = m < k ? m : k;
    for (int i = 0; i < steps; i++) {
        int len = m - i;
        double tau = 0.0;
        if (len > 1) {
            /* gsl_blas_dnrm2 on the sub-column below the diagonal (scaled 2-norm; the scaling
             * only guards over/underflow, the value is the plain Euclidean norm) */
            double scale = 0.0, ssq = 1.0;
            for (int r = i + 1; r < m; r++) {
                double x = A[(size_t)r * k + i];
                if (x != 0.0) {
                    double ax = fabs(x);
                    if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
                    else            { ssq += (ax / scale) * (ax / scale); }
                }
            }
            double xnorm = scale * sqrt(ssq);
            if (xnorm != 0.0) {
                double alpha = A[(size_t)i * k + i];
                double beta  = -(alpha >= 0.0 ? +1.0 : -1.0) * hypot(alpha, xnorm);
                tau = (beta - alpha) / beta;
                double s = alpha - beta;
                for (int r = i + 1; r < m; r++) A[(size_t)r * k + i] *= (1.0 / s);
                A[(size_t)i * k + i] = beta;
            }
        }
        if (tau != 0.0 && i + 1 < k) {
            /* householder_hm: per column j, wj = A0j + sum_r Arj*vr (r increasing), then
             * Aij -= tau*vi*wj.  Loops are ordered row-major (r outer) for speed; every wj still
             * accumulates its terms in the same r order as GSL's column loop, so results are
             * bit-identical. */
            for (int j = i + 1; j < k; j++) wrow[j] = A[(size_t)i * k + j];
            for (int r = i + 1; r < m; r++) {
                const double vr = A[(size_t)r * k + i];
                const double *Ar = A + (size_t)r * k;
                for (int j = i + 1; j < k; j++) wrow[j] += Ar[j] * vr;
            }
            for (int j = i + 1; j < k; j++) A[(size_t)i * k + j] -= tau * wrow[j];
            for (int r = i + 1; r < m; r++) {
                const double vr = A[(size_t)r * k + i];
                double *Ar = A + (size_t)r * k;
                for (int j = i + 1; j < k; j++) Ar[j] -= tau * vr * wrow[j];
            }
        }
    }
    memset(R, 0, sizeof(double) * (size_t)k * k);                 /* 2341 */
    for (int i = 0; i < k && i < m; i++)                          /* 2343-2349 */
        for (int j = i; j < k; j++) R[(size_t)i * k + j] = A[(size_t)i * k + j];
    free(A); free(wrow);
}

/* ------------------------------------------------------------------------------------------ */
/* modifiedCholeskyDecomposition, SLAM.cpp:2197-2327 (Gill-Murray-Wright, G+E = L D L^T).
 * G n*n row-major symmetric.  Outputs S = sqrt(D) L^T (2319-2321), optionally D (n) and L (n*n).
 * The dead `pneg` branch (2305-2317) computes an unused vector and is omitted.               */
ORC_API void orc_gmw(const double *G, int dim, double eps, double *S, double *D_out, double *L_out,
                     long long *n_eps_clamp, long long *n_theta_clamp)
{
    size_t nn = (size_t)dim * dim;
    double gamma_ = -INFINITY, zi = -INFINITY;
    for (int i = 0; i < dim; i++) if (G[(size_t)i * dim + i] > gamma_) gamma_ = G[(size_t)i * dim + i];   /* 2204 */
    for (int i = 0; i < dim; i++)                                                                     /* 2205: max of G - diag(diag G), diagonal entries are 0 */
        for (int j = 0; j < dim; j++) { double v = (i == j) ? 0.0 : G[(size_t)i * dim + j]; if (v > zi) zi = v; }
    double nu = fmax(1.0, sqrt((double)dim * dim - 1.0));                                             /* 2207-2208 */
    double beta2 = fmax(fmax(gamma_, zi / nu), 1e-15);                                                /* 2210-2211 */

    double *L = (double *)calloc(nn, sizeof(double));
    double *C = (double *)calloc(nn, sizeof(double));
    double *D = (double *)calloc(dim, sizeof(double));
    for (int i = 0; i < dim; i++) C[(size_t)i * dim + i] = G[(size_t)i * dim + i];                    /* 2217 */
    long long ce = 0, ct = 0;

    for (int j = 0; j < dim; j++) {
        /* row j of L: L[j,0:j] = C[j,0:j] / D[0:j]                       2224-2234 */
        for (int k = 0; k < j; k++) L[(size_t)j * dim + k] = C[(size_t)j * dim + k] / D[k];
        /* column j of C below the diagonal                                2237-2261 */
        if (j >= 1) {
            for (int i = j + 1; i < dim; i++) {
                double acc = 0.0;
                for (int k = 0; k < j; k++) acc += L[(size_t)j * dim + k] * C[(size_t)i * dim + k];
                C[(size_t)i * dim + j] = G[(size_t)i * dim + j] - acc;
            }
        } else {
            for (int i = 1; i < dim; i++) C[(size_t)i * dim + 0] = G[(size_t)i * dim + 0];
        }
        /* theta_j = max |C[j+1:,j]|                                       2264-2276 */
        double theta = 0.0;
        for (int i = j + 1; i < dim; i++) { double a = fabs(C[(size_t)i * dim + j]); if (a > theta) theta = a; }
        /* D_j = max(EPSILON, |C_jj|, theta^2/beta2)                       2279-2285 */
        double cjj = fabs(C[(size_t)j * dim + j]);
        double t2  = theta * theta / beta2;
        double d   = eps; int which = 0;
        if (cjj > d) { d = cjj; which = 1; }
        if (t2  > d) { d = t2;  which = 2; }
        D[j] = d;
        if (which == 0) ce++; else if (which == 2) ct++;
        /* C_ii -= C_ij^2 / D_j  (i > j)                                   2291-2295 */
        for (int i = j + 1; i < dim; i++)
            C[(size_t)i * dim + i] = C[(size_t)i * dim + i] - C[(size_t)i * dim + j] * C[(size_t)i * dim + j] / D[j];
    }
    for (int i = 0; i < dim; i++) L[(size_t)i * dim + i] = 1.0;                                       /* 2299-2302 */
    /* S = sqrt(D) * L^T                                                   2319-2321 */
    if (S) {
        memset(S, 0, nn * sizeof(double));
        for (int j = 0; j < dim; j++) { double sd = sqrt(D[j]); for (int i = j; i < dim; i++) S[(size_t)j * dim + i] = sd * L[(size_t)i * dim + j]; }
    }
    if (D_out) memcpy(D_out, D, sizeof(double) * dim);
    if (L_out) memcpy(L_out, L, sizeof(double) * nn);
    if (n_eps_clamp) *n_eps_clamp += ce;
    if (n_theta_clamp) *n_theta_clamp += ct;
    free(L); free(C); free(D);
}

/* ------------------------------------------------------------------------------------------ */
/* camera model helpers                                                                         */

/* getTransferMatrix, SLAM.cpp:1031-1037 */
static void transfer_matrix(double Rwc[9], double theta)
{
    Rwc[0] = cos(theta); Rwc[1] = -sin(theta); Rwc[2] = 0;
    Rwc[3] = sin(theta); Rwc[4] =  cos(theta); Rwc[5] = 0;
    Rwc[6] = 0;          Rwc[7] = 0;           Rwc[8] = 1;
}

/* getTransferMatrix with correctly rounded cos / sin (binary128 libquadmath value rounded once to double): used by
 * wrapPatch only, whose uchar output depends on the last bit of these two numbers (see the header). */
static void transfer_matrix_cr(double Rwc[9], double theta)
{
    const double c = (double)cosq((__float128)theta), s = (double)sinq((__float128)theta);
    Rwc[0] = c; Rwc[1] = -s; Rwc[2] = 0;
    Rwc[3] = s; Rwc[4] =  c; Rwc[5] = 0;
    Rwc[6] = 0; Rwc[7] = 0;  Rwc[8] = 1;
}

/* cv::Mat::inv() on a 3x3 CV_64F (SLAM.cpp:1643): OpenCV 2.4.3 cv::invert closed form
 * (determinant, then cofactors scaled by 1/det).  Un-vendored dependency, restated. */
static void inv3(const double a[9], double t[9])
{
    double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (d != 0.0) {
        d = 1.0 / d;
        t[0] = (a[4] * a[8] - a[5] * a[7]) * d;  t[1] = (a[2] * a[7] - a[1] * a[8]) * d;  t[2] = (a[1] * a[5] - a[2] * a[4]) * d;
        t[3] = (a[5] * a[6] - a[3] * a[8]) * d;  t[4] = (a[0] * a[8] - a[2] * a[6]) * d;  t[5] = (a[2] * a[3] - a[0] * a[5]) * d;
        t[6] = (a[3] * a[7] - a[4] * a[6]) * d;  t[7] = (a[1] * a[6] - a[0] * a[7]) * d;  t[8] = (a[0] * a[4] - a[1] * a[3]) * d;
    } else memset(t, 0, 9 * sizeof(double));
}

/* cv::Mat::inv() on a 2x2 CV_64F (SLAM.cpp:2077): OpenCV 2.4.3 closed form. */
static void inv2(const double a[4], double t[4])
{
    double d = a[0] * a[3] - a[1] * a[2];
    if (d != 0.0) {
        d = 1.0 / d;
        t[0] = a[3] * d; t[1] = -a[1] * d; t[2] = -a[2] * d; t[3] = a[0] * d;
    } else memset(t, 0, 4 * sizeof(double));
}

/* distortOnePointRW, SLAM.cpp:3177-3213 */
static void distort_rw(const orc_state *st, const srukf_params *p, double uvu_x, double uvu_y, double *uvd_x, double *uvd_y)
{
    double f, ff;
    double xu = (uvu_x - p->cam_cx) * p->cam_dx;                                  /* 3181 */
    double yu = (uvu_y - p->cam_cy) * p->cam_dy;                                  /* 3182 */
    double ru = sqrt(xu * xu + yu * yu);                                          /* 3183 */
    double rd = ru / (1 + p->cam_k1 * ru * ru + p->cam_k2 * pow_di(ru, 4));          /* 3184 */
    int iters = p->newton_iters;                                                  /* 3186 */
    double rd_prev = NAN;
    for (int i = 0; i < iters; i++) {                                             /* 3188-3193 */
        f  = rd + p->cam_k1 * pow_di(rd, 3) + p->cam_k2 * pow_di(rd, 5) - ru;
        ff = 1.0 + 3.0 * p->cam_k1 * rd * rd + 5.0 * p->cam_k2 * pow_di(rd, 4);
        double rd_new = rd - f / ff;
        if (st && st->newton_early_exit) {
            if (rd_new == rd) { rd = rd_new; break; }                             /* fixed point: later iterations are idempotent */
            /* 2-cycle between neighbouring doubles (the usual end of this iteration in floating point): the sequence alternates
             * from here on, so the value after the last iteration follows from the parity of the iterations left */
            if (rd_new == rd_prev) { if ((iters - i) & 1) rd = rd_new; break; }
        }
        rd_prev = rd;
        rd = rd_new;
    }
    double d = 1 + p->cam_k1 * rd * rd + p->cam_k2 * pow_di(rd, 4);                  /* 3195 */
    if (d == 0) d = p->epsilon;                                                   /* 3197-3198 */
    double xd = xu / d, yd = yu / d;                                              /* 3200-3201 */
    *uvd_x = p->cam_cx + xd / p->cam_dx;                                          /* 3203 */
    *uvd_y = p->cam_cy + yd / p->cam_dy;                                          /* 3204 */
    int vis = (*uvd_x >= 0) && (*uvd_x <= p->image_w) && (*uvd_y >= 0) && (*uvd_y <= p->image_h);   /* 3206 */
    if (!vis) { *uvd_x = 0; *uvd_y = 0; }                                         /* 3208-3212 */
}

/* undistortOnePointRW, SLAM.cpp:3224-3236 */
static void undistort_rw(const srukf_params *p, double uvd_x, double uvd_y, double *uvu_x, double *uvu_y)
{
    double xd = (uvd_x - p->cam_cx) * p->cam_dx;
    double yd = (uvd_y - p->cam_cy) * p->cam_dy;
    double rd = sqrt(xd * xd + yd * yd);
    double d  = 1 + p->cam_k1 * pow_di(rd, 2) + p->cam_k2 * pow_di(rd, 4);
    double xu = xd * d, yu = yd * d;
    *uvu_x = p->cam_cx + xu / p->cam_dx;
    *uvu_y = p->cam_cy + yu / p->cam_dy;
}

/* One projection: the body of the inner loop of passSigmaThroughMesaurementFunction,
 * SLAM.cpp:1662-1670 = coordinatesState2World (3250-3276) -> coordinatesWorld2Camera (3289-3292)
 * -> coordinatesCamera2Image (3324-3347) -> distortOnePointRW (3177-3213), with Rcw = Rwc.inv()
 * (1642-1643).  feat = (xi yi zi theta phi rho), pos = robot (x y z), psi = robot theta,
 * err = the two pixel-noise sigma rows.  out = (uvd.x, uvd.y). */
static void project_one(const orc_state *st, const srukf_params *p, const double feat[6], const double pos[3],
                        double psi, const double err[2], double out[2])
{
    double Rwc[9], Rcw[9];
    transfer_matrix(Rwc, psi);                                                     /* 1642 */
    inv3(Rwc, Rcw);                                                                /* 1643 */
    double xi = feat[0], yi = feat[1], zi = feat[2], theta = feat[3], phi = feat[4], rho = feat[5];
    double Hlw[3];                                                                 /* 3272-3275 */
    Hlw[0] = xi + 1 / rho * cos(phi) * sin(theta) - pos[0];
    Hlw[1] = yi - 1 / rho * sin(phi) - pos[1];
    Hlw[2] = zi + 1 / rho * cos(phi) * cos(theta) - pos[2];
    double Hlr[3];                                                                 /* 3292: Hlr = Rcw*Hlw (cv gemm: row . column in order) */
    for (int r = 0; r < 3; r++) Hlr[r] = Rcw[3 * r + 0] * Hlw[0] + Rcw[3 * r + 1] * Hlw[1] + Rcw[3 * r + 2] * Hlw[2];
    double uvu_x, uvu_y;
    double f1 = p->cam_f / p->cam_dx, f2 = p->cam_f / p->cam_dy;                   /* 336-337 */
    if (Hlr[2] == 0) { uvu_x = 0; uvu_y = 0; }                                      /* 3331-3335 */
    else {
        uvu_y = p->cam_cx + f1 * Hlr[0] / Hlr[2] + err[0];                         /* 3338 (x/y swap) */
        uvu_x = p->cam_cy + f2 * Hlr[1] / Hlr[2] + err[1];                         /* 3339 */
        if (uvu_x < 10 || uvu_x > p->image_w - 10 || uvu_y < 10 || uvu_y > p->image_h - 10) { uvu_x = 0; uvu_y = 0; }   /* 3341-3345 */
    }
    distort_rw(st, p, uvu_x, uvu_y, &out[0], &out[1]);                             /* 1667 */
}

ORC_API void orc_project(const srukf_params *p, int count, const double *feat6, const double *pos3,
                         const double *psi, const double *err2, double *uv_out, int early_exit)
{
    orc_state tmp; memset(&tmp, 0, sizeof tmp); tmp.newton_early_exit = early_exit;
    for (int i = 0; i < count; i++) project_one(&tmp, p, feat6 + 6 * i, pos3 + 3 * i, psi[i], err2 + 2 * i, uv_out + 2 * i);
}

/* ------------------------------------------------------------------------------------------ */
/* expandMatrix (1123-1135) + generateSigmaPoints (1148-1162):  sigma is Na x (2Na+1) row-major;
 * col0 = mu; col(i+1) = mu + gamma*sr.row(i)^T; col(Na+i+1) = mu - gamma*sr.row(i)^T
 * (addWeighted: mu*1 + element*(+-gamma) + 0). sr is Na x Na row-major.                      */
static void generate_sigma(double *sigma, const double *mu, const double *sr, int Na, double gamma)
{
    int L = 2 * Na + 1;
    for (int r = 0; r < Na; r++) sigma[(size_t)r * L + 0] = mu[r];                 /* 1152 */
    for (int i = 0; i < Na; i++)                                                   /* 1155-1161 */
        for (int r = 0; r < Na; r++) {
            double e = sr[(size_t)i * Na + r];
            sigma[(size_t)r * L + (i + 1)]      = mu[r] * 1 + e * gamma + 0;
            sigma[(size_t)r * L + (Na + i + 1)] = mu[r] * 1 + e * ((-1) * gamma) + 0;
        }
}

/* ------------------------------------------------------------------------------------------ */
ORC_API orc_state *orc_create(int N, const srukf_params *p)
{
    orc_state *st = (orc_state *)calloc(1, sizeof(orc_state));
    st->p = *p; st->N = N; st->n = 6 * N + 4;
    int n = st->n;
    st->X = (double *)calloc(n, sizeof(double));
    st->S = (double *)calloc((size_t)n * n, sizeof(double));
    /* initializeParameters, SLAM.cpp:226-231 (robot block) */
    st->S[(size_t)(n - 4) * n + (n - 4)] = p->sigma_x;
    st->S[(size_t)(n - 3) * n + (n - 3)] = p->sigma_y;
    st->S[(size_t)(n - 2) * n + (n - 2)] = p->sigma_z;
    st->S[(size_t)(n - 1) * n + (n - 1)] = p->sigma_theta;
    st->Qt[0] = st->Qt[1] = p->sigma_measure;                                       /* 238: Qt = eye(2)*m_sigmaMeasure */
    st->Na = n + 5; st->L = 2 * st->Na + 1;
    st->sigma = (double *)calloc((size_t)st->Na * st->L, sizeof(double));
    st->Z = (double *)calloc((size_t)2 * (N > 0 ? N : 1) * st->L, sizeof(double));
    st->h = (double *)calloc(2 * (N > 0 ? N : 1), sizeof(double));
    st->Si = (double *)calloc(4 * (N > 0 ? N : 1), sizeof(double));
    st->visible = (int *)calloc((N > 0 ? N : 1), sizeof(int));
    st->newton_early_exit = 1;
    return st;
}

ORC_API void orc_destroy(orc_state *st)
{
    if (!st) return;
    free(st->X); free(st->S); free(st->sigma); free(st->Z); free(st->h); free(st->Si); free(st->visible); free(st);
}

ORC_API void orc_set_state(orc_state *st, const double *X, const double *S)
{
    memcpy(st->X, X, sizeof(double) * st->n);
    memcpy(st->S, S, sizeof(double) * (size_t)st->n * st->n);
}
ORC_API void orc_get_state(const orc_state *st, double *X, double *S)
{
    if (X) memcpy(X, st->X, sizeof(double) * st->n);
    if (S) memcpy(S, st->S, sizeof(double) * (size_t)st->n * st->n);
}
ORC_API void orc_set_newton_early_exit(orc_state *st, int on) { st->newton_early_exit = on; }
ORC_API void orc_set_frozen_center(orc_state *st, int on) { st->frozen_center = on; }
ORC_API void orc_get_clamp_stats(const orc_state *st, long long out[3]) { out[0] = st->clamp_eps; out[1] = st->clamp_theta; out[2] = st->pivots; }
ORC_API const double *orc_sigma_ptr(const orc_state *st) { return st->sigma; }
ORC_API const double *orc_Z_ptr(const orc_state *st) { return st->Z; }

/* ------------------------------------------------------------------------------------------ */
/* predictMotion numeric tail, SLAM.cpp:1430-1465, with passSigmaThroughMotionFunction
 * (1476-1532) and QrAndCholeskyForMotion (1539-1555).  odo = (x, y, theta) of two consecutive
 * odometry samples (m_odoXY / m_odoTheta row 1).                                              */
ORC_API int orc_predict_motion(orc_state *st, const double odo_prev[3], const double odo_cur[3])
{
    const srukf_params *p = &st->p;
    if (p->noise_type != 0) return SRUKF_ERR_UNSUPPORTED;   /* FLAG_4_NOISE2/3 draw random numbers (1505-1516) */
    int dim = st->n, Na = dim + 3 + 2, L = 2 * Na + 1;                              /* 1431-1433 */
    st->Na = Na; st->L = L;

    double dx = odo_cur[0] - odo_prev[0];                                          /* 1446 */
    double dy = odo_cur[1] - odo_prev[1];                                          /* 1447 */
    double rot1  = atan2(dy, dx) - odo_prev[2];                                    /* 1448 (no angle wrap) */
    double trans = sqrt(dy * dy + dx * dx);                                        /* 1449 */
    double rot2  = odo_cur[2] - odo_prev[2] - rot1;                                /* 1450 */
    st->Ut[0] = rot1; st->Ut[1] = trans; st->Ut[2] = rot2;                         /* 1452-1454 */
    st->Mt[0] = p->a1 * rot1 * rot1 + p->a2 * trans * trans;                       /* 1456 */
    st->Mt[1] = p->a3 * trans * trans + p->a4 * rot1 * rot1 + p->a4 * rot2 * rot2; /* 1457 */
    st->Mt[2] = p->a1 * rot2 * rot2 + p->a2 * trans * trans;                       /* 1458 */

    set_weights(st, Na);                                                           /* 1460 */
    /* expandMatrix x2 (1461-1462): mu=[X;0_3;0_2], sr=blockdiag(S, Mt, Qt) */
    double *mu = (double *)calloc(Na, sizeof(double));
    double *sr = (double *)calloc((size_t)Na * Na, sizeof(double));
    memcpy(mu, st->X, sizeof(double) * dim);
    for (int i = 0; i < dim; i++) memcpy(sr + (size_t)i * Na, st->S + (size_t)i * dim, sizeof(double) * dim);
    for (int d = 0; d < 3; d++) sr[(size_t)(dim + d) * Na + (dim + d)] = st->Mt[d];
    for (int d = 0; d < 2; d++) sr[(size_t)(dim + 3 + d) * Na + (dim + 3 + d)] = st->Qt[d];
    generate_sigma(st->sigma, mu, sr, Na, st->gamma);                              /* 1463 */
    free(mu); free(sr);

    /* passSigmaThroughMotionFunction, 1476-1532 (FLAG_4_NOISE1 branch 1490-1494) */
    double *sg = st->sigma;
    double mean[4] = {0, 0, 0, 0};
    for (int i = 0; i < L; i++) {
        double r1 = st->Ut[0] - sg[(size_t)(dim + 0) * L + i];
        double tr = st->Ut[1] - sg[(size_t)(dim + 1) * L + i];
        double r2 = st->Ut[2] - sg[(size_t)(dim + 2) * L + i];
        double th = sg[(size_t)(dim - 1) * L + i];
        double upd[4] = { tr * cos(th + r1), tr * sin(th + r1), 0, r1 + r2 };      /* 1518-1521 */
        for (int d = 0; d < 4; d++) sg[(size_t)(dim - 4 + d) * L + i] += upd[d];   /* 1523 */
        for (int d = 0; d < 4; d++) {                                              /* 1526-1529 addWeighted */
            double e = sg[(size_t)(dim - 4 + d) * L + i];
            if (!i) mean[d] = e * st->wm0 + mean[d] * 0 + 0;
            else    mean[d] = e * st->wi + mean[d] * 1 + 0;
        }
    }
    for (int d = 0; d < 4; d++) st->X[dim - 4 + d] = mean[d];                      /* 1531 */

    /* QrAndCholeskyForMotion, 1539-1555: QR rows = wi_sr*(sigma_{i+1}[0:dim] - sigma_0[0:dim])^T */
    int dimx = 2 * Na, dimy = dim;
    double *QR = (double *)malloc(sizeof(double) * (size_t)dimx * dimy);
    for (int i = 0; i < dimx; i++)
        for (int r = 0; r < dim; r++)
            QR[(size_t)i * dimy + r] = st->wi_sr * (sg[(size_t)r * L + (i + 1)] - sg[(size_t)r * L + 0]);   /* 1552 */
    orc_qr_r(QR, dimx, dimy, st->S);                                               /* 1555 */
    free(QR);
    return SRUKF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* predictMeasurement, SLAM.cpp:1604-1608 = passSigmaThroughMesaurementFunction (1615-1691)
 * + QrAndCholeskyForMeasurement (1700-1748) + calculateOneFeatureCovariance (1759-1795).      */
ORC_API int orc_predict_measurement(orc_state *st, double *h_out, double *Si_out, int *vis_out)
{
    int dim = st->n, Na = st->Na, L = st->L, N = st->N;
    const double *sg = st->sigma;
    double *Z = st->Z;
    for (int i = 0; i < L; i++) {                                                  /* 1634 */
        double err[2] = { sg[(size_t)(dim + 3) * L + i], sg[(size_t)(dim + 4) * L + i] };                   /* 1637 */
        double pos[3] = { sg[(size_t)(dim - 4) * L + i], sg[(size_t)(dim - 3) * L + i], sg[(size_t)(dim - 2) * L + i] };   /* 1640 */
        double psi = sg[(size_t)(dim - 1) * L + i];                                /* 1642 */
        for (int id = 0; id < N; id++) {                                           /* 1647-1674 */
            double feat[6], uv[2];
            for (int d = 0; d < 6; d++) feat[d] = sg[(size_t)(6 * id + d) * L + i];   /* 1662 */
            project_one(st, &st->p, feat, pos, psi, err, uv);
            Z[(size_t)(2 * id + 0) * L + i] = uv[0];                               /* 1669 */
            Z[(size_t)(2 * id + 1) * L + i] = uv[1];                               /* 1670 */
        }
        for (int r = 0; r < 2 * N; r++) {                                          /* 1678-1681 addWeighted */
            double e = Z[(size_t)r * L + i];
            if (!i) st->h[r] = e * st->wm0 + st->h[r] * 0 + 0;
            else    st->h[r] = e * st->wi + st->h[r] * 1 + 0;
        }
    }
    /* QrAndCholeskyForMeasurement, 1700-1748 */
    double *QR = (double *)malloc(sizeof(double) * (size_t)2 * Na * 2);
    for (int id = 0; id < N; id++) {
        double px = st->h[2 * id + 0], py = st->h[2 * id + 1];                     /* 1724-1725 */
        st->visible[id] = 0;
        memset(st->Si + 4 * id, 0, 4 * sizeof(double));
        if (px != 0 && py != 0) {                                                  /* 1727 */
            st->visible[id] = 1;
            /* calculateOneFeatureCovariance, 1759-1775 */
            for (int i = 0; i < 2 * Na; i++)
                for (int c = 0; c < 2; c++)
                    QR[(size_t)i * 2 + c] = st->wi_sr * (Z[(size_t)(2 * id + c) * L + (i + 1)] - Z[(size_t)(2 * id + c) * L + 0]);   /* 1773 */
            orc_qr_r(QR, 2 * Na, 2, st->Si + 4 * id);                              /* 1775 */
        }
    }
    free(QR);
    if (h_out)  memcpy(h_out, st->h, sizeof(double) * 2 * N);
    if (Si_out) memcpy(Si_out, st->Si, sizeof(double) * 4 * N);
    if (vis_out) memcpy(vis_out, st->visible, sizeof(int) * N);
    return SRUKF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* P = S^T S with S upper triangular (m_S_k.t()*m_S_k, SLAM.cpp:2118, 2404).  The reference
 * runs a full cv::gemm; the zero lower triangle only adds exact zeros, so skipping those terms
 * changes no bit of any product (only possibly the sign of a zero).                          */
static void sts(const double *S, int n, double *P)
{
    /* row-major rank-1 accumulation; each P[r][c] still sums its terms in increasing k */
    for (int r = 0; r < n; r++) memset(P + (size_t)r * n + r, 0, sizeof(double) * (size_t)(n - r));
    for (int k = 0; k < n; k++) {
        const double *Sk = S + (size_t)k * n;
        for (int r = k; r < n; r++) {
            const double a = Sk[r];
            double *Pr = P + (size_t)r * n;
            for (int c = r; c < n; c++) Pr[c] += a * Sk[c];
        }
    }
    for (int r = 0; r < n; r++) for (int c = r + 1; c < n; c++) P[(size_t)c * n + r] = P[(size_t)r * n + c];
}
ORC_API void orc_sts(const double *S, int n, double *P) { sts(S, n, P); }

/* getPermutationMatrix, SLAM.cpp:1303-1334.  perm[r] = column c with Pi[r][c] = 1, i.e.
 * X_normal[r] = X_disordered[perm[r]].  dim = new dimension, K = number of newly added
 * landmarks (m_nFilters).                                                                    */
static void permutation(int dim, int K, int *perm)
{
    int dimOld = dim - 6 * K;
    for (int i = 0; i < dimOld - 4; i++) perm[i] = i;                              /* 1312-1316 */
    for (int d = 0; d < 4; d++) perm[dim - 4 + d] = dimOld - 4 + d;                /* 1319-1322 */
    for (int id = 0; id < K; id++) {                                               /* 1324-1333 */
        for (int d = 0; d < 3; d++) perm[dimOld - 4 + 6 * id + d]     = dimOld + 3 * K + 3 * id + d;
        for (int d = 0; d < 3; d++) perm[dimOld - 4 + 6 * id + 3 + d] = dimOld + 3 * id + d;
    }
}

/* CholeskyDecompositionWithPivoting, SLAM.cpp:2158-2179 ("rank-aware"): Cov dim x dim in the
 * disordered order, rank = m_covRank.  sr = [R11 R12; 0 0], R11 = gmw(Cov11),
 * R12 = R11^{-T} Cov12 (2175, cv inv of an upper-triangular matrix == back substitution up to
 * rounding; restated as a triangular solve).                                                 */
static void chol_with_pivoting(orc_state *st, double *sr, const double *Cov, int dim, int rank)
{
    memset(sr, 0, sizeof(double) * (size_t)dim * dim);                             /* 2161 */
    if (dim == rank) { orc_gmw(Cov, dim, st->p.epsilon, sr, NULL, NULL, &st->clamp_eps, &st->clamp_theta); return; }   /* 2163-2166 */
    double *C11 = (double *)malloc(sizeof(double) * (size_t)rank * rank);
    double *R11 = (double *)malloc(sizeof(double) * (size_t)rank * rank);
    for (int i = 0; i < rank; i++) memcpy(C11 + (size_t)i * rank, Cov + (size_t)i * dim, sizeof(double) * rank);   /* 2170 */
    orc_gmw(C11, rank, st->p.epsilon, R11, NULL, NULL, &st->clamp_eps, &st->clamp_theta);                          /* 2173 */
    /* R12 = R11^{-T} * Cov12 : solve R11^T Y = Cov12, forward substitution (R11^T lower) */
    for (int c = rank; c < dim; c++) {
        for (int i = 0; i < rank; i++) {
            double acc = Cov[(size_t)i * dim + c];
            for (int k = 0; k < i; k++) acc -= R11[(size_t)k * rank + i] * sr[(size_t)k * dim + c];
            sr[(size_t)i * dim + c] = acc / R11[(size_t)i * rank + i];
        }
    }
    for (int i = 0; i < rank; i++) memcpy(sr + (size_t)i * dim, R11 + (size_t)i * rank, sizeof(double) * rank);   /* 2176 */
    free(C11); free(R11);
}

/* GSLCholeskyUpdate, SLAM.cpp:2106-2155, FLAG_4_DOWNDATING (sign=-1) or UPDATING (sign=+1),
 * one column u at a time: src1 = S^T S; dst = src1 -+ u u^T; S = gmw(dst) or the reorder path. */
static void cholesky_update_col(orc_state *st, const double *u, int sign, int reorder, int K_new, double *P /* n*n scratch */)
{
    int n = st->n;
    sts(st->S, n, P);                                                              /* 2118 */
    for (int r = 0; r < n; r++)                                                    /* 2120, 2144/2149 */
        for (int c = 0; c < n; c++) P[(size_t)r * n + c] = P[(size_t)r * n + c] + sign * (u[r] * u[c]);
    if (reorder == SRUKF_NEEDNOT_REORDER) {
        orc_gmw(P, n, st->p.epsilon, st->S, NULL, NULL, &st->clamp_eps, &st->clamp_theta);   /* 2152 */
        st->pivots += n;
    } else {
        /* 2122-2138: dst = Pi^T (..) Pi ; S_dis = pivoted(dst) ; S = R(QR(Pi S_dis Pi^T)) */
        int *perm = (int *)malloc(sizeof(int) * n);
        permutation(n, K_new, perm);
        int rank = n - 3 * K_new;                                                  /* 2131/2126 */
        double *dst = (double *)malloc(sizeof(double) * (size_t)n * n);
        double *Sd  = (double *)malloc(sizeof(double) * (size_t)n * n);
        /* (Pi^T A Pi)[a][b] = A[r][c] with perm[r]=a, perm[c]=b */
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) dst[(size_t)perm[r] * n + perm[c]] = P[(size_t)r * n + c];
        chol_with_pivoting(st, Sd, dst, n, rank);                                  /* 2136 */
        /* (Pi B Pi^T)[r][c] = B[perm[r]][perm[c]] */
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) dst[(size_t)r * n + c] = Sd[(size_t)perm[r] * n + perm[c]];
        orc_qr_r(dst, n, n, st->S);                                                /* 2137 */
        st->pivots += rank;
        free(perm); free(dst); free(Sd);
    }
}

/* calculateOneFeatureCrossCovariance, SLAM.cpp:2020-2038: Pxy (dim x 2, row-major) with the
 * CURRENT X and the fixed sigma set.                                                          */
static void cross_cov(const orc_state *st, int id, const double hi[2], double *Pxy)
{
    int dim = st->n, L = st->L;
    const double *sg = st->sigma, *Z = st->Z;
    for (int i = 0; i < L; i++) {                                                  /* 2028 */
        double s2[2] = { Z[(size_t)(2 * id) * L + i] - hi[0], Z[(size_t)(2 * id + 1) * L + i] - hi[1] };   /* 2031 */
        double w = i ? st->wi : st->wc0;
        for (int r = 0; r < dim; r++) {
            double s1 = sg[(size_t)r * L + i] - st->Xcenter[r];                    /* 2030 (Xcenter = m_X_k) */
            if (!i) { Pxy[2 * r] = w * s1 * s2[0] + 0; Pxy[2 * r + 1] = w * s1 * s2[1] + 0; }   /* 2034 */
            else    { Pxy[2 * r] += w * s1 * s2[0];    Pxy[2 * r + 1] += w * s1 * s2[1]; }      /* 2036 */
        }
    }
}

/* KalmanUpdate, SLAM.cpp:2048-2104.  z[2N] = matchLocation, matched[N] = isMatching.
 * reorder: SRUKF_NEED_REORDER iff landmarks were added before this frame (m_nAddings != 0,
 * 2083-2090), then K_new = m_nFilters.  mode SEQUENTIAL is the reference; mode BATCHED applies
 * all gains first and refactors once:  S <- gmw(S^T S - U U^T).                               */
ORC_API int orc_update(orc_state *st, const double *z, const int *matched, int reorder, int K_new, int mode)
{
    int n = st->n, N = st->N;
    int nm = 0; for (int k = 0; k < N; k++) nm += matched[k] ? 1 : 0;
    if (nm == 0) return SRUKF_OK;                                                  /* 2050-2051 */
    if (mode == SRUKF_UPDATE_BATCHED && reorder != SRUKF_NEEDNOT_REORDER) return SRUKF_ERR_UNSUPPORTED;
    double *Pxy = (double *)malloc(sizeof(double) * 2 * n);
    double *Ki  = (double *)malloc(sizeof(double) * 2 * n);
    double *U   = (double *)malloc(sizeof(double) * 2 * n);
    double *u   = (double *)malloc(sizeof(double) * n);
    double *P   = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *Uall = NULL; int ncols = 0;
    if (mode == SRUKF_UPDATE_BATCHED) Uall = (double *)malloc(sizeof(double) * (size_t)2 * nm * n);
    double *Xfrozen = NULL;
    st->Xcenter = st->X;                                                           /* the reference: the running m_X_k */
    if (st->frozen_center) { Xfrozen = (double *)malloc(sizeof(double) * n); memcpy(Xfrozen, st->X, sizeof(double) * n); st->Xcenter = Xfrozen; }

    for (int id = 0; id < N; id++) {                                               /* 2066 */
        if (!matched[id]) continue;                                                /* 2068 */
        double zi[2] = { z[2 * id], z[2 * id + 1] };                               /* 2070 */
        double hi[2] = { st->h[2 * id], st->h[2 * id + 1] };                       /* 2071 (predictLocation) */
        const double *si = st->Si + 4 * id;                                        /* 2073 */
        cross_cov(st, id, hi, Pxy);                                                /* 2075 */
        double sii[4]; inv2(si, sii);                                              /* 2077 */
        /* Ki = Pxy*sii*sii^T  (2078): T = Pxy*sii, then T*sii^T */
        double siit[4] = { sii[0], sii[2], sii[1], sii[3] };
        double sit[4]  = { si[0], si[2], si[1], si[3] };
        for (int r = 0; r < n; r++) {
            double t0 = Pxy[2 * r] * sii[0] + Pxy[2 * r + 1] * sii[2];
            double t1 = Pxy[2 * r] * sii[1] + Pxy[2 * r + 1] * sii[3];
            Ki[2 * r]     = t0 * siit[0] + t1 * siit[2];
            Ki[2 * r + 1] = t0 * siit[1] + t1 * siit[3];
        }
        double inn[2] = { zi[0] - hi[0], zi[1] - hi[1] };
        for (int r = 0; r < n; r++) st->X[r] += Ki[2 * r] * inn[0] + Ki[2 * r + 1] * inn[1];   /* 2079 */
        for (int r = 0; r < n; r++) {                                              /* 2080: U = Ki*si^T */
            U[2 * r]     = Ki[2 * r] * sit[0] + Ki[2 * r + 1] * sit[2];
            U[2 * r + 1] = Ki[2 * r] * sit[1] + Ki[2 * r + 1] * sit[3];
        }
        for (int c = 0; c < 2; c++) {                                              /* 2116 */
            for (int r = 0; r < n; r++) u[r] = U[2 * r + c];                       /* 2119 */
            if (mode == SRUKF_UPDATE_SEQUENTIAL) cholesky_update_col(st, u, -1, reorder, K_new, P);   /* 2083-2090 */
            else { memcpy(Uall + (size_t)ncols * n, u, sizeof(double) * n); ncols++; }
        }
    }
    if (mode == SRUKF_UPDATE_BATCHED) {
        sts(st->S, n, P);
        for (int r = 0; r < n; r++)
            for (int c = r; c < n; c++) {
                double acc = 0.0;
                for (int m = 0; m < ncols; m++) acc += Uall[(size_t)m * n + r] * Uall[(size_t)m * n + c];
                P[(size_t)r * n + c] -= acc; P[(size_t)c * n + r] = P[(size_t)r * n + c];
            }
        orc_gmw(P, n, st->p.epsilon, st->S, NULL, NULL, &st->clamp_eps, &st->clamp_theta);
        st->pivots += n;
        free(Uall);
    }
    free(Pxy); free(Ki); free(U); free(u); free(P); free(Xfrozen);
    st->Xcenter = st->X;
    return SRUKF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Landmark augmentation (needed to build a faithful rank-deficient S0): the numeric part of
 * integrateFeaturesInformation (SLAM.cpp:826-871) = expandMatrix + generateSigmaPoints +
 * passSigmaThroughMapingFunction (1177-1250) + QrAndCholeskyForInitilization (1260-1300) +
 * getPermutationMatrix (1303-1334).
 * In: X (dim), S (dim x dim), K new features with pixel uv[K][2] (m_keyPoints[i].pt).
 * Out: X_new, S_new of dimension dim + 6K in NORMAL order (new landmarks inserted before the
 * robot block).  Uses st->p only.                                                             */
ORC_API int orc_joint_init(const srukf_params *p, int dim, const double *X, const double *S, int K, const double *uv,
                           double *X_new, double *S_new)
{
    orc_state tmp; memset(&tmp, 0, sizeof tmp); tmp.p = *p; tmp.newton_early_exit = 1;
    int Na = dim + 3 * K, L = 2 * Na + 1;                                          /* 827-828 */
    int dimn = dim + 6 * K;
    set_weights(&tmp, Na);                                                         /* 867 */
    double *mu = (double *)calloc(Na, sizeof(double));
    double *sr = (double *)calloc((size_t)Na * Na, sizeof(double));
    memcpy(mu, X, sizeof(double) * dim);
    for (int i = 0; i < dim; i++) memcpy(sr + (size_t)i * Na, S + (size_t)i * dim, sizeof(double) * dim);
    for (int i = 0; i < K; i++) {                                                  /* 847-858 */
        mu[dim + 3 * i + 0] = uv[2 * i + 0]; mu[dim + 3 * i + 1] = uv[2 * i + 1]; mu[dim + 3 * i + 2] = p->rho0;
        sr[(size_t)(dim + 3 * i + 0) * Na + dim + 3 * i + 0] = p->sigma_measure;
        sr[(size_t)(dim + 3 * i + 1) * Na + dim + 3 * i + 1] = p->sigma_measure;
        sr[(size_t)(dim + 3 * i + 2) * Na + dim + 3 * i + 2] = p->sigma_rho;
    }
    double *sin_ = (double *)calloc((size_t)Na * L, sizeof(double));
    generate_sigma(sin_, mu, sr, Na, tmp.gamma);                                   /* 869 */
    /* passSigmaThroughMapingFunction, 1177-1250 */
    double *sout = (double *)calloc((size_t)dimn * L, sizeof(double));
    double *mu_angle = (double *)calloc(3 * K, sizeof(double));
    for (int r = 0; r < dim; r++) memcpy(sout + (size_t)r * L, sin_ + (size_t)r * L, sizeof(double) * L);   /* 1185 */
    for (int i = 0; i < L; i++) {                                                  /* 1201 */
        double pos[3] = { sin_[(size_t)(dim - 4) * L + i], sin_[(size_t)(dim - 3) * L + i], sin_[(size_t)(dim - 2) * L + i] };   /* 1203 */
        double Rwc[9]; transfer_matrix(Rwc, sin_[(size_t)(dim - 1) * L + i]);     /* 1204 */
        for (int id = 0; id < K; id++) {                                           /* 1206 */
            int index_in = dim + 3 * id, out1 = index_in, out2 = index_in + 3 * K; /* 1208-1210 */
            double uvd_x = sin_[(size_t)(index_in + 0) * L + i], uvd_y = sin_[(size_t)(index_in + 1) * L + i];
            double rho = sin_[(size_t)(index_in + 2) * L + i];                     /* 1213-1215 */
            double uvu_x, uvu_y; undistort_rw(p, uvd_x, uvd_y, &uvu_x, &uvu_y);    /* 1217 */
            double f1 = p->cam_f / p->cam_dx, f2 = p->cam_f / p->cam_dy;
            double Hlr[3] = { (uvu_y - p->cam_cx) / f1, (uvu_x - p->cam_cy) / f2, 1 };   /* 1218 -> 3360-3363 */
            double Hlw[3];                                                         /* 1219 -> 3386 */
            for (int r = 0; r < 3; r++) Hlw[r] = Rwc[3 * r] * Hlr[0] + Rwc[3 * r + 1] * Hlr[1] + Rwc[3 * r + 2] * Hlr[2];
            double state[3] = { atan2(Hlw[0], Hlw[2]), atan2(-Hlw[1], sqrt(Hlw[0] * Hlw[0] + Hlw[2] * Hlw[2])), rho };   /* 1220 -> 3411-3419 */
            for (int d = 0; d < 3; d++) sout[(size_t)(out1 + d) * L + i] = state[d];   /* 1222 */
            for (int d = 0; d < 3; d++) sout[(size_t)(out2 + d) * L + i] = pos[d];     /* 1223 */
            for (int d = 0; d < 3; d++) {                                          /* 1232-1241 */
                if (!i) mu_angle[3 * id + d] = state[d] * tmp.wm0 + mu_angle[3 * id + d] * 0 + 0;
                else    mu_angle[3 * id + d] = state[d] * tmp.wi + mu_angle[3 * id + d] * 1 + 0;
            }
        }
    }
    /* 1245-1249: x_new (disordered) = [X; mu_angle; repeat(cam_position, K)] */
    double *xdis = (double *)calloc(dimn, sizeof(double));
    memcpy(xdis, X, sizeof(double) * dim);
    memcpy(xdis + dim, mu_angle, sizeof(double) * 3 * K);
    for (int id = 0; id < K; id++) for (int d = 0; d < 3; d++) xdis[Na + 3 * id + d] = X[dim - 4 + d];
    /* QrAndCholeskyForInitilization, 1260-1300 */
    double *QR = (double *)malloc(sizeof(double) * (size_t)2 * Na * dimn);
    for (int i = 0; i < 2 * Na; i++)
        for (int r = 0; r < dimn; r++)
            QR[(size_t)i * dimn + r] = tmp.wi_sr * (sout[(size_t)r * L + (i + 1)] - sout[(size_t)r * L + 0]);   /* 1274 */
    double *Sdis = (double *)malloc(sizeof(double) * (size_t)dimn * dimn);
    orc_qr_r(QR, 2 * Na, dimn, Sdis);                                              /* 1276 */
    int *perm = (int *)malloc(sizeof(int) * dimn);
    permutation(dimn, K, perm);                                                    /* 1280-1290 */
    for (int r = 0; r < dimn; r++) X_new[r] = xdis[perm[r]];                       /* 1293 */
    double *PSP = (double *)malloc(sizeof(double) * (size_t)dimn * dimn);
    for (int r = 0; r < dimn; r++) for (int c = 0; c < dimn; c++) PSP[(size_t)r * dimn + c] = Sdis[(size_t)perm[r] * dimn + perm[c]];
    orc_qr_r(PSP, dimn, dimn, S_new);                                              /* 1294 */
    free(mu); free(sr); free(sin_); free(sout); free(mu_angle); free(xdis); free(QR); free(Sdis); free(perm); free(PSP);
    return SRUKF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* deleteOneFeature, numeric part (SLAM.cpp:2637-2663): landmark `id` (0-based, state order) leaves
 * the state: X loses its 6 entries, S loses its 6 rows and columns, and the 6 removed rows (without
 * the removed columns), V (6 x (dim-6)), are folded back in by GSLCholeskyUpdate(V^T,
 * FLAG_4_UPDATING, FLAG_4_NEEDNOT_REORDER): one  S <- gmw(S^T S + u u^T)  per row u of V.
 * In: X (dim), S (dim x dim).  Out: X_new (dim-6), S_new ((dim-6) x (dim-6)).                     */
ORC_API int orc_delete_feature(const srukf_params *p, int dim, const double *X, const double *S, int id,
                               double *X_new, double *S_new)
{
    int N = (dim - 4) / 6, dn = dim - 6;
    if (id < 0 || id >= N) return SRUKF_ERR_BAD_ARG;
    orc_state *st = orc_create(N - 1 > 0 ? N - 1 : 1, p);
    if (!st) return SRUKF_ERR_NOMEM;
    /* orc_create sized the state for max(N-1,1) landmarks; the update below only needs n = dn */
    st->N = N - 1; st->n = dn;
    double *V = (double *)malloc(sizeof(double) * 6 * (size_t)dn);
    double *P = (double *)malloc(sizeof(double) * (size_t)dn * dn);
    int lo = 6 * id, hi = 6 * (id + 1);
    for (int r = 0, a = 0; r < dim; r++) {                                          /* 2645-2661 */
        if (r >= lo && r < hi) continue;
        X_new[a] = X[r];
        for (int c = 0, b = 0; c < dim; c++) { if (c >= lo && c < hi) continue; st->S[(size_t)a * dn + b] = S[(size_t)r * dim + c]; b++; }
        a++;
    }
    for (int q = 0; q < 6; q++)
        for (int c = 0, b = 0; c < dim; c++) { if (c >= lo && c < hi) continue; V[(size_t)q * dn + b] = S[(size_t)(lo + q) * dim + c]; b++; }
    for (int q = 0; q < 6; q++) cholesky_update_col(st, V + (size_t)q * dn, +1, SRUKF_NEEDNOT_REORDER, 0, P);   /* 2667-2668, 2116-2154 */
    memcpy(S_new, st->S, sizeof(double) * (size_t)dn * dn);
    free(V); free(P); orc_destroy(st);
    return SRUKF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Data association: wrapPatch (SLAM.cpp:1803-1906), dataAssociation (1915-2009),
 * calculateCrossCorrelation (3141-3166).  Constants HP_INIT_W = HP_INIT_H = 10, HP_MATCH_W =
 * HP_MATCH_H = 8 (SLAM.cpp:41-44), chi2inv(0.95, 2) = 5.99146454710798 (54),
 * THRESHOLD_MATCH_PATCH = 0.8 (184).
 * Un-vendored dependency: the 4x4 cv::Mat::inv() of wrapPatch (OpenCV 2.4.3, LU with partial
 * pivoting); restated as Gauss-Jordan with partial pivoting.  3x3 / 2x2 inverses are OpenCV's
 * closed forms (inv3 / inv2 above).                                                           */
#define ORC_HP_INIT 10
#define ORC_HP_MATCH 8
static int inv4(const double a[16], double out[16])
{
    double m[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { m[r][c] = a[4 * r + c]; m[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int piv = c; double big = fabs(m[c][c]);
        for (int r = c + 1; r < 4; r++) if (fabs(m[r][c]) > big) { big = fabs(m[r][c]); piv = r; }
        if (big == 0.0) return 0;
        if (piv != c) for (int k = 0; k < 8; k++) { double t = m[c][k]; m[c][k] = m[piv][k]; m[piv][k] = t; }
        double d = 1.0 / m[c][c];
        for (int k = 0; k < 8; k++) m[c][k] *= d;
        for (int r = 0; r < 4; r++) if (r != c) { double f = m[r][c]; if (f != 0.0) for (int k = 0; k < 8; k++) m[r][k] -= f * m[c][k]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[4 * r + c] = m[r][4 + c];
    return 1;
}
static void mat4mul(const double a[16], const double b[16], double o[16])
{
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { double t = 0; for (int k = 0; k < 4; k++) t += a[4 * r + k] * b[4 * k + c]; o[4 * r + c] = t; }
}
static void mat3mul(const double a[9], const double b[9], double o[9])
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double t = 0; for (int k = 0; k < 3; k++) t += a[3 * r + k] * b[3 * k + c]; o[3 * r + c] = t; }
}
/* wrapPatch for ONE landmark.  robot = (x, y, z, theta) = X[dim-4..dim-1]; initR (3x3), initT (3),
 * initPixel (2): the PointsMap fields set at creation (SLAM.cpp:920-925); xyz: current Cartesian
 * mean (PointsMap::xyz); predict: predictLocation; initPatch[21][21] (uchar, cv::Mat rows);
 * matchPatch[17][17] in/out — pixels whose warp falls outside the init patch KEEP their old value. */
ORC_API void orc_warp_patch(const srukf_params *p, const double robot[4], const double initR[9], const double initT[3],
                            const double initPixel[2], const double xyz[3], const double predict[2],
                            const unsigned char *initPatch, unsigned char *matchPatch)
{
    orc_state tmp; memset(&tmp, 0, sizeof tmp); tmp.newton_early_exit = 1;
    const double f1 = p->cam_f / p->cam_dx, f2 = p->cam_f / p->cam_dy;
    double Rwc[9]; transfer_matrix_cr(Rwc, robot[3]);                               /* 1806-1807 */
    double C0W[16] = { 0 }, C1W[16] = { 0 };                                        /* 1821-1827 */
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { C0W[4 * r + c] = initR[3 * r + c] + 0; C1W[4 * r + c] = Rwc[3 * r + c] + 0; }
        C0W[4 * r + 3] = initR[3 * r] * initT[0] + initR[3 * r + 1] * initT[1] + initR[3 * r + 2] * initT[2] + 0;
        C1W[4 * r + 3] = Rwc[3 * r] * robot[0] + Rwc[3 * r + 1] * robot[1] + Rwc[3 * r + 2] * robot[2] + 0;
    }
    C0W[15] = 1; C1W[15] = 1;
    double C0Wi[16], C1C0[16];
    inv4(C0W, C0Wi);
    mat4mul(C0Wi, C1W, C1C0);                                                       /* 1829 */
    double R[9], r3[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = C1C0[4 * r + c]; r3[r] = C1C0[4 * r + 3]; }
    double n0[3] = { initPixel[0] - p->cam_cx, initPixel[1] - p->cam_cy, -f1 };     /* 1833 */
    double t0[4] = { predict[0] - p->cam_cx, predict[1] - p->cam_cy, -f1, 1 };      /* 1834-1835 */
    double t1[4];
    for (int r = 0; r < 4; r++) t1[r] = C1C0[4 * r] * t0[0] + C1C0[4 * r + 1] * t0[1] + C1C0[4 * r + 2] * t0[2] + C1C0[4 * r + 3] * t0[3];
    for (int r = 0; r < 4; r++) t1[r] /= t1[3];   /* 1837: the division runs over the elements in order, element 3 last */
    double n1[3] = { t1[0], t1[1], t1[2] };
    double nn = sqrt(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
    for (int r = 0; r < 3; r++) n0[r] /= nn;                                        /* 1839 */
    nn = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    for (int r = 0; r < 3; r++) n1[r] /= nn;                                        /* 1840 */
    double n[3] = { n0[0] + n1[0], n0[1] + n1[1], n0[2] + n1[2] };                  /* 1841 */
    nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int r = 0; r < 3; r++) n[r] /= nn;                                         /* 1842 */
    double w4[4] = { xyz[0], xyz[1], xyz[2], 1 }, c0[4];                            /* 1844-1847 */
    for (int r = 0; r < 4; r++) c0[r] = C0Wi[4 * r] * w4[0] + C0Wi[4 * r + 1] * w4[1] + C0Wi[4 * r + 2] * w4[2] + C0Wi[4 * r + 3] * w4[3];
    for (int r = 0; r < 4; r++) c0[r] /= c0[3];
    double d = ((-1) * n[0]) * c0[0] + ((-1) * n[1]) * c0[1] + ((-1) * n[2]) * c0[2];   /* 1848-1849 */
    /* H = K (R - r n^T / d) K^{-1}                                                     1855, 1876 */
    double K[9] = { f1, 0, p->cam_cx, 0, f2, p->cam_cy, 0, 0, 1 }, Ki[9], M[9], KM[9], H[9], Hi[9];
    inv3(K, Ki);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = R[3 * r + c] - (r3[r] * n[c]) / d;
    mat3mul(K, M, KM); mat3mul(KM, Ki, H);
    double uu_x, uu_y;
    undistort_rw(p, initPixel[0], initPixel[1], &uu_x, &uu_y);                       /* 1852 */
    inv3(H, Hi);                                                                    /* 1856 */
    double q[3] = { Hi[0] * uu_x + Hi[1] * uu_y + Hi[2] * 1, Hi[3] * uu_x + Hi[4] * uu_y + Hi[5] * 1, Hi[6] * uu_x + Hi[7] * uu_y + Hi[8] * 1 };
    q[0] /= q[2]; q[1] /= q[2];                                                     /* 1857 */
    double uv_x, uv_y;
    distort_rw(&tmp, p, q[0], q[1], &uv_x, &uv_y);                                   /* 1861 */
    for (int i = 0; i < 2 * ORC_HP_MATCH + 1; i++)                                  /* 1865 */
        for (int j = 0; j < 2 * ORC_HP_MATCH + 1; j++) {
            double ax = uv_x - ORC_HP_MATCH + i, ay = uv_y - ORC_HP_MATCH + j;      /* 1869-1870 */
            double bx, by;
            undistort_rw(p, ax, ay, &bx, &by);                                       /* 1871 */
            double t[3] = { H[0] * bx + H[1] * by + H[2] * 1, H[3] * bx + H[4] * by + H[5] * 1, H[6] * bx + H[7] * by + H[8] * 1 };   /* 1874 */
            t[0] /= t[2]; t[1] /= t[2];                                             /* 1875 */
            double cx1, cy1;
            distort_rw(&tmp, p, t[0], t[1], &cx1, &cy1);                             /* 1879 */
            cx1 -= (initPixel[0] - ORC_HP_INIT - 1);                                /* 1881 */
            cy1 -= (initPixel[1] - ORC_HP_INIT - 1);                                /* 1882 */
            int lx = (int)floor(cx1), ly = (int)floor(cy1), rx = (int)ceil(cx1), ry = (int)ceil(cy1);   /* 1884-1887 */
            if (lx >= 0 && rx < 2 * ORC_HP_INIT && ly >= 0 && ry < 2 * ORC_HP_INIT) {   /* 1889 */
                double rate_lx = rx - cx1, rate_ly = ry - cy1, rate_rx = 1.0 - rate_lx, rate_ry = 1.0 - rate_ly;
                const int PW = 2 * ORC_HP_INIT + 1;
                unsigned char ll = initPatch[lx * PW + ly], lr = initPatch[lx * PW + ry];       /* at<uchar>(row = x index, col = y index) */
                unsigned char rl = initPatch[rx * PW + ly], rr = initPatch[rx * PW + ry];
                matchPatch[i * (2 * ORC_HP_MATCH + 1) + j] =
                    (unsigned char)(ll * rate_lx * rate_ly + lr * rate_lx * rate_ry + rl * rate_rx * rate_ly + rr * rate_rx * rate_ry);   /* 1899-1900 */
            }
        }
}

/* calculateCrossCorrelation, SLAM.cpp:3141-3166: roi = 17x17 window of the gray image (row stride `stride`). */
static double cross_correlation(const unsigned char *roi, int stride, const unsigned char *patch)
{
    const int PW = 2 * ORC_HP_MATCH + 1, NP = PW * PW;
    double s1 = 0, s2 = 0;
    for (int r = 0; r < PW; r++) for (int c = 0; c < PW; c++) { s1 += roi[r * stride + c]; s2 += patch[r * PW + c]; }
    double a1 = s1 / NP, a2 = s2 / NP, q1 = 0, q2 = 0, dot = 0;
    for (int r = 0; r < PW; r++) for (int c = 0; c < PW; c++) {
        double v1 = roi[r * stride + c] - a1, v2 = patch[r * PW + c] - a2;
        q1 += v1 * v1; q2 += v2 * v2; dot += v1 * v2;
    }
    double std1 = sqrt(q1), std2 = sqrt(q2);
    if (std1 == 0 || std2 == 0) return 0;
    return dot / std1 / std2;
}
/* dataAssociation for ONE visible landmark, SLAM.cpp:1947-2001.  image: H x W uchar, row-major.
 * Returns 1 and match[2] if the best correlation exceeds THRESHOLD_MATCH_PATCH; *best = maxVal.  */
ORC_API int orc_associate_one(const srukf_params *p, const unsigned char *image, const double predict[2], const double Si[4],
                              const unsigned char *matchPatch, double *best, double match[2])
{
    const int W = p->image_w, H = p->image_h;
    const double px = predict[0], py = predict[1];
    double pi[4] = { Si[0] * Si[0] + Si[2] * Si[2], Si[0] * Si[1] + Si[2] * Si[3], Si[1] * Si[0] + Si[3] * Si[2], Si[1] * Si[1] + Si[3] * Si[3] };   /* 1951: Si^T Si */
    double pinv[4]; inv2(pi, pinv);
    int half_x = (int)ceil(2 * Si[0]), half_y = (int)ceil(2 * Si[3]);               /* 1953-1954 */
    half_x = ORC_HP_INIT < (ORC_HP_MATCH > half_x ? ORC_HP_MATCH : half_x) ? ORC_HP_INIT : (ORC_HP_MATCH > half_x ? ORC_HP_MATCH : half_x);   /* 1955 */
    half_y = ORC_HP_INIT < (ORC_HP_MATCH > half_y ? ORC_HP_MATCH : half_y) ? ORC_HP_INIT : (ORC_HP_MATCH > half_y ? ORC_HP_MATCH : half_y);
    double maxVal = 0.0; int mi = 0, mj = 0;                                        /* minMaxLoc over a zero-initialised matrix: first maximum in row-major order */
    int found = 0;
    for (int j = (int)py - half_y; j <= (int)py + half_y; j++) {                     /* row-major scan of `correlation` = j outer */
        for (int i = (int)px - half_x; i <= (int)px + half_x; i++) {
            double c = 0.0;
            if (!(i < ORC_HP_MATCH || i > W - ORC_HP_MATCH - 1) && !(j < ORC_HP_MATCH || j > H - ORC_HP_MATCH - 1)) {   /* 1962, 1969 */
                double ex = i - px, ey = j - py;
                double pii = (ex * pinv[0] + ey * pinv[2]) * ex + (ex * pinv[1] + ey * pinv[3]) * ey;   /* 1975 */
                if (pii < 5.99146454710798)                                          /* 1977 */
                    c = cross_correlation(image + (size_t)(j - ORC_HP_MATCH) * W + (i - ORC_HP_MATCH), W, matchPatch);   /* 1979-1981 */
            }
            if (!found || c > maxVal) { if (!found) { maxVal = c; mi = i; mj = j; found = 1; } else { maxVal = c; mi = i; mj = j; } }
        }
    }
    if (maxVal < 0.0) maxVal = maxVal;                                              /* (maxVal initialised to 0.0 at 1986 is overwritten by minMaxLoc) */
    *best = maxVal;
    if (maxVal > 0.8) {                                                             /* 1989 */
        match[0] = (mi - ((int)px - half_x)) - half_x + px;                          /* 1991: maxLoc.x - half_x + px */
        match[1] = (mj - ((int)py - half_y)) - half_y + py;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Whole-sequence driver used by the trajectory tests and by bench.py's cpu_baseline leg:
 * runs F frames  predictMotion -> predictMeasurement -> KalmanUpdate  (CSLAM::SLAM,
 * SLAM.cpp:87-112, minus image I/O, association and display) and records per frame
 * (x, y, z, theta, P00, P01, P10, P11) — the RobotPath.txt columns (3549-3556) with
 * P = S^T S robot block (2404).                                                               */
ORC_API int orc_run_frames(orc_state *st, int F, const double *odo /* (F+1)*3 */, const double *z /* F*2N */,
                           const int *matched /* F*N */, int mode, double *traj /* F*8 or NULL */)
{
    int n = st->n, N = st->N;
    for (int f = 0; f < F; f++) {
        int rc = orc_predict_motion(st, odo + 3 * f, odo + 3 * (f + 1)); if (rc) return rc;
        rc = orc_predict_measurement(st, NULL, NULL, NULL); if (rc) return rc;
        rc = orc_update(st, z + (size_t)f * 2 * N, matched + (size_t)f * N, SRUKF_NEEDNOT_REORDER, 0, mode); if (rc) return rc;
        if (traj) {
            double *t = traj + 8 * f;
            for (int d = 0; d < 4; d++) t[d] = st->X[n - 4 + d];
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) {
                double acc = 0; int r = n - 4 + a, c = n - 4 + b;
                for (int k = 0; k < n; k++) acc += st->S[(size_t)k * n + r] * st->S[(size_t)k * n + c];
                t[4 + 2 * a + b] = acc;
            }
        }
    }
    return SRUKF_OK;
}

/* Timing helper for the "faithful" baseline at large N: cost of `cols` measurement columns of
 * the reference's per-column refactor (S^T S, -uu^T, gmw) on the current S with a tiny u. */
ORC_API int orc_time_refactor_columns(orc_state *st, int cols)
{
    int n = st->n;
    double *u = (double *)calloc(n, sizeof(double));
    double *P = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (int c = 0; c < cols; c++) { u[n - 1] = 1e-6; cholesky_update_col(st, u, -1, SRUKF_NEEDNOT_REORDER, 0, P); }
    free(u); free(P);
    return SRUKF_OK;
}
