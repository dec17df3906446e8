// srukf_assoc.hip — data association on the device: wrapPatch (SLAM.cpp:1803-1906),
// dataAssociation (1915-2009), calculateCrossCorrelation (3141-3166).  gfx950 only.
//
// The step sits between predictMeasurement and KalmanUpdate: with it on the device the per-frame
// host traffic is one 640x480 gray frame in and 3 N numbers out.  Byte work on a 300 KB frame that
// stays in L2: one workgroup per landmark, the 17x17 template and its statistics in LDS.
//
// Per landmark the host stores, at creation (SLAM.cpp:920-925): initPixel, initRotation (Rwc),
// initTrans (camera position) and initPatch = the 21x21 gray window around the rounded initPixel.
// k_warp_patch predicts how that patch looks from the current pose (plane-induced homography through
// the landmark, normal = bisector of the two viewing rays) into the 17x17 matchPatch;
// k_associate scans the chi-square gated window around the predicted pixel for the best normalised
// cross correlation.  matchPatch persists from frame to frame: pixels whose warp leaves the init
// patch keep their previous value, as in the reference (the Mat is only zeroed at creation).
#include "srukf_device.h"
#include "srukf_crtrig.h"

#define HP_INIT 10             // SLAM.cpp:41-42
#define HP_MATCH 8             // SLAM.cpp:43-44
#define PATCH_W (2 * HP_INIT + 1)
#define TMPL_W (2 * HP_MATCH + 1)
#define APP_PATCH_STRIDE 448   // bytes per landmark in the initPatch array (441 used)
#define APP_TMPL_STRIDE 320    // bytes per landmark in the matchPatch array (289 used)

// Everything below feeds floor / ceil / a truncating uchar cast (wrapPatch 1884-1900): this file is compiled with
// -ffp-contract=off and every expression keeps the reference's association, operation for operation, so that the
// bytes of matchPatch are the ones the reference's arithmetic produces.  pow(x, int literal) is the Visual C++ 2010
// overload pow(double, int) = repeated multiplication by squaring: (x,2) = x*x, (x,3) = x*(x*x), (x,4) = (x*x)*(x*x),
// (x,5) = x*((x*x)*(x*x)).
// undistortOnePointRW, SLAM.cpp:3224-3236
__device__ __forceinline__ void dev_undistort(const srukf_params& p, double dx, double dy, double& ux, double& uy)
{
    const double xd = (dx - p.cam_cx) * p.cam_dx, yd = (dy - p.cam_cy) * p.cam_dy;
    const double rd = sqrt(xd * xd + yd * yd);
    const double rd2 = rd * rd;
    const double d = 1 + p.cam_k1 * rd2 + p.cam_k2 * (rd2 * rd2);                                        // 3230
    const double xu = xd * d, yu = yd * d;
    ux = p.cam_cx + xu / p.cam_dx;
    uy = p.cam_cy + yu / p.cam_dy;
}
// distortOnePointRW, SLAM.cpp:3177-3213.  The 100 Newton iterations stop at the first fixed point (every later
// iteration would reproduce it bit for bit).
__device__ __forceinline__ void dev_distort(const srukf_params& p, double ux, double uy, double& ox, double& oy)
{
    const double k1 = p.cam_k1, k2 = p.cam_k2;
    const double xu = (ux - p.cam_cx) * p.cam_dx, yu = (uy - p.cam_cy) * p.cam_dy;
    const double ru = sqrt(xu * xu + yu * yu);
    const double ru2 = ru * ru;
    double rd = ru / (1 + k1 * ru * ru + k2 * (ru2 * ru2));                                              // 3184
    double rprev = __builtin_nan("");                                                                   // 2-cycle between neighbouring doubles: see srukf_project
    for (int it = 0; it < p.newton_iters; it++) {
        const double rd2 = rd * rd, rd4 = rd2 * rd2;
        const double f = rd + k1 * (rd * rd2) + k2 * (rd * rd4) - ru;                                    // 3190
        const double ff = 1.0 + 3.0 * k1 * rd * rd + 5.0 * k2 * rd4;                                     // 3191
        const double rn = rd - f / ff;                                                                   // 3192
        if (rn == rd) break;
        if (rn == rprev) { if ((p.newton_iters - it) & 1) rd = rn; break; }
        rprev = rd;
        rd = rn;
    }
    const double rdsq = rd * rd;
    double d = 1 + k1 * rd * rd + k2 * (rdsq * rdsq);                                                    // 3195
    if (d == 0.0) d = p.epsilon;
    const double xd = xu / d, yd = yu / d;
    const double vx = p.cam_cx + xd / p.cam_dx, vy = p.cam_cy + yd / p.cam_dy;
    const bool vis = (vx >= 0.0) && (vx <= p.image_w) && (vy >= 0.0) && (vy <= p.image_h);
    ox = vis ? vx : 0.0; oy = vis ? vy : 0.0;
}
__device__ __forceinline__ void dev_inv3(const double a[9], double t[9])
{
    // OpenCV's closed form for 3x3 (cofactors / determinant)
    const double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    const double id = (d != 0.0) ? 1.0 / d : 0.0;
    t[0] = (a[4] * a[8] - a[5] * a[7]) * id; t[1] = (a[2] * a[7] - a[1] * a[8]) * id; t[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    t[3] = (a[5] * a[6] - a[3] * a[8]) * id; t[4] = (a[0] * a[8] - a[2] * a[6]) * id; t[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    t[6] = (a[3] * a[7] - a[4] * a[6]) * id; t[7] = (a[1] * a[6] - a[0] * a[7]) * id; t[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
// 4x4 inverse, Gauss-Jordan with partial pivoting (the oracle's restatement of cv::Mat::inv for 4x4)
__device__ void dev_inv4(const double a[16], double out[16])
{
    double m[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { m[r][c] = a[4 * r + c]; m[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int piv = c; double big = fabs(m[c][c]);
        for (int r = c + 1; r < 4; r++) if (fabs(m[r][c]) > big) { big = fabs(m[r][c]); piv = r; }
        if (big == 0.0) { for (int e = 0; e < 16; e++) out[e] = 0.0; return; }
        if (piv != c) for (int k = 0; k < 8; k++) { const double t = m[c][k]; m[c][k] = m[piv][k]; m[piv][k] = t; }
        const double d = 1.0 / m[c][c];
        for (int k = 0; k < 8; k++) m[c][k] *= d;
        for (int r = 0; r < 4; r++) if (r != c) { const double f = m[r][c]; if (f != 0.0) for (int k = 0; k < 8; k++) m[r][k] -= f * m[c][k]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[4 * r + c] = m[r][4 + c];
}

// k_warp_patch: wrapPatch for every landmark.  One workgroup (320 threads, 289 used) per landmark: thread 0 builds the
// homography H = K (R - r n^T / d) K^-1 and the template centre, then thread (i, j) warps its pixel.
__global__ __launch_bounds__(320) void k_warp_patch(KDims d, srukf_params p, const double* __restrict__ X, const double* __restrict__ xyz,
                                                    const double* __restrict__ h, const double* __restrict__ appR, const double* __restrict__ appT,
                                                    const double* __restrict__ appPx, const unsigned char* __restrict__ initPatch,
                                                    const int* __restrict__ has_app, unsigned char* __restrict__ matchPatch)
{
    __shared__ double sh[12];          // H[9], uv[2]
    const int k = blockIdx.x;
    if (!has_app[k]) return;
    const double f1 = p.cam_f / p.cam_dx, f2 = p.cam_f / p.cam_dy;
    const double ipx = appPx[2 * k], ipy = appPx[2 * k + 1];
    if (threadIdx.x == 0) {
        const int n = d.n;
        const double rob[4] = { X[n - 4], X[n - 3], X[n - 2], X[n - 1] };
        double sn, cs;
        crt_sincos(rob[3], &sn, &cs);                                // correctly rounded (srukf_crtrig.h): the bytes below depend on the last bit
        const double Rwc[9] = { cs, -sn, 0, sn, cs, 0, 0, 0, 1 };                                     // getTransferMatrix, 1031-1037
        const double* iR = appR + 9 * k; const double* iT = appT + 3 * k;
        double C0W[16], C1W[16];
        for (int e = 0; e < 16; e++) { C0W[e] = 0.0; C1W[e] = 0.0; }
        for (int r = 0; r < 3; r++) {                                                                  // 1821-1827
            for (int c = 0; c < 3; c++) { C0W[4 * r + c] = iR[3 * r + c] + 0; C1W[4 * r + c] = Rwc[3 * r + c] + 0; }
            C0W[4 * r + 3] = iR[3 * r] * iT[0] + iR[3 * r + 1] * iT[1] + iR[3 * r + 2] * iT[2] + 0;
            C1W[4 * r + 3] = Rwc[3 * r] * rob[0] + Rwc[3 * r + 1] * rob[1] + Rwc[3 * r + 2] * rob[2] + 0;
        }
        C0W[15] = 1; C1W[15] = 1;
        double C0Wi[16], C1C0[16];
        dev_inv4(C0W, C0Wi);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { double t = 0; for (int q = 0; q < 4; q++) t += C0Wi[4 * r + q] * C1W[4 * q + c]; C1C0[4 * r + c] = t; }   // 1829
        double n0[3] = { ipx - p.cam_cx, ipy - p.cam_cy, -f1 };                                        // 1833
        const double t0[4] = { h[2 * k] - p.cam_cx, h[2 * k + 1] - p.cam_cy, -f1, 1 };                 // 1834-1835
        double t1[4];
        for (int r = 0; r < 4; r++) t1[r] = C1C0[4 * r] * t0[0] + C1C0[4 * r + 1] * t0[1] + C1C0[4 * r + 2] * t0[2] + C1C0[4 * r + 3] * t0[3];
        const double t13 = t1[3];
        double n1[3] = { t1[0] / t13, t1[1] / t13, t1[2] / t13 };                                       // 1837-1838
        double nn = sqrt(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
        for (int r = 0; r < 3; r++) n0[r] /= nn;                                                        // 1839
        nn = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
        for (int r = 0; r < 3; r++) n1[r] /= nn;                                                        // 1840
        double nv[3] = { n0[0] + n1[0], n0[1] + n1[1], n0[2] + n1[2] };                                 // 1841
        nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
        for (int r = 0; r < 3; r++) nv[r] /= nn;                                                        // 1842
        const double w4[4] = { xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], 1 };                         // 1844-1847
        double c0[4];
        for (int r = 0; r < 4; r++) c0[r] = C0Wi[4 * r] * w4[0] + C0Wi[4 * r + 1] * w4[1] + C0Wi[4 * r + 2] * w4[2] + C0Wi[4 * r + 3] * w4[3];
        const double c03 = c0[3];
        for (int r = 0; r < 3; r++) c0[r] /= c03;
        const double dd = ((-1) * nv[0]) * c0[0] + ((-1) * nv[1]) * c0[1] + ((-1) * nv[2]) * c0[2];     // 1848-1849
        const double K[9] = { f1, 0, p.cam_cx, 0, f2, p.cam_cy, 0, 0, 1 };
        double Ki[9], M[9], KM[9], H[9], Hi[9];
        dev_inv3(K, Ki);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = C1C0[4 * r + c] - (C1C0[4 * r + 3] * nv[c]) / dd;     // 1855
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double t = 0; for (int q = 0; q < 3; q++) t += K[3 * r + q] * M[3 * q + c]; KM[3 * r + c] = t; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double t = 0; for (int q = 0; q < 3; q++) t += KM[3 * r + q] * Ki[3 * q + c]; H[3 * r + c] = t; }
        double ux, uy;
        dev_undistort(p, ipx, ipy, ux, uy);                                                             // 1852
        dev_inv3(H, Hi);                                                                                // 1856
        double q[3] = { Hi[0] * ux + Hi[1] * uy + Hi[2] * 1, Hi[3] * ux + Hi[4] * uy + Hi[5] * 1, Hi[6] * ux + Hi[7] * uy + Hi[8] * 1 };
        q[0] /= q[2]; q[1] /= q[2];                                                                     // 1857
        double vx, vy;
        dev_distort(p, q[0], q[1], vx, vy);                                                             // 1861
        for (int e = 0; e < 9; e++) sh[e] = H[e];
        sh[9] = vx; sh[10] = vy;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= TMPL_W * TMPL_W) return;
    const int i = t / TMPL_W, j = t % TMPL_W;
    const double ax = sh[9] - HP_MATCH + i, ay = sh[10] - HP_MATCH + j;                                 // 1869-1870
    double bx, by;
    dev_undistort(p, ax, ay, bx, by);                                                                   // 1871
    double w0 = sh[0] * bx + sh[1] * by + sh[2] * 1, w1 = sh[3] * bx + sh[4] * by + sh[5] * 1;         // 1874
    const double w2 = sh[6] * bx + sh[7] * by + sh[8] * 1;
    w0 /= w2; w1 /= w2;                                                                                 // 1875
    double cx1, cy1;
    dev_distort(p, w0, w1, cx1, cy1);                                                                   // 1879
    cx1 -= (ipx - HP_INIT - 1);                                                                         // 1881
    cy1 -= (ipy - HP_INIT - 1);                                                                         // 1882
    const int lx = (int)floor(cx1), ly = (int)floor(cy1), rx = (int)ceil(cx1), ry = (int)ceil(cy1);     // 1884-1887
    if (lx >= 0 && rx < 2 * HP_INIT && ly >= 0 && ry < 2 * HP_INIT) {                                   // 1889
        const double rate_lx = rx - cx1, rate_ly = ry - cy1, rate_rx = 1.0 - rate_lx, rate_ry = 1.0 - rate_ly;
        const unsigned char* ip = initPatch + (size_t)k * APP_PATCH_STRIDE;
        const unsigned char ll = ip[lx * PATCH_W + ly], lr = ip[lx * PATCH_W + ry];                     // at<uchar>(row = x index, col = y index)
        const unsigned char rl = ip[rx * PATCH_W + ly], rr = ip[rx * PATCH_W + ry];
        matchPatch[(size_t)k * APP_TMPL_STRIDE + i * TMPL_W + j] =
            (unsigned char)(ll * rate_lx * rate_ly + lr * rate_lx * rate_ry + rl * rate_rx * rate_ly + rr * rate_rx * rate_ry);   // 1899-1900
    }
}

// k_associate: dataAssociation for every visible landmark.  One workgroup per landmark: the template minus its mean
// and its norm are staged in LDS; every thread takes candidate centres of the (2 half_y + 1) x (2 half_x + 1) window
// (<= 21 x 21), gates them with the Mahalanobis distance under Si^T Si and correlates; the block keeps the FIRST
// maximum in row-major order (cv::minMaxLoc).  Out: z (matchLocation), matched (isMatching), corr (maxVal).
__global__ __launch_bounds__(256) void k_associate(KDims d, srukf_params p, const unsigned char* __restrict__ image,
                                                   const double* __restrict__ h, const double* __restrict__ Si, const int* __restrict__ vis,
                                                   const int* __restrict__ has_app, const unsigned char* __restrict__ matchPatch,
                                                   double* __restrict__ z, int* __restrict__ matched, double* __restrict__ corr)
{
    __shared__ double tm[TMPL_W * TMPL_W];
    __shared__ double red[8];
    __shared__ double bestv[256];
    __shared__ int besti[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int W = p.image_w, H = p.image_h, NP = TMPL_W * TMPL_W;
    if (!vis[k] || !has_app[k]) { if (tid == 0) { matched[k] = 0; corr[k] = 0.0; z[2 * k] = 0.0; z[2 * k + 1] = 0.0; } return; }   // 1946
    // template statistics (calculateCrossCorrelation, 3151-3161): cv::mean, subtract, cv::norm
    const unsigned char* mp = matchPatch + (size_t)k * APP_TMPL_STRIDE;
    double s[1] = { 0.0 };
    for (int e = tid; e < NP; e += 256) s[0] += mp[e];
    block_sum<1>(s, red);
    const double a2 = s[0] / NP;
    double q[1] = { 0.0 };
    for (int e = tid; e < NP; e += 256) { const double v = mp[e] - a2; tm[e] = v; q[0] += v * v; }
    block_sum<1>(q, red);
    const double std2 = sqrt(q[0]);
    __syncthreads();
    const double px = h[2 * k], py = h[2 * k + 1];
    const double s00 = Si[4 * k], s01 = Si[4 * k + 1], s10 = Si[4 * k + 2], s11 = Si[4 * k + 3];
    const double p00 = s00 * s00 + s10 * s10, p01 = s00 * s01 + s10 * s11, p10 = s01 * s00 + s11 * s10, p11 = s01 * s01 + s11 * s11;   // Si^T Si, 1951
    double det = p00 * p11 - p01 * p10, i00 = 0, i01 = 0, i10 = 0, i11 = 0;                             // cv 2x2 closed-form inverse
    if (det != 0.0) { det = 1.0 / det; i00 = p11 * det; i01 = -p01 * det; i10 = -p10 * det; i11 = p00 * det; }
    int half_x = (int)ceil(2 * s00), half_y = (int)ceil(2 * s11);                                       // 1953-1954
    half_x = min(HP_INIT, max(HP_MATCH, half_x)); half_y = min(HP_INIT, max(HP_MATCH, half_y));         // 1955-1956
    const int wx = 2 * half_x + 1, wy = 2 * half_y + 1;
    const int x0 = (int)px - half_x, y0 = (int)py - half_y;
    // The pixels every candidate of this landmark can touch — the window plus the template's half width on every side, <= 37 x 37 — go to LDS once (as doubles: what the sums
    // take); the candidates' two passes over their 17 x 17 patch then read LDS instead of issuing 578 single-byte global loads each.  Same values, same order of the sums.
    // Pixels outside the image are staged as 0: a candidate whose patch would touch them is skipped by the border test below.
    constexpr int RG_W = 2 * HP_INIT + 1 + 2 * HP_MATCH;       // 37
    __shared__ double rg[RG_W * RG_W];
    const int rw = wx + 2 * HP_MATCH, rh = wy + 2 * HP_MATCH, rx0 = x0 - HP_MATCH, ry0 = y0 - HP_MATCH;
    for (int e = tid; e < rw * rh; e += 256) {
        const int yy = ry0 + e / rw, xx = rx0 + e % rw;
        rg[(e / rw) * RG_W + e % rw] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (double)image[(size_t)yy * W + xx] : 0.0;
    }
    __syncthreads();
    double bv = -1.0; int bi = 0x7fffffff;
    for (int c = tid; c < wx * wy; c += 256) {
        const int j = y0 + c / wx, i = x0 + c % wx;                                                     // row-major index of `correlation`
        double cc = 0.0;
        if (!(i < HP_MATCH || i > W - HP_MATCH - 1) && !(j < HP_MATCH || j > H - HP_MATCH - 1)) {       // 1962, 1969
            const double ex = i - px, ey = j - py;
            const double pii = (ex * i00 + ey * i10) * ex + (ex * i01 + ey * i11) * ey;                 // 1975
            if (pii < 5.99146454710798) {                                                               // 1977
                const double* roi = rg + (j - y0) * RG_W + (i - x0);                                    // 1979: image(j - HP_MATCH .., i - HP_MATCH ..) = rg(j - y0 .., i - x0 ..)
                double s1 = 0.0;
                for (int r = 0; r < TMPL_W; r++) for (int cl = 0; cl < TMPL_W; cl++) s1 += roi[r * RG_W + cl];
                const double a1 = s1 / NP;
                double q1 = 0.0, dot = 0.0;
                for (int r = 0; r < TMPL_W; r++) for (int cl = 0; cl < TMPL_W; cl++) { const double v1 = roi[r * RG_W + cl] - a1; q1 += v1 * v1; dot += v1 * tm[r * TMPL_W + cl]; }
                const double std1 = sqrt(q1);
                cc = (std1 == 0.0 || std2 == 0.0) ? 0.0 : dot / std1 / std2;                            // 3163-3166
            }
        }
        if (cc > bv) { bv = cc; bi = c; }                       // candidates of one thread come in increasing index: first maximum kept
    }
    bestv[tid] = bv; besti[tid] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            const double ov = bestv[tid + st]; const int oi = besti[tid + st];
            if (ov > bestv[tid] || (ov == bestv[tid] && oi < besti[tid])) { bestv[tid] = ov; besti[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double maxVal = bestv[0];
        const int c = besti[0];
        corr[k] = maxVal;
        if (maxVal > 0.8) {                                                                             // 1989
            z[2 * k] = (c % wx) - half_x + px;                                                          // 1991-1992
            z[2 * k + 1] = (c / wx) - half_y + py;
            matched[k] = 1;
        } else { z[2 * k] = 0.0; z[2 * k + 1] = 0.0; matched[k] = 0; }
    }
}

extern "C" {
void srukf_launch_warp_patch(hipStream_t st, KDims d, srukf_params p, const double* X, const double* xyz, const double* h,
                             const double* appR, const double* appT, const double* appPx, const unsigned char* initPatch,
                             const int* has_app, unsigned char* matchPatch)
{
    hipLaunchKernelGGL(k_warp_patch, dim3(d.N), dim3(320), 0, st, d, p, X, xyz, h, appR, appT, appPx, initPatch, has_app, matchPatch);
}
void srukf_launch_associate(hipStream_t st, KDims d, srukf_params p, const unsigned char* image, const double* h, const double* Si,
                            const int* vis, const int* has_app, const unsigned char* matchPatch, double* z, int* matched, double* corr)
{
    hipLaunchKernelGGL(k_associate, dim3(d.N), dim3(256), 0, st, d, p, image, h, Si, vis, has_app, matchPatch, z, matched, corr);
}
int srukf_app_patch_stride(void) { return APP_PATCH_STRIDE; }
int srukf_app_tmpl_stride(void) { return APP_TMPL_STRIDE; }
}  // extern "C"
