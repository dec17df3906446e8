// srukf_augment.hip — landmark augmentation on the device: the numeric part of
// integrateFeaturesInformation (SLAM.cpp:826-871) = expandMatrix + generateSigmaPoints (1123-1162) +
// passSigmaThroughMapingFunction (1177-1250) + QrAndCholeskyForInitilization (1260-1300) +
// getPermutationMatrix (1303-1334).  gfx950 only.
//
// In : X (dim), S (dim x dim upper), K new features at distorted pixels uv[K][2].
// Out: X_new, S_new of dimension dimn = dim + 6K in NORMAL order (new landmarks before the robot block).
//
// The reference builds the (dim + 3K) x L sigma matrix, maps it, and runs two Householder QRs:
// S_dis = R(QR(A)), A[i] = wi_sr (sigma_out_{i+1} - sigma_out_0)^T  (2Na x dimn), then
// S_new = R(QR(Pi S_dis Pi^T)).  R^T R of the second QR is Pi (A^T A) Pi^T, and P = S^T S is what the
// filter consumes, so the device forms the Gram matrix A^T A with the MFMA tile kernel, permutes it
// and factors it with the same blocked GMW as every update (rank dim + 3K: the K duplicated anchors
// are pivoted with EPSILON, exactly like the null directions of every later refactor).
// The sigma matrix itself is never materialised: only the 3K mapped angle rows (ang) are stored.
#include "srukf_device.h"
#include "srukf_tiles.h"

// sigma column i of the augmented state: which row of the sqrt matrix it is built from, and its sign
__device__ __forceinline__ void aug_column(int i, int Na, double gamma, int& row, double& g)
{
    if (i == 0) { row = -1; g = 0.0; }
    else if (i <= Na) { row = i - 1; g = gamma; }
    else { row = i - 1 - Na; g = (-1) * gamma; }
}
// element c of sigma column i for the OLD state part (c < dim): mu*1 + e*g + 0 (addWeighted, SLAM.cpp:1159-1160)
__device__ __forceinline__ double aug_old(const double* __restrict__ X, const double* __restrict__ S, int ld, int dim, int row, double g, int c)
{
    if (row < 0) return X[c];
    const double e = (row < dim && c >= row) ? S[(size_t)row * ld + c] : 0.0;      // S upper triangular; new-noise rows do not touch the old state
    return X[c] * 1 + e * g + 0;
}

// k_aug_map: passSigmaThroughMapingFunction (1201-1243).  Thread (i, id): sigma column i, new feature id.
//   ang[i][3 id + d] = (theta, phi, rho) of the new landmark as seen from the robot pose of sigma column i.
__global__ __launch_bounds__(256) void k_aug_map(srukf_params p, int dim, int ld, int K, int Na, double gamma,
                                                 const double* __restrict__ X, const double* __restrict__ S,
                                                 const double* __restrict__ uv, double* __restrict__ ang)
{
    const int L = 2 * Na + 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L * K) return;
    const int i = t / K, id = t % K;
    int row; double g;
    aug_column(i, Na, gamma, row, g);
    double pos[4];
#pragma unroll
    for (int e = 0; e < 4; e++) pos[e] = aug_old(X, S, ld, dim, row, g, dim - 4 + e);          // 1203-1204
    const int in0 = dim + 3 * id;                                                                // 1208
    // new-feature rows of mu / sr: (u, v, rho0) with sqrt diag (sigma_measure, sigma_measure, sigma_rho), 847-858
    double uvd_x = uv[2 * id], uvd_y = uv[2 * id + 1], rho = p.rho0;
    if (row >= 0) {
        uvd_x = uvd_x * 1 + ((row == in0) ? p.sigma_measure : 0.0) * g + 0;
        uvd_y = uvd_y * 1 + ((row == in0 + 1) ? p.sigma_measure : 0.0) * g + 0;
        rho   = rho * 1 + ((row == in0 + 2) ? p.sigma_rho : 0.0) * g + 0;
    }
    // undistortOnePointRW, 3224-3236
    const double xd = (uvd_x - p.cam_cx) * p.cam_dx, yd = (uvd_y - p.cam_cy) * p.cam_dy;
    const double rd = sqrt(xd * xd + yd * yd);
    const double dd = 1 + p.cam_k1 * (rd * rd) + p.cam_k2 * ((rd * rd) * (rd * rd));
    const double uvu_x = p.cam_cx + (xd * dd) / p.cam_dx, uvu_y = p.cam_cy + (yd * dd) / p.cam_dy;
    const double f1 = p.cam_f / p.cam_dx, f2 = p.cam_f / p.cam_dy;
    const double h0 = (uvu_y - p.cam_cx) / f1, h1 = (uvu_x - p.cam_cy) / f2, h2 = 1.0;         // 1218 -> 3360-3363 (x/y swap)
    double sn, cs;
    sincos(pos[3], &sn, &cs);                                                                    // getTransferMatrix, 1031-1037
    const double w0 = cs * h0 + (-sn) * h1 + 0.0 * h2;                                            // 1219 -> 3386
    const double w1 = sn * h0 + cs * h1 + 0.0 * h2;
    const double w2 = 0.0 * h0 + 0.0 * h1 + 1.0 * h2;
    double* o = ang + (size_t)i * (3 * K) + 3 * id;
    o[0] = atan2(w0, w2);                                                                        // 1220 -> 3411-3419
    o[1] = atan2(-w1, sqrt(w0 * w0 + w2 * w2));
    o[2] = rho;
}

// k_aug_x: weighted mean of the mapped angles (1232-1241: mu = state*wm0, then mu = state*wi + mu*1 + 0 in column
// order) and the new state in normal order, X_new[r] = x_dis[perm[r]] (1245-1249, 1293).
__global__ __launch_bounds__(256) void k_aug_x(int dim, int K, int Na, double wm0, double wi, const double* __restrict__ X,
                                               const double* __restrict__ ang, const int* __restrict__ perm,
                                               double* __restrict__ mu_ang, double* __restrict__ Xn, int dimn, int ldn)
{
    const int L = 2 * Na + 1;
    for (int j0 = 0; j0 < 3 * K; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < 3 * K) {
            double m = ang[j] * wm0 + 0.0 * 0 + 0;
            for (int i = 1; i < L; i++) m = ang[(size_t)i * (3 * K) + j] * wi + m * 1 + 0;
            mu_ang[j] = m;
        }
    }
    __syncthreads();                                           // mu_ang (global) written above is read below by other threads
    for (int r = threadIdx.x; r < ldn; r += 256) {
        double v = 0.0;
        if (r < dimn) {
            const int c = perm[r];
            if (c < dim) v = X[c];
            else if (c < dim + 3 * K) v = mu_ang[c - dim];
            else v = X[dim - 4 + (c - dim - 3 * K) % 3];                                         // repeat(cam_position, K), 1247
        }
        Xn[r] = v;
    }
}

// k_aug_build: the QR matrix of QrAndCholeskyForInitilization (1268-1275), rows_p x ldn, row i = sigma column i+1,
// in the disordered column layout [old state | new angles 3K | new anchors 3K]; zero padding.
__global__ __launch_bounds__(256) void k_aug_build(int dim, int ld, int K, int Na, double gamma, double wi_sr,
                                                   const double* __restrict__ X, const double* __restrict__ S,
                                                   const double* __restrict__ ang, double* __restrict__ A, int rows_p, int dimn, int ldn)
{
    const int i = blockIdx.x;
    int row; double g;
    aug_column(i + 1, Na, gamma, row, g);
    for (int c = threadIdx.x; c < ldn; c += 256) {
        double v = 0.0;
        if (i < 2 * Na && c < dimn) {
            if (c < dim) v = aug_old(X, S, ld, dim, row, g, c) - X[c];
            else if (c < dim + 3 * K) v = ang[(size_t)(i + 1) * (3 * K) + (c - dim)] - ang[c - dim];
            else { const int e = dim - 4 + (c - dim - 3 * K) % 3; v = aug_old(X, S, ld, dim, row, g, e) - X[e]; }
            v = wi_sr * v;
        }
        A[(size_t)i * ldn + c] = v;
    }
}

// k_gram: G = A^T A on the upper 32x32 tiles (A: rows_p x ld, K-major, rows_p a multiple of 16); one workgroup per
// tile, 4-way split-K over the rows.  grid = (T, T), tiles below the diagonal exit.
__global__ __launch_bounds__(256) void k_gram(int rows_p, int ld, const double* __restrict__ A, double* __restrict__ G)
{
    if (blockIdx.x < blockIdx.y) return;
    __shared__ double red[3][64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    d4 acc[2][2];
    zero_acc(acc);
    const int ng = rows_p >> 4;
    const int g0 = (ng * wv) >> 2, g1 = (ng * (wv + 1)) >> 2;
    tile32_tn<false>(acc, A, ld, A, ld, m0, n0, g0 << 4, g1 << 4, lane);
    splitk_reduce(acc, red, wv, lane);
    if (wv != 0) return;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + n0 + 16 * b + lr] = acc[a][b][t];
}

extern "C" {
void srukf_launch_aug_map(hipStream_t st, srukf_params p, int dim, int ld, int K, int Na, double gamma,
                          const double* X, const double* S, const double* uv, double* ang)
{
    const int L = 2 * Na + 1;
    hipLaunchKernelGGL(k_aug_map, dim3((L * K + 255) / 256), dim3(256), 0, st, p, dim, ld, K, Na, gamma, X, S, uv, ang);
}
void srukf_launch_aug_x(hipStream_t st, int dim, int K, int Na, double wm0, double wi, const double* X, const double* ang,
                        const int* perm, double* mu_ang, double* Xn, int dimn, int ldn)
{
    hipLaunchKernelGGL(k_aug_x, dim3(1), dim3(256), 0, st, dim, K, Na, wm0, wi, X, ang, perm, mu_ang, Xn, dimn, ldn);
}
void srukf_launch_aug_build(hipStream_t st, int dim, int ld, int K, int Na, double gamma, double wi_sr, const double* X, const double* S,
                            const double* ang, double* A, int rows_p, int dimn, int ldn)
{
    hipLaunchKernelGGL(k_aug_build, dim3(rows_p), dim3(256), 0, st, dim, ld, K, Na, gamma, wi_sr, X, S, ang, A, rows_p, dimn, ldn);
}
void srukf_launch_gram(hipStream_t st, int rows_p, int ld, const double* A, double* G)
{
    hipLaunchKernelGGL(k_gram, dim3(ld / 32, ld / 32), dim3(256), 0, st, rows_p, ld, A, G);
}
}  // extern "C"
