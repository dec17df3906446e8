// srukf_predict.hip — sigma points, motion model, structured re-triangularisation, camera
// projection of every sigma point, measurement statistics (h, Si, robot rows of Pxy).
// gfx950 only.  See srukf_device.h for the HBM layout.
#include "srukf_device.h"
#include "srukf_meas.h"
#include "srukf_rank.h"
#include "srukf_motion.h"
#include "srukf_motion_reduce.h"

// control from two odometry poses (SLAM.cpp:1444-1458): srukf_motion_control_a in srukf_device.h
__device__ __forceinline__ void srukf_motion_control(const srukf_params& p, const double* o, double (&ut)[3], double (&mt)[3])
{
    const double a[4] = { p.a1, p.a2, p.a3, p.a4 };
    srukf_motion_control_a(a, o, ut, mt);
}

// ------------------------------------------------------------------------------------------------
// Motion step: predictMotion numeric tail (SLAM.cpp:1430-1465) fused:
//   control Ut/Mt from two odometry poses (1444-1458);
//   robot rows of all L sigma points through the odometry model (generateSigmaPoints 1148-1162
//   restricted to rows n-4..n-1 and the control-noise rows, passSigmaThroughMotionFunction
//   1476-1532); weighted mean -> X[n-4:n] (1526-1531);
//   QrAndCholeskyForMotion (1539-1555) in its structured form: the QR matrix
//   A = wi_sr*(sigma_{i+1}-sigma_0)^T has A[:, :n-4] = (1/sqrt2) [E; -E] S[:n-4,:n-4], so
//   R11 = S11 (unchanged), R12[i] = wi_sr/sqrt2 * (dev+_i - dev-_i), and R22^T R22 = C^T C with C the
//   (n+14) x 4 matrix of residual rows wi_sr/sqrt2 * (dev+_i + dev-_i) (i < n-4) and
//   wi_sr*dev+-_i (n-4 <= i < Na).  R22 = chol(C^T C): a 4x4 Gram matrix from ONE block reduction
//   (backward stable for P = S^T S, which is what the filter consumes; its rows equal the
//   Householder R's up to sign).  Only the last four columns of S are rewritten: O(n) instead of
//   the reference's 2n^2(2Na - n/3) flop Householder QR (DESIGN.md "Motion step").
// One workgroup of 512 threads, one pass over the sigma directions, ONE block reduction (16-lane DPP rows, then 32 row sums
// per value through LDS: fixed order, 3.6 KB of LDS instead of a 57 KB tree).
//
// REPLAY = false: k_motion, the step-wise API (odometry pair from the host) and the exact path: results go straight into X, S.
// REPLAY = true : workgroup 0 of k_project_motion (staged replay).  The projection threads of the same launch evaluate the
//   robot part of their own sigma points from the state BEFORE the motion step, so nothing they read may change under
//   them: the new robot mean goes to fs->Xr1 and the new last four columns of S to Cm (n x 4; k_gain commits them, the dX
//   job of the k_syrk launch adds the update to Xr1).  The control comes from fs->ctl (srukf_prepare_control).
// ------------------------------------------------------------------------------------------------
template <bool REPLAY>
__device__ __forceinline__ void motion_body(const KDims& d, const KWeights& w, const srukf_params& p,
                                            double* __restrict__ X, double* __restrict__ S,
                                            double* __restrict__ sigR, double* __restrict__ Cm,
                                            FrameScalars* __restrict__ fs,
                                            const double* __restrict__ odo_seq, const double* odo_pair, const RankArgs& ra, double* smem)
{
    double (*part)[32] = (double (*)[32])smem;                 // [14][32]
    double* red = smem + 14 * 32;                              // [16]
    double* sh = red + 16;                                     // [24]
    const int tid = threadIdx.x;
    const int n = d.n, Na = d.Na, L = d.L, ld = d.np;
    if (!odo_pair && fs->frozen) return;                       // staged replay behind a flagged frame: nothing to compute from

    STAMP(0);
    // Replay path with the rank-aware form: the directions are walked in PERMUTED order — kept rows (positions < r, read from the
    // permuted copy A), then the five noise rows, then the structurally null rows.  A null row of S has nothing in the robot
    // columns, so both its sigma points are the centre point, bit for bit: they are filled in without the two sincos — half of
    // all directions — and, walked in this order, whole waves do nothing else.
    const bool permuted = REPLAY && ra.A != nullptr;
    if constexpr (REPLAY) __builtin_amdgcn_s_setprio(3);        // this workgroup shares its CU with projection waves: the chain goes first
    // the S rows of this thread's directions are requested before the control block: their round trip then overlaps
    // thread 0's dependent chain  frame counter -> odometry pair  instead of following it
    // prefetched slots: step-wise API the first 3 * 512 directions (N <= 254); replay only the first 512 (in permuted order the
    // later passes are mostly fills, and the kernel shares its register budget with the projection threads); the rest load in the loop
    constexpr int MAXIT = REPLAY ? 1 : 3;
    double2 pre[MAXIT][2];
    int pidx[MAXIT];                                           // direction (row of the augmented sqrt matrix) of this slot; -1: none
    int prow[MAXIT];                                           // step-wise API, rank-aware: row of the shadow copy that mirrors S row i (or none)
    bool pnull[MAXIT];                                         // structurally null row: fill only
    auto slot = [&](const int pos, int& i, int& arow, bool& isnull, double2& u0, double2& u1) {
        u0 = make_double2(0.0, 0.0); u1 = u0; arow = 0x7fffffff; isnull = false; i = -1;
        if (pos >= Na) return;
        if (permuted) {
            if (pos < ra.r) {
                i = ra.perm[pos];
                const double* ap = ra.A + (size_t)pos * ld + (ra.r - 4);       // r - 4 may be odd: four 8-byte loads
                u0 = make_double2(ap[0], ap[1]); u1 = make_double2(ap[2], ap[3]);
            } else if (pos < ra.r + 5) i = n + (pos - ra.r);
            else { i = ra.perm[pos - 5]; isnull = true; }
            return;
        }
        i = pos;
        arow = (ra.A && i < n - 4) ? ra.iperm[i] : 0x7fffffff;
        if (i < n) {
            // S[i][n-4..n-1]: 16-byte aligned (n-4 = 6N is even, ld a multiple of 64); the strictly lower
            // triangle of S is kept zero, so rows inside the robot block need no masking
            const double2* sp2 = reinterpret_cast<const double2*>(S + (size_t)i * ld + (n - 4));
            u0 = sp2[0]; u1 = sp2[1];
        }
    };
#pragma unroll
    for (int it = 0; it < MAXIT; it++) slot(tid + it * 512, pidx[it], prow[it], pnull[it], pre[it][0], pre[it][1]);
    // ---- control (SLAM.cpp:1444-1458) ----
    bool freeze = false;
    if (tid == 0) {
        if constexpr (REPLAY) {
            for (int q = 0; q < 3; q++) { sh[q] = fs->ctl[q]; sh[3 + q] = fs->ctl[5 + q]; }
            sh[10] = fs->ctl[3]; sh[11] = fs->ctl[4];
        } else {
            const double* o = odo_pair ? odo_pair : (odo_seq + 3 * fs->frame);
            double ut[3], mt[3];
            srukf_motion_control(p, o, ut, mt);
            for (int q = 0; q < 3; q++) { sh[q] = ut[q]; sh[3 + q] = mt[q]; }
            sh[10] = cos(ut[2]); sh[11] = sin(ut[2]);
        }
        for (int q = 0; q < 4; q++) sh[6 + q] = X[n - 4 + q];
        for (int q = 0; q < 3; q++) { fs->Ut[q] = sh[q]; fs->Mt[q] = sh[3 + q]; }
        for (int q = 0; q < 4; q++) fs->Xr0[q] = sh[6 + q];
        // staged replay: the previous frame's refactorisation was flagged (theta clamp / abandoned launch) -> remember which
        if (!odo_pair && fs->clamp_rows > 0) {
            if (fs->clamp_frame == 0x7fffffff) fs->clamp_frame = fs->frame - 1;
            freeze = true;
        }
        if (odo_pair) fs->frozen = 0;                          // step-wise API: the host has taken over
    }
    __syncthreads();
    // written only now: every wave has read the flag at the top before it reached this barrier, so the early return there is
    // uniform over the workgroup.  The rest of this frame and all later frames of the run return at once.
    if (freeze) fs->frozen = 1;
    const double rot1 = sh[0], trans = sh[1], rot2 = sh[2];
    const double xr[4] = { sh[6], sh[7], sh[8], sh[9] };
    const double crot2 = sh[10], srot2 = sh[11];
    const double mts[3] = { sh[3], sh[4], sh[5] };

    STAMP(1);
    // ---- one fused pass over the Na sigma directions: each thread takes direction i, reads row i of the
    //      augmented sqrt matrix once and pushes BOTH sigma points (mu +- gamma*row) through the motion
    //      model; the deviations from sigma_0 feed R12 and the Gram matrix of the residual rows ----
    // sigma_0 (centre point through the motion model, no noise): every thread needs it, so every thread computes it
    double s0[4], c0s, s0s;
    const MotionCtl mc = { rot1, trans, rot2, crot2, srot2 };
    srukf_motion_centre(mc, xr, s0, c0s, s0s);
    const double k2 = w.wi_sr * 0.70710678118654752440;
    double acc[14];                // [0..3] weighted mean of the robot rows, [4..13] upper triangle of C^T C
#pragma unroll
    for (int q = 0; q < 14; q++) acc[q] = 0.0;
    if (tid == 0) {
        double* o = sigR;
        o[0] = s0[0]; o[1] = s0[1]; o[2] = s0[2]; o[3] = s0[3]; o[4] = c0s; o[5] = s0s; o[6] = 0; o[7] = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = w.wm0 * s0[e];
    }
    auto direction = [&](const int i, const double2 u0, const double2 u1, const int arow) {
        double srow[4] = { 0, 0, 0, 0 }, mnoise[3] = { 0, 0, 0 };
        if (i < n) {
            srow[0] = u0.x; srow[1] = u0.y; srow[2] = u1.x; srow[3] = u1.y;
        } else if (i < n + 3) {
            mnoise[i - n] = mts[i - n];                        // control-noise rows (sr = blockdiag(S, Mt, Qt))
        }
        double dev[2][4];
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            const double gs = sg ? -w.gamma : w.gamma;
            double r[4], c2, s2;
            srukf_motion_point(mc, xr, srow, mnoise, gs, r, c2, s2);
            double4* o = reinterpret_cast<double4*>(sigR + (size_t)(1 + sg * Na + i) * 8);
            o[0] = make_double4(r[0], r[1], r[2], r[3]);
            o[1] = make_double4(c2, s2, 0.0, 0.0);
#pragma unroll
            for (int e = 0; e < 4; e++) { acc[e] += w.wi * r[e]; dev[sg][e] = r[e] - s0[e]; }
            if constexpr (REPLAY) __builtin_amdgcn_sched_barrier(0);   // one sigma point after the other: the kernel shares its register budget with the projection threads
        }
        if (i < n - 4) {
            double c[4];
#pragma unroll
            for (int e = 0; e < 4; e++) c[e] = k2 * (dev[0][e] + dev[1][e]);
            if constexpr (REPLAY) {
                *reinterpret_cast<double4*>(Cm + (size_t)i * 4) = make_double4(k2 * (dev[0][0] - dev[1][0]), k2 * (dev[0][1] - dev[1][1]),
                                                                               k2 * (dev[0][2] - dev[1][2]), k2 * (dev[0][3] - dev[1][3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) S[(size_t)i * ld + (n - 4 + e)] = k2 * (dev[0][e] - dev[1][e]);
                if (arow < ra.r) {
#pragma unroll
                    for (int e = 0; e < 4; e++) ra.A[(size_t)arow * ld + (ra.r - 4 + e)] = k2 * (dev[0][e] - dev[1][e]);
                }
            }
            int q = 4;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++) acc[q++] += c[a] * c[b];
        } else {
            int q = 4;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++) acc[q++] += (w.wi_sr * dev[0][a]) * (w.wi_sr * dev[0][b]) + (w.wi_sr * dev[1][a]) * (w.wi_sr * dev[1][b]);
        }
    };
    // a structurally null row: srukf_motion_point with a zero row IS srukf_motion_centre (same operations on the same values)
    auto fill = [&](const int i) {
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            double4* o = reinterpret_cast<double4*>(sigR + (size_t)(1 + sg * Na + i) * 8);
            o[0] = make_double4(s0[0], s0[1], s0[2], s0[3]);
            o[1] = make_double4(c0s, s0s, 0.0, 0.0);
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] += w.wi * s0[e];
        }
        *reinterpret_cast<double4*>(Cm + (size_t)i * 4) = make_double4(0.0, 0.0, 0.0, 0.0);
    };
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        if (pidx[it] < 0) continue;
        if (pnull[it]) fill(pidx[it]); else direction(pidx[it], pre[it][0], pre[it][1], prow[it]);
    }
    for (int pos = tid + MAXIT * 512; pos < Na; pos += 512) {   // the slots that were not prefetched: plain loads
        int i, arow; bool isnull; double2 u0, u1;
        slot(pos, i, arow, isnull, u0, u1);
        if (isnull) fill(i); else direction(i, u0, u1, arow);
    }
    motion_finish<REPLAY, 512>(acc, s0, d, w, X, S, sigR, Cm, fs, ra, smem);
    STAMP(5);
}
// The table of robot poses for ALL directions from the state as it stands (S in state order): the first frame of a staged run in
// "table" mode, whose predecessor's tail did not prepare it.  One thread per direction, thread 0 also the centre point.
__global__ __launch_bounds__(256) void k_sigr_rows(KDims d, KWeights w, const double* __restrict__ X, const double* __restrict__ S,
                                                   double* __restrict__ sigR, const FrameScalars* __restrict__ fs, const int* __restrict__ iperm, int rkeep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = d.n, Na = d.Na, ld = d.np;
    if (i >= Na || fs->frozen) return;
    const MotionCtl mc = { fs->ctl[0], fs->ctl[1], fs->ctl[2], fs->ctl[3], fs->ctl[4] };
    double xr[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xr[e] = X[n - 4 + e];
    if (i == 0) {
        double s0[4], c0s, s0s;
        srukf_motion_centre(mc, xr, s0, c0s, s0s);
        double4* o = reinterpret_cast<double4*>(sigR);
        o[0] = make_double4(s0[0], s0[1], s0[2], s0[3]); o[1] = make_double4(c0s, s0s, 0.0, 0.0);
    }
    double srow[4] = { 0, 0, 0, 0 }, mnoise[3] = { 0, 0, 0 };
    if (i < n) {
        const double2* sp2 = reinterpret_cast<const double2*>(S + (size_t)i * ld + (n - 4));
        const double2 u0 = sp2[0], u1 = sp2[1];
        srow[0] = u0.x; srow[1] = u0.y; srow[2] = u1.x; srow[3] = u1.y;
    } else if (i < n + 3) mnoise[i - n] = fs->ctl[5 + (i - n)];
    const bool isnull = i < n && iperm[i] >= rkeep;             // a structurally null row: both points ARE the centre point (same bits: NullSkip rests on it)
#pragma unroll
    for (int sg = 0; sg < 2; sg++) {
        double r[4], c2, s2;
        if (isnull) { srukf_motion_centre(mc, xr, r, c2, s2); }
        else srukf_motion_point(mc, xr, srow, mnoise, sg ? -w.gamma : w.gamma, r, c2, s2);
        double4* o = reinterpret_cast<double4*>(sigR + (size_t)(1 + sg * Na + i) * 8);
        o[0] = make_double4(r[0], r[1], r[2], r[3]); o[1] = make_double4(c2, s2, 0.0, 0.0);
    }
}

__global__ __launch_bounds__(512) void k_motion(KDims d, KWeights w, srukf_params p,
                                                double* __restrict__ X, double* __restrict__ S,
                                                double* __restrict__ sigR, double* __restrict__ Cmat,
                                                FrameScalars* __restrict__ fs,
                                                const double* __restrict__ odo_seq, const double* odo_pair, const RankArgs ra)
{
    __shared__ double sm[MOTION_SM_DOUBLES];
    motion_body<false>(d, w, p, X, S, sigR, Cmat, fs, odo_seq, odo_pair, ra, sm);
}

// ------------------------------------------------------------------------------------------------
// Projection: passSigmaThroughMesaurementFunction (SLAM.cpp:1615-1674).  Flat item g = (direction ii, landmark k):
// ii = 0 is the centre column, ii >= 1 handles the +/- pair built from row i = ii-1 of the augmented sqrt matrix, so
// every S row is read once and DZ[i] = Z[1+i] - Z[1+Na+i] falls out for the cross-covariance contraction.
// INLINE_ROBOT = false (k_project): the robot part of every sigma point comes from the table sigR (k_motion ran before).
// INLINE_ROBOT = true (k_project_motion): the thread pushes the robot rows of its two sigma points through the motion
//   model itself — the same device functions on the same inputs as the motion workgroup, so the same bits as its table —
//   from the state before the motion step and the prepared control fs->ctl; it waits for nothing.
// ------------------------------------------------------------------------------------------------
template <bool INLINE_ROBOT>
__device__ __forceinline__ void project_dir(const int ii, const int k, const KDims& d, const KWeights& w, const srukf_params& p,
                                            const double* __restrict__ X, const double* __restrict__ S,
                                            const double* __restrict__ sigR,
                                            double* __restrict__ Z, double* __restrict__ DZ, const FrameScalars* __restrict__ fs,
                                            const int* __restrict__ dzperm);
template <bool INLINE_ROBOT>
__device__ __forceinline__ void project_item(const int g, const KDims& d, const KWeights& w, const srukf_params& p,
                                             const double* __restrict__ X, const double* __restrict__ S,
                                             const double* __restrict__ sigR,
                                             double* __restrict__ Z, double* __restrict__ DZ, const FrameScalars* __restrict__ fs,
                                             const int* __restrict__ dzperm = nullptr)
{
    // flat (direction, landmark) index: no idle lanes when N is not a multiple of the wave size (N = 200: 78 % -> 100 %)
    const int ii = g / d.N, k = g - ii * d.N;
    if (ii > d.Na) return;
    project_dir<INLINE_ROBOT>(ii, k, d, w, p, X, S, sigR, Z, DZ, fs, dzperm);
}
// landmark k under direction ii (0: the centre point, ii >= 1: the +/- pair of row ii - 1 of the augmented sqrt matrix)
template <bool INLINE_ROBOT>
__device__ __forceinline__ void project_dir(const int ii, const int k, const KDims& d, const KWeights& w, const srukf_params& p,
                                            const double* __restrict__ X, const double* __restrict__ S,
                                            const double* __restrict__ sigR,
                                            double* __restrict__ Z, double* __restrict__ DZ, const FrameScalars* __restrict__ fs,
                                            const int* __restrict__ dzperm)
{
    const int n = d.n, Na = d.Na, ld = d.np, mp = d.mp;
    const double f1 = p.cam_f / p.cam_dx, f2 = p.cam_f / p.cam_dy;
    double base[6];
#pragma unroll
    for (int e = 0; e < 6; e++) base[e] = X[6 * k + e];
    MotionCtl mc = { 0, 0, 0, 1, 0 };
    double xr[4] = { 0, 0, 0, 0 }, mts[3] = { 0, 0, 0 };
    if constexpr (INLINE_ROBOT) {
        mc.rot1 = fs->ctl[0]; mc.trans = fs->ctl[1]; mc.rot2 = fs->ctl[2]; mc.crot2 = fs->ctl[3]; mc.srot2 = fs->ctl[4];
#pragma unroll
        for (int e = 0; e < 3; e++) mts[e] = fs->ctl[5 + e];
#pragma unroll
        for (int e = 0; e < 4; e++) xr[e] = X[n - 4 + e];
    }

    if (ii == 0) {
        double ox, oy;
        if constexpr (INLINE_ROBOT) {
            double s0[4], c0s, s0s;
            srukf_motion_centre(mc, xr, s0, c0s, s0s);
            srukf_project(p, f1, f2, base, s0[0], s0[1], s0[2], c0s, s0s, 0.0, 0.0, ox, oy);
        } else {
            const double* r = sigR;
            srukf_project(p, f1, f2, base, r[0], r[1], r[2], r[4], r[5], 0.0, 0.0, ox, oy);
        }
        *reinterpret_cast<double2*>(Z + 2 * k) = make_double2(ox, oy);
        return;
    }
    const int i = ii - 1;
    double dev[6] = { 0, 0, 0, 0, 0, 0 };
    double srow[4] = { 0, 0, 0, 0 }, mnoise[3] = { 0, 0, 0 };
    if (i < n) {
#pragma unroll
        for (int e = 0; e < 6; e++) { const int col = 6 * k + e; dev[e] = (col >= i) ? S[(size_t)i * ld + col] : 0.0; }
        if constexpr (INLINE_ROBOT) {
            const double2* sp2 = reinterpret_cast<const double2*>(S + (size_t)i * ld + (n - 4));
            const double2 u0 = sp2[0], u1 = sp2[1];
            srow[0] = u0.x; srow[1] = u0.y; srow[2] = u1.x; srow[3] = u1.y;
        }
    } else if (INLINE_ROBOT && i < n + 3) {
        mnoise[i - n] = mts[i - n];
    }
    double e0 = 0.0, e1 = 0.0;                       // pixel-noise sigma rows n+3, n+4 (Qt on the sqrt diagonal)
    if (i == n + 3) e0 = p.sigma_measure; else if (i == n + 4) e1 = p.sigma_measure;
    double zp[2], zm[2];
#pragma unroll
    for (int sgn = 0; sgn < 2; sgn++) {
        const double gq = sgn ? -w.gamma : w.gamma;
        const int c = sgn ? (1 + Na + i) : (1 + i);
        double ox, oy;
        if constexpr (INLINE_ROBOT) {
            double feat[6], q0, q1;
            srukf_sigma_feat(base, dev, e0, e1, gq, feat, q0, q1);
            double r[4], c2, s2;
            srukf_motion_point(mc, xr, srow, mnoise, gq, r, c2, s2);
            srukf_project(p, f1, f2, feat, r[0], r[1], r[2], c2, s2, q0, q1, ox, oy);
        } else {
            srukf_project_sigma(p, f1, f2, base, dev, e0, e1, gq, sigR + (size_t)c * 8, ox, oy);
        }
        *reinterpret_cast<double2*>(Z + (size_t)c * mp + 2 * k) = make_double2(ox, oy);
        if (sgn) { zm[0] = ox; zm[1] = oy; } else { zp[0] = ox; zp[1] = oy; }
    }
    // dzperm ("table" mode): the rows of DZ in the permuted order of the rank-aware form, the K order of k_pxy2
    if (i < n) *reinterpret_cast<double2*>(DZ + (size_t)(dzperm ? dzperm[i] : i) * mp + 2 * k) = make_double2(zp[0] - zm[0], zp[1] - zm[1]);
}
__global__ __launch_bounds__(256) void k_project(KDims d, KWeights w, srukf_params p,
                                                 const double* __restrict__ X, const double* __restrict__ S,
                                                 const double* __restrict__ sigR,
                                                 double* __restrict__ Z, double* __restrict__ DZ, const FrameScalars* __restrict__ fs)
{
    project_item<false>(blockIdx.x * 256 + threadIdx.x, d, w, p, X, S, sigR, Z, DZ, fs);
}
// Replay path: workgroup 0 is the motion step of the frame (a one-workgroup latency chain of ~14 us that the whole GPU used to
// wait for), every other workgroup projects 512 (direction, landmark) items and needs nothing from it.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_project_motion(KDims d, KWeights w, srukf_params p,
                                                        double* __restrict__ X, double* __restrict__ S,
                                                        double* __restrict__ sigR, double* __restrict__ Cm,
                                                        double* __restrict__ Z, double* __restrict__ DZ, FrameScalars* __restrict__ fs, const RankArgs ra)
{
    __shared__ double sm[MOTION_SM_DOUBLES];
    if (blockIdx.x == 0) { motion_body<true>(d, w, p, X, S, sigR, Cm, fs, nullptr, nullptr, ra, sm); return; }
    project_item<true>((blockIdx.x - 1) * 512 + threadIdx.x, d, w, p, X, S, sigR, Z, DZ, fs);
}
// "Table" mode of the replay (rank-aware form): workgroup 0 reduces the prepared table of robot poses to the motion step's results,
// every other workgroup is k_project as in the step-wise API (256 items from the table: 86 VGPRs, five waves per SIMD).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_project_table(KDims d, KWeights w, srukf_params p,
                                                       double* __restrict__ X, double* __restrict__ S,
                                                       double* __restrict__ sigR, double* __restrict__ Cm,
                                                       double* __restrict__ Z, double* __restrict__ DZ, FrameScalars* __restrict__ fs, const RankArgs ra,
                                                       const NullSkip ns)
{
    __shared__ double sm[MOTION_SM_DOUBLES];
    if (blockIdx.x == 0) { motion_reduce_body<256>(d, w, X, S, sigR, Cm, fs, ra, sm); return; }
    const int g = (blockIdx.x - 1) * 256 + threadIdx.x;
    const int* dzperm = ra.dzperm ? ra.iperm : nullptr;
    if (!ns.dirs) { project_item<false>(g, d, w, p, X, S, sigR, Z, DZ, fs, dzperm); return; }
    // NullSkip (srukf_device.h): the full directions x all landmarks, then one item per structurally null direction (its own landmark)
    const int nf = (1 + ns.nfull) * d.N;                        // the centre point first
    if (g < nf) { const int q = g / d.N; project_dir<false>(q ? 1 + ns.dirs[q - 1] : 0, g - q * d.N, d, w, p, X, S, sigR, Z, DZ, fs, dzperm); }
    else if (g - nf < ns.nnull) { const int i = ns.nulls[g - nf]; project_dir<false>(1 + i, i / 6, d, w, p, X, S, sigR, Z, DZ, fs, dzperm); }
}

// ------------------------------------------------------------------------------------------------
// Measurement statistics (QrAndCholeskyForMeasurement / calculateOneFeatureCovariance,
// SLAM.cpp:1700-1775, and the robot rows of calculateOneFeatureCrossCovariance, 2028-2037).
// One pass over Z.  All sums use deviations from the centre column Z_0, so nothing depends on the
// mean h inside the pass:
//   s[0..1]  = sum_{c>=1} (Z_c - Z_0)                                  -> h = Z_0 + wi * s   (sum of weights = 1)
//   s[2..4]  = sum_{c>=1} a_c^2, a_c b_c, b_c^2   (a, b = wi_sr * (Z_c - Z_0))
//   s[5..12] = sum_c w_c (sigR_c[e] - Xr[e]) (Z_c[col] - Z_0[col])     e = 0..3, col = x, y
// k_meas_partial: block (32 landmarks x 8 row sub-slices), grid.y = MEAS_SLICES row slices; writes
// per-slice partial sums (fixed order => run-to-run deterministic).  k_meas_final reduces the slices
// and finishes h, Si (GSL Householder sign rule), visible, PxyR.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_meas_partial(KDims d, KWeights w, const double* __restrict__ X,
                                                      const double* __restrict__ sigR, const double* __restrict__ Z, double* __restrict__ part)
{
    __shared__ double sm[MEAS_SM_DOUBLES];
    meas_partial_job<false>(d, w, X + d.n - 4, sigR, Z, part, blockIdx.x, blockIdx.y, sm);
}
__global__ __launch_bounds__(256) void k_meas_final(KDims d, KWeights w, const double* __restrict__ X, const double* __restrict__ sigR,
                                                    const double* __restrict__ Z, const double* __restrict__ part,
                                                    double* __restrict__ h, double* __restrict__ Si, int* __restrict__ vis,
                                                    double* __restrict__ PxyR)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < d.N) meas_final_one<false>(d, w, X, sigR, Z, part, h, Si, vis, PxyR, k);
}
// ------------------------------------------------------------------------------------------------
// k_gain: KalmanUpdate gains for all landmarks + state update (SLAM.cpp:2070-2080):
//   sii = Si^{-1} (OpenCV closed-form 2x2 inverse), U = Ki*Si^T = Pxy*sii, y = sii^T (z - h),
//   X += sum_k Ki (z - h) = sum_k U_k y_k.
// In : Ut rows 2k, 2k+1 hold S^T DZ for r < n-4 (k_pxy; scaled here by wi*gamma), PxyR rows n-4..n-1.
// Out: Ut rows become U^T (zero for unmatched / invisible landmarks); per-slice partial dX.
// grid = (np/64, GAIN_SLICES): block = 64 state rows x 4 sub-slices of one landmark slice; the k_syrk
// launch that follows adds the slice partials to X in fixed order (deterministic) in a few extra
// workgroups (srukf_gain_dx_job).  Block (0,0) also clears the gamma / xi accumulators of that k_syrk.
// ------------------------------------------------------------------------------------------------
#define GAIN_LM_MAX 64
__device__ __forceinline__ void gain_body(const KDims& d, const KWeights& w,
                                          double* __restrict__ Ut, const double* __restrict__ PxyR,
                                          const double* __restrict__ Si, const int* __restrict__ vis,
                                          const double* __restrict__ h, const double* __restrict__ z_seq,
                                          const double* z_cur, const int* __restrict__ m_seq, const int* m_cur,
                                          FrameScalars* __restrict__ fs, double* __restrict__ dxp /* [GAIN_SLICES][np] */, const RankArgs& ra,
                                          const double* __restrict__ Cm, double* __restrict__ S,
                                          const double* __restrict__ P1, int split_b0, const double* __restrict__ DZp, double sqeps,
                                          const double* __restrict__ sigR, const double* __restrict__ Z0, int fmode, const int bx, const int by)
{
    __shared__ double red[4][64];
    if (bx == 0 && by == 0 && threadIdx.x == 0) { fs->gmax_bits = 0ull; fs->ximax_bits = 0ull; }
    const int rl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int r = bx * 64 + rl;
    const int n = d.n, ld = d.np, mp = d.mp, N = d.N;
    // replay path: the motion step of this frame (workgroup 0 of k_project_motion) left the new last four columns of S in Cm —
    // the projection threads of its launch were reading the old ones.  Committed here, one row per thread, before the
    // k_syrk launch reads them: R12 rows (r < n-4), R22 rows (the robot block); the permuted copy of the rank-aware form too.
    if (Cm && by == 0 && sl == 0 && r < n) {
        const double4 v = *reinterpret_cast<const double4*>(Cm + (size_t)r * 4);
        *reinterpret_cast<double2*>(S + (size_t)r * ld + (n - 4)) = make_double2(v.x, v.y);
        *reinterpret_cast<double2*>(S + (size_t)r * ld + (n - 2)) = make_double2(v.z, v.w);
        if (ra.A) {
            const int arow = (r < n - 4) ? ra.iperm[r] : ra.r - 4 + (r - (n - 4));
            if (arow < ra.r) { double* o = ra.A + (size_t)arow * ld + (ra.r - 4); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
        }
    }
    const double* z = z_cur ? z_cur : (z_seq + (size_t)fs->frame * 2 * N);
    const int* mt = m_cur ? m_cur : (m_seq + (size_t)fs->frame * N);
    const double sc = w.wi * w.gamma;
    const int per = (N + GAIN_SLICES - 1) / GAIN_SLICES;
    const int k_beg = by * per, k_end = min(N, k_beg + per);
    const int rp = ra.Utp ? ra.iperm[r] : 0;                  // rank-aware replay: U^T also with permuted columns
    // per-landmark constants of this slice once per workgroup (they are a chain of small dependent loads: fetched per
    // thread and iteration they cost more than the streaming of Ut itself): Si^{-1}, Si^{-T}(z - h), "matched and visible"
    __shared__ double lk[GAIN_LM_MAX][6];
    __shared__ int lon[GAIN_LM_MAX];
    double dx = 0.0;
    for (int k0 = k_beg; k0 < k_end; k0 += GAIN_LM_MAX) {
        __syncthreads();
        const int cnt = min(GAIN_LM_MAX, k_end - k0);
        if ((int)threadIdx.x < cnt) {
            const int k = k0 + threadIdx.x;
            const bool on = (mt[k] != 0) && (vis[k] != 0);
            const GainLm g = srukf_gain_lm(Si[4 * k], Si[4 * k + 1], Si[4 * k + 2], Si[4 * k + 3], z[2 * k], z[2 * k + 1], h[2 * k], h[2 * k + 1], on ? 1 : 0);
            lk[threadIdx.x][0] = g.i00; lk[threadIdx.x][1] = g.i01; lk[threadIdx.x][2] = g.i10; lk[threadIdx.x][3] = g.i11;
            lk[threadIdx.x][4] = g.y0; lk[threadIdx.x][5] = g.y1;
            lon[threadIdx.x] = on ? 1 : 0;
        }
        __syncthreads();
        for (int q = sl; q < cnt; q += 4) {
            const int k = k0 + q;
            double u0 = 0.0, u1 = 0.0;
            if (lon[q] && r < n) {
                double p0, p1;
                if (r < n - 4 && P1) {
                    // "table" mode: the raw product arrives in permuted columns (k_pxy2): first K half in Utp, second in P1 where the
                    // range was cut; a structurally null row of S contributes sqrt(EPSILON) DZ[r] to its own column only
                    const bool split = rp >= split_b0, nullrow = rp >= ra.r && k == r / 6;
                    // (its DZ row is zero outside its own landmark r / 6 — exactly, see NullSkip — and may not even be written there)
                    p0 = srukf_gain_pxy(ra.Utp[(size_t)(2 * k) * ld + rp], split ? P1[(size_t)(2 * k) * ld + rp] : 0.0, split, nullrow ? DZp[(size_t)rp * mp + 2 * k] : 0.0, nullrow, sqeps, sc);
                    p1 = srukf_gain_pxy(ra.Utp[(size_t)(2 * k + 1) * ld + rp], split ? P1[(size_t)(2 * k + 1) * ld + rp] : 0.0, split, nullrow ? DZp[(size_t)rp * mp + 2 * k + 1] : 0.0, nullrow, sqeps, sc);
                } else if (r < n - 4) {
                    p0 = sc * Ut[(size_t)(2 * k) * ld + r];
                    p1 = sc * Ut[(size_t)(2 * k + 1) * ld + r];
                } else {
                    const int e = r - (n - 4);
                    p0 = PxyR[(size_t)e * mp + 2 * k];
                    p1 = PxyR[(size_t)e * mp + 2 * k + 1];
                    if (fmode) {
                        // "fused tail" mode: the statistics left the sums around the centre point's robot part r_0 (srukf_meas.h, meas_final_tail); the frame's
                        // motion reduction has run since: re-centre on the mean xr and on h
                        const double dxs = fs->Xr1[e] - sigR[e], rse = sigR[(size_t)d.L * 8 + e];
                        p0 = srukf_gain_recentre(p0, dxs, PxyR[(size_t)4 * mp + 2 * k], h[2 * k] - Z0[2 * k], rse);
                        p1 = srukf_gain_recentre(p1, dxs, PxyR[(size_t)4 * mp + 2 * k + 1], h[2 * k + 1] - Z0[2 * k + 1], rse);
                    }
                }
                const GainLm g = { lk[q][0], lk[q][1], lk[q][2], lk[q][3], lk[q][4], lk[q][5], 1 };
                double cq;
                srukf_gain_apply(g, p0, p1, u0, u1, cq);
                dx += cq;
            }
            if (!P1) {                                         // ("table" mode consumes U^T in permuted columns only)
                Ut[(size_t)(2 * k) * ld + r] = u0;
                Ut[(size_t)(2 * k + 1) * ld + r] = u1;
            }
            if (ra.Utp) { ra.Utp[(size_t)(2 * k) * ld + rp] = u0; ra.Utp[(size_t)(2 * k + 1) * ld + rp] = u1; }
        }
    }
    red[sl][rl] = dx;
    __syncthreads();
    if (sl == 0) dxp[(size_t)by * ld + r] = (red[0][rl] + red[1][rl]) + (red[2][rl] + red[3][rl]);
}
struct GainNextPose { double v[3]; double* odo; };
__global__ __launch_bounds__(256) void k_gain(KDims d, KWeights w,
                                              double* __restrict__ Ut, const double* __restrict__ PxyR,
                                              const double* __restrict__ Si, const int* __restrict__ vis,
                                              const double* __restrict__ h, const double* __restrict__ z_seq,
                                              const double* z_cur, const int* __restrict__ m_seq, const int* m_cur,
                                              FrameScalars* __restrict__ fs, double* __restrict__ dxp /* [GAIN_SLICES][np] */, const RankArgs ra,
                                              const double* __restrict__ Cm, double* __restrict__ S,
                                              const double* __restrict__ P1, int split_b0, const double* __restrict__ DZp, double sqeps,
                                              const double* __restrict__ sigR, const double* __restrict__ Z0, int fmode, GainNextPose np3)
{
    // step-wise API: the pose after this frame's, announced by the host after the frame's first launch had gone out (third pose of the two-frame sequence the frame scalars
    // point at; the state-update job of the NEXT launch prepares the next control from it).  Nothing in this launch reads the sequence.
    if (np3.odo && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { np3.odo[6] = np3.v[0]; np3.odo[7] = np3.v[1]; np3.odo[8] = np3.v[2]; fs->odo_seq = np3.odo; fs->seqF = 2; }
    gain_body(d, w, Ut, PxyR, Si, vis, h, z_seq, z_cur, m_seq, m_cur, fs, dxp, ra, Cm, S, P1, split_b0, DZp, sqeps, sigR, Z0, fmode, (int)blockIdx.x, (int)blockIdx.y);
}
// batched form (srukf_run_frames_batch): grid (np / 64, GAIN_SLICES B), filter f = blockIdx.y / GAIN_SLICES; staged inputs only, "fused tail" mode
__global__ __launch_bounds__(256) void k_gain_b(KDims d, KWeights w, const GainArgs* __restrict__ tab, int split_b0, double sqeps)
{
    const int f = (int)blockIdx.y / GAIN_SLICES, by = (int)blockIdx.y - f * GAIN_SLICES;
    const GainArgs a = tab[f];
    gain_body(d, w, a.Ut, a.PxyR, a.Si, a.vis, a.h, a.z_seq, nullptr, a.m_seq, nullptr, a.fs, a.dxp, a.ra, a.Cm, a.S, a.P1, split_b0, a.DZp, sqeps, a.sigR, a.Z0, 1, (int)blockIdx.x, by);
}
// k_gain_center: weight types with wc0 != wm0 (FLAG_4_WEIGHT2, SLAM.cpp:1077-1088: wc0 = wm0 + 1 - alpha^2 + beta).
// calculateOneFeatureCrossCovariance centres on the RUNNING state (s1 = sigma_i - m_X_k, 2030) which KalmanUpdate has
// already moved by dX_<k = sum of the earlier landmarks' Ki (z - h) (2079).  With sum_c wm_c (Z_c - h) = 0 the
// dependence on the running state collapses to the centre column:
//     Pxy_k = Pxy_k(frozen X) - (wc0 - wm0) dX_<k (Z_0k - h_k)^T
//  => U_k = Pxy_k Si^-1 = U_k(frozen) - (wc0 - wm0) dX_<k g_k^T,   g_k = Si^-T (Z_0k - h_k)
//     dX_<=k = dX_<k + U_k Si^-T (z_k - h_k)
// a recurrence over the matched landmarks in list order (2066) whose coefficients are per-landmark scalars, so every
// state row runs it on its own: one thread per row, Ut rows read and rewritten coalesced.  For wc0 == wm0 (types 0
// and 2) the term vanishes and the kernel is not launched.  Replaces the dX slice partials of k_gain.
__global__ __launch_bounds__(256) void k_gain_center(KDims d, KWeights w, double* __restrict__ Ut, const double* __restrict__ Z,
                                                     const double* __restrict__ Si, const int* __restrict__ vis,
                                                     const double* __restrict__ h, const double* __restrict__ z_seq,
                                                     const double* z_cur, const int* __restrict__ m_seq, const int* m_cur,
                                                     const FrameScalars* __restrict__ fs, double* __restrict__ dxp)
{
    __shared__ double lk[GAIN_LM_MAX][4];
    __shared__ int lon[GAIN_LM_MAX];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int n = d.n, ld = d.np, N = d.N;
    const double* z = z_cur ? z_cur : (z_seq + (size_t)fs->frame * 2 * N);
    const int* mt = m_cur ? m_cur : (m_seq + (size_t)fs->frame * N);
    const double cw = w.wc0 - w.wm0;
    double acc = 0.0;
    for (int k0 = 0; k0 < N; k0 += GAIN_LM_MAX) {
        __syncthreads();
        const int cnt = min(GAIN_LM_MAX, N - k0);
        if ((int)threadIdx.x < cnt) {
            const int k = k0 + threadIdx.x;
            const double s00 = Si[4 * k], s01 = Si[4 * k + 1], s10 = Si[4 * k + 2], s11 = Si[4 * k + 3];
            double det = s00 * s11 - s01 * s10;
            double i00 = 0, i01 = 0, i10 = 0, i11 = 0;
            if (det != 0.0) { det = 1.0 / det; i00 = s11 * det; i01 = -s01 * det; i10 = -s10 * det; i11 = s00 * det; }
            const double v0 = z[2 * k] - h[2 * k], v1 = z[2 * k + 1] - h[2 * k + 1];
            const double a0 = Z[2 * k] - h[2 * k], a1 = Z[2 * k + 1] - h[2 * k + 1];          // centre column Z_0 - h
            lk[threadIdx.x][0] = i00 * v0 + i10 * v1; lk[threadIdx.x][1] = i01 * v0 + i11 * v1;   // Si^-T (z - h)
            lk[threadIdx.x][2] = a0 * i00 + a1 * i10; lk[threadIdx.x][3] = a0 * i01 + a1 * i11;   // (Z_0 - h)^T Si^-1
            lon[threadIdx.x] = ((mt[k] != 0) && (vis[k] != 0)) ? 1 : 0;
        }
        __syncthreads();
        if (r < n) {
            for (int q = 0; q < cnt; q++) {
                if (!lon[q]) continue;
                const int k = k0 + q;
                const double u0 = Ut[(size_t)(2 * k) * ld + r] - cw * acc * lk[q][2];
                const double u1 = Ut[(size_t)(2 * k + 1) * ld + r] - cw * acc * lk[q][3];
                Ut[(size_t)(2 * k) * ld + r] = u0;
                Ut[(size_t)(2 * k + 1) * ld + r] = u1;
                acc += u0 * lk[q][0] + u1 * lk[q][1];
            }
        }
    }
    if (r < ld) {
        dxp[r] = (r < n) ? acc : 0.0;
        for (int u = 1; u < GAIN_SLICES; u++) dxp[(size_t)u * ld + r] = 0.0;
    }
}
// k_traj: per-frame record (x, y, z, theta, P00, P01, P10, P11) of the robot = RobotPath.txt
// columns (SLAM.cpp:3549-3556) with P = S^T S restricted to the robot x/y block (2404); also
// advances the staged-sequence frame counter.  One workgroup.
__global__ __launch_bounds__(256) void k_traj(KDims d, const double* __restrict__ X, const double* __restrict__ S,
                                              FrameScalars* __restrict__ fs, double* __restrict__ traj, int advance)
{
    __shared__ double red[16 * 3];
    const int n = d.n, ld = d.np;
    double v[3] = { 0, 0, 0 };
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double a = S[(size_t)k * ld + (n - 4)], b = S[(size_t)k * ld + (n - 3)];
        v[0] += a * a; v[1] += a * b; v[2] += b * b;
    }
    block_sum<3>(v, red);
    if (threadIdx.x == 0) {
        if (!traj) traj = fs->traj_base;
        if (traj) {
            double* t = traj + (size_t)8 * fs->frame;
            for (int e = 0; e < 4; e++) t[e] = X[n - 4 + e];
            t[4] = v[0]; t[5] = v[1]; t[6] = v[1]; t[7] = v[2];
        }
        if (advance) { fs->frame += 1; srukf_prepare_control(fs); }
    }
}

// k_block_cov: small diagonal blocks of P = S^T S for the accessors (robot 4x4: SLAM.cpp:3539-3556;
// landmark 6x6: 2748).  out[bs*bs], block starts at row/col `off`.  One workgroup.
// X (may be null): the bs state entries of the block are appended behind the bs x bs values (the robot view the step-wise API hands back with a frame's status)
__global__ __launch_bounds__(256) void k_block_cov(KDims d, const double* __restrict__ S, int off, int bs, double* __restrict__ out, const double* __restrict__ X)
{
    // one pass over the rows k <= off + bs - 1 for all bs (bs + 1) / 2 column pairs at once (bs <= 6: 21 sums), one block reduction — round 4 ran one reduction per pair,
    // ten dependent passes for the robot block that the step-wise API hands back with every frame's status
    __shared__ double red[16 * 21];
    const int ld = d.np;
    if (X && (int)threadIdx.x < bs) out[bs * bs + threadIdx.x] = X[off + threadIdx.x];
    double v[21];
#pragma unroll
    for (int q = 0; q < 21; q++) v[q] = 0.0;
    // eight rows per thread and trip, all their loads requested before the first product (a row per trip was one memory round trip per 256 rows: 12 us at n = 1204)
    for (int k0 = threadIdx.x; k0 < off + bs; k0 += 8 * 256) {  // S upper triangular: S[k][off + a] = 0 for k > off + a
        double s[8][6];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + 256 * u;
#pragma unroll
            for (int e = 0; e < 6; e++) s[u][e] = (e < bs && k < off + bs && off + e >= k) ? S[(size_t)k * ld + off + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) v[q++] += s[u][a] * s[u][b];
        }
    }
    block_sum<21>(v, red);
    if (threadIdx.x != 0) return;
    int q = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) { if (a < bs && b < bs) { out[a * bs + b] = v[q]; out[b * bs + a] = v[q]; } q++; }
}

// k_landmarks_cartesian: getFeatureCartesianInformation (SLAM.cpp:2721-2751) for ALL landmarks in one launch — the
// per-paint loop of OpenGlDisplay.cpp:449-583 reads xyz and the 3x3 Cartesian covariance of every landmark, which the
// reference takes from m_P_k = S^T S (a 2 n^3 gemm per frame, 2404).  One workgroup per landmark: the 6x6 block of P
// (21 column dot products over the rows k <= 6 id + 5), then cov = J P66 J^T with J = [I3 | d(xyz)/d(theta, phi, rho)].
__global__ __launch_bounds__(256) void k_landmarks_cartesian(KDims d, const double* __restrict__ X, const double* __restrict__ S,
                                                             double* __restrict__ xyz, double* __restrict__ cov)
{
    __shared__ double red[16 * 21];
    const int id = blockIdx.x, off = 6 * id, ld = d.np;
    double v[21];
#pragma unroll
    for (int q = 0; q < 21; q++) v[q] = 0.0;
    // S upper triangular: rows below the block do not contribute.  Four rows per thread and trip, their loads requested before the first product (a row per trip is one
    // memory round trip per 256 rows; the order of a thread's sums is unchanged)
    // (cov == null — srukf_associate, whose warp needs the points only: no walk over S, the same xyz code)
    if (cov)
    for (int k0 = threadIdx.x; k0 < off + 6; k0 += 4 * 256) {
        double s[4][6];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = k0 + 256 * u;
#pragma unroll
            for (int e = 0; e < 6; e++) s[u][e] = (k < off + 6 && off + e >= k) ? S[(size_t)k * ld + off + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) v[q++] += s[u][a] * s[u][b];
        }
    }
    if (cov) block_sum<21>(v, red);
    if (threadIdx.x != 0) return;
    double P[6][6];
    { int q = 0; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { P[a][b] = v[q]; P[b][a] = v[q]; q++; } }
    const double xi = X[off], yi = X[off + 1], zi = X[off + 2], th = X[off + 3], ph = X[off + 4], rho = X[off + 5];
    double sth, cth, sph, cph;
    sincos(th, &sth, &cth); sincos(ph, &sph, &cph);
    xyz[3 * id + 0] = xi + cph * sth / rho;                                                      // 2738-2740
    xyz[3 * id + 1] = yi - sph / rho;
    xyz[3 * id + 2] = zi + cph * cth / rho;
    if (!cov) return;
    double J[3][6] = { { 1, 0, 0,  cph * cth / rho, -sph * sth / rho, -cph * sth / (rho * rho) },   // 2742-2747
                       { 0, 1, 0,  0.0,             -cph / rho,        sph / (rho * rho) },
                       { 0, 0, 1, -cph * sth / rho, -sph * cth / rho, -cph * cth / (rho * rho) } };
    double JP[3][6];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 6; b++) { double t = 0.0; for (int k = 0; k < 6; k++) t += J[a][k] * P[k][b]; JP[a][b] = t; }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double t = 0.0; for (int k = 0; k < 6; k++) t += JP[a][k] * J[b][k]; cov[9 * id + 3 * a + b] = t; }   // 2749
}

// ---- host-callable launchers -------------------------------------------------------------------
extern "C" {
void srukf_launch_landmarks_cartesian(hipStream_t st, KDims d, const double* X, const double* S, double* xyz, double* cov)
{
    hipLaunchKernelGGL(k_landmarks_cartesian, dim3(d.N), dim3(256), 0, st, d, X, S, xyz, cov);
}
void srukf_launch_motion(hipStream_t st, KDims d, KWeights w, srukf_params p, double* X, double* S, double* sigR, double* Cmat,
                         FrameScalars* fs, const double* odo_seq, const double* odo_pair, RankArgs ra)
{
    hipLaunchKernelGGL(k_motion, dim3(1), dim3(512), 0, st, d, w, p, X, S, sigR, Cmat, fs, odo_seq, odo_pair, ra);
}
void srukf_launch_project_motion(hipStream_t st, KDims d, KWeights w, srukf_params p, double* X, double* S, double* sigR, double* Cm,
                                 double* Z, double* DZ, FrameScalars* fs, RankArgs ra)
{
    dim3 grid(1 + ((d.Na + 1) * d.N + 511) / 512);
    hipLaunchKernelGGL(k_project_motion, grid, dim3(512), 0, st, d, w, p, X, S, sigR, Cm, Z, DZ, fs, ra);
}
void srukf_launch_project_table(hipStream_t st, KDims d, KWeights w, srukf_params p, double* X, double* S, double* sigR, double* Cm,
                                double* Z, double* DZ, FrameScalars* fs, RankArgs ra, NullSkip ns)
{
    const int items = ns.dirs ? (1 + ns.nfull) * d.N + ns.nnull : (d.Na + 1) * d.N;
    dim3 grid(1 + (items + 255) / 256);
    hipLaunchKernelGGL(k_project_table, grid, dim3(256), 0, st, d, w, p, X, S, sigR, Cm, Z, DZ, fs, ra, ns);
}
void srukf_launch_sigr_rows(hipStream_t st, KDims d, KWeights w, const double* X, const double* S, double* sigR, const FrameScalars* fs, const int* iperm, int rkeep)
{
    hipLaunchKernelGGL(k_sigr_rows, dim3((d.Na + 255) / 256), dim3(256), 0, st, d, w, X, S, sigR, fs, iperm, rkeep);
}
void srukf_launch_project(hipStream_t st, KDims d, KWeights w, srukf_params p, const double* X, const double* S, const double* sigR,
                          double* Z, double* DZ, const FrameScalars* fs)
{
    dim3 grid(((d.Na + 1) * d.N + 255) / 256);
    hipLaunchKernelGGL(k_project, grid, dim3(256), 0, st, d, w, p, X, S, sigR, Z, DZ, fs);
}
void srukf_launch_meas_stats(hipStream_t st, KDims d, KWeights w, const double* X, const double* sigR, const double* Z,
                             double* part, double* h, double* Si, int* vis, double* PxyR)
{
    hipLaunchKernelGGL(k_meas_partial, dim3((d.N + 31) / 32, MEAS_SLICES), dim3(256), 0, st, d, w, X, sigR, Z, part);
    hipLaunchKernelGGL(k_meas_final, dim3((d.N + 255) / 256), dim3(256), 0, st, d, w, X, sigR, Z, part, h, Si, vis, PxyR);
}
int srukf_meas_part_doubles(int mp) { return MEAS_SLICES * MEAS_NS * (mp / 2); }
void srukf_launch_gain_b(hipStream_t st, KDims d, KWeights w, const void* tab, int B, int split_b0, double sqeps)
{
    hipLaunchKernelGGL(k_gain_b, dim3(d.np / 64, GAIN_SLICES * B), dim3(256), 0, st, d, w, (const GainArgs*)tab, split_b0, sqeps);
}
void srukf_launch_gain(hipStream_t st, KDims d, KWeights w, double* Ut, const double* PxyR, const double* Si, const int* vis,
                       const double* h, const double* z_seq, const double* z_cur, const int* m_seq, const int* m_cur,
                       FrameScalars* fs, double* dxp, double* X, const double* Z, RankArgs ra, const double* Cm, double* S,
                       const double* P1, int split_b0, const double* DZp, double sqeps, const double* sigR, int fmode, const double* next_pose, double* odo_dev)
{
    GainNextPose np3 = { { 0, 0, 0 }, nullptr };
    if (next_pose && odo_dev) { np3.v[0] = next_pose[0]; np3.v[1] = next_pose[1]; np3.v[2] = next_pose[2]; np3.odo = odo_dev; }
    hipLaunchKernelGGL(k_gain, dim3(d.np / 64, GAIN_SLICES), dim3(256), 0, st, d, w, Ut, PxyR, Si, vis, h, z_seq, z_cur, m_seq, m_cur, fs, dxp, ra, Cm, S,
                       P1, split_b0, DZp, sqeps, sigR, Z, fmode, np3);
    if (w.wc0 != w.wm0)
        hipLaunchKernelGGL(k_gain_center, dim3(d.np / 256 + 1), dim3(256), 0, st, d, w, Ut, Z, Si, vis, h, z_seq, z_cur, m_seq, m_cur, fs, dxp);
}
int srukf_gain_part_doubles(int np) { return GAIN_SLICES * np; }
void srukf_launch_traj(hipStream_t st, KDims d, const double* X, const double* S, FrameScalars* fs, double* traj, int advance)
{
    hipLaunchKernelGGL(k_traj, dim3(1), dim3(256), 0, st, d, X, S, fs, traj, advance);
}
void srukf_launch_block_cov(hipStream_t st, KDims d, const double* S, int off, int bs, double* out, const double* X)
{
    hipLaunchKernelGGL(k_block_cov, dim3(1), dim3(256), 0, st, d, S, off, bs, out, X);
}
}  // extern "C"

// stand-alone projection kernel for the parity tests (srukf_project_host)
__global__ void k_project_points(srukf_params p, int count, const double* feat6, const double* pos3, const double* psi,
                                 const double* err2, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double f[6];
    for (int e = 0; e < 6; e++) f[e] = feat6[6 * i + e];
    double sn, cs;
    sincos(psi[i], &sn, &cs);
    double ox, oy;
    srukf_project(p, p.cam_f / p.cam_dx, p.cam_f / p.cam_dy, f, pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2], cs, sn,
                  err2[2 * i], err2[2 * i + 1], ox, oy);
    out[2 * i] = ox; out[2 * i + 1] = oy;
}
extern "C" void srukf_launch_project_points(hipStream_t st, srukf_params p, int count, const double* feat6, const double* pos3,
                                            const double* psi, const double* err2, double* out)
{
    hipLaunchKernelGGL(k_project_points, dim3((count + 255) / 256), dim3(256), 0, st, p, count, feat6, pos3, psi, err2, out);
}
