// srukf_device.h — shared definitions for the gfx950 SRUKF kernels.
//
// Data layout in HBM (one filter; all fp64, row-major, leading dimensions padded):
//   X    [np]            state, n = 6N+4 live entries            (m_X_k, SLAM.h:271)
//   S    [np][np]        upper-triangular sqrt covariance, P = S^T S; rows/cols >= n and the
//                        strictly lower triangle are kept zero   (m_S_k, SLAM.h:272)
//   sigR [L][8]          robot part of every sigma point after the motion model:
//                        (x, y, z, theta, cos theta, sin theta, -, -)   L = 2*Na+1, Na = n+5
//   Z    [L][mp]         projected pixels, row = sigma point, col = 2k / 2k+1 of landmark k
//                        (m_sigma_allPixel, SLAM.h:285, stored transposed so that the
//                        contraction over sigma points is K-major for MFMA)
//   DZ   [np][mp]        DZ[i] = Z[1+i] - Z[1+Na+i]  for i < n (rows >= n zero)
//   Ut   [mp][np]        U^T: row c = measurement column, col r = state row
//   G    [np][np]        S^T S - U U^T (upper), factorised in place by the GMW kernels
// The full Na x L sigma matrix (m_sigma, SLAM.h:284; 23 MB at N = 200) is never materialised:
// sigma_c[0:n-4] = X +- gamma * S.row(i) is read straight from S.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/srukf.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define SRUKF_NB 32            // GMW panel height
#define SRUKF_PAD 64           // np, mp are multiples of this

struct KDims {
    int N, n, Na, L;           // landmarks, state dim, augmented dim, sigma count
    int np, mp;                // padded n, padded 2N
};

struct KWeights { double wm0, wc0, wi, wi_sr, gamma; };

#define SRUKF_STAT_GROUPS 64   // landmark groups of 32 the riding statistics jobs count: N <= 2048 (srukf_create refuses more)
// device-resident per-frame scalars
struct FrameScalars {
    double Ut[3];              // rot1, trans, rot2                    SLAM.cpp:1452-1454
    double Mt[3];              // control "sqrt" noise diag            SLAM.cpp:1456-1458
    double Xr0[4];             // robot mean before the motion step
    unsigned long long gmax_bits;   // max diag(G)      (as ordered bits of a non-negative double)
    unsigned long long ximax_bits;  // max(0, max offdiag(G))
    int clamp_rows;            // rows where the GMW theta clamp would have been active
    int clamp_first;           // first such row
    int frame;                 // frame counter for staged sequences
    int stat_count;            // landmark groups whose final pass has reached the host mirror (MeasArgs::hmirror; step-wise API only; the last one clears it)
    int stat_cnt[SRUKF_STAT_GROUPS];          // measurement-statistics jobs finished per landmark group of 32 in the current contraction launch: the last one
                               // of a group runs its final pass
    double* traj_base;         // device trajectory buffer of the current replay (row = absolute frame), or null
    int gmw_aborts;            // persistent GMW launches abandoned on an expired wait (their frames are flagged like clamp rows)
    int clamp_frame;           // staged replay: index of the FIRST frame whose refactorisation was flagged (0x7fffffff: none).  Set by
                               // the k_motion of the following frame (or by the host at the end of a run): frames before it are valid
    // ---- replay path (k_project_motion): the motion step rides on the projection launch ----
    double Xr1[4];             // robot mean after the motion step; the dX job of the k_syrk launch commits it (X_robot = Xr1 + dX)
    double ctl[8];             // control of the frame `frame` (rot1, trans, rot2, cos rot2, sin rot2, Mt[0..2]; SLAM.cpp:1444-1458), prepared by
                               // whoever sets or advances `frame` (srukf_prepare_control), so that no projection thread waits on the odometry
    const double* odo_seq;     // staged odometry (3 doubles per pose) and its frame count, a1..a4: what srukf_prepare_control needs
    double a[4];
    int seqF;
    int const_rows_ok, const_rows_pending;   // "table" mode: the structurally null rows of S already hold sqrt(EPSILON) e_k (k_rank_expand skips them); pending: as of the
                               // launch that is running — promoted by the next frame's first launch, so that no workgroup of the writing launch sees it
    int gate_timeouts;         // k_gmw_gate gave up waiting for a slot (SRUKF_GPU_SHARED) and went ahead: reported, never silent
    int ctl_next_valid;        // "table" mode: k_gain prepared fs->ctl for frame + 1 (0: the staged sequence ends with this frame)
    int frozen;                // staged replay: a frame was flagged -> k_motion and the persistent factorisation of the later frames of the run
                               // return at once (three quarters of a frame's time; the other kernels would pay a memory round trip per launch
                               // for the test); cleared with the clamp counters (k_set_frame) and by the step-wise API
    int export_cnt;            // step-wise API: workgroups of the frame's last launch (k_rank_expand<2>) that are through; the last one exports status + robot view (StepExport)
};

// control from two odometry poses (SLAM.cpp:1444-1458): Ut = (rot1, trans, rot2), Mt = control-noise sigmas
__device__ __forceinline__ void srukf_motion_control_a(const double (&a)[4], const double* o, double (&ut)[3], double (&mt)[3])
{
    const double dx = o[3] - o[0], dy = o[4] - o[1];
    const double rot1 = atan2(dy, dx) - o[2];
    const double trans = sqrt(dy * dy + dx * dx);
    const double rot2 = o[5] - o[2] - rot1;
    ut[0] = rot1; ut[1] = trans; ut[2] = rot2;
    mt[0] = a[0] * rot1 * rot1 + a[1] * trans * trans;
    mt[1] = a[2] * trans * trans + a[3] * rot1 * rot1 + a[3] * rot2 * rot2;
    mt[2] = a[0] * rot2 * rot2 + a[1] * trans * trans;
}
// control of the staged frame f as fs->ctl holds it (rot1, trans, rot2, cos rot2, sin rot2, Mt[0..2]); false: no such frame
__device__ __forceinline__ bool srukf_control_values(const FrameScalars* fs, int f, double (&ctl)[8])
{
    const double* os = fs->odo_seq;
    if (!os || f < 0 || f >= fs->seqF) return false;
    double ut[3], mt[3];
    const double a[4] = { fs->a[0], fs->a[1], fs->a[2], fs->a[3] };
    srukf_motion_control_a(a, os + 3 * f, ut, mt);
    ctl[0] = ut[0]; ctl[1] = ut[1]; ctl[2] = ut[2];
    ctl[3] = cos(ut[2]); ctl[4] = sin(ut[2]);
    ctl[5] = mt[0]; ctl[6] = mt[1]; ctl[7] = mt[2];
    return true;
}
// fs->ctl for the staged frame f (one thread); false: no such frame
__device__ __forceinline__ bool srukf_prepare_control(FrameScalars* fs, int f)
{
    double ctl[8];
    if (!srukf_control_values(fs, f, ctl)) return false;
#pragma unroll
    for (int q = 0; q < 8; q++) fs->ctl[q] = ctl[q];
    return true;
}
// ... for the frame the counter stands at (call after `frame` was set or advanced)
__device__ __forceinline__ void srukf_prepare_control(FrameScalars* fs) { srukf_prepare_control(fs, fs->frame); }

// sum over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1: VALU only, no LDS traffic); every lane of the row gets the total
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_mov_f64<0x128>(v); v += dpp_mov_f64<0x124>(v); v += dpp_mov_f64<0x122>(v); v += dpp_mov_f64<0x121>(v);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// Block-wide sum of NV values per thread (blockDim.x multiple of 64, <= 1024).  red: LDS scratch
// of at least 16*NV doubles.  Every thread gets the totals.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; q++) v[q] = wave_sum(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; q++) red[wid * NV + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; q++) {
        double s = 0.0;
        for (int w = 0; w < nw; w++) s += red[w * NV + q];
        v[q] = s;
    }
}

// ---- camera model (device restatement of SLAM.cpp:1634-1674, 3177-3213, 3250-3347) ----------
// feat = (xi yi zi theta phi rho), robot = (x y z), cs/sn = cos/sin of robot theta,
// err = pixel-noise sigma rows.  Returns uvd = (x, y) = Z[2k], Z[2k+1].
// Every fused multiply-add of this function is written out and the compiler's own contraction is off: which products it fuses
// depends on the code around an inlined call, and the same sigma point projected by two kernels (k_project_table; the frame tail
// k_rank_expand<2>) must give the same bits.
__device__ __forceinline__ void srukf_project(const srukf_params& p, double f1, double f2,
                                              const double feat[6], double px, double py, double pz,
                                              double cs, double sn, double e0, double e1,
                                              double& ox, double& oy)
{
#pragma clang fp contract(off)
    const double xi = feat[0], yi = feat[1], zi = feat[2], th = feat[3], ph = feat[4], rho = feat[5];
    double sth, cth, sph, cph;
    sincos(th, &sth, &cth);
    sincos(ph, &sph, &cph);
    const double ir = 1.0 / rho;
    // coordinatesState2World, SLAM.cpp:3272-3275
    const double ic = ir * cph;
    const double hx = fma(ic, sth, xi) - px;
    const double hy = fma(-ir, sph, yi) - py;
    const double hz = fma(ic, cth, zi) - pz;
    // Rcw = Rwc.inv() (SLAM.cpp:1642-1643): closed-form cofactor inverse, det = c^2 + s^2
    const double det = fma(cs, cs, sn * sn);
    const double id = 1.0 / det;
    const double rx = fma(cs * id, hx, (sn * id) * hy);       // coordinatesWorld2Camera, 3292
    const double ry = fma(-sn * id, hx, (cs * id) * hy);
    const double rz = (det * id) * hz;
    double ux, uy;
    if (rz == 0.0) { ux = 0.0; uy = 0.0; }                    // coordinatesCamera2Image, 3331-3335
    else {
        uy = p.cam_cx + f1 * rx / rz + e0;                    // 3338 (x/y swap)
        ux = p.cam_cy + f2 * ry / rz + e1;                    // 3339
        if (ux < 10.0 || ux > p.image_w - 10.0 || uy < 10.0 || uy > p.image_h - 10.0) { ux = 0.0; uy = 0.0; }   // 3341-3345
    }
    // distortOnePointRW, 3177-3213
    const double k1 = p.cam_k1, k2 = p.cam_k2;
    const double xu = (ux - p.cam_cx) * p.cam_dx;
    const double yu = (uy - p.cam_cy) * p.cam_dy;
    const double ru = sqrt(fma(xu, xu, yu * yu));
    const double ru2 = ru * ru;
    double rd = ru / fma(k2 * ru2, ru2, fma(k1, ru2, 1.0));
    // 100 Newton iterations (3188-3193); leaving the loop once rd is a fixed point is bit-exact because every later iteration
    // reproduces the same rd.  In floating point the iteration often does not reach a fixed point but ends in a 2-cycle between
    // two neighbouring doubles (then all 100 iterations ran, on every wave that had one such lane: that was 40 % of k_project's
    // time): once the new iterate equals the one before the current, the sequence alternates for good, and the value the
    // reference holds after its last iteration follows from the parity of the iterations left.
    double rprev = __builtin_nan("");
    const double k1_3 = 3.0 * k1, k2_5 = 5.0 * k2;
    for (int it = 0; it < p.newton_iters; it++) {
        const double rd2 = rd * rd;
        const double f  = fma(k2 * rd2 * rd2, rd, fma(k1 * rd2, rd, rd)) - ru;
        const double ff = fma(k2_5 * rd2, rd2, fma(k1_3, rd2, 1.0));
        const double rn = rd - f / ff;
        if (rn == rd) break;
        if (rn == rprev) { if ((p.newton_iters - it) & 1) rd = rn; break; }
        rprev = rd;
        rd = rn;
    }
    double d = fma(k2 * rd * rd * rd, rd, fma(k1 * rd, rd, 1.0));
    if (d == 0.0) d = p.epsilon;
    const double vx = p.cam_cx + (xu / d) / p.cam_dx;
    const double vy = p.cam_cy + (yu / d) / p.cam_dy;
    const bool vis = (vx >= 0.0) && (vx <= p.image_w) && (vy >= 0.0) && (vy <= p.image_h);
    ox = vis ? vx : 0.0;
    oy = vis ? vy : 0.0;
}

// generateSigmaPoints' addWeighted (SLAM.cpp:1159-1160: src1 * 1 + src2 * (+-gamma) + 0) on a landmark's six entries and the two pixel-noise rows
__device__ __forceinline__ void srukf_sigma_feat(const double (&base)[6], const double (&dev)[6], double e0, double e1, double gq,
                                                 double (&feat)[6], double& q0, double& q1)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 6; e++) feat[e] = fma(dev[e], gq, base[e] * 1) + 0;
    q0 = fma(e0, gq, 0.0 * 1) + 0; q1 = fma(e1, gq, 0.0 * 1) + 0;
}
// One sigma point of a direction (generateSigmaPoints' addWeighted, SLAM.cpp:1159-1160, on the landmark's six entries and the two
// pixel-noise rows), projected; r = its robot part (x, y, z, theta, cos theta, sin theta).  gq = +gamma / -gamma.
__device__ __forceinline__ void srukf_project_sigma(const srukf_params& p, double f1, double f2, const double (&base)[6], const double (&dev)[6],
                                                    double e0, double e1, double gq, const double* r, double& ox, double& oy)
{
    double feat[6], q0, q1;
    srukf_sigma_feat(base, dev, e0, e1, gq, feat, q0, q1);
    srukf_project(p, f1, f2, feat, r[0], r[1], r[2], r[4], r[5], q0, q1, ox, oy);
}

// ---- state update X += sum of the k_gain slice partials (fixed order): 256 state rows per workgroup ----
#define GAIN_SLICES 32
// xr1 (replay path, may be null): the robot mean after the motion step, which k_project_motion left beside X because the
// projection threads of its launch were still reading the mean before it
// f32: fp32 storage, "fused tail" mode: the new state is rounded to float here (what k_quantize does after the refactorisation in the other modes)
// lmN > 0: dxp holds per-landmark shares dxk[k][row] (the gain fold of k_pxy2) instead of slice partials: summed here in exactly the order k_gain and the loop above take —
// per slice of ceil(N / 32) landmarks four interleaved sub-slices, (s0 + s1) + (s2 + s3), then the slices in order — so that both forms give the same bits
__device__ __forceinline__ void srukf_gain_dx_job(int n, int np, const double* __restrict__ dxp, double* __restrict__ X, int job, const double* xr1 = nullptr, int f32 = 0, int lmN = 0)
{
    const int r = job * 256 + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0;
    if (lmN > 0) {
        const int per = (lmN + GAIN_SLICES - 1) / GAIN_SLICES;
        for (int u = 0; u < GAIN_SLICES; u++) {
            const int kb = u * per, cnt = min(lmN, kb + per) - kb;
            double s[4] = { 0.0, 0.0, 0.0, 0.0 };
            for (int q = 0; q < cnt; q++) s[q & 3] += dxp[(size_t)(kb + q) * np + r];
            acc += (s[0] + s[1]) + (s[2] + s[3]);
        }
    } else
#pragma unroll
    for (int u = 0; u < GAIN_SLICES; u++) acc += dxp[(size_t)u * np + r];
    const double x = (xr1 && r >= n - 4) ? xr1[r - (n - 4)] : X[r];
    double v = x + acc;
    if (f32) v = (double)(float)v;
    X[r] = v;
}

// ---- the arithmetic of KalmanUpdate's gains (SLAM.cpp:2070-2080), one function per formula ----------------------------------------------------------------------
// k_gain forms U = Pxy Si^-1 and the state update; since round 6 the staged replay forms them in the tile epilogue of k_pxy2 instead (GainFold, srukf_factor.hip).  Both
// must give the same bits (the step-wise API, which keeps k_gain, equals the staged replay bit for bit): every multiply-add is written out, the compiler's own
// contraction is off, and both kernels call these.
struct GainLm { double i00, i01, i10, i11, y0, y1; int on; };      // Si^-1 (OpenCV's closed-form 2 x 2 inverse), y = Si^-T (z - h), matched && visible
__device__ __forceinline__ GainLm srukf_gain_lm(double s00, double s01, double s10, double s11, double z0, double z1, double h0, double h1, int on)
{
#pragma clang fp contract(off)
    GainLm g;
    double det = fma(s00, s11, -(s01 * s10));
    g.i00 = 0.0; g.i01 = 0.0; g.i10 = 0.0; g.i11 = 0.0;
    if (det != 0.0) { det = 1.0 / det; g.i00 = s11 * det; g.i01 = -s01 * det; g.i10 = -s10 * det; g.i11 = s00 * det; }
    const double v0 = z0 - h0, v1 = z1 - h1;
    g.y0 = fma(g.i10, v1, g.i00 * v0); g.y1 = fma(g.i11, v1, g.i01 * v0);
    g.on = on;
    return g;
}
// raw product of one measurement row and one state row -> Pxy entry: the two K halves of k_pxy2, the sqrt(EPSILON) DZ term of a structurally null row, the wi gamma scale
__device__ __forceinline__ double srukf_gain_pxy(double q0, double q1, bool split, double dz, bool nullrow, double sqeps, double sc)
{
#pragma clang fp contract(off)
    double q = q0;
    if (split) q = q + q1;
    if (nullrow) q = fma(sqeps, dz, q);
    return sc * q;
}
// robot rows in "fused tail" mode: the statistics left the sums around the centre point's robot part; re-centred on the mean and on h
__device__ __forceinline__ double srukf_gain_recentre(double p, double dxs, double s4, double hz, double rse)
{
#pragma clang fp contract(off)
    return fma(-hz, rse, fma(-dxs, s4, p));
}
// U = Pxy Si^-1 for one state row (the landmark's two measurement rows), and the row's share of K (z - h)
__device__ __forceinline__ void srukf_gain_apply(const GainLm& g, double p0, double p1, double& u0, double& u1, double& c)
{
#pragma clang fp contract(off)
    u0 = fma(p1, g.i10, p0 * g.i00);
    u1 = fma(p1, g.i11, p0 * g.i01);
    c = fma(u1, g.y1, u0 * g.y0);
}

// "Gain fold" (round 6; the staged replay in "fused tail" mode): k_gain's work rides on k_pxy2.  A 64 x 64 tile of the cross covariances holds 32 whole landmarks x 64
// state rows, and what U = Pxy Si^-1 needs besides the tile — Si, h, visible of those 32 landmarks — is exactly one landmark group of the statistics jobs of the same
// launch: the workgroup that finishes a tile (the second of the two when its K range is cut) waits for that group's final pass and writes U^T for the tile itself; the
// state update's per-landmark shares go to dxk[k][state row] and the job that applies the update sums them in k_gain's order (srukf_gain_dx_job); the motion workgroup
// commits the motion step's columns once the tiles that read them are through.  One launch (7.7 us + a gap of the 182-us frame at N = 200) less.
// sync (uints, zero between frames: the frame tail clears them): see FOLD_* below.
struct GainFold {
    unsigned int* sync;                                        // null: no fold (k_gain follows)
    double* Utp; double* dxk;
    const double* z_seq; const int* m_seq;
    const double* DZp; const int* perm; const int* iperm; int r; int split_b0; double sqeps; double sc;
    double* S; double* A;                                      // the motion step's columns are committed here (what k_gain did with Cm)
    int nmt, nbt, bt_r0, bt_r1, robot_tiles;                   // tiles per column block, column blocks, the block(s) of the robot columns and how many tile workgroups read them
};
// Layout: the pair counters first (four words apart), then the flags that many workgroups POLL, each 4 KB from the next: ~130 waiting tile workgroups polling seven
// words in neighbouring cache lines queued at one memory channel, and the statistics' final passes — chains of dependent device-scope round trips — queued behind
// them (their flags moved from 17 to 23 us: scripts/fold_stamps.py)
#define FOLD_FLAG_STRIDE 1024
#define FOLD_PAIR(mt, bt, nbt) (4 * ((mt) * (nbt) + (bt)))
__host__ __device__ inline int srukf_fold_flag_base(int nmt, int nbt) { return (4 * nmt * nbt + FOLD_FLAG_STRIDE - 1) / FOLD_FLAG_STRIDE * FOLD_FLAG_STRIDE; }
#define FOLD_MOTION(nmt, nbt) (srukf_fold_flag_base(nmt, nbt))                                  // 1: the motion reduction's results are visible; 2: the run is frozen behind a flagged frame
#define FOLD_ROBOT(nmt, nbt) (srukf_fold_flag_base(nmt, nbt) + FOLD_FLAG_STRIDE)                // tile workgroups of the robot columns' block(s) that are through
#define FOLD_STAT(g, nmt, nbt) (srukf_fold_flag_base(nmt, nbt) + FOLD_FLAG_STRIDE * (2 + (g)))  // 1: landmark group g's h / Si / visible / PxyR are visible
__host__ __device__ inline int srukf_fold_words(int nmt, int nbt) { return srukf_fold_flag_base(nmt, nbt) + FOLD_FLAG_STRIDE * (2 + SRUKF_STAT_GROUPS); }
// diagnostic builds (-DSRUKF_FOLD_DBG): 8192 time stamps (s_memrealtime, 10 ns ticks) behind the sync words, which the frame tail does not clear (srukf_debug_copy "fold_dbg")
#define FOLD_DBG_STAMPS 8192
#ifdef SRUKF_FOLD_DBG
#define FOLD_TS(gf, slot) do { if ((gf).sync && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    ((unsigned long long*)((gf).sync + ((srukf_fold_words((gf).nmt, (gf).nbt) + 1) & ~1)))[slot] = t_; } } while (0)
#else
#define FOLD_TS(gf, slot) do { } while (0)
#endif

// measurement-statistics work attached to a k_pxy launch (replay path): Z == null -> none
// "Table" mode of the rank-aware replay: a structurally null row of S (sqrt(EPSILON) e_i, nothing in the robot columns) moves ONE landmark
// only, by sqrt(EPSILON) gamma in one anchor coordinate — for every other landmark both its sigma points project exactly where the centre
// point does (same function, same inputs), so their Z rows equal Z_0, their DZ entries are 0 and their terms in every statistic are
// exact zeros.  They are neither projected nor read: dirs = the directions that are projected for all landmarks (kept rows, the noise
// rows, and rows 0 / 1, whose Z rows the Si factor names explicitly), nulls = the others (one item each: their own landmark i / 6),
// rows = the rows of Z the statistics walk (0, then 1 + i and 1 + Na + i of every direction in dirs).  All null: every direction is full.
struct NullSkip { const int* dirs; const int* nulls; const int* rows; int nfull, nnull, nrows; const int* iperm; int r; };

struct MeasArgs {
    const double* X; const double* xrob; const double* sigR; const double* Z; double* part;   // xrob: robot mean after the motion step (4 doubles)
    double* h; double* Si; int* vis; double* PxyR;
    FrameScalars* fs; int gx;                                  // gx = (N + 31) / 32 landmark groups
    NullSkip ns;                                               // ns.rows != null: the statistics walk that row list (+ every landmark's own null rows)
    int preamble;                                              // "tail" mode: k_pxy2 is the frame's first launch -> its first thread runs srukf_frame_preamble
    int fmode;                                                 // "fused tail" mode: workgroup 0 of k_pxy2 is the frame's motion reduction (the table is complete, nothing else of the frame
                                                               // has run); the statistics are centred on the CENTRE point's robot part (xrob = sigR row 0) and their final pass leaves the raw
                                                               // sums (PxyR rows 0..3) and wi * sum(Z_c - Z_0) (row 4): k_gain, which runs after the reduction, re-centres them on the mean
    double* Cm;                                                // fmode: where the motion reduction leaves R12 / R22 (k_gain commits them)
    // step-wise API (null / 0 in the replay): h | Si | visible are ONE device allocation starting at h; hmirror is a pinned HOST buffer of the same layout that every landmark
    // group's final pass fills as well, and the last group stores hseq to *hflag behind it (system scope): the host has the statistics while the launch still forms its tiles
    char* hmirror; unsigned long long* hflag; unsigned long long hseq;
    unsigned long long* hstamp;                                // (may be null) two pinned words: s_memrealtime when the launch's first workgroup starts / when the flag is raised
};

// Step-wise API (dst null in the replay): the frame's LAST launch hands the frame's status and robot view to the host itself.  Its frame-tail workgroup forms the 4 x 4 robot
// block of P = S^T S from the factor rows it walks anyway and leaves it, with the pose, in `view` (20 doubles); every workgroup counts itself (cnt, then fs->export_cnt) when its
// updates of *fs are through, and the last one copies *fs (nfs 8-byte words) and the view to the pinned host buffer dst and stores seq to *flag behind them (system scope).
// cnt: 64 counters 256 B apart (zero between launches) in front of fs->export_cnt: ~1 250 relaxed device-scope increments of ONE word cost the launch 29 us (they
// serialise at the memory side); 64 words take ~20 each, the last arrival of each word then counts in fs->export_cnt.
// set: the exporting workgroup — the last thing that runs in the frame — also starts the NEXT frame's scalars (what k_set_step does: the host has announced that frame's
// odometry, poses = prev | cur): no launch for it between this frame's tail and the next frame's first launch.
struct StepExport { unsigned long long* dst; int nfs; double* view; unsigned long long* flag; unsigned long long seq; int* cnt; int set; double* odo; double poses[6]; double a[4]; };
// start of a step-wise frame whose predecessor's tail prepared the control and projected it (k_set_step with fresh = 0; one thread)
__device__ __forceinline__ void srukf_step_scalars(FrameScalars* fs, double* odo, int seqF, const double (&a)[4], bool clear_frozen = true)
{
    fs->odo_seq = odo; fs->seqF = seqF;
    fs->a[0] = a[0]; fs->a[1] = a[1]; fs->a[2] = a[2]; fs->a[3] = a[3];
    fs->frame = 0;
    fs->traj_base = nullptr;
    fs->stat_count = 0;
    for (int q = 0; q < SRUKF_STAT_GROUPS; q++) fs->stat_cnt[q] = 0;
    fs->clamp_rows = 0; fs->clamp_first = 0x7fffffff; fs->clamp_frame = 0x7fffffff; fs->gmw_aborts = 0;
    if (clear_frozen) fs->frozen = 0;
}

// ---- agent-scope (device-coherent) accesses: data handed from one workgroup to another INSIDE a launch ----
// The eight XCDs have private, mutually non-coherent L2s; plain stores stay dirty in the writer's L2 until the kernel
// ends.  Relaxed agent-scope atomics compile to sc1 loads / stores, which bypass / write through it
// (scripts/mb/mb_xwg.hip: an 8 KB tile + flag hand-off between two workgroups costs ~2 us, never a stale word).
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool DEV> __device__ __forceinline__ double ld_g(const double* p) { if constexpr (DEV) return ld_dev(p); else return *p; }
template <bool DEV> __device__ __forceinline__ void st_d4(double* p, d4 v)
{
    if constexpr (DEV) { st_dev(p, v[0]); st_dev(p + 1, v[1]); st_dev(p + 2, v[2]); st_dev(p + 3, v[3]); }
    else *(d4*)p = v;
}

// ---- synchronisation block of the persistent GMW launch (k_gmw_persist) ----
// Flags carry (epoch << GMW_EPOCH_SHIFT) + count, so the T*T tile versions never need clearing: values of older runs
// compare as "not set".  The last workgroup to leave a launch re-arms the block and advances the epoch.
#define GMW_EPOCH_SHIFT 12
#define GMW_FLAG_COPIES 16          // the two panel flags are polled by every worker at once: one copy per 16 workgroups,
#define GMW_FLAG_STRIDE 512         // 4 KB apart (unsigned long longs), so that the polls do not all queue on one memory channel
struct GmwSync {
    unsigned long long epoch;        // run counter (starts at 1)
    unsigned int exited;             // workgroups that have left the current launch
    int abort;                       // a bounded wait expired: everybody leaves, the frame is flagged for the exact path
    unsigned long long* dbg;         // diagnostic builds: host-visible progress markers (null in the product)
    unsigned int head_done;          // head fold: 32 x 32 tiles of the head rows of S^T S - U U^T finished by the helper workgroups of this launch
    unsigned int head_crit;          // ... and the first ha.ncrit of them: what the pivot needs before its first panel
    unsigned int resident;           // split form (k_gmw_pivslab_persist + k_gmw_tiles_persist): pivot / slab workgroups that have started — the tile launch waits for all of them
    unsigned int pad32;
    unsigned long long pad[59];
    unsigned long long panel_ready[GMW_FLAG_COPIES * GMW_FLAG_STRIDE];  // copy c at [c * STRIDE]: (epoch << SHIFT) + panels published
    unsigned long long half_ready[GMW_FLAG_COPIES * GMW_FLAG_STRIDE];   // same for the first half of a panel buffer (Tt1, E, pivots of sub-panel 1)
    // followed by unsigned long long ver[T*T] (one word per 128-byte line: GMW_VIDX): (epoch << SHIFT) + number of panel updates applied to tile (I, J)
    // followed by unsigned long long slabver[T*T] (split form): (epoch << SHIFT) + 1 once the slabs of panel k for column block J are in Wslab / Lslab
};
// The two head-fold counters live in words of their own, 2 KB behind copy 0 of the panel flags / of the half flags (unused padding of those arrays: zero like the rest of the
// block): as fields next to epoch / exited / abort they shared one line with everything every workgroup of the launch touches.
__device__ __forceinline__ unsigned int* gmw_head_crit(GmwSync* sy) { return (unsigned int*)&sy->panel_ready[GMW_FLAG_STRIDE / 2]; }
__device__ __forceinline__ unsigned int* gmw_head_done(GmwSync* sy) { return (unsigned int*)&sy->half_ready[GMW_FLAG_STRIDE / 2]; }
__device__ __forceinline__ unsigned long long* gmw_sync_ver(GmwSync* sy) { return (unsigned long long*)(sy + 1); }
// Every version word has a 128-byte line to itself: published by one workgroup, polled by others, and neighbours in one line queue at the memory side behind each
// other's polls (the head fold's counters: 10 us).  GMW_VIDX(I, J, T): index of tile (I, J)'s word.
#define GMW_VER_STRIDE 16
#define GMW_VIDX(I, J, T) (((size_t)(I) * (T) + (J)) * GMW_VER_STRIDE)
__device__ __forceinline__ unsigned long long* gmw_sync_slabver(GmwSync* sy, int T) { return (unsigned long long*)(sy + 1) + (size_t)T * T * GMW_VER_STRIDE; }
// ... followed by unsigned long long formver[T*T] (split fold): (epoch << SHIFT) + the 32 x 32 quarters of tile (I, J) that the forming jobs of the tile launch have stored
__device__ __forceinline__ unsigned long long* gmw_sync_formver(GmwSync* sy, int T) { return (unsigned long long*)(sy + 1) + (size_t)2 * T * T * GMW_VER_STRIDE; }

#ifdef SRUKF_GMW_DBG
#define GMW_DBG(sy, slot, val) do { if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0 && (sy)->dbg) __hip_atomic_store(&(sy)->dbg[blockIdx.x * 8 + (slot)], (unsigned long long)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#define GMW_DBG2(sy, slot, val) do { if ((sy)->dbg) __hip_atomic_store(&(sy)->dbg[blockIdx.x * 8 + (slot)], (unsigned long long)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// time stamp (s_memtime) of pivot iteration p, slot 0..7, written by whichever wave executes it
#define GMW_TS(sy, p, slot) do { if ((sy)->dbg) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); (sy)->dbg[2048 + (p) * 8 + (slot)] = t_; } } while (0)
#else
#define GMW_DBG(sy, slot, val)
#define GMW_DBG2(sy, slot, val)
#define GMW_TS(sy, p, slot)
#endif

// ---- optional cycle stamps (diagnostic builds under scripts/mb only; compiled out of the product) ----
#ifdef SRUKF_STAMPS
__device__ unsigned long long srukf_stamps[32];
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); srukf_stamps[i] = t_; } } while (0)
#define STAMPW(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); srukf_stamps[i] = t_; } } while (0)
#else
#define STAMP(i)
#define STAMPW(i)
#endif

// ---- batched launches (srukf_run_frames_batch): per-filter arguments of the B-wide kernels, one table entry per filter in device memory ----
struct Step64Args { double* G; double* Sout; double* D; void* pan[2]; double* Wb; double* Lb; };   // Wb / Lb: the panel's slabs in global memory (64 x ld each)
struct SyrkOwnArgs { const double* S0; const double* Ut0; double* G; FrameScalars* fs; };

// ---- launch prototypes (host side, implemented in the .hip files) ---------------------------
struct LaunchCtx;   // defined in srukf_api.hip
