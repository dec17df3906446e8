// srukf_rank.hip — rank-aware refactorisation: the structurally null pivots are not factored.
//
// The anchors (xi, yi, zi) of landmarks initialised in one batch are copies of ONE robot position (SLAM.cpp:1223, 1247):
// identical random variables for the rest of the filter's life, since every Kalman update preserves the equality.  Of the
// n = 6N + 4 pivots of G = S^T S - U U^T, 3 (K - 1) per batch of K landmarks are therefore null by construction; the
// reference's modifiedCholeskyDecomposition meets them as c_jj = 0 (+ rounding) and ends at D_j = EPSILON (SLAM.cpp:2279-2285),
// their multipliers L = C / D are rounding noise, and their rank-1 updates of the trailing matrix are of size
// (1e-17)^2 / 1e-13.  They cost the dependent pivot chain exactly as much as any other pivot.
//
// Rows of S whose energy sum_i S[k][i]^2 is below SRUKF_NULL_ENERGY = 1e-12 are such directions (dropping the row changes no entry
// of P = S^T S by more than 1e-12, Cauchy-Schwarz: inside the 1e-11 the parity tests hold P to).  The refactorisation then runs
// on G permuted so that these indices come last, in the same relative order otherwise:
//     Gp = Pi^T G Pi  ->  only the leading Tp = ceil(r / 64) panels are pivoted, all n columns are carried along
//     (k_gmw_persist with Tp < T)  ->  k_rank_expand: kept rows go back to state order (an upper triangular row: what stood left
//     of the diagonal in state order is the residual of a column that is a linear combination of earlier ones, i.e. zero),
//     dropped rows become sqrt(EPSILON) e_k — what the reference's clamp leaves there.
// The kept rows are, operation for operation, the rows the reference's state-order factorisation produces (the skipped pivots'
// updates are the (1e-17)^2 / 1e-13 terms), so the sigma set of the next frame is the reference's.  k_rank_expand also verifies
// the assumption for every dropped index: G_kk - sum_{a<r} Sp[a][k]^2 <= 1e-12, else the frame is flagged like a theta-clamp
// frame and repeated on the exact column path.
//
// Replay path (k_rank_shadow, "shadow" below): the kept rows are also held in PERMUTED column order in a second buffer A
// (r rows, upper triangular in that order, rows >= r zero), which k_motion and k_rank_expand keep in step with S, and k_gain
// writes U^T with permuted columns next to U^T.  S^T S - U U^T is then formed directly in permuted order — the head rows by
// k_syrk, the other tiles by their owners inside the persistent launch, with K = r instead of n — and neither the full k_syrk
// nor the permutation pass runs.  The diagonal of the dropped indices, which no tile covers, comes from srukf_rank_gdiag_job.
#include <hip/hip_runtime.h>
#include "srukf_device.h"
#include "srukf_rank.h"
#include "srukf_motion.h"
#include "srukf_motion_reduce.h"

// A[a][b] = S[perm[a]][perm[b]] for a < r (the kept rows in permuted column order), zero rows below
__global__ __launch_bounds__(256) void k_rank_shadow(int n, int ld, int r, const double* __restrict__ S, const int* __restrict__ perm, double* __restrict__ A)
{
    const int a = blockIdx.x;
    double* out = A + (size_t)a * ld;
    if (a >= r) { for (int b = threadIdx.x; b < ld; b += 256) out[b] = 0.0; return; }
    const double* src = S + (size_t)perm[a] * ld;
    for (int b = threadIdx.x; b < ld; b += 256) out[b] = (b >= a && b < n) ? src[perm[b]] : 0.0;
}

// e[k] = sum_i S[k][i]^2, one workgroup per row
__global__ __launch_bounds__(256) void k_row_energy(int n, int ld, const double* __restrict__ S, double* __restrict__ e)
{
    __shared__ double red[16];
    const int k = blockIdx.x;
    double v[1] = { 0.0 };
    for (int i = k + threadIdx.x; i < n; i += 256) { const double s = S[(size_t)k * ld + i]; v[0] += s * s; }
    block_sum<1>(v, red);
    if (threadIdx.x == 0) e[k] = v[0];
}

// gdiag[a] = G[perm[a]][perm[a]] (the diagonal is overwritten by the in-place factorisation)
__global__ __launch_bounds__(256) void k_rank_diag(int n, int ld, const double* __restrict__ G, const int* __restrict__ perm, double* __restrict__ gdiag)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < n) { const int j = perm[a]; gdiag[a] = G[(size_t)j * ld + j]; }
}

// One workgroup per row (permuted position a; state row perm[a]), one for the frame tail, and one per 16 dropped indices for the null-direction check.
// Sp: factor rows in permuted order (row a < r valid for columns b >= a), D: pivots in permuted order, perm[a] = state index at
// permuted position a, iperm = inverse.  A (may be null): the shadow copy of the kept rows in permuted order.
// MODE 2 ("fused tail", the default of the exclusive rank-aware replay): every workgroup ALSO projects the next frame's sigma points of its direction
// (passSigmaThroughMesaurementFunction, SLAM.cpp:1615-1690) — the row is in flight anyway, its robot part has just gone through the motion model, the
// new mean is final — so that the next frame needs no projection launch: the row through LDS, one thread per landmark and +- pair (project_dir's
// arithmetic, srukf_project_sigma).  Null rows: their own landmark; the frame tail: the centre point; five more workgroups: the noise rows.
// The frame's motion reduction then rides on k_pxy2 (MeasArgs::fmode).  Dynamic LDS: ld doubles.
template <int MODE>
__device__ __forceinline__ void rank_expand_body(int n, int ld, int r, double eps, const double* __restrict__ Sp, const double* __restrict__ D,
                                                 const int* __restrict__ perm, const int* __restrict__ iperm, const double* __restrict__ gdiag,
                                                 FrameScalars* __restrict__ fs, const double* __restrict__ X, int do_traj, double* __restrict__ S,
                                                 double* __restrict__ A, double* __restrict__ sigR, double gamma,
                                                 const KDims& d, const KWeights& w,
                                                 const srukf_params& p, double* __restrict__ Z, double* __restrict__ DZ, int f32, const int bid, const StepExport& ex, const double null_rel = 0.0,
                                                 unsigned int* __restrict__ fold_sync = nullptr, const int fold_words = 0)
{
    constexpr bool PROJ = MODE == 2;
    // f32 (fp32 storage, "fused tail" mode): every value this launch writes into S / the permuted copy — and reads back for the table, the projection and
    // the trajectory row — is rounded to float first: the rounding points of k_quantize / k_rank_round, without their launches
    auto rnd = [&](double v) { return f32 ? (double)(float)v : v; };
    extern __shared__ double lrow[];                           // PROJ: the workgroup's row of the factor (permuted order)
    __shared__ double prow[2][8];                              // PROJ: robot part of the direction's two sigma points (what goes into the table)
    __shared__ double red[16 * 3];
    const int j = bid;                                         // < n: row (permuted position), n: frame tail, > n: null checks (+ PROJ: noise rows)
    // "Table" mode of the replay (sigR != null): the workgroup that writes row j of S also pushes the NEXT frame's two sigma points
    // of direction j through the motion model — robot part only: pose before the step X[n-4..], the row's entries in the robot
    // columns, the control k_gain prepared in fs->ctl — and leaves them in the table the next k_project_table launch reads.
    // Two lanes per workgroup, hidden behind the row copy; the frame tail does the centre point and the five noise rows.
    const int Na = n + 5;
    const bool table = sigR && fs->ctl_next_valid && !fs->frozen;
    auto table_rows = [&](const int i, const int sg, const double (&srow)[4], const double (&mnoise)[3], const bool isnull) {
        const MotionCtl mc = { fs->ctl[0], fs->ctl[1], fs->ctl[2], fs->ctl[3], fs->ctl[4] };
        const double xr[4] = { X[n - 4], X[n - 3], X[n - 2], X[n - 1] };
        double rr[4], c2, s2;
        if (isnull) srukf_motion_centre(mc, xr, rr, c2, s2);    // a structurally null row: both points ARE the centre point (same bits: NullSkip rests on it)
        else srukf_motion_point(mc, xr, srow, mnoise, sg ? -gamma : gamma, rr, c2, s2);
        double4* o = reinterpret_cast<double4*>(sigR + (size_t)(1 + sg * Na + i) * 8);
        o[0] = make_double4(rr[0], rr[1], rr[2], rr[3]); o[1] = make_double4(c2, s2, 0.0, 0.0);
        if constexpr (PROJ) { double* q = prow[sg]; q[0] = rr[0]; q[1] = rr[1]; q[2] = rr[2]; q[3] = rr[3]; q[4] = c2; q[5] = s2; }
    };
    // PROJ: landmark k under direction i (dev = the direction's six entries for that landmark, e0 / e1 = pixel-noise rows): both points, Z rows, DZ row
    const double f1 = PROJ ? p.cam_f / p.cam_dx : 0.0, f2 = PROJ ? p.cam_f / p.cam_dy : 0.0;
    auto project_pair = [&](const int i, const int k, const double (&dev)[6], const double e0, const double e1, const double* r0, const double* r1, const int dzrow) {
        double base[6];
#pragma unroll
        for (int e = 0; e < 6; e++) base[e] = X[6 * k + e];
        double zp[2], zm[2];
        srukf_project_sigma(p, f1, f2, base, dev, e0, e1, gamma, r0, zp[0], zp[1]);
        srukf_project_sigma(p, f1, f2, base, dev, e0, e1, -gamma, r1, zm[0], zm[1]);
        *reinterpret_cast<double2*>(Z + (size_t)(1 + i) * d.mp + 2 * k) = make_double2(zp[0], zp[1]);
        *reinterpret_cast<double2*>(Z + (size_t)(1 + Na + i) * d.mp + 2 * k) = make_double2(zm[0], zm[1]);
        if (dzrow >= 0) *reinterpret_cast<double2*>(DZ + (size_t)dzrow * d.mp + 2 * k) = make_double2(zp[0] - zm[0], zp[1] - zm[1]);
    };
    if constexpr (PROJ) {
        // five extra workgroups behind the null checks: the noise rows (directions n .. n+4) for every landmark
        const int nchk = (n - r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;
        if (j > n + nchk) {
            if (!table) return;
            const int q = j - n - nchk - 1;
            if (threadIdx.x < 2) {
                const double zero4[4] = { 0, 0, 0, 0 };
                double mnoise[3] = { 0, 0, 0 };
                if (q < 3) mnoise[q] = fs->ctl[5 + q];
                table_rows(n + q, threadIdx.x, zero4, mnoise, false);
            }
            __syncthreads();
            const double zero6[6] = { 0, 0, 0, 0, 0, 0 };
            const double e0 = (q == 3) ? p.sigma_measure : 0.0, e1 = (q == 4) ? p.sigma_measure : 0.0;
            for (int k = threadIdx.x; k < d.N; k += 256) project_pair(n + q, k, zero6, e0, e1, prow[0], prow[1], -1);
            return;
        }
    }
    if (j == n) {
        // (the gain fold of k_pxy2 counts in these words; they are zero between frames: cleared here, in the launch that follows the frame's k_pxy2 and precedes the next one's)
        if (fold_sync) {                                       // (fold_words: the pair counters, contiguous; the polled flags stand FOLD_FLAG_STRIDE apart behind them)
            const int base = (fold_words + FOLD_FLAG_STRIDE - 1) / FOLD_FLAG_STRIDE * FOLD_FLAG_STRIDE;
            for (int q = threadIdx.x; q < fold_words; q += 256) fold_sync[q] = 0u;
            if (threadIdx.x < 2 + SRUKF_STAT_GROUPS) fold_sync[base + FOLD_FLAG_STRIDE * threadIdx.x] = 0u;
        }
        if (table && threadIdx.x < (PROJ ? 1 : 11)) {
            const double zero4[4] = { 0, 0, 0, 0 };
            if (threadIdx.x == 0) {
                const MotionCtl mc = { fs->ctl[0], fs->ctl[1], fs->ctl[2], fs->ctl[3], fs->ctl[4] };
                const double xr[4] = { X[n - 4], X[n - 3], X[n - 2], X[n - 1] };
                double s0[4], c0s, s0s;
                srukf_motion_centre(mc, xr, s0, c0s, s0s);
                double4* o = reinterpret_cast<double4*>(sigR);
                o[0] = make_double4(s0[0], s0[1], s0[2], s0[3]); o[1] = make_double4(c0s, s0s, 0.0, 0.0);
                if constexpr (PROJ) { double* q = prow[0]; q[0] = s0[0]; q[1] = s0[1]; q[2] = s0[2]; q[3] = s0[3]; q[4] = c0s; q[5] = s0s; }
            } else {
                const int q = (threadIdx.x - 1) >> 1, sg = (threadIdx.x - 1) & 1;       // noise row n + q (control noise: q < 3)
                double mnoise[3] = { 0, 0, 0 };
                if (q < 3) mnoise[q] = fs->ctl[5 + q];
                table_rows(n + q, sg, zero4, mnoise, false);
            }
        }
        if constexpr (PROJ) {
            // the centre point of the next frame for every landmark (row 0 of Z)
            __syncthreads();
            if (table) {
                const double f1c = p.cam_f / p.cam_dx, f2c = p.cam_f / p.cam_dy;
                for (int k = threadIdx.x; k < d.N; k += 256) {
                    double base[6];
#pragma unroll
                    for (int e = 0; e < 6; e++) base[e] = X[6 * k + e];
                    double ox, oy;
                    srukf_project(p, f1c, f2c, base, prow[0][0], prow[0][1], prow[0][2], prow[0][4], prow[0][5], 0.0, 0.0, ox, oy);
                    *reinterpret_cast<double2*>(Z + 2 * k) = make_double2(ox, oy);
                }
            }
        }
        // frame tail: RobotPath.txt row (SLAM.cpp:3549-3556) with P = S^T S restricted to the robot x / y block (2404): the
        // dropped rows have no entry in the robot columns
        const int bx = iperm[n - 4], by = iperm[n - 3];
        double v[3] = { 0, 0, 0 };
        for (int a = threadIdx.x; a < r; a += 256) {
            const double p = (bx >= a) ? rnd(Sp[(size_t)a * ld + bx]) : 0.0, q = (by >= a) ? rnd(Sp[(size_t)a * ld + by]) : 0.0;
            v[0] += p * p; v[1] += p * q; v[2] += q * q;
        }
        block_sum<3>(v, red);
        if (ex.dst) {
            // step-wise API: the whole 4 x 4 robot block (what srukf_get_robot hands out, SLAM.cpp:3539-3556) from the same walk over the kept rows, and the pose: the view
            // the host gets with the frame's status.  Device-scope stores: the workgroup that exports them may sit on another XCD.
            __shared__ double red4[16 * 10];
            int bc[4];
#pragma unroll
            for (int e = 0; e < 4; e++) bc[e] = iperm[n - 4 + e];
            double v4[10];
#pragma unroll
            for (int q = 0; q < 10; q++) v4[q] = 0.0;
            for (int a = threadIdx.x; a < r; a += 256) {
                double s4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) s4[e] = (bc[e] >= a) ? rnd(Sp[(size_t)a * ld + bc[e]]) : 0.0;
                int q = 0;
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int g = e; g < 4; g++) v4[q++] += s4[e] * s4[g];
            }
            block_sum<10>(v4, red4);
            if (threadIdx.x == 0) {
                int q = 0;
                for (int e = 0; e < 4; e++)
                    for (int g = e; g < 4; g++) { st_dev(ex.view + 4 * e + g, v4[q]); st_dev(ex.view + 4 * g + e, v4[q]); q++; }
                for (int e = 0; e < 4; e++) st_dev(ex.view + 16 + e, X[n - 4 + e]);
            }
        }
        if (threadIdx.x == 0 && do_traj) {
            double* traj = fs->traj_base;
            if (traj) {
                double* t = traj + (size_t)8 * fs->frame;
                for (int e = 0; e < 4; e++) t[e] = X[n - 4 + e];
                t[4] = v[0]; t[5] = v[1]; t[6] = v[1]; t[7] = v[2];
            }
            if (ex.dst) {                                      // (the workgroup that exports *fs may sit on another XCD: device-scope stores)
                __hip_atomic_store(&fs->frame, fs->frame + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&fs->const_rows_pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                fs->frame += 1;
                fs->const_rows_pending = 1;                     // this launch wrote (or found) every structurally null row of S as sqrt(EPSILON) e_k
            }
            if (!sigR) srukf_prepare_control(fs);              // control of the next staged frame (k_project_motion); "table" mode: k_gain did it
        }
        return;
    }
    if (j > n) {
        // the check that 16 dropped directions really are null in this frame's G: G_aa - sum_{k<r} Sp[k][a]^2 <= 1e-12
        const int a = r + SRUKF_RANK_COLS * (j - n - 1) + (threadIdx.x & 15), kl = threadIdx.x >> 4;
        __shared__ double cs[16][17];
        cs[kl][threadIdx.x & 15] = (a < n) ? srukf_rank_colsq(Sp, ld, r, a, kl) : 0.0;
        __syncthreads();
        if (threadIdx.x < 16 && a < n) {
            double t = 0.0;
            for (int q = 0; q < 16; q++) t += cs[q][threadIdx.x];
            // (null_rel: the mixed-precision downdate — its G carries fp32 product rounding, ~1e-7 of the entries' scale: 1e-12 absolute is below what it can resolve)
            if (gdiag[a] - t > 1e-12 + null_rel * fabs(gdiag[a])) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, perm[a]); }
        }
        return;
    }
    // One workgroup per PERMUTED position a = blockIdx (row a of the factor, state row j = perm[a]): the row's loads depend on nothing but the
    // block index, and j and the columns' state indices perm[b] — needed for the store addresses only — arrive with them: one memory round trip
    // in front of the stores.  (Indexed by the state row, the chain was iperm[j] -> iperm[c] -> src[iperm[c]]: three of them, ~2 us each in a freshly
    // launched grid.)
    const int a = j;                                           // (the dispatch index above)
    const int jj = perm[a];                                    // state row
    double* out = S + (size_t)jj * ld;
    if (table && threadIdx.x < 2) {
        double srow[4] = { 0, 0, 0, 0 };
        const double zero3[3] = { 0, 0, 0 };
        if (a < r) {
#pragma unroll
            for (int e = 0; e < 4; e++) srow[e] = (r - 4 + e >= a) ? rnd(Sp[(size_t)a * ld + (r - 4 + e)]) : 0.0;    // the robot columns: permuted positions r-4 .. r-1
        }
        table_rows(jj, threadIdx.x, srow, zero3, a >= r);
    }
    if (a >= r) {                                              // dropped direction: what the reference's clamp leaves
        if constexpr (PROJ) {
            // its two sigma points move ONE landmark (NullSkip, srukf_device.h): jj / 6, by sqrt(EPSILON) gamma in one anchor coordinate
            if (table) {
                __syncthreads();                               // prow: the centre point's robot part (table_rows above, isnull)
                if (threadIdx.x < 2) {                         // one lane per sign (the pair in one lane is twice the latency, and 597 workgroups do nothing else)
                    const int k = jj / 6, sg = threadIdx.x;
                    double dev[6], base[6];
#pragma unroll
                    for (int e = 0; e < 6; e++) { dev[e] = (6 * k + e == jj) ? rnd(sqrt(eps)) : 0.0; base[e] = X[6 * k + e]; }
                    double ox, oy;
                    srukf_project_sigma(p, f1, f2, base, dev, 0.0, 0.0, sg ? -gamma : gamma, prow[sg], ox, oy);
                    *reinterpret_cast<double2*>(Z + (size_t)(1 + sg * Na + jj) * d.mp + 2 * k) = make_double2(ox, oy);
                    const double pox = __shfl_xor(ox, 1), poy = __shfl_xor(oy, 1);
                    if (sg == 0) *reinterpret_cast<double2*>(DZ + (size_t)a * d.mp + 2 * k) = make_double2(ox - pox, oy - poy);
                }
            }
        }
        // the same row every frame: written by the first frame of a staged run only (fs->const_rows_ok: set for the frames after it
        // by the next launch, cleared by k_set_run / k_set_frame; whoever else rewrites S goes through one of those first)
        if (sigR && fs->const_rows_ok) return;
        for (int c = threadIdx.x; c < ld; c += 256) out[c] = (c == jj) ? rnd(sqrt(eps)) : 0.0;
        return;
    }
    const double* src = Sp + (size_t)a * ld;
    double* sh = A ? A + (size_t)a * ld : nullptr;
    double mx = 0.0;
    // columns left of the diagonal hold zeros in S (every writer keeps the strictly lower triangle zero) and in the permuted copy:
    // only the row's upper part is written ("table" mode; the other callers rewrite the whole row as before).  A state column c = perm[b]
    // left of the diagonal (c < jj) can only be a dropped one (b >= r): the residual of a column that is a copy of an earlier one, zero.
    // (four trips' loads are requested before the first store: a row is one or two memory round trips, not one per 256 columns)
    for (int b0 = (sigR ? (a & ~3) : 0) + threadIdx.x; b0 < ld; b0 += 4 * 256) {
        double sv[4]; int cc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = b0 + 256 * u;
            sv[u] = (b >= a && b < n) ? src[b] : 0.0;
            cc[u] = (b < n) ? perm[b] : b;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = b0 + 256 * u;
            if (b >= ld) break;
            const bool upper = b >= a && b < n && cc[u] >= jj;
            if (upper && cc[u] > jj) mx = fmax(mx, fabs(sv[u]));          // (the theta check looks at the factor itself)
            const double rv = rnd(sv[u]);
            if (sh) sh[b] = rv;
            if constexpr (PROJ) lrow[b] = rv;
            if (upper || !sigR) out[cc[u]] = upper ? rv : 0.0;
        }
    }
    // theta clamp of the reference evaluated afterwards, as k_gmw_check does (SLAM.cpp:2204-2211, 2264-2285)
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double gamma = __longlong_as_double((long long)fs->gmax_bits);
        const double xi = __longlong_as_double((long long)fs->ximax_bits);
        const double nu = fmax(1.0, sqrt((double)n * n - 1.0));
        const double beta2 = fmax(fmax(gamma, xi / nu), 1e-15);
        const double dj = D[a];
        const double th = mx * sqrt(dj);
        if (th * th / beta2 > dj) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, jj); }
    }
    if constexpr (PROJ) {
        // the next frame's sigma points of direction jj for every landmark (the barrier above: lrow and prow are complete)
        if (table) {
            for (int k = threadIdx.x; k < d.N; k += 256) {
                double dev[6];
#pragma unroll
                for (int e = 0; e < 6; e++) { const int col = 6 * k + e; dev[e] = (col >= jj) ? lrow[iperm[col]] : 0.0; }
                project_pair(jj, k, dev, 0.0, 0.0, prow[0], prow[1], a);
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rank_expand(int n, int ld, int r, double eps, const double* __restrict__ Sp, const double* __restrict__ D,
                                                     const int* __restrict__ perm, const int* __restrict__ iperm, const double* __restrict__ gdiag,
                                                     FrameScalars* __restrict__ fs, const double* __restrict__ X, int do_traj, double* __restrict__ S,
                                                     double* __restrict__ A, double* __restrict__ sigR, double gamma,
                                                     KDims d, KWeights w,
                                                     srukf_params p, double* __restrict__ Z, double* __restrict__ DZ, int f32, StepExport ex, double null_rel,
                                                     unsigned int* __restrict__ fold_sync, int fold_words)
{
    // dispatch order: the frame tail, the null checks and the noise rows (the longest chains of round trips: a column walk over every kept row) first, then the rows
    const int extra = (int)gridDim.x - n;
    const int bid = (int)blockIdx.x < extra ? n + (int)blockIdx.x : (int)blockIdx.x - extra;
    rank_expand_body<MODE>(n, ld, r, eps, Sp, D, perm, iperm, gdiag, fs, X, do_traj, S, A, sigR, gamma, d, w, p, Z, DZ, f32, bid, ex, null_rel, fold_sync, fold_words);
    if (!ex.dst) return;
    // step-wise API: this is the frame's last launch.  A workgroup's updates of *fs are device-scope atomics (the clamp counters) or device-scope stores (the frame
    // tail's), complete once its s_waitcnt returns.  The last workgroup through copies *fs and the view to the host with loads that bypass its own L2.
    // (The workgroups of the dropped rows — dispatch indices from extra + r on — never write *fs: they are not counted, and do not wait for their stores.)
    const int ncount = extra + r;
    if ((int)blockIdx.x >= ncount) return;
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = (int)blockIdx.x & 63, gsize = (ncount - g + 63) >> 6, groups = ncount < 64 ? ncount : 64;
        int* c1 = ex.cnt + 64 * g;
        last = 0;
        if (__hip_atomic_fetch_add(c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(c1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&fs->export_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
                __hip_atomic_store(&fs->export_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
    }
    __syncthreads();
    if (!last) return;
    const unsigned long long* fw = (const unsigned long long*)fs;
    for (int i = threadIdx.x; i < ex.nfs; i += 256) ex.dst[i] = __hip_atomic_load(fw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 20) ((double*)(ex.dst + ex.nfs))[threadIdx.x] = ld_dev(ex.view + threadIdx.x);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // every other COUNTED workgroup of the frame's last launch is through, *fs is with the host: the next frame starts here.  The workgroups of the dropped rows
        // are not counted and may still be starting: what they read of *fs (ctl, ctl_next_valid, frozen, const_rows_ok) is not touched here — `frozen` is left as it is
        // (it can only be set behind a flagged frame, and a flagged frame's successor is set up by k_set_step after the rewind, not here)
        if (ex.set && __hip_atomic_load(&fs->clamp_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            for (int e = 0; e < 6; e++) ex.odo[e] = ex.poses[e];
            srukf_step_scalars(fs, ex.odo, 1, ex.a, false);
        }
        if (ex.flag) __hip_atomic_store(ex.flag, ex.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// batched form (srukf_run_frames_batch; "fused tail" mode, fp64 storage): filter f owns workgroups [f per, (f + 1) per)
__global__ __launch_bounds__(256) void k_rank_expand_b(int n, int ld, int r, double eps, const ExpandArgs* __restrict__ tab, int per, double gamma, KDims d, KWeights w, srukf_params p)
{
    const int f = (int)blockIdx.x / per, bid = (int)blockIdx.x - f * per;
    const ExpandArgs a = tab[f];
    rank_expand_body<2>(n, ld, r, eps, a.Sp, a.D, a.perm, a.iperm, a.gdiag, a.fs, a.X, 1, a.S, a.A, a.sigR, gamma, d, w, p, a.Z, a.DZ, 0, bid, StepExport{});
}

// fp32 storage: the permuted copy holds what the stored (float) state holds, like S after k_quantize
__global__ __launch_bounds__(256) void k_rank_round(int ld, double* __restrict__ A)
{
    double* row = A + (size_t)blockIdx.x * ld;
    for (int b = blockIdx.x + threadIdx.x; b < ld; b += 256) row[b] = (double)(float)row[b];
}

extern "C" {
void srukf_launch_rank_round(hipStream_t st, int ld, int r, double* A)
{
    hipLaunchKernelGGL(k_rank_round, dim3(r), dim3(256), 0, st, ld, A);
}
void srukf_launch_row_energy(hipStream_t st, int n, int ld, const double* S, double* e)
{
    hipLaunchKernelGGL(k_row_energy, dim3(n), dim3(256), 0, st, n, ld, S, e);
}
void srukf_launch_rank_diag(hipStream_t st, int n, int ld, const double* G, const int* perm, double* gdiag)
{
    hipLaunchKernelGGL(k_rank_diag, dim3((n + 255) / 256), dim3(256), 0, st, n, ld, G, perm, gdiag);
}
void srukf_launch_rank_expand(hipStream_t st, int n, int ld, int r, double eps, const double* Sp, const double* D, const int* perm, const int* iperm,
                              const double* gdiag, void* fs, const double* X, int do_traj, double* S, double* A, double* sigR, double gamma,
                              int fuse, KDims d, KWeights w, srukf_params p, double* Z, double* DZ, int f32, const StepExport* exp, double null_rel,
                              unsigned int* fold_sync, int fold_words)
{
    const StepExport ex = (exp && fuse) ? *exp : StepExport{};
    // fuse: "fused tail" mode (projection of the next frame: five more workgroups, the row in LDS)
    const int nchk = (n - r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;
    const dim3 grid(n + 1 + nchk + (fuse ? 5 : 0));
    if (fuse) hipLaunchKernelGGL(k_rank_expand<2>, grid, dim3(256), sizeof(double) * (size_t)ld, st, n, ld, r, eps, Sp, D, perm, iperm, gdiag, (FrameScalars*)fs, X, do_traj, S, A, sigR, gamma, d, w, p, Z, DZ, f32, ex, null_rel, fold_sync, fold_words);
    else hipLaunchKernelGGL(k_rank_expand<0>, grid, dim3(256), 0, st, n, ld, r, eps, Sp, D, perm, iperm, gdiag, (FrameScalars*)fs, X, do_traj, S, A, sigR, gamma, d, w, p, Z, DZ, 0, StepExport{}, null_rel, fold_sync, fold_words);
}
void srukf_launch_rank_expand_b(hipStream_t st, int n, int ld, int r, double eps, const void* tab, int B, double gamma, KDims d, KWeights w, srukf_params p)
{
    const int nchk = (n - r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;
    const int per = n + 1 + nchk + 5;
    hipLaunchKernelGGL(k_rank_expand_b, dim3(per * B), dim3(256), sizeof(double) * (size_t)ld, st, n, ld, r, eps, (const ExpandArgs*)tab, per, gamma, d, w, p);
}
// The structurally null rows of S made what the reference's clamp leaves there and every rank-aware frame tail rewrites: sqrt(EPSILON) e_k (a factor from a path that
// factors every pivot — NEED_REORDER, the map operations — holds rounding noise / sqrt(EPSILON) beside that diagonal).  One workgroup per dropped row.
__global__ __launch_bounds__(256) void k_rank_canon(int n, int ld, int r, double sq, const int* __restrict__ perm, double* __restrict__ S)
{
    const int k = perm[r + blockIdx.x];
    if (k < 0 || k >= n) return;
    for (int c = threadIdx.x; c < ld; c += 256) S[(size_t)k * ld + c] = (c == k) ? sq : 0.0;
}
void srukf_launch_rank_canon(hipStream_t st, int n, int ld, int r, double sq, const int* perm, double* S)
{
    if (n - r > 0) hipLaunchKernelGGL(k_rank_canon, dim3(n - r), dim3(256), 0, st, n, ld, r, sq, perm, S);
}
void srukf_launch_rank_shadow(hipStream_t st, int n, int ld, int r, const double* S, const int* perm, double* A)
{
    hipLaunchKernelGGL(k_rank_shadow, dim3(ld), dim3(256), 0, st, n, ld, r, S, perm, A);
}
}  // extern "C"
