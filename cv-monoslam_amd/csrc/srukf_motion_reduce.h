// srukf_motion_reduce.h — tail of the motion step (block reduction, robot mean, R22 = chol(C^T C)) and its "table" form, a reduction
// over the prepared table of robot poses: shared by srukf_predict.hip (k_motion, k_project_motion, k_project_table) and by the frame
// tail of the rank-aware replay (k_rank_expand, srukf_rank.hip), which runs the NEXT frame's reduction when the persistent
// factorisation launch has projected that frame already.  See srukf_predict.hip for the structured re-triangularisation.
#pragma once
#include "srukf_device.h"
#include "srukf_rank.h"

// Tail of the motion step: block reduction of the 14 sums (fixed order: 16-lane DPP rows on the VALU, then the NT/16 row sums of
// each value by one thread), robot mean, the constant the statistics need, R22 = chol(C^T C).  NT = threads of the workgroup.
template <bool REPLAY, int NT>
__device__ __forceinline__ void motion_finish(double (&acc)[14], const double (&s0)[4], const KDims& d, const KWeights& w,
                                              double* __restrict__ X, double* __restrict__ S, double* __restrict__ sigR, double* __restrict__ Cm,
                                              FrameScalars* __restrict__ fs, const RankArgs& ra, double* smem)
{
#pragma clang fp contract(off)
    // (every fused multiply-add of this function and of motion_reduce_body is written out: they are compiled into several kernels — k_motion, k_project_motion,
    //  k_project_table, k_pxy2, k_rank_expand — and must give the same bits in each)
    double (*part)[32] = (double (*)[32])smem;                 // [14][32]
    double* red = smem + 14 * 32;                              // [16]
    const int tid = threadIdx.x;
    const int n = d.n, Na = d.Na, L = d.L, ld = d.np;
    STAMP(2);
    // block reduction in fixed order (deterministic): 16-lane DPP rows on the VALU, then the 32 row sums of each value by one thread
#pragma unroll
    for (int q = 0; q < 14; q++) acc[q] = row16_sum(acc[q]);
    if ((tid & 15) == 0) {
#pragma unroll
        for (int q = 0; q < 14; q++) part[q][tid >> 4] = acc[q];
    }
    __syncthreads();
    if (tid < 14) {
        double t = 0.0;
        for (int e = 0; e < NT / 16; e++) t += part[tid][e];
        red[tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 14; q++) acc[q] = red[q];
    STAMP(3);
    if (tid < 4) {
        if constexpr (REPLAY) fs->Xr1[tid] = acc[tid]; else X[n - 4 + tid] = acc[tid];   // 1531
        // rs[e] = sum_c wc_c (sigma_c[e] - X[e]) with the covariance weights (wc0 for the centre): the constant
        // k_meas_final needs to re-centre the robot rows of Pxy on the mean h.
        //   sum_c wc_c sigma_c = mean + (wc0 - wm0) sigma_0,  sum_c wc_c = wc0 + 2 Na wi
        sigR[(size_t)L * 8 + tid] = fma(acc[tid], 1.0 - (w.wc0 + 2.0 * Na * w.wi), (w.wc0 - w.wm0) * s0[tid]);
    }
    STAMP(4);
    double* g = acc + 4;
    // ---- R22 = chol(C^T C), upper triangular ----
    if (tid == 0) {
        double G4[4][4], R[4][4];
        int q = 0;
        for (int a = 0; a < 4; a++) for (int b = a; b < 4; b++) { G4[a][b] = g[q]; G4[b][a] = g[q]; q++; }
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) R[a][b] = 0.0;
        for (int a = 0; a < 4; a++) {
            double dsum = G4[a][a];
            for (int k = 0; k < a; k++) dsum = fma(-R[k][a], R[k][a], dsum);
            const double raa = sqrt(fmax(dsum, 0.0));
            R[a][a] = raa;
            for (int b = a + 1; b < 4; b++) {
                double v = G4[a][b];
                for (int k = 0; k < a; k++) v = fma(-R[k][a], R[k][b], v);
                R[a][b] = (raa > 0.0) ? v / raa : 0.0;
            }
        }
        if constexpr (REPLAY) {
            for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) Cm[(size_t)(n - 4 + a) * 4 + b] = R[a][b];
        } else {
            for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) S[(size_t)(n - 4 + a) * ld + (n - 4 + b)] = R[a][b];
            if (ra.A) for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) ra.A[(size_t)(ra.r - 4 + a) * ld + (ra.r - 4 + b)] = R[a][b];
        }
    }
}

#define MOTION_SM_DOUBLES (14 * 32 + 16 + 24)

// Replay path of the rank-aware form ("table" mode): the robot part of every sigma point of this frame is already in sigR — the
// previous frame's tail (k_rank_expand) wrote each row's pair when it wrote the row of S, the run's first frame gets it from
// k_sigr_rows — so what is left of the motion step is sums over that table: no sincos, one pass, then motion_finish.  Workgroup 0
// of k_project_table; results wait in fs->Xr1 / Cm like those of motion_body<true>.
// PREAMBLE = false: the frame tail runs the reduction for the NEXT frame ("tail" mode, k_rank_expand): the previous frame's flags
// are still being written by that very launch, so what the first launch of a frame does about them (freeze the run behind a flagged
// frame, promote const_rows_pending) is srukf_frame_preamble, called by the frame's first launch (k_pxy2).
__device__ __forceinline__ void srukf_frame_preamble(FrameScalars* __restrict__ fs)
{
    fs->const_rows_ok = fs->const_rows_pending;
    if (fs->clamp_rows > 0) {
        if (fs->clamp_frame == 0x7fffffff) fs->clamp_frame = fs->frame - 1;
        fs->frozen = 1;
    }
}
template <int NT, bool PREAMBLE = true>
__device__ __forceinline__ void motion_reduce_body(const KDims& d, const KWeights& w, double* __restrict__ X, double* __restrict__ S,
                                                   double* __restrict__ sigR, double* __restrict__ Cm, FrameScalars* __restrict__ fs, const RankArgs& ra, double* smem)
{
#pragma clang fp contract(off)
    const int tid = threadIdx.x;
    const int n = d.n, Na = d.Na;
    if (fs->frozen) return;
    __builtin_amdgcn_s_setprio(3);
    bool freeze = false;
    if (tid == 0) {
        if constexpr (PREAMBLE) fs->const_rows_ok = fs->const_rows_pending;   // the previous frame's tail has written the constant rows of S: its successors may skip them
        for (int q = 0; q < 3; q++) { fs->Ut[q] = fs->ctl[q]; fs->Mt[q] = fs->ctl[5 + q]; }
        for (int q = 0; q < 4; q++) fs->Xr0[q] = X[n - 4 + q];
        if (PREAMBLE && fs->clamp_rows > 0) {                  // the previous frame's refactorisation was flagged -> remember which
            if (fs->clamp_frame == 0x7fffffff) fs->clamp_frame = fs->frame - 1;
            freeze = true;
        }
    }
    const double4 c0 = *reinterpret_cast<const double4*>(sigR);
    const double s0[4] = { c0.x, c0.y, c0.z, c0.w };
    const double k2 = w.wi_sr * 0.70710678118654752440;
    double acc[14];
#pragma unroll
    for (int q = 0; q < 14; q++) acc[q] = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = w.wm0 * s0[e];
    }
    // one direction per trip, the next trip's two 32-byte rows requested before this trip's arithmetic
    double4 rp = make_double4(0, 0, 0, 0), rm = rp;
    if (tid < Na) { rp = *reinterpret_cast<const double4*>(sigR + (size_t)(1 + tid) * 8); rm = *reinterpret_cast<const double4*>(sigR + (size_t)(1 + Na + tid) * 8); }
    for (int i = tid; i < Na; i += NT) {
        const double r0[4] = { rp.x, rp.y, rp.z, rp.w }, r1[4] = { rm.x, rm.y, rm.z, rm.w };
        if (i + NT < Na) { rp = *reinterpret_cast<const double4*>(sigR + (size_t)(1 + i + NT) * 8); rm = *reinterpret_cast<const double4*>(sigR + (size_t)(1 + Na + i + NT) * 8); }
        double dm[4], dp[4];                                   // dev+ - dev-, dev+ + dev-
#pragma unroll
        for (int e = 0; e < 4; e++) { acc[e] = fma(w.wi, r0[e], acc[e]); acc[e] = fma(w.wi, r1[e], acc[e]); }
        if (i < n - 4) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const double d0 = r0[e] - s0[e], d1 = r1[e] - s0[e]; dm[e] = k2 * (d0 - d1); dp[e] = k2 * (d0 + d1); }
            *reinterpret_cast<double4*>(Cm + (size_t)i * 4) = make_double4(dm[0], dm[1], dm[2], dm[3]);
            int q = 4;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++) { acc[q] = fma(dp[a], dp[b], acc[q]); q++; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) { dm[e] = w.wi_sr * (r0[e] - s0[e]); dp[e] = w.wi_sr * (r1[e] - s0[e]); }
            int q = 4;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++) { acc[q] += fma(dm[a], dm[b], dp[a] * dp[b]); q++; }
        }
    }
    __syncthreads();                                           // every wave has passed the frozen test at the top
    if (freeze) fs->frozen = 1;
    motion_finish<true, NT>(acc, s0, d, w, X, S, sigR, Cm, fs, ra, smem);
}

