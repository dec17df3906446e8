// srukf_rank.h — rank-aware refactorisation (srukf_rank.hip): pieces other kernels carry along.
#pragma once
#include <hip/hip_runtime.h>
#include "srukf_device.h"

// What the replay path of the rank-aware form hands to k_motion / k_gain / k_syrk (all null / 0: the plain path).
struct RankArgs {
    double* A;             // shadow of the kept rows of S in permuted column order (r rows, ld columns)
    double* Utp;           // U^T with permuted columns
    double* gdiag;         // [ld] diagonal of G at the dropped positions (permuted index)
    const int* iperm;      // state index -> permuted position
    const int* perm;       // permuted position -> state index
    int r;                 // kept pivots (robot last: positions r-4 .. r-1)
    int dzperm;            // "table" mode: k_project_table writes the rows of DZ in permuted order (k_pxy2 contracts over them)
    int f32round;          // fp32 storage in "fused tail" mode: the state update rounds X to float as it writes it (k_rank_expand<2> rounds the rows of S it writes:
                           // the rounding points of k_quantize, without its launch)
    int prep_next;         // "table" mode of the replay: the k_syrk launch also prepares the NEXT frame's control in fs->ctl (this frame's motion
                           // step has consumed it; the tail that needs it must not read the frame counter it advances itself)
    int dxN;               // > 0: the pending state update arrives as per-landmark shares dxk[k][row] of dxN landmarks (the gain fold of k_pxy2) instead of slice partials
};

// per-filter arguments of the batched launches that carry RankArgs / MeasArgs (srukf_run_frames_batch; tables in device memory)
struct Pxy2Args { const double* DZp; const double* A; double* P0; double* P1; MeasArgs ms; };
struct SyrkArgs { const double* S; const double* Ut; double* G; FrameScalars* fs; const double* dxp; double* X; RankArgs ra; const double* xr1; };
struct GainArgs {
    double* Ut; const double* PxyR; const double* Si; const int* vis; const double* h; const double* z_seq; const int* m_seq; FrameScalars* fs; double* dxp;
    RankArgs ra; const double* Cm; double* S; const double* P1; const double* DZp; const double* sigR; const double* Z0;
};
struct ExpandArgs {
    const double* Sp; const double* D; const int* perm; const int* iperm; const double* gdiag; FrameScalars* fs; const double* X; double* S; double* A;
    double* sigR; double* Z; double* DZ;
};

// Head fold of the persistent factorisation launch (srukf_gmw_persist.hip): what its helper workgroups take over from k_syrk.
struct HeadArgs {
    const int2* tiles; int ntiles;                             // k_syrk's head tile list (row tile, col tile); the first ncrit: rows / columns < 128
    int ncrit;
    const double* dxp; double* X; const double* xr1; int ndx;  // dX slice partials -> X (srukf_gain_dx_job), ndx jobs
    int ngd;                                                   // dropped-diagonal jobs (srukf_rank_gdiag_job)
    int nhelp;                                                 // helper workgroups in the grid: they share one job queue (<= the CUs the pivot and the workers leave free)
    RankArgs ra;
};

// gdiag[a] = sum_{k<r} A[k][a]^2 - sum_{m<mu} Utp[m][a]^2 for 16 dropped positions a = r + 16 blk .. (16 columns x 16 row lanes,
// eight independent loads in flight per lane: a plain loop is one memory round trip per iteration); also feeds
// gamma = max diag(G) of the GMW bound (SLAM.cpp:2204-2211).  256 threads.
#define SRUKF_RANK_COLS 16
__device__ __forceinline__ double srukf_rank_colsq(const double* __restrict__ M, int ld, int rows, int a, int kl)
{
    // 20 loads in flight per lane, clamped onto the last row (in bounds, masked out of the sum): a column of 640 rows is two memory round trips
    // (with eight in flight and a scalar tail, 607 rows were five + six of them, and the null check was the longest chain of k_rank_expand)
    double v = 0.0;
    for (int k = kl; k < rows; k += 20 * 16) {
        double t[20];
#pragma unroll
        for (int u = 0; u < 20; u++) t[u] = M[(size_t)min(k + 16 * u, rows - 1) * ld + a];
#pragma unroll
        for (int u = 0; u < 20; u++) v += (k + 16 * u < rows) ? t[u] * t[u] : 0.0;
    }
    return v;
}
__device__ __forceinline__ void srukf_rank_gdiag_job(int n, int ld, int mu, const RankArgs ra, unsigned long long* gmax_bits, int blk)
{
    __shared__ double cs[16][17];
    const int a = ra.r + SRUKF_RANK_COLS * blk + (threadIdx.x & 15), kl = threadIdx.x >> 4;
    double v = 0.0;
    if (a < n) v = srukf_rank_colsq(ra.A, ld, ra.r, a, kl) - srukf_rank_colsq(ra.Utp, ld, mu, a, kl);
    cs[kl][threadIdx.x & 15] = v;
    __syncthreads();
    if (threadIdx.x < 16 && a < n) {
        double t = 0.0;
        for (int q = 0; q < 16; q++) t += cs[q][threadIdx.x];
        ra.gdiag[a] = t;
        if (t > 0.0) atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(t));
    }
}
