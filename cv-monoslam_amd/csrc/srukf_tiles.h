// srukf_tiles.h — FP64-MFMA tile helpers shared by the contraction kernels (srukf_factor.hip, srukf_augment.hip).
// All contractions are "TN": D[m][nn] = sum_k A[k][m] * B[k][nn] with both operands K-major (row k contiguous).
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A_op[i = l&15][k = l>>4]  -> A[k0 + (l>>4)][m0 + (l&15)]   (16 contiguous doubles per k)
//   B: lane l holds B_op[k = l>>4][j = l&15]  -> B[k0 + (l>>4)][n0 + (l&15)]
//   D: reg t of lane l is D[row = (l>>4) + 4t][col = l&15]
#pragma once
#include "srukf_device.h"

// one wave: 32x32 output tile at (m0, n0), K range [kb, ke) — (ke - kb) a multiple of 16 —, accumulate.
// Software-pipelined: the 16 fragment loads of the next group of four k-steps are in flight while the
// 16 MFMAs of the current group issue (hipcc otherwise waits for each group's loads before its MFMAs).
template <bool NEG>
__device__ __forceinline__ void tile32_tn(d4 (&acc)[2][2], const double* __restrict__ A, int lda,
                                          const double* __restrict__ B, int ldb, int m0, int n0, int kb, int ke, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    const double* pa = A + (size_t)(kb + lk) * lda + m0 + lr;
    const double* pb = B + (size_t)(kb + lk) * ldb + n0 + lr;
    const size_t sa = (size_t)4 * lda, sb = (size_t)4 * ldb;
    const int ng = (ke - kb) >> 4;
    if (ng <= 0) return;
    double ca0[4], ca1[4], cb0[4], cb1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { ca0[u] = pa[0]; ca1[u] = pa[16]; cb0[u] = pb[0]; cb1[u] = pb[16]; pa += sa; pb += sb; }
    for (int g = 0; g + 1 < ng; g++) {
        double na0[4], na1[4], nb0[4], nb1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { na0[u] = pa[0]; na1[u] = pa[16]; nb0[u] = pb[0]; nb1[u] = pb[16]; pa += sa; pb += sb; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double a0 = NEG ? -ca0[u] : ca0[u], a1 = NEG ? -ca1[u] : ca1[u];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, cb0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, cb1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, cb0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, cb1[u], acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { ca0[u] = na0[u]; ca1[u] = na1[u]; cb0[u] = nb0[u]; cb1[u] = nb1[u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const double a0 = NEG ? -ca0[u] : ca0[u], a1 = NEG ? -ca1[u] : ca1[u];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, cb0[u], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, cb1[u], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, cb0[u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, cb1[u], acc[1][1], 0, 0, 0);
    }
}

// The same contraction for a wave that has its SIMD to itself (the tile owners of the persistent GMW launch: one
// workgroup per CU): nobody else hides the operand latency, so FOUR rotating fragment buffers keep three groups of loads
// (48 requests) in flight while one group is multiplied.  sched_barrier pins the order (hipcc otherwise sinks the loads
// down to their first use).  Costs 128 VGPRs of fragments — an occupancy step in k_syrk / k_pxy, free here.
struct TileFrag { double a0[4], a1[4], b0[4], b1[4]; };
__device__ __forceinline__ void tile_frag_load(TileFrag& f, const double* __restrict__ pa, const double* __restrict__ pb, size_t sa, size_t sb)
{
#pragma unroll
    for (int u = 0; u < 4; u++) { f.a0[u] = pa[u * sa]; f.a1[u] = pa[u * sa + 16]; f.b0[u] = pb[u * sb]; f.b1[u] = pb[u * sb + 16]; }
    __builtin_amdgcn_sched_barrier(0);
}
template <bool NEG>
__device__ __forceinline__ void tile_frag_mma(d4 (&acc)[2][2], const TileFrag& f)
{
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const double a0 = NEG ? -f.a0[u] : f.a0[u], a1 = NEG ? -f.a1[u] : f.a1[u];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f.b0[u], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f.b1[u], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f.b0[u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f.b1[u], acc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <bool NEG>
__device__ __forceinline__ void tile32_tn_deep(d4 (&acc)[2][2], const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, int m0, int n0, int kb, int ke, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    const int ng = (ke - kb) >> 4;
    if (ng <= 0) return;
    const double* pa = A + (size_t)(kb + lk) * lda + m0 + lr;
    const double* pb = B + (size_t)(kb + lk) * ldb + n0 + lr;
    const size_t sa = (size_t)4 * lda, sb = (size_t)4 * ldb, ga = 4 * sa, gb = 4 * sb;
    const int last = ng - 1;                                   // loads past the last group are clamped onto it (in bounds, unused)
    TileFrag f0, f1, f2, f3;
    tile_frag_load(f0, pa, pb, sa, sb);
    { const int h = min(1, last); tile_frag_load(f1, pa + h * ga, pb + h * gb, sa, sb); }
    { const int h = min(2, last); tile_frag_load(f2, pa + h * ga, pb + h * gb, sa, sb); }
    int g = 0;
    for (; g + 4 <= ng; g += 4) {
        tile_frag_load(f3, pa + (g + 3) * ga, pb + (g + 3) * gb, sa, sb);
        tile_frag_mma<NEG>(acc, f0);
        { const int h = min(g + 4, last); tile_frag_load(f0, pa + h * ga, pb + h * gb, sa, sb); }
        tile_frag_mma<NEG>(acc, f1);
        { const int h = min(g + 5, last); tile_frag_load(f1, pa + h * ga, pb + h * gb, sa, sb); }
        tile_frag_mma<NEG>(acc, f2);
        { const int h = min(g + 6, last); tile_frag_load(f2, pa + h * ga, pb + h * gb, sa, sb); }
        tile_frag_mma<NEG>(acc, f3);
    }
    if (g < ng) tile_frag_mma<NEG>(acc, f0);
    if (g + 1 < ng) tile_frag_mma<NEG>(acc, f1);
    if (g + 2 < ng) tile_frag_mma<NEG>(acc, f2);
}

__device__ __forceinline__ void zero_acc(d4 (&acc)[2][2])
{
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4){0, 0, 0, 0};
}

// Split-K helper: the four waves of a workgroup share ONE 32x32 output tile, each contracting a
// quarter of the K range (in groups of 16); partial tiles are summed through LDS in fixed wave
// order (deterministic) and wave 0 owns the result.  Balances the triangular K ranges (S is upper
// triangular, so K grows with the tile's row index) and quadruples the waves in flight.
__device__ __forceinline__ void splitk_reduce(d4 (&acc)[2][2], double (*red)[64][17], int wv, int lane)
{
    if (wv > 0) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) red[wv - 1][lane][(a * 2 + b) * 4 + t] = acc[a][b][t];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int u = 0; u < 3; u++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] += red[u][lane][(a * 2 + b) * 4 + t];
    }
}

