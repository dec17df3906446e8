// srukf_factor.hip — FP64-MFMA contractions (cross covariance, S^T S - U U^T, trailing updates)
// and the Gill-Murray-Wright modified Cholesky that maintains the sqrt covariance.  gfx950 only.
//
// All contractions are "TN": D[m][nn] = sum_k A[k][m] * B[k][nn] with both operands K-major
// (row k contiguous), which is how S (rows = sigma directions), DZ, U^T and the GMW panels
// sit in HBM.  v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A_op[i = l&15][k = l>>4]  -> A[k0 + (l>>4)][m0 + (l&15)]   (16 contiguous doubles per k)
//   B: lane l holds B_op[k = l>>4][j = l&15]  -> B[k0 + (l>>4)][n0 + (l&15)]
//   D: reg t of lane l is D[row = (l>>4) + 4t][col = l&15]
#include "srukf_device.h"
#include "srukf_tiles.h"
#include "srukf_gmw_cols.h"
#include "srukf_meas.h"

// ------------------------------------------------------------------------------------------------
// k_pxy: landmark rows of all cross covariances in one contraction
//   Ut[c][r] = sum_{i <= r, i < n} DZ[i][c] * S[i][r]          c < 2N (mp padded), r < np
// = (S^T DZ)^T, i.e. the L rank-1 updates per landmark of calculateOneFeatureCrossCovariance
// (SLAM.cpp:2028-2037) for ALL landmarks at once; the wi*gamma scale and the robot rows are
// applied in k_gain.  S upper triangular => K range truncated at r0+32.
// grid = one workgroup per 32x32 tile (XCD-aware order from the tile table), 4-way split-K.
// ------------------------------------------------------------------------------------------------
// In the replay path the measurement statistics of the same frame ride along: the first workgroups of the
// grid run the MEAS_SLICES x gx partial-sum jobs, and the one that finishes last (device-scope counter)
// reduces the slices into h, Si, visible, PxyR — two launches fewer on the per-frame chain.
__global__ __launch_bounds__(256) void k_pxy(KDims d, const double* __restrict__ DZ, const double* __restrict__ S,
                                             double* __restrict__ Ut, const int2* __restrict__ tiles, int ntiles, KWeights w, MeasArgs ms)
{
    __shared__ double shm[MEAS_SM_DOUBLES];                    // >= 3*64*17: split-K scratch or statistics scratch
    const int nstat = ms.Z ? MEAS_SLICES * ms.gx : 0;         // statistics jobs first: they start with the launch, the tiles fill in behind
    if ((int)blockIdx.x < nstat) {
        const int job = blockIdx.x;
        meas_partial_job<true>(d, w, ms.X, ms.sigR, ms.Z, ms.part, job % ms.gx, job / ms.gx, shm);
        __shared__ int last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's device-scope stores have landed ...
        __syncthreads();                                       // ... and so have the whole workgroup's, before the count
        if (threadIdx.x == 0) {
            const int done = __hip_atomic_fetch_add(&ms.fs->stat_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (done == ms.gx * MEAS_SLICES - 1);
            if (last) __hip_atomic_store(&ms.fs->stat_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next frame
        }
        __syncthreads();
        if (!last) return;
        for (int k = threadIdx.x; k < d.N; k += 256) meas_final_one<true>(d, w, ms.X, ms.sigR, ms.Z, ms.part, ms.h, ms.Si, ms.vis, ms.PxyR, k);
        return;
    }
    double (*red)[64][17] = (double (*)[64][17])shm;
    const int2 tl = tiles[blockIdx.x - nstat];  // XCD-aware tile order (srukf_api.hip build_tile_tables)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (tl.x < 0) return;
    const int m0 = tl.x * 32;    // c
    const int n0 = tl.y * 32;    // r
    d4 acc[2][2];
    zero_acc(acc);
    int ke = n0 + 32; if (ke > d.np) ke = d.np;
    const int ng = ke >> 4;
    const int g0 = (ng * wv) >> 2, g1 = (ng * (wv + 1)) >> 2;
    tile32_tn<false>(acc, DZ, d.mp, S, d.np, m0, n0, g0 << 4, g1 << 4, lane);
    splitk_reduce(acc, red, wv, lane);
    if (wv != 0) return;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                Ut[(size_t)(m0 + 16 * a + lk + 4 * t) * d.np + n0 + 16 * b + lr] = acc[a][b][t];
}

// ------------------------------------------------------------------------------------------------
// k_syrk: G = S^T S - U U^T on the upper triangle (SLAM.cpp:2118-2120, 2149 batched over all
// measurement columns; rows [ub, ue) of Ut select the columns that are downdated — all of them
// in BATCHED mode, a single one in SEQUENTIAL mode).  Also accumulates gamma = max diag(G) and
// xi = max(0, max offdiag(G)) for the GMW bound beta^2 (SLAM.cpp:2204-2211).
// grid = one workgroup per upper-triangle 32x32 tile (XCD-aware order from the tile table), 4-way
// split-K over the concatenated K range [S rows 0..r0+32) ++ [Ut rows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_syrk(KDims d, const double* __restrict__ S, const double* __restrict__ Ut,
                                              int ub, int ue, double* __restrict__ G, FrameScalars* __restrict__ fs,
                                              const int2* __restrict__ tiles, int ntiles, const double* __restrict__ dxp, double* __restrict__ X)
{
    // workgroups past the tile list: the state update X += sum_k K_k (z_k - h_k) left over by k_gain
    if ((int)blockIdx.x >= ntiles) { srukf_gain_dx_job(d.n, d.np, dxp, X, blockIdx.x - ntiles); return; }
    __shared__ double red[3][64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int2 tl = tiles[blockIdx.x];  // upper-triangle tiles only, XCD-aware order
    if (tl.x < 0) return;
    const int m0 = tl.x * 32;    // row r
    const int n0 = tl.y * 32;    // col c
    d4 acc[2][2];
    zero_acc(acc);
    int ke = m0 + 32; if (ke > d.np) ke = d.np;          // S[k][r] = 0 for k > r
    const int u0 = ub & ~3, u1 = (ue + 3) & ~3;
    const bool full = (ub == u0) && (ue == u1) && (((u1 - u0) & 15) == 0);
    const int ngs = ke >> 4, ngu = full ? ((u1 - u0) >> 4) : 0, ng = ngs + ngu;
    const int g0 = (ng * wv) >> 2, g1 = (ng * (wv + 1)) >> 2;
    if (g0 < ngs) tile32_tn<false>(acc, S, d.np, S, d.np, m0, n0, g0 << 4, min(g1, ngs) << 4, lane);
    if (g1 > ngs) tile32_tn<true>(acc, Ut, d.np, Ut, d.np, m0, n0, u0 + ((max(g0, ngs) - ngs) << 4), u0 + ((g1 - ngs) << 4), lane);
    if (!full && wv == 0) {
        // partial K group (single-column downdate of SEQUENTIAL mode): mask rows outside [ub, ue)
        const int lr = lane & 15, lk = lane >> 4;
        for (int k = u0; k < u1; k += 4) {
            const int kk = k + lk;
            const bool in = (kk >= ub) && (kk < ue);
            const double a0 = in ? -Ut[(size_t)kk * d.np + m0 + lr] : 0.0, a1 = in ? -Ut[(size_t)kk * d.np + m0 + 16 + lr] : 0.0;
            const double b0 = in ? Ut[(size_t)kk * d.np + n0 + lr] : 0.0, b1 = in ? Ut[(size_t)kk * d.np + n0 + 16 + lr] : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    splitk_reduce(acc, red, wv, lane);
    if (wv != 0) return;
    const int lr = lane & 15, lk = lane >> 4;
    double gmax = 0.0, xmax = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int r = m0 + 16 * a + lk + 4 * t, c = n0 + 16 * b + lr;
                const double v = acc[a][b][t];
                G[(size_t)r * d.np + c] = v;
                if (r < d.n && c < d.n) {
                    if (r == c) gmax = fmax(gmax, v); else xmax = fmax(xmax, v);
                }
            }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if (lane == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}

// ------------------------------------------------------------------------------------------------
// GMW modified Cholesky (modifiedCholeskyDecomposition, SLAM.cpp:2197-2327), blocked, right-looking.
// Storage: W[j][i] = C[i][j] (the reference's column j below the diagonal is our row j right of
// the diagonal), so the final S[j][i] = sqrt(D_j) * L[i][j] = sqrt(D_j) * (W[j][i] / D_j) is a
// row scaling of W.  Fast path pivots with D_j = max(EPSILON, |C_jj|); the third candidate
// theta_j^2 / beta^2 (2279-2285) is evaluated afterwards from the row maxima theta_j collected
// here (k_gmw_check) — if it never wins, the result equals the reference algorithm's exactly;
// if it does, the caller reruns the frame on the column-by-column path (k_gmw_col_*).
//
// A 32x32 diagonal block is factored by srukf_gmw_cols.h (pivot chain on the vector ALU).  For the
// panel rows right of it, with M[kk][jj] = L[kk][jj] = W[kk][jj]/D_kk (kk < jj) and
// T = (I + M^T)^{-1} = L^{-1}, the forward substitution  w[jj] = g[jj] - sum_{kk<jj} L[kk][jj] w[kk]
// of a panel column is  w = T g,  so the panel "TRSM" is an MFMA product with Tt[kk][jj] = T[jj][kk].
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// 64-row panels: one launch per TWO 32-row sub-panels.  (One launch per 32-row panel was the first
// design: 37 launches, 2154 frames/s at N = 200.)  Every launch pays ~2.5 us of dispatch gap plus ~1.5 us of kernarg / first-load latency before any
// arithmetic starts, and the factorisation is one long dependent chain of launches; with 32-row
// panels that overhead was as large as the work.  Here the critical-path workgroup factors both
// 32x32 diagonal blocks of the NEXT 64-row panel inside one launch, and the trailing update runs
// with K = 64 (G is read and written half as often).
//
// Panel buffer: sub-panel 1 = rows j0..j0+31, sub-panel 2 = rows j0+32..j0+63.
//   Tt1, Tt2 : Tt[kk][jj] = T[jj][kk], T = L^{-1} of the sub-panel's diagonal block
//   E        : E[k][r] = L[k][32 + r] = W1d[k][r] / D_k, the multipliers that couple sub-panel 2 to the
//              pivots of sub-panel 1
//   W1 = T1 G1;   G2' = G2 - E^T W1;   W2 = T2 G2'        (three K = 32 MFMA stages per column slab,
//   register resident: a C-layout accumulator tile is a valid B operand of the next stage)
// ------------------------------------------------------------------------------------------------
struct GmwPanel64 { double Tt1[1024]; double Tt2[1024]; double E[1024]; double D[64]; double sq[64]; double rD[64]; };

// acc[a][b] (+)= sum_k A[k][16a + i] * B[k][16b + j],  k < 32:  A from a 32x32 K-major global array (row stride 32),
// B from a C-layout register tile (rows = k).  All 16 A fragments are requested before the first MFMA.
// TRI: A = Tt of a unit lower triangular T (A[k][j] = 0 for k > j): output rows 0..15 only see k < 16.
// DEV: operands written by another workgroup of the same launch (agent-scope loads).
template <bool NEG, bool TRI, bool DEV>
__device__ __forceinline__ void stage32_regB(d4 (&acc)[2][2], const double* __restrict__ A, const d4 (&B)[2][2], int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    double fa0[8], fa1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { if (!TRI || u < 4) fa0[u] = ld_g<DEV>(&A[(4 * u + lk) * 32 + lr]); fa1[u] = ld_g<DEV>(&A[(4 * u + lk) * 32 + 16 + lr]); }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int a2 = u >> 2, t = u & 3;                      // k = 16 a2 + 4 t + lk
        const double a1 = NEG ? -fa1[u] : fa1[u];
        if (!TRI || u < 4) {
            const double a0 = NEG ? -fa0[u] : fa0[u];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, B[a2][0][t], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, B[a2][1][t], acc[0][1], 0, 0, 0);
        }
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, B[a2][0][t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, B[a2][1][t], acc[1][1], 0, 0, 0);
    }
}

// first stage of a slab, W1 = T1 G1: A = Tt1 (global, row stride 32, triangular as above), B = 32 rows of G from
// row pointer Bp (row stride ldb), columns n0 .. n0+31.  All fragments are requested before the first MFMA.
template <bool DEV>
__device__ __forceinline__ void stage32_tri_globalB(d4 (&acc)[2][2], const double* __restrict__ A, const double* __restrict__ Bp, int ldb, int n0, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    double fa0[4], fa1[8], fb0[8], fb1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < 4) fa0[u] = ld_g<DEV>(&A[(4 * u + lk) * 32 + lr]);
        fa1[u] = ld_g<DEV>(&A[(4 * u + lk) * 32 + 16 + lr]);
        fb0[u] = ld_g<DEV>(&Bp[(size_t)(4 * u + lk) * ldb + n0 + lr]); fb1[u] = ld_g<DEV>(&Bp[(size_t)(4 * u + lk) * ldb + n0 + 16 + lr]);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < 4) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[u], fb0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[u], fb1[u], acc[0][1], 0, 0, 0);
        }
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[u], fb0[u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[u], fb1[u], acc[1][1], 0, 0, 0);
    }
}

#define G64_LS 80              // LDS row stride of the slabs (doubles): lanes l / l+16 land on opposite bank halves

// Critical-path workgroup of a 64-row step: owns the next panel's 64x64 diagonal region R = [base, base+64),
// tiles (0,0), (0,1), (1,1).
//   A   Tt1 | E | Tt2 are staged through LDS once (each wave needs all three); each wave then runs the
//       three-stage slab for 16 of R's 64 columns -> LDS (W and L = W/D)
//   B   quarters of tile (0,0) with K = 64 -> Xm;  factor 1: wave 0 pivots, wave 2 follows with T1'; waves 1 / 3
//       update tiles (0,1) / (1,1) with K = 64, park them in LDS, and write the panel's S rows for R's columns
//       from the LDS slab
//   C   W1d = T1' X01 (quarters), E' = W1d / D';  X11 -= E'^T W1d (quarters) -> Xm;  factor 2, in its own LDS
//       workspace: waves 1 / 3 first write E', the (0,1) tile of S and factor 1's outputs, then factor 2's outputs
// Global stores cost ~85 cycles of issue each on this path, so the pivot wave never stores to global memory.
// first != 0: there is no current panel (R is the first 64 rows): phase A and the K = 64 updates are skipped.
__device__ __forceinline__ void gmw_step64_block00(int n, int ld, int j0, int first, double eps, double* __restrict__ G,
                                                   const GmwPanel64* __restrict__ cur, GmwPanel64* __restrict__ nxt,
                                                   double* __restrict__ Dall, double* __restrict__ Sout,
                                                   double (*Lr)[G64_LS], double (*Wc)[G64_LS], double* facreg,
                                                   double* xreg, int tid)
{
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int base = j0 + 64;
    const int qa = wv >> 1, qb = wv & 1;
    // two factor workspaces: factor 1's rows stay readable while factor 2 runs, so its outputs can be written then
    const GmwColsLds ws = gmw_cols_carve(facreg), ws2 = gmw_cols_carve(facreg + GMW_FAC_DOUBLES);
    double (*X01)[32] = (double (*)[32])xreg;                  // tiles (0,1) / (1,1) after the K = 64 update
    double (*X11)[32] = (double (*)[32])(xreg + 1024);
    double* Tl = xreg + 2048;                                  // T1' [kk][33]
    double* PT = xreg;                                         // phase A only: Tt1 | E | Tt2, element (row, col) at row*32 + (col ^ 16*(row&1))
    if (tid < 32) { ws.Dv[tid] = 0.0; ws2.Dv[tid] = 0.0; }
    // prefetch: this wave's quarter of tile (0,0), its share of the panel matrices, its panel rows
    d4 g;
#pragma unroll
    for (int t = 0; t < 4; t++) g[t] = G[(size_t)(base + 16 * qa + lk + 4 * t) * ld + base + 16 * qb + lr];
    const int cw = base + 16 * wv;
    const int ro = (wv == 1) ? 0 : 32;                         // waves 1 / 3: tile rows base + ro .., columns base + 32 ..
    d4 acc[2][2];
    zero_acc(acc);
    if (!first) {
        d4 pt[3], X2[2];
        double fb[8], dr[4][4];
        pt[0] = *(const d4*)&cur->Tt1[4 * tid]; pt[1] = *(const d4*)&cur->E[4 * tid]; pt[2] = *(const d4*)&cur->Tt2[4 * tid];
#pragma unroll
        for (int u = 0; u < 8; u++) fb[u] = G[(size_t)(j0 + 4 * u + lk) * ld + cw + lr];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) X2[a][t] = G[(size_t)(j0 + 32 + 16 * a + lk + 4 * t) * ld + cw + lr];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) dr[q][t] = cur->rD[16 * q + lk + 4 * t];     // 1/D of rows 16 q + lk + 4 t
        STAMP(1);
        {
            const int row = tid >> 3, c4 = ((tid & 7) * 4) ^ (16 * (row & 1));
#pragma unroll
            for (int m = 0; m < 3; m++) *(d4*)&PT[m * 1024 + row * 32 + c4] = pt[m];
        }
        __syncthreads();
        if (wv & 1) {                                          // requested now, needed after phase B
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] = G[(size_t)(base + ro + 16 * a + lk + 4 * t) * ld + base + 32 + 16 * b + lr];
        }
        // ---- A: slab for columns cw .. cw+15; A operand (k = 4u + lk, column 16a + lr) of matrix m from LDS ----
        d4 W1[2] = { (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0} }, W2[2] = { (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0} };
        const int sw = 16 * (lk & 1);                          // (4u + lk) & 1 == lk & 1
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double* rowp = &PT[(4 * u + lk) * 32];
            if (u < 4) W1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[lr ^ sw], fb[u], W1[0], 0, 0, 0);   // Tt[k][j] = 0 for k > j
            W1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[(16 + lr) ^ sw], fb[u], W1[1], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double* rowp = &PT[1024 + (4 * u + lk) * 32];
            X2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-rowp[lr ^ sw], W1[u >> 2][u & 3], X2[0], 0, 0, 0);
            X2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-rowp[(16 + lr) ^ sw], W1[u >> 2][u & 3], X2[1], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double* rowp = &PT[2048 + (4 * u + lk) * 32];
            if (u < 4) W2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[lr ^ sw], X2[u >> 2][u & 3], W2[0], 0, 0, 0);
            W2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[(16 + lr) ^ sw], X2[u >> 2][u & 3], W2[1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int jj = 16 * a + lk + 4 * t, cc = 16 * wv + lr;
                Wc[jj][cc] = W1[a][t];       Lr[jj][cc] = W1[a][t] * dr[a][t];
                Wc[32 + jj][cc] = W2[a][t];  Lr[32 + jj][cc] = W2[a][t] * dr[2 + a][t];
            }
    } else {
        STAMP(1);
        if (wv & 1) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] = G[(size_t)(base + ro + 16 * a + lk + 4 * t) * ld + base + 32 + 16 * b + lr];
        }
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // ---- B: quarter (qa, qb) of tile (0,0), K = 64 ----
    if (!first) {
#pragma unroll
        for (int k = 0; k < 64; k += 4)
            g = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lr[k + lk][16 * qa + lr], Wc[k + lk][16 * qb + lr], g, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) ws.Xm[16 * qa + lk + 4 * t][16 * qb + lr] = g[t];
    __syncthreads();
    STAMP(4);
    // ---- factor 1 (+ panel S rows, tiles (0,1), (1,1)) ----
    if (wv == 0) gmw_cols_pivot_wave(ws, eps, lane);
    else if (wv == 2) { gmw_cols_t_wave<false>(ws, lane, nxt->Tt1, Tl); STAMPW(13); }
    else {
        if (!first) {
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const double a0 = -Lr[k + lk][ro + lr], a1 = -Lr[k + lk][ro + 16 + lr];
                const double b0 = Wc[k + lk][32 + lr], b1 = Wc[k + lk][48 + lr];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        double (*X)[32] = (wv == 1) ? X01 : X11;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) X[16 * a + lk + 4 * t][16 * b + lr] = acc[a][b][t];
        if (!first) {
            // S rows of the panel for R's columns, from the LDS slab: wave 1 rows j0..j0+31, wave 3 rows j0+32..j0+63
            // (rows / columns >= n carry zeros: G is zero there)
            const int c4 = lr * 4;
#pragma unroll 4
            for (int i = 0; i < 8; i++) {
                const int row = ro + 4 * i + lk;
                const double sq = cur->sq[row];
                d4 w = *(const d4*)&Wc[row][c4];
                w[0] *= sq; w[1] *= sq; w[2] *= sq; w[3] *= sq;
                *(d4*)&Sout[(size_t)(j0 + row) * ld + base + c4] = w;
            }
        }
        STAMPW(11 + wv);
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    // ---- C1: quarter (qa, qb) of W1d = T1' X01 and of E' = W1d / D' -> LDS (the slabs are dead by now) ----
    {
        d4 wq = (d4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; u++)
            wq = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(4 * u + lk) * 33 + 16 * qa + lr], X01[4 * u + lk][16 * qb + lr], wq, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int k = 16 * qa + lk + 4 * t, cc = 16 * qb + lr;
            Wc[k][cc] = wq[t];                                 // Wd
            Lr[k][cc] = wq[t] * gmw_pivot_rcp(ws.Dv[k]);       // Ld = E'
        }
    }
    __syncthreads();
    // ---- C2: quarter of X11 -= E'^T W1d -> Xm; reset the row flags ----
    {
        d4 x;
#pragma unroll
        for (int t = 0; t < 4; t++) x[t] = X11[16 * qa + lk + 4 * t][16 * qb + lr];
#pragma unroll
        for (int k = 0; k < 32; k += 4)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lr[k + lk][16 * qa + lr], Wc[k + lk][16 * qb + lr], x, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) ws2.Xm[16 * qa + lk + 4 * t][16 * qb + lr] = x[t];
    }
    __syncthreads();
    STAMP(7);
    // ---- factor 2 ----
    if (wv == 0) gmw_cols_pivot_wave(ws2, eps, lane);
    else if (wv == 2) gmw_cols_t_wave<true>(ws2, lane, nxt->Tt2, xreg);          // X01 / X11 are dead: staging buffer
    else {
        const int c4 = (lane & 7) * 4;
        if (wv == 1) {                                         // E' and T1' for the next launch
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 8 * i + (lane >> 3);
                *(d4*)&nxt->E[row * 32 + c4] = *(const d4*)&Lr[row][c4];
            }
            gmw_copy_t(Tl, nxt->Tt1, lane);
        } else {                                               // S rows base..base+31, columns base+32..base+63
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 8 * i + (lane >> 3);
                const double Dr = ws.Dv[row];
                const double sq = (base + row < n) ? sqrt(Dr) * gmw_pivot_rcp(Dr) : 0.0;
                d4 w = *(const d4*)&Wc[row][c4];
                w[0] *= sq; w[1] *= sq; w[2] *= sq; w[3] *= sq;
                *(d4*)&Sout[(size_t)(base + row) * ld + base + 32 + c4] = w;
            }
        }
        gmw_cols_out_wave(ws, wv == 1 ? 0 : 1, lane, n, ld, base, nxt->D, nxt->sq, nxt->rD, Dall, Sout);             // factor 1 (all rows published long ago)
        gmw_cols_out_wave(ws2, wv == 1 ? 0 : 1, lane, n, ld, base + 32, nxt->D + 32, nxt->sq + 32, nxt->rD + 32, Dall, Sout);
    }
    STAMP(8);
}

// One 64x64 block (by, bx), by <= bx and not (0,0), of the trailing square behind panel JJ = [j0, j0+64):
//   1. recomputes the panel rows it needs for its row slab and its column slab by the three MFMA stages
//      above (one 32-column half slab per wave, register resident) and keeps L = W/D and W in LDS;
//   2. updates its tile  G[r][c] -= sum_{kk<64} L[kk][r] W[kk][c];
//   3. first block row only: writes the final S rows j0..j0+63 for its column slab.
// The work is ordered by what it needs of the panel buffer, so that a worker of the persistent launch can start while
// the pivot workgroup is still factoring the panel's second half:
//   nothing   : the G tile and the panel's G rows (loads)
//   half one  : Tt1, E, 1/D[0..31]  ->  W1 = T1 G1,  G2' = G2 - E^T W1,  rows 0..31 of the slabs,  K = 0..31 of the update
//   half two  : Tt2, 1/D[32..63], sqrt(D)/D  ->  W2 = T2 G2',  rows 32..63,  K = 32..63,  tile store,  S rows
// wait_half() / wait_full(): called once each by ALL threads (they hold the workgroup barrier that separates this
// call's LDS writes from the previous call's reads); false abandons the tile.  stored(): called by all threads right
// after the tile store — the persistent kernel raises the tile's flag there, before the S rows nobody waits for.
// acc: this wave's 32x32 quadrant.  load_tile / store_tile: read it from / write it back to G (the persistent kernel
// keeps a tile in registers from its first update to its last).
// DEV: G tiles are exchanged with other workgroups of the SAME launch (agent-scope accesses).  The panel buffer is read
// with plain loads in both cases: it is written once per launch (agent-scope stores, before its flags) and read only
// after them, so no L2 can hold an older copy — and the ~170 workgroups that want the same 24 KB at the same moment
// are served by their XCD's L2 instead of one memory channel.
template <bool DEV, class WaitHalf, class WaitFull, class Stored>
__device__ __forceinline__ bool gmw_tile_update(int n, int ld, int j0, int by, int bx, double* __restrict__ G,
                                                const GmwPanel64* cur, double* __restrict__ Sout,
                                                double (*Lr)[G64_LS], double (*Wc)[G64_LS], int tid, d4 (&acc)[2][2],
                                                bool load_tile, bool store_tile, WaitHalf&& wait_half, WaitFull&& wait_full, Stored&& stored)
{
    const int lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int base = j0 + 64;
    const int R0 = base + 64 * by, C0 = base + 64 * bx;
    const bool diagblk = bx == by;
    const int m0 = R0 + 32 * (wv >> 1), c0 = C0 + 32 * (wv & 1);
    const bool live = (m0 < ld) && (c0 < ld) && (c0 + 32 > m0);
    if (live && load_tile) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[a][b][t] = ld_g<DEV>(&G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr]);
    }
    // half slab of this wave: waves 0,1 -> row slab halves, waves 2,3 -> column slab halves
    const int which = wv >> 1, half = wv & 1;
    const int n0 = (which ? C0 : R0) + 32 * half;
    const bool slab = n0 < ld && !(diagblk && which == 1);
    const bool write_s = slab && (by == 0) && (which == 1 || diagblk);
    const int ro = m0 - R0, co = c0 - C0;
    d4 X2[2][2], W1[2][2], W2[2][2];
    double fb0[8], fb1[8];
    if (slab) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) X2[a][b][t] = ld_g<DEV>(&G[(size_t)(j0 + 32 + 16 * a + lk + 4 * t) * ld + n0 + 16 * b + lr]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            fb0[u] = ld_g<DEV>(&G[(size_t)(j0 + 4 * u + lk) * ld + n0 + lr]); fb1[u] = ld_g<DEV>(&G[(size_t)(j0 + 4 * u + lk) * ld + n0 + 16 + lr]);
        }
    }
    // ---- first half of the panel ----
    if (!wait_half()) return false;
    if (slab) {
        double ta0[4], ta1[8], ea0[8], ea1[8], dr[2][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {                          // every fragment is requested before the first MFMA
            const int o = (4 * u + lk) * 32 + lr;
            if (u < 4) ta0[u] = cur->Tt1[o];
            ta1[u] = cur->Tt1[o + 16];
            ea0[u] = cur->E[o]; ea1[u] = cur->E[o + 16];
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) dr[q][t] = (which == 0) ? cur->rD[16 * q + lk + 4 * t] : 0.0;
        zero_acc(W1);
        // W1 = T1 G1 (T1 unit lower triangular: output rows 0..15 only see k < 16)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u < 4) {
                W1[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0[u], fb0[u], W1[0][0], 0, 0, 0);
                W1[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0[u], fb1[u], W1[0][1], 0, 0, 0);
            }
            W1[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1[u], fb0[u], W1[1][0], 0, 0, 0);
            W1[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1[u], fb1[u], W1[1][1], 0, 0, 0);
        }
        // G2' = G2 - E^T W1
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int a2 = u >> 2, t = u & 3;
            X2[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea0[u], W1[a2][0][t], X2[0][0], 0, 0, 0);
            X2[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea0[u], W1[a2][1][t], X2[0][1], 0, 0, 0);
            X2[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea1[u], W1[a2][0][t], X2[1][0], 0, 0, 0);
            X2[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea1[u], W1[a2][1][t], X2[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int jj = 16 * a + lk + 4 * t, cc = 32 * half + 16 * b + lr;
                    const double w1 = W1[a][b][t];
                    if (which == 0) { Lr[jj][cc] = w1 * dr[a][t]; if (diagblk) Wc[jj][cc] = w1; }
                    else Wc[jj][cc] = w1;
                }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const double a0 = -Lr[k + lk][ro + lr], a1 = -Lr[k + lk][ro + 16 + lr];
            const double b0 = Wc[k + lk][co + lr], b1 = Wc[k + lk][co + 16 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // ---- second half ----
    if (!wait_full()) return false;
    double sqr[4][4];
    if (slab) {
        double tb0[4], tb1[8], dr[2][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int o = (4 * u + lk) * 32 + lr;
            if (u < 4) tb0[u] = cur->Tt2[o];
            tb1[u] = cur->Tt2[o + 16];
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) dr[q][t] = (which == 0) ? cur->rD[32 + 16 * q + lk + 4 * t] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) sqr[q][t] = write_s ? cur->sq[16 * q + lk + 4 * t] : 0.0;
        zero_acc(W2);
        // W2 = T2 G2'
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int a2 = u >> 2, t = u & 3;
            if (u < 4) {
                W2[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb0[u], X2[a2][0][t], W2[0][0], 0, 0, 0);
                W2[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb0[u], X2[a2][1][t], W2[0][1], 0, 0, 0);
            }
            W2[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb1[u], X2[a2][0][t], W2[1][0], 0, 0, 0);
            W2[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb1[u], X2[a2][1][t], W2[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int jj = 32 + 16 * a + lk + 4 * t, cc = 32 * half + 16 * b + lr;
                    const double w2 = W2[a][b][t];
                    if (which == 0) { Lr[jj][cc] = w2 * dr[a][t]; if (diagblk) Wc[jj][cc] = w2; }
                    else Wc[jj][cc] = w2;
                }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 32; k < 64; k += 4) {
            const double a0 = -Lr[k + lk][ro + lr], a1 = -Lr[k + lk][ro + 16 + lr];
            const double b0 = Wc[k + lk][co + lr], b1 = Wc[k + lk][co + 16 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (store_tile) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        double* gp = &G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr];
                        if constexpr (DEV) st_dev(gp, acc[a][b][t]); else *gp = acc[a][b][t];
                    }
        }
    }
    stored();
    // final S rows j0 .. j0+63 of this wave's half slab (first block row only)
    if (write_s) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int jj = 16 * a + lk + 4 * t;
                    if (j0 + jj < n) Sout[(size_t)(j0 + jj) * ld + n0 + 16 * b + lr] = W1[a][b][t] * sqr[a][t];
                    if (j0 + 32 + jj < n) Sout[(size_t)(j0 + 32 + jj) * ld + n0 + 16 * b + lr] = W2[a][b][t] * sqr[2 + a][t];
                }
    }
    return true;
}

// k_gmw_step64: one launch per 64-row panel JJ = [j0, j0+64)  (j0 = -64, first = 1: only the first region is factored).
// Every 64x64 block of the trailing square (base = j0+64):
//   1. recomputes the panel rows it needs for its row slab and its column slab by the three MFMA stages
//      above (one 32-column half slab per wave, register resident) and keeps L = W/D and W in LDS;
//   2. updates its tile  G[r][c] -= sum_{kk<64} L[kk][r] W[kk][c];
//   3. first block row only: writes the final S rows j0..j0+63 for its column slab;
//   4. block (0,0) takes its own route (gmw_step64_block00).
// grid = (T, T), T = (ld - base)/64; blocks strictly below the diagonal exit.
__global__ __launch_bounds__(256) void k_gmw_step64(int n, int ld, int j0, int first, double* __restrict__ G,
                                                    const GmwPanel64* __restrict__ cur, double* __restrict__ Sout, GmwPanel64* __restrict__ nxt,
                                                    double* __restrict__ Dall, double eps)
{
    if (blockIdx.x < blockIdx.y) return;
    STAMP(0);
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];           // block (0,0): one column-factor workspace per sub-panel
    __shared__ double xreg[1024 + 1024 + 32 * 33];            // block (0,0): staged panel matrices, then X01 | X11 | T1'
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        gmw_step64_block00(n, ld, j0, first, eps, G, cur, nxt, Dall, Sout, Lr, Wc, facreg, xreg, tid);
        return;
    }
    d4 acc[2][2];
    zero_acc(acc);
    gmw_tile_update<false>(n, ld, j0, blockIdx.y, blockIdx.x, G, cur, Sout, Lr, Wc, tid, acc, true, true, [] { return true; }, [] { return true; }, [] {});
}

// ------------------------------------------------------------------------------------------------
// Persistent form: the whole factorisation in ONE launch (k_gmw_persist).
//
// With one launch per panel the chain of T = ld/64 launches pays, per panel, a dispatch gap (~1.5 us) and a cold
// start (kernarg + first loads from HBM, ~2.3 us: every launch begins with an invalidated L2) on top of the ~9.5 us
// of arithmetic of the critical-path workgroup.  Here that workgroup ("pivot", blockIdx 0) stays resident: it keeps
// the panel it has just factored in LDS (T1', E', T2', 1/D), applies it to the next 64x64 diagonal region itself and
// factors that, panel after panel.  Every other 64x64 tile (I, J) of the trailing matrix is OWNED by one worker
// workgroup, which holds it in registers from its first update to its last: a tile that went back to memory after
// every panel would need  store + acknowledge + flag + poll + load  (~2 us, scripts/mb/mb_xwg.hip) plus the update
// itself (~4 us) per panel — longer than the pivot's period, and the tile chains, not the pivot, would set the pace
// (measured with a task-queue version: 13 us per panel against 9.6 us).  Hand-off through global memory:
//   pivot  -> workers : panel buffer pans[k] (agent-scope stores), then panel_ready = k + 1
//   owner  -> anybody : the tile, once, when it has received its last update: G tile (agent-scope stores), then
//                       ver[I][J] = number of panel updates it carries (I for I < J; I - 1 on the diagonal, where the
//                       pivot applies the last panel itself)
//   step k of tile (I, J) needs panel k and the finished row-panel tiles (k, I), (k, J);
//   the pivot, before panel p, needs tiles (p-1, p) and (p, p).
// Every workgroup of the grid must be resident (1 + workers <= CUs; the launcher sees to that, and a filter that
// shares the GPU with others uses the one-launch-per-panel path).  Every wait is bounded: on expiry the launch is
// abandoned and the frame flagged, and the caller repeats it on the other path.
// ------------------------------------------------------------------------------------------------
// Who polls and who raises flags: WAVE 0 as a whole, under wave-uniform (scalar) conditions, never "if (tid == 0)".
// A divergent single-thread branch just before the back edge of the task loop and another one right after its head
// get merged by the structurizer into a lane-divergent loop around the workgroup barrier (wave 0 then executes
// s_barrier more often than the other waves: hang).  Uniform branches leave EXEC alone; 64 lanes loading or storing
// the same flag word are one memory request.
#define GMW_XWG_LIMIT (1 << 16)                 // ~50 ms; a legitimate wait is over in microseconds
__device__ __forceinline__ unsigned long long gmw_uniform64(unsigned long long v)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ bool gmw_wait_ge(const unsigned long long* f, unsigned long long want, const int* abort_flag)
{
    for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
        if (gmw_uniform64(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= want) return true;
        if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__device__ __forceinline__ void gmw_set_flag(unsigned long long* f, unsigned long long v)
{
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all copies of a panel flag: lane c < COPIES stores copy c (one instruction; called by a whole wave)
__device__ __forceinline__ void gmw_set_panel_flag(unsigned long long* f, unsigned long long v, int lane)
{
    if (lane < GMW_FLAG_COPIES) __hip_atomic_store(&f[lane * GMW_FLAG_STRIDE], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all agent-scope stores of this workgroup have landed -> wave 0 raises the flag (wv0: wave-uniform "this is wave 0")
__device__ __forceinline__ void gmw_publish(unsigned long long* f, unsigned long long v, bool wv0)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wv0) gmw_set_flag(f, v);
}

// LDS of the pivot workgroup that survives from one panel to the next
struct GmwPivotKeep {
    double* T1;      // [kk][33]   T1'[jj][kk] of the panel factored last (xreg + 2048, written by the T wave of factor 1)
    double* T2;      // [kk][33]   T2'                                     (xreg, staging buffer of the T wave of factor 2)
    double* Ep;      // [k][32], column ^ 16*(k&1): E' = W1d / D'
    double* rD;      // [64] 1/D
    double* sq;      // [64] sqrt(D)/D
};

// One wave copies the 64x64 tile at (row0, col0) of G into an LDS array (row stride G64_LS): each request is one
// 512-byte row, all 64 in flight (one memory round trip).  Agent-scope loads: the tile was written by a worker of this launch.
__device__ __forceinline__ void gmw_stage_tile(double (*dst)[G64_LS], const double* __restrict__ G, int ld, int row0, int col0, int lane)
{
    const double* src = G + (size_t)row0 * ld + col0 + lane;
    double v[64];
#pragma unroll
    for (int r = 0; r < 64; r++) v[r] = ld_dev(src + (size_t)r * ld);
#pragma unroll
    for (int r = 0; r < 64; r++) dst[r][lane] = v[r];
}

// Pivot workgroup: panels p = 0 .. T-1.  Same phases as gmw_step64_block00 (A slab, B tile (0,0), factor 1, C, factor 2);
// what differs is where the operands come from: the previous panel from LDS, G tiles through agent-scope loads after
// their version flags, and the panel buffer is published for the workers.
__device__ __forceinline__ void gmw_pivot_persist(int n, int ld, int T, double eps, double* __restrict__ G, GmwPanel64* __restrict__ pans,
                                                  double* __restrict__ Dall, double* __restrict__ Sout, GmwSync* sy, unsigned long long ebase,
                                                  double (*Lr)[G64_LS], double (*Wc)[G64_LS], double* facreg, double* xreg, double* keepreg,
                                                  int* okp, int* halfcnt, int* stageok, int tid)
{
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int qa = wv >> 1, qb = wv & 1;
    const GmwColsLds ws = gmw_cols_carve(facreg), ws2 = gmw_cols_carve(facreg + GMW_FAC_DOUBLES);
    double (*X01)[32] = (double (*)[32])xreg;
    double (*X11)[32] = (double (*)[32])(xreg + 1024);
    GmwPivotKeep kp;
    kp.T1 = xreg + 2048; kp.T2 = xreg; kp.Ep = keepreg; kp.rD = keepreg + 1024; kp.sq = keepreg + 1088;
    unsigned long long* ver = gmw_sync_ver(sy);
    const int ro = (wv == 1) ? 0 : 32;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);        // wave-uniform copy: scalar branches around everything that polls or raises flags
    const bool wv0 = wvu == 0, wv1 = wvu == 1, wv3 = wvu == 3;
    if (wv0) *okp = 1;
    if (wv3) gmw_stage_tile(Wc, G, ld, 0, 0, lane);            // region R_0 as k_syrk left it
    __syncthreads();
    for (int p = 0; p < T; p++) {
        const int j0 = 64 * (p - 1), base = 64 * p;
        const bool first = (p == 0);
        GmwPanel64* nxt = pans + p;
        if (wv0) GMW_TS(sy, p, 0);
        // operands staged in LDS by waves 1 / 3 during factor 2 of the previous panel: Lr = tile (p-1, p) (rows of the
        // current panel, columns of R), Wc = tile (p, p) (R itself)
        if (wv0) { ws.Dv[lane & 31] = 0.0; ws2.Dv[lane & 31] = 0.0; *halfcnt = 0; stageok[lane & 1] = 0; }
        d4 g;
#pragma unroll
        for (int t = 0; t < 4; t++) g[t] = Wc[16 * qa + lk + 4 * t][16 * qb + lr];
        d4 acc[2][2];
        zero_acc(acc);
        if (wv & 1) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] = Wc[ro + 16 * a + lk + 4 * t][32 + 16 * b + lr];
        }
        d4 X2[2];
        double fb[8], dr[4][4];
        if (!first) {
#pragma unroll
            for (int u = 0; u < 8; u++) fb[u] = Lr[4 * u + lk][16 * wv + lr];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int t = 0; t < 4; t++) X2[a][t] = Lr[32 + 16 * a + lk + 4 * t][16 * wv + lr];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) dr[q][t] = kp.rD[16 * q + lk + 4 * t];
        }
        __syncthreads();                                       // staged tiles are in registers: Lr / Wc may be rewritten
        if (!first) {
            // ---- A: slab for columns cw .. cw+15, panel matrices from LDS ----
            d4 W1[2] = { (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0} }, W2[2] = { (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0} };
            const int sw = 16 * (lk & 1);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double* rowp = &kp.T1[(4 * u + lk) * 33];
                if (u < 4) W1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[lr], fb[u], W1[0], 0, 0, 0);
                W1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[16 + lr], fb[u], W1[1], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double* rowp = &kp.Ep[(4 * u + lk) * 32];
                X2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-rowp[lr ^ sw], W1[u >> 2][u & 3], X2[0], 0, 0, 0);
                X2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-rowp[(16 + lr) ^ sw], W1[u >> 2][u & 3], X2[1], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double* rowp = &kp.T2[(4 * u + lk) * 33];
                if (u < 4) W2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[lr], X2[u >> 2][u & 3], W2[0], 0, 0, 0);
                W2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rowp[16 + lr], X2[u >> 2][u & 3], W2[1], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int jj = 16 * a + lk + 4 * t, cc = 16 * wv + lr;
                    Wc[jj][cc] = W1[a][t];       Lr[jj][cc] = W1[a][t] * dr[a][t];
                    Wc[32 + jj][cc] = W2[a][t];  Lr[32 + jj][cc] = W2[a][t] * dr[2 + a][t];
                }
        }
        if (wv0) GMW_TS(sy, p, 1);
        __syncthreads();
        // ---- B: quarter (qa, qb) of tile (0,0), K = 64 ----
        if (!first) {
#pragma unroll
            for (int k = 0; k < 64; k += 4)
                g = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lr[k + lk][16 * qa + lr], Wc[k + lk][16 * qb + lr], g, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) ws.Xm[16 * qa + lk + 4 * t][16 * qb + lr] = g[t];
        __syncthreads();
        if (wv0) GMW_TS(sy, p, 2);
        // ---- factor 1 (+ panel S rows, tiles (0,1), (1,1)) ----
        if (wv0) gmw_cols_pivot_wave(ws, eps, lane);
        else if (wvu == 2) gmw_cols_t_wave<0>(ws, lane, nullptr, kp.T1);
        else {
            if (!first) {
#pragma unroll
                for (int k = 0; k < 64; k += 4) {
                    const double a0 = -Lr[k + lk][ro + lr], a1 = -Lr[k + lk][ro + 16 + lr];
                    const double b0 = Wc[k + lk][32 + lr], b1 = Wc[k + lk][48 + lr];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            double (*X)[32] = (wv == 1) ? X01 : X11;           // X01 overwrites T2 of the previous panel: dead since phase A
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) X[16 * a + lk + 4 * t][16 * b + lr] = acc[a][b][t];
            if (!first) {
                const int c4 = lr * 4;
#pragma unroll 4
                for (int i = 0; i < 8; i++) {
                    const int row = ro + 4 * i + lk;
                    const double sq = kp.sq[row];
                    d4 w = *(const d4*)&Wc[row][c4];
                    w[0] *= sq; w[1] *= sq; w[2] *= sq; w[3] *= sq;
                    *(d4*)&Sout[(size_t)(j0 + row) * ld + base + c4] = w;
                }
            }
        }
        __syncthreads();
        if (wv0) GMW_TS(sy, p, 3);
        // ---- C1: quarter (qa, qb) of W1d = T1' X01 and of E' = W1d / D' ----
        {
            d4 wq = (d4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; u++)
                wq = __builtin_amdgcn_mfma_f64_16x16x4f64(kp.T1[(4 * u + lk) * 33 + 16 * qa + lr], X01[4 * u + lk][16 * qb + lr], wq, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 16 * qa + lk + 4 * t, cc = 16 * qb + lr;
                const double e = wq[t] * gmw_pivot_rcp(ws.Dv[k]);
                Wc[k][cc] = wq[t];
                Lr[k][cc] = e;
                kp.Ep[k * 32 + (cc ^ (16 * (k & 1)))] = e;
            }
        }
        __syncthreads();
        // ---- C2: quarter of X11 -= E'^T W1d -> Xm of factor 2 ----
        {
            d4 x;
#pragma unroll
            for (int t = 0; t < 4; t++) x[t] = X11[16 * qa + lk + 4 * t][16 * qb + lr];
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lr[k + lk][16 * qa + lr], Wc[k + lk][16 * qb + lr], x, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) ws2.Xm[16 * qa + lk + 4 * t][16 * qb + lr] = x[t];
        }
        __syncthreads();
        if (wv0) GMW_TS(sy, p, 4);
        // ---- factor 2.  Wave 0 pivots, wave 2 follows with T2'.  Waves 1 / 3, in the order of who is waiting for what:
        //   1. the first half of the panel buffer (wave 1: E' and the pivots of sub-panel 1; wave 3: T1'); whoever sees its
        //      stores acknowledged last raises half_ready — the workers run their first two MFMA stages while factor 2 is busy;
        //   2. the (0,1) tile of S and the S rows / pivots of both factors as the pivot wave produces them.
        // Then EVERY wave fetches 32 rows of the two tiles the NEXT panel needs (tile (p, p+1) -> Lr by waves 1 / 0,
        // tile (p+1, p+1) -> Wc by waves 3 / 2; finished by their owners with the updates of panels 0 .. p-1).  Lr / Wc are
        // free: wave 1 is the only reader of Lr's E' corner, wave 3 of Wc's W1d corner, both in rows 0..31 which they
        // overwrite themselves.  Waiting for those loads also waits for the wave's earlier stores, so after the closing
        // barrier the panel buffer is complete in memory and panel_ready can be raised at once.
        if (wv0) { gmw_cols_pivot_wave(ws2, eps, lane); GMW_TS(sy, p + 64, 0); }
        else if (wvu == 2) {
            // this wave has slack while it follows the pivots: it asks early whether the owners of the two tiles of the NEXT
            // panel have finished them (normally yes) and tells the others through LDS — saves every wave the ~1 us poll
            // round trip at the end of the iteration
            unsigned long long fa = 0, fb = 0;
            if (p >= 1 && p + 1 < T) {
                fa = __hip_atomic_load(&ver[(size_t)p * T + p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fb = __hip_atomic_load(&ver[(size_t)(p + 1) * T + p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gmw_cols_t_wave<2>(ws2, lane, nxt->Tt2, kp.T2, [&] {
                if (p == 0 || gmw_uniform64(fa) >= ebase + p) stageok[0] = 1;
                if (p == 0 || gmw_uniform64(fb) >= ebase + p) stageok[1] = 1;
            });
        } else {
            const int c4 = (lane & 7) * 4;
            if (wv1) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = 8 * i + (lane >> 3);
                    st_d4<true>(&nxt->E[row * 32 + c4], *(const d4*)&Lr[row][c4]);
                }
                if (lane < 32) {
                    const double D = ws.Dv[lane], rc = gmw_pivot_rcp(D), sq = sqrt(D) * rc;
                    st_dev(&nxt->D[lane], D); st_dev(&nxt->sq[lane], sq); st_dev(&nxt->rD[lane], rc);
                }
            } else gmw_copy_t<true>(kp.T1, nxt->Tt1, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {
                int prev = 0;
                if (lane == 0) prev = __hip_atomic_fetch_add(halfcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(prev) == 1) gmw_set_panel_flag(sy->half_ready, ebase + p + 1, lane);
            }
            if (!wv1) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = 8 * i + (lane >> 3);
                    const double Dr = ws.Dv[row];
                    const double sq = (base + row < n) ? sqrt(Dr) * gmw_pivot_rcp(Dr) : 0.0;
                    d4 w = *(const d4*)&Wc[row][c4];
                    w[0] *= sq; w[1] *= sq; w[2] *= sq; w[3] *= sq;
                    *(d4*)&Sout[(size_t)(base + row) * ld + base + 32 + c4] = w;
                }
            }
            gmw_cols_out_wave<true>(ws, wv1 ? 0 : 1, lane, n, ld, base, nxt->D, nxt->sq, nxt->rD, Dall, Sout, kp.sq, kp.rD);
            gmw_cols_out_wave<true>(ws2, wv1 ? 0 : 1, lane, n, ld, base + 32, nxt->D + 32, nxt->sq + 32, nxt->rD + 32, Dall, Sout, kp.sq + 32, kp.rD + 32);
            if (wv1) GMW_TS(sy, p + 64, 1); else GMW_TS(sy, p + 64, 2);
        }
        if (p + 1 < T) {
            const bool tileA = wvu < 2;                        // waves 0, 1: tile (p, p+1) -> Lr;  waves 2, 3: tile (p+1, p+1) -> Wc
            const int tr = tileA ? p : p + 1, r0 = (wvu & 1) ? 0 : 32;
            if (wv1) GMW_TS(sy, p, 5);
            const bool ready = p == 0 || __builtin_amdgcn_readfirstlane(stageok[tileA ? 0 : 1]) != 0 || gmw_wait_ge(&ver[(size_t)tr * T + p + 1], ebase + p, &sy->abort);
            if (wv1) GMW_TS(sy, p, 6);
            if (!ready) *okp = 0;
            else {
                double (*dst)[G64_LS] = tileA ? Lr : Wc;
                const double* src = G + (size_t)(64 * tr + r0) * ld + 64 * (p + 1) + lane;
                double v[32];
#pragma unroll
                for (int r = 0; r < 32; r++) v[r] = ld_dev(src + (size_t)r * ld);
#pragma unroll
                for (int r = 0; r < 32; r++) dst[r0 + r][lane] = v[r];
            }
        }
        __syncthreads();                                       // closes the iteration: staged tiles visible, LDS arrays reusable
        if (wv3 && p + 1 < T) { gmw_set_panel_flag(sy->panel_ready, ebase + p + 1, lane); GMW_TS(sy, p + 64, 3); }
        if (wv0) GMW_TS(sy, p, 7);
        if (!*okp) { if (wv0) __hip_atomic_store(&sy->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
    // the last panel buffers are never read by a worker (steps T-2 and T-1 have no trailing tiles)
}

// Worker side of one update step of an owned tile; returns false when a wait expired.
struct GmwOwned { int I, J, nsteps; bool computed; };
struct KDimsLite { int n, ld; };
__device__ __forceinline__ bool gmw_owner_step(int n, int ld, int T, int k, const GmwOwned& tl, d4 (&acc)[2][2], double* __restrict__ G,
                                               GmwPanel64* pans, double* __restrict__ Sout, GmwSync* sy, unsigned long long ebase,
                                               double (*Lr)[G64_LS], double (*Wc)[G64_LS], int* okp, bool wv0, int tid)
{
    unsigned long long* ver = gmw_sync_ver(sy);
    // the two row-panel tiles (k, I), (k, J) are finished (k updates each) — both flags in one round trip
    if (wv0) {
        bool good = true;
        if (k > 0) {
            const unsigned long long want = ebase + k;
            unsigned long long a = 0, b = 0;
            for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
                a = gmw_uniform64(__hip_atomic_load(&ver[(size_t)k * T + tl.I], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b = gmw_uniform64(__hip_atomic_load(&ver[(size_t)k * T + tl.J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (a >= want && b >= want) break;
                if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                __builtin_amdgcn_s_sleep(1);
            }
            good = a >= want && b >= want;
        }
        *okp = good;
    }
    __syncthreads();
    if (!*okp) return false;
    const bool last = k == tl.nsteps - 1;
    return gmw_tile_update<true>(n, ld, 64 * k, tl.I - k - 1, tl.J - k - 1, G, pans + k, Sout, Lr, Wc, tid, acc, k == 0 && !tl.computed, last,
        [&] {
            if (wv0) *okp = gmw_wait_ge(&sy->half_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], ebase + k + 1, &sy->abort);
            __syncthreads();
            return *okp != 0;
        },
        [&] {
            if (wv0) *okp = gmw_wait_ge(&sy->panel_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], ebase + k + 1, &sy->abort);
            __syncthreads();
            return *okp != 0;
        },
        [&] { if (last) gmw_publish(&ver[(size_t)tl.I * T + tl.J], ebase + tl.nsteps, wv0); });
}

// k_gmw_persist: grid = 1 + workers; worker w owns tiles[w - 1] and tiles[w - 1 + workers] (if any).
#define GMW_OWNED_MAX 2
struct GmwTile { short I, J, nsteps, pad; };
// S0 != null: the tiles of block rows I >= GMW_HEAD_ROWS are not read from G but COMPUTED by their owners,
//   G[r][c] = sum_k S0[k][r] S0[k][c] - sum_{u0 <= m < u1} Ut0[m][r] Ut0[m][c]      (what k_syrk does, SLAM.cpp:2118-2120, 2149),
// while the pivot is already factoring the first panels (k_syrk then only runs for block rows 0 and 1: a few
// microseconds instead of ~37).  The owner of tile (I, J) needs 12.5 + 1.7 I us for it and has ~12 I us until somebody waits
// for the tile.  S0 must not be the buffer the factor is written to (Sout).
#define GMW_HEAD_ROWS 2
__device__ __forceinline__ void gmw_owner_syrk(const KDimsLite d, const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1,
                                               int I, int J, d4 (&acc)[2][2], FrameScalars* __restrict__ fs, int tid)
{
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = 64 * I + 32 * (wv >> 1), c0 = 64 * J + 32 * (wv & 1);
    zero_acc(acc);
    if (m0 >= d.ld || c0 >= d.ld || c0 + 32 <= m0) return;
    tile32_tn<false>(acc, S0, d.ld, S0, d.ld, m0, c0, 0, min(m0 + 32, d.ld), lane);      // S0[k][r] = 0 for k > r
    tile32_tn<true>(acc, Ut0, d.ld, Ut0, d.ld, m0, c0, u0, u1, lane);
    double gmax = 0.0, xmax = 0.0;                             // gamma / xi of the GMW bound, as in k_syrk
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int r = m0 + 16 * a + lk + 4 * t, c = c0 + 16 * b + lr;
                if (r < d.n && c < d.n && c >= r) { if (r == c) gmax = fmax(gmax, acc[a][b][t]); else xmax = fmax(xmax, acc[a][b][t]); }
            }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if (lane == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}

__global__ __launch_bounds__(256) void k_gmw_persist(int n, int ld, int T, double* __restrict__ G, GmwPanel64* __restrict__ pans,
                                                     double* __restrict__ Sout, double* __restrict__ Dall, double eps,
                                                     GmwSync* __restrict__ sy, const GmwTile* __restrict__ tiles, int ntiles,
                                                     FrameScalars* __restrict__ fs,
                                                     const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1)
{
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];
    __shared__ double xreg[1024 + 1024 + 32 * 33];
    __shared__ double keepreg[1024 + 64 + 64];
    __shared__ int ok, halfcnt, stageok[2];
    const int tid = threadIdx.x;
    const unsigned long long ebase = sy->epoch << GMW_EPOCH_SHIFT;      // written by the previous launch's last workgroup
    if (blockIdx.x == 0) {
        gmw_pivot_persist(n, ld, T, eps, G, pans, Dall, Sout, sy, ebase, Lr, Wc, facreg, xreg, keepreg, &ok, &halfcnt, stageok, tid);
    } else {
        const bool wv0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
        const int workers = gridDim.x - 1, w = blockIdx.x - 1;
        GmwOwned ta = { 0, 0, 0, false }, tb = { 0, 0, 0, false };
        if (w < ntiles) { const GmwTile t = tiles[w]; ta.I = t.I; ta.J = t.J; ta.nsteps = t.nsteps; }
        if (w + workers < ntiles) { const GmwTile t = tiles[w + workers]; tb.I = t.I; tb.J = t.J; tb.nsteps = t.nsteps; }
        d4 acca[2][2], accb[2][2];
        zero_acc(acca); zero_acc(accb);
        if (S0) {
            const KDimsLite dl = { n, ld };
            if (ta.nsteps > 0 && ta.I >= GMW_HEAD_ROWS) { gmw_owner_syrk(dl, S0, Ut0, u0, u1, ta.I, ta.J, acca, fs, tid); ta.computed = true; }
            if (tb.nsteps > 0 && tb.I >= GMW_HEAD_ROWS) { gmw_owner_syrk(dl, S0, Ut0, u0, u1, tb.I, tb.J, accb, fs, tid); tb.computed = true; }
        }
        const int kmax = max(ta.nsteps, tb.nsteps);
        bool good = true;
        for (int k = 0; k < kmax && good; k++) {
            if (k < ta.nsteps) good = gmw_owner_step(n, ld, T, k, ta, acca, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid);
            if (good && k < tb.nsteps) good = gmw_owner_step(n, ld, T, k, tb, accb, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid);
        }
        if (!good && wv0) __hip_atomic_store(&sy->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last workgroup out re-arms the block for the next launch and reports an abandoned run
    GMW_DBG(sy, 6, 7777);
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, 0); atomicAdd(&fs->gmw_aborts, 1); }
            __hip_atomic_store(&sy->abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->epoch, (ebase >> GMW_EPOCH_SHIFT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// k_gmw_check: was the reference's third pivot candidate theta_j^2/beta^2 ever the largest?
// theta_j = max_{i>j} |C[i][j]| = sqrt(D_j) * max_{i>j} |S[j][i]|  (S[j][i] = C[i][j]/sqrt(D_j)),
// beta^2 = max(gamma, xi/nu, 1e-15), nu = max(1, sqrt(n^2-1))   (SLAM.cpp:2204-2211, 2264-2285).
// One workgroup per pivot row.
// Scopy != null: S is a scratch buffer the factorisation wrote its rows to (its input was still being read from the
// filter's own S, see k_gmw_persist): row j, columns j.., is copied into the filter's S on the way.
__global__ __launch_bounds__(256) void k_gmw_check(int n, int ld, const double* __restrict__ D, const double* __restrict__ S,
                                                   FrameScalars* __restrict__ fs, const double* __restrict__ X, int do_traj,
                                                   double* __restrict__ Scopy)
{
    __shared__ double red[4];
    const int j = blockIdx.x;
    if (j == n) {
        // extra block: per-frame record (x, y, z, theta, P00, P01, P10, P11) = RobotPath.txt columns
        // (SLAM.cpp:3549-3556), P = S^T S robot x/y block (2404); advances the staged frame counter
        __shared__ double r3[16 * 3];
        double v[3] = { 0, 0, 0 };
        for (int k = threadIdx.x; k < n; k += 256) {
            const double a = S[(size_t)k * ld + (n - 4)], b = S[(size_t)k * ld + (n - 3)];
            v[0] += a * a; v[1] += a * b; v[2] += b * b;
        }
        block_sum<3>(v, r3);
        if (threadIdx.x == 0 && do_traj) {
            double* traj = fs->traj_base;
            if (traj) {
                double* t = traj + (size_t)8 * fs->frame;
                for (int e = 0; e < 4; e++) t[e] = X[n - 4 + e];
                t[4] = v[0]; t[5] = v[1]; t[6] = v[1]; t[7] = v[2];
            }
            fs->frame += 1;
        }
        return;
    }
    double mx = 0.0;
    if (Scopy) {
        for (int i = j + threadIdx.x; i < n; i += 256) {
            const double v = S[(size_t)j * ld + i];
            Scopy[(size_t)j * ld + i] = v;
            if (i > j) mx = fmax(mx, fabs(v));
        }
    } else {
        for (int i = j + 1 + threadIdx.x; i < n; i += 256) mx = fmax(mx, fabs(S[(size_t)j * ld + i]));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double gamma = __longlong_as_double((long long)fs->gmax_bits);
        const double xi = __longlong_as_double((long long)fs->ximax_bits);
        const double nu = fmax(1.0, sqrt((double)n * n - 1.0));
        const double beta2 = fmax(fmax(gamma, xi / nu), 1e-15);
        const double dj = D[j];
        const double th = mx * sqrt(dj);
        if (th * th / beta2 > dj) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, j); }
    }
}

// ---- column-by-column path: evaluates every pivot exactly as the reference does -----------------
// k_gmw_col_a(j): W[j][i] = G[j][i] - sum_{k<j} (W[k][j]/D[k]) * W[k][i]  for i >= j; theta_j.
// W is stored in Wf (full n x n, upper).  grid over i.
__global__ __launch_bounds__(256) void k_gmw_col_a(int ld, int j, const double* __restrict__ G, double* __restrict__ Wf,
                                                   const double* __restrict__ D, unsigned long long* __restrict__ theta_bits)
{
    const int i = j + blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < ld) {
        double acc = 0.0;
        for (int k = 0; k < j; k++) acc += (Wf[(size_t)k * ld + j] / D[k]) * Wf[(size_t)k * ld + i];
        v = G[(size_t)j * ld + i] - acc;
        Wf[(size_t)j * ld + i] = v;
    }
    double mx = (i < ld && i > j) ? fabs(v) : 0.0;
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(&theta_bits[j], (unsigned long long)__double_as_longlong(mx));
}
// k_gmw_col_b(j): D_j = max(EPSILON, |C_jj|, theta_j^2/beta^2); S row j.
__global__ __launch_bounds__(256) void k_gmw_col_b(int n, int ld, int j, double eps, const double* __restrict__ Wf, double* __restrict__ D,
                                                   const unsigned long long* __restrict__ theta_bits, FrameScalars* __restrict__ fs,
                                                   double* __restrict__ Sout)
{
    const double gamma = __longlong_as_double((long long)fs->gmax_bits);
    const double xi = __longlong_as_double((long long)fs->ximax_bits);
    const double nu = fmax(1.0, sqrt((double)n * n - 1.0));
    const double beta2 = fmax(fmax(gamma, xi / nu), 1e-15);
    const double th = __longlong_as_double((long long)theta_bits[j]);
    const double cjj = fabs(Wf[(size_t)j * ld + j]);
    const double t2 = th * th / beta2;
    const double dj = fmax(fmax(eps, cjj), t2);
    const int i = j + blockIdx.x * 256 + threadIdx.x;
    if (i == j) { D[j] = dj; if (t2 > fmax(eps, cjj)) atomicAdd(&fs->clamp_rows, 1); }
    if (i < n && j < n) Sout[(size_t)j * ld + i] = (i == j) ? sqrt(dj) : sqrt(dj) * (Wf[(size_t)j * ld + i] / dj);
}

// k_gmw_stats: gamma / xi of an arbitrary symmetric G (stand-alone GMW entry point)
__global__ __launch_bounds__(256) void k_gmw_stats(int n, int ld, const double* __restrict__ G, FrameScalars* __restrict__ fs)
{
    const int r = blockIdx.x;
    double gmax = 0.0, xmax = 0.0;
    for (int c = threadIdx.x; c < n; c += 256) {
        const double v = (c >= r) ? G[(size_t)r * ld + c] : G[(size_t)c * ld + r];
        if (c == r) gmax = fmax(gmax, v); else xmax = fmax(xmax, v);
    }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if ((threadIdx.x & 63) == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}

extern "C" {
// ms.Z != null: MEAS_SLICES * ms.gx extra workgroups compute the measurement statistics of the frame
void srukf_launch_pxy(hipStream_t st, KDims d, const double* DZ, const double* S, double* Ut, const void* tiles, int ntiles, KWeights w, MeasArgs ms)
{
    const int extra = ms.Z ? MEAS_SLICES * ms.gx : 0;
    hipLaunchKernelGGL(k_pxy, dim3(ntiles + extra), dim3(256), 0, st, d, DZ, S, Ut, (const int2*)tiles, ntiles, w, ms);
}
// dxp != null: (n + 255)/256 extra workgroups apply the pending state update
void srukf_launch_syrk(hipStream_t st, KDims d, const double* S, const double* Ut, int ub, int ue, double* G, FrameScalars* fs,
                       const void* tiles, int ntiles, const double* dxp, double* X)
{
    const int extra = dxp ? (d.n + 255) / 256 : 0;
    hipLaunchKernelGGL(k_syrk, dim3(ntiles + extra), dim3(256), 0, st, d, S, Ut, ub, ue, G, fs, (const int2*)tiles, ntiles, dxp, X);
}
int srukf_gmw_sync_bytes(int T) { return (int)(sizeof(GmwSync) + sizeof(unsigned long long) * (size_t)T * T); }
// host-side tile list of the persistent launch: every tile (I, J), 1 <= I <= J < T, with the number of panel updates
// its owner applies (I off the diagonal; I - 1 on it: the pivot applies the last one itself), ordered by the step at
// which it is finished, so that worker w and worker w + workers hold tiles that retire at different times.
// Returns the number of tiles; out (4 shorts per tile) may be null.
int srukf_gmw_build_tiles(int T, short* out)
{
    int cnt = 0;
    for (int I = 1; I < T; I++)
        for (int J = I; J < T; J++) {
            const int ns = (I == J) ? I - 1 : I;
            if (ns < 1) continue;
            if (out) { out[4 * cnt] = (short)I; out[4 * cnt + 1] = (short)J; out[4 * cnt + 2] = (short)ns; out[4 * cnt + 3] = 0; }
            cnt++;
        }
    return cnt;
}
// workers the persistent launch needs for T block rows (each owns at most GMW_OWNED_MAX tiles); -1: too many tiles
int srukf_gmw_persist_workers(int T, int max_workers)
{
    const int nt = srukf_gmw_build_tiles(T, nullptr);
    if (nt == 0) return 0;
    if (nt <= max_workers) return nt;
    if (nt <= GMW_OWNED_MAX * max_workers) return max_workers;
    return -1;
}
// S0 / Ut0 / [u0, u1): see k_gmw_persist (null: every tile is read from G)
void srukf_launch_gmw_persist(hipStream_t st, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout,
                              void* sync, const void* tiles, int ntiles, int workers, void* fs,
                              const double* S0, const double* Ut0, int u0, int u1)
{
    const int T = ld / 64;
    hipLaunchKernelGGL(k_gmw_persist, dim3(1 + workers), dim3(256), 0, st, n, ld, T, G, (GmwPanel64*)pans, Sout, D, eps,
                       (GmwSync*)sync, (const GmwTile*)tiles, ntiles, (FrameScalars*)fs, S0, Ut0, u0, u1);
}
int srukf_gmw_head_rows(void) { return 64 * GMW_HEAD_ROWS; }
// 64-row panel step; j0 = -64 factors the first 64x64 region only (one workgroup)
void srukf_launch_gmw_step64(hipStream_t st, int n, int ld, int j0, double eps, double* G, const void* cur, void* nxt, double* D, double* Sout)
{
    const int rem = ld - j0 - 64;
    if (rem <= 0) return;
    const int T = (j0 < 0) ? 1 : rem / 64;
    hipLaunchKernelGGL(k_gmw_step64, dim3(T, T), dim3(256), 0, st, n, ld, j0, j0 < 0 ? 1 : 0, G, (const GmwPanel64*)cur, Sout, (GmwPanel64*)nxt, D, eps);
}
int srukf_gmw_panel_bytes(void) { return (int)sizeof(GmwPanel64); }
void srukf_launch_gmw_check(hipStream_t st, int n, int ld, const double* D, const double* S, FrameScalars* fs, const double* X, int do_traj, double* Scopy)
{
    hipLaunchKernelGGL(k_gmw_check, dim3(n + (do_traj ? 1 : 0)), dim3(256), 0, st, n, ld, D, S, fs, X, do_traj, Scopy);
}
void srukf_launch_gmw_col(hipStream_t st, int n, int ld, int j, double eps, const double* G, double* Wf, double* D,
                          unsigned long long* theta_bits, FrameScalars* fs, double* Sout)
{
    const int blocks = (ld - j + 255) / 256;
    hipLaunchKernelGGL(k_gmw_col_a, dim3(blocks), dim3(256), 0, st, ld, j, G, Wf, D, theta_bits);
    hipLaunchKernelGGL(k_gmw_col_b, dim3(blocks), dim3(256), 0, st, n, ld, j, eps, Wf, D, theta_bits, fs, Sout);
}
void srukf_launch_gmw_stats(hipStream_t st, int n, int ld, const double* G, FrameScalars* fs)
{
    hipLaunchKernelGGL(k_gmw_stats, dim3(n), dim3(256), 0, st, n, ld, G, fs);
}
}  // extern "C"
