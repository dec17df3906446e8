// srukf_gmw_panel.h — what the two forms of the blocked GMW factorisation share: the panel buffer, the three MFMA
// stages of a panel slab and the update of one 64x64 trailing tile (srukf_factor.hip: one launch per 64-row panel;
// srukf_gmw_persist.hip: one persistent launch).  gfx950 only.
#pragma once
#include "srukf_device.h"
#include "srukf_tiles.h"
#include "srukf_gmw_cols.h"

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
// early2: the whole panel buffer is known to be complete already (a worker that is behind): the second half's operands
// are requested together with the first half's — one memory round trip per step instead of two.
// early2_light: of the second half only the T2 fragments are requested early, the per-row scales follow after wait_full (fewer live registers).
// rows32: the panel has only its first 32 pivots (the rank-aware form's last pivoted panel when the kept pivots end there) and the
// tile's own values are not needed: the call ends after the first half with the S rows j0 .. j0+31.
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
                                                bool load_tile, bool store_tile, WaitHalf&& wait_half, WaitFull&& wait_full, Stored&& stored,
                                                bool early2 = false, bool rows32 = false, bool early2_light = false)
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
    double tb0[4], tb1[8], dr2[2][4], sqr[4][4];
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
        if (early2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int o = (4 * u + lk) * 32 + lr;
                if (u < 4) tb0[u] = cur->Tt2[o];
                tb1[u] = cur->Tt2[o + 16];
            }
          if (!early2_light) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) dr2[q][t] = (which == 0) ? cur->rD[32 + 16 * q + lk + 4 * t] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) sqr[q][t] = write_s ? cur->sq[16 * q + lk + 4 * t] : 0.0;
          }
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) dr[q][t] = (which == 0) ? cur->rD[16 * q + lk + 4 * t] : 0.0;
        if (rows32 && !early2) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) sqr[q][t] = write_s ? cur->sq[16 * q + lk + 4 * t] : 0.0;
        }
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
    if (rows32) {
        // the panel ends after its first 32 pivots: S rows j0 .. j0+31 of this wave's half slab, nothing else
        if (write_s) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int jj = 16 * a + lk + 4 * t;
                        if (j0 + jj < n) Sout[(size_t)(j0 + jj) * ld + n0 + 16 * b + lr] = W1[a][b][t] * sqr[a][t];
                    }
        }
        __syncthreads();                                       // the LDS slabs may be rewritten by the caller's next step
        return true;
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
    if (slab) {
        if (!early2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int o = (4 * u + lk) * 32 + lr;
                if (u < 4) tb0[u] = cur->Tt2[o];
                tb1[u] = cur->Tt2[o + 16];
            }
        }
        if (!early2 || early2_light) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) dr2[q][t] = (which == 0) ? cur->rD[32 + 16 * q + lk + 4 * t] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) sqr[q][t] = write_s ? cur->sq[16 * q + lk + 4 * t] : 0.0;
        }
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
                    if (which == 0) { Lr[jj][cc] = w2 * dr2[a][t]; if (diagblk) Wc[jj][cc] = w2; }
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
                    double* s1 = &Sout[(size_t)(j0 + jj) * ld + n0 + 16 * b + lr];
                    double* s2 = &Sout[(size_t)(j0 + 32 + jj) * ld + n0 + 16 * b + lr];
                    if (j0 + jj < n) *s1 = W1[a][b][t] * sqr[a][t];
                    if (j0 + 32 + jj < n) *s2 = W2[a][b][t] * sqr[2 + a][t];
                }
    }
    return true;
}


// ---- batched replay (srukf_run_frames_batch): the panel's slabs ONCE per column block, in global memory ----
// In the two forms above every trailing tile recomputes the slabs it needs (no workgroup waits for another: latency first).  With B filters per launch the
// trailing tiles are throughput work, and the slab stages are 80 of the 144 MFMAs of a tile step: here one wave per 32-column half forms W (64 x 32) and
// L = W / D with exactly the instruction sequence of gmw_tile_update's slab part and leaves them in the filter's slab rows Wb / Lb (row kk of the panel at
// [kk * ld + column]); the tiles then are plain K = 64 updates from those rows (k_gmw_trail_b).  write_s: the panel's final S rows for these columns.
__device__ __forceinline__ void gmw_slab_to_global(int n, int ld, int j0, int n0, const double* __restrict__ G, const GmwPanel64* __restrict__ cur,
                                                   double* __restrict__ Sout, double* __restrict__ Wb, double* __restrict__ Lb, bool write_s, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    if (n0 >= ld) return;
    d4 X2[2][2], W1[2][2], W2[2][2];
    double fb0[8], fb1[8];
    double ta0[4], ta1[8], ea0[8], ea1[8], tb0[4], tb1[8], dr[2][4], dr2[2][4], sqr[4][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) X2[a][b][t] = G[(size_t)(j0 + 32 + 16 * a + lk + 4 * t) * ld + n0 + 16 * b + lr];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        fb0[u] = G[(size_t)(j0 + 4 * u + lk) * ld + n0 + lr]; fb1[u] = G[(size_t)(j0 + 4 * u + lk) * ld + n0 + 16 + lr];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int o = (4 * u + lk) * 32 + lr;
        if (u < 4) { ta0[u] = cur->Tt1[o]; tb0[u] = cur->Tt2[o]; }
        ta1[u] = cur->Tt1[o + 16]; tb1[u] = cur->Tt2[o + 16];
        ea0[u] = cur->E[o]; ea1[u] = cur->E[o + 16];
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int t = 0; t < 4; t++) { dr[q][t] = cur->rD[16 * q + lk + 4 * t]; dr2[q][t] = cur->rD[32 + 16 * q + lk + 4 * t]; }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int t = 0; t < 4; t++) sqr[q][t] = write_s ? cur->sq[16 * q + lk + 4 * t] : 0.0;
    zero_acc(W1);
    // W1 = T1 G1
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
                const int jj = 16 * a + lk + 4 * t, col = n0 + 16 * b + lr;
                const double w1 = W1[a][b][t], w2 = W2[a][b][t];
                Wb[(size_t)jj * ld + col] = w1;        Lb[(size_t)jj * ld + col] = w1 * dr[a][t];
                Wb[(size_t)(32 + jj) * ld + col] = w2; Lb[(size_t)(32 + jj) * ld + col] = w2 * dr2[a][t];
                if (write_s) {
                    if (j0 + jj < n) Sout[(size_t)(j0 + jj) * ld + col] = w1 * sqr[a][t];
                    if (j0 + 32 + jj < n) Sout[(size_t)(j0 + 32 + jj) * ld + col] = w2 * sqr[2 + a][t];
                }
            }
}

// ---- split form of the persistent factorisation (srukf_gmw_persist.hip, N >= 340): a panel's slabs by SLAB workgroups, 16 columns per wave ----
// The same three stages on a 16-column quarter (the per-element MFMA sequences do not depend on how many column blocks a wave carries: bit-identical to the
// half-slab forms above).  DEV: the panel's G rows were written by a tile workgroup of the SAME launch pair (agent-scope loads), the slab rows are read by tile
// workgroups of it (write-through stores).  rows32: the panel has only its first 32 pivots (last pivoted panel of the rank-aware form): S rows j0 .. j0+31 only.
template <bool DEV>
__device__ __forceinline__ void gmw_slab16_to_global(int n, int ld, int j0, int n0, const double* __restrict__ G, const GmwPanel64* __restrict__ cur,
                                                     double* __restrict__ Sout, double* __restrict__ Wb, double* __restrict__ Lb, bool write_s, bool rows32, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    if (n0 >= ld) return;
    d4 X2[2], W1[2], W2[2];
    double fb[8], ta0[4], ta1[8], ea0[8], ea1[8], tb0[4], tb1[8], dr[2][4], dr2[2][4], sqr[4][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) X2[a][t] = rows32 ? 0.0 : ld_g<DEV>(&G[(size_t)(j0 + 32 + 16 * a + lk + 4 * t) * ld + n0 + lr]);
#pragma unroll
    for (int u = 0; u < 8; u++) fb[u] = ld_g<DEV>(&G[(size_t)(j0 + 4 * u + lk) * ld + n0 + lr]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int o = (4 * u + lk) * 32 + lr;
        if (u < 4) { ta0[u] = cur->Tt1[o]; tb0[u] = rows32 ? 0.0 : cur->Tt2[o]; }
        ta1[u] = cur->Tt1[o + 16]; tb1[u] = rows32 ? 0.0 : cur->Tt2[o + 16];
        ea0[u] = rows32 ? 0.0 : cur->E[o]; ea1[u] = rows32 ? 0.0 : cur->E[o + 16];
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int t = 0; t < 4; t++) { dr[q][t] = cur->rD[16 * q + lk + 4 * t]; dr2[q][t] = rows32 ? 0.0 : cur->rD[32 + 16 * q + lk + 4 * t]; }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int t = 0; t < 4; t++) sqr[q][t] = (write_s && !(rows32 && q >= 2)) ? cur->sq[16 * q + lk + 4 * t] : 0.0;
    W1[0] = (d4){0, 0, 0, 0}; W1[1] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < 4) W1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0[u], fb[u], W1[0], 0, 0, 0);
        W1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1[u], fb[u], W1[1], 0, 0, 0);
    }
    if (rows32) {
        if (write_s) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int t = 0; t < 4; t++) { const int jj = 16 * a + lk + 4 * t; if (j0 + jj < n) Sout[(size_t)(j0 + jj) * ld + n0 + lr] = W1[a][t] * sqr[a][t]; }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int a2 = u >> 2, t = u & 3;
        X2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea0[u], W1[a2][t], X2[0], 0, 0, 0);
        X2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ea1[u], W1[a2][t], X2[1], 0, 0, 0);
    }
    W2[0] = (d4){0, 0, 0, 0}; W2[1] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int a2 = u >> 2, t = u & 3;
        if (u < 4) W2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb0[u], X2[a2][t], W2[0], 0, 0, 0);
        W2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb1[u], X2[a2][t], W2[1], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int jj = 16 * a + lk + 4 * t, col = n0 + lr;
            const double w1 = W1[a][t], w2 = W2[a][t];
            if constexpr (DEV) {
                st_dev(&Wb[(size_t)jj * ld + col], w1);        st_dev(&Lb[(size_t)jj * ld + col], w1 * dr[a][t]);
                st_dev(&Wb[(size_t)(32 + jj) * ld + col], w2); st_dev(&Lb[(size_t)(32 + jj) * ld + col], w2 * dr2[a][t]);
            } else {
                Wb[(size_t)jj * ld + col] = w1;        Lb[(size_t)jj * ld + col] = w1 * dr[a][t];
                Wb[(size_t)(32 + jj) * ld + col] = w2; Lb[(size_t)(32 + jj) * ld + col] = w2 * dr2[a][t];
            }
            if (write_s) {
                if (j0 + jj < n) Sout[(size_t)(j0 + jj) * ld + col] = w1 * sqr[a][t];
                if (j0 + 32 + jj < n) Sout[(size_t)(j0 + 32 + jj) * ld + col] = w2 * sqr[2 + a][t];
            }
        }
}
