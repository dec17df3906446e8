// srukf_gmw_persist.hip — the blocked GMW factorisation (modifiedCholeskyDecomposition, SLAM.cpp:2197-2327) as ONE
// persistent launch: a resident pivot workgroup and worker workgroups that own the trailing tiles.  gfx950 only.
// The per-panel form, the theta-clamp check and the exact column path are in srukf_factor.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "srukf_device.h"
#include "srukf_tiles.h"
#include "srukf_gmw_cols.h"
#include "srukf_gmw_panel.h"
#include "srukf_rank.h"

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
#ifndef GMW_XWG_LIMIT
#define GMW_XWG_LIMIT (1 << 16)                 // ~50 ms; a legitimate wait is over in microseconds
#endif
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
// Abandon the launch (a bounded wait expired): everybody leaves, the frame is flagged.  The FIRST to give up also leaves who and where (site << 32 | blockIdx + 1) in the
// sync block's pad word 0 (srukf_debug_get "abort_code"): cold path, diagnostic only.
// sites: 1 pivot (operands of the next panel), 2 pivot (critical head tiles), 3 worker (head tiles), 4 worker (a tile step), 5 slab workgroup, 6 tile workgroup, 7 residency gate
__device__ __forceinline__ void gmw_abandon(GmwSync* sy, unsigned int site)
{
    __hip_atomic_store(&sy->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((threadIdx.x & 63) == 0) atomicCAS(&sy->pad[0], 0ull, ((unsigned long long)site << 32) | (unsigned long long)(blockIdx.x + 1));
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
// Tp <= T: number of 64-row panels that are actually pivoted.  Tp < T = rank-aware form: the matrix arrives permuted so that its
// structurally null directions come last, only the leading Tp panels are factored, and the column blocks behind them are carried
// along as ordinary off-diagonal tiles (row block Tp exists in the tile list so that somebody writes the last panel's S rows).
__device__ __forceinline__ void gmw_pivot_persist(int n, int ld, int T, int Tp, double eps, double* __restrict__ G, GmwPanel64* __restrict__ pans,
                                                  double* __restrict__ Dall, double* __restrict__ Sout, GmwSync* sy, unsigned long long ebase,
                                                  double (*Lr)[G64_LS], double (*Wc)[G64_LS], double* facreg, double* xreg, double* keepreg,
                                                  int* okp, int* halfcnt, int* stageok, int tid, int klim)
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
    for (int p = 0; p < Tp; p++) {
        const int j0 = 64 * (p - 1), base = 64 * p;
        const bool first = (p == 0);
        // rank-aware form: the kept pivots end inside the last pivoted panel.  When they end in its first half (klim <= base + 32)
        // the second 32-pivot factor would work on null directions only (rows nobody reads): it is not run — ~4 us of the chain.
        const bool half_only = (p == Tp - 1) && (Tp < T) && (klim <= base + 32);
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
        if (wv0) { gmw_cols_pivot_wave(ws, eps, lane); GMW_TS(sy, p + 64, 4); }
        else if (wvu == 2) { gmw_cols_t_wave<0>(ws, lane, nullptr, kp.T1); GMW_TS(sy, p + 64, 5); }
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
            if (wv1) GMW_TS(sy, p + 64, 6); else GMW_TS(sy, p + 64, 7);
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
        if (!half_only) {
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
        unsigned long long early = 0;                          // waves 1 / 3: the flag of the tile they stage next, requested under their last output group
        if (wv0) { if (!half_only) gmw_cols_pivot_wave(ws2, eps, lane); GMW_TS(sy, p + 64, 0); }
        else if (wvu == 2) {
            if (!half_only) {
            // this wave has slack while it follows the pivots: it asks early whether the owners of the two tiles of the NEXT
            // panel have finished them (normally yes) and tells the others through LDS — saves every wave the ~1 us poll
            // round trip at the end of the iteration
            unsigned long long fa = 0, fb = 0;
            if (p >= 1 && p + 1 < Tp) {
                fa = __hip_atomic_load(&ver[GMW_VIDX(p, p + 1, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fb = __hip_atomic_load(&ver[GMW_VIDX((p + 1), p + 1, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gmw_cols_t_wave<2>(ws2, lane, nxt->Tt2, kp.T2, [&] {
                if (p == 0 || gmw_uniform64(fa) >= ebase + p) stageok[0] = 1;
                if (p == 0 || gmw_uniform64(fb) >= ebase + p) stageok[1] = 1;
            });
            GMW_TS(sy, p + 192, 0);
            }
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
            if (!half_only) gmw_cols_out_wave<true>(ws2, wv1 ? 0 : 1, lane, n, ld, base + 32, nxt->D + 32, nxt->sq + 32, nxt->rD + 32, Dall, Sout, kp.sq + 32, kp.rD + 32,
                                                    [&] { if (p >= 1 && p + 1 < Tp) early = __hip_atomic_load(&ver[GMW_VIDX((wv1 ? p : p + 1), p + 1, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
            if (wv1) GMW_TS(sy, p + 64, 1); else GMW_TS(sy, p + 64, 2);
        }
        if (p + 1 < Tp) {
            const bool tileA = wvu < 2;                        // waves 0, 1: tile (p, p+1) -> Lr;  waves 2, 3: tile (p+1, p+1) -> Wc
            const int tr = tileA ? p : p + 1, r0 = (wvu & 1) ? 0 : 32;
            if (wv1) GMW_TS(sy, p, 5);
            const bool ready = p == 0 || gmw_uniform64(early) >= ebase + p || __builtin_amdgcn_readfirstlane(stageok[tileA ? 0 : 1]) != 0 ||
                               gmw_wait_ge(&ver[GMW_VIDX(tr, p + 1, T)], ebase + p, &sy->abort);
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
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // last pivoted panel: nothing to stage, but its stores must have landed before panel_ready
        __syncthreads();                                       // closes the iteration: staged tiles visible, LDS arrays reusable
        if (wv3 && p + 1 < T) { gmw_set_panel_flag(sy->panel_ready, ebase + p + 1, lane); GMW_TS(sy, p + 64, 3); }
        if (wv0) GMW_TS(sy, p, 7);
        if (!*okp) { if (wv0) gmw_abandon(sy, 1); return; }
    }
    // the last panel buffers are never read by a worker (steps T-2 and T-1 have no trailing tiles)
}

// A memory-tile worker that is behind the pivot: 2 = it requests ALL operands of a step at once (one memory round trip per step, as the register form does;
// with the accumulator set, the three-stage slab and both halves' operands live the instance needs ~40 VGPR spills, 164 B of scratch), 1 = only the second
// half's T fragments early (27 spills), 0 = the second half requested after the first: no spills, no scratch (256 VGPRs + 236 AGPRs), one more round trip per step.
// Measured at N = 500 (round 4, k_gmw_persist<MEM> per launch / frames per second, fp64 and fp32 storage alike): 0: 506 us / 1 092;  1: 543 us / 1 048;
// 2: 587 us / 999 — the scratch traffic in the step loop costs more than the round trip it saves.  (Round 3's 553 us / 1 048 was form 2 with 22 spills.)
#ifndef GMW_MEM_EARLY2
#define GMW_MEM_EARLY2 0
#endif
// Worker side of one update step of an owned tile; returns false when a wait expired.
// kfirst: first panel step the owner takes part in.  passon: a tile of row block Tp of the rank-aware form — its own values are never used, it only
// turns the LAST pivoted panel into S rows for its columns, so it joins at that step (kfirst = Tp - 1), neither loads nor stores a tile and raises no flag.
struct GmwOwned { int I, J, nsteps; bool computed; int kfirst; bool passon; };
struct KDimsLite { int n, ld; };
// memtile: the tile lives in G between the steps (k_gmw_persist<true>: a worker owns more tiles than accumulator sets would
// fit): it is read at the start of every step and written back at the end — by the same lanes, with agent-scope accesses, so
// every lane sees its own earlier stores.
__device__ __forceinline__ bool gmw_owner_step(int n, int ld, int T, int k, const GmwOwned& tl, d4 (&acc)[2][2], double* __restrict__ G,
                                               GmwPanel64* pans, double* __restrict__ Sout, GmwSync* sy, unsigned long long ebase,
                                               double (*Lr)[G64_LS], double (*Wc)[G64_LS], int* okp, bool wv0, int tid, bool memtile = false, bool rows32 = false)
{
    unsigned long long* ver = gmw_sync_ver(sy);
    // the two row-panel tiles (k, I), (k, J) are finished (k updates each) — both flags, and the panel flag, in one round
    // trip.  A worker that finds panel k complete already is behind the pivot: it then asks for ALL operands of the step
    // at once and skips the two panel waits (6-7 us per step instead of ~11: it catches up).
    if (wv0) {
        const unsigned long long want = ebase + k;
        unsigned long long a = want, b = want, pr = 0;
        for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
            if (k > 0) {
                a = __hip_atomic_load(&ver[GMW_VIDX(k, tl.I, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(&ver[GMW_VIDX(k, tl.J, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pr = __hip_atomic_load(&sy->panel_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a = gmw_uniform64(a); b = gmw_uniform64(b); pr = gmw_uniform64(pr);
            if (a >= want && b >= want) break;
            if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
            __builtin_amdgcn_s_sleep(1);
        }
        *okp = (a >= want && b >= want) ? ((pr >= ebase + k + 1) ? 2 : 1) : 0;
    }
    __syncthreads();
    if (!*okp) return false;
    const bool behind = *okp == 2;
    const bool last = k == tl.nsteps - 1;
    __syncthreads();                                           // *okp is rewritten by the panel waits below
    return gmw_tile_update<true>(n, ld, 64 * k, tl.I - k - 1, tl.J - k - 1, G, pans + k, Sout, Lr, Wc, tid, acc, !tl.passon && (memtile || (k == 0 && !tl.computed)),
                                 !tl.passon && (memtile || last),
        [&] {
            if (!behind) { if (wv0) *okp = gmw_wait_ge(&sy->half_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], ebase + k + 1, &sy->abort); }
            __syncthreads();
            return *okp != 0;
        },
        [&] {
            if (!behind) { if (wv0) *okp = gmw_wait_ge(&sy->panel_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], ebase + k + 1, &sy->abort); }
            __syncthreads();
            return *okp != 0;
        },
        [&] { if (last && !tl.passon) gmw_publish(&ver[GMW_VIDX(tl.I, tl.J, T)], ebase + tl.nsteps, wv0); },
        behind && (!memtile || GMW_MEM_EARLY2 > 0), rows32 && tl.passon, memtile && GMW_MEM_EARLY2 == 1);
}

// k_gmw_persist: grid = 1 + workers; worker w owns tiles[w - 1] and tiles[w - 1 + workers] (if any), both kept in accumulator
// registers from their first update to their last.
// k_gmw_persist<MEM = true>: matrices with more tiles than 2 x workers (N >= 340: 1 081 tiles at N = 500).  Worker w owns
// tiles[w - 1 + m workers], m = 0 .. GMW_OWNED_MEM - 1; every tile lives in G and passes through one accumulator set per step
// (read, update, write back: ~2 us more per step than a register-resident tile, against a dispatch gap, a cold L2 and the pivot's
// reload per panel in the one-launch-per-panel form).  Tiles are listed in the order they retire, so within a step a worker
// takes the tile of the smallest block row first: the next step's row-panel tiles are the ones everybody waits for.
#define GMW_OWNED_MAX 2
#define GMW_OWNED_MEM 6
struct GmwTile { short I, J, nsteps, pad; };
// S0 != null: the tiles of block rows I >= GMW_HEAD_ROWS are not read from G but COMPUTED by their owners,
//   G[r][c] = sum_k S0[k][r] S0[k][c] - sum_{u0 <= m < u1} Ut0[m][r] Ut0[m][c]      (what k_syrk does, SLAM.cpp:2118-2120, 2149),
// while the pivot is already factoring the first panels (k_syrk then only runs for block rows 0 and 1: a few
// microseconds instead of ~37).  The owner of tile (I, J) needs 12.5 + 1.7 I us for it and has ~12 I us until somebody waits
// for the tile.  S0 must not be the buffer the factor is written to (Sout).
#define GMW_HEAD_ROWS 2
// ... except the diagonal tile (GMW_HEAD_ROWS, GMW_HEAD_ROWS), which k_syrk forms as well: the pivot workgroup needs it — one panel
// update applied — at the end of its second panel, 28 us into the launch, and an owner that first has to form it (16 us) and then
// apply panel 0 publishes it ~10 us too late (time stamps of scripts/mb/dbg_persist: the second panel's iteration took 23.5 us instead of 13.9).
// GMW_HEAD_EXTRA_DIAG = 2: the 2 x 2 block of tiles behind the head rows, (2,2), (2,3), (3,3) — the third panel's iteration still waited ~2 us for (3,3).
#define GMW_HEAD_EXTRA_DIAG 2
__device__ __forceinline__ bool gmw_owner_computes(int I, int J) { return I >= GMW_HEAD_ROWS && !(I < GMW_HEAD_ROWS + GMW_HEAD_EXTRA_DIAG && J < GMW_HEAD_ROWS + GMW_HEAD_EXTRA_DIAG); }
// DEEP: four rotating fragment buffers (a wave that has its SIMD to itself: the owners of the persistent launch); otherwise the double-buffered loop of k_syrk (128
// registers less: four waves per SIMD hide the latency instead — the batched launch, where a SIMD is shared).  Same products in the same order either way.
template <bool DEEP = true>
__device__ __forceinline__ void gmw_owner_syrk(const KDimsLite d, const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1,
                                               int I, int J, d4 (&acc)[2][2], FrameScalars* __restrict__ fs, int tid, int krows)
{
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = 64 * I + 32 * (wv >> 1), c0 = 64 * J + 32 * (wv & 1);
    zero_acc(acc);
    if (m0 >= d.ld || c0 >= d.ld || c0 + 32 <= m0) return;
    if constexpr (DEEP) {
        tile32_tn_deep<false>(acc, S0, d.ld, S0, d.ld, m0, c0, 0, min(m0 + 32, krows), lane);     // S0[k][r] = 0 for k > r and for k >= krows
        tile32_tn_deep<true>(acc, Ut0, d.ld, Ut0, d.ld, m0, c0, u0, u1, lane);
    } else {
        tile32_tn<false>(acc, S0, d.ld, S0, d.ld, m0, c0, 0, min(m0 + 32, krows), lane);
        tile32_tn<true>(acc, Ut0, d.ld, Ut0, d.ld, m0, c0, u0, u1, lane);
    }
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

// k_syrk_own: the tiles of S^T S - U U^T that the owners of the persistent launch would form themselves (gmw_owner_syrk), as a launch of its own — same device
// function, same summation order (K ascending per 32 x 32 quadrant, no split-K), so the factorisation that then READS its tiles from G gives, bit for bit, the
// result of the launch whose owners fold.  Used where a worker owns two register tiles (filters that share the GPU with three or four tenants: forming both tiles
// before the first step would hold up the pivot chain), so that a filter's results do not depend on how many filters run beside it.
// One workgroup per tile of the persistent launch's list, longest K first; tiles the head launch covers leave at once.
template <bool DEEP>
__device__ __forceinline__ void syrk_own_body(int n, int ld, const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1, int krows,
                                              double* __restrict__ G, FrameScalars* __restrict__ fs, const GmwTile* __restrict__ tiles, int nreal, const int bid)
{
    const GmwTile t = tiles[nreal - 1 - bid];
    if (!gmw_owner_computes(t.I, t.J)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    d4 acc[2][2];
    const KDimsLite dl = { n, ld };
    gmw_owner_syrk<DEEP>(dl, S0, Ut0, u0, u1, t.I, t.J, acc, fs, tid, krows);
    const int m0 = 64 * t.I + 32 * (wv >> 1), c0 = 64 * t.J + 32 * (wv & 1);
    if (m0 >= ld || c0 >= ld || c0 + 32 <= m0) return;         // (what gmw_tile_update's `live` loads)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int tt = 0; tt < 4; tt++) G[(size_t)(m0 + 16 * a + lk + 4 * tt) * ld + c0 + 16 * b + lr] = acc[a][b][tt];
}
__global__ __launch_bounds__(256) void k_syrk_own(int n, int ld, const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1, int krows,
                                                  double* __restrict__ G, FrameScalars* __restrict__ fs, const GmwTile* __restrict__ tiles, int nreal)
{
    syrk_own_body<true>(n, ld, S0, Ut0, u0, u1, krows, G, fs, tiles, nreal, (int)blockIdx.x);
}
// batched form: workgroup index = tile B + f (the B filters' longest tiles first)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_syrk_own_b(int n, int ld, const SyrkOwnArgs* __restrict__ tab, int B, int u0, int u1, int krows, const GmwTile* __restrict__ tiles, int nreal)
{
    const int f = (int)blockIdx.x % B, bid = (int)blockIdx.x / B;
    const SyrkOwnArgs a = tab[f];
    syrk_own_body<false>(n, ld, a.S0, a.Ut0, u0, u1, krows, a.G, a.fs, tiles, nreal, bid);
}

// Admission of persistent launches when several filters share the GPU (srukf_set_exclusive(ctx, 0)): every such launch keeps to
// half the CUs (GmwPlan) and is preceded on its stream by k_gmw_gate, which lets it start only while fewer than `limit` = 2 gated
// launches are in flight on the device — so the launches that run together are always resident together, in whatever order the
// hardware dispatches their workgroups (two partly resident launches holding each other's CUs would otherwise sit in the
// bounded wait until it expires).  The gate is one wave that fits beside a resident factorisation workgroup (few registers, no
// LDS), so it can always start; the last workgroup of a gated launch gives the slot back.
__device__ int g_gmw_admitted = 0;
__global__ void k_gmw_gate(FrameScalars* __restrict__ fs, int limit)
{
    if (fs->frozen) return;                                    // the launch behind this gate returns at once as well
    if (threadIdx.x != 0) return;
    for (int spins = 0; spins < (1 << 22); spins++) {          // ~ 1 s: a slot leaked by a lost launch must not hang the stream for good
        const int cur = __hip_atomic_load(&g_gmw_admitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < limit && atomicCAS(&g_gmw_admitted, cur, cur + 1) == cur) return;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(&g_gmw_admitted, 1);
    atomicAdd(&fs->gate_timeouts, 1);
}

// ---- head fold (exclusive replay of the rank-aware form) -------------------------------------------------------------------------
// What the k_syrk launch in front of the factorisation used to do rides on the persistent launch as HELPER workgroups in front of
// the pivot and the workers (blockIdx >= 1 + workers): the 32 x 32 tiles of the head rows of S^T S - U U^T (4-way split-K over the waves, as
// in k_syrk), the pending state update X += dX, the diagonal of the dropped positions.  Helpers are short (~5 us) and come first in
// dispatch order; the pivot and every worker's first step wait for sy->head_done == ha.ntiles.  One launch and its boundary (~13 us)
// less per frame, ~6 us more inside this one.
__device__ __forceinline__ void gmw_head_tile_job(int n, int ld, int krows, const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1,
                                                  int2 tl, double* __restrict__ G, FrameScalars* __restrict__ fs, double* smem, int tid, GmwSync* sy = nullptr, bool stamp = false)
{
    if (tl.x < 0) return;
    double (*red)[64][17] = (double (*)[64][17])smem;
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = tl.x * 32, n0 = tl.y * 32;
    d4 acc[2][2];
    zero_acc(acc);
    const int ke = min(m0 + 32, krows);
    const int ngs = ke >> 4, ngu = (u1 - u0) >> 4, ng = ngs + ngu;
    const int g0 = (ng * wv) >> 2, g1 = (ng * (wv + 1)) >> 2;
    // (a helper wave has its SIMD to itself, like an owner: three groups of loads in flight — with one group of look-ahead the critical
    //  tiles took 18 us under the owners' operand traffic, and the pivot's first panel waits for them; same summation order)
    if (g0 < ngs) tile32_tn_deep<false>(acc, S0, ld, S0, ld, m0, n0, g0 << 4, min(g1, ngs) << 4, lane);
    if (stamp && wv == 0) GMW_TS(sy, 132, 1);
    if (g1 > ngs) tile32_tn_deep<true>(acc, Ut0, ld, Ut0, ld, m0, n0, u0 + ((max(g0, ngs) - ngs) << 4), u0 + ((g1 - ngs) << 4), lane);
    if (stamp && wv == 0) GMW_TS(sy, 132, 2);
    if (stamp && wv == 3) GMW_TS(sy, 132, 6);
    splitk_reduce(acc, red, wv, lane);
    if (stamp && wv == 0) GMW_TS(sy, 132, 3);
    if (wv != 0) return;
    double gmax = 0.0, xmax = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int r = m0 + 16 * a + lk + 4 * t, c = n0 + 16 * b + lr;
                const double v = acc[a][b][t];
                st_dev(&G[(size_t)r * ld + c], v);             // read by other workgroups of this launch
                if (r < n && c < n) { if (r == c) gmax = fmax(gmax, v); else xmax = fmax(xmax, v); }
            }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if (lane == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}
__device__ __forceinline__ bool gmw_wait_head(const unsigned int* cnt, unsigned int want, const int* abort_flag)
{
    for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= want) return true;
        if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <bool MEM>
__global__ __launch_bounds__(256) void k_gmw_persist(int n, int ld, int T, int Tp, double* __restrict__ G, GmwPanel64* __restrict__ pans,
                                                     double* __restrict__ Sout, double* __restrict__ Dall, double eps,
                                                     GmwSync* __restrict__ sy, const GmwTile* __restrict__ tiles, int ntiles,
                                                     FrameScalars* __restrict__ fs,
                                                     const double* __restrict__ S0, const double* __restrict__ Ut0, int u0, int u1, int krows, int gated,
                                                     const HeadArgs ha)
{
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];
    __shared__ double xreg[1024 + 1024 + 32 * 33];
    __shared__ double keepreg[1024 + 64 + 64];
    __shared__ int ok, halfcnt, stageok[2];
    const int tid = threadIdx.x;
    const int nhelp = ha.nhelp;                                // head fold: helper workgroups BEHIND the pivot and the workers (0 without it)
    const int nmain = (int)gridDim.x - nhelp;                  // 1 + workers: dispatched first, so that the owners start forming their tiles at once
    const int role = (int)blockIdx.x < nmain ? (int)blockIdx.x : -1;      // -1: helper, 0: pivot, > 0: worker role - 1
    // What a role needs first — its entry of the tile list — is requested TOGETHER with the frozen flag and the epoch: a round trip in a freshly
    // launched grid is ~2 us (cold TLB and L2, 256 workgroups asking at once), and flag -> list entry -> operands was a chain of three of them in front
    // of the critical head tiles (time stamps: 7.5 us before the first MFMA).  The asm pins all four loads in front of the first use.
    int2 mytile = make_int2(-1, -1);
    unsigned long long t01 = 0, t23 = 0, t45 = 0;              // GmwTile entries (4 shorts each) of a worker's two tiles and of its pass-on tile
    // The pass-on tiles of the rank-aware form (row block Tp: they turn the last pivoted panel into S rows for their columns, hold no values) stand at the end
    // of the list and are a THIRD slot of the first npass workers: they join at step Tp - 1, when every register tile has retired (nsteps <= Tp - 1), and use
    // its accumulator set — T - Tp workgroups (CUs) fewer per launch than one worker per pass-on tile.
    const int npass = (!MEM && Tp < T) ? T - Tp : 0, nreal = ntiles - npass;
    if (!MEM) {
        if (role < 0) { const int hb = (int)blockIdx.x - nmain; if (hb < ha.ntiles) mytile = ha.tiles[hb]; }
        else if (role > 0) {
            const unsigned long long* tq = (const unsigned long long*)tiles;
            if (role - 1 < nreal) t01 = tq[role - 1];
            if (role - 1 + nmain - 1 < nreal) t23 = tq[role - 1 + nmain - 1];
            if (role - 1 < npass) t45 = tq[nreal + role - 1];
        }
    }
    int frozen_now = fs->frozen;
    unsigned long long epoch_now = sy->epoch;                  // written by the previous launch's last workgroup
    // (Measured, both ways, per instance.  Register-resident tiles: pinned as VECTOR registers 148-150 us per launch at N = 200, as scalar ones 152-153.5.
    //  Memory tiles (N = 500): as vector registers 46 VGPR spills and 627 us per launch, as scalar ones 27 spills and 575 us.)
    if constexpr (MEM) asm volatile("" : "+s"(frozen_now), "+s"(epoch_now));
    else asm volatile("" : "+v"(mytile.x), "+v"(mytile.y), "+v"(t01), "+v"(t23), "+v"(t45), "+v"(frozen_now), "+v"(epoch_now));
    if (frozen_now) return;                                    // staged replay behind a flagged frame: every workgroup leaves before it touches the sync block
    const unsigned long long ebase = epoch_now << GMW_EPOCH_SHIFT;
    // (Letting the critical head tiles go first — every other role sleeping 1 .. 5 us before its first operand loads — was measured: 195.5 us per
    //  frame without, 196.8 / 197.2 / 198.6 / 199.0 / 203.8 with 0.9 / 1.7 / 2.6 / 3.4 / 5.1 us: every chain of the launch is critical.  Only the
    //  non-critical HELPERS sleeping: 195.5 / 195.4 / 197.3 / 200.5 us with 0.9 / 1.7 / 2.6 / 4.3 us — nothing to gain either.  Nor from dispatching the critical
    //  helpers right behind the pivot, in front of the workers: 5 201 frames/s without, 5 195 / 5 197 / 5 177 with the first 8 / 16 / 32 helper jobs there.)
    if (role < 0) {
        if constexpr (!MEM) {                                  // (the memory-tile instance has no helpers: none of their code, none of their registers)
        // One helper workgroup per job, in this order: the head tiles (critical ones first), X += dX, the dropped diagonal.  They sit behind the
        // pivot and the workers in dispatch order; nobody waits for a helper that has not started (every wait is bounded).
        const int nhead = ha.ntiles + ha.ndx + ha.ngd;
        const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int job = (int)blockIdx.x - nmain;
#ifdef SRUKF_GMW_DBG
        if (tid == 0) atomicAdd(&sy->pad[1], 1ull);             // diagnostic builds: helpers started / finished (srukf_debug_get "pad1" / "pad2")
#endif
        if ((int)blockIdx.x == nmain) GMW_TS(sy, 131, 0);
        if (job < ha.ntiles) {
            if (job == 0) GMW_TS(sy, 132, 0);
            gmw_head_tile_job(n, ld, krows, S0, Ut0, u0, u1, mytile, G, fs, &Lr[0][0], tid, sy, job == 0);
            if (job == 0) GMW_TS(sy, 132, 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's tile stores have landed ...
            __syncthreads();
            if (job == 0) GMW_TS(sy, 132, 5);
            if (job < ha.ncrit && job < 32) GMW_TS(sy, 140 + job, 0);      // (diagnostic builds: when each critical tile has landed)
            // the list starts with the ha.ncrit tiles the pivot needs before its first panel ((0,0), (0,1), (1,1) in 64 x 64 terms)
            if (wvu == 0) { if (lane == 0) __hip_atomic_fetch_add(job < ha.ncrit ? gmw_head_crit(sy) : gmw_head_done(sy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else if (job < ha.ntiles + ha.ndx) {
            const int dj = job - ha.ntiles;
            if (ha.ra.prep_next && dj == 0 && tid == 255) fs->ctl_next_valid = srukf_prepare_control(fs, fs->frame + 1) ? 1 : 0;
            srukf_gain_dx_job(n, ld, ha.dxp, ha.X, dj, ha.xr1, ha.ra.f32round, ha.ra.dxN);
        } else if (job < nhead) srukf_rank_gdiag_job(n, ld, u1, ha.ra, &fs->gmax_bits, job - ha.ntiles - ha.ndx);
        if ((int)blockIdx.x == nmain) GMW_TS(sy, 131, 1);
        if ((int)blockIdx.x == (int)gridDim.x - 1) GMW_TS(sy, 131, 2);
#ifdef SRUKF_GMW_DBG
        __syncthreads();
        if (tid == 0) atomicAdd(&sy->pad[2], 1ull);
#endif
        }
    } else if (role == 0) {
        bool head_ok = true;
        GMW_TS(sy, 128, 0);
        if (ha.ntiles > 0) {                                   // region R_0 and the two tiles behind it come from the helpers of this launch
            if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) ok = gmw_wait_head(gmw_head_crit(sy), (unsigned)ha.ncrit, &sy->abort) ? 1 : 0;
            __syncthreads();
            head_ok = ok != 0;
            __syncthreads();
        }
        GMW_TS(sy, 128, 1);
        if (head_ok) gmw_pivot_persist(n, ld, T, Tp, eps, G, pans, Dall, Sout, sy, ebase, Lr, Wc, facreg, xreg, keepreg, &ok, &halfcnt, stageok, tid, krows);
        else if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) gmw_abandon(sy, 2);
        GMW_TS(sy, 128, 2);
    } else {
        const bool wv0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
        const int workers = nmain - 1, w = role - 1;
        bool good = true;
        const bool half_last = (Tp < T) && (krows <= 64 * (Tp - 1) + 32);     // the pivot stops after the first half of the last pivoted panel (gmw_pivot_persist)
        if constexpr (MEM) {
            int kmax = 0;
            for (int m = 0; m < GMW_OWNED_MEM; m++) if (w + m * workers < ntiles) kmax = max(kmax, (int)tiles[w + m * workers].nsteps);
            for (int k = 0; k < kmax && good; k++)
                for (int m = 0; m < GMW_OWNED_MEM && good; m++) {
                    const int ti = w + m * workers;
                    if (ti >= ntiles) break;
                    const GmwTile t = tiles[ti];
                    if (k >= t.nsteps) continue;
                    if (k < t.pad) continue;
                    const GmwOwned tm = { t.I, t.J, t.nsteps, false, t.pad, Tp < T && t.I == Tp };
                    d4 accm[2][2];
                    zero_acc(accm);
                    good = gmw_owner_step(n, ld, T, k, tm, accm, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid, true, half_last && k == Tp - 1);
                }
        } else {
        GmwOwned ta = { 0, 0, 0, false, 0, false }, tb = { 0, 0, 0, false, 0, false }, tc = { 0, 0, 0, false, 0, false };
        auto entry = [](unsigned long long q) { GmwTile t; t.I = (short)(q & 0xffff); t.J = (short)((q >> 16) & 0xffff); t.nsteps = (short)((q >> 32) & 0xffff); t.pad = (short)((q >> 48) & 0xffff); return t; };
        if (w < nreal) { const GmwTile t = entry(t01); ta.I = t.I; ta.J = t.J; ta.nsteps = t.nsteps; ta.kfirst = t.pad; ta.passon = Tp < T && t.I == Tp; }
        if (w + workers < nreal) { const GmwTile t = entry(t23); tb.I = t.I; tb.J = t.J; tb.nsteps = t.nsteps; tb.kfirst = t.pad; tb.passon = Tp < T && t.I == Tp; }
        if (w < npass) { const GmwTile t = entry(t45); tc.I = t.I; tc.J = t.J; tc.nsteps = t.nsteps; tc.kfirst = t.pad; tc.passon = true; tc.computed = true; }
        d4 acca[2][2], accb[2][2];
        zero_acc(acca); zero_acc(accb);
        if (role == 1) GMW_TS(sy, 129, 0);
        if (role == nmain - 1) GMW_TS(sy, 130, 0);
        if (S0) {
            const KDimsLite dl = { n, ld };
            // (rank-aware form: block row Tp only passes the last pivoted panel's factor rows on; its own values are never used)
            if (ta.nsteps > 0 && gmw_owner_computes(ta.I, ta.J)) { if (!ta.passon) gmw_owner_syrk(dl, S0, Ut0, u0, u1, ta.I, ta.J, acca, fs, tid, krows); ta.computed = true; }
            if (tb.nsteps > 0 && gmw_owner_computes(tb.I, tb.J)) { if (!tb.passon) gmw_owner_syrk(dl, S0, Ut0, u0, u1, tb.I, tb.J, accb, fs, tid, krows); tb.computed = true; }
        }
        if (role == 1) GMW_TS(sy, 129, 1);
        if (role == nmain - 1) GMW_TS(sy, 130, 1);
        if (ha.ntiles > 0) {                                   // step 0 reads the head rows of G (and a row-1 / 2 x 2-block tile is loaded from there)
            if (wv0) ok = (gmw_wait_head(gmw_head_crit(sy), (unsigned)ha.ncrit, &sy->abort) && gmw_wait_head(gmw_head_done(sy), (unsigned)(ha.ntiles - ha.ncrit), &sy->abort)) ? 1 : 0;
            __syncthreads();
            good = ok != 0;
            if (!good && wv0) gmw_abandon(sy, 3);
            __syncthreads();
        }
        if (role == 1) GMW_TS(sy, 129, 2);
        if (role == nmain - 1) GMW_TS(sy, 130, 2);
        const int kmax = max(ta.nsteps, tb.nsteps);
        for (int k = 0; k < kmax && good; k++) {
            if (k < ta.nsteps && k >= ta.kfirst) good = gmw_owner_step(n, ld, T, k, ta, acca, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid, false, half_last && k == Tp - 1);
            if (good && k < tb.nsteps && k >= tb.kfirst) good = gmw_owner_step(n, ld, T, k, tb, accb, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid, false, half_last && k == Tp - 1);
        }
        // the pass-on slot: step Tp - 1 only, behind every step of the register tiles (which end at Tp - 2 at the latest); its accumulator values are never used
        if (good && tc.nsteps > 0) good = gmw_owner_step(n, ld, T, tc.kfirst, tc, acca, G, pans, Sout, sy, ebase, Lr, Wc, &ok, wv0, tid, false, half_last);
        }
        if (!good && wv0) gmw_abandon(sy, 4);
        if (role == 1) GMW_TS(sy, 129, 3);
        if (role == nmain - 1) GMW_TS(sy, 130, 3);
    }
    // the last workgroup out re-arms the block for the next launch and reports an abandoned run
    GMW_DBG(sy, 6, 7777);
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, 0); atomicAdd(&fs->gmw_aborts, 1); }
#ifdef SRUKF_GMW_DBG
            sy->pad[3] = __hip_atomic_load(gmw_head_done(sy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sy->pad[4] = __hip_atomic_load(gmw_head_crit(sy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sy->pad[5] = gridDim.x; sy->pad[6] = (unsigned long long)nhelp;
#endif
            __hip_atomic_store(&sy->abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gmw_head_done(sy), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gmw_head_crit(sy), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->epoch, (ebase >> GMW_EPOCH_SHIFT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gated) atomicSub(&g_gmw_admitted, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split form (plans with more than two register tiles per worker: N >= 400 in the rank-aware form, N >= 340 with every pivot factored).  The memory-tile instance above runs every tile through one accumulator set per step — read, two slabs
// recomputed (80 of the 144 MFMAs of a step), update, written back: 10.1 GFLOP of MFMA and 784 MB of HBM traffic per launch at N = 500 for 4.67 GFLOP and 74 MB
// of algorithm, ~8 us per tile step on a lone wave per SIMD, and its early panels are bound by the workers.  Here the factorisation is TWO kernels that run side
// by side on two streams and talk through the same sync block:
//   k_gmw_pivslab_persist  1 pivot workgroup (gmw_pivot_persist, unchanged) + one SLAB workgroup per column block J: for every pivoted panel k < J it waits for
//                          the panel and for the finished row-panel tile (k, J), forms W_k(J) and L = W / D once (16 columns per wave) and leaves them in
//                          Wslab[k] / Lslab[k] (+ the panel's final S rows for its columns), then raises slabver[k][J];
//   k_gmw_tiles_persist    ONE lean workgroup per tile (few registers, no LDS: several per CU, all of the early rows resident at once): the tile stays in
//                          registers from its first update to its last, a step is the plain K = 64 update from the two slab rows it needs (L2 hits), then it
//                          is stored once and ver[I][J] raised — what the slab workgroups and the pivot wait for.
// Dependencies only point to earlier block rows and workgroups are dispatched in list (= row) order per XCD, so the tile launch makes progress whatever part of it
// is resident; the pivot / slab launch (T workgroups of one CU each) has to be resident as a whole: the tile launch sits behind k_gmw_split_gate, which waits for
// sy->resident.  Same arithmetic as the other forms (per-element MFMA sequences, multiplications by 1 / D): bit-identical.  Every wait is bounded.
// The two launches WAIT FOR EACH OTHER: their streams must sit on different hardware queues (two streams on one queue run their kernels one after the other) — the
// host side probes the pair when it creates the side stream (srukf_api.hip: streams_run_side_by_side) and uses the memory-tile instance if it finds none.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gmw_pivslab_persist(int n, int ld, int T, int Tp, double* __restrict__ G, GmwPanel64* __restrict__ pans,
                                                             double* __restrict__ Sout, double* __restrict__ Dall, double eps, GmwSync* __restrict__ sy,
                                                             FrameScalars* __restrict__ fs, int krows, double* __restrict__ Wslab, double* __restrict__ Lslab,
                                                             unsigned int total_exits)
{
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];
    __shared__ double xreg[1024 + 1024 + 32 * 33];
    __shared__ double keepreg[1024 + 64 + 64];
    __shared__ int ok, halfcnt, stageok[2];
    const int tid = threadIdx.x, lane = tid & 63;
    int frozen_now = fs->frozen;
    unsigned long long epoch_now = sy->epoch;
    asm volatile("" : "+s"(frozen_now), "+s"(epoch_now));
    if (frozen_now) return;
    const unsigned long long ebase = epoch_now << GMW_EPOCH_SHIFT;
    if (tid == 0) __hip_atomic_fetch_add(&sy->resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool wv0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    if (blockIdx.x == 0) {
        gmw_pivot_persist(n, ld, T, Tp, eps, G, pans, Dall, Sout, sy, ebase, Lr, Wc, facreg, xreg, keepreg, &ok, &halfcnt, stageok, tid, krows);
    } else {
        const int J = (int)blockIdx.x, wv = tid >> 6;
        unsigned long long* ver = gmw_sync_ver(sy);
        unsigned long long* slabver = gmw_sync_slabver(sy, T);
        const bool half_last = (Tp < T) && (krows <= 64 * (Tp - 1) + 32);
        const int kend = min(J, Tp);
        bool good = true;
        for (int k = 0; k < kend && good; k++) {
            if (wv0) {
                bool g2 = gmw_wait_ge(&sy->panel_ready[(blockIdx.x % GMW_FLAG_COPIES) * GMW_FLAG_STRIDE], ebase + k + 1, &sy->abort);
                if (g2 && k >= 1) g2 = gmw_wait_ge(&ver[GMW_VIDX(k, J, T)], ebase + k, &sy->abort);
                ok = g2 ? 1 : 0;
            }
            __syncthreads();
            good = ok != 0;
            if (good) {
                const bool r32 = half_last && k == Tp - 1;
                gmw_slab16_to_global<true>(n, ld, 64 * k, 64 * J + 16 * wv, G, pans + k, Sout, Wslab + (size_t)k * 64 * ld, Lslab + (size_t)k * 64 * ld, J >= k + 2 || (Tp < T && k == Tp - 1), r32, lane);
                gmw_publish(&slabver[GMW_VIDX(k, J, T)], ebase + 1, wv0);
            }
            __syncthreads();
        }
        if (!good && wv0) gmw_abandon(sy, 5);
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == total_exits - 1) {
            if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, 0); atomicAdd(&fs->gmw_aborts, 1); }
            __hip_atomic_store(&sy->abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->epoch, (ebase >> GMW_EPOCH_SHIFT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// one lean workgroup per tile of the list (register-resident from its first update to its last); waves = 32 x 32 quadrants
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_gmw_tiles_persist(int ld, int T, double* __restrict__ G, GmwSync* __restrict__ sy, const GmwTile* __restrict__ tiles, FrameScalars* __restrict__ fs,
                         const double* __restrict__ Wslab, const double* __restrict__ Lslab, unsigned int total_exits)
{
    __shared__ int ok;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const unsigned long long tq = ((const unsigned long long*)tiles)[blockIdx.x];
    int frozen_now = fs->frozen;
    unsigned long long epoch_now = sy->epoch;
    if (frozen_now) return;
    const unsigned long long ebase = epoch_now << GMW_EPOCH_SHIFT;
    const int I = (int)(short)(tq & 0xffff), J = (int)(short)((tq >> 16) & 0xffff), ns = (int)(short)((tq >> 32) & 0xffff);
    const bool wv0 = __builtin_amdgcn_readfirstlane(wv) == 0;
    unsigned long long* ver = gmw_sync_ver(sy);
    const unsigned long long* slabver = gmw_sync_slabver(sy, T);
    const int m0 = 64 * I + 32 * (wv >> 1), c0 = 64 * J + 32 * (wv & 1);
    const bool live = (m0 < ld) && (c0 < ld) && (c0 + 32 > m0);
    d4 acc[2][2];
    zero_acc(acc);
    if (live) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[a][b][t] = G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr];
    }
    bool good = true;
    for (int k = 0; k < ns && good; k++) {
        if (wv0) {
            const unsigned long long want = ebase + 1;
            unsigned long long a = 0, b = 0;
            for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
                a = gmw_uniform64(__hip_atomic_load(&slabver[GMW_VIDX(k, I, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b = gmw_uniform64(__hip_atomic_load(&slabver[GMW_VIDX(k, J, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (a >= want && b >= want) break;
                if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                // rows k+1, k+2 feed the next panels: tight poll.  The ~800 others have slack and poll about once per microsecond (two requests each to a
                // handful of flag lines: otherwise the pollers alone are a TB/s of fabric traffic in front of the pivot's loads)
                if (I - k <= 2) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
            }
            ok = (a >= want && b >= want) ? 1 : 0;
        }
        __syncthreads();
        good = ok != 0;
        if (good && live) {
            // (plain loads: the slab rows of panel k are written once per launch pair, before their flag, and read only behind it: no L2 can hold an older copy)
            const double* __restrict__ Lb = Lslab + (size_t)k * 64 * ld + m0 + lr;
            const double* __restrict__ Wb = Wslab + (size_t)k * 64 * ld + c0 + lr;
#pragma unroll
            for (int kk = 0; kk < 64; kk += 16) {
                double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const size_t ro = (size_t)(kk + 4 * u + lk) * ld;
                    a0[u] = Lb[ro]; a1[u] = Lb[ro + 16]; b0[u] = Wb[ro]; b1[u] = Wb[ro + 16];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b0[u], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b1[u], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b0[u], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b1[u], acc[1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // ok is rewritten by the next poll
    }
    if (good) {
        if (live) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) st_dev(&G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr], acc[a][b][t]);
        }
        gmw_publish(&ver[GMW_VIDX(I, J, T)], ebase + ns, wv0);
    } else if (wv0) gmw_abandon(sy, 6);
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == total_exits - 1) {
            if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, 0); atomicAdd(&fs->gmw_aborts, 1); }
            __hip_atomic_store(&sy->abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->epoch, (ebase >> GMW_EPOCH_SHIFT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// ------------------------------------------------------------------------------------------------
// Split FOLD (round 6; rank-aware replay of the split form, N >= 267): the tile launch also FORMS the tiles.  Until here a k_syrk launch over the kept rows
// (N = 500: 212 us at 56 TFLOP/s, the machine full) stood in front of the pair, and the pair then ran 24 panels of 13.6 us with the matrix pipes 57 % idle: a
// throughput launch in front of a latency chain, one after the other.  Here k_syrk keeps block row 0 and tile (1, 1) (what the pivot and the slab workgroups read
// unversioned) and the tile launch's grid is, per block row I = 1, 2, ..: the row's FORMING JOBS (one 32 x 32 tile of S^T S - U U^T each: k_syrk's workgroup, its
// K split over the four waves, its summation order — bit for bit what the launch in front stored), then the row's tile workgroups, which wait for their three or
// four quarters (formver) before they load the tile.  Dispatch is in grid order, so a row is formed ~9 us after the one before it while the chain needs one per
// 13.6 us; a forming job waits for nothing, a tile workgroup only for jobs in front of it in the grid and for the resident pivot / slab launch: whatever part of
// the grid is resident, it makes progress.  Jobs of one pair of columns sit on one XCD (list position % 8), so its L2 serves their operand slab once.
// entry: I, J, nsteps as GmwTile for a tile workgroup; nsteps = -1: forming job for the 32 x 32 tile (I, J) (32-row units); nsteps = -2: padding.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gmw_form_job(int n, int ld, int krows, const double* __restrict__ A, const double* __restrict__ Ut, int mp, int tr, int tc,
                                             double* __restrict__ G, FrameScalars* __restrict__ fs, double (*red)[64][17], int tid)
{
    const int lane = tid & 63, wv = tid >> 6;
    const int m0 = tr * 32, n0 = tc * 32;
    d4 acc[2][2];
    zero_acc(acc);
    const int ke = min(m0 + 32, krows);
    const int ngs = ke >> 4, ngu = mp >> 4, ng = ngs + ngu;
    const int g0 = (ng * wv) >> 2, g1 = (ng * (wv + 1)) >> 2;
    if (g0 < ngs) tile32_tn<false>(acc, A, ld, A, ld, m0, n0, g0 << 4, min(g1, ngs) << 4, lane);
    if (g1 > ngs) tile32_tn<true>(acc, Ut, ld, Ut, ld, m0, n0, (max(g0, ngs) - ngs) << 4, (g1 - ngs) << 4, lane);
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
                st_dev(&G[(size_t)r * ld + c], v);             // read by a tile workgroup of this launch, on whatever XCD
                if (r < n && c < n) { if (r == c) gmax = fmax(gmax, v); else xmax = fmax(xmax, v); }
            }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if (lane == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}
// A tile workgroup of the split fold: tile (I, J) in registers from its first update to its last (k_gmw_tiles_persist), behind the wait for its formed quarters.
__device__ __forceinline__ void gmw_fold_tile_wg(int ld, int T, int I, int J, int ns, double* __restrict__ G, GmwSync* __restrict__ sy, unsigned long long ebase,
                                                 const double* __restrict__ Wslab, const double* __restrict__ Lslab, int* okp, int head_rows)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const bool wv0 = __builtin_amdgcn_readfirstlane(wv) == 0;
    unsigned long long* ver = gmw_sync_ver(sy);
    const unsigned long long* slabver = gmw_sync_slabver(sy, T);
    unsigned long long* formver = gmw_sync_formver(sy, T);
    int& ok = *okp;
    {
        // ---- tile workgroup: k_gmw_tiles_persist behind the wait for its quarters ----
        // (its waves share their SIMDs with forming jobs, which keep the matrix pipe busy: the update steps of the rows the pivot waits for go first)
        __builtin_amdgcn_s_setprio(3);
        const int m0 = 64 * I + 32 * (wv >> 1), c0 = 64 * J + 32 * (wv & 1);
        const bool live = (m0 < ld) && (c0 < ld) && (c0 + 32 > m0);
        if (wv0) {
            // (the row's jobs stand right in front of it in the grid: 10 - 20 us away; ~800 of these workgroups are resident at N = 500, so they ask about once a microsecond)
            // (quarters the grid's jobs form: none for the block rows the launch in front has formed)
            const int nq = (I < head_rows || (I == 1 && J == 1)) ? 0 : (I == J ? 3 : 4);
            const unsigned long long want = ebase + nq;
            const unsigned long long* w = &formver[GMW_VIDX(I, J, T)];
            int got = nq == 0 ? 1 : 0;
            for (int spins = 0; spins < GMW_XWG_LIMIT && !got; spins++) {
                if (gmw_uniform64(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= want) got = 1;
                else {
                    if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                    if (I <= 2) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
                }
            }
            ok = got;
        }
        __syncthreads();
        bool good = ok != 0;
        __syncthreads();                                       // ok is rewritten by the first poll below
        d4 acc[2][2];
        zero_acc(acc);
        if (good && live) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] = ld_dev(&G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr]);
        }
        for (int k = 0; k < ns && good; k++) {
            if (wv0) {
                const unsigned long long want = ebase + 1;
                unsigned long long a = 0, b = 0;
                for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
                    a = gmw_uniform64(__hip_atomic_load(&slabver[GMW_VIDX(k, I, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    b = gmw_uniform64(__hip_atomic_load(&slabver[GMW_VIDX(k, J, T)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (a >= want && b >= want) break;
                    if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                    if (I - k <= 2) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
                }
                ok = (a >= want && b >= want) ? 1 : 0;
            }
            __syncthreads();
            good = ok != 0;
            if (good && live) {
                const double* __restrict__ Lb = Lslab + (size_t)k * 64 * ld + m0 + lr;
                const double* __restrict__ Wb = Wslab + (size_t)k * 64 * ld + c0 + lr;
                // all 64 operand values of the step requested at once: under the forming jobs' traffic a round trip is several microseconds, and four of them
                // one behind the other (k_gmw_tiles_persist's loop) were most of a panel's slack — same MFMA order
                double a0[16], a1[16], b0[16], b1[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const size_t ro = (size_t)(4 * u + lk) * ld;
                    a0[u] = Lb[ro]; a1[u] = Lb[ro + 16]; b0[u] = Wb[ro]; b1[u] = Wb[ro + 16];
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b0[u], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[u], b1[u], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b0[u], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[u], b1[u], acc[1][1], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        if (good) {
            if (live) {
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int t = 0; t < 4; t++) st_dev(&G[(size_t)(m0 + 16 * a + lk + 4 * t) * ld + c0 + 16 * b + lr], acc[a][b][t]);
            }
            gmw_publish(&ver[GMW_VIDX(I, J, T)], ebase + ns, wv0);
        } else if (wv0) gmw_abandon(sy, 6);
        }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_gmw_tiles_fold(int n, int ld, int T, double* __restrict__ G, GmwSync* __restrict__ sy, const GmwTile* __restrict__ list, FrameScalars* __restrict__ fs,
                      const double* __restrict__ Wslab, const double* __restrict__ Lslab, unsigned int total_exits,
                      const double* __restrict__ A, const double* __restrict__ Ut, int mp, int krows, int head_rows)
{
    __shared__ int ok;
    __shared__ double red[3][64][17];
    const int tid = threadIdx.x;
    const unsigned long long tq = ((const unsigned long long*)list)[blockIdx.x];
    int frozen_now = fs->frozen;
    unsigned long long epoch_now = sy->epoch;
    if (frozen_now) return;
    const unsigned long long ebase = epoch_now << GMW_EPOCH_SHIFT;
    const int I = (int)(short)(tq & 0xffff), J = (int)(short)((tq >> 16) & 0xffff), ns = (int)(short)((tq >> 32) & 0xffff);
    unsigned long long* formver = gmw_sync_formver(sy, T);
    if (ns == -1) {
        // ---- forming job: the 32 x 32 tile (I, J) of 32-row units ----
        gmw_form_job(n, ld, krows, A, Ut, mp, I, J, G, fs, red, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // wave 0's stores have landed
        if (tid == 0) {
            // count the quarter in its 64 x 64 tile's word: values of older launches are replaced (the words never need clearing), the quarters of this one add up
            unsigned long long* w = &formver[GMW_VIDX(I >> 1, J >> 1, T)];
            unsigned long long cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                const unsigned long long nxt = (cur >= ebase ? cur : ebase) + 1;
                if (__hip_atomic_compare_exchange_strong(w, &cur, nxt, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
        }
    } else if (ns >= 0) {
        gmw_fold_tile_wg(ld, T, I, J, ns, G, sy, ebase, Wslab, Lslab, &ok, head_rows);
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == total_exits - 1) {
            if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&fs->clamp_rows, 1); atomicMin(&fs->clamp_first, 0); atomicAdd(&fs->gmw_aborts, 1); }
            __hip_atomic_store(&sy->abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->epoch, (ebase >> GMW_EPOCH_SHIFT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// In front of k_gmw_tiles_persist on its stream: lets it start only when every workgroup of the pivot / slab launch is resident (their CUs are then taken; the
// tile workgroups get the others).  One wave; gives up after ~50 ms and abandons the launch pair (the frame is flagged).
__global__ void k_gmw_split_gate(GmwSync* __restrict__ sy, const FrameScalars* __restrict__ fs, unsigned int want)
{
    if (fs->frozen) return;
    for (int spins = 0; spins < GMW_XWG_LIMIT; spins++) {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sy->resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= want) return;
        __builtin_amdgcn_s_sleep(4);
    }
    if (threadIdx.x == 0) gmw_abandon(sy, 7);
}

extern "C" {
int srukf_gmw_sync_bytes(int T) { return (int)(sizeof(GmwSync) + 3 * sizeof(unsigned long long) * (size_t)T * T * GMW_VER_STRIDE); }
// host-side tile list of the persistent launch: every tile (I, J), 1 <= I <= J < T, with the number of panel updates
// its owner applies (I off the diagonal; I - 1 on it: the pivot applies the last one itself), ordered by the step at
// which it is finished, so that worker w and worker w + workers hold tiles that retire at different times.
// Returns the number of tiles; out (4 shorts per tile) may be null.
int srukf_gmw_build_tiles(int T, int Tp, short* out)
{
    int cnt = 0;
    const int Imax = (Tp < T) ? Tp : T - 1;                     // rank-aware form: row block Tp carries the last pivoted panel's S rows
    for (int I = 1; I <= Imax; I++)
        for (int J = I; J < T; J++) {
            const int ns = (I == J && I < Tp) ? I - 1 : I;      // the pivot applies the last update of the diagonal tiles it factors
            if (ns < 1) continue;
            // row block Tp of the rank-aware form only passes the last pivoted panel's factor rows on: it joins at that step
            if (out) { out[4 * cnt] = (short)I; out[4 * cnt + 1] = (short)J; out[4 * cnt + 2] = (short)ns; out[4 * cnt + 3] = (short)((Tp < T && I == Tp) ? Tp - 1 : 0); }
            cnt++;
        }
    return cnt;
}
// workers the persistent launch needs for T block rows (each owns at most GMW_OWNED_MAX tiles); -1: too many tiles
int srukf_gmw_persist_workers(int T, int Tp, int max_workers)
{
    const int nt = srukf_gmw_build_tiles(T, Tp, nullptr);
    if (nt == 0) return 0;
    // register form: the pass-on tiles (row block Tp of the rank-aware form, the last T - Tp entries of the list) ride as a third, register-free slot
    const int npass = (Tp > 0 && Tp < T) ? T - Tp : 0, nreal = nt - npass;
    if (nreal >= 1 && nreal <= GMW_OWNED_MAX * max_workers) {
        const int w = nreal <= max_workers ? nreal : max_workers;
        if (npass <= w) return w;
    }
    if (nt <= GMW_OWNED_MEM * max_workers) return max_workers;   // > GMW_OWNED_MAX tiles per worker: the memory-tile form (no third slot: every tile is a list entry)
    return -1;
}
// register form (true) or memory-tile form (false) for `workers` workers: what srukf_launch_gmw_persist_head decides by
static bool gmw_register_form(int T, int Tp, int ntiles, int workers)
{
    const int npass = (Tp > 0 && Tp < T) ? T - Tp : 0, nreal = ntiles - npass;
    return workers > 0 && nreal <= GMW_OWNED_MAX * workers && npass <= workers;
}
// S0 / Ut0 / [u0, u1): see k_gmw_persist (null: every tile is read from G); gate_limit > 0: behind k_gmw_gate
void srukf_launch_gmw_persist_head(hipStream_t st, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout,
                                   void* sync, const void* tiles, int ntiles, int workers, void* fs,
                                   const double* S0, const double* Ut0, int u0, int u1, int Tp, int krows, int gate_limit, const HeadArgs* hap)
{
    if (gate_limit > 0) hipLaunchKernelGGL(k_gmw_gate, dim3(1), dim3(64), 0, st, (FrameScalars*)fs, gate_limit);
    const int T = ld / 64;
    if (Tp <= 0 || Tp > T) Tp = T;
    if (krows <= 0 || krows > ld) krows = ld;
    HeadArgs ha = {};
    if (hap) ha = *hap;
    if (workers > 0 && !gmw_register_form(T, Tp, ntiles, workers))
        hipLaunchKernelGGL((k_gmw_persist<true>), dim3(1 + workers), dim3(256), 0, st, n, ld, T, Tp, G, (GmwPanel64*)pans, Sout, D, eps,
                           (GmwSync*)sync, (const GmwTile*)tiles, ntiles, (FrameScalars*)fs, S0, Ut0, u0, u1, krows, gate_limit > 0 ? 1 : 0, HeadArgs{});
    else
        hipLaunchKernelGGL((k_gmw_persist<false>), dim3(ha.nhelp + 1 + workers), dim3(256), 0, st, n, ld, T, Tp, G, (GmwPanel64*)pans, Sout, D, eps,
                           (GmwSync*)sync, (const GmwTile*)tiles, ntiles, (FrameScalars*)fs, S0, Ut0, u0, u1, krows, gate_limit > 0 ? 1 : 0, ha);
}
void srukf_launch_gmw_persist(hipStream_t st, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout,
                              void* sync, const void* tiles, int ntiles, int workers, void* fs,
                              const double* S0, const double* Ut0, int u0, int u1, int Tp, int krows, int gate_limit)
{
    srukf_launch_gmw_persist_head(st, n, ld, eps, G, pans, D, Sout, sync, tiles, ntiles, workers, fs, S0, Ut0, u0, u1, Tp, krows, gate_limit, nullptr);
}
// the owners' tiles as a launch of its own (k_syrk_own): tiles / ntiles = the persistent launch's list, Tp / T its shape
void srukf_launch_syrk_own(hipStream_t st, int n, int ld, const double* S0, const double* Ut0, int u0, int u1, int krows, double* G, void* fs, const void* tiles, int ntiles, int Tp)
{
    const int T = ld / 64;
    const int nreal = ntiles - ((Tp > 0 && Tp < T) ? T - Tp : 0);
    if (krows <= 0 || krows > ld) krows = ld;
    if (nreal > 0) hipLaunchKernelGGL(k_syrk_own, dim3(nreal), dim3(256), 0, st, n, ld, S0, Ut0, u0, u1, krows, G, (FrameScalars*)fs, (const GmwTile*)tiles, nreal);
}
void srukf_launch_syrk_own_b(hipStream_t st, int n, int ld, const void* tab, int B, int u0, int u1, int krows, const void* tiles, int ntiles, int Tp)
{
    const int T = ld / 64;
    const int nreal = ntiles - ((Tp > 0 && Tp < T) ? T - Tp : 0);
    if (krows <= 0 || krows > ld) krows = ld;
    if (nreal > 0) hipLaunchKernelGGL(k_syrk_own_b, dim3(nreal * B), dim3(256), 0, st, n, ld, (const SyrkOwnArgs*)tab, B, u0, u1, krows, (const GmwTile*)tiles, nreal);
}
int srukf_gmw_register_form(int T, int Tp, int ntiles, int workers) { return gmw_register_form(T, Tp, ntiles, workers) ? 1 : 0; }
// split form: stA = the filter's stream (k_gmw_pivslab_persist), stB = its side stream (gate + k_gmw_tiles_persist); the caller orders stB behind whatever produced
// G (event) and stA behind stB afterwards.  tiles / ntiles: the plan's list (the pass-on tiles at its end are not launched: the slab workgroups write their S rows).
void srukf_launch_gmw_split(hipStream_t stA, hipStream_t stB, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout, void* sync,
                            const void* tiles, int ntiles, void* fs, int Tp, int krows, double* Wslab, double* Lslab, int starve)
{
    const int T = ld / 64;
    if (Tp <= 0 || Tp > T) Tp = T;
    if (krows <= 0 || krows > ld) krows = ld;
    const int nreal = ntiles - ((Tp < T) ? T - Tp : 0);
    const unsigned int total = (unsigned)(T + (starve ? 0 : nreal));
    hipLaunchKernelGGL(k_gmw_pivslab_persist, dim3(T), dim3(256), 0, stA, n, ld, T, Tp, G, (GmwPanel64*)pans, Sout, D, eps, (GmwSync*)sync, (FrameScalars*)fs, krows, Wslab, Lslab, total);
    hipLaunchKernelGGL(k_gmw_split_gate, dim3(1), dim3(64), 0, stB, (GmwSync*)sync, (const FrameScalars*)fs, (unsigned)T);
    // starve (tests): the tile launch never arrives, as if the GPU were taken — the bounded waits of the pivot and the slab workgroups expire, the frame is flagged
    if (nreal > 0 && !starve) hipLaunchKernelGGL(k_gmw_tiles_persist, dim3(nreal), dim3(256), 0, stB, ld, T, G, (GmwSync*)sync, (const GmwTile*)tiles, (FrameScalars*)fs, Wslab, Lslab, total);
}
// Measurement only (srukf_debug_split_replay): ONE launch of the pair, alone, against the buffers and flags a real run of the pair left behind (epoch set back by
// the caller): which = 0 the pivot / slab launch, 1 the tile launch without its residency gate.  The last workgroup out re-arms the block as in the real pair.
void srukf_launch_gmw_split_alone(hipStream_t st, int which, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout, void* sync,
                                  const void* tiles, int ntiles, void* fs, int Tp, int krows, double* Wslab, double* Lslab)
{
    const int T = ld / 64;
    if (Tp <= 0 || Tp > T) Tp = T;
    if (krows <= 0 || krows > ld) krows = ld;
    const int nreal = ntiles - ((Tp < T) ? T - Tp : 0);
    if (which == 0) hipLaunchKernelGGL(k_gmw_pivslab_persist, dim3(T), dim3(256), 0, st, n, ld, T, Tp, G, (GmwPanel64*)pans, Sout, D, eps, (GmwSync*)sync, (FrameScalars*)fs, krows, Wslab, Lslab, (unsigned)T);
    else if (nreal > 0) hipLaunchKernelGGL(k_gmw_tiles_persist, dim3(nreal), dim3(256), 0, st, ld, T, G, (GmwSync*)sync, (const GmwTile*)tiles, (FrameScalars*)fs, Wslab, Lslab, (unsigned)nreal);
}
// Split fold: the grid of k_gmw_tiles_fold (4 shorts per entry; out may be null; returns the number of entries) for T block columns, Tp pivoted panels.  Per block row
// I >= 1 with tile workgroups: the row's forming jobs — every 32 x 32 tile (tr, tc), tr in {2 I, 2 I + 1}, tc >= tr, that srukf_gmw_fold_head_tile() does not give to the
// k_syrk launch in front — with the jobs of one column pair tc / 2 on one XCD (entry index % 8), then the row's tile workgroups as srukf_gmw_build_tiles lists them
// (without the pass-on row); every segment padded to a multiple of 8 entries (nsteps = -2).
// How many block rows the launch in front forms: row 0 and tile (1, 1) at least — read unversioned by the pivot and the slab workgroups.  More than that because
// forming inside the pair is worth less than forming in front of it (the jobs share SIMDs and places with the tile workgroups: §10 of DESIGN.md) and the more so the
// fuller the machine is with tile workgroups: frames/s by head rows — N = 400 (Tp = 19): 1 2 168, 4 2 188, 8 2 130, 12 2 058 (all in front: 1 958); N = 450 (Tp = 22): 1 1 682,
// 4 1 702, 8 1 712, 10 1 720, 14 1 686 (1 587); N = 500 (Tp = 24, fp32 storage): 1 1 370, 4 1 394, 10 1 426, 12 1 433, 14 1 450, 16 1 440, 18 1 420 (1 342).  2 (Tp - 17) fits the three.
static int g_fold_head_override = 0;
void srukf_gmw_fold_head_override(int v) { g_fold_head_override = v; }
int srukf_gmw_fold_head_rows(int Tp)
{
    int v = 2 * (Tp - 17);
    if (g_fold_head_override > 0) v = g_fold_head_override;                          // (measurements: srukf_debug_set(0, "fold_head", v))
    return std::max(1, std::min(v, Tp - 4));
}
int srukf_gmw_fold_head_tile(int tr, int tc, int head_rows) { return (tr < 2 * head_rows || (tr < 4 && tc < 4)) ? 1 : 0; }
int srukf_gmw_build_fold_list(int T, int Tp, short* out)
{
    if (Tp <= 0 || Tp > T) Tp = T;
    const int nt = srukf_gmw_build_tiles(T, Tp, nullptr);
    std::vector<short> tk((size_t)4 * (nt > 0 ? nt : 1));
    srukf_gmw_build_tiles(T, Tp, tk.data());
    const int nreal = nt - ((Tp < T) ? T - Tp : 0);
    const int Ilast = (Tp < T) ? Tp - 1 : T - 1;
    const int head_rows = srukf_gmw_fold_head_rows(Tp);
    int cnt = 0;
    auto emit = [&](int a, int b, int c) { if (out) { out[4 * cnt] = (short)a; out[4 * cnt + 1] = (short)b; out[4 * cnt + 2] = (short)c; out[4 * cnt + 3] = 0; } cnt++; };
    auto pad8 = [&]() { while (cnt % 8) emit(0, 0, -2); };
    // Where a row's tile workgroups stand.  A tile workgroup holds its registers from its dispatch to its row's last update, working or waiting: with every row's
    // workgroups right behind the row's jobs, the rows that were formed ahead of the chain (forming: ~9 us per row, the chain: 13.6) filled the machine with
    // waiting workgroups — 805 of 836 places at N = 500 — and the forming jobs behind them got what was left: the pair took chain + forming (measured: 510 us
    // against 327 + 212 one after the other).  The slabs of every panel stay in Wslab / Lslab, so a tile workgroup may arrive late and catch up: row I stands
    // behind the jobs of row place(I) = max(I, lead_a I - lead_b) — about three panels (+ its catch-up time) before the chain needs it.
    const double lead_a = 1.38, lead_b = 4.6;
    auto place = [&](int I) { const int f = (int)floor(lead_a * I - lead_b); return std::min(Ilast, std::max(I, f)); };
    // Segments of `rows` block rows (the first `single` rows one by one: the chain waits for them): within a segment the jobs of one column pair stand together on
    // their XCD, row after row, so the pair's operand slab crosses the fabric once per segment instead of once per row.
    const int rows = 2, single = 2;
    for (int F0 = 1; F0 <= Ilast;) {
        const int F1 = std::min(Ilast, F0 <= single ? F0 : F0 + rows - 1);
        std::vector<std::pair<short, short>> q[8];
        for (int Jc = F0; Jc < T; Jc++)
            for (int F = F0; F <= F1 && F <= Jc; F++)
                for (int tc = 2 * Jc; tc <= 2 * Jc + 1; tc++)
                    for (int tr = 2 * F; tr <= 2 * F + 1 && tr <= tc; tr++)
                        if (!srukf_gmw_fold_head_tile(tr, tc, head_rows)) q[Jc % 8].push_back({ (short)tr, (short)tc });
        size_t longest = 0;
        for (int x = 0; x < 8; x++) longest = std::max(longest, q[x].size());
        for (size_t r = 0; r < longest; r++)
            for (int x = 0; x < 8; x++) { if (r < q[x].size()) emit(q[x][r].first, q[x][r].second, -1); else emit(0, 0, -2); }
        for (int I = 1; I <= Ilast; I++) {
            const int pl = place(I);
            if (pl < F0 || pl > F1) continue;
            for (int t = 0; t < nreal; t++) if (tk[4 * t] == I) emit(tk[4 * t], tk[4 * t + 1], tk[4 * t + 2]);
        }
        pad8();
        F0 = F1 + 1;
    }
    return cnt;
}
// the pair of the split form with the fold: A / Ut (mp rows) / krows = the operands k_syrk would have read (the permuted copy of the kept rows, U^T with permuted columns)
void srukf_launch_gmw_split_fold(hipStream_t stA, hipStream_t stB, int n, int ld, double eps, double* G, void* pans, double* D, double* Sout, void* sync,
                                 const void* list, int nlist, void* fs, int Tp, int krows, double* Wslab, double* Lslab, const double* A, const double* Ut, int mp)
{
    const int T = ld / 64;
    if (Tp <= 0 || Tp > T) Tp = T;
    if (krows <= 0 || krows > ld) krows = ld;
    const unsigned int total = (unsigned)(T + nlist);
    hipLaunchKernelGGL(k_gmw_pivslab_persist, dim3(T), dim3(256), 0, stA, n, ld, T, Tp, G, (GmwPanel64*)pans, Sout, D, eps, (GmwSync*)sync, (FrameScalars*)fs, krows, Wslab, Lslab, total);
    hipLaunchKernelGGL(k_gmw_split_gate, dim3(1), dim3(64), 0, stB, (GmwSync*)sync, (const FrameScalars*)fs, (unsigned)T);
    if (nlist > 0) hipLaunchKernelGGL(k_gmw_tiles_fold, dim3(nlist), dim3(256), 0, stB, n, ld, T, G, (GmwSync*)sync, (const GmwTile*)list, (FrameScalars*)fs, Wslab, Lslab, total, A, Ut, mp, krows, srukf_gmw_fold_head_rows(Tp));
}
int srukf_gmw_head_rows(void) { return 64 * GMW_HEAD_ROWS; }
int srukf_gmw_head_extra_diag(void) { return GMW_HEAD_EXTRA_DIAG; }
}  // extern "C"
