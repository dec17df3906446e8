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
#include "srukf_rank.h"
#include "srukf_gmw_cols.h"
#include "srukf_meas.h"
#include "srukf_motion_reduce.h"
#include <vector>
#include <algorithm>

// A statistics job that rides on a contraction launch has stored its partial sums: the LAST of the MEAS_SLICES jobs of a landmark
// group (device-scope counter per group) runs that group's final pass with its whole workgroup.  (One last workgroup for all
// landmarks, 208 dependent loads per thread, was an 8 us tail on the launch's critical path.)
// fold_flag (may be null): the gain fold of k_pxy2 — the group's h / Si / visible / PxyR are made visible device-wide and the word is raised behind them
__device__ __forceinline__ void meas_job_done(const KDims& d, const KWeights& w, const MeasArgs& ms, int bx, double* shm, unsigned int* fold_flag = nullptr)
{
    __shared__ int last;
    // what the final pass will want from memory besides the partial sums — rows 0, 1, 2 of Z at the group's columns — is requested by EVERY job before it counts itself:
    // for the one that turns out last it has arrived with the counter's reply (a dependent round trip less on the launch's longest chain)
#ifdef SRUKF_PXY2_DBG
#define PXY2_TS(slot) do { if (ms.hstamp && bx == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[slot] = t_; } } while (0)
#else
#define PXY2_TS(slot) do { } while (0)
#endif
    PXY2_TS(4);                                                // (group 0: the last writer is the last job to get here)
    MeasPre pre;
    const int kq = bx * 32 + (int)threadIdx.x;
    const bool mine = threadIdx.x < 32 && kq < d.N;
#pragma unroll
    for (int e = 0; e < 3; e++) pre.z[e] = mine ? *reinterpret_cast<const double2*>(ms.Z + (size_t)e * d.mp + 2 * kq) : make_double2(0.0, 0.0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's device-scope stores have landed ...
    __syncthreads();                                           // ... and so have the whole workgroup's, before the count
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&ms.fs->stat_cnt[bx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (done == MEAS_SLICES - 1);
        if (last) __hip_atomic_store(&ms.fs->stat_cnt[bx], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next frame
    }
    __syncthreads();
    if (!last) return;
    PXY2_TS(5);
    MeasOut mo;
    meas_final_group(d, w, ms.sigR, ms.Z, ms.part, ms.h, ms.Si, ms.vis, ms.PxyR, bx, shm, ms.fmode, &pre, &mo, fold_flag ? 1 : 0);      // (always handed over: a conditional pointer keeps the struct in scratch)
    PXY2_TS(3);
    if (fold_flag && threadIdx.x < 64) {                       // (the 32 lanes that stored the group's results — write-through — are in wave 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(fold_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!ms.hmirror) return;
    // step-wise API: the 32 lanes that stored this group's h / Si / visible (one wave) repeat them into the host's pinned buffer; the group that completes the count raises
    // the flag.  A wave's fence covers all its lanes' stores, and every group fences at system scope BEFORE it counts.
    if (threadIdx.x >= 32) return;
    if (mine) {
        double* hh = (double*)ms.hmirror;
        double* hSi = (double*)(ms.hmirror + ((const char*)ms.Si - (const char*)ms.h));
        int* hvis = (int*)(ms.hmirror + ((const char*)ms.vis - (const char*)ms.h));
        *(double2*)(hh + 2 * kq) = make_double2(mo.h[0], mo.h[1]);
        *(double2*)(hSi + 4 * kq) = make_double2(mo.si[0], mo.si[1]); *(double2*)(hSi + 4 * kq + 2) = make_double2(mo.si[2], mo.si[3]);
        hvis[kq] = mo.vis;
    }
    __threadfence_system();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&ms.fs->stat_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == ms.gx - 1) {
            __hip_atomic_store(&ms.fs->stat_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ms.hstamp) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[1] = t_; }      // (10 ns ticks; scripts: where in the launch the host gets its statistics)
            __hip_atomic_store(ms.hflag, ms.hseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

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
        meas_partial_job<true>(d, w, ms.xrob, ms.sigR, ms.Z, ms.part, job % ms.gx, job / ms.gx, shm);
        meas_job_done(d, w, ms, job % ms.gx, shm);
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
// k_pxy2: the same contraction for the rank-aware replay in "table" mode, on the PERMUTED operands:
//   P[m][b] = sum_{a <= b, a < r} DZp[a][m] * A[a][b]            m < mp, b < np (permuted column), a = permuted row
// DZp = DZ with its rows in permuted order (k_project_table writes it so), A = the kept rows of S in permuted column order.  The
// structurally null rows of S (sqrt(EPSILON) e_i) are not in A: their contribution, sqrt(EPSILON) DZ[i][m] to column i only, is
// added by k_gain.  K ends at r instead of n.
// k_pxy moves 158 MB from L2 to the CUs per launch (one fragment load per MFMA and lane) and is bound there (6.3 TB/s, 25 us).
// Here a workgroup owns a 64 x 64 tile of P; each 16-row K group of both operand slabs goes global -> registers -> LDS ONCE per
// workgroup (double buffered, one barrier per group) and feeds all four waves (32 x 32 sub-tile each): 16 MB from L2.  To fill
// the GPU the long K ranges are cut in two halves that go to two workgroups; the second half lands in P1 and k_gain adds the two
// (a + b in a fixed order: deterministic, no atomics).  The measurement statistics ride along as in k_pxy.
// tiles[w] = (mt, bt, half, halves).
// ------------------------------------------------------------------------------------------------
#define PXY2_LS 80
// 512 threads: EIGHT waves, two per SIMD — a single wave per SIMD issues FP64 MFMAs at 44 % of the rate (34 of 78 TFLOP/s,
// scripts/mb/mb_mfma_f64.hip; the four-wave form of this kernel took 20.6 us for its 8.5 us of MFMA).  Waves w and w + 4 share a
// 32 x 32 sub-tile and take k-steps 0-1 / 2-3 of every group; the upper four hand their accumulators over through LDS at the end.
// (the body takes the workgroup's index explicitly: k_pxy2 passes blockIdx.x, the batched launch k_pxy2_b the index within its filter's share of the grid)
// bounded wait of the gain fold (a whole wave polls under a wave-uniform condition): the word becomes non-zero / reaches `want`
__device__ __forceinline__ unsigned fold_wait(const unsigned int* p, unsigned want)
{
    unsigned v = 0;
    for (int spins = 0; spins < (1 << 15); spins++) {
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= want) break;
        __builtin_amdgcn_s_sleep(12);                          // (~0.3 us between polls: one poll per 27 ns from each of ~130 waiting workgroups slowed what they were waiting for)
    }
    return v;
}
// The gain of one finished tile (mt, bt), by the workgroup that finished it, from the tile's values IN ITS REGISTERS (v: this workgroup's K half, or the whole range;
// other: the other half, read back by the caller): U^T for 32 landmarks x 64 permuted columns (the block's robot columns from the statistics' robot rows), the landmarks'
// shares of the state update.  Waves 0 .. 3 of the tile's workgroup; a lane holds 16 (measurement row, column) elements: row m0 + mo + 16 a + lk + 4 t, column
// b0 + bo + 16 b + lr — the two rows of a landmark sit in lanes l and l ^ 16.  sm: >= 32 * 8 doubles + 32 ints of LDS.
// (First form, measured with time stamps (scripts/fold_stamps.py): every pair's products written through and read back by the job — the reads' traffic reached the
//  memory side exactly when the statistics' final passes, chains of dependent round trips, ran: their flags moved from 17 to 19 - 21 us and the launch got longer.)
__device__ __forceinline__ void pxy2_gain_job(const KDims& d, const MeasArgs& ms, const GainFold& gf, const d4 (&v)[2][2], const d4 (&other)[2][2], const bool split, const int half,
                                              const int mt, const int bt, const int mo, const int bo, double* sm)
{
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4, N = d.N, np = d.np, mp = d.mp, n = d.n;
    double (*glm)[8] = (double (*)[8])sm;
    int* gon = (int*)(sm + 32 * 8);
    __shared__ unsigned fold_state[2];
    const bool robot_blk = bt == gf.bt_r0 || bt == gf.bt_r1;
    // what does not depend on the statistics is requested BEFORE the wait for them: the columns' state rows, this frame's measurements, a null row's DZ term
    int rpc[2], rc[2];
#pragma unroll
    for (int b = 0; b < 2; b++) { rpc[b] = 64 * bt + bo + 16 * b + lr; rc[b] = gf.perm[rpc[b]]; }
    const int frame = ms.fs->frame;
    double zk0 = 0.0, zk1 = 0.0; int mk = 0;
    if (tid < 32 && 32 * mt + tid < N) {
        const int k = 32 * mt + tid;
        const double* z = gf.z_seq + (size_t)frame * 2 * N;
        zk0 = z[2 * k]; zk1 = z[2 * k + 1]; mk = gf.m_seq[(size_t)frame * N + k];
    }
    if (tid < 64) {                                            // wave 0 polls
        const unsigned st = fold_wait(gf.sync + FOLD_STAT(mt, gf.nmt, gf.nbt), 1u);
        const unsigned mq = robot_blk ? fold_wait(gf.sync + FOLD_MOTION(gf.nmt, gf.nbt), 1u) : 1u;
        if (tid == 0) { fold_state[0] = st; fold_state[1] = mq; }
    }
    __syncthreads();
    if (fold_state[0] == 0 || fold_state[1] != 1) {            // a wait expired (flag the frame) or the run is frozen behind a flagged frame (nothing to compute from)
        if (tid == 0 && (fold_state[0] == 0 || fold_state[1] == 0)) { atomicAdd(&ms.fs->clamp_rows, 1); atomicMin(&ms.fs->clamp_first, 0); }
        return;
    }
    if (tid < 64) FOLD_TS(gf, 512 + 4 * (mt * gf.nbt + bt) + 2);                                    // [2]: the waits are over
    if (tid < 32) {
        const int k = 32 * mt + tid;
        GainLm g = { 0, 0, 0, 0, 0, 0, 0 };
        if (k < N) {
            const int on = (mk != 0) && (__hip_atomic_load(&ms.vis[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
            g = srukf_gain_lm(ld_dev(&ms.Si[4 * k]), ld_dev(&ms.Si[4 * k + 1]), ld_dev(&ms.Si[4 * k + 2]), ld_dev(&ms.Si[4 * k + 3]), zk0, zk1,
                              ld_dev(&ms.h[2 * k]), ld_dev(&ms.h[2 * k + 1]), on ? 1 : 0);
        }
        glm[tid][0] = g.i00; glm[tid][1] = g.i01; glm[tid][2] = g.i10; glm[tid][3] = g.i11; glm[tid][4] = g.y0; glm[tid][5] = g.y1;
        gon[tid] = g.on;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int rp = rpc[b], r = rc[b];
        const int e = rp - (gf.r - 4);                         // 0 .. 3: a robot column
        const bool robot = e >= 0 && e < 4 && r < n;
        double dxs = 0.0, rse = 0.0;
        if (robot) { dxs = ld_dev(&ms.fs->Xr1[e]) - ms.sigR[e]; rse = ld_dev(&ms.sigR[(size_t)d.L * 8 + e]); }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int ml = mo + 16 * a + lk + 4 * t, m = 64 * mt + ml, kl = ml >> 1, k = 32 * mt + kl;       // measurement row (local / global), landmark (local / global)
                // this lane's row of Pxy
                double pown;
                if (robot) pown = (k < N) ? srukf_gain_recentre(ld_dev(&ms.PxyR[(size_t)e * mp + m]), dxs, ld_dev(&ms.PxyR[(size_t)4 * mp + m]), ld_dev(&ms.h[m]) - ms.Z[m], rse) : 0.0;
                else {
                    const bool nullrow = rp >= gf.r && r < n && k == r / 6;
                    const double h0v = half == 0 ? v[a][b][t] : other[a][b][t], h1v = half == 0 ? other[a][b][t] : v[a][b][t];
                    pown = srukf_gain_pxy(h0v, h1v, split, nullrow ? gf.DZp[(size_t)rp * mp + m] : 0.0, nullrow, gf.sqeps, gf.sc);
                }
                const double ppar = __shfl_xor(pown, 16);      // the landmark's other row (rows 2 k, 2 k + 1: lanes l, l ^ 16)
                const bool odd = (ml & 1) != 0;
                double u0 = 0.0, u1 = 0.0, cq = 0.0;
                if (k < N && gon[kl] && r < n) {
                    const GainLm g = { glm[kl][0], glm[kl][1], glm[kl][2], glm[kl][3], glm[kl][4], glm[kl][5], 1 };
                    srukf_gain_apply(g, odd ? ppar : pown, odd ? pown : ppar, u0, u1, cq);
                }
                if (k < N) {
                    gf.Utp[(size_t)m * np + rp] = odd ? u1 : u0;
                    if (!odd && r < n) gf.dxk[(size_t)k * np + r] = cq;
                }
            }
    }
}

template <bool FOLD>
__device__ __forceinline__ void pxy2_body(const KDims& d, const double* __restrict__ DZp, const double* __restrict__ A, double* __restrict__ P0, double* __restrict__ P1,
                                          const int4* __restrict__ tiles, int ntiles, int kr, const KWeights& w, const MeasArgs& ms, const int bid, const GainFold& gf_in)
{
    const GainFold& gf = gf_in;                                // (the motion workgroup and the statistics jobs; the tiles' epilogue reads its own copy late)
    __shared__ double shm[2 * 2 * 16 * PXY2_LS];               // [buf][A|B][k][PXY2_LS]; >= MEAS_SM_DOUBLES (statistics scratch), >= 4 x 64 x 17 (hand-over)
    static_assert(2 * 2 * 16 * PXY2_LS >= MEAS_SM_DOUBLES && 2 * 2 * 16 * PXY2_LS >= 4 * 64 * 17, "scratch");
    const int nstat = ms.Z ? MEAS_SLICES * ms.gx : 0;         // statistics jobs first: they start with the launch, the tiles fill in behind
    if (ms.preamble && bid == 0 && threadIdx.x == 0) srukf_frame_preamble(ms.fs);   // "tail" mode: nothing of this frame runs before this launch
    const int nmot = ms.fmode ? 1 : 0;                         // "fused tail" mode: workgroup 0 is the frame's motion step (sums over the table the previous frame's tail completed)
    if (nmot && bid == 0) {
        if (ms.hstamp && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[0] = t_; }
        if (threadIdx.x >= 256) return;
        const RankArgs ra0 = {};
        if (FOLD && threadIdx.x == 0) { ms.fs->gmax_bits = 0ull; ms.fs->ximax_bits = 0ull; }      // (what k_gain's first workgroup does for the refactorisation that follows)
        if (threadIdx.x < 64) FOLD_TS(gf, 0);
        motion_reduce_body<256, false>(d, w, const_cast<double*>(ms.X), nullptr, const_cast<double*>(ms.sigR), ms.Cm, ms.fs, ra0, shm);
        if (threadIdx.x < 64) FOLD_TS(gf, 1);
        if (FOLD) {
            // gain fold: the reduction's results (Cm, fs->Xr1, the rs row of the table) become visible to the tile workgroups' gain jobs, then — once every tile that
            // reads the robot columns of the permuted copy is through — the motion step's columns are committed (k_gain's first duty)
            __syncthreads();
            if (threadIdx.x < 4) {                             // (all that other workgroups read of the reduction: the mean and the rs row; write-through, no fence)
                st_dev(&ms.fs->Xr1[threadIdx.x], ms.fs->Xr1[threadIdx.x]);
                double* rs = const_cast<double*>(ms.sigR) + (size_t)d.L * 8 + threadIdx.x;
                st_dev(rs, *rs);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // the commit's operands are requested now, its stores wait for the tiles that still read the robot columns
            double4 cv[5]; int ca[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int r = threadIdx.x + 256 * i;
                cv[i] = make_double4(0, 0, 0, 0); ca[i] = 0;
                if (r < d.n) { cv[i] = *reinterpret_cast<const double4*>(ms.Cm + (size_t)r * 4); ca[i] = (r < d.n - 4) ? gf.iperm[r] : gf.r - 4 + (r - (d.n - 4)); }
            }
            __syncthreads();
            __shared__ unsigned mo_state;
            if (threadIdx.x < 64) {
                const int fz = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ms.fs->frozen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (threadIdx.x == 0) __hip_atomic_store(gf.sync + FOLD_MOTION(gf.nmt, gf.nbt), fz ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                FOLD_TS(gf, 2);
                const unsigned got = fz ? 0u : fold_wait(gf.sync + FOLD_ROBOT(gf.nmt, gf.nbt), (unsigned)gf.robot_tiles);
                FOLD_TS(gf, 3);
                if (threadIdx.x == 0) mo_state = fz ? 2u : (got >= (unsigned)gf.robot_tiles ? 1u : 0u);
            }
            __syncthreads();
            if (mo_state == 0 && threadIdx.x == 0) { atomicAdd(&ms.fs->clamp_rows, 1); atomicMin(&ms.fs->clamp_first, 0); }
            if (mo_state == 1) {
                const int n = d.n, ld = d.np;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int r = threadIdx.x + 256 * i;
                    if (r >= n) continue;
                    const double4 v = cv[i];
                    *reinterpret_cast<double2*>(gf.S + (size_t)r * ld + (n - 4)) = make_double2(v.x, v.y);
                    *reinterpret_cast<double2*>(gf.S + (size_t)r * ld + (n - 2)) = make_double2(v.z, v.w);
                    if (ca[i] < gf.r) { double* o = gf.A + (size_t)ca[i] * ld + (gf.r - 4); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
                for (int r = threadIdx.x + 256 * 5; r < n; r += 256) {          // (n > 1280: the rows beyond the prefetched ones)
                    const double4 v = *reinterpret_cast<const double4*>(ms.Cm + (size_t)r * 4);
                    *reinterpret_cast<double2*>(gf.S + (size_t)r * ld + (n - 4)) = make_double2(v.x, v.y);
                    *reinterpret_cast<double2*>(gf.S + (size_t)r * ld + (n - 2)) = make_double2(v.z, v.w);
                    const int arow = (r < n - 4) ? gf.iperm[r] : gf.r - 4 + (r - (n - 4));
                    if (arow < gf.r) { double* o = gf.A + (size_t)arow * ld + (gf.r - 4); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
            }
            if (threadIdx.x < 64) FOLD_TS(gf, 4);
        }
#ifdef SRUKF_PXY2_DBG
        if (ms.hstamp && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[2] = t_; }
#endif
        return;
    }
    // order of the grid: motion job, statistics jobs, tiles (PXY2_STATS_LAST = 1: the statistics behind the tiles)
#ifndef PXY2_STATS_LAST
#define PXY2_STATS_LAST 0                              // measured both ways: 5 207 / 5 207 frames/s at N = 200, 1 024 (last) / 1 033 (first) at N = 500
#endif
    const int stat0 = PXY2_STATS_LAST ? nmot + ntiles : nmot, tile0 = PXY2_STATS_LAST ? nmot : nmot + nstat;
    if (bid >= stat0 && bid < stat0 + nstat) {
        if (threadIdx.x >= 256) return;
#ifdef SRUKF_PXY2_DBG
        if (ms.hstamp && bid == stat0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[6] = t_; }
#endif
        const int job = bid - stat0;               // the statistics jobs are written for 256 threads: the upper half of the workgroup leaves (no barrier waits for it: s_barrier counts the waves still alive)
        meas_partial_job<true>(d, w, ms.xrob, ms.sigR, ms.Z, ms.part, job % ms.gx, job / ms.gx, shm, ms.ns);
        if (threadIdx.x < 64 && job / ms.gx == 0) FOLD_TS(gf, 16 + 4 * (job % ms.gx));
        meas_job_done(d, w, ms, job % ms.gx, shm, FOLD ? gf.sync + FOLD_STAT(job % ms.gx, gf.nmt, gf.nbt) : nullptr);
#ifdef SRUKF_FOLD_DBG
        if (FOLD && threadIdx.x < 64 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gf.sync + FOLD_STAT(job % ms.gx, gf.nmt, gf.nbt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) FOLD_TS(gf, 17 + 4 * (job % ms.gx));
#endif
        return;
    }
    if (bid - tile0 >= ntiles) return;                         // (padding of the batched grid)
    const int4 tl = tiles[bid - tile0];
    if (tl.x < 0) return;                                      // empty slot of the XCD-aware list
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = 64 * tl.x, b0 = 64 * tl.y;
    // A[a][b] = 0 for a > b and for a >= r.  Every range is a whole number of FOUR-group rounds (kr is rounded up to 64 rows — the
    // rows of A behind the kept ones are zero — and the cut falls on a multiple of four): the rounds below are straight-line code
    const int ng = min(4 * (tl.y + 1), ((kr + 63) >> 6) << 2);
    const int gh = (ng / 2) & ~3;
    const int g0 = (tl.w == 2 && tl.z == 1) ? gh : 0, g1 = (tl.w == 2 && tl.z == 0) ? gh : ng;
    double (*sA)[16][PXY2_LS] = (double (*)[16][PXY2_LS])shm;                          // sA[buf][k][m]
    double (*sB)[16][PXY2_LS] = (double (*)[16][PXY2_LS])(shm + 2 * 16 * PXY2_LS);     // sB[buf][k][b]
    // staging: thread t carries row t >> 5 of the group, two doubles at column 2 (t & 31), of both slabs
    const int srow = tid >> 5, scol = (tid & 31) * 2;
    const double* gA = DZp + (size_t)srow * d.mp + m0 + scol;
    const double* gB = A + (size_t)srow * d.np + b0 + scol;
    // The slab rows of FOUR groups are in flight at any time (a register ring): nothing else hides the ~2 us a load takes under
    // this kernel's own traffic, and one group of MFMAs covers 0.2-0.4 us (a single group of look-ahead measured 57 us for the launch).
    d2 ra[4], rb[4];                                        // (ext vectors: a ring of HIP_vector_type structs ends up in scratch)
#pragma unroll
    for (int u = 0; u < 4; u++) { ra[u] = *(const d2*)(gA + (size_t)(16 * (g0 + u)) * d.mp); rb[u] = *(const d2*)(gB + (size_t)(16 * (g0 + u)) * d.np); }
    *(d2*)&sA[0][srow][scol] = ra[0]; *(d2*)&sB[0][srow][scol] = rb[0];
    __syncthreads();
    d4 acc[2][2];
    zero_acc(acc);
    const int mo = 32 * ((wv >> 1) & 1), bo = 32 * (wv & 1), kh = 8 * (wv >> 2);
    for (int gb = g0; gb < g1; gb += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = gb + u;
            const int buf = u & 1;                             // (g - g0) & 1: gb - g0 is a multiple of 4
            // slot u held group g, which went to LDS at the end of the previous step: free for group g + 4.  Unconditional (the
            // last round re-reads the last group): a conditional load makes the compiler wait for ALL loads before every LDS write
            const int gn = min(g + 4, g1 - 1);
            ra[u] = *(const d2*)(gA + (size_t)(16 * gn) * d.mp); rb[u] = *(const d2*)(gB + (size_t)(16 * gn) * d.np);
            double fa0[2], fa1[2], fb0[2], fb1[2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                fa0[ks] = sA[buf][kh + 4 * ks + lk][mo + lr]; fa1[ks] = sA[buf][kh + 4 * ks + lk][mo + 16 + lr];
                fb0[ks] = sB[buf][kh + 4 * ks + lk][bo + lr]; fb1[ks] = sB[buf][kh + 4 * ks + lk][bo + 16 + lr];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[ks], fb0[ks], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[ks], fb1[ks], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[ks], fb0[ks], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[ks], fb1[ks], acc[1][1], 0, 0, 0);
            }
            *(d2*)&sA[buf ^ 1][srow][scol] = ra[(u + 1) & 3]; *(d2*)&sB[buf ^ 1][srow][scol] = rb[(u + 1) & 3];
            __syncthreads();
        }
    }
    // hand-over: waves 4..7 -> LDS -> waves 0..3 (fixed order: lower K steps + upper K steps)
    double (*ho)[64][17] = (double (*)[64][17])shm;
    if (wv >= 4) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) ho[wv - 4][lane][(a * 2 + b) * 4 + t] = acc[a][b][t];
    }
    __syncthreads();
    if (wv >= 4) return;
    double* P = (tl.w == 2 && tl.z == 1) ? P1 : P0;
    d4 vfin[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) vfin[a][b][t] = acc[a][b][t] + ho[wv][lane][(a * 2 + b) * 4 + t];
    if (!FOLD || tl.w == 2) {
        // (gain fold: only a K half is written — through, for the workgroup that finishes the pair; a tile whose K range is not cut never leaves the registers)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    double* pp = &P[(size_t)(m0 + mo + 16 * a + lk + 4 * t) * d.np + b0 + bo + 16 * b + lr];
                    if (FOLD) st_dev(pp, vfin[a][b][t]); else *pp = vfin[a][b][t];
                }
    }
    if (FOLD) {
        // (the fold's arguments are read from the kernel's argument segment HERE: held in scalar registers across the contraction they spilled — 31 SGPRs into vector lanes,
        //  and 104 VGPRs behind them)
        const GainFold* gq = &gf_in;
        asm volatile("" : "+s"(gq) :: "memory");
        const GainFold gf = *gq;
        // gain fold: count the tile (and, for the block(s) of the robot columns, tell the motion workgroup that one reader fewer is left); the workgroup that completes the
        // pair of K halves writes U^T for the tile
        __shared__ int fold_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // (waves 4 .. 7 have left: the barrier counts the waves still alive)
        if (tid == 0) {
            if (tl.y == gf.bt_r0 || tl.y == gf.bt_r1) __hip_atomic_fetch_add(gf.sync + FOLD_ROBOT(gf.nmt, gf.nbt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned old = __hip_atomic_fetch_add(gf.sync + FOLD_PAIR(tl.x, tl.y, gf.nbt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fold_last = (int)old == tl.w - 1;
        }
        __syncthreads();
        if (tid < 64) FOLD_TS(gf, 512 + 4 * (tl.x * gf.nbt + tl.y) + (fold_last ? 1 : 0));       // [0]: the first half's end (or nothing), [1]: the pair complete
        if (fold_last) {
            d4 oth[2][2];
            zero_acc(oth);
            if (tl.w == 2) {                                   // the other half, as the workgroup that finished first wrote it
                const double* Po = (tl.z == 1) ? P0 : P1;
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int t = 0; t < 4; t++) oth[a][b][t] = ld_dev(&Po[(size_t)(m0 + mo + 16 * a + lk + 4 * t) * d.np + b0 + bo + 16 * b + lr]);
            }
            pxy2_gain_job(d, ms, gf, vfin, oth, tl.w == 2, tl.w == 2 ? tl.z : 0, tl.x, tl.y, mo, bo, shm);
            if (tid < 64) FOLD_TS(gf, 512 + 4 * (tl.x * gf.nbt + tl.y) + 3);
        }
    }
#ifdef SRUKF_PXY2_DBG
    if (ms.hstamp && threadIdx.x == 0) {
        const int ti = bid - tile0;
        const int slot = ti == 0 ? 7 : -1;
        if (slot >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ms.hstamp[slot] = t_; }
    }
#endif
}
// (four waves per SIMD = two of these workgroups per CU: 128 VGPRs.  The tiles need 124; the statistics jobs' loads-in-flight would take 142 and halve the tiles' occupancy —
//  N = 500: 122 -> 142 us — so the cap is stated and the few registers over it in the statistics path go to scratch)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pxy2(KDims d, const double* __restrict__ DZp, const double* __restrict__ A, double* __restrict__ P0, double* __restrict__ P1,
                                              const int4* __restrict__ tiles, int ntiles, int kr, KWeights w, MeasArgs ms)
{
    pxy2_body<false>(d, DZp, A, P0, P1, tiles, ntiles, kr, w, ms, (int)blockIdx.x, GainFold{});
}
// the same with the gain fold compiled in (the "gain_fold" switch, off by default: DESIGN.md 10 — the fold's code in the plain kernel cost it 4 us, so it is a kernel of its own)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pxy2_fold(KDims d, const double* __restrict__ DZp, const double* __restrict__ A, double* __restrict__ P0, double* __restrict__ P1,
                                              const int4* __restrict__ tiles, int ntiles, int kr, KWeights w, MeasArgs ms, GainFold gf)
{
    pxy2_body<true>(d, DZp, A, P0, P1, tiles, ntiles, kr, w, ms, (int)blockIdx.x, gf);
}
// Batched form (srukf_run_frames_batch, B filters of one shape in ONE launch per stage): filter f owns workgroups [f per, (f + 1) per) — per is a multiple of 8, so
// the XCD-aware tile list keeps its meaning — and takes its pointers from tab[f] (device memory: one scalar load round trip at the head of the launch).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pxy2_b(KDims d, const Pxy2Args* __restrict__ tab, int per, const int4* __restrict__ tiles, int ntiles, int kr, KWeights w)
{
    const int f = (int)blockIdx.x / per, bid = (int)blockIdx.x - f * per;
    const Pxy2Args a = tab[f];
    pxy2_body<false>(d, a.DZp, a.A, a.P0, a.P1, tiles, ntiles, kr, w, a.ms, bid, GainFold{});
}

// ------------------------------------------------------------------------------------------------
// k_syrk: G = S^T S - U U^T on the upper triangle (SLAM.cpp:2118-2120, 2149 batched over all
// measurement columns; rows [ub, ue) of Ut select the columns that are downdated — all of them
// in BATCHED mode, a single one in SEQUENTIAL mode).  Also accumulates gamma = max diag(G) and
// xi = max(0, max offdiag(G)) for the GMW bound beta^2 (SLAM.cpp:2204-2211).
// grid = one workgroup per upper-triangle 32x32 tile (XCD-aware order from the tile table), 4-way
// split-K over the concatenated K range [S rows 0..r0+32) ++ [Ut rows).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void syrk_body(const KDims& d, const double* __restrict__ S, const double* __restrict__ Ut,
                                          int ub, int ue, double* __restrict__ G, FrameScalars* __restrict__ fs,
                                          const int2* __restrict__ tiles, int ntiles, const double* __restrict__ dxp, double* __restrict__ X,
                                          int krows, int ndx, const RankArgs& ra, const double* __restrict__ xr1, const int bid, const int nblocks)
{
    // workgroups past the tile list: the state update X += sum_k K_k (z_k - h_k) left over by k_gain, then (rank-aware replay)
    // the diagonal of G at the dropped positions
    if (bid >= nblocks) return;                                // (padding of the batched grid)
    if (bid >= ntiles + ndx) { srukf_rank_gdiag_job(d.n, d.np, ue, ra, &fs->gmax_bits, bid - ntiles - ndx); return; }
    if (bid >= ntiles) {
        if (ra.prep_next && bid == ntiles && threadIdx.x == 255) fs->ctl_next_valid = srukf_prepare_control(fs, fs->frame + 1) ? 1 : 0;
        srukf_gain_dx_job(d.n, d.np, dxp, X, bid - ntiles, xr1, ra.f32round, ra.dxN);
        return;
    }
    __shared__ double red[3][64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int2 tl = tiles[bid];  // upper-triangle tiles only, XCD-aware order
    if (tl.x < 0) return;
    const int m0 = tl.x * 32;    // row r
    const int n0 = tl.y * 32;    // col c
    d4 acc[2][2];
    zero_acc(acc);
    int ke = m0 + 32; if (ke > krows) ke = krows;        // S[k][r] = 0 for k > r; krows: rows of S that hold anything (a multiple of 16)
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
// (occupancy: 120 registers instead of 100 + 32 accumulator registers, four waves per SIMD instead of three: N = 500 244 -> 238 us per launch, 32 batched filters 16 320 -> 16 470 frames/s)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_syrk(KDims d, const double* __restrict__ S, const double* __restrict__ Ut,
                                              int ub, int ue, double* __restrict__ G, FrameScalars* __restrict__ fs,
                                              const int2* __restrict__ tiles, int ntiles, const double* __restrict__ dxp, double* __restrict__ X,
                                              int krows, int ndx, const RankArgs ra, const double* __restrict__ xr1)
{
    syrk_body(d, S, Ut, ub, ue, G, fs, tiles, ntiles, dxp, X, krows, ndx, ra, xr1, (int)blockIdx.x, (int)gridDim.x);
}
// batched form: filter f owns workgroups [f per, (f + 1) per), nblocks of them live; all downdate rows [0, ue) (the replay's launch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_syrk_b(KDims d, const SyrkArgs* __restrict__ tab, int per, int nblocks, int ue, const int2* __restrict__ tiles, int ntiles, int krows, int ndx)
{
    const int f = (int)blockIdx.x / per, bid = (int)blockIdx.x - f * per;
    const SyrkArgs a = tab[f];
    syrk_body(d, a.S, a.Ut, 0, ue, a.G, a.fs, tiles, ntiles, a.dxp, a.X, krows, ndx, a.ra, a.xr1, bid, nblocks);
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

#include "srukf_gmw_panel.h"

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

// k_gmw_step64: one launch per 64-row panel JJ = [j0, j0+64)  (j0 = -64, first = 1: only the first region is factored).
// Every 64x64 block of the trailing square (base = j0+64):
//   1. recomputes the panel rows it needs for its row slab and its column slab by the three MFMA stages
//      above (one 32-column half slab per wave, register resident) and keeps L = W/D and W in LDS;
//   2. updates its tile  G[r][c] -= sum_{kk<64} L[kk][r] W[kk][c];
//   3. first block row only: writes the final S rows j0..j0+63 for its column slab;
//   4. block (0,0) takes its own route (gmw_step64_block00).
// grid = (T, T), T = (ld - base)/64; blocks strictly below the diagonal exit.
__device__ __forceinline__ void gmw_step64_body(int n, int ld, int j0, int first, double* __restrict__ G,
                                                const GmwPanel64* __restrict__ cur, double* __restrict__ Sout, GmwPanel64* __restrict__ nxt,
                                                double* __restrict__ Dall, double eps, const int bx, const int by)
{
    if (bx < by) return;
    STAMP(0);
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];           // block (0,0): one column-factor workspace per sub-panel
    __shared__ double xreg[1024 + 1024 + 32 * 33];            // block (0,0): staged panel matrices, then X01 | X11 | T1'
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (bx == 0 && by == 0) {
        gmw_step64_block00(n, ld, j0, first, eps, G, cur, nxt, Dall, Sout, Lr, Wc, facreg, xreg, tid);
        return;
    }
    d4 acc[2][2];
    zero_acc(acc);
    gmw_tile_update<false>(n, ld, j0, by, bx, G, cur, Sout, Lr, Wc, tid, acc, true, true, [] { return true; }, [] { return true; }, [] {});
}
__global__ __launch_bounds__(256) void k_gmw_step64(int n, int ld, int j0, int first, double* __restrict__ G,
                                                    const GmwPanel64* __restrict__ cur, double* __restrict__ Sout, GmwPanel64* __restrict__ nxt,
                                                    double* __restrict__ Dall, double eps, const FrameScalars* __restrict__ fs)
{
    gmw_step64_body(n, ld, j0, first, G, cur, Sout, nxt, Dall, eps, (int)blockIdx.x, (int)blockIdx.y);
}
// Batched form: one launch per 64-row panel for B filters.  Workgroup index = idx B + f, so that the B critical-path workgroups (idx 0: block (0,0) of every filter)
// are dispatched first; idx -> (by, bx) over `rows` block rows x `cols` block columns of the trailing square (the rank-aware form only needs the rows of the kept
// pivots: the caller passes rows < cols), tiles below the diagonal leave at once.  cur / nxt alternate between the two panel buffers of a filter (flip).
__global__ __launch_bounds__(256) void k_gmw_step64_b(int n, int ld, int j0, int first, const Step64Args* __restrict__ tab, int B, int cols, int flip, double eps)
{
    const int f = (int)blockIdx.x % B, idx = (int)blockIdx.x / B;
    const Step64Args a = tab[f];
    gmw_step64_body(n, ld, j0, first, a.G, (const GmwPanel64*)a.pan[flip ^ 1], a.Sout, (GmwPanel64*)a.pan[flip], a.D, eps, idx % cols, idx / cols);
}

// Batched replay, split form of a panel step (srukf_gmw_panel.h, gmw_slab_to_global): launch A = the B critical-path workgroups (block (0,0): apply the panel to the
// next diagonal region, factor it) beside one slab workgroup per column block of every filter; launch B = the trailing tiles as plain K = 64 updates from the slab
// rows — no LDS, few registers, many waves per SIMD.  Same instruction sequences on the same values as k_gmw_step64: bit-identical.
// A: workgroup index = idx B + f; idx 0: block (0,0); idx 1 + q: the slabs of column blocks 2q and 2q + 1 (four waves: their 32-column halves).
// (sel: which of the two slab buffers behind a.Wb / a.Lb this panel's slabs go to — the K = 128 trailing update reads two panels' slabs)
__global__ __launch_bounds__(256) void k_gmw_pivslab_b(int n, int ld, int j0, int first, const Step64Args* __restrict__ tab, int B, int flip, double eps, int sel)
{
    __shared__ double Lr[64][G64_LS];
    __shared__ double Wc[64][G64_LS];
    __shared__ double facreg[2 * GMW_FAC_DOUBLES];
    __shared__ double xreg[1024 + 1024 + 32 * 33];
    const int f = (int)blockIdx.x % B, idx = (int)blockIdx.x / B;
    const Step64Args a = tab[f];
    const GmwPanel64* cur = (const GmwPanel64*)a.pan[flip ^ 1];
    if (idx == 0) {
        gmw_step64_block00(n, ld, j0, first, eps, a.G, cur, (GmwPanel64*)a.pan[flip], a.D, a.Sout, Lr, Wc, facreg, xreg, threadIdx.x);
        return;
    }
    const int wv = threadIdx.x >> 6, bx = 2 * (idx - 1) + (wv >> 1);
    gmw_slab_to_global(n, ld, j0, j0 + 64 + 64 * bx + 32 * (wv & 1), a.G, cur, a.Sout, a.Wb + (size_t)sel * 64 * ld, a.Lb + (size_t)sel * 64 * ld, bx >= 1, threadIdx.x & 63);
}
// B: workgroup index = idx B + f, idx -> (by, bx) over rows x cols; tiles below the diagonal and (0,0) leave at once
// thin: only what the NEXT panel step reads — block row 0 and tile (1, 1) (the K = 128 form: k_gmw_trail2_b applies this panel to everything else together with the next one)
__global__ __launch_bounds__(256) void k_gmw_trail_b(int ld, int j0, const Step64Args* __restrict__ tab, int B, int cols, int sel, int thin)
{
    const int f = (int)blockIdx.x % B, idx = (int)blockIdx.x / B;
    const int by = idx / cols, bx = idx % cols;
    if (bx < by || (bx == 0 && by == 0)) return;
    if (thin && !(by == 0 || (by == 1 && bx == 1))) return;
    const Step64Args a = tab[f];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const int base = j0 + 64;
    const int m0 = base + 64 * by + 32 * (wv >> 1), c0 = base + 64 * bx + 32 * (wv & 1);
    if (!((m0 < ld) && (c0 < ld) && (c0 + 32 > m0))) return;
    double* __restrict__ G = a.G;
    const double* __restrict__ Lb = a.Lb + (size_t)sel * 64 * ld; const double* __restrict__ Wb = a.Wb + (size_t)sel * 64 * ld;
    d4 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[q][b][t] = G[(size_t)(m0 + 16 * q + lk + 4 * t) * ld + c0 + 16 * b + lr];
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
        const double a0 = -Lb[(size_t)(k + lk) * ld + m0 + lr], a1 = -Lb[(size_t)(k + lk) * ld + m0 + 16 + lr];
        const double b0 = Wb[(size_t)(k + lk) * ld + c0 + lr], b1 = Wb[(size_t)(k + lk) * ld + c0 + 16 + lr];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) G[(size_t)(m0 + 16 * q + lk + 4 * t) * ld + c0 + 16 * b + lr] = acc[q][b][t];
}
// K = 128: the trailing tiles of panel step j0 (k_gmw_trail_b's set) take the PREVIOUS panel's update too — slabs sel ^ 1, which the thin launch of step j0 - 64 applied
// only to what this step's pivots and slabs read — and then this panel's (slabs sel): one pass over G for two panels, the products per element in the same order.
__global__ __launch_bounds__(256) void k_gmw_trail2_b(int ld, int j0, const Step64Args* __restrict__ tab, int B, int cols, int sel)
{
    const int f = (int)blockIdx.x % B, idx = (int)blockIdx.x / B;
    const int by = idx / cols, bx = idx % cols;
    if (bx < by || (bx == 0 && by == 0)) return;
    const Step64Args a = tab[f];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const int base = j0 + 64;
    const int m0 = base + 64 * by + 32 * (wv >> 1), c0 = base + 64 * bx + 32 * (wv & 1);
    if (!((m0 < ld) && (c0 < ld) && (c0 + 32 > m0))) return;
    double* __restrict__ G = a.G;
    d4 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[q][b][t] = G[(size_t)(m0 + 16 * q + lk + 4 * t) * ld + c0 + 16 * b + lr];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int sl = pass == 0 ? (sel ^ 1) : sel;
        const double* __restrict__ Lb = a.Lb + (size_t)sl * 64 * ld; const double* __restrict__ Wb = a.Wb + (size_t)sl * 64 * ld;
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            const double a0 = -Lb[(size_t)(k + lk) * ld + m0 + lr], a1 = -Lb[(size_t)(k + lk) * ld + m0 + 16 + lr];
            const double b0 = Wb[(size_t)(k + lk) * ld + c0 + lr], b1 = Wb[(size_t)(k + lk) * ld + c0 + 16 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) G[(size_t)(m0 + 16 * q + lk + 4 * t) * ld + c0 + 16 * b + lr] = acc[q][b][t];
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
            srukf_prepare_control(fs);                         // control of the next staged frame (k_project_motion)
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
// k_gmw_col(j), one launch per pivot j = 0 .. n (left-looking, SLAM.cpp:2220-2296):
//   1. row j - 1 is complete (the previous launch): D_{j-1} = max(EPSILON, |C_{j-1,j-1}|, theta_{j-1}^2 / beta^2) — every workgroup evaluates it for itself, the first
//      one records it — and S row j - 1 = sqrt(D) (W / D) goes out;
//   2. the multipliers m_k = W[k][j] / D[k], k < j, ONCE per workgroup into LDS;
//   3. W[j][i] = G[j][i] - sum_{k<j} m_k W[k][i] for i >= j (same terms in the same order as the reference's recurrence), theta_j = max_{i>j} |W[j][i]|.
// The launch with j = n only finishes row n - 1.  W is stored in Wf (full n x n, upper).  grid over i.
// (Until round 6 every thread of every column divided W[k][j] / D[k] itself inside the k loop — j dependent fp64 divisions in front of j dependent loads — and a second
//  launch per column wrote the S row: 126 ms per flagged frame at N = 200, 2 408 launches; bench.py "theta_clamp".)
__global__ __launch_bounds__(256) void k_gmw_col(int n, int ld, int j, double eps, const double* __restrict__ G, double* __restrict__ Wf, double* __restrict__ D,
                                                 unsigned long long* __restrict__ theta_bits, FrameScalars* __restrict__ fs, double* __restrict__ Sout)
{
    extern __shared__ double mk[];                              // j multipliers
    const int i = j + blockIdx.x * 256 + threadIdx.x;
    double dprev = 0.0;
    if (j > 0) {
        const double gamma = __longlong_as_double((long long)fs->gmax_bits);
        const double xi = __longlong_as_double((long long)fs->ximax_bits);
        const double nu = fmax(1.0, sqrt((double)n * n - 1.0));
        const double beta2 = fmax(fmax(gamma, xi / nu), 1e-15);
        const double th = __longlong_as_double((long long)theta_bits[j - 1]);
        const double cjj = fabs(Wf[(size_t)(j - 1) * ld + (j - 1)]);
        const double t2 = th * th / beta2;
        dprev = fmax(fmax(eps, cjj), t2);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            D[j - 1] = dprev;
            if (t2 > fmax(eps, cjj)) atomicAdd(&fs->clamp_rows, 1);
            if (j - 1 < n) Sout[(size_t)(j - 1) * ld + (j - 1)] = sqrt(dprev);
        }
        if (i < n && j - 1 < n) Sout[(size_t)(j - 1) * ld + i] = sqrt(dprev) * (Wf[(size_t)(j - 1) * ld + i] / dprev);
    }
    if (j >= n) return;
    for (int k = threadIdx.x; k < j; k += 256) mk[k] = Wf[(size_t)k * ld + j] / (k == j - 1 ? dprev : D[k]);
    __syncthreads();
    double v = 0.0;
    if (i < ld) {
        double acc = 0.0;
        const double* col = Wf + i;
        // (32 rows requested before the first product: the loop is a chain of dependent additions over independent loads — one row per trip was a memory round trip per row)
        int k = 0;
        for (; k + 32 <= j; k += 32) {
            double wv[32];
#pragma unroll
            for (int u = 0; u < 32; u++) wv[u] = col[(size_t)(k + u) * ld];
#pragma unroll
            for (int u = 0; u < 32; u++) acc += mk[k + u] * wv[u];
        }
        for (; k < j; k++) acc += mk[k] * col[(size_t)k * ld];
        v = G[(size_t)j * ld + i] - acc;
        Wf[(size_t)j * ld + i] = v;
    }
    double mx = (i < ld && i > j) ? fabs(v) : 0.0;
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(&theta_bits[j], (unsigned long long)__double_as_longlong(mx));
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
void srukf_launch_pxy2(hipStream_t st, KDims d, const double* DZp, const double* A, double* P0, double* P1, const void* tiles, int ntiles, int kr, KWeights w, MeasArgs ms, GainFold gf)
{
    const int extra = ms.Z ? MEAS_SLICES * ms.gx : 0;
    const dim3 grid(ntiles + extra + (ms.fmode ? 1 : 0));
    if (gf.sync) hipLaunchKernelGGL(k_pxy2_fold, grid, dim3(512), 0, st, d, DZp, A, P0, P1, (const int4*)tiles, ntiles, kr, w, ms, gf);
    else hipLaunchKernelGGL(k_pxy2, grid, dim3(512), 0, st, d, DZp, A, P0, P1, (const int4*)tiles, ntiles, kr, w, ms);
}
// host-side tile list of k_pxy2 (4 ints per workgroup: mt, bt, half, halves; mt < 0: empty slot).  K ranges of at least PXY2_SPLIT
// groups are cut in two.  XCD-aware: workgroup w runs on XCD w % 8 (round-robin dispatch; the statistics jobs in front of the
// list are a multiple of 8), and each XCD has its own L2 — so all mp / 64 tiles that contract the same slab of A (one (bt, half)
// pair: 0.16-0.33 MB) go to ONE XCD, pairs are dealt to the XCDs longest first onto the least loaded one, and each XCD walks its
// pairs longest first.  (In plain bt-major order every slab of A was fetched by seven XCDs: 72 MB from the Infinity Cache per launch.)
// Returns the number of slots; out may be null.
#ifndef PXY2_SPLIT
#define PXY2_SPLIT 16
#endif
int srukf_pxy2_build_tiles(int mp, int np, int kr, int* out)
{
    const int ngmax = ((kr + 63) / 64) * 4, nmt = mp / 64;
    struct Pair { int bt, h, halves, groups; };
    std::vector<Pair> pairs;                                   // sized by the matrix: every (bt, half) pair appears exactly once, whatever N
    pairs.reserve(2 * (size_t)(np / 64));
    for (int bt = np / 64 - 1; bt >= 0; bt--) {
        const int ng = (4 * (bt + 1) < ngmax) ? 4 * (bt + 1) : ngmax;
        const int halves = ng >= PXY2_SPLIT ? 2 : 1, gh = (ng / 2) & ~3;
        for (int h = 0; h < halves; h++) pairs.push_back({ bt, h, halves, halves == 1 ? ng : (h == 0 ? gh : ng - gh) });
    }
    const int npairs = (int)pairs.size();
    std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.groups > b.groups; });   // longest first
    int load[8] = { 0 };
    std::vector<int> lst[8];
    for (int a = 0; a < npairs; a++) {
        int x = 0;
        for (int q = 1; q < 8; q++) if (load[q] < load[x]) x = q;
        lst[x].push_back(a); load[x] += pairs[a].groups;
    }
    int maxlen = 0;
    for (int x = 0; x < 8; x++) maxlen = (int)lst[x].size() * nmt > maxlen ? (int)lst[x].size() * nmt : maxlen;
    if (out) {
        for (int q = 0; q < maxlen * 8; q++) { out[4 * q] = -1; out[4 * q + 1] = 0; out[4 * q + 2] = 0; out[4 * q + 3] = 1; }
        for (int x = 0; x < 8; x++)
            for (int a = 0; a < (int)lst[x].size(); a++)
                for (int mt = 0; mt < nmt; mt++) {
                    const Pair& pr = pairs[lst[x][a]];
                    int* o = out + 4 * ((a * nmt + mt) * 8 + x);
                    o[0] = mt; o[1] = pr.bt; o[2] = pr.h; o[3] = pr.halves;
                }
    }
    return maxlen * 8;
}
int srukf_pxy2_split_groups(void) { return PXY2_SPLIT; }
// dxp != null: (n + 255)/256 extra workgroups apply the pending state update
// ra.A != null (rank-aware replay): S / Ut are the permuted operands, ceil((n - r) / 16) more workgroups form the dropped diagonal
void srukf_launch_syrk(hipStream_t st, KDims d, const double* S, const double* Ut, int ub, int ue, double* G, FrameScalars* fs,
                       const void* tiles, int ntiles, const double* dxp, double* X, RankArgs ra, const double* xr1)
{
    const int ndx = dxp ? (d.n + 255) / 256 : 0;
    const int krows = ra.A ? min(d.np, (ra.r + 15) & ~15) : d.np;
    const int extra = ndx + (ra.A ? (d.n - ra.r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS : 0);
    hipLaunchKernelGGL(k_syrk, dim3(ntiles + extra), dim3(256), 0, st, d, S, Ut, ub, ue, G, fs, (const int2*)tiles, ntiles, dxp, X, krows, ndx, ra, xr1);
}
// 64-row panel step; j0 = -64 factors the first 64x64 region only (one workgroup)
void srukf_launch_gmw_step64(hipStream_t st, int n, int ld, int j0, double eps, double* G, const void* cur, void* nxt, double* D, double* Sout,
                             const FrameScalars* fs)
{
    const int rem = ld - j0 - 64;
    if (rem <= 0) return;
    const int T = (j0 < 0) ? 1 : rem / 64;
    hipLaunchKernelGGL(k_gmw_step64, dim3(T, T), dim3(256), 0, st, n, ld, j0, j0 < 0 ? 1 : 0, G, (const GmwPanel64*)cur, Sout, (GmwPanel64*)nxt, D, eps, fs);
}
int srukf_gmw_panel_bytes(void) { return (int)sizeof(GmwPanel64); }
// split form of a batched panel step: A (critical-path workgroups + slabs of every column block of the trailing square), B (trailing tiles of `rows` block rows)
void srukf_launch_gmw_pivslab_b(hipStream_t st, int n, int ld, int j0, double eps, const void* tab, int B, int flip, int sel)
{
    const int rem = ld - j0 - 64;
    if (rem <= 0) return;
    const int cols = (j0 < 0) ? 0 : rem / 64;                  // j0 = -64: no panel yet, block (0,0) only
    hipLaunchKernelGGL(k_gmw_pivslab_b, dim3((1 + (cols + 1) / 2) * B), dim3(256), 0, st, n, ld, j0, j0 < 0 ? 1 : 0, (const Step64Args*)tab, B, flip, eps, sel);
}
// form: 0 the plain K = 64 update of `rows` block rows; 1 thin (block row 0 and tile (1, 1) only: two block rows of workgroups); 2 K = 128 (this panel's slabs `sel` behind the
// previous panel's `sel ^ 1`)
void srukf_launch_gmw_trail_b(hipStream_t st, int ld, int j0, const void* tab, int B, int rows, int sel, int form)
{
    const int cols = (ld - j0 - 64) / 64;
    if (j0 < 0 || cols <= 0 || rows < 1) return;
    if (rows > cols) rows = cols;
    if (form == 1) rows = std::min(rows, 2);
    if (form == 2) hipLaunchKernelGGL(k_gmw_trail2_b, dim3(rows * cols * B), dim3(256), 0, st, ld, j0, (const Step64Args*)tab, B, cols, sel);
    else hipLaunchKernelGGL(k_gmw_trail_b, dim3(rows * cols * B), dim3(256), 0, st, ld, j0, (const Step64Args*)tab, B, cols, sel, form == 1 ? 1 : 0);
}
// ---- batched launches (srukf_run_frames_batch): B filters of one shape, one launch per stage, arguments per filter in device tables ----
int srukf_pxy2_b_per(int ntiles, int gx) { return (ntiles + MEAS_SLICES * gx + 1 + 7) & ~7; }
void srukf_launch_pxy2_b(hipStream_t st, KDims d, const void* tab, int B, const void* tiles, int ntiles, int kr, KWeights w, int gx)
{
    const int per = srukf_pxy2_b_per(ntiles, gx);
    hipLaunchKernelGGL(k_pxy2_b, dim3(per * B), dim3(512), 0, st, d, (const Pxy2Args*)tab, per, (const int4*)tiles, ntiles, kr, w);
}
void srukf_launch_syrk_b(hipStream_t st, KDims d, const void* tab, int B, const void* tiles, int ntiles, int krows, int ndx, int ngd)
{
    const int nblocks = ntiles + ndx + ngd, per = (nblocks + 7) & ~7;
    hipLaunchKernelGGL(k_syrk_b, dim3(per * B), dim3(256), 0, st, d, (const SyrkArgs*)tab, per, nblocks, d.mp, (const int2*)tiles, ntiles, krows, ndx);
}
// j0 = -64: the first 64 x 64 region only; rows = block rows of the trailing square that are updated (>= 1), flip = which panel buffer receives the new panel
void srukf_launch_gmw_step64_b(hipStream_t st, int n, int ld, int j0, double eps, const void* tab, int B, int rows, int flip)
{
    const int rem = ld - j0 - 64;
    if (rem <= 0) return;
    const int cols = (j0 < 0) ? 1 : rem / 64;
    if (j0 < 0 || rows < 1) rows = 1;
    if (rows > cols) rows = cols;
    hipLaunchKernelGGL(k_gmw_step64_b, dim3(rows * cols * B), dim3(256), 0, st, n, ld, j0, j0 < 0 ? 1 : 0, (const Step64Args*)tab, B, cols, flip, eps);
}
void srukf_launch_gmw_check(hipStream_t st, int n, int ld, const double* D, const double* S, FrameScalars* fs, const double* X, int do_traj, double* Scopy)
{
    hipLaunchKernelGGL(k_gmw_check, dim3(n + (do_traj ? 1 : 0)), dim3(256), 0, st, n, ld, D, S, fs, X, do_traj, Scopy);
}
// The exact path, right-looking and blocked (round 6; SLAM.cpp:2246-2262 is written right-looking: C[r][c] -= C[j][r] C[j][c] / D_j for every r, c behind pivot j).
// k_gmw_col above forms row j from ALL rows above it with (ld - j) / 256 <= 5 workgroups — j dependent memory round trips per thread, 37 us per pivot at n = 1 204 with
// the machine empty, 45 ms per flagged frame.  Here one launch takes B pivots j0 .. j0 + B - 1: EVERY workgroup loads those B rows (columns j0 .. n) into LDS and factors the
// B x (n - j0) panel itself — theta_j = the largest |C[j][c]|, c > j, over the whole row as the reference takes it, D_j = max(EPSILON, |C_jj|, theta_j^2 / beta^2), the
// pivot applied to the panel rows behind it — redundantly, so nobody waits for anybody; then it applies the B pivots, in order, to its 32 x 256 tile of the trailing triangle.
// Same operations on every element in the same order as one pivot per launch (B = 1 is that form: bit-identical); n / B launches and 1 / B of the trailing traffic.
#define GMW_RL_ROWS 32
#define GMW_RL_BMAX 8
__global__ __launch_bounds__(256) void k_gmw_rl(int n, int ld, int j0, int B, double eps, double* __restrict__ C, double* __restrict__ D,
                                                FrameScalars* __restrict__ fs, double* __restrict__ Sout)
{
    extern __shared__ double P[];                              // B rows of W = n - j0 doubles
    __shared__ double red[4], Dsh[GMW_RL_BMAX], mrow[GMW_RL_BMAX][GMW_RL_ROWS];
    const int tid = threadIdx.x, W = n - j0, Bn = min(B, W);
    const int c = j0 + Bn + blockIdx.x * 256 + tid, r0 = j0 + Bn + blockIdx.y * GMW_RL_ROWS;
    if (j0 + Bn + (int)blockIdx.x * 256 + 255 < r0) return;    // the whole tile lies under the diagonal
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    const double gamma = __longlong_as_double((long long)fs->gmax_bits);
    const double xi = __longlong_as_double((long long)fs->ximax_bits);
    const double nu = fmax(1.0, sqrt((double)n * n - 1.0));
    const double beta2 = fmax(fmax(gamma, xi / nu), 1e-15);
    for (int b = 0; b < Bn; b++)
        for (int q = tid; q < W; q += 256) P[b * W + q] = C[(size_t)(j0 + b) * ld + j0 + q];
    __syncthreads();
    for (int b = 0; b < Bn; b++) {
        double mx = 0.0;
        for (int q = b + 1 + tid; q < W; q += 256) mx = fmax(mx, fabs(P[b * W + q]));
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        const double th = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double cjj = fabs(P[b * W + b]);
        const double t2 = th * th / beta2;
        const double Db = fmax(fmax(eps, cjj), t2);           // SLAM.cpp:2279-2285
        if (tid == 0) {
            Dsh[b] = Db;
            if (first) { D[j0 + b] = Db; if (t2 > fmax(eps, cjj)) atomicAdd(&fs->clamp_rows, 1); }
        }
        for (int b2 = b + 1; b2 < Bn; b2++) {
            const double m = P[b * W + b2] / Db;
            for (int q = b2 + tid; q < W; q += 256) P[b2 * W + q] = P[b2 * W + q] - m * P[b * W + q];
        }
        __syncthreads();
    }
    if (first) {
        for (int b = 0; b < Bn; b++) {
            const double Db = Dsh[b], sq = sqrt(Db);
            for (int q = b + tid; q < W; q += 256) Sout[(size_t)(j0 + b) * ld + j0 + q] = (q == b) ? sq : sq * (P[b * W + q] / Db);
        }
    }
    if (tid < GMW_RL_ROWS) {
        const int r = r0 + tid;
        for (int b = 0; b < Bn; b++) mrow[b][tid] = (r < n) ? P[b * W + (r - j0)] / Dsh[b] : 0.0;
    }
    double cv[GMW_RL_BMAX];
#pragma unroll
    for (int b = 0; b < GMW_RL_BMAX; b++) cv[b] = (b < Bn && c < n) ? P[b * W + (c - j0)] : 0.0;
    __syncthreads();
    if (c >= n) return;
    for (int rr = 0; rr < GMW_RL_ROWS; rr++) {
        const int r = r0 + rr;
        if (r < n && c >= r) {
            double v = C[(size_t)r * ld + c];
#pragma unroll
            for (int b = 0; b < GMW_RL_BMAX; b++) if (b < Bn) v = v - mrow[b][rr] * cv[b];
            C[(size_t)r * ld + c] = v;
        }
    }
}
static bool g_exact_right_looking = true;
static int g_exact_rl_block = 0;                              // 0: as many pivots per launch as fit                     // srukf_debug_set(0, "exact_rl", 0): k_gmw_col (the left-looking form, one row per launch)
void srukf_set_exact_right_looking(int on) { g_exact_right_looking = on != 0; g_exact_rl_block = on > 1 ? std::min(on - 1, GMW_RL_BMAX) : 0; }
int srukf_get_exact_right_looking(void) { return g_exact_right_looking ? 1 : 0; }
// The first launch of a kernel with more than 64 KB of dynamic LDS costs ~75 ms once per stream (measured: the first flagged frame at N = 200 took 82 ms, the others 6.5).
// srukf_create pays it: an empty launch (n = 0: every thread leaves at once) with the largest panel the exact path asks for.
void srukf_warm_exact_path(hipStream_t st, FrameScalars* fs)
{
    // (per stream, not per process: the cost comes back on a stream — hardware queue — that has not run such a launch yet; on one that has, the empty launch is ~10 us)
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_gmw_rl, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    hipLaunchKernelGGL(k_gmw_rl, dim3(8, 64), dim3(256), 144 * 1024, st, 0, 0, 0, 1, 0.0, (double*)nullptr, (double*)nullptr, fs, (double*)nullptr);      // (every CU: 512 workgroups that leave at once)
}
void srukf_launch_gmw_col(hipStream_t st, int n, int ld, int j, double eps, const double* G, double* Wf, double* D,
                          unsigned long long* theta_bits, FrameScalars* fs, double* Sout)
{
    if (g_exact_right_looking) {
        // (callers loop j = 0 .. n - 1: the first call submits the whole sequence on a working copy of G, the others nothing)
        if (j != 0) return;
        hipMemcpyAsync(Wf, G, sizeof(double) * (size_t)ld * ld, hipMemcpyDeviceToDevice, st);
        // B pivots per launch: as many rows of n doubles as fit in ~144 KB of LDS, at most 8 (srukf_debug_set(0, "exact_rl", 1 + B) forces B: tests)
        int B = g_exact_rl_block > 0 ? g_exact_rl_block : (int)std::min<size_t>(GMW_RL_BMAX, (144 * 1024) / (sizeof(double) * (size_t)n));
        if (B < 1) B = 1;
        { static bool attr2 = false; if (!attr2) { (void)hipFuncSetAttribute((const void*)k_gmw_rl, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr2 = true; } }
        for (int q = 0; q < n; q += B) {
            const int Bn = std::min(B, n - q), rem = n - q - Bn;
            hipLaunchKernelGGL(k_gmw_rl, dim3(rem > 0 ? (rem + 255) / 256 : 1, rem > 0 ? (rem + GMW_RL_ROWS - 1) / GMW_RL_ROWS : 1), dim3(256), sizeof(double) * (size_t)Bn * (n - q), st,
                               n, ld, q, B, eps, Wf, D, fs, Sout);
        }
        return;
    }
    // (callers loop j = 0 .. n - 1; the launch behind the last pivot finishes its row)
    hipLaunchKernelGGL(k_gmw_col, dim3((ld - j + 255) / 256), dim3(256), sizeof(double) * (size_t)(j > 0 ? j : 1), st, n, ld, j, eps, G, Wf, D, theta_bits, fs, Sout);
    if (j == n - 1) hipLaunchKernelGGL(k_gmw_col, dim3((ld - n + 255) / 256 > 0 ? (ld - n + 255) / 256 : 1), dim3(256), sizeof(double), st, n, ld, n, eps, G, Wf, D, theta_bits, fs, Sout);
}
void srukf_launch_gmw_stats(hipStream_t st, int n, int ld, const double* G, FrameScalars* fs)
{
    hipLaunchKernelGGL(k_gmw_stats, dim3(n), dim3(256), 0, st, n, ld, G, fs);
}
}  // extern "C"
