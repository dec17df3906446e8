// srukf_tail.h — "tail fold" of the rank-aware replay: the NEXT frame's sigma points are projected inside the persistent factorisation
// launch of THIS frame (passSigmaThroughMesaurementFunction, SLAM.cpp:1615-1690, and the robot part of generateSigmaPoints /
// passSigmaThroughMotionFunction, 1148-1162, 1476-1532).  gfx950 only.
//
// Row i of the new sqrt factor is final long before the launch ends (its 64-row panel is factored, its trailing columns are scaled by
// the owners of the tiles below it), and everything else a sigma point of direction i needs is known early in the launch: the new
// mean X (the state-update jobs), the next frame's control (staged odometry).  The helper workgroups of the launch (head tiles,
// X += dX, dropped diagonal — srukf_gmw_persist.hip) therefore stay once those jobs are done and take TAIL JOBS from the same queue:
//     centre point; the five noise rows; the structurally null rows (their own landmark only, NullSkip in srukf_device.h);
//     then, panel by panel, the kept rows: (row a of the factor in permuted order, chunk of TAIL_LM landmarks), two lanes per
//     landmark (the + and the - point), after  sy->rows_done[a / 64] == T - a / 64  (every writer of that row panel has finished).
// Each job also leaves the robot part of its sigma points in the table sigR (what k_rank_expand did in "table" mode), so the frame
// tail (k_rank_expand) can run the next frame's motion reduction and the next frame starts with k_pxy2: the projection launch
// (11.6 us at N = 200) and its boundary are gone from the frame, ~4 us come back as the last panel's rows after the last pivot.
// Arithmetic: the same device functions as k_project_table / k_rank_expand (srukf_project_sigma, srukf_motion_point) on the same
// values — Z, DZ and the table are bit-identical to "table" mode.
#pragma once
#include "srukf_device.h"
#include "srukf_motion.h"
#include "srukf_rank.h"

#define TAIL_LM 100                  // landmarks per row job (200 of the 256 threads busy)
#define TAIL_SM_DOUBLES 40           // LDS: ctl[8], xr[4], centre row[8], the two rows of the current job [16], spare

struct TailArgs {
    int on;                          // 1: tail jobs.  Measurement only (srukf_debug_set "tail_fold"): 2 = the writers' protocol without any job, + 4 / + 8 = the
                                     // pivot's / the workers' rows with plain stores and no counting, + 16 = inside the "table" flow (k_project_table still runs)
    int N, n, Na, mp, r, nnull, nchunk;
    srukf_params p; KWeights w;
    const double* X;                 // the state the dX jobs of this launch complete
    double* sigR; double* Z; double* DZ;
    const int* perm; const int* iperm; const int* nulls;
};
__host__ __device__ __forceinline__ int tail_jobs_centre(const TailArgs& ta) { return (ta.N + 255) / 256; }
__host__ __device__ __forceinline__ int tail_jobs_noise(const TailArgs& ta) { return 5 * ta.nchunk; }
__host__ __device__ __forceinline__ int tail_jobs_null(const TailArgs& ta) { return (2 * ta.nnull + 255) / 256; }
__host__ __device__ __forceinline__ int tail_jobs_total(const TailArgs& ta)
{
    return (ta.on & 3) == 1 ? tail_jobs_centre(ta) + tail_jobs_noise(ta) + tail_jobs_null(ta) + ta.r * ta.nchunk : 0;
}

__device__ __forceinline__ bool tail_wait_count(const unsigned int* cnt, unsigned int want, const int* abort_flag)
{
    for (int spins = 0; spins < (1 << 16); spins++) {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= want) return true;
        if ((spins & 31) == 31 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

// per-helper state between jobs (registers, the same in every thread of the workgroup)
struct TailState { int ready_p; bool init; };

// One tail job (index q within the tail part of the queue).  Sp: the factor rows this launch writes (permuted order, agent-scope
// stores).  sm: TAIL_SM_DOUBLES of LDS that survive between the jobs of a helper.  Returns false when a bounded wait expired.
__device__ __forceinline__ bool tail_job(const TailArgs& ta, int q, const double* __restrict__ Sp, int ld, int T, GmwSync* sy, const FrameScalars* __restrict__ fs,
                                         unsigned int ndx, double* sm, double* __restrict__ ldx, double* __restrict__ lrow, int* okp, TailState& st, int tid)
{
    const int nC = tail_jobs_centre(ta), nN = tail_jobs_noise(ta), nU = tail_jobs_null(ta);
    const bool wv0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int need_p = (q >= nC + nN + nU) ? ((q - nC - nN - nU) / ta.nchunk) >> 6 : -1;
    if (!st.init || need_p > st.ready_p) {
        if (wv0) {
            bool good = st.init || tail_wait_count(&sy->dx_done, ndx, &sy->abort);
            if (good && need_p >= 0) good = tail_wait_count(&sy->rows_done[need_p], (unsigned)(T - need_p), &sy->abort);
            *okp = good ? 1 : 0;
        }
        __syncthreads();
        const bool good = *okp != 0;
        __syncthreads();
        if (!good) return false;
        asm volatile("" ::: "memory");                         // nothing below is read before the counters
        if (!st.init) {
            // once per helper: the next frame's control, the robot mean it starts from, the centre point's robot part
            if (tid == 0) {
                double ctl[8];
                srukf_control_values(fs, fs->frame + 1, ctl);
#pragma unroll
                for (int e = 0; e < 8; e++) sm[e] = ctl[e];
                double xr[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { xr[e] = ld_dev(ta.X + ta.n - 4 + e); sm[8 + e] = xr[e]; }
                const MotionCtl mc = { ctl[0], ctl[1], ctl[2], ctl[3], ctl[4] };
                double s0[4], c0s, s0s;
                srukf_motion_centre(mc, xr, s0, c0s, s0s);
                sm[12] = s0[0]; sm[13] = s0[1]; sm[14] = s0[2]; sm[15] = s0[3]; sm[16] = c0s; sm[17] = s0s; sm[18] = 0.0; sm[19] = 0.0;
            }
            // ... and the mean itself: every job of this helper reads its landmarks from LDS (one agent-scope pass over X per helper; six
            // such loads per sigma point made the X lines a hot spot of the whole launch: + 20 us on the pivot chain)
            for (int c = tid; c < ta.n; c += 256) ldx[c] = ld_dev(ta.X + c);
            __syncthreads();
            st.init = true;
        }
        if (need_p > st.ready_p) st.ready_p = need_p;
    }
    const int N = ta.N, n = ta.n, Na = ta.Na, mp = ta.mp;
    const double f1 = ta.p.cam_f / ta.p.cam_dx, f2 = ta.p.cam_f / ta.p.cam_dy;
    const double* cen = sm + 12;
    double* rows = sm + 20;                                    // [2][8]
    auto put_row = [&](double* dst, const double (&rr)[4], double c2, double s2) {
        double4* o = reinterpret_cast<double4*>(dst);
        o[0] = make_double4(rr[0], rr[1], rr[2], rr[3]); o[1] = make_double4(c2, s2, 0.0, 0.0);
    };
    if (q < nC) {                                              // ---- centre point (row 0 of Z, row 0 of the table) ----
        const int k = 256 * q + tid;
        if (q == 0 && tid == 0) { const double rr[4] = { cen[0], cen[1], cen[2], cen[3] }; put_row(ta.sigR, rr, cen[4], cen[5]); }
        if (k < N) {
            double base[6];
#pragma unroll
            for (int e = 0; e < 6; e++) base[e] = ldx[6 * k + e];
            double ox, oy;
            srukf_project(ta.p, f1, f2, base, cen[0], cen[1], cen[2], cen[4], cen[5], 0.0, 0.0, ox, oy);
            *reinterpret_cast<double2*>(ta.Z + 2 * k) = make_double2(ox, oy);
        }
        return true;
    }
    q -= nC;
    const double zero4[4] = { 0, 0, 0, 0 }, zero3[3] = { 0, 0, 0 };
    if (q >= nN && q < nN + nU) {                              // ---- structurally null rows: their own landmark, both points' robot part = the centre ----
        q -= nN;
        const int it = 128 * q + (tid >> 1), sg = tid & 1;
        const bool act = it < ta.nnull;
        const int i = act ? ta.nulls[it] : 0, k = i / 6;
        double ox = 0.0, oy = 0.0;
        if (act) {
            double base[6], dev[6];
#pragma unroll
            for (int e = 0; e < 6; e++) { base[e] = ldx[6 * k + e]; dev[e] = (6 * k + e == i) ? sqrt(ta.p.epsilon) : 0.0; }
            srukf_project_sigma(ta.p, f1, f2, base, dev, 0.0, 0.0, sg ? -ta.w.gamma : ta.w.gamma, cen, ox, oy);
            const int c = 1 + sg * Na + i;
            *reinterpret_cast<double2*>(ta.Z + (size_t)c * mp + 2 * k) = make_double2(ox, oy);
            const double rr[4] = { cen[0], cen[1], cen[2], cen[3] };
            put_row(ta.sigR + (size_t)c * 8, rr, cen[4], cen[5]);
        }
        const double pox = __shfl_xor(ox, 1), poy = __shfl_xor(oy, 1);
        if (act && sg == 0) *reinterpret_cast<double2*>(ta.DZ + (size_t)ta.iperm[i] * mp + 2 * k) = make_double2(ox - pox, oy - poy);
        return true;
    }
    // ---- a full direction: noise row n + q / nchunk, or kept row a of the factor ----
    const bool noise = q < nN;
    if (!noise) q -= nN + nU;
    const int a = q / ta.nchunk, hch = q - a * ta.nchunk;       // noise: a = 0 .. 4
    const int i = noise ? n + a : ta.perm[a];
    // the row of the factor goes through LDS: one coalesced pass (columns >= a; what stands left of the diagonal is never used) instead of
    // six scattered loads per sigma point.  Plain loads: the row was written with write-through stores before its panel's counter, and
    // nothing of it was read in this launch before — no cache can hold an older copy (the panel buffer is read the same way).
    if (!noise) {
        const double* src = Sp + (size_t)a * ld;
        for (int b = (a & ~1) + 2 * tid; b < ld; b += 512) *reinterpret_cast<double2*>(lrow + b) = *reinterpret_cast<const double2*>(src + b);
        __syncthreads();
    }
    if (tid < 2) {
        const MotionCtl mc = { sm[0], sm[1], sm[2], sm[3], sm[4] };
        const double xr[4] = { sm[8], sm[9], sm[10], sm[11] };
        double srow[4] = { 0, 0, 0, 0 }, mnoise[3] = { 0, 0, 0 };
        if (noise) { if (a < 3) mnoise[a] = sm[5 + a]; }
        else {
#pragma unroll
            for (int e = 0; e < 4; e++) srow[e] = (ta.r - 4 + e >= a) ? lrow[ta.r - 4 + e] : 0.0;     // the robot columns: permuted positions r-4 .. r-1
        }
        double rr[4], c2, s2;
        srukf_motion_point(mc, xr, srow, mnoise, tid ? -ta.w.gamma : ta.w.gamma, rr, c2, s2);
        put_row(rows + 8 * tid, rr, c2, s2);
        if (hch == 0) put_row(ta.sigR + (size_t)(1 + tid * Na + i) * 8, rr, c2, s2);
    }
    __syncthreads();
    {
        const int k = TAIL_LM * hch + (tid >> 1), sg = tid & 1;
        const bool act = tid < 2 * TAIL_LM && k < N;
        double ox = 0.0, oy = 0.0;
        if (act) {
            double base[6], dev[6];
#pragma unroll
            for (int e = 0; e < 6; e++) {
                const int col = 6 * k + e;
                base[e] = ldx[col];
                dev[e] = (!noise && col >= i) ? lrow[ta.iperm[col]] : 0.0;
            }
            const double e0 = (i == n + 3) ? ta.p.sigma_measure : 0.0, e1 = (i == n + 4) ? ta.p.sigma_measure : 0.0;
            srukf_project_sigma(ta.p, f1, f2, base, dev, e0, e1, sg ? -ta.w.gamma : ta.w.gamma, rows + 8 * sg, ox, oy);
            *reinterpret_cast<double2*>(ta.Z + (size_t)(1 + sg * Na + i) * mp + 2 * k) = make_double2(ox, oy);
        }
        const double pox = __shfl_xor(ox, 1), poy = __shfl_xor(oy, 1);
        if (act && sg == 0 && !noise) *reinterpret_cast<double2*>(ta.DZ + (size_t)a * mp + 2 * k) = make_double2(ox - pox, oy - poy);   // row dzperm[i] = a
    }
    __syncthreads();                                           // the job's two rows in LDS may be rewritten
    return true;
}
