// srukf_meas.h — measurement statistics jobs (device functions shared by the stand-alone kernels
// k_meas_partial / k_meas_final and by the extra workgroups of the k_pxy launch).  See srukf_predict.hip
// for the definition of the 13 sums.
#pragma once
#include "srukf_device.h"

#define MEAS_SLICES 16
#define MEAS_NS 13
#define MEAS_CHUNK 10                                  // rows of Z per thread and chunk: all of them in flight at once
#define MEAS_SM_DOUBLES (8 * 32 * MEAS_NS + 8 * MEAS_CHUNK * 4)
// one workgroup: 32 landmarks (bx) x one of MEAS_SLICES row slices (by); smem: MEAS_SM_DOUBLES doubles of LDS
// xrob: the robot mean AFTER the motion step (X + n - 4, or fs->Xr1 in the replay path where X still holds the mean before it)
// ns.rows != null (NullSkip, srukf_device.h): the job's rows are a slice of that list instead of a slice of 0 .. L-1, and slice 0 adds,
// for each of its landmarks, the six rows of the landmark's own structurally null directions (all other null rows are exact zeros).
template <bool COHERENT>
__device__ __forceinline__ void meas_partial_job(const KDims& d, const KWeights& w, const double* __restrict__ xrob,
                                                 const double* __restrict__ sigR, const double* __restrict__ Z,
                                                 double* __restrict__ part /* [MEAS_SLICES][MEAS_NS][mp/2] */, int bx, int by, double* smem,
                                                 const NullSkip ns = NullSkip{})
{
    double (*sm)[32][MEAS_NS] = (double (*)[32][MEAS_NS])smem;
    const int lx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int k = bx * 32 + lx;
    const int kk = (k < d.N) ? k : 0;
    const int L = ns.rows ? ns.nrows : d.L, mp = d.mp;          // (list positions when a row list is given)
    const int rows = (L + MEAS_SLICES - 1) / MEAS_SLICES;
    const int c_beg = by * rows, c_end = min(L, c_beg + rows);
    const double2 z0 = *reinterpret_cast<const double2*>(Z + 2 * kk);
    double xr[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xr[e] = xrob[e];
    double s[MEAS_NS];
#pragma unroll
    for (int q = 0; q < MEAS_NS; q++) s[q] = 0.0;
    // The job walks its rows in chunks of 8 x MEAS_CHUNK: the robot parts of the chunk's sigma points (32 B each, the same for all
    // 32 landmarks) are staged in LDS by one coalesced pass, then every thread requests ALL its MEAS_CHUNK rows of Z before it
    // touches the first — one memory round trip per chunk (one row per trip was a round trip per row: 9.8 us per job, the long
    // pole of the launch the statistics ride on).
    double (*rs)[4] = (double (*)[4])(smem + 8 * 32 * MEAS_NS);
    // slice 0 also takes the landmark's own structurally null directions (below, behind the chunks: the order of the sums is what it was).  Their loads go out HERE, with
    // the first chunk's — the row index does not depend on the test, only the decision does: behind the loop they were two more dependent round trips on the job every
    // group's final pass waits for, and the statistics are the long pole of the launch they ride on (24 of its 25 us; scripts/pxy2_stamps.py)
    const bool own = ns.rows && by == 0 && sl < 6 && k < d.N && 6 * k + (sl >> 1) >= 2;
    int own_ip = 0; double2 own_z = make_double2(0.0, 0.0); double4 own_r = make_double4(0.0, 0.0, 0.0, 0.0);
    if (own) {
        const int i = 6 * k + (sl >> 1), c = 1 + (sl & 1) * d.Na + i;
        own_ip = ns.iperm[i];
        own_z = *reinterpret_cast<const double2*>(Z + (size_t)c * mp + 2 * k);
        own_r = *reinterpret_cast<const double4*>(sigR + (size_t)c * 8);
    }
    for (int cb = c_beg; cb < c_end; cb += 8 * MEAS_CHUNK) {
        const int cn = min(8 * MEAS_CHUNK, c_end - cb);
        __syncthreads();
        // Two memory round trips per chunk — every row index this thread needs first (list mode), then every value —, in that order in the program: written as a
        // staging loop (index -> value -> LDS, twice) followed by the Z rows (index -> value) it was SIX dependent round trips, 13 of the launch's 25 us (scripts/pxy2_stamps.py)
        static_assert(8 * MEAS_CHUNK * 4 <= 2 * 256, "two staging elements per thread");
        int ce[2], cz[MEAS_CHUNK];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = threadIdx.x + 256 * j, cp = cb + (e >> 2);
            ce[j] = (e < cn * 4) ? (ns.rows ? ns.rows[cp] : cp) : -1;
        }
#pragma unroll
        for (int u = 0; u < MEAS_CHUNK; u++) {
            const int cp = min(cb + sl + 8 * u, c_end - 1);
            cz[u] = ns.rows ? ns.rows[cp] : cp;
        }
        double rv[2];
#pragma unroll
        for (int j = 0; j < 2; j++) rv[j] = (ce[j] >= 0) ? sigR[(size_t)ce[j] * 8 + (threadIdx.x & 3)] : 0.0;
        double2 zz[MEAS_CHUNK];
#pragma unroll
        for (int u = 0; u < MEAS_CHUNK; u++) zz[u] = *reinterpret_cast<const double2*>(Z + (size_t)cz[u] * mp + 2 * kk);
#pragma unroll
        for (int j = 0; j < 2; j++) { const int e = threadIdx.x + 256 * j; if (ce[j] >= 0) rs[e >> 2][e & 3] = rv[j]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MEAS_CHUNK; u++) {
            const int c = cb + sl + 8 * u;
            if (c < c_end) {
                const double2 z = zz[u];
                const double* r = rs[sl + 8 * u];
                const double dx = z.x - z0.x, dy = z.y - z0.y;
                const double wt = (c == 0) ? w.wc0 : w.wi;                 // (list position 0 is row 0 as well)
                s[0] += dx; s[1] += dy;
                const double a = w.wi_sr * dx, b = w.wi_sr * dy;
                s[2] += a * a; s[3] += a * b; s[4] += b * b;
#pragma unroll
                for (int e = 0; e < 4; e++) { const double dr = wt * (r[e] - xr[e]); s[5 + e] += dr * dx; s[9 + e] += dr * dy; }
            }
        }
    }
    if (own) {
        // the landmark's own null directions (anchor coordinates 6 k + e): sub-slice sl takes (e, sign) = (sl >> 1, sl & 1)
        if (own_ip >= ns.r) {
            const double2 z = own_z;
            const double4 rr = own_r;
            const double r[4] = { rr.x, rr.y, rr.z, rr.w };
            const double dx = z.x - z0.x, dy = z.y - z0.y;
            s[0] += dx; s[1] += dy;
            const double a = w.wi_sr * dx, b = w.wi_sr * dy;
            s[2] += a * a; s[3] += a * b; s[4] += b * b;
#pragma unroll
            for (int e = 0; e < 4; e++) { const double dr = w.wi * (r[e] - xr[e]); s[5 + e] += dr * dx; s[9 + e] += dr * dy; }
        }
    }
#pragma unroll
    for (int q = 0; q < MEAS_NS; q++) sm[sl][lx][q] = s[q];
    __syncthreads();
    const int half = mp / 2;
    for (int e = threadIdx.x; e < 32 * MEAS_NS; e += 256) {
        const int q = e / 32, l2 = e % 32;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) t += sm[u][l2][q];
        const int k2 = bx * 32 + l2;
        if (k2 < d.N) {
            double* dst = &part[((size_t)by * MEAS_NS + q) * half + k2];
            // COHERENT: device-scope store (write-through, sc1) so that the last workgroup of the same launch can read it
            // without a release fence — a fence would write back the whole L2, where the k_pxy tiles are writing Ut
            if (COHERENT) __hip_atomic_store(dst, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = t;
        }
    }
}

// The MEAS_SLICES partial sums of one value of landmark k in the fixed order both forms of the final pass use: slice pairs first,
// then the eight pair sums one after the other.
template <bool COHERENT>
__device__ __forceinline__ double meas_slice_pair(const double* __restrict__ part, int half, int k, int q, int pr)
{
    const double* s0 = &part[((size_t)(2 * pr) * MEAS_NS + q) * half + k];
    const double* s1 = &part[((size_t)(2 * pr + 1) * MEAS_NS + q) * half + k];
    if (COHERENT) return __hip_atomic_load(s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *s0 + *s1;
}
// landmark k, given the reduced sums t[]: h, Si, visible, PxyR
// pre (may be null): rows 0, 1, 2 of Z at the landmark's columns, requested by the caller before it learnt that the final pass is its to run (one memory round trip less
// behind the last partial sum); out (may be null): what was stored to h / Si / visible, for a caller that passes it on (the step-wise API's host mirror)
struct MeasPre { double2 z[3]; };
struct MeasOut { double h[2]; double si[4]; int vis; };
__device__ __forceinline__ void meas_final_tail(const KDims& d, const KWeights& w, const double* __restrict__ sigR,
                                                const double* __restrict__ Z, const double (&t)[MEAS_NS],
                                                double* __restrict__ h, double* __restrict__ Si, int* __restrict__ vis,
                                                double* __restrict__ PxyR, int k, int defer = 0, const MeasPre* pre = nullptr, MeasOut* out = nullptr, int devst = 0);
// landmark k: reduce the slices, finish h, Si, visible, PxyR
template <bool COHERENT>
__device__ __forceinline__ void meas_final_one(const KDims& d, const KWeights& w, const double* __restrict__ X, const double* __restrict__ sigR,
                                               const double* __restrict__ Z, const double* __restrict__ part,
                                               double* __restrict__ h, double* __restrict__ Si, int* __restrict__ vis,
                                               double* __restrict__ PxyR, int k)
{
    static_assert(MEAS_SLICES == 16, "eight slice pairs");
    const int half = d.mp / 2;
    double t[MEAS_NS];
#pragma unroll
    for (int q = 0; q < MEAS_NS; q++) {
        double acc = 0.0;
        for (int pr = 0; pr < MEAS_SLICES / 2; pr++) acc += meas_slice_pair<COHERENT>(part, half, k, q, pr);
        t[q] = acc;
    }
    meas_final_tail(d, w, sigR, Z, t, h, Si, vis, PxyR, k);
}
// One landmark group (32 landmarks) by a whole workgroup of 256 threads: thread (landmark l, pair p) sums its slice pair of all
// 13 values, the pair sums meet in LDS (sm: MEAS_SM_DOUBLES doubles), 32 threads finish.  Same summation order as meas_final_one.
__device__ __forceinline__ void meas_final_group(const KDims& d, const KWeights& w, const double* __restrict__ sigR,
                                                 const double* __restrict__ Z, const double* __restrict__ part,
                                                 double* __restrict__ h, double* __restrict__ Si, int* __restrict__ vis,
                                                 double* __restrict__ PxyR, int bx, double* smem, int defer = 0, const MeasPre* pre = nullptr, MeasOut* out = nullptr, int devst = 0)
{
    double (*sm)[32][MEAS_NS] = (double (*)[32][MEAS_NS])smem;
    const int lx = threadIdx.x & 31, pr = threadIdx.x >> 5;
    const int k = bx * 32 + lx, half = d.mp / 2;
    __syncthreads();                                           // the partial job's use of the scratch is over
    if (k < d.N) {
        // all 26 device-scope loads go out before the first sum (the compiler keeps atomic loads where they are written: pair by pair it was 13 dependent round trips,
        // 5.8 us of the launch's longest chain)
        double p0[MEAS_NS], p1[MEAS_NS];
#pragma unroll
        for (int q = 0; q < MEAS_NS; q++) {
            p0[q] = __hip_atomic_load(&part[((size_t)(2 * pr) * MEAS_NS + q) * half + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p1[q] = __hip_atomic_load(&part[((size_t)(2 * pr + 1) * MEAS_NS + q) * half + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < MEAS_NS; q++) sm[pr][lx][q] = p0[q] + p1[q];        // (meas_slice_pair's sum)
    }
    __syncthreads();
    if (threadIdx.x < 32 && k < d.N) {
        double t[MEAS_NS];
#pragma unroll
        for (int q = 0; q < MEAS_NS; q++) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++) acc += sm[u][lx][q];
            t[q] = acc;
        }
        meas_final_tail(d, w, sigR, Z, t, h, Si, vis, PxyR, k, defer, pre, out, devst);
    }
}
__device__ __forceinline__ void meas_final_tail(const KDims& d, const KWeights& w, const double* __restrict__ sigR,
                                                const double* __restrict__ Z, const double (&t)[MEAS_NS],
                                                double* __restrict__ h, double* __restrict__ Si, int* __restrict__ vis,
                                                double* __restrict__ PxyR, int k, int defer, const MeasPre* pre, MeasOut* out, int devst)
{
#pragma clang fp contract(off)
    // devst (the gain fold of k_pxy2): the results are read by other workgroups of the SAME launch — write-through stores (a release fence instead would write the whole
    // L2 back: 2 us on the launch's longest chain)
    auto stv = [&](double* q, double v) { if (devst) st_dev(q, v); else *q = v; };
    // (every fused multiply-add written out, contraction off: this function is compiled into three kernels — k_meas_final, k_pxy, k_pxy2 — and which of two
    //  products the compiler fuses depends on the code around an inlined call; the step-wise API and the replay must give the same bits)
    const int mp = d.mp;
    const double2 z0 = pre ? pre->z[0] : *reinterpret_cast<const double2*>(Z + 2 * k);
    // h = wm0*Z0 + wi*sum_{c>=1} Z_c = Z0*(wm0 + 2Na*wi) + wi*sum (Z_c - Z0)      (SLAM.cpp:1678-1681)
    const double wsum = w.wm0 + 2.0 * d.Na * w.wi;
    const double hx = fma(wsum, z0.x, w.wi * t[0]), hy = fma(wsum, z0.y, w.wi * t[1]);
    // robot rows of Pxy: sum_c w_c (r_c - xr)(Z_c - h) = sum_c w_c (r_c - xr)(Z_c - Z0) - (h - Z0) * sum_c w_c (r_c - xr)
    const bool v = (hx != 0.0) && (hy != 0.0);
    stv(&h[2 * k], hx); stv(&h[2 * k + 1], hy);
    if (devst) __hip_atomic_store(&vis[k], v ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else vis[k] = v ? 1 : 0;
    if (defer) {
        // "fused tail" mode: the motion reduction of this frame runs in the same launch, so neither the mean xr nor rs exist yet.  The sums were
        // taken around the centre point's robot part r_0 instead of xr; with sum_c w_c (Z_c - Z_0) = wi t[0..1] (the centre's own term is zero)
        //     sum_c w_c (r_c - xr)(Z_c - Z_0) = sum_c w_c (r_c - r_0)(Z_c - Z_0) - (xr - r_0) wi t[0..1]
        // and k_gain applies that and the (h - Z_0) rs term: raw sums here, wi t[0..1] in row 4
#pragma unroll
        for (int e = 0; e < 4; e++) { stv(&PxyR[(size_t)e * mp + 2 * k], t[5 + e]); stv(&PxyR[(size_t)e * mp + 2 * k + 1], t[9 + e]); }
        stv(&PxyR[(size_t)4 * mp + 2 * k], w.wi * t[0]); stv(&PxyR[(size_t)4 * mp + 2 * k + 1], w.wi * t[1]);
    } else {
        double rs[4];
#pragma unroll
        for (int e = 0; e < 4; e++) rs[e] = sigR[(size_t)d.L * 8 + e];      // sum_c w_c (r_c - xr), from k_motion
#pragma unroll
        for (int e = 0; e < 4; e++) {
            PxyR[(size_t)e * mp + 2 * k]     = fma(-(hx - z0.x), rs[e], t[5 + e]);
            PxyR[(size_t)e * mp + 2 * k + 1] = fma(-(hy - z0.y), rs[e], t[9 + e]);
        }
    }
    // Householder R of the 2Na x 2 matrix [a b] (GSL: beta = -sign(alpha) hypot(alpha, xnorm); tau = 0 if xnorm == 0)
    const double2 z1 = pre ? pre->z[1] : *reinterpret_cast<const double2*>(Z + (size_t)1 * mp + 2 * k);
    const double2 z2 = pre ? pre->z[2] : *reinterpret_cast<const double2*>(Z + (size_t)2 * mp + 2 * k);
    const double a0 = w.wi_sr * (z1.x - z0.x), b0 = w.wi_sr * (z1.y - z0.y);
    const double a1 = w.wi_sr * (z2.x - z0.x), b1 = w.wi_sr * (z2.y - z0.y);
    const double saa = t[2], sab = t[3], sbb = t[4];
    const double xn2 = fmax(fma(-a0, a0, saa), 0.0);
    double R00 = a0, R01 = b0, tau = 0.0, wv = 0.0, inv_s = 0.0;
    if (xn2 > 0.0) {
        const double beta = -(a0 >= 0.0 ? 1.0 : -1.0) * sqrt(saa);
        tau = (beta - a0) / beta;
        inv_s = 1.0 / (a0 - beta);
        wv = fma(fma(-a0, b0, sab), inv_s, b0);         // w = B_0 + sum_{r>=1} B_r v_r
        R00 = beta;
        R01 = fma(-tau, wv, b0);
    }
    // second column: b' = H1 b; |b'[1:]|^2 = |b|^2 - R01^2 (H1 orthogonal); R11 = -sign(b'_1) |b'[1:]|,
    // or b'_1 itself when the rest of the sub-column is zero
    const double bp1 = fma(-(tau * (a1 * inv_s)), wv, b1);
    const double nrm2 = fmax(fma(-R01, R01, sbb), 0.0);
    const double rest2 = fma(-bp1, bp1, nrm2);
    double R11 = bp1;
    if (rest2 > 0.0) R11 = -(bp1 >= 0.0 ? 1.0 : -1.0) * sqrt(nrm2);
    stv(&Si[4 * k + 0], v ? R00 : 0.0); stv(&Si[4 * k + 1], v ? R01 : 0.0); stv(&Si[4 * k + 2], 0.0); stv(&Si[4 * k + 3], v ? R11 : 0.0);
    if (out) { out->h[0] = hx; out->h[1] = hy; out->si[0] = v ? R00 : 0.0; out->si[1] = v ? R01 : 0.0; out->si[2] = 0.0; out->si[3] = v ? R11 : 0.0; out->vis = v ? 1 : 0; }
}

