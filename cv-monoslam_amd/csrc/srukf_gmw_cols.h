// srukf_gmw_cols.h — the 32-pivot chain of the GMW factorisation (one 32x32 diagonal block) in
// COLUMN layout on the vector ALU.  gfx950 only.
//
// Why not MFMA here: on MI355X the FP64 matrix pipe has the same peak as the FP64 vector pipe (one
// v_mfma_f64_16x16x4 = 1024 FMAs in 64 cycles = one v_fma_f64 per 4 cycles), and a dependent MFMA
// costs 64-92 cycles against 4.8 for a dependent v_fma_f64 (scripts/mb/mb_lat.hip).  The pivot chain
// is latency bound, so it runs as plain FMAs; what made a VALU version slow before — two v_readlane
// per multiplier — is replaced by DPP row_newbcast operands, one instruction per row update:
//
//   lane c16 of EVERY 16-lane DPP row holds columns c16 (P[r] = C[r][c16], r < 16) and 16 + c16
//   (Q[r] = C[r][16 + c16], r < 32) of the block — row_newbcast only reaches the own DPP row.
//   pivot j:  xp,xq = P[j], Q[j]                      the pivot row entries of this lane's two columns
//             d   = xp|xq[lane j & 15]  (v_mov_b64_dpp row_newbcast),  D = max(eps, |d|),  rc = 1/D
//             ntp,ntq = -xp * rc, -xq * rc             = -L[j][c]
//             P[r] += xp[lane r] * ntp, Q[r] += (xp|xq)[lane r & 15] * ntq   for r > j   (v_fmac_f64_dpp)
//   i.e. C[r][c] -= C[j][r] * C[j][c] / D_j, the reference's recurrence (SLAM.cpp:2246-2262) with
//   D_j = max(EPSILON, |C_jj|) (the theta clamp is verified afterwards by k_gmw_check).
//
// The pivot wave publishes row j of -L (and D_j, which doubles as the "row j ready" flag) in LDS as
// soon as it exists.  A second wave follows one pivot behind and builds T = (I + M^T)^{-1} = L^{-1}
// (column c' per lane, multipliers as uniform LDS reads), so the panel "TRSM" of the next step stays
// an MFMA product.  After a barrier every wave of the workgroup helps to write S rows / D outputs.
#pragma once
#include "srukf_device.h"

struct GmwPanel;

template <int N> __device__ __forceinline__ void fmac_bcast16(double& acc, double src, double nt)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(nt), "n"(N));
}
// the s_nop covers the "VALU writes VGPR -> DPP reads it" hazard (2 wait states), which the compiler
// cannot see inside inline asm; src is produced right before this instruction on the pivot chain
template <int N> __device__ __forceinline__ double mov_bcast16(double src)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(N));
    return r;
}

// ---- LDS workspace ---------------------------------------------------------------------------
// Xm: the block to factor (row-major, upper triangle valid).
// Lm: row J of -L as published by the pivot wave, one 528-byte strip per lane c16 (conflict-free for
//     128-bit accesses): strip[2J] = -L[J][c16], strip[2J+1] = -L[J][16 + c16].
// Dv: D_J; zero before the factorisation starts, D_J > 0 afterwards — doubles as the "row J published" flag.
#define GMW_LM_STRIDE 66                       // doubles per lane strip
#define GMW_XM_DOUBLES 1024
#define GMW_LM_DOUBLES (16 * GMW_LM_STRIDE)
#define GMW_FAC_DOUBLES (GMW_XM_DOUBLES + GMW_LM_DOUBLES + 32)
struct GmwColsLds {                            // pointers, so that a kernel can alias LDS arrays it no longer needs
    double (*Xm)[32];
    double* Lm;
    double* Dv;
};
// carve the three arrays out of one LDS region of >= GMW_XM_DOUBLES + GMW_LM_DOUBLES + 32 doubles
__device__ __forceinline__ GmwColsLds gmw_cols_carve(double* region)
{
    GmwColsLds w;
    w.Xm = (double (*)[32])region; w.Lm = region + GMW_XM_DOUBLES; w.Dv = region + GMW_XM_DOUBLES + GMW_LM_DOUBLES;
    return w;
}
__device__ __forceinline__ double gmw_lm(const double* Lm, int j, int c) { return Lm[(c & 15) * GMW_LM_STRIDE + 2 * j + (c >> 4)]; }

typedef double d2 __attribute__((ext_vector_type(2)));
typedef volatile __attribute__((address_space(3))) double lds_vdouble;
typedef volatile __attribute__((address_space(3))) d2 lds_vd2;
__device__ __forceinline__ unsigned lds_off(const volatile void* p)
{
    return (unsigned)(__UINTPTR_TYPE__)(const volatile __attribute__((address_space(3))) void*)p;
}
// Publishing stores: asm volatile statements keep their order, and LDS executes one wave's accesses in
// issue order, so "row, then D" needs no s_waitcnt on the pivot chain.
template <int OFF> __device__ __forceinline__ void lds_publish(unsigned off, double v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(off), "v"(v), "n"(OFF) : "memory");
}
template <int O0, int O1> __device__ __forceinline__ void lds_publish2(unsigned off, double v0, double v1)
{
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(off), "v"(v0), "v"(v1), "n"(O0), "n"(O1) : "memory");
}

// 1/D on the pivot chain: v_rcp_f64 (relative error < 2^-26, measured 1.5e-8) and ONE Newton step:
// e = 1 - D r is exact to the last bit in fma arithmetic, r (1 + e) then carries e^2 < 2^-52 plus one rounding, i.e.
// the multipliers L = C * (1/D) are within ~2 ulp of the reference's IEEE quotients C/D — three dependent
// instructions on the chain instead of the ~12 of a division (a second step would add 16 cycles to each of the 32 pivots).
__device__ __forceinline__ double gmw_pivot_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// rows R..RE-1 of one register set: acc[r] += src[lane r & 15 of this DPP row] * nt
template <int R, int RE, int NA> struct GmwRowUpd {
    static __device__ __forceinline__ void run(double (&acc)[NA], double src, double nt)
    {
        if constexpr (R < RE) {
            fmac_bcast16<(R & 15)>(acc[R], src, nt);
            GmwRowUpd<R + 1, RE, NA>::run(acc, src, nt);
        }
    }
};

// Pivot J.  Every DPP row of 16 lanes holds the whole block: lane c16 owns column c16 (P[r], rows 0..15 — the
// lower-left 16x16 quarter is never needed) and column 16 + c16 (Q[r], rows 0..31), so the pivot row is
// available to row_newbcast in every DPP row without any cross-row traffic.
// Software-pipelined by one pivot: step J arrives with D_J and -1 / D_J already computed, updates ROW J + 1 first, starts pivot J + 1's chain (broadcast of the new
// diagonal, clamp, reciprocal + Newton step: ~90 cycles of dependent instructions) and only then issues the other updates of pivot J, which fill those cycles — a
// lone wave issues in order, so what is not interleaved in the instruction stream is not overlapped (the compiler keeps this order; it kept the serial one too:
// chain, then 16 - 46 updates, per pivot).  Same operations on the same operands: bit-identical.
// The updates of pivot J other than row J + 1, as one list with a compile-time index (so that the chain of pivot J + 1 can be threaded through it):
//   J < 16:  P rows J+2..15 (src xp, nt ntp), Q rows J+2..15 (src xp, nt ntq), Q rows 16..31 (src xq, nt ntq; row 16 is "row J + 1" when J = 15)
//   J >= 16: Q rows J+2..31 (src xq, nt ntq)
template <int J> struct GmwOps {
    static constexpr int NP = (J < 16) ? (14 - J > 0 ? 14 - J : 0) : 0;          // P rows J+2..15, and as many Q rows
    static constexpr int Q16 = (J < 16) ? (J == 15 ? 15 : 16) : 0;               // Q rows 16..31 (17..31 when J = 15)
    static constexpr int NQ = (J >= 16) ? (30 - J > 0 ? 30 - J : 0) : 0;         // Q rows J+2..31
    static constexpr int N = (J < 16) ? 2 * NP + Q16 : NQ;
    template <int I> static __device__ __forceinline__ void one(double (&P)[16], double (&Q)[32], double xp, double xq, double ntp, double ntq)
    {
        if constexpr (J < 16) {
            if constexpr (I < NP) fmac_bcast16<((J + 2 + I) & 15)>(P[J + 2 + I], xp, ntp);
            else if constexpr (I < 2 * NP) fmac_bcast16<((J + 2 + I - NP) & 15)>(Q[J + 2 + I - NP], xp, ntq);
            else { constexpr int r = 32 - Q16 + (I - 2 * NP); fmac_bcast16<(r & 15)>(Q[r], xq, ntq); }
        } else {
            constexpr int r = J + 2 + I;
            fmac_bcast16<(r & 15)>(Q[r], xq, ntq);
        }
    }
    // ops [I0, I1) clipped to the list
    template <int I0, int I1> static __device__ __forceinline__ void run(double (&P)[16], double (&Q)[32], double xp, double xq, double ntp, double ntq)
    {
        if constexpr (I0 < I1 && I0 < N) {
            one<I0>(P, Q, xp, xq, ntp, ntq);
            run<I0 + 1, I1>(P, Q, xp, xq, ntp, ntq);
        }
    }
};
template <int J> struct GmwPivot {
    // lm: LDS offset of this lane's Lm strip, dv: LDS offset of Dv[0]; D, nrc: pivot J and -1 / pivot J
    static __device__ __forceinline__ void run(double (&P)[16], double (&Q)[32], double eps, unsigned lm, unsigned dv, double D, double nrc)
    {
        if constexpr (J < 32) {
            const double xq = Q[J];
            double xp = 0.0;
            if constexpr (J < 16) xp = P[J];
            const double ntq = xq * nrc;
            double ntp = 0.0;
            if constexpr (J < 16) {
                ntp = xp * nrc;
                lds_publish2<2 * J, 2 * J + 1>(lm, ntp, ntq);       // row J of -L (entries c <= J are not used)
            } else {
                lds_publish<16 * J + 8>(lm, ntq);
            }
            lds_publish<J * 8>(dv, D);
            double D1 = 0.0, nrc1 = 0.0;
            using Ops = GmwOps<J>;
            if constexpr (J + 1 < 32) {
                // row J + 1, then the next pivot's chain, one link at a time with other updates of pivot J between the links (scheduling barriers: the compiler
                // otherwise sinks the whole chain behind the updates again)
                if constexpr (J + 1 < 16) { fmac_bcast16<((J + 1) & 15)>(P[J + 1], xp, ntp); fmac_bcast16<((J + 1) & 15)>(Q[J + 1], xp, ntq); }
                else fmac_bcast16<((J + 1) & 15)>(Q[J + 1], xq, ntq);
                Ops::template run<0, 2>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                const double d1 = mov_bcast16<((J + 1) & 15)>(J + 1 < 16 ? P[(J + 1) & 15] : Q[J + 1 < 32 ? J + 1 : 0]);
                Ops::template run<2, 4>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                asm("v_max_f64 %0, %1, |%2|" : "=v"(D1) : "v"(eps), "v"(d1));     // NaN-proof: max(eps, NaN) = eps
                Ops::template run<4, 6>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                double r = __builtin_amdgcn_rcp(D1);                // gmw_pivot_rcp's three links
                Ops::template run<6, 10>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                const double e = fma(-D1, r, 1.0);
                Ops::template run<10, 12>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                r = fma(e, r, r);
                Ops::template run<12, 14>(P, Q, xp, xq, ntp, ntq);
                __builtin_amdgcn_sched_barrier(0);
                nrc1 = -r;
                Ops::template run<14, 64>(P, Q, xp, xq, ntp, ntq);
            }
            GmwPivot<J + 1>::run(P, Q, eps, lm, dv, D1, nrc1);
        }
    }
};

// Pivot wave.
__device__ __forceinline__ void gmw_cols_pivot_wave(const GmwColsLds& w, double eps, int lane)
{
    const int c = lane & 15;
    double P[16], Q[32];
#pragma unroll
    for (int r = 0; r < 16; r++) P[r] = w.Xm[r][c];
#pragma unroll
    for (int r = 0; r < 32; r++) Q[r] = w.Xm[r][16 + c];
    double D0, nrc0;
    {
        const double d0 = mov_bcast16<0>(P[0]);
        asm("v_max_f64 %0, %1, |%2|" : "=v"(D0) : "v"(eps), "v"(d0));
        nrc0 = -gmw_pivot_rcp(D0);
    }
    GmwPivot<0>::run(P, Q, eps, lds_off(&w.Lm[c * GMW_LM_STRIDE]), lds_off(w.Dv), D0, nrc0);
}

// ---- follower waves ----------------------------------------------------------------------------
// Both followers consume the published rows in groups (8, 8, 8, 4, rest): one wave-uniform poll per group on
// the D of its last row, then ordinary LDS loads, so the compiler pipelines them; a follower that polled on
// every row would pay an LDS round trip per pivot and fall behind the pivot wave.
#define GMW_POLL_LIMIT (1 << 22)               // a stuck pivot wave must never hang the GPU: give up (k_gmw_check then flags the frame)
__device__ __forceinline__ void gmw_wait_row(unsigned dv, int j)
{
    int spins = 0;
    double f;
    // (explicit ds_read: a volatile C++ load would make the compiler drain all outstanding global stores first)
    // (the s_sleep of the retries keeps three polling waves from crowding the pivot wave's LDS traffic)
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(dv + j * 8) : "memory");
    while (__builtin_amdgcn_readfirstlane(__double2hiint(f)) <= 0 && spins++ < GMW_POLL_LIMIT)
        asm volatile("s_sleep 1\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(dv + j * 8) : "memory");
    asm volatile("" ::: "memory");             // nothing below may be read before the flag
}

// T[r][c'] -= L[J][r] T[J][c'] for r > J, J = J0..J1-1.  a/b = row J of -L (columns lane&15 / 16 + lane&15, i.e.
// replicated into every DPP row).
template <int J, int J1> struct GmwTStep {
    static __device__ __forceinline__ void run(double (&t)[32], const d2 (&ab)[8], int)
    {
        if constexpr (J < J1) {
            const double tj = t[J];
            if constexpr (J < 15) GmwRowUpd<J + 1, 16, 32>::run(t, ab[J & 7][0], tj);
            GmwRowUpd<(J < 15 ? 16 : J + 1), 32, 32>::run(t, ab[J & 7][1], tj);
            GmwTStep<J + 1, J1>::run(t, ab, 0);
        }
    }
};
// Flag and rows of group [J0, J1) as read (speculatively) while the previous group was being processed: valid if f > 0
// (the flag is read BEFORE the rows, and LDS executes a wave's reads in order).
struct GmwTPre { double f; d2 ab[8]; };
template <int J0, int J1> __device__ __forceinline__ void gmw_t_prefetch(GmwTPre& p, const GmwColsLds& w, int lane)
{
    asm volatile("" ::: "memory");
    p.f = w.Dv[J1 - 1];
    asm volatile("" ::: "memory");             // the compiler must not hoist a row read above the flag read
    const d2* strip = (const d2*)&w.Lm[(lane & 15) * GMW_LM_STRIDE];
#pragma unroll
    for (int j = J0; j < J1; j++) p.ab[j & 7] = strip[j];
    asm volatile("" ::: "memory");
}
// Group [J0, J1); [N0, N1) is the group after it (N1 == N0: none), read ahead into nxt before this group's FMAs so that a
// follower that has fallen behind pays no LDS round trip per group.
template <int J0, int J1, int N0, int N1, int GLOBAL>
__device__ __forceinline__ void gmw_t_group(double (&t)[32], const GmwColsLds& w, unsigned dv, int lane,
                                            double* __restrict__ Tt, double* Tl, GmwTPre& cur, GmwTPre& nxt)
{
    // not published yet when we looked: read flag + rows again (one LDS round trip per attempt, bounded)
    for (int spins = 0; __builtin_amdgcn_readfirstlane(__double2hiint(cur.f)) <= 0 && spins < GMW_POLL_LIMIT; spins++) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        gmw_t_prefetch<J0, J1>(cur, w, lane);
    }
    if constexpr (N1 > N0) gmw_t_prefetch<N0, N1>(nxt, w, lane);
    GmwTStep<J0, J1>::run(t, cur.ab, 0);
#pragma unroll
    for (int r = J0 + 1; r < 32; r++) asm volatile("" : "+v"(t[r]));   // keep the FMAs in this group (no sinking past the next poll)
    // rows < J1 of T are final now (row 31 after the last group): publish rows [J0, J1') while the pivot wave works on
    // the next group.  Column per lane is the wrong shape for a global store (32 lanes x 8 B, 256 B apart: ~85 cycles
    // of issue per row), so the rows go to the LDS copy Tl[kk][33] first and come back transposed: one 128-bit store
    // per lane for the whole group.
    constexpr int R0 = J0, R1 = (J1 == 31) ? 32 : J1, NR = R1 - R0;     // 8 or 4 rows
    if (lane < 32) {
#pragma unroll
        for (int r = R0; r < R1; r++) Tl[lane * 33 + r] = t[r];
    }
    if constexpr (GLOBAL != 0) {
        const int kk = (NR == 8) ? (lane >> 1) : lane, jj = R0 + ((NR == 8) ? 4 * (lane & 1) : 0);
        if (NR == 8 || lane < 32) {
            const double* src = &Tl[kk * 33 + jj];
            d4 v = { src[0], src[1], src[2], src[3] };
            st_d4<GLOBAL == 2>(&Tt[kk * 32 + jj], v);
        }
    }
}

// Follower wave 1: T = L^{-1}, column c' = lane & 31 per lane, into the LDS array Tl[kk][33] = T[jj][kk] (32 x 33 doubles)
// and, if GLOBAL, through it to Tt[kk*32 + jj] in global memory (otherwise gmw_copy_t does that later).
// GLOBAL: 0 = LDS copy only, 1 = also to Tt with plain stores, 2 = with agent-scope stores (read by other workgroups of the same launch)
struct GmwNoMid { __device__ __forceinline__ void operator()() const {} };
// mid(): called once between the groups 16..23 and 24..27, where this wave is ahead of the pivot wave anyway
template <int GLOBAL, class Mid = GmwNoMid>
__device__ __forceinline__ void gmw_cols_t_wave(const GmwColsLds& w, int lane, double* __restrict__ Tt, double* Tl, Mid&& mid = Mid())
{
    const int c = lane & 31;
    const unsigned dv = lds_off(w.Dv);
    double t[32];
#pragma unroll
    for (int r = 0; r < 32; r++) t[r] = (r == c) ? 1.0 : 0.0;
    GmwTPre pa, pb;
    pa.f = 0.0;                                                         // nothing read ahead for the first group
    gmw_t_group<0, 4, 4, 8, GLOBAL>(t, w, dv, lane, Tt, Tl, pa, pb);
    gmw_t_group<4, 8, 8, 16, GLOBAL>(t, w, dv, lane, Tt, Tl, pb, pa);
    gmw_t_group<8, 16, 16, 24, GLOBAL>(t, w, dv, lane, Tt, Tl, pa, pb);
    gmw_t_group<16, 24, 24, 28, GLOBAL>(t, w, dv, lane, Tt, Tl, pb, pa);
    mid();
    gmw_t_group<24, 28, 28, 31, GLOBAL>(t, w, dv, lane, Tt, Tl, pa, pb);
    gmw_t_group<28, 31, 0, 0, GLOBAL>(t, w, dv, lane, Tt, Tl, pb, pa);
}
// Tl[kk][33] -> Tt[kk*32 + jj], one wave, coalesced 128-bit stores
template <bool DEV = false>
__device__ __forceinline__ void gmw_copy_t(const double* Tl, double* __restrict__ Tt, int lane)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int kk = 8 * i + (lane >> 3), jj = (lane & 7) * 4;
        const double* src = &Tl[kk * 33 + jj];
        d4 v = { src[0], src[1], src[2], src[3] };
        st_d4<DEV>(&Tt[kk * 32 + jj], v);
    }
}

// Follower wave 2: outputs of rows J0..J1-1 — S rows j0+J (diagonal-block part), pivots, per-row scales of the
// panel buffer.  Lane l handles row J0 + (l >> 3) (when the group has 8 rows) and four columns.
template <int J0, int J1, bool DEV> __device__ __forceinline__ void gmw_out_group(const GmwColsLds& w, unsigned dv, int lane, int n, int ld, int j0,
                                                                        double* __restrict__ pD, double* __restrict__ psq, double* __restrict__ prD,
                                                                        double* __restrict__ Dall, double* __restrict__ Sout, double* lsq, double* lrc)
{
    gmw_wait_row(dv, J1 - 1);
    const int j = J0 + (lane >> 3), c0 = (lane & 7) * 4;
    if (j >= J1) return;
    const double D = w.Dv[j];
    double l[4];
#pragma unroll
    for (int q = 0; q < 4; q++) l[q] = gmw_lm(w.Lm, j, c0 + q);          // unconditional: all four LDS reads in flight together
    const double sq = sqrt(D);
    if ((lane & 7) == 0) {
        const double rc = gmw_pivot_rcp(D);
        if constexpr (DEV) { st_dev(&pD[j], D); st_dev(&psq[j], sq * rc); st_dev(&prD[j], rc); }
        else { pD[j] = D; psq[j] = sq * rc; prD[j] = rc; }
        Dall[j0 + j] = D;
        if (lsq) { lsq[j] = sq * rc; lrc[j] = rc; }                    // workgroup-local copies
    }
    // the whole 32-byte chunk is stored: zeros below the diagonal and in the padding rows / columns are what S holds there anyway
    d4 v;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c = c0 + q;
        v[q] = (c > j && j0 + c < n && j0 + j < n) ? -(l[q] * sq) : ((c == j && j0 + j < n) ? sq : 0.0);
    }
    *(d4*)&Sout[(size_t)(j0 + j) * ld + j0 + c0] = v;
}
// Two output waves (which = 0 / 1) take alternate groups, so the last group starts the moment its rows exist.
// before_last(): called once, before the wave's last group — the persistent kernel requests the flag of the tile it stages next
// there, so that the flag's round trip runs under the last group instead of after it.
struct GmwNoHook { __device__ __forceinline__ void operator()() const {} };
template <bool DEV = false, class Hook = GmwNoHook>
__device__ __forceinline__ void gmw_cols_out_wave(const GmwColsLds& w, int which, int lane, int n, int ld, int j0,
                                                  double* __restrict__ pD, double* __restrict__ psq, double* __restrict__ prD,
                                                  double* __restrict__ Dall, double* __restrict__ Sout, double* lsq = nullptr, double* lrc = nullptr,
                                                  Hook&& before_last = Hook())
{
    const unsigned dv = lds_off(w.Dv);
    if (which == 0) {
        gmw_out_group<0, 8, DEV>(w, dv, lane, n, ld, j0, pD, psq, prD, Dall, Sout, lsq, lrc);
        gmw_out_group<16, 24, DEV>(w, dv, lane, n, ld, j0, pD, psq, prD, Dall, Sout, lsq, lrc);
        before_last();
        gmw_out_group<28, 32, DEV>(w, dv, lane, n, ld, j0, pD, psq, prD, Dall, Sout, lsq, lrc);
    } else {
        gmw_out_group<8, 16, DEV>(w, dv, lane, n, ld, j0, pD, psq, prD, Dall, Sout, lsq, lrc);
        before_last();
        gmw_out_group<24, 28, DEV>(w, dv, lane, n, ld, j0, pD, psq, prD, Dall, Sout, lsq, lrc);
    }
}
