// srukf_mixed.hip — SRUKF_STORAGE_F32_MIXED (BASELINE configs[4]: "fp32 SRUKF with mixed-precision sqrt-S downdate").
//
// What is formed in single precision is the downdated covariance itself, G = S^T S - U U^T (SLAM.cpp:2118-2120, 2149
// batched over the measurement columns): two thirds of the refactorisation's flops.  Its operands are already fp32 in
// this mode — S32 is the stored filter state, U^T is rounded once — and the products run on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 2x the FP64 rate), in K chunks of at most 1024 whose partial sums are
// added in FP64, in fixed order.  The pivots, the diagonal blocks and the trailing updates of the modified Cholesky
// (SLAM.cpp:2197-2327) stay FP64 (srukf_factor.hip / srukf_gmw_persist.hip): the clamp max(EPSILON, |c_jj|) decides on
// differences of nearly equal numbers.
//
// k_syrk32: one workgroup per (128 x 128 macro tile of the upper triangle, K chunk).  Operand slabs of 32 rows go
// global -> registers -> LDS (double buffered); each of the four waves owns a 64 x 64 quadrant = 2 x 2 MFMA tiles.
// k_syrk32_reduce: sums the chunk partials of every tile in FP64 (chunk order), writes G and the gamma / xi of the GMW
// bound (SLAM.cpp:2204-2211).
#include <hip/hip_runtime.h>
#include "srukf_device.h"

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define MX_TILE 128
#define MX_KS 32                       // rows per slab
#ifndef MX_KCHUNK
#define MX_KCHUNK 1024                 // K range of one task (a multiple of MX_KS * MX_FLUSH)
#endif
// fp32 accumulation length in slabs of MX_KS rows: after every MX_FLUSH slabs the fp32 accumulators are added to FP64 ones held beside them in registers and cleared.
// Measured with the accumulation length as the K chunk (round 6, N = 500, the hybrid form: scripts/mixed_drift_probe.py): the kept pivots of the filter drift away
// from the fp64 filter's LINEARLY in the frame count, in proportion to that length — 1024 rows: 3 % after 240 frames, then a flagged frame and divergence; 256: the same
// at frame 637; 128: 1 % after 800 frames.  An fp32 sum of K products carries ~ sqrt(K) eps32 of sum |a b|, and it does so in every entry of P in every frame.
#ifndef MX_FLUSH
#define MX_FLUSH 1
#endif
#define MX_LS (MX_TILE + 32)           // LDS row stride in floats: the two 32-lane halves of a fragment read (rows k, k + 1) land on opposite bank halves

struct MxTask { short I, J, chunk, nchunks; };   // macro tile (I <= J) and K chunk; partial slot = task index

// C[k][m] layout: A operand of lane l = As[k + (l >> 5)][m0 + (l & 31)], B likewise; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
// krows (a multiple of MX_KS): rows of S32 that hold anything — np for the state itself; for the rank-aware form S32 is the permuted copy of the KEPT rows
// (r of them, upper triangular in permuted order, zero rows behind): K ends at the kept rows
__global__ __launch_bounds__(256) void k_syrk32(int np, int ue, const float* __restrict__ S32, const float* __restrict__ U32,
                                                const MxTask* __restrict__ tasks, double* __restrict__ part, int krows)
{
    __shared__ float As[2][MX_KS][MX_LS];
    __shared__ float Bs[2][MX_KS][MX_LS];
    const MxTask tk = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mb = tk.I * MX_TILE, nb = tk.J * MX_TILE;
    const int ks = min(mb + MX_TILE, krows);                     // S[k][m] = 0 for k > m: rows of S that contribute
    const int ktot = ks + ue;                                    // concatenated K range: S rows, then U^T rows
    const int kbeg = tk.chunk * MX_KCHUNK, kend = min(ktot, kbeg + MX_KCHUNK);
    const int lr = tid >> 5, lc = (tid & 31) * 4;                // this thread's part of a slab: rows lr + 8 i, columns lc .. lc + 3
    f16v acc[2][2];
    double acc64[2][2][16];                                      // what the fp32 accumulators have been flushed into (MX_FLUSH)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) { acc[a][b][t] = 0.f; acc64[a][b][t] = 0.0; }
    const int wm = 64 * (wv >> 1), wn = 64 * (wv & 1);
    const bool active = (mb + wm < np) && (nb + wn < np) && (nb + wn + 64 > mb + wm);   // quadrant inside the matrix and not strictly below the diagonal
    f4v ra[4], rb[4];
    auto fetch = [&](int k0) {                                   // rows k0 .. k0 + 31 of the concatenated operand
        const bool inS = k0 < ks;
        const float* src = inS ? S32 + (size_t)k0 * np : U32 + (size_t)(k0 - ks) * np;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float* row = src + (size_t)(lr + 8 * i) * np;
            ra[i] = (mb + lc < np) ? *(const f4v*)(row + mb + lc) : (f4v){0.f, 0.f, 0.f, 0.f};
            rb[i] = (nb + lc < np) ? *(const f4v*)(row + nb + lc) : (f4v){0.f, 0.f, 0.f, 0.f};
            if (!inS) ra[i] = -ra[i];                            // - U U^T
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) { *(f4v*)&As[buf][lr + 8 * i][lc] = ra[i]; *(f4v*)&Bs[buf][lr + 8 * i][lc] = rb[i]; }
    };
    int buf = 0, nslab = 0;
    if (kbeg < kend) { fetch(kbeg); stash(0); }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += MX_KS) {
        const bool more = k0 + MX_KS < kend;
        if (more) fetch(k0 + MX_KS);                             // next slab in flight while this one is multiplied
        if (active) {
            // fragments of step kk + 2 are requested before the MFMAs of step kk issue (hipcc otherwise waits for each step's
            // LDS reads right before its first MFMA: one LDS round trip per 256 cycles of matrix work)
            const int h = lane >> 5, l31 = lane & 31;
            const float* ap = &As[buf][h][wm + l31];
            const float* bp = &Bs[buf][h][wn + l31];
            float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll
            for (int kk = 0; kk < MX_KS; kk += 2) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (kk + 2 < MX_KS) {
                    na0 = ap[(kk + 2) * MX_LS]; na1 = ap[(kk + 2) * MX_LS + 32];
                    nb0 = bp[(kk + 2) * MX_LS]; nb1 = bp[(kk + 2) * MX_LS + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
        if (active && (++nslab == MX_FLUSH || !more)) {
            nslab = 0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 16; t++) { acc64[a][b][t] += (double)acc[a][b][t]; acc[a][b][t] = 0.f; }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // partial tile of this chunk: part[task][128][128] doubles
    double* out = part + (size_t)blockIdx.x * MX_TILE * MX_TILE;
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int r = wm + 32 * a + (t & 3) + 8 * (t >> 2) + 4 * h, c = wn + 32 * b + l31;
                out[r * MX_TILE + c] = acc64[a][b][t];
            }
}

// ---- the same product from bf16 pieces (round 6) ----------------------------------------------------------------------------------------------------------------
// On MI355X the FP32 matrix peak is 2 x the FP64 one and 1/16 of the bf16 one.  A float splits EXACTLY into three bf16 pieces of eight significant bits each
// (x = h + m + l: h = x truncated to its upper 16 bits, m = (x - h) truncated, l = the rest), and a product of two floats is then six bf16 products up to 2^-24 of itself
// (h h', h m', m h', m m', h l', l h'; the three dropped ones are below 2^-24): six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block of multiply-adds at 32 cycles each
// against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 — 2.7 x the fp32 pipe's rate, 5.3 x the FP64 one's —, accumulated in fp32 and flushed into FP64 registers every
// 32 rows like k_syrk32.  Operands: X^T as three planes of bf16, [column][K] with K contiguous (the A rows first, the U^T rows behind them at krows): a lane's MFMA
// fragment — eight consecutive k of one column — is one 16-byte load from global memory and one ds_read_b128 from the LDS image, which is a plain copy of it.
// k_split_bf3: rows [0, rows) of a K-major fp64 matrix -> rounded to float (one rounding: the stored state is float already, U^T is rounded as for k_syrk32), split,
// transposed into the planes at K offset koff.  grid (ld / 64, rows / 32), 256 threads.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned short us8v __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_split_bf3(int rows, int ld, int ktot, int koff, const double* __restrict__ src, unsigned short* __restrict__ planes, size_t plane_stride)
{
    __shared__ float t[32][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 32, tid = threadIdx.x;
    {
        const int r = tid >> 3, c8 = (tid & 7) * 8;
        const double* q = src + (size_t)(r0 + r) * ld + c0 + c8;
#pragma unroll
        for (int u = 0; u < 8; u++) t[r][c8 + u] = (r0 + r < rows) ? (float)q[u] : 0.f;
    }
    __syncthreads();
    const int col = tid & 63, kg = tid >> 6;
    us8v h, m, l;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float x = t[kg * 8 + j][col];
        const unsigned xb = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(xb);                 // exact
        const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb);                // exact, at most eight significant bits
        h[j] = (unsigned short)(xb >> 16); m[j] = (unsigned short)(mb >> 16); l[j] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
    unsigned short* o = planes + (size_t)(c0 + col) * ktot + koff + r0 + kg * 8;
    *(us8v*)o = h; *(us8v*)(o + plane_stride) = m; *(us8v*)(o + 2 * plane_stride) = l;
}

#define BF3_LROW 40                    // LDS row of one column: 32 k (64 bytes) + 16 bytes of padding (ushorts): 16 lanes' ds_read_b128 cover the 64 banks once
__global__ __launch_bounds__(256) void k_syrk_bf3(int np, int ktot, int krows, int ue, const unsigned short* __restrict__ planes, size_t plane_stride,
                                                  const MxTask* __restrict__ tasks, double* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) unsigned short sA[3][MX_TILE][BF3_LROW];
    __shared__ __attribute__((aligned(16))) unsigned short sB[3][MX_TILE][BF3_LROW];
    const MxTask tk = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mb = tk.I * MX_TILE, nb = tk.J * MX_TILE;
    const int ks = min(mb + MX_TILE, krows);
    const int klen = ks + ue;
    const int kbeg = tk.chunk * MX_KCHUNK, kend = min(klen, kbeg + MX_KCHUNK);
    f16v acc[2][2];
    double acc64[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) { acc[a][b][t] = 0.f; acc64[a][b][t] = 0.0; }
    const int wm = 64 * (wv >> 1), wn = 64 * (wv & 1);
    const bool active = (mb + wm < np) && (nb + wn < np) && (nb + wn + 64 > mb + wm);
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    u4v ra[6], rb[6];
    auto fetch = [&](int k0) {
        const bool inS = k0 < ks;
        const int kk = inS ? k0 : krows + (k0 - ks);            // position in the concatenated K of the planes
        const unsigned flip = inS ? 0u : 0x80008000u;           // - U U^T: the row operand's pieces change sign
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int q = tid + 256 * i, p = q >> 9, rem = q & 511, col = rem >> 2, pt = rem & 3;
            const unsigned short* base = planes + (size_t)p * plane_stride + kk + pt * 8;
            ra[i] = (mb + col < np) ? *(const u4v*)(base + (size_t)(mb + col) * ktot) : (u4v){0u, 0u, 0u, 0u};
            rb[i] = (nb + col < np) ? *(const u4v*)(base + (size_t)(nb + col) * ktot) : (u4v){0u, 0u, 0u, 0u};
            ra[i] ^= (u4v){flip, flip, flip, flip};
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int q = tid + 256 * i, p = q >> 9, rem = q & 511, col = rem >> 2, pt = rem & 3;
            *(u4v*)&sA[p][col][pt * 8] = ra[i]; *(u4v*)&sB[p][col][pt * 8] = rb[i];
        }
    };
    if (kbeg < kend) fetch(kbeg);
    const int l31 = lane & 31, h = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += MX_KS) {
        __syncthreads();                                        // the previous slab's fragments have been read
        stash();
        __syncthreads();
        if (k0 + MX_KS < kend) fetch(k0 + MX_KS);               // next slab in flight under this one's products
        if (active) {
#pragma unroll
            for (int st = 0; st < 2; st++) {
                bf8v fa[2][3], fb[2][3];
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int p = 0; p < 3; p++) {
                        fa[a][p] = *(const bf8v*)&sA[p][wm + 32 * a + l31][16 * st + 8 * h];
                        fb[a][p] = *(const bf8v*)&sB[p][wn + 32 * a + l31][16 * st + 8 * h];
                    }
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const int pa = (c == 2 || c == 3) ? 1 : (c == 5 ? 2 : 0), pb = (c == 1 || c == 3) ? 1 : (c == 4 ? 2 : 0);      // (h h) (h m) (m h) (m m) (h l) (l h)
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][pa], fb[b][pb], acc[a][b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 16; t++) { acc64[a][b][t] += (double)acc[a][b][t]; acc[a][b][t] = 0.f; }
        }
    }
    double* out = part + (size_t)blockIdx.x * MX_TILE * MX_TILE;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int r = wm + 32 * a + (t & 3) + 8 * (t >> 2) + 4 * h, c = wn + 32 * b + l31;
                out[r * MX_TILE + c] = acc64[a][b][t];
            }
}

// one workgroup per macro tile: G = sum over the tile's chunks (FP64, chunk order), gamma = max diag, xi = max(0, max offdiag)
__global__ __launch_bounds__(256) void k_syrk32_reduce(int n, int np, const int2* __restrict__ tiles /* (first task, nchunks) per tile */,
                                                       const MxTask* __restrict__ tasks, const double* __restrict__ part,
                                                       double* __restrict__ G, FrameScalars* __restrict__ fs)
{
    const int2 tl = tiles[blockIdx.x];
    const MxTask tk = tasks[tl.x];
    const int mb = tk.I * MX_TILE, nb = tk.J * MX_TILE;
    double gmax = 0.0, xmax = 0.0;
    for (int e = threadIdx.x; e < MX_TILE * MX_TILE / 4; e += 256) {
        const int r = e / (MX_TILE / 4), c = (e % (MX_TILE / 4)) * 4;
        if (mb + r >= np || nb + c >= np) continue;
        double s[4] = { 0.0, 0.0, 0.0, 0.0 };
        for (int q = 0; q < tl.y; q++) {
            const d4 v = *(const d4*)(part + ((size_t)(tl.x + q) * MX_TILE + r) * MX_TILE + c);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        const int gr = mb + r;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int gc = nb + c + u;
            if (gc < gr) continue;                               // upper triangle only
            G[(size_t)gr * np + gc] = s[u];
            if (gr < n && gc < n) { if (gr == gc) gmax = fmax(gmax, s[u]); else xmax = fmax(xmax, s[u]); }
        }
    }
    gmax = wave_max(gmax); xmax = wave_max(xmax);
    if ((threadIdx.x & 63) == 0) {
        if (gmax > 0.0) atomicMax(&fs->gmax_bits, (unsigned long long)__double_as_longlong(gmax));
        if (xmax > 0.0) atomicMax(&fs->ximax_bits, (unsigned long long)__double_as_longlong(xmax));
    }
}

// the state update X += sum_k K_k (z_k - h_k) that k_gain left in slices (k_syrk adds it in its spare workgroups; here its own launch)
__global__ __launch_bounds__(256) void k_gain_dx(int n, int np, const double* __restrict__ dxp, double* __restrict__ X, const double* __restrict__ xr1)
{
    srukf_gain_dx_job(n, np, dxp, X, blockIdx.x, xr1);
}

// The motion step rewrites the last four columns of the FP64 working copy of S after the state was rounded (k_motion:
// R12 rows and the 4 x 4 block R22): bring the fp32 operand up to date — rows 0..n-1, columns n-4..n-1, one rounding each.
__global__ __launch_bounds__(256) void k_cvt_robot_cols(int n, int np, const double* __restrict__ S, float* __restrict__ S32)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
#pragma unroll
    for (int e = 0; e < 4; e++) S32[(size_t)r * np + n - 4 + e] = (float)S[(size_t)r * np + n - 4 + e];
}

// U^T (fp64, rows [0, ue)) -> fp32, one rounding per entry
__global__ __launch_bounds__(256) void k_cvt_f32(size_t count, const double* __restrict__ src, float* __restrict__ dst)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < count) {
        const d4 v = *(const d4*)(src + i);
        *(f4v*)(dst + i) = (f4v){ (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
    } else for (size_t j = i; j < count; j++) dst[j] = (float)src[j];
}

extern "C" {
// host-side task list for an np x np matrix with ue rows of U^T: tiles longest-K first, chunks of a tile adjacent.
// out_tasks: 4 shorts per task; out_tiles: (first task, nchunks) per tile.  Returns the number of tasks; *ntiles gets the tile count.
// krows / rows_lim (rank-aware form): K of the first operand ends at krows, and only the macro tiles that hold rows < rows_lim (the pivoted panels) are formed
int srukf_mixed_build_tasks_red(int np, int ue, int krows, int rows_lim, short* out_tasks, int* out_tiles, int* ntiles)
{
    const int T = (np + MX_TILE - 1) / MX_TILE;
    int nt = 0, ntl = 0;
    for (int I = T - 1; I >= 0; I--)                             // large I = long K first
        for (int J = I; J < T; J++) {
            if (I * MX_TILE >= rows_lim) continue;
            const int ks = (I + 1) * MX_TILE < krows ? (I + 1) * MX_TILE : krows;
            const int nch = (ks + ue + MX_KCHUNK - 1) / MX_KCHUNK;
            if (out_tiles) { out_tiles[2 * ntl] = nt; out_tiles[2 * ntl + 1] = nch; }
            for (int c = 0; c < nch; c++) {
                if (out_tasks) { out_tasks[4 * nt] = (short)I; out_tasks[4 * nt + 1] = (short)J; out_tasks[4 * nt + 2] = (short)c; out_tasks[4 * nt + 3] = (short)nch; }
                nt++;
            }
            ntl++;
        }
    if (ntiles) *ntiles = ntl;
    return nt;
}
int srukf_mixed_build_tasks(int np, int ue, short* out_tasks, int* out_tiles, int* ntiles) { return srukf_mixed_build_tasks_red(np, ue, np, np, out_tasks, out_tiles, ntiles); }
int srukf_mixed_krows(int r) { return (r + MX_KS - 1) / MX_KS * MX_KS; }
size_t srukf_mixed_part_bytes(int ntasks) { return (size_t)ntasks * MX_TILE * MX_TILE * sizeof(double); }
void srukf_launch_cvt_f32(hipStream_t st, size_t count, const double* src, float* dst)
{
    hipLaunchKernelGGL(k_cvt_f32, dim3((unsigned)((count / 4 + 255) / 256 + 1)), dim3(256), 0, st, count, src, dst);
}
void srukf_launch_gain_dx(hipStream_t st, int n, int np, const double* dxp, double* X, const double* xr1)
{
    hipLaunchKernelGGL(k_gain_dx, dim3((n + 255) / 256), dim3(256), 0, st, n, np, dxp, X, xr1);
}
void srukf_launch_cvt_robot_cols(hipStream_t st, int n, int np, const double* S, float* S32)
{
    hipLaunchKernelGGL(k_cvt_robot_cols, dim3((n + 255) / 256), dim3(256), 0, st, n, np, S, S32);
}
void srukf_launch_syrk32(hipStream_t st, int n, int np, int ue, const float* S32, const float* U32, const void* tasks, int ntasks,
                         const void* tiles, int ntiles, double* part, double* G, void* fs, int krows)
{
    hipLaunchKernelGGL(k_syrk32, dim3(ntasks), dim3(256), 0, st, np, ue, S32, U32, (const MxTask*)tasks, part, krows > 0 ? krows : np);
    hipLaunchKernelGGL(k_syrk32_reduce, dim3(ntiles), dim3(256), 0, st, n, np, (const int2*)tiles, (const MxTask*)tasks, part, G, (FrameScalars*)fs);
}
// the bf16-piece form: operands as X^T planes (k_split_bf3), products by k_syrk_bf3, the same reduction
void srukf_launch_split_bf3(hipStream_t st, int rows, int ld, int ktot, int koff, const double* src, void* planes, size_t plane_stride)
{
    hipLaunchKernelGGL(k_split_bf3, dim3(ld / 64, (rows + 31) / 32), dim3(256), 0, st, rows, ld, ktot, koff, src, (unsigned short*)planes, plane_stride);
}
void srukf_launch_syrk_bf3(hipStream_t st, int n, int np, int ue, int krows, int ktot, const void* planes, size_t plane_stride, const void* tasks, int ntasks,
                           const void* tiles, int ntiles, double* part, double* G, void* fs)
{
    hipLaunchKernelGGL(k_syrk_bf3, dim3(ntasks), dim3(256), 0, st, np, ktot, krows, ue, (const unsigned short*)planes, plane_stride, (const MxTask*)tasks, part);
    hipLaunchKernelGGL(k_syrk32_reduce, dim3(ntiles), dim3(256), 0, st, n, np, (const int2*)tiles, (const MxTask*)tasks, part, G, (FrameScalars*)fs);
}
}  // extern "C"
