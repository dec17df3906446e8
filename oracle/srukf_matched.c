/*
 * srukf_matched.c — "B-matched" CPU baseline: the SAME formulation the
 * MI355X path runs — sigma set never materialised, structured O(n) motion re-triangularisation, one dense
 * cross-covariance contraction, ONE batched refactor  S <- gmw(S^T S - U U^T)  per frame with a blocked right-looking
 * modified Cholesky whose theta clamp is verified afterwards — written for the host: OpenMP over all cores, 8 x 24
 * register tiles on K-major operands (AVX-512 / AVX2 clones picked at load time).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY, like srukf_oracle.c (which it includes for the camera model, the 2x2 helpers, the
 * Householder R of the Si blocks and the exact orc_gmw fallback).  Nothing under cv-monoslam_amd/ links or calls it.
 * It exists so that bench.py can time an algorithm-matched, multi-threaded CPU implementation next to the GPU, beside
 * the reference-structured single-thread port; tests/test_oracle.py holds it to the oracle's results.
 *
 * Build:  make -C oracle   (gcc -O3 -fopenmp; no -march: the hot loops carry target_clones)
 */
#include "srukf_oracle.c"
#include <omp.h>
#include <immintrin.h>

#define MT_NB 48                       /* GMW panel height: a multiple of the 8 x 24 register tile in both directions */
#define MT_PAD 48                      /* leading dimensions are padded to this, the padding stays zero */

#define MT_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

typedef struct mt_state {
    orc_state *o;                      /* parameters, weights, X, and the exact fallback                              */
    int N, n, Na, L, np, mp;           /* np = padded n, mp = padded 2N                                               */
    double *S;                         /* np x np upper, padded copy of the filter's sqrt covariance                  */
    double *G, *Gbak;                  /* np x np: S^T S - U U^T (upper), factorised in place / its backup            */
    double *sigR;                      /* L x 4 robot part of every sigma point after the motion model                */
    double *Z;                         /* L x mp projected pixels, row = sigma point                                  */
    double *DZ;                        /* np x mp: Z[1+i] - Z[1+Na+i]                                                 */
    double *Ut;                        /* mp x np: U^T                                                                */
    double *Lw;                        /* MT_NB x np scaled panel rows                                                */
    double *D;                         /* np pivots                                                                   */
    double *h, *Si; int *vis;
    long long clamp_fallbacks;
    /* rank-aware form (mt_set_rank_aware; the GPU path's default, DESIGN.md "Rank-aware refactorisation"): the r kept directions first, same relative order */
    int rank_aware, r;                 /* r = 0: the full-rank form runs                                              */
    int *perm, *iperm;                 /* permuted position -> state index and back (np entries)                      */
    double *A;                         /* np x np: kept rows of S in permuted column order (rows >= r unused)         */
    long long rank_fallbacks;          /* frames whose skipped directions were found not to be null (full path taken) */
    double t_phase[6];                 /* seconds spent in motion, measurement, gains, S^T S - U U^T, factorisation, fallback */
} mt_state;

static int mt_round(int v, int m) { return (v + m - 1) / m * m; }

ORC_API mt_state *mt_create(int N, const srukf_params *p, int threads)
{
    mt_state *m = (mt_state *)calloc(1, sizeof *m);
    m->o = orc_create(N, p);
    m->N = N; m->n = 6 * N + 4; m->Na = m->n + 5; m->L = 2 * m->Na + 1;
    m->np = mt_round(m->n, MT_PAD); m->mp = mt_round(2 * N > 0 ? 2 * N : 1, MT_PAD);
    size_t np = m->np, mp = m->mp;
    m->S = (double *)calloc(np * np, sizeof(double)); m->G = (double *)calloc(np * np, sizeof(double));
    m->Gbak = (double *)calloc(np * np, sizeof(double));
    m->sigR = (double *)calloc((size_t)m->L * 4, sizeof(double));
    m->Z = (double *)calloc((size_t)m->L * mp, sizeof(double)); m->DZ = (double *)calloc(np * mp, sizeof(double));
    m->Ut = (double *)calloc(mp * np, sizeof(double)); m->Lw = (double *)calloc((size_t)MT_NB * np, sizeof(double));
    m->D = (double *)calloc(np, sizeof(double));
    m->h = (double *)calloc(mp, sizeof(double)); m->Si = (double *)calloc(4 * (size_t)(N > 0 ? N : 1), sizeof(double));
    m->vis = (int *)calloc(N > 0 ? N : 1, sizeof(int));
    m->perm = (int *)calloc(np, sizeof(int)); m->iperm = (int *)calloc(np, sizeof(int));
    m->A = (double *)calloc(np * np, sizeof(double));
    if (threads > 0) omp_set_num_threads(threads);
    return m;
}
ORC_API void mt_destroy(mt_state *m)
{
    if (!m) return;
    orc_destroy(m->o);
    free(m->S); free(m->G); free(m->Gbak); free(m->sigR); free(m->Z); free(m->DZ); free(m->Ut); free(m->Lw); free(m->D);
    free(m->h); free(m->Si); free(m->vis); free(m->perm); free(m->iperm); free(m->A); free(m);
}
ORC_API int mt_threads(void) { int t = 1;
#pragma omp parallel
    {
#pragma omp master
        t = omp_get_num_threads();
    }
    return t; }
ORC_API void mt_set_state(mt_state *m, const double *X, const double *S)
{
    int n = m->n; size_t np = m->np;
    memcpy(m->o->X, X, sizeof(double) * n);
    memset(m->S, 0, sizeof(double) * np * np);
    for (int r = 0; r < n; r++) memcpy(m->S + r * np + r, S + (size_t)r * n + r, sizeof(double) * (n - r));
}
ORC_API void mt_get_state(const mt_state *m, double *X, double *S)
{
    int n = m->n; size_t np = m->np;
    if (X) memcpy(X, m->o->X, sizeof(double) * n);
    if (S) for (int r = 0; r < n; r++) { memset(S + (size_t)r * n, 0, sizeof(double) * n); memcpy(S + (size_t)r * n + r, m->S + r * np + r, sizeof(double) * (n - r)); }
}
ORC_API long long mt_clamp_fallbacks(const mt_state *m) { return m->clamp_fallbacks; }
ORC_API void mt_phase_times(const mt_state *m, double out[6]) { memcpy(out, m->t_phase, sizeof m->t_phase); }

/* C[8][24] (+)= sgn * sum_k A[k][0..8)^T B[k][0..24): both operands K-major (row k contiguous), no packing.
 * Three builds of the same tile, chosen once at load time from the host's ISA. */
__attribute__((target("avx512f"))) static void mt_tile_tn_512(int K, const double *restrict A, size_t lda, const double *restrict B, size_t ldb,
                                                              double *restrict C, size_t ldc, double sgn, int accumulate)
{
    __m512d c0a = _mm512_setzero_pd(), c0b = c0a, c0c = c0a, c1a = c0a, c1b = c0a, c1c = c0a, c2a = c0a, c2b = c0a, c2c = c0a, c3a = c0a, c3b = c0a, c3c = c0a;
    __m512d c4a = c0a, c4b = c0a, c4c = c0a, c5a = c0a, c5b = c0a, c5c = c0a, c6a = c0a, c6b = c0a, c6c = c0a, c7a = c0a, c7b = c0a, c7c = c0a;
    for (int k = 0; k < K; k++) {
        const double *b = B + (size_t)k * ldb, *a = A + (size_t)k * lda;
        const __m512d b0 = _mm512_loadu_pd(b), b1 = _mm512_loadu_pd(b + 8), b2 = _mm512_loadu_pd(b + 16);
#define MT_ROW(r, x, y, z) { const __m512d av = _mm512_set1_pd(a[r]); x = _mm512_fmadd_pd(av, b0, x); y = _mm512_fmadd_pd(av, b1, y); z = _mm512_fmadd_pd(av, b2, z); }
        MT_ROW(0, c0a, c0b, c0c) MT_ROW(1, c1a, c1b, c1c) MT_ROW(2, c2a, c2b, c2c) MT_ROW(3, c3a, c3b, c3c)
        MT_ROW(4, c4a, c4b, c4c) MT_ROW(5, c5a, c5b, c5c) MT_ROW(6, c6a, c6b, c6c) MT_ROW(7, c7a, c7b, c7c)
#undef MT_ROW
    }
    const __m512d sv = _mm512_set1_pd(sgn);
#define MT_OUT(r, x, y, z) { double *cr = C + (size_t)(r) * ldc; __m512d t0 = _mm512_mul_pd(sv, x), t1 = _mm512_mul_pd(sv, y), t2 = _mm512_mul_pd(sv, z); \
        if (accumulate) { t0 = _mm512_add_pd(t0, _mm512_loadu_pd(cr)); t1 = _mm512_add_pd(t1, _mm512_loadu_pd(cr + 8)); t2 = _mm512_add_pd(t2, _mm512_loadu_pd(cr + 16)); } \
        _mm512_storeu_pd(cr, t0); _mm512_storeu_pd(cr + 8, t1); _mm512_storeu_pd(cr + 16, t2); }
    MT_OUT(0, c0a, c0b, c0c) MT_OUT(1, c1a, c1b, c1c) MT_OUT(2, c2a, c2b, c2c) MT_OUT(3, c3a, c3b, c3c)
    MT_OUT(4, c4a, c4b, c4c) MT_OUT(5, c5a, c5b, c5c) MT_OUT(6, c6a, c6b, c6c) MT_OUT(7, c7a, c7b, c7c)
#undef MT_OUT
}
/* AVX2 + FMA: the 8 x 24 tile as four 4 x 12 register blocks (12 ymm accumulators each) */
__attribute__((target("avx2,fma"))) static void mt_tile_tn_256(int K, const double *restrict A, size_t lda, const double *restrict B, size_t ldb,
                                                               double *restrict C, size_t ldc, double sgn, int accumulate)
{
    for (int rb = 0; rb < 8; rb += 4)
        for (int cb = 0; cb < 24; cb += 12) {
            __m256d c0a = _mm256_setzero_pd(), c0b = c0a, c0c = c0a, c1a = c0a, c1b = c0a, c1c = c0a, c2a = c0a, c2b = c0a, c2c = c0a, c3a = c0a, c3b = c0a, c3c = c0a;
            for (int k = 0; k < K; k++) {
                const double *b = B + (size_t)k * ldb + cb, *a = A + (size_t)k * lda + rb;
                const __m256d b0 = _mm256_loadu_pd(b), b1 = _mm256_loadu_pd(b + 4), b2 = _mm256_loadu_pd(b + 8);
#define MT_ROW(r, x, y, z) { const __m256d av = _mm256_set1_pd(a[r]); x = _mm256_fmadd_pd(av, b0, x); y = _mm256_fmadd_pd(av, b1, y); z = _mm256_fmadd_pd(av, b2, z); }
                MT_ROW(0, c0a, c0b, c0c) MT_ROW(1, c1a, c1b, c1c) MT_ROW(2, c2a, c2b, c2c) MT_ROW(3, c3a, c3b, c3c)
#undef MT_ROW
            }
            const __m256d sv = _mm256_set1_pd(sgn);
#define MT_OUT(r, x, y, z) { double *cr = C + (size_t)(rb + r) * ldc + cb; __m256d t0 = _mm256_mul_pd(sv, x), t1 = _mm256_mul_pd(sv, y), t2 = _mm256_mul_pd(sv, z); \
                if (accumulate) { t0 = _mm256_add_pd(t0, _mm256_loadu_pd(cr)); t1 = _mm256_add_pd(t1, _mm256_loadu_pd(cr + 4)); t2 = _mm256_add_pd(t2, _mm256_loadu_pd(cr + 8)); } \
                _mm256_storeu_pd(cr, t0); _mm256_storeu_pd(cr + 4, t1); _mm256_storeu_pd(cr + 8, t2); }
            MT_OUT(0, c0a, c0b, c0c) MT_OUT(1, c1a, c1b, c1c) MT_OUT(2, c2a, c2b, c2c) MT_OUT(3, c3a, c3b, c3c)
#undef MT_OUT
        }
}
static void mt_tile_tn_c(int K, const double *restrict A, size_t lda, const double *restrict B, size_t ldb,
                         double *restrict C, size_t ldc, double sgn, int accumulate)
{
    double c[8][24];
    memset(c, 0, sizeof c);
    for (int k = 0; k < K; k++)
        for (int r = 0; r < 8; r++) { const double ar = A[(size_t)k * lda + r]; for (int q = 0; q < 24; q++) c[r][q] += ar * B[(size_t)k * ldb + q]; }
    for (int r = 0; r < 8; r++) for (int q = 0; q < 24; q++) C[r * ldc + q] = (accumulate ? C[r * ldc + q] : 0.0) + sgn * c[r][q];
}
typedef void (*mt_tile_fn)(int, const double *restrict, size_t, const double *restrict, size_t, double *restrict, size_t, double, int);
static mt_tile_fn mt_tile_tn = mt_tile_tn_c;
static const char *mt_isa = "generic";
__attribute__((constructor)) static void mt_pick_isa(void)
{
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) { mt_tile_tn = mt_tile_tn_512; mt_isa = "avx512f"; }
    else if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) { mt_tile_tn = mt_tile_tn_256; mt_isa = "avx2+fma"; }
}
ORC_API const char *mt_isa_name(void) { return mt_isa; }

/* Macro tile: C[8 nrt][24 nct] (+)= sgn * A[0..K)^T B[0..K) in K chunks that keep the two operand slabs in L2 while all
 * nrt x nct register tiles consume them (8 x 24 tiles straight from L3 / DRAM are bandwidth bound: 1.5 flop per byte).
 * klim_mode 1: S is upper triangular and the ROW tile rt only has terms k < klim0 + 8 (rt + 1);  2: the COLUMN tile ct
 * only k < klim0 + 24 (ct + 1);  0: no limit.  skip_below: tiles strictly below the diagonal (c0 + 24 <= r0) are skipped. */
#define MT_KC 256
static void mt_macro_tn(int K, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc,
                        int nrt, int nct, double sgn, int accumulate, int klim_mode, int klim0, int skip_below, int r0, int c0)
{
    for (int k0 = 0; k0 < K; k0 += MT_KC) {
        const int kc = (K - k0 < MT_KC) ? K - k0 : MT_KC;
        for (int rt = 0; rt < nrt; rt++)
            for (int ct = 0; ct < nct; ct++) {
                if (skip_below && c0 + 24 * ct + 24 <= r0 + 8 * rt) continue;
                int kk = kc;
                if (klim_mode == 1) { const int lim = klim0 + 8 * (rt + 1) - k0; if (lim < kk) kk = lim; }
                else if (klim_mode == 2) { const int lim = klim0 + 24 * (ct + 1) - k0; if (lim < kk) kk = lim; }
                if (kk <= 0 && (accumulate || k0 > 0)) continue;
                if (kk < 0) kk = 0;
                mt_tile_tn(kk, A + (size_t)k0 * lda + 8 * rt, lda, B + (size_t)k0 * ldb + 24 * ct, ldb,
                           C + (size_t)(8 * rt) * ldc + 24 * ct, ldc, sgn, accumulate || k0 > 0);
            }
    }
}

/* ---- motion: control, robot sigma rows through the odometry model, structured update of the last 4 columns of S ---- */
static void mt_motion(mt_state *m, const double odo_prev[3], const double odo_cur[3])
{
    orc_state *o = m->o; const srukf_params *p = &o->p;
    const int n = m->n, Na = m->Na, L = m->L; const size_t np = m->np;
    double dx = odo_cur[0] - odo_prev[0], dy = odo_cur[1] - odo_prev[1];
    double rot1 = atan2(dy, dx) - odo_prev[2], trans = sqrt(dy * dy + dx * dx), rot2 = odo_cur[2] - odo_prev[2] - rot1;
    o->Ut[0] = rot1; o->Ut[1] = trans; o->Ut[2] = rot2;
    o->Mt[0] = p->a1 * rot1 * rot1 + p->a2 * trans * trans;
    o->Mt[1] = p->a3 * trans * trans + p->a4 * rot1 * rot1 + p->a4 * rot2 * rot2;
    o->Mt[2] = p->a1 * rot2 * rot2 + p->a2 * trans * trans;
    o->Na = Na; o->L = L;
    set_weights(o, Na);
    const double g = o->gamma;
    double xr[4]; for (int e = 0; e < 4; e++) xr[e] = o->X[n - 4 + e];
    double *sigR = m->sigR;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < L; c++) {
        double rob[4] = { xr[0], xr[1], xr[2], xr[3] }, noise[3] = { 0, 0, 0 };
        if (c > 0) {
            const int i = (c - 1) % Na; const double sg = (c - 1) < Na ? g : -g;
            if (i < n) for (int e = 0; e < 4; e++) rob[e] = xr[e] * 1 + m->S[(size_t)i * np + (n - 4 + e)] * sg + 0;   /* addWeighted, 1159-1160 */
            else if (i < n + 3) noise[i - n] = 0.0 * 1 + o->Mt[i - n] * sg + 0;
        }
        const double r1 = o->Ut[0] - noise[0], tr = o->Ut[1] - noise[1], r2 = o->Ut[2] - noise[2], th = rob[3];
        rob[0] += tr * cos(th + r1); rob[1] += tr * sin(th + r1); rob[2] += 0; rob[3] += r1 + r2;                     /* 1518-1523 */
        for (int e = 0; e < 4; e++) sigR[(size_t)c * 4 + e] = rob[e];
    }
    double mean[4];
    for (int e = 0; e < 4; e++) { double s = 0; for (int c = 1; c < L; c++) s += sigR[(size_t)c * 4 + e] - sigR[e]; mean[e] = sigR[e] * (o->wm0 + 2.0 * Na * o->wi) + o->wi * s; }
    /* A = wi_sr (sigma_{c} - sigma_0)^T, 2Na x n.  A[:, :n-4] = (1/sqrt 2)[E; -E] S11 (wi gamma^2 = 1/2), so R11 = S11,
     * R12[i] = wi_sr/sqrt2 (dev+_i - dev-_i) and R22^T R22 = A2^T A2 - R12^T R12. */
    double gram[4][4]; memset(gram, 0, sizeof gram);
    for (int c = 1; c < L; c++) {
        double d[4]; for (int e = 0; e < 4; e++) d[e] = o->wi_sr * (sigR[(size_t)c * 4 + e] - sigR[e]);
        for (int a = 0; a < 4; a++) for (int b = a; b < 4; b++) gram[a][b] += d[a] * d[b];
    }
    const double is2 = o->wi_sr / sqrt(2.0);
    for (int i = 0; i < n - 4; i++) {
        double r12[4];
        for (int e = 0; e < 4; e++) r12[e] = is2 * (sigR[(size_t)(1 + i) * 4 + e] - sigR[(size_t)(1 + Na + i) * 4 + e]);
        for (int e = 0; e < 4; e++) m->S[(size_t)i * np + (n - 4 + e)] = r12[e];
        for (int a = 0; a < 4; a++) for (int b = a; b < 4; b++) gram[a][b] -= r12[a] * r12[b];
    }
    double R[4][4]; memset(R, 0, sizeof R);
    for (int a = 0; a < 4; a++) {
        double ds = gram[a][a]; for (int k = 0; k < a; k++) ds -= R[k][a] * R[k][a];
        const double raa = sqrt(fmax(ds, 0.0)); R[a][a] = raa;
        for (int b = a + 1; b < 4; b++) { double v = gram[a][b]; for (int k = 0; k < a; k++) v -= R[k][a] * R[k][b]; R[a][b] = raa > 0 ? v / raa : 0.0; }
    }
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) m->S[(size_t)(n - 4 + a) * np + (n - 4 + b)] = R[a][b];
    for (int e = 0; e < 4; e++) o->X[n - 4 + e] = mean[e];
}

/* project_one of the oracle with Rcw = Rwc(psi)^-1 hoisted out of the landmark loop (one per sigma point) */
static inline void mt_project(const orc_state *o, const srukf_params *p, const double feat[6], const double pos[3],
                              const double Rcw[9], const double err[2], double out[2])
{
    const double xi = feat[0], yi = feat[1], zi = feat[2], theta = feat[3], phi = feat[4], rho = feat[5];
    const double cph = cos(phi), sph = sin(phi), cth = cos(theta), sth = sin(theta);
    const double Hlw[3] = { xi + 1 / rho * cph * sth - pos[0], yi - 1 / rho * sph - pos[1], zi + 1 / rho * cph * cth - pos[2] };
    double Hlr[3];
    for (int r = 0; r < 3; r++) Hlr[r] = Rcw[3 * r + 0] * Hlw[0] + Rcw[3 * r + 1] * Hlw[1] + Rcw[3 * r + 2] * Hlw[2];
    const double f1 = p->cam_f / p->cam_dx, f2 = p->cam_f / p->cam_dy;
    double ux, uy;
    if (Hlr[2] == 0) { ux = 0; uy = 0; }
    else {
        uy = p->cam_cx + f1 * Hlr[0] / Hlr[2] + err[0];                           /* 3338 (x/y swap) */
        ux = p->cam_cy + f2 * Hlr[1] / Hlr[2] + err[1];
        if (ux < 10 || ux > p->image_w - 10 || uy < 10 || uy > p->image_h - 10) { ux = 0; uy = 0; }
    }
    distort_rw(o, p, ux, uy, &out[0], &out[1]);
}

/* ---- measurement: L x N projections straight from S, h, Si, visible ---- */
static void mt_measure(mt_state *m)
{
    orc_state *o = m->o; const srukf_params *p = &o->p;
    const int n = m->n, Na = m->Na, L = m->L, N = m->N; const size_t np = m->np, mp = m->mp;
    const double g = o->gamma;
    const int ra = m->r > 0;
    if (ra) {                                                                     /* the centre point first: structurally null directions copy its row */
        double err0[2] = { 0, 0 }, Rwc[9], Rcw[9];
        const double *rob = m->sigR;
        transfer_matrix(Rwc, rob[3]); inv3(Rwc, Rcw);
        for (int k = 0; k < N; k++) mt_project(o, p, o->X + 6 * k, rob, Rcw, err0, m->Z + 2 * k);
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = ra ? 1 : 0; c < L; c++) {
        const int i = c > 0 ? (c - 1) % Na : -1; const double sg = (c - 1) < Na ? g : -g;
        double err[2] = { 0, 0 };
        if (ra && i < n && m->iperm[i] >= m->r) {
            /* NullSkip (srukf_device.h): row i of S is sqrt(EPSILON) e_i and moves ONE landmark; every other landmark's sigma point is the centre
             * point's, bit for bit (same function, same inputs: the robot part of this sigma point equals the centre's) */
            const int k = i / 6;
            const double *rob = m->sigR + (size_t)c * 4;
            double *zr = m->Z + (size_t)c * mp, Rwc[9], Rcw[9], feat[6];
            memcpy(zr, m->Z, sizeof(double) * 2 * N);
            transfer_matrix(Rwc, rob[3]); inv3(Rwc, Rcw);
            for (int e = 0; e < 6; e++) { const int col = 6 * k + e; feat[e] = o->X[col] * 1 + ((col >= i) ? m->S[(size_t)i * np + col] : 0.0) * sg + 0; }
            mt_project(o, p, feat, rob, Rcw, err, zr + 2 * k);
            continue;
        }
        if (i == n + 3) err[0] = 0.0 * 1 + o->Qt[0] * sg + 0; else if (i == n + 4) err[1] = 0.0 * 1 + o->Qt[1] * sg + 0;
        const double *rob = m->sigR + (size_t)c * 4;
        double *zr = m->Z + (size_t)c * mp;
        double Rwc[9], Rcw[9];
        transfer_matrix(Rwc, rob[3]); inv3(Rwc, Rcw);                             /* 1642-1643 */
        for (int k = 0; k < N; k++) {
            double feat[6];
            for (int e = 0; e < 6; e++) {
                const int col = 6 * k + e;
                const double dv = (i >= 0 && i < n && col >= i) ? m->S[(size_t)i * np + col] : 0.0;
                feat[e] = (c > 0) ? o->X[col] * 1 + dv * sg + 0 : o->X[col];
            }
            mt_project(o, p, feat, rob, Rcw, err, zr + 2 * k);
        }
    }
#pragma omp parallel for schedule(static)
    for (int a = 0; a < n; a++) {                                                 /* rank-aware form: the rows of DZ in permuted order */
        const int i = ra ? m->perm[a] : a;
        const double *zp = m->Z + (size_t)(1 + i) * mp, *zm = m->Z + (size_t)(1 + Na + i) * mp;
        double *d = m->DZ + (size_t)a * mp;
        for (int q = 0; q < 2 * N; q++) d[q] = zp[q] - zm[q];
    }
    const double wsum = o->wm0 + 2.0 * Na * o->wi;
#pragma omp parallel
    {
        double *QR = (double *)malloc(sizeof(double) * (size_t)2 * Na * 2);
#pragma omp for schedule(static)
        for (int k = 0; k < N; k++) {
            const double z0x = m->Z[2 * k], z0y = m->Z[2 * k + 1];
            double sx = 0, sy = 0;
            for (int c = 1; c < L; c++) { sx += m->Z[(size_t)c * mp + 2 * k] - z0x; sy += m->Z[(size_t)c * mp + 2 * k + 1] - z0y; }
            const double hx = wsum * z0x + o->wi * sx, hy = wsum * z0y + o->wi * sy;
            m->h[2 * k] = hx; m->h[2 * k + 1] = hy;
            m->vis[k] = (hx != 0 && hy != 0) ? 1 : 0;                                         /* 1727 */
            memset(m->Si + 4 * k, 0, 4 * sizeof(double));
            if (m->vis[k]) {
                for (int c = 0; c < 2 * Na; c++) {
                    QR[2 * c] = o->wi_sr * (m->Z[(size_t)(c + 1) * mp + 2 * k] - z0x);
                    QR[2 * c + 1] = o->wi_sr * (m->Z[(size_t)(c + 1) * mp + 2 * k + 1] - z0y);
                }
                orc_qr_r(QR, 2 * Na, 2, m->Si + 4 * k);                                       /* 1775 */
            }
        }
        free(QR);
    }
}

/* ---- gains: U^T = (Pxy Si^-1)^T for every matched landmark, X += sum K (z - h) ---- */
static void mt_gain(mt_state *m, const double *z, const int *matched)
{
    orc_state *o = m->o;
    const int n = m->n, L = m->L, N = m->N; const size_t np = m->np, mp = m->mp;
    const double sc = o->wi * o->gamma;
    /* landmark rows of all cross covariances: Ut_raw[q][r] = sum_{i <= r} S[i][r] DZ[i][q]  (S upper: K truncated) */
    const int MR = ((int)np + 95) / 96, MQ = (int)mp / 48;
    const int ra = m->r > 0, kr = m->r;
    const double *Sop = ra ? m->A : m->S;                                         /* rank-aware form: kept rows, permuted columns, K <= r */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int mr = MR - 1; mr >= 0; mr--)
        for (int mq = 0; mq < MQ; mq++) {
            const int r0 = mr * 96, q0 = mq * 48;
            const int nct = ((int)np - r0 < 96 ? (int)np - r0 : 96) / 24;
            int K = r0 + 24 * nct; if (K > n) K = n;
            if (ra && K > kr) K = kr;
            mt_macro_tn(K, m->DZ + q0, mp, Sop + r0, np, m->Ut + (size_t)q0 * np + r0, np, 6, nct, 1.0, 0, (ra && r0 + 24 * nct > kr) ? 0 : 2, r0, 0, 0, 0);
        }
    if (ra) {                                                                     /* the null rows' share: sqrt(EPSILON) DZ[i] into column i only */
        const double sqeps = sqrt(o->p.epsilon);
#pragma omp parallel for schedule(static)
        for (int b = kr; b < n; b++) {
            const int k = m->perm[b] / 6;
            for (int e = 0; e < 2; e++) m->Ut[(size_t)(2 * k + e) * np + b] += sqeps * m->DZ[(size_t)b * mp + 2 * k + e];
        }
    }
    /* robot rows of Pxy: sum_c w_c (r_c - X_r)(Z_c - h) */
    double *pr = (double *)calloc((size_t)4 * mp, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int q = 0; q < 2 * N; q++) {
        double acc[4] = { 0, 0, 0, 0 };
        for (int c = 0; c < L; c++) {
            const double w = c ? o->wi : o->wc0, dz = m->Z[(size_t)c * mp + q] - m->h[q];
            for (int e = 0; e < 4; e++) acc[e] += w * (m->sigR[(size_t)c * 4 + e] - o->X[n - 4 + e]) * dz;
        }
        for (int e = 0; e < 4; e++) pr[(size_t)e * mp + q] = acc[e];
    }
    /* per-landmark 2x2 constants */
    double *lk = (double *)calloc((size_t)8 * (N > 0 ? N : 1), sizeof(double));
    for (int k = 0; k < N; k++) {
        double sii[4]; inv2(m->Si + 4 * k, sii);
        const double v0 = z[2 * k] - m->h[2 * k], v1 = z[2 * k + 1] - m->h[2 * k + 1];
        const double a0 = m->Z[2 * k] - m->h[2 * k], a1 = m->Z[2 * k + 1] - m->h[2 * k + 1];
        double *l = lk + 8 * k;
        l[0] = sii[0]; l[1] = sii[1]; l[2] = sii[2]; l[3] = sii[3];
        l[4] = sii[0] * v0 + sii[2] * v1; l[5] = sii[1] * v0 + sii[3] * v1;              /* Si^-T (z - h) */
        l[6] = a0 * sii[0] + a1 * sii[2]; l[7] = a0 * sii[1] + a1 * sii[3];              /* (Z0 - h)^T Si^-1: centre term, wc0 != wm0 */
    }
    const double cw = o->wc0 - o->wm0;
#pragma omp parallel for schedule(static)
    for (int rq = 0; rq < n; rq++) {
        const int r = ra ? m->perm[rq] : rq;                                      /* state row; rq = its column of U^T */
        double dxr = 0.0;
        for (int k = 0; k < N; k++) {
            double *u0p = m->Ut + (size_t)(2 * k) * np + rq, *u1p = m->Ut + (size_t)(2 * k + 1) * np + rq;
            if (!(matched[k] && m->vis[k])) { *u0p = 0.0; *u1p = 0.0; continue; }
            const double *l = lk + 8 * k;
            double p0, p1;
            if (r < n - 4) { p0 = sc * *u0p; p1 = sc * *u1p; } else { p0 = pr[(size_t)(r - (n - 4)) * mp + 2 * k]; p1 = pr[(size_t)(r - (n - 4)) * mp + 2 * k + 1]; }
            const double u0 = p0 * l[0] + p1 * l[2] - cw * dxr * l[6], u1 = p0 * l[1] + p1 * l[3] - cw * dxr * l[7];
            *u0p = u0; *u1p = u1;
            dxr += u0 * l[4] + u1 * l[5];
        }
        o->X[r] += dxr;
    }
    for (int q = 2 * N; q < (int)mp; q++) memset(m->Ut + (size_t)q * np, 0, sizeof(double) * np);
    free(pr); free(lk);
}

/* ---- G = S^T S - U U^T (upper triangle, 24-wide tile rows) ---- */
static void mt_syrk(mt_state *m)
{
    const int n = m->n; const size_t np = m->np, mp = m->mp;
    const int MI = (int)np / 48, MJ = ((int)np + 95) / 96;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int mi = MI - 1; mi >= 0; mi--)
        for (int mj = 0; mj < MJ; mj++) {
            const int i0 = mi * 48, j0 = mj * 96;
            if (j0 + 96 <= i0) continue;                                      /* strictly below the diagonal */
            const int nct = ((int)np - j0 < 96 ? (int)np - j0 : 96) / 24;
            int K = i0 + 48; if (K > n) K = n;
            double *C = m->G + (size_t)i0 * np + j0;
            mt_macro_tn(K, m->S + i0, np, m->S + j0, np, C, np, 6, nct, 1.0, 0, 1, i0, 1, i0, j0);
            mt_macro_tn((int)mp, m->Ut + i0, np, m->Ut + j0, np, C, np, 6, nct, -1.0, 1, 0, 0, 1, i0, j0);
        }
}

/* ---- blocked right-looking modified Cholesky on the upper triangle, W[j][i] = C[i][j] in place; returns clamp rows ---- */
MT_CLONES static void mt_panel_rows(int nb, int j0, int i_beg, int i_end, double *restrict G, size_t np, const double *restrict D)
{
    /* rows j0..j0+nb of the columns [i_beg, i_end): forward substitution with the unit lower factor of the diagonal block */
    for (int j = j0; j < j0 + nb; j++) {
        const double *wj = G + (size_t)j * np;
        for (int k = j + 1; k < j0 + nb; k++) {
            const double f = wj[k] / D[j];
            double *wk = G + (size_t)k * np;
            for (int i = i_beg; i < i_end; i++) wk[i] -= f * wj[i];
        }
    }
}
static int mt_gmw(mt_state *m)
{
    const int n = m->n; const size_t np = m->np;
    double *G = m->G, *D = m->D; const double eps = m->o->p.epsilon;
    /* gamma, xi of the GMW bound (2204-2211) */
    double gmax = -INFINITY, xi = -INFINITY;
#pragma omp parallel for reduction(max : gmax, xi) schedule(static)
    for (int r = 0; r < n; r++) {
        if (G[(size_t)r * np + r] > gmax) gmax = G[(size_t)r * np + r];
        for (int c = r + 1; c < n; c++) if (G[(size_t)r * np + c] > xi) xi = G[(size_t)r * np + c];
    }
    xi = fmax(xi, 0.0);                                                       /* the reference's max runs over G - diag(diag G): zeros on the diagonal */
    const double nu = fmax(1.0, sqrt((double)n * n - 1.0)), beta2 = fmax(fmax(gmax, xi / nu), 1e-15);
    for (int j0 = 0; j0 < n; j0 += MT_NB) {
        const int nb = (n - j0 < MT_NB) ? n - j0 : MT_NB, t0 = j0 + MT_NB;
        /* (a) diagonal block: the reference's recurrence, pivots D_j = max(eps, |C_jj|)          2279-2295 */
        for (int j = j0; j < j0 + nb; j++) {
            double *wj = G + (size_t)j * np;
            const double dj = fmax(eps, fabs(wj[j]));
            D[j] = dj;
            for (int k = j + 1; k < j0 + nb; k++) {
                const double f = wj[k] / dj;
                double *wk = G + (size_t)k * np;
                for (int i = k; i < j0 + nb; i++) wk[i] -= f * wj[i];
            }
        }
        if (t0 >= n) break;
        /* (b) the panel's rows right of the diagonal block, in column slabs */
        const int ncol = (int)np - t0, slab = 96, nsl = (ncol + slab - 1) / slab;
#pragma omp parallel for schedule(static)
        for (int s = 0; s < nsl; s++) {
            const int ib = t0 + s * slab, ie = (ib + slab < (int)np) ? ib + slab : (int)np;
            mt_panel_rows(nb, j0, ib, ie, G, np, D);
            for (int j = j0; j < j0 + nb; j++) {                                              /* Lw = W / D */
                const double id = 1.0 / D[j];
                const double *wj = G + (size_t)j * np; double *lj = m->Lw + (size_t)(j - j0) * np;
                for (int i = ib; i < ie; i++) lj[i] = wj[i] * id;
            }
        }
        /* (c) trailing update: W[k][i] -= sum_j Lw[j][k] W[j][i],  k >= t0, i >= k */
        const int T8 = ((int)np - t0) / 8, T24 = ((int)np - t0) / 24;
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
        for (int tk = 0; tk < T8; tk++)
            for (int ti = 0; ti < T24; ti++) {
                const int k0 = t0 + tk * 8, i0 = t0 + ti * 24;
                if (i0 + 24 <= k0) continue;
                mt_tile_tn(nb, m->Lw + k0, np, G + (size_t)j0 * np + i0, np, G + (size_t)k0 * np + i0, np, -1.0, 1);
            }
    }
    /* S = sqrt(D) L^T; theta check after the fact (as k_gmw_check): would the third pivot candidate have won? */
    int clamp = 0;
#pragma omp parallel for schedule(static) reduction(+ : clamp)
    for (int j = 0; j < n; j++) {
        const double sd = sqrt(D[j]), is = 1.0 / sd;
        double *sj = m->S + (size_t)j * np; const double *wj = G + (size_t)j * np;
        double mx = 0.0;
        for (int i = 0; i < j; i++) sj[i] = 0.0;
        sj[j] = sd;
        for (int i = j + 1; i < n; i++) { const double v = wj[i] * is; sj[i] = v; if (fabs(v) > mx) mx = fabs(v); }
        for (int i = n; i < (int)np; i++) sj[i] = 0.0;
        const double th = mx * sd;
        if (th * th / beta2 > D[j]) clamp++;
    }
    return clamp;
}


/* ==== rank-aware form (the GPU path's default; cv-monoslam_amd/csrc/srukf_rank.hip): the structurally null pivots are not factored ====
 * Rows of S with energy below 1e-12 (never the robot's) are structurally null directions (anchors of jointly initialised landmarks are
 * copies of one robot position, SLAM.cpp:1223, 1247): the reference's modified Cholesky clamps their pivots to EPSILON (2279-2285).  The
 * refactorisation runs on the permuted matrix (kept indices first, same relative order), pivots only the r kept rows, carries all n columns
 * along, and writes sqrt(EPSILON) e_k for the rest; the contractions end at K = r.  Same rule, same check, same fallback as the device. */
#define MT_NULL_ENERGY 1e-12
static void mt_rebuild_A(mt_state *m)
{
    const int n = m->n, r = m->r; const size_t np = m->np;
#pragma omp parallel for schedule(static)
    for (int a = 0; a < r; a++) {
        const double *src = m->S + (size_t)m->perm[a] * np; double *dst = m->A + (size_t)a * np;
        for (int b = 0; b < (int)np; b++) dst[b] = (b >= a && b < n) ? src[m->perm[b]] : 0.0;
    }
}
ORC_API int mt_set_rank_aware(mt_state *m, int on)
{
    const int n = m->n; const size_t np = m->np;
    m->rank_aware = on ? 1 : 0; m->r = 0;
    if (!on || n < 128) return 0;
    int r = 0, nd = 0;
    int *drop = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n; k++) {
        double e = 0.0; for (int i = k; i < n; i++) e += m->S[(size_t)k * np + i] * m->S[(size_t)k * np + i];
        if (k < n - 4 && e < MT_NULL_ENERGY) drop[nd++] = k; else m->perm[r++] = k;
    }
    if (nd == 0) { free(drop); return 0; }
    for (int q = 0; q < nd; q++) m->perm[r + q] = drop[q];
    for (int k = n; k < (int)np; k++) m->perm[k] = k;
    for (int a = 0; a < (int)np; a++) m->iperm[m->perm[a]] = a;
    const double sqeps = sqrt(m->o->p.epsilon);
    for (int q = 0; q < nd; q++) { double *row = m->S + (size_t)drop[q] * np; memset(row, 0, sizeof(double) * np); row[drop[q]] = sqeps; }   /* k_rank_const_rows */
    free(drop);
    m->r = r;
    mt_rebuild_A(m);
    return nd;
}
ORC_API int mt_null_directions(const mt_state *m) { return m->r > 0 ? m->n - m->r : 0; }
ORC_API long long mt_rank_fallbacks(const mt_state *m) { return m->rank_fallbacks; }

/* Gp = A^T A - Up Up^T on the kept rows (a < r) x all permuted columns b >= a, K <= r; gdiag[b] for the dropped positions into D[b] (scratch) */
static void mt_syrk_ra(mt_state *m, double *gdiag)
{
    const int n = m->n, r = m->r; const size_t np = m->np, mp = m->mp;
    const int MI = (r + 47) / 48, MJ = ((int)np + 95) / 96;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int mi = MI - 1; mi >= 0; mi--)
        for (int mj = 0; mj < MJ; mj++) {
            const int i0 = mi * 48, j0 = mj * 96;
            if (j0 + 96 <= i0) continue;
            const int nct = ((int)np - j0 < 96 ? (int)np - j0 : 96) / 24;
            int K = i0 + 48; if (K > r) K = r;
            double *C = m->G + (size_t)i0 * np + j0;
            mt_macro_tn(K, m->A + i0, np, m->A + j0, np, C, np, 6, nct, 1.0, 0, (i0 + 48 > r) ? 0 : 1, i0, 1, i0, j0);
            mt_macro_tn((int)mp, m->Ut + i0, np, m->Ut + j0, np, C, np, 6, nct, -1.0, 1, 0, 0, 1, i0, j0);
        }
#pragma omp parallel for schedule(static)
    for (int b = r; b < n; b++) {
        double t = 0.0;
        for (int k = 0; k < r; k++) { const double v = m->A[(size_t)k * np + b]; t += v * v; }
        for (int q = 0; q < (int)mp; q++) { const double v = m->Ut[(size_t)q * np + b]; t -= v * v; }
        gdiag[b] = t;
    }
}
/* the blocked modified Cholesky of mt_gmw on the leading r pivots of Gp; then back to state order.  Returns clamp rows + failed null checks. */
static int mt_gmw_ra(mt_state *m, const double *gdiag)
{
    const int n = m->n, r = m->r; const size_t np = m->np;
    double *G = m->G, *D = m->D; const double eps = m->o->p.epsilon;
    double gmax = -INFINITY, xi = -INFINITY;
#pragma omp parallel for reduction(max : gmax, xi) schedule(static)
    for (int a = 0; a < n; a++) {
        if (a >= r) { if (gdiag[a] > gmax) gmax = gdiag[a]; continue; }
        if (G[(size_t)a * np + a] > gmax) gmax = G[(size_t)a * np + a];
        for (int c = a + 1; c < n; c++) if (G[(size_t)a * np + c] > xi) xi = G[(size_t)a * np + c];   /* kept rows x all columns: holds every value of the dropped block (copies) */
    }
    xi = fmax(xi, 0.0);
    const double nu = fmax(1.0, sqrt((double)n * n - 1.0)), beta2 = fmax(fmax(gmax, xi / nu), 1e-15);
    for (int j0 = 0; j0 < r; j0 += MT_NB) {
        const int nb = (r - j0 < MT_NB) ? r - j0 : MT_NB, t0 = j0 + nb;
        for (int j = j0; j < j0 + nb; j++) {
            double *wj = G + (size_t)j * np;
            const double dj = fmax(eps, fabs(wj[j]));
            D[j] = dj;
            for (int k = j + 1; k < j0 + nb; k++) {
                const double f = wj[k] / dj;
                double *wk = G + (size_t)k * np;
                for (int i = k; i < j0 + nb; i++) wk[i] -= f * wj[i];
            }
        }
        /* the panel's rows right of the diagonal block: all n columns are carried along */
        const int ncol = (int)np - t0, slab = 96, nsl = (ncol + slab - 1) / slab;
#pragma omp parallel for schedule(static)
        for (int s = 0; s < nsl; s++) {
            const int ib = t0 + s * slab, ie = (ib + slab < (int)np) ? ib + slab : (int)np;
            mt_panel_rows(nb, j0, ib, ie, G, np, D);
            for (int j = j0; j < j0 + nb; j++) {
                const double id = 1.0 / D[j];
                const double *wj = G + (size_t)j * np; double *lj = m->Lw + (size_t)(j - j0) * np;
                for (int i = ib; i < ie; i++) lj[i] = wj[i] * id;
            }
        }
        if (t0 >= r) break;                                   /* (a full panel: t0 = j0 + 48 keeps the 8 x 24 tile grid) */
        /* trailing update of the KEPT rows only: k in [t0, r), i >= k */
        const int T8 = (r - t0 + 7) / 8, T24 = ((int)np - t0) / 24;
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
        for (int tk = 0; tk < T8; tk++)
            for (int ti = 0; ti < T24; ti++) {
                const int k0 = t0 + tk * 8, i0 = t0 + ti * 24;
                if (i0 + 24 <= k0) continue;
                mt_tile_tn(nb, m->Lw + k0, np, G + (size_t)j0 * np + i0, np, G + (size_t)k0 * np + i0, np, -1.0, 1);
            }
    }
    /* checks BEFORE anything is written (the state before the refactorisation stays intact for the fallback): theta clamp of the reference on the kept
     * rows (as k_gmw_check / k_rank_expand do), and every skipped direction null in THIS frame's G: G_bb - sum_{a<r} Sp[a][b]^2 <= 1e-12 */
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int a = 0; a < r; a++) {
        const int j = m->perm[a];
        const double sd = sqrt(D[a]), is = 1.0 / sd;
        const double *wa = G + (size_t)a * np;
        double mx = 0.0;
        for (int b = a + 1; b < n; b++) if (m->perm[b] > j) { const double v = fabs(wa[b] * is); if (v > mx) mx = v; }
        const double th = mx * sd;
        if (th * th / beta2 > D[a]) bad++;
    }
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int b = r; b < n; b++) {
        double t = 0.0;
        for (int a = 0; a < r; a++) { const double v = G[(size_t)a * np + b]; t += v * v / D[a]; }
        if (gdiag[b] - t > MT_NULL_ENERGY) bad++;
    }
    return bad;
}
/* back to state order (k_rank_expand): kept rows are upper triangular there (what stands left of the diagonal belongs to a column that is a copy of an
 * earlier one: residual zero), dropped rows sqrt(EPSILON) e_k; the permuted copy A alongside */
static void mt_expand_ra(mt_state *m)
{
    const int n = m->n, r = m->r; const size_t np = m->np;
    const double *G = m->G, *D = m->D; const double eps = m->o->p.epsilon;
#pragma omp parallel for schedule(static)
    for (int a = 0; a < n; a++) {
        const int j = m->perm[a];
        double *sj = m->S + (size_t)j * np;
        memset(sj, 0, sizeof(double) * np);
        if (a >= r) { sj[j] = sqrt(eps); continue; }
        const double sd = sqrt(D[a]), is = 1.0 / sd;
        const double *wa = G + (size_t)a * np; double *aa = m->A + (size_t)a * np;
        for (int b = 0; b < a; b++) aa[b] = 0.0;
        aa[a] = sd; sj[j] = sd;
        for (int b = a + 1; b < n; b++) {
            const double v = wa[b] * is; const int c = m->perm[b];
            aa[b] = v;
            if (c > j) sj[c] = v;
        }
        for (int b = n; b < (int)np; b++) aa[b] = 0.0;
    }
}

ORC_API int mt_frame(mt_state *m, const double odo_prev[3], const double odo_cur[3], const double *z, const int *matched)
{
    const int n = m->n, N = m->N; const size_t np = m->np;
    double t0 = omp_get_wtime(), t1;
#define MT_LAP(i) do { t1 = omp_get_wtime(); m->t_phase[i] += t1 - t0; t0 = t1; } while (0)
    mt_motion(m, odo_prev, odo_cur);
    if (m->r > 0) {                                            /* the permuted copy follows the motion step: robot columns = permuted positions r-4 .. r-1 */
        for (int a = 0; a < m->r; a++) for (int e = 0; e < 4; e++) m->A[(size_t)a * np + (m->r - 4 + e)] = (m->perm[a] <= n - 4 + e) ? m->S[(size_t)m->perm[a] * np + (n - 4 + e)] : 0.0;
    }
    MT_LAP(0);
    mt_measure(m); MT_LAP(1);
    int nm = 0; for (int k = 0; k < N; k++) nm += matched[k] ? 1 : 0;
    if (nm == 0) return SRUKF_OK;
    mt_gain(m, z, matched); MT_LAP(2);
    if (m->r > 0) {
        double *gdiag = (double *)malloc(sizeof(double) * np);
        mt_syrk_ra(m, gdiag); MT_LAP(3);
        const int bad = mt_gmw_ra(m, gdiag);
        free(gdiag);
        if (bad == 0) { mt_expand_ra(m); MT_LAP(4); return SRUKF_OK; }
        /* a skipped direction was not null in this frame's S^T S - U U^T, or the theta clamp would have been active: the refactorisation is repeated on
         * the full-rank path (S is untouched; U^T back to state order), and the null set is taken again from its result */
        m->rank_fallbacks++;
#pragma omp parallel for schedule(static)
        for (int q = 0; q < (int)m->mp; q++) {
            double *tmp = m->Gbak + (size_t)q * np, *u = m->Ut + (size_t)q * np;
            memcpy(tmp, u, sizeof(double) * np);
            for (int b = 0; b < n; b++) u[m->perm[b]] = tmp[b];
        }
        m->r = 0;
    }
    const int redo_rank = m->rank_aware && m->r == 0;
    mt_syrk(m); MT_LAP(3);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; r++) memcpy(m->Gbak + (size_t)r * np + r, m->G + (size_t)r * np + r, sizeof(double) * (n - r));
    const int clamp_rows = mt_gmw(m); MT_LAP(4);
    if (clamp_rows > 0) {
        /* the theta clamp would have been active: redo this refactor exactly as the reference does */
        double *Gd = (double *)malloc(sizeof(double) * (size_t)n * n), *Sd = (double *)malloc(sizeof(double) * (size_t)n * n);
        for (int r = 0; r < n; r++) for (int c = r; c < n; c++) { Gd[(size_t)r * n + c] = m->Gbak[(size_t)r * np + c]; Gd[(size_t)c * n + r] = Gd[(size_t)r * n + c]; }
        orc_gmw(Gd, n, m->o->p.epsilon, Sd, NULL, NULL, &m->o->clamp_eps, &m->o->clamp_theta);
        for (int r = 0; r < n; r++) memcpy(m->S + (size_t)r * np + r, Sd + (size_t)r * n + r, sizeof(double) * (n - r));
        free(Gd); free(Sd);
        m->clamp_fallbacks++;
        MT_LAP(5);
    }
    if (redo_rank) mt_set_rank_aware(m, 1);
#undef MT_LAP
    return SRUKF_OK;
}

ORC_API int mt_run_frames(mt_state *m, int F, const double *odo, const double *z, const int *matched, double *traj)
{
    const int n = m->n, N = m->N; const size_t np = m->np;
    for (int f = 0; f < F; f++) {
        int rc = mt_frame(m, odo + 3 * f, odo + 3 * (f + 1), z + (size_t)f * 2 * N, matched + (size_t)f * N); if (rc) return rc;
        if (traj) {
            double *t = traj + 8 * f;
            for (int d = 0; d < 4; d++) t[d] = m->o->X[n - 4 + d];
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) {
                double acc = 0; for (int k = 0; k < n; k++) acc += m->S[k * np + (n - 4 + a)] * m->S[k * np + (n - 4 + b)];
                t[4 + 2 * a + b] = acc;
            }
        }
    }
    return SRUKF_OK;
}
