// srukf_map.hip — map changes on the device (srukf_add_landmarks: integrateFeaturesInformation, SLAM.cpp:818-871; srukf_delete_landmark: deleteOneFeature, 2637-2668)
// and data association (wrapPatch + dataAssociation, 1803-2009).  A map change rebuilds the context behind the handle (adopt_context).

#include "srukf_ctx.h"
#include <chrono>
using namespace srukf_impl;

// srukf_debug_set(0, "timing", 1): wall time of the phases of a map change on stderr (measurement: where do the milliseconds of srukf_add_landmarks /
// srukf_delete_landmark go — the device work or the context that is rebuilt around it?)
namespace {
struct MapTimer {
    bool on; const char* what; std::chrono::steady_clock::time_point t0, tl; std::string line;
    explicit MapTimer(const char* w) : on(g_dbg_timing.load() != 0), what(w), t0(std::chrono::steady_clock::now()), tl(t0) {}
    void mark(const char* label) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        char b[64]; snprintf(b, sizeof b, " %s %.0f", label, std::chrono::duration<double, std::micro>(t - tl).count()); line += b; tl = t;
    }
    ~MapTimer() { if (on) fprintf(stderr, "[map timing] %s: total %.0f us:%s\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), line.c_str()); }
};
}

static int ensure_appearance(srukf_ctx* c)
{
    if (c->app_patch) return SRUKF_OK;
    const size_t N = c->d.N > 0 ? c->d.N : 1;
    const size_t img = (size_t)c->p.image_w * c->p.image_h;
    HIPCHK(c, srukf_dmalloc((void**)&c->app_patch, N * srukf_app_patch_stride()));
    HIPCHK(c, srukf_dmalloc((void**)&c->app_tmpl, N * srukf_app_tmpl_stride()));
    HIPCHK(c, srukf_dmalloc((void**)&c->d_image, img));
    HIPCHK(c, srukf_dmalloc((void**)&c->appR, sizeof(double) * 9 * N));
    HIPCHK(c, srukf_dmalloc((void**)&c->appT, sizeof(double) * 3 * N));
    HIPCHK(c, srukf_dmalloc((void**)&c->appPx, sizeof(double) * 2 * N));
    HIPCHK(c, srukf_dmalloc((void**)&c->corr, sizeof(double) * N));
    HIPCHK(c, srukf_dmalloc((void**)&c->has_app, sizeof(int) * N));
    HIPCHK(c, hipMemsetAsync(c->app_patch, 0, N * srukf_app_patch_stride(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->app_tmpl, 0, N * srukf_app_tmpl_stride(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->has_app, 0, sizeof(int) * N, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SRUKF_OK;
}

// the appearance record of landmark `from` of `a` becomes the one of landmark `to` of `b` (map changes)
static void copy_appearance(srukf_ctx* a, int from, srukf_ctx* b, int to)
{
    const size_t ps = srukf_app_patch_stride(), ts = srukf_app_tmpl_stride();
    hipMemcpyAsync(b->app_patch + to * ps, a->app_patch + from * ps, ps, hipMemcpyDeviceToDevice, b->stream);
    hipMemcpyAsync(b->app_tmpl + to * ts, a->app_tmpl + from * ts, ts, hipMemcpyDeviceToDevice, b->stream);
    hipMemcpyAsync(b->appR + 9 * to, a->appR + 9 * from, sizeof(double) * 9, hipMemcpyDeviceToDevice, b->stream);
    hipMemcpyAsync(b->appT + 3 * to, a->appT + 3 * from, sizeof(double) * 3, hipMemcpyDeviceToDevice, b->stream);
    hipMemcpyAsync(b->appPx + 2 * to, a->appPx + 2 * from, sizeof(double) * 2, hipMemcpyDeviceToDevice, b->stream);
    hipMemcpyAsync(b->has_app + to, a->has_app + from, sizeof(int), hipMemcpyDeviceToDevice, b->stream);
}

namespace srukf_impl {

// the handle keeps its identity when the map changes size: swap the guts of a freshly built context in, keep the
// stream ownership and the profile, destroy the old buffers
void adopt_context(srukf_ctx* c, srukf_ctx* c2)
{
    const bool own = c->own_stream;
    std::vector<srukf_ctx*> kept = std::move(c->retired);       // (the handle's list of retired contexts stays with the handle)
    std::swap(*c, *c2);
    c->retired = std::move(kept); c2->retired.clear();
    c->spare_stage = c2->spare_stage; c->spare_stage_bytes = c2->spare_stage_bytes; c2->spare_stage = nullptr; c2->spare_stage_bytes = 0;      // (the handle's spare staging area too)
    c->own_stream = own; c2->own_stream = false;
    c->profiling = c2->profiling; c->use_graph = c2->use_graph;
    // per-context switches the caller set on the handle survive the rebuild (before srukf_set_storage / update_null_set run on it)
    c->rank_aware = c2->rank_aware; c->debug_allow_mixed = c2->debug_allow_mixed; c->debug_starve = c2->debug_starve; c->dbg = c2->dbg; c->split_off = c2->split_off;
    const int shared = c2->gmw_shared, tenants = c2->shared_tenants;
    memcpy(c->prof_ms, c2->prof_ms, sizeof c->prof_ms); memcpy(c->prof_n, c2->prof_n, sizeof c->prof_n);
    memcpy(c->prof_flops, c2->prof_flops, sizeof c->prof_flops); memcpy(c->prof_bytes, c2->prof_bytes, sizeof c->prof_bytes);
    c2->profiling = false; c2->pev.clear();
    ctx_retire(c, c2);                                           // (not destroyed: revived when the map has this size again — ctx_obtain)
    c->phase = 0;
    if (shared != c->gmw_shared) set_shared(c, shared, tenants);
}

}  // namespace srukf_impl

extern "C" {

// PointsMap::initPatch / initRotation / initTrans / initPixel as set at creation (SLAM.cpp:920-925): patch = the
// (2 HP_INIT + 1)^2 = 21 x 21 gray window image(Rect(round(u) - 10, round(v) - 10, 21, 21)), row-major as cv::Mat;
// R = Rwc (3x3 row-major), t = camera position, px = the distorted pixel.  matchPatch is zeroed (926).
int srukf_set_landmark_appearance(srukf_ctx* c, int k, const unsigned char* patch, const double R[9], const double t[3], const double px[2])
{
    if (!c || !patch || !R || !t || !px || k < 0 || k >= c->d.N) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_appearance(c); if (rc) return rc;
    const size_t ps = srukf_app_patch_stride(), ts = srukf_app_tmpl_stride();
    const int one = 1;
    HIPCHK(c, hipMemcpy(c->app_patch + k * ps, patch, 441, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->app_tmpl + k * ts, 0, ts));
    HIPCHK(c, hipMemcpy(c->appR + 9 * k, R, sizeof(double) * 9, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->appT + 3 * k, t, sizeof(double) * 3, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->appPx + 2 * k, px, sizeof(double) * 2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->has_app + k, &one, sizeof(int), hipMemcpyHostToDevice));
    return SRUKF_OK;
}

int srukf_get_match_patch(srukf_ctx* c, int k, unsigned char* out)
{
    if (!c || !out || k < 0 || k >= c->d.N) return SRUKF_ERR_BAD_ARG;
    if (!c->app_tmpl) { c->err = "get_match_patch: no appearance records"; return SRUKF_ERR_SEQUENCE; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->app_tmpl + (size_t)k * srukf_app_tmpl_stride(), 289, hipMemcpyDeviceToHost));
    return SRUKF_OK;
}

// wrapPatch + dataAssociation (SLAM.cpp:1803-2009) between srukf_predict_measurement and srukf_update: gray = the
// image_h x image_w frame (row-major uchar).  Out (host, any may be NULL): z[2N] = matchLocation, matched[N] =
// isMatching, corr[N] = best normalised cross correlation.  Landmarks without an appearance record never match.
int srukf_associate(srukf_ctx* c, const unsigned char* gray, double* z, int* matched, double* corr)
{
    if (!c || !gray) return SRUKF_ERR_BAD_ARG;
    if (c->phase < 2) { c->err = "associate before predict_measurement"; return SRUKF_ERR_SEQUENCE; }
    const int N = c->d.N;
    if (N == 0) return SRUKF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_appearance(c); if (rc) return rc;
    const size_t img = (size_t)c->p.image_w * c->p.image_h;
    const size_t mp = c->d.mp, zm = mp + ((size_t)N + 1) / 2;             // zcur | mcur: one device allocation of zm doubles
    // What does not need the frame goes out first — the warp uses the PREDICTED robot pose (wrapPatch reads m_X_k after predictMotion, SLAM.cpp:1812-1830) — and runs while
    // the host copies the frame into pinned memory: a hipMemcpyAsync from the caller's pageable buffer is staged by the runtime, synchronously, in front of everything.
    step_commit_motion(c);
    double* dxyz = c->G;                                                 // G is free outside the refactorisation
    srukf_launch_landmarks_cartesian(c->stream, c->d, c->X, c->S, dxyz, nullptr);                           // PointsMap::xyz (2574): the points only, no covariances
    srukf_launch_warp_patch(c->stream, c->d, c->p, c->X, dxyz, c->h, c->appR, c->appT, c->appPx, c->app_patch, c->has_app, c->app_tmpl);
    double* hs = c->hstage;
    const size_t img_off = (zm + (size_t)N + 7) & ~(size_t)7;           // (doubles) behind the results' place in the staging buffer
    if ((img_off + (img + 7) / 8) * sizeof(double) <= c->hstage_bytes) {
        unsigned char* himg = (unsigned char*)(hs + img_off);
        memcpy(himg, gray, img);
        HIPCHK(c, hipMemcpyAsync(c->d_image, himg, img, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_image, gray, img, hipMemcpyHostToDevice, c->stream));
    }
    srukf_launch_associate(c->stream, c->d, c->p, c->d_image, c->h, c->Si, c->vis, c->has_app, c->app_tmpl, c->zcur, c->mcur, c->corr);
    // z | matched | corr written into the pinned buffer by ONE short launch, flag behind them: three small device-to-host copies were three blit kernels with their gaps
    const unsigned long long seq = ++c->step_seq;
    launch_export(c->stream, c->zcur, sizeof(double) * zm, c->corr, sizeof(double) * N, hs, c->dbg.step_spin ? step_flag(c) : nullptr, seq);
    rc = step_wait_export(c, seq); if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    if (z) memcpy(z, hs, sizeof(double) * 2 * N);
    if (matched) memcpy(matched, hs + mp, sizeof(int) * N);
    if (corr) memcpy(corr, hs + zm, sizeof(double) * N);
    return SRUKF_OK;
}

// integrateFeaturesInformation, numeric part (SLAM.cpp:826-871): K new landmarks at the distorted pixels uv[K][2] are
// appended to the map (normal order: before the robot block).  The context is rebuilt for N + K landmarks in place
// (the handle stays valid; staged sequences and captured graphs are dropped) and K_new = K is armed for the
// FLAG_4_NEED_REORDER update that follows (SLAM.cpp:2083-2090).  See srukf_augment.hip.
int srukf_add_landmarks(srukf_ctx* c, int K, const double* uv)
{
    if (!c || K < 1 || !uv) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    MapTimer mt("add_landmarks");
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mt.mark("sync");
    const int dim = c->d.n, ld = c->d.np;
    const int Na = dim + 3 * K, L = 2 * Na + 1, dimn = dim + 6 * K;                          // 827-828
    srukf_ctx* c2 = nullptr;
    int rc = ctx_obtain(c, &c2, c->d.N + K);
    if (rc) { c->err = std::string("add_landmarks: ") + g_create_error; return rc; }
    mt.mark("create");
    const int ldn = c2->d.np, rows_p = round_up(2 * Na, 16);
    KWeights wa; host_weights(Na, c->p, wa);                                                 // 867
    std::vector<int> perm(dimn);
    {   // getPermutationMatrix, 1303-1334 (dim = new dimension)
        const int dimOld = dimn - 6 * K;
        for (int i = 0; i < dimOld - 4; i++) perm[i] = i;
        for (int e = 0; e < 4; e++) perm[dimn - 4 + e] = dimOld - 4 + e;
        for (int id = 0; id < K; id++) {
            for (int e = 0; e < 3; e++) perm[dimOld - 4 + 6 * id + e] = dimOld + 3 * K + 3 * id + e;
            for (int e = 0; e < 3; e++) perm[dimOld - 4 + 6 * id + 3 + e] = dimOld + 3 * id + e;
        }
    }
    double *d_uv = nullptr, *d_ang = nullptr, *d_A = nullptr, *d_mu = nullptr; int* d_perm = nullptr;
    auto cleanup = [&]() { for (void* b : { (void*)d_uv, (void*)d_ang, (void*)d_A, (void*)d_mu, (void*)d_perm }) if (b) srukf_dfree(b); };
    if (srukf_dmalloc((void**)&d_uv, sizeof(double) * 2 * K) != hipSuccess || srukf_dmalloc((void**)&d_ang, sizeof(double) * (size_t)L * 3 * K) != hipSuccess ||
        srukf_dmalloc((void**)&d_A, sizeof(double) * (size_t)rows_p * ldn) != hipSuccess || srukf_dmalloc((void**)&d_mu, sizeof(double) * 3 * K) != hipSuccess ||
        srukf_dmalloc((void**)&d_perm, sizeof(int) * dimn) != hipSuccess) {
        cleanup(); srukf_destroy(c2); c->err = "add_landmarks: out of device memory"; return SRUKF_ERR_NOMEM;
    }
    hipMemcpyAsync(d_uv, uv, sizeof(double) * 2 * K, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * dimn, hipMemcpyHostToDevice, c->stream);
    hipStreamSynchronize(c->stream);                                                         // uv / perm are pageable host memory
    srukf_launch_aug_map(c->stream, c->p, dim, ld, K, Na, wa.gamma, c->X, c->S, d_uv, d_ang);
    srukf_launch_aug_x(c->stream, dim, K, Na, wa.wm0, wa.wi, c->X, d_ang, d_perm, d_mu, c2->X, dimn, ldn);
    srukf_launch_aug_build(c->stream, dim, ld, K, Na, wa.gamma, wa.wi_sr, c->X, c->S, d_ang, d_A, rows_p, dimn, ldn);
    srukf_launch_gram(c->stream, rows_p, ldn, d_A, c2->G);                                   // A^T A, disordered layout
    for (int slow = 0; slow < 2; slow++) {
        launch_set_frame(c->stream, c2->fs, 0, 1);
        launch_refactor_reset(c->stream, ldn, c2->theta, c2->fs, 1);
        launch_sym_permute(c->stream, dimn, ldn, c2->G, ldn, c2->Gbak, d_perm);      // Pi (A^T A) Pi^T
        srukf_launch_gmw_stats(c->stream, dimn, ldn, c2->Gbak, c2->fs);
        if (slow) hipMemsetAsync(c2->S, 0, sizeof(double) * (size_t)ldn * ldn, c->stream);
        run_gmw(c2, c2->Gbak, c2->S, slow != 0);
        if (slow) break;
        rc = read_fs(c2);
        if (rc) { c->err = c2->err; cleanup(); srukf_destroy(c2); return rc; }
        if (c2->hfs->clamp_rows == 0) break;
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    cleanup();
    mt.mark("numeric");
    if (e != hipSuccess) { srukf_destroy(c2); c->err = std::string("add_landmarks: ") + hipGetErrorString(e); return SRUKF_ERR_HIP; }
    if (c->app_patch) {                                      // the old landmarks keep their appearance records
        rc = ensure_appearance(c2);
        if (rc) { c->err = c2->err; srukf_destroy(c2); return rc; }
        for (int k = 0; k < c->d.N; k++) copy_appearance(c, k, c2, k);
        hipStreamSynchronize(c->stream);
    }
    const int storage = c->storage;
    mt.mark("appearance");
    adopt_context(c, c2);
    mt.mark("adopt+destroy");
    rc = srukf_set_storage(c, storage); if (rc) return rc;
    mt.mark("set_storage");
    rc = update_null_set(c); if (rc) return rc;
    canonicalize_null_rows(c);
    mt.mark("null_set");
    return srukf_set_new_landmarks(c, K);
}

// deleteOneFeature, numeric part (SLAM.cpp:2637-2668): landmark id (0-based, state order) leaves the state.  The
// reference drops its 6 rows and columns from S and folds the 6 removed rows V back in with six
// S <- gmw(S^T S + v v^T); the sum of those is the remaining block of P = S^T S, so the device takes S^T S
// (k_syrk), compacts it and factors it once (the batched form of the six updates, as in srukf_update).
int srukf_delete_landmark(srukf_ctx* c, int id)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    const int N = c->d.N, n = c->d.n, np = c->d.np;
    if (id < 0 || id >= N) { c->err = "delete_landmark: no such landmark"; return SRUKF_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    MapTimer mt("delete_landmark");
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mt.mark("sync");
    srukf_ctx* c2 = nullptr;
    int rc = ctx_obtain(c, &c2, N - 1);
    if (rc) { c->err = std::string("delete_landmark: ") + g_create_error; return rc; }
    mt.mark("create");
    const int nn = n - 6, ldn = c2->d.np;
    std::vector<int> map(nn);
    for (int a = 0; a < nn; a++) map[a] = a < 6 * id ? a : a + 6;
    int* d_map = nullptr;
    if (srukf_dmalloc((void**)&d_map, sizeof(int) * nn) != hipSuccess) { srukf_destroy(c2); c->err = "delete_landmark: out of device memory"; return SRUKF_ERR_NOMEM; }
    hipMemcpy(d_map, map.data(), sizeof(int) * nn, hipMemcpyHostToDevice);
    launch_refactor_reset(c->stream, np, c->theta, c->fs, 1);
    srukf_launch_syrk(c->stream, c->d, c->S, c->Ut, 0, 0, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, nullptr, c->X, RankArgs{}, nullptr);   // P = S^T S
    launch_gather(c->stream, nn, ldn, c->X, c2->X, d_map);
    for (int slow = 0; slow < 2; slow++) {
        launch_set_frame(c->stream, c2->fs, 0, 1);
        launch_refactor_reset(c->stream, ldn, c2->theta, c2->fs, 1);
        launch_sym_permute(c->stream, nn, ldn, c->G, np, c2->Gbak, d_map);
        srukf_launch_gmw_stats(c->stream, nn, ldn, c2->Gbak, c2->fs);
        hipMemsetAsync(c2->S, 0, sizeof(double) * (size_t)ldn * ldn, c->stream);
        run_gmw(c2, c2->Gbak, c2->S, slow != 0);
        if (slow) break;
        rc = read_fs(c2);
        if (rc) { c->err = c2->err; srukf_dfree(d_map); srukf_destroy(c2); return rc; }
        if (c2->hfs->clamp_rows == 0) break;
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    srukf_dfree(d_map);
    mt.mark("numeric");
    if (e != hipSuccess) { srukf_destroy(c2); c->err = std::string("delete_landmark: ") + hipGetErrorString(e); return SRUKF_ERR_HIP; }
    // m_nFilters-- when one of the landmarks added last is the one that goes (SLAM.cpp:2468-2492)
    const int k_new = c->K_new > 0 ? (id >= N - c->K_new ? c->K_new - 1 : c->K_new) : 0;
    if (c->app_patch) {
        rc = ensure_appearance(c2);
        if (rc) { c->err = c2->err; srukf_destroy(c2); return rc; }
        for (int k = 0, a = 0; k < N; k++) if (k != id) copy_appearance(c, k, c2, a++);
        hipStreamSynchronize(c->stream);
    }
    const int storage = c->storage;
    mt.mark("appearance");
    adopt_context(c, c2);
    mt.mark("adopt+destroy");
    rc = srukf_set_storage(c, storage); if (rc) return rc;
    mt.mark("set_storage");
    rc = update_null_set(c); if (rc) return rc;
    canonicalize_null_rows(c);
    mt.mark("null_set");
    return srukf_set_new_landmarks(c, k_new);
}

}  // extern "C"
