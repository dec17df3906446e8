// srukf_api.hip — C-ABI (include/srukf.h) over the gfx950 kernels: context lifetime and HBM buffers, state accessors, storage, profiling read-out (the step-wise
// calls of a frame — predictMotion / predictMeasurement / KalmanUpdate, SLAM.cpp:87-112 — and their fast path: srukf_step.hip).  No CPU fallback: every numeric
// result is produced by the kernels in srukf_predict.hip / srukf_factor.hip / srukf_gmw_persist.hip / srukf_rank.hip.  The launch sequences of a frame and the
// staged replay are in srukf_replay.hip, the batched replay in srukf_batch.hip, map changes and data association in srukf_map.hip, the split form's plumbing in
// srukf_split.hip, the debug hooks in srukf_debug.hip (round 4 had all of it in this file).

#include "srukf_ctx.h"
using namespace srukf_impl;

// ---- device memory: the stream-ordered pool of the device instead of hipMalloc / hipFree ------------------------------
// A context is rebuilt whenever the map changes size (srukf_add_landmarks / srukf_delete_landmark): ~25 buffers freed and
// ~25 allocated.  hipMalloc / hipFree go to the driver every time (and hipFree synchronises the whole device): 12.6 ms per
// augmentation at N = 200, almost all of it there.  The default memory pool keeps freed blocks (release threshold raised
// to "never") and hands them out again in microseconds.  Allocation is made visible to every stream by synchronising the
// null stream it is ordered on; every free below happens after the streams that used the block have been synchronised.

namespace srukf_impl {

void srukf_pool_init()
{
    static std::mutex mu;                                      // a second thread must not get its first allocation before the first has raised the threshold
    static unsigned long long done_mask = 0;                   // one bit per device
    int dev = 0; hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    const unsigned long long bit = 1ull << dev;
    std::lock_guard<std::mutex> lk(mu);
    if (done_mask & bit) return;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
        unsigned long long keep = ~0ull;
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    done_mask |= bit;
}

hipError_t srukf_dmalloc_raw(void** p, size_t bytes)
{
    srukf_pool_init();
    hipError_t e = hipMallocAsync(p, bytes ? bytes : 8, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

hipError_t srukf_dfree(void* p) { return p ? hipFreeAsync(p, nullptr) : hipSuccess; }
hipError_t srukf_dfree_on(void* p, hipStream_t st) { return p ? hipFreeAsync(p, st) : hipSuccess; }

thread_local std::string g_create_error;

}  // namespace srukf_impl

static thread_local double* g_spare_stage = nullptr;       // one pinned staging buffer handed from a destroyed context to the next one

static thread_local size_t g_spare_stage_bytes = 0;

namespace srukf_impl {

const char* const kclass_name[KC_COUNT] = { "k_motion", "k_project", "k_meas_stats", "k_pxy", "k_gain", "k_syrk",
                                             "k_gmw_step64", "k_gmw_persist", "k_gmw_check", "k_gmw_col", "k_rank_expand", "k_project_motion", "k_project_table", "k_pxy2",
                                             "misc" };

void gmw_plan_destroy(GmwPlan& g, hipStream_t st)
{
    if (g.pans) srukf_dfree_on(g.pans, st);
    if (g.sync) srukf_dfree_on(g.sync, st);
    if (g.tiles) srukf_dfree_on(g.tiles, st);
    g = GmwPlan();
}

}  // namespace srukf_impl

// workers = -1 afterwards: the matrix has more tiles than resident workgroups can own (the per-panel launches are used)
// tenants = 2: the plan of a filter that shares the GPU (gmw_shared = 1): at most half the CUs, so that two admitted launches are resident together
// XCD-aware order of the tile list when every worker owns ONE tile.  Workgroup b runs on XCD b % 8, and
// in the fused replay the owner of tile (I, J) streams the operand columns of blocks I and J through its XCD's L2: with the tiles dealt out in
// list order every XCD touches every column block (8 copies of the 9.8 MB operand set through 4 MB L2s — the 17 bandwidth-bound us at the head
// of the launch).  Here the tiles whose owners compute them are cut into 8 compact 2D regions (two bands of block rows x four ranges of block
// columns), one per XCD; the others (head rows, pass-on row) fill the XCDs up to equal counts.  Which worker owns which tile changes nothing else.
static void gmw_tiles_xcd_order(std::vector<short>& tk, int ntiles_all, int workers, int T, int Tp)
{
    const int ntiles = ntiles_all - ((Tp > 0 && Tp < T) ? T - Tp : 0);         // the pass-on tiles stay at the end of the list (third slot of the first workers)
    if (ntiles > workers || ntiles < 16) return;
    struct Tl { short v[4]; };
    std::vector<Tl> comp, rest;
    const int h0 = srukf_gmw_head_rows() / 64, hx = h0 + srukf_gmw_head_extra_diag();
    for (int q = 0; q < ntiles; q++) {
        Tl t; for (int e = 0; e < 4; e++) t.v[e] = tk[4 * q + e];
        const int I = t.v[0], J = t.v[1];
        const bool computes = I >= h0 && !(I < hx && J < hx) && !(Tp < T && I == Tp);
        (computes ? comp : rest).push_back(t);
    }
    if (comp.size() < 16) return;
    // two bands of block rows with about half of the computed tiles each, each band in J-major order cut into four ranges
    std::sort(comp.begin(), comp.end(), [](const Tl& a, const Tl& b) { return a.v[0] != b.v[0] ? a.v[0] < b.v[0] : a.v[1] < b.v[1]; });
    size_t cut = comp.size() / 2;
    while (cut < comp.size() && cut > 0 && comp[cut].v[0] == comp[cut - 1].v[0]) cut++;      // bands end at row boundaries
    std::vector<std::vector<Tl>> grp(8);
    for (int band = 0; band < 2; band++) {
        std::vector<Tl> b(comp.begin() + (band ? cut : 0), band ? comp.end() : comp.begin() + cut);
        std::sort(b.begin(), b.end(), [](const Tl& x, const Tl& y) { return x.v[1] != y.v[1] ? x.v[1] < y.v[1] : x.v[0] < y.v[0]; });
        for (size_t q = 0; q < b.size(); q++) grp[4 * band + std::min<size_t>(3, q * 4 / b.size())].push_back(b[q]);
    }
    // positions of XCD x: list index w with (w + 1) % 8 == x (blockIdx = w + 1: the pivot is workgroup 0)
    int cap[8] = { 0 };
    for (int w = 0; w < ntiles; w++) cap[(w + 1) % 8]++;
    std::vector<Tl> spill(rest);
    for (int x = 0; x < 8; x++) while ((int)grp[x].size() > cap[x]) { spill.push_back(grp[x].back()); grp[x].pop_back(); }
    for (int x = 0; x < 8; x++) while ((int)grp[x].size() < cap[x] && !spill.empty()) { grp[x].push_back(spill.back()); spill.pop_back(); }
    size_t pos[8] = { 0 };
    for (int w = 0; w < ntiles; w++) {
        const int x = (w + 1) % 8;
        const Tl t = grp[x][pos[x]++];
        for (int e = 0; e < 4; e++) tk[4 * w + e] = t.v[e];
    }
}

namespace srukf_impl {

int gmw_plan_create(GmwPlan& g, int np, hipStream_t st, int Tp, int tenants)
{
    g.T = np / 64;
    g.Tp = (Tp > 0 && Tp < g.T) ? Tp : g.T;
    g.tenants = tenants > 1 ? tenants : 1;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2) cus = 2;
    g.cus = cus;
    const int cap = cus / (tenants > 1 ? tenants : 1) - 1;    // one workgroup per CU (registers), all of them resident
    g.workers = cap >= 1 ? srukf_gmw_persist_workers(g.T, g.Tp, cap) : -1;
    g.ntiles = srukf_gmw_build_tiles(g.T, g.Tp, nullptr);
    g.nreal = g.ntiles - (g.Tp < g.T ? g.T - g.Tp : 0);
    std::vector<short> tk((size_t)4 * (g.ntiles > 0 ? g.ntiles : 1), 0);
    srukf_gmw_build_tiles(g.T, g.Tp, tk.data());
    gmw_tiles_xcd_order(tk, g.ntiles, g.workers, g.T, g.Tp);
    const size_t sync_bytes = (size_t)srukf_gmw_sync_bytes(g.T);
    if (srukf_dmalloc_on(&g.pans, (size_t)srukf_gmw_panel_bytes() * g.T, st) != hipSuccess ||
        srukf_dmalloc_on(&g.sync, sync_bytes, st) != hipSuccess ||
        srukf_dmalloc_on(&g.tiles, sizeof(short) * tk.size(), st) != hipSuccess) { gmw_plan_destroy(g, st); return SRUKF_ERR_NOMEM; }
    const unsigned long long epoch1 = 1;                         // everything else starts at zero
    if (hipMemsetAsync(g.pans, 0, (size_t)srukf_gmw_panel_bytes() * g.T, st) != hipSuccess ||
        hipMemsetAsync(g.sync, 0, sync_bytes, st) != hipSuccess ||
        hipMemcpyAsync((char*)g.sync + offsetof(GmwSync, epoch), &epoch1, sizeof epoch1, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(g.tiles, tk.data(), sizeof(short) * tk.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { gmw_plan_destroy(g, st); return SRUKF_ERR_HIP; }
    return SRUKF_OK;
}

void host_weights(int Na, const srukf_params& p, KWeights& w)
{
    // calculateSampleParameter, SLAM.cpp:1050-1103
    const double alpha = p.ut_alpha, beta = p.ut_beta;
    const double Lammda = alpha * alpha * Na - Na;
    switch (p.weight_type) {
    case 0:
        w.wm0 = 1.0 - Na / 3.0; w.wc0 = 1.0 - Na / 3.0; w.wi = (1.0 - w.wc0) / (2 * Na); w.wi_sr = sqrt(w.wi);
        w.gamma = sqrt(Na / (1.0 - w.wm0));
        break;
    case 1:
        w.gamma = sqrt(Na + Lammda); w.wm0 = Lammda / (Na + Lammda); w.wc0 = w.wm0 + (1 - alpha * alpha + beta);
        w.wi = 1.0 / (2 * (Na + Lammda)); w.wi_sr = sqrt(fabs(w.wi));
        break;
    default:
        w.gamma = sqrt(3.0 * Na / 2.0); w.wm0 = 1.0 / 3.0; w.wc0 = 1.0 / 3.0; w.wi = 1.0 / (3.0 * Na); w.wi_sr = sqrt(w.wi);
        break;
    }
}

void prof_collect(srukf_ctx* c)
{
    for (auto& e : c->pev) {
        float ms = 0.f;
        hipEventSynchronize(e.b);
        hipEventElapsedTime(&ms, e.a, e.b);
        c->prof_ms[e.kc] += ms; c->prof_n[e.kc] += 1;
        hipEventDestroy(e.a); hipEventDestroy(e.b);
    }
    c->pev.clear();
}

}  // namespace srukf_impl

// ---- XCD-aware tile order -------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group, MI355X_MICROARCH.md);
// each XCD has its own 4 MiB L2 and S (11.6 MB at N = 200) does not fit one.  Give every XCD a fixed set of
// operand panels (tile rows `own`) and walk the other index in the same order on all XCDs, so that an
// own-panel stays L2-resident and the streamed panel is shared by the tiles that run next to each other.
//   own(t) = t % 8;  list per XCD: for other = 0.. : for own-tiles of this XCD: (own, other) if valid.
// The table maps linear workgroup id -> tile; unused slots hold (-1, -1).

namespace srukf_impl {

std::vector<int> build_tile_table(int n_own, int n_other, bool upper, bool own_is_row, int k_index /* 0: K grows with tile.x, 1: with tile.y */)
{
    std::vector<std::vector<int>> lst(8);
    for (int x = 0; x < 8; x++)
        for (int o = 0; o < n_other; o++)
            for (int w = x; w < n_own; w += 8) {
                const int r = own_is_row ? w : o, c = own_is_row ? o : w;
                if (upper && c < r) continue;
                lst[x].push_back(r); lst[x].push_back(c);
            }
    {
        // longest K first (S is upper triangular, so the K range grows with the state tile index): the long tiles
        // must not be the last ones dispatched
        for (auto& l : lst) {
            std::vector<std::pair<int, int>> t;
            for (size_t q = 0; q < l.size() / 2; q++) t.push_back({ l[2 * q], l[2 * q + 1] });
            std::stable_sort(t.begin(), t.end(), [k_index](const std::pair<int, int>& a, const std::pair<int, int>& b) {
                return k_index ? a.second > b.second : a.first > b.first; });
            for (size_t q = 0; q < t.size(); q++) { l[2 * q] = t[q].first; l[2 * q + 1] = t[q].second; }
        }
    }
    size_t mx = 0;
    for (auto& l : lst) mx = l.size() / 2 > mx ? l.size() / 2 : mx;
    std::vector<int> tab(mx * 8 * 2, -1);
    for (int x = 0; x < 8; x++)
        for (size_t q = 0; q < lst[x].size() / 2; q++) { tab[(q * 8 + x) * 2] = lst[x][2 * q]; tab[(q * 8 + x) * 2 + 1] = lst[x][2 * q + 1]; }
    return tab;
}

}  // namespace srukf_impl

static int alloc_zero(srukf_ctx* c, void** p, size_t bytes)
{
    HIPCHK(c, srukf_dmalloc_on(p, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
    return SRUKF_OK;
}

#define ALLOC(ptr, count) do { int rc_ = alloc_zero(c, (void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)); if (rc_) { g_create_error = c->err; srukf_destroy(c); return rc_; } } while (0)

static int block_cov(srukf_ctx* c, int off, int bs, double* out)
{
    srukf_launch_block_cov(c->stream, c->d, c->S, off, bs, c->small, nullptr);
    HIPCHK(c, hipMemcpyAsync(c->hstage, c->small, sizeof(double) * bs * bs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->hstage, sizeof(double) * bs * bs);
    return SRUKF_OK;
}

extern "C" {

int srukf_abi_version(void) { return SRUKF_ABI_VERSION; }

int srukf_default_params(srukf_params* p)
{
    if (!p) return SRUKF_ERR_BAD_ARG;
    memset(p, 0, sizeof *p);
    p->cam_dx = 0.0028; p->cam_dy = 0.0028; p->cam_cx = 310.1129; p->cam_cy = 236.7526;
    p->cam_k1 = 0.0001; p->cam_k2 = 0.0; p->cam_f = 2.1735; p->image_w = 640; p->image_h = 480;
    p->a1 = p->a2 = p->a3 = p->a4 = 8.0; p->sigma_measure = 3.0; p->rho0 = 1.0 / 3.0; p->sigma_rho = p->rho0 / 2.0;
    p->sigma_x = 0.02; p->sigma_y = 0.02; p->sigma_z = 0.005; p->sigma_theta = 0.02;
    p->epsilon = 1e-13; p->ut_alpha = 1e-3; p->ut_beta = 2.0;
    p->weight_type = 0; p->noise_type = 0; p->newton_iters = 100;
    return SRUKF_OK;
}

const char* srukf_last_error(const srukf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int srukf_create(srukf_ctx** out, int N, const srukf_params* p, int device, void* stream)
{
    if (!out || !p || N < 0) { g_create_error = "bad argument"; return SRUKF_ERR_BAD_ARG; }      // N = 0: the robot block only (SLAM.cpp:226-231), until landmarks are added
    // the per-group counters of the measurement statistics (FrameScalars::stat_cnt) serve (N + 31) / 32 <= 64 landmark groups
    if (N > 32 * SRUKF_STAT_GROUPS) { g_create_error = "more than 2048 landmarks: the per-group statistics counters (FrameScalars::stat_cnt) serve 64 groups of 32"; return SRUKF_ERR_UNSUPPORTED; }
    if (p->noise_type != 0) { g_create_error = "noise_type != 0 draws random numbers (SLAM.cpp:1505-1516) and is not built"; return SRUKF_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_error = "no HIP device (this library has no CPU fallback)";
        return SRUKF_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return SRUKF_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return SRUKF_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return SRUKF_ERR_NO_DEVICE;
    }
    srukf_ctx* c = new srukf_ctx();
    c->device = device; c->p = *p;
    if (!g_dbg_graphs) c->use_graph = false;                 // eager launches (profilers): srukf_debug_set(0, "graphs", 0)
    memset(c->prof_ms, 0, sizeof c->prof_ms); memset(c->prof_n, 0, sizeof c->prof_n);
    memset(c->prof_flops, 0, sizeof c->prof_flops); memset(c->prof_bytes, 0, sizeof c->prof_bytes);
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return SRUKF_ERR_HIP; }
        c->own_stream = true;
    }
    KDims& d = c->d;
    d.N = N; d.n = 6 * N + 4; d.Na = d.n + 5; d.L = 2 * d.Na + 1;
    d.np = round_up(d.n, SRUKF_PAD); d.mp = round_up(2 * N > 0 ? 2 * N : 1, SRUKF_PAD);
    host_weights(d.Na, c->p, c->w);
    const size_t np = d.np, mp = d.mp;
    ALLOC(c->X, np); ALLOC(c->S, np * np); ALLOC(c->G, np * np); ALLOC(c->Gbak, np * np); ALLOC(c->Wf, np * np);
    ALLOC(c->sigR, (size_t)d.L * 8 + 8); ALLOC(c->mpart, srukf_meas_part_doubles(d.mp)); ALLOC(c->dxp, srukf_gain_part_doubles(d.np)); ALLOC(c->Cmat, (np + 64) * 4); ALLOC(c->Z, (size_t)d.L * mp); ALLOC(c->DZ, np * mp);
    ALLOC(c->Ut, mp * np); ALLOC(c->PxyR, 5 * mp);
    // h | Si | visible in ONE allocation, z | matched in another: what srukf_predict_measurement hands to the host / srukf_update takes from it is one copy each
    { const size_t Nn = (size_t)(N > 0 ? N : 1); ALLOC(c->h, mp + 4 * Nn + (Nn + 1) / 2); c->Si = c->h + mp; c->vis = (int*)(c->h + mp + 4 * Nn);
      ALLOC(c->zcur, mp + (Nn + 1) / 2); c->mcur = (int*)(c->zcur + mp); }                       // rows 0..3: robot rows of the cross covariances; row 4: scratch of the "fused tail" statistics
    ALLOC(c->D, np); ALLOC(c->odocur, 8); ALLOC(c->small, 64);
    ALLOC(c->theta, np); ALLOC(c->fs, 1);
    { char* pb0 = nullptr; char* pb1 = nullptr; ALLOC(pb0, srukf_gmw_panel_bytes()); ALLOC(pb1, srukf_gmw_panel_bytes()); c->pan[0] = pb0; c->pan[1] = pb1; }
    { const int rcg = gmw_plan_create(c->gplan, d.np, c->stream); if (rcg) { g_create_error = "persistent GMW resources: allocation failed"; srukf_destroy(c); return rcg; } }
    split_ensure(c, c->gplan);
    {
        // k_syrk: tile (row r, col c >= r); A panel = S columns of r, B panel = S columns of c.  XCD owns rows.
        std::vector<int> ts = build_tile_table(d.np / 32, d.np / 32, true, true, 0);
        // k_pxy: tile (m = measurement tile, n = state tile); XCD owns the S panel (n), DZ panels stream.
        std::vector<int> tp = build_tile_table(d.np / 32, d.mp / 32, false, false, 1);
        c->n_syrk_tiles = (int)ts.size() / 2; c->n_pxy_tiles = (int)tp.size() / 2;
        ALLOC(c->syrk_tiles, ts.size()); ALLOC(c->pxy_tiles, tp.size());
        // the same order, restricted to the tile rows the persistent launch does not compute itself
        std::vector<int> th;
        // ... plus the diagonal 64 x 64 tile right behind them (srukf_gmw_head_extra_diag): the pivot workgroup needs it, with one panel
        // update applied, at the end of its second panel — its owner would still be forming it then
        const int hd = srukf_gmw_head_rows(), hx = hd + 64 * srukf_gmw_head_extra_diag();
        // (the tiles of the leading 128 x 128 block first: the pivot workgroup waits for nothing else before its first panel — head fold)
        for (int pass = 0; pass < 2; pass++)
            for (size_t q = 0; q + 1 < ts.size(); q += 2) {
                if (!(ts[q] >= 0 && (ts[q] * 32 < hd || (ts[q] * 32 < hx && ts[q + 1] * 32 < hx)))) continue;
                const bool crit = ts[q] * 32 < 128 && ts[q + 1] * 32 < 128;
                if (crit == (pass == 0)) { th.push_back(ts[q]); th.push_back(ts[q + 1]); }
            }
        c->n_syrk_head_crit = 0;
        for (size_t q = 0; q + 1 < th.size(); q += 2) if (th[q] * 32 < 128 && th[q + 1] * 32 < 128) c->n_syrk_head_crit++;
        c->n_syrk_head_tiles = (int)th.size() / 2;
        ALLOC(c->syrk_head_tiles, th.size() ? th.size() : 2);
        {
            // the batched launch (k_syrk_b: B filters x these tiles, no pivot waiting for the first ones) is bound by its operand traffic — 35 MB per filter for 10 MB of
            // operands: in the table's order an XCD owns tile ROWS, so the four tile rows of the head fetch every column slab (S and U^T, K x 32 doubles) four times over
            // four L2s.  Here the tiles of one pair of tile columns go to one XCD (list position % 8): a slab crosses the fabric once.
            std::vector<int> qx[8], tb;
            for (size_t q = 0; q + 1 < th.size(); q += 2) { const int x = (th[q + 1] / 2) % 8; qx[x].push_back(th[q]); qx[x].push_back(th[q + 1]); }
            size_t longest = 0;
            for (int x = 0; x < 8; x++) longest = std::max(longest, qx[x].size() / 2);
            for (size_t rnd = 0; rnd < longest; rnd++)
                for (int x = 0; x < 8; x++) { tb.push_back(2 * rnd + 1 < qx[x].size() ? qx[x][2 * rnd] : -1); tb.push_back(2 * rnd + 1 < qx[x].size() ? qx[x][2 * rnd + 1] : -1); }
            c->n_syrk_head_tiles_b = (int)tb.size() / 2;
            ALLOC(c->syrk_head_tiles_b, tb.size() ? tb.size() : 2);
            if (!tb.empty() && hipMemcpyAsync(c->syrk_head_tiles_b, tb.data(), sizeof(int) * tb.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                g_create_error = "tile table upload failed"; srukf_destroy(c); return SRUKF_ERR_HIP;
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) { g_create_error = "tile table upload failed"; srukf_destroy(c); return SRUKF_ERR_HIP; }      // (tb is a local)
        }
        if (!th.empty() && hipMemcpyAsync(c->syrk_head_tiles, th.data(), sizeof(int) * th.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            g_create_error = "tile table upload failed"; srukf_destroy(c); return SRUKF_ERR_HIP;
        }
        // same stream as the zero-fill of ALLOC (a copy on the null stream could be overtaken by it)
        if (hipMemcpyAsync(c->syrk_tiles, ts.data(), sizeof(int) * ts.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->pxy_tiles, tp.data(), sizeof(int) * tp.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            g_create_error = "tile table upload failed"; srukf_destroy(c); return SRUKF_ERR_HIP;
        }
    }
    c->hstage_bytes = sizeof(double) * (np * np + 4096);
    if (g_spare_stage && g_spare_stage_bytes >= c->hstage_bytes) {         // pinned staging of a context that was just rebuilt (map change)
        c->hstage = g_spare_stage; c->hstage_bytes = g_spare_stage_bytes; g_spare_stage = nullptr; g_spare_stage_bytes = 0;
    }
    if ((!c->hstage && hipHostMalloc((void**)&c->hstage, c->hstage_bytes) != hipSuccess) || hipHostMalloc((void**)&c->hfs, sizeof(FrameScalars) + sizeof(double) * (40 + c->d.mp + 5 * (size_t)c->d.N + 8)) != hipSuccess) {
        g_create_error = "hipHostMalloc failed"; srukf_destroy(c); return SRUKF_ERR_NOMEM;
    }
    memset(c->hfs, 0, sizeof(FrameScalars) + sizeof(double) * 40);      // (the flag word behind the robot view starts below every sequence number)
    c->hmeas = (double*)((char*)c->hfs + sizeof(FrameScalars) + sizeof(double) * 40);      // h | Si | visible as the statistics jobs of the fast path's k_pxy2 leave them (a buffer of its own:
                                                                                          // hstage is every accessor's staging area, and an accessor may run between predict_motion and predict_measurement)
    int rc = srukf_reset(c);
    if (rc) { g_create_error = c->err; srukf_destroy(c); return rc; }
    srukf_warm_exact_path(c->stream, c->fs);                     // (the exact path's first large-LDS launch on a stream costs ~75 ms: here, not inside somebody's flagged frame)
    *out = c;
    return SRUKF_OK;
}

int srukf_destroy(srukf_ctx* c)
{
    if (!c) return SRUKF_OK;
    hipSetDevice(c->device);
    for (srukf_ctx* r : c->retired) srukf_destroy(r);            // (they launch on this handle's stream: before it goes)
    c->retired.clear();
    if (c->spare_stage) { hipHostFree(c->spare_stage); c->spare_stage = nullptr; c->spare_stage_bytes = 0; }
    batch_plan_forget(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    // the side streams too, BEFORE any buffer goes back to the pool: the step-wise fast path returns as soon as the tail raises its pinned flag and has by then queued
    // the S -> ckS2 copy on ck_stream behind that tail; nothing on c->stream follows it, so the synchronisation above can return while the copy still reads S
    if (c->ck_stream) hipStreamSynchronize(c->ck_stream);
    if (c->side) hipStreamSynchronize(c->side);
    prof_collect(c);
    if (c->graph_exec) hipGraphExecDestroy(c->graph_exec);
    if (c->graph) hipGraphDestroy(c->graph);
    if (c->graph8_exec) hipGraphExecDestroy(c->graph8_exec);
    if (c->graph8) hipGraphDestroy(c->graph8);
    if (c->graphN_exec) hipGraphExecDestroy(c->graphN_exec);
    if (c->graphN) hipGraphDestroy(c->graphN);
    void* bufs[] = { c->X, c->S, c->G, c->Gbak, c->Wf, c->sigR, c->Cmat, c->Z, c->DZ, c->Ut, c->h /* + Si, vis */, c->PxyR, c->D,
                     c->zcur /* + mcur */, c->odocur, c->small, c->theta, c->fs, c->odo_seq, c->z_seq, c->m_seq, c->pan[0], c->pan[1], c->mpart, c->dxp, c->syrk_tiles, c->pxy_tiles, c->syrk_head_tiles,
                     c->perm, c->iperm, c->Sdis, c->ckS, c->ckX, c->ckS2, c->ckX2, c->odo_step, c->export_cnt, c->red_perm, c->red_iperm, c->gdiag, c->red_syrk_tiles, c->syrk_head_tiles_b, c->red_head0_tiles, c->split_fold_list, c->shadowA, c->Utp, c->P1, c->pxy2_tiles, c->nskip, c->slabW, c->slabL, c->gsW, c->gsL, c->S32, c->X32, c->U32, c->mx_part, c->mx_tasks, c->mx_tiles, c->fold_sync, c->dxk, c->A32, c->mxr_part, c->mxr_tasks, c->mxr_tiles, c->mxr_f64_tiles, c->mxr_xt, c->app_patch, c->app_tmpl, c->d_image, c->appR, c->appT, c->appPx, c->corr, c->has_app };
    for (void* b : bufs) if (b) srukf_dfree_on(b, c->stream);
    gmw_plan_destroy(c->gplan, c->stream);
    gmw_plan_destroy(c->gplan_red, c->stream);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
    if (c->ck_stream) { hipStreamSynchronize(c->ck_stream); hipStreamDestroy(c->ck_stream); hipEventDestroy(c->ck_e1); hipEventDestroy(c->ck_e2); if (c->ck_e3) hipEventDestroy(c->ck_e3); }
    if (c->own_stream && c->stream) hipStreamSynchronize(c->stream);
    if (c->hstage) {
        // keep ONE pinned staging buffer for the next context (pinning 16 MB costs milliseconds; map changes rebuild contexts)
        if (!g_spare_stage || g_spare_stage_bytes < c->hstage_bytes) { if (g_spare_stage) hipHostFree(g_spare_stage); g_spare_stage = c->hstage; g_spare_stage_bytes = c->hstage_bytes; }
        else hipHostFree(c->hstage);
    }
    if (c->hfs) hipHostFree(c->hfs);
    if (c->hview) hipHostFree(c->hview);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return SRUKF_OK;
}

}  // extern "C"

namespace srukf_impl {

// A context the handle outgrew at a map change, kept for the next time the map has that size.  Everything of it that depends on the STATE is rebuilt by whoever revives
// it (srukf_reset here; the caller's factorisation, srukf_set_storage and update_null_set afterwards); what depends only on N, the device and the parameters — buffers,
// plans, tile tables, pinned areas, side streams — is what it is kept for.
void ctx_retire(srukf_ctx* handle, srukf_ctx* old)
{
    drop_graphs(old);
    old->own_stream = false;
    // its pinned staging area goes to the handle: the next context this handle creates or revives takes it (srukf_create's hipHostMalloc of np^2 doubles was 2.5 of the
    // 3.5 ms a map change to a size not seen before cost)
    if (old->hstage) {
        if (!handle->spare_stage || handle->spare_stage_bytes < old->hstage_bytes) {
            if (handle->spare_stage) hipHostFree(handle->spare_stage);
            handle->spare_stage = old->hstage; handle->spare_stage_bytes = old->hstage_bytes;
        } else hipHostFree(old->hstage);
        old->hstage = nullptr; old->hstage_bytes = 0;
    }
    handle->retired.push_back(old);
    // How many: a map that breathes by +- 1 around a size revisits ~16 sizes (bench.py's churn leg: 200 -> 185 landmarks), and destroying the context that falls out of
    // the list is the expensive part of a miss (2.6 of 3.6 ms at N = 200: ~60 device frees).  Up to 24 contexts ("ctx_keep") or ~16 GB of them (a context is ~14 matrices of np^2 doubles).  Not 32: a process that
    // exits with exactly 32 retired contexts behind a handle — 950 to 1 100 frames of the churn leg — crashed in the HIP runtime's own exit handler after every srukf call
    // had returned (SIGSEGV under __cxa_finalize in libamdhip64; 30, 31, 33, 36 and 40 contexts: no crash; trimming the memory pool first: no difference).
    const double ctx_bytes = 14.0 * 8.0 * (double)old->d.np * old->d.np;
    const size_t keep = (size_t)std::min((double)g_dbg_ctx_keep.load(), std::max(4.0, 16e9 / ctx_bytes));
    while (handle->retired.size() > keep) { srukf_destroy(handle->retired.front()); handle->retired.erase(handle->retired.begin()); }
}

static int ctx_revive(srukf_ctx* r)
{
    if (r->ck_stream) HIPCHK(r, hipStreamSynchronize(r->ck_stream));
    if (r->side) HIPCHK(r, hipStreamSynchronize(r->side));
    r->ck_pending = false; r->ck3_inflight = false; r->ck_valid = false;
    if (r->odo_seq) { srukf_dfree_on(r->odo_seq, r->stream); srukf_dfree_on(r->z_seq, r->stream); srukf_dfree_on(r->m_seq, r->stream); r->odo_seq = nullptr; r->z_seq = nullptr; r->m_seq = nullptr; }
    r->seqF = 0;
    r->storage = SRUKF_STORAGE_F64;                              // (as a fresh context: the caller sets the handle's mode; S32 / X32 / A32 stay allocated)
    r->K_new = 0; r->dx_pending = false; r->dx_lm = false; r->xr1_pending = false;
    if (r->fold_sync) HIPCHK(r, hipMemsetAsync(r->fold_sync, 0, sizeof(unsigned int) * (size_t)srukf_fold_words(r->d.mp / 64, r->d.np / 64), r->stream));
    r->null_canonical = false; r->tail_ok = false;
    r->clamp_frame_host = r->clamp_row_host = -1;
    r->next_odo_valid = false; r->fs_seq_step = false; r->last_update_sequential = false;
    r->step_export_attached = false; r->step_export = StepExport{}; r->mirror_next = false; r->meas_seq = 0;
    r->view_auto = false; r->view_unused = 0; r->view_hits = 0;
    r->step_fast_frames = r->step_slow_frames = 0; r->exact_frames = 0;
    r->profiling = false; r->pev.clear();
    r->err.clear();
    HIPCHK(r, hipMemsetAsync(r->fs, 0, sizeof(FrameScalars), r->stream));
    return srukf_reset(r);                                       // step-path flags, X / S as srukf_create leaves them, frame scalars, null set off, graphs dropped
}

int ctx_obtain(srukf_ctx* handle, srukf_ctx** out, int N)
{
    for (size_t q = 0; q < handle->retired.size(); q++) {
        srukf_ctx* r = handle->retired[q];
        if (r->d.N != N || r->device != handle->device || r->stream != handle->stream || memcmp(&r->p, &handle->p, sizeof(srukf_params)) != 0) continue;
        handle->retired.erase(handle->retired.begin() + (long)q);
        if (!r->hstage) {                                       // (retired contexts keep no staging area: ctx_retire)
            const size_t need = sizeof(double) * ((size_t)r->d.np * r->d.np + 4096);
            if (handle->spare_stage && handle->spare_stage_bytes >= need) { r->hstage = handle->spare_stage; r->hstage_bytes = handle->spare_stage_bytes; handle->spare_stage = nullptr; handle->spare_stage_bytes = 0; }
            else if (hipHostMalloc((void**)&r->hstage, need) == hipSuccess) r->hstage_bytes = need;
            else { r->hstage = nullptr; srukf_destroy(r); break; }
        }
        if (ctx_revive(r) == SRUKF_OK) { *out = r; return SRUKF_OK; }
        srukf_destroy(r);
        break;
    }
    // (nothing to revive: a new context.  The retired one keeps its side stream for its next life; the new one finds its own — split_ensure)
    if (handle->spare_stage && !g_spare_stage) { g_spare_stage = handle->spare_stage; g_spare_stage_bytes = handle->spare_stage_bytes; handle->spare_stage = nullptr; handle->spare_stage_bytes = 0; }
    return srukf_create(out, N, &handle->p, handle->device, handle->stream);
}

}  // namespace srukf_impl

extern "C" {

int srukf_reset(srukf_ctx* c)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const KDims& d = c->d;
    const size_t np = d.np;
    step_state_replaced(c);
    HIPCHK(c, hipMemsetAsync(c->X, 0, sizeof(double) * np, c->stream));
    HIPCHK(c, hipMemsetAsync(c->S, 0, sizeof(double) * np * np, c->stream));
    // initializeParameters, SLAM.cpp:226-231
    double* hs = c->hstage;
    hs[0] = c->p.sigma_x; hs[1] = c->p.sigma_y; hs[2] = c->p.sigma_z; hs[3] = c->p.sigma_theta;
    for (int e = 0; e < 4; e++)
        HIPCHK(c, hipMemcpyAsync(c->S + (size_t)(d.n - 4 + e) * np + (d.n - 4 + e), hs + e, sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_set_frame(c->stream, c->fs, 0, 1);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->phase = 0; c->async_pending = false;
    if (c->red_r) { c->red_r = 0; drop_graphs(c); }            // the state is the robot block only: nothing to reduce until a state arrives
    return SRUKF_OK;
}

int srukf_dims(const srukf_ctx* c, int* N, int* n, int* Na, int* L)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    if (N) *N = c->d.N; if (n) *n = c->d.n; if (Na) *Na = c->d.Na; if (L) *L = c->d.L;
    return SRUKF_OK;
}

int srukf_set_state(srukf_ctx* c, const double* X, const double* S)
{
    if (!c || !X || !S) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n; const size_t np = c->d.np;
    step_state_replaced(c);
    double* hs = c->hstage;
    memset(hs, 0, sizeof(double) * np * np);
    for (int r = 0; r < n; r++) for (int cc = r; cc < n; cc++) hs[(size_t)r * np + cc] = S[(size_t)r * n + cc];   // upper triangle only
    HIPCHK(c, hipMemcpyAsync(c->S, hs, sizeof(double) * np * np, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memset(hs, 0, sizeof(double) * np);
    memcpy(hs, X, sizeof(double) * n);
    HIPCHK(c, hipMemcpyAsync(c->X, hs, sizeof(double) * np, hipMemcpyHostToDevice, c->stream));
    quantize_state(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->phase = 0;
    return update_null_set(c);
}

int srukf_get_state(srukf_ctx* c, double* X, double* S)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n; const size_t np = c->d.np;
    step_commit_motion(c);
    double* hs = c->hstage;
    if (X) {
        HIPCHK(c, hipMemcpyAsync(hs, c->X, sizeof(double) * np, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(X, hs, sizeof(double) * n);
    }
    if (S) {
        HIPCHK(c, hipMemcpyAsync(hs, c->S, sizeof(double) * np * np, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int r = 0; r < n; r++) memcpy(S + (size_t)r * n, hs + (size_t)r * np, sizeof(double) * n);
    }
    return SRUKF_OK;
}

int srukf_set_state_device(srukf_ctx* c, const double* dX, const double* dS, int S_ld)
{
    if (!c || !dX || !dS || S_ld < c->d.n) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n; const size_t np = c->d.np;
    step_state_replaced(c);
    HIPCHK(c, hipMemsetAsync(c->S, 0, sizeof(double) * np * np, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->S, sizeof(double) * np, dS, sizeof(double) * S_ld, sizeof(double) * n, n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->X, 0, sizeof(double) * np, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->X, dX, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    quantize_state(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->phase = 0;
    return update_null_set(c);
}

int srukf_get_state_device(srukf_ctx* c, double* dX, double* dS, int S_ld)
{
    if (!c || S_ld < c->d.n) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n; const size_t np = c->d.np;
    step_commit_motion(c);
    if (dX) HIPCHK(c, hipMemcpyAsync(dX, c->X, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    if (dS) HIPCHK(c, hipMemcpy2DAsync(dS, sizeof(double) * S_ld, c->S, sizeof(double) * np, sizeof(double) * n, n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SRUKF_OK;
}

int srukf_get_robot(srukf_ctx* c, double pose4[4], double P4[16])
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n;
    if (c->robot_cached) {                                     // the fast path of srukf_update fetched it with the frame's status
        const double* hr = (const double*)((const char*)c->hfs + sizeof(FrameScalars));
        if (P4) memcpy(P4, hr, sizeof(double) * 16);
        if (pose4) memcpy(pose4, hr + 16, sizeof(double) * 4);
        return SRUKF_OK;
    }
    step_commit_motion(c);
    if (pose4) {
        HIPCHK(c, hipMemcpyAsync(c->hstage + 64, c->X + (n - 4), sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(pose4, c->hstage + 64, sizeof(double) * 4);
    }
    if (P4) return block_cov(c, n - 4, 4, P4);
    return SRUKF_OK;
}

int srukf_get_landmark_block(srukf_ctx* c, int k, double X6[6], double P66[36])
{
    if (!c || k < 0 || k >= c->d.N) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c);
    if (X6) {
        HIPCHK(c, hipMemcpyAsync(c->hstage + 64, c->X + 6 * k, sizeof(double) * 6, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(X6, c->hstage + 64, sizeof(double) * 6);
    }
    if (P66) return block_cov(c, 6 * k, 6, P66);
    return SRUKF_OK;
}

int srukf_get_landmarks_cartesian(srukf_ctx* c, double* xyz, double* cov)
{
    if (!c || (!xyz && !cov)) return SRUKF_ERR_BAD_ARG;
    const int N = c->d.N;
    if (N == 0) return SRUKF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c);
    // G is scratch outside the refactorisation (the tail has consumed the factor rows it held): 12 N doubles for the results.  (Not Z: between predict and update the
    // fast path of the step-wise API still needs the centre point's row there.)
    double* dx = c->G; double* dc = c->G + 3 * (size_t)N;
    srukf_launch_landmarks_cartesian(c->stream, c->d, c->X, c->S, dx, dc);
    HIPCHK(c, hipMemcpyAsync(c->hstage, dx, sizeof(double) * 12 * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (xyz) memcpy(xyz, c->hstage, sizeof(double) * 3 * (size_t)N);
    if (cov) memcpy(cov, c->hstage + 3 * (size_t)N, sizeof(double) * 9 * (size_t)N);
    return SRUKF_OK;
}

// What the reference's SLAM() refreshes after every update for the display (updateFeaturesInformation, SLAM.cpp:2397-2621: m_X_k, per-landmark xyz and 3 x 3 Cartesian
// covariance; recordRobotInformation 3539-3556: the robot block of m_P_k) in ONE device round trip: X[n], xyz[3N], cov[9N], pose4[4], P4[16] (any may be NULL).
int srukf_get_frame_view(srukf_ctx* c, double* X, double* xyz, double* cov, double pose4[4], double P4[16])
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->d.N, n = c->d.n;
    if (c->view_cached) {                                      // exported with the last update's status (step_update_fast)
        c->view_unused = 0; c->view_hits++;
        const double* hv = c->hview; const double* hr = (const double*)((const char*)c->hfs + sizeof(FrameScalars));
        if (xyz) memcpy(xyz, hv, sizeof(double) * 3 * (size_t)N);
        if (cov) memcpy(cov, hv + 3 * (size_t)N, sizeof(double) * 9 * (size_t)N);
        if (X) memcpy(X, hv + 12 * (size_t)N, sizeof(double) * n);
        if (P4) memcpy(P4, hr, sizeof(double) * 16);
        if (pose4) memcpy(pose4, hr + 16, sizeof(double) * 4);
        return SRUKF_OK;
    }
    if (c->robot_cached && c->dbg.view_auto && N > 0) {        // asked right after an update of the fast path: the next updates bring the view along
        if (c->hview_doubles < 12 * (size_t)N + n) {
            if (c->hview) hipHostFree(c->hview);
            c->hview = nullptr; c->hview_doubles = 0;
            if (hipHostMalloc((void**)&c->hview, sizeof(double) * (12 * (size_t)N + c->d.np)) == hipSuccess) c->hview_doubles = 12 * (size_t)N + c->d.np;
            else (void)hipGetLastError();
        }
        c->view_auto = c->hview != nullptr; c->view_unused = 0;
    }
    step_commit_motion(c);
    double* dx = c->G; double* dc = c->G + 3 * (size_t)N; double* dr = c->G + 12 * (size_t)N;        // G is scratch outside the refactorisation
    if (N > 0 && (xyz || cov)) srukf_launch_landmarks_cartesian(c->stream, c->d, c->X, c->S, dx, dc);
    const bool robot = (pose4 || P4) && !c->robot_cached;
    if (robot) srukf_launch_block_cov(c->stream, c->d, c->S, n - 4, 4, dr, c->X);
    double* hs = c->hstage;
    if (N > 0 && (xyz || cov)) HIPCHK(c, hipMemcpyAsync(hs, dx, sizeof(double) * (12 * (size_t)N + (robot ? 20 : 0)), hipMemcpyDeviceToHost, c->stream));
    else if (robot) HIPCHK(c, hipMemcpyAsync(hs + 12 * (size_t)N, dr, sizeof(double) * 20, hipMemcpyDeviceToHost, c->stream));
    if (X) HIPCHK(c, hipMemcpyAsync(hs + 12 * (size_t)N + 32, c->X, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (xyz) memcpy(xyz, hs, sizeof(double) * 3 * (size_t)N);
    if (cov) memcpy(cov, hs + 3 * (size_t)N, sizeof(double) * 9 * (size_t)N);
    const double* hr = c->robot_cached ? (const double*)((const char*)c->hfs + sizeof(FrameScalars)) : hs + 12 * (size_t)N;
    if (P4) memcpy(P4, hr, sizeof(double) * 16);
    if (pose4) memcpy(pose4, hr + 16, sizeof(double) * 4);
    if (X) memcpy(X, hs + 12 * (size_t)N + 32, sizeof(double) * n);
    return SRUKF_OK;
}

int srukf_get_covariance(srukf_ctx* c, double* P)
{
    // m_P_k = S^T S (SLAM.cpp:2404): k_syrk with an empty downdate range
    if (!c || !P) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n; const size_t np = c->d.np;
    step_commit_motion(c);
    srukf_launch_syrk(c->stream, c->d, c->S, c->Ut, 0, 0, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, nullptr, c->X, RankArgs{}, nullptr);
    HIPCHK(c, hipMemcpyAsync(c->hstage, c->G, sizeof(double) * np * np, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < n; r++) for (int cc = r; cc < n; cc++) { const double v = c->hstage[(size_t)r * np + cc]; P[(size_t)r * n + cc] = v; P[(size_t)cc * n + r] = v; }
    return SRUKF_OK;
}

int srukf_set_storage(srukf_ctx* c, int storage)
{
    if (!c || (storage != SRUKF_STORAGE_F64 && storage != SRUKF_STORAGE_F32 && storage != SRUKF_STORAGE_F32_MIXED)) return SRUKF_ERR_BAD_ARG;
    // (Rounds 2 - 5 refused SRUKF_STORAGE_F32_MIXED below epsilon = 1e-9: the mode diverged within ten frames at the reference's 1e-13.  Round 6 found the cause —
    //  fp32 accumulation over K = 1024 products per chunk, whose error grew linearly in the kept pivots from frame to frame (scripts/mixed_drift_probe.py) — and
    //  two remedies: the fp32 accumulators are flushed into FP64 ones every 32 rows (k_syrk32, MX_FLUSH), and in the rank-aware form the tiles of the robot block and
    //  of the map's shared anchor are formed in FP64 (their pivots are 2e-6 .. 9e-6 of the marginal variance: scripts/pivot_ratio_probe.py).  With both the mode
    //  tracks the fp64 filter to 1.2e-6 m over the reference's whole capacity at N = 500, epsilon = 1e-13 (scripts/mixed_rank_study.py, DESIGN.md row g).)
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c); step_state_replaced(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t np = c->d.np, mp = c->d.mp;
    if (storage != SRUKF_STORAGE_F64 && !c->S32) {
        HIPCHK(c, srukf_dmalloc((void**)&c->S32, sizeof(float) * np * np));
        HIPCHK(c, srukf_dmalloc((void**)&c->X32, sizeof(float) * np));
        HIPCHK(c, hipMemsetAsync(c->S32, 0, sizeof(float) * np * np, c->stream));
        HIPCHK(c, hipMemsetAsync(c->X32, 0, sizeof(float) * np, c->stream));
    }
    if (storage == SRUKF_STORAGE_F32_MIXED && !c->U32) {
        int ntiles = 0;
        const int ntasks = srukf_mixed_build_tasks((int)np, (int)mp, nullptr, nullptr, &ntiles);
        std::vector<short> tk((size_t)4 * ntasks); std::vector<int> tl((size_t)2 * ntiles);
        srukf_mixed_build_tasks((int)np, (int)mp, tk.data(), tl.data(), &ntiles);
        HIPCHK(c, srukf_dmalloc((void**)&c->U32, sizeof(float) * mp * np));
        HIPCHK(c, srukf_dmalloc((void**)&c->mx_part, srukf_mixed_part_bytes(ntasks)));
        HIPCHK(c, srukf_dmalloc(&c->mx_tasks, sizeof(short) * tk.size()));
        HIPCHK(c, srukf_dmalloc(&c->mx_tiles, sizeof(int) * tl.size()));
        HIPCHK(c, hipMemcpy(c->mx_tasks, tk.data(), sizeof(short) * tk.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->mx_tiles, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice));
        c->mx_ntasks = ntasks; c->mx_ntiles = ntiles;
    }
    if (storage != c->storage) drop_graphs(c);             // the captured frames do or do not contain the rounding pass / the fp32 contraction
    c->storage = storage;
    quantize_state(c);
    shadow_rebuild(c);                                     // the permuted copy of the rank-aware form holds what S holds
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return mixed_red_ensure(c);
}

int srukf_get_state_f32(srukf_ctx* c, float* X, float* S)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    if (c->storage == SRUKF_STORAGE_F64) { c->err = "get_state_f32: the context stores fp64 (srukf_set_storage)"; return SRUKF_ERR_SEQUENCE; }
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c);
    if (c->f32_stale) { quantize_state(c); HIPCHK(c, hipStreamSynchronize(c->stream)); c->f32_stale = false; }      // (the step-wise fast path rounds S and X as it writes them; the float copies on demand)
    const int n = c->d.n; const size_t np = c->d.np;
    if (X) HIPCHK(c, hipMemcpy(X, c->X32, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (S) HIPCHK(c, hipMemcpy2D(S, sizeof(float) * n, c->S32, sizeof(float) * np, sizeof(float) * n, n, hipMemcpyDeviceToHost));
    return SRUKF_OK;
}

int srukf_set_new_landmarks(srukf_ctx* c, int K_new)
{
    if (!c || K_new < 0) return SRUKF_ERR_BAD_ARG;
    const int n = c->d.n;
    if (6 * K_new > n - 4) { c->err = "K_new larger than the map"; return SRUKF_ERR_DIM_MISMATCH; }
    HIPCHK(c, hipSetDevice(c->device));
    c->K_new = K_new;
    if (K_new == 0) return SRUKF_OK;
    // getPermutationMatrix, SLAM.cpp:1303-1334: X_normal[r] = X_disordered[perm[r]]
    std::vector<int> perm(n), iperm(n);
    const int dimOld = n - 6 * K_new;
    for (int i = 0; i < dimOld - 4; i++) perm[i] = i;
    for (int e = 0; e < 4; e++) perm[n - 4 + e] = dimOld - 4 + e;
    for (int id = 0; id < K_new; id++) {
        for (int e = 0; e < 3; e++) perm[dimOld - 4 + 6 * id + e] = dimOld + 3 * K_new + 3 * id + e;
        for (int e = 0; e < 3; e++) perm[dimOld - 4 + 6 * id + 3 + e] = dimOld + 3 * id + e;
    }
    for (int r2 = 0; r2 < n; r2++) iperm[perm[r2]] = r2;
    if (!c->perm) { HIPCHK(c, srukf_dmalloc((void**)&c->perm, sizeof(int) * c->d.np)); HIPCHK(c, srukf_dmalloc((void**)&c->iperm, sizeof(int) * c->d.np)); }
    HIPCHK(c, hipMemcpy(c->perm, perm.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->iperm, iperm.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    return SRUKF_OK;
}

int srukf_set_profiling(srukf_ctx* c, int on)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    c->profiling = on != 0;
    return SRUKF_OK;
}

int srukf_profile_count(srukf_ctx* c) { return c ? KC_COUNT : SRUKF_ERR_BAD_ARG; }

int srukf_profile_get(srukf_ctx* c, int i, const char** name, double* total_ms, long long* launches, double* alg_flops, double* alg_bytes)
{
    if (!c || i < 0 || i >= KC_COUNT) return SRUKF_ERR_BAD_ARG;
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (name) *name = kclass_name[i];
    if (total_ms) *total_ms = c->prof_ms[i];
    if (launches) *launches = c->prof_n[i];
    if (alg_flops) *alg_flops = c->prof_flops[i];
    if (alg_bytes) *alg_bytes = c->prof_bytes[i];
    return SRUKF_OK;
}

int srukf_profile_reset(srukf_ctx* c)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    memset(c->prof_ms, 0, sizeof c->prof_ms); memset(c->prof_n, 0, sizeof c->prof_n);
    memset(c->prof_flops, 0, sizeof c->prof_flops); memset(c->prof_bytes, 0, sizeof c->prof_bytes);
    return SRUKF_OK;
}

}  // extern "C"
