// srukf_batch.hip — srukf_run_frames_batch: B filters of one shape through the same block of staged frames, one launch per stage for a group of filters.

#include "srukf_ctx.h"
using namespace srukf_impl;

// ---- batched replay: B filters of ONE shape, ONE launch per stage, ONE stream, one graph (srukf_run_frames_batch) --------------------------------------------
// The filters of a Monte-Carlo run (MonoSLAMView.cpp:526-572 once per sequence: same map, own measurements) have the same dimensions, the same null set and
// the same launch grids; what differs are the buffers.  Every stage of the staged frame then runs as one launch over all of them — k_pxy2_b, k_gain_b, k_syrk_b
// (head tiles, X += dX, dropped diagonal), k_syrk_own_b (the other tiles of S^T S - U U^T in the summation order of the owners' fold), one k_gmw_step64_b per
// 64-row panel (B critical-path workgroups side by side, the trailing tiles of all filters around them: no workgroup waits for another inside a launch, so no
// CU is held idle — the persistent launch's workers are, three quarters of the time —, no gate, no residency assumption), k_rank_expand_b — with the per-filter
// pointers in small device tables.  Same device functions on the same values as the filter that runs alone (the per-panel and the persistent factorisation are
// bit-identical given the same tiles; k_syrk_own is the owners' arithmetic): trajectories and states are bit-identical to solo runs.
struct BatchPlan {
    std::vector<srukf_ctx*> cs;
    int B = 0;
    void *t_pxy2 = nullptr, *t_gain = nullptr, *t_syrk = nullptr, *t_own = nullptr, *t_step = nullptr, *t_exp = nullptr;
    hipGraph_t g1 = nullptr, g8 = nullptr; hipGraphExec_t e1 = nullptr, e8 = nullptr;
    std::vector<unsigned long long> sig;                    // what the captured launches depend on besides the tables' CONTENTS
};
// (one plan per group of filters: srukf_run_frames_batch cuts B filters into groups that run side by side, each on a stream of its own)
// Plans and group streams are kept per host thread AND per device: a stream belongs to the device that was current when it was created, and a thread may run
// batches for filters on several devices (round-4 advisor finding: streams created once on whichever device came first).
// The groups' streams: created together, once per device, so that they sit on different hardware queues whatever the filters' own streams map to (streams that share a
// queue serialise: with the groups on their first filters' streams, 4 + 4 filters ran slower than 4 alone).  They go when the device's last plan goes.
struct BatchDev { BatchPlan* plans[SRUKF_BATCH_GROUPS_MAX] = {}; hipStream_t streams[SRUKF_BATCH_GROUPS_MAX] = {}; };
static thread_local std::map<int, BatchDev> g_batch_dev;
static hipStream_t batch_stream(int device, int grp)
{
    BatchDev& bd = g_batch_dev[device];
    if (!bd.streams[0]) {
        if (hipSetDevice(device) != hipSuccess) return nullptr;
        for (int q = 0; q < SRUKF_BATCH_GROUPS_MAX; q++) if (hipStreamCreateWithFlags(&bd.streams[q], hipStreamNonBlocking) != hipSuccess) bd.streams[q] = nullptr;
    }
    return bd.streams[grp];
}
static void batch_plan_drop_graphs(BatchPlan* bp)
{
    if (bp->e1) { hipGraphExecDestroy(bp->e1); bp->e1 = nullptr; }
    if (bp->g1) { hipGraphDestroy(bp->g1); bp->g1 = nullptr; }
    if (bp->e8) { hipGraphExecDestroy(bp->e8); bp->e8 = nullptr; }
    if (bp->g8) { hipGraphDestroy(bp->g8); bp->g8 = nullptr; }
}
static void batch_plan_destroy(int device, int grp, bool keep_streams = false)
{
    auto it = g_batch_dev.find(device);
    if (it == g_batch_dev.end()) return;
    BatchDev& bd = it->second;
    BatchPlan* bp = bd.plans[grp];
    if (bp) {
        hipSetDevice(device);
        batch_plan_drop_graphs(bp);
        for (void* t : { bp->t_pxy2, bp->t_gain, bp->t_syrk, bp->t_own, bp->t_step, bp->t_exp }) if (t) srukf_dfree(t);
        delete bp;
        bd.plans[grp] = nullptr;
    }
    if (keep_streams) return;
    for (int q = 0; q < SRUKF_BATCH_GROUPS_MAX; q++) if (bd.plans[q]) return;
    for (int q = 0; q < SRUKF_BATCH_GROUPS_MAX; q++) if (bd.streams[q]) { hipStreamSynchronize(bd.streams[q]); hipStreamDestroy(bd.streams[q]); }
    g_batch_dev.erase(it);                                      // the device's last plan: its streams go too
}
static void batch_plan_forget_impl(const srukf_ctx* c)
{
    auto it = g_batch_dev.find(c->device);
    if (it == g_batch_dev.end()) return;
    for (int grp = 0; grp < SRUKF_BATCH_GROUPS_MAX; grp++) {
        BatchPlan* bp = it->second.plans[grp];
        if (!bp) continue;
        bool mine = false;
        for (const srukf_ctx* q : bp->cs) mine = mine || q == c;
        if (!mine) continue;
        if (it->second.streams[grp]) hipStreamSynchronize(it->second.streams[grp]);
        batch_plan_destroy(c->device, grp);
        it = g_batch_dev.find(c->device);
        if (it == g_batch_dev.end()) return;
    }
}
namespace srukf_impl {
void batch_plan_forget(const srukf_ctx* c) { batch_plan_forget_impl(c); }
// srukf_debug_set "batch_split": the captured batch frames of this thread contain one or the other launch sequence
void batch_drop_all_graphs()
{
    for (auto& kv : g_batch_dev)
        for (int grp = 0; grp < SRUKF_BATCH_GROUPS_MAX; grp++) if (kv.second.plans[grp]) { hipSetDevice(kv.first); hipStreamSynchronize(kv.second.streams[grp]); batch_plan_drop_graphs(kv.second.plans[grp]); }
}
}  // namespace srukf_impl
// Can these filters run as one batch?  Same device and shape, the default launch sequence of a filter that has the GPU to itself ("fused tail" mode on the permuted
// operands, fp64 storage), canonical null rows, nothing pending.
static bool batch_eligible(srukf_ctx* const* cs, int B, bool ignore_canonical = false)
{
    if (B < 2 || B > 64 || !g_dbg_batch_wide.load()) return false;
    const srukf_ctx* a = cs[0];
    for (int b = 0; b < B; b++) {
        const srukf_ctx* c = cs[b];
        for (int q = 0; q < b; q++) if (cs[q] == c) return false;
        if (c->device != a->device || c->d.N != a->d.N || c->d.N < 1 || c->storage != SRUKF_STORAGE_F64 || c->w.wc0 != c->w.wm0) return false;
        if (c->red_r <= 0 || c->red_r != a->red_r || c->red_Tp != a->red_Tp || !c->shadowA || (!c->null_canonical && !ignore_canonical) || !c->nskip || !c->tail_ok) return false;
        if (c->ns_full != a->ns_full || c->ns_null != a->ns_null || c->ns_rows != a->ns_rows || c->n_pxy2_tiles != a->n_pxy2_tiles) return false;
        // (the batched launches take these from the group's first filter: shape-only quantities today — checked, not assumed)
        if (c->n_syrk_head_tiles != a->n_syrk_head_tiles || c->gplan_red.ntiles != a->gplan_red.ntiles || c->pxy2_split_b0 != a->pxy2_split_b0 || memcmp(&c->w, &a->w, sizeof c->w) != 0) return false;
        if (!c->dbg.pxy2 || !c->dbg.nullskip || !c->dbg.tail_fuse || c->dbg.fused_motion != 2 || !c->dbg.table_perm || c->profiling || c->use_graph != a->use_graph || c->debug_starve) return false;
        if (memcmp(&c->p, &a->p, sizeof c->p) != 0 || c->gplan_red.T < 16 || (size_t)c->d.np * sizeof(double) > 48 * 1024 || !rank_fused_mode()) return false;
        if (!c->odo_seq || c->seqF != a->seqF) return false;
    }
    return true;
}
static void batch_frame(const BatchPlan* bp, hipStream_t st)
{
    const srukf_ctx* c = bp->cs[0];
    const KDims& d = c->d;
    const int n = d.n, np = d.np, r = c->red_r, Tp = c->red_Tp, B = bp->B, kr = (r + 15) & ~15;
    srukf_launch_pxy2_b(st, d, bp->t_pxy2, B, c->pxy2_tiles, c->n_pxy2_tiles, kr, c->w, (d.N + 31) / 32);
    srukf_launch_gain_b(st, d, c->w, bp->t_gain, B, c->pxy2_split_b0, sqrt(c->p.epsilon));
    srukf_launch_syrk_b(st, d, bp->t_syrk, B, g_dbg_batch_xcd.load() ? c->syrk_head_tiles_b : c->syrk_head_tiles, g_dbg_batch_xcd.load() ? c->n_syrk_head_tiles_b : c->n_syrk_head_tiles, std::min(np, kr), (n + 255) / 256,
                        (n - r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS);
    srukf_launch_syrk_own_b(st, n, np, bp->t_own, B, 0, d.mp, kr, c->gplan_red.tiles, c->gplan_red.ntiles, Tp);
    int pb = 0;
    for (int j0 = -64; j0 + 64 < np && j0 + 64 <= 64 * Tp; j0 += 64, pb ^= 1) {
        if (!g_dbg_batch_split.load()) {
            srukf_launch_gmw_step64_b(st, n, np, j0, c->p.epsilon, bp->t_step, B, std::max(1, Tp - j0 / 64 - 1), pb);    // rows of the kept pivots only; the last panel: the pass-on row
            continue;
        }
        // split form: A = critical-path workgroups + the panel's slabs (and S rows) once per column block, B = the trailing tiles of the kept rows as plain K = 64
        // updates.  The last pivoted panel has no tiles to update (the pass-on row's values are never used: its S rows come from the slab workgroups).
        // K = 128 (round 6): panels in pairs (a, b = a + 64).  Step a's trailing update only touches what step b's pivots and slabs read (block row 0, tile (1, 1): "thin");
        // step b's pass applies panel a's slabs and then panel b's to everything else — one pass over G instead of two, the products per element in the same order
        // (bit-identical).  A pair needs step b to have trailing tiles; the slabs of a pair live in the two halves of slabW / slabL.
        const int rows = Tp - j0 / 64 - 1, pair = j0 >= 0 ? (j0 / 64) & 1 : 0;
        const bool k128 = g_dbg_batch_k128.load() != 0;
        const bool first_of_pair = k128 && j0 >= 0 && pair == 0 && rows - 1 >= 1 && j0 + 128 < np && j0 + 128 <= 64 * Tp;
        const bool second_of_pair = k128 && j0 >= 64 && pair == 1 && rows >= 1;
        srukf_launch_gmw_pivslab_b(st, n, np, j0, c->p.epsilon, bp->t_step, B, pb, second_of_pair ? 1 : 0);
        if (j0 >= 0 && rows >= 1) srukf_launch_gmw_trail_b(st, np, j0, bp->t_step, B, rows, second_of_pair ? 1 : 0, first_of_pair ? 1 : second_of_pair ? 2 : 0);
    }
    srukf_launch_rank_expand_b(st, n, np, r, c->p.epsilon, bp->t_exp, B, c->w.gamma, d, c->w, c->p);
}
static int batch_capture(BatchPlan* bp, hipStream_t st, int nframes, hipGraph_t* g, hipGraphExec_t* ge)
{
    srukf_ctx* c = bp->cs[0];
    HIPCHK(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int q = 0; q < nframes; q++) batch_frame(bp, st);
    const hipError_t launch_err = hipGetLastError();
    HIPCHK(c, hipStreamEndCapture(st, g));
    HIPCHK(c, launch_err);
    HIPCHK(c, hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
    return SRUKF_OK;
}
// frames [first, first + count) of all filters on cs[0]'s stream; dt[b]: device trajectory buffers (count rows).  Asynchronous; the caller synchronises that stream.
static int batch_run(srukf_ctx* const* cs, int B, int first, int count, double* const* dt, int grp)
{
    srukf_ctx* c0 = cs[0];
    HIPCHK(c0, hipSetDevice(c0->device));
    hipStream_t st = batch_stream(c0->device, grp);
    if (!st) { c0->err = "run_frames_batch: no stream for the group"; return SRUKF_ERR_HIP; }
    BatchPlan* bp = g_batch_dev[c0->device].plans[grp];
    bool same = bp && bp->B == B;
    for (int b = 0; same && b < B; b++) same = bp->cs[b] == cs[b];
    if (!same) {
        batch_plan_destroy(c0->device, grp, true);
        bp = g_batch_dev[c0->device].plans[grp] = new BatchPlan();
        bp->B = B; bp->cs.assign(cs, cs + B);
        if (srukf_dmalloc(&bp->t_pxy2, sizeof(Pxy2Args) * B) != hipSuccess || srukf_dmalloc(&bp->t_gain, sizeof(GainArgs) * B) != hipSuccess ||
            srukf_dmalloc(&bp->t_syrk, sizeof(SyrkArgs) * B) != hipSuccess || srukf_dmalloc(&bp->t_own, sizeof(SyrkOwnArgs) * B) != hipSuccess ||
            srukf_dmalloc(&bp->t_step, sizeof(Step64Args) * B) != hipSuccess || srukf_dmalloc(&bp->t_exp, sizeof(ExpandArgs) * B) != hipSuccess) {
            batch_plan_destroy(c0->device, grp); c0->err = "run_frames_batch: out of device memory (argument tables)"; return SRUKF_ERR_NOMEM;
        }
    }
    // the tables' contents (buffers may have been re-staged or rebuilt since the last call: rewritten every call, the captured launches only hold the tables' addresses)
    std::vector<Pxy2Args> a1(B); std::vector<GainArgs> a2(B); std::vector<SyrkArgs> a3(B); std::vector<SyrkOwnArgs> a4(B); std::vector<Step64Args> a5(B); std::vector<ExpandArgs> a6(B);
    std::vector<unsigned long long> sig;
    for (int b = 0; b < B; b++) {
        srukf_ctx* c = cs[b];
        const KDims& d = c->d;
        a1[b] = Pxy2Args{ c->DZ, c->shadowA, c->Utp, c->P1,
                          MeasArgs{ c->X, c->sigR, c->sigR, c->Z, c->mpart, c->h, c->Si, c->vis, c->PxyR, c->fs, (d.N + 31) / 32, null_skip(c), 1, 1, c->Cmat } };
        a2[b] = GainArgs{ c->Ut, c->PxyR, c->Si, c->vis, c->h, c->z_seq, c->m_seq, c->fs, c->dxp, rank_args(c), c->Cmat, c->S, c->P1, c->DZ, c->sigR, c->Z };
        a3[b] = SyrkArgs{ c->shadowA, c->Utp, c->Wf, c->fs, c->dxp, c->X, rank_args(c, true), (const double*)((const char*)c->fs + offsetof(FrameScalars, Xr1)) };
        a4[b] = SyrkOwnArgs{ c->shadowA, c->Utp, c->Wf, c->fs };
        if (!c->slabW) {
            HIPCHK(c, srukf_dmalloc(&c->slabW, sizeof(double) * 2 * 64 * (size_t)d.np)); HIPCHK(c, srukf_dmalloc(&c->slabL, sizeof(double) * 2 * 64 * (size_t)d.np));
            HIPCHK(c, hipMemset(c->slabW, 0, sizeof(double) * 2 * 64 * (size_t)d.np)); HIPCHK(c, hipMemset(c->slabL, 0, sizeof(double) * 2 * 64 * (size_t)d.np));
        }
        a5[b] = Step64Args{ c->Wf, c->G, c->D, { c->pan[0], c->pan[1] }, c->slabW, c->slabL };
        a6[b] = ExpandArgs{ c->G, c->D, c->red_perm, c->red_iperm, c->gdiag, c->fs, c->X, c->S, c->shadowA, c->sigR, c->Z, c->DZ };
    }
    {
        const srukf_ctx* c = c0;
        for (unsigned long long v : { (unsigned long long)(size_t)st, (unsigned long long)c->d.N, (unsigned long long)c->red_r, (unsigned long long)c->red_Tp, (unsigned long long)(size_t)c->pxy2_tiles,
                                      (unsigned long long)c->n_pxy2_tiles, (unsigned long long)(size_t)c->syrk_head_tiles, (unsigned long long)c->n_syrk_head_tiles,
                                      (unsigned long long)(size_t)c->gplan_red.tiles, (unsigned long long)c->gplan_red.ntiles, (unsigned long long)c->pxy2_split_b0 }) sig.push_back(v);
    }
    {
        // (the captured launches also embed the parameters and the weights by value)
        unsigned long long h = 1469598103934665603ull;
        const unsigned char* pb = (const unsigned char*)&c0->p;
        for (size_t q = 0; q < sizeof c0->p; q++) h = (h ^ pb[q]) * 1099511628211ull;
        const unsigned char* wb = (const unsigned char*)&c0->w;
        for (size_t q = 0; q < sizeof c0->w; q++) h = (h ^ wb[q]) * 1099511628211ull;
        sig.push_back(h); sig.push_back((unsigned long long)g_dbg_batch_split.load());
    }
    if (sig != bp->sig) { batch_plan_drop_graphs(bp); bp->sig = sig; }
    HIPCHK(c0, hipSetDevice(c0->device));
    HIPCHK(c0, hipMemcpyAsync(bp->t_pxy2, a1.data(), sizeof(Pxy2Args) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipMemcpyAsync(bp->t_gain, a2.data(), sizeof(GainArgs) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipMemcpyAsync(bp->t_syrk, a3.data(), sizeof(SyrkArgs) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipMemcpyAsync(bp->t_own, a4.data(), sizeof(SyrkOwnArgs) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipMemcpyAsync(bp->t_step, a5.data(), sizeof(Step64Args) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipMemcpyAsync(bp->t_exp, a6.data(), sizeof(ExpandArgs) * B, hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipStreamSynchronize(st));                      // (the host vectors are pageable and go out of scope)
    const bool graphs = c0->use_graph;                         // ("graphs" 0 / "use_graph" 0: eager launches, which rocprofv3 --pmc needs)
    if (graphs && !bp->e1) {
        int rc = batch_capture(bp, st, 1, &bp->g1, &bp->e1); if (rc) return rc;
        rc = batch_capture(bp, st, SRUKF_GRAPH_FRAMES, &bp->g8, &bp->e8); if (rc) return rc;
    }
    // start of the run, per filter: frame counter / flags / trajectory base, the first frame's table of robot poses and its projection (the frames behind it are
    // projected by their predecessors' tails)
    for (int b = 0; b < B; b++) {
        srukf_ctx* c = cs[b];
        double* traj = dt[b] ? dt[b] - (size_t)8 * first : nullptr;
        launch_set_run(st, c->fs, first, c->async_pending ? 0 : 1, traj);
        srukf_launch_sigr_rows(st, c->d, c->w, c->X, c->S, c->sigR, c->fs, c->red_iperm, c->red_r);
        srukf_launch_project_table(st, c->d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->Z, c->DZ, c->fs, rank_args(c, false, true), null_skip(c));
        c->xr1_pending = false; c->dx_pending = false;          // (the batched launches apply both themselves, every frame)
        c->async_pending = true; c->phase = 0;
    }
    int f = 0;
    if (graphs) {
        for (; f + SRUKF_GRAPH_FRAMES <= count; f += SRUKF_GRAPH_FRAMES) HIPCHK(c0, hipGraphLaunch(bp->e8, st));
        for (; f < count; f++) HIPCHK(c0, hipGraphLaunch(bp->e1, st));
    } else for (; f < count; f++) batch_frame(bp, st);
    HIPCHK(c0, hipGetLastError());
    return SRUKF_OK;
}


extern "C" {

// B filters (independent sequences: Monte-Carlo runs, several cameras) through the same block of staged frames, concurrently on one
// GPU.  Every filter keeps its own context and stream; the frames are issued round-robin in chunks of two captured 8-frame graphs,
// so that the filters' launches interleave on the device, then all are awaited.  Filters that were left in SRUKF_GPU_EXCLUSIVE are
// switched to SRUKF_GPU_SHARED first (two exclusive persistent launches do not fit the GPU together).  A filter whose block holds
// a flagged frame (theta clamp) is rerun alone through srukf_run_frames, which recovers by itself.
// traj_host: [B][count][8] or null; status: per-filter return codes or null.  Returns the first error.
int srukf_run_frames_batch(srukf_ctx* const* ctxs, int B, int first, int count, int mode, double* traj_host, int* status)
{
    if (!ctxs || B < 1 || count < 1) return SRUKF_ERR_BAD_ARG;
    for (int b = 0; b < B; b++) if (!ctxs[b]) return SRUKF_ERR_BAD_ARG;
    std::vector<double*> dt(B, nullptr);
    std::vector<int> rcs(B, SRUKF_OK), canon0(B, 0);
    int rc = SRUKF_OK;
    for (int b = 0; b < B && rc == SRUKF_OK; b++) {
        srukf_ctx* c = ctxs[b];
        if (hipSetDevice(c->device) != hipSuccess) rc = SRUKF_ERR_HIP;
        if (rc == SRUKF_OK) { step_commit_motion(c); step_state_replaced(c); }      // (a filter that was stepped frame by frame before: nothing of that chain survives a replay)
        if (rc == SRUKF_OK && srukf_dmalloc((void**)&dt[b], sizeof(double) * 8 * (size_t)count) != hipSuccess) { c->err = "run_frames_batch: out of device memory"; rc = SRUKF_ERR_NOMEM; }
        if (rc == SRUKF_OK && !c->ckS) {                        // the state before the block, for the recovery of a flagged filter
            const size_t np = c->d.np;
            if (srukf_dmalloc((void**)&c->ckS, sizeof(double) * np * np) != hipSuccess || srukf_dmalloc((void**)&c->ckX, sizeof(double) * np) != hipSuccess) { c->err = "run_frames_batch: out of device memory (checkpoint)"; rc = SRUKF_ERR_NOMEM; }
        }
        if (rc == SRUKF_OK) {
            const size_t np = c->d.np;
            hipMemcpyAsync(c->ckS, c->S, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->stream);
            hipMemcpyAsync(c->ckX, c->X, sizeof(double) * np, hipMemcpyDeviceToDevice, c->stream);
            canon0[b] = c->null_canonical ? 1 : 0;
        }
    }
    // The batched launches (one launch per stage for all filters, one stream: batch_run) where the filters have one shape and run the default launch sequence.
    // Filters whose structurally null rows are not canonical yet (a fresh state) run their first frame on their own, one after the other.
    int done0 = 0;
    bool wide = false;
    // (decided before anything runs: a filter that changes its launch sequence in the middle of a block — exclusive for its first frame, shared behind it — rebuilds its
    //  permuted copy from S in between and is then no longer bit-identical to the same filter running alone)
    bool fresh = false, others_ok = true;
    for (int b = 0; b < B; b++) { fresh = fresh || (ctxs[b]->red_r > 0 && !ctxs[b]->null_canonical); others_ok = others_ok && ctxs[b]->odo_seq && ctxs[b]->seqF >= first + count; }
    if (rc == SRUKF_OK && B > 1 && mode == SRUKF_UPDATE_BATCHED && g_dbg_batch_wide.load() && others_ok && batch_eligible(ctxs, B, true) && !(fresh && count < 2)) {
        for (int b = 0; b < B; b++) hipStreamSynchronize(ctxs[b]->stream);       // the checkpoint copies; whatever the filters did before
        if (fresh) {
            for (int b = 0; b < B; b++) {
                rcs[b] = srukf_run_frames_async(ctxs[b], first, 1, mode, dt[b]);
                if (rcs[b] == SRUKF_OK) rcs[b] = srukf_synchronize(ctxs[b]);
            }
            done0 = 1;
        }
        // the filters that are still clean (one flagged in its first frame is rerun alone below) go on as one batch — or, if what is left cannot be batched, one
        // after the other: filters in exclusive mode cannot share the GPU, and switching them to the shared form in the middle of a block would cost the bit-identity
        std::vector<srukf_ctx*> sub; std::vector<double*> dtb; std::vector<int> idx;
        for (int b = 0; b < B; b++) if (rcs[b] == SRUKF_OK) { sub.push_back(ctxs[b]); dtb.push_back(dt[b] + (size_t)8 * done0); idx.push_back(b); }
        const int nb = (int)sub.size();
        if (nb >= 2 && batch_eligible(sub.data(), nb)) {
            // groups of filters side by side, each group one batch on a stream of its own: while one group sits in a launch that cannot fill the GPU
            // (the pivot chains of a panel step), the other groups' launches do
            // (measured at N = 200, round 4, aggregate frames/s with 1 / 2 / 3 / 4 groups: 8 filters 10 270 / 11 450 / 11 280 / 11 700; 16: 12 710 / 14 260 / 14 300 / 14 790;
            //  32: 14 600 / 15 560 / 15 790 / 16 430; 48: 14 840 / 15 710 / 16 460 / 16 650)
            int G = g_dbg_batch_groups.load() > 0 ? g_dbg_batch_groups.load() : SRUKF_BATCH_GROUPS_MAX;
            G = std::max(1, std::min(std::min(G, SRUKF_BATCH_GROUPS_MAX), nb / 2));
            for (int grp = 0; grp < G && rc == SRUKF_OK; grp++) {
                const int b0 = (int)((long long)nb * grp / G), b1 = (int)((long long)nb * (grp + 1) / G);
                rc = batch_run(sub.data() + b0, b1 - b0, first + done0, count - done0, dtb.data() + b0, grp);
            }
            for (int grp = 0; grp < G; grp++) {
                const int b0 = (int)((long long)nb * grp / G);
                if (hipStreamSynchronize(batch_stream(sub[b0]->device, grp)) != hipSuccess && rc == SRUKF_OK) { sub[b0]->err = "run_frames_batch: the batched launches failed"; rc = SRUKF_ERR_HIP; }
            }
        } else {
            for (int q = 0; q < nb && rc == SRUKF_OK; q++) {
                rcs[idx[q]] = srukf_run_frames_async(sub[q], first + done0, count - done0, mode, dtb[q]);
                if (rcs[idx[q]] != SRUKF_OK && rcs[idx[q]] != SRUKF_ERR_CLAMP_PENDING) rc = rcs[idx[q]];
                hipStreamSynchronize(sub[q]->stream);
            }
        }
        wide = true;
    }
    if (!wide) {
        // one stream per filter, persistent launches behind the admission gate: one tenant per filter up to SRUKF_MAX_TENANTS (every filter's persistent launch
        // admitted at once, each on cus / tenants CUs); a filter in per-panel mode (forced, or after an abandoned persistent launch) stays there
        for (int b = 0; b < B && rc == SRUKF_OK; b++)
            if (B > 1 && ctxs[b]->gmw_shared != 2) rc = set_shared(ctxs[b], 1, std::min(std::max(B, 2), SRUKF_MAX_TENANTS));
        const int chunk = 2 * SRUKF_GRAPH_FRAMES;
        for (int k0 = done0; k0 < count && rc == SRUKF_OK; k0 += chunk)
            for (int b = 0; b < B && rc == SRUKF_OK; b++) {
                if (rcs[b] != SRUKF_OK) continue;                          // (flagged in its first frame: rerun alone below)
                rcs[b] = srukf_run_frames_async(ctxs[b], first + k0, std::min(chunk, count - k0), mode, dt[b] + (size_t)8 * k0);
                if (rcs[b] != SRUKF_OK && rcs[b] != SRUKF_ERR_CLAMP_PENDING) rc = rcs[b];
            }
    }
    for (int b = 0; b < B; b++) {
        srukf_ctx* c = ctxs[b];
        int r = srukf_synchronize(c);
        if (rcs[b] == SRUKF_OK) rcs[b] = r;
        if (rcs[b] == SRUKF_ERR_CLAMP_PENDING) {
            // rewind this filter and let the synchronous form (checkpoint, exact path for the flagged frame) run its block alone
            const size_t np = c->d.np;
            hipMemcpyAsync(c->S, c->ckS, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->stream);
            hipMemcpyAsync(c->X, c->ckX, sizeof(double) * np, hipMemcpyDeviceToDevice, c->stream);
            if (c->null_canonical != (canon0[b] != 0)) { c->null_canonical = canon0[b] != 0; drop_graphs(c); }
            quantize_state(c); shadow_rebuild(c);
            std::vector<double> th((size_t)8 * count);
            rcs[b] = srukf_run_frames(c, first, count, mode, th.data());
            if (rcs[b] == SRUKF_OK && dt[b]) hipMemcpy(dt[b], th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice);
        }
        if (rcs[b] == SRUKF_OK && traj_host && dt[b]) hipMemcpy(traj_host + (size_t)b * 8 * count, dt[b], sizeof(double) * 8 * (size_t)count, hipMemcpyDeviceToHost);
        if (rcs[b] != SRUKF_OK && rc == SRUKF_OK) rc = rcs[b];
        if (dt[b]) srukf_dfree(dt[b]);
        if (status) status[b] = rcs[b];
    }
    return rc;
}

}  // extern "C"
