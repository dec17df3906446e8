// srukf_debug.hip — srukf_debug_* (measurement / test switches, diagnostic read-outs; none of them is needed to use the library) and the stand-alone numeric
// primitives the parity tests call (srukf_gmw_host, srukf_project_host).

#include "srukf_ctx.h"
using namespace srukf_impl;

// A/B switches of srukf_debug_set (process-wide; measurement and test knobs, all 1 in the product):
//   gmw_persist  1 = one persistent launch per factorisation, 0 = one launch per 64-row panel
//   gmw_fused    0 = the persistent launch reads every tile from G (k_syrk computes all of them)
//   rank_fused   0 = the rank-aware form always goes through the full k_syrk + permutation pass
//   rank_fold    0 = the owners never form their tiles themselves in the rank-aware replay (k_syrk over all kept rows instead)
//   rank_aware   0 = no context looks for structurally null directions (per filter: srukf_set_rank_aware)
//   graphs       0 = contexts created from now on launch eagerly (profilers with --pmc; per filter: key "use_graph")
// (atomics: another thread's context may be launching while a switch is set; a switch applies to whatever is built or captured afterwards)

// Diagnostic copy of a device work buffer (synchronises the stream): "Z" (L x mp), "DZ" (np x mp), "sigR" ((L + 1) x 8), "Cmat" (n x 4),
// "Xr1" (4), "Utp" / "P1" (mp x np), "h" (2N), "Si" (4N).  count doubles from the start of the buffer.
namespace srukf_impl {
std::atomic<int> g_dbg_gmw_persist{1}, g_dbg_gmw_fused{1}, g_dbg_rank_fused{1}, g_dbg_rank_fold{1}, g_dbg_rank_aware{1}, g_dbg_graphs{1};
std::atomic<int> g_dbg_mem_split{1};          // "mem_split" 0: sizes beyond two tiles per worker keep the memory-tile instance of k_gmw_persist instead of the split form; 2 (measurements): also
                                              // the plans whose workers own two register tiles each
std::atomic<int> g_dbg_shared_tenants{2};     // "shared_tenants": persistent launches that share the GPU after srukf_set_exclusive(SRUKF_GPU_SHARED)
// srukf_run_frames_batch: "batch_wide" 0: never the batched launches (one stream per filter, persistent launches behind the gate: round 3's form); "batch_groups": groups the batched
// filters are cut into (0: as many as pay); "batch_split" 0: one k_gmw_step64_b launch per panel (every tile recomputes its slabs) instead of slabs + plain updates
std::atomic<int> g_dbg_batch_wide{1}, g_dbg_batch_groups{0}, g_dbg_batch_split{1};
std::atomic<int> g_dbg_timing{0}, g_dbg_fold_head{0}, g_dbg_fold_force{0}, g_dbg_ctx_keep{24}, g_dbg_batch_xcd{1}, g_dbg_batch_k128{1};
// "head_fold_free": CUs a plan must leave beside the pivot and the workers for the head fold (helper workgroups of the persistent launch); default SRUKF_HEAD_FOLD_MIN_FREE_CUS
std::atomic<int> g_dbg_head_fold_free{0};
}  // namespace srukf_impl

static bool debug_buffer(srukf_ctx* c, const char* key, double** ptr, long long* cap)
{
    const KDims& d = c->d;
    const GmwPlan& gp = c->red_r > 0 ? c->gplan_red : c->gplan;
    double* src = nullptr; long long n = 0;
    if (!strcmp(key, "Z")) { src = c->Z; n = (long long)d.L * d.mp; }
    else if (!strcmp(key, "DZ")) { src = c->DZ; n = (long long)d.np * d.mp; }
    else if (!strcmp(key, "sigR")) { src = c->sigR; n = (long long)(d.L + 1) * 8; }
    else if (!strcmp(key, "Cmat")) { src = c->Cmat; n = (long long)d.n * 4; }
    else if (!strcmp(key, "Xr1")) { src = (double*)((char*)c->fs + offsetof(FrameScalars, Xr1)); n = 4; }
    else if (!strcmp(key, "Utp")) { src = c->Utp; n = c->Utp ? (long long)d.mp * d.np : 0; }
    else if (!strcmp(key, "P1")) { src = c->P1; n = c->P1 ? (long long)d.mp * d.np : 0; }
    else if (!strcmp(key, "h")) { src = c->h; n = 2LL * d.N; }
    else if (!strcmp(key, "Si")) { src = c->Si; n = 4LL * d.N; }
    // the operands of one factorisation (scripts/split_replay.py: a split-form pair recorded from a real frame, each launch then replayed alone under the counters)
    else if (!strcmp(key, "Wf")) { src = c->Wf; n = (long long)d.np * d.np; }
    else if (!strcmp(key, "Gbak")) { src = c->Gbak; n = (long long)d.np * d.np; }
    else if (!strcmp(key, "G")) { src = c->G; n = (long long)d.np * d.np; }
    else if (!strcmp(key, "D")) { src = c->D; n = d.np; }
    else if (!strcmp(key, "fold_dbg")) { src = c->fold_sync ? (double*)(c->fold_sync + ((srukf_fold_words(d.mp / 64, d.np / 64) + 1) & ~1)) : nullptr; n = c->fold_sync ? FOLD_DBG_STAMPS : 0; }      // -DSRUKF_FOLD_DBG: time stamps of the gain fold
    else if (!strcmp(key, "gsW")) { src = c->gsW; n = c->gsW ? (long long)c->gs_panels * 64 * d.np : 0; }
    else if (!strcmp(key, "gsL")) { src = c->gsL; n = c->gsL ? (long long)c->gs_panels * 64 * d.np : 0; }
    else if (!strcmp(key, "pans")) { src = (double*)gp.pans; n = gp.pans ? (long long)srukf_gmw_panel_bytes() * gp.T / 8 : 0; }
    else if (!strcmp(key, "sync")) { src = (double*)gp.sync; n = gp.sync ? (long long)srukf_gmw_sync_bytes(gp.T) / 8 : 0; }
    else return false;
    *ptr = src; *cap = n;
    return true;
}

extern "C" {

// Tolerance study only (scripts/mixed_eps_study.py): lets srukf_set_storage accept SRUKF_STORAGE_F32_MIXED below epsilon 1e-9,
// where it is known to diverge — that divergence is what the study documents.
int srukf_debug_allow_mixed(srukf_ctx* c, int on)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    c->debug_allow_mixed = on ? 1 : 0;
    return SRUKF_OK;
}

// Tests only: persistent factorisation launches of this context start WITHOUT their worker workgroups, as if another
// process held the GPU — exercises the bounded waits and the fallback to per-panel launches.
// Test hook: S[row][col] = value on the device, behind the back of everything that tracks S (the null set of the rank-aware
// refactorisation, the permuted copy): the next frame has to notice by itself.
int srukf_debug_poke_state(srukf_ctx* c, int row, int col, double value)
{
    if (!c || row < 0 || col < row || col >= c->d.n) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->S + (size_t)row * c->d.np + col, &value, sizeof(double), hipMemcpyHostToDevice));
    return SRUKF_OK;
}

// Measurement / test switches behind ONE entry point (none of them is needed to use the library; all default to the product
// path).  ctx may be NULL for the process-wide keys listed above g_dbg_*; per-context keys: "use_graph" (0: eager launches),
// "fused_motion" (0: the replay keeps k_motion and k_project as two launches).  Captured graphs are dropped.
int srukf_debug_set(srukf_ctx* c, const char* key, int value)
{
    if (!key) return SRUKF_ERR_BAD_ARG;
    struct { const char* k; std::atomic<int>* v; } globals[] = { { "gmw_persist", &g_dbg_gmw_persist }, { "gmw_fused", &g_dbg_gmw_fused }, { "rank_fused", &g_dbg_rank_fused },
                                                    { "rank_fold", &g_dbg_rank_fold }, { "rank_aware", &g_dbg_rank_aware }, { "graphs", &g_dbg_graphs }, { "mem_split", &g_dbg_mem_split }
                                                  };
    if (!strcmp(key, "batch_split")) { g_dbg_batch_split = value ? 1 : 0; batch_drop_all_graphs(); return SRUKF_OK; }
    if (!strcmp(key, "head_fold_free")) { if (value < 1) return SRUKF_ERR_BAD_ARG; g_dbg_head_fold_free = value; if (c) { hipSetDevice(c->device); step_commit_motion(c); step_invalidate(c); hipStreamSynchronize(c->stream); drop_graphs(c); } return SRUKF_OK; }
    if (!strcmp(key, "batch_groups")) { if (value < 0 || value > SRUKF_BATCH_GROUPS_MAX) return SRUKF_ERR_BAD_ARG; g_dbg_batch_groups = value; return SRUKF_OK; }
    if (!strcmp(key, "batch_wide")) { g_dbg_batch_wide = value ? 1 : 0; return SRUKF_OK; }
    if (!strcmp(key, "timing")) { g_dbg_timing = value ? 1 : 0; return SRUKF_OK; }              // phases of map changes and of flagged frames on stderr
    if (!strcmp(key, "fold_head")) { g_dbg_fold_head = value < 0 ? 0 : value; srukf_gmw_fold_head_override(g_dbg_fold_head); return SRUKF_OK; }   // split fold: block rows formed in front of the pair (0: the rule); before the state is set
    if (!strcmp(key, "batch_k128")) { g_dbg_batch_k128 = value ? 1 : 0; batch_drop_all_graphs(); return SRUKF_OK; }    // 0: every panel's trailing update a pass of its own over G (K = 64)
    if (!strcmp(key, "batch_xcd")) { g_dbg_batch_xcd = value ? 1 : 0; batch_drop_all_graphs(); return SRUKF_OK; }      // 0: k_syrk_b walks the head tiles in the solo launch's order
    if (!strcmp(key, "ctx_keep")) { g_dbg_ctx_keep = value < 0 ? 0 : value; return SRUKF_OK; }       // contexts a handle keeps across map changes (default 24, and at most ~16 GB of them)
    if (!strcmp(key, "fold_force")) { g_dbg_fold_force = value ? 1 : 0; return SRUKF_OK; }         // split fold also where the tile workgroups do not all fit
    if (!strcmp(key, "exact_rl")) { srukf_set_exact_right_looking(value); return SRUKF_OK; }      // 0: k_gmw_col, the left-looking exact path (one row per launch)
    if (!strcmp(key, "shared_tenants")) {                      // applies to filters switched to SRUKF_GPU_SHARED afterwards
        if (value < 2 || value > 8) return SRUKF_ERR_BAD_ARG;
        g_dbg_shared_tenants = value;
        return SRUKF_OK;
    }
    for (auto& g : globals)
        if (!strcmp(key, g.k)) {
            g.v->store(!strcmp(key, "mem_split") ? value : (value ? 1 : 0));
            if (c) { hipSetDevice(c->device); step_commit_motion(c); step_invalidate(c); hipStreamSynchronize(c->stream); drop_graphs(c); if (!strcmp(key, "rank_aware")) return update_null_set(c); }
            return SRUKF_OK;
        }
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!strcmp(key, "use_graph")) c->use_graph = value != 0;
    else if (!strcmp(key, "pxy2")) c->dbg.pxy2 = value ? 1 : 0;
    else if (!strcmp(key, "nullskip")) c->dbg.nullskip = value ? 1 : 0;
    else if (!strcmp(key, "head_fold")) c->dbg.head_fold = value ? 1 : 0;
    else if (!strcmp(key, "tail_fuse")) c->dbg.tail_fuse = value ? 1 : 0;
    else if (!strcmp(key, "table_perm")) c->dbg.table_perm = value ? 1 : 0;
    else if (!strcmp(key, "f32_fuse")) c->dbg.f32_fuse = value ? 1 : 0;
    else if (!strcmp(key, "split_record")) c->dbg.split_record = value ? 1 : 0;
    else if (!strcmp(key, "gain_fold")) c->dbg.gain_fold = value ? 1 : 0;
    else if (!strcmp(key, "null_canon")) c->dbg.null_canon = value ? 1 : 0;
    else if (!strcmp(key, "split_fold")) c->dbg.split_fold = value < 0 ? 0 : value > 2 ? 2 : value;      // (2: measurements — the fold's launch behind a k_syrk that has formed everything)
    else if (!strcmp(key, "mixed_rank")) c->dbg.mixed_rank = value ? 1 : 0;
    else if (!strcmp(key, "mixed_f64_robot")) c->dbg.mixed_f64_robot = value ? 1 : 0;
    else if (!strcmp(key, "mixed_bf16")) c->dbg.mixed_bf16 = value ? 1 : 0;
    else if (!strcmp(key, "mixed_null_ppm")) c->dbg.mixed_null_ppm = value < 0 ? 0 : value;
    else if (!strcmp(key, "step_fast")) c->dbg.step_fast = value ? 1 : 0;
    else if (!strcmp(key, "step_spin")) c->dbg.step_spin = value ? 1 : 0;
    else if (!strcmp(key, "step_early")) c->dbg.step_early = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(key, "view_auto")) { c->dbg.view_auto = value ? 1 : 0; if (!value) { c->view_auto = false; c->view_cached = false; } }
    else if (!strcmp(key, "step_fuse_export")) c->dbg.step_fuse_export = value ? 1 : 0;
    else if (!strcmp(key, "fused_motion")) c->dbg.fused_motion = value < 0 ? 0 : value > 2 ? 2 : value;
    else { c->err = std::string("srukf_debug_set: unknown key ") + key; return SRUKF_ERR_BAD_ARG; }
    drop_graphs(c);
    return SRUKF_OK;
}

// Diagnostic builds only (make EXTRA=-DSRUKF_GMW_DBG): host-visible time stamps of the persistent factorisation launch of THIS context's rank-aware
// plan (GMW_TS in srukf_gmw_persist.hip).  buf receives 4096 unsigned long longs: [2048 + 8 p + slot] = s_memrealtime (10 ns ticks) of pivot iteration p.
int srukf_debug_gmw_stamps(srukf_ctx* c, unsigned long long* buf)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    // (device memory: stamps written to pinned host memory cross PCIe, and every later s_waitcnt vmcnt(0) of the stamping wave waits for them — the timeline of the
    //  diagnostic build then shows 17.9 us per panel where the product runs 14.5)
    static unsigned long long* dbuf = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    GmwPlan& g = c->red_r > 0 ? c->gplan_red : c->gplan;
    if (!g.sync) return SRUKF_ERR_SEQUENCE;
    if (!dbuf) { HIPCHK(c, hipMalloc((void**)&dbuf, 8 * 4096)); HIPCHK(c, hipMemset(dbuf, 0, 8 * 4096)); }
    if (buf) HIPCHK(c, hipMemcpy(buf, dbuf, 8 * 4096, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy((char*)g.sync + offsetof(GmwSync, dbg), &dbuf, 8, hipMemcpyHostToDevice));      // armed for the launches that follow
    return SRUKF_OK;
}

// Diagnostic read-out of the device-resident frame scalars (synchronises the stream): "gmw_aborts", "clamp_rows", "frame", "frozen", "gate_timeouts"; "gmw_shared", "split_form"
int srukf_debug_get(srukf_ctx* c, const char* key, long long* value)
{
    if (!c || !key || !value) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->hfs, c->fs, sizeof(FrameScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!strcmp(key, "gmw_aborts")) *value = c->hfs->gmw_aborts;
    else if (!strcmp(key, "clamp_rows")) *value = c->hfs->clamp_rows;
    else if (!strcmp(key, "frame")) *value = c->hfs->frame;
    else if (!strcmp(key, "frozen")) *value = c->hfs->frozen;
    else if (!strcmp(key, "gate_timeouts")) *value = c->hfs->gate_timeouts;
    else if (!strcmp(key, "gmw_shared")) *value = c->gmw_shared;
    else if (!strcmp(key, "split_off")) *value = c->split_off ? 1 : 0;
    else if (!strncmp(key, "pad", 3) && key[3] >= '1' && key[3] <= '7' && !key[4]) {
        // diagnostic builds (-DSRUKF_GMW_DBG): words 1..7 of the sync block's pad (helpers started / finished, head counters at the last exit, grid, helpers); cleared by the read
        const GmwPlan& gp = c->red_r > 0 ? c->gplan_red : c->gplan;
        unsigned long long v = 0, zero = 0;
        const size_t off = offsetof(GmwSync, pad) + 8 * (size_t)(key[3] - '0');
        if (gp.sync) { HIPCHK(c, hipMemcpy(&v, (char*)gp.sync + off, sizeof v, hipMemcpyDeviceToHost)); HIPCHK(c, hipMemcpy((char*)gp.sync + off, &zero, sizeof zero, hipMemcpyHostToDevice)); }
        *value = (long long)v;
    }
    else if (!strcmp(key, "abort_code")) {
        // who abandoned a persistent launch first, and where (gmw_abandon, srukf_gmw_persist.hip: site << 32 | blockIdx + 1; 0: nobody since the last read); cleared by the read
        const GmwPlan& gp = c->red_r > 0 ? c->gplan_red : c->gplan;
        unsigned long long code = 0, zero = 0;
        if (gp.sync) {
            HIPCHK(c, hipMemcpy(&code, (char*)gp.sync + offsetof(GmwSync, pad), sizeof code, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy((char*)gp.sync + offsetof(GmwSync, pad), &zero, sizeof zero, hipMemcpyHostToDevice));
        }
        *value = (long long)code;
    }
    else if (!strcmp(key, "step_fast")) *value = c->step_fast_frames;          // frames the step-wise API ran on the staged replay's launch sequence / on its own
    else if (!strcmp(key, "step_slow")) *value = c->step_slow_frames;
    else if (!strcmp(key, "exact_frames")) *value = c->exact_frames;
    else if (!strcmp(key, "fold_seqs")) *value = c->fold_seqs;
    else if (!strcmp(key, "split_fold_seqs")) *value = c->split_fold_seqs;
    else if (!strncmp(key, "pxy2_stamp", 10) && key[10] >= '1' && key[10] <= '7') { const unsigned long long* t = (const unsigned long long*)(c->hmeas + c->d.mp + 5 * (size_t)c->d.N); *value = (long long)(t[key[10] - '0'] - t[0]); }   // diagnostic build (-DSRUKF_PXY2_DBG): 2 motion end, 4..7 sampled tiles' ends
    else if (!strcmp(key, "meas_flag_ticks")) { const unsigned long long* t = (const unsigned long long*)(c->hmeas + c->d.mp + 5 * (size_t)c->d.N); *value = (long long)(t[1] - t[0]); }   // last fast-path k_pxy2: first workgroup's start -> statistics flag, 10 ns ticks
    else if (!strcmp(key, "view_hits")) *value = c->view_hits;                     // srukf_get_frame_view calls served from the view an update exported with its status
    else if (!strcmp(key, "view_auto")) *value = c->view_auto ? 1 : 0;
    else if (!strcmp(key, "split_form")) *value = split_form(c, c->red_r > 0 ? c->gplan_red : c->gplan) ? 1 : 0;       // would the next persistent factorisation be the split form?
    else if (!strncmp(key, "plan_", 5)) {
        // which launch plan the next staged frame takes (tests assert it next to the oracle comparison: every N is a product size, SLAM.cpp:552-562, 2443-2460)
        const GmwPlan& gp = c->red_r > 0 ? c->gplan_red : c->gplan;
        const bool persist = gmw_use_persist(c) && gmw_plan_persists(c, gp);
        const char* k = key + 5;
        if (!strcmp(k, "T")) *value = gp.T;
        else if (!strcmp(k, "Tp")) *value = gp.Tp;
        else if (!strcmp(k, "tiles")) *value = gp.nreal;
        else if (!strcmp(k, "workers")) *value = gp.workers;
        else if (!strcmp(k, "persist")) *value = persist ? 1 : 0;                                        // 0: one launch per 64-row panel
        else if (!strcmp(k, "register_form")) *value = (persist && !split_form(c, gp) && srukf_gmw_register_form(gp.T, gp.Tp, gp.ntiles, gp.workers)) ? 1 : 0;
        else if (!strcmp(k, "tiles_per_worker")) *value = gp.workers > 0 ? (gp.nreal + gp.workers - 1) / gp.workers : -1;
        else if (!strcmp(k, "fold")) *value = replay_red_fused(c) ? 1 : 0;                               // the owners form their tiles of S^T S - U U^T themselves
        else if (!strcmp(k, "head_fold")) *value = (replay_red_fused(c) && head_fold_ok(c)) ? 1 : 0;     // ... and the head tiles ride on the persistent launch
        else if (!strcmp(k, "red_perm")) *value = (!replay_red_fused(c) && replay_red_perm(c)) ? 1 : 0;  // k_syrk over the kept rows in permuted order
        else if (!strcmp(k, "motion")) *value = replay_motion_mode(c);                                   // 2: "table" mode
        else if (!strcmp(k, "fuse")) *value = replay_fuse_mode(c) ? 1 : 0;                               // "fused tail" mode
        else if (!strcmp(k, "kept")) *value = c->red_r;
        else if (!strcmp(k, "sync_doubles")) *value = gp.sync ? srukf_gmw_sync_bytes(gp.T) / 8 : 0;      // sizes of the byte buffers srukf_debug_copy counts in doubles
        else if (!strcmp(k, "pans_doubles")) *value = gp.pans ? (long long)srukf_gmw_panel_bytes() * gp.T / 8 : 0;
        else if (!strcmp(k, "slab_panels")) *value = c->gs_panels;
        else return SRUKF_ERR_BAD_ARG;
    }
    else return SRUKF_ERR_BAD_ARG;
    return SRUKF_OK;
}

int srukf_debug_copy(srukf_ctx* c, const char* key, double* out, long long count)
{
    if (!c || !key || !out || count < 0) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double* src = nullptr; long long cap = 0;
    if (!debug_buffer(c, key, &src, &cap)) return SRUKF_ERR_BAD_ARG;
    if (!src || count > cap) return SRUKF_ERR_DIM_MISMATCH;
    HIPCHK(c, hipMemcpy(out, src, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    return SRUKF_OK;
}

// the other direction (same keys): `count` doubles to the start of the buffer
int srukf_debug_upload(srukf_ctx* c, const char* key, const double* in, long long count)
{
    if (!c || !key || !in || count < 0) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double* dst = nullptr; long long cap = 0;
    if (!debug_buffer(c, key, &dst, &cap)) return SRUKF_ERR_BAD_ARG;
    if (!dst || count > cap) return SRUKF_ERR_DIM_MISMATCH;
    HIPCHK(c, hipMemcpy(dst, in, sizeof(double) * (size_t)count, hipMemcpyHostToDevice));
    return SRUKF_OK;
}

int srukf_debug_starve_workers(srukf_ctx* c, int on)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    c->debug_starve = on == 2 ? 2 : on ? 1 : 0;              // 2: only split-form pairs start without their tile launch (the single persistent launch below them is not starved)
    step_invalidate(c);
    drop_graphs(c);                                    // the captured frames contain one or the other launch sequence
    return SRUKF_OK;
}

int srukf_gmw_host(int device, int n, const double* G, double* S_out, double* D_out, double epsilon, int force_slow, int* clamp_hit)
{
    if (n < 1 || !G || !S_out) return SRUKF_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SRUKF_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SRUKF_ERR_NO_DEVICE;
    const int np = round_up(n, SRUKF_PAD);
    const size_t bytes = sizeof(double) * (size_t)np * np;
    std::vector<double> hG((size_t)np * np, 0.0), hS((size_t)np * np, 0.0), hD(np, 0.0);
    for (int r = 0; r < n; r++) for (int c = r; c < n; c++) hG[(size_t)r * np + c] = G[(size_t)r * n + c];
    // every device resource of the call in one holder: released on every path out
    struct Res {
        double *dG = nullptr, *dS = nullptr, *dD = nullptr, *dWf = nullptr; unsigned long long* dTh = nullptr; FrameScalars* dFs = nullptr;
        void* pan[2] = { nullptr, nullptr }; GmwPlan gp;
        ~Res() { for (void* b : { (void*)dG, (void*)dS, (void*)dD, (void*)dWf, (void*)dTh, (void*)dFs, pan[0], pan[1] }) if (b) srukf_dfree(b); gmw_plan_destroy(gp); }
    } r;
#define GH(call) do { if ((call) != hipSuccess) return SRUKF_ERR_HIP; } while (0)
    GH(srukf_dmalloc((void**)&r.dG, bytes)); GH(srukf_dmalloc((void**)&r.dS, bytes)); GH(srukf_dmalloc((void**)&r.dWf, bytes));
    GH(srukf_dmalloc((void**)&r.dD, sizeof(double) * np)); GH(srukf_dmalloc((void**)&r.dTh, sizeof(unsigned long long) * np)); GH(srukf_dmalloc((void**)&r.dFs, sizeof(FrameScalars)));
    GH(hipMemcpy(r.dG, hG.data(), bytes, hipMemcpyHostToDevice));
    GH(hipMemset(r.dS, 0, bytes)); GH(hipMemset(r.dWf, 0, bytes)); GH(hipMemset(r.dTh, 0, sizeof(unsigned long long) * np));
    GH(hipMemset(r.dFs, 0, sizeof(FrameScalars))); GH(hipMemset(r.dD, 0, sizeof(double) * np));
    hipStream_t st = nullptr;
    srukf_launch_gmw_stats(st, n, np, r.dG, r.dFs);
    FrameScalars fs;
    if (!force_slow) {
        // the plan knows how many workgroups THIS device can keep resident (CU count); workers < 0: per-panel launches
        if (gmw_persist_mode()) { const int rc = gmw_plan_create(r.gp, np, st); if (rc) return rc; }
        if (gmw_persist_mode() && r.gp.workers >= 0) {
            srukf_launch_gmw_persist(st, n, np, epsilon, r.dG, r.gp.pans, r.dD, r.dS, r.gp.sync, r.gp.tiles, r.gp.ntiles, r.gp.workers, r.dFs, nullptr, nullptr, 0, 0, 0, 0, 0);
        } else {
            GH(srukf_dmalloc(&r.pan[0], srukf_gmw_panel_bytes())); GH(srukf_dmalloc(&r.pan[1], srukf_gmw_panel_bytes()));
            GH(hipMemset(r.pan[0], 0, srukf_gmw_panel_bytes())); GH(hipMemset(r.pan[1], 0, srukf_gmw_panel_bytes()));
            int pb = 0;
            for (int j0 = -64; j0 + 64 < np; j0 += 64, pb ^= 1)
                srukf_launch_gmw_step64(st, n, np, j0, epsilon, r.dG, r.pan[pb ^ 1], r.pan[pb], r.dD, r.dS, nullptr);
        }
        GH(hipDeviceSynchronize());
        srukf_launch_gmw_check(st, n, np, r.dD, r.dS, r.dFs, nullptr, 0, nullptr);
        GH(hipMemcpy(&fs, r.dFs, sizeof fs, hipMemcpyDeviceToHost));
        if (clamp_hit) *clamp_hit = fs.clamp_rows;
        if (fs.clamp_rows > 0) force_slow = 2;   // same contract as srukf_update: redo on the exact path
    }
    if (force_slow) {
        GH(hipMemcpy(r.dG, hG.data(), bytes, hipMemcpyHostToDevice));
        GH(hipMemset(r.dTh, 0, sizeof(unsigned long long) * np));
        GH(hipMemset(r.dS, 0, bytes));
        for (int j = 0; j < n; j++) srukf_launch_gmw_col(st, n, np, j, epsilon, r.dG, r.dWf, r.dD, r.dTh, r.dFs, r.dS);
        GH(hipMemcpy(&fs, r.dFs, sizeof fs, hipMemcpyDeviceToHost));
        if (clamp_hit && force_slow == 1) *clamp_hit = fs.clamp_rows;
    }
    GH(hipDeviceSynchronize());
    GH(hipMemcpy(hS.data(), r.dS, bytes, hipMemcpyDeviceToHost));
    GH(hipMemcpy(hD.data(), r.dD, sizeof(double) * np, hipMemcpyDeviceToHost));
#undef GH
    for (int rr = 0; rr < n; rr++) memcpy(S_out + (size_t)rr * n, hS.data() + (size_t)rr * np, sizeof(double) * n);
    if (D_out) memcpy(D_out, hD.data(), sizeof(double) * n);
    return SRUKF_OK;
}

int srukf_project_host(int device, const srukf_params* p, int count, const double* feat6, const double* pos3, const double* psi,
                       const double* err2, double* uv_out)
{
    if (!p || count < 1 || !feat6 || !pos3 || !psi || !err2 || !uv_out) return SRUKF_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SRUKF_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SRUKF_ERR_NO_DEVICE;
    double *df, *dp, *ds, *de, *dout;
    srukf_dmalloc((void**)&df, sizeof(double) * 6 * count); srukf_dmalloc((void**)&dp, sizeof(double) * 3 * count);
    srukf_dmalloc((void**)&ds, sizeof(double) * count); srukf_dmalloc((void**)&de, sizeof(double) * 2 * count); srukf_dmalloc((void**)&dout, sizeof(double) * 2 * count);
    hipMemcpy(df, feat6, sizeof(double) * 6 * count, hipMemcpyHostToDevice); hipMemcpy(dp, pos3, sizeof(double) * 3 * count, hipMemcpyHostToDevice);
    hipMemcpy(ds, psi, sizeof(double) * count, hipMemcpyHostToDevice); hipMemcpy(de, err2, sizeof(double) * 2 * count, hipMemcpyHostToDevice);
    srukf_launch_project_points(nullptr, *p, count, df, dp, ds, de, dout);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(uv_out, dout, sizeof(double) * 2 * count, hipMemcpyDeviceToHost);
    srukf_dfree(df); srukf_dfree(dp); srukf_dfree(ds); srukf_dfree(de); srukf_dfree(dout);
    return e == hipSuccess ? SRUKF_OK : SRUKF_ERR_HIP;
}

}  // extern "C"
