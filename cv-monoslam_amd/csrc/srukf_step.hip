// srukf_step.hip — the step-wise API (srukf_predict_motion / srukf_predict_measurement / srukf_update: one frame at a time with the host's association in between,
// SLAM.cpp:87-112) and its fast path: the staged replay's own frame cut at the association step.  gfx950 only.
#include "srukf_ctx.h"
#include <chrono>
using namespace srukf_impl;

namespace srukf_impl {

// the state is about to be replaced or read by somebody outside the step-wise fast path
// the checkpoint copy of the frame in flight (its own stream) must be complete before anything on the filter's stream changes S or X, or reads the checkpoint
void step_ck_join(srukf_ctx* c)
{
    if (c->ck_pending) { hipStreamWaitEvent(c->stream, c->ck_e2, 0); c->ck_pending = false; }
    // ... and the copy an update submitted ahead for the NEXT frame (second buffer pair, ck_e3: it becomes ck_e2 / ck_pending once its frame turns out clean; until
    // then — a flagged frame that is rewound, a state that is replaced — it is still reading S and X)
    if (c->ck3_inflight) { hipStreamWaitEvent(c->stream, c->ck_e3, 0); c->ck3_inflight = false; }
}
void step_state_replaced(srukf_ctx* c) { step_ck_join(c); c->step_uncommitted = false; c->step_fast = false; c->xr1_pending = false; step_invalidate(c); c->f32_stale = false; c->robot_cached = false; }

}  // namespace srukf_impl

// ---- step-wise API: the fast path ---------------------------------------------------------------------------------------------------------------
// The step-wise calls used to run launch sequences of their own (k_motion, k_project, k_meas_*, k_pxy, k_gain, a full k_syrk, the permutation pass, the persistent launch
// reading its tiles from memory, k_rank_expand, the rebuild of the permuted copy): ~2 x the staged replay's time per frame before the host round trips.  Where the replay's
// "fused tail" mode applies (replay_fuse_mode: rank-aware form with canonical null rows; BATCHED, NEEDNOT_REORDER) a step-wise frame now IS a frame of the staged replay, cut
// in two at the host's association step:
//   srukf_predict_motion       [k_set_step; unless the previous frame's tail projected this very odometry pair: k_sigr_rows + k_project_table;] k_pxy2 (motion reduction,
//                              measurement statistics, cross covariances);  the state before the frame is kept (ckS / ckX, copied on a stream of its own).  Nothing at
//                              all when the previous srukf_update submitted this frame ahead (announced odometry, below)
//   srukf_predict_measurement  waits on a pinned flag word: the statistics jobs inside k_pxy2 write h, Si, visible to the host themselves (MeasArgs::hmirror)
//   srukf_update               k_gain (z / matched read in place from pinned memory), the persistent factorisation launch, k_rank_expand<2> — which, when the host has
//                              announced the next frame's odometry (srukf_predict_motion_next), also projects the next frame, and whose last workgroup hands the frame
//                              scalars and the robot view to the host (StepExport) —; then, while the host waits for that: the NEXT frame's checkpoint copy, and with
//                              announced odometry its frame scalars and first launch ("step_early").  A flagged frame (theta clamp, abandoned launch, a null direction
//                              that is not) is rewound and repeated on the other path, as srukf_run_frames does; what was submitted ahead is then ignored
// Same kernels on the same values as the staged replay: bit-identical states (tests/test_gpu_parity_r5.py::test_step_api_equals_staged_replay).

namespace srukf_impl {

void step_invalidate(srukf_ctx* c) { step_ck_join(c); c->step_chain = false; c->proj_valid = false; c->robot_cached = false; c->view_cached = false; c->ck_valid = false; c->setstep_done = false; c->next_pose_pending = false; c->pre_issued = false; }

}  // namespace srukf_impl

static bool step_fast_eligible(const srukf_ctx* c) { return c->dbg.step_fast && !c->last_update_sequential && c->d.N > 0 && replay_fuse_mode(c); }

namespace srukf_impl {

// a state getter between predict and update (or a frame that ends without an update): the motion step's results go where k_gain / the state update would put them
void step_commit_motion(srukf_ctx* c)
{
    if (!c->step_uncommitted) return;
    step_ck_join(c);
    const RankArgs ra = rank_args(c);
    launch_commit_motion(c->stream, c->d.n, c->d.np, c->X, c->S, c->Cmat, c->fs, ra.A, ra.iperm, ra.r);
    c->step_uncommitted = false;
}

}  // namespace srukf_impl

// Wait for an export of the fast path: spin on the pinned flag word the export kernel writes behind its data (a completion signal through hipStreamSynchronize costs
// ~10 us more per round trip); after 2 ms on the steady clock without it — or with the switch off — the stream is synchronised the ordinary way (which also surfaces a
// faulted launch).  A spin that succeeds never asks the runtime anything, so every 256th of them queries the stream: a launch that faulted is then reported within 256
// frames of the one that caused it instead of at some later synchronising call.
namespace srukf_impl {
unsigned long long* step_flag(srukf_ctx* c) { return (unsigned long long*)((char*)c->hfs + sizeof(FrameScalars) + sizeof(double) * 32); }
int step_wait_export(srukf_ctx* c, unsigned long long seq)
{
    if (c->dbg.step_spin) {
        volatile unsigned long long* f = step_flag(c);
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        for (unsigned spins = 1;; spins++) {
            if (*f >= seq) {                                   // (>=: the next frame's pre-issued first launch may already have raised it further)
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                if ((++c->spin_ok & 255) == 0) { const hipError_t q = hipStreamQuery(c->stream); if (q != hipSuccess && q != hipErrorNotReady) HIPCHK(c, q); }
                return SRUKF_OK;
            }
            __builtin_ia32_pause();
            if ((spins & 255) == 0 && std::chrono::steady_clock::now() >= t_end) break;       // (the pause instruction is 40 - 140 cycles depending on the CPU: the bound is the clock's)
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SRUKF_OK;
}
}  // namespace srukf_impl
static int step_predict_fast(srukf_ctx* c, const double odo_prev[3], const double odo_cur[3])
{
    const KDims& d = c->d;
    const size_t np = d.np;
    if (!c->odo_step) HIPCHK(c, srukf_dmalloc(&c->odo_step, sizeof(double) * 16));
    if (!c->ckS) {
        if (srukf_dmalloc((void**)&c->ckS, sizeof(double) * np * np) != hipSuccess || srukf_dmalloc((void**)&c->ckX, sizeof(double) * np) != hipSuccess) { c->err = "predict_motion: out of device memory (checkpoint)"; return SRUKF_ERR_NOMEM; }
    }
    for (int e = 0; e < 3; e++) { c->step_odo[e] = odo_prev[e]; c->step_odo[3 + e] = odo_cur[e]; }
    const bool projected = c->step_chain && c->proj_valid && memcmp(c->proj_odo, c->step_odo, sizeof c->step_odo) == 0;
    // the next pose, if the host has announced it already (it may still do so before srukf_update)
    const bool hint = c->next_odo_valid && memcmp(c->next_odo, odo_cur, sizeof(double) * 3) == 0;
    double poses[9];
    for (int e = 0; e < 6; e++) poses[e] = c->step_odo[e];
    for (int e = 0; e < 3; e++) poses[6 + e] = hint ? c->next_odo[3 + e] : 0.0;
    c->step_seqF = hint ? 2 : 1;
    // the state before the frame (a flagged frame is repeated from it on the other path): copied on a stream of its own, beside the frame's first launch and the host's
    // association step — nothing writes S or X before k_gain, which waits for the copy (step_ck_join)
    if (!c->ck_stream) {
        if (hipStreamCreateWithFlags(&c->ck_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ck_e1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ck_e2, hipEventDisableTiming) != hipSuccess) { c->err = "predict_motion: no stream for the checkpoint copy"; return SRUKF_ERR_HIP; }
    }
    if (c->ck_valid && c->step_chain) {
        // the update that produced this state submitted the copy behind its last launch (ck_pending / ck_e2 are that copy's)
        c->ck_valid = false;
    } else {
        step_ck_join(c);
        HIPCHK(c, hipEventRecord(c->ck_e1, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->ck_stream, c->ck_e1, 0));
        HIPCHK(c, hipMemcpyAsync(c->ckS, c->S, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->ck_stream));
        HIPCHK(c, hipMemcpyAsync(c->ckX, c->X, sizeof(double) * np, hipMemcpyDeviceToDevice, c->ck_stream));
        HIPCHK(c, hipEventRecord(c->ck_e2, c->ck_stream));
        c->ck_pending = true;
    }
    const bool preset = projected && c->setstep_done && memcmp(c->setstep_odo, c->step_odo, sizeof c->step_odo) == 0;
    c->setstep_done = false;
    const bool pre = preset && c->pre_issued && memcmp(c->pre_odo, c->step_odo, sizeof c->step_odo) == 0;      // k_pxy2 of this very frame is in flight already
    c->pre_issued = false;
    if (!preset) launch_set_step(c->stream, c->fs, c->odo_step, c->step_seqF, c->p.a1, c->p.a2, c->p.a3, c->p.a4, c->step_chain ? 0 : 1, poses);
    c->fs_seq_step = true;
    if (!projected) {
        if (c->step_chain) launch_set_frame_control(c->stream, c->fs);      // (the tail prepared the control of ANOTHER pair, or none)
        srukf_launch_sigr_rows(c->stream, d, c->w, c->X, c->S, c->sigR, c->fs, c->red_iperm, c->red_r);
        seq_predict_fused(c, 2);
    }
    c->xr1_pending = true;
    // h | Si | visible reach the host's pinned buffer from the statistics jobs of this very launch (their final passes: the first ~10 us of it), flag behind them: the host
    // runs its association, and queues the update's launches, while the cross-covariance tiles are still being formed
    if (!pre) {
        c->meas_seq = c->dbg.step_fuse_export ? ++c->step_seq : 0;
        c->mirror_next = c->meas_seq != 0;
        seq_pxy(c, true, true, true, true, true);
        c->mirror_next = false;
    }
    c->next_pose_pending = preset && hint;                     // (k_set_step went out with two poses; the third rides on the update's k_gain launch: its successor needs it)
    c->step_fast = true; c->step_uncommitted = true;
    c->proj_valid = false;
    HIPCHK(c, hipGetLastError());
    c->phase = 1;
    return SRUKF_OK;
}

static int step_predict_slow(srukf_ctx* c, const double odo_prev[3], const double odo_cur[3])
{
    double* hs = c->hstage;
    for (int e = 0; e < 3; e++) { hs[e] = odo_prev[e]; hs[3 + e] = odo_cur[e]; }
    HIPCHK(c, hipMemcpyAsync(c->odocur, hs, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    seq_predict_motion(c, c->odocur);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->step_fast = false;
    c->phase = 1;
    return SRUKF_OK;
}

static int step_update_slow(srukf_ctx* c, const double* z, const int* matched, int reorder, int mode, int nm);

// the frame in flight leaves the fast path: the state before the frame comes back and the frame's predict half runs again on the other path
static int step_rewind_to_slow(srukf_ctx* c)
{
    const size_t np = c->d.np;
    step_ck_join(c);
    HIPCHK(c, hipMemcpyAsync(c->S, c->ckS, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->X, c->ckX, sizeof(double) * np, hipMemcpyDeviceToDevice, c->stream));
    quantize_state(c); shadow_rebuild(c);
    c->step_uncommitted = false; c->xr1_pending = false; c->dx_pending = false;
    step_invalidate(c);
    int rc = step_predict_slow(c, c->step_odo, c->step_odo + 3);
    if (rc) return rc;
    seq_predict_measurement(c, false);
    c->phase = 2;
    return SRUKF_OK;
}

static int step_update_fast(srukf_ctx* c, const double* z, const int* matched, int nm)
{
    const KDims& d = c->d;
    const int N = d.N;
    c->step_fast = false;
    if (nm == 0) {
        // KalmanUpdate returns at once (SLAM.cpp:2050-2051): the frame ends with its motion step, which the fast path still holds beside the state
        step_commit_motion(c);
        c->xr1_pending = false;
        step_invalidate(c);
        c->step_fast_frames++;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        return SRUKF_OK;
    }
    double* hs = c->hstage;
    const size_t mp = d.mp;
    // z | matched stay in the pinned host buffer: k_gain reads its 2N doubles and N ints from there (the buffer is device-accessible and is not touched again before this call's
    // synchronisation) — a host-to-device copy in front of it cost 12 us of stream time
    memcpy(hs, z, sizeof(double) * 2 * N);
    memcpy(hs + mp, matched, sizeof(int) * N);
    if (c->step_seqF == 1 && c->next_odo_valid && memcmp(c->next_odo, c->step_odo + 3, sizeof(double) * 3) == 0) {
        // the host announced the next frame's odometry after srukf_predict_motion: the tail of this frame can still project it
        c->next_pose_pending = true;                           // (rides on k_gain)
        c->step_seqF = 2;
    }
    step_ck_join(c);
    seq_gain_only(c, hs, (const int*)(hs + mp), true, true, true);
    c->step_uncommitted = false;                               // (k_gain and the state update commit the motion step)
    // The frame's status and the robot view (pose, 4 x 4 block of P: what the host records per frame, SLAM.cpp:3539-3556; srukf_get_robot then costs no round trip) reach the
    // pinned buffer from the frame's LAST launch: k_rank_expand<2>'s frame tail forms the block from the factor rows it walks anyway, the last workgroup through exports
    // (StepExport).  A host that fetched the display view (srukf_get_frame_view: the facade's refreshFeaturesDisplay, SLAM.cpp:2721-2751 per landmark) after its last update
    // gets it with this one: the landmark launch and one export launch behind the tail, which then raises the flag — no round trip and no copies of its own.
    const unsigned long long seq = ++c->step_seq;
    const bool view = c->dbg.step_fuse_export && c->view_auto && c->dbg.view_auto && c->hview && c->hview_doubles >= 12 * (size_t)N + d.n;
    if (c->dbg.step_fuse_export && !c->export_cnt) {
        if (srukf_dmalloc((void**)&c->export_cnt, sizeof(int) * 64 * 64) == hipSuccess) HIPCHK(c, hipMemsetAsync(c->export_cnt, 0, sizeof(int) * 64 * 64, c->stream));
        else { (void)hipGetLastError(); c->export_cnt = nullptr; }
    }
    bool early_set = false;
    if (c->dbg.step_fuse_export && c->export_cnt) {
        c->step_export = StepExport{ (unsigned long long*)c->hfs, (int)(sizeof(FrameScalars) / 8), c->small, view ? nullptr : step_flag(c), seq, c->export_cnt, 0, nullptr, { 0, 0, 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
        if (c->dbg.step_early && c->step_seqF == 2) {         // this frame's tail projects the pair (cur, next): the next frame's poses are known, its scalars start in the exporter
            StepExport& ex = c->step_export;
            ex.set = 1; ex.odo = c->odo_step;
            for (int e = 0; e < 3; e++) { ex.poses[e] = c->step_odo[3 + e]; ex.poses[3 + e] = c->next_odo[3 + e]; }
            ex.a[0] = c->p.a1; ex.a[1] = c->p.a2; ex.a[2] = c->p.a3; ex.a[3] = c->p.a4;
        }
    }
    c->step_export_attached = false;
    seq_refactor(c, 0, d.mp, false, false, false, true, true, true);
    early_set = c->step_export_attached && c->step_export.set;
    c->step_export = StepExport{};
    if (!c->step_export_attached) {                            // (the form with launches of their own: "step_fuse_export" 0, or a tail that is not k_rank_expand<2>)
        srukf_launch_block_cov(c->stream, d, c->S, d.n - 4, 4, c->small, c->X);
        launch_export(c->stream, c->fs, sizeof(FrameScalars), c->small, sizeof(double) * 20, c->hfs, (c->dbg.step_spin && !view) ? step_flag(c) : nullptr, seq);
    }
    if (view) {
        srukf_launch_landmarks_cartesian(c->stream, d, c->X, c->S, c->G, c->G + 3 * (size_t)N);
        launch_export(c->stream, c->G, sizeof(double) * 12 * (size_t)N, c->X, sizeof(double) * d.n, c->hview, step_flag(c), seq);
    }
    // While the host waits anyway: what the NEXT srukf_predict_motion would have to submit in front of its first launch.  The copy of the state this frame leaves (the
    // checkpoint of the next frame; second pair of buffers: this frame's own checkpoint is still needed if it turns out flagged), and, when the host has announced the next
    // frame's odometry, that frame's k_set_step.
    bool early_ck = false, early_pxy = false;
    if (c->dbg.step_early && c->ck_stream) {
        const size_t np = d.np;
        if (!c->ckS2 && (srukf_dmalloc((void**)&c->ckS2, sizeof(double) * np * np) != hipSuccess || srukf_dmalloc((void**)&c->ckX2, sizeof(double) * np) != hipSuccess || hipEventCreateWithFlags(&c->ck_e3, hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError();
            if (c->ckS2) { srukf_dfree_on(c->ckS2, c->stream); c->ckS2 = nullptr; }
        }
        if (c->ckS2 && c->ckX2 && c->ck_e3) {
            HIPCHK(c, hipEventRecord(c->ck_e1, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->ck_stream, c->ck_e1, 0));
            HIPCHK(c, hipMemcpyAsync(c->ckS2, c->S, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->ck_stream));
            HIPCHK(c, hipMemcpyAsync(c->ckX2, c->X, sizeof(double) * np, hipMemcpyDeviceToDevice, c->ck_stream));
            HIPCHK(c, hipEventRecord(c->ck_e3, c->ck_stream));
            early_ck = true; c->ck3_inflight = true;
        }
        if (c->step_seqF == 2) {                                // this frame's tail projects the pair (cur, next): the next frame's poses are known
            if (!early_set) {                                  // (the tail did not export: a launch of its own)
                double poses[9];
                for (int e = 0; e < 3; e++) { poses[e] = c->step_odo[3 + e]; poses[3 + e] = c->next_odo[3 + e]; poses[6 + e] = 0.0; }
                launch_set_step(c->stream, c->fs, c->odo_step, 1, c->p.a1, c->p.a2, c->p.a3, c->p.a4, 0, poses);
                early_set = true;
            }
            if (c->dbg.step_early >= 2 && early_ck) {
                // ... and that frame's first launch: it reads what this frame's tail leaves (the table, the projected sigma points) and writes per-frame scratch only —
                // nothing of the state — so a host that then does something else (another pair, a new state, a map change) just has it ignored and repeated
                c->meas_seq = c->dbg.step_fuse_export ? ++c->step_seq : 0;
                c->mirror_next = c->meas_seq != 0;
                seq_pxy(c, true, true, true, true, true);
                c->mirror_next = false;
                early_pxy = true;
            }
        }
    }
    int rc = step_wait_export(c, seq); if (rc) return rc;
    rc = read_fs_host(c); if (rc) return rc;
    if (c->hfs->clamp_rows > 0) {
        // flagged (the reference's theta clamp would have been active, a skipped direction was not null, a persistent launch was abandoned): the frame is repeated
        // from the state before it on the path that evaluates the clamp pivot by pivot
        rc = step_rewind_to_slow(c); if (rc) return rc;
        c->phase = 0;
        return step_update_slow(c, z, matched, SRUKF_NEEDNOT_REORDER, SRUKF_UPDATE_BATCHED, nm);
    }
    set_null_canonical(c);
    c->step_chain = true;
    c->proj_valid = c->step_seqF == 2 && c->hfs->ctl_next_valid != 0;
    if (c->proj_valid) { for (int e = 0; e < 3; e++) { c->proj_odo[e] = c->step_odo[3 + e]; c->proj_odo[3 + e] = c->next_odo[3 + e]; } }
    if (early_ck) {                                            // the frame is clean: its end state's copy becomes the next frame's checkpoint
        std::swap(c->ckS, c->ckS2); std::swap(c->ckX, c->ckX2); std::swap(c->ck_e2, c->ck_e3);
        c->ck_pending = true; c->ck_valid = true; c->ck3_inflight = false;
    }
    c->setstep_done = early_set && c->proj_valid;
    if (c->setstep_done) memcpy(c->setstep_odo, c->proj_odo, sizeof c->setstep_odo);
    c->pre_issued = early_pxy && c->setstep_done;
    if (c->pre_issued) memcpy(c->pre_odo, c->proj_odo, sizeof c->pre_odo);
    c->next_odo_valid = false;
    c->robot_cached = true;
    c->view_cached = view;
    if (view && ++c->view_unused >= 3) c->view_auto = false;   // (nobody reads them)
    c->f32_stale = storage_f32_like(c);
    c->step_fast_frames++;
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

static int step_update_slow(srukf_ctx* c, const double* z, const int* matched, int reorder, int mode, int nm)
{
    const KDims& d = c->d;
    const int N = d.N;
    step_invalidate(c);
    c->step_slow_frames++;
    if (nm == 0) return SRUKF_OK;                                        // SLAM.cpp:2050-2051
    double* hs = c->hstage;
    memcpy(hs, z, sizeof(double) * 2 * N);
    int* hm = (int*)(hs + 2 * N);
    memcpy(hm, matched, sizeof(int) * N);
    HIPCHK(c, hipMemcpyAsync(c->zcur, hs, sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->mcur, hm, sizeof(int) * N, hipMemcpyHostToDevice, c->stream));
    launch_set_frame(c->stream, c->fs, 0, 1);
    bool exact_ran = false;
    seq_gain(c, c->zcur, c->mcur, false);
    // visibility is needed on the host only to skip no-op refactors in SEQUENTIAL mode
    if (reorder == SRUKF_NEED_REORDER) {
        if (mode == SRUKF_UPDATE_BATCHED) { int rc = refactor_reorder(c, 0, d.mp); if (rc) return rc; }
        else {
            for (int k = 0; k < N; k++) {
                if (!matched[k]) continue;                               // SLAM.cpp:2068
                for (int col = 0; col < 2; col++) { int rc = refactor_reorder(c, 2 * k + col, 2 * k + col + 1); if (rc) return rc; }
            }
        }
    } else if (mode == SRUKF_UPDATE_BATCHED) {
        seq_refactor(c, 0, d.mp, false, true, false, false);
        int rc = read_fs(c); if (rc) return rc;
        if (c->hfs->clamp_rows > 0) {
            // the reference's theta clamp would have been active: redo this refactor on the exact path
            exact_ran = true;
            launch_set_frame(c->stream, c->fs, 0, 1);
            launch_refactor_reset(c->stream, d.np, c->theta, c->fs, 0);
            HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
            ProfScope ps(c, KC_GMW_COL, 0, 0);
            exact_path(c, c->G, c->S);
            quantize_state(c);
        }
    } else {
        std::vector<int> visible(N);
        HIPCHK(c, hipMemcpyAsync(hs + 4 * N, c->vis, sizeof(int) * N, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(visible.data(), hs + 4 * N, sizeof(int) * N);
        for (int k = 0; k < N; k++) {
            if (!matched[k]) continue;                                   // SLAM.cpp:2068
            for (int col = 0; col < 2; col++) {                          // SLAM.cpp:2116
                const int m = 2 * k + col;
                seq_refactor(c, m, m + 1, false, true, true, false);
                int rc = read_fs(c); if (rc) return rc;
                if (c->hfs->clamp_rows > 0) {
                    exact_ran = true;
                    launch_set_frame(c->stream, c->fs, 0, 1);
                    launch_refactor_reset(c->stream, d.np, c->theta, c->fs, 0);
                    HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
                    exact_path(c, c->G, c->S);
                    quantize_state(c);
                    launch_set_frame(c->stream, c->fs, 0, 1);
                }
            }
        }
    }
    // rank-aware form: the reorder path and the exact column path write S without the permuted copy, and their factor may have
    // other null rows (a frame that went to the exact path because a skipped direction was found not to be null must not meet the
    // same null set again)
    if (reorder == SRUKF_NEED_REORDER || exact_ran) { const int rc = update_null_set(c); if (rc) return rc; canonicalize_null_rows(c); }
    else {
        shadow_rebuild(c);
        if (c->storage != SRUKF_STORAGE_F32_MIXED || storage_f32_like(c)) set_null_canonical(c);     // the rank-aware tail (k_rank_expand) has written sqrt(EPSILON) e_k into every skipped row
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

extern "C" {

int srukf_predict_motion(srukf_ctx* c, const double odo_prev[3], const double odo_cur[3])
{
    if (!c || !odo_prev || !odo_cur) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->step_uncommitted) { step_commit_motion(c); step_invalidate(c); }      // a frame that was predicted and never updated: its motion step stands (as on the other path)
    c->step_fast = false; c->robot_cached = false; c->view_cached = false;
    if (step_fast_eligible(c)) return step_predict_fast(c, odo_prev, odo_cur);
    step_invalidate(c);
    return step_predict_slow(c, odo_prev, odo_cur);
}

int srukf_predict_motion_next(srukf_ctx* c, const double odo_prev[3], const double odo_cur[3])
{
    if (!c || !odo_prev || !odo_cur) return SRUKF_ERR_BAD_ARG;
    for (int e = 0; e < 3; e++) { c->next_odo[e] = odo_prev[e]; c->next_odo[3 + e] = odo_cur[e]; }
    c->next_odo_valid = true;
    return SRUKF_OK;
}

int srukf_predict_measurement(srukf_ctx* c, double* h, double* Si, int* visible)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    if (c->phase < 1) { c->err = "predict_measurement before predict_motion"; return SRUKF_ERR_SEQUENCE; }
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->d.N;
    if (N == 0) { c->phase = 2; return SRUKF_OK; }                       // empty map: nothing to predict
    if (!c->step_fast) seq_predict_measurement(c, false);                // (fast path: the statistics rode on srukf_predict_motion's k_pxy2 launch: this call is a copy)
    double* hs = c->hstage;
    const size_t mp = c->d.mp;                                           // (h | Si | visible are one device allocation: one transfer)
    const size_t out_bytes = sizeof(double) * (mp + 4 * (size_t)N) + sizeof(int) * N;
    if (c->step_fast && c->meas_seq) {                                   // the statistics jobs of k_pxy2 have written (or are writing) the pinned buffer themselves
        const int rcw = step_wait_export(c, c->meas_seq); if (rcw) return rcw;
        hs = c->hmeas;
    } else if (c->step_fast) {                                           // a kernel writes the pinned buffer: no blit, no gap behind it
        const unsigned long long seq = ++c->step_seq;
        launch_export(c->stream, c->h, out_bytes, nullptr, 0, hs, c->dbg.step_spin ? step_flag(c) : nullptr, seq);
        const int rcw = step_wait_export(c, seq); if (rcw) return rcw;
    } else {
        HIPCHK(c, hipMemcpyAsync(hs, c->h, out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipGetLastError());
    if (h) memcpy(h, hs, sizeof(double) * 2 * N);
    if (Si) memcpy(Si, hs + mp, sizeof(double) * 4 * N);
    if (visible) memcpy(visible, hs + mp + 4 * (size_t)N, sizeof(int) * N);
    c->phase = 2;
    return SRUKF_OK;
}

int srukf_update(srukf_ctx* c, const double* z, const int* matched, int reorder, int mode)
{
    if (!c || !z || !matched) return SRUKF_ERR_BAD_ARG;
    if (c->phase < 2) { c->err = "update before predict_measurement"; return SRUKF_ERR_SEQUENCE; }
    if (reorder != SRUKF_NEEDNOT_REORDER && reorder != SRUKF_NEED_REORDER) return SRUKF_ERR_BAD_ARG;
    if (reorder == SRUKF_NEED_REORDER && c->K_new <= 0) { c->err = "NEED_REORDER without srukf_set_new_landmarks (m_nFilters = 0)"; return SRUKF_ERR_SEQUENCE; }
    if (mode != SRUKF_UPDATE_SEQUENTIAL && mode != SRUKF_UPDATE_BATCHED) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->d.N;
    int nm = 0; for (int k = 0; k < N; k++) nm += matched[k] ? 1 : 0;
    c->last_update_sequential = mode == SRUKF_UPDATE_SEQUENTIAL;
    if (c->step_fast) {
        if (reorder == SRUKF_NEEDNOT_REORDER && mode == SRUKF_UPDATE_BATCHED) { c->phase = 0; return step_update_fast(c, z, matched, nm); }
        const int rc = step_rewind_to_slow(c); if (rc) return rc;       // predicted on the fast path, updated in a mode it does not have
        c->step_fast = false;
    }
    c->phase = 0;
    return step_update_slow(c, z, matched, reorder, mode, nm);
}

}  // extern "C"
