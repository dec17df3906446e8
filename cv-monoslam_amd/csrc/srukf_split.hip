// srukf_split.hip — plumbing of the split form of the persistent factorisation (k_gmw_pivslab_persist + k_gmw_tiles_persist, srukf_gmw_persist.hip): when it applies,
// its slab buffers, the side stream its second launch runs on (probed: the two launches wait for each other and must sit on different hardware queues), the hand-over
// of that stream across context rebuilds, and the measurement hook that replays one launch of the pair alone.

#include "srukf_ctx.h"
using namespace srukf_impl;

// ... and one verified side stream of the split form (with its events): a map change rebuilds the context on the SAME filter stream, and probing candidates again
// (up to eight streams, two launches and three synchronisations each) would sit on the latency-critical path of every srukf_add_landmarks / srukf_delete_landmark
struct SpareSide { int device = -1; hipStream_t main = nullptr, side = nullptr; hipEvent_t fork = nullptr, join = nullptr; };

static thread_local SpareSide g_spare_side;

static void spare_side_drop()
{
    if (g_spare_side.side) { hipStreamSynchronize(g_spare_side.side); hipStreamDestroy(g_spare_side.side); hipEventDestroy(g_spare_side.fork); hipEventDestroy(g_spare_side.join); }
    g_spare_side = SpareSide();
}

static bool split_wanted(const GmwPlan& gp)
{
    if (!srukf_gmw_register_form(gp.T, gp.Tp, gp.ntiles, gp.workers)) return true;
    return g_dbg_mem_split == 2 && gp.nreal > gp.workers;
}

// Do kernels on streams a and b run side by side?  HIP maps streams onto a handful of hardware queues (four by default: GPU_MAX_HW_QUEUES) and two streams that share
// one run their kernels one after the other — the split form's two launches wait for each other, so it must never be given such a pair (measured: with several
// filters in a process the SECOND one's stream pair shared a queue; its first pair of launches sat out the 50 ms wait bound and the filter fell back to per-panel
// launches).  Probe: a one-wave kernel on a that waits (bounded, ~2 ms) for a word a kernel on b sets.
__global__ void k_stream_probe_wait(int* w)
{
    int seen = 0;
    for (int spins = 0; spins < (1 << 15) && !seen; spins++) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (!seen) __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) w[1] = seen;
}

__global__ void k_stream_probe_set(int* w) { __hip_atomic_store(&w[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static bool streams_run_side_by_side(hipStream_t a, hipStream_t b)
{
    int* w = nullptr;
    if (srukf_dmalloc(&w, 2 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return false; }
    int seen = 0;
    bool ok = hipMemsetAsync(w, 0, 2 * sizeof(int), a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_stream_probe_wait, dim3(1), dim3(64), 0, a, w);
        hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(1), 0, b, w);
        ok = hipStreamSynchronize(b) == hipSuccess && hipStreamSynchronize(a) == hipSuccess && hipMemcpy(&seen, w + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    srukf_dfree(w);
    if (!ok) (void)hipGetLastError();
    return ok && seen != 0;
}

namespace srukf_impl {

// Buffers / side stream of the split form for a plan with Tp pivoted panels (not inside a capture).  Failure is not an error: the memory-tile form is used.
void split_ensure(srukf_ctx* c, const GmwPlan& gp)
{
    if (!g_dbg_mem_split || gp.T < 16 || !split_wanted(gp)) return;           // (gp.workers < 0 — more tiles than the memory-tile form can own — included: the split form has no such limit)
    if (gp.T + 1 > gp.cus) return;                              // the pivot / slab launch must be resident as a whole with CUs left for the tiles
    if (c->gs_panels < gp.Tp) {
        if (c->gsW) srukf_dfree_on(c->gsW, c->stream);
        if (c->gsL) srukf_dfree_on(c->gsL, c->stream);
        c->gsW = c->gsL = nullptr; c->gs_panels = 0;
        const size_t bytes = sizeof(double) * 64 * (size_t)c->d.np * gp.Tp;
        if (srukf_dmalloc(&c->gsW, bytes) != hipSuccess || srukf_dmalloc(&c->gsL, bytes) != hipSuccess || hipMemset(c->gsW, 0, bytes) != hipSuccess || hipMemset(c->gsL, 0, bytes) != hipSuccess) {
            if (c->gsW) srukf_dfree_on(c->gsW, c->stream);
            if (c->gsL) srukf_dfree_on(c->gsL, c->stream);
            c->gsW = c->gsL = nullptr; (void)hipGetLastError();
            return;
        }
        c->gs_panels = gp.Tp;
    }
    if (!c->side && g_spare_side.side && g_spare_side.device == c->device && g_spare_side.main == c->stream) {
        c->side = g_spare_side.side; c->ev_fork = g_spare_side.fork; c->ev_join = g_spare_side.join;      // probed against this very stream by the context that was just rebuilt
        g_spare_side = SpareSide();
    }
    if (!c->side) {
        // a side stream whose kernels really run beside the filter's stream's: candidates are created until one passes the probe (they are kept alive until then,
        // so that the runtime hands out another hardware queue), the others are destroyed; none in eight tries: no split form for this filter
        hipStream_t tried[8]; int ntried = 0;
        while (!c->side && ntried < 8) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            if (streams_run_side_by_side(c->stream, s)) c->side = s; else tried[ntried++] = s;
        }
        for (int q = 0; q < ntried; q++) hipStreamDestroy(tried[q]);
        if (!c->side) return;
        if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            hipStreamDestroy(c->side); c->side = nullptr; (void)hipGetLastError();
        }
    }
}

bool split_form(const srukf_ctx* c, const GmwPlan& gp, bool ignore_starve)
{
    return g_dbg_mem_split && !c->split_off && c->gmw_shared == 0 && (ignore_starve || !c->debug_starve) && c->side && c->gsW && c->gs_panels >= gp.Tp && gp.T >= 16 && gp.T + 1 <= gp.cus && split_wanted(gp);
}

// a context about to be rebuilt (map change) offers its verified side stream to the context srukf_create builds next on the same filter stream (split_ensure)
void side_stream_lend(srukf_ctx* c)
{
    if (!c->side) return;
    hipStreamSynchronize(c->side);
    spare_side_drop();
    g_spare_side.device = c->device; g_spare_side.main = c->stream; g_spare_side.side = c->side; g_spare_side.fork = c->ev_fork; g_spare_side.join = c->ev_join;
    c->side = nullptr; c->ev_fork = c->ev_join = nullptr;
}

}  // namespace srukf_impl

extern "C" {

// Measurement only (scripts/split_replay.py).  The two launches of the split form wait for each other, and rocprofv3's counter passes serialise dispatches: the pair cannot
// run under them.  Everything the launches exchange lives in HBM — G tiles and their version flags, the slabs of every panel and theirs, the panel buffers and flags — so
// ONE launch of the pair can be replayed ALONE against the buffers a real frame left behind (uploaded with srukf_debug_upload: "Gbak" = the matrix before the factorisation,
// "Wf" = its tiles after it, "gsW" / "gsL", "pans", "sync"): every wait finds its flag at its final value, every load the value the real run delivered, and the launch
// executes the instructions and moves the bytes of the real one.  which = 0: k_gmw_pivslab_persist, 1: k_gmw_tiles_persist (its tiles restored from "Gbak" first); `reps` launches.
int srukf_debug_split_replay(srukf_ctx* c, int which, int reps)
{
    if (!c || which < 0 || which > 1 || reps < 1) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const bool reduced = c->red_r > 0;
    const GmwPlan& gp = reduced ? c->gplan_red : c->gplan;
    // (not split_form(): under a profiler that serialises dispatches the side-stream probe fails, and the replay needs no side stream — only the slab buffers)
    if (!c->gsW || !c->gsL || c->gs_panels < gp.Tp || gp.T < 16) { c->err = "split_replay: this context has no slab buffers (its size does not factor with the split form)"; return SRUKF_ERR_SEQUENCE; }
    const int np = c->d.np, n = c->d.n, Tp = reduced ? c->red_Tp : np / 64;
    unsigned long long epoch = 0;
    HIPCHK(c, hipMemcpy(&epoch, (char*)gp.sync + offsetof(GmwSync, epoch), sizeof epoch, hipMemcpyDeviceToHost));
    if (epoch < 2) { c->err = "split_replay: the sync block holds no finished run"; return SRUKF_ERR_SEQUENCE; }
    const unsigned long long prev = epoch - 1;                   // the run whose flags the block holds
    for (int r = 0; r < reps; r++) {
        HIPCHK(c, hipMemcpyAsync((char*)gp.sync + offsetof(GmwSync, epoch), &prev, sizeof prev, hipMemcpyHostToDevice, c->stream));
        if (which == 1) HIPCHK(c, hipMemcpyAsync(c->Wf, c->Gbak, sizeof(double) * (size_t)np * np, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));              // (&prev is pageable)
        srukf_launch_gmw_split_alone(c->stream, which, n, np, c->p.epsilon, c->Wf, gp.pans, c->D, c->G, gp.sync, gp.tiles, gp.ntiles, c->fs, Tp, reduced ? ((c->red_r + 15) & ~15) : 0, c->gsW, c->gsL);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipGetLastError());
    return read_fs(c);
}

}  // extern "C"
