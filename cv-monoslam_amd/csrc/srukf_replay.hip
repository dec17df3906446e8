// srukf_replay.hip — the launch sequences of a frame (seq_*: which kernels run, in which order, for which launch plan), the rank-aware null set, the graph cache
// and the staged replay (srukf_stage_sequence / srukf_run_frames*: the OnBnClickedAuto loop of MonoSLAMView.cpp:526-572 with inputs resident in HBM).
// Also the small host-layer kernels (frame scalars, permutations, rounding) behind their launchers.

#include "srukf_ctx.h"
using namespace srukf_impl;

// resets the per-refactor accumulators (theta row maxima, gamma/xi)
__global__ void k_refactor_reset(int np, unsigned long long* theta_bits, FrameScalars* fs, int reset_stats)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) theta_bits[i] = 0ull;
    if (i == 0 && reset_stats) { fs->gmax_bits = 0ull; fs->ximax_bits = 0ull; }
}

// staged sequence: what srukf_prepare_control needs to prepare the control of a frame on the device
__global__ void k_set_seq(FrameScalars* fs, const double* odo_seq, int seqF, double a1, double a2, double a3, double a4)
{
    fs->odo_seq = odo_seq; fs->seqF = seqF;
    fs->a[0] = a1; fs->a[1] = a2; fs->a[2] = a3; fs->a[3] = a4;
}

__global__ void k_set_frame(FrameScalars* fs, int frame, int clear_clamp)
{
    fs->frame = frame;
    srukf_prepare_control(fs);
    fs->stat_count = 0;
    fs->const_rows_ok = 0; fs->const_rows_pending = 0;         // whatever happened to S since the last staged frame: its first tail writes every row again
    for (int q = 0; q < SRUKF_STAT_GROUPS; q++) fs->stat_cnt[q] = 0;
    fs->traj_base = nullptr;
    if (clear_clamp) { fs->clamp_rows = 0; fs->clamp_first = 0x7fffffff; fs->clamp_frame = 0x7fffffff; fs->frozen = 0; fs->gmw_aborts = 0; }
}

__global__ void k_set_traj(FrameScalars* fs, double* traj_base) { fs->traj_base = traj_base; }

// start of a staged replay: frame counter, flags and trajectory base in one launch
__global__ void k_set_run(FrameScalars* fs, int frame, int clear_clamp, double* traj_base)
{
    fs->frame = frame;
    srukf_prepare_control(fs);                                 // the first frame's control (k_project_motion); later ones by the frame tails
    fs->stat_count = 0;
    fs->const_rows_ok = 0; fs->const_rows_pending = 0;         // whatever happened to S since the last staged frame: its first tail writes every row again
    for (int q = 0; q < SRUKF_STAT_GROUPS; q++) fs->stat_cnt[q] = 0;
    fs->traj_base = traj_base;
    if (clear_clamp) { fs->clamp_rows = 0; fs->clamp_first = 0x7fffffff; fs->clamp_frame = 0x7fffffff; fs->frozen = 0; fs->gmw_aborts = 0; }
}

// Step-wise API, fast path: start of a frame.  odo = (prev, cur[, next]) poses on the device: a staged sequence of one (two) frames the frame scalars point at.
// fresh: nothing prepared this frame (the control, the flags of the constant rows); otherwise the previous frame's tail prepared fs->ctl and projected the frame.
// (the poses arrive as kernel arguments and are written to the device buffer here: a 72-byte host-to-device copy in front of the frame's first launch cost 12 us of stream
//  time — 4.5 for the blit kernel, 7.5 of gap behind it; scripts/profile_step.sh)
struct StepPoses { double v[9]; };
__global__ void k_set_step(FrameScalars* fs, double* odo, int seqF, double a1, double a2, double a3, double a4, int fresh, StepPoses po)
{
    for (int e = 0; e < 9; e++) odo[e] = po.v[e];
    const double a[4] = { a1, a2, a3, a4 };
    srukf_step_scalars(fs, odo, seqF, a);
    if (fresh) { srukf_prepare_control(fs); fs->const_rows_ok = 0; fs->const_rows_pending = 0; }
}

__global__ void k_set_frame_control(FrameScalars* fs) { srukf_prepare_control(fs); }
// Small results for the host (h | Si | visible after the predict half; frame scalars + robot view after the update half) written straight into its pinned buffer by a kernel:
// a hipMemcpyAsync of a few KB is a blit kernel plus ~7.5 us of gap behind it on the stream (scripts/profile_step.sh); this is one short launch.  Two segments, 8-byte words.
// flag (may be null): a word of the same pinned buffer that receives `seq` AFTER the data — every wave's stores made visible at system scope first — so that the host can
// spin on it instead of going through hipStreamSynchronize (one workgroup then: the flag needs a barrier over all stores).
__global__ __launch_bounds__(256) void k_export(const unsigned long long* __restrict__ a, int na, const unsigned long long* __restrict__ b, int nb, unsigned long long* __restrict__ dst,
                                                unsigned long long* flag, unsigned long long seq)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nb; i += gridDim.x * 256) dst[i] = i < na ? a[i] : b[i - na];
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ... and the commit of its motion step on demand (a state getter or srukf_associate between predict and update; a frame without a match): what k_gain does with Cmat /
// the state update with fs->Xr1 — the new last four columns of S (and of the permuted copy), the new robot mean.  Idempotent: k_gain / the update write the same values again.
__global__ __launch_bounds__(256) void k_commit_motion(int n, int ld, double* __restrict__ X, double* __restrict__ S, const double* __restrict__ Cm, const FrameScalars* __restrict__ fs,
                                                       double* __restrict__ A, const int* __restrict__ iperm, int rk)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const double4 v = *reinterpret_cast<const double4*>(Cm + (size_t)r * 4);
    *reinterpret_cast<double2*>(S + (size_t)r * ld + (n - 4)) = make_double2(v.x, v.y);
    *reinterpret_cast<double2*>(S + (size_t)r * ld + (n - 2)) = make_double2(v.z, v.w);
    if (A) {
        const int arow = (r < n - 4) ? iperm[r] : rk - 4 + (r - (n - 4));
        if (arow < rk) { double* o = A + (size_t)arow * ld + (rk - 4); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    }
    if (r >= n - 4) X[r] = fs->Xr1[r - (n - 4)];
}

// dst = P^T src P on the upper triangle of a symmetric matrix stored upper: dst[a][b] = src[ip[a]][ip[b]] (a <= b),
// ip[a] = the row r of the source with perm[r] = a.  forward = 0 swaps the roles (dst[r][c] = src[perm[r]][perm[c]]).
// Entries outside the upper n x n part are zeroed.
// The same kernel compacts a covariance when landmarks leave the state (map skips their indices, lds > ld).
__global__ __launch_bounds__(256) void k_sym_permute(int n, int ld, const double* __restrict__ src, int lds, double* __restrict__ dst,
                                                     const int* __restrict__ map)
{
    const int a = blockIdx.x;
    for (int b = threadIdx.x; b < ld; b += 256) {
        double v = 0.0;
        if (a < n && b < n && b >= a) {
            const int r = map[a], c = map[b];
            v = (r <= c) ? src[(size_t)r * lds + c] : src[(size_t)c * lds + r];
        }
        dst[(size_t)a * ld + b] = v;
    }
}

__global__ __launch_bounds__(256) void k_gather(int n, int ld, const double* __restrict__ src, double* __restrict__ dst, const int* __restrict__ map)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < ld) dst[a] = (a < n) ? src[map[a]] : 0.0;
}

// "bottom rows zero" of CholeskyDecompositionWithPivoting (SLAM.cpp:2161, 2176): rows >= rank of the disordered factor
__global__ __launch_bounds__(256) void k_zero_rows(int ld, int r0, double* __restrict__ A)
{
    const int r = r0 + blockIdx.x;
    for (int b = threadIdx.x; b < ld; b += 256) A[(size_t)r * ld + b] = 0.0;
}

// SRUKF_STORAGE_F32 (BASELINE configs[4]: fp32 filter state, fp64 arithmetic): the state that lives from frame to frame
// is X32 / S32; the fp64 working copies are rounded to the stored values at the end of every refactorisation, so the
// next frame computes from exactly what fp32 storage holds.  One workgroup per row of S (+ one for X).
__global__ __launch_bounds__(256) void k_quantize(int n, int ld, double* __restrict__ S, double* __restrict__ X,
                                                  float* __restrict__ S32, float* __restrict__ X32)
{
    const int r = blockIdx.x;
    if (r == n) {
        for (int c = threadIdx.x; c < n; c += 256) { const float f = (float)X[c]; X32[c] = f; X[c] = (double)f; }
        return;
    }
    for (int c = r + threadIdx.x; c < n; c += 256) {
        const float f = (float)S[(size_t)r * ld + c];
        S32[(size_t)r * ld + c] = f;
        S[(size_t)r * ld + c] = (double)f;
    }
}

namespace srukf_impl {
void launch_refactor_reset(hipStream_t st, int np, unsigned long long* theta_bits, FrameScalars* fs, int reset_stats) { hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, st, np, theta_bits, fs, reset_stats); }
void launch_set_seq(hipStream_t st, FrameScalars* fs, const double* odo_seq, int seqF, double a1, double a2, double a3, double a4) { hipLaunchKernelGGL(k_set_seq, dim3(1), dim3(1), 0, st, fs, odo_seq, seqF, a1, a2, a3, a4); }
void launch_set_frame(hipStream_t st, FrameScalars* fs, int frame, int clear_clamp) { hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, st, fs, frame, clear_clamp); }
void launch_set_traj(hipStream_t st, FrameScalars* fs, double* traj_base) { hipLaunchKernelGGL(k_set_traj, dim3(1), dim3(1), 0, st, fs, traj_base); }
void launch_set_run(hipStream_t st, FrameScalars* fs, int frame, int clear_clamp, double* traj_base) { hipLaunchKernelGGL(k_set_run, dim3(1), dim3(1), 0, st, fs, frame, clear_clamp, traj_base); }
void launch_set_step(hipStream_t st, FrameScalars* fs, double* odo, int seqF, double a1, double a2, double a3, double a4, int fresh, const double poses[9])
{
    StepPoses po; for (int e = 0; e < 9; e++) po.v[e] = poses[e];
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, st, fs, odo, seqF, a1, a2, a3, a4, fresh, po);
}
void launch_export(hipStream_t st, const void* a, size_t bytes_a, const void* b, size_t bytes_b, void* host_pinned, unsigned long long* flag, unsigned long long seq)
{
    const int na = (int)((bytes_a + 7) / 8), nb = (int)((bytes_b + 7) / 8);
    const int blocks = flag ? 1 : ((na + nb + 255) / 256 > 8 ? 8 : (na + nb + 255) / 256);
    hipLaunchKernelGGL(k_export, dim3(blocks), dim3(256), 0, st, (const unsigned long long*)a, na, (const unsigned long long*)b, nb, (unsigned long long*)host_pinned, flag, seq);
}
void launch_set_frame_control(hipStream_t st, FrameScalars* fs) { hipLaunchKernelGGL(k_set_frame_control, dim3(1), dim3(1), 0, st, fs); }
void launch_commit_motion(hipStream_t st, int n, int ld, double* X, double* S, const double* Cm, const FrameScalars* fs, double* A, const int* iperm, int rk) { hipLaunchKernelGGL(k_commit_motion, dim3((n + 255) / 256), dim3(256), 0, st, n, ld, X, S, Cm, fs, A, iperm, rk); }
void launch_sym_permute(hipStream_t st, int n, int ld, const double* src, int lds, double* dst, const int* map) { hipLaunchKernelGGL(k_sym_permute, dim3(ld), dim3(256), 0, st, n, ld, src, lds, dst, map); }
void launch_gather(hipStream_t st, int n, int ld, const double* src, double* dst, const int* map) { hipLaunchKernelGGL(k_gather, dim3((ld + 255) / 256), dim3(256), 0, st, n, ld, src, dst, map); }
void launch_zero_rows(hipStream_t st, int ld, int r0, double* A) { if (ld > r0) hipLaunchKernelGGL(k_zero_rows, dim3(ld - r0), dim3(256), 0, st, ld, r0, A); }
void launch_quantize(hipStream_t st, int n, int ld, double* S, double* X, float* S32, float* X32) { hipLaunchKernelGGL(k_quantize, dim3(n + 1), dim3(256), 0, st, n, ld, S, X, S32, X32); }
}  // namespace srukf_impl

namespace srukf_impl {

int gmw_persist_mode() { return g_dbg_gmw_persist; }

void quantize_state(srukf_ctx* c)
{
    if (c->storage != SRUKF_STORAGE_F64)
        hipLaunchKernelGGL(k_quantize, dim3(c->d.n + 1), dim3(256), 0, c->stream, c->d.n, c->d.np, c->S, c->X, c->S32, c->X32);
}

// rank-aware replay form: what k_motion / k_gain / k_syrk carry along (all null when the shadow copy does not exist)
// the pending state update as the launch that applies it takes it: k_gain's slice partials, or the per-landmark shares the gain fold of k_pxy2 left (RankArgs::dxN)
static const double* dx_src(const srukf_ctx* c) { return c->dx_pending ? (c->dx_lm ? c->dxk : c->dxp) : nullptr; }

RankArgs rank_args(const srukf_ctx* c, bool prep_next, bool dzperm, bool f32round)
{
    RankArgs ra = {};
    ra.prep_next = prep_next ? 1 : 0; ra.dzperm = dzperm ? 1 : 0; ra.f32round = f32round ? 1 : 0;
    ra.dxN = (c->dx_pending && c->dx_lm) ? c->d.N : 0;
    if (c->red_r > 0 && c->shadowA) { ra.A = c->shadowA; ra.Utp = c->Utp; ra.gdiag = c->gdiag; ra.iperm = c->red_iperm; ra.perm = c->red_perm; ra.r = c->red_r; }
    return ra;
}

// NullSkip of the "table" mode (all null: every direction is projected and read in full)
NullSkip null_skip(const srukf_ctx* c)
{
    NullSkip ns = {};
    if (c->dbg.nullskip && c->dbg.pxy2 && c->nskip && c->red_r > 0) {
        ns.dirs = c->nskip; ns.nulls = c->nskip + c->ns_full; ns.rows = c->nskip + c->ns_full + c->ns_null;
        ns.nfull = c->ns_full; ns.nnull = c->ns_null; ns.nrows = c->ns_rows; ns.iperm = c->red_iperm; ns.r = c->red_r;
    }
    return ns;
}

// fs->Xr1 for the launch that applies the pending state update (and only once)
const double* take_xr1(srukf_ctx* c)
{
    if (!c->xr1_pending) return nullptr;
    c->xr1_pending = false;
    return (const double*)((const char*)c->fs + offsetof(FrameScalars, Xr1));
}

// Replay path: motion step + projection of all sigma points in ONE launch (k_project_motion): workgroup 0 is the motion step,
// whose results wait beside the state (fs->Xr1, Cmat) until k_gain / the dX job commit them.
void seq_predict_fused(srukf_ctx* c, int mode)
{
    const KDims& d = c->d;
    ProfScope ps(c, mode == 2 ? KC_PROJECT_TABLE : KC_PROJECT_MOTION, 2.0 * 60.0 * d.Na * d.N + 60.0 * d.L, 8.0 * ((double)d.n * d.n / 2 + 2.0 * d.L * 2 * d.N + (double)d.n * 2 * d.N + 8.0 * d.L + 8.0 * d.n));
    if (mode == 2) srukf_launch_project_table(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->Z, c->DZ, c->fs, rank_args(c, false, c->dbg.pxy2 != 0), null_skip(c));
    else srukf_launch_project_motion(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->Z, c->DZ, c->fs, rank_args(c));
    c->xr1_pending = true;
}

void seq_predict_motion(srukf_ctx* c, const double* odo_pair_dev)
{
    const KDims& d = c->d;
    ProfScope ps(c, KC_MOTION, 60.0 * d.L, 8.0 * (4.0 * d.n + 8.0 * d.L + 4.0 * d.n));
    srukf_launch_motion(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->fs, c->odo_seq, odo_pair_dev, rank_args(c));
}

// fused_stats: the statistics ride on the k_pxy launch of seq_gain (replay path, no host in between)
void seq_predict_measurement(srukf_ctx* c, bool fused_stats)
{
    const KDims& d = c->d;
    {
        ProfScope ps(c, KC_PROJECT, 2.0 * 60.0 * d.Na * d.N, 8.0 * ((double)d.n * d.n / 2 + 2.0 * d.L * 2 * d.N + (double)d.n * 2 * d.N));
        srukf_launch_project(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Z, c->DZ, c->fs);
    }
    if (!fused_stats) {
        ProfScope ps(c, KC_STATS, 30.0 * d.L * d.N, 8.0 * 3.0 * d.L * 2 * d.N);
        srukf_launch_meas_stats(c->stream, d, c->w, c->X, c->sigR, c->Z, c->mpart, c->h, c->Si, c->vis, c->PxyR);
    }
}

// one refactorisation  S <- gmw(S^T S - U[ub:ue] U[ub:ue]^T);  slow = column-by-column path.
// need_reset: the gamma/xi accumulators were not just cleared by k_gain (SEQUENTIAL mode, fallbacks).
// frame_tail: the check kernel also records the trajectory row and advances the staged frame counter.
int gmw_fused_mode() { return g_dbg_gmw_fused; }

int rank_fused_mode() { return g_dbg_rank_fused; }

int rank_fold_mode() { return g_dbg_rank_fold; }

void shadow_rebuild(srukf_ctx* c)
{
    if (c->red_r > 0 && c->shadowA) srukf_launch_rank_shadow(c->stream, c->d.n, c->d.np, c->red_r, c->S, c->red_perm, c->shadowA);
}

bool gmw_plan_persists(const srukf_ctx* c, const GmwPlan& gp) { return gp.workers >= 0 || split_form(c, gp, true); }

bool gmw_use_persist(const srukf_ctx* c) { return gmw_persist_mode() && c->gmw_shared != 2 && gmw_plan_persists(c, c->gplan); }

}  // namespace srukf_impl

// SRUKF_GPU_SHARED: how many persistent launches share the GPU (each keeps to 1 / tenants of the CUs; the gate admits that many)
// (per context: srukf_run_frames_batch picks it from the number of filters it runs — one tenant per filter up to SRUKF_MAX_TENANTS; srukf_set_exclusive alone uses the
//  process-wide default of srukf_debug_set "shared_tenants")

namespace srukf_impl {

int plan_tenants(const srukf_ctx* c) { return c->gmw_shared == 1 ? c->shared_tenants : 1; }

int gate_limit(const srukf_ctx* c) { return c->gmw_shared == 1 ? c->shared_tenants : 0; }

// Tail of every rank-aware refactorisation: factor rows (c->G, permuted order) -> S and the permuted copy, checks, frame tail.
// fp32 storage: S, X and the permuted copy are rounded to the stored values first, and the trajectory row is taken from those
// (as the full-rank form does: quantize_state before the tail).
// table: "table" mode of the replay — the tail also prepares the next frame's table of robot poses (k_rank_expand)
// fuse: "fused tail" mode — this launch also projects the next frame's sigma points (k_rank_expand<2>)
void rank_expand(srukf_ctx* c, bool frame_tail, bool table, bool fuse)
{
    const int n = c->d.n, np = c->d.np;
    const bool f32s = c->storage != SRUKF_STORAGE_F64;            // (the mixed mode stores floats as well: the launches behind the tail round, as for fp32 storage without "fused tail" mode)
    const bool f32fuse = storage_f32_like(c) && fuse && table && frame_tail;      // fp32 storage in "fused tail" mode: the launch rounds what it writes (no k_quantize / k_rank_round / k_traj behind it)
    const bool f32 = f32s && !f32fuse;
    const bool tt = table && frame_tail && !f32;
    const bool exports = c->step_export.dst && tt && fuse;         // step-wise fast path: this launch is the frame's last and hands status + robot view to the host itself
    srukf_launch_rank_expand(c->stream, n, np, c->red_r, c->p.epsilon, c->G, c->D, c->red_perm, c->red_iperm, c->gdiag, c->fs, c->X, (frame_tail && !f32) ? 1 : 0, c->S, c->shadowA,
                             tt ? c->sigR : nullptr, c->w.gamma, (tt && fuse) ? 1 : 0, c->d, c->w, c->p, c->Z, c->DZ, f32fuse ? 1 : 0, exports ? &c->step_export : nullptr,
                             c->storage == SRUKF_STORAGE_F32_MIXED ? 1e-6 * c->dbg.mixed_null_ppm : 0.0,
                             (tt && fuse) ? c->fold_sync : nullptr, c->fold_sync ? 4 * (c->d.mp / 64) * (c->d.np / 64) : 0);
    c->step_export_attached = exports;
    if (f32) {
        quantize_state(c);
        srukf_launch_rank_round(c->stream, np, c->red_r, c->shadowA);
        if (frame_tail) srukf_launch_traj(c->stream, c->d, c->X, c->S, c->fs, nullptr, 1);
    }
}

}  // namespace srukf_impl

// the rank-aware replay whose owners form their tiles of S^T S - U U^T themselves (seq_refactor below): what a whole staged frame takes
// How many tiles per worker (in percent) the owners' fold accepts.  A filter that has the GPU to itself: 106 = about one tile per worker (measured in round 2: with two
// tiles per worker, both to be formed before the first step, the exclusive replay loses — N = 300: 1 544 against 1 663 frames/s; srukf_debug_set "fold_tiles_pct").
// A filter that shares the GPU (three or four tenants of 256 / tenants CUs: two register tiles per worker): 200 — measured in round 4 at N = 200, four filters and four
// tenants, aggregate frames/s: owners fold both tiles 12 260; the same tiles from a launch of their own in the owners' summation order (k_syrk_own) 8 500 - 11 900;
// split-K k_syrk over the kept rows 13 100 but then the results differ in rounding from the same filter running alone.
static int fold_tiles_pct(const srukf_ctx* c) { return c->gmw_shared == 1 ? 200 : 106; }

namespace srukf_impl {

bool replay_red_fused(const srukf_ctx* c)
{
    return c->red_r > 0 && c->storage != SRUKF_STORAGE_F32_MIXED && c->shadowA && c->w.wc0 == c->w.wm0 && gmw_use_persist(c) &&
           c->gplan_red.workers >= 0 && c->gplan_red.nreal <= c->gplan_red.workers * fold_tiles_pct(c) / 100 && c->gplan_red.T >= 16 &&
           !c->debug_starve && gmw_fused_mode() && rank_fused_mode() && rank_fold_mode();
}

// 0: k_motion + k_project; 1: k_project_motion (motion workgroup + projection with the robot part inline); 2: "table" (k_project_table: the
// previous frame's tail prepared the robot part of every sigma point) — only where the tail is k_rank_expand on fp64 storage
// ... or forms them with k_syrk over the kept rows, still in permuted order (memory tiles, two tiles per worker, one launch per panel: seq_refactor's second branch)
bool replay_red_perm(const srukf_ctx* c)
{
    return c->red_r > 0 && (c->storage != SRUKF_STORAGE_F32_MIXED || (c->dbg.mixed_rank && c->A32)) && c->shadowA && c->w.wc0 == c->w.wm0 && rank_fused_mode() && c->dbg.table_perm;
}

int replay_motion_mode(const srukf_ctx* c)
{
    // fp32 storage: only as "fused tail" mode (k_rank_expand<2> and the state update round what they write; "table" mode alone has no such form)
    const bool st_ok = c->storage == SRUKF_STORAGE_F64 ||
                       (storage_f32_like(c) && c->dbg.f32_fuse && c->dbg.tail_fuse && c->dbg.pxy2 && c->dbg.nullskip && c->nskip && c->tail_ok &&
                        (size_t)c->d.np * sizeof(double) <= 48 * 1024);
    // (null_canonical: "table" mode and everything on top of it read the structurally null rows of S as sqrt(EPSILON) e_k without looking)
    if (c->dbg.fused_motion == 2 && !((replay_red_fused(c) || replay_red_perm(c)) && st_ok && c->null_canonical)) return 1;
    return c->dbg.fused_motion;
}

// "fused tail" mode (default where "table" mode runs with k_pxy2 and NullSkip): k_rank_expand also projects the next frame (k_rank_expand<2>), the frame's motion reduction
// rides on k_pxy2 (MeasArgs::fmode), k_gain re-centres the robot rows: a frame is k_pxy2, k_gain, k_gmw_persist, k_rank_expand, and only a run's first frame has a projection launch
bool replay_fuse_mode(const srukf_ctx* c)
{
    return replay_motion_mode(c) == 2 && c->dbg.pxy2 && c->dbg.nullskip && c->nskip && c->tail_ok && c->dbg.tail_fuse &&
           (size_t)c->d.np * sizeof(double) <= 48 * 1024;      // (k_rank_expand<2> keeps a row of the factor in dynamic LDS)
}

}  // namespace srukf_impl

// Round 5 measured where the head fold pays and where it cannot run at all (scripts/head_fold_sweep.py, scripts/abort_probe.py with the diagnostic build's time stamps):
//   * every worker's first step waits for ALL head tiles, and the helpers only get the CUs that pivot + workers leave free: with 261 helper jobs on 34 free CUs (N = 266) the
//     last one ends 125 us into the launch, the workers idle from 57 us on and the pivot behind them — frames/s with / without the fold: N = 200 5 269 / 5 098, 215 4 847 / 4 713,
//     230 4 488 / 4 487, 240 4 256 / 4 299, 250 3 681 / 4 179, 266 3 224 / 3 790.  The fold is kept while the helpers are at most 2.25 rounds on the free CUs (N <= 223);
//   * with 22 free CUs (N = 267 .. 275, pivot + 233 workers) the launch never completes by itself: some helpers run, then no further one starts until the workers give up
//     (all 270 finish right after the 10.7 ms bound: stamps).  29 or 30 main workgroups on an XCD's 32 CUs leave one of its shader engines without a free CU, and the
//     dispatcher, which deals workgroups round-robin to the engines, apparently does not skip a full one: with at most 28 per XCD (7 per engine: <= 224 main workgroups, >= 32
//     CUs free) it never happened.  Until this sweep the threshold was 16 free CUs and nothing between 9 and 34 had ever run: N = 267 .. 275 fell back to one launch per
//     panel without saying so.
#define SRUKF_HEAD_FOLD_MIN_FREE_CUS 32

namespace srukf_impl {

bool head_fold_ok(const srukf_ctx* c)
{
    const int need = g_dbg_head_fold_free.load() > 0 ? g_dbg_head_fold_free.load() : SRUKF_HEAD_FOLD_MIN_FREE_CUS;
    const int free_cus = c->gplan_red.cus - 1 - c->gplan_red.workers;
    const int nhelp = c->n_syrk_head_tiles + (c->d.n + 255) / 256 + (c->d.n - c->red_r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;       // as seq_refactor counts them
    const bool pays = g_dbg_head_fold_free.load() > 0 || 4 * nhelp <= 9 * free_cus;      // (the A/B switch "head_fold_free" overrides the rounds rule: measurements)
    return c->dbg.head_fold && c->gmw_shared == 0 && free_cus >= need && pays;
}

void seq_refactor(srukf_ctx* c, int ub, int ue, bool slow, bool keep_backup, bool need_reset, bool frame_tail, bool table, bool fuse)
{
    const KDims& d = c->d;
    const int np = d.np, n = d.n;
    if (need_reset || slow) {
        ProfScope ps(c, KC_MISC, 0, 8.0 * np);
        hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 1);
    }
    // Fused form (replay path): k_syrk only for the first block rows, the persistent launch computes the other tiles of
    // S^T S - U U^T itself while it is already factoring; it reads the filter's S for that, so the factor goes to the
    // scratch buffer Wf and k_gmw_check copies it into S.
    // Measured (frames/s, fused against not fused): N = 200 2 965 / 2 910, N = 100 5 247 / 5 262, N = 50 9 256 / 9 465,
    // N = 300 (two tiles per worker, both to be computed first) 1 544 / 1 663 — so only with one tile per worker and T >= 16.
    // rank-aware form (srukf_rank.hip): permute the null directions to the end, pivot only the leading red_Tp panels
    // (the mixed-precision downdate too, round 6: the null pivots its fp32-formed G cannot resolve — what made the full-rank form of the mode diverge — are not factored at all)
    const bool reduced = !slow && c->red_r > 0 && (c->storage != SRUKF_STORAGE_F32_MIXED || (c->dbg.mixed_rank && c->A32));
    // ... and on the replay path directly in permuted order from the shadow copy (no full k_syrk, no permutation pass)
    // (the owners' fold pays with about one tile per worker and T >= 16, as in the full-rank form: frames/s fold / k_syrk over the kept
    //  rows: N = 100 7 360 / 7 610, N = 200 4 360 / 4 300, N = 300 — two tiles per worker — 2 400 / 2 580)
    const bool red_fused = reduced && !keep_backup && ub == 0 && ue == d.mp && replay_red_fused(c);
    if (red_fused) {
        const double rr = c->red_r, hr = srukf_gmw_head_rows();
        const double head_flop = 2.0 * hr * n * (hr / 2.0 + d.mp) + 2.0 * (n - rr) * (rr + d.mp), head_byte = 8.0 * ((hr + d.mp) * n + (n - rr) * (rr + d.mp));
        // head fold (a filter that has the GPU to itself): the head tiles, the pending X += dX and the dropped diagonal are helper
        // workgroups of the persistent launch instead of a k_syrk launch in front of it (srukf_debug_set "head_fold", 0: two launches)
        // Only with CUs to spare: the helpers are dispatched behind the pivot and the workers, which spin on their tiles — and the launch's static LDS allows one
        // workgroup per CU.  A plan whose pivot + workers (nearly) fill the GPU (255-270 tiles) would leave the helpers waiting for a main workgroup to exit:
        // the pivot's bounded wait would expire.  Such plans keep the k_syrk launch in front (head_fold_ok).
        const bool head_fold = head_fold_ok(c);
        if (!head_fold) {
            // head rows of Gp: K = hr rows of the shadow copy (upper triangular) + the 2N measurement rows; + the dropped diagonal
            ProfScope ps(c, KC_SYRK, head_flop, head_byte);
            srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->syrk_head_tiles, c->n_syrk_head_tiles, dx_src(c), c->X, rank_args(c, table, false, fuse && storage_f32_like(c)), take_xr1(c));
            c->dx_pending = false;
        }
        {
            // factorisation of the leading red_Tp panels (all n columns carried along) + the owners' tiles of S^T S - U U^T
            // (kept rows below the head x all columns, K <= r and 2N): red_*_flop, update_null_set
            ProfScope ps(c, KC_GMW_PERSIST, c->red_fac_flop + c->red_own_flop + (head_fold ? head_flop : 0.0), 8.0 * (2.0 * rr * n + (double)d.mp * n));
            HeadArgs ha = {};
            if (head_fold) {
                ha.tiles = (const int2*)c->syrk_head_tiles; ha.ntiles = c->n_syrk_head_tiles; ha.ncrit = c->n_syrk_head_crit;
                ha.dxp = dx_src(c); ha.X = c->X; ha.xr1 = take_xr1(c); ha.ndx = c->dx_pending ? (n + 255) / 256 : 0;
                ha.ra = rank_args(c, table, false, fuse && storage_f32_like(c)); ha.ngd = (n - c->red_r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;
                ha.nhelp = ha.ntiles + ha.ndx + ha.ngd;             // one helper workgroup per job, behind the pivot and the workers in dispatch order
                c->dx_pending = false;
            }
            srukf_launch_gmw_persist_head(c->stream, n, np, c->p.epsilon, c->Wf, c->gplan_red.pans, c->D, c->G, c->gplan_red.sync, c->gplan_red.tiles, c->gplan_red.ntiles,
                                          c->gplan_red.workers, c->fs, c->shadowA, c->Utp, 0, d.mp, c->red_Tp, (c->red_r + 15) & ~15, gate_limit(c), head_fold ? &ha : nullptr);
        }
        ProfScope ps(c, KC_RANK_EXPAND, 0, 8.0 * 2.5 * (double)n * n);
        rank_expand(c, frame_tail, table, fuse);
        return;
    }
    // ... or, where the owners cannot fold (memory tiles: more than two tiles per worker; one launch per panel), still in permuted
    // order: k_syrk over the tiles of the kept rows only, K <= r, straight into Gp
    const bool red_perm = reduced && !red_fused && !keep_backup && ub == 0 && ue == d.mp && c->shadowA && c->w.wc0 == c->w.wm0 && rank_fused_mode();
    if (red_perm) {
        const double rr = c->red_r, rp = 64.0 * c->red_Tp;
        // A filter that shares the GPU and whose workers own two register tiles (three or four tenants): the head rows by k_syrk, every other tile by k_syrk_own in
        // the summation order of the owners' fold — bit for bit what the same filter computes when it runs alone (its owners fold) — then the persistent launch reads
        // its tiles from Gp.  (The memory-tile form and the launches per panel keep the split-K k_syrk over the kept rows: nothing to be identical to.)
        const bool own_order = c->gmw_shared == 1 && gmw_use_persist(c) && c->gplan_red.workers > 0 && c->gplan_red.T >= 16 && !c->debug_starve && gmw_fused_mode() && rank_fold_mode() &&
                               srukf_gmw_register_form(c->gplan_red.T, c->gplan_red.Tp, c->gplan_red.ntiles, c->gplan_red.workers);
        const bool sfold = !own_order && split_fold_ok(c);
        const double syrk_flop_all = rr * rr * rr / 3.0 + rr * rr * (n - rr) + 2.0 * d.mp * (rr * n - rr * rr / 2.0) + 2.0 * (n - rr) * (rr + d.mp);
        {
            ProfScope ps(c, KC_SYRK, sfold ? c->red_head0_flop + 2.0 * (n - rr) * (rr + d.mp) : syrk_flop_all, 8.0 * (rr * n + (double)d.mp * n + rp * n));
            if (c->storage == SRUKF_STORAGE_F32_MIXED) {
                // the mixed-precision downdate in the rank-aware form: the kept rows of S in permuted column order (what the stored floats hold: the permuted copy is
                // rounded with S) and U^T with permuted columns as fp32 operands, K <= r, products on the fp32 matrix pipe, chunk sums in FP64 — only the macro tiles of the
                // pivoted panels; the state update and the dropped diagonal (FP64, from the same operands) by k_syrk's spare workgroups with an empty tile list
                if (c->dbg.mixed_bf16 && c->mxr_xt) {
                    // operands as three bf16 pieces each, transposed (K contiguous per column): the products on the bf16 matrix pipe, six per fp32 product
                    srukf_launch_split_bf3(c->stream, c->mxr_krows, np, c->mxr_ktot, 0, c->shadowA, c->mxr_xt, c->mxr_xt_stride);
                    srukf_launch_split_bf3(c->stream, d.mp, np, c->mxr_ktot, c->mxr_krows, c->Utp, c->mxr_xt, c->mxr_xt_stride);
                    srukf_launch_syrk_bf3(c->stream, n, np, d.mp, c->mxr_krows, c->mxr_ktot, c->mxr_xt, c->mxr_xt_stride, c->mxr_tasks, c->mxr_ntasks, c->mxr_tiles, c->mxr_ntiles,
                                          c->mxr_part, c->Wf, c->fs);
                } else {
                    srukf_launch_cvt_f32(c->stream, (size_t)c->mxr_krows * np, c->shadowA, c->A32);
                    srukf_launch_cvt_f32(c->stream, (size_t)d.mp * np, c->Utp, c->U32);
                    srukf_launch_syrk32(c->stream, n, np, d.mp, c->A32, c->U32, c->mxr_tasks, c->mxr_ntasks, c->mxr_tiles, c->mxr_ntiles, c->mxr_part, c->Wf, c->fs, c->mxr_krows);
                }
                // ... and BEHIND it, in FP64 from the FP64 operands, the few tiles whose pivots an fp32-formed product cannot resolve (the robot block, the shared anchor:
                // mxr_f64_tiles) — they overwrite what the fp32 launch left there
                srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->mxr_f64_tiles, c->dbg.mixed_f64_robot ? c->mxr_n_f64_tiles : 0,
                                  dx_src(c), c->X, rank_args(c, table, false, fuse && storage_f32_like(c)), take_xr1(c));
            } else if (own_order) {
                srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->syrk_head_tiles, c->n_syrk_head_tiles, dx_src(c), c->X, rank_args(c, table, false, fuse && storage_f32_like(c)), take_xr1(c));
                srukf_launch_syrk_own(c->stream, n, np, c->shadowA, c->Utp, 0, d.mp, (c->red_r + 15) & ~15, c->Wf, c->fs, c->gplan_red.tiles, c->gplan_red.ntiles, c->red_Tp);
            } else if (sfold && c->dbg.split_fold != 2)
                // split fold: block row 0 and tile (1, 1) here (+ the state update and the dropped diagonal, as always); the tile launch of the pair forms the rest
                srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->red_head0_tiles, c->n_red_head0_tiles, dx_src(c), c->X, rank_args(c, table, false, fuse && storage_f32_like(c)), take_xr1(c));
            else
            srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->red_syrk_tiles, c->n_red_syrk_tiles, dx_src(c), c->X, rank_args(c, table, false, fuse && storage_f32_like(c)), take_xr1(c));
            c->dx_pending = false;
        }
        {
            ProfScope ps(c, gmw_use_persist(c) && gmw_plan_persists(c, c->gplan_red) ? KC_GMW_PERSIST : KC_GMW_TRAIL, c->red_fac_flop + (sfold ? syrk_flop_all - c->red_head0_flop : 0.0), 8.0 * 2.0 * rp * n);
            launch_gmw_fast(c, c->Wf, c->G, true, sfold);
        }
        ProfScope ps(c, KC_RANK_EXPAND, 0, 8.0 * 2.5 * (double)n * n);
        rank_expand(c, frame_tail, table, fuse);
        return;
    }
    const bool fused = !reduced && !slow && !keep_backup && gmw_use_persist(c) && ub == 0 && ue == d.mp && c->storage == SRUKF_STORAGE_F64 &&
                       c->gplan.ntiles <= c->gplan.workers && c->gplan.T >= 16 &&
                       !c->debug_starve && gmw_fused_mode();
    const double nn = n;
    const double syrk_flop = nn * nn * nn / 3.0 + nn * nn * (ue - ub), syrk_byte = 8.0 * (nn * nn + (double)(ue - ub) * nn);
    const double head_frac = fused ? fmin(1.0, 2.0 * srukf_gmw_head_rows() / nn) : 1.0;      // share of the tiles k_syrk still computes (rows / n, upper triangle)
    if (c->storage == SRUKF_STORAGE_F32_MIXED && ub == 0 && ue == d.mp && !(c->dbg.mixed_rank && c->A32)) {
        // mixed precision, round 2's full-rank form (study: "mixed_rank" 0): the fp32 state S32 and U^T rounded once, products on the fp32 matrix pipe, chunk sums in FP64.
        // (In the rank-aware form of the mode only the staged replay's branch above forms the product in fp32; whatever comes through here — the step-wise calls, a
        //  flagged frame's repeat on the exact path — forms it in FP64 from the FP64 working copies of the stored floats: the exact path must not divide fp32 noise.)
        ProfScope ps(c, KC_SYRK, syrk_flop, 4.0 * (nn * nn + (double)(ue - ub) * nn) + 8.0 * nn * nn / 2);
        if (c->dx_pending) srukf_launch_gain_dx(c->stream, n, np, c->dxp, c->X, take_xr1(c));
        c->dx_pending = false;
        srukf_launch_cvt_f32(c->stream, (size_t)d.mp * np, c->Ut, c->U32);
        srukf_launch_cvt_robot_cols(c->stream, n, np, c->S, c->S32);          // the motion step's columns, computed after the state was rounded
        srukf_launch_syrk32(c->stream, n, np, d.mp, c->S32, c->U32, c->mx_tasks, c->mx_ntasks, c->mx_tiles, c->mx_ntiles, c->mx_part, c->G, c->fs, np);
    } else {
        ProfScope ps(c, KC_SYRK, syrk_flop * head_frac, syrk_byte * head_frac);
        srukf_launch_syrk(c->stream, d, c->S, c->Ut, ub, ue, c->G, c->fs, fused ? c->syrk_head_tiles : c->syrk_tiles,
                          fused ? c->n_syrk_head_tiles : c->n_syrk_tiles, dx_src(c), c->X, RankArgs{}, take_xr1(c));
        c->dx_pending = false;
    }
    if (keep_backup) hipMemcpyAsync(c->Gbak, c->G, sizeof(double) * (size_t)np * np, hipMemcpyDeviceToDevice, c->stream);
    if (reduced) {
        // Gp = Pi^T G Pi into Wf (+ its diagonal), factor the leading red_Tp panels of Gp with the factor rows going to G (scratch
        // now), then back to state order with the theta check, the null-direction check and the frame tail in one kernel
        {
            ProfScope ps(c, KC_MISC, 0, 16.0 * nn * nn);
            hipLaunchKernelGGL(k_sym_permute, dim3(np), dim3(256), 0, c->stream, n, np, c->G, np, c->Wf, c->red_perm);
            srukf_launch_rank_diag(c->stream, n, np, c->G, c->red_perm, c->gdiag);
        }
        {
            const double rr = 64.0 * c->red_Tp;
            ProfScope ps(c, gmw_use_persist(c) && gmw_plan_persists(c, c->gplan_red) ? KC_GMW_PERSIST : KC_GMW_TRAIL, rr * rr * rr / 3.0 + rr * rr * (nn - rr) + rr * (nn - rr) * (nn - rr) / 2.0,
                         8.0 * (rr * nn));
            launch_gmw_fast(c, c->Wf, c->G, true);
        }
        ProfScope ps(c, KC_GMW_CHECK, 0, 8.0 * nn * nn);
        rank_expand(c, frame_tail);
        return;
    }
    if (!slow) {
        // 64-row panels: j0 = -64 factors the first 64x64 region, then one launch per panel
        // per panel: trailing update 64*r2^2 (upper half, 2 flop) + three-stage slab recompute + next 64x64 diagonal region
        auto panel_flop = [&](int j0) { const double r2 = np - j0 - 64; return j0 < 0 ? 64.0 * 64.0 * 64.0 / 3.0 : 64.0 * r2 * r2 + 3.0 * 2.0 * 32.0 * 32.0 * r2 + 64.0 * 64.0 * 64.0 / 3.0; };
        auto panel_byte = [&](int j0) { const double r2 = np - j0 - 64; return 8.0 * (r2 * r2 + 2.0 * 64.0 * r2); };
        if (gmw_use_persist(c)) {
            double fl = 0.0, by = 0.0;
            for (int j0 = -64; j0 + 64 < np; j0 += 64) { fl += panel_flop(j0); by += panel_byte(j0); }
            ProfScope ps(c, KC_GMW_PERSIST, fl + syrk_flop * (1.0 - head_frac), by + syrk_byte * (1.0 - head_frac));
            if (fused) srukf_launch_gmw_persist(c->stream, n, np, c->p.epsilon, c->G, c->gplan.pans, c->D, c->Wf, c->gplan.sync, c->gplan.tiles, c->gplan.ntiles,
                                                c->gplan.workers, c->fs, c->S, c->Ut, ub, ue, 0, 0, gate_limit(c));
            else launch_gmw_fast(c, c->G, c->S);
        } else {
            int pb = 0;
            for (int j0 = -64; j0 + 64 < np; j0 += 64, pb ^= 1) {
                ProfScope ps(c, KC_GMW_TRAIL, panel_flop(j0), panel_byte(j0));
                srukf_launch_gmw_step64(c->stream, n, np, j0, c->p.epsilon, c->G, c->pan[pb ^ 1], c->pan[pb], c->D, c->S, c->fs);
            }
        }
        quantize_state(c);
        ProfScope ps(c, KC_GMW_CHECK, 0, 8.0 * (double)n * n / 2);
        srukf_launch_gmw_check(c->stream, n, np, c->D, fused ? c->Wf : c->S, c->fs, c->X, frame_tail ? 1 : 0, fused ? c->S : nullptr);
    } else {
        ProfScope ps(c, KC_GMW_COL, (double)n * n * n / 3.0, 8.0 * (double)n * n * n / 3.0);
        exact_path(c, c->G, c->S);
        quantize_state(c);
        if (frame_tail) srukf_launch_traj(c->stream, d, c->X, c->S, c->fs, nullptr, 1);
    }
    shadow_rebuild(c);                                 // S was rewritten by a path that does not keep the permuted copy in step
}

// Blocked fast path (or, slow, the exact column path) on an arbitrary matrix buffer: Gbuf (upper triangle,
// destroyed) -> upper-triangular factor rows in Sout (whose lower triangle must already be zero).
// split fold (srukf_gmw_persist.hip, k_gmw_tiles_fold): the rank-aware replay's k_syrk over the kept rows becomes jobs of the split form's tile launch
bool split_fold_ok(const srukf_ctx* c)
{
    // Only while every tile workgroup of the plan can be resident beside the pivot / slab launch (four per CU): they hold their places from dispatch to their row's
    // last update, and beyond that the forming jobs queue behind waiting workgroups — frames/s with / without the fold: N = 400 2 150 / 1 960, 500 1 385 / 1 350,
    // 600 (1 190 tile workgroups for 796 places) 855 / 893, 800 393 / 459.
    const GmwPlan& gp = c->gplan_red;
    const bool fits = gp.nreal <= 4 * (gp.cus - gp.T) || g_dbg_fold_force.load() != 0;      // ("fold_force": measurements)
    return c->dbg.split_fold && (fits || c->dbg.split_fold == 2) && c->red_r > 0 && c->storage != SRUKF_STORAGE_F32_MIXED && c->split_fold_list && c->n_split_fold > 0 && c->red_head0_tiles &&
           !c->debug_starve && !c->dbg.split_record && gmw_use_persist(c) && gmw_plan_persists(c, gp) && split_form(c, gp, true);
}

void launch_gmw_fast(srukf_ctx* c, double* Gbuf, double* Sout, bool reduced, bool fold)
{
    const int np = c->d.np, n = c->d.n;
    const GmwPlan& gp = reduced ? c->gplan_red : c->gplan;
    const int Tp = reduced ? c->red_Tp : np / 64;
    if (gmw_use_persist(c) && gmw_plan_persists(c, gp)) {
        // srukf_debug_starve_workers (tests only): launch without workers, as if the GPU were taken — the pivot's bounded wait
        // expires, the frame is flagged and repeated on the exact path, and the context falls back to one launch per panel
        const int workers = c->debug_starve == 1 ? 0 : gp.workers;   // (2: only a split-form pair is starved — the tier below it then runs undisturbed)
        if (split_form(c, gp, true)) {                         // (srukf_debug_starve_workers: the pair without its tile launch)
            // the tile launch depends on what produced Gbuf, not on the pivot / slab launch: fork before, join after (in a capture: two parallel branches)
            if (c->dbg.split_record) hipMemcpyAsync(c->Gbak, Gbuf, sizeof(double) * (size_t)np * np, hipMemcpyDeviceToDevice, c->stream);
            hipEventRecord(c->ev_fork, c->stream);
            hipStreamWaitEvent(c->side, c->ev_fork, 0);
            if (fold && reduced) c->split_fold_seqs++;
            if (fold && reduced)
                srukf_launch_gmw_split_fold(c->stream, c->side, n, np, c->p.epsilon, Gbuf, gp.pans, c->D, Sout, gp.sync, c->split_fold_list, c->n_split_fold, c->fs, Tp, (c->red_r + 15) & ~15,
                                            c->gsW, c->gsL, c->shadowA, c->Utp, c->d.mp);
            else
            srukf_launch_gmw_split(c->stream, c->side, n, np, c->p.epsilon, Gbuf, gp.pans, c->D, Sout, gp.sync, gp.tiles, gp.ntiles, c->fs, Tp, reduced ? ((c->red_r + 15) & ~15) : 0, c->gsW, c->gsL, c->debug_starve ? 1 : 0);
            hipEventRecord(c->ev_join, c->side);
            hipStreamWaitEvent(c->stream, c->ev_join, 0);
            return;
        }
        // (krows: where the kept pivots end — the last pivoted panel is not factored beyond them)
        srukf_launch_gmw_persist(c->stream, n, np, c->p.epsilon, Gbuf, gp.pans, c->D, Sout, gp.sync, gp.tiles, gp.ntiles, workers, c->fs, nullptr, nullptr, 0, 0, Tp,
                                 reduced ? ((c->red_r + 15) & ~15) : 0, gate_limit(c));
        return;
    }
    // one launch per panel; rank-aware form: the step after the last pivoted panel still runs (it writes that panel's S rows)
    int pb = 0;
    for (int j0 = -64; j0 + 64 < np && j0 + 64 <= 64 * Tp; j0 += 64, pb ^= 1)
        srukf_launch_gmw_step64(c->stream, n, np, j0, c->p.epsilon, Gbuf, c->pan[pb ^ 1], c->pan[pb], c->D, Sout, c->fs);
}

// The exact path: n + 1 launches behind one call (srukf_launch_gmw_col submits the right-looking sequence on its first call).  Launch-bound: 13.6 ms per flagged frame at
// N = 200 for ~5 ms of kernels.  As ONE captured graph per context it measured worse where it counts (38.8 ms per flagged frame over bench.py's 20-frame leg with three flagged
// frames: instantiating 1 205 nodes costs more than the two replays save), so the launches stay eager.
void exact_path(srukf_ctx* c, const double* Gbuf, double* Sout)
{
    const int np = c->d.np, n = c->d.n;
    for (int j = 0; j < n; j++) srukf_launch_gmw_col(c->stream, n, np, j, c->p.epsilon, Gbuf, c->Wf, c->D, c->theta, c->fs, Sout);
}

void run_gmw(srukf_ctx* c, double* Gbuf, double* Sout, bool slow)
{
    const int np = c->d.np, n = c->d.n;
    if (!slow) {
        launch_gmw_fast(c, Gbuf, Sout);
        srukf_launch_gmw_check(c->stream, n, np, c->D, Sout, c->fs, c->X, 0, nullptr);
    } else {
        hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 0);
        exact_path(c, Gbuf, Sout);
    }
}

// GSLCholeskyUpdate with FLAG_NEED_REORDER (SLAM.cpp:2122-2138) for the columns [ub, ue) of U:
//   dst = S^T S - U U^T;  dst_dis = Pi^T dst Pi  (disordered layout: the rank-deficient new-anchor block last);
//   S_dis = [R11 R12; 0 0],  R11 = gmw(dst_dis[0:r, 0:r]),  R12 = R11^{-T} dst_dis[0:r, r:n]   (2158-2179, r = n - 3 K_new);
//   S = R factor of QR(Pi S_dis Pi^T).
// [R11 R12] is what the right-looking GMW leaves in its first r rows whatever stands in the lower right block, so the
// full factorisation runs and rows >= r are zeroed.  R^T R = Pi (S_dis^T S_dis) Pi^T, so the QR is a second
// SYRK + permutation + GMW (P = S^T S is what the filter consumes; row signs of R are a convention).
int refactor_reorder(srukf_ctx* c, int ub, int ue)
{
    const KDims& d = c->d;
    const int np = d.np, n = d.n, r = n - 3 * c->K_new;
    const size_t bytes = sizeof(double) * (size_t)np * np;
    if (!c->Sdis) { if (srukf_dmalloc((void**)&c->Sdis, bytes) != hipSuccess) { c->err = "out of device memory (NEED_REORDER buffer)"; return SRUKF_ERR_NOMEM; } }
    hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 1);
    srukf_launch_syrk(c->stream, d, c->S, c->Ut, ub, ue, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, dx_src(c), c->X, RankArgs{}, take_xr1(c));
    c->dx_pending = false;
    for (int stage = 0; stage < 2; stage++) {
        double* out = stage == 0 ? c->Sdis : c->S;
        if (stage == 1) {
            hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 1);
            srukf_launch_syrk(c->stream, d, c->Sdis, c->Ut, 0, 0, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, nullptr, c->X, RankArgs{}, nullptr);
        }
        for (int slow = 0; slow < 2; slow++) {
            hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
            hipLaunchKernelGGL(k_sym_permute, dim3(np), dim3(256), 0, c->stream, n, np, c->G, np, c->Gbak, stage == 0 ? c->iperm : c->perm);
            if (stage == 0) HIPCHK(c, hipMemsetAsync(c->Sdis, 0, bytes, c->stream));
            run_gmw(c, c->Gbak, out, slow != 0);
            if (stage == 0 && r < np) hipLaunchKernelGGL(k_zero_rows, dim3(np - r), dim3(256), 0, c->stream, np, r, c->Sdis);
            if (slow) break;
            int rc = read_fs(c); if (rc) return rc;
            if (c->hfs->clamp_rows == 0) break;        // the theta clamp never won: the blocked result is the reference's
        }
    }
    quantize_state(c);
    return SRUKF_OK;
}

// fused_motion: the frame's motion step ran inside k_project_motion: the statistics take the robot mean from fs->Xr1, k_gain commits Cmat
// table: "table" mode of the replay — the product on the permuted operands (k_pxy2), k_gain takes it from there
// first half: the cross covariances (and, riding on the launch, the measurement statistics h / Si / visible; in "fused tail" mode the frame's motion reduction)
void seq_pxy(srukf_ctx* c, bool fused_stats, bool fused_motion, bool table, bool preamble, bool fmode, bool fold)
{
    const KDims& d = c->d;
    const double nn = d.n;
    ProfScope ps(c, table ? KC_PXY2 : KC_PXY, nn * nn * 2.0 * d.N, 8.0 * (nn * nn / 2 + 2.0 * nn * 2 * d.N));
    MeasArgs ms = {};
    // ("fused tail" mode: the statistics are centred on the centre point's robot part, row 0 of the table: the mean does not exist yet)
    const double* xrob = fmode ? c->sigR : fused_motion ? (const double*)((const char*)c->fs + offsetof(FrameScalars, Xr1)) : c->X + (d.n - 4);
    if (fused_stats) ms = MeasArgs{ c->X, xrob, c->sigR, c->Z, c->mpart, c->h, c->Si, c->vis, c->PxyR, c->fs, (d.N + 31) / 32, table ? null_skip(c) : NullSkip{}, preamble ? 1 : 0,
                                    fmode ? 1 : 0, c->Cmat };
    if (fused_stats && c->mirror_next) {                       // step-wise API: the host's pinned copy of h | Si | visible is filled by the statistics jobs themselves
        ms.hmirror = (char*)c->hmeas;
        ms.hflag = (unsigned long long*)((char*)c->hfs + sizeof(FrameScalars) + sizeof(double) * 32);
        ms.hseq = c->meas_seq;
        ms.hstamp = (unsigned long long*)(c->hmeas + c->d.mp + 5 * (size_t)c->d.N);       // (the spare words behind h | Si | visible)
    }
    GainFold gf = {};
    if (fold && table && fmode) {
        gf.sync = c->fold_sync; gf.Utp = c->Utp; gf.dxk = c->dxk; gf.z_seq = c->z_seq; gf.m_seq = c->m_seq;
        gf.DZp = c->DZ; gf.perm = c->red_perm; gf.iperm = c->red_iperm; gf.r = c->red_r; gf.split_b0 = c->pxy2_split_b0;
        gf.sqeps = storage_f32_like(c) ? (double)(float)sqrt(c->p.epsilon) : sqrt(c->p.epsilon);      // (the null rows of S as they are stored: what seq_gain_only hands k_gain)
        gf.sc = c->w.wi * c->w.gamma;
        gf.S = c->S; gf.A = c->shadowA;
        gf.nmt = d.mp / 64; gf.nbt = d.np / 64; gf.bt_r0 = (c->red_r - 4) / 64; gf.bt_r1 = (c->red_r - 1) / 64; gf.robot_tiles = c->fold_robot_tiles;
    }
    if (table) srukf_launch_pxy2(c->stream, d, c->DZ, c->shadowA, c->Utp, c->P1, c->pxy2_tiles, c->n_pxy2_tiles, (c->red_r + 15) & ~15, c->w, ms, gf);
    else srukf_launch_pxy(c->stream, d, c->DZ, c->S, c->Ut, c->pxy_tiles, c->n_pxy_tiles, c->w, ms);
}

// second half: gains, U^T, slice partials of the state update; z_dev / m_dev: this frame's measurements and matches on the device (null: the staged sequence's)
void seq_gain_only(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_motion, bool table, bool fmode)
{
    const KDims& d = c->d;
    ProfScope ps(c, KC_GAIN, 8.0 * d.n * 2 * d.N, 8.0 * 2.0 * d.n * 2 * d.N);
    srukf_launch_gain(c->stream, d, c->w, c->Ut, c->PxyR, c->Si, c->vis, c->h, c->z_seq, z_dev, c->m_seq, m_dev, c->fs, c->dxp, c->X, c->Z, rank_args(c),
                      fused_motion ? c->Cmat : nullptr, c->S, table ? c->P1 : nullptr, c->pxy2_split_b0, c->DZ,
                      (fmode && storage_f32_like(c)) ? (double)(float)sqrt(c->p.epsilon) : sqrt(c->p.epsilon),   // (the null rows of S as they are stored)
                      c->sigR, fmode ? 1 : 0, c->next_pose_pending ? c->next_odo + 3 : nullptr, c->odo_step);
    c->next_pose_pending = false;
    c->dx_pending = true; c->dx_lm = false;           // applied by the next k_syrk launch (seq_refactor)
}

void seq_gain(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_stats, bool fused_motion, bool table, bool preamble, bool fmode)
{
    // gain fold: the staged replay's "fused tail" frames (staged measurements, nothing for the host in between) form U^T and the state update inside k_pxy2
    const bool fold = c->dbg.gain_fold && fmode && table && fused_stats && fused_motion && !z_dev && !m_dev && !c->mirror_next && !c->next_pose_pending &&
                      c->fold_sync && c->dxk && c->w.wc0 == c->w.wm0 && c->z_seq && c->m_seq;
    seq_pxy(c, fused_stats, fused_motion, table, preamble, fmode, fold);
    if (fold) { c->dx_pending = true; c->dx_lm = true; c->fold_seqs++; }
    else seq_gain_only(c, z_dev, m_dev, fused_motion, table, fmode);
}

}  // namespace srukf_impl

// Which directions of the state are structurally null (srukf_rank.hip)?  Called whenever a state arrives from outside
// (srukf_set_state*, map changes): row energies of S on the device, the lists on the host.  srukf_debug_set(0, "rank_aware", 0) switches it off.

namespace srukf_impl {

int update_null_set(srukf_ctx* c)
{
    const int enabled = g_dbg_rank_aware;
    const int n = c->d.n, np = c->d.np, T = np / 64;
    const int was = c->red_r;
    step_invalidate(c);
    c->red_r = 0;
    if (enabled && c->rank_aware && n >= 128) {
        srukf_launch_row_energy(c->stream, n, np, c->S, c->D);
        HIPCHK(c, hipMemcpyAsync(c->hstage, c->D, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(c->hstage + np, sizeof(double), c->S, sizeof(double) * (np + 1), sizeof(double), n, hipMemcpyDeviceToHost, c->stream));   // diag S
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<int> perm, drop;
        for (int k = 0; k < n; k++) ((k < n - 4 && c->hstage[k] < SRUKF_NULL_ENERGY) ? drop : perm).push_back(k);
        // NullSkip and the sqrt(EPSILON) DZ term of k_gain assume that every structurally null row IS sqrt(EPSILON) e_k (what the reference's clamp leaves there and
        // every frame tail rewrites).  A state from outside only promises energy < 1e-12 (zero rows after joint initialisation, another small diagonal): then the
        // first staged frame runs the launch sequence that reads the rows as they are, and its tail makes them canonical (null_canonical, run_frames_async).
        {
            const double sq = c->storage != SRUKF_STORAGE_F64 ? (double)(float)sqrt(c->p.epsilon) : sqrt(c->p.epsilon);
            bool canon = true;
            for (int k : drop) canon = canon && c->hstage[np + k] == sq && c->hstage[k] == sq * sq;
            c->null_canonical = canon;
        }
        const int r = (int)perm.size(), Tp = (r + 63) / 64;
        if (!drop.empty() && Tp < T) {                            // worth it only if at least one whole panel leaves the pivot chain
            perm.insert(perm.end(), drop.begin(), drop.end());
            for (int k = n; k < np; k++) perm.push_back(k);
            std::vector<int> iperm(np);
            for (int a = 0; a < np; a++) iperm[perm[a]] = a;
            if (!c->red_perm) {
                HIPCHK(c, srukf_dmalloc(&c->red_perm, sizeof(int) * np)); HIPCHK(c, srukf_dmalloc(&c->red_iperm, sizeof(int) * np));
                HIPCHK(c, srukf_dmalloc(&c->gdiag, sizeof(double) * np));
            }
            HIPCHK(c, hipMemcpy(c->red_perm, perm.data(), sizeof(int) * np, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->red_iperm, iperm.data(), sizeof(int) * np, hipMemcpyHostToDevice));
            if (c->gplan_red.Tp != Tp || !c->gplan_red.pans || c->gplan_red.tenants != plan_tenants(c)) {
                gmw_plan_destroy(c->gplan_red, c->stream);
                const int rc = gmw_plan_create(c->gplan_red, np, c->stream, Tp, plan_tenants(c));
                if (rc) { c->err = "rank-aware refactorisation: allocation failed"; return rc; }
            }
            split_ensure(c, c->gplan_red);
            c->red_r = r; c->red_Tp = Tp;
            {
                // algorithmic flop of the rank-aware refactorisation (DESIGN.md "flop model"): pivots j < rp update rows (j, rp) x
                // columns [row, n) of the upper triangle; the owners form the tiles of rows [head, rp) from K = min(row + 32, r) + 2N terms
                const double rp = 64.0 * Tp, nn = n, kr = (r + 15) & ~15;
                c->red_fac_flop = (nn - rp) * rp * rp + rp * rp * rp / 3.0;
                c->red_own_flop = 0.0;
                for (int I = srukf_gmw_head_rows() / 64; I < Tp; I++)
                    for (int J = I; J < T; J++)
                        for (int h = 0; h < 2; h++) c->red_own_flop += 2.0 * 32.0 * 64.0 * (fmin(64.0 * I + 32.0 * h + 32.0, kr) + c->d.mp) * (I == J ? 0.75 : 1.0);
            }
            {
                // k_syrk tiles of block rows < Tp in the XCD-aware order of the full table (build_tile_table)
                std::vector<int> ts = build_tile_table(np / 32, np / 32, true, true, 0), tr;
                for (size_t q = 0; q + 1 < ts.size(); q += 2) if (ts[q] >= 0 && ts[q] * 32 < 64 * Tp) { tr.push_back(ts[q]); tr.push_back(ts[q + 1]); }
                if (c->red_syrk_tiles) srukf_dfree_on(c->red_syrk_tiles, c->stream);
                c->red_syrk_tiles = nullptr; c->n_red_syrk_tiles = (int)tr.size() / 2;
                HIPCHK(c, srukf_dmalloc(&c->red_syrk_tiles, sizeof(int) * tr.size()));
                HIPCHK(c, hipMemcpy(c->red_syrk_tiles, tr.data(), sizeof(int) * tr.size(), hipMemcpyHostToDevice));
                // split fold: what of that list stays with k_syrk, and the grid of the tile launch that forms the rest (srukf_gmw_persist.hip)
                std::vector<int> th;
                c->red_head0_flop = 0.0;
                const int kr16 = (r + 15) & ~15;
                const int fold_head = srukf_gmw_fold_head_rows(Tp);
                for (size_t q = 0; q + 1 < tr.size(); q += 2) if (srukf_gmw_fold_head_tile(tr[q], tr[q + 1], fold_head)) {
                    th.push_back(tr[q]); th.push_back(tr[q + 1]);
                    c->red_head0_flop += 2.0 * 32.0 * 32.0 * (std::min(32 * tr[q] + 32, kr16) + c->d.mp);
                }
                if (c->red_head0_tiles) srukf_dfree_on(c->red_head0_tiles, c->stream);
                if (c->split_fold_list) srukf_dfree_on(c->split_fold_list, c->stream);
                c->red_head0_tiles = nullptr; c->split_fold_list = nullptr;
                c->n_red_head0_tiles = (int)th.size() / 2;
                HIPCHK(c, srukf_dmalloc(&c->red_head0_tiles, sizeof(int) * std::max<size_t>(th.size(), 2)));
                HIPCHK(c, hipMemcpy(c->red_head0_tiles, th.data(), sizeof(int) * th.size(), hipMemcpyHostToDevice));
                c->n_split_fold = srukf_gmw_build_fold_list(T, Tp, nullptr);
                std::vector<short> fl((size_t)4 * std::max(c->n_split_fold, 1));
                srukf_gmw_build_fold_list(T, Tp, fl.data());
                HIPCHK(c, srukf_dmalloc(&c->split_fold_list, sizeof(short) * fl.size()));
                HIPCHK(c, hipMemcpy(c->split_fold_list, fl.data(), sizeof(short) * fl.size(), hipMemcpyHostToDevice));
            }
            if (!c->shadowA) {
                HIPCHK(c, srukf_dmalloc(&c->shadowA, sizeof(double) * (size_t)np * np)); HIPCHK(c, srukf_dmalloc(&c->Utp, sizeof(double) * (size_t)c->d.mp * np));
                HIPCHK(c, hipMemsetAsync(c->Utp, 0, sizeof(double) * (size_t)c->d.mp * np, c->stream));
                HIPCHK(c, srukf_dmalloc(&c->P1, sizeof(double) * (size_t)c->d.mp * np));
                HIPCHK(c, hipMemsetAsync(c->P1, 0, sizeof(double) * (size_t)c->d.mp * np, c->stream));
                // gain fold of k_pxy2: sync words (zero between frames) and the per-landmark shares of the state update
                const size_t fw = sizeof(unsigned int) * (size_t)(srukf_fold_words(c->d.mp / 64, np / 64) + 2) + sizeof(unsigned long long) * FOLD_DBG_STAMPS;
                HIPCHK(c, srukf_dmalloc(&c->fold_sync, fw)); HIPCHK(c, hipMemsetAsync(c->fold_sync, 0, fw, c->stream));
                HIPCHK(c, srukf_dmalloc(&c->dxk, sizeof(double) * (size_t)(c->d.N > 0 ? c->d.N : 1) * np));
                HIPCHK(c, hipMemsetAsync(c->dxk, 0, sizeof(double) * (size_t)(c->d.N > 0 ? c->d.N : 1) * np, c->stream));
            }
            {
                // k_pxy2 ("table" mode): 64 x 64 tiles of the permuted product, K ends at the kept rows, long K ranges in two halves
                const int kr = (r + 15) & ~15;
                const int nt = srukf_pxy2_build_tiles(c->d.mp, np, kr, nullptr);
                std::vector<int> tl((size_t)4 * nt);
                srukf_pxy2_build_tiles(c->d.mp, np, kr, tl.data());
                if (c->pxy2_tiles) srukf_dfree_on(c->pxy2_tiles, c->stream);
                c->pxy2_tiles = nullptr; c->n_pxy2_tiles = nt;
                c->fold_robot_tiles = 0;                          // tile workgroups that read the robot columns (permuted positions r-4 .. r-1) of the permuted copy: gain fold
                for (int q = 0; q < nt; q++) if (tl[4 * q] >= 0 && (tl[4 * q + 1] == (r - 4) / 64 || tl[4 * q + 1] == (r - 1) / 64)) c->fold_robot_tiles++;
                HIPCHK(c, srukf_dmalloc(&c->pxy2_tiles, sizeof(int) * tl.size()));
                HIPCHK(c, hipMemcpy(c->pxy2_tiles, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice));
                {
                    // NullSkip: which directions are projected for all landmarks, which only for their own, which rows of Z the statistics walk
                    std::vector<int> dirs, nulls, rows;
                    const int Na = n + 5;
                    for (int i = 0; i < Na; i++) ((i >= n || i < 2 || iperm[i] < r) ? dirs : nulls).push_back(i);
                    rows.push_back(0);
                    for (int i : dirs) rows.push_back(1 + i);
                    for (int i : dirs) rows.push_back(1 + Na + i);
                    std::vector<int> all(dirs); all.insert(all.end(), nulls.begin(), nulls.end()); all.insert(all.end(), rows.begin(), rows.end());
                    if (c->nskip) srukf_dfree_on(c->nskip, c->stream);
                    c->nskip = nullptr;
                    HIPCHK(c, srukf_dmalloc(&c->nskip, sizeof(int) * all.size()));
                    HIPCHK(c, hipMemcpy(c->nskip, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice));
                    c->ns_full = (int)dirs.size(); c->ns_null = (int)nulls.size(); c->ns_rows = (int)rows.size();
                    // (directions 0 and 1 are projected for every landmark even when they are structurally null — the Si factor names their Z rows —
                    //  and the frame tail (k_rank_expand<2>) only does that for kept rows: such a state stays with k_project_table)
                    c->tail_ok = iperm[0] < r && iperm[1] < r;
                }
                c->pxy2_split_b0 = np;                            // first permuted column whose K range is cut in two
                for (int bt = 0; bt < np / 64; bt++) if (std::min(4 * (bt + 1), ((kr + 63) / 64) * 4) >= srukf_pxy2_split_groups()) { c->pxy2_split_b0 = 64 * bt; break; }
            }
            shadow_rebuild(c);
        }
    }
    if (was || c->red_r) drop_graphs(c);                          // the captured frames contain one or the other launch sequence
    return mixed_red_ensure(c);
}

// SRUKF_STORAGE_F32_MIXED in the rank-aware form: the fp32 copy of the kept rows, the task list of k_syrk32 over the pivoted panels (shape: r, Tp).  Called whenever the
// null set or the storage mode changes — never inside a capture.
int mixed_red_ensure(srukf_ctx* c)
{
    if (c->storage != SRUKF_STORAGE_F32_MIXED || c->red_r <= 0 || !c->shadowA) return SRUKF_OK;
    const int np = c->d.np, mp = c->d.mp;
    if (!c->A32) HIPCHK(c, srukf_dmalloc(&c->A32, sizeof(float) * (size_t)np * np));
    if (c->mxr_for_r == c->red_r && c->mxr_tasks) return SRUKF_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (void* b : { (void*)c->mxr_part, c->mxr_tasks, c->mxr_tiles, (void*)c->mxr_f64_tiles, (void*)c->mxr_xt }) if (b) srukf_dfree_on(b, c->stream);
    c->mxr_part = nullptr; c->mxr_tasks = nullptr; c->mxr_tiles = nullptr; c->mxr_f64_tiles = nullptr; c->mxr_xt = nullptr;
    c->mxr_krows = std::min(np, srukf_mixed_krows(c->red_r));
    int ntiles = 0;
    const int ntasks = srukf_mixed_build_tasks_red(np, mp, c->mxr_krows, 64 * c->red_Tp, nullptr, nullptr, &ntiles);
    std::vector<short> tk((size_t)4 * ntasks); std::vector<int> tl((size_t)2 * ntiles);
    srukf_mixed_build_tasks_red(np, mp, c->mxr_krows, 64 * c->red_Tp, tk.data(), tl.data(), &ntiles);
    HIPCHK(c, srukf_dmalloc(&c->mxr_part, srukf_mixed_part_bytes(ntasks)));
    HIPCHK(c, srukf_dmalloc(&c->mxr_tasks, sizeof(short) * tk.size()));
    HIPCHK(c, srukf_dmalloc(&c->mxr_tiles, sizeof(int) * tl.size()));
    HIPCHK(c, hipMemcpy(c->mxr_tasks, tk.data(), sizeof(short) * tk.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->mxr_tiles, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice));
    c->mxr_ntasks = ntasks; c->mxr_ntiles = ntiles; c->mxr_for_r = c->red_r;
    c->mxr_ktot = c->mxr_krows + mp; c->mxr_xt_stride = (size_t)np * c->mxr_ktot;
    HIPCHK(c, srukf_dmalloc(&c->mxr_xt, sizeof(unsigned short) * 3 * c->mxr_xt_stride));
    HIPCHK(c, hipMemsetAsync(c->mxr_xt, 0, sizeof(unsigned short) * 3 * c->mxr_xt_stride, c->stream));
    {
        // the FP64 tiles (32 x 32, permuted order, upper triangle, rows of the pivoted panels): tile row / column 0 (the shared anchor: permuted positions 0 .. 2) and the
        // one or two tile rows / columns of the robot block (r-4 .. r-1)
        const int T32 = np / 32, rows32 = 2 * c->red_Tp, q0 = 0, q1 = (c->red_r - 4) / 32, q2 = (c->red_r - 1) / 32;
        std::vector<int> tl;
        for (int I = 0; I < rows32 && I < T32; I++)
            for (int J = I; J < T32; J++)
                if (I == q0 || I == q1 || I == q2 || J == q1 || J == q2) { tl.push_back(I); tl.push_back(J); }
        HIPCHK(c, srukf_dmalloc(&c->mxr_f64_tiles, sizeof(int) * (tl.size() + 2)));
        HIPCHK(c, hipMemcpy(c->mxr_f64_tiles, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice));
        c->mxr_n_f64_tiles = (int)tl.size() / 2;
    }
    drop_graphs(c);
    return SRUKF_OK;
}

int read_fs(srukf_ctx* c)
{
    HIPCHK(c, hipMemcpyAsync(c->hfs, c->fs, sizeof(FrameScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return read_fs_host(c);
}
// the frame scalars are in *c->hfs already (the step-wise fast path has a kernel write them there with the robot view)
int read_fs_host(srukf_ctx* c)
{
    if (c->hfs->clamp_rows > 0 && c->hfs->clamp_frame == 0x7fffffff) c->hfs->clamp_frame = c->hfs->frame - 1;   // the run's last frame
    if (c->hfs->gmw_aborts > 0 && c->gmw_shared != 2) {
        // a persistent launch did not get all its workgroups onto the GPU in time (somebody else is using it, or the two launches of a split-form pair were not
        // run side by side): the flagged frame is repeated on the exact path like a clamp frame, and the filter steps down ONE tier — from the split form to the
        // memory-tile instance of k_gmw_persist (one launch, no second hardware queue needed), from any single persistent launch to one launch per panel
        const GmwPlan& gp = c->red_r > 0 ? c->gplan_red : c->gplan;
        if (split_form(c, gp, true)) { c->split_off = true; c->err = "a split-form factorisation pair was abandoned: this filter continues with the memory-tile persistent launch (srukf_debug_get \"split_off\")"; }
        else { c->gmw_shared = 2; c->err = "a persistent factorisation launch was abandoned: this filter continues with one launch per panel (srukf_debug_get \"gmw_shared\" = 2)"; }
        drop_graphs(c);
    }
    return SRUKF_OK;
}

void drop_graphs(srukf_ctx* c)
{
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->graph8_exec) { hipGraphExecDestroy(c->graph8_exec); c->graph8_exec = nullptr; }
    if (c->graph8) { hipGraphDestroy(c->graph8); c->graph8 = nullptr; }
    if (c->graphN_exec) { hipGraphExecDestroy(c->graphN_exec); c->graphN_exec = nullptr; }
    if (c->graphN) { hipGraphDestroy(c->graphN); c->graphN = nullptr; }
    c->graphN_frames = 0;
}

// shared: 0 exclusive, 1 shared (tenants persistent launches at a time), 2 one launch per panel
int set_shared(srukf_ctx* c, int shared, int tenants)
{
    if (tenants < 2) tenants = 2;
    if (shared == c->gmw_shared && (shared != 1 || tenants == c->shared_tenants)) return SRUKF_OK;
    const int was = plan_tenants(c);
    step_invalidate(c);
    c->gmw_shared = shared;
    if (shared == 1) c->shared_tenants = tenants;
    drop_graphs(c);
    if (plan_tenants(c) != was) {                              // the persistent launches keep to half the CUs / may use all of them again
        gmw_plan_destroy(c->gplan, c->stream);
        const int rc = gmw_plan_create(c->gplan, c->d.np, c->stream, 0, plan_tenants(c));
        if (rc) { c->err = "set_exclusive: persistent GMW resources: allocation failed"; return rc; }
        split_ensure(c, c->gplan);
        return update_null_set(c);                             // the rank-aware plan with the same limit
    }
    return SRUKF_OK;
}

// srukf_debug_set(ctx, "fused_motion", 0): the replay keeps k_motion and k_project as two launches (A/B runs)
void replay_one_frame(srukf_ctx* c)
{
    const int mode = replay_motion_mode(c);
    const bool fuse = replay_fuse_mode(c);
    if (fuse) {
        // the previous frame's tail (or, for a run's first frame, run_frames_async) projected this frame; its motion reduction rides on k_pxy2
        c->xr1_pending = true;
        seq_gain(c, nullptr, nullptr, true, true, true, true, true);
    } else if (mode) {
        seq_predict_fused(c, mode);
        seq_gain(c, nullptr, nullptr, true, true, mode == 2 && c->dbg.pxy2);
    } else {
        seq_predict_motion(c, nullptr);
        seq_predict_measurement(c, true);
        seq_gain(c, nullptr, nullptr, true);
    }
    seq_refactor(c, 0, c->d.mp, false, false, false, true, mode == 2, fuse);
}

}  // namespace srukf_impl

static int capture_frames(srukf_ctx* c, int nframes, hipGraph_t* g, hipGraphExec_t* ge)
{
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int q = 0; q < nframes; q++) replay_one_frame(c);
    const hipError_t launch_err = hipGetLastError();            // a failed launch inside the capture must not leave the stream capturing
    HIPCHK(c, hipStreamEndCapture(c->stream, g));
    HIPCHK(c, launch_err);
    HIPCHK(c, hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
    return SRUKF_OK;
}

namespace srukf_impl {

// the null rows are canonical from here on (a rank-aware frame tail has been issued): captured frames of the other launch sequence are stale
// fp64 storage only (the float-stored modes keep their rounded copies in step elsewhere); "null_canon" 0: the first frame behind such a factor runs the launch sequence that
// reads the rows as they are and its tail rewrites them (rounds 2 - 5)
void canonicalize_null_rows(srukf_ctx* c)
{
    if (!c->dbg.null_canon || c->red_r <= 0 || c->null_canonical || c->storage != SRUKF_STORAGE_F64 || !c->red_perm) return;
    srukf_launch_rank_canon(c->stream, c->d.n, c->d.np, c->red_r, sqrt(c->p.epsilon), c->red_perm, c->S);
    c->null_canonical = true;
    drop_graphs(c);
}

void set_null_canonical(srukf_ctx* c)
{
    if (c->red_r > 0 && !c->null_canonical) { c->null_canonical = true; drop_graphs(c); }
}

}  // namespace srukf_impl

// One staged frame (index `frame`) through the path that checks the theta clamp on the host and repeats the
// refactorisation column by column when the reference's third pivot candidate would have won — what srukf_update does,
// with the staged inputs.  traj_row: device pointer of this frame's trajectory row, or null.
static int run_staged_frame_exact(srukf_ctx* c, int frame, double* traj_row)
{
    const KDims& d = c->d;
    c->exact_frames++;
    const bool timing = g_dbg_timing.load() != 0;              // (srukf_debug_set(0, "timing", 1): where a flagged frame's milliseconds go, on stderr)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char* what) { if (timing) { hipStreamSynchronize(c->stream); auto t1 = now(); fprintf(stderr, "[exact frame] %s %.0f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count()); t0 = t1; } };
    double* tb = traj_row ? traj_row - (size_t)8 * frame : nullptr;
    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, frame, 1);
    hipLaunchKernelGGL(k_set_traj, dim3(1), dim3(1), 0, c->stream, c->fs, tb);
    seq_predict_motion(c, nullptr);
    seq_predict_measurement(c, true);
    lap("predict");
    seq_gain(c, nullptr, nullptr, true);
    lap("gain");
    seq_refactor(c, 0, d.mp, false, true, false, false);
    int rc = read_fs(c); if (rc) return rc;
    lap("blocked refactor");
    if (c->hfs->clamp_rows > 0) {
        hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, frame, 1);
        hipLaunchKernelGGL(k_set_traj, dim3(1), dim3(1), 0, c->stream, c->fs, tb);
        hipLaunchKernelGGL(k_refactor_reset, dim3((d.np + 255) / 256), dim3(256), 0, c->stream, d.np, c->theta, c->fs, 0);
        HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
        ProfScope ps(c, KC_GMW_COL, 0, 0);
        exact_path(c, c->G, c->S);
        quantize_state(c);
        lap("exact path");
        rc = update_null_set(c); if (rc) return rc;      // the null set is re-derived from the exact factor (see srukf_update)
        lap("null set");
    } else if (c->storage != SRUKF_STORAGE_F32_MIXED || storage_f32_like(c)) set_null_canonical(c);
    srukf_launch_traj(c->stream, d, c->X, c->S, c->fs, nullptr, 1);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

extern "C" {

int srukf_set_exclusive(srukf_ctx* c, int exclusive)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));                         // the plans below size themselves on the CURRENT device's CU count
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int shared = exclusive == SRUKF_GPU_SHARED ? 1 : exclusive == SRUKF_GPU_SHARED_PER_PANEL ? 2 : 0;
    return set_shared(c, shared, g_dbg_shared_tenants.load());
}

int srukf_stage_sequence(srukf_ctx* c, int F, const double* odo, const double* z, const int* matched)
{
    if (!c || F < 1 || !odo || !z || !matched) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->d.N;
    HIPCHK(c, hipStreamSynchronize(c->stream));             // frames in flight may still read the staged inputs
    if (c->odo_seq) { srukf_dfree(c->odo_seq); srukf_dfree(c->z_seq); srukf_dfree(c->m_seq); c->odo_seq = nullptr; c->z_seq = nullptr; c->m_seq = nullptr; }
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->graph8_exec) { hipGraphExecDestroy(c->graph8_exec); c->graph8_exec = nullptr; }
    if (c->graph8) { hipGraphDestroy(c->graph8); c->graph8 = nullptr; }
    if (c->graphN_exec) { hipGraphExecDestroy(c->graphN_exec); c->graphN_exec = nullptr; }
    if (c->graphN) { hipGraphDestroy(c->graphN); c->graphN = nullptr; }
    c->graphN_frames = 0;
    HIPCHK(c, srukf_dmalloc((void**)&c->odo_seq, sizeof(double) * 3 * (F + 1)));
    HIPCHK(c, srukf_dmalloc((void**)&c->z_seq, sizeof(double) * (size_t)F * 2 * N));
    HIPCHK(c, srukf_dmalloc((void**)&c->m_seq, sizeof(int) * (size_t)F * N));
    HIPCHK(c, hipMemcpy(c->odo_seq, odo, sizeof(double) * 3 * (F + 1), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->z_seq, z, sizeof(double) * (size_t)F * 2 * N, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->m_seq, matched, sizeof(int) * (size_t)F * N, hipMemcpyHostToDevice));
    c->seqF = F;
    hipLaunchKernelGGL(k_set_seq, dim3(1), dim3(1), 0, c->stream, c->fs, c->odo_seq, F, c->p.a1, c->p.a2, c->p.a3, c->p.a4);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SRUKF_OK;
}

// A block of `count` staged frames as ONE captured graph for the next srukf_run_frames_async(ctx, *, count, ...) calls (the default
// is graphs of 8 frames + single frames; between two graph launches the device idles for ~10 us, which shows in short blocks).
// Nothing runs; the graph is dropped with the others whenever the launch sequence changes.  count <= 512.
int srukf_prepare_frames(srukf_ctx* c, int count)
{
    if (!c || count < 1 || count > 512) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->use_graph || (c->graphN_exec && c->graphN_frames == count)) return SRUKF_OK;
    if (c->graphN_exec) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipGraphExecDestroy(c->graphN_exec); c->graphN_exec = nullptr; }   // a launch of the old one may still be in flight
    if (c->graphN) { hipGraphDestroy(c->graphN); c->graphN = nullptr; }
    c->graphN_frames = 0;
    const int rc = capture_frames(c, count, &c->graphN, &c->graphN_exec);
    if (rc) return rc;
    c->graphN_frames = count;
    return SRUKF_OK;
}

int srukf_run_frames_async(srukf_ctx* c, int first, int count, int mode, double* d_traj)
{
    if (!c || first < 0 || count < 1) return SRUKF_ERR_BAD_ARG;
    if (!c->odo_seq || first + count > c->seqF) { c->err = "frames outside the staged sequence"; return SRUKF_ERR_DIM_MISMATCH; }
    if (mode != SRUKF_UPDATE_BATCHED) { c->err = "run_frames_async supports BATCHED only (SEQUENTIAL needs a host check per column)"; return SRUKF_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    const KDims& d = c->d;
    step_commit_motion(c); step_state_replaced(c);
    if (c->fs_seq_step) {                                      // the step-wise fast path pointed the frame scalars at its own three poses
        hipLaunchKernelGGL(k_set_seq, dim3(1), dim3(1), 0, c->stream, c->fs, c->odo_seq, c->seqF, c->p.a1, c->p.a2, c->p.a3, c->p.a4);
        c->fs_seq_step = false;
    }
    // traj rows are indexed by the absolute frame counter; offset so that frame `first` lands in row 0
    double* traj = d_traj ? d_traj - (size_t)8 * first : nullptr;
    int clear = c->async_pending ? 0 : 1;
    if (c->red_r > 0 && !c->null_canonical) {
        // A state that arrived from outside with structurally null rows that are not (yet) sqrt(EPSILON) e_k — zero rows after the joint initialisation, say: the
        // run's first frame takes the launch sequence that reads those rows as they are (k_project_motion, k_pxy; replay_motion_mode), eagerly; its tail writes the
        // canonical rows, and the frames behind it run the default sequence.
        hipLaunchKernelGGL(k_set_run, dim3(1), dim3(1), 0, c->stream, c->fs, first, clear, traj);
        replay_one_frame(c);
        set_null_canonical(c);
        c->async_pending = true; c->phase = 0; clear = 0;
        first += 1; count -= 1;
        HIPCHK(c, hipGetLastError());
        if (count == 0) return SRUKF_OK;
    }
    hipLaunchKernelGGL(k_set_run, dim3(1), dim3(1), 0, c->stream, c->fs, first, clear, traj);
    // "table" mode: the first frame's table of robot poses (the frames after it get theirs from their predecessor's tail)
    if (replay_motion_mode(c) == 2) srukf_launch_sigr_rows(c->stream, d, c->w, c->X, c->S, c->sigR, c->fs, c->red_iperm, c->red_r);
    // "fused tail" mode: ... and the first frame's projection (k_project_table); every later frame is projected by its predecessor's tail
    if (replay_fuse_mode(c)) seq_predict_fused(c, 2);
    if (c->use_graph && !c->profiling) {
        // every per-frame argument lives in HBM (frame counter, staged inputs, trajectory base), so ONE
        // captured frame replays for all frames: the 45 launches cost one hipGraphLaunch on the host
        if (!c->graph_exec) {
            int rc = capture_frames(c, 1, &c->graph, &c->graph_exec); if (rc) return rc;
            // and a graph of SRUKF_GRAPH_FRAMES consecutive frames: one host launch per 8 frames keeps the host
            // ahead of the device when several filters share one host thread
            rc = capture_frames(c, SRUKF_GRAPH_FRAMES, &c->graph8, &c->graph8_exec); if (rc) return rc;
        }
        if (c->graphN_exec && c->graphN_frames == count) HIPCHK(c, hipGraphLaunch(c->graphN_exec, c->stream));   // srukf_prepare_frames
        else {
            int f = 0;
            for (; f + SRUKF_GRAPH_FRAMES <= count; f += SRUKF_GRAPH_FRAMES) HIPCHK(c, hipGraphLaunch(c->graph8_exec, c->stream));
            for (; f < count; f++) HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
        }
    } else {
        for (int f = 0; f < count; f++) replay_one_frame(c);
    }
    // fp32 storage in "fused tail" mode: S and X are rounded as they are written; the float copies (srukf_get_state_f32) once per run
    if (storage_f32_like(c) && replay_fuse_mode(c)) quantize_state(c);
    c->async_pending = true;
    c->phase = 0;
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

// Synchronous form.  Unlike the asynchronous replay it never returns SRUKF_ERR_CLAMP_PENDING: the state before the
// block is kept, and when a frame is flagged (theta clamp of the modified Cholesky, SLAM.cpp:2279-2285, or an abandoned
// persistent launch) the block is rewound to that state, the frames before the flagged one are replayed, the flagged
// frame runs on the exact path, and the replay continues behind it.
int srukf_run_frames(srukf_ctx* c, int first, int count, int mode, double* traj_host)
{
    if (!c || count < 1) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t np = c->d.np;
    if (!c->ckS) {
        if (srukf_dmalloc((void**)&c->ckS, sizeof(double) * np * np) != hipSuccess || srukf_dmalloc((void**)&c->ckX, sizeof(double) * np) != hipSuccess) {
            c->err = "run_frames: out of device memory (checkpoint)"; return SRUKF_ERR_NOMEM;
        }
    }
    double* dt = nullptr;
    HIPCHK(c, srukf_dmalloc((void**)&dt, sizeof(double) * 8 * (size_t)count));
    bool ck_canon = c->null_canonical;
    auto checkpoint = [&](bool save) {
        hipMemcpyAsync(save ? c->ckS : c->S, save ? c->S : c->ckS, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->stream);
        hipMemcpyAsync(save ? c->ckX : c->X, save ? c->X : c->ckX, sizeof(double) * np, hipMemcpyDeviceToDevice, c->stream);
        if (save) ck_canon = c->null_canonical;
        else { if (c->null_canonical != ck_canon) { c->null_canonical = ck_canon; drop_graphs(c); } quantize_state(c); shadow_rebuild(c); }
    };
    int rc = SRUKF_OK, done = 0;
    const bool timing = g_dbg_timing.load() != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) { hipStreamSynchronize(c->stream); auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[run_frames] %s %.0f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count()); t0 = t1; } };
    while (done < count) {
        checkpoint(true);
        rc = srukf_run_frames_async(c, first + done, count - done, mode, dt + (size_t)8 * done);
        if (rc == SRUKF_OK) rc = srukf_synchronize(c);
        if (rc != SRUKF_ERR_CLAMP_PENDING) break;
        lap("flagged attempt");
        const int fc = c->clamp_frame_host;                               // absolute index of the first flagged frame
        if (fc < first + done || fc >= first + count) { c->err = "run_frames: flagged frame outside the block"; rc = SRUKF_ERR_HIP; break; }
        checkpoint(false);
        lap("rewind");
        const int good = fc - (first + done);
        if (good > 0) {
            rc = srukf_run_frames_async(c, first + done, good, mode, dt + (size_t)8 * done);
            if (rc == SRUKF_OK) rc = srukf_synchronize(c);
            if (rc != SRUKF_OK) break;                                    // (the same frames passed a moment ago)
        }
        rc = run_staged_frame_exact(c, fc, dt + (size_t)8 * (fc - first));
        if (rc != SRUKF_OK) break;
        done = fc - first + 1;
    }
    if (traj_host && rc == SRUKF_OK) hipMemcpy(traj_host, dt, sizeof(double) * 8 * (size_t)count, hipMemcpyDeviceToHost);
    srukf_dfree(dt);
    return rc;
}

// What the last SRUKF_ERR_CLAMP_PENDING of srukf_synchronize was about: the first flagged staged frame (frames before it
// are valid) and the first flagged pivot row.  -1 / -1 if there was none.
int srukf_clamp_info(srukf_ctx* c, int* frame, int* row)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    if (frame) *frame = c->clamp_frame_host;
    if (row) *row = c->clamp_row_host;
    return SRUKF_OK;
}

// Rank-aware refactorisation on / off (default on); re-derives the null set from the current state.
int srukf_set_rank_aware(srukf_ctx* c, int on)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rank_aware = on ? 1 : 0;
    return update_null_set(c);
}

// How many of the n pivots the refactorisation skips (0: the rank-aware form is off or found nothing to skip).
int srukf_null_directions(srukf_ctx* c) { return c ? (c->red_r > 0 ? c->d.n - c->red_r : 0) : SRUKF_ERR_BAD_ARG; }

int srukf_synchronize(srukf_ctx* c)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (c->async_pending) {
        c->async_pending = false;
        int rc = read_fs(c); if (rc) return rc;
        if (c->hfs->clamp_rows > 0) {
            c->clamp_frame_host = c->hfs->clamp_frame; c->clamp_row_host = c->hfs->clamp_first;
            char b[220]; snprintf(b, sizeof b, "GMW theta clamp active on %d pivot rows (first row %d), first in staged frame %d, during async frames%s", c->hfs->clamp_rows, c->hfs->clamp_first, c->hfs->clamp_frame,
                                 c->hfs->gmw_aborts > 0 ? " (a persistent factorisation launch was abandoned: the GPU is shared; see srukf_set_exclusive)" : "");
            c->err = b;
            return SRUKF_ERR_CLAMP_PENDING;
        }
    }
    return SRUKF_OK;
}

}  // extern "C"
