// srukf_api.hip — C-ABI (include/srukf.h) over the gfx950 kernels: context, HBM buffers,
// per-frame launch sequences, profiling.  No CPU fallback: every numeric result is produced by
// the kernels in srukf_predict.hip / srukf_factor.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <cstddef>
#include <vector>
#include <algorithm>
#include <utility>
#include <atomic>
#include <mutex>
#include <map>
#include "srukf_device.h"
#include "srukf_rank.h"

extern "C" {
void srukf_launch_motion(hipStream_t, KDims, KWeights, srukf_params, double*, double*, double*, double*, FrameScalars*, const double*, const double*, RankArgs);
void srukf_launch_project(hipStream_t, KDims, KWeights, srukf_params, const double*, const double*, const double*, double*, double*, const FrameScalars*);
void srukf_launch_meas_stats(hipStream_t, KDims, KWeights, const double*, const double*, const double*, double*, double*, double*, int*, double*);
int srukf_meas_part_doubles(int);
void srukf_launch_gain(hipStream_t, KDims, KWeights, double*, const double*, const double*, const int*, const double*, const double*,
                       const double*, const int*, const int*, FrameScalars*, double*, double*, const double*, RankArgs, const double*, double*,
                       const double*, int, const double*, double, const double*, int);
void srukf_launch_project_motion(hipStream_t, KDims, KWeights, srukf_params, double*, double*, double*, double*, double*, double*, FrameScalars*, RankArgs);
int srukf_gain_part_doubles(int);
void srukf_launch_traj(hipStream_t, KDims, const double*, const double*, FrameScalars*, double*, int);
void srukf_launch_block_cov(hipStream_t, KDims, const double*, int, int, double*);
void srukf_launch_project_points(hipStream_t, srukf_params, int, const double*, const double*, const double*, const double*, double*);
void srukf_launch_pxy(hipStream_t, KDims, const double*, const double*, double*, const void*, int, KWeights, MeasArgs);
void srukf_launch_pxy2(hipStream_t, KDims, const double*, const double*, double*, double*, const void*, int, int, KWeights, MeasArgs);
int srukf_pxy2_build_tiles(int mp, int np, int kr, int* out);
int srukf_pxy2_split_groups(void);
void srukf_launch_syrk(hipStream_t, KDims, const double*, const double*, int, int, double*, FrameScalars*, const void*, int, const double*, double*, RankArgs, const double*);
void srukf_launch_gmw_step64(hipStream_t, int, int, int, double, double*, const void*, void*, double*, double*, const FrameScalars*);
int srukf_gmw_panel_bytes(void);
int srukf_gmw_sync_bytes(int T);
int srukf_gmw_build_tiles(int T, int Tp, short* out);
int srukf_gmw_persist_workers(int T, int Tp, int max_workers);
void srukf_launch_gmw_persist(hipStream_t, int, int, double, double*, void*, double*, double*, void*, const void*, int, int, void*, const double*, const double*, int, int, int, int, int);
void srukf_launch_gmw_persist_head(hipStream_t, int, int, double, double*, void*, double*, double*, void*, const void*, int, int, void*, const double*, const double*, int, int, int, int, int, const HeadArgs*);
void srukf_launch_gmw_split(hipStream_t, hipStream_t, int, int, double, double*, void*, double*, double*, void*, const void*, int, void*, int, int, double*, double*, int);
void srukf_launch_gmw_split_alone(hipStream_t, int, int, int, double, double*, void*, double*, double*, void*, const void*, int, void*, int, int, double*, double*);
void srukf_launch_row_energy(hipStream_t, int, int, const double*, double*);
void srukf_launch_rank_diag(hipStream_t, int, int, const double*, const int*, double*);
void srukf_launch_rank_expand(hipStream_t, int, int, int, double, const double*, const double*, const int*, const int*, const double*, void*, const double*, int, double*, double*, double*, double, int, KDims, KWeights, srukf_params, double*, double*, int);
void srukf_launch_project_table(hipStream_t, KDims, KWeights, srukf_params, double*, double*, double*, double*, double*, double*, FrameScalars*, RankArgs, NullSkip);
void srukf_launch_sigr_rows(hipStream_t, KDims, KWeights, const double*, const double*, double*, const FrameScalars*, const int*, int);
void srukf_launch_rank_shadow(hipStream_t, int, int, int, const double*, const int*, double*);
void srukf_launch_rank_round(hipStream_t, int, int, double*);
void srukf_launch_syrk_own(hipStream_t, int, int, const double*, const double*, int, int, int, double*, void*, const void*, int, int);
int srukf_gmw_register_form(int, int, int, int);
int srukf_pxy2_b_per(int, int);
void srukf_launch_pxy2_b(hipStream_t, KDims, const void*, int, const void*, int, int, KWeights, int);
void srukf_launch_gain_b(hipStream_t, KDims, KWeights, const void*, int, int, double);
void srukf_launch_syrk_b(hipStream_t, KDims, const void*, int, const void*, int, int, int, int);
void srukf_launch_syrk_own_b(hipStream_t, int, int, const void*, int, int, int, int, const void*, int, int);
void srukf_launch_gmw_step64_b(hipStream_t, int, int, int, double, const void*, int, int, int);
void srukf_launch_gmw_pivslab_b(hipStream_t, int, int, int, double, const void*, int, int);
void srukf_launch_gmw_trail_b(hipStream_t, int, int, const void*, int, int);
void srukf_launch_rank_expand_b(hipStream_t, int, int, int, double, const void*, int, double, KDims, KWeights, srukf_params);
int srukf_gmw_head_rows(void);
int srukf_gmw_head_extra_diag(void);
void srukf_launch_gmw_check(hipStream_t, int, int, const double*, const double*, FrameScalars*, const double*, int, double*);
void srukf_launch_gmw_col(hipStream_t, int, int, int, double, const double*, double*, double*, unsigned long long*, FrameScalars*, double*);
void srukf_launch_gmw_stats(hipStream_t, int, int, const double*, FrameScalars*);
void srukf_launch_landmarks_cartesian(hipStream_t, KDims, const double*, const double*, double*, double*);
void srukf_launch_aug_map(hipStream_t, srukf_params, int, int, int, int, double, const double*, const double*, const double*, double*);
void srukf_launch_aug_x(hipStream_t, int, int, int, double, double, const double*, const double*, const int*, double*, double*, int, int);
void srukf_launch_aug_build(hipStream_t, int, int, int, int, double, double, const double*, const double*, const double*, double*, int, int, int);
void srukf_launch_gram(hipStream_t, int, int, const double*, double*);
void srukf_launch_warp_patch(hipStream_t, KDims, srukf_params, const double*, const double*, const double*, const double*, const double*, const double*,
                             const unsigned char*, const int*, unsigned char*);
void srukf_launch_associate(hipStream_t, KDims, srukf_params, const unsigned char*, const double*, const double*, const int*, const int*,
                            const unsigned char*, double*, int*, double*);
int srukf_mixed_build_tasks(int np, int ue, short* out_tasks, int* out_tiles, int* ntiles);
size_t srukf_mixed_part_bytes(int ntasks);
void srukf_launch_cvt_f32(hipStream_t, size_t, const double*, float*);
void srukf_launch_cvt_robot_cols(hipStream_t, int, int, const double*, float*);
void srukf_launch_gain_dx(hipStream_t, int, int, const double*, double*, const double*);
void srukf_launch_syrk32(hipStream_t, int, int, int, const float*, const float*, const void*, int, const void*, int, float*, double*, void*);
int srukf_app_patch_stride(void);
int srukf_app_tmpl_stride(void);
}

// ---- device memory: the stream-ordered pool of the device instead of hipMalloc / hipFree ------------------------------
// A context is rebuilt whenever the map changes size (srukf_add_landmarks / srukf_delete_landmark): ~25 buffers freed and
// ~25 allocated.  hipMalloc / hipFree go to the driver every time (and hipFree synchronises the whole device): 12.6 ms per
// augmentation at N = 200, almost all of it there.  The default memory pool keeps freed blocks (release threshold raised
// to "never") and hands them out again in microseconds.  Allocation is made visible to every stream by synchronising the
// null stream it is ordered on; every free below happens after the streams that used the block have been synchronised.
static void srukf_pool_init()
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
static hipError_t srukf_dmalloc_raw(void** p, size_t bytes)
{
    srukf_pool_init();
    hipError_t e = hipMallocAsync(p, bytes ? bytes : 8, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}
template <class T> static hipError_t srukf_dmalloc(T** p, size_t bytes) { return srukf_dmalloc_raw((void**)p, bytes); }
static hipError_t srukf_dfree(void* p) { return p ? hipFreeAsync(p, nullptr) : hipSuccess; }
// the same, ordered on a context's own stream (no synchronisation: everything that touches the block is on that stream)
template <class T> static hipError_t srukf_dmalloc_on(T** p, size_t bytes, hipStream_t st) { srukf_pool_init(); return hipMallocAsync((void**)p, bytes ? bytes : 8, st); }
static hipError_t srukf_dfree_on(void* p, hipStream_t st) { return p ? hipFreeAsync(p, st) : hipSuccess; }

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
__global__ void k_set_step(FrameScalars* fs, const double* odo, int seqF, double a1, double a2, double a3, double a4, int fresh)
{
    fs->odo_seq = odo; fs->seqF = seqF;
    fs->a[0] = a1; fs->a[1] = a2; fs->a[2] = a3; fs->a[3] = a4;
    fs->frame = 0;
    fs->traj_base = nullptr;
    fs->stat_count = 0;
    for (int q = 0; q < SRUKF_STAT_GROUPS; q++) fs->stat_cnt[q] = 0;
    fs->clamp_rows = 0; fs->clamp_first = 0x7fffffff; fs->clamp_frame = 0x7fffffff; fs->frozen = 0; fs->gmw_aborts = 0;
    if (fresh) { srukf_prepare_control(fs); fs->const_rows_ok = 0; fs->const_rows_pending = 0; }
}
__global__ void k_set_frame_control(FrameScalars* fs) { srukf_prepare_control(fs); }
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

// ---- NEED_REORDER helpers (GSLCholeskyUpdate, SLAM.cpp:2122-2138) -------------------------------
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

#define SRUKF_GRAPH_FRAMES 8
static thread_local std::string g_create_error;
static thread_local double* g_spare_stage = nullptr;       // one pinned staging buffer handed from a destroyed context to the next one
static thread_local size_t g_spare_stage_bytes = 0;
// ... and one verified side stream of the split form (with its events): a map change rebuilds the context on the SAME filter stream, and probing candidates again
// (up to eight streams, two launches and three synchronisations each) would sit on the latency-critical path of every srukf_add_landmarks / srukf_delete_landmark
struct SpareSide { int device = -1; hipStream_t main = nullptr, side = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static thread_local SpareSide g_spare_side;
static void spare_side_drop()
{
    if (g_spare_side.side) { hipStreamSynchronize(g_spare_side.side); hipStreamDestroy(g_spare_side.side); hipEventDestroy(g_spare_side.fork); hipEventDestroy(g_spare_side.join); }
    g_spare_side = SpareSide();
}

enum KClass { KC_MOTION = 0, KC_PROJECT, KC_STATS, KC_PXY, KC_GAIN, KC_SYRK, KC_GMW_TRAIL, KC_GMW_PERSIST, KC_GMW_CHECK,
              KC_GMW_COL, KC_RANK_EXPAND, KC_PROJECT_MOTION, KC_PROJECT_TABLE, KC_PXY2, KC_MISC, KC_COUNT };
static const char* kclass_name[KC_COUNT] = { "k_motion", "k_project", "k_meas_stats", "k_pxy", "k_gain", "k_syrk",
                                             "k_gmw_step64", "k_gmw_persist", "k_gmw_check", "k_gmw_col", "k_rank_expand", "k_project_motion", "k_project_table", "k_pxy2",
                                             "misc" };

struct ProfEvent { hipEvent_t a, b; int kc; };

// ---- persistent GMW launch (k_gmw_persist): per-matrix-size resources --------------------------------
// nreal: the tiles that hold values (ntiles minus the T - Tp pass-on tiles of the rank-aware form, which ride as a register-free third slot of the first workers)
struct GmwPlan { void* pans = nullptr; void* sync = nullptr; void* tiles = nullptr; int ntiles = 0, nreal = 0, T = 0, Tp = 0, workers = -1, tenants = 1, cus = 0; };
static void gmw_plan_destroy(GmwPlan& g, hipStream_t st = nullptr)
{
    if (g.pans) srukf_dfree_on(g.pans, st);
    if (g.sync) srukf_dfree_on(g.sync, st);
    if (g.tiles) srukf_dfree_on(g.tiles, st);
    g = GmwPlan();
}
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
// srukf_debug_set "batch_wide" 0: srukf_run_frames_batch never takes the batched launches (one stream per filter, persistent launches behind the gate: round 3's form)
static std::atomic<int> g_dbg_batch_wide{1};
static std::atomic<int> g_dbg_batch_groups{0};
static std::atomic<int> g_dbg_batch_split{1};                 // "batch_split" 0: one k_gmw_step64_b launch per panel (every tile recomputes its slabs) instead of slabs + plain updates                // "batch_groups": groups the batched filters are cut into (0: two from eight filters on)
static int gmw_plan_create(GmwPlan& g, int np, hipStream_t st, int Tp = 0, int tenants = 1)
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
// A/B switches of srukf_debug_set (process-wide; measurement and test knobs, all 1 in the product):
//   gmw_persist  1 = one persistent launch per factorisation, 0 = one launch per 64-row panel
//   gmw_fused    0 = the persistent launch reads every tile from G (k_syrk computes all of them)
//   rank_fused   0 = the rank-aware form always goes through the full k_syrk + permutation pass
//   rank_fold    0 = the owners never form their tiles themselves in the rank-aware replay (k_syrk over all kept rows instead)
//   rank_aware   0 = no context looks for structurally null directions (per filter: srukf_set_rank_aware)
//   graphs       0 = contexts created from now on launch eagerly (profilers with --pmc; per filter: key "use_graph")
// (atomics: another thread's context may be launching while a switch is set; a switch applies to whatever is built or captured afterwards)
static std::atomic<int> g_dbg_gmw_persist{1}, g_dbg_gmw_fused{1}, g_dbg_rank_fused{1}, g_dbg_rank_fold{1}, g_dbg_rank_aware{1}, g_dbg_graphs{1};
static int gmw_persist_mode() { return g_dbg_gmw_persist; }

struct srukf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    srukf_params p;
    KDims d;
    KWeights w;
    // HBM buffers
    double *X = nullptr, *S = nullptr, *G = nullptr, *Gbak = nullptr, *Wf = nullptr;
    double *sigR = nullptr, *Cmat = nullptr, *Z = nullptr, *DZ = nullptr, *Ut = nullptr;
    double *h = nullptr, *Si = nullptr, *PxyR = nullptr, *D = nullptr;
    double *zcur = nullptr, *odocur = nullptr, *small = nullptr, *mpart = nullptr, *dxp = nullptr;
    int *vis = nullptr, *mcur = nullptr;
    unsigned long long* theta = nullptr;
    bool dx_pending = false;               // k_gain left slice partials of dX that the next k_syrk must add to X
    bool xr1_pending = false;              // replay path: the robot mean after the motion step waits in fs->Xr1 for the same launch
    // NEED_REORDER (frames that follow a landmark addition): K_new = m_nFilters, permutation between the normal and the
    // disordered layout (getPermutationMatrix, SLAM.cpp:1303-1334), disordered factor
    int K_new = 0;
    // data association (srukf_assoc.hip): per-landmark appearance records, allocated on first use
    unsigned char *app_patch = nullptr, *app_tmpl = nullptr, *d_image = nullptr;
    double *appR = nullptr, *appT = nullptr, *appPx = nullptr, *corr = nullptr;
    int* has_app = nullptr;
    int storage = SRUKF_STORAGE_F64;       // SRUKF_STORAGE_F32 / _F32_MIXED: X32 / S32 hold the inter-frame state
    float *S32 = nullptr, *X32 = nullptr;
    // SRUKF_STORAGE_F32_MIXED: S^T S - U U^T on the fp32 matrix pipe (srukf_mixed.hip)
    float *U32 = nullptr, *mx_part = nullptr; void *mx_tasks = nullptr, *mx_tiles = nullptr; int mx_ntasks = 0, mx_ntiles = 0;
    int *perm = nullptr, *iperm = nullptr;
    double* Sdis = nullptr;
    void* pan[2] = { nullptr, nullptr };   // GMW panel hand-off buffers (double-buffered), one launch per panel
    GmwPlan gplan;                         // persistent GMW launch: panel buffers, sync block, task list
    // rank-aware refactorisation (srukf_rank.hip): red_r > 0 = the n - red_r structurally null directions are not pivoted
    int red_r = 0, red_Tp = 0;
    int rank_aware = 1;                                // srukf_set_rank_aware
    int *red_perm = nullptr, *red_iperm = nullptr;     // permuted position <-> state index, kept indices first
    double* gdiag = nullptr;                           // diagonal of G in permuted order (the factorisation overwrites it)
    double *shadowA = nullptr, *Utp = nullptr;         // replay form: kept rows of S / U^T in permuted column order (srukf_rank.hip)
    double *slabW = nullptr, *slabL = nullptr;         // batched replay: the current panel's slabs W and L = W / D (64 x np each)
    // split form of the persistent factorisation (memory-tile sizes, a filter that has the GPU to itself): the slabs of every pivoted panel (gs_panels x 64 x np
    // each), the side stream the tile launch runs on and the events that fork it off / join it to the filter's stream
    double *gsW = nullptr, *gsL = nullptr; int gs_panels = 0;
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool split_off = false;                // a split-form pair of this context was abandoned (its two launches did not run side by side — e.g. the branches of a captured
                                           // graph sharing a hardware queue): the context keeps to the memory-tile instance of k_gmw_persist (read_fs; srukf_debug_get "split_off")
    double* P1 = nullptr; int* pxy2_tiles = nullptr; int n_pxy2_tiles = 0, pxy2_split_b0 = 0;   // "table" mode: k_pxy2's second K half, its tile list
    int* nskip = nullptr; int ns_full = 0, ns_null = 0, ns_rows = 0;   // NullSkip lists (srukf_device.h): [dirs | nulls | rows] in one buffer
    int* red_syrk_tiles = nullptr; int n_red_syrk_tiles = 0;   // k_syrk tiles of the kept rows (rows < 64 red_Tp) in permuted order: replay form without the owners' fold
    double red_fac_flop = 0, red_own_flop = 0;         // algorithmic flop of the rank-aware persistent launch: factorisation / owners' tiles of S^T S - U U^T
    GmwPlan gplan_red;                                 // tile list / sync block of the persistent launch with red_Tp pivoted panels
    int shared_tenants = 2;                // SRUKF_GPU_SHARED: how many persistent launches share the GPU (each keeps to cus / tenants CUs; the gate admits that many)
    int gmw_shared = 0;                    // 0: the filter has the GPU to itself; 1: shared with other filters — persistent launches of at most half the CUs behind
                                           // the admission gate (k_gmw_gate); 2: one launch per panel (forced, or after an abandoned persistent launch)
    int debug_allow_mixed = 0;             // srukf_debug_allow_mixed: the tolerance study runs the mixed mode below its epsilon floor on purpose
    // Measurement / test switches of srukf_debug_set (all default to the product path); one struct, so that a rebuilt context (map change) inherits them in one assignment
    struct DbgSwitches {
        int fused_motion = 2;              // "fused_motion": the replay's motion step — 0: its own launch (k_motion + k_project), 1: inside the projection launch
                                           // (k_project_motion), 2: "table" mode where the rank-aware tail allows it (replay_motion_mode)
        int f32_fuse = 1;                  // "f32_fuse": fp32 storage also runs in "fused tail" mode (rounding inside k_rank_expand<2> and the state update)
        int table_perm = 1;                // "table_perm": "table" / "fused tail" mode also where the owners do not fold (k_syrk over the kept rows: N >= 300); 0: k_project_motion + k_pxy there
        int tail_fuse = 1;                 // "tail_fuse": k_rank_expand also projects the next frame ("fused tail" mode); 0: k_project_table in front of every frame
        int head_fold = 1;                 // "head_fold": exclusive rank-aware replay without the k_syrk launch (helper workgroups of the persistent launch)
        int nullskip = 1;                  // "nullskip": with pxy2, structurally null directions are projected for their own landmark only (NullSkip)
        int pxy2 = 1;                      // "pxy2": "table" mode forms the cross covariances on the permuted operands (k_pxy2); 0: k_pxy
        int step_fast = 1;                 // "step_fast": 0: the step-wise API keeps to its own launch sequences (k_motion, k_project, k_meas_*, k_pxy, ...: round 4's path)
        int split_record = 0;              // "split_record": every split-form factorisation first copies its input matrix to Gbak (scripts/split_replay.py)
    } dbg;
    bool null_canonical = false;           // every structurally null row of S is exactly sqrt(EPSILON) e_k (update_null_set checks; true behind every rank-aware frame tail)
    bool tail_ok = false;                  // "fused tail" mode is possible: directions 0 and 1 are kept rows (the Si factor names their Z rows: they are projected for every landmark, which
                                           // the frame tail only does for kept rows — a state where they are structurally null stays with k_project_table)
    int debug_starve = 0;                  // srukf_debug_starve_workers: persistent launches start without their workers (tests of the fallback)
    int clamp_frame_host = -1, clamp_row_host = -1;   // what the last SRUKF_ERR_CLAMP_PENDING was about (srukf_clamp_info)
    double *ckS = nullptr, *ckX = nullptr; // srukf_run_frames: state before the block of frames in flight (recovery from a theta-clamp frame)
    int *syrk_tiles = nullptr, *pxy_tiles = nullptr;   // (by, bx) per workgroup, XCD-aware order
    int *syrk_head_tiles = nullptr;                    // k_syrk tiles of the first srukf_gmw_head_rows() rows only (fused refactor)
    int n_syrk_tiles = 0, n_pxy_tiles = 0, n_syrk_head_tiles = 0, n_syrk_head_crit = 0;
    FrameScalars* fs = nullptr;
    // staged sequence
    int seqF = 0;
    double *odo_seq = nullptr, *z_seq = nullptr;
    int* m_seq = nullptr;
    // pinned staging
    double* hstage = nullptr; size_t hstage_bytes = 0;
    FrameScalars* hfs = nullptr;
    // state machine
    int phase = 0;   // 0 idle, 1 after predict_motion, 2 after predict_measurement
    double next_odo[6] = { 0, 0, 0, 0, 0, 0 }; bool next_odo_valid = false;   // srukf_predict_motion_next: the pair the next srukf_predict_motion will bring
    // Fast path of the step-wise API (step_* below): a frame of the staged replay's own launch sequence ("fused tail" mode) cut in two at the host's association step
    double* odo_step = nullptr;            // device: (prev, cur, next) poses of the frame in flight — a three-pose "staged sequence" fs->odo_seq points at
    double step_odo[6] = { 0, 0, 0, 0, 0, 0 };   // the pair srukf_predict_motion was called with (the fallback to the other path needs it again)
    int step_seqF = 1;                     // 2: odo_step holds the next pose too (hint), the tail prepares and projects the next frame
    bool step_fast = false;                // the frame in flight runs on the fast path
    bool step_uncommitted = false;         // ... and its motion step still waits beside the state (fs->Xr1, Cmat): state getters commit it first (k_commit_motion)
    bool step_chain = false;               // X, S, the permuted copy and the frame scalars are exactly what the last fast-path tail left: its constant rows stand
    bool proj_valid = false; double proj_odo[6] = { 0, 0, 0, 0, 0, 0 };   // ... and that tail projected the frame with this odometry pair (Z, DZ, the table, fs->ctl)
    bool fs_seq_step = false;              // fs->odo_seq points at odo_step (srukf_run_frames_async points it back at the staged sequence)
    bool last_update_sequential = false;   // a host that updates in SRUKF_UPDATE_SEQUENTIAL mode never takes the fast path (decided at predict time)
    bool f32_stale = false;                // fp32 storage: X32 / S32 (srukf_get_state_f32) are behind the rounded fp64 working copies (refreshed on demand)
    int step_fast_frames = 0, step_slow_frames = 0;   // srukf_debug_get "step_fast" / "step_slow"
    bool async_pending = false;
    std::string err;
    // one captured frame (BATCHED, staged inputs): replayed by srukf_run_frames_async
    hipGraph_t graph = nullptr, graph8 = nullptr;          // one frame / SRUKF_GRAPH_FRAMES frames
    hipGraphExec_t graph_exec = nullptr, graph8_exec = nullptr;
    hipGraph_t graphN = nullptr; hipGraphExec_t graphN_exec = nullptr; int graphN_frames = 0;   // srukf_prepare_frames: a whole block of frames in ONE graph
    bool use_graph = true;
    // profiling
    bool profiling = false;
    std::vector<ProfEvent> pev;
    double prof_ms[KC_COUNT]; long long prof_n[KC_COUNT]; double prof_flops[KC_COUNT]; double prof_bytes[KC_COUNT];
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    char b_[256]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    (ctx)->err = b_; return SRUKF_ERR_HIP; } } while (0)

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static void host_weights(int Na, const srukf_params& p, KWeights& w)
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

// ---- profiling helpers -------------------------------------------------------------------------
struct ProfScope {
    srukf_ctx* c; int kc; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(srukf_ctx* c_, int kc_, double flops, double bytes) : c(c_), kc(kc_) {
        if (c->profiling) {
            hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->stream);
            c->prof_flops[kc] += flops; c->prof_bytes[kc] += bytes;
        }
    }
    ~ProfScope() {
        if (c->profiling) { hipEventRecord(b, c->stream); c->pev.push_back({ a, b, kc }); }
    }
};
static void prof_collect(srukf_ctx* c)
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

// ---- XCD-aware tile order -------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group, MI355X_MICROARCH.md);
// each XCD has its own 4 MiB L2 and S (11.6 MB at N = 200) does not fit one.  Give every XCD a fixed set of
// operand panels (tile rows `own`) and walk the other index in the same order on all XCDs, so that an
// own-panel stays L2-resident and the streamed panel is shared by the tiles that run next to each other.
//   own(t) = t % 8;  list per XCD: for other = 0.. : for own-tiles of this XCD: (own, other) if valid.
// The table maps linear workgroup id -> tile; unused slots hold (-1, -1).
static std::vector<int> build_tile_table(int n_own, int n_other, bool upper, bool own_is_row, int k_index /* 0: K grows with tile.x, 1: with tile.y */)
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

// ---- launch sequences --------------------------------------------------------------------------
static void step_commit_motion(srukf_ctx* c);
static void step_invalidate(srukf_ctx* c);
// the state is about to be replaced or read by somebody outside the step-wise fast path
static void step_state_replaced(srukf_ctx* c) { c->step_uncommitted = false; c->step_fast = false; c->xr1_pending = false; step_invalidate(c); c->f32_stale = false; }
static void quantize_state(srukf_ctx* c)
{
    if (c->storage != SRUKF_STORAGE_F64)
        hipLaunchKernelGGL(k_quantize, dim3(c->d.n + 1), dim3(256), 0, c->stream, c->d.n, c->d.np, c->S, c->X, c->S32, c->X32);
}
// rank-aware replay form: what k_motion / k_gain / k_syrk carry along (all null when the shadow copy does not exist)
static RankArgs rank_args(const srukf_ctx* c, bool prep_next = false, bool dzperm = false, bool f32round = false)
{
    RankArgs ra = {};
    ra.prep_next = prep_next ? 1 : 0; ra.dzperm = dzperm ? 1 : 0; ra.f32round = f32round ? 1 : 0;
    if (c->red_r > 0 && c->shadowA) { ra.A = c->shadowA; ra.Utp = c->Utp; ra.gdiag = c->gdiag; ra.iperm = c->red_iperm; ra.perm = c->red_perm; ra.r = c->red_r; }
    return ra;
}
// NullSkip of the "table" mode (all null: every direction is projected and read in full)
static NullSkip null_skip(const srukf_ctx* c)
{
    NullSkip ns = {};
    if (c->dbg.nullskip && c->dbg.pxy2 && c->nskip && c->red_r > 0) {
        ns.dirs = c->nskip; ns.nulls = c->nskip + c->ns_full; ns.rows = c->nskip + c->ns_full + c->ns_null;
        ns.nfull = c->ns_full; ns.nnull = c->ns_null; ns.nrows = c->ns_rows; ns.iperm = c->red_iperm; ns.r = c->red_r;
    }
    return ns;
}
// fs->Xr1 for the launch that applies the pending state update (and only once)
static const double* take_xr1(srukf_ctx* c)
{
    if (!c->xr1_pending) return nullptr;
    c->xr1_pending = false;
    return (const double*)((const char*)c->fs + offsetof(FrameScalars, Xr1));
}
// Replay path: motion step + projection of all sigma points in ONE launch (k_project_motion): workgroup 0 is the motion step,
// whose results wait beside the state (fs->Xr1, Cmat) until k_gain / the dX job commit them.
static void seq_predict_fused(srukf_ctx* c, int mode)
{
    const KDims& d = c->d;
    ProfScope ps(c, mode == 2 ? KC_PROJECT_TABLE : KC_PROJECT_MOTION, 2.0 * 60.0 * d.Na * d.N + 60.0 * d.L, 8.0 * ((double)d.n * d.n / 2 + 2.0 * d.L * 2 * d.N + (double)d.n * 2 * d.N + 8.0 * d.L + 8.0 * d.n));
    if (mode == 2) srukf_launch_project_table(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->Z, c->DZ, c->fs, rank_args(c, false, c->dbg.pxy2 != 0), null_skip(c));
    else srukf_launch_project_motion(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->Z, c->DZ, c->fs, rank_args(c));
    c->xr1_pending = true;
}
static void seq_predict_motion(srukf_ctx* c, const double* odo_pair_dev)
{
    const KDims& d = c->d;
    ProfScope ps(c, KC_MOTION, 60.0 * d.L, 8.0 * (4.0 * d.n + 8.0 * d.L + 4.0 * d.n));
    srukf_launch_motion(c->stream, d, c->w, c->p, c->X, c->S, c->sigR, c->Cmat, c->fs, c->odo_seq, odo_pair_dev, rank_args(c));
}
// fused_stats: the statistics ride on the k_pxy launch of seq_gain (replay path, no host in between)
static void seq_predict_measurement(srukf_ctx* c, bool fused_stats)
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
static int gmw_fused_mode() { return g_dbg_gmw_fused; }
static int rank_fused_mode() { return g_dbg_rank_fused; }
static int rank_fold_mode() { return g_dbg_rank_fold; }
static void shadow_rebuild(srukf_ctx* c)
{
    if (c->red_r > 0 && c->shadowA) srukf_launch_rank_shadow(c->stream, c->d.n, c->d.np, c->red_r, c->S, c->red_perm, c->shadowA);
}
static bool split_form(const srukf_ctx* c, const GmwPlan& gp, bool ignore_starve = false);
// (a plan with workers < 0 — more tiles than the workers of k_gmw_persist can own — still has the split form)
static bool gmw_plan_persists(const srukf_ctx* c, const GmwPlan& gp) { return gp.workers >= 0 || split_form(c, gp, true); }
static bool gmw_use_persist(const srukf_ctx* c) { return gmw_persist_mode() && c->gmw_shared != 2 && gmw_plan_persists(c, c->gplan); }
// SRUKF_GPU_SHARED: how many persistent launches share the GPU (each keeps to 1 / tenants of the CUs; the gate admits that many)
// (per context: srukf_run_frames_batch picks it from the number of filters it runs — one tenant per filter up to SRUKF_MAX_TENANTS; srukf_set_exclusive alone uses the
//  process-wide default of srukf_debug_set "shared_tenants")
static std::atomic<int> g_dbg_shared_tenants{2};
#define SRUKF_MAX_TENANTS 4                                    // 4 x (1 pivot + 63 workers with two register tiles each) fill 256 CUs at N = 200
static int plan_tenants(const srukf_ctx* c) { return c->gmw_shared == 1 ? c->shared_tenants : 1; }
static int gate_limit(const srukf_ctx* c) { return c->gmw_shared == 1 ? c->shared_tenants : 0; }
// "mem_split" 0: sizes beyond two tiles per worker keep the memory-tile instance of k_gmw_persist (round 3's form) instead of the split form
static std::atomic<int> g_dbg_mem_split{1};
// (2, measurements: also the plans whose workers own two register tiles each)
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
// Buffers / side stream of the split form for a plan with Tp pivoted panels (not inside a capture).  Failure is not an error: the memory-tile form is used.
static void split_ensure(srukf_ctx* c, const GmwPlan& gp)
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
static bool split_form(const srukf_ctx* c, const GmwPlan& gp, bool ignore_starve)
{
    return g_dbg_mem_split && !c->split_off && c->gmw_shared == 0 && (ignore_starve || !c->debug_starve) && c->side && c->gsW && c->gs_panels >= gp.Tp && gp.T >= 16 && gp.T + 1 <= gp.cus && split_wanted(gp);
}
static void launch_gmw_fast(srukf_ctx* c, double* Gbuf, double* Sout, bool reduced = false);
// Tail of every rank-aware refactorisation: factor rows (c->G, permuted order) -> S and the permuted copy, checks, frame tail.
// fp32 storage: S, X and the permuted copy are rounded to the stored values first, and the trajectory row is taken from those
// (as the full-rank form does: quantize_state before the tail).
// table: "table" mode of the replay — the tail also prepares the next frame's table of robot poses (k_rank_expand)
// fuse: "fused tail" mode — this launch also projects the next frame's sigma points (k_rank_expand<2>)
static void rank_expand(srukf_ctx* c, bool frame_tail, bool table = false, bool fuse = false)
{
    const int n = c->d.n, np = c->d.np;
    const bool f32s = c->storage == SRUKF_STORAGE_F32;
    const bool f32fuse = f32s && fuse && table && frame_tail;      // fp32 storage in "fused tail" mode: the launch rounds what it writes (no k_quantize / k_rank_round / k_traj behind it)
    const bool f32 = f32s && !f32fuse;
    const bool tt = table && frame_tail && !f32;
    srukf_launch_rank_expand(c->stream, n, np, c->red_r, c->p.epsilon, c->G, c->D, c->red_perm, c->red_iperm, c->gdiag, c->fs, c->X, (frame_tail && !f32) ? 1 : 0, c->S, c->shadowA,
                             tt ? c->sigR : nullptr, c->w.gamma, (tt && fuse) ? 1 : 0, c->d, c->w, c->p, c->Z, c->DZ, f32fuse ? 1 : 0);
    if (f32) {
        quantize_state(c);
        srukf_launch_rank_round(c->stream, np, c->red_r, c->shadowA);
        if (frame_tail) srukf_launch_traj(c->stream, c->d, c->X, c->S, c->fs, nullptr, 1);
    }
}
// the rank-aware replay whose owners form their tiles of S^T S - U U^T themselves (seq_refactor below): what a whole staged frame takes
// How many tiles per worker (in percent) the owners' fold accepts.  A filter that has the GPU to itself: 106 = about one tile per worker (measured in round 2: with two
// tiles per worker, both to be formed before the first step, the exclusive replay loses — N = 300: 1 544 against 1 663 frames/s; srukf_debug_set "fold_tiles_pct").
// A filter that shares the GPU (three or four tenants of 256 / tenants CUs: two register tiles per worker): 200 — measured in round 4 at N = 200, four filters and four
// tenants, aggregate frames/s: owners fold both tiles 12 260; the same tiles from a launch of their own in the owners' summation order (k_syrk_own) 8 500 - 11 900;
// split-K k_syrk over the kept rows 13 100 but then the results differ in rounding from the same filter running alone.
static int fold_tiles_pct(const srukf_ctx* c) { return c->gmw_shared == 1 ? 200 : 106; }
static bool replay_red_fused(const srukf_ctx* c)
{
    return c->red_r > 0 && c->storage != SRUKF_STORAGE_F32_MIXED && c->shadowA && c->w.wc0 == c->w.wm0 && gmw_use_persist(c) &&
           c->gplan_red.workers >= 0 && c->gplan_red.nreal <= c->gplan_red.workers * fold_tiles_pct(c) / 100 && c->gplan_red.T >= 16 &&
           !c->debug_starve && gmw_fused_mode() && rank_fused_mode() && rank_fold_mode();
}
// 0: k_motion + k_project; 1: k_project_motion (motion workgroup + projection with the robot part inline); 2: "table" (k_project_table: the
// previous frame's tail prepared the robot part of every sigma point) — only where the tail is k_rank_expand on fp64 storage
// ... or forms them with k_syrk over the kept rows, still in permuted order (memory tiles, two tiles per worker, one launch per panel: seq_refactor's second branch)
static bool replay_red_perm(const srukf_ctx* c)
{
    return c->red_r > 0 && c->storage != SRUKF_STORAGE_F32_MIXED && c->shadowA && c->w.wc0 == c->w.wm0 && rank_fused_mode() && c->dbg.table_perm;
}
static int replay_motion_mode(const srukf_ctx* c)
{
    // fp32 storage: only as "fused tail" mode (k_rank_expand<2> and the state update round what they write; "table" mode alone has no such form)
    const bool st_ok = c->storage == SRUKF_STORAGE_F64 ||
                       (c->storage == SRUKF_STORAGE_F32 && c->dbg.f32_fuse && c->dbg.tail_fuse && c->dbg.pxy2 && c->dbg.nullskip && c->nskip && c->tail_ok &&
                        (size_t)c->d.np * sizeof(double) <= 48 * 1024);
    // (null_canonical: "table" mode and everything on top of it read the structurally null rows of S as sqrt(EPSILON) e_k without looking)
    if (c->dbg.fused_motion == 2 && !((replay_red_fused(c) || replay_red_perm(c)) && st_ok && c->null_canonical)) return 1;
    return c->dbg.fused_motion;
}
// "fused tail" mode (default where "table" mode runs with k_pxy2 and NullSkip): k_rank_expand also projects the next frame (k_rank_expand<2>), the frame's motion reduction
// rides on k_pxy2 (MeasArgs::fmode), k_gain re-centres the robot rows: a frame is k_pxy2, k_gain, k_gmw_persist, k_rank_expand, and only a run's first frame has a projection launch
static bool replay_fuse_mode(const srukf_ctx* c)
{
    return replay_motion_mode(c) == 2 && c->dbg.pxy2 && c->dbg.nullskip && c->nskip && c->tail_ok && c->dbg.tail_fuse &&
           (size_t)c->d.np * sizeof(double) <= 48 * 1024;      // (k_rank_expand<2> keeps a row of the factor in dynamic LDS)
}
#define SRUKF_HEAD_FOLD_MIN_FREE_CUS 16
static bool head_fold_ok(const srukf_ctx* c)
{
    return c->dbg.head_fold && c->gmw_shared == 0 && c->gplan_red.cus - 1 - c->gplan_red.workers >= SRUKF_HEAD_FOLD_MIN_FREE_CUS;
}
static void seq_refactor(srukf_ctx* c, int ub, int ue, bool slow, bool keep_backup, bool need_reset, bool frame_tail, bool table = false, bool fuse = false)
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
    const bool reduced = !slow && c->red_r > 0 && c->storage != SRUKF_STORAGE_F32_MIXED;
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
            srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->syrk_head_tiles, c->n_syrk_head_tiles, c->dx_pending ? c->dxp : nullptr, c->X, rank_args(c, table, false, fuse && c->storage == SRUKF_STORAGE_F32), take_xr1(c));
            c->dx_pending = false;
        }
        {
            // factorisation of the leading red_Tp panels (all n columns carried along) + the owners' tiles of S^T S - U U^T
            // (kept rows below the head x all columns, K <= r and 2N): red_*_flop, update_null_set
            ProfScope ps(c, KC_GMW_PERSIST, c->red_fac_flop + c->red_own_flop + (head_fold ? head_flop : 0.0), 8.0 * (2.0 * rr * n + (double)d.mp * n));
            HeadArgs ha = {};
            if (head_fold) {
                ha.tiles = (const int2*)c->syrk_head_tiles; ha.ntiles = c->n_syrk_head_tiles; ha.ncrit = c->n_syrk_head_crit;
                ha.dxp = c->dx_pending ? c->dxp : nullptr; ha.X = c->X; ha.xr1 = take_xr1(c); ha.ndx = c->dx_pending ? (n + 255) / 256 : 0;
                ha.ra = rank_args(c, table, false, fuse && c->storage == SRUKF_STORAGE_F32); ha.ngd = (n - c->red_r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS;
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
        {
            ProfScope ps(c, KC_SYRK, rr * rr * rr / 3.0 + rr * rr * (n - rr) + 2.0 * d.mp * (rr * n - rr * rr / 2.0) + 2.0 * (n - rr) * (rr + d.mp), 8.0 * (rr * n + (double)d.mp * n + rp * n));
            if (own_order) {
                srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->syrk_head_tiles, c->n_syrk_head_tiles, c->dx_pending ? c->dxp : nullptr, c->X, rank_args(c, table, false, fuse && c->storage == SRUKF_STORAGE_F32), take_xr1(c));
                srukf_launch_syrk_own(c->stream, n, np, c->shadowA, c->Utp, 0, d.mp, (c->red_r + 15) & ~15, c->Wf, c->fs, c->gplan_red.tiles, c->gplan_red.ntiles, c->red_Tp);
            } else
            srukf_launch_syrk(c->stream, d, c->shadowA, c->Utp, 0, d.mp, c->Wf, c->fs, c->red_syrk_tiles, c->n_red_syrk_tiles, c->dx_pending ? c->dxp : nullptr, c->X, rank_args(c, table, false, fuse && c->storage == SRUKF_STORAGE_F32), take_xr1(c));
            c->dx_pending = false;
        }
        {
            ProfScope ps(c, gmw_use_persist(c) && gmw_plan_persists(c, c->gplan_red) ? KC_GMW_PERSIST : KC_GMW_TRAIL, c->red_fac_flop, 8.0 * 2.0 * rp * n);
            launch_gmw_fast(c, c->Wf, c->G, true);
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
    if (c->storage == SRUKF_STORAGE_F32_MIXED && ub == 0 && ue == d.mp) {
        // mixed precision: the fp32 state S32 and U^T rounded once, products on the fp32 matrix pipe, chunk sums in FP64
        ProfScope ps(c, KC_SYRK, syrk_flop, 4.0 * (nn * nn + (double)(ue - ub) * nn) + 8.0 * nn * nn / 2);
        if (c->dx_pending) srukf_launch_gain_dx(c->stream, n, np, c->dxp, c->X, take_xr1(c));
        c->dx_pending = false;
        srukf_launch_cvt_f32(c->stream, (size_t)d.mp * np, c->Ut, c->U32);
        srukf_launch_cvt_robot_cols(c->stream, n, np, c->S, c->S32);          // the motion step's columns, computed after the state was rounded
        srukf_launch_syrk32(c->stream, n, np, d.mp, c->S32, c->U32, c->mx_tasks, c->mx_ntasks, c->mx_tiles, c->mx_ntiles, c->mx_part, c->G, c->fs);
    } else {
        ProfScope ps(c, KC_SYRK, syrk_flop * head_frac, syrk_byte * head_frac);
        srukf_launch_syrk(c->stream, d, c->S, c->Ut, ub, ue, c->G, c->fs, fused ? c->syrk_head_tiles : c->syrk_tiles,
                          fused ? c->n_syrk_head_tiles : c->n_syrk_tiles, c->dx_pending ? c->dxp : nullptr, c->X, RankArgs{}, take_xr1(c));
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
        for (int j = 0; j < n; j++)
            srukf_launch_gmw_col(c->stream, n, np, j, c->p.epsilon, c->G, c->Wf, c->D, c->theta, c->fs, c->S);
        quantize_state(c);
        if (frame_tail) srukf_launch_traj(c->stream, d, c->X, c->S, c->fs, nullptr, 1);
    }
    shadow_rebuild(c);                                 // S was rewritten by a path that does not keep the permuted copy in step
}
// Blocked fast path (or, slow = true, the exact column path) on an arbitrary matrix buffer: Gbuf (upper triangle,
// destroyed) -> upper-triangular factor rows in Sout (whose lower triangle must already be zero).
static void launch_gmw_fast(srukf_ctx* c, double* Gbuf, double* Sout, bool reduced)
{
    const int np = c->d.np, n = c->d.n;
    const GmwPlan& gp = reduced ? c->gplan_red : c->gplan;
    const int Tp = reduced ? c->red_Tp : np / 64;
    if (gmw_use_persist(c) && gmw_plan_persists(c, gp)) {
        // srukf_debug_starve_workers (tests only): launch without workers, as if the GPU were taken — the pivot's bounded wait
        // expires, the frame is flagged and repeated on the exact path, and the context falls back to one launch per panel
        const int workers = c->debug_starve ? 0 : gp.workers;
        if (split_form(c, gp, true)) {                         // (srukf_debug_starve_workers: the pair without its tile launch)
            // the tile launch depends on what produced Gbuf, not on the pivot / slab launch: fork before, join after (in a capture: two parallel branches)
            if (c->dbg.split_record) hipMemcpyAsync(c->Gbak, Gbuf, sizeof(double) * (size_t)np * np, hipMemcpyDeviceToDevice, c->stream);
            hipEventRecord(c->ev_fork, c->stream);
            hipStreamWaitEvent(c->side, c->ev_fork, 0);
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
static void run_gmw(srukf_ctx* c, double* Gbuf, double* Sout, bool slow)
{
    const int np = c->d.np, n = c->d.n;
    if (!slow) {
        launch_gmw_fast(c, Gbuf, Sout);
        srukf_launch_gmw_check(c->stream, n, np, c->D, Sout, c->fs, c->X, 0, nullptr);
    } else {
        hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 0);
        for (int j = 0; j < n; j++) srukf_launch_gmw_col(c->stream, n, np, j, c->p.epsilon, Gbuf, c->Wf, c->D, c->theta, c->fs, Sout);
    }
}
static int read_fs(srukf_ctx* c);
// GSLCholeskyUpdate with FLAG_NEED_REORDER (SLAM.cpp:2122-2138) for the columns [ub, ue) of U:
//   dst = S^T S - U U^T;  dst_dis = Pi^T dst Pi  (disordered layout: the rank-deficient new-anchor block last);
//   S_dis = [R11 R12; 0 0],  R11 = gmw(dst_dis[0:r, 0:r]),  R12 = R11^{-T} dst_dis[0:r, r:n]   (2158-2179, r = n - 3 K_new);
//   S = R factor of QR(Pi S_dis Pi^T).
// [R11 R12] is what the right-looking GMW leaves in its first r rows whatever stands in the lower right block, so the
// full factorisation runs and rows >= r are zeroed.  R^T R = Pi (S_dis^T S_dis) Pi^T, so the QR is a second
// SYRK + permutation + GMW (P = S^T S is what the filter consumes; row signs of R are a convention).
static int refactor_reorder(srukf_ctx* c, int ub, int ue)
{
    const KDims& d = c->d;
    const int np = d.np, n = d.n, r = n - 3 * c->K_new;
    const size_t bytes = sizeof(double) * (size_t)np * np;
    if (!c->Sdis) { if (srukf_dmalloc((void**)&c->Sdis, bytes) != hipSuccess) { c->err = "out of device memory (NEED_REORDER buffer)"; return SRUKF_ERR_NOMEM; } }
    hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 1);
    srukf_launch_syrk(c->stream, d, c->S, c->Ut, ub, ue, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, c->dx_pending ? c->dxp : nullptr, c->X, RankArgs{}, take_xr1(c));
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
static void seq_pxy(srukf_ctx* c, bool fused_stats, bool fused_motion = false, bool table = false, bool preamble = false, bool fmode = false)
{
    const KDims& d = c->d;
    const double nn = d.n;
    ProfScope ps(c, table ? KC_PXY2 : KC_PXY, nn * nn * 2.0 * d.N, 8.0 * (nn * nn / 2 + 2.0 * nn * 2 * d.N));
    MeasArgs ms = {};
    // ("fused tail" mode: the statistics are centred on the centre point's robot part, row 0 of the table: the mean does not exist yet)
    const double* xrob = fmode ? c->sigR : fused_motion ? (const double*)((const char*)c->fs + offsetof(FrameScalars, Xr1)) : c->X + (d.n - 4);
    if (fused_stats) ms = MeasArgs{ c->X, xrob, c->sigR, c->Z, c->mpart, c->h, c->Si, c->vis, c->PxyR, c->fs, (d.N + 31) / 32, table ? null_skip(c) : NullSkip{}, preamble ? 1 : 0,
                                    fmode ? 1 : 0, c->Cmat };
    if (table) srukf_launch_pxy2(c->stream, d, c->DZ, c->shadowA, c->Utp, c->P1, c->pxy2_tiles, c->n_pxy2_tiles, (c->red_r + 15) & ~15, c->w, ms);
    else srukf_launch_pxy(c->stream, d, c->DZ, c->S, c->Ut, c->pxy_tiles, c->n_pxy_tiles, c->w, ms);
}
// second half: gains, U^T, slice partials of the state update; z_dev / m_dev: this frame's measurements and matches on the device (null: the staged sequence's)
static void seq_gain_only(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_motion = false, bool table = false, bool fmode = false)
{
    const KDims& d = c->d;
    ProfScope ps(c, KC_GAIN, 8.0 * d.n * 2 * d.N, 8.0 * 2.0 * d.n * 2 * d.N);
    srukf_launch_gain(c->stream, d, c->w, c->Ut, c->PxyR, c->Si, c->vis, c->h, c->z_seq, z_dev, c->m_seq, m_dev, c->fs, c->dxp, c->X, c->Z, rank_args(c),
                      fused_motion ? c->Cmat : nullptr, c->S, table ? c->P1 : nullptr, c->pxy2_split_b0, c->DZ,
                      (fmode && c->storage == SRUKF_STORAGE_F32) ? (double)(float)sqrt(c->p.epsilon) : sqrt(c->p.epsilon),   // (the null rows of S as they are stored)
                      c->sigR, fmode ? 1 : 0);
    c->dx_pending = true;                             // applied by the next k_syrk launch (seq_refactor)
}
static void seq_gain(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_stats, bool fused_motion = false, bool table = false, bool preamble = false, bool fmode = false)
{
    seq_pxy(c, fused_stats, fused_motion, table, preamble, fmode);
    seq_gain_only(c, z_dev, m_dev, fused_motion, table, fmode);
}

// Which directions of the state are structurally null (srukf_rank.hip)?  Called whenever a state arrives from outside
// (srukf_set_state*, map changes): row energies of S on the device, the lists on the host.  srukf_debug_set(0, "rank_aware", 0) switches it off.
static void drop_graphs(srukf_ctx* c);
#define SRUKF_NULL_ENERGY 1e-12
static int update_null_set(srukf_ctx* c)
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
            }
            if (!c->shadowA) {
                HIPCHK(c, srukf_dmalloc(&c->shadowA, sizeof(double) * (size_t)np * np)); HIPCHK(c, srukf_dmalloc(&c->Utp, sizeof(double) * (size_t)c->d.mp * np));
                HIPCHK(c, hipMemsetAsync(c->Utp, 0, sizeof(double) * (size_t)c->d.mp * np, c->stream));
                HIPCHK(c, srukf_dmalloc(&c->P1, sizeof(double) * (size_t)c->d.mp * np));
                HIPCHK(c, hipMemsetAsync(c->P1, 0, sizeof(double) * (size_t)c->d.mp * np, c->stream));
            }
            {
                // k_pxy2 ("table" mode): 64 x 64 tiles of the permuted product, K ends at the kept rows, long K ranges in two halves
                const int kr = (r + 15) & ~15;
                const int nt = srukf_pxy2_build_tiles(c->d.mp, np, kr, nullptr);
                std::vector<int> tl((size_t)4 * nt);
                srukf_pxy2_build_tiles(c->d.mp, np, kr, tl.data());
                if (c->pxy2_tiles) srukf_dfree_on(c->pxy2_tiles, c->stream);
                c->pxy2_tiles = nullptr; c->n_pxy2_tiles = nt;
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
    return SRUKF_OK;
}

// ---- C-ABI --------------------------------------------------------------------------------------
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

static int alloc_zero(srukf_ctx* c, void** p, size_t bytes)
{
    HIPCHK(c, srukf_dmalloc_on(p, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
    return SRUKF_OK;
}
#define ALLOC(ptr, count) do { int rc_ = alloc_zero(c, (void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)); if (rc_) { g_create_error = c->err; srukf_destroy(c); return rc_; } } while (0)

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
    ALLOC(c->Ut, mp * np); ALLOC(c->h, mp); ALLOC(c->Si, 4 * (size_t)(N > 0 ? N : 1)); ALLOC(c->PxyR, 5 * mp);                       // rows 0..3: robot rows of the cross covariances; row 4: scratch of the "fused tail" statistics
    ALLOC(c->D, np); ALLOC(c->zcur, mp); ALLOC(c->odocur, 8); ALLOC(c->small, 64);
    ALLOC(c->vis, N > 0 ? N : 1); ALLOC(c->mcur, N > 0 ? N : 1); ALLOC(c->theta, np); ALLOC(c->fs, 1);
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
    if ((!c->hstage && hipHostMalloc((void**)&c->hstage, c->hstage_bytes) != hipSuccess) || hipHostMalloc((void**)&c->hfs, sizeof(FrameScalars)) != hipSuccess) {
        g_create_error = "hipHostMalloc failed"; srukf_destroy(c); return SRUKF_ERR_NOMEM;
    }
    int rc = srukf_reset(c);
    if (rc) { g_create_error = c->err; srukf_destroy(c); return rc; }
    *out = c;
    return SRUKF_OK;
}

static void batch_plan_forget(const srukf_ctx* c);
int srukf_destroy(srukf_ctx* c)
{
    if (!c) return SRUKF_OK;
    hipSetDevice(c->device);
    batch_plan_forget(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (c->graph_exec) hipGraphExecDestroy(c->graph_exec);
    if (c->graph) hipGraphDestroy(c->graph);
    if (c->graph8_exec) hipGraphExecDestroy(c->graph8_exec);
    if (c->graph8) hipGraphDestroy(c->graph8);
    if (c->graphN_exec) hipGraphExecDestroy(c->graphN_exec);
    if (c->graphN) hipGraphDestroy(c->graphN);
    void* bufs[] = { c->X, c->S, c->G, c->Gbak, c->Wf, c->sigR, c->Cmat, c->Z, c->DZ, c->Ut, c->h, c->Si, c->PxyR, c->D,
                     c->zcur, c->odocur, c->small, c->vis, c->mcur, c->theta, c->fs, c->odo_seq, c->z_seq, c->m_seq, c->pan[0], c->pan[1], c->mpart, c->dxp, c->syrk_tiles, c->pxy_tiles, c->syrk_head_tiles,
                     c->perm, c->iperm, c->Sdis, c->ckS, c->ckX, c->odo_step, c->red_perm, c->red_iperm, c->gdiag, c->red_syrk_tiles, c->shadowA, c->Utp, c->P1, c->pxy2_tiles, c->nskip, c->slabW, c->slabL, c->gsW, c->gsL, c->S32, c->X32, c->U32, c->mx_part, c->mx_tasks, c->mx_tiles, c->app_patch, c->app_tmpl, c->d_image, c->appR, c->appT, c->appPx, c->corr, c->has_app };
    for (void* b : bufs) if (b) srukf_dfree_on(b, c->stream);
    gmw_plan_destroy(c->gplan, c->stream);
    gmw_plan_destroy(c->gplan_red, c->stream);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
    if (c->own_stream && c->stream) hipStreamSynchronize(c->stream);
    if (c->hstage) {
        // keep ONE pinned staging buffer for the next context (pinning 16 MB costs milliseconds; map changes rebuild contexts)
        if (!g_spare_stage || g_spare_stage_bytes < c->hstage_bytes) { if (g_spare_stage) hipHostFree(g_spare_stage); g_spare_stage = c->hstage; g_spare_stage_bytes = c->hstage_bytes; }
        else hipHostFree(c->hstage);
    }
    if (c->hfs) hipHostFree(c->hfs);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return SRUKF_OK;
}

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
    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
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

static int block_cov(srukf_ctx* c, int off, int bs, double* out)
{
    srukf_launch_block_cov(c->stream, c->d, c->S, off, bs, c->small);
    HIPCHK(c, hipMemcpyAsync(c->hstage, c->small, sizeof(double) * bs * bs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->hstage, sizeof(double) * bs * bs);
    return SRUKF_OK;
}

int srukf_get_robot(srukf_ctx* c, double pose4[4], double P4[16])
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->d.n;
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

// ---- step-wise API: the fast path ---------------------------------------------------------------------------------------------------------------
// The step-wise calls used to run launch sequences of their own (k_motion, k_project, k_meas_*, k_pxy, k_gain, a full k_syrk, the permutation pass, the persistent launch
// reading its tiles from memory, k_rank_expand, the rebuild of the permuted copy): ~2 x the staged replay's time per frame before the host round trips.  Where the replay's
// "fused tail" mode applies (replay_fuse_mode: rank-aware form with canonical null rows; BATCHED, NEEDNOT_REORDER) a step-wise frame now IS a frame of the staged replay, cut
// in two at the host's association step:
//   srukf_predict_motion       [k_set_step; unless the previous frame's tail projected this very odometry pair: k_sigr_rows + k_project_table;] k_pxy2 (motion reduction,
//                              measurement statistics, cross covariances);  the state before the frame is kept (ckS / ckX)
//   srukf_predict_measurement  D->H copies of h, Si, visible
//   srukf_update               H->D z / matched; k_gain, the persistent factorisation launch, k_rank_expand<2> (which, when the host has announced the next frame's odometry —
//                              srukf_predict_motion_next — also projects the next frame); the frame scalars come back, and a flagged frame (theta clamp, abandoned launch,
//                              a null direction that is not) is rewound and repeated on the other path, as srukf_run_frames does
// Same kernels on the same values as the staged replay: bit-identical states (tests/test_gpu_parity_r5.py::test_step_api_equals_staged_replay).
static void step_invalidate(srukf_ctx* c) { c->step_chain = false; c->proj_valid = false; }
static bool step_fast_eligible(const srukf_ctx* c) { return c->dbg.step_fast && !c->last_update_sequential && c->d.N > 0 && replay_fuse_mode(c); }
// a state getter between predict and update (or a frame that ends without an update): the motion step's results go where k_gain / the state update would put them
static void step_commit_motion(srukf_ctx* c)
{
    if (!c->step_uncommitted) return;
    const RankArgs ra = rank_args(c);
    hipLaunchKernelGGL(k_commit_motion, dim3((c->d.n + 255) / 256), dim3(256), 0, c->stream, c->d.n, c->d.np, c->X, c->S, c->Cmat, c->fs, ra.A, ra.iperm, ra.r);
    c->step_uncommitted = false;
}
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
    double* hs = c->hstage;
    for (int e = 0; e < 6; e++) hs[e] = c->step_odo[e];
    for (int e = 0; e < 3; e++) hs[6 + e] = hint ? c->next_odo[3 + e] : 0.0;
    c->step_seqF = hint ? 2 : 1;
    HIPCHK(c, hipMemcpyAsync(c->odo_step, hs, sizeof(double) * 9, hipMemcpyHostToDevice, c->stream));
    // the state before the frame: a flagged frame is repeated from it on the other path
    HIPCHK(c, hipMemcpyAsync(c->ckS, c->S, sizeof(double) * np * np, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ckX, c->X, sizeof(double) * np, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, c->stream, c->fs, c->odo_step, c->step_seqF, c->p.a1, c->p.a2, c->p.a3, c->p.a4, c->step_chain ? 0 : 1);
    c->fs_seq_step = true;
    if (!projected) {
        if (c->step_chain) hipLaunchKernelGGL(k_set_frame_control, dim3(1), dim3(1), 0, c->stream, c->fs);      // (the tail prepared the control of ANOTHER pair, or none)
        srukf_launch_sigr_rows(c->stream, d, c->w, c->X, c->S, c->sigR, c->fs, c->red_iperm, c->red_r);
        seq_predict_fused(c, 2);
    }
    c->xr1_pending = true;
    seq_pxy(c, true, true, true, true, true);
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
int srukf_predict_motion(srukf_ctx* c, const double odo_prev[3], const double odo_cur[3])
{
    if (!c || !odo_prev || !odo_cur) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->step_uncommitted) { step_commit_motion(c); step_invalidate(c); }      // a frame that was predicted and never updated: its motion step stands (as on the other path)
    c->step_fast = false;
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
    HIPCHK(c, hipMemcpyAsync(hs, c->h, sizeof(double) * 2 * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + 2 * N, c->Si, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + 6 * N, c->vis, sizeof(int) * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (h) memcpy(h, hs, sizeof(double) * 2 * N);
    if (Si) memcpy(Si, hs + 2 * N, sizeof(double) * 4 * N);
    if (visible) memcpy(visible, hs + 6 * N, sizeof(int) * N);
    c->phase = 2;
    return SRUKF_OK;
}

static void drop_graphs(srukf_ctx* c);
static void set_null_canonical(srukf_ctx* c);
static int read_fs(srukf_ctx* c)
{
    HIPCHK(c, hipMemcpyAsync(c->hfs, c->fs, sizeof(FrameScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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

static int step_update_slow(srukf_ctx* c, const double* z, const int* matched, int reorder, int mode, int nm);
// the frame in flight leaves the fast path: the state before the frame comes back and the frame's predict half runs again on the other path
static int step_rewind_to_slow(srukf_ctx* c)
{
    const size_t np = c->d.np;
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
    memcpy(hs, z, sizeof(double) * 2 * N);
    int* hm = (int*)(hs + 2 * N);
    memcpy(hm, matched, sizeof(int) * N);
    HIPCHK(c, hipMemcpyAsync(c->zcur, hs, sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->mcur, hm, sizeof(int) * N, hipMemcpyHostToDevice, c->stream));
    if (c->step_seqF == 1 && c->next_odo_valid && memcmp(c->next_odo, c->step_odo + 3, sizeof(double) * 3) == 0) {
        // the host announced the next frame's odometry after srukf_predict_motion: the tail of this frame can still project it
        double* ho = hs + 2 * N + N;                            // (behind z and matched in the pinned buffer)
        for (int e = 0; e < 3; e++) ho[e] = c->next_odo[3 + e];
        HIPCHK(c, hipMemcpyAsync(c->odo_step + 6, ho, sizeof(double) * 3, hipMemcpyHostToDevice, c->stream));
        c->step_seqF = 2;
        hipLaunchKernelGGL(k_set_seq, dim3(1), dim3(1), 0, c->stream, c->fs, c->odo_step, 2, c->p.a1, c->p.a2, c->p.a3, c->p.a4);
    }
    seq_gain_only(c, c->zcur, c->mcur, true, true, true);
    c->step_uncommitted = false;                               // (k_gain and the state update commit the motion step)
    seq_refactor(c, 0, d.mp, false, false, false, true, true, true);
    int rc = read_fs(c); if (rc) return rc;
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
    c->next_odo_valid = false;
    c->f32_stale = c->storage == SRUKF_STORAGE_F32;
    c->step_fast_frames++;
    HIPCHK(c, hipGetLastError());
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
    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
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
            hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
            hipLaunchKernelGGL(k_refactor_reset, dim3((d.np + 255) / 256), dim3(256), 0, c->stream, d.np, c->theta, c->fs, 0);
            HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
            ProfScope ps(c, KC_GMW_COL, 0, 0);
            for (int j = 0; j < d.n; j++) srukf_launch_gmw_col(c->stream, d.n, d.np, j, c->p.epsilon, c->G, c->Wf, c->D, c->theta, c->fs, c->S);
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
                    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
                    hipLaunchKernelGGL(k_refactor_reset, dim3((d.np + 255) / 256), dim3(256), 0, c->stream, d.np, c->theta, c->fs, 0);
                    HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
                    for (int j = 0; j < d.n; j++) srukf_launch_gmw_col(c->stream, d.n, d.np, j, c->p.epsilon, c->G, c->Wf, c->D, c->theta, c->fs, c->S);
                    quantize_state(c);
                    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, 0, 1);
                }
            }
        }
    }
    // rank-aware form: the reorder path and the exact column path write S without the permuted copy, and their factor may have
    // other null rows (a frame that went to the exact path because a skipped direction was found not to be null must not meet the
    // same null set again)
    if (reorder == SRUKF_NEED_REORDER || exact_ran) { const int rc = update_null_set(c); if (rc) return rc; }
    else {
        shadow_rebuild(c);
        if (c->storage != SRUKF_STORAGE_F32_MIXED) set_null_canonical(c);     // the rank-aware tail (k_rank_expand) has written sqrt(EPSILON) e_k into every skipped row
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

static void drop_graphs(srukf_ctx* c)
{
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->graph8_exec) { hipGraphExecDestroy(c->graph8_exec); c->graph8_exec = nullptr; }
    if (c->graph8) { hipGraphDestroy(c->graph8); c->graph8 = nullptr; }
    if (c->graphN_exec) { hipGraphExecDestroy(c->graphN_exec); c->graphN_exec = nullptr; }
    if (c->graphN) { hipGraphDestroy(c->graphN); c->graphN = nullptr; }
    c->graphN_frames = 0;
}
static int set_shared(srukf_ctx* c, int shared, int tenants);
int srukf_set_exclusive(srukf_ctx* c, int exclusive)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));                         // the plans below size themselves on the CURRENT device's CU count
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int shared = exclusive == SRUKF_GPU_SHARED ? 1 : exclusive == SRUKF_GPU_SHARED_PER_PANEL ? 2 : 0;
    return set_shared(c, shared, g_dbg_shared_tenants.load());
}
// shared: 0 exclusive, 1 shared (tenants persistent launches at a time), 2 one launch per panel
static int set_shared(srukf_ctx* c, int shared, int tenants)
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
int srukf_set_storage(srukf_ctx* c, int storage)
{
    if (!c || (storage != SRUKF_STORAGE_F64 && storage != SRUKF_STORAGE_F32 && storage != SRUKF_STORAGE_F32_MIXED)) return SRUKF_ERR_BAD_ARG;
    if (storage == SRUKF_STORAGE_F32_MIXED && c->p.epsilon < 1e-9 && !c->debug_allow_mixed) {
        // S^T S - U U^T formed from fp32 products carries ~1e-7 * max diag of rounding in the entries that are exactly zero in
        // exact arithmetic (P is permanently rank deficient: the anchors of jointly initialised landmarks are copies of the
        // robot position).  The reference's EPSILON = 1e-13 clamp sits far below that noise: null pivots |c_jj| ~ 1e-9 divide
        // off-diagonal noise of the same size, the multipliers are O(1) garbage and the filter diverges within ten frames
        // (scripts/mixed_eps_study.py).  The mode is only offered with a clamp above the fp32 noise floor.
        c->err = "SRUKF_STORAGE_F32_MIXED needs params.epsilon >= 1e-9 (fp32-formed S^T S - U U^T cannot resolve the reference's 1e-13 clamp)";
        return SRUKF_ERR_UNSUPPORTED;
    }
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
    return SRUKF_OK;
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

static void adopt_context(srukf_ctx* c, srukf_ctx* c2);
// a context about to be rebuilt (map change) offers its verified side stream to the context srukf_create builds next on the same filter stream (split_ensure)
static void side_stream_lend(srukf_ctx* c)
{
    if (!c->side) return;
    hipStreamSynchronize(c->side);
    spare_side_drop();
    g_spare_side.device = c->device; g_spare_side.main = c->stream; g_spare_side.side = c->side; g_spare_side.fork = c->ev_fork; g_spare_side.join = c->ev_join;
    c->side = nullptr; c->ev_fork = c->ev_join = nullptr;
}
// ---- data association ------------------------------------------------------------------------------
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
    HIPCHK(c, hipMemcpyAsync(c->d_image, gray, img, hipMemcpyHostToDevice, c->stream));
    step_commit_motion(c);                                               // the warp uses the PREDICTED robot pose (wrapPatch reads m_X_k after predictMotion, SLAM.cpp:1812-1830)
    double* dxyz = c->G; double* dcov = c->G + 3 * (size_t)N;            // G is free outside the refactorisation
    srukf_launch_landmarks_cartesian(c->stream, c->d, c->X, c->S, dxyz, dcov);                              // PointsMap::xyz (2574)
    srukf_launch_warp_patch(c->stream, c->d, c->p, c->X, dxyz, c->h, c->appR, c->appT, c->appPx, c->app_patch, c->has_app, c->app_tmpl);
    srukf_launch_associate(c->stream, c->d, c->p, c->d_image, c->h, c->Si, c->vis, c->has_app, c->app_tmpl, c->zcur, c->mcur, c->corr);
    double* hs = c->hstage;
    HIPCHK(c, hipMemcpyAsync(hs, c->zcur, sizeof(double) * 2 * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + 2 * N, c->corr, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs + 3 * N, c->mcur, sizeof(int) * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (z) memcpy(z, hs, sizeof(double) * 2 * N);
    if (corr) memcpy(corr, hs + 2 * N, sizeof(double) * N);
    if (matched) memcpy(matched, hs + 3 * N, sizeof(int) * N);
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
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int dim = c->d.n, ld = c->d.np;
    const int Na = dim + 3 * K, L = 2 * Na + 1, dimn = dim + 6 * K;                          // 827-828
    srukf_ctx* c2 = nullptr;
    side_stream_lend(c);
    int rc = srukf_create(&c2, c->d.N + K, &c->p, c->device, c->stream);
    if (rc) { c->err = std::string("add_landmarks: ") + g_create_error; return rc; }
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
        hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c2->fs, 0, 1);
        hipLaunchKernelGGL(k_refactor_reset, dim3((ldn + 255) / 256), dim3(256), 0, c->stream, ldn, c2->theta, c2->fs, 1);
        hipLaunchKernelGGL(k_sym_permute, dim3(ldn), dim3(256), 0, c->stream, dimn, ldn, c2->G, ldn, c2->Gbak, d_perm);      // Pi (A^T A) Pi^T
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
    if (e != hipSuccess) { srukf_destroy(c2); c->err = std::string("add_landmarks: ") + hipGetErrorString(e); return SRUKF_ERR_HIP; }
    if (c->app_patch) {                                      // the old landmarks keep their appearance records
        rc = ensure_appearance(c2);
        if (rc) { c->err = c2->err; srukf_destroy(c2); return rc; }
        for (int k = 0; k < c->d.N; k++) copy_appearance(c, k, c2, k);
        hipStreamSynchronize(c->stream);
    }
    const int storage = c->storage;
    adopt_context(c, c2);
    rc = srukf_set_storage(c, storage); if (rc) return rc;
    rc = update_null_set(c); if (rc) return rc;
    return srukf_set_new_landmarks(c, K);
}

// the handle keeps its identity when the map changes size: swap the guts of a freshly built context in, keep the
// stream ownership and the profile, destroy the old buffers
static void adopt_context(srukf_ctx* c, srukf_ctx* c2)
{
    const bool own = c->own_stream;
    std::swap(*c, *c2);
    c->own_stream = own; c2->own_stream = false;
    c->profiling = c2->profiling; c->use_graph = c2->use_graph;
    // per-context switches the caller set on the handle survive the rebuild (before srukf_set_storage / update_null_set run on it)
    c->rank_aware = c2->rank_aware; c->debug_allow_mixed = c2->debug_allow_mixed; c->debug_starve = c2->debug_starve; c->dbg = c2->dbg; c->split_off = c2->split_off;
    const int shared = c2->gmw_shared, tenants = c2->shared_tenants;
    memcpy(c->prof_ms, c2->prof_ms, sizeof c->prof_ms); memcpy(c->prof_n, c2->prof_n, sizeof c->prof_n);
    memcpy(c->prof_flops, c2->prof_flops, sizeof c->prof_flops); memcpy(c->prof_bytes, c2->prof_bytes, sizeof c->prof_bytes);
    c2->profiling = false; c2->pev.clear();
    srukf_destroy(c2);
    c->phase = 0;
    if (shared != c->gmw_shared) set_shared(c, shared, tenants);
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
    step_commit_motion(c); step_invalidate(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    srukf_ctx* c2 = nullptr;
    side_stream_lend(c);
    int rc = srukf_create(&c2, N - 1, &c->p, c->device, c->stream);
    if (rc) { c->err = std::string("delete_landmark: ") + g_create_error; return rc; }
    const int nn = n - 6, ldn = c2->d.np;
    std::vector<int> map(nn);
    for (int a = 0; a < nn; a++) map[a] = a < 6 * id ? a : a + 6;
    int* d_map = nullptr;
    if (srukf_dmalloc((void**)&d_map, sizeof(int) * nn) != hipSuccess) { srukf_destroy(c2); c->err = "delete_landmark: out of device memory"; return SRUKF_ERR_NOMEM; }
    hipMemcpy(d_map, map.data(), sizeof(int) * nn, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_refactor_reset, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, c->theta, c->fs, 1);
    srukf_launch_syrk(c->stream, c->d, c->S, c->Ut, 0, 0, c->G, c->fs, c->syrk_tiles, c->n_syrk_tiles, nullptr, c->X, RankArgs{}, nullptr);   // P = S^T S
    hipLaunchKernelGGL(k_gather, dim3((ldn + 255) / 256), dim3(256), 0, c->stream, nn, ldn, c->X, c2->X, d_map);
    for (int slow = 0; slow < 2; slow++) {
        hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c2->fs, 0, 1);
        hipLaunchKernelGGL(k_refactor_reset, dim3((ldn + 255) / 256), dim3(256), 0, c->stream, ldn, c2->theta, c2->fs, 1);
        hipLaunchKernelGGL(k_sym_permute, dim3(ldn), dim3(256), 0, c->stream, nn, ldn, c->G, np, c2->Gbak, d_map);
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
    adopt_context(c, c2);
    rc = srukf_set_storage(c, storage); if (rc) return rc;
    rc = update_null_set(c); if (rc) return rc;
    return srukf_set_new_landmarks(c, k_new);
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

// srukf_debug_set(ctx, "fused_motion", 0): the replay keeps k_motion and k_project as two launches (A/B runs)
static void replay_one_frame(srukf_ctx* c)
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
// the null rows are canonical from here on (a rank-aware frame tail has been issued): captured frames of the other launch sequence are stale
static void set_null_canonical(srukf_ctx* c)
{
    if (c->red_r > 0 && !c->null_canonical) { c->null_canonical = true; drop_graphs(c); }
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
    if (c->storage == SRUKF_STORAGE_F32 && replay_fuse_mode(c)) quantize_state(c);
    c->async_pending = true;
    c->phase = 0;
    HIPCHK(c, hipGetLastError());
    return SRUKF_OK;
}

// One staged frame (index `frame`) through the path that checks the theta clamp on the host and repeats the
// refactorisation column by column when the reference's third pivot candidate would have won — what srukf_update does,
// with the staged inputs.  traj_row: device pointer of this frame's trajectory row, or null.
static int run_staged_frame_exact(srukf_ctx* c, int frame, double* traj_row)
{
    const KDims& d = c->d;
    double* tb = traj_row ? traj_row - (size_t)8 * frame : nullptr;
    hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, frame, 1);
    hipLaunchKernelGGL(k_set_traj, dim3(1), dim3(1), 0, c->stream, c->fs, tb);
    seq_predict_motion(c, nullptr);
    seq_predict_measurement(c, true);
    seq_gain(c, nullptr, nullptr, true);
    seq_refactor(c, 0, d.mp, false, true, false, false);
    int rc = read_fs(c); if (rc) return rc;
    if (c->hfs->clamp_rows > 0) {
        hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(1), 0, c->stream, c->fs, frame, 1);
        hipLaunchKernelGGL(k_set_traj, dim3(1), dim3(1), 0, c->stream, c->fs, tb);
        hipLaunchKernelGGL(k_refactor_reset, dim3((d.np + 255) / 256), dim3(256), 0, c->stream, d.np, c->theta, c->fs, 0);
        HIPCHK(c, hipMemcpyAsync(c->G, c->Gbak, sizeof(double) * (size_t)d.np * d.np, hipMemcpyDeviceToDevice, c->stream));
        ProfScope ps(c, KC_GMW_COL, 0, 0);
        for (int j = 0; j < d.n; j++) srukf_launch_gmw_col(c->stream, d.n, d.np, j, c->p.epsilon, c->G, c->Wf, c->D, c->theta, c->fs, c->S);
        quantize_state(c);
        rc = update_null_set(c); if (rc) return rc;      // the null set is re-derived from the exact factor (see srukf_update)
    } else if (c->storage != SRUKF_STORAGE_F32_MIXED) set_null_canonical(c);
    srukf_launch_traj(c->stream, d, c->X, c->S, c->fs, nullptr, 1);
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    while (done < count) {
        checkpoint(true);
        rc = srukf_run_frames_async(c, first + done, count - done, mode, dt + (size_t)8 * done);
        if (rc == SRUKF_OK) rc = srukf_synchronize(c);
        if (rc != SRUKF_ERR_CLAMP_PENDING) break;
        const int fc = c->clamp_frame_host;                               // absolute index of the first flagged frame
        if (fc < first + done || fc >= first + count) { c->err = "run_frames: flagged frame outside the block"; rc = SRUKF_ERR_HIP; break; }
        checkpoint(false);
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
#define SRUKF_BATCH_GROUPS_MAX 4
// Plans and group streams are kept per host thread AND per device: a stream belongs to the device that was current when it was created, and a thread may run
// batches for filters on several devices (round-4 advisor finding: streams created once on whichever device came first).
// The groups' streams: created together, once per device, so that they sit on different hardware queues whatever the filters' own streams map to (streams that share a
// queue serialise: with the groups on their first filters' streams, 4 + 4 filters ran slower than 4 alone).  They go when the device's last plan goes.
struct BatchDev { BatchPlan* plans[SRUKF_BATCH_GROUPS_MAX] = { nullptr, nullptr, nullptr, nullptr }; hipStream_t streams[SRUKF_BATCH_GROUPS_MAX] = { nullptr, nullptr, nullptr, nullptr }; };
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
static void batch_plan_forget(const srukf_ctx* c)
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
    srukf_launch_syrk_b(st, d, bp->t_syrk, B, c->syrk_head_tiles, c->n_syrk_head_tiles, std::min(np, kr), (n + 255) / 256, (n - r + SRUKF_RANK_COLS - 1) / SRUKF_RANK_COLS);
    srukf_launch_syrk_own_b(st, n, np, bp->t_own, B, 0, d.mp, kr, c->gplan_red.tiles, c->gplan_red.ntiles, Tp);
    int pb = 0;
    for (int j0 = -64; j0 + 64 < np && j0 + 64 <= 64 * Tp; j0 += 64, pb ^= 1) {
        if (!g_dbg_batch_split.load()) {
            srukf_launch_gmw_step64_b(st, n, np, j0, c->p.epsilon, bp->t_step, B, std::max(1, Tp - j0 / 64 - 1), pb);    // rows of the kept pivots only; the last panel: the pass-on row
            continue;
        }
        // split form: A = critical-path workgroups + the panel's slabs (and S rows) once per column block, B = the trailing tiles of the kept rows as plain K = 64
        // updates.  The last pivoted panel has no tiles to update (the pass-on row's values are never used: its S rows come from the slab workgroups).
        srukf_launch_gmw_pivslab_b(st, n, np, j0, c->p.epsilon, bp->t_step, B, pb);
        if (j0 >= 0 && Tp - j0 / 64 - 1 >= 1) srukf_launch_gmw_trail_b(st, np, j0, bp->t_step, B, Tp - j0 / 64 - 1);
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
            HIPCHK(c, srukf_dmalloc(&c->slabW, sizeof(double) * 64 * (size_t)d.np)); HIPCHK(c, srukf_dmalloc(&c->slabL, sizeof(double) * 64 * (size_t)d.np));
            HIPCHK(c, hipMemset(c->slabW, 0, sizeof(double) * 64 * (size_t)d.np)); HIPCHK(c, hipMemset(c->slabL, 0, sizeof(double) * 64 * (size_t)d.np));
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
        hipLaunchKernelGGL(k_set_run, dim3(1), dim3(1), 0, st, c->fs, first, c->async_pending ? 0 : 1, traj);
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
    if (!strcmp(key, "batch_split")) {
        g_dbg_batch_split = value ? 1 : 0;
        for (auto& kv : g_batch_dev)
            for (int grp = 0; grp < SRUKF_BATCH_GROUPS_MAX; grp++) if (kv.second.plans[grp]) { hipSetDevice(kv.first); hipStreamSynchronize(kv.second.streams[grp]); batch_plan_drop_graphs(kv.second.plans[grp]); }
        return SRUKF_OK;
    }
    if (!strcmp(key, "batch_groups")) { if (value < 0 || value > SRUKF_BATCH_GROUPS_MAX) return SRUKF_ERR_BAD_ARG; g_dbg_batch_groups = value; return SRUKF_OK; }
    if (!strcmp(key, "batch_wide")) { g_dbg_batch_wide = value ? 1 : 0; return SRUKF_OK; }
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
    else if (!strcmp(key, "step_fast")) c->dbg.step_fast = value ? 1 : 0;
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
    else if (!strcmp(key, "step_fast")) *value = c->step_fast_frames;          // frames the step-wise API ran on the staged replay's launch sequence / on its own
    else if (!strcmp(key, "step_slow")) *value = c->step_slow_frames;
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
// Diagnostic copy of a device work buffer (synchronises the stream): "Z" (L x mp), "DZ" (np x mp), "sigR" ((L + 1) x 8), "Cmat" (n x 4),
// "Xr1" (4), "Utp" / "P1" (mp x np), "h" (2N), "Si" (4N).  count doubles from the start of the buffer.
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
    else if (!strcmp(key, "gsW")) { src = c->gsW; n = c->gsW ? (long long)c->gs_panels * 64 * d.np : 0; }
    else if (!strcmp(key, "gsL")) { src = c->gsL; n = c->gsL ? (long long)c->gs_panels * 64 * d.np : 0; }
    else if (!strcmp(key, "pans")) { src = (double*)gp.pans; n = gp.pans ? (long long)srukf_gmw_panel_bytes() * gp.T / 8 : 0; }
    else if (!strcmp(key, "sync")) { src = (double*)gp.sync; n = gp.sync ? (long long)srukf_gmw_sync_bytes(gp.T) / 8 : 0; }
    else return false;
    *ptr = src; *cap = n;
    return true;
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
    if (!split_form(c, gp, true)) { c->err = "split_replay: this context does not factor with the split form"; return SRUKF_ERR_SEQUENCE; }
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
int srukf_debug_starve_workers(srukf_ctx* c, int on)
{
    if (!c) return SRUKF_ERR_BAD_ARG;
    c->debug_starve = on ? 1 : 0;
    step_invalidate(c);
    drop_graphs(c);                                    // the captured frames contain one or the other launch sequence
    return SRUKF_OK;
}

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

// ---- stand-alone primitives for the parity tests ------------------------------------------------
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
