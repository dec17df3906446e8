// srukf_ctx.h — what the host-side translation units of libsrukf_hip.so share: the context (struct srukf_ctx), the launchers of the kernel files, the
// helpers behind the C-ABI.  Internal: include/srukf.h is the boundary.
//   srukf_api.hip     context lifetime, state accessors, the step-wise calls (predictMotion / predictMeasurement / KalmanUpdate), storage, profiling read-out
//   srukf_replay.hip  the launch sequences of a frame (seq_*), the rank-aware null set, graph cache, staged replay (srukf_run_frames*)
//   srukf_split.hip   split form of the persistent factorisation: side stream, probe, buffers
//   srukf_batch.hip   batched replay (srukf_run_frames_batch)
//   srukf_map.hip     map changes (srukf_add_landmarks / srukf_delete_landmark) and data association
//   srukf_debug.hip   srukf_debug_*, stand-alone primitives for the parity tests
#pragma once
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
                       const double*, int, const double*, double, const double*, int, const double*, double*);
void srukf_launch_project_motion(hipStream_t, KDims, KWeights, srukf_params, double*, double*, double*, double*, double*, double*, FrameScalars*, RankArgs);
int srukf_gain_part_doubles(int);
void srukf_launch_traj(hipStream_t, KDims, const double*, const double*, FrameScalars*, double*, int);
void srukf_launch_block_cov(hipStream_t, KDims, const double*, int, int, double*, const double*);
void srukf_launch_project_points(hipStream_t, srukf_params, int, const double*, const double*, const double*, const double*, double*);
void srukf_launch_pxy(hipStream_t, KDims, const double*, const double*, double*, const void*, int, KWeights, MeasArgs);
void srukf_launch_pxy2(hipStream_t, KDims, const double*, const double*, double*, double*, const void*, int, int, KWeights, MeasArgs, GainFold);
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
void srukf_launch_gmw_split_fold(hipStream_t, hipStream_t, int, int, double, double*, void*, double*, double*, void*, const void*, int, void*, int, int, double*, double*, const double*, const double*, int);
int srukf_gmw_build_fold_list(int T, int Tp, short* out);
int srukf_gmw_fold_head_tile(int tr, int tc, int head_rows);
int srukf_gmw_fold_head_rows(int Tp);
void srukf_gmw_fold_head_override(int v);
void srukf_launch_gmw_split_alone(hipStream_t, int, int, int, double, double*, void*, double*, double*, void*, const void*, int, void*, int, int, double*, double*);
void srukf_launch_row_energy(hipStream_t, int, int, const double*, double*);
void srukf_launch_rank_diag(hipStream_t, int, int, const double*, const int*, double*);
void srukf_launch_rank_expand(hipStream_t, int, int, int, double, const double*, const double*, const int*, const int*, const double*, void*, const double*, int, double*, double*, double*, double, int, KDims, KWeights, srukf_params, double*, double*, int, const StepExport*, double, unsigned int*, int);
void srukf_launch_project_table(hipStream_t, KDims, KWeights, srukf_params, double*, double*, double*, double*, double*, double*, FrameScalars*, RankArgs, NullSkip);
void srukf_launch_sigr_rows(hipStream_t, KDims, KWeights, const double*, const double*, double*, const FrameScalars*, const int*, int);
void srukf_launch_rank_shadow(hipStream_t, int, int, int, const double*, const int*, double*);
void srukf_launch_rank_canon(hipStream_t, int, int, int, double, const int*, double*);
void srukf_launch_rank_round(hipStream_t, int, int, double*);
void srukf_launch_syrk_own(hipStream_t, int, int, const double*, const double*, int, int, int, double*, void*, const void*, int, int);
int srukf_gmw_register_form(int, int, int, int);
int srukf_pxy2_b_per(int, int);
void srukf_launch_pxy2_b(hipStream_t, KDims, const void*, int, const void*, int, int, KWeights, int);
void srukf_launch_gain_b(hipStream_t, KDims, KWeights, const void*, int, int, double);
void srukf_launch_syrk_b(hipStream_t, KDims, const void*, int, const void*, int, int, int, int);
void srukf_launch_syrk_own_b(hipStream_t, int, int, const void*, int, int, int, int, const void*, int, int);
void srukf_launch_gmw_step64_b(hipStream_t, int, int, int, double, const void*, int, int, int);
void srukf_launch_gmw_pivslab_b(hipStream_t, int, int, int, double, const void*, int, int, int);
void srukf_launch_gmw_trail_b(hipStream_t, int, int, const void*, int, int, int, int);
void srukf_launch_rank_expand_b(hipStream_t, int, int, int, double, const void*, int, double, KDims, KWeights, srukf_params);
int srukf_gmw_head_rows(void);
int srukf_gmw_head_extra_diag(void);
void srukf_launch_gmw_check(hipStream_t, int, int, const double*, const double*, FrameScalars*, const double*, int, double*);
void srukf_launch_gmw_col(hipStream_t, int, int, int, double, const double*, double*, double*, unsigned long long*, FrameScalars*, double*);
void srukf_set_exact_right_looking(int on);
void srukf_warm_exact_path(hipStream_t st, FrameScalars* fs);
int srukf_get_exact_right_looking(void);
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
int srukf_mixed_build_tasks_red(int np, int ue, int krows, int rows_lim, short* out_tasks, int* out_tiles, int* ntiles);
int srukf_mixed_krows(int r);
void srukf_launch_split_bf3(hipStream_t, int, int, int, int, const double*, void*, size_t);
void srukf_launch_syrk_bf3(hipStream_t, int, int, int, int, int, const void*, size_t, const void*, int, const void*, int, double*, double*, void*);
size_t srukf_mixed_part_bytes(int ntasks);
void srukf_launch_cvt_f32(hipStream_t, size_t, const double*, float*);
void srukf_launch_cvt_robot_cols(hipStream_t, int, int, const double*, float*);
void srukf_launch_gain_dx(hipStream_t, int, int, const double*, double*, const double*);
void srukf_launch_syrk32(hipStream_t, int, int, int, const float*, const float*, const void*, int, const void*, int, double*, double*, void*, int);
int srukf_app_patch_stride(void);
int srukf_app_tmpl_stride(void);
}

#define SRUKF_GRAPH_FRAMES 8
#define SRUKF_MAX_TENANTS 4                                    // 4 x (1 pivot + 63 workers with two register tiles each) fill 256 CUs at N = 200
#define SRUKF_NULL_ENERGY 1e-12
#ifndef SRUKF_BATCH_GROUPS_MAX
#define SRUKF_BATCH_GROUPS_MAX 4
#endif
//                                                          // groups of filters srukf_run_frames_batch runs side by side, each on a stream of its own

enum KClass { KC_MOTION = 0, KC_PROJECT, KC_STATS, KC_PXY, KC_GAIN, KC_SYRK, KC_GMW_TRAIL, KC_GMW_PERSIST, KC_GMW_CHECK,
              KC_GMW_COL, KC_RANK_EXPAND, KC_PROJECT_MOTION, KC_PROJECT_TABLE, KC_PXY2, KC_MISC, KC_COUNT };
struct ProfEvent { hipEvent_t a, b; int kc; };

// ---- persistent GMW launch (k_gmw_persist): per-matrix-size resources --------------------------------
// nreal: the tiles that hold values (ntiles minus the T - Tp pass-on tiles of the rank-aware form, which ride as a register-free third slot of the first workers)
struct GmwPlan { void* pans = nullptr; void* sync = nullptr; void* tiles = nullptr; int ntiles = 0, nreal = 0, T = 0, Tp = 0, workers = -1, tenants = 1, cus = 0; };

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
    bool dx_lm = false;                    // ... as per-landmark shares in dxk (the gain fold of k_pxy2) instead of k_gain's slice partials in dxp
    // gain fold (round 6: k_gain's work inside k_pxy2 in the staged replay's "fused tail" mode): its sync words (zero between frames), the per-landmark shares of the
    // state update (N x np), how many tile workgroups read the robot columns of the permuted copy
    unsigned int* fold_sync = nullptr; double* dxk = nullptr; int fold_robot_tiles = 0;
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
    float *U32 = nullptr; double* mx_part = nullptr; void *mx_tasks = nullptr, *mx_tiles = nullptr; int mx_ntasks = 0, mx_ntiles = 0;
    // ... in the rank-aware form (round 6): the kept rows of S in permuted column order as float (the permuted copy's values are the stored floats), K <= r, only the
    // macro tiles of the pivoted panels; task list / partials of that shape (mixed_red_ensure)
    float* A32 = nullptr; double* mxr_part = nullptr; void *mxr_tasks = nullptr, *mxr_tiles = nullptr; int mxr_ntasks = 0, mxr_ntiles = 0, mxr_krows = 0, mxr_for_r = 0;
    // ... and which 32 x 32 tiles of it are formed in FP64 after all (k_syrk over this list, behind the fp32 launch): the tile rows / columns that hold the robot block
    // and the map's shared anchor (permuted positions r-4 .. r-1 and 0 .. 2).  The robot's pivots are its variance GIVEN the map — 2e-6 .. 9e-6 of its marginal variance
    // in the benchmark scene, the z coordinate exactly null (scripts/pivot_ratio_probe.py) — and the regression that produces them has weight ~1 on the anchor the
    // robot position was copied into: an fp32-formed entry there (relative error ~1e-7 .. 1e-6) is as large as the pivot itself.  Every other kept pivot is >= 0.1 of
    // its marginal variance
    int* mxr_f64_tiles = nullptr; int mxr_n_f64_tiles = 0;
    // ... and the operands of the bf16-piece form of the product (k_split_bf3 / k_syrk_bf3): X^T = [kept rows of S | U^T]^T as three planes of bf16, [np columns][mxr_ktot]
    unsigned short* mxr_xt = nullptr; size_t mxr_xt_stride = 0; int mxr_ktot = 0;
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
    double *slabW = nullptr, *slabL = nullptr;         // batched replay: two panels' slabs W and L = W / D (2 x 64 x np each: the K = 128 trailing update reads a pair)
    // split form of the persistent factorisation (memory-tile sizes, a filter that has the GPU to itself): the slabs of every pivoted panel (gs_panels x 64 x np
    // each), the side stream the tile launch runs on and the events that fork it off / join it to the filter's stream
    double *gsW = nullptr, *gsL = nullptr; int gs_panels = 0;
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool split_off = false;                // a split-form pair of this context was abandoned (its two launches did not run side by side — e.g. the branches of a captured
                                           // graph sharing a hardware queue): the context keeps to the memory-tile instance of k_gmw_persist (read_fs; srukf_debug_get "split_off")
    double* P1 = nullptr; int* pxy2_tiles = nullptr; int n_pxy2_tiles = 0, pxy2_split_b0 = 0;   // "table" mode: k_pxy2's second K half, its tile list
    int* nskip = nullptr; int ns_full = 0, ns_null = 0, ns_rows = 0;   // NullSkip lists (srukf_device.h): [dirs | nulls | rows] in one buffer
    int* red_head0_tiles = nullptr; int n_red_head0_tiles = 0; // split fold: the k_syrk tiles that stay with the launch in front (block row 0, tile (1, 1)) ...
    void* split_fold_list = nullptr; int n_split_fold = 0;     // ... and the grid of k_gmw_tiles_fold (forming jobs + tile workgroups, row by row: srukf_gmw_build_fold_list)
    double red_head0_flop = 0.0;                               // flop of the tiles in red_head0_tiles (the rest of k_syrk's count moves to the persistent launch's line of the profile)
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
        int step_fuse_export = 1;          // "step_fuse_export": the step-wise fast path's results reach the host from the launches that form them (the statistics' final passes inside
                                           // k_pxy2, the status + robot view from k_block_cov) instead of two k_export launches behind them (0: round 5's first form, for A/B)
        int view_auto = 1;                 // "view_auto": a host that fetched srukf_get_frame_view after its last update gets the view exported with the next update's status (0: never)
        int step_early = 2;                // "step_early": the update submits the next frame's checkpoint copy and (announced odometry) its k_set_step behind its own last launch (0: the predict
                                           // does), 2: and that frame's first launch, k_pxy2, too — srukf_predict_motion for the announced pair then has nothing left to submit
                                           // (the statistics as a launch of their own in front of it were tried: alone they still take 16 us — eight dependent memory round trips —, and
                                           //  16 + 25 us of launches lose to 25 us + the 12-us round trip they were meant to hide)
        int step_spin = 1;                 // "step_spin": the step-wise fast path waits for its two exports by spinning on a pinned flag word (0: hipStreamSynchronize)
        int step_fast = 1;                 // "step_fast": 0: the step-wise API keeps to its own launch sequences (k_motion, k_project, k_meas_*, k_pxy, ...: round 4's path)
        int split_record = 0;              // "split_record": every split-form factorisation first copies its input matrix to Gbak (scripts/split_replay.py)
        int null_canon = 1;                // "null_canon": a factor of the library's own making that was not produced by a rank-aware tail (NEED_REORDER, map changes) gets its null rows rewritten as sqrt(EPSILON) e_k at once
        int split_fold = 1;                // "split_fold": the split form's tile launch forms the tiles of S^T S - U U^T itself (k_gmw_tiles_fold), k_syrk in front keeps block row 0; 0: k_syrk forms everything first
        int gain_fold = 0;                 // "gain_fold": the staged replay's "fused tail" frames form U^T and the state update in the tile epilogue of k_pxy2 (three launches per frame); 0: k_gain
        int mixed_rank = 1;                // "mixed_rank": SRUKF_STORAGE_F32_MIXED runs the rank-aware refactorisation (fp32-formed S^T S - U U^T over the kept rows, FP64 factorisation
                                           // of the kept pivots only); 0: round 2's full-rank form, in which the null pivots divide fp32 noise (the negative study of rounds 2 / 5)
        int mixed_null_ppm = 1;            // "mixed_null_ppm": ... and its null-direction check allows this many 1e-6 of G_aa on top of 1e-12 (an fp32-formed G cannot resolve 1e-12)
        int mixed_bf16 = 0;                // "mixed_bf16": 1: the fp32 products from three bf16 pieces per operand on the bf16 matrix pipe (k_syrk_bf3) instead of the fp32 pipe
                                           // (k_syrk32).  Built and measured in round 6 (N = 500): 168 us + 2 x 19 us of splitting against 187 us + 2 x 9 us of conversion —
                                           // no gain: at one workgroup per CU (the FP64 accumulators beside the fp32 ones: 256 + 65 registers) every 32-row slab waits for its
                                           // operands (~1 us of load latency against 0.64 us of matrix work), so neither form is bound by its matrix pipe; and its error is
                                           // 1.6 x the fp32 pipe's (12.5 against 7.7 eps32 units over fixture g9).  Kept behind the switch, held to the same fixture
        int mixed_f64_robot = 1;           // "mixed_f64_robot": ... with the tiles of the robot block and of the shared anchor in FP64 (mxr_f64_tiles); 0: every kept tile from the fp32 pipe (study)
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
    int* syrk_head_tiles_b = nullptr; int n_syrk_head_tiles_b = 0;      // the same tiles in the batched launch's order (k_syrk_b): one pair of tile columns per XCD, empty slots (-1) where a share is shorter
    FrameScalars* fs = nullptr;
    // staged sequence
    int seqF = 0;
    double *odo_seq = nullptr, *z_seq = nullptr;
    int* m_seq = nullptr;
    // pinned staging
    double* hstage = nullptr; size_t hstage_bytes = 0;
    FrameScalars* hfs = nullptr;
    double* hmeas = nullptr;               // inside the hfs allocation, behind the robot view and the flag word
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
    // The copy for the NEXT frame is submitted by the update that ends this one, right behind its last launch (into the second pair of buffers; pair and event swap
    // when the frame turns out clean): the next srukf_predict_motion then finds its checkpoint made (ck_valid) and submits its first launch at once
    double *ckS2 = nullptr, *ckX2 = nullptr; hipEvent_t ck_e3 = nullptr; bool ck_valid = false;
    bool ck3_inflight = false;             // ... and until then the copy is in flight on ck_stream with nothing but ck_e3 to wait on (step_ck_join)
    unsigned spin_ok = 0;                  // successful spin waits (step_wait_export queries the stream every 256th)
    bool pre_issued = false; double pre_odo[6] = { 0, 0, 0, 0, 0, 0 };   // the NEXT frame's first launch (k_pxy2) went out behind this frame's tail, for this odometry pair
    bool next_pose_pending = false;        // the next k_gain launch carries next_odo[3..5] as the sequence's third pose (no launch of its own)
    bool setstep_done = false; double setstep_odo[6] = { 0, 0, 0, 0, 0, 0 };   // k_set_step for the announced next frame went out behind this frame's tail (poses: prev, cur)
    hipStream_t ck_stream = nullptr; hipEvent_t ck_e1 = nullptr, ck_e2 = nullptr; bool ck_pending = false;   // the copy of the state before the frame runs BESIDE the frame's
                                           // first launch on a stream of its own (it only has to be complete before k_gain touches S): step_ck_join
    unsigned long long meas_seq = 0;       // != 0: k_pxy2's statistics jobs mirror h | Si | visible into hstage and raise the flag word with this number (srukf_predict_measurement waits for it)
    bool step_export_attached = false;     // the last rank_expand carried step_export
    int* export_cnt = nullptr;             // 64 x 64 ints (zero): first-level counters of the exporting launch (StepExport::cnt)
    StepExport step_export = {};           // dst != null: the next rank_expand is the step-wise fast path's and exports the frame's status + robot view itself
    bool mirror_next = false;              // the next seq_pxy is the step-wise fast path's: its MeasArgs carry the host mirror
    unsigned long long step_seq = 0;       // sequence number of the step-wise fast path's exports: the host spins on a pinned word (behind the robot view) that receives it
    double* hview = nullptr; size_t hview_doubles = 0;      // pinned: xyz (3N) | cov (9N) | X (n) of the state an update of the fast path left, when the host is known to ask for it
    bool view_auto = false; int view_unused = 0;   // the host called srukf_get_frame_view after its last update -> the following updates export the view with their status; three views nobody read end it
    int view_hits = 0;
    bool view_cached = false;              // *hview is the view of the CURRENT state (same lifetime as robot_cached)
    bool robot_cached = false;             // the 20 doubles behind *hfs hold P4 and the pose of the CURRENT state (fast path: fetched with the frame's status)
    bool f32_stale = false;                // fp32 storage: X32 / S32 (srukf_get_state_f32) are behind the rounded fp64 working copies (refreshed on demand)
    int step_fast_frames = 0, step_slow_frames = 0;   // srukf_debug_get "step_fast" / "step_slow"
    double* spare_stage = nullptr; size_t spare_stage_bytes = 0;   // pinned staging of the context retired last (a handle's retired contexts keep none: pinning 11.8 MB costs ~2.5 ms)
    int split_fold_seqs = 0;               // split-form pairs enqueued (or captured) with the split fold: "split_fold_seqs"
    int fold_seqs = 0;                     // frame sequences enqueued (or captured) with the gain fold: "fold_seqs" (tests: the switch took effect)
    int exact_frames = 0;                  // staged frames srukf_run_frames repeated on the exact column path (flagged: theta clamp, a skipped direction that is not null, an abandoned launch): "exact_frames"
    bool async_pending = false;
    // Map changes rebuild the context behind the handle (srukf_add_landmarks / srukf_delete_landmark: adopt_context).  A rebuilt context used to be destroyed and the
    // next one created from nothing — ~0.3 ms of allocations, plans and tile tables, 0.5 - 2.4 ms of frees (pinned host memory among them) per map change at N = 200,
    // where the reference's map changes every few frames (SLAM.cpp:552-562, 2443-2460) and N only moves by +- 1.  The handle keeps the last few contexts it outgrew
    // (shape, device, stream and parameters identical when N comes back) and revives one instead of creating it: ctx_obtain / ctx_retire (srukf_api.hip)
    std::vector<srukf_ctx*> retired;
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

namespace srukf_impl {

// fp32 storage of the state with everything that goes with it in "fused tail" mode (the tail and the state update round what they write).  The mixed-precision downdate in
// its rank-aware form (round 6) IS that mode with one difference: where the staged frame forms S^T S - U U^T over the kept rows (seq_refactor, red_perm branch)
inline bool storage_f32_like(const srukf_ctx* c) { return c->storage == SRUKF_STORAGE_F32 || (c->storage == SRUKF_STORAGE_F32_MIXED && c->dbg.mixed_rank && c->A32); }

extern const char* const kclass_name[KC_COUNT];
extern thread_local std::string g_create_error;
// srukf_debug_set switches (process-wide; srukf_debug.hip)
extern std::atomic<int> g_dbg_gmw_persist, g_dbg_gmw_fused, g_dbg_rank_fused, g_dbg_rank_fold, g_dbg_rank_aware, g_dbg_graphs, g_dbg_mem_split, g_dbg_shared_tenants;
extern std::atomic<int> g_dbg_batch_wide, g_dbg_batch_groups, g_dbg_batch_split, g_dbg_head_fold_free;
extern std::atomic<int> g_dbg_timing, g_dbg_fold_head, g_dbg_fold_force, g_dbg_ctx_keep, g_dbg_batch_xcd, g_dbg_batch_k128;      // process-wide measurement switches (srukf_debug_set(0, "timing" / "fold_head" / "fold_force", v)): the library reads no environment variable

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- device memory: the stream-ordered pool of the device (srukf_api.hip) ----
void srukf_pool_init();
hipError_t srukf_dmalloc_raw(void** p, size_t bytes);
template <class T> inline hipError_t srukf_dmalloc(T** p, size_t bytes) { return srukf_dmalloc_raw((void**)p, bytes); }
hipError_t srukf_dfree(void* p);
// the same, ordered on a context's own stream (no synchronisation: everything that touches the block is on that stream)
template <class T> inline hipError_t srukf_dmalloc_on(T** p, size_t bytes, hipStream_t st) { srukf_pool_init(); return hipMallocAsync((void**)p, bytes ? bytes : 8, st); }
hipError_t srukf_dfree_on(void* p, hipStream_t st);

// ---- small kernels of the host layer behind launchers (srukf_replay.hip) ----
void launch_refactor_reset(hipStream_t st, int np, unsigned long long* theta_bits, FrameScalars* fs, int reset_stats);
void launch_set_seq(hipStream_t st, FrameScalars* fs, const double* odo_seq, int seqF, double a1, double a2, double a3, double a4);
void launch_set_frame(hipStream_t st, FrameScalars* fs, int frame, int clear_clamp);
void launch_set_traj(hipStream_t st, FrameScalars* fs, double* traj_base);
void launch_set_run(hipStream_t st, FrameScalars* fs, int frame, int clear_clamp, double* traj_base);
void launch_set_step(hipStream_t st, FrameScalars* fs, double* odo, int seqF, double a1, double a2, double a3, double a4, int fresh, const double poses[9]);
// device -> pinned host memory, two segments of 8-byte words; flag (pinned too, may be null) receives seq behind the data
void launch_export(hipStream_t st, const void* a, size_t bytes_a, const void* b, size_t bytes_b, void* host_pinned, unsigned long long* flag = nullptr, unsigned long long seq = 0);
void launch_set_frame_control(hipStream_t st, FrameScalars* fs);
void launch_commit_motion(hipStream_t st, int n, int ld, double* X, double* S, const double* Cm, const FrameScalars* fs, double* A, const int* iperm, int rk);
void launch_sym_permute(hipStream_t st, int n, int ld, const double* src, int lds, double* dst, const int* map);
void launch_gather(hipStream_t st, int n, int ld, const double* src, double* dst, const int* map);
void launch_zero_rows(hipStream_t st, int ld, int r0, double* A);
void launch_quantize(hipStream_t st, int n, int ld, double* S, double* X, float* S32, float* X32);

// ---- plans, tables (srukf_api.hip) ----
void gmw_plan_destroy(GmwPlan& g, hipStream_t st = nullptr);
int gmw_plan_create(GmwPlan& g, int np, hipStream_t st, int Tp = 0, int tenants = 1);
void host_weights(int Na, const srukf_params& p, KWeights& w);
std::vector<int> build_tile_table(int n_own, int n_other, bool upper, bool own_is_row, int k_index /* 0: K grows with tile.x, 1: with tile.y */);
void prof_collect(srukf_ctx* c);
void adopt_context(srukf_ctx* c, srukf_ctx* c2);
int ctx_obtain(srukf_ctx* handle, srukf_ctx** out, int N);     // a context for N landmarks with the handle's device / stream / parameters: a retired one revived, or srukf_create
void ctx_retire(srukf_ctx* handle, srukf_ctx* old);

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

// ---- the step-wise fast path's bookkeeping (srukf_api.hip) ----
void step_commit_motion(srukf_ctx* c);
unsigned long long* step_flag(srukf_ctx* c);          // the pinned word exports raise behind their data
int step_wait_export(srukf_ctx* c, unsigned long long seq);   // spin on it ("step_spin"), or synchronise the stream
void step_invalidate(srukf_ctx* c);
void step_state_replaced(srukf_ctx* c);
void step_ck_join(srukf_ctx* c);

// ---- launch sequences (srukf_replay.hip) ----
void quantize_state(srukf_ctx* c);
RankArgs rank_args(const srukf_ctx* c, bool prep_next = false, bool dzperm = false, bool f32round = false);
NullSkip null_skip(const srukf_ctx* c);
const double* take_xr1(srukf_ctx* c);
void seq_predict_fused(srukf_ctx* c, int mode);
void seq_predict_motion(srukf_ctx* c, const double* odo_pair_dev);
void seq_predict_measurement(srukf_ctx* c, bool fused_stats);
void shadow_rebuild(srukf_ctx* c);
bool gmw_plan_persists(const srukf_ctx* c, const GmwPlan& gp);
bool gmw_use_persist(const srukf_ctx* c);
int gmw_persist_mode();
int plan_tenants(const srukf_ctx* c);
int gate_limit(const srukf_ctx* c);
void rank_expand(srukf_ctx* c, bool frame_tail, bool table = false, bool fuse = false);
bool replay_red_fused(const srukf_ctx* c);
bool replay_red_perm(const srukf_ctx* c);
int replay_motion_mode(const srukf_ctx* c);
bool replay_fuse_mode(const srukf_ctx* c);
bool head_fold_ok(const srukf_ctx* c);
int gmw_fused_mode();
int rank_fused_mode();
int rank_fold_mode();
void seq_refactor(srukf_ctx* c, int ub, int ue, bool slow, bool keep_backup, bool need_reset, bool frame_tail, bool table = false, bool fuse = false);
void launch_gmw_fast(srukf_ctx* c, double* Gbuf, double* Sout, bool reduced = false, bool fold = false);
bool split_fold_ok(const srukf_ctx* c);
void run_gmw(srukf_ctx* c, double* Gbuf, double* Sout, bool slow);
int refactor_reorder(srukf_ctx* c, int ub, int ue);
void seq_pxy(srukf_ctx* c, bool fused_stats, bool fused_motion = false, bool table = false, bool preamble = false, bool fmode = false, bool fold = false);
void seq_gain_only(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_motion = false, bool table = false, bool fmode = false);
void seq_gain(srukf_ctx* c, const double* z_dev, const int* m_dev, bool fused_stats, bool fused_motion = false, bool table = false, bool preamble = false, bool fmode = false);
int update_null_set(srukf_ctx* c);
int mixed_red_ensure(srukf_ctx* c);
void drop_graphs(srukf_ctx* c);
void exact_path(srukf_ctx* c, const double* Gbuf, double* Sout);      // the exact column path for Gbuf -> Sout (D, theta, clamp count as side effects)
void set_null_canonical(srukf_ctx* c);
void canonicalize_null_rows(srukf_ctx* c);      // after update_null_set on a factor of the library's own making (NEED_REORDER, map changes): the next frame may take the fast path
int read_fs(srukf_ctx* c);
int read_fs_host(srukf_ctx* c);
int set_shared(srukf_ctx* c, int shared, int tenants);
void replay_one_frame(srukf_ctx* c);

// ---- split form of the persistent factorisation (srukf_split.hip) ----
bool split_form(const srukf_ctx* c, const GmwPlan& gp, bool ignore_starve = false);
void split_ensure(srukf_ctx* c, const GmwPlan& gp);
void side_stream_lend(srukf_ctx* c);

// ---- batched replay (srukf_batch.hip) ----
void batch_plan_forget(const srukf_ctx* c);
void batch_drop_all_graphs();

}  // namespace srukf_impl
