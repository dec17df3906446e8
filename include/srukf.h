/*
 * srukf.h — C-ABI of the MI355X-native SRUKF predict/update hot path.
 *
 * This is the drop-in boundary for ONE path of junliu111/CV-MonoSLAM: the per-frame
 * square-root unscented Kalman filter inside class CSLAM (reference MonoSLAM/SLAM.cpp,
 * MonoSLAM/SLAM.h:118-398).  The reference has no plugin/FFI interface: CSLAM is a concrete
 * class embedded by value in the MFC view (MonoSLAMView.h:44) whose methods are called and
 * whose public fields are read directly.  The entry points below are what a CSLAM facade
 * (cv-monoslam_amd/host/cslam.hpp) binds to; each one cites the reference member(s) it
 * replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns an int: SRUKF_OK (0) or a negative srukf_status; never aborts,
 *     never prints (the reference prints + system("pause"), SLAM.cpp:297,324,2702,3526).
 *   - one context = one filter = one HIP stream; a context is not thread-safe, independent
 *     contexts are fully concurrent (multi-GPU / Monte-Carlo model).
 *   - the caller owns every pointer it passes; nothing is retained past the call.
 *   - all matrices are row-major fp64.  State layout (SLAM.cpp:226-231,1371,2427-2435):
 *       X = [ N x (xi yi zi theta phi rho) | x y z theta ]      n = 6N+4
 *       S = n x n upper triangular, P = S^T S                   (SLAM.h:272, SLAM.cpp:2118,2404)
 *   - there is NO CPU fallback: if the HIP device/kernels are unavailable srukf_create fails
 *     with SRUKF_ERR_NO_DEVICE.
 */
#ifndef SRUKF_H_
#define SRUKF_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SRUKF_ABI_VERSION 6


typedef enum srukf_status {
    SRUKF_OK                =  0,
    SRUKF_ERR_BAD_ARG       = -1,  /* NULL pointer, negative size, unknown enum            */
    SRUKF_ERR_DIM_MISMATCH  = -2,  /* sizes inconsistent with the context                  */
    SRUKF_ERR_HIP           = -3,  /* a HIP runtime call failed; see srukf_last_error      */
    SRUKF_ERR_NO_DEVICE     = -4,  /* no gfx950 device / code object not loadable          */
    SRUKF_ERR_SEQUENCE      = -5,  /* call order violated (e.g. update before predict)     */
    SRUKF_ERR_UNSUPPORTED   = -6,  /* feature outside the built hot path   */
    SRUKF_ERR_CLAMP_PENDING = -7,  /* async replay hit the GMW theta-clamp; rerun sync      */
    SRUKF_ERR_NOMEM         = -8
} srukf_status;

/* Tunables of the hot path.  Defaults = the reference's debug-model values
 * (SLAM.cpp:172-198, 221-224, 238, 241-242, 263-264, 329-337, 52, 3186). */
typedef struct srukf_params {
    double cam_dx, cam_dy;          /* 0.0028, 0.0028          SLAM.cpp:329-330 */
    double cam_cx, cam_cy;          /* 310.1129, 236.7526      SLAM.cpp:331-332 */
    double cam_k1, cam_k2;          /* 1e-4, 0                 SLAM.cpp:333-334 */
    double cam_f;                   /* 2.1735                  SLAM.cpp:335     */
    double image_w, image_h;        /* 640, 480                SLAM.cpp:312-313 */
    double a1, a2, a3, a4;          /* 8,8,8,8                 SLAM.cpp:195-198 */
    double sigma_measure;           /* 3.0 (Qt = 3*I2)         SLAM.cpp:189,238 */
    double rho0, sigma_rho;         /* 1/3, 1/6                SLAM.cpp:173,190 */
    double sigma_x, sigma_y, sigma_z, sigma_theta; /* .02 .02 .005 .02  SLAM.cpp:221-224 */
    double epsilon;                 /* 1e-13 (EPSILON)         SLAM.cpp:52      */
    double ut_alpha, ut_beta;       /* 1e-3, 2                 SLAM.cpp:263-264 */
    int    weight_type;             /* 0 (FLAG_4_WEIGHT1)      SLAM.cpp:241     */
    int    noise_type;              /* 0 (FLAG_4_NOISE1); others unsupported  SLAM.cpp:242 */
    int    newton_iters;            /* 100                     SLAM.cpp:3186    */
    int    reserved_;
} srukf_params;

/* How the square-root covariance is maintained in srukf_update.
 *  SEQUENTIAL: the reference's structure — for every matched landmark and each of its two
 *              measurement columns u:  S <- gmw(S^T S - u u^T)   (SLAM.cpp:2066-2095, 2116-2154).
 *  BATCHED   : one  S <- gmw(S^T S - U U^T)  per frame (gains do not depend on the S updates,
 *              SLAM.cpp:2070-2080; see DESIGN.md for the equivalence evidence).               */
typedef enum srukf_update_mode { SRUKF_UPDATE_SEQUENTIAL = 0, SRUKF_UPDATE_BATCHED = 1 } srukf_update_mode;

/* SLAM.cpp:36-37 FLAG_4_NEED_REORDER(0) / FLAG_4_NEEDNOT_REORDER(1) */
typedef enum srukf_reorder { SRUKF_NEED_REORDER = 0, SRUKF_NEEDNOT_REORDER = 1 } srukf_reorder;

/* Precision of the filter state that lives from frame to frame (arithmetic is fp64 in both).  F32 is BASELINE
 * configs[4] ("500 landmarks fp32 SRUKF ..., tolerance study"): X and S are kept as float and every frame computes
 * from exactly those rounded values. */
/* F32_MIXED adds the mixed-precision downdate of that config: the covariance that is re-factorised, S^T S - U U^T
 * (SLAM.cpp:2118-2120, 2149), is formed on the fp32 matrix pipe from the fp32 state and U^T rounded once, in K chunks of
 * <= 1024 summed in FP64; pivots, diagonal blocks and trailing updates of the modified Cholesky stay FP64.  BATCHED
 * updates only use it (a SEQUENTIAL / single-column refactor keeps the FP64 contraction).
 * Round 6: offered at every epsilon, the reference's 1e-13 included (rounds 2 - 5 refused it below 1e-9).  Its fp32 accumulators are flushed into FP64 every 32
 * rows of K, and in the rank-aware form (the default wherever a null set exists) only the kept pivots are factored and the tiles of the robot block and of the
 * map's shared anchor are formed in FP64: within ~1e-6 m of the fp64 filter over 3 000 frames at N = 500 (7.3e-7 / 1.2e-6 in two runs) (fp32 storage with FP64 arithmetic: 6.8e-7;
 * DESIGN.md, row g).  On MI355X the FP32 matrix peak is 2 x the FP64 one and the mode's extra passes use that up: it is NOT faster than SRUKF_STORAGE_F32. */
typedef enum srukf_storage { SRUKF_STORAGE_F64 = 0, SRUKF_STORAGE_F32 = 1, SRUKF_STORAGE_F32_MIXED = 2 } srukf_storage;

typedef struct srukf_ctx srukf_ctx;   /* opaque; owns all device buffers + pinned staging */

int  srukf_abi_version(void);

/* Fill *p with the reference defaults.  Replaces the constant block of
 * CSLAM::initializeParameters (SLAM.cpp:158-343). */
int  srukf_default_params(srukf_params* p);

/* Create a filter for N >= 0 landmarks (n = 6N+4) on HIP device `device`.
 * `stream` is an existing hipStream_t to launch on (so a host can time with its own events) or
 * NULL to let the context create one.  Replaces CSLAM::CSLAM + the allocation side of
 * initializeParameters (SLAM.cpp:21-57, 226-239).  Initial state: robot X=0, S=diag(sigma_x,
 * sigma_y, sigma_z, sigma_theta) in the robot block, landmark rows zero until srukf_set_state. */
int  srukf_create(srukf_ctx** out, int n_landmarks, const srukf_params* p, int device, void* stream);
int  srukf_destroy(srukf_ctx* ctx);                                  /* CSLAM::~CSLAM, SLAM.cpp:64-78      */
int  srukf_reset(srukf_ctx* ctx);                                    /* resetAllParameters, SLAM.cpp:3090-3128 */
const char* srukf_last_error(const srukf_ctx* ctx);                  /* NULL ctx -> last create() error    */

/* Upload / download the filter state (host pointers).  m_X_k / m_S_k mirrors, SLAM.h:271-272.
 * S is n*n row-major; only the upper triangle is read.  S may be NULL in get. */
int  srukf_set_state(srukf_ctx* ctx, const double* X, const double* S);
int  srukf_get_state(srukf_ctx* ctx, double* X, double* S);
/* Same, from/to DEVICE pointers on the context's device (e.g. a tensor that arrived by RCCL
 * broadcast); S_ld = row stride of the device matrix in elements (>= n). */
int  srukf_set_state_device(srukf_ctx* ctx, const double* dX, const double* dS, int S_ld);
int  srukf_get_state_device(srukf_ctx* ctx, double* dX, double* dS, int S_ld);

/* Robot pose X[n-4:n] and the 4x4 robot block of P = S^T S.  Replaces the reads at
 * SLAM.cpp:2963-2965, 3539-3556 and OpenGlDisplay.cpp:386-391 without forming the full
 * m_P_k = S^T S of SLAM.cpp:2404. */
int  srukf_get_robot(srukf_ctx* ctx, double pose4[4], double P4[16]);
/* Landmark k: its 6 state rows and the 6x6 diagonal block of P (SLAM.cpp:2427-2432, 2748). */
int  srukf_get_landmark_block(srukf_ctx* ctx, int k, double X6[6], double P66[36]);
/* getFeatureCartesianInformation (SLAM.cpp:2721-2751) for every landmark at once: xyz[3N] = anchor + m(theta, phi)/rho,
 * cov[9N] = J P66 J^T (row-major 3x3 each) with P66 the landmark's block of P = S^T S.  What the OpenGL view reads on
 * every paint (OpenGlDisplay.cpp:449-583) without ever forming the n x n P.  Either pointer may be NULL. */
int  srukf_get_landmarks_cartesian(srukf_ctx* ctx, double* xyz, double* cov);

/* What CSLAM::SLAM() refreshes for the display after every update (updateFeaturesInformation 2397-2621: m_X_k, xyz / Cartesian covariance of every landmark;
 * recordRobotInformation 3539-3556: the robot block) in ONE device round trip instead of one per accessor: X[n], xyz[3N], cov[9N], pose4[4], P4[16]; any may be NULL.
 * A host that calls it right after srukf_update gets the view exported with the status of its following updates (the call is then a copy from pinned memory; three
 * views nobody fetched end that).  Same values either way. */
int  srukf_get_frame_view(srukf_ctx* ctx, double* X, double* xyz, double* cov, double pose4[4], double P4[16]);

/* Full covariance m_P_k = S^T S (SLAM.cpp:2404), n*n row-major, for hosts that want it. */
int  srukf_get_covariance(srukf_ctx* ctx, double* P);

/* ---- the per-frame numeric seam (CSLAM::SLAM, SLAM.cpp:87-112) -------------------------- */

/* predictMotion numeric tail (SLAM.cpp:1430-1465): control from two consecutive odometry
 * poses (x, y, theta), sigma points, motion model, re-triangularisation of S.            */
int  srukf_predict_motion(srukf_ctx* ctx, const double odo_prev[3], const double odo_cur[3]);
/* Optional look-ahead for hosts that know their odometry in advance (the reference does: loadOdometryData reads the whole file into m_odoXY / m_odoTheta
 * before the first frame, SLAM.cpp:363-496): the pair srukf_predict_motion will be called with for the NEXT frame, announced any time before this frame's
 * srukf_update.  The update's frame tail then also projects the next frame's sigma points (as the staged replay's tail does) and srukf_update submits the
 * next frame's first launch behind its own: the next srukf_predict_motion has nothing left to launch and srukf_predict_measurement only waits for the
 * statistics.  A next call whose pair differs from the announced one (or a state / map change in between) has that work ignored and projects again:
 * results never depend on the hint.  No reference counterpart (predictMotion reads m_odoXY[counter], SLAM.cpp:1444-1450). */
int  srukf_predict_motion_next(srukf_ctx* ctx, const double odo_prev[3], const double odo_cur[3]);

/* predictMeasurement (SLAM.cpp:1604-1608): h[2N] predicted pixels (m_allPredictSet),
 * Si[4N] per-landmark 2x2 upper-triangular sqrt innovation covariance (PointsMap::Si),
 * visible[N] (PointsMap::isVisible, SLAM.cpp:1727).  Outputs are host pointers, any may be NULL. */
int  srukf_predict_measurement(srukf_ctx* ctx, double* h, double* Si, int* visible);

/* KalmanUpdate (SLAM.cpp:2048-2104): z[2N] matched pixels (PointsMap::matchLocation),
 * matched[N] (PointsMap::isMatching).  reorder = SRUKF_NEEDNOT_REORDER for steady state;
 * SRUKF_NEED_REORDER on the frames that follow a landmark addition (the reference passes
 * `m_nAddings != 0 ? NEED_REORDER : NEEDNOT_REORDER`, SLAM.cpp:2084-2090): the rank-aware path
 * GSLCholeskyUpdate 2122-2138 + CholeskyDecompositionWithPivoting 2158-2179 with
 * m_covRank = n - 3*K_new; K_new must have been set with srukf_set_new_landmarks
 * (SRUKF_ERR_SEQUENCE otherwise). */
int  srukf_update(srukf_ctx* ctx, const double* z, const int* matched, int reorder, int mode);

/* m_nFilters (SLAM.cpp:826-830, 2126-2131): the number of landmarks added by the last augmentation, i.e. the LAST
 * K_new landmarks of the map.  Defines the permutation of getPermutationMatrix (SLAM.cpp:1303-1334) and the rank
 * n - 3*K_new used by SRUKF_NEED_REORDER updates.  0 clears it. */
int  srukf_set_new_landmarks(srukf_ctx* ctx, int K_new);

/* integrateFeaturesInformation, numeric part (SLAM.cpp:826-871, with passSigmaThroughMapingFunction 1177-1250,
 * QrAndCholeskyForInitilization 1260-1300, getPermutationMatrix 1303-1334): K new landmarks first seen at the
 * distorted pixels uv[K][2] (m_keyPoints[i].pt) are joint-initialised and appended to the map (inverse depth rho0,
 * sigma_rho; pixel noise sigma_measure).  The context grows to N + K landmarks in place — the handle stays valid,
 * staged sequences are dropped — and K_new = K is armed for the SRUKF_NEED_REORDER update that follows.  A context
 * created with N = 0 holds the robot block only (initializeParameters 221-231) and is the reference's frame-1 state. */
int  srukf_add_landmarks(srukf_ctx* ctx, int K, const double* uv);

/* ---- data association on the device (wrapPatch 1803-1906, dataAssociation 1915-2009, calculateCrossCorrelation
 * 3141-3166) ----
 * srukf_set_landmark_appearance: the PointsMap fields set when landmark k was created (SLAM.cpp:920-925): patch =
 *   initPatch, the 21 x 21 gray window image(Rect(cvRound(u) - 10, cvRound(v) - 10, 21, 21)) row-major; R = initRotation
 *   (Rwc, 3x3 row-major); t = initTrans (camera position); px = initPixel (distorted).  Records follow their landmark
 *   through srukf_add_landmarks / srukf_delete_landmark.
 * srukf_associate: between srukf_predict_measurement and srukf_update.  gray = the image_h x image_w frame (uchar,
 *   row-major).  Warps every init patch to the current pose (matchPatch, persistent as in the reference) and searches
 *   the chi-square gated window around the predicted pixel for the best normalised cross correlation; a landmark
 *   matches when it exceeds THRESHOLD_MATCH_PATCH = 0.8.  Out (host, any may be NULL): z[2N] matchLocation,
 *   matched[N] isMatching, corr[N] the best correlation.
 * srukf_get_match_patch: the 17 x 17 matchPatch of landmark k (row-major, 289 bytes). */
int  srukf_set_landmark_appearance(srukf_ctx* ctx, int k, const unsigned char* patch, const double R[9], const double t[3], const double px[2]);
int  srukf_associate(srukf_ctx* ctx, const unsigned char* gray, double* z, int* matched, double* corr);
int  srukf_get_match_patch(srukf_ctx* ctx, int k, unsigned char* out);

/* Select the storage precision (default SRUKF_STORAGE_F64).  With SRUKF_STORAGE_F32 the state is rounded to float at
 * the end of every refactorisation (and by srukf_set_state); srukf_get_state returns those values widened to double,
 * srukf_get_state_f32 the float arrays themselves (X[n], S[n*n] row-major). */
/* How the filter shares the GPU.  No reference counterpart (the reference is single-threaded host code).
 * SRUKF_GPU_EXCLUSIVE (default, any value other than the two below): this filter has the GPU to itself while it runs; the
 *   refactorisation is ONE persistent launch that may use every CU and whose workgroups all have to be resident.
 * SRUKF_GPU_SHARED: several filters replay concurrently on this GPU (one context and stream each).  The persistent launch keeps
 *   to half the CUs and starts behind an admission gate that lets at most two such launches of the process run at a time, so
 *   the launches that run together are always resident together (measured at N = 200: one filter 4 500 frames/s, two 8 000,
 *   three 10 600 aggregate).  Each filter's stream should have a hardware queue of its own: the ROCm runtime maps streams onto
 *   GPU_MAX_HW_QUEUES = 4 queues by default, and streams that share one serialise — export GPU_MAX_HW_QUEUES=8 before the process
 *   initialises HIP when it holds more than four streams.
 * SRUKF_GPU_SHARED_PER_PANEL: one launch per 64-row panel, no residency assumption at all (other kernels of unknown size and
 *   duration on the GPU).  A persistent launch that cannot get its workgroups in time gives up (bounded waits), its frame is
 *   repeated on the exact path, and the context switches to this mode by itself. */
#define SRUKF_GPU_SHARED 0
#define SRUKF_GPU_EXCLUSIVE 1
#define SRUKF_GPU_SHARED_PER_PANEL 2
int  srukf_set_exclusive(srukf_ctx* ctx, int exclusive);
/* Rank-aware refactorisation (default on).  The anchors of landmarks initialised in one batch are copies of one robot
 * position (SLAM.cpp:1223, 1247) and stay identical random variables for the life of the filter, so 3 (K - 1) pivots per
 * batch are null by construction: the reference's modified Cholesky meets them as c_jj = 0 and clamps them to EPSILON
 * (SLAM.cpp:2279-2285).  Whenever a state arrives (srukf_set_state*, map changes) the rows of S with energy < 1e-12 are
 * taken as such directions; the refactorisation then pivots only the others (same relative order: the kept rows are the
 * reference's rows) and writes sqrt(EPSILON) e_k for the rest, which changes no entry of P = S^T S by more than 1e-12.
 * Every frame verifies that the skipped directions are still null in its S^T S - U U^T and otherwise takes the exact path.
 * srukf_null_directions: how many pivots are skipped at present.  No reference counterpart. */
int  srukf_set_rank_aware(srukf_ctx* ctx, int on);
int  srukf_null_directions(srukf_ctx* ctx);
int  srukf_set_storage(srukf_ctx* ctx, int storage);
int  srukf_get_state_f32(srukf_ctx* ctx, float* X, float* S);

/* deleteOneFeature, numeric part (SLAM.cpp:2637-2668): landmark `id` (0-based position in the state) leaves the map:
 * X and S lose its 6 entries / rows / columns and the removed rows are folded back in (GSLCholeskyUpdate with
 * FLAG_4_UPDATING), i.e. S becomes the factor of the remaining block of P = S^T S.  The context shrinks to N - 1
 * landmarks in place. */
int  srukf_delete_landmark(srukf_ctx* ctx, int id);

/* ---- benchmark seam: whole frames with inputs pre-staged in HBM --------------------------- */

/* Stage F frames of inputs on the device: odo[(F+1)*3] odometry poses (frame f uses f, f+1),
 * z[F*2N] measurements, matched[F*N].  Host pointers. */
int  srukf_stage_sequence(srukf_ctx* ctx, int n_frames, const double* odo, const double* z, const int* matched);
/* Run frames [first, first+count) back to back (predictMotion + predictMeasurement +
 * KalmanUpdate per frame, no host round trip in between), asynchronously on the context's
 * stream.  traj, if not NULL, is a DEVICE buffer of count*8 doubles receiving per frame
 * (x, y, z, theta, P00, P01, P10, P11) — the RobotPath.txt columns of SLAM.cpp:3549-3556. */
int  srukf_run_frames_async(srukf_ctx* ctx, int first, int count, int mode, double* d_traj);
/* Synchronous form: runs the frames, waits, and copies the trajectory to the HOST buffer traj_host[count*8] (may be
 * NULL).  Never returns SRUKF_ERR_CLAMP_PENDING: the state before the block is kept on the device, and when a frame
 * needs the reference's theta-clamp branch of the modified Cholesky (SLAM.cpp:2279-2285) the block is rewound, the
 * frames before it are replayed, that frame runs on the exact column-by-column path (as srukf_update does) and the
 * replay continues behind it. */
int  srukf_run_frames(srukf_ctx* ctx, int first, int count, int mode, double* traj_host);
/* Wait for the stream; reports SRUKF_ERR_CLAMP_PENDING if any async frame needed the reference's theta-clamp branch of
 * the modified Cholesky: the frames BEFORE the first such frame are valid, that frame and the later ones are not
 * (srukf_clamp_info names it; restore the state the block started from and use srukf_run_frames or the step-wise API). */
int  srukf_synchronize(srukf_ctx* ctx);
/* Optional: capture a block of `count` staged frames as ONE graph for the following srukf_run_frames_async calls with that count
 * (default: graphs of 8 frames + single frames; the device idles ~10 us between two graph launches, which shows in short blocks).
 * Nothing is executed.  1 <= count <= 512. */
int  srukf_prepare_frames(srukf_ctx* ctx, int count);
/* B filters (independent sequences over the same frame range: the Monte-Carlo use, MonoSLAMView.cpp:526-572 once per run) replayed
 * concurrently on one GPU: frames are issued round-robin in chunks so that the filters' launches interleave, then all are awaited;
 * filters still in SRUKF_GPU_EXCLUSIVE are switched to SRUKF_GPU_SHARED.  A filter with a flagged frame in its block is rerun
 * alone through srukf_run_frames.  ctxs[b] must have its own sequence staged (srukf_stage_sequence) and must live on the same
 * device.  traj_host: [B][count][8] or NULL; status: [B] per-filter return codes or NULL.  Returns the first error. */
int  srukf_run_frames_batch(srukf_ctx* const* ctxs, int B, int first, int count, int mode, double* traj_host, int* status);
/* First flagged staged frame / pivot row of the last SRUKF_ERR_CLAMP_PENDING (-1, -1: none).  Either may be NULL. */
int  srukf_clamp_info(srukf_ctx* ctx, int* frame, int* row);

/* ---- instrumentation ------------------------------------------------------------------------ */

/* When on, every kernel launch is bracketed by HIP events on the context's stream and the
 * durations are accumulated per kernel class (no graph replay in this mode). */
int  srukf_set_profiling(srukf_ctx* ctx, int on);
/* Number of kernel classes; then name / total ms / launch count / algorithmic flops and bytes
 * (summed over launches) of class i since the last srukf_profile_reset. */
int  srukf_profile_count(srukf_ctx* ctx);
int  srukf_profile_get(srukf_ctx* ctx, int i, const char** name, double* total_ms, long long* launches,
                       double* alg_flops, double* alg_bytes);
int  srukf_profile_reset(srukf_ctx* ctx);

/* Test hooks, not for hosts.  poke_state: one entry of S behind the back of everything that tracks S.  starve_workers: persistent
 * factorisation launches of this context start without their worker workgroups, as if another process held the GPU (exercises the
 * bounded waits and the fallback to per-panel launches). */
int  srukf_debug_poke_state(srukf_ctx* ctx, int row, int col, double value);
int  srukf_debug_starve_workers(srukf_ctx* ctx, int on);      /* on = 2: only the split form's pair is starved (the tier it falls back to runs undisturbed) */
/* Kept for ABI compatibility (rounds 2 - 5: let srukf_set_storage accept SRUKF_STORAGE_F32_MIXED below its then epsilon floor): no effect since round 6. */
int  srukf_debug_allow_mixed(srukf_ctx* ctx, int on);

/* Measurement / test switches, not for hosts (every one defaults to the product path; captured graphs are dropped).
 * Process-wide keys (ctx may be NULL; they apply to what is built or captured afterwards):
 *   "gmw_persist" (0: one launch per 64-row panel), "gmw_fused", "rank_fused", "rank_fold" (0: the owners of the persistent launch never form
 *   their tiles of S^T S - U U^T themselves), "rank_aware", "graphs" (0: contexts created afterwards launch eagerly, which rocprofv3 --pmc needs),
 *   "shared_tenants" (2..8: persistent launches that share the GPU after srukf_set_exclusive(SRUKF_GPU_SHARED); srukf_run_frames_batch picks its own),
 *   "batch_wide" (0: srukf_run_frames_batch never takes the batched launches), "batch_groups" (1..4: groups the batched filters are cut into),
 *   "batch_split" (0: one launch per panel in the batched replay instead of slabs + plain trailing updates),
 *   "mem_split" (0: sizes beyond two register tiles per worker — N >= 400 — keep the memory-tile instance of k_gmw_persist instead of the split form
 *   k_gmw_pivslab_persist + k_gmw_tiles_persist; 2: the split form also where a worker would own two register tiles),
 *   round 6: "exact_rl" (0: the left-looking exact path k_gmw_col, one row per launch; 1: right-looking, as many pivots per launch as fit — default; 1 + B: B pivots per
 *   launch), "fold_head" (block rows the k_syrk launch in front of the split fold forms; 0: the rule 2 (Tp - 17)), "fold_force" (the split fold also where the tile workgroups
 *   do not all fit beside the pivot / slab launch), "timing" (1: phases of map changes and of flagged frames on stderr), "ctx_keep" (contexts a handle keeps across map
 *   changes: default 24), "batch_xcd" (0: k_syrk_b walks the head tiles in the solo launch's order), "batch_k128" (0: one pass over G per panel in the batched replay's
 *   trailing updates instead of one per pair of panels).
 * Per-context keys: "use_graph" (0: eager launches), "fused_motion" (0: k_motion + k_project as two launches, 1: k_project_motion, 2: "table"
 *   mode), "pxy2" (0: k_pxy instead of k_pxy2), "nullskip", "head_fold" (0: k_syrk launch in front of the persistent launch), "tail_fuse"
 *   (0: k_project_table in front of every frame), "table_perm", "f32_fuse", "step_fast" (0: the step-wise API keeps to its own launch sequences instead of the
 *   staged replay's cut at the association step), "step_fuse_export" (0: the fast path's results leave through export launches of their own instead of from the
 *   launches that form them), "step_early" (what srukf_update submits of the NEXT frame behind its own tail: 0 nothing, 1 checkpoint copy + frame scalars, 2 (default)
 *   also the announced frame's first launch), "step_spin" (0: wait with hipStreamSynchronize instead of spinning on the pinned flag word), "view_auto" (0: the display
 *   view is never exported with an update's status), "split_record"; round 6: "gain_fold" (1: k_gain's work in the tile epilogue of k_pxy2_fold — three launches per frame,
 *   slower: off), "split_fold" (0: k_syrk over all kept rows in front of the split form's pair instead of forming jobs inside its tile launch), "mixed_rank" (0: round 2's
 *   full-rank form of SRUKF_STORAGE_F32_MIXED), "mixed_f64_robot" (0: every tile of the mixed downdate on the fp32 pipe), "mixed_bf16" (1: its fp32 products from three bf16
 *   pieces), "mixed_null_ppm", "null_canon" (0: the null rows of a factor that NEED_REORDER or a map operation produced are left as they are until a frame tail rewrites them).
 *   See srukf_ctx.h (srukf_ctx::DbgSwitches). */
int  srukf_debug_set(srukf_ctx* ctx, const char* key, int value);
/* Diagnostic read-out of device-resident counters ("gmw_aborts", "clamp_rows", "frame", "frozen", "gate_timeouts", "gmw_shared", "split_form", "split_off",
 * "step_fast" / "step_slow": frames the step-wise API ran on the fast / the other path, "view_hits": srukf_get_frame_view calls served from an exported view,
 * "meas_flag_ticks": 10-ns ticks from the start of the fast path's last k_pxy2 to the moment the host had its statistics) and of the launch plan the next staged frame takes ("plan_persist",
 * "plan_register_form", "plan_tiles_per_worker", "plan_fold", "plan_head_fold", "plan_red_perm", "plan_motion", "plan_fuse", "plan_T", "plan_Tp", "plan_tiles",
 * "plan_workers", "plan_kept"). */
int  srukf_debug_get(srukf_ctx* ctx, const char* key, long long* value);
/* Diagnostic copy of a device work buffer (tests compare the launch sequences stage by stage): key = "Z", "DZ", "sigR", "Cmat", "Xr1",
 * "Utp", "P1", "h", "Si"; `count` doubles from the start of the buffer. */
int  srukf_debug_copy(srukf_ctx* ctx, const char* key, double* out, long long count);
/* ... and the other way (same keys, plus the operands of one factorisation: "Wf", "Gbak", "G", "D", "gsW", "gsL", "pans", "sync" — byte buffers counted in doubles).
 * srukf_debug_split_replay: ONE launch of the split form's pair (0: k_gmw_pivslab_persist, 1: k_gmw_tiles_persist) `reps` times ALONE against the buffers and flags a
 * real frame left behind (recorded with the "split_record" switch, moved between processes with debug_copy / debug_upload): the pair itself cannot run under
 * rocprofv3's counter passes, which serialise dispatches (scripts/split_replay.py). */
int  srukf_debug_upload(srukf_ctx* ctx, const char* key, const double* in, long long count);
int  srukf_debug_split_replay(srukf_ctx* ctx, int which, int reps);
/* Diagnostic builds (-DSRUKF_GMW_DBG) only: arms / reads the time stamps of the persistent factorisation launch (4096 values). */
int  srukf_debug_gmw_stamps(srukf_ctx* ctx, unsigned long long* buf);

/* Problem sizes of a context: N, n, Na, L. */
int  srukf_dims(const srukf_ctx* ctx, int* N, int* n, int* Na, int* L);

/* Stand-alone numeric primitives on DEVICE buffers (used by the parity tests; same kernels the
 * frame path launches).
 *   gmw   : S_out(upper) = modifiedCholeskyDecomposition(G)      SLAM.cpp:2197-2327
 *           G is n*n row-major symmetric (upper triangle read). force_slow=1 takes the
 *           column-by-column path that evaluates theta_j exactly as the reference does.
 *           clamp_hit (host int*, may be NULL) reports whether the theta clamp was active.  */
int  srukf_gmw_host(int device, int n, const double* G, double* S_out, double* D_out, double epsilon,
                    int force_slow, int* clamp_hit);
/*   project: one camera projection, SLAM.cpp:1634-1674 + 3177-3347 (host in/out, runs the device fn) */
int  srukf_project_host(int device, const srukf_params* p, int count, const double* feat6, const double* pos3,
                        const double* psi, const double* err2, double* uv_out);

#ifdef __cplusplus
}
#endif
#endif /* SRUKF_H_ */
