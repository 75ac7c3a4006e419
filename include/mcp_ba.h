/*
 * mcp_ba.h -- C ABI of the MI355X (gfx950) ChainBundle back end.
 *
 * Drop-in boundary for the bundle-adjustment hot path of MCPTAM.  The reference has no
 * FFI; its seam is the public surface of class ChainBundle
 * (/root/reference/include/mcptam/ChainBundle.h:106-186), which is everything
 * BundleAdjuster{Multi,Single,Calib} call (BundleAdjusterMulti.cc:75,90,101,116,126,148,
 * 160,196,272-274,298,310,328-330,246).  Each entry point below names the member it
 * replaces; INTEGRATION.md shows the ChainBundle.cc shim a maintainer would compile
 * against this header.
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no exceptions cross the ABI; the library copies
 *    every input at add_* time (caller keeps ownership).
 *  - ids start at 1 and poses and points share one counter (ChainBundle.cc:1145,1201,1215).
 *  - one handle lives for one BundleAdjust call like the stack ChainBundle object
 *    (BundleAdjusterMulti.cc:75); mcp_ba_compute may be called on it repeatedly (two-step
 *    mode, :210-224); every call re-initialises the optimiser and re-seeds lambda
 *    (ChainBundle.cc:1307).
 *  - rotation matrices are row-major double[9]; an SE3 is (R,t) with x' = R x + t.
 *  - all arithmetic is fp64 on the device.  There is NO CPU fallback: if no gfx950 device
 *    (or no HIP runtime) is usable, mcp_ba_create returns NULL and
 *    mcp_last_error() says why.
 *  - threading: a handle is driven from one thread (the MapMaker thread in MCPTAM);
 *    abort_flag may be written asynchronously by another thread AND is written by the
 *    solver itself on convergence (ChainBundle.cc:1027-1028,1105-1113); it is polled once
 *    per LM trial.
 */
#ifndef MCP_BA_H
#define MCP_BA_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCP_MAX_CHAIN 8      /* longest pose chain accepted (the reference's adapters build chains of 1 and 2 poses) */
#define MCP_MAX_INV   31     /* MAX_INV_DEGREE+1, include/mcptam/TaylorCamera.h:74 */

/* The already-fitted TaylorCamera, i.e. the state after TaylorCamera::RefreshParams
 * (src/TaylorCamera.cc:84-198), so the device never needs the root finder / SVD fit. */
typedef struct mcp_camera {
  double params[9];      /* mv9CameraParams: a0,a2,a3,a4,xc,yc,c,d,e   :89-99   */
  double image_size[2];  /* mv2ImageSize                                         */
  double affine[4];      /* mm2Affine, row-major                        :183-186 */
  double center[2];      /* mv2Center                                   :127-128 */
  double min_theta;      /* mdMinTheta                                  :152     */
  double max_rho;        /* mdMaxRho                                    :147     */
  double theta_mean;     /* mdThetaMean                                 :551-565 */
  double theta_std;      /* mdThetaStd                                  :569     */
  int    n_inv;          /* number of inverse polynomial coefficients (mbUsingInversePoly); 0 = the reference's
                            Newton fallback (:161-176, 258-270): inv_coeffs[0..1] then hold the linear
                            inverse model mv2LinearInvCoeffs that seeds FindRootWithNewton (:293-315)    */
  int    pad_;
  double inv_coeffs[MCP_MAX_INV];   /* mvxPolyInvCoeffs, x^0 first       :157     */
} mcp_camera;

/* ChainBundle statics (src/ChainBundle.cc:1132-1136; ROS overrides
 * include/mcptam/LoadStaticParamsServer.h:63-66).  Pass NULL for the defaults. */
typedef struct mcp_ba_params {
  int    max_iterations;             /* snMaxIterations = 100 (default n_iter)  */
  int    max_trials_after_failure;   /* snMaxTrialsAfterFailure = 100           */
  double update_percent_limit;       /* sdUpdatePercentConvergenceLimit = 1e-10 */
  double update_rms_limit;           /* sdUpdateRMSConvergenceLimit = 1e-10     */
  double min_mestimator_sigma;       /* sdMinMEstimatorSigma = 0.5              */
  int    disable_convergence;        /* 0; nonzero: convergence actions never fire
                                        (timing runs)            */
  int    device;                     /* HIP device ordinal, -1 = current device */
  int    profile;                    /* nonzero: bracket every stage with HIP events
                                        on the solver stream (mcp_ba_get_timing)  */
} mcp_ba_params;

typedef struct mcp_ba mcp_ba;

/* per-outer-iteration trace (what the reference only prints with ROS_DEBUG) */
typedef struct mcp_ba_iter_log {
  double chi2_start, chi2_end, lambda_end, sigma_sq, rms_update;
  int trials, accepted;
} mcp_ba_iter_log;

/* per-stage device time of the last mcp_ba_compute, milliseconds (HIP events on the
 * solver's stream) -- the analogue of the reference's timing topics (MapMakerTiming.msg) */
typedef struct mcp_ba_timing {
  double total_ms;
  double structure_ms;    /* host: initializeOptimization analogue + upload */
  double eval_ms, select_ms, linearize_ms, schur_ms, cholesky_ms, solve_ms, update_ms;
  int    n_linearize, n_trials;
  int    n_solves;        /* reduced systems built + factored (a solve may carry a second, speculative lambda) */
  int    n_spec_hits;     /* trials served by such a speculative solve */
  /* several ranks: collectives enqueued by this compute() per lane (main stream / speculative stream) and their payload */
  int    n_collectives_main, n_collectives_spec;
  double collective_bytes_main, collective_bytes_spec;
  int    n_median_fast;   /* medians that needed one collective (digit histograms rode on the accepted trial's all-reduce) */
  int    n_persist_fallbacks;   /* solves redone with the per-step kernels after a hand-off of the one-launch factorisation timed out (0 in a healthy run) */
  /* profile != 0 only.  The Schur complement of ONE system as the kernel executes it: v_mfma_f64_16x16x4 instructions per launch
   * (zero-filled 16 x 16 tile pairs of every 16-point chunk, 12 instructions each; 2048 flop per instruction), and the flops of the
   * structurally non-zero 6x3 . 3x3 . 3x6 block products it stands for (sum over the free points of k (k + 1) / 2 x 324, k = poses
   * that see the point: SURVEY 8(d)) */
  double schur_mfma_per_system, schur_flops_structural;
  /* flops of ONE factorisation as its plan executes it: 32 x 32 tile operations of the block-sparse plan after symbolic fill (a product
   * 2 x 32^3, a triangular solve 32^3, a diagonal tile's factorisation + inverse 2/3 x 32^3); the dense (6P)^3 / 3 of SURVEY 8(d) is
   * the upper bound a banded trajectory stays well below (0 if no plan was built) */
  double chol_flops_plan;
  /* chains of block columns the one-launch factorisation walks beside each other (1: the poses in add order; 3: a trajectory's band cut into
   * two halves + the separator between them; 0 if no plan was built) */
  int    chol_chains;
} mcp_ba_timing;

const char* mcp_last_error(void);
/* number of usable gfx950 devices (0 if none / runtime missing) */
int mcp_device_count(void);

/* ChainBundle::ChainBundle(TaylorCameraMap&, bool, bool, bool)   ChainBundle.cc:1139-1181 */
mcp_ba* mcp_ba_create(const mcp_camera* cams, int ncam, int use_robust, int use_tukey,
                      int verbose, const mcp_ba_params* params);
/* ChainBundle::~ChainBundle                                       ChainBundle.cc:1183-1195
 * (the reference builds one ChainBundle per BundleAdjust call, src/BundleAdjusterMulti.cc:75-76: the device blocks and pinned
 *  blocks of a destroyed handle go to a process-wide cache and the next handle on that device takes them from there;
 *  MCP_DEV_CACHE_MB bounds the cache per device, 0 switches it off -- INTEGRATION.md section 7) */
void    mcp_ba_destroy(mcp_ba*);

/* int ChainBundle::AddPose(SE3<> se3PoseFromRef, bool bFixed)     ChainBundle.cc:1198-1208 */
int mcp_ba_add_pose(mcp_ba*, const double R[9], const double t[3], int fixed);
/* int ChainBundle::AddPoint(Vector<3>, std::vector<int> vCams, bool bFixed)   :1211-1236
 * returns the id, or -1 (bad chain: unknown pose id, n<1 or n>MCP_MAX_CHAIN) */
int mcp_ba_add_point(mcp_ba*, const double x[3], const int* chain, int n, int fixed);
/* void ChainBundle::AddMeas(vector<int> vCams, int nPointIdx, Vector<2> v2Pos,
 *                           double dNoiseSigmaSquared, std::string cameraName) :1239-1281
 * cam_index indexes the cams[] passed at create (the reference keys cameras by name; the
 * shim maps name -> index).  returns 0, or -1 on a bad argument. */
int mcp_ba_add_meas(mcp_ba*, const int* chain, int n, int point_id, const double uv[2],
                    double sigma_sq, int cam_index);
/* Batched forms of the two calls above (same semantics, one ABI crossing): `chains` holds
 * `count` rows of `stride` ints of which chain_len[i] are used; add_points writes the new
 * ids to ids_out[count] (may be NULL). */
int mcp_ba_add_points(mcp_ba*, int count, const double* x /*count*3*/, const int* chains,
                      int stride, const int* chain_len, const unsigned char* fixed, int* ids_out);
int mcp_ba_add_measurements(mcp_ba*, int count, const int* chains, int stride,
                            const int* chain_len, const int* point_ids, const double* uv /*count*2*/,
                            const double* sigma_sq, const int* cam_index);

/* int ChainBundle::Compute(bool* pAbortSignal, int nNumIter, double dUserLambda) :1305-1451
 * n_iter < 0 selects params.max_iterations; n_iter == 0 runs no iteration (g2o optimize(0)), which the reference
 * reports as -1 unless the abort flag is up.  Return value as the reference: number of outer iterations run (>0),
 * 0 = aborted before any step, -1 = no iteration could be run (:1355-1366).  MCP_BA_ERR_RUNTIME (-2) is NOT a
 * reference outcome: a HIP / RCCL call failed (mcp_last_error() says which), the state was not written back.
 *
 * A SIZE LIMIT THE REFERENCE DOES NOT HAVE: the reduced pose system is factored as one dense-in-tiles matrix whose solution vector
 * the back-substitution keeps in LDS, so a map may hold at most MCP_BA_MAX_FREE_POSES free poses (6 P <= 6144 unknowns; every
 * BASELINE configuration is below it: c4 has 499).  Beyond it mcp_ba_prepare returns -1 and mcp_ba_compute MCP_BA_ERR_RUNTIME, with
 * "too many free poses ..." in mcp_last_error() and nothing is solved (g2o + CHOLMOD would go on, src/ChainBundle.cc:1150-1158); the
 * caller's remedy is what MCPTAM does anyway for large maps -- BundleAdjustRecent's window with the rest of the poses fixed. */
#define MCP_BA_ERR_RUNTIME (-2)
#define MCP_BA_MAX_FREE_POSES 1024
int mcp_ba_compute(mcp_ba*, volatile unsigned char* abort_flag, int n_iter, double user_lambda);

int    mcp_ba_converged(mcp_ba*);                 /* Converged()          ChainBundle.h:146 */
int    mcp_ba_total_iterations(mcp_ba*);          /* TotalIterations()    ChainBundle.h:148 */
int    mcp_ba_get_point(mcp_ba*, int id, double x[3]);               /* GetPoint  :1453-1457 */
int    mcp_ba_get_pose(mcp_ba*, int id, double R[9], double t[3]);   /* GetPose   :1459-1463 */
/* bulk read-back: ids[count] -> x[count*3] / R[count*9], t[count*3] */
int    mcp_ba_get_points(mcp_ba*, int count, const int* ids, double* x);
int    mcp_ba_get_poses(mcp_ba*, int count, const int* ids, double* R, double* t);
/* GetOutlierMeasurements()  :1465-1468, filled at :1368-1399.  out = triples
 * (point id, id of the FIRST pose of the observer chain, cam index). */
int    mcp_ba_num_outliers(mcp_ba*);
int    mcp_ba_get_outliers(mcp_ba*, int* out, int cap);
double mcp_ba_sigma_squared(mcp_ba*);             /* GetSigmaSquared()    :1470-1474 */
double mcp_ba_mean_chi_squared(mcp_ba*);          /* GetMeanChiSquared()  :1476-1481 */
double mcp_ba_max_cov(mcp_ba*);                   /* GetMaxCov()          ChainBundle.h:173 */
double mcp_ba_lambda(mcp_ba*);                    /* GetLambda()          :1483-1487 */

int    mcp_ba_num_iter_logs(mcp_ba*);
int    mcp_ba_get_iter_logs(mcp_ba*, mcp_ba_iter_log* out, int cap);
int    mcp_ba_get_timing(mcp_ba*, mcp_ba_timing* out);

/* ---- multi-GPU: points/measurements are sharded across ranks, poses
 * are replicated, and the reduced pose system is summed once per linearisation and once
 * per trial.  The library is collective-agnostic: the host installs a hook that performs
 * an in-place SUM all-reduce of `count` doubles at device pointer `buf` on `stream`
 * (RCCL in production, see mcptam_amd/dist.py).  It returns 0 on success.
 * hook == NULL (default) means single rank. */
typedef int (*mcp_allreduce_fn)(void* user, void* device_buf, size_t count, void* hip_stream);
int mcp_ba_set_allreduce(mcp_ba*, mcp_allreduce_fn hook, void* user, int rank, int world_size);

/* Native transport: an RCCL communicator over the GPUs of the node.  The all-reduces are enqueued on the
 * solver's HIP stream (stream-ordered, no host round trip).  Bootstrap: rank 0 calls mcp_comm_unique_id and
 * ships the MCP_COMM_ID_BYTES bytes to the other ranks by any means (torch.distributed, MPI, a file); every
 * rank then calls mcp_comm_init on its own device.  One communicator serves any number of handles. */
#define MCP_COMM_ID_BYTES 128
typedef struct mcp_comm mcp_comm;
int       mcp_comm_unique_id(void* id_out);
mcp_comm* mcp_comm_init(const void* id, int rank, int world_size, int device);
void      mcp_comm_destroy(mcp_comm*);
int       mcp_ba_set_comm(mcp_ba*, mcp_comm*);
/* SUM all-reduce of `count` doubles at a device pointer on the communicator (test hook); _lane: 0 = the main stream's
 * RCCL communicator, 1 = the speculative stream's (split off the first at init) */
int       mcp_comm_allreduce(mcp_comm*, void* device_buf, size_t count);
int       mcp_comm_allreduce_lane(mcp_comm*, void* device_buf, size_t count, int lane);

/* ---- introspection used by the parity tests (no reference counterpart) ---- */
/* structure build + upload without solving; returns the number of unknowns 6P+3N */
int mcp_ba_prepare(mcp_ba*);
/* computeActiveErrors at the current state: chi2 (signed as EdgeChainMeas::chi2, :401-417)
 * per measurement in ADD ORDER; err = 2 doubles per measurement (may be NULL) */
int mcp_ba_eval(mcp_ba*, double* chi2_out, double* err_out);
/* sigma^2 (raw Huber, MEstimator.h:194-204 over |chi2|) and robust chi2 sum at the current state */
int mcp_ba_robust_chi2(mcp_ba*, double* sigma_sq_raw, double* chi2_sum);
/* build the normal equations at the current state and solve (H + lambda I) x = b on the
 * device; x has mcp_ba_prepare() entries: free poses (id order) then free points (id order) */
int mcp_ba_debug_solve(mcp_ba*, double lambda, double* x_out);

/* dense SPD solve A x = b (row-major n x n, lower triangle read) with the reduced-system
 * Cholesky kernels; returns -1 if A is not positive definite.  n <= 6144. */
int mcp_dense_spd_solve(const double* A, int n, const double* b, double* x);
/* reproducibility check of the factorisation kernels: solves (A + q I) x_q = b for q = 0..nsys-1 (nsys <= 4) in one
 * batched launch chain, `reps` times from the same device-resident input; x (nsys*n) receives the first repetition's
 * solutions and *n_mismatch the number of later repetitions whose solutions differ from it in any bit. */
int mcp_dense_spd_stress(const double* A, int n, const double* b, int nsys, int reps, double* x, int* n_mismatch);
/* Structure cache (round 5).  MCPTAM builds a fresh ChainBundle per BundleAdjust call (src/BundleAdjusterMulti.cc:75) and calls
 * again while the map has not converged (src/MapMaker.cc run loop): a handle whose TOPOLOGY -- chains, which point hangs off which
 * chain, who measures what, what is fixed -- equals that of an earlier Prepare() on this device adopts that Prepare()'s structure
 * (host results + a device-to-device clone of the packed structure block) and uploads only its own numbers.  Results are bit for
 * bit those of a cold Prepare().  MCP_BA_STRUCT_CACHE=0 switches it off, MCP_BA_STRUCT_CACHE_MB sets the budget (default 512).
 * NEAR MISS (round 6): MCPTAM erases the measurements an adjustment flagged as outliers and adjusts again
 * (/root/reference/src/MapMaker.cc:225-230, 283-287 -> MapMakerServerBase::HandleOutliers, src/MapMakerServerBase.cc:1198-1238), so the
 * next ChainBundle brings the poses, points and chains of the call before and its measurements MINUS a few, in the same order.  Such a
 * handle adopts the cached structure of the superset (if no pose or point lost its last measurement); the erased measurements keep
 * their place in the device arrays with weight 0.  The mathematics is that of a cold Prepare() of the smaller map, the order of some
 * floating-point sums is the superset's: results agree with a cold Prepare() to rounding, not bit for bit (MCP_BA_NEAR_MISS=0: off).
 * A map that GAINED poses, points or measurements (a new keyframe) is built cold.
 * The entries below are diagnostics: hits / misses so far in this process (a near miss counts as a miss AND in near_hits); drop every entry. */
void mcp_ba_struct_cache_stats(long long* hits, long long* misses);
long long mcp_ba_struct_cache_near_hits(void);
void mcp_ba_struct_cache_clear(void);
/* the one-launch factorisation (ba_chol2.h) seen from outside: L (n*n row-major, lower; its diagonal 32x32 BLOCKS hold
 * L_kk^-1, which is what the kernels keep), y = L^-1 b (n), info[0] = hand-off error word, info[1] = failure flag */
int mcp_chol_debug_factor(const double* A, int n, const double* b, double* L_out, double* y_out, int* info);
/* device time (HIP events, ms per solve, first repetition left out) of the factorisation and the back-substitution launches
 * for (A + q I) x_q = b, q < nsys; band > 0: banded + bordered tile plan (i - j <= band, last `band` block rows dense) */
int mcp_chol_time(const double* A, int n, const double* b, int nsys, int reps, int band, double* ms_factor, double* ms_back, double* x);
/* the cut of the pose coupling graph Prepare() gives the factorisation its chains with (mcptam_amd/csrc/ba_cut.h; host code, no device
 * needed): adjacency[u*nf + v] != 0 = poses u and v (free poses in add order) couple.  order_out[nf]: position -> pose; segs_out[8]:
 * first tile (32 unknowns = 16/3 poses) of every chain, the separator's last, -1 behind them (segs_out[1] < 0: one chain);
 * info_out[14]: found, taken, relabelled, arcs, ring opened at, block columns on the longest path, of one chain, separator poses,
 * arc lengths [6].  `threads` only splits the search (the result does not depend on it).  Returns the number of chains (1 = none). */
int mcp_debug_pose_cut(const unsigned char* adjacency, int nf, int max_arcs, int threads, int* order_out, int* segs_out, int* info_out);
/* the reduced pose system the solver would factor at the current state for `lambda`: S (np*np, row-major, lower
 * triangle meaningful inside the tiles of the factorisation plan, other entries 0), rhs (np) and J^T r (np) behind it;
 * np = 6 * free poses.  Returns np (buffers may be NULL to query it), < 0 on error. */
int mcp_ba_debug_system(mcp_ba*, double lambda, double* S_rhs_b_out);

#ifdef __cplusplus
}
#endif
#endif
