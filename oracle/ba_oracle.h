/*
 * ba_oracle.h -- CPU ORACLE for the ChainBundle bundle-adjustment hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (mcptam_amd/csrc,
 * libmcptam_hip.so) never links, loads or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm
 *   /root/reference/src/ChainBundle.cc:67-901,1132-1487
 *   /root/reference/include/mcptam/MEstimator.h:84-236
 *   /root/reference/src/TaylorCamera.cc:202-287,353-383,472-486,617-669
 * plus the third-party semantics the reference leans on (g2o LM schedule, TooN SE3/SO3),
 * restated from their published algorithms.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this
 * path and cannot be built here (g2o, CHOLMOD, TooN, libCVD, Eigen, ROS absent, no
 * network).  The oracle is therefore pinned only by self-consistency checks (analytic
 * vs central-difference Jacobians as in ChainBundle.cc:688-740, Schur vs full-system
 * solve, zero-noise ground-truth recovery, hand-computable cases) -- see tests/.
 */
#ifndef MCPTAM_BA_ORACLE_H
#define MCPTAM_BA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CHAIN 8
#define ORC_MAX_INV   31

/* Post-RefreshParams state of a TaylorCamera (TaylorCamera.cc:84-198). */
typedef struct orc_camera {
  double params[9];      /* a0,a2,a3,a4,xc,yc,c,d,e                     :89-99   */
  double image_size[2];  /* mv2ImageSize                                          */
  double affine[4];      /* mm2Affine row-major                          :183-186 */
  double center[2];      /* mv2Center                                    :127-128 */
  double min_theta;      /* mdMinTheta                                   :152     */
  double max_rho;        /* mdMaxRho                                     :147     */
  double theta_mean;     /* mdThetaMean                                  :551-565 */
  double theta_std;      /* mdThetaStd                                   :569     */
  int    n_inv;          /* number of inverse-poly coefficients (degree+1)        */
  int    pad_;
  double inv_coeffs[ORC_MAX_INV]; /* mvxPolyInvCoeffs                    :157     */
} orc_camera;

typedef struct orc_ba orc_ba;

/* per-outer-iteration trace, for trajectory comparison with the HIP path */
typedef struct orc_iter_log {
  double chi2_start;   /* currentChi after sigma recompute               */
  double chi2_end;     /* last trial's tempChi                           */
  double lambda_end;   /* lambda after the iteration                     */
  double sigma_sq;     /* raw Huber sigma^2 used in this iteration       */
  double rms_update;   /* RMS of last x                                  */
  int    trials;       /* qmax                                           */
  int    accepted;     /* 1 if the last trial was accepted               */
} orc_iter_log;

orc_ba* orc_ba_create(const orc_camera* cams, int ncam, int use_robust, int use_tukey, int verbose);
void    orc_ba_destroy(orc_ba*);

/* statics of ChainBundle (ChainBundle.cc:1132-1136) */
void orc_ba_set_limits(orc_ba*, int max_trials, double pct_limit, double rms_limit, double min_sigma);
/* when nonzero the convergence actions never fire (timing runs) */
void orc_ba_disable_convergence(orc_ba*, int disable);
/* oracle-only switch (sensitivity report): 1 = a pose vertex that occurs at two positions of one edge receives both cross terms
 * (symmetric Gauss-Newton block) instead of g2o's one-sided block (ba_oracle.c build_system) */
void orc_ba_set_dup_symmetric(orc_ba*, int on);
/* [3P-memory] constants / rules of g2o's Levenberg schedule as oracle-only switches (scripts/oracle_sensitivity.py):
 * key 0 initial-lambda factor tau (1e-5), 1 constant in the rho denominator (1e-3), 2 rejection rule (0: lambda *= ni, ni *= 2;
 * 1: lambda *= 2), 3 acceptance rule (0: max(1/3, min(1 - (2 rho - 1)^3, 2/3)); 1: 1/3; 2: no 2/3 cap) */
void orc_ba_set_variant(orc_ba*, int key, double value);
/* test switch: the k-th LM trial of the handle behaves as if the linear solver had failed (g2o: x keeps its previous content, is
 * applied, the trial is rejected) */
void orc_ba_set_fail_trial(orc_ba*, int k);
/* CPU-baseline variants for bench.py (ba_baseline.inc): solver 0 = the oracle proper, 1 = A "reference-shaped" (sparse
 * L D L^T of the un-marginalised system, one thread), 2 = B "best CPU" (Schur, OpenMP over `threads`) */
void orc_ba_set_solver(orc_ba*, int solver, int threads);
int  orc_ba_threads(orc_ba*);
int  orc_built_with_openmp(void);

int  orc_ba_add_pose (orc_ba*, const double R[9], const double t[3], int fixed);
int  orc_ba_add_point(orc_ba*, const double x[3], const int* chain, int n, int fixed);
int  orc_ba_add_meas (orc_ba*, const int* chain, int n, int point_id, const double uv[2],
                      double sigma_sq, int cam_index);

/* ChainBundle::Compute (ChainBundle.cc:1305-1451) */
int  orc_ba_compute(orc_ba*, volatile unsigned char* abort_flag, int n_iter, double user_lambda);

int    orc_ba_converged(orc_ba*);
int    orc_ba_total_iterations(orc_ba*);
int    orc_ba_get_point(orc_ba*, int id, double x[3]);
int    orc_ba_get_pose (orc_ba*, int id, double R[9], double t[3]);
int    orc_ba_num_outliers(orc_ba*);
/* out = n*3 ints: point id, front pose id, cam index */
int    orc_ba_get_outliers(orc_ba*, int* out, int cap);
double orc_ba_sigma_squared(orc_ba*);
double orc_ba_mean_chi_squared(orc_ba*);
double orc_ba_max_cov(orc_ba*);
double orc_ba_lambda(orc_ba*);
int    orc_ba_num_iter_logs(orc_ba*);
int    orc_ba_get_iter_logs(orc_ba*, orc_iter_log* out, int cap);

/* ---- introspection hooks used only by the self-consistency tests ---- */
int  orc_ba_num_meas(orc_ba*);
/* initialise structure (what g2o initializeOptimization does); returns #unknowns */
int  orc_ba_prepare(orc_ba*);
/* evaluate e, chi2 at the current state (computeActiveErrors); chi2 signed as chi2() */
void orc_ba_eval(orc_ba*, double* chi2_out /*M or NULL*/, double* err_out /*2M or NULL*/);
/* analytic Jacobians of measurement m: J_obs[link] (2x6 row-major, ORC_MAX_CHAIN of them),
 * J_src[link], J_pt (2x3).  mask bits: obs link i -> bit i, src link i -> bit 4+i,
 * point -> bit 8 (set = free and nonzero) */
int  orc_ba_jacobian(orc_ba*, int m, double* J_obs, double* J_src, double* J_pt);
/* the reference's disabled central-difference check (ChainBundle.cc:688-740) */
int  orc_ba_numeric_jacobian(orc_ba*, int m, double delta, double* J_obs, double* J_src, double* J_pt);
/* build H,b at the current state (sigma recomputed) and solve (H+lambda I)x=b two ways:
 * points-first block elimination (what compute uses) and one dense Cholesky of the whole
 * un-marginalised system.  x arrays have orc_ba_prepare() entries.  returns 0 on success */
int  orc_ba_debug_solve(orc_ba*, double lambda, double* x_schur, double* x_dense);
/* reduced pose system at the current state: out = [S (np*np, symmetric) | rhs (np) | pose part of J^T r (np)]; returns np
 * (out may be NULL to query it), -1 if a point block is not positive definite */
int  orc_ba_debug_system(orc_ba*, double lambda, double* out);
/* robust chi2 sum at current state with freshly recomputed sigma */
double orc_ba_debug_robust_chi2(orc_ba*, double* sigma_sq_raw);

/* camera primitives */
double orc_atan(double x);      /* correctly rounded arctangent (binary128 atanq rounded once) */
void orc_set_atan_libm(int on); /* oracle-only: 1 = the platform's libm atan (what the reference calls), 0 = correctly rounded (default) */
int  orc_cam_project(const orc_camera*, const double xc[3], double uv[2], double D[4]);
void orc_cam_sphere_deriv(const double xc[3], double dtheta[3], double dphi[3]);
/* TooN [3P-memory] */
void orc_se3_exp(const double mu[6], double R[9], double t[3]);
void orc_so3_exp(const double w[3], double R[9]);
/* MEstimator.h */
double orc_huber_sigma_squared(double* v, int n);   /* sorts v in place */
double orc_tukey_sigma_squared(double* v, int n);   /* sorts v in place */
double orc_tukey_weight(double e2, double s2);

#ifdef __cplusplus
}
#endif
#endif
