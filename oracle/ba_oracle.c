/*
 * ba_oracle.c -- CPU ORACLE (test infrastructure only; see ba_oracle.h header note).
 * PARITY UNPINNED (no reference tests/golden vectors exist; reference unbuildable here).
 *
 * Plain-C restatement of /root/reference/src/ChainBundle.cc and the pieces of
 * TaylorCamera.cc / MEstimator.h it calls.  Every function cites the lines it follows.
 * Third-party semantics (g2o, TooN) are restated from their published algorithms and
 * marked [3P-memory] as in SURVEY.md Appendix A.
 *
 * Linear algebra: g2o solves the un-marginalised (6P+3N) system with CHOLMOD
 * (ChainBundle.cc:1150-1158,1218).  Here the same system is factored by block Cholesky
 * with the points ordered first (what a fill-reducing ordering does to a BA matrix);
 * orc_ba_debug_solve() cross-checks that against one dense Cholesky of the whole matrix.
 */
#include "ba_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <float.h>

typedef struct { double R[9], t[3]; } se3_t;

typedef struct { int id, fixed, unk, active; se3_t T, Tbak; } opose;
typedef struct { int id, fixed, unk, active, chain; double x[3], xbak[3];
                 int ms, mn;      /* measurement CSR (sorted list) */
                 int is, in;      /* incidence range */ } opoint;
typedef struct { int len; int v[ORC_MAX_CHAIN]; se3_t first[ORC_MAX_CHAIN];
                 double second[ORC_MAX_CHAIN][9]; } ochain;
typedef struct { int chain, point, cam; double z[2], omega;
                 double e[2], D[4], xc[3]; } omeas;

struct orc_ba {
  orc_camera* cams; int ncam;
  int robust, tukey, verbose;
  int max_trials; double pct_limit, rms_limit, min_sigma; int no_converge;

  opose* poses; int npose, cpose;
  opoint* points; int npoint, cpoint;
  omeas* meas; int nmeas, cmeas;
  ochain* chains; int nchain, cchain;
  int* id_kind; int* id_index; int cid; int next_id;   /* kind: 0 none, 1 pose, 2 point */

  /* structure (orc_ba_prepare) */
  int prepared;
  int nfp, nfl;            /* free active poses / points */
  int np, nx;              /* 6*nfp, 6*nfp+3*nfl */
  int* pt_meas;            /* measurement indices grouped by point */
  int* inc_pose;           /* incidence -> pose unk */
  int ninc;
  int* fl_point;           /* free point unk -> point index */
  int* fp_pose;            /* free pose unk -> pose index */

  /* linear system */
  double* Hpp; double* bp; /* np x np, np */
  double* V; double* g;    /* nfl x 9, nfl x 3 */
  double* W;               /* ninc x 18 (6x3 row-major) */
  double* S; double* x; double* ball;   /* work */

  /* robust data (RobustKernelData) */
  int need_recompute; double sigma_sq, sigma_sq_lim, sigma_lim;

  /* results */
  int converged, total_iterations; double lambda, max_cov;
  int* outliers; int noutliers, coutliers;
  orc_iter_log* logs; int nlogs, clogs;
  double last_chi2_action;

  /* CPU-baseline variants (ba_baseline.inc): 0 = the oracle proper (Schur, dense Cholesky, 1 thread), 1 = A (sparse
   * L D L^T of the un-marginalised system, 1 thread), 2 = B (Schur, OpenMP) */
  int solver, threads; void* sparse; void* par;
  /* [3P-memory] pieces of g2o's OptimizationAlgorithmLevenberg as switches (sensitivity report, scripts/oracle_sensitivity.py):
   * var_tau: initial lambda = tau * max diag (1e-5); var_rho_eps: the constant added to the rho denominator (1e-3);
   * var_reject: 0 = lambda *= ni, ni *= 2; 1 = lambda *= 2 every time; var_accept: 0 = max(1/3, min(1 - (2 rho - 1)^3, 2/3)),
   * 1 = 1/3 always, 2 = no 2/3 cap */
  double var_tau, var_rho_eps; int var_reject, var_accept;
  int fail_trial, trial_no;   /* test switch: the fail_trial-th trial is treated as a failed factorisation (CHOLMOD not positive definite) */
  int dup_symmetric;          /* 1: a vertex that occurs twice in an edge gets both cross terms (see build_system) */
};

static void baseline_free(orc_ba* h);
static int solve_system_sparse(orc_ba* h, double lambda, double* x);
static int solve_system_par(orc_ba* h, double lambda, double* x);
static void build_system_par(orc_ba* h);
static void compute_active_errors_par(orc_ba* h);
static double active_robust_chi2_par(orc_ba* h);
static void apply_update_par(orc_ba* h, const double* x);
static double select_kth(double* v, int n, int k);

/* ------------------------------------------------------------------ small math */
static void m3mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    T[3*i+j] = A[3*i]*B[j] + A[3*i+1]*B[3+j] + A[3*i+2]*B[6+j];
  memcpy(C, T, sizeof T);
}
static void m3tmul(const double* A, const double* B, double* C) {   /* A^T B */
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    T[3*i+j] = A[i]*B[j] + A[3+i]*B[3+j] + A[6+i]*B[6+j];
  memcpy(C, T, sizeof T);
}
static void m3v(const double* A, const double* v, double* o) {
  double a = A[0]*v[0] + A[1]*v[1] + A[2]*v[2];
  double b = A[3]*v[0] + A[4]*v[1] + A[5]*v[2];
  double c = A[6]*v[0] + A[7]*v[1] + A[8]*v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void m3tv(const double* A, const double* v, double* o) {
  double a = A[0]*v[0] + A[3]*v[1] + A[6]*v[2];
  double b = A[1]*v[0] + A[4]*v[1] + A[7]*v[2];
  double c = A[2]*v[0] + A[5]*v[1] + A[8]*v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void se3_identity(se3_t* T) { memset(T, 0, sizeof *T); T->R[0] = T->R[4] = T->R[8] = 1.0; }
/* TooN SE3 product: (R1 R2, R1 t2 + t1) [3P-memory] */
static void se3_mul(const se3_t* A, const se3_t* B, se3_t* C) {
  se3_t T; m3mul(A->R, B->R, T.R); m3v(A->R, B->t, T.t);
  T.t[0] += A->t[0]; T.t[1] += A->t[1]; T.t[2] += A->t[2]; *C = T;
}
static void se3_inv(const se3_t* A, se3_t* C) {
  se3_t T; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.R[3*i+j] = A->R[3*j+i];
  m3v(T.R, A->t, T.t); T.t[0] = -T.t[0]; T.t[1] = -T.t[1]; T.t[2] = -T.t[2]; *C = T;
}
static void se3_apply(const se3_t* A, const double* v, double* o) {
  double r[3]; m3v(A->R, v, r); o[0] = r[0] + A->t[0]; o[1] = r[1] + A->t[1]; o[2] = r[2] + A->t[2];
}

/* TooN rodrigues_so3_exp [3P-memory] */
static void rodrigues(const double* w, double A, double B, double* R) {
  { const double wx2 = w[0]*w[0], wy2 = w[1]*w[1], wz2 = w[2]*w[2];
    R[0] = 1.0 - B*(wy2 + wz2); R[4] = 1.0 - B*(wx2 + wz2); R[8] = 1.0 - B*(wx2 + wy2); }
  { const double a = A*w[2], b = B*(w[0]*w[1]); R[1] = b - a; R[3] = b + a; }
  { const double a = A*w[1], b = B*(w[0]*w[2]); R[2] = b + a; R[6] = b - a; }
  { const double a = A*w[0], b = B*(w[1]*w[2]); R[5] = b - a; R[7] = b + a; }
}
/* TooN SO3::exp [3P-memory] */
void orc_so3_exp(const double w[3], double R[9]) {
  const double one_6th = 1.0/6.0, one_20th = 1.0/20.0;
  const double theta_sq = w[0]*w[0] + w[1]*w[1] + w[2]*w[2];
  const double theta = sqrt(theta_sq);
  double A, B;
  if (theta_sq < 1e-8) { A = 1.0 - one_6th*theta_sq; B = 0.5; }
  else if (theta_sq < 1e-6) { B = 0.5 - 0.25*one_6th*theta_sq; A = 1.0 - theta_sq*one_6th*(1.0 - one_20th*theta_sq); }
  else { const double inv_theta = 1.0/theta; A = sin(theta)*inv_theta; B = (1 - cos(theta))*(inv_theta*inv_theta); }
  rodrigues(w, A, B, R);
}
/* TooN SE3::exp, mu = (t0,t1,t2,w0,w1,w2) [3P-memory]; used by VertexPoseSE3::oplusImpl
 * ChainBundle.cc:82-86 */
void orc_se3_exp(const double mu[6], double R[9], double t[3]) {
  const double one_6th = 1.0/6.0, one_20th = 1.0/20.0;
  const double* w = mu + 3;
  const double theta_sq = w[0]*w[0] + w[1]*w[1] + w[2]*w[2];
  const double theta = sqrt(theta_sq);
  double A, B;
  const double cr[3] = { w[1]*mu[2] - w[2]*mu[1], w[2]*mu[0] - w[0]*mu[2], w[0]*mu[1] - w[1]*mu[0] };
  if (theta_sq < 1e-8) {
    A = 1.0 - one_6th*theta_sq; B = 0.5;
    t[0] = mu[0] + 0.5*cr[0]; t[1] = mu[1] + 0.5*cr[1]; t[2] = mu[2] + 0.5*cr[2];
  } else {
    double C;
    if (theta_sq < 1e-6) {
      C = one_6th*(1.0 - one_20th*theta_sq); A = 1.0 - theta_sq*C; B = 0.5 - 0.25*one_6th*theta_sq;
    } else {
      const double inv_theta = 1.0/theta;
      A = sin(theta)*inv_theta; B = (1 - cos(theta))*(inv_theta*inv_theta); C = (1 - A)*(inv_theta*inv_theta);
    }
    const double wc[3] = { w[1]*cr[2] - w[2]*cr[1], w[2]*cr[0] - w[0]*cr[2], w[0]*cr[1] - w[1]*cr[0] };
    t[0] = mu[0] + B*cr[0] + C*wc[0]; t[1] = mu[1] + B*cr[1] + C*wc[1]; t[2] = mu[2] + B*cr[2] + C*wc[2];
  }
  rodrigues(w, A, B, R);
}
/* TooN SE3/SO3::generator_field: i<3 -> e_i * w ; i>=3 -> e_{i-3} x p [3P-memory] */
static void gen_field(int i, const double* p, double* o) {
  o[0] = o[1] = o[2] = 0.0;
  if (i < 3) { o[i] = 1.0; return; }
  const int a = i - 3;
  o[(a+1)%3] = -p[(a+2)%3];
  o[(a+2)%3] =  p[(a+1)%3];
}

/* ------------------------------------------------------------------ camera */
/* theta = atan(z / n), TaylorCamera.cc:243.  In the tracker the last ulp of this value decides template bytes (CVD::transform
 * truncates), and libm implementations differ in it (glibc 2.35's atan is within 1 ulp but not correctly
 * rounded: 0.07 % of arguments).  The oracle therefore takes the CORRECTLY ROUNDED arctangent -- the one platform-
 * independent definition -- by evaluating atanq in binary128 (libquadmath) and rounding once; the device reaches the same
 * double by a different route (double-double arithmetic, mcptam_amd/csrc/atan_cr.h).  tests/test_oracle_cpu.py pins it
 * against libm (never more than 1 ulp apart). */
#include <quadmath.h>
/* Oracle-only switch (round 5): the platform's libm atan instead -- what a real MCPTAM build calls (src/TaylorCamera.cc:216-217
 * uses ::atan).  Not a parity mode of the product (the device cannot reproduce a particular libm); scripts/oracle_sensitivity.py
 * uses it to bound what the deliberate "correctly rounded" deviation can move: template bytes, found positions, LM state. */
static int g_atan_libm = 0;
void orc_set_atan_libm(int on) { g_atan_libm = on; }
double orc_atan(double x) { return g_atan_libm ? atan(x) : (double)atanq((__float128)x); }

/* TaylorCamera::PolyVal, TaylorCamera.cc:472-486 */
static double polyval(const double* c, int n, double x) {
  double val = 0;
  for (int i = n - 1; i > 0; i--) { val += c[i]; val *= x; }
  val += c[0];
  return val;
}
/* TaylorCamera::Project (:202-287) + GetProjectionDerivs (:353-383).
 * returns 1 if the projection is flagged invalid (mbInvalid), 0 otherwise. */
int orc_cam_project(const orc_camera* cam, const double xc[3], double uv[2], double D[4]) {
  const double dNorm = sqrt(xc[0]*xc[0] + xc[1]*xc[1]);
  double dTheta, rho, cphi, sphi;
  if (dNorm == 0) dTheta = M_PI_2;                       /* :209-213 */
  else dTheta = orc_atan(xc[2]/dNorm);                        /* :216-217 */
  int invalid = (dTheta < cam->min_theta);                /* :223 */
  if (dNorm == 0) { rho = 0; cphi = 0; sphi = 0; }        /* :225-230 */
  else {
    if (cam->n_inv > 0) rho = polyval(cam->inv_coeffs, cam->n_inv, (dTheta - cam->theta_mean)/cam->theta_std); /* :261-262 */
    else {                                                /* :263-268 + FindRootWithNewton :293-315 (dErrorLimit 0.01, nMaxIter 50) */
      const double dTanTheta = xc[2]/dNorm;
      double c5[5] = { cam->params[0], 0 - dTanTheta, cam->params[1], cam->params[2], cam->params[3] };
      double d4[4] = { 0 - dTanTheta, 2*cam->params[1], 3*cam->params[2], 4*cam->params[3] };
      double prev = polyval(cam->inv_coeffs, 2, (dTheta - cam->theta_mean)/cam->theta_std);
      rho = prev;
      for (int i = 0; i < 50; i++) {
        rho = prev - polyval(c5, 5, prev)/polyval(d4, 4, prev);
        const double dError = fabs(rho - prev);             /* abs(double) [3P-memory: resolves to std::abs through TooN] */
        prev = rho;
        if (!(dError > 0.01)) break;
      }
    }
    cphi = xc[0]/dNorm; sphi = xc[1]/dNorm;               /* :271-272 */
  }
  const double dc0 = cphi*rho, dc1 = sphi*rho;            /* :279-280 */
  uv[0] = cam->affine[0]*dc0 + cam->affine[1]*dc1 + cam->center[0];   /* :282 */
  uv[1] = cam->affine[2]*dc0 + cam->affine[3]*dc1 + cam->center[1];
  if (!(uv[0] >= 0 && uv[0] < cam->image_size[0] && uv[1] >= 0 && uv[1] < cam->image_size[1]))
    invalid = 1;                                          /* :284, Utility.h:230-237 */
  if (D) {
    const double c5[5] = { cam->params[0], 0, cam->params[1], cam->params[2], cam->params[3] };  /* :102-106 */
    const double m5[5] = { -c5[0], 0, c5[2], 2*c5[3], 3*c5[4] };                                  /* :107-110 */
    const double w = polyval(c5, 5, rho);                                                        /* :355 */
    const double dRho_dTheta = (rho*rho + w*w) / polyval(m5, 5, rho);                             /* :358 */
    const double t0 = cphi*dRho_dTheta, t1 = sphi*dRho_dTheta;                                    /* :366-367 */
    const double p0 = -sphi*rho, p1 = cphi*rho;                                                   /* :370-371 */
    D[0] = cam->affine[0]*t0 + cam->affine[1]*t1;  D[2] = cam->affine[2]*t0 + cam->affine[3]*t1;  /* col 0 */
    D[1] = cam->affine[0]*p0 + cam->affine[1]*p1;  D[3] = cam->affine[2]*p0 + cam->affine[3]*p1;  /* col 1 */
  }
  return invalid;
}
/* TaylorCamera::GetCamSphereDeriv, TaylorCamera.cc:617-669 */
void orc_cam_sphere_deriv(const double v[3], double dT[3], double dP[3]) {
  const double x = v[0], y = v[1], z = v[2];
  const double x2 = x*x, y2 = y*y, z2 = z*z;
  const double n = sqrt(x*x + y*y), n2 = n*n, n3 = n2*n;
  if (n == 0) { dT[0] = dT[1] = dT[2] = 0; }
  else { dT[0] = -z*x/(n3 + n*z2); dT[1] = -z*y/(n3 + n*z2); dT[2] = n/(n2 + z2); }
  if (x == 0 && y == 0) { dP[0] = dP[1] = dP[2] = 0; }
  else { dP[0] = -y/(x2 + y2); dP[1] = x/(x2 + y2); dP[2] = 0; }
}

/* ------------------------------------------------------------------ M-estimators */
static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y);
}
/* Huber::FindSigmaSquared, MEstimator.h:194-204 */
double orc_huber_sigma_squared(double* v, int n) {
  qsort(v, n, sizeof(double), cmp_double);
  const double med = v[n/2];
  double dSigma = 1.4826 * (1 + 5.0/(double)((unsigned long long)n*2ull - 6ull)) * sqrt(med);
  dSigma = 1.345 * dSigma;
  return dSigma*dSigma;
}
/* Tukey::FindSigmaSquared, MEstimator.h:109-124 */
double orc_tukey_sigma_squared(double* v, int n) {
  qsort(v, n, sizeof(double), cmp_double);
  const double med = v[n/2];
  double dSigma = 1.4826 * (1 + 5.0/(double)((unsigned long long)n*2ull - 6ull)) * sqrt(med);
  dSigma = 4.6851 * dSigma;
  return dSigma*dSigma;
}
/* Tukey::Weight / SquareRootWeight, MEstimator.h:84-96 */
double orc_tukey_weight(double e2, double s2) {
  double s = (e2 > s2) ? 0.0 : 1.0 - (e2/s2);
  return s*s;
}

/* ------------------------------------------------------------------ container */
#define GROW(ptr, cap, need, type) do { if ((need) > (cap)) { int nc_ = (cap) ? (cap)*2 : 64; \
  while (nc_ < (need)) { nc_ *= 2; } \
  ptr = (type*)realloc(ptr, (size_t)nc_*sizeof(type)); cap = nc_; } } while (0)

orc_ba* orc_ba_create(const orc_camera* cams, int ncam, int use_robust, int use_tukey, int verbose) {
  orc_ba* h = (orc_ba*)calloc(1, sizeof *h);
  h->cams = (orc_camera*)malloc(sizeof(orc_camera)*ncam);
  memcpy(h->cams, cams, sizeof(orc_camera)*ncam);   /* ChainBundle copies the cameras, :1140 */
  h->ncam = ncam; h->robust = use_robust; h->tukey = use_tukey; h->verbose = verbose;
  h->max_trials = 100; h->pct_limit = 1e-10; h->rms_limit = 1e-10; h->min_sigma = 0.5;   /* :1132-1136 */
  h->next_id = 1;                                   /* mnCurrId = 1, :1145 */
  h->max_cov = DBL_MAX;                             /* :1179 */
  h->last_chi2_action = DBL_MAX;                    /* _dLastChi2, :1068 */
  h->var_tau = 1e-5; h->var_rho_eps = 1e-3;          /* g2o defaults [3P-memory] */
  return h;
}
static void free_structure(orc_ba* h) {
  baseline_free(h);
  free(h->pt_meas); free(h->inc_pose); free(h->fl_point); free(h->fp_pose);
  free(h->Hpp); free(h->bp); free(h->V); free(h->g); free(h->W); free(h->S); free(h->x); free(h->ball);
  h->pt_meas = h->inc_pose = h->fl_point = h->fp_pose = NULL;
  h->Hpp = h->bp = h->V = h->g = h->W = h->S = h->x = h->ball = NULL;
  h->prepared = 0;
}
void orc_ba_destroy(orc_ba* h) {
  if (!h) return;
  free_structure(h);
  free(h->cams); free(h->poses); free(h->points); free(h->meas); free(h->chains);
  free(h->id_kind); free(h->id_index); free(h->outliers); free(h->logs); free(h);
}
void orc_ba_set_limits(orc_ba* h, int max_trials, double pct, double rms, double min_sigma) {
  h->max_trials = max_trials; h->pct_limit = pct; h->rms_limit = rms; h->min_sigma = min_sigma;
}
void orc_ba_disable_convergence(orc_ba* h, int d) { h->no_converge = d; }
void orc_ba_set_dup_symmetric(orc_ba* h, int on) { h->dup_symmetric = on; }
void orc_ba_set_variant(orc_ba* h, int key, double value) {
  switch (key) { case 0: h->var_tau = value; break; case 1: h->var_rho_eps = value; break; case 2: h->var_reject = (int)value; break; case 3: h->var_accept = (int)value; break; default: break; }
}
void orc_ba_set_fail_trial(orc_ba* h, int k) { h->fail_trial = k; h->trial_no = 0; }

static int new_id(orc_ba* h, int kind, int index) {
  int id = h->next_id++;
  if (id >= h->cid) {
    int nc = h->cid ? h->cid*2 : 256; while (nc <= id) nc *= 2;
    h->id_kind = (int*)realloc(h->id_kind, sizeof(int)*nc);
    h->id_index = (int*)realloc(h->id_index, sizeof(int)*nc);
    memset(h->id_kind + h->cid, 0, sizeof(int)*(nc - h->cid));
    h->cid = nc;
  }
  h->id_kind[id] = kind; h->id_index[id] = index;
  return id;
}
/* ChainBundle::AddPose, ChainBundle.cc:1198-1208 */
int orc_ba_add_pose(orc_ba* h, const double R[9], const double t[3], int fixed) {
  GROW(h->poses, h->cpose, h->npose + 1, opose);
  opose* p = &h->poses[h->npose];
  memset(p, 0, sizeof *p);
  memcpy(p->T.R, R, 72); memcpy(p->T.t, t, 24); p->fixed = fixed ? 1 : 0; p->unk = -1;
  p->id = new_id(h, 1, h->npose);
  h->npose++; h->prepared = 0;
  return p->id;
}
/* mmHelpers lookup, ChainBundle.cc:1220-1230 / 1247-1273: one helper per distinct chain */
static int find_chain(orc_ba* h, const int* ids, int n) {
  if (n < 1 || n > ORC_MAX_CHAIN) return -1;
  int v[ORC_MAX_CHAIN];
  for (int i = 0; i < n; i++) {
    if (ids[i] <= 0 || ids[i] >= h->next_id || h->id_kind[ids[i]] != 1) return -1;
    v[i] = h->id_index[ids[i]];
  }
  for (int c = h->nchain - 1; c >= 0; c--) {
    if (h->chains[c].len != n) continue;
    int same = 1; for (int i = 0; i < n; i++) if (h->chains[c].v[i] != v[i]) { same = 0; break; }
    if (same) return c;
  }
  GROW(h->chains, h->cchain, h->nchain + 1, ochain);
  ochain* c = &h->chains[h->nchain]; memset(c, 0, sizeof *c);
  c->len = n; for (int i = 0; i < n; i++) c->v[i] = v[i];
  return h->nchain++;
}
/* ChainBundle::AddPoint, ChainBundle.cc:1211-1236 */
int orc_ba_add_point(orc_ba* h, const double x[3], const int* chain, int n, int fixed) {
  int c = find_chain(h, chain, n); if (c < 0) return -1;
  GROW(h->points, h->cpoint, h->npoint + 1, opoint);
  opoint* p = &h->points[h->npoint]; memset(p, 0, sizeof *p);
  memcpy(p->x, x, 24); p->chain = c; p->fixed = fixed ? 1 : 0; p->unk = -1;
  p->id = new_id(h, 2, h->npoint);
  h->npoint++; h->prepared = 0;
  return p->id;
}
/* ChainBundle::AddMeas, ChainBundle.cc:1239-1281 */
int orc_ba_add_meas(orc_ba* h, const int* chain, int n, int point_id, const double uv[2],
                    double sigma_sq, int cam_index) {
  if (point_id <= 0 || point_id >= h->next_id || h->id_kind[point_id] != 2) return -1;
  if (cam_index < 0 || cam_index >= h->ncam) return -1;
  int c = find_chain(h, chain, n); if (c < 0) return -1;
  GROW(h->meas, h->cmeas, h->nmeas + 1, omeas);
  omeas* m = &h->meas[h->nmeas]; memset(m, 0, sizeof *m);
  m->chain = c; m->point = h->id_index[point_id]; m->cam = cam_index;
  m->z[0] = uv[0]; m->z[1] = uv[1];
  m->omega = 1/sqrt(sigma_sq);          /* information = I * 1/sqrt(sigma^2), :1244-1245 */
  h->nmeas++; h->prepared = 0;
  return 0;
}
int orc_ba_num_meas(orc_ba* h) { return h->nmeas; }

/* PoseChainHelper::UpdateTransforms, ChainBundle.cc:120-150 */
static void update_chains(orc_ba* h) {
  for (int c = 0; c < h->nchain; c++) {
    ochain* ch = &h->chains[c];
    se3_t BfW; se3_identity(&BfW);
    for (int i = 0; i < ch->len; i++) {
      se3_mul(&h->poses[ch->v[i]].T, &BfW, &BfW);        /* :129 */
      ch->first[i] = BfW;
    }
    se3_t CfB; se3_identity(&CfB);
    for (int i = ch->len - 1; i >= 0; i--) {
      memcpy(ch->second[i], CfB.R, 72);                   /* :146-147 (rotation only) */
      se3_mul(&CfB, &h->poses[ch->v[i]].T, &CfB);         /* :148 */
    }
  }
}
/* PoseChainHelper::MoveTogether, ChainBundle.cc:157-199 */
static int move_together(const orc_ba* h, const ochain* a, const ochain* b, int depth) {
  int furthest = -1;
  for (;;) {
    int t = furthest + 1;
    if (a->len <= t || b->len <= t) break;
    if (a->v[t] != b->v[t]) break;
    furthest = t;
    if (furthest == depth) return 1;
  }
  if (furthest == -1) return 0;
  for (int i = furthest; i <= depth; i++) if (!h->poses[a->v[i]].fixed) return 0;
  return 1;
}

/* EdgeChainMeas::computeError, ChainBundle.cc:376-397 with
 * VertexRelPoint::estimateInGlobalCartesian :312-324 */
static void compute_error(orc_ba* h, omeas* m) {
  const opoint* p = &h->points[m->point];
  const ochain* sc = &h->chains[p->chain];
  const ochain* oc = &h->chains[m->chain];
  se3_t inv; se3_inv(&sc->first[sc->len-1], &inv);
  double xw[3]; se3_apply(&inv, p->x, xw);                       /* :323 */
  se3_apply(&oc->first[oc->len-1], xw, m->xc);                    /* :389 */
  double uv[2];
  orc_cam_project(&h->cams[m->cam], m->xc, uv, m->D);            /* :390-392 */
  m->e[0] = m->z[0] - uv[0]; m->e[1] = m->z[1] - uv[1];          /* :394-396 */
}
/* EdgeChainMeas::chi2, ChainBundle.cc:401-417 */
static double meas_chi2(const orc_ba* h, const omeas* m) {
  double val = m->omega*(m->e[0]*m->e[0] + m->e[1]*m->e[1]);
  if (h->points[m->point].fixed && h->robust) val *= -1;
  return val;
}
/* g2o computeActiveErrors + UpdateHelpersAction (:925-947) */
static void compute_active_errors(orc_ba* h) {
  if (h->solver == 2) { compute_active_errors_par(h); return; }
  update_chains(h);
  for (int i = 0; i < h->nmeas; i++) compute_error(h, &h->meas[i]);
}
/* RobustKernelData::RecomputeNow, ChainBundle.cc:810-833 */
static void recompute_sigma(orc_ba* h) {
  h->need_recompute = 0;
  double* v = (double*)malloc(sizeof(double)*(h->nmeas > 0 ? h->nmeas : 1));
  for (int i = 0; i < h->nmeas; i++) v[i] = fabs(meas_chi2(h, &h->meas[i]));
  double s;
  if (h->solver == 2 && h->nmeas > 0) {     /* baseline B: the same element [n/2] by quick-select instead of a full sort */
    const int n = h->nmeas;
    double dSigma = 1.4826 * (1 + 5.0/(double)((unsigned long long)n*2ull - 6ull)) * sqrt(select_kth(v, n, n/2));
    dSigma = 1.345 * dSigma; s = dSigma*dSigma;
  } else s = orc_huber_sigma_squared(v, h->nmeas);
  free(v);
  h->sigma_sq = s; h->sigma_sq_lim = s;
  const double mins = h->min_sigma*h->min_sigma;                  /* :1148 */
  if (h->sigma_sq_lim < mins) h->sigma_sq_lim = mins;
  h->sigma_lim = sqrt(h->sigma_sq_lim);
}
/* RobustKernelAdaptive::robustify, ChainBundle.cc:871-897 */
static void robustify(orc_ba* h, double e2, double rho[3]) {
  if (h->need_recompute) recompute_sigma(h);
  const double s2 = h->sigma_sq_lim, s = h->sigma_lim;
  if (e2 <= s2) { rho[0] = fabs(e2); rho[1] = 1.; rho[2] = 0.; }
  else { double e = sqrt(e2); rho[0] = 2*s*e - s2; rho[1] = s/e; rho[2] = -0.5*rho[1]/e2; }
}
/* g2o SparseOptimizer::activeRobustChi2 [3P-memory] */
static double active_robust_chi2(orc_ba* h) {
  if (h->solver == 2) return active_robust_chi2_par(h);
  double chi = 0.0, rho[3];
  for (int i = 0; i < h->nmeas; i++) {
    const double c = meas_chi2(h, &h->meas[i]);
    if (h->robust) { robustify(h, c, rho); chi += rho[0]; } else chi += c;
  }
  return chi;
}

/* point frame used by oplusImpl (:253-265) and linearizeOplus (:595-606) */
static void point_frame(const double* x, double* Rp, double* dir, double* rho) {
  const double len = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
  *rho = 1.0/len;
  dir[0] = x[0]*(*rho); dir[1] = x[1]*(*rho); dir[2] = x[2]*(*rho);
  double ax[3] = { dir[1], -dir[0], 0.0 };                /* dir ^ (0,0,1) */
  const double nrm = sqrt(ax[0]*ax[0] + ax[1]*ax[1] + ax[2]*ax[2]);
  const double angle = asin(nrm);
  ax[0] /= nrm; ax[1] /= nrm; ax[2] /= nrm;               /* TooN::normalize; no zero guard in the live code */
  ax[0] *= angle; ax[1] *= angle; ax[2] *= angle;
  orc_so3_exp(ax, Rp);
}
/* VertexRelPoint::oplusImpl, ChainBundle.cc:237-281 */
static void point_oplus(double* x, const double* u) {
  double Rp[9], dir[3], rho;
  point_frame(x, Rp, dir, &rho);
  const double w[3] = { u[0], u[1], 0.0 };
  double E[9]; orc_so3_exp(w, E);
  double M[9]; m3tmul(Rp, E, M); m3mul(M, Rp, M);          /* Rp^-1 * exp * Rp, left to right */
  double v[3]; m3v(M, dir, v);
  const double s = 1/(rho + u[2]);
  x[0] = s*v[0]; x[1] = s*v[1]; x[2] = s*v[2];
  const double d = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
  if (d > 1e5) { const double f = 1e5/d; x[0] *= f; x[1] *= f; x[2] *= f; }
  if (d < 1e-5) { const double f = 1e-5/d; x[0] *= f; x[1] *= f; x[2] *= f; }
}
/* VertexPoseSE3::oplusImpl, ChainBundle.cc:82-86 */
static void pose_oplus(se3_t* T, const double* u) {
  se3_t E; orc_se3_exp(u, E.R, E.t); se3_mul(&E, T, T);
}

/* EdgeChainMeas::linearizeOplus, ChainBundle.cc:449-749.  J blocks are 2x6 / 2x3 row-major.
 * returns mask of nonzero free blocks. */
static int linearize(const orc_ba* h, const omeas* m, double Jo[ORC_MAX_CHAIN][12],
                     double Js[ORC_MAX_CHAIN][12], double Jp[6]) {
  const opoint* p = &h->points[m->point];
  const ochain* sc = &h->chains[p->chain];
  const ochain* oc = &h->chains[m->chain];
  int mask = 0;
  se3_t inv; se3_inv(&sc->first[sc->len-1], &inv);
  double xw[3]; se3_apply(&inv, p->x, xw);                               /* :476 */
  double xc[3]; se3_apply(&oc->first[oc->len-1], xw, xc);                /* :477 */
  double dT[3], dP[3]; orc_cam_sphere_deriv(xc, dT, dP);                 /* :480 */
  const double* D = m->D;                                                /* _m2CamDerivs from computeError */
  for (int i = 0; i < oc->len; i++) {                                    /* :485-532 */
    if (h->poses[oc->v[i]].fixed) continue;
    if (move_together(h, oc, sc, i)) { memset(Jo[i], 0, 96); continue; }
    double base[3]; se3_apply(&oc->first[i], xw, base);                  /* :505 */
    for (int k = 0; k < 6; k++) {
      double mb[3], mc[3]; gen_field(k, base, mb);                       /* :512 */
      m3v(oc->second[i], mb, mc);                                        /* :515 */
      const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2];
      const double s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
      Jo[i][k]     = -1*(D[0]*s0 + D[1]*s1);                             /* :523-530 */
      Jo[i][6 + k] = -1*(D[2]*s0 + D[3]*s1);
    }
    mask |= 1 << i;
  }
  for (int i = 0; i < sc->len; i++) {                                    /* :535-586 */
    if (h->poses[sc->v[i]].fixed) continue;
    if (move_together(h, sc, oc, i)) { memset(Js[i], 0, 96); continue; }
    double base[3]; se3_apply(&sc->first[i], xw, base);                  /* :555 */
    se3_t bi, cfb; se3_inv(&sc->first[i], &bi); se3_mul(&oc->first[oc->len-1], &bi, &cfb);  /* :567 */
    for (int k = 0; k < 6; k++) {
      double mb[3], mc[3]; gen_field(k, base, mb);
      mb[0] = -mb[0]; mb[1] = -mb[1]; mb[2] = -mb[2];                    /* :564 */
      m3v(cfb.R, mb, mc);                                                /* :568-569 */
      const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2];
      const double s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
      Js[i][k]     = -1*(D[0]*s0 + D[1]*s1);
      Js[i][6 + k] = -1*(D[2]*s0 + D[3]*s1);
    }
    mask |= 1 << (ORC_MAX_CHAIN + i);
  }
  if (!p->fixed) {                                                       /* :589-684 */
    double Rp[9], dir[3], rho;
    point_frame(p->x, Rp, dir, &rho);
    double rx[3]; m3v(Rp, p->x, rx);
    double cols[3][3], gtmp[3];
    gen_field(3, rx, gtmp); m3tv(Rp, gtmp, cols[0]);                     /* :614 */
    gen_field(4, rx, gtmp); m3tv(Rp, gtmp, cols[1]);                     /* :617 */
    cols[2][0] = -1*p->x[0]/rho; cols[2][1] = -1*p->x[1]/rho; cols[2][2] = -1*p->x[2]/rho;   /* :620 */
    se3_t si, cfs; se3_inv(&sc->first[sc->len-1], &si); se3_mul(&oc->first[oc->len-1], &si, &cfs);  /* :659 */
    for (int k = 0; k < 3; k++) {
      double mc[3]; m3v(cfs.R, cols[k], mc);                             /* :662 */
      const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2];
      const double s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
      Jp[k]     = -1*(D[0]*s0 + D[1]*s1);                                /* :676-682 */
      Jp[3 + k] = -1*(D[2]*s0 + D[3]*s1);
    }
    mask |= 1 << (2*ORC_MAX_CHAIN);
  }
  return mask;
}

int orc_ba_jacobian(orc_ba* h, int m, double* J_obs, double* J_src, double* J_pt) {
  if (!h->prepared) orc_ba_prepare(h);
  double Jo[ORC_MAX_CHAIN][12], Js[ORC_MAX_CHAIN][12], Jp[6];
  memset(Jo, 0, sizeof Jo); memset(Js, 0, sizeof Js); memset(Jp, 0, sizeof Jp);
  update_chains(h); compute_error(h, &h->meas[m]);      /* linearizeOplus uses _m2CamDerivs cached by computeError */
  int mask = linearize(h, &h->meas[m], Jo, Js, Jp);
  memcpy(J_obs, Jo, sizeof Jo); memcpy(J_src, Js, sizeof Js); memcpy(J_pt, Jp, sizeof Jp);
  return mask;
}
/* the reference's disabled numeric check, ChainBundle.cc:688-740 (central differences via oplus) */
int orc_ba_numeric_jacobian(orc_ba* h, int mi, double delta, double* J_obs, double* J_src, double* J_pt) {
  if (!h->prepared) orc_ba_prepare(h);
  omeas* m = &h->meas[mi];
  opoint* p = &h->points[m->point];
  const ochain* sc = &h->chains[p->chain];
  const ochain* oc = &h->chains[m->chain];
  const double scalar = 1.0/(2*delta);
  memset(J_obs, 0, sizeof(double)*ORC_MAX_CHAIN*12);
  memset(J_src, 0, sizeof(double)*ORC_MAX_CHAIN*12);
  memset(J_pt, 0, sizeof(double)*6);
  int mask = 0;
  for (int side = 0; side < 2; side++) {
    const ochain* c = side ? sc : oc; double* J = side ? J_src : J_obs;
    for (int i = 0; i < c->len; i++) {
      opose* ps = &h->poses[c->v[i]];
      if (ps->fixed) continue;
      for (int d = 0; d < 6; d++) {
        double add[6] = {0,0,0,0,0,0}, ep[2];
        se3_t bak = ps->T;
        add[d] = delta; pose_oplus(&ps->T, add); update_chains(h); compute_error(h, m);
        ep[0] = m->e[0]; ep[1] = m->e[1]; ps->T = bak;
        add[d] = -delta; pose_oplus(&ps->T, add); update_chains(h); compute_error(h, m);
        ps->T = bak;
        J[i*12 + d]     = scalar*(ep[0] - m->e[0]);
        J[i*12 + 6 + d] = scalar*(ep[1] - m->e[1]);
      }
      mask |= 1 << (side*ORC_MAX_CHAIN + i);
    }
  }
  if (!p->fixed) {
    for (int d = 0; d < 3; d++) {
      double add[3] = {0,0,0}, ep[2], bak[3];
      memcpy(bak, p->x, 24);
      add[d] = delta; point_oplus(p->x, add); update_chains(h); compute_error(h, m);
      ep[0] = m->e[0]; ep[1] = m->e[1]; memcpy(p->x, bak, 24);
      add[d] = -delta; point_oplus(p->x, add); compute_error(h, m);
      memcpy(p->x, bak, 24);
      J_pt[d] = scalar*(ep[0] - m->e[0]); J_pt[3 + d] = scalar*(ep[1] - m->e[1]);
    }
    mask |= 1 << (2*ORC_MAX_CHAIN);
  }
  update_chains(h); compute_error(h, m);
  return mask;
}

/* ------------------------------------------------------------------ structure */
/* what g2o initializeOptimization/buildStructure do [3P-memory]: active vertices are those
 * touched by an edge; unknowns = non-fixed active vertices in id order (none marginalised,
 * ChainBundle.cc:1218). */
int orc_ba_prepare(orc_ba* h) {
  free_structure(h);
  for (int i = 0; i < h->npose; i++) { h->poses[i].active = 0; h->poses[i].unk = -1; }
  for (int i = 0; i < h->npoint; i++) { h->points[i].active = 0; h->points[i].unk = -1; h->points[i].mn = 0; }
  for (int i = 0; i < h->nmeas; i++) {
    const omeas* m = &h->meas[i]; opoint* p = &h->points[m->point];
    p->active = 1; p->mn++;
    const ochain* oc = &h->chains[m->chain]; const ochain* sc = &h->chains[p->chain];
    for (int k = 0; k < oc->len; k++) h->poses[oc->v[k]].active = 1;
    for (int k = 0; k < sc->len; k++) h->poses[sc->v[k]].active = 1;
  }
  h->nfp = h->nfl = 0;
  h->fp_pose = (int*)malloc(sizeof(int)*(h->npose + 1));
  h->fl_point = (int*)malloc(sizeof(int)*(h->npoint + 1));
  for (int i = 0; i < h->npose; i++) if (h->poses[i].active && !h->poses[i].fixed) { h->fp_pose[h->nfp] = i; h->poses[i].unk = h->nfp++; }
  for (int i = 0; i < h->npoint; i++) if (h->points[i].active && !h->points[i].fixed) { h->fl_point[h->nfl] = i; h->points[i].unk = h->nfl++; }
  h->np = 6*h->nfp; h->nx = h->np + 3*h->nfl;
  /* measurement CSR by point */
  h->pt_meas = (int*)malloc(sizeof(int)*(h->nmeas + 1));
  int acc = 0;
  for (int i = 0; i < h->npoint; i++) { h->points[i].ms = acc; acc += h->points[i].mn; h->points[i].mn = 0; }
  for (int i = 0; i < h->nmeas; i++) { opoint* p = &h->points[h->meas[i].point]; h->pt_meas[p->ms + p->mn++] = i; }
  /* incidences: distinct free poses touched by each free point */
  int cap = 0; h->ninc = 0; h->inc_pose = NULL;
  for (int i = 0; i < h->npoint; i++) {
    opoint* p = &h->points[i]; p->is = h->ninc; p->in = 0;
    if (p->unk < 0) continue;
    const ochain* sc = &h->chains[p->chain];
    for (int pass = -1; pass < p->mn; pass++) {
      const ochain* c = (pass < 0) ? sc : &h->chains[h->meas[h->pt_meas[p->ms + pass]].chain];
      for (int k = 0; k < c->len; k++) {
        int u = h->poses[c->v[k]].unk; if (u < 0) continue;
        int found = 0; for (int q = 0; q < p->in; q++) if (h->inc_pose[p->is + q] == u) { found = 1; break; }
        if (!found) { GROW(h->inc_pose, cap, h->ninc + 1, int); h->inc_pose[h->ninc++] = u; p->in++; }
      }
    }
  }
  h->Hpp = (double*)calloc((size_t)h->np*h->np + 1, 8); h->S = (double*)calloc((size_t)h->np*h->np + 1, 8);
  h->bp = (double*)calloc(h->np + 1, 8);
  h->V = (double*)calloc((size_t)h->nfl*9 + 1, 8); h->g = (double*)calloc((size_t)h->nfl*3 + 1, 8);
  h->W = (double*)calloc((size_t)h->ninc*18 + 1, 8);
  h->x = (double*)calloc(h->nx + 1, 8); h->ball = (double*)calloc(h->nx + 1, 8);
  h->prepared = 1;
  update_chains(h);
  return h->nx;
}

static int find_inc(const orc_ba* h, const opoint* p, int unk) {
  for (int q = 0; q < p->in; q++) if (h->inc_pose[p->is + q] == unk) return p->is + q;
  return -1;
}
/* g2o BlockSolver::buildSystem + BaseMultiEdge::constructQuadraticForm [3P-memory]:
 * H += Ji^T (rho' Omega) Jj, b += -Ji^T (rho' Omega) e over the free vertices of each edge, i < j in the edge's vertex order
 * (observer chain, then source chain, then the point: ChainBundle.cc:1258-1267).
 * A pose vertex may sit at TWO positions i < j of one edge (BundleAdjusterCalib: the relative camera pose is link 2 of the
 * observer chain and of the source chain, BundleAdjusterCalib.cc:166-199).  g2o maps the (i, j) block of such an edge onto the
 * vertex's own diagonal block (BlockSolver::buildStructure: _Hpp->block(ind1, ind2) with ind1 == ind2, not transposed) and adds
 * Ji^T Omega Jj to it ONCE -- the transposed term is never added, so the block is not symmetric -- and LinearSolverCholmod reads
 * the upper triangle of diagonal blocks (fillCCS(..., upperTriangle = true)).  The system solved is therefore the symmetric matrix
 * whose entries (r, c), r <= c, of that block are Ji^T Omega Ji + Jj^T Omega Jj + Ji^T Omega Jj.  Reproduced here
 * (dup_symmetric = 0, the default); dup_symmetric = 1 adds the transposed term as well (the mathematically complete Gauss-Newton
 * block; round 1-2 behaviour, kept as a switch for the sensitivity report). */
static void build_system(orc_ba* h) {
  if (h->solver == 2) { build_system_par(h); return; }
  const int np = h->np;
  memset(h->Hpp, 0, sizeof(double)*(size_t)np*np); memset(h->bp, 0, sizeof(double)*np);
  memset(h->V, 0, sizeof(double)*(size_t)h->nfl*9); memset(h->g, 0, sizeof(double)*(size_t)h->nfl*3);
  memset(h->W, 0, sizeof(double)*(size_t)h->ninc*18);
  for (int mi = 0; mi < h->nmeas; mi++) {
    const omeas* m = &h->meas[mi]; const opoint* p = &h->points[m->point];
    const ochain* sc = &h->chains[p->chain]; const ochain* oc = &h->chains[m->chain];
    double Jo[ORC_MAX_CHAIN][12], Js[ORC_MAX_CHAIN][12], Jp[6];
    const int mask = linearize(h, m, Jo, Js, Jp);
    double w = m->omega;
    if (h->robust) { double rho[3]; robustify(h, meas_chi2(h, m), rho); w *= rho[1]; }
    /* slots */
    const double* J[2*ORC_MAX_CHAIN]; int U[2*ORC_MAX_CHAIN]; int ns = 0;
    for (int i = 0; i < oc->len; i++) if (mask & (1 << i)) { J[ns] = Jo[i]; U[ns++] = h->poses[oc->v[i]].unk; }
    for (int i = 0; i < sc->len; i++) if (mask & (1 << (ORC_MAX_CHAIN+i))) { J[ns] = Js[i]; U[ns++] = h->poses[sc->v[i]].unk; }
    for (int a = 0; a < ns; a++) {
      double* b = h->bp + 6*U[a];
      for (int r = 0; r < 6; r++) b[r] += -w*(J[a][r]*m->e[0] + J[a][6+r]*m->e[1]);
      for (int c2 = a; c2 < ns; c2++) {
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
          const double v = w*(J[a][r]*J[c2][c] + J[a][6+r]*J[c2][6+c]);
          if (c2 != a && U[a] == U[c2] && !h->dup_symmetric) {      /* one vertex at two positions: upper triangle of Ja^T Omega Jb, mirrored */
            if (r <= c) { h->Hpp[(size_t)(6*U[a]+r)*np + 6*U[a]+c] += v; if (r < c) h->Hpp[(size_t)(6*U[a]+c)*np + 6*U[a]+r] += v; }
            continue;
          }
          h->Hpp[(size_t)(6*U[a]+r)*np + 6*U[c2]+c] += v;
          if (c2 != a || U[a] != U[c2]) { if (c2 != a) h->Hpp[(size_t)(6*U[c2]+c)*np + 6*U[a]+r] += v; }
        }
      }
    }
    if (mask & (1 << (2*ORC_MAX_CHAIN))) {
      double* V = h->V + 9*(size_t)p->unk; double* g = h->g + 3*(size_t)p->unk;
      for (int r = 0; r < 3; r++) {
        g[r] += -w*(Jp[r]*m->e[0] + Jp[3+r]*m->e[1]);
        for (int c = 0; c < 3; c++) V[3*r+c] += w*(Jp[r]*Jp[c] + Jp[3+r]*Jp[3+c]);
      }
      for (int a = 0; a < ns; a++) {
        double* Wb = h->W + 18*(size_t)find_inc(h, p, U[a]);
        for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++)
          Wb[3*r+c] += w*(J[a][r]*Jp[c] + J[a][6+r]*Jp[3+c]);
      }
    }
  }
}

/* in-place lower Cholesky of a dense symmetric n x n row-major matrix; returns 0 if SPD */
static int chol_dense(double* A, int n) {
  for (int i = 0; i < n; i++) {
    double* Ai = A + (size_t)i*n;
    for (int j = 0; j <= i; j++) {
      const double* Aj = A + (size_t)j*n;
      double s = Ai[j];
      for (int k = 0; k < j; k++) s -= Ai[k]*Aj[k];
      if (i == j) { if (!(s > 0.0)) return -1; Ai[j] = sqrt(s); }
      else Ai[j] = s/Aj[j];
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; i++) { double s = b[i]; const double* Li = L + (size_t)i*n; for (int k = 0; k < i; k++) s -= Li[k]*b[k]; b[i] = s/Li[i]; }
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k*n + i]*b[k]; b[i] = s/L[(size_t)i*n + i]; }
}
/* symmetric 3x3 inverse through its Cholesky factor; returns 0 if SPD */
static int inv3_spd(const double* A, double* I) {
  double L[9]; memcpy(L, A, 72);
  if (chol_dense(L, 3)) return -1;
  for (int c = 0; c < 3; c++) { double e[3] = {0,0,0}; e[c] = 1; chol_solve(L, 3, e); I[c] = e[0]; I[3+c] = e[1]; I[6+c] = e[2]; }
  return 0;
}

/* (H + lambda I) x = b with the points eliminated first.  returns 0 ok, -1 = not SPD (ok2=false) */
/* points eliminated: S = Hpp + lambda I - sum W (V + lambda I)^-1 W^T, r = bp - sum W (V + lambda I)^-1 g  (Vinv: nfl x 9) */
static int reduce_system(orc_ba* h, double lambda, double* S, double* r, double* Vinv) {
  const int np = h->np;
  memcpy(S, h->Hpp, sizeof(double)*(size_t)np*np); memcpy(r, h->bp, sizeof(double)*np);
  for (int i = 0; i < np; i++) S[(size_t)i*np + i] += lambda;
  for (int l = 0; l < h->nfl; l++) {
    const opoint* p = &h->points[h->fl_point[l]];
    double Vl[9]; memcpy(Vl, h->V + 9*(size_t)l, 72); Vl[0] += lambda; Vl[4] += lambda; Vl[8] += lambda;
    double* Vi = Vinv + 9*(size_t)l;
    if (inv3_spd(Vl, Vi)) return -1;
    const double* g = h->g + 3*(size_t)l;
    for (int a = 0; a < p->in; a++) {
      const double* Wa = h->W + 18*(size_t)(p->is + a); const int ua = h->inc_pose[p->is + a];
      double Y[18];
      for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 3; c++)
        Y[3*rr+c] = Wa[3*rr]*Vi[c] + Wa[3*rr+1]*Vi[3+c] + Wa[3*rr+2]*Vi[6+c];
      for (int rr = 0; rr < 6; rr++) r[6*ua+rr] -= Y[3*rr]*g[0] + Y[3*rr+1]*g[1] + Y[3*rr+2]*g[2];
      for (int b2 = 0; b2 < p->in; b2++) {
        const double* Wb = h->W + 18*(size_t)(p->is + b2); const int ub = h->inc_pose[p->is + b2];
        for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 6; c++)
          S[(size_t)(6*ua+rr)*np + 6*ub+c] -= Y[3*rr]*Wb[3*c] + Y[3*rr+1]*Wb[3*c+1] + Y[3*rr+2]*Wb[3*c+2];
      }
    }
  }
  return 0;
}
static int solve_system(orc_ba* h, double lambda, double* x) {
  if (h->solver == 1) return solve_system_sparse(h, lambda, x);
  if (h->solver == 2) return solve_system_par(h, lambda, x);
  const int np = h->np;
  double* S = h->S; double* r = (double*)malloc(sizeof(double)*(np + 1));
  double* Vinv = (double*)malloc(sizeof(double)*((size_t)h->nfl*9 + 1));
  int fail = reduce_system(h, lambda, S, r, Vinv) ? 1 : 0;
  if (!fail && np > 0 && chol_dense(S, np)) fail = 1;
  if (!fail) {
    if (np > 0) chol_solve(S, np, r);
    memcpy(x, r, sizeof(double)*np);
    for (int l = 0; l < h->nfl; l++) {
      const opoint* p = &h->points[h->fl_point[l]];
      const double* Vi = Vinv + 9*(size_t)l; const double* g = h->g + 3*(size_t)l;
      double t[3] = { g[0], g[1], g[2] };
      for (int a = 0; a < p->in; a++) {
        const double* Wa = h->W + 18*(size_t)(p->is + a); const double* xa = x + 6*h->inc_pose[p->is + a];
        for (int c = 0; c < 3; c++) for (int rr = 0; rr < 6; rr++) t[c] -= Wa[3*rr+c]*xa[rr];
      }
      m3v(Vi, t, x + np + 3*(size_t)l);
    }
  }
  free(Vinv); free(r);
  return fail ? -1 : 0;
}
static void gather_b(orc_ba* h) {
  memcpy(h->ball, h->bp, sizeof(double)*h->np);
  memcpy(h->ball + h->np, h->g, sizeof(double)*(size_t)h->nfl*3);
}

int orc_ba_debug_solve(orc_ba* h, double lambda, double* x_schur, double* x_dense) {
  if (!h->prepared) orc_ba_prepare(h);
  compute_active_errors(h); h->need_recompute = 1; (void)active_robust_chi2(h);
  build_system(h);
  int rc = solve_system(h, lambda, x_schur);
  if (rc) return rc;
  /* dense: assemble the whole (6P+3N)^2 matrix */
  const int n = h->nx, np = h->np;
  double* H = (double*)calloc((size_t)n*n, 8);
  for (int i = 0; i < np; i++) for (int j = 0; j < np; j++) H[(size_t)i*n + j] = h->Hpp[(size_t)i*np + j];
  for (int l = 0; l < h->nfl; l++) {
    const opoint* p = &h->points[h->fl_point[l]];
    for (int rr = 0; rr < 3; rr++) for (int c = 0; c < 3; c++) H[(size_t)(np+3*l+rr)*n + np+3*l+c] = h->V[9*(size_t)l + 3*rr + c];
    for (int a = 0; a < p->in; a++) {
      const double* Wa = h->W + 18*(size_t)(p->is + a); const int ua = h->inc_pose[p->is + a];
      for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 3; c++) {
        H[(size_t)(6*ua+rr)*n + np+3*l+c] = Wa[3*rr+c]; H[(size_t)(np+3*l+c)*n + 6*ua+rr] = Wa[3*rr+c];
      }
    }
  }
  for (int i = 0; i < n; i++) H[(size_t)i*n + i] += lambda;
  gather_b(h); memcpy(x_dense, h->ball, sizeof(double)*n);
  rc = chol_dense(H, n);
  if (!rc) chol_solve(H, n, x_dense);
  free(H);
  return rc;
}
/* the reduced pose system at the current state: out = [S (np x np, full symmetric) | rhs (np) | J^T r of the poses (np)] */
int orc_ba_debug_system(orc_ba* h, double lambda, double* out) {
  if (!h->prepared) orc_ba_prepare(h);
  const int np = h->np;
  if (!out) return np;
  compute_active_errors(h); h->need_recompute = 1; (void)active_robust_chi2(h);
  build_system(h);
  double* Vinv = (double*)malloc(sizeof(double)*((size_t)h->nfl*9 + 1));
  const int rc = reduce_system(h, lambda, out, out + (size_t)np*np, Vinv);
  free(Vinv);
  memcpy(out + (size_t)np*np + np, h->bp, sizeof(double)*np);
  return rc ? -1 : np;
}
double orc_ba_debug_robust_chi2(orc_ba* h, double* sigma_sq_raw) {
  if (!h->prepared) orc_ba_prepare(h);
  compute_active_errors(h); h->need_recompute = 1;
  double c = active_robust_chi2(h);
  if (sigma_sq_raw) *sigma_sq_raw = h->sigma_sq;
  return c;
}
void orc_ba_eval(orc_ba* h, double* chi2_out, double* err_out) {
  if (!h->prepared) orc_ba_prepare(h);
  compute_active_errors(h);
  for (int i = 0; i < h->nmeas; i++) {
    if (chi2_out) chi2_out[i] = meas_chi2(h, &h->meas[i]);
    if (err_out) { err_out[2*i] = h->meas[i].e[0]; err_out[2*i+1] = h->meas[i].e[1]; }
  }
}

/* g2o push / pop / discardTop on all active vertices [3P-memory] */
static void push_state(orc_ba* h) {
  for (int i = 0; i < h->nfp; i++) { opose* p = &h->poses[h->fp_pose[i]]; p->Tbak = p->T; }
  for (int i = 0; i < h->nfl; i++) { opoint* p = &h->points[h->fl_point[i]]; memcpy(p->xbak, p->x, 24); }
}
static void pop_state(orc_ba* h) {
  for (int i = 0; i < h->nfp; i++) { opose* p = &h->poses[h->fp_pose[i]]; p->T = p->Tbak; }
  for (int i = 0; i < h->nfl; i++) { opoint* p = &h->points[h->fl_point[i]]; memcpy(p->x, p->xbak, 24); }
}
static void apply_update(orc_ba* h, const double* x) {
  if (h->solver == 2) { apply_update_par(h, x); return; }
  for (int i = 0; i < h->nfp; i++) pose_oplus(&h->poses[h->fp_pose[i]].T, x + 6*i);
  for (int i = 0; i < h->nfl; i++) point_oplus(h->points[h->fl_point[i]].x, x + h->np + 3*(size_t)i);
}
static void add_log(orc_ba* h, const orc_iter_log* l) {
  GROW(h->logs, h->clogs, h->nlogs + 1, orc_iter_log); h->logs[h->nlogs++] = *l;
}
static int terminate_flag(volatile unsigned char* f) { return f && *f; }

/* ChainBundle::Compute, ChainBundle.cc:1305-1451, with g2o SparseOptimizer::optimize and
 * OptimizationAlgorithmLevenberg::solve restated [3P-memory]. */
int orc_ba_compute(orc_ba* h, volatile unsigned char* abort_flag, int n_iter, double user_lambda) {
  /* Initialize(), :1284-1298 */
  h->noutliers = 0; h->nlogs = 0;
  int conv_mag = 0, conv_res = 0;
  orc_ba_prepare(h);                    /* initializeOptimization */
  h->converged = 0;
  /* NOTE: CheckConvergedResidualAction::_dLastChi2 (:1068) is set once at construction and is NOT
   * reset by SetAbortFlag/SetNotConverged, so a second Compute on the same object (two-step mode,
   * BundleAdjusterMulti.cc:210-224) compares against the last chi2 of the first.  Mirrored: the
   * member is initialised in orc_ba_create only. */
  /* :1317-1323 */
  compute_active_errors(h); h->need_recompute = 1;
  if (h->nmeas > 0) (void)active_robust_chi2(h);
  h->total_iterations = 0;
  int nCounter;
  /* ---- optimize(nNumIter) ---- */
  if (h->nx == 0) nCounter = -1;        /* "0 vertices to optimize" */
  else {
    int cj = 0; int ok = 1; double ni = 2; int lev_its = 0;
    double lastx_rms = 0;
    for (int it = 0; it < n_iter && !terminate_flag(abort_flag) && ok; it++) {
      h->need_recompute = 1;                                    /* preIteration: UpdateSigmaSquaredAction :913-917 */
      orc_iter_log lg; memset(&lg, 0, sizeof lg);
      /* ---- solve(it) ---- */
      compute_active_errors(h);
      double currentChi = active_robust_chi2(h);
      double tempChi = currentChi;
      lg.chi2_start = currentChi; lg.sigma_sq = h->sigma_sq;
      build_system(h);
      if (it == 0) {                                            /* computeLambdaInit */
        if (user_lambda > 0) h->lambda = user_lambda;
        else {
          double maxd = 0;
          for (int i = 0; i < h->np; i++) { double d = fabs(h->Hpp[(size_t)i*h->np + i]); if (d > maxd) maxd = d; }
          for (int l = 0; l < h->nfl; l++) for (int k = 0; k < 3; k++) { double d = fabs(h->V[9*(size_t)l + 4*k]); if (d > maxd) maxd = d; }
          h->lambda = h->var_tau*maxd;
        }
        ni = 2;
      }
      double rho = 0; int qmax = 0; int accepted = 0;
      gather_b(h);
      do {
        push_state(h);
        int ok2 = (h->fail_trial > 0 && ++h->trial_no == h->fail_trial) ? 0       /* test switch: this trial's factorisation "fails" */
                  : (solve_system(h, h->lambda, h->x) == 0);     /* x untouched on failure */
        apply_update(h, h->x);
        compute_active_errors(h);
        tempChi = active_robust_chi2(h);
        if (!ok2) tempChi = DBL_MAX;
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < h->nx; j++) scale += h->x[j]*(h->lambda*h->x[j] + h->ball[j]);
        scale += h->var_rho_eps;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2*rho - 1), 3);
          if (h->var_accept != 2) alpha = fmin(alpha, 2./3.);
          double sf = fmax(1./3., alpha);
          if (h->var_accept == 1) sf = 1./3.;
          h->lambda *= sf; ni = 2; currentChi = tempChi; accepted = 1;
        } else {
          if (h->var_reject == 1) h->lambda *= 2; else { h->lambda *= ni; ni *= 2; }
          pop_state(h); accepted = 0;
        }
        qmax++;
      } while (rho < 0 && qmax < h->max_trials && !terminate_flag(abort_flag));
      lev_its = qmax;
      ok = !(qmax == h->max_trials || rho == 0);               /* Terminate */
      if (h->verbose) compute_active_errors(h);                 /* optimize(): verbose recomputes errors */
      ++cj;
      /* ---- postIteration actions ---- */
      { /* CheckConvergedUpdateMagAction :1009-1047 */
        double ss = 0; for (int i = 0; i < h->nx; i++) ss += h->x[i]*h->x[i];
        double rms = sqrt(ss/h->nx); lastx_rms = rms;
        if (rms < h->rms_limit && !h->no_converge) { conv_mag = 1; if (abort_flag) *abort_flag = 1; }
      }
      { /* CheckConvergedResidualAction :1091-1118 */
        double cur = active_robust_chi2(h);
        double pct = (h->last_chi2_action - cur)/h->last_chi2_action;
        if (!h->no_converge) {
          if (pct >= 0 && pct <= h->pct_limit) { conv_res = 1; if (abort_flag) *abort_flag = 1; }
          else if (cur == 0) { conv_res = 1; if (abort_flag) *abort_flag = 1; }
        }
        h->last_chi2_action = cur;
        lg.chi2_end = cur;
      }
      h->total_iterations += lev_its;                            /* UpdateTotalIterationsAction :958-963 */
      lg.lambda_end = h->lambda; lg.trials = qmax; lg.accepted = accepted; lg.rms_update = lastx_rms;
      add_log(h, &lg);
    }
    nCounter = cj;
  }
  /* :1339-1345 */
  compute_active_errors(h); h->need_recompute = 1;
  if (h->nmeas > 0) (void)active_robust_chi2(h);
  h->converged = (conv_mag || conv_res);                          /* :1347 */
  int external_abort = 0;
  if (terminate_flag(abort_flag) && !h->converged) external_abort = 1;   /* :1355-1360 */
  if (nCounter == 0 && !external_abort) return -1;                 /* :1362-1363 */
  if (nCounter == 0 && terminate_flag(abort_flag)) return 0;       /* :1365-1366 */
  if (h->tukey && h->nmeas > 0) {                                  /* :1368-1399 */
    double* v = (double*)malloc(sizeof(double)*h->nmeas);
    for (int i = 0; i < h->nmeas; i++) v[i] = fabs(meas_chi2(h, &h->meas[i]));
    double s = orc_tukey_sigma_squared(v, h->nmeas); free(v);
    const double mins = h->min_sigma*h->min_sigma;
    if (s < mins) s = mins;
    for (int i = 0; i < h->nmeas; i++) {
      const omeas* m = &h->meas[i];
      if (orc_tukey_weight(fabs(meas_chi2(h, m)), s) == 0) {
        GROW(h->outliers, h->coutliers, 3*(h->noutliers + 1), int);
        h->outliers[3*h->noutliers]   = h->points[m->point].id;
        h->outliers[3*h->noutliers+1] = h->poses[h->chains[m->chain].v[0]].id;   /* vertices().front() :1394 */
        h->outliers[3*h->noutliers+2] = m->cam;
        h->noutliers++;
      }
    }
  }
  /* point-depth covariance, :1401-1448.  computeMarginals uses the Hessian of the last
   * buildSystem (no lambda) [3P-memory]; (H^-1)_ll = V^-1 + V^-1 W^T S^-1 W V^-1. */
  if (h->nfp < 3 && nCounter > 0) {
    int okm = 1; const int np = h->np;
    double* S = h->S; memcpy(S, h->Hpp, sizeof(double)*(size_t)np*np);
    double* Vinv = (double*)malloc(sizeof(double)*((size_t)h->nfl*9 + 1));
    for (int l = 0; l < h->nfl && okm; l++) {
      const opoint* p = &h->points[h->fl_point[l]]; double* Vi = Vinv + 9*(size_t)l;
      if (inv3_spd(h->V + 9*(size_t)l, Vi)) { okm = 0; break; }
      for (int a = 0; a < p->in; a++) {
        const double* Wa = h->W + 18*(size_t)(p->is + a); const int ua = h->inc_pose[p->is + a];
        double Y[18];
        for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 3; c++) Y[3*rr+c] = Wa[3*rr]*Vi[c] + Wa[3*rr+1]*Vi[3+c] + Wa[3*rr+2]*Vi[6+c];
        for (int b2 = 0; b2 < p->in; b2++) {
          const double* Wb = h->W + 18*(size_t)(p->is + b2); const int ub = h->inc_pose[p->is + b2];
          for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 6; c++)
            S[(size_t)(6*ua+rr)*np + 6*ub+c] -= Y[3*rr]*Wb[3*c] + Y[3*rr+1]*Wb[3*c+1] + Y[3*rr+2]*Wb[3*c+2];
        }
      }
    }
    if (okm && np > 0 && chol_dense(S, np)) okm = 0;
    if (okm) {
      double* cov = (double*)malloc(sizeof(double)*(h->nfl + 1)); int nc = 0;
      for (int l = 0; l < h->nfl; l++) {
        const opoint* p = &h->points[h->fl_point[l]]; const double* Vi = Vinv + 9*(size_t)l;
        /* u = W * Vi[:,2] stacked over incidences (np-vector), c22 = Vi[2][2] + u^T S^-1 u */
        double* u = (double*)calloc(np + 1, 8);
        for (int a = 0; a < p->in; a++) {
          const double* Wa = h->W + 18*(size_t)(p->is + a); const int ua = h->inc_pose[p->is + a];
          for (int rr = 0; rr < 6; rr++) u[6*ua+rr] += Wa[3*rr]*Vi[2] + Wa[3*rr+1]*Vi[5] + Wa[3*rr+2]*Vi[8];
        }
        double* s = (double*)malloc(sizeof(double)*(np + 1)); memcpy(s, u, sizeof(double)*np);
        if (np > 0) chol_solve(S, np, s);
        double c22 = Vi[8]; for (int i = 0; i < np; i++) c22 += u[i]*s[i];
        cov[nc++] = c22; free(u); free(s);
      }
      if (nc > 0) { qsort(cov, nc, 8, cmp_double); h->max_cov = cov[nc/2]; }   /* :1431-1437 */
      else h->max_cov = DBL_MAX;                                                 /* :1441 */
      free(cov);
    } else h->max_cov = 0;                                                       /* :1447 */
    free(Vinv);
  } else h->max_cov = 0;                                                         /* :1444-1448 */
  return nCounter;
}

int orc_ba_converged(orc_ba* h) { return h->converged; }
int orc_ba_total_iterations(orc_ba* h) { return h->total_iterations; }
int orc_ba_get_point(orc_ba* h, int id, double x[3]) {
  if (id <= 0 || id >= h->next_id || h->id_kind[id] != 2) return -1;
  memcpy(x, h->points[h->id_index[id]].x, 24); return 0;
}
int orc_ba_get_pose(orc_ba* h, int id, double R[9], double t[3]) {
  if (id <= 0 || id >= h->next_id || h->id_kind[id] != 1) return -1;
  memcpy(R, h->poses[h->id_index[id]].T.R, 72); memcpy(t, h->poses[h->id_index[id]].T.t, 24); return 0;
}
int orc_ba_num_outliers(orc_ba* h) { return h->noutliers; }
int orc_ba_get_outliers(orc_ba* h, int* out, int cap) {
  int n = h->noutliers < cap ? h->noutliers : cap; memcpy(out, h->outliers, sizeof(int)*3*(size_t)n); return n;
}
double orc_ba_sigma_squared(orc_ba* h) { return h->sigma_sq; }                 /* :1470-1474 */
double orc_ba_mean_chi_squared(orc_ba* h) { return active_robust_chi2(h)/h->nmeas; }   /* :1476-1481 */
double orc_ba_max_cov(orc_ba* h) { return h->max_cov; }
double orc_ba_lambda(orc_ba* h) { return h->lambda; }                          /* :1483-1487 */
int orc_ba_num_iter_logs(orc_ba* h) { return h->nlogs; }
int orc_ba_get_iter_logs(orc_ba* h, orc_iter_log* out, int cap) {
  int n = h->nlogs < cap ? h->nlogs : cap; memcpy(out, h->logs, sizeof(orc_iter_log)*(size_t)n); return n;
}

#include "ba_baseline.inc"
