// ba_kernels.h -- HIP kernels of the ChainBundle LM iteration (gfx950, fp64).
//
// One LM outer iteration (g2o OptimizationAlgorithmLevenberg::solve as configured at
// src/ChainBundle.cc:1156-1161) maps to:
//   k_chains      PoseChainHelper::UpdateTransforms               ChainBundle.cc:120-150
//   k_eval        EdgeChainMeas::computeError + chi2              :376-417
//   select_*      Huber/Tukey::FindSigmaSquared median            MEstimator.h:109-124,194-204
//   k_robust_sum  SparseOptimizer::activeRobustChi2 + robustify   ChainBundle.cc:871-897
//   k_linearize   EdgeChainMeas::linearizeOplus + constructQuadraticForm  :449-749
//   k_schur*      point elimination of (H + lambda I) x = b
//   chol_*        dense Cholesky of the reduced pose system (ba_chol.h)
//   k_backsub / k_update_poses   back-substitution + oplus        :82-86, :237-281
#pragma once
#include "ba_device.h"
#include "ba_select.h"

namespace mcp {

// device-resident problem description (SoA, sorted by point)
struct DevProblem {
  // cameras
  const mcp_camera* cams; int ncam;
  // chains
  int nchain;
  const int* chain_len;       // [nchain]
  const int* chain_pose;      // [nchain*MAXC] pose array index
  // poses
  int npose;
  const int* pose_unk;        // [npose] unknown index or -1
  // points
  int npoint;
  const int* pt_chain;        // [npoint]
  const int* pt_unk;          // [npoint] free-point index or -1
  const unsigned char* pt_fixed;
  // measurements (sorted by point)
  int nmeas;
  const int* m_pt; const int* m_chain; const unsigned char* m_cam; const unsigned short* m_mask;
  const double* m_u; const double* m_v; const double* m_omega;
  const int* slot_start;      // [nmeas+1]
  const int* slot_unk;        // [nslot] pose unknown of the slot
  const int* slot_inc;        // [nslot] incidence index (W block) or -1 (fixed point)
  // incidences (point-pose blocks), CSR by free point
  int nfl, ninc, np;
  const int* l_i0;            // [nfl] incidence range of free point l
  const int* l_i1;
  const int* inc_unk;         // [ninc]
  const int* fl_point;        // [nfl] free point -> point index
  int robust;
  // points in group order ("sp" = sorted point slot); measurements and incidences are contiguous per sp
  int nsp, ngroup;
  const int* sp_pt;           // [nsp] point index
  const int* sp_m;            // [nsp+1] measurement range
  const int* sp_i;            // [nsp+1] incidence range
  const unsigned char* sp_big;// [nsp] 1: more than GRP_LMAX poses -> generic (global atomics) path
  const int* m_sp;            // [nmeas] sp of the measurement
  const int* l_sp;            // [nfl] sp of free point l
  const int* g_sp0;           // [ngroup+1] sp range of the group
  const int* g_pose;          // [ngroup*GRP_LMAX] pose unknowns of the group (ascending), -1 padded
  const unsigned char* slot_lp;   // [nslot] local pose index of the slot inside its group
  const unsigned char* slot_first;// [nslot] 1: first contribution to its incidence (store), 0: accumulate
  const unsigned char* inc_lp;    // [ninc] local pose index of the incidence
  const unsigned char* inc_mixed; // [ninc] 1: block fed both from registers (first source link) and through memory
  // staging of the groups' local tiles (ba_group.h): group g owns the 6x6 blocks [g_blk0[g], g_blk0[g+1])
  const int* g_blk0;              // [ngroup+1]
  const unsigned char* blk_pair;  // [nblk] local pose pair of the block, la << 4 | lb (la >= lb)
  const int* blk_dst;             // [nblk] where the block is staged: position in the destination-ordered staging array
  const int* rhs_dst;             // [ngroup*GRP_LMAX] staged rhs row of (group, local pose), or -1
  const int* m_last;              // [nmeas] m_chain[m]*MAXC + chain_len - 1: the observer chain's last link (its transform from world) without the hop over chain_len
  const int* sp_unk;              // [nsp] free-point index of the sorted point (-1: fixed, or a point of the generic path) = pt_unk[sp_pt[sp]]
};
constexpr int MAXC = MCP_MAX_CHAIN;             // links per pose chain; per-chain arrays are strided by it
constexpr int MAXC_LOG = 3;
static_assert((1 << MAXC_LOG) == MAXC && 2*MAXC <= 16, "m_mask is 16 bits: observer links in bits [0, MAXC), source links in [MAXC, 2 MAXC)");
constexpr int GRP_LMAX = 16;      // poses per group (6*16 = 96 local dof)
constexpr int GRP_PTS = 64;       // points per group (one lane each in k_linearize_group)
constexpr int GRP_DOF = 6*GRP_LMAX;

// chain transforms: first[c*4+i] = link i from world (12 doubles), second[c*4+i] = rotation of last from link i
__global__ void k_chains(DevProblem P, const double* __restrict__ pose_T, double* __restrict__ first,
                         double* __restrict__ second, double* __restrict__ last) {
  const int c = blockIdx.x*blockDim.x + threadIdx.x;
  if (c >= P.nchain) return;
  const int len = P.chain_len[c];
  Se3 acc; se3_identity(acc);
  for (int i = 0; i < len; ++i) {
    Se3 v; const double* p = pose_T + 12*(size_t)P.chain_pose[c*MAXC+i];
#pragma unroll
    for (int k = 0; k < 9; ++k) v.R[k] = p[k];
    v.t[0] = p[9]; v.t[1] = p[10]; v.t[2] = p[11];
    se3_compose(v, acc, acc);
    double* o = first + 12*(size_t)(c*MAXC+i);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = acc.R[k];
    o[9] = acc.t[0]; o[10] = acc.t[1]; o[11] = acc.t[2];
  }
  {
    double* o = last + 12*(size_t)c;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = acc.R[k];
    o[9] = acc.t[0]; o[10] = acc.t[1]; o[11] = acc.t[2];
  }
  double Rc[9] = {1,0,0, 0,1,0, 0,0,1};
  for (int i = len - 1; i >= 0; --i) {
    double* o = second + 9*(size_t)(c*MAXC+i);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = Rc[k];
    mat3_mul(Rc, pose_T + 12*(size_t)P.chain_pose[c*MAXC+i], Rc);
  }
}

__device__ inline void load_se3(const double* p, Se3& T) {
#pragma unroll
  for (int k = 0; k < 9; ++k) T.R[k] = p[k];
  T.t[0] = p[9]; T.t[1] = p[10]; T.t[2] = p[11];
}

// block-wide deterministic sum (fixed tree): returns the total in thread 0
template <int BLOCK>
__device__ inline double block_sum(double v, double* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) { for (int i = 0; i < BLOCK/64; ++i) t += lds[i]; }
  __syncthreads();
  return t;
}
template <int BLOCK>
__device__ inline double block_max(double v, double* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) { for (int i = 0; i < BLOCK/64; ++i) t = fmax(t, lds[i]); }
  __syncthreads();
  return t;
}

// Huber on chi2 with the adaptive sigma (RobustKernelAdaptive::robustify, ChainBundle.cc:871-897)
__device__ inline void robustify(double e2, double s2, double s, double& rho0, double& rho1) {
  if (e2 <= s2) { rho0 = fabs(e2); rho1 = 1.0; }
  else { const double e = sqrt(e2); rho0 = 2.0*s*e - s2; rho1 = s/e; }
}

// sigma block in device memory: [0] raw sigma^2, [1] limited sigma^2, [2] limited sigma, [3] median
constexpr int EVAL_BLOCK = 256;

// computeError + chi2 for every measurement; optional fused robust sum (sigma known: LM trials).
// err_out (2 per measurement) only for introspection.
template <bool SUM>
__device__ __forceinline__ void eval_body(const DevProblem& P, const double* __restrict__ pt_x, const double* __restrict__ last,
       double* __restrict__ chi2, double* __restrict__ err_out, const double* __restrict__ sigma,
       double* __restrict__ partial) {
  __shared__ double lds[EVAL_BLOCK/64];
  const int m = blockIdx.x*EVAL_BLOCK + threadIdx.x;
  double contrib = 0.0;
  if (m < P.nmeas) {
    const int pt = P.m_pt[m];
    Se3 Ts, To;
    load_se3(last + 12*(size_t)P.pt_chain[pt], Ts);
    load_se3(last + 12*(size_t)P.m_chain[m], To);
    const double x[3] = { pt_x[3*(size_t)pt], pt_x[3*(size_t)pt+1], pt_x[3*(size_t)pt+2] };
    double xw[3], xc[3];
    se3_apply_inv(Ts, x, xw);
    se3_apply(To, xw, xc);
    Projection pr;
    cam_project<false>(P.cams[P.m_cam[m]], xc, pr);
    const double e0 = P.m_u[m] - pr.u, e1 = P.m_v[m] - pr.v;
    double c = P.m_omega[m]*(e0*e0 + e1*e1);
    if (P.pt_fixed[pt] && P.robust) c = -c;
    chi2[m] = c;
    if (err_out) { err_out[2*(size_t)m] = e0; err_out[2*(size_t)m+1] = e1; }
    if (SUM) {
      if (P.robust) { double r0, r1; robustify(c, sigma[1], sigma[2], r0, r1); contrib = r0; }
      else contrib = c;
    }
  }
  if (SUM) {
    const double t = block_sum<EVAL_BLOCK>(contrib, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
  }
}

template <bool SUM>
__global__ void __launch_bounds__(EVAL_BLOCK)
k_eval(DevProblem P, const double* __restrict__ pt_x, const double* __restrict__ last,
       double* __restrict__ chi2, double* __restrict__ err_out, const double* __restrict__ sigma,
       double* __restrict__ partial) {
  eval_body<SUM>(P, pt_x, last, chi2, err_out, sigma, partial);
}

// robust sum over an existing chi2 array (first evaluation of an iteration, after the median)
__global__ void __launch_bounds__(EVAL_BLOCK)
k_robust_sum(int n, int robust, const double* __restrict__ chi2, const double* __restrict__ sigma,
             double* __restrict__ partial) {
  __shared__ double lds[EVAL_BLOCK/64];
  const int m = blockIdx.x*EVAL_BLOCK + threadIdx.x;
  double contrib = 0.0;
  if (m < n) {
    const double c = chi2[m];
    if (robust) { double r0, r1; robustify(c, sigma[1], sigma[2], r0, r1); contrib = r0; }
    else contrib = c;
  }
  const double t = block_sum<EVAL_BLOCK>(contrib, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// fixed-order final reduction of up to 3 partial arrays into out[off..off+2] (single block); out[off+3] = failure flag
// mail (may be null): device-visible pinned host memory.  The first mail_count entries of `out` are forwarded there, then the
// ticket mail[MAIL_TICKET] is released at system scope -- the host polls the ticket instead of enqueueing a copy and waiting on
// the stream (the LM accept/reject decision is one PCIe write away instead of a blit kernel + a stream wait).
constexpr int MAIL_TICKET = 31;
__device__ __forceinline__ void final_sums_body(int n0, const double* p0, int n1, const double* p1, int n2, const double* p2,
             double* __restrict__ out, int off, const int* __restrict__ fail, double* mail, int mail_count, unsigned long long ticket,
             const double* ride_src, int ride_dst, const double* start_src = nullptr /* entries [24, 29) of the mailbox from here instead of out */) {
  __shared__ double lds[4];
  const double* ps[3] = { p0, p1, p2 }; const int ns[3] = { n0, n1, n2 };
  for (int a = 0; a < 3; ++a) {
    if (!ps[a]) continue;
    double v = 0.0;
    for (int i = threadIdx.x; i < ns[a]; i += 256) v += ps[a][i];
    const double t = block_sum<256>(v, lds);
    if (threadIdx.x == 0) out[off + a] = t;
  }
  // (bit 2: a hand-off of the one-launch factorisation timed out, ba_chol2.h -- the host redoes the solve with the per-step kernels;
  //  1e9 survives the sum over ranks as "some rank had it")
  if (fail && threadIdx.x == 0) out[off + 3] = (fail[0] & 4) ? 1e9 : ((fail[0] != 0) ? 1.0 : 0.0);
  if (ride_src && threadIdx.x == 0) out[ride_dst] = ride_src[0];
  if (mail) {
    __syncthreads();                                 // thread 0's stores to `out` above; the other entries come from earlier kernels of the stream
    for (int i = threadIdx.x; i < mail_count; i += 256) mail[i] = (start_src && i >= 24 && i < 29) ? start_src[i - 24] : out[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((unsigned long long*)(mail + MAIL_TICKET), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void __launch_bounds__(256)
k_final_sums(int n0, const double* p0, int n1, const double* p1, int n2, const double* p2,
             double* __restrict__ out, int off, const int* __restrict__ fail, double* mail = nullptr, int mail_count = 0, unsigned long long ticket = 0,
             const double* ride_src = nullptr, int ride_dst = 0 /* out[ride_dst] = ride_src[0]: a value of an earlier kernel joins this block's all-reduce */,
             const double* start_src = nullptr /* the iteration-start block forwarded to the host lives here (a per-trial head's, ba_solver.hip enqueue_head) */) {
  final_sums_body(n0, p0, n1, p1, n2, p2, out, off, fail, mail, mail_count, ticket, ride_src, ride_dst, start_src);
}

// ------------------------------------------------------------------------------------------
// linearize: Jacobians + weighted normal-equation accumulation, one thread per measurement.
// U (np x np, lower triangle maintained), bp, V (6 per free point: xx,xy,xz,yy,yz,zz), g, W (18 per incidence).
struct SlotGeom { double B[6]; double base[3]; double sign; };

__device__ inline void slot_jacobian(const SlotGeom& s, double* J /*2x6 row-major*/) {
  // J[:,k] = sign * B * gen_k(base);  k<3: B[:,k];  k>=3: B * (e_{k-3} x base)
  const double* B = s.B; const double* p = s.base; const double sg = s.sign;
  J[0] = sg*B[0]; J[1] = sg*B[1]; J[2] = sg*B[2];
  J[6] = sg*B[3]; J[7] = sg*B[4]; J[8] = sg*B[5];
  J[3]  = sg*(-B[1]*p[2] + B[2]*p[1]);  J[9]  = sg*(-B[4]*p[2] + B[5]*p[1]);
  J[4]  = sg*( B[0]*p[2] - B[2]*p[0]);  J[10] = sg*( B[3]*p[2] - B[5]*p[0]);
  J[5]  = sg*(-B[0]*p[1] + B[1]*p[0]);  J[11] = sg*(-B[3]*p[1] + B[4]*p[0]);
}

__device__ inline void make_slot(int side, int link, const double* A /*2x3*/, const double* xw,
                                 const double* first, const double* second, int oc, int sc,
                                 const double* Robs_last, SlotGeom& s) {
  Se3 F;
  if (side == 0) {
    load_se3(first + 12*(size_t)(oc*MAXC+link), F);
    se3_apply(F, xw, s.base);
    const double* R2 = second + 9*(size_t)(oc*MAXC+link);
    // B = A * R2
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s.B[3*r+c] = A[3*r]*R2[c] + A[3*r+1]*R2[3+c] + A[3*r+2]*R2[6+c];
    s.sign = 1.0;
  } else {
    load_se3(first + 12*(size_t)(sc*MAXC+link), F);
    se3_apply(F, xw, s.base);
    double Rrel[9];
    mat3_mul_t(Robs_last, F.R, Rrel);          // R(T_obs * T_src_i^-1)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s.B[3*r+c] = A[3*r]*Rrel[c] + A[3*r+1]*Rrel[3+c] + A[3*r+2]*Rrel[6+c];
    s.sign = -1.0;
  }
}

constexpr int LIN_BLOCK = 128;

__global__ void __launch_bounds__(LIN_BLOCK)
k_linearize(DevProblem P, int only_big, const double* __restrict__ pt_x, const double* __restrict__ first,
            const double* __restrict__ second, const double* __restrict__ sigma,
            double* __restrict__ U, double* __restrict__ bp, double* __restrict__ V,
            double* __restrict__ g, double* __restrict__ W) {
  const int m = blockIdx.x*LIN_BLOCK + threadIdx.x;
  if (m >= P.nmeas) return;
  if (only_big && !P.sp_big[P.m_sp[m]]) return;
  const int pt = P.m_pt[m];
  const int oc = P.m_chain[m], sc = P.pt_chain[pt];
  const int olen = P.chain_len[oc], slen = P.chain_len[sc];
  const int mask = P.m_mask[m];
  const int lpt = P.pt_unk[pt];
  if (mask == 0 && lpt < 0) return;
  Se3 Ts, To;
  load_se3(first + 12*(size_t)(sc*MAXC + slen - 1), Ts);
  load_se3(first + 12*(size_t)(oc*MAXC + olen - 1), To);
  const double x[3] = { pt_x[3*(size_t)pt], pt_x[3*(size_t)pt+1], pt_x[3*(size_t)pt+2] };
  double xw[3], xc[3];
  se3_apply_inv(Ts, x, xw);
  se3_apply(To, xw, xc);
  Projection pr;
  cam_project<true>(P.cams[P.m_cam[m]], xc, pr);
  const double e0 = P.m_u[m] - pr.u, e1 = P.m_v[m] - pr.v;
  const double omega = P.m_omega[m];
  double c2 = omega*(e0*e0 + e1*e1);
  if (P.pt_fixed[pt] && P.robust) c2 = -c2;
  double w = omega;
  if (P.robust) { double r0, r1; robustify(c2, sigma[1], sigma[2], r0, r1); w *= r1; }
  double dT[3], dP[3];
  cam_sphere_deriv(xc, dT, dP);
  // A = -D * [dT; dP]  (2x3): change of the ERROR per unit motion of the camera-frame point
  double A[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    A[c]     = -(pr.D[0]*dT[c] + pr.D[1]*dP[c]);
    A[3 + c] = -(pr.D[2]*dT[c] + pr.D[3]*dP[c]);
  }
  // point block
  double Jp[6];
  if (lpt >= 0) {
    double Rp[9], dir[3], rho;
    point_frame(x, Rp, dir, rho);
    double rx[3], g0[3], g1[3], c0[3], c1[3];
    mat3_vec(Rp, x, rx);
    generator(3, rx, g0); generator(4, rx, g1);
    mat3t_vec(Rp, g0, c0); mat3t_vec(Rp, g1, c1);
    const double c2v[3] = { -x[0]/rho, -x[1]/rho, -x[2]/rho };
    double Rcs[9];
    mat3_mul_t(To.R, Ts.R, Rcs);
    double AR[6];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) AR[3*r+c] = A[3*r]*Rcs[c] + A[3*r+1]*Rcs[3+c] + A[3*r+2]*Rcs[6+c];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      Jp[3*r]   = AR[3*r]*c0[0]  + AR[3*r+1]*c0[1]  + AR[3*r+2]*c0[2];
      Jp[3*r+1] = AR[3*r]*c1[0]  + AR[3*r+1]*c1[1]  + AR[3*r+2]*c1[2];
      Jp[3*r+2] = AR[3*r]*c2v[0] + AR[3*r+1]*c2v[1] + AR[3*r+2]*c2v[2];
    }
    double* Vp = V + 6*(size_t)lpt; double* gp = g + 3*(size_t)lpt;
    unsafeAtomicAdd(Vp + 0, w*(Jp[0]*Jp[0] + Jp[3]*Jp[3]));
    unsafeAtomicAdd(Vp + 1, w*(Jp[0]*Jp[1] + Jp[3]*Jp[4]));
    unsafeAtomicAdd(Vp + 2, w*(Jp[0]*Jp[2] + Jp[3]*Jp[5]));
    unsafeAtomicAdd(Vp + 3, w*(Jp[1]*Jp[1] + Jp[4]*Jp[4]));
    unsafeAtomicAdd(Vp + 4, w*(Jp[1]*Jp[2] + Jp[4]*Jp[5]));
    unsafeAtomicAdd(Vp + 5, w*(Jp[2]*Jp[2] + Jp[5]*Jp[5]));
#pragma unroll
    for (int c = 0; c < 3; ++c) unsafeAtomicAdd(gp + c, -w*(Jp[c]*e0 + Jp[3+c]*e1));
  }
  // pose slots
  const int s0 = P.slot_start[m], ns = P.slot_start[m+1] - s0;
  const int np = P.np;
  int ia = 0;
  for (int bit_a = 0; bit_a < 2*MAXC && ia < ns; ++bit_a) {
    if (!(mask & (1 << bit_a))) continue;
    SlotGeom sa; double Ja[12];
    make_slot(bit_a >> MAXC_LOG, bit_a & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sa);
    slot_jacobian(sa, Ja);
    const int ua = P.slot_unk[s0 + ia];
    double* b = bp + 6*(size_t)ua;
#pragma unroll
    for (int r = 0; r < 6; ++r) unsafeAtomicAdd(b + r, -w*(Ja[r]*e0 + Ja[6+r]*e1));
    // diagonal block, lower triangle
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c)
        unsafeAtomicAdd(U + (size_t)(6*ua + r)*np + 6*ua + c, w*(Ja[r]*Ja[c] + Ja[6+r]*Ja[6+c]));
    if (lpt >= 0) {
      double* Wb = W + 18*(size_t)P.slot_inc[s0 + ia];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) unsafeAtomicAdd(Wb + 3*r + c, w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]));
    }
    // cross blocks with later slots
    int ib = ia + 1;
    for (int bit_b = bit_a + 1; bit_b < 2*MAXC && ib < ns; ++bit_b) {
      if (!(mask & (1 << bit_b))) continue;
      SlotGeom sb; double Jb[12];
      make_slot(bit_b >> MAXC_LOG, bit_b & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sb);
      slot_jacobian(sb, Jb);
      const int ub = P.slot_unk[s0 + ib];
      if (ua > ub) {
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
          unsafeAtomicAdd(U + (size_t)(6*ua + r)*np + 6*ub + c, w*(Ja[r]*Jb[c] + Ja[6+r]*Jb[6+c]));
      } else if (ub > ua) {
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
          unsafeAtomicAdd(U + (size_t)(6*ub + r)*np + 6*ua + c, w*(Jb[r]*Ja[c] + Jb[6+r]*Ja[6+c]));
      } else {
        // The same pose vertex at two positions of the edge (BundleAdjusterCalib: the relative camera pose in both chains).  g2o adds
        // Ja^T Omega Jb ONCE to the vertex's diagonal block and its linear solver reads that block's upper triangle
        // (BaseMultiEdge::constructQuadraticForm / BlockSolver::buildStructure / fillCCS [3P-memory]): the symmetric matrix that is solved has (Ja^T Omega Jb)(c, r), c <= r, at the lower entry (r, c).
        for (int r = 0; r < 6; ++r) for (int c = 0; c <= r; ++c)
          unsafeAtomicAdd(U + (size_t)(6*ua + r)*np + 6*ua + c, w*(Ja[c]*Jb[r] + Ja[6+c]*Jb[6+r]));
      }
      ++ib;
    }
    ++ia;
  }
}

// max |diag| of U and V  (computeLambdaInit [g2o])
__global__ void __launch_bounds__(256)
k_max_diag(int np, const double* __restrict__ U, int stride, int nfl, const double* __restrict__ V, double* out) {
  __shared__ double lds[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) v = fmax(v, fabs(U[(size_t)i*stride]));
  for (int i = threadIdx.x; i < nfl; i += 256) {
    const double* Vp = V + 6*(size_t)i;
    v = fmax(v, fmax(fabs(Vp[0]), fmax(fabs(Vp[3]), fabs(Vp[5]))));
  }
  const double t = block_max<256>(v, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// the same over many workgroups (a maximum does not care in which order it is taken): 50 000 points' diagonals were 57 us of ONE workgroup in the
// first iteration of every call at the metric size.  part[b] = block b's maximum; k_max_of takes the maximum of those.
__global__ void __launch_bounds__(256)
k_max_diag_part(int np, const double* __restrict__ U, int stride, int nfl, const double* __restrict__ V, double* __restrict__ part) {
  __shared__ double lds[4];
  double v = 0.0;
  const int nt = gridDim.x*256, t0 = blockIdx.x*256 + threadIdx.x;
  for (int i = t0; i < np; i += nt) v = fmax(v, fabs(U[(size_t)i*stride]));
  for (int i = t0; i < nfl; i += nt) {
    const double* Vp = V + 6*(size_t)i;
    v = fmax(v, fmax(fabs(Vp[0]), fmax(fabs(Vp[3]), fabs(Vp[5]))));
  }
  const double t = block_max<256>(v, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256)
k_max_of(int n, const double* __restrict__ part, double* out) {
  __shared__ double lds[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) v = fmax(v, part[i]);
  const double t = block_max<256>(v, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// A trial solve can carry speculative systems: the same linearisation damped with the lambdas the LM schedule will use
// next if this trial (and the following ones) are rejected (lambda*ni, lambda*ni*2ni, ...).  Kernels that build or factor
// the reduced system take the batch index q from blockIdx.y; system q lives q*sstride doubles behind system 0 in the
// [S | rhs | bp] buffer, its V^-1 blocks q*vstride doubles behind in Vinv, its failure flag in fail[q].
constexpr int MAX_SYS = 4;
struct SysBatch { double lambda[MAX_SYS]; double lambda_init[MAX_SYS]; size_t sstride; size_t vstride;
                  size_t ststride, strstride; /* per-system strides of the staged Schur blocks / local right-hand sides */ };

// gather (pack = 1) / scatter (pack = 0) of the structurally non-zero 32x32 tiles of the reduced system plus its
// two trailing vectors [rhs | bp] (2*np doubles) to / from a contiguous buffer: the payload of the per-trial all-reduce.
// block b < ntiles handles tile b, block ntiles handles the vectors.  `src`/`dst` roles swap with `pack`.
__global__ void __launch_bounds__(256)
k_pack_tiles(const double* __restrict__ src, int np, const int* __restrict__ tiles, int ntiles, double* __restrict__ dst, int pack,
             size_t full_stride, size_t pack_stride) {
  if (blockIdx.y) { src += blockIdx.y*(pack ? full_stride : pack_stride); dst += blockIdx.y*(pack ? pack_stride : full_stride); }
  const int b = blockIdx.x;
  if (b == ntiles) {
    const size_t off_full = (size_t)np*np, off_pack = (size_t)ntiles*1024;
    for (int i = threadIdx.x; i < 2*np; i += 256) {
      if (pack) dst[off_pack + i] = src[off_full + i]; else dst[off_full + i] = src[off_pack + i];
    }
    return;
  }
  const int ti = tiles[b] >> 16, tj = tiles[b] & 0xffff;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int r = 32*ti + (e >> 5), c = 32*tj + (e & 31);
    if (r < np && c < np) {
      if (pack) dst[(size_t)b*1024 + e] = src[(size_t)r*np + c]; else dst[(size_t)r*np + c] = src[(size_t)b*1024 + e];
    } else if (pack) dst[(size_t)b*1024 + e] = 0.0;
  }
}

// inverse of the symmetric 3x3 (V + lambda I); returns false if not positive definite
__device__ inline bool inv_sym3(const double* V6, double lambda, double* I6) {
  const double a = V6[0] + lambda, b = V6[1], c = V6[2], d = V6[3] + lambda, e = V6[4], f = V6[5] + lambda;
  const double c00 = d*f - e*e, c01 = c*e - b*f, c02 = b*e - c*d;
  const double det = a*c00 + b*c01 + c*c02;
  const double m2 = a*d - b*b;
  const bool ok = (a > 0.0) && (m2 > 0.0) && (det > 0.0);
  const double id = 1.0/det;
  I6[0] = c00*id; I6[1] = c01*id; I6[2] = c02*id;
  I6[3] = (a*f - c*c)*id; I6[4] = (b*c - a*e)*id; I6[5] = m2*id;
  return ok;
}

// point elimination: S -= W Vinv W^T, rhs -= W Vinv g.  One wave per free point.
__global__ void __launch_bounds__(256)
k_schur(DevProblem P, int only_big, double lambda, const double* __restrict__ V, const double* __restrict__ g,
        const double* __restrict__ W, double* __restrict__ Vinv, double* __restrict__ S,
        double* __restrict__ rhs, int* __restrict__ fail, SysBatch sb) {
  if (blockIdx.y) { const int q = blockIdx.y; lambda = sb.lambda[q]; Vinv += q*sb.vstride; S += q*sb.sstride; rhs += q*sb.sstride; fail += q; }
  const int l = blockIdx.x*4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (l >= P.nfl) return;
  if (only_big && !P.sp_big[P.l_sp[l]]) return;
  double I6[6];
  const bool ok = inv_sym3(V + 6*(size_t)l, lambda, I6);
  if (lane == 0) {
    if (!ok) atomicOr(fail, 1);
#pragma unroll
    for (int k = 0; k < 6; ++k) Vinv[6*(size_t)l + k] = I6[k];
  }
  const int i0 = P.l_i0[l], q = P.l_i1[l] - i0;
  const double Vi[9] = { I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5] };
  const double g0 = g[3*(size_t)l], g1 = g[3*(size_t)l+1], g2 = g[3*(size_t)l+2];
  const int np = P.np;
  // rhs: q*6 items
  for (int it = lane; it < q*6; it += 64) {
    const int a = it / 6, r = it % 6;
    const double* Wa = W + 18*(size_t)(i0 + a) + 3*r;
    const double y0 = Wa[0]*Vi[0] + Wa[1]*Vi[3] + Wa[2]*Vi[6];
    const double y1 = Wa[0]*Vi[1] + Wa[1]*Vi[4] + Wa[2]*Vi[7];
    const double y2 = Wa[0]*Vi[2] + Wa[1]*Vi[5] + Wa[2]*Vi[8];
    unsafeAtomicAdd(rhs + 6*(size_t)P.inc_unk[i0 + a] + r, -(y0*g0 + y1*g1 + y2*g2));
  }
  const int items = q*q*36;
  for (int it = lane; it < items; it += 64) {
    const int pair = it / 36, e = it % 36;
    const int a = pair / q, b = pair % q, r = e / 6, c = e % 6;
    const int ua = P.inc_unk[i0 + a], ub = P.inc_unk[i0 + b];
    if (ua < ub || (ua == ub && c > r)) continue;
    const double* Wa = W + 18*(size_t)(i0 + a) + 3*r;
    const double* Wb = W + 18*(size_t)(i0 + b) + 3*c;
    const double y0 = Wa[0]*Vi[0] + Wa[1]*Vi[3] + Wa[2]*Vi[6];
    const double y1 = Wa[0]*Vi[1] + Wa[1]*Vi[4] + Wa[2]*Vi[7];
    const double y2 = Wa[0]*Vi[2] + Wa[1]*Vi[5] + Wa[2]*Vi[8];
    unsafeAtomicAdd(S + (size_t)(6*ua + r)*np + 6*ub + c, -(y0*Wb[0] + y1*Wb[1] + y2*Wb[2]));
  }
}

// pose update: T_trial = exp(x) * T_cur for free poses; pose part of sum x(lambda x + b) and sum x^2
__global__ void __launch_bounds__(256)
k_update_poses(DevProblem P, double lambda, const double* __restrict__ xp, const double* __restrict__ bp,
               const double* __restrict__ T_cur, double* __restrict__ T_trial, double* __restrict__ out /*[2]*/,
               double* __restrict__ xp_keep /* copy of the pose update: the solver's x if this trial is accepted */) {
  __shared__ double lds[4];
  double sc = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < P.np; i += 256) xp_keep[i] = xp[i];
  for (int i = threadIdx.x; i < P.npose; i += 256) {
    const int u = P.pose_unk[i];
    if (u < 0) continue;
    const double* d = xp + 6*(size_t)u;
    Se3 E, T, R;
    se3_exp(d, E);
    load_se3(T_cur + 12*(size_t)i, T);
    se3_compose(E, T, R);
    double* o = T_trial + 12*(size_t)i;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = R.R[k];
    o[9] = R.t[0]; o[10] = R.t[1]; o[11] = R.t[2];
    for (int k = 0; k < 6; ++k) { sc += d[k]*(lambda*d[k] + bp[6*(size_t)u + k]); ss += d[k]*d[k]; }
  }
  const double t0 = block_sum<256>(sc, lds);
  const double t1 = block_sum<256>(ss, lds);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

constexpr int BS_BLOCK = 256;
constexpr int BS_TPP = 4;          // threads per point in k_backsub
// back-substitution for the points + oplus; partial sums of x(lambda x + b) and x^2
// (body: block `blk` of the launch -- k_backsub passes blockIdx.x, k_trial_apply of ba_small.h blockIdx.x - 1)
__device__ __forceinline__ void backsub_body(const DevProblem& P, int blk, double lambda, const double* __restrict__ xp, const double* __restrict__ g,
          const double* __restrict__ W, const double* __restrict__ Vinv, const double* __restrict__ pt_cur,
          double* __restrict__ pt_trial, double* __restrict__ xl, double* __restrict__ part_scale,
          double* __restrict__ part_ss) {
  __shared__ double lds[BS_BLOCK/64];
  // BS_TPP threads per point share its incidences (each walk is a chain of dependent index -> pose-update loads), partial
  // sums are combined with a fixed shuffle tree
  const int l = (blk*BS_BLOCK + threadIdx.x)/BS_TPP, q = threadIdx.x & (BS_TPP - 1);
  double sc = 0.0, ss = 0.0;
  const bool valid = l < P.nfl;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
  if (valid) {
    const int i0 = P.l_i0[l], i1 = P.l_i1[l];
    for (int i = i0 + q; i < i1; i += BS_TPP) {
      const double* Wa = W + 18*(size_t)i; const double* xa = xp + 6*(size_t)P.inc_unk[i];
#pragma unroll
      for (int r = 0; r < 6; ++r) { t0 -= Wa[3*r]*xa[r]; t1 -= Wa[3*r+1]*xa[r]; t2 -= Wa[3*r+2]*xa[r]; }
    }
  }
#pragma unroll
  for (int o = 1; o < BS_TPP; o <<= 1) { t0 += __shfl_xor(t0, o, 64); t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
  if (valid && q == 0) {
    const double b0 = g[3*(size_t)l], b1 = g[3*(size_t)l+1], b2 = g[3*(size_t)l+2];
    t0 += b0; t1 += b1; t2 += b2;
    const double* I6 = Vinv + 6*(size_t)l;
    const double d[3] = { I6[0]*t0 + I6[1]*t1 + I6[2]*t2, I6[1]*t0 + I6[3]*t1 + I6[4]*t2, I6[2]*t0 + I6[4]*t1 + I6[5]*t2 };
    xl[3*(size_t)l] = d[0]; xl[3*(size_t)l+1] = d[1]; xl[3*(size_t)l+2] = d[2];
    sc = d[0]*(lambda*d[0] + b0) + d[1]*(lambda*d[1] + b1) + d[2]*(lambda*d[2] + b2);
    ss = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
    const int pt = P.fl_point[l];
    double o[3];
    point_oplus(pt_cur + 3*(size_t)pt, d, o);
    pt_trial[3*(size_t)pt] = o[0]; pt_trial[3*(size_t)pt+1] = o[1]; pt_trial[3*(size_t)pt+2] = o[2];
  }
  const double a = block_sum<BS_BLOCK>(sc, lds);
  const double b = block_sum<BS_BLOCK>(ss, lds);
  if (threadIdx.x == 0) { part_scale[blk] = a; part_ss[blk] = b; }
}
__global__ void __launch_bounds__(BS_BLOCK)
k_backsub(DevProblem P, double lambda, const double* __restrict__ xp, const double* __restrict__ g,
          const double* __restrict__ W, const double* __restrict__ Vinv, const double* __restrict__ pt_cur,
          double* __restrict__ pt_trial, double* __restrict__ xl, double* __restrict__ part_scale,
          double* __restrict__ part_ss) {
  backsub_body(P, blockIdx.x, lambda, xp, g, W, Vinv, pt_cur, pt_trial, xl, part_scale, part_ss);
}

// VertexRelPoint::oplusImpl (ChainBundle.cc:237-281) with a GIVEN update per free point: what g2o's update(_solver->x()) does with the
// x its solver still holds after a failed factorisation
__global__ void k_apply_point_step(DevProblem P, const double* __restrict__ xl, const double* __restrict__ pt_cur, double* __restrict__ pt_trial) {
  const int l = blockIdx.x*blockDim.x + threadIdx.x;
  if (l >= P.nfl) return;
  const int pt = P.fl_point[l];
  double o[3];
  point_oplus(pt_cur + 3*(size_t)pt, xl + 3*(size_t)l, o);
  pt_trial[3*(size_t)pt] = o[0]; pt_trial[3*(size_t)pt+1] = o[1]; pt_trial[3*(size_t)pt+2] = o[2];
}

// Tukey outlier flags in sorted order (ChainBundle.cc:1385-1398 with MEstimator.h:84-96)
__global__ void k_tukey_flags(int n, const double* __restrict__ chi2, double s2, unsigned char* __restrict__ flag) {
  const int m = blockIdx.x*blockDim.x + threadIdx.x;
  if (m >= n) return;
  const double e = fabs(chi2[m]);
  const double sq = (e > s2) ? 0.0 : 1.0 - (e/s2);
  flag[m] = (sq*sq == 0.0) ? 1 : 0;
}

// The same with the threshold taken on the device from the median the head of the last iteration left in the result block (res[med_idx]):
// Tukey::FindSigmaSquared (MEstimator.h:109-124) with the floor of ChainBundle.cc:1377-1383, the host's expression operation for operation --
// the flags need no round trip through the host between the median and this kernel.
__global__ void k_tukey_flags_dev(int n, const double* __restrict__ chi2, const double* __restrict__ res, int med_idx, double m_total, double min_sigma,
                                  unsigned long long* __restrict__ mask) {
  // one BIT per measurement, a 64-bit word per wavefront (round 6: a byte per measurement was 400 KB for the host to walk through at the
  // end of every solve -- 0.5 ms of a 15 ms call at the metric size; the words of a map that size are 50 KB, and the host visits set bits only)
  const int m = blockIdx.x*blockDim.x + threadIdx.x;
  bool out = false;
  if (m < n) {
    double s = 1.4826*(1 + 5.0/mest_denom(m_total))*sqrt(res[med_idx]);
    s = 4.6851*s;
    double s2 = s*s;
    const double mins = min_sigma*min_sigma;
    if (s2 < mins) s2 = mins;
    const double e = fabs(chi2[m]);
    const double sq = (e > s2) ? 0.0 : 1.0 - (e/s2);
    out = sq*sq == 0.0;
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(out);
  if ((threadIdx.x & 63) == 0 && m < n) mask[m >> 6] = b;
}
// the state of a solve (poses 12 doubles each, points 3) to ONE destination -- pinned host memory: two copies into pageable vectors were
// ~90 us of copy-engine operations and queue switches at the end of every BundleAdjustRecent call
__global__ void k_export_state(size_t nps, size_t npt, const double* __restrict__ pose, const double* __restrict__ pt, double* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
  if (i < nps) dst[i] = pose[i];
  else if (i < nps + npt) dst[i] = pt[i - nps];
}

// (2,2) entry of each free point's marginal covariance: Vi[2][2] + u^T Sinv u, u = W Vi[:,2]
// (ChainBundle.cc:1419-1437); np <= 12 here.
__global__ void k_point_cov22(DevProblem P, const double* __restrict__ Vinv, const double* __restrict__ W,
                              const double* __restrict__ Sinv, double* __restrict__ cov) {
  const int l = blockIdx.x*blockDim.x + threadIdx.x;
  if (l >= P.nfl) return;
  const double* I6 = Vinv + 6*(size_t)l;
  const double v2[3] = { I6[2], I6[4], I6[5] };
  double u[12];
  for (int i = 0; i < 12; ++i) u[i] = 0.0;
  for (int i = P.l_i0[l]; i < P.l_i1[l]; ++i) {
    const double* Wa = W + 18*(size_t)i; const int ua = P.inc_unk[i];
    for (int r = 0; r < 6; ++r) u[6*ua + r] += Wa[3*r]*v2[0] + Wa[3*r+1]*v2[1] + Wa[3*r+2]*v2[2];
  }
  double c = I6[5];
  const int np = P.np;
  for (int i = 0; i < np; ++i) { double s = 0.0; for (int j = 0; j < np; ++j) s += Sinv[i*np + j]*u[j]; c += u[i]*s; }
  cov[l] = c;
}

}  // namespace mcp
