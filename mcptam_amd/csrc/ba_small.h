// ba_small.h -- fused kernels for bundles whose every stage is one workgroup's latency (gfx950, fp64).
//
// BundleAdjustRecent (src/BundleAdjusterBase.cc:188-265) hands ChainBundle a window of a few thousand points and ~20 k
// measurements ten times a second: on 256 compute units every kernel of an LM iteration is then a handful of workgroups and the
// iteration is the sum of ~20 launch-to-launch latencies.  For such bundles (one rank, <= SMALL_MEAS measurements) what can be
// merged is:
//   k_update_chains  the trial's pose update (oplus) and the chain transforms that hang off it (PoseChainHelper::UpdateTransforms,
//                    ChainBundle.cc:120-150) -- in place of k_update_poses + k_chains.
// Same arithmetic per element as the kernels they replace; the sums are taken in a different (fixed) order.
#pragma once
#include "ba_kernels.h"
#include "ba_select.h"

namespace mcp {

constexpr int SMALL_MEAS = 32768;
constexpr int SMALL_CHAINS = 2048;

// one workgroup: T_trial = exp(x) * T_cur for the free poses, the pose part of sum x(lambda x + b) and sum x^2 (as k_update_poses),
// then the chain transforms of the trial state (as k_chains).  The poses written in the first half are read back by other
// threads of the same workgroup in the second: nothing has cached those lines before (a kernel starts with a clean L1).
__global__ void __launch_bounds__(256)
k_update_chains(DevProblem P, double lambda, const double* __restrict__ xp, const double* __restrict__ bp,
                const double* __restrict__ T_cur, double* T_trial, double* __restrict__ out /*[2]*/, double* __restrict__ xp_keep,
                double* __restrict__ first, double* __restrict__ second, double* __restrict__ last) {
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
  __threadfence();
  __syncthreads();
  const double* pose_T = T_trial;
  for (int c = threadIdx.x; c < P.nchain; c += 256) {
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
      double Rp[9]; const double* p = pose_T + 12*(size_t)P.chain_pose[c*MAXC+i];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = p[k];
      mat3_mul(Rc, Rp, Rc);
    }
  }
}

}  // namespace mcp
