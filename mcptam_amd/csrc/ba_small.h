// ba_small.h -- fused kernels for bundles whose every stage is one workgroup's latency (gfx950, fp64).
//
// BundleAdjustRecent (src/BundleAdjusterBase.cc:188-265) hands ChainBundle a window of a few thousand points and ~20 k
// measurements ten times a second: on 256 compute units every kernel of an LM iteration is then a handful of workgroups and the
// iteration is the sum of ~20 launch-to-launch latencies.  For such bundles (one rank, <= SMALL_MEAS measurements) what can be
// merged is:
//   k_head_small     the head of an iteration (median of |chi2|, sigma block, robust chi2 of the state): see below;
//   k_trial_apply    the step of a trial in one launch: workgroup 0 the pose update (oplus) and the chain transforms that hang off it
//                    (PoseChainHelper::UpdateTransforms, ChainBundle.cc:120-150), the others the point back-substitution -- in place
//                    of k_update_poses + k_chains + k_backsub.
// Same arithmetic per element and the same order of every sum as the kernels they replace: a bundle gives the same bits on either
// path (tests/test_ba_gpu.py::test_small_bundle_scheduling_does_not_change_a_single_bit).
#pragma once
#include "ba_kernels.h"
#include "ba_select.h"

namespace mcp {

constexpr int SMALL_MEAS = 32768;
constexpr int SMALL_CHAINS = 2048;

// ---- k_head_small: the head of an LM iteration in one workgroup -- exact median of |chi2| (Huber::FindSigmaSquared,
// include/mcptam/MEstimator.h:194-204), the sigma block, the robust chi2 of the state (activeRobustChi2, src/ChainBundle.cc:871-897)
// -> out[off], in place of memset + two histogram passes + gather + finish + robust sum + final sum (7 stream operations whose
// launch-to-launch latencies, not their work, are what the head of a 20 k-measurement bundle costs).
// The select is the radix select of ba_select.h, arranged so that the array is swept three times (+ once for the sum):
//   sweep 1   with the digit 0 (sign + exponent bits) of the LAST median as the guess (sig_prev[3]: the median moves little between
//             iterations): how many elements lie below that digit, how many share it, and digit 1 of those that do; a wrong
//             guess costs two more sweeps;
//   sweep 2   the elements that share the selected 22-bit prefix (a handful) are copied to LDS; the remaining digits are resolved there;
//   sum       robustified chi2 in the order k_robust_sum + k_final_sums take it, so that a bundle gives the same bits whichever
//             path evaluates it: 256 consecutive measurements = one partial (four wavefront trees, added in order), then the
//             partials, one per thread of a 256-thread block, through the same tree.
// (hs_hist_add, the long way for digit 0: a chi2 array has a handful of distinct digit-0 values -- left to plain LDS atomics, 20 k
// updates of the same few counters serialise; the wavefront first counts its lanes per digit, HS_AGG rounds, one atomic each.)
#ifdef MCP_HS_PROF      // phase stamps of k_head_small (build variant hsprof; printed by the solver's event trace)
__device__ unsigned long long g_hs_prof[16];
#define HS_STAMP(i) do { if (threadIdx.x == 0) g_hs_prof[i] = clock64(); } while (0)
#else
#define HS_STAMP(i) do {} while (0)
#endif
constexpr int HS_AGG = 6;
constexpr int HS_CAND = 2048;
// the elements that share the guessed digit 0 (sign + ten exponent bits: two binades, a third to a half of a chi2 array) are kept in LDS as the
// first sweep finds them: the second sweep -- the same 155 KB through ONE compute unit's ~128 cache lines in flight, 11 k cycles -- then runs over
// that stash (dynamic LDS, one region per wavefront; a region that overflows: the sweep over the array, as before)
constexpr int HS_STASH = 14336;
// sum over the wavefront in the association of `for (o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64)` (lane 0's value; the other
// lanes hold partial trees), without the LDS crossbar: lanes i+32 and i+16 through v_permlane32_swap / v_permlane16_swap, the four
// steps inside a 16-lane row through DPP row shifts
__device__ inline double hs_tree64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  { const auto r = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); const auto q = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v += __hiloint2double((int)q[1], (int)r[1]); }
  lo = __double2loint(v); hi = __double2hiint(v);
  { const auto r = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); const auto q = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v += __hiloint2double((int)q[1], (int)r[1]); }
#define HS_ROWSHL(n) { lo = __double2loint(v); hi = __double2hiint(v); \
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x100 + n, 0xf, 0xf, false), h2 = __builtin_amdgcn_update_dpp(0, hi, 0x100 + n, 0xf, 0xf, false); \
    v += __hiloint2double(h2, l2); }
  HS_ROWSHL(8) HS_ROWSHL(4) HS_ROWSHL(2) HS_ROWSHL(1)
#undef HS_ROWSHL
  return v;
}
__device__ inline void hs_hist_add(unsigned int* hist, bool has, unsigned int bin, int lane) {
  unsigned long long todo = __ballot(has);
#pragma unroll 1
  for (int r = 0; r < HS_AGG && todo; ++r) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned int b = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);      // (leader is wavefront-uniform)
    const unsigned long long same = __ballot(has && bin == b);
    if (lane == leader) atomicAdd(&hist[b], (unsigned int)__popcll(same));
    todo &= ~same;
  }
  if (has && ((todo >> lane) & 1ull)) atomicAdd(&hist[bin], 1u);
}
// every thread of the 1024: f(valid, x[i]) for i = t, t + 1024, ...; HS_DEPTH loads in flight per thread (one workgroup pulls the
// array through one compute unit: what a sweep costs is the number of dependent round trips, 2 for 20 k elements); all threads make
// the same number of calls (f may vote across the wavefront)
constexpr int HS_DEPTH = 16;
template <class F>
__device__ inline void hs_sweep(const double* __restrict__ x, int n, F f) {
  const int t = threadIdx.x;
#pragma unroll 1
  for (int j0 = 0; j0 < n; j0 += 1024*HS_DEPTH) {
    double v[HS_DEPTH];
#pragma unroll
    for (int u = 0; u < HS_DEPTH; ++u) v[u] = x[min(j0 + 1024*u + t, n - 1)];
#pragma unroll
    for (int u = 0; u < HS_DEPTH; ++u) {
      if (j0 + 1024*u >= n) break;                    // (uniform)
      f(j0 + 1024*u + t < n, v[u]);
    }
  }
}
__global__ void __launch_bounds__(1024)
k_head_small(int n, int robust, const double* __restrict__ chi2, unsigned long long k, double n_total, double min_sigma_sq,
             const double* __restrict__ sig_prev, double* __restrict__ med_out, double* __restrict__ sig, double* __restrict__ sig_copy,
             double* __restrict__ out, int off, int do_sum /* 0: median + sigma block only (the sum is taken elsewhere, off the critical path) */) {
  __shared__ unsigned int hist0[SEL_BINS], hist1[SEL_BINS];
  __shared__ unsigned int wcnt[16][2];
  __shared__ unsigned long long sc[1024/64 + 3];
  __shared__ unsigned long long s_st[2];
  __shared__ double cand[HS_CAND];
  __shared__ unsigned int ncand;
  __shared__ double s_sig[2];
  __shared__ double red[4];
  __shared__ double wpart[SMALL_MEAS/64];
  __shared__ unsigned int nstash;
  extern __shared__ double stash[];              // HS_STASH doubles
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  HS_STAMP(0);
  if (robust) {
    const int sh0 = sel_shift(0), sh1 = sel_shift(1);
    // Digit 0 (sign + exponent bits) is not histogrammed at all if the guess holds: with the digit 0 of the LAST median (first
    // iteration: of some element) as the guess, all the select needs to know is how many elements lie below that digit and how many
    // share it -- two votes and two scalar additions per element, no shared counter (the few digit-0 values of a chi2 array as LDS
    // atomics, even pre-counted per wavefront, were 23 us of this kernel).  Digit 1 is counted under the guess in the same sweep.
    const double guess = (sig_prev[3] != 0.0) ? sig_prev[3] : chi2[n/2];
    const unsigned int pred = (unsigned int)(((unsigned long long)__double_as_longlong(fabs(guess))) >> sh0) & (SEL_BINS - 1);
    for (int b = t; b < SEL_BINS; b += 1024) hist1[b] = 0u;
    if (t == 0) { ncand = 0u; nstash = 0u; }
    __syncthreads();
    unsigned int cl = 0u, ce = 0u, mystash = 0u;     // (wavefront-uniform)
    hs_sweep(chi2, n, [&](bool valid, double v) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(v));
      const unsigned int b0 = (unsigned int)(key >> sh0) & (SEL_BINS - 1);
      cl += (unsigned int)__popcll(__ballot(valid && b0 < pred));
      const unsigned long long me = __ballot(valid && b0 == pred);
      ce += (unsigned int)__popcll(me);
      if (valid && b0 == pred) {
        atomicAdd(&hist1[(unsigned int)(key >> sh1) & (SEL_BINS - 1)], 1u);
        // the wavefront's own region of the stash, its lanes' slots by their rank in the vote: no shared counter, no wait
        const unsigned int slot = mystash + __builtin_amdgcn_mbcnt_hi((unsigned int)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)me, 0u));
        if (slot < (unsigned int)(HS_STASH/16)) stash[wave*(HS_STASH/16) + slot] = fabs(v);
      }
      mystash += (unsigned int)__popcll(me);
    });
    if (lane == 0) { wcnt[wave][0] = cl; wcnt[wave][1] = ce; if (mystash > (unsigned int)(HS_STASH/16)) atomicOr(&nstash, 1u); }      // (nstash: "a region overflowed")
    __syncthreads();
    HS_STAMP(1);
    unsigned long long below = 0ull, equal = 0ull;
#pragma unroll
    for (int w = 0; w < 16; ++w) { below += wcnt[w][0]; equal += wcnt[w][1]; }
    int bin0 = (int)pred; unsigned long long k1 = k - below;
    bool stashed = nstash == 0u;                     // (uniform; complete: the barrier above)
    HS_STAMP(2);
    if (!(below <= k && k < below + equal)) {         // (uniform) the guess was wrong: both digits the long way
      stashed = false;
      for (int b = t; b < SEL_BINS; b += 1024) { hist0[b] = 0u; hist1[b] = 0u; }
      __syncthreads();
      hs_sweep(chi2, n, [&](bool valid, double v) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(v));
        hs_hist_add(hist0, valid, (unsigned int)(key >> sh0) & (SEL_BINS - 1), lane);
      });
      __syncthreads();
      unsigned int in0;
      lds_find_bin<1024>(hist0, k, bin0, k1, in0, sc);
      hs_sweep(chi2, n, [&](bool valid, double v) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(v));
        if (valid && ((unsigned int)(key >> sh0) & (SEL_BINS - 1)) == (unsigned int)bin0) atomicAdd(&hist1[(unsigned int)(key >> sh1) & (SEL_BINS - 1)], 1u);
      });
      __syncthreads();
    }
    HS_STAMP(3);
    int bin1; unsigned long long k2; unsigned int in1;
    lds_find_bin<1024>(hist1, k1, bin1, k2, in1, sc);
    HS_STAMP(4);
    const unsigned long long prefix = ((unsigned long long)bin0 << sh0) | ((unsigned long long)bin1 << sh1);
    const unsigned long long himask = ~0ull << sh1;
    if (stashed) {
      const unsigned int ns = wcnt[wave][1];           // (this wavefront's matches = the fill of its region)
      for (unsigned int i = lane; i < ns; i += 64u) {
        const double a = stash[wave*(HS_STASH/16) + i];
        if ((((unsigned long long)__double_as_longlong(a)) & himask) == prefix) {
          const unsigned int idx = atomicAdd(&ncand, 1u);
          if (idx < (unsigned int)HS_CAND) cand[idx] = a;
        }
      }
    } else
    hs_sweep(chi2, n, [&](bool valid, double v) {
      const double a = fabs(v);
      if (valid && (((unsigned long long)__double_as_longlong(a)) & himask) == prefix) {
        const unsigned int idx = atomicAdd(&ncand, 1u);
        if (idx < (unsigned int)HS_CAND) cand[idx] = a;
      }
    });
    __syncthreads();
    HS_STAMP(5);
    const unsigned int nc = ncand;
    unsigned long long sel;
    if (nc <= 64u) {
      // a handful of candidates (the usual case): candidate j's rank is the number of candidates below it (equal ones by position); the
      // one of rank k2 is the median -- no histogram passes for a wavefront's worth of keys
      if (t == 0) s_st[0] = 0ull;
      __syncthreads();
      if (t < (int)nc) {
        const unsigned long long mine = (unsigned long long)__double_as_longlong(cand[t]);
        unsigned int r = 0u;
        for (unsigned int j = 0; j < nc; ++j) { const unsigned long long o = (unsigned long long)__double_as_longlong(cand[j]); r += (o < mine || (o == mine && j < (unsigned int)t)) ? 1u : 0u; }
        if ((unsigned long long)r == k2) s_st[0] = mine;
      }
      __syncthreads();
      sel = s_st[0];
    } else if (nc <= (unsigned int)HS_CAND)
      sel = lds_radix_select<1024>((int)nc, k2, 2, prefix, [&](int i, unsigned long long& key) { key = (unsigned long long)__double_as_longlong(cand[i]); return true; }, hist0, sc, s_st);
    else      // thousands of values equal in their top 22 bits: the remaining digits over the array itself
      sel = lds_radix_select<1024>(n, k2, 2, prefix, [&](int i, unsigned long long& key) { key = (unsigned long long)__double_as_longlong(fabs(chi2[i])); return true; }, hist0, sc, s_st);
    HS_STAMP(6);
    if (t == 0) {
      const double md = __longlong_as_double((long long)sel);
      double s = 1.4826*(1 + 5.0/mest_denom(n_total))*sqrt(md);
      s = 1.345*s;
      const double s2 = s*s;
      const double lim = (s2 < min_sigma_sq) ? min_sigma_sq : s2;
      const double sl = sqrt(lim);
      med_out[0] = md;
      sig[0] = s2; sig[1] = lim; sig[2] = sl; sig[3] = md;
      if (sig_copy) { sig_copy[0] = s2; sig_copy[1] = lim; sig_copy[2] = sl; sig_copy[3] = md; }
      s_sig[0] = lim; s_sig[1] = sl;
    }
    __syncthreads();
  }
  HS_STAMP(7);
  if (do_sum) {
    const double lim = robust ? s_sig[0] : 0.0, sl = robust ? s_sig[1] : 0.0;
    int j = 0;
    hs_sweep(chi2, n, [&](bool valid, double v) {
      // call number j of this thread holds element 64 (wave + 16 j) + lane: segment wave + 16 j of the array
      double c = 0.0;
      if (valid) { c = v; if (robust) { double r0, r1; robustify(v, lim, sl, r0, r1); c = r0; } }
      c = hs_tree64(c);
      const int sgm = wave + 16*j;
      if (lane == 0 && 64*sgm < n) wpart[sgm] = c;
      ++j;
    });
    __syncthreads();
    HS_STAMP(8);
    const int nseg = (n + 63)/64, nbe = (n + 255)/256;
    double v = 0.0;
    if (t < 256) {
      if (t < nbe) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) a += (4*t + q < nseg) ? wpart[4*t + q] : 0.0;
        v = a;
      }
      v = hs_tree64(v);
      if (lane == 0) red[wave] = v;
    }
    __syncthreads();
    if (t == 0) { double tot = 0.0; for (int q = 0; q < 4; ++q) tot += red[q]; out[off] = tot; }
  }
  HS_STAMP(9);
#ifdef MCP_HS_PROF
  if (t == 0 && robust) g_hs_prof[10] = ncand;
#endif
}

// T_trial = exp(x) * T_cur for the free poses, the pose part of sum x(lambda x + b) and sum x^2 (as k_update_poses), then the chain
// transforms of the trial state (as k_chains).  One workgroup (ncb = 1: small bundles), or ncb of them with TA_CHAINS chains each
// (round 6, large maps): EVERY chain workgroup computes all trial poses into its LDS (a pose is ~1 us of arithmetic for one thread,
// and there are at most TA_MAX_POSES of them) and takes the chains' links from there; workgroup 0 also stores the poses and their
// two sums.  Nothing crosses a workgroup, so the pose update, the chain transforms and the points' back-substitution -- three
// dependent launches before -- run side by side in one.  Same arithmetic per pose and per chain as the separate kernels.
constexpr int TA_CHAINS = 64;             // chains per chain workgroup when there are several
constexpr int TA_MAX_POSES = 256;         // poses a chain workgroup keeps in LDS (24 KB); beyond: the separate kernels
__device__ __forceinline__ void update_chains_body(const DevProblem& P, int cb, int ncb, double* __restrict__ tl, double lambda, const double* __restrict__ xp, const double* __restrict__ bp,
                const double* __restrict__ T_cur, double* T_trial, double* __restrict__ out /*[2]*/, double* __restrict__ xp_keep,
                double* __restrict__ first, double* __restrict__ second, double* __restrict__ last) {
  __shared__ double lds[4];
  double sc = 0.0, ss = 0.0;
  if (cb == 0) for (int i = threadIdx.x; i < P.np; i += 256) xp_keep[i] = xp[i];
  for (int i = threadIdx.x; i < P.npose; i += 256) {
    const int u = P.pose_unk[i];
    double* tp = tl + 12*(size_t)i;
    if (u < 0) {          // (a fixed pose: every candidate state holds it)
      const double* p = T_trial + 12*(size_t)i;
#pragma unroll
      for (int k = 0; k < 12; ++k) tp[k] = p[k];
      continue;
    }
    const double* d = xp + 6*(size_t)u;
    Se3 E, T, R;
    se3_exp(d, E);
    load_se3(T_cur + 12*(size_t)i, T);
    se3_compose(E, T, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) tp[k] = R.R[k];
    tp[9] = R.t[0]; tp[10] = R.t[1]; tp[11] = R.t[2];
    if (cb == 0) {
      double* o = T_trial + 12*(size_t)i;
#pragma unroll
      for (int k = 0; k < 9; ++k) o[k] = R.R[k];
      o[9] = R.t[0]; o[10] = R.t[1]; o[11] = R.t[2];
      for (int k = 0; k < 6; ++k) { sc += d[k]*(lambda*d[k] + bp[6*(size_t)u + k]); ss += d[k]*d[k]; }
    }
  }
  if (cb == 0) {
    const double t0 = block_sum<256>(sc, lds);
    const double t1 = block_sum<256>(ss, lds);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; }
  }
  __syncthreads();
  const double* pose_T = tl;
  const int c_lo = ncb == 1 ? 0 : cb*TA_CHAINS, c_hi = ncb == 1 ? P.nchain : min(P.nchain, (cb + 1)*TA_CHAINS);
  if (ncb > 1 && threadIdx.x >= TA_CHAINS) return;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += 256) {
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

// The step of a trial applied in ONE launch: workgroups [0, ncb) update the poses and recompute the chain transforms (above), the
// others back-substitute and update the points (k_backsub's body) -- the two halves read the same solution vector and touch
// disjoint state, so nothing orders them; k_eval, which needs both, is the next launch of the stream.
// Dynamic LDS: 12 npose doubles.
static_assert(BS_BLOCK == 256, "k_trial_apply runs both bodies with 256 threads");
__global__ void __launch_bounds__(256)
k_trial_apply(DevProblem P, int ncb, double lambda, const double* __restrict__ xp, const double* __restrict__ bp,
              const double* __restrict__ T_cur, double* T_trial, double* __restrict__ out /*[2]*/, double* __restrict__ xp_keep,
              double* __restrict__ first, double* __restrict__ second, double* __restrict__ last,
              const double* __restrict__ g, const double* __restrict__ W, const double* __restrict__ Vinv, const double* __restrict__ pt_cur,
              double* __restrict__ pt_trial, double* __restrict__ xl, double* __restrict__ part_scale, double* __restrict__ part_ss) {
  extern __shared__ __attribute__((aligned(16))) double ta_poses[];
  if ((int)blockIdx.x < ncb) update_chains_body(P, (int)blockIdx.x, ncb, ta_poses, lambda, xp, bp, T_cur, T_trial, out, xp_keep, first, second, last);
  else backsub_body(P, (int)blockIdx.x - ncb, lambda, xp, g, W, Vinv, pt_cur, pt_trial, xl, part_scale, part_ss);
}

}  // namespace mcp
