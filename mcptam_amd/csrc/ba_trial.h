// ba_trial.h -- the tail of an LM trial when the map is sharded over several ranks.
//
// Every trial ends with ONE all-reduce of a small block: the four result scalars (robust chi2, the point parts of
// sum x(lambda x + b) and sum x^2, the failure flag; g2o's rho test of OptimizationAlgorithmLevenberg::solve needs their sums
// over all edges / unknowns), the iteration-start robust chi2 when it still has to be summed, and -- riding in
// the same payload -- the first two radix digits of the NEXT iteration's Huber median (RobustKernelData::RecomputeNow,
// src/ChainBundle.cc:810-833 with MEstimator.h:194-204): if this trial is accepted, its chi2 array is the one the next
// iteration takes the median of, so its digit histograms are that selection's first two passes.  The second digit can only be
// histogrammed for a KNOWN first digit; the first digit of |chi2|'s bit pattern is sign + the upper ten exponent bits (a bin
// spans a factor of four), and the median stays in the bin of the previous iteration's median or moves to a neighbour, so
// the second-digit histograms of those three bins are taken speculatively (SEL_PRED).  After the all-reduce k_trial_post scans
// the summed histograms, leaves the selection state of a 22-bit prefix behind (or says that the prediction missed / that the
// selected bin holds more candidates than the gather table takes: the next median then runs the plain three-collective
// selection), and forwards the trial's result block to the host's mailbox.
//
// Trial buffer (doubles):  [0] robust chi2  [1] sum x(lambda x + b), points  [2] sum x^2, points  [3] failure flag
//                          [4] iteration-start robust chi2 (first trial of an iteration, else 0)  [5..7] 0
//                          [8 ..) coarse | fine(pred) | fine(pred - 1) | fine(pred + 1)   (SEL_BINS counters each)
#pragma once
#include "ba_kernels.h"
#include "ba_select.h"

namespace mcp {

constexpr int SEL_PRED = 3;                               // second-digit histograms taken speculatively: pred, pred - 1, pred + 1
constexpr int TRIAL_HDR = 8;
constexpr int TRIAL_HIST = (1 + SEL_PRED)*SEL_BINS;
constexpr int TRIAL_LEN = TRIAL_HDR + TRIAL_HIST;         // doubles per trial buffer = payload of the trial's all-reduce
constexpr int MAIL_PRED_OK = 29, MAIL_OVERFLOW = 30;      // mailbox entries next to the result block (ticket at MAIL_TICKET = 31)

__host__ __device__ inline int sel_coarse_bin(double v) {
  union { double d; unsigned long long u; } c; c.d = v < 0 ? -v : v;
  return (int)(c.u >> sel_shift(0));
}

// digits 0 and (for the predicted first digits) 1 of |x| in one sweep; hist zeroed beforehand
static __global__ void __launch_bounds__(SEL_BLOCK)
k_select_hist2(int n, const double* __restrict__ x, int pred_bin, double* __restrict__ hist) {
  __shared__ unsigned int lh[TRIAL_HIST];
  for (int i = threadIdx.x; i < TRIAL_HIST; i += SEL_BLOCK) lh[i] = 0;
  __syncthreads();
  const int sh0 = sel_shift(0), sh1 = sel_shift(1);
  for (size_t i = blockIdx.x*(size_t)SEL_BLOCK + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x*SEL_BLOCK) {
    const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
    const int c = (int)(key >> sh0);
    atomicAdd(&lh[c], 1u);
    const int d = c - pred_bin;
    if (pred_bin >= 0 && d >= -1 && d <= 1) {
      const int slot = (d == 0) ? 1 : (d < 0 ? 2 : 3);
      atomicAdd(&lh[slot*SEL_BINS + ((unsigned int)(key >> sh1) & (SEL_BINS - 1))], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TRIAL_HIST; i += SEL_BLOCK) if (lh[i]) unsafeAtomicAdd(hist + i, (double)lh[i]);
}

// one workgroup, after the trial's all-reduce.  T: the summed trial buffer.  Leaves state[0], state[1] (and the second-digit
// histogram of the selected first digit in the pass-1 slot T + TRIAL_HDR + SEL_BINS) as two k_select_pass calls would, so that
// k_select_gather_slot continues from there.  Composes the result block (layout of mcp_ba::h_res) in `out_dev` (may be null) and
// forwards it to `mail` (may be null) with the ticket, as k_final_sums does on a single rank.
static __global__ void __launch_bounds__(SEL_BLOCK)
k_trial_post(double* __restrict__ T, SelState* __restrict__ state, unsigned long long k0, int pred_bin, double cap, int do_select,
             const double* __restrict__ pose_parts /* 2 */, int pose_off, const double* __restrict__ start_block /* 5 or null */,
             int start_rides, double* __restrict__ out_dev, double* mail, int mail_count, unsigned long long ticket) {
  __shared__ unsigned long long sc[SEL_BLOCK + 2];
  __shared__ double B[32];
  double pred_ok = 0.0, ovf = 0.0;
  if (do_select) {
    double* coarse = T + TRIAL_HDR;
    int bin0; unsigned long long kin0;
    sel_find_bin(coarse, SEL_BINS, k0, bin0, kin0, sc);
    const int d = bin0 - pred_bin;
    const bool hit = pred_bin >= 0 && d >= -1 && d <= 1;
    if (threadIdx.x == 0) {
      SelState s0; s0.prefix = 0; s0.k = k0; state[0] = s0;
      SelState s1; s1.prefix = (unsigned long long)bin0 << sel_shift(0); s1.k = kin0; state[1] = s1;
    }
    if (hit) {
      double* fine = coarse + SEL_BINS;
      if (d != 0) {
        const double* src = coarse + (size_t)(d < 0 ? 2 : 3)*SEL_BINS;
        for (int i = threadIdx.x; i < SEL_BINS; i += SEL_BLOCK) fine[i] = src[i];
        __syncthreads();
      }
      int bin1; unsigned long long kin1;
      sel_find_bin(fine, SEL_BINS, kin0, bin1, kin1, sc);
      pred_ok = 1.0;
      ovf = (fine[bin1] > cap) ? 1.0 : 0.0;
    }
  }
  for (int i = threadIdx.x; i < 32; i += SEL_BLOCK) B[i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    B[0] = T[0]; B[1] = T[1]; B[2] = T[2]; B[3] = T[3];
    B[pose_off] = pose_parts[0]; B[pose_off + 1] = pose_parts[1];
    if (start_block) { for (int i = 0; i < 5; ++i) B[24 + i] = start_block[i]; if (start_rides) B[24] = T[4]; }
    B[MAIL_PRED_OK] = pred_ok; B[MAIL_OVERFLOW] = ovf;
  }
  __syncthreads();
  if (out_dev) for (int i = threadIdx.x; i < MAIL_TICKET; i += SEL_BLOCK) if (i < 8 || i >= 24) out_dev[i] = B[i];
  if (mail) {
    for (int i = threadIdx.x; i < mail_count; i += SEL_BLOCK) mail[i] = B[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((unsigned long long*)(mail + MAIL_TICKET), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace mcp
