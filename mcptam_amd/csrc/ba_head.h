// ba_head.h -- the head of the NEXT LM iteration riding on a trial's evaluation (one rank, maps beyond the small-bundle limit; round 6).
//
// g2o recomputes the Huber kernel's sigma^2 at the start of every outer iteration from the median of the |chi2| of the current
// state (RobustKernelData::RecomputeNow, src/ChainBundle.cc:810-833, 913-917 with MEstimator.h:194-204).  The current state of
// iteration k + 1 is the state of the trial iteration k accepts, whose chi2 values that trial's evaluation has just written -- so the
// median can be taken right there, before the host knows whether the trial will be accepted:
//
//   * k_head_hist, one sweep over the chi2 array: the coarse digit of the selection (ba_select.h: sign + ten exponent bits, a bin
//     spans a factor of four) in a window around the bin the CURRENT median lies in, and the next HEAD_FBITS bits of that bin and
//     of its two neighbours (the prediction of ba_trial.h, which the multi-rank path lets ride on the trial's all-reduce).  Integer
//     counters, LDS per workgroup, non-zero bins flushed with integer atomics: order-free, exact.  (Counting inside the evaluation
//     kernel itself was measured: its 1563 workgroups flushed 300 k device-scope atomics onto 1600 addresses -- the evaluation took
//     110 us instead of 70.)
//   * k_head_finish, one launch behind the trial's result block: every workgroup finds the median's first two digits from the
//     histograms, gathers the values sharing those 11 + HEAD_FBITS bits; the LAST workgroup to finish resolves the remaining digits in LDS,
//     writes the sigma block and the sigma part of the iteration-start block exactly as k_select_small does, leaves the counters
//     zero for the next use and tells the host (pinned status word + ticket) whether the prediction held.
//
// If it did not (the median moved by more than one coarse bin: a factor of >= 4 between two iterations) or that prefix is
// shared by more values than the table holds, the host takes the plain selection (mcp_ba::median_sigma) as before.  The numbers are
// the plain selection's: the same [n/2] order statistic, the same sigma arithmetic.
#pragma once
#include "ba_trial.h"

namespace mcp {

// Counter block of a head slot (u32): [0] values below the coarse window, [1 .. HEAD_WIN] the coarse bins pred - HEAD_WIN/2 ..
// pred + HEAD_WIN/2 - 1, [HEAD_WIN + 1] values above the window; then three fine histograms (the next HEAD_FBITS bits) of the coarse
// bins pred, pred - 1, pred + 1; then [0] the gather counter, [1] workgroups done.  (ba_trial.h's layout -- 2048 coarse bins + 3 x 2048
// fine ones -- is 32 KB of LDS and an LDS atomic per lane on one of two or three hot coarse bins; here: 1.7 KB, the coarse count
// aggregated per wavefront by ballots.)
constexpr int HEAD_WIN = 32;
constexpr int HEAD_FBITS = 7;
constexpr int HEAD_FBINS = 1 << HEAD_FBITS;
constexpr int HEAD_COARSE = HEAD_WIN + 2;
constexpr int HEAD_HIST = HEAD_COARSE + SEL_PRED*HEAD_FBINS;
constexpr int HEAD_CTL = 4;
constexpr int HEAD_GRID = 128;
constexpr int HEAD_THREADS = 256;
constexpr int HEAD_CAND = 4096;                           // candidates sharing the median's first 11 + HEAD_FBITS bits (typically one or two thousand): staged in LDS by the last workgroup
__host__ __device__ inline int head_fshift() { return sel_shift(0) - HEAD_FBITS; }

// the digit counters of |x| in one sweep, HEAD_GRID workgroups (hist zero before the first one starts)
__global__ void __launch_bounds__(HEAD_THREADS)
k_head_hist(int n, const double* __restrict__ x, const double* __restrict__ sigma_cur /* [3]: the current median */, unsigned int* __restrict__ hist) {
  __shared__ unsigned int lh[HEAD_HIST];
  for (int i = threadIdx.x; i < HEAD_HIST; i += HEAD_THREADS) lh[i] = 0u;
  __syncthreads();
  const int pred = sel_coarse_bin(sigma_cur[3]);
  const size_t stride = (size_t)gridDim.x*HEAD_THREADS;
  const size_t n_up = ((size_t)n + stride - 1)/stride*stride;          // (every lane takes part in the ballots of every round)
  for (size_t i = blockIdx.x*(size_t)HEAD_THREADS + threadIdx.x; i < n_up; i += stride) {
    int wb = -1;                     // this lane's coarse counter, -1 = no value
    if (i < (size_t)n) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
      const int d = (int)(key >> sel_shift(0)) - pred;
      wb = d < -HEAD_WIN/2 ? 0 : (d >= HEAD_WIN/2 ? HEAD_WIN + 1 : d + HEAD_WIN/2 + 1);
      if (d >= -1 && d <= 1) {
        const int slot = (d == 0) ? 0 : (d < 0 ? 1 : 2);
        atomicAdd(&lh[HEAD_COARSE + slot*HEAD_FBINS + ((unsigned int)(key >> head_fshift()) & (HEAD_FBINS - 1))], 1u);
      }
    }
    // coarse counters, aggregated per wavefront: nearly every lane of a wavefront has one of two or three values
    unsigned long long todo = __builtin_amdgcn_ballot_w64(wb >= 0);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      const int b = __builtin_amdgcn_readlane(wb, src);
      const unsigned long long same = __builtin_amdgcn_ballot_w64(wb == b);
      if ((int)(threadIdx.x & 63) == src) atomicAdd(&lh[b], (unsigned int)__builtin_popcountll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HEAD_HIST; i += HEAD_THREADS) { const unsigned int v = lh[i]; if (v) atomicAdd(hist + i, v); }
}

// status word of a head in the host's pinned block: [0] 1 = sigma block written, 2 = prediction missed / table overflow (nothing
// written: take the plain selection); [1] the ticket
__global__ void __launch_bounds__(HEAD_THREADS)
k_head_finish(int n, const double* __restrict__ x, unsigned int* __restrict__ hist /* HEAD_HIST + HEAD_CTL */, double* __restrict__ vals /* HEAD_CAND */,
              const double* __restrict__ sigma_cur, unsigned long long k, double n_total, double min_sigma_sq,
              double* __restrict__ sig_out, double* __restrict__ res_out /* 4 doubles, as k_select_small's sig_copy */, double* __restrict__ med_out,
              double* mail, unsigned long long ticket) {
  __shared__ unsigned int lh[SEL_BINS];
  __shared__ unsigned long long ck[HEAD_CAND];          // the last workgroup's candidates
  __shared__ unsigned long long sc[HEAD_THREADS/64 + 3];
  __shared__ unsigned long long st[2];
  __shared__ long long s_c[3];
  __shared__ unsigned int s_last, s_n, s_base;
  const int t = threadIdx.x;
  unsigned int* ctl = hist + HEAD_HIST;
  const int pred = sel_coarse_bin(sigma_cur[3]);
  // coarse window: which bin holds rank k?  (the counters are fetched by 34 lanes at once: thread 0 walking them in global memory
  // was 34 dependent round trips, 24 us of this kernel)
  if (t < HEAD_COARSE) lh[t] = hist[t];
  __syncthreads();
  if (t == 0) {
    unsigned long long acc = lh[0];
    long long j = -1, k0 = 0;
    if (k >= acc) {
      for (int b = 1; b <= HEAD_WIN; ++b) { const unsigned long long c = lh[b]; if (acc + c > k) { j = b; k0 = (long long)(k - acc); break; } acc += c; }
    }
    s_c[0] = j; s_c[1] = k0;
  }
  __syncthreads();
  const int d = (s_c[0] < 0) ? 99 : (int)s_c[0] - 1 - HEAD_WIN/2;
  bool ok = d >= -1 && d <= 1;
  const int bin0 = pred + d;
  unsigned long long prefix = 0, k1 = 0; unsigned int in1 = 0;
  if (ok) {
    const int slot = (d == 0) ? 0 : (d < 0 ? 1 : 2);
    const unsigned int* fh = hist + HEAD_COARSE + slot*HEAD_FBINS;
    static_assert(2*HEAD_THREADS >= HEAD_FBINS, "at most two fine bins per thread");
    const unsigned int h0 = (2*t < HEAD_FBINS) ? fh[2*t] : 0u, h1 = (2*t + 1 < HEAD_FBINS) ? fh[2*t + 1] : 0u;
    int tt; unsigned long long acc;
    block_find_rank<HEAD_THREADS>((unsigned long long)h0 + h1, (unsigned long long)s_c[1], tt, acc, sc);
    if (t == tt) {
      const unsigned long long kk = (unsigned long long)s_c[1];
      int b = 0; unsigned int inb = h0;
      if (!(acc + h0 > kk)) { acc += h0; b = 1; inb = h1; }
      st[0] = (unsigned long long)(2*t + b); st[1] = kk - acc; s_c[2] = (long long)inb;
    }
    __syncthreads();
    const unsigned long long bin9 = st[0]; k1 = st[1]; in1 = (unsigned int)s_c[2];
    __syncthreads();
    prefix = ((unsigned long long)bin0 << sel_shift(0)) | (bin9 << head_fshift());
    ok = in1 <= (unsigned int)HEAD_CAND;
  }
  if (t == 0) s_n = 0u;
  __syncthreads();
  if (ok) {
    // gather: a workgroup collects its matches in LDS and reserves its range of the table with ONE atomic (an atomic per match on the one
    // counter -- 1-2 k of them, device scope, each waiting for its index -- was 25 us of this kernel)
    const unsigned long long himask = ~0ull << head_fshift();
    for (size_t i = blockIdx.x*(size_t)HEAD_THREADS + t; i < (size_t)n; i += (size_t)gridDim.x*HEAD_THREADS) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
      if ((key & himask) == prefix) { const unsigned int j = atomicAdd(&s_n, 1u); if (j < (unsigned int)HEAD_CAND) ck[j] = key; }
    }
    __syncthreads();
    const unsigned int mine = min(s_n, (unsigned int)HEAD_CAND);
    if (t == 0 && mine) s_base = atomicAdd(&ctl[0], mine);
    __syncthreads();
    for (unsigned int j = t; j < mine; j += HEAD_THREADS) {
      const unsigned int idx = s_base + j;
      if (idx < (unsigned int)HEAD_CAND) __hip_atomic_store(reinterpret_cast<unsigned long long*>(vals) + idx, ck[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the last workgroup to get here finishes (the candidates were stored write-through; the counter is an agent-scope atomic)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) s_last = (atomicAdd(&ctl[1], 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  if (ok) {
    // the candidates share digit 0 and the upper HEAD_FBITS bits of digit 1: the selection resumes AT digit 1, rank k1 among them
    const int m = (int)in1;          // (every value with the prefix has been gathered: in1 of them)
    // ... staged in LDS first (16 independent write-through-coherent loads per thread in flight at once, then every pass at LDS speed)
    constexpr int PER = HEAD_CAND/HEAD_THREADS;
    unsigned long long kv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int i = t + j*HEAD_THREADS; kv[j] = (i < m) ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(vals) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull; }
#pragma unroll
    for (int j = 0; j < PER; ++j) ck[t + j*HEAD_THREADS] = kv[j];
    __syncthreads();
    const unsigned long long sel = lds_radix_select<HEAD_THREADS>(m, k1, 1, (unsigned long long)bin0 << sel_shift(0), [&](int i, unsigned long long& key) {
      key = ck[i];
      return true; }, lh, sc, st);
    if (t == 0) {
      const double md = __longlong_as_double((long long)sel);
      med_out[0] = md;
      double s = 1.4826*(1 + 5.0/mest_denom(n_total))*sqrt(md);       // Huber::FindSigmaSquared, MEstimator.h:194-204 (k_select_small's arithmetic)
      s = 1.345*s;
      const double s2 = s*s;
      const double lim = (s2 < min_sigma_sq) ? min_sigma_sq : s2;
      sig_out[0] = s2; sig_out[1] = lim; sig_out[2] = sqrt(lim); sig_out[3] = md;
      res_out[0] = s2; res_out[1] = lim; res_out[2] = sqrt(lim); res_out[3] = md;
    }
  }
  __syncthreads();
  for (int i = t; i < HEAD_HIST + HEAD_CTL; i += HEAD_THREADS) hist[i] = 0u;       // armed for the next trial that uses this slot
  if (t == 0 && mail) {
    mail[0] = ok ? 1.0 : 2.0;
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(mail + 1), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace mcp
