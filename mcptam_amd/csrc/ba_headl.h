// ba_headl.h -- the head of an LM iteration of a LARGE map in ONE launch (round 6), gfx950.
//
// What the head computes (RobustKernelData::RecomputeNow, /root/reference/src/ChainBundle.cc:810-833: the [size/2] order statistic
// of |chi2|, Huber sigma^2 with its limits; then activeRobustChi2, :871-897) was six dependent launches at the metric size -- two
// digit histograms, the candidate gather, the one-workgroup finish (ba_select.h), the robust sum per block and its final sum
// (ba_kernels.h) -- 49 us of device time of which the bytes are ~1 us: every launch is a drain, a dispatch and a cold first load.
// Here the same steps run in one launch of HL_GRID co-resident workgroups that meet at four counter barriers:
//
//   A  digit 0 (11 bits) of every |chi2| -> LDS histogram -> integer atomics into the launch's global histogram       | barrier
//   B  every workgroup finds the bin of rank k in the SAME global histogram; digit 1 of the matching values, as A      | barrier
//   C  every workgroup finds the second bin; the values sharing the 22-bit prefix are gathered (atomic slot counter)   | barrier
//   D  workgroup 0: remaining digits of the candidates in LDS (lds_radix_select), the sigma block                      | barrier
//   E  robust sum: workgroup w takes blocks w, w + HL_GRID, ... of EVAL_BLOCK values, one partial per block exactly as
//      k_robust_sum writes it; the workgroup that finishes LAST adds the partials in k_final_sums' fixed order.
//
// The median is exact whichever way it is found, the sigma block is k_select_small's expression, the partials and their sum are
// the separate kernels' operation for operation: the result block is bit-identical to the six launches' (knob MCP_BA_HEAD_LARGE in
// the scheduling test).  MEASURED (round 6, metric map): 39 us alone on the device, but no faster in the solver (1611 / 1603 vs
// 1606 / 1611 it/s): its fat workgroups wait for room beside the trials evaluated ahead on the second stream -- DEFAULT OFF.  Cross-workgroup data goes through device-scope atomics (the per-XCD L2s are not coherent with each other);
// the scratch is left zeroed by the last workgroup.  A barrier that waits longer than the wall-clock bound of the one-launch
// factorisation (cp_deadline, 20 ms) raises the scratch's error word: every later wait ends at once, the robust chi2 is reported
// as NaN and the host fails the solve by name (nothing hangs, nothing is silently wrong).
#pragma once
#include "ba_kernels.h"
#include "ba_chol2.h"

namespace mcp {

#ifndef HL_GRID_N
#define HL_GRID_N 64
#endif
// Few, fat workgroups: a barrier is one same-address device-scope atomic per workgroup, and those serialise at ~20 ns each (256
// workgroups of 256 threads: 5.5 us per barrier, 61 us for the whole head -- slower than the six launches; measured, round 6).
constexpr int HL_GRID = HL_GRID_N;        // workgroups at most (fewer for a small array)
constexpr int HL_THREADS = 1024;
constexpr int HL_CAP = 16384;             // candidates kept after two digits (expected: tens); beyond: the finish walks the array itself
struct HeadLScratch {
  unsigned int sync, err, cand_n, ticket;
  unsigned int hist[2][SEL_BINS];
  double cand[HL_CAP];
};
constexpr unsigned long long HL_FAILED_BITS = 0x7ff80000deadbeefull;      // the robust chi2 of a head that gave up (a NaN no arithmetic yields)
static_assert(EVAL_BLOCK == 256 && HL_THREADS == 4*EVAL_BLOCK, "phase E writes k_robust_sum's partials, four blocks of values per round");

__device__ inline unsigned int hl_ld(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double hl_ldd(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void hl_std(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads of all workgroups; `epoch` counts the barriers passed (uniform).  Returns false once the error word is up.
__device__ inline bool hl_barrier(HeadLScratch* G, unsigned int& epoch, unsigned int nwg, int* flag /* LDS */) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every thread's device-scope stores / atomics have completed ...
  __syncthreads();                                            // ... before thread 0 says that this workgroup has arrived
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&G->sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int want = (epoch + 1)*nwg;
    unsigned int spins = 0; long long t0 = 0; bool ok = true;
    while (hl_ld(&G->sync) < want) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63) == 63) {
        if (hl_ld(&G->err)) { ok = false; break; }
        if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { __hip_atomic_store(&G->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
      }
    }
    flag[0] = ok ? 1 : 0;
  }
  ++epoch;
  __syncthreads();
  return flag[0] != 0;
}

// the launch's global histogram of one digit into LDS, then the bin of rank k (uniform; the same on every workgroup)
__device__ inline void hl_find_bin(const unsigned int* gh, unsigned int* lh, unsigned long long k, int& bin, unsigned long long& k_in, unsigned long long* sc) {
  for (int i = threadIdx.x; i < SEL_BINS; i += HL_THREADS) lh[i] = hl_ld(gh + i);
  __syncthreads();
  unsigned int in_bin;
  lds_find_bin<HL_THREADS>(lh, k, bin, k_in, in_bin, sc);
}

#ifdef MCP_HL_PROF
__device__ long long g_hl_prof[16];
__device__ long long g_hl_wg[256][4];
#define HL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_hl_prof[i] = wall_clock64(); } while (0)
#else
#define HL_STAMP(i) do {} while (0)
#endif
__global__ void __launch_bounds__(HL_THREADS)
k_head_large(int n, const double* __restrict__ x, unsigned long long k0, double n_total, double min_sigma_sq,
             double* __restrict__ sig, double* __restrict__ sig_copy, double* __restrict__ sig_copy2, double* __restrict__ med_out,
             double* __restrict__ part, double* __restrict__ out, int off, HeadLScratch* __restrict__ G) {
  __shared__ unsigned int lh[SEL_BINS];
  __shared__ unsigned long long sc[HL_THREADS/64 + 3], s_st[2];
  __shared__ double red[HL_THREADS/64];
  static_assert(HL_THREADS/64 >= 4, "block_sum<256> reads four words");
  __shared__ int flag[2];
  const int t = threadIdx.x;
  const unsigned int nwg = gridDim.x, wg = blockIdx.x;
  unsigned int epoch = 0;
  bool alive = true;
  HL_STAMP(0);
  // ---- A: digit 0
  for (int i = t; i < SEL_BINS; i += HL_THREADS) lh[i] = 0u;
  __syncthreads();
  for (size_t i = wg*(size_t)HL_THREADS + t; i < (size_t)n; i += (size_t)nwg*HL_THREADS) {
    const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
    atomicAdd(&lh[(unsigned int)(key >> sel_shift(0))], 1u);
  }
  __syncthreads();
#ifdef MCP_HL_PROF
  if (t == 0) { g_hl_wg[wg][0] = wall_clock64(); }
#endif
  for (int i = t; i < SEL_BINS; i += HL_THREADS) if (lh[i]) __hip_atomic_fetch_add(&G->hist[0][i], lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef MCP_HL_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
  if (t == 0) { g_hl_wg[wg][1] = wall_clock64(); }
#endif
  HL_STAMP(1);
  alive = hl_barrier(G, epoch, nwg, flag);
  HL_STAMP(2);
#ifdef MCP_HL_PROF
  if (t == 0) { g_hl_wg[wg][2] = wall_clock64(); }
#endif
  // ---- B: digit 1 of the values in the bin of rank k
  unsigned long long prefix = 0, k = k0;
  if (alive) {
    int bin; unsigned long long kin;
    hl_find_bin(G->hist[0], lh, k, bin, kin, sc);
    prefix = (unsigned long long)bin << sel_shift(0); k = kin;
    __syncthreads();
    for (int i = t; i < SEL_BINS; i += HL_THREADS) lh[i] = 0u;
    __syncthreads();
    const unsigned long long himask = ~0ull << sel_shift(0);
    const unsigned int dmask = (1u << sel_nbits(1)) - 1u;
    for (size_t i = wg*(size_t)HL_THREADS + t; i < (size_t)n; i += (size_t)nwg*HL_THREADS) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
      if ((key & himask) == prefix) atomicAdd(&lh[(unsigned int)(key >> sel_shift(1)) & dmask], 1u);
    }
    __syncthreads();
    for (int i = t; i < SEL_BINS; i += HL_THREADS) if (lh[i]) __hip_atomic_fetch_add(&G->hist[1][i], lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HL_STAMP(3);
    alive = hl_barrier(G, epoch, nwg, flag);
    HL_STAMP(4);
  }
  // ---- C: gather the values that share the 22-bit prefix
  if (alive) {
    int bin; unsigned long long kin;
    hl_find_bin(G->hist[1], lh, k, bin, kin, sc);
    prefix |= (unsigned long long)bin << sel_shift(1); k = kin;
    const unsigned long long himask = ~0ull << sel_shift(1);
    for (size_t i = wg*(size_t)HL_THREADS + t; i < (size_t)n; i += (size_t)nwg*HL_THREADS) {
      const double a = fabs(x[i]);
      if (((unsigned long long)__double_as_longlong(a) & himask) == prefix) {
        const unsigned int idx = __hip_atomic_fetch_add(&G->cand_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (idx < (unsigned int)HL_CAP) hl_std(&G->cand[idx], a);
      }
    }
    HL_STAMP(5);
    alive = hl_barrier(G, epoch, nwg, flag);
    HL_STAMP(6);
  }
  // ---- D: workgroup 0 resolves the remaining digits and writes the sigma block (k_select_small's expressions)
  if (alive && wg == 0) {
    const unsigned int c = hl_ld(&G->cand_n);
    const bool overflow = c > (unsigned int)HL_CAP;
    const int m = overflow ? n : (int)c;
    const double* cand = G->cand;
    const unsigned long long hm = ~0ull << sel_shift(1);
    const unsigned long long sel = lds_radix_select<HL_THREADS>(m, k, 2, prefix, [&](int i, unsigned long long& key) {
      if (overflow) { key = (unsigned long long)__double_as_longlong(fabs(x[i])); return (key & hm) == prefix; }      // (the filter lds_radix_select applies anyway; kept explicit)
      key = (unsigned long long)__double_as_longlong(hl_ldd(cand + i));
      return true; }, lh, sc, s_st);
    if (t == 0) {
      const double md = __longlong_as_double((long long)sel);
      if (med_out) med_out[0] = md;
      double s = 1.4826*(1 + 5.0/mest_denom(n_total))*sqrt(md);
      s = 1.345*s;
      const double s2 = s*s;
      const double lim = (s2 < min_sigma_sq) ? min_sigma_sq : s2;
      const double sl = sqrt(lim);
      hl_std(sig + 0, s2); hl_std(sig + 1, lim); hl_std(sig + 2, sl); hl_std(sig + 3, md);
      if (sig_copy) { sig_copy[0] = s2; sig_copy[1] = lim; sig_copy[2] = sl; sig_copy[3] = md; }
      if (sig_copy2) { sig_copy2[0] = s2; sig_copy2[1] = lim; sig_copy2[2] = sl; sig_copy2[3] = md; }
    }
  }
  HL_STAMP(7);
  if (alive) alive = hl_barrier(G, epoch, nwg, flag);
  HL_STAMP(8);
  // ---- E: the robust sum, k_robust_sum's partial per block of EVAL_BLOCK values: four blocks per round, one per quarter of the
  //      workgroup, each summed as block_sum<EVAL_BLOCK> sums it (lane tree per wavefront, then the four wavefronts in order)
  const int nbe = (n + EVAL_BLOCK - 1)/EVAL_BLOCK;
  if (alive) {
    const double s1 = hl_ldd(sig + 1), s2 = hl_ldd(sig + 2);
    const int qd = t >> 8, tt = t & 255;
    for (int b0 = 4*(int)wg; b0 < nbe; b0 += 4*(int)nwg) {
      const int b = b0 + qd, m = b*EVAL_BLOCK + tt;
      double v = 0.0;
      if (b < nbe && m < n) { double r0, r1; robustify(x[m], s1, s2, r0, r1); v = r0; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if ((t & 63) == 0) red[t >> 6] = v;
      __syncthreads();
      if (tt == 0 && b < nbe) { double tot = 0.0; for (int i = 0; i < EVAL_BLOCK/64; ++i) tot += red[4*qd + i]; hl_std(part + b, tot); }
      __syncthreads();
    }
  }
  HL_STAMP(9);
  // the workgroup that arrives last sums the partials (k_final_sums' order) and leaves the scratch clean for the next launch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) flag[1] = (__hip_atomic_fetch_add(&G->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) ? 1 : 0;
  __syncthreads();
  if (!flag[1]) return;
  const bool failed = hl_ld(&G->err) != 0u;
  {
    double v = 0.0;
    if (t < 256) for (int i = t; i < nbe; i += 256) v += hl_ldd(part + i);
    const double tot = block_sum<256>(v, red);        // (threads 256.. add nothing: k_final_sums' 256-thread sum)
    if (t == 0) out[off] = failed ? __longlong_as_double((long long)HL_FAILED_BITS) : tot;
  }
  for (int i = t; i < 2*SEL_BINS; i += HL_THREADS) (&G->hist[0][0])[i] = 0u;
  if (t == 0) { G->sync = 0u; G->err = 0u; G->cand_n = 0u; G->ticket = 0u; }
#ifdef MCP_HL_PROF
  if (t == 0) g_hl_prof[10] = wall_clock64();
#endif
}

}  // namespace mcp
