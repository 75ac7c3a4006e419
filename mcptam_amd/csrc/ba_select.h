// ba_select.h -- exact k-th order statistic of |x| over a device array (fp64), gfx950.
//
// The reference takes medians with a full std::sort and indexes element [size/2]
// (include/mcptam/MEstimator.h:109-124,194-204; src/ChainBundle.cc:1433-1434).  On the
// device the same element is found without sorting: a most-significant-digit radix select
// over the IEEE-754 bit pattern of |x| (monotone for non-negative doubles), 11 bits per
// pass, 6 passes.  Histograms are integer-valued doubles so that a multi-GPU run can sum
// them with the same all-reduce hook as everything else (exact below 2^53).
#pragma once
#include <hip/hip_runtime.h>

namespace mcp {

constexpr int SEL_BITS = 11;
constexpr int SEL_BINS = 1 << SEL_BITS;      // 2048
constexpr int SEL_PASSES = 6;                // 5 x 11 + 9 bits
constexpr int SEL_BLOCK = 256;

struct SelState { unsigned long long prefix; unsigned long long k; };

// Denominator of the small-sample factor 1 + 5/(2n - 6) of Huber/Tukey::FindSigmaSquared (include/mcptam/MEstimator.h:121,201).
// The reference evaluates `vErrorSquared.size()*2 - 6` in size_t, so it wraps for n = 1, 2 (factor ~ 1) and is 0 for n = 3
// (sigma = inf); reproduced here in 64-bit unsigned arithmetic.
__host__ __device__ inline double mest_denom(double n) {
  const unsigned long long u = (unsigned long long)n*2ull - 6ull;
  return (double)u;
}

__host__ __device__ inline int sel_shift(int pass) { return pass < 5 ? 64 - SEL_BITS*(pass + 1) : 0; }
__host__ __device__ inline int sel_nbits(int pass) { return pass < 5 ? SEL_BITS : 9; }

// All NT threads of the block (thread order = bin order): `loc` = the count this thread owns.  Finds the thread whose span
// holds rank k -- exclusive prefix <= k < inclusive prefix -- by a wavefront scan (lane shuffles) + one hop through LDS; a k
// beyond the total is clamped to the last thread.  t_out / excl_out are uniform on return.  lds: NT/64 + 2 words.
template <int NT>
__device__ inline void block_find_rank(unsigned long long loc, unsigned long long k, int& t_out, unsigned long long& excl_out,
                                       unsigned long long* lds) {
  constexpr int NW = NT/64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned long long v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) lds[wave] = inc;
  if (threadIdx.x == 0) lds[NW] = ~0ull;
  __syncthreads();
  unsigned long long base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const unsigned long long v = lds[w]; if (w < wave) base += v; total += v; }
  const unsigned long long incl = base + inc, excl = incl - loc;
  if ((excl <= k && k < incl) || (k >= total && threadIdx.x == NT - 1)) { lds[NW] = (unsigned long long)threadIdx.x; lds[NW + 1] = excl; }
  __syncthreads();
  t_out = (int)lds[NW]; excl_out = lds[NW + 1];
  __syncthreads();
}

// all threads of the block: locate the bin of `hist` (nbins doubles) holding rank k; returns via refs
__device__ inline void sel_find_bin(const double* __restrict__ hist, int nbins, unsigned long long k,
                                    int& bin_out, unsigned long long& k_in, unsigned long long* lds /*SEL_BLOCK+2*/) {
  const int per = (nbins + SEL_BLOCK - 1)/SEL_BLOCK;
  const int b0 = threadIdx.x*per;
  unsigned long long loc = 0;
  for (int i = 0; i < per; ++i) if (b0 + i < nbins) loc += (unsigned long long)hist[b0 + i];
  int t; unsigned long long acc;
  block_find_rank<SEL_BLOCK>(loc, k, t, acc, lds);
  if ((int)threadIdx.x == t) {        // the owner walks its own few bins
    int b = t*per; const int be = min(nbins, b + per);
    for (; b < be; ++b) { const unsigned long long c = (unsigned long long)hist[b]; if (acc + c > k) break; acc += c; }
    if (b >= be) b = be - 1;
    lds[0] = (unsigned long long)b; lds[1] = k - acc;
  }
  __syncthreads();
  bin_out = (int)lds[0];
  k_in = lds[1];
  __syncthreads();
}

// NT threads (a divisor of SEL_BINS), one digit histogram of SEL_BINS counters in LDS: the bin holding rank k, the rank inside it and the
// bin's count (uniform on return).  sc: NT/64 + 3 words.
template <int NT>
__device__ inline void lds_find_bin(const unsigned int* hist, unsigned long long k, int& bin, unsigned long long& k_in, unsigned int& in_bin,
                                    unsigned long long* sc) {
  constexpr int BPT = SEL_BINS/NT;
  static_assert(BPT*NT == SEL_BINS && BPT >= 1, "threads must divide the bins");
  const int t = threadIdx.x;
  unsigned int h[BPT]; unsigned long long loc = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) { h[q] = hist[BPT*t + q]; loc += h[q]; }
  int tt; unsigned long long acc;
  block_find_rank<NT>(loc, k, tt, acc, sc);
  if (t == tt) {
    int b = 0;
#pragma unroll
    for (int q = 0; q < BPT - 1; ++q) if (b == q && !(acc + h[q] > k)) { acc += h[q]; b = q + 1; }
    unsigned int inb = h[0];
#pragma unroll
    for (int q = 1; q < BPT; ++q) if (b == q) inb = h[q];
    sc[0] = (unsigned long long)(BPT*t + b); sc[1] = k - acc; sc[2] = inb;
  }
  __syncthreads();
  bin = (int)sc[0]; k_in = sc[1]; in_bin = (unsigned int)sc[2];
  __syncthreads();
}
__device__ inline void lds_find_bin_1024(const unsigned int* hist, unsigned long long k, int& bin, unsigned long long& k_in, unsigned int& in_bin,
                                         unsigned long long* sc) { lds_find_bin<1024>(hist, k, bin, k_in, in_bin, sc); }

// One workgroup of NT threads: the rank-k element (bit pattern) among the keys keyfn(i, key) yields for i in [0, m), continuing
// a most-significant-digit radix select at digit `pass0` with the bits above it fixed to `prefix0`.  Digit histograms live in
// LDS; the bin search is a parallel scan; as soon as the selected bin holds a single key, that key is the answer and its
// remaining bits are read off directly.  hist: SEL_BINS counters, sc: NT/64 + 3 words, st: 2 words (all LDS).  Uniform result.
template <int NT, class KeyFn>
__device__ inline unsigned long long lds_radix_select(int m, unsigned long long k, int pass0, unsigned long long prefix0, KeyFn keyfn,
                                                           unsigned int* hist, unsigned long long* sc, unsigned long long* st) {
  const int t = threadIdx.x;
  if (t == 0) { st[0] = prefix0; st[1] = k; }
  __syncthreads();
  for (int pass = pass0; pass < SEL_PASSES; ++pass) {
    const int sh = sel_shift(pass);
    const unsigned int dmask = (1u << sel_nbits(pass)) - 1u;
    const unsigned long long himask = (pass == 0) ? 0ull : (~0ull << sel_shift(pass - 1));
    for (int b = t; b < SEL_BINS; b += NT) hist[b] = 0u;
    __syncthreads();
    const unsigned long long prefix = st[0];
    for (int i = t; i < m; i += NT) {
      unsigned long long key;
      if (!keyfn(i, key)) continue;
      if ((key & himask) == prefix) atomicAdd(&hist[(unsigned int)(key >> sh) & dmask], 1u);
    }
    __syncthreads();
    int bin; unsigned long long kin; unsigned int in_bin;
    lds_find_bin<NT>(hist, st[1], bin, kin, in_bin, sc);
    const unsigned long long np_ = prefix | ((unsigned long long)bin << sh);
    if (in_bin == 1u && sh > 0) {
      const unsigned long long hm2 = ~0ull << sh;
      for (int i = t; i < m; i += NT) {
        unsigned long long key;
        if (!keyfn(i, key)) continue;
        if ((key & hm2) == np_) st[0] = key;
      }
      __syncthreads();
      break;
    }
    if (t == 0) { st[0] = np_; st[1] = kin; }
    __syncthreads();
  }
  const unsigned long long r = st[0];
  __syncthreads();
  return r;
}
template <class KeyFn>
__device__ inline unsigned long long lds_radix_select_1024(int m, unsigned long long k, int pass0, unsigned long long prefix0, KeyFn keyfn,
                                                           unsigned int* hist, unsigned long long* sc, unsigned long long* st) {
  return lds_radix_select<1024>(m, k, pass0, prefix0, keyfn, hist, sc, st);
}

// ---- register-held keys, one workgroup: the rank-k key by VOTES, without shared counters -------------------------------
// NT threads hold up to PPT keys each (bit patterns of non-negative doubles; ok[] marks the live ones).  The search keeps a
// bracket [lo, lo + 16 * 2^s) of KEY space (the bit pattern is monotone in the value, so any split of key space is a valid split
// of the values): a key's bin is (key - lo) >> s, the 16 bins are counted by ballots (bin p's count is a scalar; lane p keeps
// it), summed over the wavefronts through a small table in LDS and scanned by every wavefront itself -- one barrier per step,
// no atomics.  The bin holding rank k becomes the next bracket (s - 4).  The first bracket is the caller's guess: s = 52 makes
// the bins binades, s = 49 eighths of a binade around the last median.  As soon as the bracket holds <= SELV_MAXC keys they are
// compacted into LDS and ranked (count of smaller keys; the wavefronts share the comparisons).
// Returns false (out untouched) when rank k lies outside the first bracket: the caller widens it or falls back to histogram passes.
// Exact for any input: equal keys share a bracket down to s = 0, where a bin IS a key.
// has[q]: wavefront-uniform "some lane holds a live key in slot q" (slots no lane of the wavefront uses cost nothing).
// vt: [2][NT/64][SELV_NB + 1] words, cand: SELV_MAXC keys, rk: SELV_MAXC words (LDS; NT >= SELV_MAXC, NT/64 >= SELV_MAXC/8).  Uniform result; contains barriers (all threads call it).
constexpr int SELV_NB = 16, SELV_MAXC = 64;
#ifdef MCP_PRR_PROF
__device__ unsigned long long g_selv_prof[8];      // last call: entry, votes of step 0 done, its barrier passed, its scan done, loop left, gather barrier passed, ranked; [7] = steps
#define SELV_STAMP(i) do { if (threadIdx.x == 0) g_selv_prof[i] = clock64(); } while (0)
#else
#define SELV_STAMP(i) do {} while (0)
#endif
template <int NT, int PPT>
__device__ inline bool regs_vote_select(const unsigned long long (&key)[PPT], const bool (&ok)[PPT], const bool (&has)[PPT], unsigned int kk,
                                        unsigned long long lo, int s,
                                        unsigned int (*vt)[NT/64][SELV_NB + 1], unsigned long long* cand, unsigned int* rk, unsigned long long& out) {
  constexpr int NW = NT/64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int buf = 0, p = 0;
  unsigned int cnt = 0u;
  bool hs[PPT];                                     // (made scalar for the compiler: the counts below stay in scalar registers)
#pragma unroll
  for (int q = 0; q < PPT; ++q) hs[q] = __builtin_amdgcn_readfirstlane((int)has[q]) != 0;
  SELV_STAMP(0);
  for (int step = 0; ; ++step) {
    // bin of a key: 0..15 inside the bracket, 16 below it (first step only: later brackets hold the rank by construction), 99 not counted
    const unsigned long long width = (unsigned long long)SELV_NB << s;         // (s <= 52: no overflow)
    int d[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const unsigned long long rel = key[q] - lo;
      d[q] = !ok[q] ? 99 : (key[q] < lo ? SELV_NB : (rel < width ? (int)(rel >> s) : 99));
    }
    // Counting without a round trip through the scalar unit per bin: every lane adds 1 to its key's bin in four dwords of packed
    // 8-bit counters (a wavefront holds at most 64 PPT < 256 keys), the dwords are summed over the wavefront by DPP (row scans, then
    // row_bcast:15 / :31 -- the total lands in lane 63), and lane p extracts bin p's byte.
    static_assert(64*PPT < 256, "8-bit packed counters");
    unsigned int pk[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int q = 0; q < PPT; ++q) if (hs[q]) {
      const unsigned int one = d[q] < SELV_NB ? 1u << ((d[q] & 3) << 3) : 0u;
      const int j = d[q] >> 2;
#pragma unroll
      for (int x = 0; x < 4; ++x) pk[x] += (j == x) ? one : 0u;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      int v = (int)pk[x];
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1 and 3
      v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2 and 3
      pk[x] = (unsigned int)__builtin_amdgcn_readlane(v, 63);
    }
    const unsigned int word = (lane >> 2) == 0 ? pk[0] : (lane >> 2) == 1 ? pk[1] : (lane >> 2) == 2 ? pk[2] : pk[3];
    int mine = (int)((word >> ((lane & 3) << 3)) & 0xffu);
    if (step == 0) {                                 // the keys below the first bracket: lane SELV_NB
      int c = 0;
#pragma unroll
      for (int q = 0; q < PPT; ++q) if (hs[q]) c += __popcll(__ballot(d[q] == SELV_NB));
      if (lane == SELV_NB) mine = c;
    }
    if (step == 0) SELV_STAMP(1);
    if (lane <= SELV_NB) vt[buf][wave][lane] = (unsigned int)mine;
    __syncthreads();
    if (step == 0) SELV_STAMP(2);
    unsigned int sum = 0u;
    if (lane <= SELV_NB) {
      unsigned int pw[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) pw[w] = vt[buf][w][lane];
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += pw[w];
    }
    const unsigned int below = (step == 0) ? (unsigned int)__builtin_amdgcn_readlane((int)sum, SELV_NB) : 0u;
    // inclusive scan over the 16 bins = one DPP row (row_shr with zero fill), no LDS crossbar
    int inc = lane < SELV_NB ? (int)sum : 0;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
    const unsigned int incl = below + (unsigned int)inc, excl = incl - sum;
    const unsigned long long hit = __ballot(lane < SELV_NB && excl <= kk && kk < incl);
    if (!hit) return false;                          // (only the first bracket can miss)
    p = __ffsll((long long)hit) - 1;
    cnt = (unsigned int)__builtin_amdgcn_readlane((int)sum, p);
    kk -= (unsigned int)__builtin_amdgcn_readlane((int)excl, p);
    lo += (unsigned long long)p << s;                // the bin [lo, lo + 2^s) is the new bracket
    if (step == 0) SELV_STAMP(3);
#ifdef MCP_PRR_PROF
    if (threadIdx.x == 0) g_selv_prof[7] = (unsigned long long)(step + 1) | ((unsigned long long)cnt << 32);
#endif
    if (cnt <= (unsigned int)SELV_MAXC) break;
    if (s == 0) { out = lo; return true; }           // more than SELV_MAXC keys equal in every bit
    const int ns = s >= 4 ? s - 4 : 0;               // 16 bins of 2^ns cover the 2^s keys of the bracket (s < 4: with room to spare)
    s = ns;
    buf ^= 1;
  }
  // the bracket is [lo, lo + 2^s_final) where s_final is the bin width of the LAST step: compact its keys (wavefront order, then key
  // slot, then lane) behind a padding of all-ones keys, rank them
  SELV_STAMP(4);
  const unsigned long long bw = 1ull << s;
  unsigned int off = 0u;
  {
    unsigned int pw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) pw[w] = vt[buf][w][p];
#pragma unroll
    for (int w = 0; w < NW; ++w) if (w < wave) off += pw[w];
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    if (!hs[q]) continue;
    const bool c = ok[q] && key[q] >= lo && key[q] - lo < bw;
    const unsigned long long m = __ballot(c);
    if (c) cand[off + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = key[q];
    off += (unsigned int)__popcll(m);
  }
  if (threadIdx.x >= cnt && threadIdx.x < ((cnt + 7u) & ~7u)) cand[threadIdx.x] = ~0ull;
  if (threadIdx.x < (unsigned int)SELV_MAXC) rk[threadIdx.x] = 0u;
  __syncthreads();
  SELV_STAMP(5);
  // lane j of every wavefront holds candidate j; wavefront w counts, of the candidates 8 w .. 8 w + 7, those below it (low half) and those
  // not above it (high half); the counts meet in LDS.  Rank k belongs to the key with  #below <= k < #not-above  (equal keys all qualify).
  const unsigned long long mykey = (unsigned int)lane < cnt ? cand[lane] : ~0ull;
  if ((unsigned int)(8*wave) < cnt) {
    unsigned long long ci[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ci[j] = cand[8*wave + j];
    unsigned int add = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) add += (ci[j] < mykey ? 1u : 0u) + (ci[j] <= mykey ? 0x10000u : 0u);
    if ((unsigned int)lane < cnt) atomicAdd(&rk[lane], add);
  }
  __syncthreads();
  const unsigned int rv = (unsigned int)lane < cnt ? rk[lane] : 0u;
  const unsigned long long hit = __ballot((unsigned int)lane < cnt && (rv & 0xffffu) <= kk && kk < (rv >> 16));
  const int L = __ffsll((long long)hit) - 1;
  out = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(mykey >> 32), L) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)mykey, L);
  SELV_STAMP(6);
  return true;
}

// pass kernel: derive state[pass] from state[pass-1] and hist[pass-1], then histogram digit `pass`
// of the elements that match the prefix.  hist: SEL_PASSES x SEL_BINS doubles, zeroed beforehand.
static __global__ void __launch_bounds__(SEL_BLOCK)
k_select_pass(int pass, int n, const double* __restrict__ x, double* __restrict__ hist, SelState* __restrict__ state,
              unsigned long long k0) {
  __shared__ unsigned int lh[SEL_BINS];
  __shared__ unsigned long long sc[SEL_BLOCK + 2];
  SelState st;
  if (pass == 0) { st.prefix = 0; st.k = k0; }
  else {
    const SelState prev = state[pass - 1];
    int bin; unsigned long long kin;
    sel_find_bin(hist + (size_t)(pass - 1)*SEL_BINS, 1 << sel_nbits(pass - 1), prev.k, bin, kin, sc);
    st.prefix = prev.prefix | ((unsigned long long)bin << sel_shift(pass - 1));
    st.k = kin;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) state[pass] = st;
  for (int i = threadIdx.x; i < SEL_BINS; i += SEL_BLOCK) lh[i] = 0;
  __syncthreads();
  const int sh = sel_shift(pass);
  const unsigned long long himask = (pass == 0) ? 0ull : (~0ull << sel_shift(pass - 1));
  const unsigned int dmask = (1u << sel_nbits(pass)) - 1u;
  for (size_t i = blockIdx.x*(size_t)SEL_BLOCK + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x*SEL_BLOCK) {
    const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x[i]));
    if ((key & himask) == st.prefix) atomicAdd(&lh[(unsigned int)(key >> sh) & dmask], 1u);
  }
  __syncthreads();
  double* gh = hist + (size_t)pass*SEL_BINS;
  for (int i = threadIdx.x; i < SEL_BINS; i += SEL_BLOCK) if (lh[i]) unsafeAtomicAdd(gh + i, (double)lh[i]);
}

// final: resolve the last digit; out[0] = the k-th smallest |x|
static __global__ void __launch_bounds__(SEL_BLOCK)
k_select_final(const double* __restrict__ hist, const SelState* __restrict__ state, double* __restrict__ out) {
  __shared__ unsigned long long sc[SEL_BLOCK + 2];
  const SelState prev = state[SEL_PASSES - 1];
  int bin; unsigned long long kin;
  sel_find_bin(hist + (size_t)(SEL_PASSES - 1)*SEL_BINS, 1 << sel_nbits(SEL_PASSES - 1), prev.k, bin, kin, sc);
  if (threadIdx.x == 0) {
    const unsigned long long key = prev.prefix | ((unsigned long long)bin << sel_shift(SEL_PASSES - 1));
    out[0] = __longlong_as_double((long long)key);
  }
}

// sigma block from the median (Huber::FindSigmaSquared + RobustKernelData::RecomputeNow limits,
// MEstimator.h:194-204, ChainBundle.cc:822-830):  sig[0] raw sigma^2, [1] limited, [2] sqrt(limited), [3] median
static __global__ void k_sigma_from_median(const double* __restrict__ med, double n_total, double min_sigma_sq,
                                    double* __restrict__ sig, double* __restrict__ sig_copy /* second destination or null */) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double m = med[0];
    double s = 1.4826*(1 + 5.0/mest_denom(n_total))*sqrt(m);
    s = 1.345*s;
    const double s2 = s*s;
    const double lim = (s2 < min_sigma_sq) ? min_sigma_sq : s2;
    sig[0] = s2; sig[1] = lim; sig[2] = sqrt(lim); sig[3] = m;
    if (sig_copy) { sig_copy[0] = s2; sig_copy[1] = lim; sig_copy[2] = sqrt(lim); sig_copy[3] = m; }
  }
}


// ---- short form for a single GPU: two histogram passes over the whole array, then the few elements that share the
// selected 22-bit prefix are gathered and the remaining 42 bits are resolved by one workgroup in LDS (4 launches in place
// of 8; the launches, not the bytes, are what a 400k-element selection costs).
constexpr int SEL_GATHER_CAP = 1 << 16;

// derive state[2] from pass 1 (as k_select_pass(pass = 2) would) and gather |x| of the matching elements
static __global__ void __launch_bounds__(SEL_BLOCK)
k_select_gather(int n, const double* __restrict__ x, const double* __restrict__ hist, SelState* __restrict__ state,
                unsigned int* __restrict__ cnt, double* __restrict__ vals) {
  __shared__ unsigned long long sc[SEL_BLOCK + 2];
  const SelState prev = state[1];
  int bin; unsigned long long kin;
  sel_find_bin(hist + (size_t)SEL_BINS, 1 << sel_nbits(1), prev.k, bin, kin, sc);
  SelState st;
  st.prefix = prev.prefix | ((unsigned long long)bin << sel_shift(1));
  st.k = kin;
  if (blockIdx.x == 0 && threadIdx.x == 0) state[2] = st;
  const unsigned long long himask = ~0ull << sel_shift(1);
  for (size_t i = blockIdx.x*(size_t)SEL_BLOCK + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x*SEL_BLOCK) {
    const double a = fabs(x[i]);
    const unsigned long long key = (unsigned long long)__double_as_longlong(a);
    if ((key & himask) == st.prefix) {
      const unsigned int idx = atomicAdd(cnt, 1u);
      if (idx < (unsigned int)SEL_GATHER_CAP) vals[idx] = a;
    }
  }
}

// ---- multi-rank short form: two all-reduced histogram passes, then every rank writes the elements that share the selected
// 22-bit prefix into its own slot of a (ranks x cap) table; summing the zero-filled tables over the ranks gathers them, and
// every rank finishes locally with k_select_small (3 collectives in place of 6).  The all-reduced histogram tells every rank
// the same global candidate count; beyond `cap` the table is abandoned and the remaining histogram passes run instead.
static __global__ void __launch_bounds__(SEL_BLOCK)
k_select_gather_slot(int n, const double* __restrict__ x, const double* __restrict__ hist, SelState* __restrict__ state,
                     unsigned int* __restrict__ cnt, double* __restrict__ slot_vals, int cap, double* __restrict__ flag /* rank 0; else null */) {
  __shared__ unsigned long long sc[SEL_BLOCK + 2];
  const SelState prev = state[1];
  int bin; unsigned long long kin;
  sel_find_bin(hist + (size_t)SEL_BINS, 1 << sel_nbits(1), prev.k, bin, kin, sc);
  SelState st;
  st.prefix = prev.prefix | ((unsigned long long)bin << sel_shift(1));
  st.k = kin;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state[2] = st;
    if (flag) flag[0] = (hist[SEL_BINS + bin] > (double)cap) ? 1.0 : 0.0;
  }
  const unsigned long long himask = ~0ull << sel_shift(1);
  for (size_t i = blockIdx.x*(size_t)SEL_BLOCK + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x*SEL_BLOCK) {
    const double a = fabs(x[i]);
    const unsigned long long key = (unsigned long long)__double_as_longlong(a);
    if ((key & himask) == st.prefix) {
      const unsigned int idx = atomicAdd(cnt, 1u);
      if (idx < (unsigned int)cap) slot_vals[idx] = a;
    }
  }
}
static __global__ void k_select_publish(const unsigned int* __restrict__ cnt, int cap, double* __restrict__ count_slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) count_slot[0] = (double)min(cnt[0], (unsigned int)cap);
}

// one workgroup: passes 2..5 over the gathered candidates (or, if they overflowed the buffer -- tens of thousands of
// values equal in their top 22 bits -- over the original array with the prefix filter), then the sigma block
static __global__ void __launch_bounds__(1024)
k_select_small(int n, const double* __restrict__ x, const unsigned int* cnt, const double* __restrict__ vals,
               const SelState* __restrict__ state, double n_total, double min_sigma_sq, double* __restrict__ med_out,
               double* __restrict__ sig, double* __restrict__ sig_copy,
               const double* __restrict__ slot_counts = nullptr, int nslot = 0, int slot_cap = 0,
               double* clear = nullptr, int nclear = 0 /* the histograms + gather counter of this selection (cnt lies inside): left zero for the
               next one, which then needs no fill launch in front of it */) {
  __shared__ unsigned int hist[SEL_BINS];
  __shared__ unsigned long long sc[1024/64 + 3];
  __shared__ unsigned long long s_st[2];
  const int t = threadIdx.x;
  // slot mode (multi-rank, k_select_gather_slot): vals is an nslot x slot_cap table, slot r holds slot_counts[r] candidates
  const bool slots = slot_counts != nullptr;
  const unsigned int c = slots ? 0u : cnt[0];
  const bool overflow = !slots && c > (unsigned int)SEL_GATHER_CAP;
  const double* src = overflow ? x : vals;
  const int m = slots ? nslot*slot_cap : (overflow ? n : (int)c);
  const unsigned long long sel = lds_radix_select_1024(m, state[2].k, 2, state[2].prefix, [&](int i, unsigned long long& key) {
    if (slots && (double)(i % slot_cap) >= slot_counts[i/slot_cap]) return false;
    key = (unsigned long long)__double_as_longlong(fabs(src[i]));
    return true; }, hist, sc, s_st);
  if (t == 0) {
    const double md = __longlong_as_double((long long)sel);
    med_out[0] = md;
    if (sig) {
      double s = 1.4826*(1 + 5.0/mest_denom(n_total))*sqrt(md);
      s = 1.345*s;
      const double s2 = s*s;
      const double lim = (s2 < min_sigma_sq) ? min_sigma_sq : s2;
      sig[0] = s2; sig[1] = lim; sig[2] = sqrt(lim); sig[3] = md;
      if (sig_copy) { sig_copy[0] = s2; sig_copy[1] = lim; sig_copy[2] = sqrt(lim); sig_copy[3] = md; }
    }
  }
  // (every thread has read cnt[0] long ago: the select above is full of barriers)
  for (int i = t; i < nclear; i += 1024) clear[i] = 0.0;
}

}  // namespace mcp
