// ba_chol.h -- dense fp64 Cholesky solve of the reduced pose system on gfx950.
//
// Replaces g2o::LinearSolverCholmod (src/ChainBundle.cc:1156) for the (6P x 6P) system that
// remains after the points are eliminated.  Layout: S is row-major n x n (leading dimension
// n, lower triangle meaningful) and the right-hand side is stored directly behind it, i.e. it
// is row n of an (n+1) x n matrix.  Factoring that augmented matrix gives the forward
// substitution for free: after the last step row n holds y = L^-1 rhs.
//
// One launch per 32-wide block step (launch boundaries are ~1.5 us on MI355X, cheaper than
// any grid barrier).  Launch k:
//   every tile (ti >= tj >= k):  C -= L(ti,k-1) L(tj,k-1)^T      fp64 MFMA 16x16x4
//   tiles of block column k:     redundantly update + factor the diagonal tile in registers
//                                (one wavefront, v_readlane broadcasts), then X L_kk^T = C.
// A non-positive pivot (CHOLMOD failure in the reference == rejected LM trial) raises *fail.
#pragma once
#include <hip/hip_runtime.h>
#include "ba_pool.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <cstdlib>
#include <utility>
#include <vector>

namespace mcp {

constexpr int CH_NB = 32;
// The diagonal workgroup of launch k writes its result (L_kk^-T, all the back-substitution needs of the tile) to a side array
// of diagonal tiles (Dg, read by k_chol_back) instead of over tile (k,k): the other workgroups of block column k load tile
// (k,k) in the same launch, and nothing orders a store against those loads inside a launch.  Tile (k,k) of S therefore
// stays untouched (pre-factorisation values) for good.
#define CH_DIAG_PARAMS(cv) , cv double* __restrict__ Dg /* [systems][block columns][32 x 32] factored diagonal tiles */, size_t diag_stride
#define CH_DIAG_ARGS(plan) , plan.d_diag, plan.diag_stride
#define CH_DIAG_OFFSET(b) Dg += (b)*diag_stride
#define CH_BACK_DG , Dg, diag_stride
typedef double chol_d4 __attribute__((ext_vector_type(4)));

__device__ inline double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

constexpr int CH_STEP_THREADS = 256;    // 4 wavefronts load and multiply; wavefront 0 alone runs the panel factorisation

// 32x32 tile (rows r0.., cols c0..) -> 4 registers per thread (zero outside [nrows) x [ncols)); issuing the
// global loads of ALL tiles of a step before the first LDS write keeps them in flight together (one memory
// round trip per step instead of one per tile)
typedef double chol_d2 __attribute__((ext_vector_type(2)));
// CH_LOAD16: a thread fetches two neighbouring columns with one 16-byte load (rows 16 i + (t >> 4), columns 2 (t & 15) ..):
// half the memory requests per tile; needs an even leading dimension (n = 6 P always is; odd n -- test matrices -- takes the 8-byte form)
#ifndef CH_LOAD16
#define CH_LOAD16 0
#endif
#ifndef CH_STORE16       // trailing tiles written back as 16-byte column pairs
#define CH_STORE16 0
#endif
#ifndef CH_PSTORE16      // panel rows leave through LDS as 16-byte column pairs
#define CH_PSTORE16 0
#endif
__device__ inline void chol_load_tile_regs(const double* __restrict__ A, int ld, int nrows, int ncols, int r0, int c0, double* v) {
  const int t = threadIdx.x;
  if (CH_LOAD16 && !(ld & 1)) {
    const int c = (t & 15)*2, rb = t >> 4;      // rb 0..15
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 16*i + rb;
      chol_d2 x = {0.0, 0.0};
      if (r0 + r < nrows) {
        if (c0 + c + 1 < ncols) x = *reinterpret_cast<const chol_d2*>(A + (size_t)(r0 + r)*ld + c0 + c);
        else if (c0 + c < ncols) x[0] = A[(size_t)(r0 + r)*ld + c0 + c];
      }
      v[2*i] = x[0]; v[2*i + 1] = x[1];
    }
    return;
  }
  const int c = t & 31, rb = t >> 5;          // rb 0..7
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8*i + rb;
    v[i] = (r0 + r < nrows && c0 + c < ncols) ? A[(size_t)(r0 + r)*ld + c0 + c] : 0.0;
  }
}
__device__ inline void chol_regs_to_lds(const double* v, double (*T)[CH_NB + 1], int ld) {
  const int t = threadIdx.x;
  if (CH_LOAD16 && !(ld & 1)) {
    const int c = (t & 15)*2, rb = t >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) { T[16*i + rb][c] = v[2*i]; T[16*i + rb][c + 1] = v[2*i + 1]; }
    return;
  }
  const int c = t & 31, rb = t >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) T[8*i + rb][c] = v[i];
}

// one 16x16 quadrant (bi, bj) per wavefront:  acc (C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15,
// row = (lane >> 4) + 4*reg)  =  T[quadrant] - Pi[16 bi ..] Pj[16 bj ..]^T  over K = 32
__device__ inline void chol_quadrant_update(double (*T)[CH_NB + 1], double (*Pi)[CH_NB + 1], double (*Pj)[CH_NB + 1], bool mma, chol_d4& acc) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bi = w >> 1, bj = w & 1;
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = T[16*bi + rq + 4*g][16*bj + c];
  if (mma) {
#pragma unroll
    for (int kk = 0; kk < CH_NB; kk += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pi[16*bi + c][kk + rq], Pj[16*bj + c][kk + rq], acc, 0, 0, 0);
  }
}
__device__ inline void chol_quadrant_store(double (*T)[CH_NB + 1], const chol_d4& acc) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bi = w >> 1, bj = w & 1;
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) T[16*bi + rq + 4*g][16*bj + c] = acc[g];
}

// ---- panel factorisation: one row per lane, columns in registers, software-pipelined over the pivots -------------
// Column K of L (final after the scaling at pivot K) updates the columns to its right in two ways:
//   * fast path, at pivot K, for columns K+1 and K+2: the two multipliers are fetched with v_readlane.  This is
//     what the next pivot's critical path needs (its diagonal entry, then its reciprocal square root: v_rsq_f64 +
//     one third-order correction, a chain of dependent fp64 instructions);
//   * bulk, at pivot K+1, for columns >= K+3: the lanes of the diagonal tile write column K to LDS and every lane
//     reads the multipliers back with broadcast ds_read (uniform address, no VALU issue slots -- a v_readlane
//     pair per multiplier costs as much issue time as the fp64 FMA it feeds).  The reads are issued at pivot K and
//     consumed one pivot later, interleaved with the dependency chain of pivot K+1, so neither the LDS round trip
//     nor the chain latency is exposed.
// Every d[c] still receives its updates in column order, so the arithmetic is that of the plain right-looking loop.
#define CH_SB() __builtin_amdgcn_sched_barrier(0)
template <int K, int G> struct ChBulk {        // group G (of 6) of the bulk update by column K: columns [lo, hi)
  static constexpr int n = (CH_NB - K - 3 > 0) ? CH_NB - K - 3 : 0;
  static constexpr int lo = K + 3 + (n*G)/6, hi = K + 3 + (n*(G + 1))/6;
  static __device__ inline void fm(double* d, const double* m) {
#if !defined(CH_ABL)
#pragma unroll
    for (int c = lo; c < hi; ++c) d[c] -= d[K]*m[c];          // entries above the diagonal: unused garbage
#endif
  }
};
template <int J>
__device__ inline void chol_panel_pivot(double* d, double& inv, bool& bad, double* mE, double* mO, double* colbuf) {
  double* cur = (J & 1) ? mO : mE;          // multipliers of column J (requested here, used at pivot J+1)
  double* prev = (J & 1) ? mE : mO;         // multipliers of column J-1
  d[J] *= inv;                              // lane J holds the pivot itself: pivot * rsqrt(pivot) = L_JJ
#if !defined(CH_ABL) || CH_ABL < 2
  if constexpr (J + 3 < CH_NB) colbuf[threadIdx.x] = d[J];
#endif
  CH_SB();
  if constexpr (J + 1 < CH_NB) {
    // the next pivot, lane-locally: on lane J+1, d[J+1] - d[J]^2 is the updated diagonal entry
    const double w = __builtin_fma(-d[J], d[J], d[J + 1]);
    const double l1 = readlane_f64(d[J], J + 1);
    double l2 = 0.0;
    if constexpr (J + 2 < CH_NB) l2 = readlane_f64(d[J], J + 2);
    const double pn = readlane_f64(w, J + 1);
    bad |= !(pn > 0.0);                     // off the critical path: a failed factorisation is flagged, its numbers are not used
#if !defined(CH_ABL) || CH_ABL < 2
    if constexpr (J + 3 < CH_NB) {
#pragma unroll
      for (int c = J + 3; c < CH_NB; ++c) cur[c] = colbuf[c];
    }
#endif
    CH_SB();
    const double y0 = __builtin_amdgcn_rsq(pn);
#if defined(CH_RSQ2)
    // second-order (Newton) correction, one dependent instruction shorter: g = pn y0, h = y0/2, r = 1/2 - g h, inv = y0 + y0 r
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 0>::fm(d, prev); CH_SB();
    const double g = pn*y0;
    const double h = 0.5*y0;
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 1>::fm(d, prev); if constexpr (J >= 1) ChBulk<J - 1, 2>::fm(d, prev); CH_SB();
    const double r = __builtin_fma(-g, h, 0.5);
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 3>::fm(d, prev); CH_SB();
    inv = __builtin_fma(y0, r, y0);
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 4>::fm(d, prev); CH_SB();
#else
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 0>::fm(d, prev); CH_SB();
    const double t = y0*(-pn);
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 1>::fm(d, prev); CH_SB();
    const double e = __builtin_fma(t, y0, 1.0);
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 2>::fm(d, prev); CH_SB();
    const double u = y0*e;
    const double q = __builtin_fma(e, 0.375, 0.5);
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 3>::fm(d, prev); CH_SB();
    inv = __builtin_fma(u, q, y0);          // rsqrt(pn)
    CH_SB(); if constexpr (J >= 1) ChBulk<J - 1, 4>::fm(d, prev); CH_SB();
#endif
    d[J + 1] -= d[J]*l1;
    if constexpr (J >= 1) ChBulk<J - 1, 5>::fm(d, prev);
    CH_SB();
    if constexpr (J + 2 < CH_NB) d[J + 2] -= d[J]*l2;      // after the bulk update of d[J+2] by column J-1 (group 0)
    CH_SB();
  }
}
#undef CH_SB
template <int... Js>
__device__ inline void chol_panel_pivots(double* d, double& inv, bool& bad, double* colbuf, std::integer_sequence<int, Js...>) {
  double mE[CH_NB], mO[CH_NB];
  (chol_panel_pivot<Js>(d, inv, bad, mE, mO, colbuf), ...);
}

#ifdef MCP_CHOL_PROF
__device__ unsigned long long g_chol_prof[256*2*8];
#define CHOL_STAMP(i) do { if (blockIdx.x < 2 && blockIdx.y == 0 && threadIdx.x == 0) g_chol_prof[(k*2 + blockIdx.x)*8 + (i)] = clock64(); } while (0)
#else
#define CHOL_STAMP(i) do {} while (0)
#endif
__global__ void __launch_bounds__(CH_STEP_THREADS)
k_chol_step(double* __restrict__ S, int n, int nrows, int k, const int* __restrict__ tiles, int* __restrict__ fail, size_t sys_stride CH_DIAG_PARAMS(), int fuse_back /* single-tile system: the back-substitution here too */) {
  if (blockIdx.y) { S += blockIdx.y*sys_stride; fail += blockIdx.y; CH_DIAG_OFFSET(blockIdx.y); }     // further systems of a multi-lambda batch
  // tiles: the structurally non-zero tiles this step touches, packed (ti << 16 | tj), block column k first
  // bit 31: both tiles of panel k-1 the update of this tile multiplies -- (ti, k-1) and (tj, k-1) -- are tiles of the plan; bit 30:
  // so is (k, k-1), which updates the diagonal tile.  A tile outside the plan is structurally zero and S holds NOTHING for it (the
  // assembly writes the plan's tiles only): it must not be read -- whatever the buffer held before would be taken for numbers.
  const unsigned int packed = (unsigned int)tiles[blockIdx.x];
  const int ti = (int)((packed >> 16) & 0x3fffu), tj = (int)(packed & 0xffffu);
  const bool upd = k > 0 && (packed >> 31) != 0u, upd_d = k > 0 && ((packed >> 30) & 1u) != 0u;
  __shared__ double Ta[CH_NB][CH_NB + 1];
  __shared__ double Tb[CH_NB][CH_NB + 1];
  __shared__ double Tc[CH_NB][CH_NB + 1];
  __shared__ double Td[CH_NB][CH_NB + 1];
  __shared__ double Te[CH_NB][CH_NB + 1];
  const int lane = threadIdx.x;
  const int r0 = ti*CH_NB, c0 = tj*CH_NB, k0 = k*CH_NB, p0 = (k - 1)*CH_NB;
  const bool panel = (tj == k), offdiag = (ti != k);
  CHOL_STAMP(0);
  // ---- all global loads up front: own tile, the two tiles of panel k-1, and (block column k) the diagonal tile
  {
    double vc[4], va[4], vb[4], vd[4], ve[4];
    chol_load_tile_regs(S, n, nrows, n, r0, c0, vc);
    if (upd) {
      chol_load_tile_regs(S, n, nrows, n, r0, p0, va);
      chol_load_tile_regs(S, n, nrows, n, c0, p0, vb);
    }
    if (panel && offdiag) {
      chol_load_tile_regs(S, n, nrows, n, k0, k0, vd);
      if (upd_d) chol_load_tile_regs(S, n, nrows, n, k0, p0, ve);
    }
    chol_regs_to_lds(vc, Tc, n);
    if (upd) { chol_regs_to_lds(va, Ta, n); chol_regs_to_lds(vb, Tb, n); }
    if (panel && offdiag) { chol_regs_to_lds(vd, Td, n); if (upd_d) chol_regs_to_lds(ve, Te, n); }
  }
  __syncthreads();
  CHOL_STAMP(1);
  chol_d4 acc, dacc;
  chol_quadrant_update(Tc, Ta, Tb, upd, acc);
  if (panel && offdiag) chol_quadrant_update(Td, Te, Te, upd_d, dacc);
  chol_quadrant_store(Tc, acc);                 // each wavefront reads and writes only its own quadrant of Tc / Td
  if (panel && offdiag) chol_quadrant_store(Td, dacc);
  __syncthreads();
  CHOL_STAMP(2);
  if (!panel) {      // plain trailing tile: write back and leave
    if (CH_STORE16 && !(n & 1) && ti > tj) {      // off-diagonal tile, even n: whole 16-byte column pairs (c0 + c + 1 < n because n is even)
      const int c = (lane & 15)*2, rb = lane >> 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 16*i + rb;
        if (r0 + r < nrows && c0 + c < n) { chol_d2 x = {Tc[r][c], Tc[r][c + 1]}; *reinterpret_cast<chol_d2*>(S + (size_t)(r0 + r)*n + c0 + c) = x; }
      }
      return;
    }
    const int c = lane & 31, rb = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8*i + rb;
      if (r0 + r < nrows && c0 + c < n && (ti > tj || c <= r)) S[(size_t)(r0 + r)*n + c0 + c] = Tc[r][c];
    }
    return;
  }
  if (lane >= 64) return;      // the panel factorisation is one wavefront's job (no barrier below this line)
  // ---- block column k: unblocked panel factorisation of [diagonal tile ; own tile], one row per lane.
  // lanes 0..31 hold the rows of the diagonal tile, lanes 32..63 the rows of the own tile (for the diagonal
  // block itself: only rows beyond the matrix, i.e. the right-hand-side row).  Applying the column operations
  // of the Cholesky factorisation to the lower rows yields X = C L_kk^-T without a separate triangular solve.
  double (*Tdiag)[CH_NB + 1] = offdiag ? Td : Tc;
  const int nbe = min(CH_NB, n - k0);
  const int rr = lane & 31;
  const bool low = lane >= 32;
  double d[CH_NB];
  if (!low) {
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) d[c] = (c <= rr && rr < nbe && c < nbe) ? Tdiag[rr][c] : ((c == rr) ? 1.0 : 0.0);
  } else if (offdiag || rr >= nbe) {
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) d[c] = (c < nbe) ? Tc[rr][c] : 0.0;
  } else {
    // diagonal workgroup: its lower lanes are free (the tile's rows sit in lanes 0..31), so they carry the identity through the
    // same column operations and come out as I L_kk^-T -- lane 32 + r ends with row r of L_kk^-T = column r of L_kk^-1, which
    // is all the back-substitution needs of this tile (a 32 x 32 matrix-vector product instead of a 32-pivot triangular solve)
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) d[c] = (c == rr) ? 1.0 : 0.0;
  }
  const double piv0 = readlane_f64(d[0], 0);
  bool bad = !(piv0 > 0.0);
  double inv = rsqrt(piv0);
  chol_panel_pivots(d, inv, bad, &Ta[0][0], std::make_integer_sequence<int, CH_NB>());     // Ta is free now: column buffer
  CHOL_STAMP(3);
  if (bad && lane == 0) atomicOr(fail, 2);
  if (CH_PSTORE16 && !(n & 1)) {
    // even n: the 32 result rows leave through LDS (row per lane in, 16-byte column pairs out: a store instruction then covers
    // four whole rows of the tile instead of one double in each of 64 rows).  Single wavefront: its LDS accesses execute in order.
    if (low) {
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) Tc[rr][c] = d[c];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i*64 + lane, r = e >> 4, c = (e & 15)*2;
      if (c >= nbe) continue;                                     // nbe is even
      double* dst;
      if (!offdiag && r < nbe) dst = Dg + (size_t)k*(CH_NB*CH_NB) + r*CH_NB + c;           // row r of L_kk^-T (entries left of the diagonal are never read)
      else if (r0 + r < nrows && (offdiag || r >= nbe)) dst = S + (size_t)(r0 + r)*n + k0 + c;
      else continue;
      const chol_d2 x = {Tc[r][c], Tc[r][c + 1]};
      *reinterpret_cast<chol_d2*>(dst) = x;
    }
    CHOL_STAMP(4);
    return;
  }
  if (fuse_back) {
    // A system of one tile (n <= 32: the BundleAdjustRecent window, <= 5 free poses): x = L^-T y is a 32 x 32 matrix-vector product of
    // what this wavefront holds -- lane 32 + r: row r of L^-T, lane 32 + nbe: y (the right-hand-side row after the column operations)
    // -- taken here, in the order k_chol_back takes it (even and odd columns apart, then their sum), instead of in a launch of its own.
    const int ly = 32 + nbe;
    double xa = 0.0, xb = 0.0;
#pragma unroll
    for (int c = 0; c < CH_NB; c += 2) {
      const double y0 = readlane_f64(d[c], ly), y1 = readlane_f64(d[c + 1], ly);
      const double l0 = (c >= rr && rr < nbe && c < nbe) ? d[c] : ((c == rr) ? 1.0 : 0.0);
      const double l1 = (c + 1 >= rr && rr < nbe && c + 1 < nbe) ? d[c + 1] : ((c + 1 == rr) ? 1.0 : 0.0);
      xa += l0*((c < nbe) ? y0 : 0.0); xb += l1*((c + 1 < nbe) ? y1 : 0.0);
    }
    if (low && rr < nbe) S[(size_t)n*n + rr] = xa + xb;
  }
  if (!low) {
    // (L_kk itself is not stored: the forward substitution rides on the factorisation as the augmented row, the backward
    // one uses the inverse below; the other tiles of block column k already hold X = C L_kk^-T)
  } else if (!offdiag && rr < nbe) {
    double* p = Dg + (size_t)k*(CH_NB*CH_NB) + rr*CH_NB;        // row rr of L_kk^-T (upper triangular)
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) if (c >= rr && c < nbe) p[c] = d[c];
  } else if (!fuse_back && r0 + rr < nrows && (offdiag || rr >= nbe)) {      // (fused back-substitution: row n holds x already)
    double* p = S + (size_t)(r0 + rr)*n + k0;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) if (c < nbe) p[c] = d[c];
  }
  CHOL_STAMP(4);
}

// backward substitution L^T x = y (y = row n of the augmented matrix); one workgroup per system.
// Wavefront 0 multiplies by L_kk^-T of the step (a by-product of the factorisation; tile prefetched into registers during the
// previous step's update); the other 7 wavefronts spread the update y[c] -= sum_r L[k0+r][c] x[k0+r] over (column pair, 16-row slice) items --
// one L2 round trip of 16 independent 16-byte loads per item (n = 6P is even, so column pairs are aligned) instead of a
// 32-long walk per column -- and combine the two slices of a column pair by a lane shuffle (fixed order, no atomics).
constexpr int CH_BACK_THREADS = 512;
constexpr int CH_BACK_RG = 2;            // row slices per column
constexpr int CH_SOLVE_MAX = 6144;       // x is staged in LDS
// tiles of step kb into LDS, spread over `nth` threads (index `u`): Lt = L_kk^-T (upper, unit-padded outside the matrix),
// Tt = the tile below it, L[k0+32 .. ][k0 ..] (zero outside); coalesced row segments; load and LDS store are split so the
// loads stay in flight during the update
constexpr int CH_STAGE_PER = 5;                            // ceil(2*1024 / (CH_BACK_THREADS - 64))
__device__ inline void chol_back_stage_load(const double* __restrict__ S CH_DIAG_PARAMS(const), int n, int nblk, int kb, int u, int nth, double* v, bool below /* tile (kb+1, kb) is a tile of the plan: S holds it */) {
  const int k0 = kb*CH_NB, nbe = min(CH_NB, n - k0);
  const int kb0 = k0 + CH_NB, nbb = (kb + 1 < nblk) ? min(CH_NB, n - kb0) : 0;
#pragma unroll
  for (int i = 0; i < CH_STAGE_PER; ++i) {
    const int e = u + i*nth;
    const int which = e >> 10, r = (e >> 5) & 31, c = e & 31;
    v[i] = 0.0;
    if (e < 2048) {
      if (which == 0) v[i] = (c >= r && r < nbe && c < nbe) ? Dg[(size_t)kb*(CH_NB*CH_NB) + r*CH_NB + c] : ((r == c) ? 1.0 : 0.0);   // L_kk^-T, upper
      else v[i] = (below && r < nbb && c < nbe) ? S[(size_t)(kb0 + r)*n + k0 + c] : 0.0;
    }
  }
}
__device__ inline void chol_back_stage_store(int u, int nth, const double* v, double (*Lt)[CH_NB + 1], double (*Tt)[CH_NB + 1]) {
#pragma unroll
  for (int i = 0; i < CH_STAGE_PER; ++i) {
    const int e = u + i*nth;
    if (e < 2048) { if ((e >> 10) == 0) Lt[(e >> 5) & 31][e & 31] = v[i]; else Tt[(e >> 5) & 31][e & 31] = v[i]; }
  }
}
#ifdef MCP_CHOL_PROF
__device__ unsigned long long g_back_prof[256*2*8];
#define BACK_STAMP(who, i) do { if (blockIdx.x == 0 && threadIdx.x == (who)*64) g_back_prof[(kb*2 + (who))*8 + (i)] = clock64(); } while (0)
#else
#define BACK_STAMP(who, i) do {} while (0)
#endif
// Time step kb: wavefront 0 (solver) first applies the one tile that links the block it solved in the previous step to
// block kb (tile column in registers, x broadcast with v_readlane), then solves the 32x32 triangle; at the same time
// the other 7 wavefronts subtract the PREVIOUS block's contribution from all column blocks further left, over
// (column pair, 16-row slice) items -- one L2 round trip of 16 independent 16-byte loads per item (n = 6P is even, so
// column pairs are aligned) -- combining the two slices of a column pair by a lane shuffle; they first stage the solver's
// two tiles of the NEXT step in LDS.  One barrier per step; the solver touches neither global memory nor the
// bulk update of the step before.
__global__ void __launch_bounds__(CH_BACK_THREADS)
k_chol_back(const double* __restrict__ S, int n, const int* __restrict__ row_start, const int* __restrict__ row_tiles, double* __restrict__ xout, size_t sys_stride CH_DIAG_PARAMS(const)) {
  if (blockIdx.x) { S += blockIdx.x*sys_stride; xout += blockIdx.x*sys_stride; CH_DIAG_OFFSET(blockIdx.x); }
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int t = threadIdx.x;
  const double* y = S + (size_t)n*n;
  for (int i = t; i < n; i += CH_BACK_THREADS) xs[i] = y[i];
  const int nblk = (n + CH_NB - 1)/CH_NB;
  const bool solver = t < 64;
  const int rr = t & 31;
  __shared__ double Lt[2][CH_NB][CH_NB + 1], Tt[2][CH_NB][CH_NB + 1];     // tiles of the current / next step (by parity)
  // the block-row index of the plan, staged once: every step would otherwise start with two dependent global index reads
  constexpr int RT_CAP = 4096;
  __shared__ int rs_l[CH_SOLVE_MAX/CH_NB + 2], rt_l[RT_CAP];
  const int n_rt = row_start[nblk];
  for (int i = t; i <= nblk; i += CH_BACK_THREADS) rs_l[i] = row_start[i];
  if (n_rt <= RT_CAP) for (int i = t; i < n_rt; i += CH_BACK_THREADS) rt_l[i] = row_tiles[i];
  const int* rt = (n_rt <= RT_CAP) ? rt_l : row_tiles;
  double xprev = 0.0;
  // the solver's two tiles travel global -> registers -> LDS two steps ahead of their use: loaded during step j+2, stored to
  // LDS at the start of step j+1 (into the buffer of the other parity), read by the solver in step j
  double sv[CH_STAGE_PER];
  // is the tile below the diagonal tile of step kb -- (kb+1, kb) -- one of the plan's?  (block row kb+1 lists its tile columns in
  // ascending order.)  If not it is structurally zero and S holds nothing for it.
  auto below_in_plan = [&](int kb) -> bool {
    if (kb + 1 >= nblk) return false;
    const int b = row_start[kb + 1], e = row_start[kb + 2];
    return e > b && row_tiles[e - 1] == kb;
  };
  if (t >= 64) {
    chol_back_stage_load(S CH_BACK_DG, n, nblk, nblk - 1, t - 64, CH_BACK_THREADS - 64, sv, false);
    chol_back_stage_store(t - 64, CH_BACK_THREADS - 64, sv, Lt[(nblk - 1) & 1], Tt[(nblk - 1) & 1]);
    if (nblk > 1) chol_back_stage_load(S CH_BACK_DG, n, nblk, nblk - 2, t - 64, CH_BACK_THREADS - 64, sv, below_in_plan(nblk - 2));
  }
  __syncthreads();
  // updaters: (tile, column pair, row slice) items of block row ub against x of block ub; item q of a thread's first round is
  // PREFETCHED one step ahead (the L values do not depend on x), so the cold-miss latency of a step's loads hides behind the
  // solver's work of the step before
  constexpr int RPI = CH_NB/CH_BACK_RG;                 // rows per item
  constexpr int IPT = (CH_NB/2)*CH_BACK_RG;             // items per tile
  constexpr int NUP = CH_BACK_THREADS - 64;             // updater threads
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 pv[RPI]; int p_c = -1, p_rg = 0;
  // issue the loads of item q of block row ub, skipping tile `skip` (the solver's own tile of the step that consumes them)
  auto load_item = [&](int ub, int q, int skip, d2* v, int& c_out, int& rg_out) {
    c_out = -1;
    const int l0 = rs_l[ub], items = (rs_l[ub + 1] - l0)*IPT;
    if (q >= items) return;
    const int tile = rt[l0 + q/IPT];
    if (tile == skip) return;
    const int rem = q % IPT;
    const int rg = rem & (CH_BACK_RG - 1), c = tile*CH_NB + 2*(rem/CH_BACK_RG);     // the row slices of a column pair sit in neighbouring lanes
    const int u0 = ub*CH_NB, nbu = min(CH_NB, n - u0), r0 = rg*RPI;
#pragma unroll
    for (int r = 0; r < RPI; ++r) v[r] = (r0 + r < nbu) ? *reinterpret_cast<const d2*>(S + (size_t)(u0 + r0 + r)*n + c) : (d2){0.0, 0.0};
    c_out = c; rg_out = rg;
  };
  // both lanes of a slice pair (2i, 2i+1: same tile, same column pair) are active together: the item counts are even
  auto apply_item = [&](int ub, const d2* v, int c, int rg) {
    const int u0 = ub*CH_NB, nbu = min(CH_NB, n - u0), r0 = rg*RPI;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int r = 0; r < RPI; ++r) { const double xr = (r0 + r < nbu) ? xs[u0 + r0 + r] : 0.0; a0 += v[r][0]*xr; a1 += v[r][1]*xr; }
    // fixed-order combination of the two slices; one lane owns the column pair in this step -- no atomics, reproducible
    static_assert(CH_BACK_RG == 2 && (NUP % 2) == 0, "slice pairing");
    a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
    if (rg == 0) { xs[c] -= a0; xs[c + 1] -= a1; }
  };
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb*CH_NB, nbe = min(CH_NB, n - k0);
    BACK_STAMP(0, 0); BACK_STAMP(1, 0);
    if (solver) {
      // column rr of the tile below and row rr of L_kk^-T, staged in LDS by the updaters during the previous step; consumed in
      // halves of 16 so that the solver's live registers stay below the updaters' (the kernel's allocation is the maximum)
      double yv = (rr < nbe) ? xs[k0 + rr] : 0.0;
      // contribution of the block solved in the previous step (the tile below is zero in the first step)
#pragma unroll
      for (int h = 0; h < CH_NB; h += 16) {
        double Tf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) Tf[r] = Tt[kb & 1][h + r][rr];
#pragma unroll
        for (int r = 0; r < 16; ++r) yv -= Tf[r]*readlane_f64(xprev, h + r);
      }
      // x = L_kk^-T y with the tile's inverse (a by-product of the factorisation, k_chol_step): x[rr] = sum_c (L^-T)[rr][c] y[c]
      {
        double xa = 0.0, xb = 0.0;
#pragma unroll
        for (int h = 0; h < CH_NB; h += 16) {
          double Lc[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) Lc[c] = Lt[kb & 1][rr][h + c];
#pragma unroll
          for (int c = 0; c < 16; c += 2) { xa += Lc[c]*readlane_f64(yv, h + c); xb += Lc[c + 1]*readlane_f64(yv, h + c + 1); }
        }
        yv = xa + xb;
      }
      if (t < nbe) xs[k0 + t] = yv;
      xprev = (rr < nbe) ? yv : 0.0;
      BACK_STAMP(0, 1);
    } else {
      if (kb > 0) chol_back_stage_store(t - 64, CH_BACK_THREADS - 64, sv, Lt[(kb - 1) & 1], Tt[(kb - 1) & 1]);      // tiles of step kb-1 (loaded a step ago)
      if (kb > 1) chol_back_stage_load(S CH_BACK_DG, n, nblk, kb - 2, t - 64, CH_BACK_THREADS - 64, sv, rs_l[kb] > rs_l[kb - 1] && rt[rs_l[kb] - 1] == kb - 2);           // tiles of step kb-2: a full step to land
      if (kb + 1 < nblk) {
        // x of block kb+1 (solved in the previous step) against the structurally non-zero tiles of block row kb+1 left of tile kb
        const int ub = kb + 1;
        if (p_c >= 0) apply_item(ub, pv, p_c, p_rg);                         // first round: prefetched during the previous step
        const int items = (rs_l[ub + 1] - rs_l[ub])*IPT;
        for (int q = t - 64 + NUP; q < items; q += NUP) {                    // further rounds (long block rows): loaded here
          d2 v[RPI]; int c, rg;
          load_item(ub, q, kb, v, c, rg);
          if (c >= 0) apply_item(ub, v, c, rg);
        }
      }
      // block row kb is consumed in the next step (against tile kb-1 the solver handles itself): fetch its first round now
      p_c = -1;
      if (kb > 0) load_item(kb, t - 64, kb - 1, pv, p_c, p_rg);
    }
    BACK_STAMP(0, 3); BACK_STAMP(1, 3);
    __syncthreads();
    BACK_STAMP(0, 4); BACK_STAMP(1, 4);
  }
  for (int i = t; i < n; i += CH_BACK_THREADS) xout[i] = xs[i];
}

}  // namespace mcp
#include "ba_chol2.h"
namespace mcp {

// Block-sparse execution plan (the job CHOLMOD's symbolic analysis does for the reference): which 32x32 tiles
// of the lower triangle are structurally non-zero after fill-in, in the given (natural) pose order.
struct CholPlan {
  int n = 0, ntc = 0, ntr = 0;
  std::vector<int> step_start, step_tiles;     // per step k: tiles to process, block column k first
  std::vector<int> row_start, row_tiles;       // per block row: non-zero tile columns left of the diagonal
  std::vector<int> all_tiles;                  // every tile of the plan after fill-in, incl. the right-hand-side row (ti << 16 | tj)
  int* d_all_tiles = nullptr;
  int* d_step_tiles = nullptr; int* d_row_start = nullptr; int* d_row_tiles = nullptr;
  double* d_diag = nullptr; size_t diag_stride = 0; static constexpr int max_sys = 4;     // factored diagonal tiles
  // the one-launch factorisation + chain back-substitution of ba_chol2.h (MCP_BA_CHOL_PERSIST=0: the per-step kernels below)
  mutable CholPersist persist; bool use_persist = false;
  const char* launch_failed = nullptr;      // name of a kernel of this plan whose launch the runtime refused (the solver reports it)
  std::vector<int> persist_segs;      // first block column of every chain of the one-launch plan (ba_chol2.h CholPersist::build); empty = one chain
  int persist_min_ntc = 3;      // smallest system (in tiles) the one-launch plan is built for (the test hooks set 1)
  ~CholPlan() { release(); }
  int arena_dev = -1;
  char* arena = nullptr; size_t arena_cap = 0;      // one cached block (ba_pool.h): [step tiles | row starts | row tiles | all tiles | factored diagonal tiles]
  void release() { if (arena) DevCache::get().put(arena, arena_cap, arena_dev); arena = nullptr; arena_cap = 0; arena_dev = -1;
                   d_diag = nullptr; d_all_tiles = nullptr;
                   d_step_tiles = d_row_start = d_row_tiles = nullptr; persist.release(); }
  // pattern: ntc x ntc lower-triangular tile occupancy of S (true = may be non-zero); empty = dense
  int build(int n_, const std::vector<unsigned char>& pattern) {
    release();
    n = n_; ntc = (n + CH_NB - 1)/CH_NB; ntr = (n + 1 + CH_NB - 1)/CH_NB;
    std::vector<unsigned char> P((size_t)ntr*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) P[(size_t)i*ntc + j] = pattern.empty() ? 1 : pattern[(size_t)i*ntc + j];
    for (int i = 0; i < ntc; ++i) P[(size_t)i*ntc + i] = 1;
    const int rhs_tile = n/CH_NB;                                  // the row tile holding the right-hand side: dense
    for (int j = 0; j < ntc; ++j) P[(size_t)rhs_tile*ntc + j] = 1;
    for (int k = 0; k < ntc; ++k) {                                // symbolic fill-in
      std::vector<int> rows;
      for (int i = k + 1; i < ntr; ++i) if (P[(size_t)i*ntc + k]) rows.push_back(i);
      for (int a : rows) for (int b : rows) if (b <= a && b < ntc) P[(size_t)a*ntc + b] = 1;
    }
    all_tiles.clear();
    for (int i = 0; i < ntr; ++i) for (int j = 0; j < ntc && j <= i; ++j) if (P[(size_t)i*ntc + j]) all_tiles.push_back((i << 16) | j);
    step_start.assign(ntc + 1, 0); step_tiles.clear();
    for (int k = 0; k < ntc; ++k) {
      step_start[k] = (int)step_tiles.size();
      for (int i = k; i < ntr; ++i) if (P[(size_t)i*ntc + k]) step_tiles.push_back((i << 16) | k);       // block column k (critical path) first
      if (k > 0) {
        std::vector<int> rows;
        for (int i = k; i < ntr; ++i) if (P[(size_t)i*ntc + k - 1]) rows.push_back(i);
        for (int a : rows) for (int b : rows) if (b <= a && b < ntc && b > k) step_tiles.push_back((a << 16) | b);
        // tiles of block column k that panel k-1 does not touch were emitted above already (they only need the panel step)
      }
    }
    step_start[ntc] = (int)step_tiles.size();
    row_start.assign(ntc + 1, 0); row_tiles.clear();
    for (int i = 0; i < ntc; ++i) { row_start[i] = (int)row_tiles.size(); for (int j = 0; j < i; ++j) if (P[(size_t)i*ntc + j]) row_tiles.push_back(j); }
    row_start[ntc] = (int)row_tiles.size();
    diag_stride = (size_t)std::max(ntc, 1)*CH_NB*CH_NB;
    {
      auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
      const size_t o_step = 0, o_rs = o_step + al(4*std::max<size_t>(step_tiles.size(), 1)), o_rt = o_rs + al(4*row_start.size()),
                   o_all = o_rt + al(4*std::max<size_t>(row_tiles.size(), 1)), o_diag = o_all + al(4*std::max<size_t>(all_tiles.size(), 1));
      arena = (char*)DevCache::get().take(o_diag + sizeof(double)*diag_stride*max_sys, &arena_cap, &arena_dev);
      if (!arena) return -1;
      std::vector<char> stage(o_diag, 0);
      {
        // the device's copy carries the update flags of k_chol_step (which tiles of panel k-1 exist)
        unsigned int* dst = reinterpret_cast<unsigned int*>(stage.data() + o_step);
        for (int k = 0; k < ntc; ++k) for (int i = step_start[k]; i < step_start[k + 1]; ++i) {
          const int ti = step_tiles[i] >> 16, tj = step_tiles[i] & 0xffff;
          unsigned int v = (unsigned int)step_tiles[i];
          if (k > 0) {
            if (P[(size_t)ti*ntc + k - 1] && P[(size_t)tj*ntc + k - 1]) v |= 1u << 31;
            if (P[(size_t)k*ntc + k - 1]) v |= 1u << 30;
          }
          dst[i] = v;
        }
      }
      std::memcpy(stage.data() + o_rs, row_start.data(), 4*row_start.size());
      if (!row_tiles.empty()) std::memcpy(stage.data() + o_rt, row_tiles.data(), 4*row_tiles.size());
      if (!all_tiles.empty()) std::memcpy(stage.data() + o_all, all_tiles.data(), 4*all_tiles.size());
      if (hipMemcpy(arena, stage.data(), o_diag, hipMemcpyHostToDevice) != hipSuccess) return -1;
      d_step_tiles = (int*)(arena + o_step); d_row_start = (int*)(arena + o_rs); d_row_tiles = (int*)(arena + o_rt); d_all_tiles = (int*)(arena + o_all);
      d_diag = (double*)(arena + o_diag);
    }
    { const char* e = getenv("MCP_BA_CHOL_PERSIST"); use_persist = !(e && atoi(e) == 0); }
    // (a reduced system of one or two tiles -- the BundleAdjustRecent window, <= 5 free poses -- is two launches either way: no plan for it)
    if (use_persist && ntc >= persist_min_ntc && persist.build(n, pattern, pattern.empty() ? std::vector<int>() : all_tiles, persist_segs)) return -1;
    return 0;
  }
  size_t tile_updates() const { return step_tiles.size(); }
  // one tile, 16-byte panel stores off (the fused path lives in the plain store branch): factorisation and back-substitution in one launch
  bool fuse_single() const { return ntc == 1 && n < CH_NB /* the right-hand-side row lies in the same tile */ && !(CH_PSTORE16 && !(n & 1)) && fuse_single_on; }
  bool fuse_single_on = [] { const char* e = getenv("MCP_BA_CHOL_FUSE1"); return !(e && atoi(e) == 0); }();
};

// factor S (n x n, lower) with the rhs in row n: afterwards row n holds y = L^-1 rhs
// nsys > 1 factors further systems stored q*sys_stride doubles behind the first in the same launches (fail[q] is their flag)
// q0: index of the first of the nsys systems inside the batch buffers (S, fail and the diagonal side array are offset by it)
// (a one-launch factorisation the runtime refuses -- hipFuncSetAttribute failed -- switches the plan to the per-step kernels and
//  goes on with them in the same call: the back-substitution must never run on vectors no factorisation has armed)
inline void chol_factor(hipStream_t st, CholPlan& plan, double* S, int* fail, int nsys = 1, size_t sys_stride = 0, int q0 = 0) {
  if (plan.use_persist && plan.persist.ok) {
    if (chol_persist_factor(st, plan.persist, S, fail, nsys, sys_stride, q0) == 0) return;
    plan.use_persist = false; (void)hipGetLastError();
  }
  const int n = plan.n, nrows = n + 1;
  S += q0*sys_stride; fail += q0;
  double* dg = plan.d_diag + q0*plan.diag_stride;
  for (int k = 0; k < plan.ntc; ++k) {
    const int cnt = plan.step_start[k + 1] - plan.step_start[k];
    if (cnt > 0) hipLaunchKernelGGL(k_chol_step, dim3(cnt, nsys), dim3(CH_STEP_THREADS), 0, st, S, n, nrows, k, (const int*)(plan.d_step_tiles + plan.step_start[k]), fail, sys_stride, dg, plan.diag_stride,
                                    plan.fuse_single() ? 1 : 0);
  }
}
// row n: y -> x = L^-T y
inline void chol_back(hipStream_t st, CholPlan& plan, double* S, int nsys = 1, size_t sys_stride = 0, int q0 = 0) {
  if (plan.use_persist && plan.persist.ok) { if (chol_persist_back(st, plan.persist, S, nsys, sys_stride, q0)) plan.launch_failed = "k_chol_back2"; return; }
  if (plan.fuse_single()) return;      // (k_chol_step has left x in row n)
  const int n = plan.n;
  S += q0*sys_stride;
  const double* dg = plan.d_diag + q0*plan.diag_stride;
  // x (up to CH_SOLVE_MAX doubles) + the staged tiles can exceed the default 64 KB of LDS; function attributes are per device
  static std::atomic<unsigned long long> attr_mask{0};
  int dev = 0; (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_mask.load(std::memory_order_relaxed) & bit)) {
    (void)hipFuncSetAttribute((const void*)k_chol_back, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SOLVE_MAX*(int)sizeof(double));
    attr_mask.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(k_chol_back, dim3(nsys), dim3(CH_BACK_THREADS), (size_t)n*sizeof(double), st, (const double*)S, n,
                     (const int*)plan.d_row_start, (const int*)plan.d_row_tiles, S + (size_t)n*n, sys_stride, dg, plan.diag_stride);
}

}  // namespace mcp
