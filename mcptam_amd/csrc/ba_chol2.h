// ba_chol2.h -- the reduced pose system factored in ONE persistent launch, solved in a second one (gfx950).
//
// Same job as k_chol_step x 38 + k_chol_back of ba_chol.h (g2o::LinearSolverCholmod, src/ChainBundle.cc:1150-1158), other
// schedule.  The per-step launches were bound by what sits between two pivots blocks: launch gap, tile loads, tile stores
// (DESIGN 4).  Here the dependent chain never leaves one compute unit:
//
//   * the CRITICAL workgroup owns the band, tiles (i, j) with i - j < CP_W, in LDS.  Per block column s: X1 = A(s+1,s) L_ss^-T
//     and A(s+1,s+1) -= X1 X1^T on the matrix cores (all four wavefronts), then wavefront 0 factors the 32 x 32 diagonal tile
//     (the panel code of ba_chol.h, identity rows carried along -> L^-1) WHILE the other three wavefronts finish column s for
//     row s+2, publish L_ss^-1 / L(s+1,s) / L(s+2,s), and take in row s+3's band tiles.  Nothing on this path waits for memory
//     that another compute unit has to produce "now": what arrives was due a whole step ago.
//   * one HELPER workgroup per other tile of the plan, left-looking: A(i,j) - sum_m L(i,m) L(j,m)^T in registers over the
//     columns m as they are published (fixed ascending order: bit-reproducible), then either X = . L_jj^-T -> published (far
//     tiles, i - j >= CP_W) or the partial sum handed to the critical workgroup (band tiles, columns m <= i - CP_W).
//   * hand-offs: payload by 16-byte write-through (sc1) stores, every storing wavefront drains, ONE lane stores the tile's flag
//     (epoch << 2 | stage, relaxed agent scope); consumers poll that word relaxed and read the payload with sc1 loads
//     (MI355X_MICROARCH.md, inter-workgroup visibility, form R1).  Flags are never reset: every launch has its own epoch.
//     Every spin is bounded; a timeout raises the system's error word, every other spinner sees it and leaves, and the host falls
//     back to the per-step kernels (S itself is never written by the factorisation, so nothing has to be restored).
//   * workgroup order = dependency order (helpers sorted by the column that completes them): a workgroup only ever waits for
//     lower-numbered ones or for the critical workgroup (block 0), so the launch makes progress with any number of resident
//     workgroups as long as blocks are dispatched in order.
//
// The right-hand side is block row R = ntc of the plan (one meaningful row): its tiles come out as y = L^-1 rhs.
// Back-substitution L^T x = y (k_chol_back2): one chain workgroup walks the block columns right to left with the near tiles
// (rows k+1 .. k+CP_BACK_NEAR) staged in LDS one step ahead; one helper workgroup per column sums the far tiles' products as
// the x blocks appear.  x and the far sums travel as data-tagged doubles (a NaN pattern no arithmetic produces = "not yet").
#pragma once

namespace mcp {

constexpr int CP_W = 3;                       // band of the critical workgroup: tiles (i, j) with i - j < CP_W
constexpr int CP_THREADS = 256;
constexpr int CP_LD = CH_NB + 1;              // LDS row stride of a tile
constexpr int CP_TILE = CH_NB*CP_LD;          // doubles per LDS tile
constexpr int CP_TQ = CH_NB*CH_NB;            // doubles per tile in global memory (thread-major quadrant layout)
// Every spin is bounded in WALL-CLOCK time (s_memrealtime: 100 MHz on gfx950, the same on every compute unit): a poll that has
// seen nothing for cp_deadline ticks (20 ms by default, MCP_BA_CHOL_DEADLINE_MS; a whole factorisation takes 0.23 ms) raises the
// error word.  (Round 4 counted 2^22 polls instead: seconds inside a kernel.)
__device__ long long cp_deadline = 2000000;
__device__ inline bool cp_expired(long long t0) { return wall_clock64() - t0 > cp_deadline; }
constexpr int CP_BACK_NEAR = 3;               // rows k+1 .. k+CP_BACK_NEAR of column k stay with the chain workgroup
constexpr unsigned int CP_SENT32 = 0xFFF5A5A5u;
constexpr unsigned long long CP_SENT = ((unsigned long long)CP_SENT32 << 32) | CP_SENT32;      // "not written yet": a NaN payload arithmetic never yields (both halves alike: hipMemsetD32 fills a buffer with it)

#ifndef CP_HELPER_BATCH
#define CP_HELPER_BATCH 1
#endif
constexpr int CP_HB = 4;              // tiles a helper keeps in flight ahead of its products (batched updates)
constexpr int CP_MAX_SEG = 7;
struct CpHelper { int ti, tj, slot, dslot, upd0, nupd, kind, in_s, pre, pre_flag, pre_diag, pad; };
// kind: 0 far tile, 1 band tile.  pre >= 0 -- far tile (i, i - CP_W): band slot its sum goes to BEFORE the solve with L_jj^-T (flag
// index pre_flag), so that the band tiles of row i, whose last update needs L(i, i - CP_W), need not wait for this tile's own
// publication: they redo the solve from that sum the moment L_jj^-1 appears (band tile: pre = that band slot, pre_diag = slot of
// L_jj^-1, and the LAST entry of its update list is the one to be taken that way)
// what the critical workgroup looks up per step, laid out for it by the host (copied to LDS once: a table look-up in global
// memory is a round trip, and a waited-for load also waits for every load issued before it -- the band row fetched ahead)
constexpr int CP_STEP_INTS = 12;
enum { CPS_SD = 0, CPS_S1, CPS_S2, CPS_DL, CPS_F0, CPS_F1, CPS_F2, CPS_B0, CPS_B1, CPS_B2 };
struct CpArgs {
  const double* S; size_t sys_stride; int n, ntc, nsys, nhelpers, nworkers;
  int nseg; int seg_start[CP_MAX_SEG + 1];      // block columns [seg_start[g], seg_start[g+1]) are critical workgroup g's (round 6: independent chains side by side)
  const int* slot_of; const int* bslot_of; const int* delta_of; const int* steps; const CpHelper* helpers; const int2* upd;
  double* Lt; size_t lt_stride;               // published L tiles / L_kk^-1, per system
  double* Bt; size_t bt_stride;               // band tiles handed to the critical workgroup, per system
  int test_fail_step;             // >= 0: the critical workgroup raises the error word at that step (test of the fall-back)
  int* flags; int nslots; int nflags; int* err; int* fail; const int* epoch;      // epoch[q]: bumped by the back-substitution launch (graph replay safe)
  double* xbuf; double* fbuf; int vec_stride;
  int* claim;                     // claim[q]: next entry of the helper list nobody has taken yet (zeroed by the launch that follows: k_chol_back2 / k_cp_bump)
};
struct CpBackArgs {
  const double* Lt; size_t lt_stride; const int* slot_of; int n, ntc, nsys, ncols;
  const int* far_start; const int* far_slot; const int* far_row;      // per block column: its far tiles, rows descending
  const int* back_tab;            // [ntc][CPB_INTS] (below)
  double* xbuf; double* fbuf; int vec_stride; double* xout; size_t sys_stride; int* err; int* epoch; int* fail; int* claim;
  // chain workgroups of the back-substitution: [0] walks the last chain of the plan (the separator) and on through the chain before it,
  // the others a chain each (their columns couple to the separator's through far tiles only): columns [ch_lo, ch_hi), right to left
  int nchain; int ch_lo[CP_MAX_SEG]; int ch_hi[CP_MAX_SEG];
};

#ifdef MCP_CP_PROF
#define MCP_CP_PROF_ARRIVE (MCP_CP_PROF == 5)
// phase stamps (100 MHz wall clock, the same on every compute unit) of system 0: critical workgroup [step + 1][16], helpers [index][4]
__device__ unsigned long long g_cp_prof[256*16];
__device__ unsigned long long g_cp_hprof[8192*4];
__device__ unsigned int g_cp_miss[8];          // [0] band row asked again, [1] late product asked again (system 0, last launch)
#define CP_MISS(i) do { if (q == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_cp_miss[i], 1u); } while (0)
// MCP_CP_PROF = 1: everything; 2: critical workgroup only; 3: step start / end only (0x101 mask); 4: wavefront 0's stamps only
#define CP_STAMP_ON(i) ((MCP_CP_PROF == 5 && ((i) == 0 || (i) == 2 || (i) == 8 || (i) == 3 || (i) == 12 || (i) == 11 || (i) == 21 || (i) == 22 || (i) == 23 || ((i) >= 16 && (i) <= 19))) || MCP_CP_PROF == 1 || MCP_CP_PROF == 2 || (MCP_CP_PROF == 3 && ((i) == 0 || (i) == 8)) || (MCP_CP_PROF == 4 && ((i) <= 2 || ((i) >= 8 && (i) <= 10) || (i) >= 14)))
#define CP_STAMP(step, i) do { if (CP_STAMP_ON(i) && q == 0 && (step) + 1 < 256) g_cp_prof[((step) + 1)*16 + ((i) >= 21 ? (i) - 8 : (i) >= 16 ? (i) - 12 : (i))] = wall_clock64(); } while (0)      // (16..19: a wavefront's arrival at the step's last barrier, kept in slots 4..7)
#define CP_HSTAMP(i) do { if (MCP_CP_PROF == 1 && q == 0 && hidx < 8192) g_cp_hprof[hidx*4 + (i)] = wall_clock64(); } while (0)
#else
#define CP_STAMP(step, i) do {} while (0)
#define CP_HSTAMP(i) do {} while (0)
#define CP_MISS(i) do {} while (0)
#define MCP_CP_PROF_ARRIVE 0
#endif
typedef unsigned int cp_u4 __attribute__((ext_vector_type(4)));
typedef double (*cp_tile)[CP_LD];

// ---- write-through / L1-bypassing 16-byte accesses (buffer instructions with sc1) ------------------------------------------
__device__ inline __amdgpu_buffer_rsrc_t cp_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ inline chol_d4 cp_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const cp_u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
  const cp_u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 16);
  const chol_d2 x = __builtin_bit_cast(chol_d2, a), y = __builtin_bit_cast(chol_d2, b);
  chol_d4 v = {x[0], x[1], y[0], y[1]};
  return v;
}
__device__ inline void cp_st4(__amdgpu_buffer_rsrc_t r, unsigned off, const chol_d4& v) {
  const chol_d2 x = {v[0], v[1]}, y = {v[2], v[3]};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_u4, x), r, off, 0, 16);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_u4, y), r, off + 16, 0, 16);
}
#define CP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// Data-tagged chunks (round 6): a band tile travels from its helper to the critical workgroup WITHOUT a flag -- the helper stores the
// chunks and is done (no drain, no flag store), the consumer loads them and looks at the data itself: a chunk is two untorn 16-byte
// stores, so "first double of either half is the sentinel" = not there yet.  Band slots exist twice, a copy per parity of the launch
// count (epoch): a launch's helpers write -- and its critical workgroup reads -- the copy of its parity, and the helper that fills a
// slot also stores the sentinel into the slot's OTHER copy, which the previous launch is done with and the next one will poll (the
// critical workgroup armed its own slots at first: 16 more stores per wavefront and step on the one in-order memory counter of the
// wavefronts whose flags everybody waits for).  Both copies are armed when a plan is built, and a launch that ends in a time-out is
// the last one of its plan (the handle falls back to the per-step kernels for good).  One hop = one round
// trip (store -> visible -> load) instead of drain + flag store + flag poll + payload load: ~0.8 us instead of ~2.2 us.
__device__ inline bool cp_chunk_missing(const chol_d4& v) {
  return (unsigned long long)__double_as_longlong(v[0]) == CP_SENT || (unsigned long long)__double_as_longlong(v[2]) == CP_SENT;
}
__device__ inline void cp_chunk_arm(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const cp_u4 x = {CP_SENT32, CP_SENT32, CP_SENT32, CP_SENT32};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, 16);
  __builtin_amdgcn_raw_buffer_store_b128(x, r, off + 16, 0, 16);
}
// wavefront-uniform: does any lane of this wavefront still miss (part of) one of its chunks?
__device__ inline bool cp_wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
__device__ inline int cp_flag_load(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void cp_flag_store(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a flag load that is ISSUED here (the compiler may not sink it to its use) and waited for by cp_flag_wait: the round trip to the
// memory side hides behind whatever is done in between
__device__ inline int cp_flag_load_early(const int* f) {
  int v;
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(f) : "memory");
  return v;
}
__device__ inline void cp_flag_wait(int& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory"); }
// one lane: spin until *f == want; false = gave up (this spin timed out, or another workgroup of the system raised the error word)
__device__ inline bool cp_poll(const int* f, int want, int* err, int code, bool patient) {
  long long t0 = 0;
  for (unsigned it = 0;; ++it) {
    if (cp_flag_load(f) == want) return true;
    if ((it & 31) == 31) {
      if (cp_flag_load(err) != 0) return false;
      if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) break;
    }
    if (patient) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
  }
  cp_flag_store(err, code);
  return false;
}

// one lane: up to three flags polled together (their loads in flight at once: a poll is a round trip to the memory side)
__device__ inline bool cp_poll3(const int* f0, const int* f1, const int* f2, int want, int* err, int code) {
  long long t0 = 0;
  for (unsigned it = 0;; ++it) {
    const int a = cp_flag_load(f0), b = cp_flag_load(f1), c = cp_flag_load(f2);
    if (a == want && b == want && c == want) return true;
    if ((it & 31) == 31) {
      if (cp_flag_load(err) != 0) return false;
      if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  cp_flag_store(err, code);
  return false;
}

// ---- a 32 x 32 tile as four 16 x 16 quadrants in the C/D layout of v_mfma_f64_16x16x4: quadrant qd, lane l, register g
//      <-> row 16 (qd >> 1) + (l >> 4) + 4 g, column 16 (qd & 1) + (l & 15); in global memory chunk (qd, l) is 32 contiguous bytes
__device__ inline void cp_regs_to_lds(cp_tile T, int qd, int l, const chol_d4& v) {
  const int r = 16*(qd >> 1) + (l >> 4), c = 16*(qd & 1) + (l & 15);
#pragma unroll
  for (int g = 0; g < 4; ++g) T[r + 4*g][c] = v[g];
}
__device__ inline chol_d4 cp_lds_to_regs(cp_tile T, int qd, int l) {
  const int r = 16*(qd >> 1) + (l >> 4), c = 16*(qd & 1) + (l & 15);
  chol_d4 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) v[g] = T[r + 4*g][c];
  return v;
}
// acc (+/-)= Pi[rows of the quadrant] Pj[columns of the quadrant]^T over K = 32.  All 2 KMAX/4 operands are asked for BEFORE the first
// matrix instruction (round 6): left to itself the compiler reused two operand registers per pair of instructions, so every pair waited
// for its own LDS round trip -- 8 instructions took ~920 cycles instead of ~512 + one round trip.
template <bool NEG, int KMAX = CH_NB>
__device__ inline void cp_mma(chol_d4& acc, cp_tile Pi, cp_tile Pj, int qd, int l) {
  const int ri = 16*(qd >> 1) + (l & 15), rj = 16*(qd & 1) + (l & 15), rq = l >> 4;
  double x[KMAX/4], y[KMAX/4];
#pragma unroll
  for (int i = 0; i < KMAX/4; ++i) { x[i] = Pi[ri][4*i + rq]; y[i] = Pj[rj][4*i + rq]; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < KMAX/4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -x[i] : x[i], y[i], acc, 0, 0, 0);
}
__device__ inline unsigned cp_chunk_off(int slot, int qd, int l) { return (unsigned)(((size_t)slot*CP_TQ + (size_t)(qd*64 + l)*4)*sizeof(double)); }

// A(ti, tj) out of S (row-major n x n, right-hand side = row n); block row R = ntc is the right-hand side (one row).
// Beyond the matrix a diagonal tile continues as the identity (the panel then needs no notion of the matrix edge), others as zero.
__device__ inline chol_d4 cp_load_A(const double* __restrict__ S, int n, int ntc, int ti, int tj, int qd, int l) {
  chol_d4 v = {0.0, 0.0, 0.0, 0.0};
  const int rl = 16*(qd >> 1) + (l >> 4), c = tj*CH_NB + 16*(qd & 1) + (l & 15);
  if (ti == ntc) { if (rl == 0 && c < n) v[0] = S[(size_t)n*n + c]; return v; }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = ti*CH_NB + rl + 4*g;
    if (r < n && c < n) v[g] = S[(size_t)r*n + c];
    else if (r == c) v[g] = 1.0;
  }
  return v;
}

// ---- the critical workgroup ------------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS only: __syncthreads() also drains the vector memory counter, i.e. would wait for the band row
// the team has just asked for (fetched ahead precisely so that nobody waits for it)
__device__ inline void cp_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// LDS-only rendezvous of the three team wavefronts (the hardware barrier would include wavefront 0, which is inside the panel)
__device__ inline void cp_team_sync(int* ctr, int& target, int lane) {
  target += 3;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// X = T Dinv^T for one quadrant; Dinv = L^-1 is lower triangular, so the left column quadrants only see k < 16
__device__ inline chol_d4 cp_trsm_quadrant(cp_tile T, cp_tile Dinv, int qd, int l) {
  chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (qd & 1) cp_mma<false>(acc, T, Dinv, qd, l); else cp_mma<false, 16>(acc, T, Dinv, qd, l);
  return acc;
}
__device__ inline void cp_pair_sync(int* ctr, int& target, int lane) {      // the same for two wavefronts
  target += 2;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// ---- the split panel (round 6) ---------------------------------------------------------------------------------------------
// Round 5's panel cost wavefront 0 ~42 instructions per pivot of which ~22 were the bulk rank-1 updates of the columns >= J + 3 (one
// fp64 multiply-add and half a broadcast read per column, ~2.3 ns each for a lone wavefront): 3.0 of the step's 4.9 us.  Now the
// columns >= CP_SPLIT are NOT wavefront 0's during the first CP_SPLIT pivots: it writes each finished column J < CP_SPLIT (all 64 lanes:
// the tile's rows and the identity's) to LDS -- as it always did, but each to its own place, Lcol[J][lane] -- and a FOLLOWER (wavefront
// 1, holding the same 64 rows' columns CP_SPLIT .. 31 in registers) applies it to those columns as it appears, in the same order and
// with the same multiply-adds wavefront 0 would have issued: every entry's arithmetic is unchanged, bit for bit.  After pivot
// CP_SPLIT - 1 the follower hands its 16 columns over through LDS and wavefront 0 goes on with a 16-column panel.  Wavefront 0's bulk:
// 435 -> 182 multiply-adds; the price is the hand-over on the chain (follower's last column, its 16 stores, wavefront 0's 16 loads).
// Lcol lives in the LDS tile that will take L^-1 at the panel's END (free until then); the hand-over buffer aliases Lcol.
constexpr int CP_SPLIT = 16;
#ifndef CP_SWAP
#define CP_SWAP 1      // the follower finishes the panel (no hand-over), wavefront 0 takes its product of the step
#endif
#define CP_SB() __builtin_amdgcn_sched_barrier(0)
template <int K, int G> struct CpBulkA {       // first half: group G (of 6) of the bulk update by column K on columns [K + 3, CP_SPLIT)
  static constexpr int n = (CP_SPLIT - K - 3 > 0) ? CP_SPLIT - K - 3 : 0;
  static constexpr int lo = K + 3 + (n*G)/6, hi = K + 3 + (n*(G + 1))/6;
  static __device__ inline void fm(double* d, const double* m) {
#pragma unroll
    for (int c = lo; c < hi; ++c) d[c] -= d[K]*m[c];
  }
};
// the rsqrt of the next pivot: v_rsq_f64 + one third-order correction (the sequence of ba_chol.h's panel, so that the numbers are its)
__device__ inline double cp_rsq3(double pn) {
  const double y0 = __builtin_amdgcn_rsq(pn);
  const double t = y0*(-pn);
  const double e = __builtin_fma(t, y0, 1.0);
  const double u = y0*e;
  const double q = __builtin_fma(e, 0.375, 0.5);
  return __builtin_fma(u, q, y0);
}
// pivots 0 .. CP_SPLIT - 1 on wavefront 0 (ba_chol.h's chol_panel_pivot restricted to the columns < CP_SPLIT)
template <int J>
__device__ inline void cp_pivot_a(double* d, double& inv, bool& bad, double* mE, double* mO, double* lcol /* Lcol + lane */, const double* lrow /* Lcol */, int* prog, int pbase) {
  double* cur = (J & 1) ? mO : mE;
  double* prev = (J & 1) ? mE : mO;
  d[J] *= inv;
  lcol[J*64] = d[J];
  __hip_atomic_store(prog + threadIdx.x, pbase + J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ("column J is there": LDS takes a wavefront's operations in order.  Every lane its OWN word -- 64 lanes storing to one word is a 64-way bank conflict that stalls the whole compute unit's LDS; the follower reads word 0)
  CP_SB();
  if constexpr (J + 1 < CP_SPLIT) {
    const double w = __builtin_fma(-d[J], d[J], d[J + 1]);
    const double l1 = readlane_f64(d[J], J + 1);
    double l2 = 0.0;
    if constexpr (J + 2 < CP_SPLIT) l2 = readlane_f64(d[J], J + 2);
    const double pn = readlane_f64(w, J + 1);
    bad |= !(pn > 0.0);
    if constexpr (J + 3 < CP_SPLIT) {
#pragma unroll
      for (int c = J + 3; c < CP_SPLIT; ++c) cur[c] = lrow[J*64 + c];
    }
    CP_SB();
    const double y0 = __builtin_amdgcn_rsq(pn);
    CP_SB(); if constexpr (J >= 1) CpBulkA<J - 1, 0>::fm(d, prev); CP_SB();
    const double t = y0*(-pn);
    CP_SB(); if constexpr (J >= 1) CpBulkA<J - 1, 1>::fm(d, prev); CP_SB();
    const double e = __builtin_fma(t, y0, 1.0);
    CP_SB(); if constexpr (J >= 1) CpBulkA<J - 1, 2>::fm(d, prev); CP_SB();
    const double u = y0*e;
    const double q = __builtin_fma(e, 0.375, 0.5);
    CP_SB(); if constexpr (J >= 1) CpBulkA<J - 1, 3>::fm(d, prev); CP_SB();
    inv = __builtin_fma(u, q, y0);
    CP_SB(); if constexpr (J >= 1) CpBulkA<J - 1, 4>::fm(d, prev); CP_SB();
    d[J + 1] -= d[J]*l1;
    if constexpr (J >= 1) CpBulkA<J - 1, 5>::fm(d, prev);
    CP_SB();
    if constexpr (J + 2 < CP_SPLIT) d[J + 2] -= d[J]*l2;
    CP_SB();
  }
}
// pivots CP_SPLIT .. 31 on wavefront 0: ba_chol.h's chol_panel_pivot, but column CP_SPLIT - 1's bulk update is not wavefront 0's
template <int J>
__device__ inline void cp_pivot_b(double* d, double& inv, bool& bad, double* mE, double* mO, double* colbuf, int lane) {
  double* cur = (J & 1) ? mO : mE;
  double* prev = (J & 1) ? mE : mO;
  constexpr bool BULK = J >= CP_SPLIT + 1;
  d[J] *= inv;
  if constexpr (J + 3 < CH_NB) colbuf[lane] = d[J];
  CP_SB();
  if constexpr (J + 1 < CH_NB) {
    const double w = __builtin_fma(-d[J], d[J], d[J + 1]);
    const double l1 = readlane_f64(d[J], J + 1);
    double l2 = 0.0;
    if constexpr (J + 2 < CH_NB) l2 = readlane_f64(d[J], J + 2);
    const double pn = readlane_f64(w, J + 1);
    bad |= !(pn > 0.0);
    if constexpr (J + 3 < CH_NB) {
#pragma unroll
      for (int c = J + 3; c < CH_NB; ++c) cur[c] = colbuf[c];
    }
    CP_SB();
    const double y0 = __builtin_amdgcn_rsq(pn);
    CP_SB(); if constexpr (BULK) ChBulk<J - 1, 0>::fm(d, prev); CP_SB();
    const double t = y0*(-pn);
    CP_SB(); if constexpr (BULK) ChBulk<J - 1, 1>::fm(d, prev); CP_SB();
    const double e = __builtin_fma(t, y0, 1.0);
    CP_SB(); if constexpr (BULK) ChBulk<J - 1, 2>::fm(d, prev); CP_SB();
    const double u = y0*e;
    const double q = __builtin_fma(e, 0.375, 0.5);
    CP_SB(); if constexpr (BULK) ChBulk<J - 1, 3>::fm(d, prev); CP_SB();
    inv = __builtin_fma(u, q, y0);
    CP_SB(); if constexpr (BULK) ChBulk<J - 1, 4>::fm(d, prev); CP_SB();
    d[J + 1] -= d[J]*l1;
    if constexpr (BULK) ChBulk<J - 1, 5>::fm(d, prev);
    CP_SB();
    if constexpr (J + 2 < CH_NB) d[J + 2] -= d[J]*l2;
    CP_SB();
  }
}
#undef CP_SB
template <int... Js>
__device__ inline void cp_pivots_a(double* d, double& inv, bool& bad, double* mE, double* mO, double* lcol, const double* lrow, int* prog, int pbase, std::integer_sequence<int, Js...>) {
  (cp_pivot_a<Js>(d, inv, bad, mE, mO, lcol, lrow, prog, pbase), ...);
}
template <int... Js>
__device__ inline void cp_pivots_b(double* d, double& inv, bool& bad, double* mE, double* mO, double* colbuf, int lane, std::integer_sequence<int, Js...>) {
  (cp_pivot_b<CP_SPLIT + Js>(d, inv, bad, mE, mO, colbuf, lane), ...);
}
__device__ inline int cp_lds_word(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wavefront 0: L L^T = D (a diagonal tile arrives with the identity beyond the matrix), L^-1 -> Dinv (row-major): lanes 0..31 carry
// the tile's rows, lanes 32..63 the identity through the same column operations.  Dinv doubles as Lcol / the hand-over buffer while the
// panel runs; ctlw[0..63] = the columns published so far (monotonic over the steps: 16 k + J + 1; a word per lane), ctlw[64..127] = the follower's hand-overs.
__device__ __forceinline__ void cp_potrf(cp_tile D, cp_tile Dinv, double* colbuf, const double* zvec, int* ctlw, int k, int* fail, int q, int s) {
  const int lane = threadIdx.x;
  int rr = lane & 31;
  asm volatile("" : "+v"(rr));       // opaque per call
  const bool low = lane >= 32;
  double d[CH_NB];
  // lanes 0..31 read their row of the tile, lanes 32..63 their row of the identity -- the SAME loads with another base address:
  // zvec[0..62] is zero but for zvec[31] = 1, so zvec + 31 - r is row r of the identity.  (Round 5 loaded the tile on every lane and
  // swapped the identity in with a compare and a select per entry: 130 instructions, 0.35 us of the 0.52 us this entry took.)
  const double* src = low ? zvec + (CH_NB - 1 - rr) : &D[rr][0];
#pragma unroll
  for (int c = 0; c < CP_SPLIT; ++c) d[c] = src[c];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double* Lcol = &Dinv[0][0];
  const double piv0 = readlane_f64(d[0], 0);
  bool bad = !(piv0 > 0.0);
  double inv = rsqrt(piv0);
  if (lane == 0) CP_STAMP(s, 9);
  double mE[CH_NB], mO[CH_NB];
  cp_pivots_a(d, inv, bad, mE, mO, Lcol + lane, Lcol, ctlw, CP_SPLIT*k, std::make_integer_sequence<int, CP_SPLIT>());
  if (lane == 0) CP_STAMP(s, 14);
#if CP_SWAP
  // The follower goes on with the pivots CP_SPLIT .. 31 itself (cp_follow): the columns they work on are the ones it holds.  Wavefront
  // 0's part of L^-1 -- its 16 columns of the identity's rows -- goes out once the follower has read the last published column (Lcol
  // and L^-1 share the tile), off the chain.
  while (cp_lds_word(ctlw + 64) != k + 1) __builtin_amdgcn_s_sleep(1);
  if (bad && lane == 0) atomicOr(fail, 2);
  if (low) {
#pragma unroll
    for (int c = 0; c < CP_SPLIT; ++c) Dinv[c][rr] = d[c];
  }
  return;
#endif
  // the follower's columns
  while (cp_lds_word(ctlw + 64) != k + 1) __builtin_amdgcn_s_sleep(0);
#pragma unroll
  for (int c = 0; c < CH_NB - CP_SPLIT; ++c) d[CP_SPLIT + c] = Lcol[c*64 + lane];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const double pn = readlane_f64(d[CP_SPLIT], CP_SPLIT);
    bad |= !(pn > 0.0);
    inv = cp_rsq3(pn);
  }
  if (lane == 0) CP_STAMP(s, 15);
  cp_pivots_b(d, inv, bad, mE, mO, colbuf, lane, std::make_integer_sequence<int, CH_NB - CP_SPLIT>());
  if (lane == 0) CP_STAMP(s, 10);
  if (bad && lane == 0) atomicOr(fail, 2);
  // lane 32 + r ends with row r of L^-T = column r of L^-1.  Its entries left of the diagonal ARE zero (0 - 0 m = 0 through every
  // column operation), so the row leaves as it is (round 5 selected 0.0 per entry: 100 more instructions on the critical wavefront).
  if (low) {
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) Dinv[c][rr] = d[c];
  }
}
// the follower (wavefront 1): columns CP_SPLIT .. 31 of the same 64 rows; applies the columns wavefront 0 publishes, hands over.
// Column K's words are asked for together with the progress word (read FIRST: LDS serves a wavefront in order, so a progress value
// that says "column K is there" vouches for the data read behind it), one column ahead of the multiply-adds; scheduling barriers keep
// the compiler from asking for ALL the columns at once (it did: 400 spills, a hand-over of 10 us).
constexpr int CP_NC = CH_NB - CP_SPLIT;
struct CpCol { int pw; double own; double m[CP_NC]; };
__device__ __forceinline__ void cp_follow_ask(CpCol& c, const double* Lcol, const int* prog, int K, int lane) {
  c.pw = cp_lds_word(prog);
  c.own = Lcol[K*64 + lane];
#pragma unroll
  for (int i = 0; i < CP_NC; ++i) c.m[i] = Lcol[K*64 + CP_SPLIT + i];
}
template <int K>
__device__ __forceinline__ void cp_follow_col(double* dh, CpCol& cur, CpCol& nxt, const double* Lcol, const int* prog, int pbase, int lane) {
  while (cur.pw - pbase <= K) { __builtin_amdgcn_s_sleep(0); cp_follow_ask(cur, Lcol, prog, K, lane); }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K + 1 < CP_SPLIT) cp_follow_ask(nxt, Lcol, prog, K + 1, lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < CP_NC; ++i) dh[i] -= cur.own*cur.m[i];
  // (pins the multiply-adds HERE: their results are only read at the hand-over, and the compiler sank all 256 of them behind the
  //  last column's poll loop, keeping sixteen columns of multipliers alive in scratch)
#pragma unroll
  for (int i = 0; i < CP_NC; ++i) asm volatile("" : "+v"(dh[i]));
  __builtin_amdgcn_sched_barrier(0);
}
template <int... Ks>
__device__ __forceinline__ void cp_follow_cols(double* dh, CpCol& ca, CpCol& cb, const double* Lcol, const int* prog, int pbase, int lane, std::integer_sequence<int, Ks...>) {
  ((Ks & 1 ? cp_follow_col<Ks>(dh, cb, ca, Lcol, prog, pbase, lane) : cp_follow_col<Ks>(dh, ca, cb, Lcol, prog, pbase, lane)), ...);
}
__device__ __forceinline__ void cp_follow(cp_tile D, cp_tile Dinv, const double* zvec, int* ctlw, int k, double* colbuf, int* fail, int q, int s) {
  const int lane = threadIdx.x & 63;
  const int rr = lane & 31;
  const bool low = lane >= 32;
#if CP_SWAP
  double d[CH_NB];                       // (columns CP_SPLIT .. 31 only: the lower half is never touched)
  double* dh = d + CP_SPLIT;
#else
  double dh[CP_NC];
#endif
  const double* src = low ? zvec + (CH_NB - 1 - rr) : &D[rr][0];
#pragma unroll
  for (int c = 0; c < CP_NC; ++c) dh[c] = src[CP_SPLIT + c];
  double* Lcol = &Dinv[0][0];
  CpCol ca, cb;
  cp_follow_ask(ca, Lcol, ctlw, 0, lane);
  __builtin_amdgcn_sched_barrier(0);
  cp_follow_cols(dh, ca, cb, Lcol, ctlw, CP_SPLIT*k, lane, std::make_integer_sequence<int, CP_SPLIT>());
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if CP_SWAP
  // Round 6, second half: the follower does NOT hand its 16 columns back (16 stores, a flag, wavefront 0's poll and 16 loads: 0.56 us on
  // the chain of every step) -- it IS the panel from here on: pivots CP_SPLIT .. 31 touch no other column.  Same instructions on the
  // same numbers as wavefront 0 would have issued.  Wavefront 0 is told that the last published column has been read (it may now
  // overwrite Lcol with its part of L^-1) and takes over this wavefront's product of the step (cp_critical).
  __hip_atomic_store(ctlw + 64 + lane, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (a word per lane, as the progress words)
  bool bad = false;
  double inv;
  {
    const double pn = readlane_f64(d[CP_SPLIT], CP_SPLIT);
    bad |= !(pn > 0.0);
    inv = cp_rsq3(pn);
  }
  if (lane == 0) CP_STAMP(s, 15);
  double mE[CH_NB], mO[CH_NB];
  cp_pivots_b(d, inv, bad, mE, mO, colbuf, lane, std::make_integer_sequence<int, CH_NB - CP_SPLIT>());
  if (lane == 0) CP_STAMP(s, 10);
  if (bad && lane == 0) atomicOr(fail, 2);
  if (low) {
#pragma unroll
    for (int c = CP_SPLIT; c < CH_NB; ++c) Dinv[c][rr] = d[c];
  }
#else
#pragma unroll
  for (int c = 0; c < CP_NC; ++c) Lcol[c*64 + lane] = dh[c];
  __hip_atomic_store(ctlw + 64 + lane, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (a word per lane, as the progress words)
#endif
}

template <bool SEG>        // (false: one chain from column 0 to the right-hand side -- the constants fold as they did before there were segments)
__device__ inline void cp_critical(const CpArgs& a, int q, double* lds, int seg) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // This workgroup's chain: block columns [c0, ntc).  The LAST segment ends with the right-hand-side row (row a.ntc) as the matrix does;
  // any other ends where its columns end -- nothing below them is this chain's (what couples to them are far tiles: helpers' work) --
  // i.e. it is walked as a matrix of `ntc` block columns without a right-hand side whose last row is R = ntc - 1.
  const int c0 = SEG ? a.seg_start[seg] : 0;
  const bool last_seg = SEG ? seg + 1 == a.nseg : true;
  const int ntc = SEG ? a.seg_start[seg + 1] : a.ntc, R = last_seg ? a.ntc : ntc - 1;
  const int ntc_all = a.ntc;
  // LDS tiles (pointers computed, never kept in an indexed array: that would live in scratch)
  auto Dv = [&](int i) { return (cp_tile)(lds + (i & 1)*CP_TILE); };             // L_kk^-1 of block k in Dv(k)
  const cp_tile Xb0 = (cp_tile)(lds + 2*CP_TILE), Xb1 = (cp_tile)(lds + 3*CP_TILE);
  const cp_tile Tc = (cp_tile)(lds + 4*CP_TILE), T2 = (cp_tile)(lds + 5*CP_TILE);
  auto Dt = [&](int i) { return (cp_tile)(lds + (6 + (i & 1))*CP_TILE); };
  double* colbuf = lds + 8*CP_TILE;
  double* zvec = colbuf + 64;                // [63]: rows of the identity for the panel's upper lanes (cp_potrf)
  int* pwords = (int*)(zvec + 64);           // [128]: the split panel's progress / hand-over words
  int* ctl = pwords + 128;                   // [0] team counter, [1] ok / abort word
  int* flags = a.flags + (size_t)q*a.nflags;
  int* err = a.err + q; int* fail = a.fail + q;
  const __amdgpu_buffer_rsrc_t rL = cp_rsrc(a.Lt + q*a.lt_stride, a.lt_stride*sizeof(double));
  const __amdgpu_buffer_rsrc_t rB = cp_rsrc(a.Bt + q*a.bt_stride + (size_t)(a.epoch[q] & 1)*(a.bt_stride/2), (a.bt_stride/2)*sizeof(double));      // (this launch's copy of the band slots)
  const int* slot_of = a.slot_of; const int* bslot_of = a.bslot_of;
  const int epoch4 = a.epoch[q] << 2, done_l = epoch4 | 2;
  const int code = 0x100;
  int* stp = ctl + 16;                       // the step table
  for (int i = t; i < (ntc_all + 1)*CP_STEP_INTS; i += CP_THREADS) stp[i] = a.steps[i];
  if (t == 0) { ctl[0] = 0; ctl[1] = 1; ctl[2] = 0; }
  if (t < 128) pwords[t] = 0;
  if (t < 64) zvec[t] = (t == CH_NB - 1) ? 1.0 : 0.0;
  __syncthreads();
  // ---- prologue: rows 0 and 1 of the band (their helpers pass A through) arrive as data-tagged chunks: wavefront w takes quadrant w of
  //      tiles (0,0), (1,0) and (1,1)
  const bool row1_diag = c0 + 1 < ntc;
  {
    const unsigned o0 = cp_chunk_off(bslot_of[(size_t)c0*ntc_all + c0], wave, lane), o1 = cp_chunk_off(bslot_of[(size_t)(c0 + 1)*ntc_all + c0], wave, lane);
    const unsigned o2 = row1_diag ? cp_chunk_off(bslot_of[(size_t)(c0 + 1)*ntc_all + c0 + 1], wave, lane) : o1;
    chol_d4 p0, p1, p2;
    unsigned it = 0; long long t0 = 0;
    for (;;) {
      p0 = cp_ld4(rB, o0); p1 = cp_ld4(rB, o1); p2 = cp_ld4(rB, o2);
      if (!cp_wave_any(cp_chunk_missing(p0) || cp_chunk_missing(p1) || cp_chunk_missing(p2))) break;
      if ((++it & 15) == 15) {
        if (cp_flag_load(err) != 0) { ctl[1] = 0; break; }
        if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { if (lane == 0) cp_flag_store(err, code | 1); ctl[1] = 0; break; }
      }
      __builtin_amdgcn_s_sleep(2);
    }
    cp_regs_to_lds(Dt(0), wave, lane, p0); cp_regs_to_lds(Tc, wave, lane, p1);
    if (row1_diag) cp_regs_to_lds(Dt(1), wave, lane, p2);
  }
  __syncthreads();
  if (!ctl[1]) return;
  // Two loops, one per role, with the same barriers: what the team carries from step to step (the band row fetched ahead) and what
  // the panel keeps in registers never share a live range that way (one loop with a branch per role spilled 63 registers).
  if (wave == 0) {
    int dcur = 0;               // Dt(dcur): diagonal tile (s+1, s+1), updated through column s - 1
    int pair_target0 = 0;
    // s = -1 is the lead-in: only the factorisation of diagonal tile 0 (the one call site of the panel code)
    for (int s = c0 - 1; s < ntc; ++s) {
      const int i1 = s + 1;
      const cp_tile Ds = Dv(s);
      if (t == 0) CP_STAMP(s, 0);
      if (s >= c0) {
        cp_regs_to_lds(Xb0, 0, lane, cp_trsm_quadrant(Tc, Ds, 0, lane));      // P1, quadrant 0
        cp_barrier();
        if (i1 < ntc) {                                                          // P2, quadrant 0
          chol_d4 acc = cp_lds_to_regs(Dt(dcur), 0, lane);
          cp_mma<true>(acc, Xb0, Xb0, 0, lane);
          cp_regs_to_lds(Dt(dcur), 0, lane, acc);
        }
        cp_barrier();
      }
      if (t == 0) CP_STAMP(s, 1);
      if (i1 < ntc) cp_potrf(Dt(dcur), Dv(i1), colbuf, zvec, pwords, i1, fail, q, s);
      if (t == 0) CP_STAMP(s, 2);
#if CP_SWAP
      if (s >= c0 && s + 2 <= R && ctl[1]) {
        // A'(s+2, s+2) -= X2 X2^T (lower triangle's quadrants: the panel never reads (0, 1)) once wavefronts 2, 3 have completed the solve:
        // wavefront 1's product until it became the panel's second half
        while (__hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < pair_target0 + 2 && ctl[1]) __builtin_amdgcn_s_sleep(1);
        pair_target0 += 4;               // (the pair meets twice per step: after its solve and before L(s+2, s)'s flag)
        asm volatile("" ::: "memory");
        if (s + 2 < ntc && ctl[1]) {
          chol_d4 a0 = cp_lds_to_regs(Dt(dcur ^ 1), 0, lane), a2 = cp_lds_to_regs(Dt(dcur ^ 1), 2, lane), a3 = cp_lds_to_regs(Dt(dcur ^ 1), 3, lane);
          cp_mma<true>(a0, Xb1, Xb1, 0, lane); cp_mma<true>(a2, Xb1, Xb1, 2, lane); cp_mma<true>(a3, Xb1, Xb1, 3, lane);
          cp_regs_to_lds(Dt(dcur ^ 1), 0, lane, a0); cp_regs_to_lds(Dt(dcur ^ 1), 2, lane, a2); cp_regs_to_lds(Dt(dcur ^ 1), 3, lane, a3);
        }
      }
#endif
      cp_barrier();
      if (t == 0) CP_STAMP(s, 8);
      if (!ctl[1]) return;
      dcur ^= 1;
    }
    return;
  }
  int dcur = 0;
  int team_target = 0, pair_target = 0;
  // The loop body starts where the team asks for the band row it needs a step later and ends where it has used it: the request and
  // its use sit in ONE iteration (carried around the loop the compiler waited for the loads at the loop head, i.e. hid nothing).
  // `s` is the step that is ending (-1: the lead-in, wavefront 0 factors diagonal tile 0).
  for (int s = c0 - 1; s < ntc; ++s) {
    chol_d4 v0 = {0.0, 0.0, 0.0, 0.0}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0;       // wavefronts 2, 3: band row s+3
    unsigned ob0 = 0, ob1 = 0, ob2 = 0;
    if (wave >= 2 && s + 3 <= R && ctl[1]) {
      // the row after next into registers (wavefronts 2, 3; also in the lead-in step), as late in the step as can be and WITHOUT asking
      // first: the chunks are data-tagged, whether they had arrived is looked at where they are used (P3 of the next step)
      const int* se = stp + (s + 1)*CP_STEP_INTS;
      const int h = wave - 2;                    // 0: quadrants 0, 1   1: quadrants 2, 3
      ob0 = cp_chunk_off(se[CPS_B0], 2*h, lane); ob1 = cp_chunk_off(se[CPS_B1], 2*h, lane); ob2 = cp_chunk_off(se[CPS_B2], 2*h, lane);
      v0 = cp_ld4(rB, ob0); v1 = cp_ld4(rB, ob0 + 2048);
      v2 = cp_ld4(rB, ob1); v3 = cp_ld4(rB, ob1 + 2048);
      v4 = cp_ld4(rB, ob2); v5 = cp_ld4(rB, ob2 + 2048);
      if (t == 128) CP_STAMP(s, 4);
    }
    if (s < c0 && wave == 1) cp_follow(Dt(0), Dv(c0), zvec, pwords, c0, colbuf, fail, q, s);      // (the lead-in: diagonal tile 0's panel has its follower too)
    if (MCP_CP_PROF_ARRIVE && lane == 0) CP_STAMP(s, 16 + wave);
    cp_barrier();                      // end of step s
    if (!ctl[1]) return;
    dcur ^= 1;
    if (s + 1 >= ntc) break;
    // ---------------- step s + 1 ----------------
    const int sn = s + 1;
    {
      const int s = sn;               // (the body below was written in terms of the step it works on)
      const int i1 = s + 1, i2 = s + 2;
      const cp_tile Ds = Dv(s);
      const int tw = wave - 1;
      const bool row2 = i2 <= R, diag2 = i2 < ntc;
      const int* se = stp + (s + 1)*CP_STEP_INTS;
      const int sd = se[CPS_SD], s1 = se[CPS_S1];
      // ---- L_ss^-1 leaves the moment the step begins (round 6; round 5 published it from P3, 1.9 us into the step): it has been in LDS
      //      since the panel ended, and the helpers of band row s+3 -- the longest chain that hangs on this step -- start from its flag.
      //      Wavefront 1 stores quadrants (0,0) and (0,1) (zeros; the back-substitution reads the whole tile), 2 and 3 the lower ones.
      if (tw == 0) { cp_st4(rL, cp_chunk_off(sd, 0, lane), cp_lds_to_regs(Ds, 0, lane)); cp_st4(rL, cp_chunk_off(sd, 1, lane), cp_lds_to_regs(Ds, 1, lane)); }
      else cp_st4(rL, cp_chunk_off(sd, wave, lane), cp_lds_to_regs(Ds, wave, lane));
      // ---- P1: X1 = A'(s+1, s) L_ss^-T (all four wavefronts, a quadrant each)
      cp_regs_to_lds(Xb0, wave, lane, cp_trsm_quadrant(Tc, Ds, wave, lane));
      cp_barrier();
      // ---- P2: A'(s+1, s+1) -= X1 X1^T, quadrants 0 | 3 | 2 on wavefronts 0, 1, 2 (the panel never reads quadrant (0, 1)); wavefront 3 sends L(s+1, s) off instead
      if (wave == 3) {
        if (i1 <= R) {                    // (a chain that is not the last ends without a row below its last column)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) cp_st4(rL, cp_chunk_off(s1, qd, lane), cp_lds_to_regs(Xb0, qd, lane));
        }
      } else if (i1 < ntc) {
        const int qd = wave == 1 ? 3 : wave;
        chol_d4 acc = cp_lds_to_regs(Dt(dcur), qd, lane);
        cp_mma<true>(acc, Xb0, Xb0, qd, lane);
        cp_regs_to_lds(Dt(dcur), qd, lane, acc);
      }
      cp_barrier();
      // ---- P3: wavefront 0 factors the next diagonal tile; the team finishes column s
      const int dl = (row2 && ctl[1]) ? se[CPS_DL] : -1;        // band slot of row s+2's late product, -1 = none
      if (a.test_fail_step == s && t == 64) { cp_flag_store(err, code | 0xf); ctl[1] = 0; }
      if (tw >= 1 && row2 && ctl[1]) {
        // wavefronts 2, 3: row s+2 of the band was asked for while the previous step ended; had it arrived?  (It nearly always has: its
        // helpers start from L_ss^-1's flag of the PREVIOUS step.)  If not, ask again until it has.
        unsigned it = 0; long long t0 = 0;
        while (cp_wave_any(cp_chunk_missing(v0) || cp_chunk_missing(v1) || cp_chunk_missing(v2) || cp_chunk_missing(v3) || cp_chunk_missing(v4) || cp_chunk_missing(v5))) {
          if ((++it & 15) == 15) {
            if (cp_flag_load(err) != 0) { ctl[1] = 0; break; }
            if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { if (lane == 0) cp_flag_store(err, code | 4); ctl[1] = 0; break; }
          }
          __builtin_amdgcn_s_sleep(1);
          CP_MISS(0);
          v0 = cp_ld4(rB, ob0); v1 = cp_ld4(rB, ob0 + 2048); v2 = cp_ld4(rB, ob1); v3 = cp_ld4(rB, ob1 + 2048); v4 = cp_ld4(rB, ob2); v5 = cp_ld4(rB, ob2 + 2048);
        }
        // now that X1 is out of Tc the row goes to LDS
        const int h = tw - 1;
        cp_regs_to_lds(T2, 2*h, lane, v0); cp_regs_to_lds(T2, 2*h + 1, lane, v1);
        cp_regs_to_lds(Tc, 2*h, lane, v2); cp_regs_to_lds(Tc, 2*h + 1, lane, v3);
        if (diag2) { cp_regs_to_lds(Dt(dcur ^ 1), 2*h, lane, v4); cp_regs_to_lds(Dt(dcur ^ 1), 2*h + 1, lane, v5); }
      }
      // (Measured and dropped, round 6: wavefront 0 -- idle once the panel's first half is done -- sending L(s+2, s) off and raising its flag
      //  instead of wavefronts 2, 3: the flag comes ~0.15 us later, 1569 re-asks, 189.6 vs 185 us.  And:)
      // (Measured and dropped, round 6: wavefronts 2, 3 computing quadrants of the solve BEFORE this drain, wavefront 1 storing nothing --
      //  the flag of L_ss^-1 then comes 0.7 us later, the band row of step s+2 is late (1038-1551 re-asks per 30 launches instead of ~100)
      //  and the step grows from 4.5 to 4.9-5.1 us: the helpers of row s+3 hang on that flag.)
      CP_DRAIN();                      // every wavefront's share of L_ss^-1 / L(s+1, s) has landed
      if (tw == 0) {
        // wavefront 1 checks in without waiting, follows the panel's first half and does its second (cp_follow)
        team_target += 3;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i1 < ntc) cp_follow(Dt(dcur), Dv(i1), zvec, pwords, i1, colbuf, fail, q, s);
      } else cp_team_sync(ctl, team_target, lane);
      if (t == 128) { cp_flag_store(flags + sd, done_l); if (i1 <= R) cp_flag_store(flags + s1, done_l); CP_STAMP(s, 3); }
      // the late product of tile (s+2, s+1), data-tagged as well: wavefronts 2, 3 ask for the two quadrants each will update (U1 below)
      // and look at them after their products.  (Round 5 waited for the product's flag HERE, before the solve: the solve then started
      // when the product's helper was done, which in turn hung on the previous step's last flag -- a cycle one step long.)
      chol_d4 dq0 = {0.0, 0.0, 0.0, 0.0}, dq1 = dq0;
      const unsigned odq = (dl >= 0 && tw >= 1) ? cp_chunk_off(dl, 2*(tw - 1), lane) : 0u;
      if (dl >= 0 && tw >= 1) { dq0 = cp_ld4(rB, odq); dq1 = cp_ld4(rB, odq + 2048); }
      if (t == 64) CP_STAMP(s, 5);
      if (row2 && ctl[1]) {
        const int s2 = se[CPS_S2];
        if (tw >= 1) {
          // (c) X2 = A'(s+2, s) L_ss^-T on wavefronts 2, 3 (a right quadrant, 8 matrix instructions, and a left one, 4, each)
          const int qr = 2*tw - 1, ql = 2*tw - 2;          // 1, 0 | 3, 2
          {
            // (d) L(s+2, s) leaves the moment it exists, each wavefront its own two quadrants straight from the registers (the late product
            //     of the NEXT step hangs on its flag); the slots this step has emptied are armed only AFTER that flag: on the one in-order
            //     memory counter every store issued before L(s+2, s)'s delays the flag by its own round trip
            const chol_d4 xr = cp_trsm_quadrant(T2, Ds, qr, lane), xl = cp_trsm_quadrant(T2, Ds, ql, lane);
            cp_st4(rL, cp_chunk_off(s2, qr, lane), xr); cp_st4(rL, cp_chunk_off(s2, ql, lane), xl);
            cp_regs_to_lds(Xb1, qr, lane, xr); cp_regs_to_lds(Xb1, ql, lane, xl);
          }
          if (MCP_CP_PROF_ARRIVE && lane == 0) CP_STAMP(s, 20 + tw);
          cp_pair_sync(ctl + 2, pair_target, lane);
          if (t == 128) { CP_STAMP(s, 6); CP_STAMP(s, 11); }
          // then A'(s+2, s+1) -= X2 X1^T + late product, two quadrants per wavefront
          chol_d4 acc = cp_lds_to_regs(Tc, ql, lane), accb = cp_lds_to_regs(Tc, qr, lane);
          cp_mma<true>(acc, Xb1, Xb0, ql, lane);
          cp_mma<true>(accb, Xb1, Xb0, qr, lane);
          if (dl >= 0) {
            unsigned it = 0; long long t0 = 0;
            while (cp_wave_any(cp_chunk_missing(dq0) || cp_chunk_missing(dq1))) {
              if ((++it & 15) == 15) {
                if (cp_flag_load(err) != 0) { ctl[1] = 0; break; }
                if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { if (lane == 0) cp_flag_store(err, code | 7); ctl[1] = 0; break; }
              }
              __builtin_amdgcn_s_sleep(1);
              CP_MISS(1);
              dq0 = cp_ld4(rB, odq); dq1 = cp_ld4(rB, odq + 2048);
            }
            acc += dq0; accb += dq1;                                     // (the helper summed -L(s+2,s-1) L(s+1,s-1)^T)
          }
          cp_regs_to_lds(Tc, ql, lane, acc); cp_regs_to_lds(Tc, qr, lane, accb);
          if (t == 128) CP_STAMP(s, 12);
          // both wavefronts' stores have landed -> one flag
          CP_DRAIN();
          cp_pair_sync(ctl + 2, pair_target, lane);
          if (t == 128) { cp_flag_store(flags + s2, done_l); CP_STAMP(s, 7); }
        } else if (!CP_SWAP) {
          // wavefront 1, back from the panel: A'(s+2, s+2) -= X2 X2^T (lower triangle's quadrants: the panel never reads (0, 1)) once
          // the solve is complete (it usually has been for a while)
          while (__hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < pair_target + 2) __builtin_amdgcn_s_sleep(1);
          pair_target += 4;                // (the pair meets twice per step: after its solve and before L(s+2, s)'s flag)
          asm volatile("" ::: "memory");
          if (diag2) {
            chol_d4 a0 = cp_lds_to_regs(Dt(dcur ^ 1), 0, lane), a2 = cp_lds_to_regs(Dt(dcur ^ 1), 2, lane), a3 = cp_lds_to_regs(Dt(dcur ^ 1), 3, lane);
            cp_mma<true>(a0, Xb1, Xb1, 0, lane); cp_mma<true>(a2, Xb1, Xb1, 2, lane); cp_mma<true>(a3, Xb1, Xb1, 3, lane);
            cp_regs_to_lds(Dt(dcur ^ 1), 0, lane, a0); cp_regs_to_lds(Dt(dcur ^ 1), 2, lane, a2); cp_regs_to_lds(Dt(dcur ^ 1), 3, lane, a3);
          }
        }
      }
    }
  }
}

// ---- a helper workgroup: one tile of the plan, left-looking -------------------------------------------------------------------
__device__ inline bool cp_helper(const CpArgs& a, int q, int hidx, double* lds) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const CpHelper h = a.helpers[hidx];
  cp_tile Pa = (cp_tile)(lds), Pb = (cp_tile)(lds + CP_TILE);
  int* ctl = (int*)(lds + 2*CP_TILE);
  int* flags = a.flags + (size_t)q*a.nflags;
  int* err = a.err + q;
  const __amdgpu_buffer_rsrc_t rL = cp_rsrc(a.Lt + q*a.lt_stride, a.lt_stride*sizeof(double));
  const int par = a.epoch[q] & 1;
  const __amdgpu_buffer_rsrc_t rB = cp_rsrc(a.Bt + q*a.bt_stride + (size_t)par*(a.bt_stride/2), (a.bt_stride/2)*sizeof(double));
  const __amdgpu_buffer_rsrc_t rBo = cp_rsrc(a.Bt + q*a.bt_stride + (size_t)(par ^ 1)*(a.bt_stride/2), (a.bt_stride/2)*sizeof(double));      // (the NEXT launch's copy)
  const double* S = a.S + q*a.sys_stride;
  const int epoch4 = a.epoch[q] << 2, done_l = epoch4 | 2;
  const int code = 0x200 | (hidx << 12);
  // the right-hand-side row's helpers also arm the back-substitution's data-tagged vectors
  if (h.ti == a.ntc && t < 64) {
    double* dst = (t < 32 ? a.xbuf : a.fbuf) + (size_t)q*a.vec_stride + h.tj*CH_NB + (t & 31);
    *reinterpret_cast<unsigned long long*>(dst) = CP_SENT;
  }
  chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (t == 0) CP_HSTAMP(0);
  if (h.in_s) acc = cp_load_A(S, a.n, a.ntc, h.ti, h.tj, wave, lane);
  const int nplain = (h.kind == 1 && h.pre >= 0) ? h.nupd - 1 : h.nupd;
  // The leading updates whose operands were published long ago -- nearly all of them for an entry claimed late, i.e. whenever the
  // workers are few -- are taken in a batch: ONE look at their flags (a lane per update), then their tiles stream in CP_HB updates
  // ahead of the products, through two LDS tile pairs with one barrier per update.  One update at a time this was a table read, a
  // poll and a tile load in a row (three round trips to the memory side per update, ~2.5 us).  Same products in the same order.
  int u_first = 0;
#if CP_HELPER_BATCH
  if (nplain > 1) {
    int2* ulist = (int2*)(lds + 5*CP_TILE);
    if (wave == 0) {
      const int nb = min(nplain, 64);
      bool late = false;
      if (lane < nb) {
        const int2 sl = a.upd[h.upd0 + lane];
        ulist[lane] = sl;
        const int fx = cp_flag_load(flags + sl.x), fy = cp_flag_load(flags + sl.y);
        late = fx != done_l || fy != done_l;
      }
      const unsigned long long m = __builtin_amdgcn_ballot_w64(late);
      if (lane == 0) ctl[2] = m ? min(nb, (int)__builtin_ctzll(m)) : nb;
    }
    __syncthreads();
    const int nready = ctl[2];
    if (nready > 1) {
      chol_d4 va[CP_HB], vb[CP_HB];
#pragma unroll
      for (int k = 0; k < CP_HB; ++k) if (k < nready) {
        const int2 sl = ulist[k];
        va[k] = cp_ld4(rL, cp_chunk_off(sl.x, wave, lane));
        vb[k] = cp_ld4(rL, cp_chunk_off(sl.y, wave, lane));
      }
      for (int base = 0; base < nready; base += CP_HB) {
#pragma unroll
        for (int k = 0; k < CP_HB; ++k) {
          const int u = base + k;
          if (u < nready) {
            cp_tile Qa = (cp_tile)(lds + ((u & 1) ? 3*CP_TILE : 0)), Qb = (cp_tile)(lds + ((u & 1) ? 4*CP_TILE : CP_TILE));
            cp_regs_to_lds(Qa, wave, lane, va[k]); cp_regs_to_lds(Qb, wave, lane, vb[k]);
            if (u + CP_HB < nready) {
              const int2 sl = ulist[u + CP_HB];
              va[k] = cp_ld4(rL, cp_chunk_off(sl.x, wave, lane));
              vb[k] = cp_ld4(rL, cp_chunk_off(sl.y, wave, lane));
            }
            cp_barrier();
            cp_mma<true>(acc, Qa, Qb, wave, lane);
          }
        }
      }
      u_first = nready;
      __syncthreads();
    }
  }
#endif
  for (int u = u_first; u < nplain; ++u) {
    const int2 sl = a.upd[h.upd0 + u];
    if (t == 0) {
      const bool patient = u + 3 < h.nupd;           // far behind the frontier of this tile: poll gently
      const bool ok = patient ? (cp_poll(flags + sl.x, done_l, err, code | 1, true) && (sl.y == sl.x || cp_poll(flags + sl.y, done_l, err, code | 2, true)))
                              : cp_poll3(flags + sl.x, flags + sl.y, flags + sl.y, done_l, err, code | 1);
      ctl[0] = ok ? 1 : 0;
      if (u + 1 == h.nupd) CP_HSTAMP(1);
    }
    __syncthreads();
    if (!ctl[0]) return false;
    const chol_d4 va = cp_ld4(rL, cp_chunk_off(sl.x, wave, lane));
    if (sl.y != sl.x) {
      const chol_d4 vb = cp_ld4(rL, cp_chunk_off(sl.y, wave, lane));
      cp_regs_to_lds(Pb, wave, lane, vb);
    }
    cp_regs_to_lds(Pa, wave, lane, va);
    __syncthreads();
    cp_mma<true>(acc, Pa, sl.y == sl.x ? Pa : Pb, wave, lane);
    __syncthreads();
  }
  if (h.kind == 1 && h.pre >= 0) {
    // last update of a band tile of row i, column m = i - CP_W: L(i, m) = (the far tile's sum, handed over early) L_mm^-T, redone here
    const int2 sl = a.upd[h.upd0 + h.nupd - 1];        // (x: the far tile itself -- not waited for; y: L(j, m), x if j = i)
    if (t == 0) {
      const bool ok = cp_poll(flags + h.pre_flag, epoch4 | 1, err, code | 4, false) &&
                      cp_poll3(flags + h.pre_diag, flags + (sl.y == sl.x ? h.pre_diag : sl.y), flags + h.pre_diag, done_l, err, code | 5);
      ctl[0] = ok ? 1 : 0;
      CP_HSTAMP(1);
    }
    __syncthreads();
    if (!ctl[0]) return false;
    const chol_d4 vp = cp_ld4(rB, cp_chunk_off(h.pre, wave, lane)), vd = cp_ld4(rL, cp_chunk_off(h.pre_diag, wave, lane));
    chol_d4 vy = vp;
    if (sl.y != sl.x) vy = cp_ld4(rL, cp_chunk_off(sl.y, wave, lane));
    cp_regs_to_lds(Pa, wave, lane, vp); cp_regs_to_lds(Pb, wave, lane, vd);
    __syncthreads();
    const chol_d4 x = cp_trsm_quadrant(Pa, Pb, wave, lane);
    __syncthreads();
    cp_regs_to_lds(Pa, wave, lane, x);
    if (sl.y != sl.x) cp_regs_to_lds(Pb, wave, lane, vy);
    __syncthreads();
    cp_mma<true>(acc, Pa, sl.y == sl.x ? Pa : Pb, wave, lane);
  }
  if (h.kind == 0 && h.pre >= 0) {
    // far tile (i, i - CP_W): the sum before the solve goes out first (see CpHelper)
    cp_st4(rB, cp_chunk_off(h.pre, wave, lane), acc);
    CP_DRAIN();
    __syncthreads();
    if (t == 0) cp_flag_store(flags + h.pre_flag, epoch4 | 1);
  }
  if (h.kind == 1) {          // band tile / late product: the sum goes to the critical workgroup, data-tagged -- no drain, no flag (CP_DRAIN's note)
    cp_st4(rB, cp_chunk_off(h.dslot, wave, lane), acc);        // (dslot of a band tile = its band slot)
    cp_chunk_arm(rBo, cp_chunk_off(h.dslot, wave, lane));      // ... and the slot's other copy is armed for the next launch (below)
    if (t == 0) CP_HSTAMP(3);
    return true;
  }
  // far tile: X = acc L_jj^-T
  if (t == 0) { ctl[0] = cp_poll(flags + h.dslot, done_l, err, code | 3, false) ? 1 : 0; CP_HSTAMP(2); }
  __syncthreads();
  if (!ctl[0]) return false;
  cp_regs_to_lds(Pb, wave, lane, cp_ld4(rL, cp_chunk_off(h.dslot, wave, lane)));
  cp_regs_to_lds(Pa, wave, lane, acc);
  __syncthreads();
  const chol_d4 x = cp_trsm_quadrant(Pa, Pb, wave, lane);
  cp_st4(rL, cp_chunk_off(h.slot, wave, lane), x);
  CP_DRAIN();
  __syncthreads();
  if (t == 0) { cp_flag_store(flags + h.slot, done_l); CP_HSTAMP(3); }
  return true;
}

constexpr int CP_LDS_DOUBLES = 8*CP_TILE + 64 + 64 + 64 + 8;          // + the step table of the critical workgroup behind it
// (two kernels, not one with a branch: the critical workgroup's loops are ~50 KB of instructions and share a 64 KB instruction cache with
//  the helper on the neighbouring compute unit -- with both forms of the critical code in one kernel the single-chain factorisation
//  went from 181 to 210 us)
template <bool SEG>
__device__ __forceinline__ void cp_persist_body(const CpArgs& a) {
  extern __shared__ __attribute__((aligned(16))) double cp_lds[];
  const int q = blockIdx.x % a.nsys, role0 = blockIdx.x / a.nsys;
  if (role0 < (SEG ? a.nseg : 1)) { cp_critical<SEG>(a, q, cp_lds, role0); return; }
  // Workers CLAIM the entries of the dependency-ordered helper list one after the other (round 5; round 4 dealt them statically,
  // worker w taking w, w + nworkers, ..., which only made progress if every workgroup of the launch was resident at once).  An entry
  // waits only for entries before it in the list and for the critical workgroup (blocks 0 .. nsys-1, dispatched first); entries are
  // claimed by workgroups that are running, in list order, so whatever an entry waits for is held by a running workgroup: the launch
  // makes progress with ANY number of resident workers -- a partitioned or shared device, a second handle factoring beside this one,
  // the tracker's kernels in between.  Which worker computes a tile does not change a bit of it.
  // (Measured and dropped: asking for the NEXT entry before starting on the current one, to hide the claim's round trip -- the
  //  factorisation went from 0.233 to 0.33 ms per solve: an entry that is ready sits behind its holder's spin on an earlier one.)
  int* ctl = (int*)(cp_lds + 2*CP_TILE);
  for (;;) {
    if (threadIdx.x == 0) ctl[1] = __hip_atomic_fetch_add(a.claim + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int h = ctl[1];
    if (h >= a.nhelpers) return;
    if (!cp_helper(a, q, h, cp_lds)) return;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(CP_THREADS, 2) k_chol_persist(CpArgs a) { cp_persist_body<false>(a); }
__global__ void __launch_bounds__(CP_THREADS, 2) k_chol_persist_seg(CpArgs a) { cp_persist_body<true>(a); }      // several chains side by side (CpArgs::nseg > 1)

// ---- back-substitution ----------------------------------------------------------------------------------------------------------
__device__ inline double cp_tagged_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline void cp_tagged_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool cp_is_sent(double v) { return (unsigned long long)__double_as_longlong(v) == CP_SENT; }
// element (row, col) of a tile in the thread-major quadrant layout
__device__ inline int cp_tq_index(int row, int col) {
  const int qd = (row >> 4)*2 + (col >> 4), ri = row & 15;
  return (qd*64 + ((ri & 3) << 4 | (col & 15)))*4 + (ri >> 2);
}

// far tiles of one block column j:  f_j = sum_i L(i, j)^T x_i  over rows i > j + CP_BACK_NEAR (descending), as the x_i appear
__device__ inline void cp_back_far(const CpBackArgs& a, int q, int cidx, double* lds) {
  const int t = threadIdx.x;
  const int c = t & 31, rg = t >> 5;                   // rows 4 rg .. 4 rg + 3 of every tile
  const int f0 = a.far_start[cidx], f1 = a.far_start[cidx + 1];
  const double* Lt = a.Lt + q*a.lt_stride;
  const double* xb = a.xbuf + (size_t)q*a.vec_stride;
  int* err = a.err + q;
  const int col = a.ntc - 1 - cidx;                    // far helpers are numbered right to left
  if (f0 == f1) return;
  double acc = 0.0;
  double lv[4];
  auto load_tile = [&](int f, double* v) {
    const double* T = Lt + (size_t)a.far_slot[f]*CP_TQ;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = T[cp_tq_index(4*rg + e, c)];
  };
  if (f0 < f1) load_tile(f0, lv);
  for (int f = f0; f < f1; ++f) {
    double nv[4] = {0.0, 0.0, 0.0, 0.0};
    if (f + 1 < f1) load_tile(f + 1, nv);
    const double* xi = xb + a.far_row[f]*CH_NB + 4*rg;
    double xv[4];
    unsigned it = 0; long long t0 = 0;
    for (;;) {
      bool ready = true;
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = cp_tagged_load(xi + e); ready &= !cp_is_sent(xv[e]); }
      if (ready) break;
      if ((++it & 31) == 31) {
        if (cp_flag_load(err) != 0) return;
        if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { cp_flag_store(err, 0x400 | (cidx << 12)); return; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_fma(lv[e], xv[e], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) lv[e] = nv[e];
  }
  lds[rg*32 + c] = acc;
  __syncthreads();
  if (t < 32) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += lds[g*32 + t];
    cp_tagged_store(a.fbuf + (size_t)q*a.vec_stride + col*CH_NB + t, s);
  }
}

// the chain: x_k = L_kk^-T (y_k - f_k - sum_{d=1..NEAR} L(k+d, k)^T x_{k+d}), k = ntc-1 .. 0
// wavefront 0 does the two dependent 32 x 32 products of a step out of LDS; wavefronts 1-3 prepare step k-1 meanwhile:
// tile (k, k-1) and L^-1 of block k-1 into LDS, y - f and the products of the rows beyond k (their x blocks are known)
// per block column k, for the chain workgroup (copied to LDS once): slots of L_kk^-1, of the near tiles (k+1..k+3, k), of y_k; far tiles?
constexpr int CPB_INTS = 8;
enum { CPB_D = 0, CPB_N1, CPB_N2, CPB_N3, CPB_Y, CPB_FAR };
__device__ inline void cp_back_chain(const CpBackArgs& a, int q, double* lds, int ci) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int n = a.n, ntc = a.ntc;
  const int klo = a.ch_lo[ci], khi = a.ch_hi[ci];       // (khi stands for the end of the matrix: no tile of these columns lies in rows khi .. khi + 2)
  const double* Lt = a.Lt + q*a.lt_stride;
  const double* fb = a.fbuf + (size_t)q*a.vec_stride;
  double* xb = a.xbuf + (size_t)q*a.vec_stride;
  int* err = a.err + q;
  // a ring of three: column k is read (wavefront 0) while column k-1 gets its partial sums (wavefronts 1, 2) and column k-2's tiles,
  // asked for at the top of the step, are written at its bottom (wavefront 3): every global round trip has a whole step to land
  auto N1 = [&](int k) { return (cp_tile)(lds + (k % 3)*CP_TILE); };
  auto Dn = [&](int k) { return (cp_tile)(lds + (3 + k % 3)*CP_TILE); };
  double* pre = lds + 6*CP_TILE;                // [3][3][64]: y - f | partial sums of rows k+2 | k+3 (two 16-row halves each)
  double* zb = pre + 3*3*64;                    // [32]
  int* ctl = (int*)(zb + 32);
  int* tab = ctl + 8;                           // [ntc][CPB_INTS]
  double* xs = (double*)(tab + ((ntc*CPB_INTS + 1) & ~1));     // [ntc*32]
  for (int i = t; i < ntc*CPB_INTS; i += CP_THREADS) tab[i] = a.back_tab[i];
  if (t == 0) ctl[0] = 1;
  __syncthreads();
  // wavefront 3, two steps ahead of the chain: tiles of column k into registers (top of a step) ...
  struct Staged { chol_d4 d[4], u[4]; double y, f; };
  auto stage_load = [&](int k, Staged& g) {
    const int* e = tab + k*CPB_INTS;
    const double* Dg = Lt + (size_t)e[CPB_D]*CP_TQ;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      g.d[qd] = *reinterpret_cast<const chol_d4*>(Dg + (qd*64 + lane)*4);
      g.u[qd] = (chol_d4){0.0, 0.0, 0.0, 0.0};
      if (e[CPB_N1] >= 0 && k + 1 < khi) g.u[qd] = *reinterpret_cast<const chol_d4*>(Lt + (size_t)e[CPB_N1]*CP_TQ + (qd*64 + lane)*4);
    }
    g.y = 0.0; g.f = 0.0;
    if (lane < 32) {
      g.y = Lt[(size_t)e[CPB_Y]*CP_TQ + cp_tq_index(0, lane)];
      if (e[CPB_FAR]) g.f = cp_tagged_load(fb + k*CH_NB + lane);       // (looked at in stage_store: usually there by then)
    }
  };
  // ... and into LDS (bottom of the step); the far sum of the column must have arrived by now
  auto stage_store = [&](int k, Staged& g) {
    const int* e = tab + k*CPB_INTS;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) { cp_regs_to_lds(Dn(k), qd, lane, g.d[qd]); cp_regs_to_lds(N1(k), qd, lane, g.u[qd]); }
    if (lane < 32) {
      double f = g.f;
      if (e[CPB_FAR]) {
        unsigned it = 0; long long t0 = 0;
        while (cp_is_sent(f)) {
          if ((++it & 31) == 31) {
            if (cp_flag_load(err) != 0) { ctl[0] = 0; break; }
            if (!t0) t0 = wall_clock64(); else if (cp_expired(t0)) { cp_flag_store(err, 0x500 | (k << 12)); ctl[0] = 0; break; }
          }
          __builtin_amdgcn_s_sleep(1);
          f = cp_tagged_load(fb + k*CH_NB + lane);
        }
      }
      pre[(k % 3)*192 + lane] = g.y - f;
    }
  };
  // wavefronts 1, 2, one step ahead: rows k+2 | k+3 of column k against their (known) x blocks, two 16-row halves per column
  auto partial = [&](int k, int w) {
    const int i = k + w + 1;
    const int c = lane & 31, hh = lane >> 5;
    double sacc = 0.0;
    const int sl = tab[k*CPB_INTS + (w == 1 ? CPB_N2 : CPB_N3)];
    if (sl >= 0 && i < khi) {
      const double* T = Lt + (size_t)sl*CP_TQ;
      double v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = T[cp_tq_index(16*hh + r, c)];
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc = __builtin_fma(v[r], xs[i*CH_NB + 16*hh + r], sacc);
    }
    pre[(k % 3)*192 + w*64 + lane] = sacc;
  };
  // lead-in: columns ntc-1 and ntc-2 staged, the partial sums of column ntc-1 (no rows below it: zeros)
  if (wave == 3) {
    Staged g;
    stage_load(khi - 1, g); stage_store(khi - 1, g);
    if (khi - 2 >= klo) { stage_load(khi - 2, g); stage_store(khi - 2, g); }
  } else if (wave >= 1) pre[((khi - 1) % 3)*192 + wave*64 + lane] = 0.0;
  __syncthreads();
  for (int k = khi - 1; k >= klo; --k) {
    if (!ctl[0]) break;
    Staged g;
    if (wave == 0) {
      const double* P = pre + (k % 3)*192;
      const int c = lane & 31, hh = lane >> 5;
      double sacc = 0.0;
      if (k + 1 < khi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc = __builtin_fma(N1(k)[16*hh + r][c], xs[(k + 1)*CH_NB + 16*hh + r], sacc);
      }
      sacc += __shfl_xor(sacc, 32, 64);
      const double z = P[c] - ((P[64 + c] + P[64 + 32 + c]) + (P[128 + c] + P[128 + 32 + c])) - sacc;
      if (lane < 32) zb[lane] = z;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double x = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) x = __builtin_fma(Dn(k)[16*hh + cc][c], zb[16*hh + cc], x);
      x += __shfl_xor(x, 32, 64);
      const int nbe = min(CH_NB, n - k*CH_NB);
      if (lane < 32) {
        if (lane >= nbe) x = 0.0;
        xs[k*CH_NB + lane] = x;
        cp_tagged_store(xb + k*CH_NB + lane, x);
      }
    } else if (wave == 3) { if (k - 2 >= klo) stage_load(k - 2, g); }
    else if (k - 1 >= klo) partial(k - 1, wave);
    cp_barrier();
    if (wave == 3 && k - 2 >= klo) stage_store(k - 2, g);
  }
  __syncthreads();
  if (t == 0) {
    if (ci == 0) {
      a.epoch[q] = a.epoch[q] + 1;        // the next factorisation of this system sees fresh flags (nothing of this launch reads it)
      a.claim[q] = 0;                     // ... and an untouched helper list
    }
    if (cp_flag_load(err) != 0 && a.fail) atomicOr(a.fail + q, 4);      // a hand-off timed out somewhere: the host falls back to the per-step kernels
  }
  if (!ctl[0]) return;
  for (int i = klo*CH_NB + t; i < min(n, khi*CH_NB); i += CP_THREADS) a.xout[q*a.sys_stride + i] = xs[i];
}

// a factorisation that is not followed by k_chol_back2 (debug hook) closes its epoch itself
__global__ void k_cp_bump(int* epoch, int* claim, int nsys) { if (threadIdx.x < (unsigned)nsys) { epoch[threadIdx.x] += 1; claim[threadIdx.x] = 0; } }

__global__ void __launch_bounds__(CP_THREADS)
k_chol_back2(CpBackArgs a) {
  extern __shared__ __attribute__((aligned(16))) double cp_lds[];
  const int q = blockIdx.x % a.nsys, role = blockIdx.x / a.nsys;
  if (role < a.nchain) cp_back_chain(a, q, cp_lds, role);
  else cp_back_far(a, q, role - a.nchain, cp_lds);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
__global__ void k_chol_persist(CpArgs a);
__global__ void k_chol_persist_seg(CpArgs a);
__global__ void k_chol_back2(CpBackArgs a);
constexpr int CP_PERSIST_LDS_MAX = CP_LDS_DOUBLES*(int)sizeof(double) + (CH_SOLVE_MAX/CH_NB + 2)*CP_STEP_INTS*(int)sizeof(int);
// Per device, once: the kernels' LDS attributes, how many workgroups of k_chol_persist the device holds at a time (occupancy query x
// compute units: 512 on a whole MI355X, a fraction of that on a partitioned one), the spin deadline in wall-clock ticks.  Progress
// does not depend on the launch being resident (workers claim their work), the footprint does: a launch is sized to leave room.
struct CpDevice { bool ok = false; int capacity = 0, ncu = 0; bool disabled = false; int real_fallbacks = 0; };
inline CpDevice& cp_device(int dev = -1) {
  static std::mutex mu; static std::map<int, CpDevice> tab;
  if (dev < 0) (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = tab.find(dev);
  if (it != tab.end()) return it->second;
  CpDevice d;
  int nb = 0, ncu = 0, rate_khz = 100000;
  d.ok = hipFuncSetAttribute((const void*)k_chol_persist, hipFuncAttributeMaxDynamicSharedMemorySize, CP_PERSIST_LDS_MAX) == hipSuccess &&
         hipFuncSetAttribute((const void*)k_chol_persist_seg, hipFuncAttributeMaxDynamicSharedMemorySize, CP_PERSIST_LDS_MAX) == hipSuccess &&
         hipFuncSetAttribute((const void*)k_chol_back2, hipFuncAttributeMaxDynamicSharedMemorySize, (6*CP_TILE + 700 + CH_SOLVE_MAX + 64)*(int)sizeof(double) + (CH_SOLVE_MAX/CH_NB + 4)*8*(int)sizeof(int)) == hipSuccess;
  if (d.ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_chol_persist, CP_THREADS, (size_t)CP_LDS_DOUBLES*sizeof(double) + 40*CP_STEP_INTS*sizeof(int)) == hipSuccess &&
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) { d.capacity = nb*ncu; d.ncu = ncu; }
  if (const char* e = getenv("MCP_BA_CHOL_CAPACITY")) d.capacity = atoi(e);      // (test hook: pretend to be a partition of that many workgroup slots)
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
  double ms = 20.0; if (const char* e = getenv("MCP_BA_CHOL_DEADLINE_MS")) ms = std::max(0.01, atof(e));
  const long long ticks = (long long)(ms*rate_khz);
  if (d.ok) d.ok = hipMemcpyToSymbol(HIP_SYMBOL(cp_deadline), &ticks, sizeof ticks) == hipSuccess;
  (void)hipGetLastError();
  return tab.emplace(dev, d).first->second;
}

struct CholPersist {
  static constexpr int max_sys = 4;
  int n = 0, ntc = 0, nslots = 0, nbslots = 0, nflags = 0, nhelpers = 0, nfarcols = 0;
  bool ok = false;
  std::vector<int> steps;            // [ntc + 1][CP_STEP_INTS], index = step + 1
  std::vector<int> slot_of, bslot_of, delta_of;       // delta_of[i]: band slot of row i's late product (below), -1 = none
  std::vector<CpHelper> helpers; std::vector<int2> upd;
  std::vector<int> far_start, far_slot, far_row, back_tab;
  int *d_steps = nullptr, *d_slot_of = nullptr, *d_bslot_of = nullptr, *d_delta_of = nullptr, *d_flags = nullptr, *d_err = nullptr, *d_epoch = nullptr, *d_far_start = nullptr, *d_far_slot = nullptr, *d_far_row = nullptr, *d_back_tab = nullptr;
  CpHelper* d_helpers = nullptr; int2* d_upd = nullptr;
  double *d_Lt = nullptr, *d_Bt = nullptr, *d_x = nullptr, *d_f = nullptr;
  size_t lt_stride = 0, bt_stride = 0; int vec_stride = 0;
  int nworkers = 0;                      // helper workgroups per system (MCP_BA_CHOL_WORKERS)
  int batch_sys = 0;                     // systems the caller has in flight beside each other, over all its streams (0 = just this launch's)
  bool back_chains = true;               // MCP_BA_CHOL_BACK_CHAINS=0: the back-substitution walks all block columns on one workgroup whatever the plan's chains
  bool spread = true;                    // MCP_BA_CHOL_SPREAD=0: the workers of a launch are not cut to one workgroup per compute unit
  double flops = 0;                      // of one factorisation as this plan executes it (tile operations; mcp_ba_timing.chol_flops_plan)
  int* fail_ptr = nullptr; int n_launch = 0, test_fail_launch = -1;      // (MCP_BA_TEST_PERSIST_FAIL=k: the k-th factorisation of this plan is made to time out)
  ~CholPersist() { release(); }
  // One device arena per plan: [tables (one upload) | flags, error words, epochs | L tiles | band tiles | x | f], a block of the
  // process-wide cache (ba_pool.h): a ChainBundle lives for one BundleAdjust call, plans are built and dropped all the time.
  char* arena = nullptr; size_t arena_bytes = 0; int arena_dev = -1;
  void release() {
    if (arena) DevCache::get().put(arena, arena_bytes, arena_dev);
    arena = nullptr; arena_bytes = 0; arena_dev = -1;
    d_steps = d_delta_of = d_epoch = d_slot_of = d_bslot_of = d_flags = d_err = d_far_start = d_far_slot = d_far_row = d_back_tab = nullptr; d_helpers = nullptr; d_upd = nullptr;
    d_Lt = d_Bt = d_x = d_f = nullptr; ok = false;
  }
  int arena_get(size_t bytes) { arena = (char*)DevCache::get().take(bytes, &arena_bytes, &arena_dev); return arena ? 0 : -1; }
  // pattern: ntc x ntc lower-triangular tile occupancy of S (empty = dense); in_old: the tiles the assembly writes (ti << 16 | tj),
  // with the right-hand side inside block row n / 32 as ba_chol.h's plan has it
  // segs: first block column of every chain (ascending, segs[0] = 0; empty = one chain).  The caller has ordered the unknowns so that the
  // chains before the last do not couple with each other: what couples to them from the last chain's rows are far tiles.
  std::vector<int> seg_start;            // [nseg + 1]
  int nseg = 1;
  int build(int n_, const std::vector<unsigned char>& pattern, const std::vector<int>& old_tiles, const std::vector<int>& segs = std::vector<int>()) {
    release();
    n = n_; ntc = (n + CH_NB - 1)/CH_NB;
    if (ntc < 1) return 0;
    const int R = ntc, nr = ntc + 1;
    seg_start.assign(1, 0);
    for (size_t g = 1; g < segs.size() && (int)seg_start.size() < CP_MAX_SEG; ++g) if (segs[g] >= seg_start.back() + CP_W && segs[g] + CP_W <= ntc) seg_start.push_back(segs[g]);
    // (the promise checked: a tile between two chains before the last would make a band tile of the later chain wait for a far tile that
    //  is behind it in the helpers' order -- such a plan is built as one chain, correct whatever the order of the unknowns)
    if (seg_start.size() > 1 && !pattern.empty()) {
      const int last0 = seg_start.back();
      auto chain_of = [&](int c) { int g = 0; while (g + 1 < (int)seg_start.size() && seg_start[g + 1] <= c) ++g; return g; };
      bool coupled = false;
      for (int i = 0; i < last0 && !coupled; ++i) for (int j = 0; j < i; ++j) if (pattern[(size_t)i*ntc + j] && chain_of(i) != chain_of(j)) { coupled = true; break; }
      if (coupled) seg_start.assign(1, 0);
    }
    nseg = (int)seg_start.size(); seg_start.push_back(ntc);
    std::vector<int> seg_of(nr, nseg - 1), loc_of(nr, 0);
    int maxlen = 0;
    for (int g = 0; g < nseg; ++g) { for (int c = seg_start[g]; c < seg_start[g + 1]; ++c) { seg_of[c] = g; loc_of[c] = c - seg_start[g]; } if (g + 1 < nseg) maxlen = std::max(maxlen, seg_start[g + 1] - seg_start[g]); }
    loc_of[R] = seg_start[nseg] - seg_start[nseg - 1];
    auto in_band = [&](int i, int j) { return i - j < CP_W && seg_of[i] == seg_of[j]; };       // (a chain's own tiles: its critical workgroup keeps them)
    auto order_key = [&](int c) { return seg_of[c] + 1 < nseg ? loc_of[c] : maxlen + loc_of[c]; };      // chains side by side: by the step inside the chain; the last chain behind them
    std::vector<unsigned char> P((size_t)nr*ntc, 0), inS((size_t)nr*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) P[(size_t)i*ntc + j] = pattern.empty() ? 1 : pattern[(size_t)i*ntc + j];
    for (int tp : old_tiles) { const int i = tp >> 16, j = tp & 0xffff; if (i < ntc && j < ntc) inS[(size_t)i*ntc + j] = 1; }
    if (old_tiles.empty()) for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) inS[(size_t)i*ntc + j] = 1;      // dense test matrices: S is all there
    for (int i = 0; i < ntc; ++i) for (int j = std::max(0, i - CP_W + 1); j <= i; ++j) if (in_band(i, j)) P[(size_t)i*ntc + j] = 1;      // the band is the critical workgroup's: always there
    for (int j = 0; j < ntc; ++j) { P[(size_t)R*ntc + j] = 1; inS[(size_t)R*ntc + j] = 1; }
    for (int k = 0; k < ntc; ++k) {                                // symbolic fill-in
      std::vector<int> rows;
      for (int i = k + 1; i < nr; ++i) if (P[(size_t)i*ntc + k]) rows.push_back(i);
      for (int a2 : rows) for (int b : rows) if (b <= a2 && b < ntc) P[(size_t)a2*ntc + b] = 1;
    }
    slot_of.assign((size_t)nr*ntc, -1); bslot_of.assign((size_t)nr*ntc, -1);
    nslots = 0; nbslots = 0;
    for (int i = 0; i < nr; ++i) for (int j = 0; j < ntc && j <= i; ++j) if (P[(size_t)i*ntc + j]) {
      slot_of[(size_t)i*ntc + j] = nslots++;
      if (in_band(i, j)) bslot_of[(size_t)i*ntc + j] = nbslots++;
    }
    // The last update of band tile (i, i-1), column m = i - 3, needs L(i-1, i-3): the critical workgroup's own tile of the step before
    // the row is taken in -- a round trip through a helper on the critical cycle.  That one product travels apart ("delta"): its own
    // helper, its own band slot and flag (index nslots + i); the critical workgroup subtracts it a step later, when it has long arrived.
    delta_of.assign(nr, -1);
    for (int i = CP_W; i < nr; ++i) if (i - 1 < ntc && seg_of[i - CP_W] == seg_of[i] && slot_of[(size_t)i*ntc + i - CP_W] >= 0 && slot_of[(size_t)(i - 1)*ntc + i - CP_W] >= 0) delta_of[i] = nbslots++;
    // ... and the far tile (i, i - CP_W) hands its sum to the band tiles of its row before its own solve (CpHelper): flag index nslots + nr + i
    std::vector<int> pre_of(nr, -1);
    for (int i = CP_W; i < nr; ++i) if (seg_of[i - CP_W] == seg_of[i] && slot_of[(size_t)i*ntc + i - CP_W] >= 0) pre_of[i] = nbslots++;
    nflags = nslots + 2*nr;
    // helpers, in the order of the column that completes them: far tile (i, j) -> j; band tile (i, j) -> i - CP_W; delta of row i -> i - CP_W
    struct Key { int key, kind, i, j; };
    std::vector<Key> ks;
    for (int i = 0; i < nr; ++i) for (int j = 0; j < ntc && j <= i; ++j) if (P[(size_t)i*ntc + j]) {
      const bool band = in_band(i, j);
      // (a band tile is complete when column i - CP_W is; at the head of a chain that is before the chain starts: with the last of the other chains' columns)
      ks.push_back({band ? (nseg == 1 ? i - CP_W : std::max(order_key(i) - CP_W, seg_of[i] + 1 < nseg ? -CP_W : maxlen)) : order_key(j), band ? 1 : 0, i, j});
    }
    for (int i = 0; i < nr; ++i) if (delta_of[i] >= 0) ks.push_back({order_key(i) - CP_W, 2, i, i - 1});
    std::stable_sort(ks.begin(), ks.end(), [](const Key& x, const Key& y) { if (x.key != y.key) return x.key < y.key; if (x.kind != y.kind) return x.kind < y.kind; return x.i < y.i; });
    helpers.clear(); upd.clear();
    for (const Key& k : ks) {
      CpHelper h; h.ti = k.i; h.tj = k.j; h.upd0 = (int)upd.size(); h.pre = -1; h.pre_flag = h.pre_diag = h.pad = 0;
      if (k.kind == 2) {       // runs the band tile's code: zero start, one product, handed over through its band slot
        h.kind = 1; h.in_s = 0; h.slot = nslots + k.i; h.dslot = delta_of[k.i];
        upd.push_back(make_int2(slot_of[(size_t)k.i*ntc + k.i - CP_W], slot_of[(size_t)(k.i - 1)*ntc + k.i - CP_W]));
        h.ti = -1 - k.i;        // (not a tile of its own: never touches S, never arms the back-substitution's vectors)
      } else {
        h.kind = k.kind; h.slot = slot_of[(size_t)k.i*ntc + k.j];
        h.dslot = k.kind ? bslot_of[(size_t)k.i*ntc + k.j] : slot_of[(size_t)k.j*ntc + k.j];
        h.in_s = inS[(size_t)k.i*ntc + k.j];
        // the columns this helper applies: all before j -- but for a band tile not the ones its chain's critical workgroup applies itself
        // (its own columns i - 2, i - 1), and not the one that travels as the row's late product
        int mlast = -1;
        for (int m = 0; m < k.j; ++m) {
          if (k.kind == 1 && m > k.i - CP_W && seg_of[m] == seg_of[k.i]) continue;
          if (k.kind == 1 && k.j == k.i - 1 && delta_of[k.i] >= 0 && m == k.i - CP_W) continue;
          const int sa = slot_of[(size_t)k.i*ntc + m], sb = slot_of[(size_t)k.j*ntc + m];
          if (sa >= 0 && sb >= 0) { upd.push_back(make_int2(sa, sb)); mlast = m; }
        }
        const int mw = k.i - CP_W;          // the column whose far tile of this row feeds the row's band tiles
        if (k.kind == 0 && k.j == mw && pre_of[k.i] >= 0) { h.pre = pre_of[k.i]; h.pre_flag = nslots + nr + k.i; }
        if (k.kind == 1 && mlast == mw && mw >= 0 && pre_of[k.i] >= 0 && slot_of[(size_t)k.j*ntc + mw] >= 0) {
          h.pre = pre_of[k.i]; h.pre_flag = nslots + nr + k.i; h.pre_diag = slot_of[(size_t)mw*ntc + mw];      // (its last list entry is column mw's)
        }
      }
      h.nupd = (int)upd.size() - h.upd0;
      helpers.push_back(h);
    }
    nhelpers = (int)helpers.size();
    {
      const double t3 = (double)CH_NB*CH_NB*CH_NB;
      int nfar = 0; for (const CpHelper& h : helpers) if (h.kind == 0) ++nfar;
      // helpers: a product per list entry, a triangular solve per far tile; critical workgroup per block column: two solves (rows s+1, s+2),
      // three products (the band's updates), a diagonal tile's factorisation + inverse
      flops = 2.0*t3*(double)upd.size() + t3*nfar + (double)ntc*(2.0*t3 + 3.0*2.0*t3 + 2.0*t3/3.0);
    }
    steps.assign((size_t)(ntc + 1)*CP_STEP_INTS, -1);
    for (int st = -1; st < ntc; ++st) {
      int* e = &steps[(size_t)(st + 1)*CP_STEP_INTS];
      const int i1 = st + 1, i2 = st + 2, i3 = st + 3;
      if (st >= 0) { e[CPS_SD] = slot_of[(size_t)st*ntc + st]; e[CPS_S1] = slot_of[(size_t)i1*ntc + st]; if (i2 <= R) { e[CPS_S2] = slot_of[(size_t)i2*ntc + st]; e[CPS_DL] = delta_of[i2]; } }
      if (i3 <= R) {
        e[CPS_F0] = slot_of[(size_t)i3*ntc + i1]; e[CPS_F1] = slot_of[(size_t)i3*ntc + i2]; e[CPS_F2] = i3 < ntc ? slot_of[(size_t)i3*ntc + i3] : e[CPS_F1];
        e[CPS_B0] = bslot_of[(size_t)i3*ntc + i1]; e[CPS_B1] = bslot_of[(size_t)i3*ntc + i2]; e[CPS_B2] = i3 < ntc ? bslot_of[(size_t)i3*ntc + i3] : e[CPS_B1];
      }
    }
    // back-substitution: far tiles per block column, columns right to left (role 1 + idx <-> column ntc - 1 - idx)
    far_start.assign(1, 0); far_slot.clear(); far_row.clear();
    for (int idx = 0; idx < ntc; ++idx) {
      const int j = ntc - 1 - idx;
      for (int i = ntc - 1; i > j + CP_BACK_NEAR; --i) if (P[(size_t)i*ntc + j]) { far_slot.push_back(slot_of[(size_t)i*ntc + j]); far_row.push_back(i); }
      far_start.push_back((int)far_slot.size());
    }
    back_tab.assign((size_t)ntc*8, -1);
    for (int k = 0; k < ntc; ++k) {
      int* e = &back_tab[(size_t)k*8];
      e[0] = slot_of[(size_t)k*ntc + k];
      for (int d = 1; d <= 3; ++d) e[d] = (k + d < ntc) ? slot_of[(size_t)(k + d)*ntc + k] : -1;
      e[4] = slot_of[(size_t)ntc*ntc + k];
      e[5] = far_start[ntc - 1 - k + 1] > far_start[ntc - 1 - k] ? 1 : 0;
    }
    lt_stride = (size_t)nslots*CP_TQ; bt_stride = (size_t)2*std::max(nbslots, 1)*CP_TQ; vec_stride = ntc*CH_NB;
    // lay the arena out (every part 256-byte aligned), stage the tables in one host block, one upload
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    struct Part { const void* src; size_t bytes; void** dst; };
    Part parts[] = {
      {slot_of.data(), slot_of.size()*4, (void**)&d_slot_of}, {bslot_of.data(), bslot_of.size()*4, (void**)&d_bslot_of}, {delta_of.data(), delta_of.size()*4, (void**)&d_delta_of},
      {steps.data(), steps.size()*4, (void**)&d_steps}, {helpers.data(), helpers.size()*sizeof(CpHelper), (void**)&d_helpers}, {upd.data(), upd.size()*sizeof(int2), (void**)&d_upd},
      {far_start.data(), far_start.size()*4, (void**)&d_far_start}, {far_slot.data(), far_slot.size()*4, (void**)&d_far_slot}, {far_row.data(), far_row.size()*4, (void**)&d_far_row},
      {back_tab.data(), back_tab.size()*4, (void**)&d_back_tab}};
    size_t tab_bytes = 0;
    for (const Part& pt : parts) tab_bytes += al(std::max<size_t>(pt.bytes, 4));
    const size_t flag_bytes = al(sizeof(int)*(size_t)nflags*max_sys), err_bytes = al(sizeof(int)*max_sys*4), ep_bytes = al(sizeof(int)*max_sys);
    const size_t lt_bytes = al(sizeof(double)*lt_stride*max_sys), bt_bytes = al(sizeof(double)*bt_stride*max_sys), vec_bytes = al(sizeof(double)*(size_t)vec_stride*max_sys);
    if (arena_get(tab_bytes + flag_bytes + err_bytes + ep_bytes + lt_bytes + bt_bytes + 2*vec_bytes)) return -1;
    std::vector<char> stage(tab_bytes + flag_bytes + err_bytes + ep_bytes, 0);
    size_t off = 0;
    for (const Part& pt : parts) { if (pt.bytes) std::memcpy(stage.data() + off, pt.src, pt.bytes); *pt.dst = arena + off; off += al(std::max<size_t>(pt.bytes, 4)); }
    d_flags = (int*)(arena + off); off += flag_bytes;
    d_err = (int*)(arena + off); off += err_bytes;
    d_epoch = (int*)(arena + off); { int* e = (int*)(stage.data() + off); for (int q = 0; q < max_sys; ++q) e[q] = 1; } off += ep_bytes;
    d_Lt = (double*)(arena + off); d_Bt = (double*)(arena + off + lt_bytes);
    // every band slot starts armed (the sentinel in every double; ba_chol2.h, CP_DRAIN's note): the null stream's fill is done before the
    // blocking copy behind it returns
    if (hipMemsetD32Async((hipDeviceptr_t)d_Bt, (int)CP_SENT32, bt_bytes/4, nullptr) != hipSuccess) return -1;
    if (hipMemcpy(arena, stage.data(), off, hipMemcpyHostToDevice) != hipSuccess) return -1;      // (tables, zeroed flags and error words, epochs = 1)
    off += lt_bytes; off += bt_bytes;
    d_x = (double*)(arena + off); off += vec_bytes;
    d_f = (double*)(arena + off); off += vec_bytes;
    {
      // enough workers to hold a few block columns' worth of tiles at once (a column's far tile, its row's band tiles and the late product
      // work side by side; the columns ahead are being summed while they wait), at most 126 so that four systems are resident together
      const char* e = getenv("MCP_BA_CHOL_WORKERS");
      const int per_col = (nhelpers + ntc - 1)/std::max(ntc, 1);
      nworkers = e && atoi(e) > 0 ? atoi(e) : std::max(16, std::min(126, 10*per_col));
      nworkers = std::min(nworkers, std::max(nhelpers, 1));
      // ... and no more than the device holds beside a reserve of an eighth of its slots (what else is running -- the tracker thread's
      // frames, the speculative stream -- finds room; a second handle's factorisation shares the workers' slots and both go on)
      const CpDevice& dv = cp_device();
      if (!dv.ok || dv.disabled || dv.capacity < 2*max_sys) { release(); return 0; }       // (no one-launch plan on this device: ok stays false, the per-step kernels run)
      nworkers = std::max(1, std::min(nworkers, (dv.capacity - dv.capacity/8)/max_sys - nseg));
    }
    { const char* e = getenv("MCP_BA_CHOL_SPREAD"); spread = !(e && atoi(e) == 0) && !getenv("MCP_BA_CHOL_WORKERS"); }
    { const char* e = getenv("MCP_BA_CHOL_BACK_CHAINS"); back_chains = !(e && atoi(e) == 0); }
    { const char* e = getenv("MCP_BA_TEST_PERSIST_FAIL"); test_fail_launch = e ? atoi(e) : -1; n_launch = 0; }
    ok = true;
    return 0;
  }
};


// one launch: factor systems [q0, q0 + nsys) of the batch (S + q * sys_stride); L tiles, L_kk^-1 and y land in P.d_Lt
inline int chol_persist_factor(hipStream_t st, CholPersist& P, const double* S, int* fail, int nsys, size_t sys_stride, int q0) {
  if (!cp_device().ok) return -1;          // (attributes, capacity and deadline of this device: set once)
  CpArgs a;
  a.S = S + q0*sys_stride; a.sys_stride = sys_stride; a.n = P.n; a.ntc = P.ntc; a.nsys = nsys; a.nhelpers = P.nhelpers; a.nworkers = P.nworkers;
  a.nseg = P.nseg; for (int g = 0; g <= CP_MAX_SEG; ++g) a.seg_start[g] = g < (int)P.seg_start.size() ? P.seg_start[g] : P.ntc;
  a.slot_of = P.d_slot_of; a.bslot_of = P.d_bslot_of; a.delta_of = P.d_delta_of; a.steps = P.d_steps; a.helpers = P.d_helpers; a.upd = P.d_upd;
  a.Lt = P.d_Lt + q0*P.lt_stride; a.lt_stride = P.lt_stride; a.Bt = P.d_Bt + q0*P.bt_stride; a.bt_stride = P.bt_stride;
  a.flags = P.d_flags + (size_t)q0*P.nflags; a.nslots = P.nslots; a.nflags = P.nflags; a.err = P.d_err + q0; a.fail = fail + q0;
  a.epoch = P.d_epoch + q0; P.fail_ptr = fail; a.claim = P.d_err + CholPersist::max_sys + q0;
  a.test_fail_step = (++P.n_launch == P.test_fail_launch) ? std::min(5, P.ntc - 1) : -1;
  a.xbuf = P.d_x + (size_t)q0*P.vec_stride; a.fbuf = P.d_f + (size_t)q0*P.vec_stride; a.vec_stride = P.vec_stride;
  // Workers per system of THIS launch: two workgroups fit a compute unit, but a critical workgroup that shares its unit -- and a
  // worker that shares one -- is slower than the same work spread out (four systems x 111 workers: 201 us, x 55: 188 us, one system
  // alone: 186 us; round 6, after the workers' batched updates made 55 of them enough).  So when more than two systems are in
  // flight together the launch is cut to what gives every workgroup of the batch a unit of its own, an eighth of the device left free.
  {
    const CpDevice& dv = cp_device();
    const int batch = std::max(P.batch_sys, nsys);
    if (P.spread && batch > 2 && dv.ncu > 0) a.nworkers = std::max(1, std::min(P.nworkers, std::max(8, (dv.ncu - dv.ncu/8)/batch - P.nseg)));
  }
  (void)hipGetLastError();          // (a leftover of an earlier, unrelated runtime call -- a stream query's hipErrorNotReady -- is not this launch's refusal)
  if (a.nseg > 1) hipLaunchKernelGGL(k_chol_persist_seg, dim3((a.nseg + a.nworkers)*nsys), dim3(CP_THREADS), CP_LDS_DOUBLES*sizeof(double) + (size_t)(P.ntc + 1)*CP_STEP_INTS*sizeof(int), st, a);
  else hipLaunchKernelGGL(k_chol_persist, dim3((1 + a.nworkers)*nsys), dim3(CP_THREADS), CP_LDS_DOUBLES*sizeof(double) + (size_t)(P.ntc + 1)*CP_STEP_INTS*sizeof(int), st, a);
  { const hipError_t e = hipGetLastError(); return (e == hipSuccess || e == hipErrorNotReady) ? 0 : -1; }       // (a launch the runtime refuses -- its LDS or grid does not fit this device -- is reported here, by name, not at the end of the solve)
}
// the second launch: x = L^-T y into row n of S (xout = S + n n)
inline int chol_persist_back(hipStream_t st, CholPersist& P, double* S, int nsys, size_t sys_stride, int q0) {
  CpBackArgs a;
  a.Lt = P.d_Lt + q0*P.lt_stride; a.lt_stride = P.lt_stride; a.slot_of = P.d_slot_of; a.n = P.n; a.ntc = P.ntc; a.nsys = nsys; a.ncols = P.ntc;
  a.far_start = P.d_far_start; a.far_slot = P.d_far_slot; a.far_row = P.d_far_row; a.back_tab = P.d_back_tab;
  a.xbuf = P.d_x + (size_t)q0*P.vec_stride; a.fbuf = P.d_f + (size_t)q0*P.vec_stride; a.vec_stride = P.vec_stride;
  a.xout = S + q0*sys_stride + (size_t)P.n*P.n; a.sys_stride = sys_stride; a.err = P.d_err + q0; a.epoch = P.d_epoch + q0; a.fail = P.fail_ptr ? P.fail_ptr + q0 : nullptr; a.claim = P.d_err + CholPersist::max_sys + q0;
  static_assert(CP_BACK_NEAR == 3 && CPB_INTS == 8, "the chain workgroup's column table holds three near tiles");
  const size_t lds = (size_t)(6*CP_TILE + 3*3*64 + 32 + 8 + P.ntc*CH_NB)*sizeof(double) + (size_t)(P.ntc*CPB_INTS + 18)*sizeof(int);
  (void)hipGetLastError();
  // (a plan of several chains: the chains before the last do not couple, and what couples them to the last are far tiles)
  a.nchain = std::max(1, P.nseg - 1);
  for (int j = 0; j < CP_MAX_SEG; ++j) { a.ch_lo[j] = 0; a.ch_hi[j] = 0; }
  a.ch_lo[0] = P.nseg >= 2 ? P.seg_start[P.nseg - 2] : 0; a.ch_hi[0] = P.ntc;
  for (int j = 1; j < a.nchain; ++j) { a.ch_lo[j] = P.seg_start[P.nseg - 2 - j]; a.ch_hi[j] = P.seg_start[P.nseg - 1 - j]; }
  if (!P.back_chains) { a.nchain = 1; a.ch_lo[0] = 0; }
  hipLaunchKernelGGL(k_chol_back2, dim3((a.nchain + P.ntc)*nsys), dim3(CP_THREADS), lds, st, a);
  { const hipError_t e = hipGetLastError(); return (e == hipSuccess || e == hipErrorNotReady) ? 0 : -1; }
}

}  // namespace mcp
