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
constexpr unsigned CP_SPIN_MAX = 1u << 22;
constexpr int CP_BACK_NEAR = 3;               // rows k+1 .. k+CP_BACK_NEAR of column k stay with the chain workgroup
constexpr unsigned long long CP_SENT = 0xFFFDEADBEEF5A5A5ull;      // "not written yet" (a NaN payload arithmetic never yields)

struct CpHelper { int ti, tj, slot, dslot, upd0, nupd, kind, in_s; };      // kind: 0 far tile, 1 band tile
struct CpArgs {
  const double* S; size_t sys_stride; int n, ntc, nsys, nhelpers;
  const int* slot_of; const int* bslot_of; const CpHelper* helpers; const int2* upd;
  double* Lt; size_t lt_stride;               // published L tiles / L_kk^-1, per system
  double* Bt; size_t bt_stride;               // band tiles handed to the critical workgroup, per system
  int* flags; int nslots; int* err; int* fail; int epoch4;
  double* xbuf; double* fbuf; int vec_stride;
};
struct CpBackArgs {
  const double* Lt; size_t lt_stride; const int* slot_of; int n, ntc, nsys, ncols;
  const int* far_start; const int* far_slot; const int* far_row;      // per block column: its far tiles, rows descending
  double* xbuf; double* fbuf; int vec_stride; double* xout; size_t sys_stride; int* err;
};

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
__device__ inline int cp_flag_load(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void cp_flag_store(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: spin until *f == want; false = gave up (this spin timed out, or another workgroup of the system raised the error word)
__device__ inline bool cp_poll(const int* f, int want, int* err, int code, bool patient) {
  for (unsigned it = 0; it < CP_SPIN_MAX; ++it) {
    if (cp_flag_load(f) == want) return true;
    if ((it & 31) == 31 && cp_flag_load(err) != 0) return false;
    if (patient) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
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
// acc (+/-)= Pi[rows of the quadrant] Pj[columns of the quadrant]^T over K = 32
template <bool NEG>
__device__ inline void cp_mma(chol_d4& acc, cp_tile Pi, cp_tile Pj, int qd, int l) {
  const int ri = 16*(qd >> 1) + (l & 15), rj = 16*(qd & 1) + (l & 15), rq = l >> 4;
#pragma unroll
  for (int kk = 0; kk < CH_NB; kk += 4) {
    const double x = Pi[ri][kk + rq];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -x : x, Pj[rj][kk + rq], acc, 0, 0, 0);
  }
}
__device__ inline unsigned cp_chunk_off(int slot, int qd, int l) { return (unsigned)(((size_t)slot*CP_TQ + (size_t)(qd*64 + l)*4)*sizeof(double)); }

// A(ti, tj) out of S (row-major n x n, right-hand side = row n); block row R = ntc is the right-hand side (one row)
__device__ inline chol_d4 cp_load_A(const double* __restrict__ S, int n, int ntc, int ti, int tj, int qd, int l) {
  chol_d4 v = {0.0, 0.0, 0.0, 0.0};
  const int rl = 16*(qd >> 1) + (l >> 4), c = tj*CH_NB + 16*(qd & 1) + (l & 15);
  if (c >= n) return v;
  if (ti == ntc) { if (rl == 0) v[0] = S[(size_t)n*n + c]; return v; }
#pragma unroll
  for (int g = 0; g < 4; ++g) { const int r = ti*CH_NB + rl + 4*g; if (r < n) v[g] = S[(size_t)r*n + c]; }
  return v;
}

// ---- the critical workgroup ------------------------------------------------------------------------------------------------
// LDS-only rendezvous of the three team wavefronts (the hardware barrier would include wavefront 0, which is inside the panel)
__device__ inline void cp_team_sync(int* ctr, int& target, int lane) {
  target += 3;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// wavefront 0: L L^T = D (identity beyond nbe), L^-1 -> Dinv (row-major); the pivot loop is ba_chol.h's panel
__device__ __forceinline__ void cp_potrf(cp_tile D, int nbe, cp_tile Dinv, double* colbuf, int* fail) {
  const int lane = threadIdx.x, rr = lane & 31;
  const bool low = lane >= 32;
  double d[CH_NB];
#pragma unroll
  for (int c = 0; c < CH_NB; ++c) d[c] = (!low && c <= rr && rr < nbe && c < nbe) ? D[rr][c] : ((c == rr) ? 1.0 : 0.0);
  const double piv0 = readlane_f64(d[0], 0);
  bool bad = !(piv0 > 0.0);
  double inv = rsqrt(piv0);
  chol_panel_pivots(d, inv, bad, colbuf, std::make_integer_sequence<int, CH_NB>());
  if (bad && lane == 0) atomicOr(fail, 2);
  // lane 32 + r ends with row r of L^-T = column r of L^-1
  if (low) {
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) Dinv[c][rr] = (c >= rr) ? d[c] : 0.0;
  }
}

__device__ inline void cp_critical(const CpArgs& a, int q, double* lds) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int n = a.n, ntc = a.ntc, R = a.ntc;
  // LDS tiles (pointers computed, never kept in an indexed array: that would live in scratch)
  auto Dv = [&](int i) { return (cp_tile)(lds + (i & 1)*CP_TILE); };             // L_kk^-1 of block k in Dv(k)
  const cp_tile Xb0 = (cp_tile)(lds + 2*CP_TILE), Xb1 = (cp_tile)(lds + 3*CP_TILE);
  const cp_tile Tc = (cp_tile)(lds + 4*CP_TILE), T2 = (cp_tile)(lds + 5*CP_TILE);
  auto Dt = [&](int i) { return (cp_tile)(lds + (6 + (i & 1))*CP_TILE); };
  double* colbuf = lds + 8*CP_TILE;
  int* ctl = (int*)(colbuf + 64);            // [0] team counter, [1] ok / abort word
  int* flags = a.flags + (size_t)q*a.nslots;
  int* err = a.err + q; int* fail = a.fail + q;
  const __amdgpu_buffer_rsrc_t rL = cp_rsrc(a.Lt + q*a.lt_stride, a.lt_stride*sizeof(double));
  const __amdgpu_buffer_rsrc_t rB = cp_rsrc(a.Bt + q*a.bt_stride, a.bt_stride*sizeof(double));
  const int* slot_of = a.slot_of; const int* bslot_of = a.bslot_of;
  const int want_band = a.epoch4 | 1, done_l = a.epoch4 | 2;
  const int code = 0x100;
  if (t == 0) { ctl[0] = 0; ctl[1] = 1; }
  __syncthreads();
  // ---- prologue: rows 0 and 1 of the band (their helpers pass A through), diagonal tile 0 factored
  const bool row1_diag = 1 < ntc;
  if (t == 0) {
    bool ok = cp_poll(flags + slot_of[0], want_band, err, code | 1, false);
    ok = ok && cp_poll(flags + slot_of[1*ntc + 0], want_band, err, code | 2, false);
    if (row1_diag) ok = ok && cp_poll(flags + slot_of[1*ntc + 1], want_band, err, code | 3, false);
    if (!ok) ctl[1] = 0;
  }
  __syncthreads();
  if (!ctl[1]) return;
  for (int j = wave; j < 12; j += 4) {
    const int which = j >> 2, qd = j & 3;
    if (which == 2 && !row1_diag) continue;
    const int bs = which == 0 ? bslot_of[0] : (which == 1 ? bslot_of[1*ntc + 0] : bslot_of[1*ntc + 1]);
    const chol_d4 v = cp_ld4(rB, cp_chunk_off(bs, qd, lane));
    cp_regs_to_lds(which == 0 ? Dt(0) : (which == 1 ? Tc : Dt(1)), qd, lane, v);
  }
  __syncthreads();
  int dcur = 0;                 // Dt(dcur): diagonal tile (s+1, s+1), updated through column s - 1
  int team_target = 0;
  // s = -1 is the lead-in: only the factorisation of diagonal tile 0 (the one call site of the panel code)
  for (int s = -1; s < ntc; ++s) {
    const int i1 = s + 1, i2 = s + 2;
    const cp_tile Ds = Dv(s);
    if (s >= 0) {
      // ---- P1: X1 = A'(s+1, s) L_ss^-T
      {
        chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
        cp_mma<false>(acc, Tc, Ds, wave, lane);
        cp_regs_to_lds(Xb0, wave, lane, acc);
      }
      __syncthreads();
      // ---- P2: A'(s+1, s+1) -= X1 X1^T
      if (i1 < ntc) {
        chol_d4 acc = cp_lds_to_regs(Dt(dcur), wave, lane);
        cp_mma<true>(acc, Xb0, Xb0, wave, lane);
        cp_regs_to_lds(Dt(dcur), wave, lane, acc);
      }
      __syncthreads();
    }
    // ---- P3: wavefront 0 factors the next diagonal tile; the team finishes column s
    if (wave == 0) {
      if (i1 < ntc) cp_potrf(Dt(dcur), min(CH_NB, n - i1*CH_NB), Dv(i1), colbuf, fail);
    } else if (s >= 0) {
      const int tw = wave - 1;
      // (a) publish L_ss^-1 and L(s+1, s)
      {
        const int sd = slot_of[s*ntc + s], s1 = slot_of[i1*ntc + s];
        for (int j = tw; j < 8; j += 3) {
          const int qd = j & 3;
          if (j < 4) cp_st4(rL, cp_chunk_off(sd, qd, lane), cp_lds_to_regs(Ds, qd, lane));
          else cp_st4(rL, cp_chunk_off(s1, qd, lane), cp_lds_to_regs(Xb0, qd, lane));
        }
        CP_DRAIN();
        cp_team_sync(ctl, team_target, lane);
        if (t == 64) { cp_flag_store(flags + sd, done_l); cp_flag_store(flags + s1, done_l); }
      }
      if (i2 <= R && ctl[1]) {
        const bool diag2 = i2 < ntc;
        // (b) row s+2 of the band, updated through column s - 1 by its helpers
        if (t == 64) {
          bool ok = cp_poll(flags + slot_of[i2*ntc + s], want_band, err, code | 4, false);
          ok = ok && cp_poll(flags + slot_of[i2*ntc + i1], want_band, err, code | 5, false);
          if (diag2) ok = ok && cp_poll(flags + slot_of[i2*ntc + i2], want_band, err, code | 6, false);
          if (!ok) ctl[1] = 0;
        }
        cp_team_sync(ctl, team_target, lane);
        if (ctl[1]) {
          for (int j = tw; j < 12; j += 3) {
            const int which = j >> 2, qd = j & 3;
            if (which == 2 && !diag2) continue;
            const int bs = bslot_of[i2*ntc + (which == 0 ? s : (which == 1 ? i1 : i2))];
            const chol_d4 v = cp_ld4(rB, cp_chunk_off(bs, qd, lane));
            cp_regs_to_lds(which == 0 ? T2 : (which == 1 ? Tc : Dt(dcur ^ 1)), qd, lane, v);
          }
          cp_team_sync(ctl, team_target, lane);
          // (c) X2 = A'(s+2, s) L_ss^-T
          for (int j = tw; j < 4; j += 3) {
            chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
            cp_mma<false>(acc, T2, Ds, j, lane);
            cp_regs_to_lds(Xb1, j, lane, acc);
          }
          cp_team_sync(ctl, team_target, lane);
          // (d) A'(s+2, s+1) -= X2 X1^T, A'(s+2, s+2) -= X2 X2^T; publish L(s+2, s)
          const int s2 = slot_of[i2*ntc + s];
          for (int j = tw; j < 12; j += 3) {
            const int which = j >> 2, qd = j & 3;
            if (which == 0) {
              chol_d4 acc = cp_lds_to_regs(Tc, qd, lane);
              cp_mma<true>(acc, Xb1, Xb0, qd, lane);
              cp_regs_to_lds(Tc, qd, lane, acc);
            } else if (which == 1) {
              if (!diag2) continue;
              chol_d4 acc = cp_lds_to_regs(Dt(dcur ^ 1), qd, lane);
              cp_mma<true>(acc, Xb1, Xb1, qd, lane);
              cp_regs_to_lds(Dt(dcur ^ 1), qd, lane, acc);
            } else cp_st4(rL, cp_chunk_off(s2, qd, lane), cp_lds_to_regs(Xb1, qd, lane));
          }
          CP_DRAIN();
          cp_team_sync(ctl, team_target, lane);
          if (t == 64) cp_flag_store(flags + s2, done_l);
        }
      }
    }
    __syncthreads();
    if (!ctl[1]) return;
    dcur ^= 1;
  }
}

// ---- a helper workgroup: one tile of the plan, left-looking -------------------------------------------------------------------
__device__ inline void cp_helper(const CpArgs& a, int q, int hidx, double* lds) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const CpHelper h = a.helpers[hidx];
  cp_tile Pa = (cp_tile)(lds), Pb = (cp_tile)(lds + CP_TILE);
  int* ctl = (int*)(lds + 2*CP_TILE);
  int* flags = a.flags + (size_t)q*a.nslots;
  int* err = a.err + q;
  const __amdgpu_buffer_rsrc_t rL = cp_rsrc(a.Lt + q*a.lt_stride, a.lt_stride*sizeof(double));
  const __amdgpu_buffer_rsrc_t rB = cp_rsrc(a.Bt + q*a.bt_stride, a.bt_stride*sizeof(double));
  const double* S = a.S + q*a.sys_stride;
  const int done_l = a.epoch4 | 2;
  const int code = 0x200 | (hidx << 12);
  // the right-hand-side row's helpers also arm the back-substitution's data-tagged vectors
  if (h.ti == a.ntc && t < 64) {
    double* dst = (t < 32 ? a.xbuf : a.fbuf) + (size_t)q*a.vec_stride + h.tj*CH_NB + (t & 31);
    *reinterpret_cast<unsigned long long*>(dst) = CP_SENT;
  }
  chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (h.in_s) acc = cp_load_A(S, a.n, a.ntc, h.ti, h.tj, wave, lane);
  for (int u = 0; u < h.nupd; ++u) {
    const int2 sl = a.upd[h.upd0 + u];
    if (t == 0) {
      const bool patient = u + 3 < h.nupd;           // far behind the frontier of this tile: poll gently
      bool ok = cp_poll(flags + sl.x, done_l, err, code | 1, patient);
      ok = ok && (sl.y == sl.x || cp_poll(flags + sl.y, done_l, err, code | 2, patient));
      ctl[0] = ok ? 1 : 0;
    }
    __syncthreads();
    if (!ctl[0]) return;
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
  if (h.kind == 1) {          // band tile: the partial sum goes to the critical workgroup
    cp_st4(rB, cp_chunk_off(h.dslot, wave, lane), acc);        // (dslot of a band tile = its band slot)
    CP_DRAIN();
    __syncthreads();
    if (t == 0) cp_flag_store(flags + h.slot, a.epoch4 | 1);
    return;
  }
  // far tile: X = acc L_jj^-T
  if (t == 0) ctl[0] = cp_poll(flags + h.dslot, done_l, err, code | 3, false) ? 1 : 0;
  __syncthreads();
  if (!ctl[0]) return;
  cp_regs_to_lds(Pb, wave, lane, cp_ld4(rL, cp_chunk_off(h.dslot, wave, lane)));
  cp_regs_to_lds(Pa, wave, lane, acc);
  __syncthreads();
  chol_d4 x = {0.0, 0.0, 0.0, 0.0};
  cp_mma<false>(x, Pa, Pb, wave, lane);
  cp_st4(rL, cp_chunk_off(h.slot, wave, lane), x);
  CP_DRAIN();
  __syncthreads();
  if (t == 0) cp_flag_store(flags + h.slot, done_l);
}

constexpr int CP_LDS_DOUBLES = 8*CP_TILE + 64 + 8;
__global__ void __launch_bounds__(CP_THREADS, 2)
k_chol_persist(CpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double cp_lds[];
  const int q = blockIdx.x % a.nsys, role = blockIdx.x / a.nsys;
  if (role == 0) cp_critical(a, q, cp_lds);
  else cp_helper(a, q, role - 1, cp_lds);
}

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
    unsigned it = 0;
    for (;;) {
      bool ready = true;
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = cp_tagged_load(xi + e); ready &= !cp_is_sent(xv[e]); }
      if (ready) break;
      if (++it >= CP_SPIN_MAX) { cp_flag_store(err, 0x400 | (cidx << 12)); return; }
      if ((it & 31) == 31 && cp_flag_load(err) != 0) return;
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
__device__ inline void cp_back_chain(const CpBackArgs& a, int q, double* lds) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int n = a.n, ntc = a.ntc;
  const double* Lt = a.Lt + q*a.lt_stride;
  const double* fb = a.fbuf + (size_t)q*a.vec_stride;
  double* xb = a.xbuf + (size_t)q*a.vec_stride;
  int* err = a.err + q;
  auto N1 = [&](int par) { return (cp_tile)(lds + (par & 1)*CP_TILE); };
  auto Dn = [&](int par) { return (cp_tile)(lds + (2 + (par & 1))*CP_TILE); };
  double* pre = lds + 4*CP_TILE;                // [2][3][64]: per step parity: y - f, and the far-of-near partial sums of two wavefronts (two halves each)
  double* zb = pre + 2*3*64;                    // [32]
  int* ctl = (int*)(zb + 32);
  double* xs = zb + 32 + 8;                     // [ntc*32]
  const int* slot_of = a.slot_of;
  if (t == 0) ctl[0] = 1;
  // prepare step k (tiles and sums of block column k) -- called by wavefronts 1..3 during step k+1 (and by everybody before the loop)
  auto prepare = [&](int k, int w) {
    const int par = k & 1;
    double* P = pre + par*3*64;
    if (w == 3) {
      // L^-1 of block k and tile (k+1, k) -> LDS (row-major), y_k - f_k
      const double* Dg = Lt + (size_t)slot_of[k*ntc + k]*CP_TQ;
      const int s1 = (k + 1 < ntc) ? slot_of[(k + 1)*ntc + k] : -1;
      for (int qd = 0; qd < 4; ++qd) {
        const chol_d4 v = *reinterpret_cast<const chol_d4*>(Dg + (qd*64 + lane)*4);
        cp_regs_to_lds(Dn(par), qd, lane, v);
        chol_d4 u = {0.0, 0.0, 0.0, 0.0};
        if (s1 >= 0) u = *reinterpret_cast<const chol_d4*>(Lt + (size_t)s1*CP_TQ + (qd*64 + lane)*4);
        cp_regs_to_lds(N1(par), qd, lane, u);
      }
      if (lane < 32) {
        const int sy = slot_of[ntc*ntc + k];
        double y = Lt[(size_t)sy*CP_TQ + cp_tq_index(0, lane)];
        // far sum of the column (a column without far tiles has none)
        if (a.far_start[ntc - 1 - k + 1] > a.far_start[ntc - 1 - k]) {
          double f; unsigned it = 0;
          for (;;) {
            f = cp_tagged_load(fb + k*CH_NB + lane);
            if (!cp_is_sent(f)) break;
            if (++it >= CP_SPIN_MAX) { cp_flag_store(err, 0x500 | (k << 12)); ctl[0] = 0; break; }
            if ((it & 31) == 31 && cp_flag_load(err) != 0) { ctl[0] = 0; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          y -= f;
        }
        P[lane] = y;
      }
    } else {
      // w = 1, 2: the tile d = w + 1 rows above: partial[c] over the two 16-row halves
      const int d = w + 1, i = k + d;
      const int c = lane & 31, hh = lane >> 5;
      double s = 0.0;
      const int sl = (i < ntc) ? slot_of[i*ntc + k] : -1;
      if (sl >= 0) {
        const double* T = Lt + (size_t)sl*CP_TQ;
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = T[cp_tq_index(16*hh + r, c)];
#pragma unroll
        for (int r = 0; r < 16; ++r) s = __builtin_fma(v[r], xs[i*CH_NB + 16*hh + r], s);
      }
      P[w*64 + lane] = s;
    }
  };
  __syncthreads();
  if (ntc > 0) { if (wave >= 1) prepare(ntc - 1, wave); }
  __syncthreads();
  for (int k = ntc - 1; k >= 0; --k) {
    if (!ctl[0]) return;
    const int par = k & 1;
    if (wave == 0) {
      const double* P = pre + par*3*64;
      const int c = lane & 31, hh = lane >> 5;
      double s = 0.0;
      if (k + 1 < ntc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s = __builtin_fma(N1(par)[16*hh + r][c], xs[(k + 1)*CH_NB + 16*hh + r], s);
      }
      s += __shfl_xor(s, 32, 64);
      const double z = P[c] - ((P[64 + c] + P[64 + 32 + c]) + (P[128 + c] + P[128 + 32 + c])) - s;
      if (lane < 32) zb[lane] = z;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double x = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) x = __builtin_fma(Dn(par)[16*hh + cc][c], zb[16*hh + cc], x);
      x += __shfl_xor(x, 32, 64);
      const int nbe = min(CH_NB, n - k*CH_NB);
      if (lane < 32) {
        if (lane >= nbe) x = 0.0;
        xs[k*CH_NB + lane] = x;
        cp_tagged_store(xb + k*CH_NB + lane, x);
      }
    } else if (k > 0) prepare(k - 1, wave);
    __syncthreads();
  }
  if (!ctl[0]) return;
  for (int i = t; i < n; i += CP_THREADS) a.xout[q*a.sys_stride + i] = xs[i];
}

__global__ void __launch_bounds__(CP_THREADS)
k_chol_back2(CpBackArgs a) {
  extern __shared__ __attribute__((aligned(16))) double cp_lds[];
  const int q = blockIdx.x % a.nsys, role = blockIdx.x / a.nsys;
  if (role == 0) cp_back_chain(a, q, cp_lds);
  else cp_back_far(a, q, role - 1, cp_lds);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
struct CholPersist {
  static constexpr int max_sys = 4;
  int n = 0, ntc = 0, nslots = 0, nbslots = 0, nhelpers = 0, nfarcols = 0;
  bool ok = false;
  std::vector<int> slot_of, bslot_of;
  std::vector<CpHelper> helpers; std::vector<int2> upd;
  std::vector<int> far_start, far_slot, far_row;
  int *d_slot_of = nullptr, *d_bslot_of = nullptr, *d_flags = nullptr, *d_err = nullptr, *d_far_start = nullptr, *d_far_slot = nullptr, *d_far_row = nullptr;
  CpHelper* d_helpers = nullptr; int2* d_upd = nullptr;
  double *d_Lt = nullptr, *d_Bt = nullptr, *d_x = nullptr, *d_f = nullptr;
  size_t lt_stride = 0, bt_stride = 0; int vec_stride = 0;
  unsigned epoch = 0;
  ~CholPersist() { release(); }
  void release() {
    void* ps[] = {d_slot_of, d_bslot_of, d_flags, d_err, d_far_start, d_far_slot, d_far_row, d_helpers, d_upd, d_Lt, d_Bt, d_x, d_f};
    for (void* p : ps) if (p) (void)hipFree(p);
    d_slot_of = d_bslot_of = d_flags = d_err = d_far_start = d_far_slot = d_far_row = nullptr; d_helpers = nullptr; d_upd = nullptr;
    d_Lt = d_Bt = d_x = d_f = nullptr; ok = false;
  }
  template <class T> static int up(T*& d, const std::vector<T>& v) {
    if (hipMalloc((void**)&d, sizeof(T)*std::max<size_t>(v.size(), 1)) != hipSuccess) return -1;
    if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T)*v.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
  }
  // pattern: ntc x ntc lower-triangular tile occupancy of S (empty = dense); in_old: the tiles the assembly writes (ti << 16 | tj),
  // with the right-hand side inside block row n / 32 as ba_chol.h's plan has it
  int build(int n_, const std::vector<unsigned char>& pattern, const std::vector<int>& old_tiles) {
    release();
    n = n_; ntc = (n + CH_NB - 1)/CH_NB;
    if (ntc < 1) return 0;
    const int R = ntc, nr = ntc + 1;
    std::vector<unsigned char> P((size_t)nr*ntc, 0), inS((size_t)nr*ntc, 0);
    for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) P[(size_t)i*ntc + j] = pattern.empty() ? 1 : pattern[(size_t)i*ntc + j];
    for (int tp : old_tiles) { const int i = tp >> 16, j = tp & 0xffff; if (i < ntc && j < ntc) inS[(size_t)i*ntc + j] = 1; }
    if (old_tiles.empty()) for (int i = 0; i < ntc; ++i) for (int j = 0; j <= i; ++j) inS[(size_t)i*ntc + j] = 1;      // dense test matrices: S is all there
    for (int i = 0; i < ntc; ++i) for (int j = std::max(0, i - CP_W + 1); j <= i; ++j) P[(size_t)i*ntc + j] = 1;      // the band is the critical workgroup's: always there
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
      if (i - j < CP_W) bslot_of[(size_t)i*ntc + j] = nbslots++;
    }
    // helpers, in the order of the column that completes them: far tile (i, j) -> j; band tile (i, j) -> i - CP_W
    struct Key { int key, kind, i, j; };
    std::vector<Key> ks;
    for (int i = 0; i < nr; ++i) for (int j = 0; j < ntc && j <= i; ++j) if (P[(size_t)i*ntc + j]) {
      const bool band = i - j < CP_W;
      ks.push_back({band ? i - CP_W : j, band ? 1 : 0, i, j});
    }
    std::stable_sort(ks.begin(), ks.end(), [](const Key& x, const Key& y) { if (x.key != y.key) return x.key < y.key; if (x.kind != y.kind) return x.kind < y.kind; return x.i < y.i; });
    helpers.clear(); upd.clear();
    for (const Key& k : ks) {
      CpHelper h; h.ti = k.i; h.tj = k.j; h.kind = k.kind; h.slot = slot_of[(size_t)k.i*ntc + k.j];
      h.dslot = k.kind ? bslot_of[(size_t)k.i*ntc + k.j] : slot_of[(size_t)k.j*ntc + k.j];
      h.in_s = inS[(size_t)k.i*ntc + k.j];
      h.upd0 = (int)upd.size();
      const int mlast = k.kind ? std::min(k.j - 1, k.i - CP_W) : k.j - 1;
      for (int m = 0; m <= mlast; ++m) {
        const int sa = slot_of[(size_t)k.i*ntc + m], sb = slot_of[(size_t)k.j*ntc + m];
        if (sa >= 0 && sb >= 0) upd.push_back(make_int2(sa, sb));
      }
      h.nupd = (int)upd.size() - h.upd0;
      helpers.push_back(h);
    }
    nhelpers = (int)helpers.size();
    // back-substitution: far tiles per block column, columns right to left (role 1 + idx <-> column ntc - 1 - idx)
    far_start.assign(1, 0); far_slot.clear(); far_row.clear();
    for (int idx = 0; idx < ntc; ++idx) {
      const int j = ntc - 1 - idx;
      for (int i = ntc - 1; i > j + CP_BACK_NEAR; --i) if (P[(size_t)i*ntc + j]) { far_slot.push_back(slot_of[(size_t)i*ntc + j]); far_row.push_back(i); }
      far_start.push_back((int)far_slot.size());
    }
    lt_stride = (size_t)nslots*CP_TQ; bt_stride = (size_t)std::max(nbslots, 1)*CP_TQ; vec_stride = ntc*CH_NB;
    if (up(d_slot_of, slot_of) || up(d_bslot_of, bslot_of) || up(d_helpers, helpers) || up(d_upd, upd) || up(d_far_start, far_start) || up(d_far_slot, far_slot) || up(d_far_row, far_row)) return -1;
    if (hipMalloc((void**)&d_Lt, sizeof(double)*lt_stride*max_sys) != hipSuccess || hipMalloc((void**)&d_Bt, sizeof(double)*bt_stride*max_sys) != hipSuccess ||
        hipMalloc((void**)&d_flags, sizeof(int)*(size_t)nslots*max_sys) != hipSuccess || hipMalloc((void**)&d_err, sizeof(int)*max_sys*2) != hipSuccess ||
        hipMalloc((void**)&d_x, sizeof(double)*vec_stride*max_sys) != hipSuccess || hipMalloc((void**)&d_f, sizeof(double)*vec_stride*max_sys) != hipSuccess) return -1;
    if (hipMemset(d_flags, 0, sizeof(int)*(size_t)nslots*max_sys) != hipSuccess || hipMemset(d_err, 0, sizeof(int)*max_sys*2) != hipSuccess) return -1;
    ok = true;
    return 0;
  }
};


// one launch: factor systems [q0, q0 + nsys) of the batch (S + q * sys_stride); L tiles, L_kk^-1 and y land in P.d_Lt
inline int chol_persist_factor(hipStream_t st, CholPersist& P, const double* S, int* fail, int nsys, size_t sys_stride, int q0) {
  static std::atomic<unsigned long long> attr_mask{0};
  int dev = 0; (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_mask.load(std::memory_order_relaxed) & bit)) {
    if (hipFuncSetAttribute((const void*)k_chol_persist, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS_DOUBLES*(int)sizeof(double)) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_chol_back2, hipFuncAttributeMaxDynamicSharedMemorySize, (4*CP_TILE + 512 + CH_SOLVE_MAX + 64)*(int)sizeof(double)) != hipSuccess) return -1;
    attr_mask.fetch_or(bit, std::memory_order_relaxed);
  }
  CpArgs a;
  a.S = S + q0*sys_stride; a.sys_stride = sys_stride; a.n = P.n; a.ntc = P.ntc; a.nsys = nsys; a.nhelpers = P.nhelpers;
  a.slot_of = P.d_slot_of; a.bslot_of = P.d_bslot_of; a.helpers = P.d_helpers; a.upd = P.d_upd;
  a.Lt = P.d_Lt + q0*P.lt_stride; a.lt_stride = P.lt_stride; a.Bt = P.d_Bt + q0*P.bt_stride; a.bt_stride = P.bt_stride;
  a.flags = P.d_flags + (size_t)q0*P.nslots; a.nslots = P.nslots; a.err = P.d_err + q0; a.fail = fail + q0;
  a.epoch4 = (int)((++P.epoch) << 2);
  a.xbuf = P.d_x + (size_t)q0*P.vec_stride; a.fbuf = P.d_f + (size_t)q0*P.vec_stride; a.vec_stride = P.vec_stride;
  hipLaunchKernelGGL(k_chol_persist, dim3((1 + P.nhelpers)*nsys), dim3(CP_THREADS), CP_LDS_DOUBLES*sizeof(double), st, a);
  return 0;
}
// the second launch: x = L^-T y into row n of S (xout = S + n n)
inline int chol_persist_back(hipStream_t st, CholPersist& P, double* S, int nsys, size_t sys_stride, int q0) {
  CpBackArgs a;
  a.Lt = P.d_Lt + q0*P.lt_stride; a.lt_stride = P.lt_stride; a.slot_of = P.d_slot_of; a.n = P.n; a.ntc = P.ntc; a.nsys = nsys; a.ncols = P.ntc;
  a.far_start = P.d_far_start; a.far_slot = P.d_far_slot; a.far_row = P.d_far_row;
  a.xbuf = P.d_x + (size_t)q0*P.vec_stride; a.fbuf = P.d_f + (size_t)q0*P.vec_stride; a.vec_stride = P.vec_stride;
  a.xout = S + q0*sys_stride + (size_t)P.n*P.n; a.sys_stride = sys_stride; a.err = P.d_err + q0;
  const size_t lds = (size_t)(4*CP_TILE + 2*3*64 + 32 + 8 + P.ntc*CH_NB)*sizeof(double);
  hipLaunchKernelGGL(k_chol_back2, dim3((1 + P.ntc)*nsys), dim3(CP_THREADS), lds, st, a);
  return 0;
}

}  // namespace mcp
