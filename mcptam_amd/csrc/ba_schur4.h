// ba_schur4.h -- point elimination for ALL systems of a multi-lambda batch in one workgroup (gfx950, fp64).
//
// What it replaces: g2o's BlockSolver::buildSystem + the Schur complement it would form if the points were marginalised
// (the reference does not even do that: src/ChainBundle.cc:1150-1158 hands the un-marginalised system to CHOLMOD), for the
// up to four damped systems S(lambda_q) = U - W (V + lambda_q I)^-1 W^T a trial solve carries (ba_solver.hip solve_trial).
//
// Round 4 launched one workgroup per (group, system): the four lambda-copies of a group fetched W, V, g and walked the
// index hops four times, staged four Z = W L arrays, and 3140 workgroups queued 12 deep per compute unit.  Here a
// workgroup owns a group for the whole batch:
//   * W of a 16-point chunk is staged ONCE, raw, in LDS (Wd[row][3 p + c], row = 6 local pose + r);
//   * wavefront q is system q.  It forms its operand Y_q = W (V + lambda_q I)^-1 on the fly: MFMA step s = 3 m + c of
//     v_mfma_f64_16x16x4 gives lane (i, kq) the k-index (point 4 m + kq, component c), so a lane holds the three W values
//     of one (row, point), multiplies them by that point's 3x3 inverse (9 fused multiply-adds on the vector pipe, beside
//     the matrix pipe) and feeds y_c as A, the raw w_c of the column tile as B.  The raw tiles are read once per m and
//     kept in registers for every tile pair: 0.3 LDS reads per MFMA instead of 2;
//   * W V^-1 g needs no pass of its own: rhs_q(row) += y . g on the same lane, summed over the four k-lanes at the end;
//   * a chunk that leaves a 16-row tile empty (most chunks of a real map see 8 poses = 3 of 5 tiles) skips its pairs;
//   * two barriers per chunk (the scatter does not wait for the inverse any more; a column triple of Wd is only ever
//     written by one quarter-wavefront, so "clear the old rows, write the new ones" needs no barrier in between);
//   * the accumulators go from registers straight to the staged blocks (no pass through LDS, no barrier).
// Groups of the fast path hold <= 13 poses (78 rows = 5 tiles, 15 tile pairs = 120 accumulator registers per lane);
// Prepare() closes groups accordingly and falls back to k_schur_group for a map in which one point alone sees 14..16.
#pragma once
#include "ba_group.h"

namespace mcp {

#ifndef S4_DEPHASE
#define S4_DEPHASE 0      // x 2048 cycles of initial delay for the workgroup in the odd wave slot of a compute unit (0 = off)
#endif
#ifndef S4_WAVES
#define S4_WAVES 2        // wavefronts per SIMD the register allocation is held to (256 registers: two workgroups per compute unit)
#endif
constexpr int S4_NT = 5;                            // 16-row tiles of a group
constexpr int S4_ROWS = 16*S4_NT;                   // 80 (+ one trash row: absent W rows are "written" there, no branches in the scatter)
constexpr int S4_LMAX = S4_ROWS/6;                  // 13 poses
constexpr int S4_NPAIR = S4_NT*(S4_NT + 1)/2;       // 15
constexpr int S4_LD = 49;                           // row stride of Wd (doubles)
constexpr int S4_EPT = (6*S4_LMAX + 15)/16;         // W rows per thread of the scatter (16 threads per point)
constexpr int S4_NBLK = S4_LMAX*(S4_LMAX + 1)/2;    // 91 local pose pairs
constexpr int S4_OFF_VI = (S4_ROWS + 1)*S4_LD;      // (V + lambda_q)^-1 of the group's points  [MAX_SYS][GRP_PTS][6]
constexpr int S4_OFF_G = S4_OFF_VI + MAX_SYS*GRP_PTS*6;     // g of the group's points  [GRP_PTS][3]
constexpr int S4_OFF_IDX = S4_OFF_G + GRP_PTS*3;            // per thread and chunk: first incidence, 6 x incidences  [4][256] int2
constexpr int S4_OFF_INT = S4_OFF_IDX + 4*256;
constexpr int S4_NINT = 92 + 16 + 8;                // bofs[92] rdst[16] tmask[4] smask[1]
constexpr size_t S4_LDS_BYTES = (size_t)S4_OFF_INT*sizeof(double) + (size_t)S4_NINT*sizeof(int);

#ifdef MCP_SCH_PROF
#define S4_T0() unsigned long long s4_t = clock64(), s4_a[8] = {0,0,0,0,0,0,0,0}
#define S4_LAP(i) do { const unsigned long long n_ = clock64(); s4_a[i] += n_ - s4_t; s4_t = n_; } while (0)
#define S4_OUT() do { if ((blockIdx.x & 127) == 100 && blockIdx.x < 1024 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_sch_prof[(blockIdx.x >> 7)*8 + i_] = s4_a[i_]; } while (0)
#else
#define S4_T0() do {} while (0)
#define S4_LAP(i) do {} while (0)
#define S4_OUT() do {} while (0)
#endif

// Which part of the batch a wavefront computes.  Every accumulator (tile pair of a system) and every right-hand-side tile belongs
// to exactly one wavefront and receives the same sequence of operations whatever the width of the batch: a system comes out bit for
// bit the same from a launch that carries it alone, beside one, or beside three others (the speculative batch may be split over
// streams and launches: ba_solver.hip solve_trial).
//   4 systems: wavefront q = system q.   3: likewise, wavefront 3 only helps staging.   2: two wavefronts per system, tile pairs and
//   right-hand-side tiles dealt by parity.   1: four wavefronts, dealt modulo 4.
struct S4Role { int q, pmask, rmask; };
__device__ inline S4Role s4_role(int nsys, int wave) {
  S4Role r{0, 0, 0};
  if (nsys >= 3) { if (wave < nsys) { r.q = wave; r.pmask = (1 << S4_NPAIR) - 1; r.rmask = (1 << S4_NT) - 1; } }
  else if (nsys == 2) { r.q = wave >> 1; for (int p = 0; p < S4_NPAIR; ++p) if ((p & 1) == (wave & 1)) r.pmask |= 1 << p; for (int t = 0; t < S4_NT; ++t) if ((t & 1) == (wave & 1)) r.rmask |= 1 << t; }
  else { for (int p = 0; p < S4_NPAIR; ++p) if ((p & 3) == wave) r.pmask |= 1 << p; for (int t = 0; t < S4_NT; ++t) if ((t & 3) == wave) r.rmask |= 1 << t; }
  return r;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S4_WAVES, S4_WAVES)))
k_schur4(DevProblem P, int nsys, const int* __restrict__ g_order /* launch order of the groups (heaviest first), or null */,
         const double* __restrict__ V, const double* __restrict__ g, const double* __restrict__ W,
         double* __restrict__ Vinv, double* __restrict__ stS, double* __restrict__ str, int* __restrict__ fail, SysBatch sb) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* const Wd = lds;
  double* const Vi = lds + S4_OFF_VI;
  double* const gt = lds + S4_OFF_G;
  int2* const idx = reinterpret_cast<int2*>(lds + S4_OFF_IDX);
  int* const bofs = reinterpret_cast<int*>(lds + S4_OFF_INT);
  int* const rdst = bofs + 92;
  int* const tmask = rdst + 16;
  const int grp = g_order ? g_order[blockIdx.x] : (int)blockIdx.x;
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int sp0 = P.g_sp0[grp], sp1 = P.g_sp0[grp + 1];
  const int b0 = P.g_blk0[grp], nb = P.g_blk0[grp + 1] - b0;
  const int nch = (sp1 - sp0 + 15) >> 4;
  const int pl = t >> 4, sub = t & 15;
  const S4Role role = s4_role(nsys, wave);
  const int wq = __builtin_amdgcn_readfirstlane(role.q), pmask = __builtin_amdgcn_readfirstlane(role.pmask), rmask = __builtin_amdgcn_readfirstlane(role.rmask);
  S4_T0();
  // ---- prologue.  (a) every thread resolves the incidence range of its point in each chunk (one exposed index hop per group; kept in
  // LDS, read back by the same thread) and requests the first chunk's W rows; (b) lane = point: every working wavefront inverts the
  // 64 (V + lambda_q I) of ITS system at once (-> Vinv_q in global memory for the back-substitution, and LDS); (c) tables of the
  // flush; Wd zero-filled once
  int i0c = 0, cntc = 0;
  // (the flush tables' entries are requested here, with everything else that hangs off the group's ranges, and land in LDS behind the barrier)
  int bp_t = 0, bd_t = 0, rd_t = -1;
  if (t < nb) { bp_t = P.blk_pair[b0 + t]; bd_t = P.blk_dst[b0 + t]; }
  if (t >= 64 && t < 64 + GRP_LMAX) rd_t = P.rhs_dst[grp*GRP_LMAX + (t - 64)];
  const int spl = sp0 + lane, splc = spl < sp1 ? spl : sp0;          // lane = point of the group (b)
  int lpt = P.sp_unk[splc];
  {
    // (all loads of the four chunks issued before any is looked at: one round trip, not eight)
    int ia[4], ib[4]; unsigned char big[4]; bool ok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int sp = sp0 + 16*c + pl;
      ok[c] = sp < sp1;
      const int spc = ok[c] ? sp : sp0;
      big[c] = P.sp_big[spc]; ia[c] = P.sp_i[spc]; ib[c] = P.sp_i[spc + 1];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool use = ok[c] && !big[c];
      const int i0 = use ? ia[c] : 0, cnt = use ? (ib[c] - ia[c])*6 : 0;
      idx[c*256 + t] = make_int2(i0, cnt);
      if (c == 0) { i0c = i0; cntc = cnt; }
    }
  }
  double pw[S4_EPT][3]; int plp[S4_EPT];            // the next chunk's W rows and the local pose of each (raw: nothing here is looked at before the scatter)
  auto prefetch = [&](int i0, int cnt) {
#pragma unroll
    for (int k = 0; k < S4_EPT; ++k) {
      const int it = sub + 16*k;
      plp[k] = -1;
      if (it < cnt) {
        const int inc = i0 + it/6;
        const double* Wr = W + 18*(size_t)inc + 3*(it%6);
        pw[k][0] = Wr[0]; pw[k][1] = Wr[1]; pw[k][2] = Wr[2];
        plp[k] = P.inc_lp[inc];
      }
    }
  };
  prefetch(i0c, cntc);
  if (spl >= sp1) lpt = -1;
  if (pmask | rmask) {
    double I6[6] = {0, 0, 0, 0, 0, 0}, g3[3] = {0, 0, 0};
    if (lpt >= 0) {
      double v6[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) v6[k] = V[6*(size_t)lpt + k];
      g3[0] = g[3*(size_t)lpt]; g3[1] = g[3*(size_t)lpt + 1]; g3[2] = g[3*(size_t)lpt + 2];
      if (!inv_sym3(v6, sb.lambda[wq], I6)) atomicOr(fail + wq, 1);
      double* vo = Vinv + wq*sb.vstride + 6*(size_t)lpt;       // (wavefronts that share a system store the same numbers)
#pragma unroll
      for (int k = 0; k < 6; ++k) vo[k] = I6[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Vi[(wq*GRP_PTS + lane)*6 + k] = I6[k];
    if (wave == 0) { gt[3*lane] = g3[0]; gt[3*lane + 1] = g3[1]; gt[3*lane + 2] = g3[2]; }
  }
  for (int i = t; i < (S4_ROWS + 1)*S4_LD; i += 256) Wd[i] = 0.0;
  if (t < 92) bofs[t] = -1;
  if (t < 5) tmask[t] = 0;                           // ([4]: tiles that hold a pose of the group at all)
  if (t >= 64 && t < 64 + GRP_LMAX) rdst[t - 64] = rd_t;
  __syncthreads();
  if (t < nb) bofs[(bp_t >> 4)*((bp_t >> 4) + 1)/2 + (bp_t & 15)] = bd_t*36;
  if (t >= 64 && t < 64 + S4_LMAX && rd_t >= 0) atomicOr(&tmask[4], (1 << ((6*(t - 64)) >> 4)) | (1 << ((6*(t - 64) + 5) >> 4)));
  sch_d4 acc[S4_NPAIR];
#pragma unroll
  for (int i = 0; i < S4_NPAIR; ++i) acc[i] = (sch_d4){0.0, 0.0, 0.0, 0.0};
  double rp[S4_NT];
#pragma unroll
  for (int i = 0; i < S4_NT; ++i) rp[i] = 0.0;
  unsigned long long orow = 0x5050505050ull;        // rows written for the chunk before (bytes; 80 = the trash row)
#if S4_DEPHASE
  // Two workgroups share a compute unit and are dispatched together: left alone they walk their phases in lockstep -- both stage,
  // both multiply (the matrix pipe takes twice as long), both flush.  The one in the odd wave slot starts its chunk loop half a
  // chunk period late, so that one workgroup's products run beside the other's staging.
  if ((__builtin_amdgcn_s_getreg((3 << 11) | 4 /* HW_REG_HW_ID, WAVE_ID[3:0] */) & 1) != 0) {
#pragma unroll 1
    for (int i = 0; i < S4_DEPHASE; ++i) __builtin_amdgcn_s_sleep(32);
  }
#endif
  S4_LAP(0);
  for (int c = 0; c < nch; ++c) {
    // ---- stage the chunk: the rows of the chunk before are cleared by the threads that wrote them (same quarter-wavefront as the
    // writers of the new rows of that column triple: program order is enough), then the prefetched rows land
    {
      int tm = 0;
      unsigned long long nrow = 0;
#pragma unroll
      for (int k = 0; k < S4_EPT; ++k) {
        double* wd = Wd + ((int)(orow >> (8*k)) & 0xff)*S4_LD + 3*pl;
        wd[0] = 0.0; wd[1] = 0.0; wd[2] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < S4_EPT; ++k) {
        const int rn = plp[k] >= 0 ? 6*plp[k] + (sub + 16*k)%6 : S4_ROWS;
        double* wd = Wd + rn*S4_LD + 3*pl;
        wd[0] = pw[k][0]; wd[1] = pw[k][1]; wd[2] = pw[k][2];
        tm |= 1 << (rn >> 4);
        nrow |= (unsigned long long)rn << (8*k);
      }
      orow = nrow;
      // which tiles the chunk fills: OR over the wavefront by ballots, one LDS atomic per wavefront (left to the compiler, the
      // per-lane atomicOr became a scalar loop over the 64 lanes: 2-3 k cycles per chunk)
      int wm = 0;
#pragma unroll
      for (int b = 0; b < S4_NT; ++b) wm |= (__ballot((tm >> b) & 1) != 0ull) ? (1 << b) : 0;
      if (lane == 0 && wm) atomicOr(&tmask[c], wm);
    }
    S4_LAP(1);
    __syncthreads();
    S4_LAP(2);
    if (c + 1 < nch) { const int2 ix = idx[(c + 1)*256 + t]; prefetch(ix.x, ix.y); }      // in flight during the matrix-core phase
    if (pmask | rmask) {
      const int i = lane & 15, kq = lane >> 4;
      const int tm = __builtin_amdgcn_readfirstlane(tmask[c]);
#pragma unroll 1
      for (int m = 0; m < 4; ++m) {
        const int p = 16*c + 4*m + kq;                     // point of the group
        const double* vip = Vi + (wq*GRP_PTS + p)*6;
        const double v0 = vip[0], v1 = vip[1], v2 = vip[2], v3 = vip[3], v4 = vip[4], v5 = vip[5];
        const double g0 = gt[3*p], g1 = gt[3*p + 1], g2 = gt[3*p + 2];
        double wc[S4_NT][3];
#pragma unroll
        for (int tr = 0; tr < S4_NT; ++tr) {
          if (!((tm >> tr) & 1)) continue;
          const double* wr = Wd + (16*tr + i)*S4_LD + 3*(4*m + kq);
          const double w0 = wr[0], w1 = wr[1], w2 = wr[2];
          wc[tr][0] = w0; wc[tr][1] = w1; wc[tr][2] = w2;
          const int rowpairs = ((1 << (tr + 1)) - 1) << (tr*(tr + 1)/2);
          if (!((pmask & rowpairs) | ((rmask >> tr) & 1))) continue;
          const double y0 = w0*v0 + w1*v1 + w2*v2;
          const double y1 = w0*v1 + w1*v3 + w2*v4;
          const double y2 = w0*v2 + w1*v4 + w2*v5;
          if ((rmask >> tr) & 1) rp[tr] += y0*g0 + y1*g1 + y2*g2;
#pragma unroll
          for (int tc = 0; tc <= tr; ++tc) {
            if (!((tm >> tc) & 1) || !((pmask >> (tr*(tr + 1)/2 + tc)) & 1)) continue;
            sch_d4 a = acc[tr*(tr + 1)/2 + tc];
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, wc[tc][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, wc[tc][1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(y2, wc[tc][2], a, 0, 0, 0);
            acc[tr*(tr + 1)/2 + tc] = a;
          }
        }
      }
    }
    S4_LAP(3);
    __syncthreads();
    S4_LAP(4);
  }
  // ---- flush: a wavefront writes the accumulators it owns to the group's staged blocks (block (la, lb), la >= lb, row-major, rows =
  // pose la; of a diagonal block only the lower triangle is written -- k_assemble / k_assemble_long never read the upper entries)
  // and its tiles of the local right-hand side; k_assemble subtracts them from the system
  if (pmask | rmask) {
    double* const stq = stS + wq*sb.ststride;
    double* const srq = str + wq*sb.strstride;
    const int col_l = lane & 15, rq = lane >> 4;
    // every staged block of the group is written, also those no free point of the group contributes to (poses only co-observed
    // through FIXED points: a zero block, not whatever the staging array held): all tiles that hold a pose of the group
    const int gm = __builtin_amdgcn_readfirstlane(tmask[4]);
#pragma unroll
    for (int tr = 0; tr < S4_NT; ++tr) {
      if (!((gm >> tr) & 1)) continue;
#pragma unroll
      for (int tc = 0; tc <= tr; ++tc) {
        if (!((gm >> tc) & 1) || !((pmask >> (tr*(tr + 1)/2 + tc)) & 1)) continue;
        const sch_d4 a = acc[tr*(tr + 1)/2 + tc];
        const int cc = 16*tc + col_l, lb = cc/6, c6 = cc - 6*lb;
        int o[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {                    // the four look-ups of the pair in flight together (entry 91 of the table stays -1)
          const int r = 16*tr + rq + 4*gq, la = r/6;
          const bool use = la < S4_LMAX && cc <= r;         // (not rows 78, 79; not the mirrored half of a diagonal tile)
          o[gq] = bofs[use ? la*(la + 1)/2 + lb : 91];
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int r = 16*tr + rq + 4*gq, r6 = r - 6*(r/6);
          if (o[gq] >= 0) stq[(size_t)o[gq] + 6*r6 + c6] = a[gq];
        }
      }
    }
#pragma unroll
    for (int tr = 0; tr < S4_NT; ++tr) {
      if (!((rmask >> tr) & 1)) continue;
      double v = rp[tr];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int row = 16*tr + (lane & 15), la = row/6;
      if (lane < 16 && la < S4_LMAX) { const int d = rdst[la]; if (d >= 0) srq[(size_t)d*6 + (row - 6*la)] = v; }
    }
  }
  S4_LAP(5);
  S4_OUT();
}

}  // namespace mcp
