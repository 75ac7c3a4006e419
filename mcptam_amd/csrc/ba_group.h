// ba_group.h -- group-tiled linearisation and point elimination (gfx950, fp64).
//
// The host packs the map points (sorted by the pose their coordinates are expressed in) into
// groups of <= 64 points that touch <= 16 free poses.  Inside a group every pose-pose block of
// the normal equations lives in a 96x96 local tile, so the 400k-measurement accumulations that
// g2o performs edge by edge (BaseMultiEdge::constructQuadraticForm, called from
// src/ChainBundle.cc via optimizer.optimize) never touch HBM with an atomic:
//   k_linearize_group : one wavefront per group, one lane per point; Jacobians of
//                       EdgeChainMeas::linearizeOplus (ChainBundle.cc:449-749); V, g, W go out
//                       with plain stores, the pose blocks are summed in an LDS tile (ds_add_f64)
//                       and flushed once per group.
//   k_schur_group     : S -= W V^-1 W^T as a dense zero-filled (96 x 3*16)(3*16 x 96) product
//                       per 16-point chunk on the fp64 matrix cores (v_mfma_f64_16x16x4).
//   k_assemble        : the reduced system, tile by tile, from the groups' staged blocks.
//
// Fixed-order accumulation (hard part 2): a group never adds into the global system.  It writes the
// structurally non-zero 6x6 blocks of its local tile (which local pose pairs those are is known at Prepare()) to its own
// slots of a staging array, and k_assemble builds every entry of S as the sum of its staged contributions in ascending
// group order -- one writer per entry, no atomics, bit-identical from run to run.
#pragma once
#include "ba_kernels.h"

namespace mcp {

// packed lower-triangular index of the local tile
__device__ inline int tri(int r, int c) { return r*(r + 1)/2 + c; }
constexpr int GRP_TRI = GRP_DOF*(GRP_DOF + 1)/2;     // 4656

// Flush of a group's local tile (packed lower triangle in LDS) and local right-hand side to staging.  Block `slot` of the
// group is the 6x6 block (la, lb), la >= lb, of the local tile, row-major, rows = pose la (for la == lb only the lower
// triangle is meaningful; the upper entries are written as zeros).  It lands at staged block blk_dst[..]: the staging array
// is ordered by DESTINATION (all contributions to one global pose pair are consecutive, ascending group), so that
// k_assemble sums a contiguous run.  The local rhs of pose slot la lands at row rhs_dst[grp*GRP_LMAX + la] likewise.
// Consecutive threads write consecutive doubles of a block.  NT = threads of the workgroup.
template <int NT>
__device__ inline void flush_group_blocks(const double* Sl, const double* bl, int grp, const DevProblem& P,
                                          double* __restrict__ st_blocks, double* __restrict__ st_rhs) {
  const int b0 = P.g_blk0[grp], nb = P.g_blk0[grp + 1] - b0;
  for (int e = threadIdx.x; e < nb*36; e += NT) {
    const int slot = e/36, w = e - 36*slot;
    const int pr = P.blk_pair[b0 + slot], la = pr >> 4, lb = pr & 15;
    const int r = w/6, c = w - 6*r;
    st_blocks[(size_t)P.blk_dst[b0 + slot]*36 + w] = (la == lb && c > r) ? 0.0 : Sl[tri(6*la + r, 6*lb + c)];
  }
  for (int i = threadIdx.x; i < GRP_DOF; i += NT) {
    const int d = P.rhs_dst[grp*GRP_LMAX + i/6];
    if (d >= 0) st_rhs[(size_t)d*6 + i%6] = bl[i];
  }
}

// k_linearize_group keeps only the group's structurally non-zero blocks in LDS, in the order of the group's block list
// (36 doubles per block, row-major, rows = the larger local pose; for a diagonal block only the lower triangle is ever
// added to, the upper entries stay zero): the tile is the staging layout, so the flush is a straight copy, the LDS a group
// needs is what its blocks need (22 blocks = 6 KB on average at the metric size instead of the 37 KB of the packed
// 96 x 96 triangle), and more groups share a compute unit.  pslot[la*16 + lb] (la >= lb) = block slot of the local pair.
// (sdst / srd: the group's blk_dst and rhs_dst entries, fetched into LDS at the head of the kernel.  Read from global memory here, every
//  pass of the loop was a look-up, a wait and a store: 70 k of a group's 305 k cycles at the metric size.)
template <int NT>
__device__ inline void flush_compact_blocks(const double* Sc, const double* bl, int nb, const int* sdst, const int* srd,
                                            double* __restrict__ st_blocks, double* __restrict__ st_rhs) {
  for (int e = threadIdx.x; e < nb*36; e += NT) {
    const int slot = e/36, w = e - 36*slot;
    st_blocks[(size_t)sdst[slot]*36 + w] = Sc[e];
  }
  for (int i = threadIdx.x; i < GRP_DOF; i += NT) {
    const int d = srd[i/6];
    if (d >= 0) st_rhs[(size_t)d*6 + i%6] = bl[i];
  }
}

#if defined(LIN_ABL) && LIN_ABL == 1
__device__ inline void lds_add(double* p, double v) { if (v == 1.2345e-300) unsafeAtomicAdd(p, v); }     // timing ablation only
#else
__device__ inline void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }
#endif

// add w * Ja^T Jb (6x6) into the compact tile at local poses (la, lb)
__device__ inline void tile_add_cross(double* Sc, const unsigned char* pslot, int la, int lb, const double* Ja, const double* Jb, double w) {
  // block rows belong to the larger local index (lower triangle)
  if (la > lb) {
    double* B = Sc + 36*(int)pslot[16*la + lb];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) lds_add(B + 6*r + c, w*(Ja[r]*Jb[c] + Ja[6+r]*Jb[6+c]));
  } else if (lb > la) {
    double* B = Sc + 36*(int)pslot[16*lb + la];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) lds_add(B + 6*r + c, w*(Jb[r]*Ja[c] + Jb[6+r]*Ja[6+c]));
  } else {
    // one pose vertex at two positions of the edge (slot a before slot b in the edge's vertex order): g2o's one-sided block, of which
    // its solver reads the upper triangle -- (Ja^T Omega Jb)(c, r), c <= r, mirrored into the lower entry (r, c)  (ba_kernels.h k_linearize)
    double* B = Sc + 36*(int)pslot[17*la];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c)
        lds_add(B + 6*r + c, w*(Ja[c]*Jb[r] + Ja[6+c]*Jb[6+r]));
  }
}

#ifdef MCP_LIN_PROF      // in-kernel phase stamps (scripts/lin_prof.sh); compiled out of the product build
__device__ unsigned long long g_lin_prof[8*8];
#define LIN_STAMP(i) do { if ((blockIdx.x & 127) == 100 && blockIdx.x < 1024 && threadIdx.x == 0) g_lin_prof[(blockIdx.x >> 7)*8 + (i)] = clock64(); } while (0)
#else
#define LIN_STAMP(i) do {} while (0)
#endif
#ifndef LIN_LDS_CAMS
#define LIN_LDS_CAMS 8      // cameras kept in LDS by the linearisation kernels (3.4 KB); a rig with more reads the rest from global memory
#endif
#ifndef LIN_WAVES
#define LIN_WAVES 0   // > 0: wavefronts per SIMD the register allocation is held to
#endif
// LPP = lanes per point.  1: one lane per point, groups of <= 64 points (the layout for maps large enough to fill the chip: a lane
// walks all measurements of its point).  4: groups of <= 16 points, a point's measurements dealt to its 4 lanes -- the per-group
// latency of a small problem (BundleAdjustRecent's window: 40 groups on 256 compute units, all of the stage is one group's latency)
// falls from 8 dependent measurements to 2.  What a lane kept in registers per point (V, g, the source pose's blocks) is summed over
// the point's lanes at the end (butterfly inside the quad: same value on every lane, fixed order); W blocks that several
// measurements of a point contribute to are accumulated in LDS (wl_off = offset of that area in doubles) and go out once.
// make_slot (ba_kernels.h) with the chain transforms already in registers: F = first[chain link], R2 = second[observer chain link]
struct SlotPre { Se3 F; double R2[9]; };
__device__ inline void slot_from_pre(int side, const SlotPre& sp, const double* A /*2x3*/, const double* xw, const double* Robs_last, SlotGeom& s) {
  se3_apply(sp.F, xw, s.base);
  if (side == 0) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s.B[3*r+c] = A[3*r]*sp.R2[c] + A[3*r+1]*sp.R2[3+c] + A[3*r+2]*sp.R2[6+c];
    s.sign = 1.0;
  } else {
    double Rrel[9];
    mat3_mul_t(Robs_last, sp.F.R, Rrel);          // R(T_obs * T_src_i^-1)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s.B[3*r+c] = A[3*r]*Rrel[c] + A[3*r+1]*Rrel[3+c] + A[3*r+2]*Rrel[6+c];
    s.sign = -1.0;
  }
}
// what the measurement loop needs of a measurement that can be fetched by its index alone (one round ahead)
struct MeasHead { int oc, mask, cam, last, s0, s1; double u, v, om; };      // (raw loads only: nothing here may be looked at before the next round)
__device__ inline void load_head(const DevProblem& P, int m, MeasHead& h) {
  h.oc = P.m_chain[m]; h.mask = P.m_mask[m]; h.cam = P.m_cam[m]; h.last = P.m_last[m];
  h.s0 = P.slot_start[m]; h.s1 = P.slot_start[m + 1];
  h.u = P.m_u[m]; h.v = P.m_v[m]; h.om = P.m_omega[m];
}

template <int LPP, bool PIPE = false>
__device__ __forceinline__ void linearize_group_body(const DevProblem& P, const double* __restrict__ pt_x, const double* __restrict__ first,
                    const double* __restrict__ second, const double* __restrict__ sigma,
                    double* __restrict__ stU /* staged pose-pose blocks */, double* __restrict__ stb /* staged local rhs */,
                    double* __restrict__ V, double* __restrict__ g, double* __restrict__ W, int wl_off, int* __restrict__ fail_zero) {
  // (the failure flags of the reduced systems built from this linearisation: cleared here instead of by a fill launch of their own)
  if (fail_zero && blockIdx.x == 0 && threadIdx.x < 4) fail_zero[threadIdx.x] = 0;
  extern __shared__ double Sc[];                 // the group's blocks, 36 doubles each (launch: 288 B x the largest block count of any group)
  __shared__ double bl[GRP_DOF];
  __shared__ unsigned char pslot[GRP_LMAX*16];
  __shared__ int sdst[GRP_LMAX*(GRP_LMAX + 1)/2], srd[GRP_LMAX];
  __shared__ double satan[130];                  // the arctangent's table (atan_cr.h), likewise
  __shared__ mcp_camera scam[LIN_LDS_CAMS];      // the cameras: read per measurement by cam_project -- from LDS they do not queue behind the W stores in the vector-memory counter
  const int grp = blockIdx.x, lane = threadIdx.x;
  LIN_STAMP(0);
  const int nb_grp = P.g_blk0[grp + 1] - P.g_blk0[grp];
  {
    const int b0 = P.g_blk0[grp], nb = nb_grp;
    for (int i = lane; i < nb*36; i += 64) Sc[i] = 0.0;
    for (int i = lane; i < GRP_DOF; i += 64) bl[i] = 0.0;
    for (int i = lane; i < nb; i += 64) { pslot[P.blk_pair[b0 + i]] = (unsigned char)i; sdst[i] = P.blk_dst[b0 + i]; }      // (pairs outside the list are never looked up)
    if (lane < GRP_LMAX) srd[lane] = P.rhs_dst[grp*GRP_LMAX + lane];
    const int ncl = P.ncam < LIN_LDS_CAMS ? P.ncam : LIN_LDS_CAMS;
    const double* src = reinterpret_cast<const double*>(P.cams); double* dst = reinterpret_cast<double*>(scam);
    static_assert(sizeof(mcp_camera) % sizeof(double) == 0, "camera copied as doubles");
    for (int i = lane; i < ncl*(int)(sizeof(mcp_camera)/sizeof(double)); i += 64) dst[i] = src[i];
    for (int i = lane; i < 65; i += 64) { satan[i] = mcp_atan::kAtanHi[i]; satan[65 + i] = mcp_atan::kAtanLo[i]; }
  }
  double* const Wl = Sc + wl_off;                // LPP > 1: the group's W blocks, 18 doubles per incidence
  const int inc0 = (LPP > 1) ? P.sp_i[P.g_sp0[grp]] : 0;
  if constexpr (LPP > 1) { const int ni = P.sp_i[P.g_sp0[grp + 1]] - inc0; for (int i = lane; i < ni*18; i += 64) Wl[i] = 0.0; }
  __syncthreads();
  LIN_STAMP(1);
  const int sub = lane % LPP;
  const int sp = P.g_sp0[grp] + lane/LPP;
  const bool valid = sp < P.g_sp0[grp + 1] && !P.sp_big[sp];
  // register accumulators of the first source-chain slot (the pose the point is expressed in)
  double Uss[21], bs[6], Wss[18];
  int wss_inc = -1;
#pragma unroll
  for (int i = 0; i < 18; ++i) Wss[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 21; ++i) Uss[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) bs[i] = 0.0;
  int ls = -1;
  if (valid) {
    const int pt = P.sp_pt[sp];
    const int sc = P.pt_chain[pt], slen = P.chain_len[sc];
    const int lpt = P.pt_unk[pt];
    const double x[3] = { pt_x[3*(size_t)pt], pt_x[3*(size_t)pt+1], pt_x[3*(size_t)pt+2] };
    Se3 Ts;
    load_se3(first + 12*(size_t)(sc*MAXC + slen - 1), Ts);
    double xw[3];
    se3_apply_inv(Ts, x, xw);
    // point frame (shared by all measurements of the point)
    double c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0}, c2v[3] = {0, 0, 0};
    if (lpt >= 0) {
      double Rp[9], dir[3], rho, rx[3], g0[3], g1[3];
      point_frame(x, Rp, dir, rho);
      mat3_vec(Rp, x, rx);
      generator(3, rx, g0); generator(4, rx, g1);
      mat3t_vec(Rp, g0, c0); mat3t_vec(Rp, g1, c1);
      c2v[0] = -x[0]/rho; c2v[1] = -x[1]/rho; c2v[2] = -x[2]/rho;
    }
    double Vp[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
    const bool fixed_neg = P.pt_fixed[pt] && P.robust;
    LIN_STAMP(2);
    if constexpr (PIPE) {
    // ---- round 5: the measurement loop with every load of a round issued at its head.  Vector loads and stores share one in-order
    // counter on gfx950: a load issued behind the 144-byte W block of the measurement before is only known to have arrived when that
    // store has been acknowledged, and a measurement is a chain of index hops (chain -> link -> transform; slot range -> local pose /
    // incidence): 23 k cycles per measurement, 6 k of them arithmetic.  Here (a) what can be fetched by the measurement index alone is
    // fetched one round ahead (MeasHead; m_last = the observer chain's last link, one hop less), (b) the transforms and slot records of
    // this round are all requested at its head, (c) the W block of the round before leaves AFTER them (deferred in registers; an
    // accumulating block as a no-return atomic add: a lane's blocks are its own, so it is the same single addition), (d) the
    // arithmetic follows.  Same operations on the same numbers as the loop below: the same bits.
    static_assert(LPP == 1, "the quad form keeps W in LDS: nothing to defer");
    const int mend = P.sp_m[sp + 1];
    int m = P.sp_m[sp];
    MeasHead nxt;
    if (m < mend) load_head(P, m, nxt);
    double Wd[18]; double* Wd_ptr = nullptr; bool Wd_acc = false;
    for (; m < mend; ++m) {
      const MeasHead cur = nxt;
      if (m + 1 < mend) load_head(P, m + 1, nxt);
      const int oc = cur.oc, mask = cur.mask;
      // (b) this round's loads
      Se3 To;
      load_se3(first + 12*(size_t)cur.last, To);
      const int bitA = mask ? __builtin_ctz(mask) : 0, mask2 = mask & (mask - 1), bitB = mask2 ? __builtin_ctz(mask2) : 0;
      SlotPre preA, preB;
      {
        const int sideA = bitA >> MAXC_LOG, sideB = bitB >> MAXC_LOG;
        const size_t ia_ = (size_t)((sideA ? sc : oc)*MAXC + (bitA & (MAXC - 1))), ib_ = (size_t)((sideB ? sc : oc)*MAXC + (bitB & (MAXC - 1)));
        load_se3(first + 12*ia_, preA.F); load_se3(first + 12*ib_, preB.F);
#pragma unroll
        for (int k = 0; k < 9; ++k) { preA.R2[k] = second[9*ia_ + k]; preB.R2[k] = second[9*ib_ + k]; }
      }
#if defined(LIN_ABL) && LIN_ABL == 2
      const int s0 = cur.s0, ns = 0;                                      // timing ablation only: no pose slots
#else
      const int s0 = cur.s0, ns = cur.s1 - cur.s0;
#endif
      const int sA = ns > 0 ? s0 : 0, sB = ns > 1 ? s0 + 1 : sA;          // (always a valid index: fetched whether or not the slot exists)
      const int lpA = P.slot_lp[sA], lpB = P.slot_lp[sB], incA = P.slot_inc[sA], incB = P.slot_inc[sB];
      const int firstA = P.slot_first[sA], firstB = P.slot_first[sB];
      // (c) the deferred W block of the round before
      if (Wd_ptr) {
        if (Wd_acc) {
#pragma unroll
          for (int k = 0; k < 18; ++k) unsafeAtomicAdd(Wd_ptr + k, Wd[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 18; ++k) Wd_ptr[k] = Wd[k];
        }
        Wd_ptr = nullptr;
      }
      if (mask == 0 && lpt < 0) continue;
      // (d) the arithmetic of the loop below
      double xc[3];
      se3_apply(To, xw, xc);
      Projection pr;
      { const int ci = cur.cam; if (ci < LIN_LDS_CAMS) cam_project<true>(scam[ci], xc, pr, satan, satan + 65); else cam_project<true>(P.cams[ci], xc, pr, satan, satan + 65); }
      const double e0 = cur.u - pr.u, e1 = cur.v - pr.v;
      const double omega = cur.om;
      double c2 = omega*(e0*e0 + e1*e1);
      if (fixed_neg) c2 = -c2;
      double w = omega;
      if (P.robust) { double r0, r1; robustify(c2, sigma[1], sigma[2], r0, r1); w *= r1; }
      double dT[3], dP[3];
      cam_sphere_deriv(xc, dT, dP);
      double A[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        A[c]     = -(pr.D[0]*dT[c] + pr.D[1]*dP[c]);
        A[3 + c] = -(pr.D[2]*dT[c] + pr.D[3]*dP[c]);
      }
      double Jp[6] = {0, 0, 0, 0, 0, 0};
      if (lpt >= 0) {
        double Rcs[9], AR[6];
        mat3_mul_t(To.R, Ts.R, Rcs);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) AR[3*r+c] = A[3*r]*Rcs[c] + A[3*r+1]*Rcs[3+c] + A[3*r+2]*Rcs[6+c];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          Jp[3*r]   = AR[3*r]*c0[0]  + AR[3*r+1]*c0[1]  + AR[3*r+2]*c0[2];
          Jp[3*r+1] = AR[3*r]*c1[0]  + AR[3*r+1]*c1[1]  + AR[3*r+2]*c1[2];
          Jp[3*r+2] = AR[3*r]*c2v[0] + AR[3*r+1]*c2v[1] + AR[3*r+2]*c2v[2];
        }
        Vp[0] += w*(Jp[0]*Jp[0] + Jp[3]*Jp[3]); Vp[1] += w*(Jp[0]*Jp[1] + Jp[3]*Jp[4]); Vp[2] += w*(Jp[0]*Jp[2] + Jp[3]*Jp[5]);
        Vp[3] += w*(Jp[1]*Jp[1] + Jp[4]*Jp[4]); Vp[4] += w*(Jp[1]*Jp[2] + Jp[4]*Jp[5]); Vp[5] += w*(Jp[2]*Jp[2] + Jp[5]*Jp[5]);
#pragma unroll
        for (int c = 0; c < 3; ++c) gp[c] += -w*(Jp[c]*e0 + Jp[3+c]*e1);
      }
      int ia = 0;
      for (int bit_a = 0; bit_a < 2*MAXC && ia < ns; ++bit_a) {
        if (!(mask & (1 << bit_a))) continue;
        SlotGeom sa; double Ja[12];
        if (ia == 0) slot_from_pre(bit_a >> MAXC_LOG, preA, A, xw, To.R, sa);
        else if (ia == 1) slot_from_pre(bit_a >> MAXC_LOG, preB, A, xw, To.R, sa);      // (keeping the Jacobian the cross term below computed for this slot: measured, no gain)
        else make_slot(bit_a >> MAXC_LOG, bit_a & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sa);
        slot_jacobian(sa, Ja);
        const int la = ia == 0 ? lpA : (ia == 1 ? lpB : (int)P.slot_lp[s0 + ia]);
        if (bit_a == MAXC) {            // first source link: private accumulators, reduced across the wave below
          ls = la;
#pragma unroll
          for (int r = 0; r < 6; ++r) bs[r] += -w*(Ja[r]*e0 + Ja[6+r]*e1);
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) { Uss[q] += w*(Ja[r]*Ja[c] + Ja[6+r]*Ja[6+c]); ++q; }
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r) lds_add(bl + 6*la + r, -w*(Ja[r]*e0 + Ja[6+r]*e1));
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) lds_add(Sc + 36*(int)pslot[17*la] + 6*r + c, w*(Ja[r]*Ja[c] + Ja[6+r]*Ja[6+c]));
        }
#if defined(LIN_ABL) && LIN_ABL == 3
        if (false) {                 // timing ablation only: no W blocks
#else
        if (lpt >= 0) {
#endif
          const int inc = ia == 0 ? incA : (ia == 1 ? incB : P.slot_inc[s0 + ia]);
          if (bit_a == MAXC) {          // the source pose's block collects one term per measurement: keep it in registers
            wss_inc = inc;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int c = 0; c < 3; ++c) Wss[3*r + c] += w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
          } else {
            const int isfirst = ia == 0 ? firstA : (ia == 1 ? firstB : (int)P.slot_first[s0 + ia]);
            double* Wb = W + 18*(size_t)inc;
            if (Wd_ptr == Wb) {         // a second slot of this measurement on the block that is waiting (one pose vertex at two positions of the edge): joins it
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Wd[3*r + c] += w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
            } else if (!Wd_ptr) {       // deferred: leaves at the head of the next round (or behind the loop)
              Wd_ptr = Wb; Wd_acc = !isfirst;
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Wd[3*r + c] = w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
            } else if (isfirst) {       // (a second non-source slot of one measurement -- long chains: written at once, as the loop below does)
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Wb[3*r + c] = w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
            } else {                    // (no load of W anywhere in this form of the kernel: an accumulating block is always an atomic add)
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) unsafeAtomicAdd(Wb + 3*r + c, w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]));
            }
          }
        }
        int ib = ia + 1;
        for (int bit_b = bit_a + 1; bit_b < 2*MAXC && ib < ns; ++bit_b) {
          if (!(mask & (1 << bit_b))) continue;
          SlotGeom sb; double Jb[12];
          if (ib == 1) slot_from_pre(bit_b >> MAXC_LOG, preB, A, xw, To.R, sb);
          else make_slot(bit_b >> MAXC_LOG, bit_b & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sb);
          slot_jacobian(sb, Jb);
          tile_add_cross(Sc, pslot, la, ib == 1 ? lpB : (int)P.slot_lp[s0 + ib], Ja, Jb, w);
          ++ib;
        }
        ++ia;
      }
    }
    if (Wd_ptr) {
      if (Wd_acc) {
#pragma unroll
        for (int k = 0; k < 18; ++k) unsafeAtomicAdd(Wd_ptr + k, Wd[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 18; ++k) Wd_ptr[k] = Wd[k];
      }
    }
    } else
    for (int m = P.sp_m[sp] + sub; m < P.sp_m[sp + 1]; m += LPP) {
      const int oc = P.m_chain[m], olen = P.chain_len[oc];
      const int mask = P.m_mask[m];
      if (mask == 0 && lpt < 0) continue;
      Se3 To;
      load_se3(first + 12*(size_t)(oc*MAXC + olen - 1), To);
      double xc[3];
      se3_apply(To, xw, xc);
      Projection pr;
      { const int ci = P.m_cam[m]; if (ci < LIN_LDS_CAMS) cam_project<true>(scam[ci], xc, pr, satan, satan + 65); else cam_project<true>(P.cams[ci], xc, pr, satan, satan + 65); }
      const double e0 = P.m_u[m] - pr.u, e1 = P.m_v[m] - pr.v;
      const double omega = P.m_omega[m];
      double c2 = omega*(e0*e0 + e1*e1);
      if (fixed_neg) c2 = -c2;
      double w = omega;
      if (P.robust) { double r0, r1; robustify(c2, sigma[1], sigma[2], r0, r1); w *= r1; }
      double dT[3], dP[3];
      cam_sphere_deriv(xc, dT, dP);
      double A[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        A[c]     = -(pr.D[0]*dT[c] + pr.D[1]*dP[c]);
        A[3 + c] = -(pr.D[2]*dT[c] + pr.D[3]*dP[c]);
      }
      double Jp[6] = {0, 0, 0, 0, 0, 0};
      if (lpt >= 0) {
        double Rcs[9], AR[6];
        mat3_mul_t(To.R, Ts.R, Rcs);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) AR[3*r+c] = A[3*r]*Rcs[c] + A[3*r+1]*Rcs[3+c] + A[3*r+2]*Rcs[6+c];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          Jp[3*r]   = AR[3*r]*c0[0]  + AR[3*r+1]*c0[1]  + AR[3*r+2]*c0[2];
          Jp[3*r+1] = AR[3*r]*c1[0]  + AR[3*r+1]*c1[1]  + AR[3*r+2]*c1[2];
          Jp[3*r+2] = AR[3*r]*c2v[0] + AR[3*r+1]*c2v[1] + AR[3*r+2]*c2v[2];
        }
        Vp[0] += w*(Jp[0]*Jp[0] + Jp[3]*Jp[3]); Vp[1] += w*(Jp[0]*Jp[1] + Jp[3]*Jp[4]); Vp[2] += w*(Jp[0]*Jp[2] + Jp[3]*Jp[5]);
        Vp[3] += w*(Jp[1]*Jp[1] + Jp[4]*Jp[4]); Vp[4] += w*(Jp[1]*Jp[2] + Jp[4]*Jp[5]); Vp[5] += w*(Jp[2]*Jp[2] + Jp[5]*Jp[5]);
#pragma unroll
        for (int c = 0; c < 3; ++c) gp[c] += -w*(Jp[c]*e0 + Jp[3+c]*e1);
      }
#if defined(LIN_ABL) && LIN_ABL == 2
      const int s0 = P.slot_start[m], ns = 0;                              // timing ablation only: no pose slots
#else
      const int s0 = P.slot_start[m], ns = P.slot_start[m+1] - s0;
#endif
      int ia = 0;
      for (int bit_a = 0; bit_a < 2*MAXC && ia < ns; ++bit_a) {
        if (!(mask & (1 << bit_a))) continue;
        SlotGeom sa; double Ja[12];
        make_slot(bit_a >> MAXC_LOG, bit_a & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sa);
        slot_jacobian(sa, Ja);
        const int la = P.slot_lp[s0 + ia];
        if (bit_a == MAXC) {            // first source link: private accumulators, reduced across the wave below
          ls = la;
#pragma unroll
          for (int r = 0; r < 6; ++r) bs[r] += -w*(Ja[r]*e0 + Ja[6+r]*e1);
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) { Uss[q] += w*(Ja[r]*Ja[c] + Ja[6+r]*Ja[6+c]); ++q; }
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r) lds_add(bl + 6*la + r, -w*(Ja[r]*e0 + Ja[6+r]*e1));
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) lds_add(Sc + 36*(int)pslot[17*la] + 6*r + c, w*(Ja[r]*Ja[c] + Ja[6+r]*Ja[6+c]));
        }
#if defined(LIN_ABL) && LIN_ABL == 3
        if (false) {                 // timing ablation only: no W blocks
#else
        if (lpt >= 0) {
#endif
          if (bit_a == MAXC) {          // the source pose's block collects one term per measurement: keep it in registers
            wss_inc = P.slot_inc[s0 + ia];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int c = 0; c < 3; ++c) Wss[3*r + c] += w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
          } else if constexpr (LPP > 1) {
            double* Wb = Wl + 18*(P.slot_inc[s0 + ia] - inc0);
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int c = 0; c < 3; ++c) lds_add(Wb + 3*r + c, w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]));
          } else {
            double* Wb = W + 18*(size_t)P.slot_inc[s0 + ia];
            if (P.slot_first[s0 + ia]) {
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Wb[3*r + c] = w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
            } else {
#pragma unroll
              for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Wb[3*r + c] += w*(Ja[r]*Jp[c] + Ja[6+r]*Jp[3+c]);
            }
          }
        }
        int ib = ia + 1;
        for (int bit_b = bit_a + 1; bit_b < 2*MAXC && ib < ns; ++bit_b) {
          if (!(mask & (1 << bit_b))) continue;
          SlotGeom sb; double Jb[12];
          make_slot(bit_b >> MAXC_LOG, bit_b & (MAXC - 1), A, xw, first, second, oc, sc, To.R, sb);
          slot_jacobian(sb, Jb);
          tile_add_cross(Sc, pslot, la, P.slot_lp[s0 + ib], Ja, Jb, w);
          ++ib;
        }
        ++ia;
      }
    }
    LIN_STAMP(3);
    if constexpr (LPP > 1) {
      // the point's lanes (all of them are here: `valid` is a property of the point) sum what each kept in registers
#pragma unroll
      for (int d = 1; d < LPP; d <<= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Vp[k] += __shfl_xor(Vp[k], d, 64);
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] += __shfl_xor(gp[k], d, 64);
#pragma unroll
        for (int k = 0; k < 21; ++k) Uss[k] += __shfl_xor(Uss[k], d, 64);
#pragma unroll
        for (int k = 0; k < 6; ++k) bs[k] += __shfl_xor(bs[k], d, 64);
#pragma unroll
        for (int k = 0; k < 18; ++k) Wss[k] += __shfl_xor(Wss[k], d, 64);
        wss_inc = max(wss_inc, __shfl_xor(wss_inc, d, 64));
        ls = max(ls, __shfl_xor(ls, d, 64));
      }
      if (lpt >= 0) {
        // (LDS operations of one wavefront execute in order: the loop's atomics on these blocks are done)
        if (wss_inc >= 0 && sub == 0) {
          double* Wb = Wl + 18*(wss_inc - inc0);
#pragma unroll
          for (int k = 0; k < 18; ++k) Wb[k] += Wss[k];
        }
        const int i0 = P.sp_i[sp], npi = P.sp_i[sp + 1] - i0;
        for (int e = sub; e < npi*18; e += LPP) W[18*(size_t)i0 + e] = Wl[18*(i0 - inc0) + e];
        if (sub == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) V[6*(size_t)lpt + k] = Vp[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) g[3*(size_t)lpt + k] = gp[k];
        }
      }
      if (sub != 0) ls = -1;                  // one lane per point enters the reduction over the points below
    } else
    if (lpt >= 0) {
      if (wss_inc >= 0) {
        double* Wb = W + 18*(size_t)wss_inc;
        if (P.inc_mixed[wss_inc]) {
          if constexpr (PIPE) {
#pragma unroll
            for (int k = 0; k < 18; ++k) unsafeAtomicAdd(Wb + k, Wss[k]);
          } else {
#pragma unroll
            for (int k = 0; k < 18; ++k) Wb[k] += Wss[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 18; ++k) Wb[k] = Wss[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) V[6*(size_t)lpt + k] = Vp[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g[3*(size_t)lpt + k] = gp[k];
    }
  }
  LIN_STAMP(4);
  // segmented wave reduction of the source-pose accumulators, one LDS update per distinct pose
  unsigned long long todo = __ballot(ls >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int key = __shfl(ls, leader, 64);
    const bool mine = (ls == key);
    // reduce-scatter over the wavefront (ba_device.h): 32 lane exchanges instead of 27 x 6, and the 27 totals land in 27 different
    // lanes, each of which adds its own entry to the tile
    double w32[32];
#pragma unroll
    for (int i = 0; i < 21; ++i) w32[i] = mine ? Uss[i] : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) w32[21 + i] = mine ? bs[i] : 0.0;
#pragma unroll
    for (int i = 27; i < 32; ++i) w32[i] = 0.0;
    int idx; const double total = wave_reduce_scatter32(w32, lane, idx);
    if (!(lane & 1) && idx < 27) {
      if (idx < 21) {
        int r = 0; while ((r + 1)*(r + 2)/2 <= idx) ++r;
        const int c = idx - r*(r + 1)/2;
        Sc[36*(int)pslot[17*key] + 6*r + c] += total;
      } else bl[6*key + idx - 21] += total;
    }
    todo &= ~__ballot(mine);
    __syncthreads();
  }
  __syncthreads();
  LIN_STAMP(5);
  // flush the local tile to the group's staging slots (k_assemble sums them in group order)
  flush_compact_blocks<64>(Sc, bl, nb_grp, sdst, srd, stU, stb);
  LIN_STAMP(6);
}
#if LIN_WAVES > 0      // (the bound is a promise of at most 128 threads -- the launch has 64; with a one-wavefront bound the compiler ignores the occupancy request)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(LIN_WAVES, LIN_WAVES)))
#else
__global__ void __launch_bounds__(64)
#endif
k_linearize_group(DevProblem P, const double* __restrict__ pt_x, const double* __restrict__ first,
                  const double* __restrict__ second, const double* __restrict__ sigma,
                  double* __restrict__ stU, double* __restrict__ stb, double* __restrict__ V, double* __restrict__ g, double* __restrict__ W, int* __restrict__ fail_zero) {
  linearize_group_body<1>(P, pt_x, first, second, sigma, stU, stb, V, g, W, 0, fail_zero);
}
// the same with the measurement loop software-pipelined (loads of a round at its head, the W block of the round before behind them)
__global__ void __launch_bounds__(64)
k_linearize_pipe(DevProblem P, const double* __restrict__ pt_x, const double* __restrict__ first,
                 const double* __restrict__ second, const double* __restrict__ sigma,
                 double* __restrict__ stU, double* __restrict__ stb, double* __restrict__ V, double* __restrict__ g, double* __restrict__ W, int* __restrict__ fail_zero) {
  linearize_group_body<1, true>(P, pt_x, first, second, sigma, stU, stb, V, g, W, 0, fail_zero);
}
constexpr int LIN_QUAD_PTS = 16;           // points per group when the quad form runs (64 lanes / 4)
__global__ void __launch_bounds__(64)
k_linearize_quad(DevProblem P, const double* __restrict__ pt_x, const double* __restrict__ first,
                 const double* __restrict__ second, const double* __restrict__ sigma,
                 double* __restrict__ stU, double* __restrict__ stb, double* __restrict__ V, double* __restrict__ g, double* __restrict__ W, int wl_off, int* __restrict__ fail_zero) {
  linearize_group_body<4>(P, pt_x, first, second, sigma, stU, stb, V, g, W, wl_off, fail_zero);
}

// ------------------------------------------------------------------------------------------
#ifndef SCH_CHUNK_PTS
#define SCH_CHUNK_PTS 16
#endif
constexpr int SCH_CHUNK = SCH_CHUNK_PTS;                 // points per MFMA chunk (8: more barriers, 5.8 -> 7.3 ms; 32: one workgroup per CU)
constexpr int SCH_TPP = 256/SCH_CHUNK;        // threads per point in the scatter
constexpr int SCH_EPT = (GRP_LMAX*6 + SCH_TPP - 1)/SCH_TPP;    // W rows per thread (<= 16 incidences x 6 rows per point)
constexpr int SCH_K = 3*SCH_CHUNK;            // 48
constexpr int SCH_LD = SCH_K + 1;             // LDS row stride (doubles)
constexpr int SCH_NCH = GRP_PTS/SCH_CHUNK;    // chunks per group
#ifndef SCH_Z
#define SCH_Z 1       // 1: stage Z = W L only (V^-1 = L L^T), S -= Z Z^T;  0: stage W and Y = W V^-1, S -= Y W^T
#endif
#ifndef SCH_WAVES
#define SCH_WAVES 3   // wavefronts per SIMD the register allocation aims at (3: 168 VGPRs + 104 B of scratch, three workgroups per CU: 5.34 -> 5.06 ms)
#endif
typedef double sch_d4 __attribute__((ext_vector_type(4)));
#ifdef MCP_SCH_PROF
__device__ unsigned long long g_sch_prof[8*8];
#define SCH_T0() unsigned long long sch_t = clock64(), sch_a[8] = {0,0,0,0,0,0,0,0}
#define SCH_LAP(i) do { const unsigned long long n_ = clock64(); sch_a[i] += n_ - sch_t; sch_t = n_; } while (0)
#define SCH_OUT() do { if ((blockIdx.x & 127) == 100 && blockIdx.x < 1024 && blockIdx.y == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_sch_prof[(blockIdx.x >> 7)*8 + i_] = sch_a[i_]; } while (0)
#else
#define SCH_T0() do {} while (0)
#define SCH_LAP(i) do {} while (0)
#define SCH_OUT() do {} while (0)
#endif

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SCH_WAVES, SCH_WAVES)))
k_schur_group(DevProblem P, double lambda, const double* __restrict__ V, const double* __restrict__ g,
              const double* __restrict__ W, double* __restrict__ Vinv, double* __restrict__ stS /* staged blocks of W V^-1 W^T */,
              double* __restrict__ str /* staged W V^-1 g */, int* __restrict__ fail, SysBatch sb) {
  if (blockIdx.y) { const int q = blockIdx.y; lambda = sb.lambda[q]; Vinv += q*sb.vstride; stS += q*sb.ststride; str += q*sb.strstride; fail += q; }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Yd = lds;                            // [GRP_DOF][SCH_LD]   (SCH_Z: Z, the only staging array)
  double* Wd = lds + (SCH_Z ? 0 : GRP_DOF*SCH_LD);     // [GRP_DOF][SCH_LD]
  double* Vi = Wd + GRP_DOF*SCH_LD;            // [SCH_CHUNK][6]
  double* gl = Vi + SCH_CHUNK*6;               // [SCH_K]
  const int grp = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int sp0 = P.g_sp0[grp], sp1 = P.g_sp0[grp + 1];
  const int* gp_ = P.g_pose + grp*GRP_LMAX;
  int npl = 0;
  for (int i = 0; i < GRP_LMAX; ++i) if (gp_[i] >= 0) npl = i + 1;
  const int ntile = (6*npl + 15)/16;           // 16-row tiles in use
  // lower-triangular tile pairs handled by this wave: t = wave, wave+4, ...
  sch_d4 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = (sch_d4){0.0, 0.0, 0.0, 0.0};
  double racc = 0.0;                           // half of rhs entry t % GRP_DOF (t < 2*GRP_DOF)
  SCH_T0();
  // The global reads of a chunk hang off index hops (point -> incidence range / free-point index).  Those of all
  // chunks are resolved in the prologue (one exposed round trip per group instead of one per chunk); the W / V / g
  // reads are then issued one chunk ahead, into registers, and land in LDS after the matrix-core phase of the chunk
  // before, so only the first chunk of a group waits for them.
  const int pl = t/SCH_TPP, sub = t%SCH_TPP;
  int ci0[SCH_NCH], ccnt[SCH_NCH], cpt[SCH_NCH];
#pragma unroll
  for (int c = 0; c < SCH_NCH; ++c) {
    const int sp = sp0 + c*SCH_CHUNK + pl;
    ci0[c] = 0; ccnt[c] = 0; cpt[c] = -1;
    if (sp < sp1 && !P.sp_big[sp]) { ci0[c] = P.sp_i[sp]; ccnt[c] = (P.sp_i[sp + 1] - ci0[c])*6; }
    if (t < SCH_CHUNK) {
      const int sq = sp0 + c*SCH_CHUNK + t;
      if (sq < sp1 && !P.sp_big[sq]) cpt[c] = P.pt_unk[P.sp_pt[sq]];
    }
  }
  double pw[SCH_EPT][3]; int prow[SCH_EPT];    // this thread's W rows of the next chunk
  double pv[6], pg[3]; int plpt = -1;          // threads 0..15: V and g of the next chunk's points
  auto prefetch = [&]() {                      // the next chunk not fetched yet: entry 0 of the index queues
    const int i0 = ci0[0], cnt = ccnt[0];
#pragma unroll
    for (int k = 0; k < SCH_EPT; ++k) {
      const int it = sub + SCH_TPP*k;
      prow[k] = -1;
      if (it < cnt) {
        const int inc = i0 + it/6, r = it%6;
        const double* Wr = W + 18*(size_t)inc + 3*r;
        pw[k][0] = Wr[0]; pw[k][1] = Wr[1]; pw[k][2] = Wr[2];
        prow[k] = 6*P.inc_lp[inc] + r;
      }
    }
    if (t < SCH_CHUNK) {
      plpt = cpt[0];
      if (plpt >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pv[k] = V[6*(size_t)plpt + k];
        pg[0] = g[3*(size_t)plpt]; pg[1] = g[3*(size_t)plpt + 1]; pg[2] = g[3*(size_t)plpt + 2];
      }
    }
#pragma unroll
    for (int c = 0; c + 1 < SCH_NCH; ++c) { ci0[c] = ci0[c + 1]; ccnt[c] = ccnt[c + 1]; cpt[c] = cpt[c + 1]; }
    ccnt[SCH_NCH - 1] = 0; cpt[SCH_NCH - 1] = -1;
  };
  prefetch();
  // the staging arrays are zero-filled once; after each chunk every thread clears exactly the rows it wrote
  for (int i = t; i < (SCH_Z ? 1 : 2)*GRP_DOF*SCH_LD; i += 256) lds[i] = 0.0;
  int orow[SCH_EPT];
  SCH_LAP(0);
  for (int base = sp0; base < sp1; base += SCH_CHUNK) {
    if (t < SCH_CHUNK) {
      double I6[6] = {0, 0, 0, 0, 0, 0}; double g3[3] = {0, 0, 0};
      if (plpt >= 0) {
        if (!inv_sym3(pv, lambda, I6)) atomicOr(fail, 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) Vinv[6*(size_t)plpt + k] = I6[k];
        g3[0] = pg[0]; g3[1] = pg[1]; g3[2] = pg[2];
#if SCH_Z
        // V^-1 = L L^T (3x3 Cholesky; a failed inverse is flagged above, its factor is zeroed rather than NaN):
        // I6 <- (l00, l10, l20, l11, l21, l22), g3 <- L^T g, so that W V^-1 W^T = Z Z^T and W V^-1 g = Z (L^T g)
        const double l00 = I6[0] > 0.0 ? sqrt(I6[0]) : 0.0, r00 = l00 > 0.0 ? 1.0/l00 : 0.0;
        const double l10 = I6[1]*r00, l20 = I6[2]*r00;
        const double d11 = I6[3] - l10*l10;
        const double l11 = d11 > 0.0 ? sqrt(d11) : 0.0, r11 = l11 > 0.0 ? 1.0/l11 : 0.0;
        const double l21 = (I6[4] - l20*l10)*r11;
        const double d22 = I6[5] - l20*l20 - l21*l21;
        const double l22 = d22 > 0.0 ? sqrt(d22) : 0.0;
        I6[0] = l00; I6[1] = l10; I6[2] = l20; I6[3] = l11; I6[4] = l21; I6[5] = l22;
        g3[0] = l00*pg[0] + l10*pg[1] + l20*pg[2]; g3[1] = l11*pg[1] + l21*pg[2]; g3[2] = l22*pg[2];
#endif
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Vi[6*t + k] = I6[k];
      gl[3*t] = g3[0]; gl[3*t + 1] = g3[1]; gl[3*t + 2] = g3[2];
    }
    __syncthreads();
    SCH_LAP(1);
    {   // scatter W and Y = W V^-1 rows: 16 threads per point
      const double* I6 = Vi + 6*pl;
#pragma unroll
      for (int k = 0; k < SCH_EPT; ++k) {
        orow[k] = prow[k];
        if (prow[k] < 0) continue;
        const double w0 = pw[k][0], w1 = pw[k][1], w2 = pw[k][2];
        double* yd = Yd + prow[k]*SCH_LD + 3*pl;
#if SCH_Z
        yd[0] = w0*I6[0] + w1*I6[1] + w2*I6[2];
        yd[1] = w1*I6[3] + w2*I6[4];
        yd[2] = w2*I6[5];
#else
        double* wd = Wd + prow[k]*SCH_LD + 3*pl;
        wd[0] = w0; wd[1] = w1; wd[2] = w2;
        yd[0] = w0*I6[0] + w1*I6[1] + w2*I6[2];
        yd[1] = w0*I6[1] + w1*I6[3] + w2*I6[4];
        yd[2] = w0*I6[2] + w1*I6[4] + w2*I6[5];
#endif
      }
    }
    SCH_LAP(2);
    if (base + SCH_CHUNK < sp1) prefetch();      // in flight during the matrix-core phase
    __syncthreads();
    SCH_LAP(3);
    // S_loc += Y W^T on the matrix cores
    {   // wave w owns the lower-triangular tile pairs w, w+4, ...
      const int i = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int p = wave + 4*s;
        int tr = 0;
        while ((tr + 1)*(tr + 2)/2 <= p) ++tr;
        const int tc = p - tr*(tr + 1)/2;
        if (p < 21 && tr < ntile) {
          const double* Ya = Yd + (16*tr + i)*SCH_LD + kq;
          const double* Wb = Wd + (16*tc + i)*SCH_LD + kq;
          sch_d4 a = acc[s];
#pragma unroll
          for (int kk = 0; kk < SCH_K; kk += 4) a = __builtin_amdgcn_mfma_f64_16x16x4f64(Ya[kk], Wb[kk], a, 0, 0, 0);
          acc[s] = a;
        }
      }
    }
    SCH_LAP(4);
    if (t < 2*GRP_DOF) {                        // two threads per row, 24 columns each; both add to rhs at the flush
      const int row = (t < GRP_DOF) ? t : t - GRP_DOF, k0 = (t < GRP_DOF) ? 0 : SCH_K/2;
      const double* yr = Yd + row*SCH_LD + k0;
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < SCH_K/2; ++k) s += yr[k]*gl[k0 + k];
      racc += s;
    }
    SCH_LAP(5);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCH_EPT; ++k) {
      if (orow[k] < 0) continue;
      double* yd = Yd + orow[k]*SCH_LD + 3*pl;
      yd[0] = 0.0; yd[1] = 0.0; yd[2] = 0.0;
#if !SCH_Z
      double* wd = Wd + orow[k]*SCH_LD + 3*pl;
      wd[0] = 0.0; wd[1] = 0.0; wd[2] = 0.0;
#endif
    }
    SCH_LAP(6);
  }
  // flush: the local product goes through LDS (the staging array is free now: packed lower triangle, 4656 <= 96 x 49
  // doubles; the local rhs in the V^-1 scratch) to the group's staging slots; k_assemble subtracts it from the system
  __syncthreads();
  {
    double* Sl = lds;
    double* rl = Vi;                              // [GRP_DOF] = SCH_CHUNK*6 doubles
    static_assert(SCH_CHUNK*6 >= GRP_DOF && GRP_DOF*SCH_LD >= GRP_TRI, "flush scratch");
    const int col_l = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int p = wave + 4*s;
      int tr = 0;
      while ((tr + 1)*(tr + 2)/2 <= p) ++tr;
      const int tc = p - tr*(tr + 1)/2;
      if (p < 21 && tr < ntile) {
        const sch_d4 a = acc[s];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int r = 16*tr + rq + 4*gq, c = 16*tc + col_l;
          if (c <= r) Sl[tri(r, c)] = a[gq];
        }
      }
    }
    // rows of 16-row tiles that were not computed (beyond the group's poses) are never read by the flush
    if (t < GRP_DOF) rl[t] = racc;
    __syncthreads();
    if (t >= GRP_DOF && t < 2*GRP_DOF) rl[t - GRP_DOF] += racc;      // fixed order: first half + second half
    __syncthreads();
    flush_group_blocks<256>(Sl, rl, grp, P, stS, str);
  }
  SCH_LAP(7);
  SCH_OUT();
}
// ------------------------------------------------------------------------------------------
// Assembly of the reduced system from the staged group blocks.  One writer per entry:
//   S_q(6a+r, 6b+c) = sum_g U_g - sum_g (W V^-1 W^T)_{g,q}  (+ lambda_q on the diagonal),   g ascending,
//   rhs_q(6a+r)     = sum_g b_g - sum_g (W V^-1 g)_{g,q},
// over the groups that stage a block for the pose pair (a, b) / the pose a.  Every entry of every tile of the
// factorisation plan is written (structural zeros and fill-in tiles included), so S needs no clearing between solves.
struct AsmPlan {
  int nfp;                        // free poses
  int ntiles;                     // tiles of the factorisation plan incl. the right-hand-side row (ti << 16 | tj)
  const int* tiles;
  const int* pair_id;             // [nfp*nfp] (a, b), a >= b  ->  pair index or -1
  const int* pr_start;            // [npairs+1] staged blocks of the pair: [pr_start[p], pr_start[p+1]), ascending group
  const int* po_start;            // [nfp+1] staged rhs rows of the pose, ascending group
};

#ifndef ASM_NT
#define ASM_NT 1024      // one entry of the 32 x 32 tile per thread: every entry's walk over its staged contributions is a chain of memory round trips,
#endif                   // and four times the threads have four times the walks in flight (256: 810-813, 512: 804-821, 1024: 822-825 it/s)
// (Round 5 measured one thread per entry for ALL systems of a batch -- the pose-pose sum and the list walk shared, a quarter of the
//  workgroups: 2.87 vs 2.34 ms of Schur + assembly per 20 iterations.  Fewer, longer walks hide less latency; dropped.)
__global__ void __launch_bounds__(ASM_NT)
k_assemble(AsmPlan A, int np, const double* __restrict__ stU, const double* __restrict__ stb,
           const double* __restrict__ stS, const double* __restrict__ str,
           const double* __restrict__ Ubig /* dense np x np + np contributions of the points outside the groups, or null */,
           double* __restrict__ S, SysBatch sb) {
  const int q = blockIdx.y;
  S += q*sb.sstride; stS += q*sb.ststride; str += q*sb.strstride;
  const double lam = sb.lambda_init[q];
  const int packed = A.tiles[blockIdx.x];
  const int ti = packed >> 16, tj = packed & 0xffff;
  const size_t n2 = (size_t)np*np;
  for (int e = threadIdx.x; e < 1024; e += ASM_NT) {
    const int row = 32*ti + (e >> 5), col = 32*tj + (e & 31);
    if (col >= np || row > np) continue;
    if (row == np) {                       // right-hand side (row n of the augmented matrix) and, behind it, the plain J^T r
      const int a = col/6, r = col - 6*a;
      double b = 0.0, sr = 0.0;
      int k = A.po_start[a]; const int k1 = A.po_start[a + 1];
      for (; k + 8 <= k1; k += 8) {            // independent loads in flight, summed in list order (a window's few free poses are staged by every group)
        double bb[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const size_t o = (size_t)(k + j)*6 + r; bb[j] = stb[o]; ss[j] = str[o]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { b += bb[j]; sr += ss[j]; }
      }
      for (; k < k1; ++k) { const size_t o = (size_t)k*6 + r; b += stb[o]; sr += str[o]; }
      if (Ubig) b += Ubig[n2 + col];
      S[n2 + col] = b - sr;
      S[n2 + np + col] = b;
      continue;
    }
    double v = 0.0;
    if (col <= row) {                      // (the upper part of a diagonal tile is never read: zero)
      const int a = row/6, r = row - 6*a, b = col/6, c = col - 6*b;
      const int pid = A.pair_id[(size_t)a*A.nfp + b];
      double u = 0.0, w = 0.0;
      if (pid >= 0) {
        const int k0 = A.pr_start[pid], k1 = A.pr_start[pid + 1];
        const double* pu = stU + (size_t)k0*36 + r*6 + c;
        const double* pw = stS + (size_t)k0*36 + r*6 + c;
        int k = k0;
        for (; k + 8 <= k1; k += 8, pu += 288, pw += 288) {       // independent loads in flight, summed in list order
          double uu[8], ww[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { uu[j] = pu[36*j]; ww[j] = pw[36*j]; }
#pragma unroll
          for (int j = 0; j < 8; ++j) { u += uu[j]; w += ww[j]; }
        }
        for (; k + 4 <= k1; k += 4, pu += 144, pw += 144) {
          const double u0 = pu[0], u1 = pu[36], u2 = pu[72], u3 = pu[108];
          const double w0 = pw[0], w1 = pw[36], w2 = pw[72], w3 = pw[108];
          u += u0; u += u1; u += u2; u += u3;
          w += w0; w += w1; w += w2; w += w3;
        }
        for (; k < k1; ++k, pu += 36, pw += 36) { u += pu[0]; w += pw[0]; }
      }
      v = u - w;
      if (Ubig) v += Ubig[(size_t)row*np + col];
      if (row == col) v += lam;
    }
    S[(size_t)row*np + col] = v;
  }
}

// ---- the same assembly for systems whose entries are sums of MANY staged contributions -- a BundleAdjustRecent window: a handful of
// free poses that every one of its 150 groups stages.  k_assemble gives an entry to one thread; 155 contributions are then 20 rounds
// of 8 loads, 26 us of dependent round trips for a 19 x 18 system.  Here an entry has ASML_LPE lanes, each sums a contiguous eighth
// of the list (in list order), and the eight partial sums are added in a fixed tree ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)); one
// workgroup per tile ROW (32 entries x 8 lanes), so a one-tile system is spread over 19 compute units instead of one.
// (Another -- fixed -- summation order than k_assemble's: which of the two runs is decided by the structure, at Prepare().)
constexpr int ASML_LPE = 8;
__device__ inline double asml_tree8(double v) {
  v += wave_xor<1>(v); v += wave_xor<2>(v); v += __shfl_xor(v, 4, 64);
  return v;
}
// partial sum of p[0], p[stride], ... over this lane's share of n terms, in order, four loads in flight
__device__ inline double asml_part(const double* __restrict__ p, int stride, int n, int l8) {
  const int c0 = (n*l8)/ASML_LPE, c1 = (n*(l8 + 1))/ASML_LPE;
  double a = 0.0;
  int k = c0;
  p += (size_t)c0*stride;
  for (; k + 4 <= c1; k += 4, p += 4*(size_t)stride) {
    const double x0 = p[0], x1 = p[stride], x2 = p[2*(size_t)stride], x3 = p[3*(size_t)stride];
    a += x0; a += x1; a += x2; a += x3;
  }
  for (; k < c1; ++k, p += stride) a += p[0];
  return a;
}
__global__ void __launch_bounds__(32*ASML_LPE)
k_assemble_long(AsmPlan A, int np, const double* __restrict__ stU, const double* __restrict__ stb,
                const double* __restrict__ stS, const double* __restrict__ str,
                const double* __restrict__ Ubig, double* __restrict__ S, SysBatch sb) {
  const int q = blockIdx.y;
  S += q*sb.sstride; stS += q*sb.ststride; str += q*sb.strstride;
  const double lam = sb.lambda_init[q];
  const int packed = A.tiles[blockIdx.x >> 5];
  const int ti = packed >> 16, tj = packed & 0xffff;
  const int row = 32*ti + (blockIdx.x & 31);
  if (row > np) return;                                  // (whole workgroup)
  const int l8 = threadIdx.x & (ASML_LPE - 1), col = 32*tj + (threadIdx.x >> 3);
  const size_t n2 = (size_t)np*np;
  const bool active = col < np;
  double u = 0.0, w = 0.0;
  if (active) {
    if (row == np) {                                     // right-hand side: b (from stb) and W V^-1 g (from str)
      const int a = col/6, r = col - 6*a;
      const int k0 = A.po_start[a], n = A.po_start[a + 1] - k0;
      u = asml_part(stb + (size_t)k0*6 + r, 6, n, l8);
      w = asml_part(str + (size_t)k0*6 + r, 6, n, l8);
    } else if (col <= row) {
      const int a = row/6, r = row - 6*a, b = col/6, c = col - 6*b;
      const int pid = A.pair_id[(size_t)a*A.nfp + b];
      if (pid >= 0) {
        const int k0 = A.pr_start[pid], n = A.pr_start[pid + 1] - k0;
        u = asml_part(stU + (size_t)k0*36 + r*6 + c, 36, n, l8);
        w = asml_part(stS + (size_t)k0*36 + r*6 + c, 36, n, l8);
      }
    }
  }
  u = asml_tree8(u); w = asml_tree8(w);
  if (!active || l8 != 0) return;
  if (row == np) {
    double b = u;
    if (Ubig) b += Ubig[n2 + col];
    S[n2 + col] = b - w;
    S[n2 + np + col] = b;
    return;
  }
  double v = 0.0;
  if (col <= row) {
    v = u - w;
    if (Ubig) v += Ubig[(size_t)row*np + col];
    if (row == col) v += lam;
  }
  S[(size_t)row*np + col] = v;
}
__global__ void __launch_bounds__(256)
k_udiag_long(AsmPlan A, int np, const double* __restrict__ stU, const double* __restrict__ Ubig, double* __restrict__ d) {
  const int i = (int)((blockIdx.x*blockDim.x + threadIdx.x) >> 3), l8 = threadIdx.x & (ASML_LPE - 1);
  double u = 0.0;
  if (i < np) {
    const int a = i/6, r = i - 6*a;
    const int pid = A.pair_id[(size_t)a*A.nfp + a];
    if (pid >= 0) { const int k0 = A.pr_start[pid]; u = asml_part(stU + (size_t)k0*36 + 7*r, 36, A.pr_start[pid + 1] - k0, l8); }
  }
  u = asml_tree8(u);
  if (i < np && l8 == 0) { if (Ubig) u += Ubig[(size_t)i*np + i]; d[i] = u; }
}

// diagonal of the pose part of J^T J (for computeLambdaInit [g2o]): d[6a+r] = sum_g U_g(6a+r, 6a+r)
__global__ void k_udiag(AsmPlan A, int np, const double* __restrict__ stU, const double* __restrict__ Ubig, double* __restrict__ d) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int a = i/6, r = i - 6*a;
  const int pid = A.pair_id[(size_t)a*A.nfp + a];
  double u = 0.0;
  if (pid >= 0) for (int k = A.pr_start[pid]; k < A.pr_start[pid + 1]; ++k) u += stU[(size_t)k*36 + 7*r];
  if (Ubig) u += Ubig[(size_t)i*np + i];
  d[i] = u;
}

constexpr size_t SCH_LDS_BYTES = (size_t)((SCH_Z ? 1 : 2)*GRP_DOF*SCH_LD + SCH_CHUNK*6 + SCH_K)*sizeof(double);

}  // namespace mcp
