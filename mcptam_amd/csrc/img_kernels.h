// img_kernels.h -- HIP kernels of the KeyFrame / Tracker image path (gfx950).
//
//   k_pyr_fast               CVD::halfSample x3 + fast_corner_detect_10 + fast_corner_score_10 + vFastFrequency, all levels of
//                            all cameras of a frame in one launch, LDS-tiled        src/KeyFrame.cc:189-190, 259-275
//   k_dilate5 / k_glare_mask cv::dilate x5 + threshold + AND         :214-238
//   k_row_count / k_row_compact   histogram knee + score/mask filter, raster-ordered vCorners, vCornerRowLUT   :279-315, 346-355
//   k_nonmax_* / k_candidates     fast_nonmax + FAST / Shi-Tomasi scores, border 10       :393-420
//   k_minipatch              MiniPatch::FindPatch                     src/MiniPatch.cc:34-113
//   k_track_search           TrackerData::Project/CalcJacobian + PatchFinder (warp, template,
//                            ZMSSD search, sub-pixel)                 src/PatchFinder.cc:69-472,511-664
//   k_pose_*                 Tracker::CalcPoseUpdate                  src/Tracker.cc:1386-1512
// This translation unit is compiled with -ffp-contract=off so that the float/double arithmetic
// of the sub-pixel iterations rounds exactly as the scalar reference code does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ba_device.h"
#include "ba_select.h"
#include "../../include/mcp_img.h"

namespace mcp {

struct LevelInfo {            // device-resident bookkeeping of one pyramid level
  int n_all;                  // corners detected at the minimum threshold (raster order)
  int n_corners;              // corners kept (Level::vCorners)
  int thresh;                 // Level::nFastThresh
  int n_cand;                 // nonmax survivors inside the border (MakeKeyFrame_Rest)
  int hist[32];               // Level::vFastFrequency (cumulative)
  int overflow;
};

__global__ void k_dilate5(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int w, int h) {
  const int x = blockIdx.x*blockDim.x + threadIdx.x, y = blockIdx.y*blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  int m = 0;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      if ((dy == -2 || dy == 2) && dx != 0) continue;      // 5x5 MORPH_ELLIPSE
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      m = max(m, (int)in[(size_t)yy*w + xx]);
    }
  out[(size_t)y*w + x] = (uint8_t)m;
}
__global__ void k_glare_mask(const uint8_t* __restrict__ dil, const uint8_t* __restrict__ internal, uint8_t* __restrict__ out, int n) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t g = dil[i] > 245 ? 0 : 255;
  out[i] = internal ? (internal[i] & g) : g;
}

// ---- FAST-10 -------------------------------------------------------------------------------
__device__ inline void fast_ring(const uint8_t* __restrict__ p, int w, int* r) {
  r[0] = p[3*w];       r[1] = p[3*w + 1];   r[2] = p[2*w + 2];   r[3] = p[w + 3];
  r[4] = p[3];         r[5] = p[-w + 3];    r[6] = p[-2*w + 2];  r[7] = p[-3*w + 1];
  r[8] = p[-3*w];      r[9] = p[-3*w - 1];  r[10] = p[-2*w - 2]; r[11] = p[-w - 3];
  r[12] = p[-3];       r[13] = p[w - 3];    r[14] = p[2*w - 2];  r[15] = p[3*w - 1];
}
__device__ inline unsigned rot16(unsigned m, int k) { return ((m >> k) | (m << (16 - k))) & 0xffffu; }
__device__ inline bool run10(unsigned m) {       // >= 10 contiguous set bits on the 16-ring
  unsigned a = m & rot16(m, 1);        // 2
  a &= rot16(a, 2);                    // 4
  unsigned b = a & rot16(a, 4);        // 8
  return (b & rot16(a, 6)) != 0;       // 8 + window shifted by 6 covers bits s..s+9
}
__device__ inline bool fast10_corner(const int* r, int c, int b) {
  unsigned br = 0, dk = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { br |= (unsigned)(r[k] > c + b) << k; dk |= (unsigned)(r[k] < c - b) << k; }
  return run10(br) || run10(dk);
}
// largest threshold that still passes the segment test == max over arcs of the arc minimum of the
// signed differences, minus one (what the reference's binary search converges to)
__device__ inline int fast10_score(const int* r, int c) {
  int best = -1000;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = pass ? (c - r[k]) : (r[k] - c);
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m8[k] = min(m4[k], m4[(k + 4) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) best = max(best, min(m8[k], m2[(k + 8) & 15]));
  }
  return min(best - 1, 254);
}
__device__ inline int ring_sad_score(const int* r, int c, int barrier) {
  const int cb = c + barrier, c_b = c - barrier;
  int sp = 0, sn = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { if (r[k] > cb) sp += r[k] - cb; else if (r[k] < c_b) sn += c_b - r[k]; }
  return max(sp, sn);
}

constexpr int FAST_BLOCK = 256;
// rank of this thread among the flagged threads of the block (raster order) and the block total
__device__ inline int block_rank(bool flag, int* total, int* lds /* FAST_BLOCK/64 + 1 */) {
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) lds[wave] = __popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < FAST_BLOCK/64; ++i) { if (i < wave) base += lds[i]; tot += lds[i]; }
  __syncthreads();
  *total = tot;
  return base + in_wave;
}
__device__ inline int block_prefix(const int* __restrict__ cnt, int b, int* lds) {   // sum of cnt[0..b)
  int s = 0;
  for (int i = threadIdx.x; i < b; i += FAST_BLOCK) s += cnt[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  int t = 0;
  for (int i = 0; i < FAST_BLOCK/64; ++i) t += lds[i];
  __syncthreads();
  return t;
}

__device__ inline int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// knee of the cumulative histogram (KeyFrame.cc:279-300)
__device__ inline int knee_threshold(const int* hist, int w, int h) {
  const double target = -1*(w*h)/500.0;
  int th = MCP_MIN_FAST_THRESH;
  for (int t = MCP_MIN_FAST_THRESH; t <= MCP_MAX_FAST_THRESH; ++t) {
    double deriv;
    if (t == MCP_MIN_FAST_THRESH) deriv = (double)hist[t + 1] - (double)hist[t];
    else if (t == MCP_MAX_FAST_THRESH) deriv = (double)hist[t] - (double)hist[t - 1];
    else deriv = ((double)hist[t + 1] - (double)hist[t - 1])/2.0;
    th = t;
    if (deriv > target) break;
  }
  return th;
}

// ---- MakeKeyFrame_Lite in three launches for all levels of all cameras of a frame ----------------------
// k_pyr_fast     one workgroup per 64x64 tile of level 0: the tile + a 24-pixel apron is staged in LDS, levels 1..3 are
//                half-sampled LDS -> LDS (the apron shrinks to the 3 pixels FAST needs at level 3; values in the aprons are
//                recomputed by the neighbouring tiles, bit for bit the same integers), every level's interior is written
//                out once, FAST-10 + its score run on the LDS tiles and leave one score byte per pixel (0 = no corner at the
//                detection threshold) and the per-score corner counts (integer atomics: order-free; k_row_count sums them into the cumulative
//                threshold histogram).
// k_row_count    threshold from the histogram knee (or the fixed one), kept corners per image row (one wavefront per row).
// k_row_compact  exclusive prefix over the rows = vCornerRowLUT; the row's corners are written in x order at that offset, so
//                vCorners comes out in raster order exactly as fast_corner_detect_10 + the filter loop produce it.
#ifndef PYR_NT
#define PYR_NT 512                        // threads per tile workgroup: a tile's phases are latency chains of one workgroup (load 15 k, half-samples 7 k, level-0
                                          // detection 23 k cycles with 256 threads), and the c3 frame has 1.25 tiles per compute unit -- 256: 34.0 / 78.2 us
                                          // per launch at c3 / c5, 512: 26.7 / 72.5, 1024: 24.7 / 79.0
#endif
constexpr int PYR_T = 64;                 // level-0 tile edge
constexpr int PYR_A = 24;                 // level-0 apron = 3 << (MCP_LEVELS - 1)
constexpr int PYR_R = PYR_T + 2*PYR_A;    // 112: staged region edge at level 0 (56, 28, 14 above)
constexpr int FRAME_WORK_INTS = MCP_LEVELS*32 + MCP_LEVELS;      // histogram[level][32], corners detected[level]
struct FrameCam {             // one camera of a frame: everything the three kernels touch
  const uint8_t* src; int src_stride;           // level-0 pixels (device); == img[0] when the upload went straight there
  int w, h;
  uint8_t* img[MCP_LEVELS]; uint8_t* score[MCP_LEVELS]; const uint8_t* mask[MCP_LEVELS];
  mcp_int2* corners[MCP_LEVELS]; int* lut[MCP_LEVELS]; int* rowcnt[MCP_LEVELS]; LevelInfo* info[MCP_LEVELS];
  LevelInfo* host_info;                          // pinned, device-visible: the four levels' bookkeeping lands on the host with the frame
  int* work; int cap[MCP_LEVELS];
  unsigned long long* scan[MCP_LEVELS]; unsigned int epoch;      // k_row_tables: kept corners per 4-row workgroup, tagged with the frame's epoch
};
struct FrameBatch { FrameCam c[MCP_MAX_FRAME_CAMS]; int ncam, adaptive, pavgb; int detect_t[MCP_LEVELS];
                    // mcp_track_frame: the search's small inputs (camera table, tracked points; pinned host memory) are copied to the device by
                    // one more z-slice of k_row_count's grid, beside the pyramid's own work: no copy-engine operation (and its two
                    // ~8 us switches between compute and copy queue) between the corner tables and the search
                    const unsigned long long* up_src[4]; unsigned long long* up_dst[4]; int up_n8[4]; };

__global__ void __launch_bounds__(PYR_NT)
k_pyr_fast(const FrameBatch B) {
  const FrameCam& C = B.c[blockIdx.y];
  const int ntx = (C.w + PYR_T - 1)/PYR_T, nty = (C.h + PYR_T - 1)/PYR_T;
  if ((int)blockIdx.x >= ntx*nty) return;
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx, tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) uint8_t r0[PYR_R*PYR_R];
  __shared__ __attribute__((aligned(16))) uint8_t r1[(PYR_R/2)*(PYR_R/2)];
  __shared__ __attribute__((aligned(16))) uint8_t r2[(PYR_R/4)*(PYR_R/4)];
  __shared__ __attribute__((aligned(16))) uint8_t r3[(PYR_R/8)*(PYR_R/8) + 4];
  __shared__ int lh[MCP_LEVELS][32];
  __shared__ int ln[MCP_LEVELS];
  __shared__ unsigned short queue[PYR_T*PYR_T];
  __shared__ int qn;
  if (tid < MCP_LEVELS*32) (&lh[0][0])[tid] = 0;
  if (tid < MCP_LEVELS) ln[tid] = 0;
  // level 0: 112 rows of 28 words
  {
    const int ox = tx*PYR_T - PYR_A, oy = ty*PYR_T - PYR_A;
    const bool al = ((((uintptr_t)C.src) | (uintptr_t)C.src_stride) & 3) == 0;
    for (int i = tid; i < PYR_R*(PYR_R/4); i += PYR_NT) {
      const int ry = i/(PYR_R/4), rx = (i % (PYR_R/4))*4, gy = oy + ry, gx = ox + rx;
      uint32_t v = 0;
      if (gy >= 0 && gy < C.h && gx + 3 >= 0 && gx < C.w) {
        const uint8_t* row = C.src + (size_t)gy*C.src_stride;
        if (al && gx >= 0 && gx + 3 < C.w) v = *(const uint32_t*)(row + gx);
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (gx + q >= 0 && gx + q < C.w) v |= (uint32_t)row[gx + q] << (8*q);
        }
      }
      *(uint32_t*)&r0[ry*PYR_R + rx] = v;
    }
  }
  __syncthreads();
  uint8_t* const reg[MCP_LEVELS] = { r0, r1, r2, r3 };
#pragma unroll
  for (int l = 1; l < MCP_LEVELS; ++l) {
    const int n = PYR_R >> l, wl = C.w >> l, hl = C.h >> l, ox = (tx*PYR_T - PYR_A) >> l, oy = (ty*PYR_T - PYR_A) >> l;     // arithmetic shift: the origin is a multiple of 8
    const uint8_t* in = reg[l - 1]; uint8_t* out = reg[l]; const int ni = n*2;
    for (int i = tid; i < n*n; i += PYR_NT) {
      const int y = i / n, x = i % n, gx = ox + x, gy = oy + y;
      int v = 0;
      if (gx >= 0 && gx < wl && gy >= 0 && gy < hl) {
        const int a = in[(2*y)*ni + 2*x], b = in[(2*y)*ni + 2*x + 1], c = in[(2*y + 1)*ni + 2*x], d = in[(2*y + 1)*ni + 2*x + 1];
        if (B.pavgb) { const int v0 = (a + c + 1) >> 1, v1 = (b + d + 1) >> 1; v = (v0 + v1 + 1) >> 1; }
        else v = (a + b + c + d)/4;
      }
      out[i] = (uint8_t)v;
    }
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < MCP_LEVELS; ++l) {
    const int n = PYR_R >> l, t = PYR_T >> l, a = PYR_A >> l, wl = C.w >> l, hl = C.h >> l, gx0 = (tx*PYR_T) >> l, gy0 = (ty*PYR_T) >> l;
    const uint8_t* R = reg[l];
    const bool write_img = l > 0 || C.src != C.img[0];
    const int b = B.detect_t[l];
    // detection: every pixel; the corners (a few per cent) are queued by tile position and scored afterwards with all lanes busy --
    // scoring inside this loop would make a whole wavefront pay for fast10_score whenever one of its 64 pixels is a corner
    if (tid == 0) qn = 0;
    __syncthreads();
    for (int i = tid; i < t*t; i += PYR_NT) {
      const int ly = i / t, lx = i % t, gx = gx0 + lx, gy = gy0 + ly;
      if (gx >= wl || gy >= hl) continue;
      const uint8_t* p = R + (a + ly)*n + a + lx;
      const int c = *p;
      bool corner = false;
      if (gy >= 3 && gy < hl - 3 && gx >= 3 && gx < wl - 3) {
        int r[16]; fast_ring(p, n, r);
        corner = fast10_corner(r, c, b);
      }
      const size_t g = (size_t)gy*wl + gx;
      if (corner) queue[atomicAdd(&qn, 1)] = (unsigned short)i;      // (order is irrelevant: scores go to their pixel, the counts are sums)
      else C.score[l][g] = 0;
      if (write_img) C.img[l][g] = (uint8_t)c;
    }
    __syncthreads();
    const int nq = qn;
    for (int j = tid; j < nq; j += PYR_NT) {
      const int i = queue[j], ly = i / t, lx = i % t;
      const uint8_t* p = R + (a + ly)*n + a + lx;
      int r[16]; fast_ring(p, n, r);
      const int sc = fast10_score(r, *p);
      C.score[l][(size_t)(gy0 + ly)*wl + gx0 + lx] = (uint8_t)sc;
      if (B.adaptive) atomicAdd(&lh[l][min(sc, MCP_MAX_FAST_THRESH)], 1);       // raw count per (capped) score; k_row_count turns it into the cumulative vFastFrequency
    }
    if (tid == 0) ln[l] = nq;
    __syncthreads();
  }
  __syncthreads();
  if (tid < MCP_LEVELS*32) { const int v = (&lh[0][0])[tid]; if (v) atomicAdd(&C.work[tid], v); }
  if (tid < MCP_LEVELS && ln[tid]) atomicAdd(&C.work[MCP_LEVELS*32 + tid], ln[tid]);
}

__device__ inline bool corner_kept(const FrameCam& C, int l, int wl, int x, int y, int th, bool use_mask) {
  const size_t g = (size_t)y*wl + x;
  const int sc = C.score[l][g];
  if (sc == 0 || sc < th) return false;
  return !use_mask || C.mask[l][g] == 255;
}
__global__ void __launch_bounds__(256)
k_row_count(const FrameBatch B) {
  if ((int)blockIdx.z == B.ncam) {               // the upload slice
    const int nth = (int)(gridDim.x*gridDim.y)*256, t0 = (int)(blockIdx.y*gridDim.x + blockIdx.x)*256 + (int)threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) for (int i = t0; i < B.up_n8[r]; i += nth) B.up_dst[r][i] = B.up_src[r][i];
    return;
  }
  const int l = blockIdx.y; const FrameCam& C = B.c[blockIdx.z];
  const int wl = C.w >> l, hl = C.h >> l, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x*4 >= hl) return;
  __shared__ int th_s;
  __shared__ int cum[32];
  // vFastFrequency[t] = corners whose score reaches t (KeyFrame.cc:264-275) = suffix sum of the raw counts per capped score
  if (threadIdx.x < 32) {
    int c = 0;
    if ((int)threadIdx.x >= MCP_MIN_FAST_THRESH) for (int q = threadIdx.x; q <= MCP_MAX_FAST_THRESH; ++q) c += C.work[l*32 + q];
    cum[threadIdx.x] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) th_s = B.adaptive ? knee_threshold(cum, wl, hl) : B.detect_t[l];
  __syncthreads();
  const int th = th_s;
  if (blockIdx.x == 0) {                        // the level's bookkeeping, completed by k_row_compact
    LevelInfo* I = C.info[l];
    if (threadIdx.x < 32) I->hist[threadIdx.x] = cum[threadIdx.x];
    if (threadIdx.x == 0) { I->thresh = th; I->n_all = C.work[MCP_LEVELS*32 + l]; I->n_cand = 0; }
  }
  const int y = blockIdx.x*4 + wave;
  if (y >= hl) return;
  const bool use_mask = B.adaptive && C.mask[l] != nullptr;         // the fixed-threshold branch keeps every detected corner, KeyFrame.cc:318-343
  int cnt = 0;
  for (int x = lane; x < wl; x += 64) cnt += corner_kept(C, l, wl, x, y, th, use_mask) ? 1 : 0;
  cnt = wave_sum_i(cnt);
  if (lane == 0) C.rowcnt[l][y] = cnt;
}
__global__ void __launch_bounds__(256)
k_row_compact(const FrameBatch B) {
  const int l = blockIdx.y; const FrameCam& C = B.c[blockIdx.z];
  const int wl = C.w >> l, hl = C.h >> l, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.x*4;
  if (y0 >= hl) return;
  __shared__ int part[4];
  if (blockIdx.x == 0) {                        // the histogram accumulators are free again: leave them zero for the next frame
    if (threadIdx.x < 32) C.work[l*32 + threadIdx.x] = 0;
    if (threadIdx.x == 0) C.work[MCP_LEVELS*32 + l] = 0;
  }
  int s = 0;
  for (int i = threadIdx.x; i < y0; i += 256) s += C.rowcnt[l][i];
  s = wave_sum_i(s);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  int off = part[0] + part[1] + part[2] + part[3];
  const int y = y0 + wave;
  if (y >= hl) return;
  for (int j = 0; j < wave; ++j) off += C.rowcnt[l][y0 + j];
  const int th = C.info[l]->thresh;
  const bool use_mask = B.adaptive && C.mask[l] != nullptr;
  if (lane == 0) C.lut[l][y] = off;
  int run = off;
  for (int x0 = 0; x0 < wl; x0 += 64) {
    const int x = x0 + lane;
    const bool keep = x < wl && corner_kept(C, l, wl, x, y, th, use_mask);
    const unsigned long long bal = __ballot(keep);
    if (keep) { const int o = run + __popcll(bal & ((1ull << lane) - 1ull)); if (o < C.cap[l]) { C.corners[l][o].x = x; C.corners[l][o].y = y; } }
    run += __popcll(bal);
  }
  if (y == hl - 1 && lane == 0) {
    LevelInfo* I = C.info[l];
    I->n_corners = min(run, C.cap[l]); I->overflow = run > C.cap[l];
    LevelInfo* H = C.host_info + l;              // k_row_count's stores to *I are visible: it ran in the previous launch
    H->n_all = I->n_all; H->n_corners = I->n_corners; H->thresh = I->thresh; H->n_cand = 0; H->overflow = I->overflow;
    for (int q = 0; q < 32; ++q) H->hist[q] = I->hist[q];
  }
}

// k_row_count + k_row_compact in ONE launch (round 6): a workgroup counts its four rows, publishes the sum (tagged with the frame's epoch: no
// clearing between frames), adds up the sums of the workgroups above it as they appear -- every workgroup publishes BEFORE it looks at
// anybody else's, and the ones it waits for have lower block indices, i.e. were dispatched before it: no chain, no deadlock -- and writes
// its rows' corners at the resulting offset.  The exclusive prefix over the rows is vCornerRowLUT as before; the corner order, the
// threshold and the bookkeeping are those of the two kernels, bit for bit.  Saves a launch boundary and k_row_compact's prologue (its
// own walk over the row counts) per frame.
constexpr unsigned long long ROWSCAN_TAG_SHIFT = 32;
__global__ void __launch_bounds__(256)
k_row_tables(const FrameBatch B) {
  if ((int)blockIdx.z == B.ncam) {               // the upload slice
    const int nth = (int)(gridDim.x*gridDim.y)*256, t0 = (int)(blockIdx.y*gridDim.x + blockIdx.x)*256 + (int)threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) for (int i = t0; i < B.up_n8[r]; i += nth) B.up_dst[r][i] = B.up_src[r][i];
    return;
  }
  const int l = blockIdx.y; const FrameCam& C = B.c[blockIdx.z];
  const int wl = C.w >> l, hl = C.h >> l, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.x*4, nblk = (hl + 3) >> 2;
  if (y0 >= hl) return;
  __shared__ int th_s;
  __shared__ int cum[32];
  __shared__ int rc[4];
  __shared__ int s_off;
  if (threadIdx.x < 32) {
    int c = 0;
    if ((int)threadIdx.x >= MCP_MIN_FAST_THRESH) for (int q = threadIdx.x; q <= MCP_MAX_FAST_THRESH; ++q) c += C.work[l*32 + q];
    cum[threadIdx.x] = c;
  }
  if (threadIdx.x < 4) rc[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) th_s = B.adaptive ? knee_threshold(cum, wl, hl) : B.detect_t[l];
  __syncthreads();
  const int th = th_s;
  const int n_all = C.work[MCP_LEVELS*32 + l];
  if (blockIdx.x == 0) {
    LevelInfo* I = C.info[l];
    if (threadIdx.x < 32) I->hist[threadIdx.x] = cum[threadIdx.x];
    if (threadIdx.x == 0) { I->thresh = th; I->n_all = n_all; I->n_cand = 0; }
  }
  const int y = y0 + wave;
  const bool use_mask = B.adaptive && C.mask[l] != nullptr;
  int cnt = 0;
  if (y < hl) for (int x = lane; x < wl; x += 64) cnt += corner_kept(C, l, wl, x, y, th, use_mask) ? 1 : 0;
  cnt = wave_sum_i(cnt);
  if (lane == 0) rc[wave] = cnt;
  __syncthreads();
  const unsigned long long tag = (unsigned long long)C.epoch << ROWSCAN_TAG_SHIFT;
  unsigned long long* scan = C.scan[l];
  if (threadIdx.x == 0) __hip_atomic_store(scan + blockIdx.x, tag | (unsigned int)(rc[0] + rc[1] + rc[2] + rc[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    // the sums of the workgroups above, 64 at a time
    int off = 0; long long t0 = 0; unsigned int it = 0; bool dead = false;
    for (int base = 0; base < (int)blockIdx.x && !dead; base += 64) {
      const int j = base + lane;
      const bool mine = j < (int)blockIdx.x;
      unsigned long long v = tag;
      for (;;) {
        if (mine) v = __hip_atomic_load(scan + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__builtin_amdgcn_ballot_w64((v >> ROWSCAN_TAG_SHIFT) != C.epoch)) break;
        if ((++it & 63) == 63) { if (!t0) t0 = wall_clock64(); else if (wall_clock64() - t0 > 2000000ll) { dead = true; break; } }      // 20 ms: give up (overflow is raised below)
        __builtin_amdgcn_s_sleep(1);
      }
      off += wave_sum_i(mine ? (int)(unsigned int)v : 0);
    }
    if (lane == 0) s_off = dead ? -1 : off;
  }
  __syncthreads();
  int off = s_off;
  const bool dead = off < 0;
  if (dead) off = 0;
  if (y >= hl) return;
  for (int j = 0; j < wave; ++j) off += rc[j];
  if (lane == 0) C.lut[l][y] = off;
  int run = off;
  for (int x0 = 0; x0 < wl; x0 += 64) {
    const int x = x0 + lane;
    const bool keep = x < wl && corner_kept(C, l, wl, x, y, th, use_mask);
    const unsigned long long bal = __ballot(keep);
    if (keep) { const int o = run + __popcll(bal & ((1ull << lane) - 1ull)); if (o < C.cap[l]) { C.corners[l][o].x = x; C.corners[l][o].y = y; } }
    run += __popcll(bal);
  }
  if (y == hl - 1 && lane == 0) {
    // the last row: every workgroup of the level has published, i.e. has read the histogram accumulators -- they are left zero for the next frame
    for (int q = 0; q < 32; ++q) C.work[l*32 + q] = 0;
    C.work[MCP_LEVELS*32 + l] = 0;
    LevelInfo* I = C.info[l];
    I->n_corners = min(run, C.cap[l]); I->overflow = (run > C.cap[l]) || dead;
    LevelInfo* H = C.host_info + l;
    H->n_all = n_all; H->n_corners = I->n_corners; H->thresh = th; H->n_cand = 0; H->overflow = I->overflow;
    for (int q = 0; q < 32; ++q) H->hist[q] = cum[q];
  }
  (void)nblk;
}

// ---- MakeKeyFrame_Rest -----------------------------------------------------------------------
__global__ void k_nonmax_scores(const uint8_t* __restrict__ img, int w, const mcp_int2* __restrict__ corners,
                                const LevelInfo* __restrict__ info, int mode, int* __restrict__ score_img) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= info->n_corners) return;
  const int x = corners[i].x, y = corners[i].y;
  int r[16]; const uint8_t* p = img + (size_t)y*w + x;
  fast_ring(p, w, r);
  const int s = mode ? ring_sad_score(r, *p, info->thresh) : fast10_score(r, *p);
  score_img[(size_t)y*w + x] = s + 1;            // 0 = not a corner
}
__device__ inline double shi_tomasi7(const uint8_t* __restrict__ img, int w, int cx, int cy) {
  double dXX = 0, dYY = 0, dXY = 0;
  for (int y = cy - 3; y <= cy + 3; ++y)
    for (int x = cx - 3; x <= cx + 3; ++x) {
      const double dx = (double)((int)img[(size_t)y*w + x + 1] - (int)img[(size_t)y*w + x - 1]);
      const double dy = (double)((int)img[(size_t)(y + 1)*w + x] - (int)img[(size_t)(y - 1)*w + x]);
      dXX += dx*dx; dYY += dy*dy; dXY += dx*dy;
    }
  dXX = dXX/(2.0*49); dYY = dYY/(2.0*49); dXY = dXY/(2.0*49);
  return 0.5*(dXX + dYY - sqrt((dXX + dYY)*(dXX + dYY) - 4*(dXX*dYY - dXY*dXY)));
}
template <bool WRITE>
__global__ void __launch_bounds__(FAST_BLOCK)
k_candidates(const uint8_t* __restrict__ img, int w, int h, const mcp_int2* __restrict__ corners, LevelInfo* __restrict__ info,
             const int* __restrict__ score_img, int use_shi, int* __restrict__ blk_cnt, mcp_int2* __restrict__ out_pos, double* __restrict__ out_score) {
  __shared__ int lds[FAST_BLOCK/64 + 1];
  int off = 0;
  if (WRITE) off = block_prefix(blk_cnt, blockIdx.x, lds);
  const int i = blockIdx.x*FAST_BLOCK + threadIdx.x;
  bool keep = false; int x = 0, y = 0;
  if (i < info->n_corners) {
    x = corners[i].x; y = corners[i].y;
    const int s = score_img[(size_t)y*w + x];
    keep = true;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      if (!dx && !dy) continue;
      const int xx = x + dx, yy = y + dy;
      if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
      if (score_img[(size_t)yy*w + xx] > s) keep = false;
    }
    if (!(x >= 10 && y >= 10 && x < w - 10 && y < h - 10)) keep = false;
  }
  int tot; const int rank = block_rank(keep, &tot, lds);
  if (!WRITE) { if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot; return; }
  if (keep) {
    double sc;
    if (use_shi) sc = shi_tomasi7(img, w, x, y);
    else { int r[16]; const uint8_t* p = img + (size_t)y*w + x; fast_ring(p, w, r); sc = (double)fast10_score(r, *p); }
    out_pos[off + rank].x = x; out_pos[off + rank].y = y; out_score[off + rank] = sc;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) info->n_cand = off + tot;
}

// ---- MiniPatch ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_minipatch(const uint8_t* __restrict__ simg, int sw, int sh, const uint8_t* __restrict__ dimg, int dw, int dh,
            const mcp_int2* __restrict__ corners, const LevelInfo* __restrict__ info, const int* __restrict__ lut, int n,
            const mcp_int2* __restrict__ src_pos, const mcp_int2* __restrict__ dst_pos, int range,
            mcp_int2* __restrict__ out_pos, uint8_t* __restrict__ out_found, int* __restrict__ out_ssd) {
  __shared__ uint8_t patch[96];
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n) return;
  const int H = 4, MAXSSD = 9999;
  const mcp_int2 sp = src_pos[i], dp = dst_pos[i];
  const bool ok = sp.x >= H && sp.y >= H && sp.x < sw - H && sp.y < sh - H;
  for (int k = lane; k < 81; k += 64) patch[k] = ok ? simg[(size_t)(sp.y - H + k/9)*sw + sp.x - H + k%9] : 0;
  __syncthreads();
  int best = MAXSSD + 1, best_c = 0x7fffffff;
  if (ok) {
    const int tlx = dp.x - range, brx = dp.x + range, bry = dp.y + range;
    int top = dp.y - range; if (top < 0) top = 0; if (top >= dh) top = dh - 1;
    const int c0 = lut[top], c1 = (bry + 1 < dh) ? ((bry + 1 < 0) ? 0 : lut[bry + 1]) : info->n_corners;
    for (int c = c0 + lane; c < c1; c += 64) {
      const mcp_int2 p = corners[c];
      if (p.x < tlx || p.x > brx) continue;
      int ssd;
      if (!(p.x >= H && p.y >= H && p.x < dw - H && p.y < dh - H)) ssd = MAXSSD + 1;
      else {
        ssd = 0;
        for (int r = 0; r < 9; ++r) { const uint8_t* q = dimg + (size_t)(p.y - H + r)*dw + p.x - H; for (int k = 0; k < 9; ++k) { const int df = (int)q[k] - (int)patch[9*r + k]; ssd += df*df; } }
      }
      if (ssd < best) { best = ssd; best_c = c; }          // ascending c within the lane: first best kept
    }
  }
  // wave arg-min with "first best" semantics: smallest ssd, then smallest corner index
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o, 64), oc = __shfl_xor(best_c, o, 64);
    if (ob < best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
  }
  if (lane == 0) {
    out_found[i] = 0; out_pos[i] = dp; if (out_ssd) out_ssd[i] = best;
    if (best < MAXSSD) { out_pos[i] = corners[best_c]; out_found[i] = 1; }
  }
}

// ---- Tracker per-point search ---------------------------------------------------------------------
struct DevKfView {            // what a kernel needs of a keyframe
  const uint8_t* img[MCP_LEVELS]; int w[MCP_LEVELS], h[MCP_LEVELS];
  const mcp_int2* corners[MCP_LEVELS]; const int* lut[MCP_LEVELS]; const LevelInfo* info[MCP_LEVELS];
};
struct DevTdIn {
  double world_pos[3], pixel_right_w[3], pixel_down_w[3];
  const uint8_t* src_img; int src_w, src_h;
  int center_x, center_y, fixed;
};


// ---- PatchFinder as a stateful object (src/PatchFinder.cc:56-65): what survives a call -- the template cache of
// MakeTemplateCoarseCont (:144-181: mpLastTemplateMapPoint, mm2LastWarpMatrix, mimTemplate, mbTemplateBad), the template the
// sub-pixel Jacobians were last made from (MakeSubPixTemplate :362-390) and mdMeanDiff.  Scalars are wave-uniform registers,
// the two templates live in LDS (tmpl / jtmpl).  Callers and their differences: see mcp_img.h MCP_PF_*.
struct PfRegs { int valid, key, bad, jvalid; double lw[4]; double mean; };
constexpr int PF_TRACK = 0, PF_REFIND = 1, PF_EPI_COARSE = 2, PF_EPI_REFINE = 3;

// IterateSubPixToConvergence (:392-410) / IterateSubPix (:415-472) from sp[]: Jacobians of jtmpl, differences against tmpl, mean
// difference carried in `mean`; returns 1 converged, 0 iterations used up, -1 left the image.  Lanes 0..35 own the interior pixels.
__device__ __forceinline__ int pf_iterate(const uint8_t* __restrict__ limg, int lw, int lh, int scale, const uint8_t* tmpl, const uint8_t* jtmpl,
                                          double sp[2], double& mean, int its, double (*dprod)[36], int lane) {
  const int sy = lane/6 + 1, sx = lane%6 + 1;
  double gx = 0, gy = 0;
  if (lane < 36) { gx = 0.5*((int)jtmpl[8*sy + sx + 1] - (int)jtmpl[8*sy + sx - 1]); gy = 0.5*((int)jtmpl[8*(sy + 1) + sx] - (int)jtmpl[8*(sy - 1) + sx]); }
  // J^T J: sums of multiples of 1/4 -- exact in any order
  double H[9];
  { const double g[3] = { gx, gy, lane < 36 ? 1.0 : 0.0 };
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) { double v = g[a]*g[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        H[3*a + b] = v; } }
  double Hi[9];
  { const double c00 = H[4]*H[8] - H[5]*H[7], c01 = H[5]*H[6] - H[3]*H[8], c02 = H[3]*H[7] - H[4]*H[6];
    const double det = H[0]*c00 + H[1]*c01 + H[2]*c02, id = 1.0/det;
    Hi[0] = c00*id; Hi[1] = (H[2]*H[7] - H[1]*H[8])*id; Hi[2] = (H[1]*H[5] - H[2]*H[4])*id;
    Hi[3] = c01*id; Hi[4] = (H[0]*H[8] - H[2]*H[6])*id; Hi[5] = (H[2]*H[3] - H[0]*H[5])*id;
    Hi[6] = c02*id; Hi[7] = (H[1]*H[6] - H[0]*H[7])*id; Hi[8] = (H[0]*H[4] - H[1]*H[3])*id; }
  int conv = 0;
  for (int it = 0; it < its && conv == 0; ++it) {
    const double cx = (sp[0] + 0.5)/scale - 0.5, cy = (sp[1] + 0.5)/scale - 0.5;
    const int rx = (int)round(cx), ry = (int)round(cy);
    if (!(rx >= 5 && ry >= 5 && rx < lw - 5 && ry < lh - 5)) { conv = -1; break; }
    const double bxs = cx - 4, bys = cy - 4;
    const double dX = bxs - floor(bxs), dY = bys - floor(bys);
    const float fTL = (float)((1.0 - dX)*(1.0 - dY)), fTR = (float)(dX*(1.0 - dY)), fBL = (float)((1.0 - dX)*dY), fBR = (float)(dX*dY);
    if (lane < 36) {
      const uint8_t* q = limg + (size_t)((int)bys + sy)*lw + (int)bxs + sx;
      float fPixel = fTL*(float)q[0] + fTR*(float)q[1];
      fPixel = fPixel + fBL*(float)q[lw];
      fPixel = fPixel + fBR*(float)q[lw + 1];
      const double d = (double)fPixel - (double)tmpl[8*sy + sx] + mean;
      dprod[0][lane] = d*gx; dprod[1][lane] = d*gy; dprod[2][lane] = d;
    }
    __syncthreads();
    double acc[3] = {0, 0, 0};          // summed in the reference's pixel order for bit-equal rounding
    for (int q = 0; q < 36; ++q) { acc[0] += dprod[0][q]; acc[1] += dprod[1][q]; acc[2] += dprod[2][q]; }
    __syncthreads();
    double up[3]; mat3_vec(Hi, acc, up);
    sp[0] -= up[0]*scale; sp[1] -= up[1]*scale; mean -= up[2];
    const double u2 = up[0]*up[0] + up[1]*up[1];
    if (u2 < 0.03*0.03) conv = 1;
  }
  return conv;
}

// one item of one finder, one wavefront (control flow is wave-uniform: every lane computes the geometry redundantly)
__device__ __forceinline__ void patch_item(int mode, const DevKfView& T, const uint8_t* __restrict__ mask0, const mcp_camera& cam, const Se3& bfw, const Se3& cfb,
                                           const DevTdIn& P, int point_key, double start_x, double start_y, PfRegs& S, uint8_t* tmpl, uint8_t* jtmpl,
                                           mcp_td_out& O, int range, int subpix_its, int exhaustive, double (*dprod)[36], int lane,
                                           mcp_td_out* O2 = nullptr /* a second copy of the record (pinned host memory), or null */,
                                           mcp_pose_point* PP = nullptr /* the record the pose iterations read (what the loops of Tracker::TrackMap take from vTD after SearchForPoints, src/Tracker.cc:1040-1075: world position, found position, noise, projection + camera derivatives at the search pose, camera, found flag), or null */, int cam_index = 0) {
  const int MAXSSD = 8*8*250;
  Se3 cfw; se3_compose(cfb, bfw, cfw);
  double xc[3]; se3_apply(cfw, P.world_pos, xc);
  Projection pr; cam_project<true>(cam, xc, pr);
  bool go = true;
  if (mode == PF_TRACK || mode == PF_REFIND) {          // TrackerData.h:102-119; MapMakerServerBase.cc:941-953
    go = !pr.invalid && !(pr.u < 0 || pr.v < 0 || pr.u > cam.image_size[0] || pr.v > cam.image_size[1]);
  } else if (mode == PF_EPI_COARSE) {                   // :757-765: Invalid(), in_image(CVD::ir(v2Image)), mask == 0
    go = !pr.invalid;
    if (go) { const int ix = (int)pr.u, iy = (int)pr.v; go = ix >= 0 && iy >= 0 && ix < T.w[0] && iy < T.h[0] && !(mask0 && mask0[(size_t)iy*T.w[0] + ix] == 0); }
  }
  const bool in_image = go ? !pr.invalid : false;
  int level = -1, template_bad = 0, searched = 0, found = 0, did_subpix = 0, bx = 0, by = 0, best = MAXSSD + 1;
  double WI[4] = {0, 0, 0, 0}, J[12], fpos[2] = {0, 0}, sinv = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) J[k] = 0;
  bool have_templ = false;
  if (go) {
    double dT[3], dP[3]; cam_sphere_deriv(xc, dT, dP);
    double xb[3]; se3_apply(bfw, P.world_pos, xb);
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      double mb[3], mc[3]; generator(m, xb, mb); mat3_vec(cfb.R, mb, mc);
      const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
      J[m] = pr.D[0]*s0 + pr.D[1]*s1; J[6 + m] = pr.D[2]*s0 + pr.D[3]*s1;
    }
    double mr[3], md[3]; mat3_vec(cfw.R, P.pixel_right_w, mr); mat3_vec(cfw.R, P.pixel_down_w, md);
    const double sr0 = dT[0]*mr[0] + dT[1]*mr[1] + dT[2]*mr[2], sr1 = dP[0]*mr[0] + dP[1]*mr[1] + dP[2]*mr[2];
    const double sd0 = dT[0]*md[0] + dT[1]*md[1] + dT[2]*md[2], sd1 = dP[0]*md[0] + dP[1]*md[1] + dP[2]*md[2];
    WI[0] = pr.D[0]*sr0 + pr.D[1]*sr1; WI[2] = pr.D[2]*sr0 + pr.D[3]*sr1;
    WI[1] = pr.D[0]*sd0 + pr.D[1]*sd1; WI[3] = pr.D[2]*sd0 + pr.D[3]*sd1;
    double dDet = WI[0]*WI[3] - WI[1]*WI[2];
    int lv = 0;
    while (dDet > 3 && lv < MCP_LEVELS - 1) { lv++; dDet *= 0.25; }
    const bool rejected = (dDet > 3 || dDet < 0.5 || !isfinite(dDet));
    if (rejected) {
      S.bad = 1;                                        // PatchFinder.cc:116-117: the member is set before -1 is returned
      if (mode == PF_TRACK || mode == PF_EPI_COARSE) { template_bad = 1; go = false; }     // FindPVS drops the point; MapMakerServerBase.cc:769-770
    }
    if (go) level = lv;
  }
  if (go) {
    const int scale = 1 << level;
    double m2[4];
    { const double det = WI[0]*WI[3] - WI[1]*WI[2], id = 1.0/det;
      m2[0] = WI[3]*id*scale; m2[3] = WI[0]*id*scale; m2[2] = -WI[2]*id*scale; m2[1] = -WI[1]*id*scale; }
    // MakeTemplateCoarseCont :135-182: keep the template while the finder works on the same map point and neither column of the
    // warp matrix has moved by more than 0.07
    bool refresh = !S.valid || S.key != point_key;
    for (int c = 0; !refresh && c < 2; ++c) { const double d0 = m2[c] - S.lw[c], d1 = m2[2 + c] - S.lw[2 + c]; if (d0*d0 + d1*d1 > 0.07*0.07) refresh = true; }
    if (refresh) {
      // CVD::transform: incremental source position, replayed up to this lane's pixel so that it rounds identically
      const int iw = P.src_w, ih = P.src_h;
      const double across[2] = { m2[0], m2[2] }, down[2] = { m2[1], m2[3] };
      double p0[2] = { (double)P.center_x - (m2[0]*4.0 + m2[1]*4.0), (double)P.center_y - (m2[2]*4.0 + m2[3]*4.0) };
      double min_x = p0[0], min_y = p0[1], max_x = min_x, max_y = min_y;
      if (across[0] < 0) min_x += 8*across[0]; else max_x += 8*across[0];
      if (down[0] < 0) min_x += 8*down[0]; else max_x += 8*down[0];
      if (across[1] < 0) min_y += 8*across[1]; else max_y += 8*across[1];
      if (down[1] < 0) min_y += 8*down[1]; else max_y += 8*down[1];
      const double cr[2] = { down[0] - 8*across[0], down[1] - 8*across[1] };
      const bool inside = (min_x >= 0 && min_y >= 0 && max_x < iw - 1 && max_y < ih - 1);
      double p[2] = { p0[0], p0[1] };
      for (int q = 0; q < lane; ++q) { p[0] += across[0]; p[1] += across[1]; if ((q & 7) == 7) { p[0] += cr[0]; p[1] += cr[1]; } }
      bool outside = false; int tv = 0;
      if (inside || (0 <= p[0] && 0 <= p[1] && p[0] < (double)(iw - 1) && p[1] < (double)(ih - 1))) {
        const int lx = (int)p[0], ly = (int)p[1];
        const double x = p[0] - lx, y = p[1] - ly;
        const uint8_t* q = P.src_img + (size_t)ly*iw + lx;
        const double v = (1 - y)*((1 - x)*q[0] + x*q[1]) + y*((1 - x)*q[iw] + x*q[iw + 1]);
        tv = (int)(uint8_t)v;
      } else outside = true;
      __syncthreads();
      tmpl[lane] = (uint8_t)tv;
      S.bad = (__ballot(outside) != 0ull) ? 1 : 0;
      S.valid = 1; S.key = point_key; S.lw[0] = m2[0]; S.lw[1] = m2[1]; S.lw[2] = m2[2]; S.lw[3] = m2[3];
      if (!S.bad) { jtmpl[lane] = (uint8_t)tv; S.jvalid = 1; S.mean = 0.0; }               // MakeSubPixTemplate :176-177
      __syncthreads();
    }
    have_templ = true;
    template_bad = S.bad;
    const int tv = tmpl[lane];
    if (!(S.bad && mode != PF_EPI_REFINE)) {            // Tracker.cc:1316; MapMakerServerBase.cc:774, 958 (the refinement loop does not look)
      const int lw = T.w[level], lh = T.h[level];
      const uint8_t* limg = T.img[level];
      if (mode == PF_EPI_REFINE) {                       // :845-851: SetSubPixPos, IterateSubPixToConvergence(kfTarget, 10)
        sinv = 1.0/scale; did_subpix = 1;
        double sp[2] = { start_x, start_y };
        const int conv = S.jvalid ? pf_iterate(limg, lw, lh, scale, tmpl, jtmpl, sp, S.mean, 10, dprod, lane) : 0;
        found = (conv == 1); fpos[0] = sp[0]; fpos[1] = sp[1];
      } else {
        const int tsum = wave_sum_i(tv), tsumsq = wave_sum_i(tv*tv);
        const bool bex = (mode == PF_TRACK) && (P.fixed || exhaustive);
        const int its = bex ? 10 : subpix_its;
        int px = (int)pr.u, py = (int)pr.v;
        px = px/scale; py = py/scale;
        const unsigned nr = ((unsigned)range + scale - 1)/scale;
        int top = py - (int)nr, bot1 = py + (int)nr + 1, left = px - (int)nr, right = px + (int)nr;
        searched = 1;
        bool early = false;
        if (top < 0) top = 0;
        if (top >= lh) early = true;
        if (bot1 <= 0) early = true;
        if (left < 0) left = 0;
        if (left >= lw) early = true;
        if (!early) {
          // candidates in reference order; one candidate per lane, first-best arg-min
          int nc, c0 = 0, bw = 0;
          if (bex) { const int yb = min(bot1, lh), xr = min(right, lw - 1); bw = xr - left + 1; nc = (bw > 0 && yb > top) ? bw*(yb - top) : 0; }
          else { c0 = T.lut[level][top]; const int c1 = (bot1 >= lh) ? T.info[level]->n_corners : T.lut[level][bot1]; nc = c1 - c0; }
          int mybest = MAXSSD + 1, myidx = 0x7fffffff, mx = 0, my = 0;
          for (int c = lane; c < nc; c += 64) {
            int x, y;
            if (bex) { x = left + c % bw; y = top + c / bw; }
            else { x = T.corners[level][c0 + c].x; y = T.corners[level][c0 + c].y; if (x < left || x > right) continue; }
            if ((unsigned)((px - x)*(px - x) + (py - y)*(py - y)) > nr*nr) continue;
            int s2;
            if (!(x >= 4 && y >= 4 && x < lw - 4 && y < lh - 4)) s2 = MAXSSD + 1;
            else {
              int isum = 0, isumsq = 0, cross = 0;
              for (int r = 0; r < 8; ++r) { const uint8_t* ip = limg + (size_t)(y - 4 + r)*lw + x - 4;
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int v = ip[k]; isum += v; isumsq += v*v; cross += v*(int)tmpl[8*r + k]; } }
              s2 = ((2*tsum*isum - tsum*tsum - isum*isum)/64 + isumsq + tsumsq - 2*cross);
            }
            if (s2 < mybest) { mybest = s2; myidx = c; mx = x; my = y; }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const int ob = __shfl_xor(mybest, o, 64), oi = __shfl_xor(myidx, o, 64), ox = __shfl_xor(mx, o, 64), oy = __shfl_xor(my, o, 64);
            if (ob < mybest || (ob == mybest && oi < myidx)) { mybest = ob; myidx = oi; mx = ox; my = oy; }
          }
          best = mybest;
          if (best < MAXSSD) {
            found = 1; bx = mx; by = my;
            const double coarse[2] = { (bx + 0.5)*scale - 0.5, (by + 0.5)*scale - 0.5 };
            sinv = 1.0/scale; fpos[0] = coarse[0]; fpos[1] = coarse[1];
            if (mode == PF_TRACK && its > 0) {                                              // Tracker.cc:1350-1366
              did_subpix = 1;
              __syncthreads(); jtmpl[lane] = tmpl[lane]; S.jvalid = 1; S.mean = 0.0; __syncthreads();      // MakeSubPixTemplate
              double sp[2] = { coarse[0], coarse[1] };
              const int conv = pf_iterate(limg, lw, lh, scale, tmpl, jtmpl, sp, S.mean, its, dprod, lane);
              if (conv != 1) found = 0; else { fpos[0] = sp[0]; fpos[1] = sp[1]; }
            } else if (mode == PF_REFIND && level > 0) {                                    // MapMakerServerBase.cc:981-987: eight iterations, result kept either way
              did_subpix = 1;
              __syncthreads(); jtmpl[lane] = tmpl[lane]; S.jvalid = 1; S.mean = 0.0; __syncthreads();
              double sp[2] = { coarse[0], coarse[1] };
              (void)pf_iterate(limg, lw, lh, scale, tmpl, jtmpl, sp, S.mean, 8, dprod, lane);
              fpos[0] = sp[0]; fpos[1] = sp[1];
            }
          }
        }
      }
    }
  }
  O.templ[lane] = have_templ ? tmpl[lane] : (uint8_t)0;
  if (lane == 0) {
    O.image[0] = pr.u; O.image[1] = pr.v;
#pragma unroll
    for (int k = 0; k < 4; ++k) { O.cam_derivs[k] = pr.D[k]; O.warp_inverse[k] = WI[k]; }
#pragma unroll
    for (int k = 0; k < 12; ++k) O.jacobian[k] = J[k];
    O.found_pos[0] = fpos[0]; O.found_pos[1] = fpos[1]; O.sqrt_inv_noise = sinv;
    O.in_image = in_image; O.search_level = level; O.template_bad = template_bad; O.searched = searched;
    O.found = found; O.did_subpix = did_subpix; O.coarse_x = bx; O.coarse_y = by; O.score = best;
  }
  if (O2) {
    O2->templ[lane] = have_templ ? tmpl[lane] : (uint8_t)0;
    if (lane == 0) {
      O2->image[0] = pr.u; O2->image[1] = pr.v;
#pragma unroll
      for (int k = 0; k < 4; ++k) { O2->cam_derivs[k] = pr.D[k]; O2->warp_inverse[k] = WI[k]; }
#pragma unroll
      for (int k = 0; k < 12; ++k) O2->jacobian[k] = J[k];
      O2->found_pos[0] = fpos[0]; O2->found_pos[1] = fpos[1]; O2->sqrt_inv_noise = sinv;
      O2->in_image = in_image; O2->search_level = level; O2->template_bad = template_bad; O2->searched = searched;
      O2->found = found; O2->did_subpix = did_subpix; O2->coarse_x = bx; O2->coarse_y = by; O2->score = best;
    }
  }
  if (PP && lane == 0) {
    PP->world_pos[0] = P.world_pos[0]; PP->world_pos[1] = P.world_pos[1]; PP->world_pos[2] = P.world_pos[2];
    PP->found_pos[0] = fpos[0]; PP->found_pos[1] = fpos[1]; PP->sqrt_inv_noise = sinv;
    PP->image[0] = pr.u; PP->image[1] = pr.v;
#pragma unroll
    for (int k = 0; k < 4; ++k) PP->cam_derivs[k] = pr.D[k];
    PP->cam = cam_index; PP->found = found;
  }
}

// one tracked point, one wavefront: Tracker::SearchForPoints with a PatchFinder that has seen nothing yet (the per-frame batch
// entries; a caller that keeps the finders across frames uses k_patch_sequences)
__device__ __forceinline__ void track_search_point(const DevKfView& T, const mcp_camera& cam, const Se3& bfw, const Se3& cfb, const DevTdIn& P, mcp_td_out& O,
                                                   int range, int subpix_its, int exhaustive, uint8_t* tmpl, uint8_t* jtmpl, double (*dprod)[36], int lane,
                                                   mcp_td_out* O2 = nullptr, mcp_pose_point* PP = nullptr, int cam_index = 0) {
  PfRegs S; S.valid = 0; S.key = -1; S.bad = 0; S.jvalid = 0; S.lw[0] = S.lw[1] = S.lw[2] = S.lw[3] = 0.0; S.mean = 0.0;
  patch_item(PF_TRACK, T, nullptr, cam, bfw, cfb, P, 0, 0.0, 0.0, S, tmpl, jtmpl, O, range, subpix_its, exhaustive, dprod, lane, O2, PP, cam_index);
}
__global__ void __launch_bounds__(64)
k_track_search(DevKfView T, mcp_camera cam, Se3 bfw, Se3 cfb, int n, const DevTdIn* __restrict__ in, int range,
               int subpix_its, int exhaustive, mcp_td_out* __restrict__ out) {
  __shared__ uint8_t tmpl[64], jtmpl[64];
  __shared__ double dprod[3][36];
  const int pi = blockIdx.x;
  if (pi >= n) return;
  track_search_point(T, cam, bfw, cfb, in[pi], out[pi], range, subpix_its, exhaustive, tmpl, jtmpl, dprod, threadIdx.x);
}
// the cameras of a frame in one launch (blockIdx.y = camera); the per-camera views, models and poses sit in a device table
struct SearchCam { DevKfView T; mcp_camera cam; Se3 cfb; int n, first; };
__global__ void __launch_bounds__(64)
k_track_search_batch(const SearchCam* __restrict__ tab, Se3 bfw, const DevTdIn* __restrict__ in, int range, int subpix_its, int exhaustive,
                     mcp_td_out* __restrict__ out, mcp_td_out* __restrict__ host_out /* the results once more, in pinned host memory (mcp_track_frame: no
                     copy back), or null */, mcp_pose_point* __restrict__ pose_pts /* the pose iterations' records, or null */) {
  __shared__ uint8_t tmpl[64], jtmpl[64];
  __shared__ double dprod[3][36];
  const SearchCam& S = tab[blockIdx.y];
  const int pi = blockIdx.x;
  if (pi >= S.n) return;
  const int idx = S.first + pi;
  track_search_point(S.T, S.cam, bfw, S.cfb, in[idx], out[idx], range, subpix_its, exhaustive, tmpl, jtmpl, dprod, threadIdx.x,
                     host_out ? host_out + idx : nullptr, pose_pts ? pose_pts + idx : nullptr, (int)blockIdx.y);
}
// Sequences of items through stateful finders: one wavefront per sequence, items in order, the finder's members in registers /
// LDS between them and in `state` before and after.  tab: the targets (keyframe view, camera, poses; SearchCam::n / first unused).
struct PfItemDev { DevTdIn p; int point_key, target; double start_x, start_y; };
struct PfTargetDev { DevKfView T; const uint8_t* mask0; mcp_camera cam; Se3 bfw, cfb; };
__global__ void __launch_bounds__(64)
k_patch_sequences(int mode, const PfTargetDev* __restrict__ tab, int n_seq, const int* __restrict__ seq_start, const PfItemDev* __restrict__ items,
                  mcp_pf_state* __restrict__ state, int range, int subpix_its, int exhaustive, mcp_td_out* __restrict__ out,
                  mcp_td_out* __restrict__ host_out = nullptr /* mcp_track_frame: the results and ... */, mcp_pf_state* __restrict__ host_state = nullptr /* ... the finders' states once more,
                  in pinned host memory (no copies back) */, mcp_pose_point* __restrict__ pose_pts = nullptr /* the pose iterations' records */) {
  __shared__ uint8_t tmpl[64], jtmpl[64];
  __shared__ double dprod[3][36];
  const int sq = blockIdx.x, lane = threadIdx.x;
  if (sq >= n_seq) return;
  mcp_pf_state& G = state[sq];
  PfRegs S; S.valid = G.valid; S.key = G.point_key; S.bad = G.template_bad; S.jvalid = G.jacs_valid; S.mean = G.mean_diff;
  S.lw[0] = G.last_warp[0]; S.lw[1] = G.last_warp[1]; S.lw[2] = G.last_warp[2]; S.lw[3] = G.last_warp[3];
  tmpl[lane] = G.templ[lane]; jtmpl[lane] = G.jac_templ[lane];
  __syncthreads();
  for (int i = seq_start[sq]; i < seq_start[sq + 1]; ++i) {
    const PfItemDev& I = items[i];
    const PfTargetDev& Tg = tab[I.target];
    patch_item(mode, Tg.T, Tg.mask0, Tg.cam, Tg.bfw, Tg.cfb, I.p, I.point_key, I.start_x, I.start_y, S, tmpl, jtmpl, out[i], range, subpix_its, exhaustive, dprod, lane,
               host_out ? host_out + i : nullptr, pose_pts ? pose_pts + i : nullptr, I.target);
    __syncthreads();
  }
  G.templ[lane] = tmpl[lane]; G.jac_templ[lane] = jtmpl[lane];
  if (lane == 0) {
    G.valid = S.valid; G.point_key = S.key; G.template_bad = S.bad; G.jacs_valid = S.jvalid; G.mean_diff = S.mean;
    G.last_warp[0] = S.lw[0]; G.last_warp[1] = S.lw[1]; G.last_warp[2] = S.lw[2]; G.last_warp[3] = S.lw[3];
  }
  if (host_state) {
    mcp_pf_state& H = host_state[sq];
    H.templ[lane] = tmpl[lane]; H.jac_templ[lane] = jtmpl[lane];
    if (lane == 0) {
      H.valid = S.valid; H.point_key = S.key; H.template_bad = S.bad; H.jacs_valid = S.jvalid; H.mean_diff = S.mean;
      H.last_warp[0] = S.lw[0]; H.last_warp[1] = S.lw[1]; H.last_warp[2] = S.lw[2]; H.last_warp[3] = S.lw[3];
    }
  }
}

// ---- Tracker::CalcPoseUpdate ------------------------------------------------------------------------
// The M-estimator Tracker::CalcPoseUpdate dispatches on (Tracker::sMEstimatorName, src/Tracker.cc:1388-1401, 1429-1468) with the
// formulas of include/mcptam/MEstimator.h: Tukey :84-124, Cauchy :131-157, Huber :164-204.  est: MCP_MEST_TUKEY 0 / CAUCHY 1 / HUBER 2.
__device__ inline double mest_sigma_sq(int est, double n, double med) {
  double s = 1.4826*(1 + 5.0/mest_denom(n))*sqrt(med);
  s = ((est == 2) ? 1.345 : 4.6851)*s;
  return s*s;
}
__device__ inline double mest_weight(int est, double e, double s2) {
  if (est == 1) return 1.0/(1.0 + e/s2);
  if (est == 2) return (e < s2) ? 1.0 : sqrt(s2/e);
  const double sq = (e > s2) ? 0.0 : 1.0 - (e/s2);
  return sq*sq;
}
__global__ void k_pose_errors(int n, const uint8_t* __restrict__ found, const double* __restrict__ fpos, const double* __restrict__ ipos,
                              const double* __restrict__ sinv, double* __restrict__ ex, double* __restrict__ e2_compact,
                              const int* __restrict__ slot) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n || !found[i]) return;
  const double a = sinv[i]*(fpos[2*i] - ipos[2*i]), b = sinv[i]*(fpos[2*i + 1] - ipos[2*i + 1]);
  ex[2*i] = a; ex[2*i + 1] = b;
  e2_compact[slot[i]] = a*a + b*b;
}
// one block: Tukey weights, WLS<6> accumulation (prior 100), 6x6 Cholesky solve.  sig[0] = sigma^2
__global__ void __launch_bounds__(256)
k_pose_solve(int n, const uint8_t* __restrict__ found, const double* __restrict__ ex, const double* __restrict__ sinv,
             const double* __restrict__ J, const double* __restrict__ sig, double* __restrict__ mu, double* __restrict__ wout, int est) {
  __shared__ double red[27][4];
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  const double s2 = sig[0];
  for (int i = threadIdx.x; i < n; i += 256) {
    double w = 0.0;
    if (found[i]) {
      const double e = ex[2*i]*ex[2*i] + ex[2*i + 1]*ex[2*i + 1];
      w = mest_weight(est, e, s2);
      if (w != 0.0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double Jr[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) Jr[k] = sinv[i]*J[12*(size_t)i + 6*r + k];
          const double m = ex[2*i + r];
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) { acc[21 + a] += w*m*Jr[a];
#pragma unroll
            for (int b = 0; b <= a; ++b) { acc[q] += w*Jr[a]*Jr[b]; ++q; } }
        }
      }
    }
    if (wout) wout[i] = w;
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double C[36], v[6], L[36], x[6];
    int q = 0;
    for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) { const double t = red[q][0] + red[q][1] + red[q][2] + red[q][3]; C[6*a + b] = t; C[6*b + a] = t; ++q; }
    for (int a = 0; a < 6; ++a) { C[7*a] += 100.0; v[a] = red[21 + a][0] + red[21 + a][1] + red[21 + a][2] + red[21 + a][3]; }
    for (int i = 0; i < 36; ++i) L[i] = C[i];
    for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) { double s = L[6*i + j]; for (int k = 0; k < j; ++k) s -= L[6*i + k]*L[6*j + k]; L[6*i + j] = (i == j) ? sqrt(s) : s/L[6*j + j]; }
    for (int i = 0; i < 6; ++i) { double s = v[i]; for (int k = 0; k < i; ++k) s -= L[6*i + k]*x[k]; x[i] = s/L[6*i + i]; }
    for (int i = 5; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < 6; ++k) s -= L[6*k + i]*x[k]; x[i] = s/L[6*i + i]; }
    for (int i = 0; i < 6; ++i) mu[i] = x[i];
  }
}
__global__ void k_tukey_sigma(const double* __restrict__ med, double n, double override_sigma, double* __restrict__ sig, int est) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (override_sigma > 0) sig[0] = override_sigma;
    else sig[0] = mest_sigma_sq(est, (double)n, med[0]);
  }
}


// ---- SmallBlurryImage / Relocaliser ------------------------------------------------------------------------------
//   k_sbi_make     SmallBlurryImage::MakeFromKF + MakeJacs      src/SmallBlurryImage.cc:67-118
//   k_sbi_score    Relocaliser::ScoreKFs (ZMSSD per candidate)  src/Relocaliser.cc:93-121, SmallBlurryImage.cc:122-134
//   k_sbi_iterate  SmallBlurryImage::IteratePosRelToTarget      src/SmallBlurryImage.cc:139-245
constexpr int SBI_W = 40, SBI_H = 30, SBI_N = SBI_W*SBI_H;
struct SbiTables {               // cv::resize fixed-point taps and the Gaussian, built on the host once per handle / blur
  int xi[SBI_W], yi[SBI_H]; short xa[SBI_W], xb[SBI_W], ya[SBI_H], yb[SBI_H];
  float k[32]; int ks;
};

// one workgroup: resize (integer, as cv::resize 8U INTER_LINEAR), exact integer sum -> float mean, separable Gaussian in
// float with the taps accumulated centre-first then outwards (zero outside the image), central differences
__global__ void __launch_bounds__(256)
k_sbi_make(const uint8_t* __restrict__ img, int iw, int ih, SbiTables tb, uint8_t* __restrict__ small_out,
           float* __restrict__ templ_out, float* __restrict__ jacs_out) {
  __shared__ float A[SBI_N], B[SBI_N];
  __shared__ unsigned int red[4];
  const int t = threadIdx.x;
  unsigned int loc = 0;
  for (int i = t; i < SBI_N; i += 256) {
    const int y = i/SBI_W, x = i%SBI_W;
    const uint8_t* r0 = img + (size_t)tb.yi[y]*iw; const uint8_t* r1 = img + (size_t)min(tb.yi[y] + 1, ih - 1)*iw;
    const int x0 = tb.xi[x], x1 = min(x0 + 1, iw - 1);
    const int h0 = r0[x0]*tb.xa[x] + r0[x1]*tb.xb[x], h1 = r1[x0]*tb.xa[x] + r1[x1]*tb.xb[x];
    const int v = (((tb.ya[y]*(h0 >> 4)) >> 16) + ((tb.yb[y]*(h1 >> 4)) >> 16) + 2) >> 2;
    const uint8_t b = (uint8_t)v;
    small_out[i] = b; A[i] = (float)b; loc += b;
  }
  for (int o = 32; o > 0; o >>= 1) loc += __shfl_xor(loc, o, 64);
  if ((t & 63) == 0) red[t >> 6] = loc;
  __syncthreads();
  const unsigned int sum = red[0] + red[1] + red[2] + red[3];
  const float mean = ((float)sum)/SBI_N;
  for (int i = t; i < SBI_N; i += 256) A[i] = A[i] - mean;
  __syncthreads();
  for (int i = t; i < SBI_N; i += 256) {           // rows
    const int y = i/SBI_W, x = i%SBI_W;
    float a = A[i]*tb.k[0];
    for (int q = 1; q <= tb.ks; ++q) { float p = 0.f; if (x - q >= 0) p += A[y*SBI_W + x - q]; if (x + q < SBI_W) p += A[y*SBI_W + x + q]; a += p*tb.k[q]; }
    B[i] = a;
  }
  __syncthreads();
  for (int i = t; i < SBI_N; i += 256) {           // columns
    const int y = i/SBI_W, x = i%SBI_W;
    float a = B[i]*tb.k[0];
    for (int q = 1; q <= tb.ks; ++q) { float p = 0.f; if (y - q >= 0) p += B[(y - q)*SBI_W + x]; if (y + q < SBI_H) p += B[(y + q)*SBI_W + x]; a += p*tb.k[q]; }
    A[i] = a; templ_out[i] = a;
  }
  __syncthreads();
  for (int i = t; i < SBI_N; i += 256) {
    const int y = i/SBI_W, x = i%SBI_W;
    float gx = 0.f, gy = 0.f;
    if (x >= 1 && y >= 1 && x < SBI_W - 1 && y < SBI_H - 1) { gx = A[i + 1] - A[i - 1]; gy = A[i + SBI_W] - A[i - SBI_W]; }
    jacs_out[2*i] = gx; jacs_out[2*i + 1] = gy;
  }
}

// one thread per candidate keyframe, raster-order double accumulation exactly as the scalar loop (so scores, and with them
// the "first smallest" winner, are bit-identical); the 1200-float templates are a few kB each
__global__ void __launch_bounds__(64)
k_sbi_score(const float* __restrict__ cur, const float* const* __restrict__ cands, int n, double* __restrict__ scores) {
  const int i = blockIdx.x*64 + threadIdx.x;
  if (i >= n) return;
  const float* o = cands[i];
  if (!o) { scores[i] = 1.7976931348623157e308; return; }
  double ssd = 0.0;
  for (int p = 0; p < SBI_N; ++p) { const double d = cur[p] - o[p]; ssd += d*d; }
  scores[i] = ssd;
}

// ESM alignment, all iterations in one launch (single workgroup): warp by closed-form source positions, 15 partial sums per
// thread in double, fixed-order tree reduction, 4x4 LDL^T solve and SE2 update on thread 0
__global__ void __launch_bounds__(256)
k_sbi_iterate(const float* __restrict__ me, const float* __restrict__ ot_templ, const float* __restrict__ ot_jacs, int iterations,
              double* __restrict__ out /* se2[6], score */) {
  __shared__ float Tm[SBI_N], warped[SBI_N];
  __shared__ double X[6], red[256][15 + 1], st[8];
  const int t = threadIdx.x;
  const int cx = SBI_W/2, cy = SBI_H/2;
  for (int i = t; i < SBI_N; i += 256) Tm[i] = me[i];
  if (t == 0) { st[0] = 1; st[1] = 0; st[2] = 0; st[3] = 1; st[4] = 0; st[5] = 0; st[6] = 0; st[7] = 0; }   // CtoC, mean offset, score
  __syncthreads();
  for (int it = 0; it < iterations; ++it) {
    if (t == 0) {
      // se2WfromC * se2CtoC * se2WfromC^-1 with WfromC = (I, centre)
      const double R0 = st[0], R1 = st[1], R2 = st[2], R3 = st[3];
      const double tx = st[4] + cx, ty = st[5] + cy;                      // WfromC * CtoC
      X[0] = R0; X[1] = R1; X[2] = R2; X[3] = R3;
      X[4] = R0*(-(double)cx) + R1*(-(double)cy) + tx; X[5] = R2*(-(double)cx) + R3*(-(double)cy) + ty;
    }
    __syncthreads();
    const double xb = SBI_W - 1, yb = SBI_H - 1;
    for (int i = t; i < SBI_N; i += 256) {
      const int y = i/SBI_W, x = i%SBI_W;
      const double px = X[4] + X[0]*x + X[1]*y, py = X[5] + X[2]*x + X[3]*y;
      float v = -9e20f;
      if (0 <= px && 0 <= py && px < xb && py < yb) {
        const int lx = (int)px, ly = (int)py;
        const double fx = px - lx, fy = py - ly;
        const float* q = Tm + ly*SBI_W + lx;
        v = (float)((1 - fy)*((1 - fx)*q[0] + fx*q[1]) + fy*((1 - fx)*q[SBI_W] + fx*q[SBI_W + 1]));
      }
      warped[i] = v;
    }
    __syncthreads();
    double a[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) a[k] = 0.0;
    const double moff = st[6];
    for (int i = t; i < SBI_N; i += 256) {
      const int y = i/SBI_W, x = i%SBI_W;
      if (!(x >= 1 && y >= 1 && x < SBI_W - 1 && y < SBI_H - 1)) continue;
      const float l = warped[i - 1], r = warped[i + 1], u = warped[i - SBI_W], d = warped[i + SBI_W], here = warped[i];
      if (l + r + u + d + here < -9999.9) continue;
      const double g0 = r - l, g1 = d - u;
      const double s0 = 0.25*(g0 + (double)ot_jacs[2*i]), s1 = 0.25*(g1 + (double)ot_jacs[2*i + 1]);
      const double J0 = s0, J1 = s1, J2 = -(y - cy)*s0 + (x - cx)*s1;
      const double diff = (here - ot_templ[i]) + moff;
      a[14] += diff*diff;
      a[10] += diff*J0; a[11] += diff*J1; a[12] += diff*J2; a[13] += diff;
      a[0] += J0*J0; a[1] += J1*J0; a[2] += J1*J1; a[3] += J2*J0; a[4] += J2*J1; a[5] += J2*J2; a[6] += J0; a[7] += J1; a[8] += J2; a[9] += 1.0;
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) red[t][k] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) {
#pragma unroll
        for (int k = 0; k < 15; ++k) red[t][k] += red[t + o][k];
      }
      __syncthreads();
    }
    if (t == 0) {
      const double* tri = red[0];
      double A[16]; int v = 0;
      for (int j = 0; j < 4; ++j) for (int i = 0; i <= j; ++i) { A[4*j + i] = A[4*i + j] = tri[v++]; }
      const double b[4] = { red[0][10], red[0][11], red[0][12], red[0][13] };
      double L[16], D[4], yv[4], upd[4];
      for (int i = 0; i < 16; ++i) L[i] = 0.0;
      for (int j = 0; j < 4; ++j) {
        double dd = A[5*j];
        for (int k = 0; k < j; ++k) dd -= L[4*j + k]*L[4*j + k]*D[k];
        D[j] = dd;
        for (int i = j + 1; i < 4; ++i) { double w = A[4*i + j]; for (int k = 0; k < j; ++k) w -= L[4*i + k]*L[4*j + k]*D[k]; L[4*i + j] = w/dd; }
      }
      for (int i = 0; i < 4; ++i) { double w = b[i]; for (int k = 0; k < i; ++k) w -= L[4*i + k]*yv[k]; yv[i] = w; }
      for (int i = 0; i < 4; ++i) yv[i] /= D[i];
      for (int i = 3; i >= 0; --i) { double w = yv[i]; for (int k = i + 1; k < 4; ++k) w -= L[4*k + i]*upd[k]; upd[i] = w; }
      const double th = -upd[2], c = cos(th), s = sin(th);
      const double U[6] = { c, -s, s, c, -upd[0], -upd[1] };
      const double R0 = st[0], R1 = st[1], R2 = st[2], R3 = st[3];
      const double n4 = R0*U[4] + R1*U[5] + st[4], n5 = R2*U[4] + R3*U[5] + st[5];
      st[0] = R0*U[0] + R1*U[2]; st[1] = R0*U[1] + R1*U[3]; st[2] = R2*U[0] + R3*U[2]; st[3] = R2*U[1] + R3*U[3];
      st[4] = n4; st[5] = n5;
      st[6] -= upd[3];
      st[7] = red[0][14];
    }
    __syncthreads();
  }
  if (t < 6) out[t] = st[t];
  if (t == 6) out[6] = st[7];
}


// One Gauss-Newton step from the 27 sums of the weighted normal equations (21 of the lower triangle row by row, 6 of the vector):
// prior 100 on the diagonal, 6x6 Cholesky, mu, BaseFromWorld <- exp(mu) BaseFromWorld (Tracker::CalcPoseUpdate, src/Tracker.cc:1499-1512:
// TooN's Cholesky<6> backsub).  One lane: a pivot costs its reciprocal square root only (v_rsq_f64 + one third-order correction, 6 dependent
// instructions; sqrt followed by a division is ~50) -- results within an ulp or two of the sqrt / divide form.
__device__ inline double rsqrt_h3(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(y0*(-x), y0, 1.0);
  return __builtin_fma(y0*e, __builtin_fma(e, 0.375, 0.5), y0);
}
// exp of the step: A = sin t / t, B = (1 - cos t) / t^2, C = (1 - A) / t^2 are entire functions of t^2 -- below t^2 = 1/4 (a pose step of
// half a radian) their series to 2^-60 (no square root, no division, no sin / cos: ~25 fused multiply-adds where libm's pair costs ~150
// instructions on the one lane everything waits for); above, the closed forms of se3_exp (TooN SE3<>::exp, what the oracle evaluates);
// below 1e-6 TooN's truncated forms, as the oracle.
__device__ __forceinline__ void se3_exp_step(const double* mu, Se3& T) {
#pragma clang fp contract(fast)
  const double* w = mu + 3;
  const double x = w[0]*w[0] + w[1]*w[1] + w[2]*w[2];
  if (!(x < 0.25)) { se3_exp(mu, T); return; }
  const double cx = w[1]*mu[2] - w[2]*mu[1], cy = w[2]*mu[0] - w[0]*mu[2], cz = w[0]*mu[1] - w[1]*mu[0];
  double A, B;
  if (x < 1e-8) {                                  // (TooN's own truncations below 1e-8 and 1e-6: the reference's values, not better ones)
    A = 1.0 - x*(1.0/6.0); B = 0.5;
    T.t[0] = mu[0] + 0.5*cx; T.t[1] = mu[1] + 0.5*cy; T.t[2] = mu[2] + 0.5*cz;
  } else {
    double C;
    if (x < 1e-6) { C = (1.0/6.0)*(1.0 - (1.0/20.0)*x); A = 1.0 - x*C; B = 0.5 - 0.25*(1.0/6.0)*x; }
    else {
      // 1/n! for n = 1 .. 19, alternating by the index of the term
      A = 1.0 + x*(-1.0/6 + x*(1.0/120 + x*(-1.0/5040 + x*(1.0/362880 + x*(-1.0/39916800 + x*(1.0/6227020800.0 + x*(-1.0/1307674368000.0 + x*(1.0/355687428096000.0))))))));
      B = 0.5 + x*(-1.0/24 + x*(1.0/720 + x*(-1.0/40320 + x*(1.0/3628800 + x*(-1.0/479001600 + x*(1.0/87178291200.0 + x*(-1.0/20922789888000.0 + x*(1.0/6402373705728000.0))))))));
      C = 1.0/6 + x*(-1.0/120 + x*(1.0/5040 + x*(-1.0/362880 + x*(1.0/39916800 + x*(-1.0/6227020800.0 + x*(1.0/1307674368000.0 + x*(-1.0/355687428096000.0)))))));
    }
    const double dx = w[1]*cz - w[2]*cy, dy = w[2]*cx - w[0]*cz, dz = w[0]*cy - w[1]*cx;
    T.t[0] = mu[0] + B*cx + C*dx; T.t[1] = mu[1] + B*cy + C*dy; T.t[2] = mu[2] + B*cz + C*dz;
  }
  rodrigues(w, A, B, T.R);
}
__device__ __forceinline__ void pose_step_from_sums(const double* tot, double* pose, double* v6) {
#pragma clang fp contract(fast)                   // (this translation unit is built with -ffp-contract=off for the bit-exact image paths; the pose step is compared to rounding)
  double C[36], v[6], mu[6], rd[6];
  int q = 0;
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int y = 0; y <= x; ++y) { C[6*x + y] = tot[q]; ++q; }
#pragma unroll
  for (int x = 0; x < 6; ++x) { v[x] = tot[21 + x]; C[7*x] += 100.0; }    // add_prior(100)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = C[7*j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= C[6*j + k]*C[6*j + k];
    rd[j] = rsqrt_h3(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) { double sum = C[6*i + j]; for (int k = 0; k < j; ++k) sum -= C[6*i + k]*C[6*j + k]; C[6*i + j] = sum*rd[j]; }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { double sum = v[i]; for (int k = 0; k < i; ++k) sum -= C[6*i + k]*mu[k]; mu[i] = sum*rd[i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { double sum = mu[i]; for (int k = i + 1; k < 6; ++k) sum -= C[6*k + i]*mu[k]; mu[i] = sum*rd[i]; }
  Se3 E, T, R;
  se3_exp_step(mu, E);
#pragma unroll
  for (int k = 0; k < 9; ++k) T.R[k] = pose[k];
  T.t[0] = pose[9]; T.t[1] = pose[10]; T.t[2] = pose[11];
  se3_compose(E, T, R);
#pragma unroll
  for (int k = 0; k < 9; ++k) pose[k] = R.R[k];
  pose[9] = R.t[0]; pose[10] = R.t[1]; pose[11] = R.t[2];
#pragma unroll
  for (int k = 0; k < 6; ++k) v6[k] = mu[k];
}

// ---- the ten Gauss-Newton pose iterations of Tracker::TrackMap in one launch ----------------------------------------
//   src/Tracker.cc:775-838 (PoseUpdateStep / PoseUpdateStepLinear), 1038-1075 (schedule), 1386-1512 (CalcPoseUpdate)
// One workgroup; the points stay on the device between iterations (image position, camera derivatives, 2x6 Jacobian), so
// an iteration costs a few barriers instead of a host round trip:  re-project or linear update -> covariance-scaled
// errors -> exact Tukey median by an in-LDS radix select -> weights, 27 partial sums per thread, fixed-order reduction ->
// 6x6 Cholesky and exp(mu) on thread 0.
constexpr int PR_THREADS = 1024;
__global__ void __launch_bounds__(PR_THREADS)
k_pose_refine(int n, mcp_pose_point* __restrict__ pts, const mcp_camera* __restrict__ cams, const double* __restrict__ cfb_all,
              double* __restrict__ bfw_io, int n_iter, const uint8_t* __restrict__ nonlinear, const double* __restrict__ override_sigma,
              double* __restrict__ J, double* __restrict__ ex /* 2n */, double* __restrict__ e2s, double* __restrict__ mu_out, double* __restrict__ w_out, int est) {
  __shared__ unsigned int hist[2048];
  __shared__ unsigned long long sel_sc[1024/64 + 3], sel_st[2];
  __shared__ double red[PR_THREADS/64][28];
  __shared__ double pose[12], v6[6], tot[27];
  __shared__ int nf_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 12) pose[t] = bfw_io[t];
  if (t < 6) v6[t] = 0.0;
  __syncthreads();
  for (int it = 0; it < n_iter; ++it) {
    const bool nl = nonlinear[it] != 0;
    int nf_loc = 0;
    for (int i = t; i < n; i += PR_THREADS) {
      mcp_pose_point& p = pts[i];
      if (!p.found) continue;
      ++nf_loc;
      double* Ji = J + 12*(size_t)i;
      if (nl) {
        const double* cfb = cfb_all + 12*(size_t)p.cam;
        double xb[3], xc[3];
        mat3_vec(pose, p.world_pos, xb); xb[0] += pose[9]; xb[1] += pose[10]; xb[2] += pose[11];
        mat3_vec(cfb, xb, xc); xc[0] += cfb[9]; xc[1] += cfb[10]; xc[2] += cfb[11];
        if (it != 0) {
          Projection pr; cam_project<true>(cams[p.cam], xc, pr);
          p.image[0] = pr.u; p.image[1] = pr.v; p.cam_derivs[0] = pr.D[0]; p.cam_derivs[1] = pr.D[1]; p.cam_derivs[2] = pr.D[2]; p.cam_derivs[3] = pr.D[3];
        }
        double dT[3], dP[3]; cam_sphere_deriv(xc, dT, dP);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          double mb[3], mc[3]; generator(m, xb, mb); mat3_vec(cfb, mb, mc);
          const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
          Ji[m] = p.cam_derivs[0]*s0 + p.cam_derivs[1]*s1; Ji[6 + m] = p.cam_derivs[2]*s0 + p.cam_derivs[3]*s1;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) { double a = 0.0; for (int k = 0; k < 6; ++k) a += Ji[6*r + k]*v6[k]; p.image[r] += a; }
      }
      const double e0 = p.sqrt_inv_noise*(p.found_pos[0] - p.image[0]), e1 = p.sqrt_inv_noise*(p.found_pos[1] - p.image[1]);
      ex[2*(size_t)i] = e0; ex[2*(size_t)i + 1] = e1; e2s[i] = e0*e0 + e1*e1;
    }
    // number of found points (constant over the iterations, recomputed for simplicity)
    for (int o = 32; o > 0; o >>= 1) nf_loc += __shfl_xor(nf_loc, o, 64);
    if (t == 0) nf_s = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&nf_s, nf_loc);
    __syncthreads();
    const int nf = nf_s;
    if (nf == 0) { if (t < 6) { v6[t] = 0.0; } __syncthreads(); continue; }           // no valid measurements: null update
    double s2 = override_sigma[it];
    if (!(s2 > 0)) {
      // Tukey::FindSigmaSquared: exact [nf/2] order statistic of the squared errors, MSD radix select in LDS
      const unsigned long long sel_prefix = lds_radix_select_1024(n, (unsigned long long)(nf/2), 0, 0ull, [&](int i, unsigned long long& key) {
        if (!pts[i].found) return false;
        key = (unsigned long long)__double_as_longlong(fabs(e2s[i]));
        return true; }, hist, sel_sc, sel_st);
      const double med = __longlong_as_double((long long)sel_prefix);
      s2 = mest_sigma_sq(est, (double)nf, med);
    }
    // weighted normal equations: 21 + 6 partial sums per thread
    double a[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) a[k] = 0.0;
    const bool last = (it == n_iter - 1);
    for (int i = t; i < n; i += PR_THREADS) {
      const mcp_pose_point& p = pts[i];
      if (!p.found) { if (last && w_out) w_out[i] = 0.0; continue; }
      const double err2 = e2s[i];
      const double w = mest_weight(est, err2, s2);
      if (last && w_out) w_out[i] = w;
      if (w == 0.0) continue;
      const double* Ji = J + 12*(size_t)i;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double Jr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) Jr[k] = p.sqrt_inv_noise*Ji[6*r + k];
        const double m = ex[2*(size_t)i + r];
        int q = 0;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          a[21 + x] += w*m*Jr[x];
#pragma unroll
          for (int y = 0; y <= x; ++y) { a[q] += w*Jr[x]*Jr[y]; ++q; }
        }
      }
    }
    {
      double w32[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w32[k] = k < 27 ? a[k] : 0.0;
      int idx; const double total = wave_reduce_scatter32(w32, lane, idx);          // ba_device.h: 32 lane exchanges instead of 27 x 6
      if (!(lane & 1) && idx < 27) red[wave][idx] = total;
    }
    __syncthreads();
    if (t < 27) { double sum = 0.0; for (int wv = 0; wv < PR_THREADS/64; ++wv) sum += red[wv][t]; tot[t] = sum; }      // the wavefronts' partials in wavefront order
    __syncthreads();
    if (t == 0) {
      pose_step_from_sums(tot, pose, v6);
    }
    __syncthreads();
  }
  if (t < 12) bfw_io[t] = pose[t];
  if (t < 6) mu_out[t] = v6[t];
}


// ---- the same ten iterations over MANY workgroups (frames with thousands of tracked points: BASELINE c5, eight cameras) -------
// k_pose_refine is one workgroup: at 8000 points its ten iterations take 0.76 ms -- 70 % of the device time of a c5 frame.  Here
// every workgroup owns a slice of the points; what the iterations need from ALL points crosses the workgroups through small
// device-scope buffers and a counter barrier:
//   * Tukey / Cauchy / Huber sigma^2: the exact [nf/2] order statistic by most-significant-digit radix select -- per digit every
//     workgroup adds its LDS histogram to a global one (integer atomics at device scope), barrier, every workgroup scans the same
//     global histogram; after two digits the few candidates are gathered (atomic slot counter), barrier, and every workgroup
//     resolves the remaining digits from that list on its own (lds_radix_select_1024) -- three barriers, all workgroups end with
//     the same median, bit for bit;
//   * the 21 + 6 WLS sums: a partial per workgroup, barrier, every workgroup adds the partials in workgroup order and solves the
//     6 x 6 itself (identical inputs, identical arithmetic: identical poses everywhere, nothing to broadcast).
// Cross-workgroup data is written and read with 8-byte device-scope atomics (the per-XCD L2s are not coherent with each other,
// MI355X_MICROARCH.md "Workgroup dispatch ..."); the barrier is one monotonic counter: drain own stores, add, poll.  Every buffer
// is per (iteration, digit), zeroed by the host once per call: no reuse, no reset races.  A spin that exceeds its bound raises
// an error word that ends every later wait (the call then fails; nothing hangs).
#ifndef PRM_NT
#define PRM_NT 512       // threads per workgroup: 1024 (four wavefronts per SIMD, 128 registers, 216 B of scratch) 0.744 ms per c5 frame, 512 (two per
#endif                   // SIMD, no spills) 0.671, 256 0.706 -- each with 512 points per workgroup
constexpr int PRM_THREADS = PRM_NT, PRM_MAX_WG = 64, PRM_CAND_CAP = 4096, PRM_MAX_ITER = 16;
struct PrmScratch {           // device memory, zeroed before the launch
  unsigned int sync;          // barrier counter
  unsigned int err;           // a wait gave up
  unsigned int nf;            // found points over all workgroups
  unsigned int pad_;
  unsigned int cand_n[PRM_MAX_ITER];
  unsigned int hist[PRM_MAX_ITER][SEL_PASSES][SEL_BINS];
  double cand[PRM_MAX_ITER][PRM_CAND_CAP];
  double acc[PRM_MAX_ITER][PRM_MAX_WG][28];
};
constexpr int PRM_GATHER_MAX = 16384;      // squared errors of all points fit one workgroup's LDS (128 KB) up to here
__device__ inline unsigned int prm_ld(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double prm_ldd(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void prm_std(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all threads of all workgroups; `epoch` counts the barriers passed (uniform)
__device__ inline void prm_barrier(PrmScratch* G, unsigned int& epoch, unsigned int nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // EVERY thread: its own device-scope stores / atomics have completed ...
  __syncthreads();                                            // ... before thread 0 tells the other workgroups that this one has arrived
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&G->sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int want = (epoch + 1)*nwg;
    // bounded in wall-clock time (s_memrealtime, 100 MHz): a workgroup of this launch that is not resident yet -- the device is shared
    // with the map maker's kernels -- arrives within microseconds to a fraction of a millisecond; after 50 ms the call gives up and is
    // redone in one workgroup (img_api.hip).  (Round 4 counted 2^24 polls: seconds.)
    unsigned int spins = 0; long long t0 = 0;
    while (prm_ld(&G->sync) < want) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63) == 63) {
        if (prm_ld(&G->err)) break;
        if (!t0) t0 = wall_clock64(); else if (wall_clock64() - t0 > 5000000ll) { __hip_atomic_store(&G->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
  }
  ++epoch;
  __syncthreads();
}
__global__ void __launch_bounds__(PRM_THREADS)
k_pose_refine_multi(int n, mcp_pose_point* __restrict__ pts, const mcp_camera* __restrict__ cams, const double* __restrict__ cfb_all,
                    double* __restrict__ bfw_io, int n_iter, const uint8_t* __restrict__ nonlinear, const double* __restrict__ override_sigma,
                    double* __restrict__ J, double* __restrict__ ex /* 2n */, double* __restrict__ e2s, double* __restrict__ mu_out, double* __restrict__ w_out,
                    int est, PrmScratch* __restrict__ G, double* __restrict__ e2_all /* 2 x n, or null: per-digit global histograms */,
                    double* __restrict__ res_host /* BaseFromWorld | mu once more, for pinned host memory (no copy back), or null */) {
  extern __shared__ double sh_e2[];            // gather mode: everybody's squared errors
  __shared__ unsigned int hist[SEL_BINS];
  __shared__ unsigned long long sel_sc[1024/64 + 3], sel_st[2];
  __shared__ double red[PRM_THREADS/64][28];
  __shared__ double pose[12], v6[6], tot[27];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned int nwg = gridDim.x, wg = blockIdx.x;
  const int i0 = (int)((long long)n*wg/nwg), i1 = (int)((long long)n*(wg + 1)/nwg);
  unsigned int epoch = 0;
  if (t < 12) pose[t] = bfw_io[t];
  if (t < 6) v6[t] = 0.0;
  // found points over all slices (constant over the iterations)
  {
    int nf_loc = 0;
    for (int i = i0 + t; i < i1; i += PRM_THREADS) nf_loc += pts[i].found ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) nf_loc += __shfl_xor(nf_loc, o, 64);
    if (lane == 0 && nf_loc) __hip_atomic_fetch_add(&G->nf, (unsigned int)nf_loc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  prm_barrier(G, epoch, nwg);
  const int nf = (int)prm_ld(&G->nf);
  for (int it = 0; it < n_iter; ++it) {
    const bool nl = nonlinear[it] != 0;
    for (int i = i0 + t; i < i1; i += PRM_THREADS) {
      mcp_pose_point& p = pts[i];
      if (!p.found) continue;
      double* Ji = J + 12*(size_t)i;
      if (nl) {
        const double* cfb = cfb_all + 12*(size_t)p.cam;
        double xb[3], xc[3];
        mat3_vec(pose, p.world_pos, xb); xb[0] += pose[9]; xb[1] += pose[10]; xb[2] += pose[11];
        mat3_vec(cfb, xb, xc); xc[0] += cfb[9]; xc[1] += cfb[10]; xc[2] += cfb[11];
        if (it != 0) {
          Projection pr; cam_project<true>(cams[p.cam], xc, pr);
          p.image[0] = pr.u; p.image[1] = pr.v; p.cam_derivs[0] = pr.D[0]; p.cam_derivs[1] = pr.D[1]; p.cam_derivs[2] = pr.D[2]; p.cam_derivs[3] = pr.D[3];
        }
        double dT[3], dP[3]; cam_sphere_deriv(xc, dT, dP);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          double mb[3], mc[3]; generator(m, xb, mb); mat3_vec(cfb, mb, mc);
          const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
          Ji[m] = p.cam_derivs[0]*s0 + p.cam_derivs[1]*s1; Ji[6 + m] = p.cam_derivs[2]*s0 + p.cam_derivs[3]*s1;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) { double a = 0.0; for (int k = 0; k < 6; ++k) a += Ji[6*r + k]*v6[k]; p.image[r] += a; }
      }
      const double e0 = p.sqrt_inv_noise*(p.found_pos[0] - p.image[0]), e1 = p.sqrt_inv_noise*(p.found_pos[1] - p.image[1]);
      ex[2*(size_t)i] = e0; ex[2*(size_t)i + 1] = e1; e2s[i] = e0*e0 + e1*e1;
    }
    __syncthreads();
    if (nf == 0) { if (t < 6) v6[t] = 0.0; __syncthreads(); continue; }           // no valid measurements: null update (the same decision everywhere)
    double s2 = override_sigma[it];
    if (!(s2 > 0) && e2_all) {
      // Few enough points for one LDS: every workgroup publishes its slice's squared errors (-1 = not found), ONE barrier, every
      // workgroup copies all of them and selects the [nf/2] order statistic on its own -- the same key set everywhere, the same
      // median.  The buffer alternates by iteration: its next writer is two barriers behind its last reader.
      double* eb = e2_all + (size_t)(it & 1)*n;
      for (int i = i0 + t; i < i1; i += PRM_THREADS) prm_std(eb + i, pts[i].found ? e2s[i] : -1.0);
      prm_barrier(G, epoch, nwg);
      for (int i = t; i < n; i += PRM_THREADS) sh_e2[i] = prm_ldd(eb + i);
      __syncthreads();
      const unsigned long long sel = lds_radix_select<PRM_THREADS>(n, (unsigned long long)(nf/2), 0, 0ull, [&](int i, unsigned long long& key) {
        const double v = sh_e2[i];
        if (v < 0.0) return false;
        key = (unsigned long long)__double_as_longlong(v);
        return true; }, hist, sel_sc, sel_st);
      s2 = mest_sigma_sq(est, (double)nf, __longlong_as_double((long long)sel));
    } else if (!(s2 > 0)) {
      // exact [nf/2] order statistic of the squared errors of ALL workgroups
      unsigned long long prefix = 0, kk = (unsigned long long)(nf/2);
      bool done = false;
      for (int pass = 0; pass < SEL_PASSES && !done; ++pass) {
        const int sh = sel_shift(pass);
        const unsigned int dmask = (1u << sel_nbits(pass)) - 1u;
        const unsigned long long himask = (pass == 0) ? 0ull : (~0ull << sel_shift(pass - 1));
        for (int b = t; b < SEL_BINS; b += PRM_THREADS) hist[b] = 0u;
        __syncthreads();
        for (int i = i0 + t; i < i1; i += PRM_THREADS) {
          if (!pts[i].found) continue;
          const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(e2s[i]));
          if ((key & himask) == prefix) atomicAdd(&hist[(unsigned int)(key >> sh) & dmask], 1u);
        }
        __syncthreads();
        unsigned int* gh = G->hist[it][pass];
        for (int b = t; b < SEL_BINS; b += PRM_THREADS) if (hist[b]) __hip_atomic_fetch_add(gh + b, hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prm_barrier(G, epoch, nwg);
        for (int b = t; b < SEL_BINS; b += PRM_THREADS) hist[b] = prm_ld(gh + b);
        __syncthreads();
        int bin; unsigned long long kin; unsigned int in_bin;
        lds_find_bin<PRM_THREADS>(hist, kk, bin, kin, in_bin, sel_sc);
        prefix |= (unsigned long long)bin << sh; kk = kin;
        if (sh == 0) { done = true; break; }
        if (in_bin <= (unsigned int)PRM_CAND_CAP) {
          // few enough keys share the prefix: gather them, every workgroup finishes on its own
          const unsigned long long hm2 = ~0ull << sh;
          for (int i = i0 + t; i < i1; i += PRM_THREADS) {
            if (!pts[i].found) continue;
            const double a = fabs(e2s[i]);
            if (((unsigned long long)__double_as_longlong(a) & hm2) == prefix) {
              const unsigned int slot = __hip_atomic_fetch_add(&G->cand_n[it], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (slot < (unsigned int)PRM_CAND_CAP) prm_std(&G->cand[it][slot], a);
            }
          }
          prm_barrier(G, epoch, nwg);
          const int m = (int)min(prm_ld(&G->cand_n[it]), (unsigned int)PRM_CAND_CAP);
          const double* cd = G->cand[it];
          prefix = lds_radix_select<PRM_THREADS>(m, kk, pass + 1, prefix, [&](int i, unsigned long long& key) {
            key = (unsigned long long)__double_as_longlong(prm_ldd(cd + i));
            return true; }, hist, sel_sc, sel_st);
          done = true;
        }
      }
      const double med = __longlong_as_double((long long)prefix);
      s2 = mest_sigma_sq(est, (double)nf, med);
    }
    // this workgroup's part of the weighted normal equations
    double a[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) a[k] = 0.0;
    const bool last = (it == n_iter - 1);
    for (int i = i0 + t; i < i1; i += PRM_THREADS) {
      const mcp_pose_point& p = pts[i];
      if (!p.found) { if (last && w_out) w_out[i] = 0.0; continue; }
      const double err2 = e2s[i];
      const double w = mest_weight(est, err2, s2);
      if (last && w_out) w_out[i] = w;
      if (w == 0.0) continue;
      const double* Ji = J + 12*(size_t)i;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double Jr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) Jr[k] = p.sqrt_inv_noise*Ji[6*r + k];
        const double m = ex[2*(size_t)i + r];
        int q = 0;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          a[21 + x] += w*m*Jr[x];
#pragma unroll
          for (int y = 0; y <= x; ++y) { a[q] += w*Jr[x]*Jr[y]; ++q; }
        }
      }
    }
    {
      double w32[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w32[k] = k < 27 ? a[k] : 0.0;
      int idx; const double total = wave_reduce_scatter32(w32, lane, idx);
      if (!(lane & 1) && idx < 27) red[wave][idx] = total;
    }
    __syncthreads();
    if (t < 27) { double sum = 0.0; for (int wv = 0; wv < PRM_THREADS/64; ++wv) sum += red[wv][t]; prm_std(&G->acc[it][wg][t], sum); }
    prm_barrier(G, epoch, nwg);
    if (t < 27) { double sum = 0.0; for (unsigned int g = 0; g < nwg; ++g) sum += prm_ldd(&G->acc[it][g][t]); tot[t] = sum; }      // workgroup order: the same sum everywhere
    __syncthreads();
    if (t == 0) {
      pose_step_from_sums(tot, pose, v6);
    }
    __syncthreads();
  }
  if (wg == 0) {
    if (t < 12) bfw_io[t] = pose[t];
    if (t < 6) mu_out[t] = v6[t];
    if (res_host) { if (t < 12) res_host[t] = pose[t]; if (t < 6) res_host[12 + t] = v6[t]; }
  }
}


// ---- the same ten iterations with the points held in registers (n <= PRR_THREADS*PRR_PPT, ncam <= PRR_CAMS) ----------------
// One workgroup of 512 threads = two wavefronts per SIMD at up to 256 registers each.  The FOUND points are compacted into the slots
// (thread t: slots t and t + 512): a thread's points' found position, noise, image position and errors live in registers across all
// iterations, their 2x6 Jacobians in LDS (96 KB), and what the re-projecting iterations read besides -- world positions, camera models,
// CamFromBase, the arctangent's table -- in LDS too (29 KB): no global reads inside an iteration.  Eight-wavefront barriers; the Tukey
// median is selected from the register-held squared errors by votes (regs_vote_select, ba_select.h), the 6x6 step is solved on one lane
// with reciprocal-square-root pivots and the exponential's series (pose_step_from_sums).  Same arithmetic per point as k_pose_refine
// up to fused multiply-adds; the order of the 27 sums differs (compacted slots, PRR_THREADS/64 wavefronts), i.e. results agree to
// rounding (tests: 1e-10 on the pose), the median exactly (tests/cpp/vote_select_check.hip, the hard populations of tests/test_img_gpu.py).
// Ten iterations at c3 (1000 points, 696 found): 94 -> 64 us in round 5.  Stamps (-DMCP_PRR_PROF, scripts/prr_prof.sh; cycles): entry to
// the first iteration 12 k (two dependent trips to cold global memory, ~5 k each for one compute unit), a re-projecting pass 16 k (three
// wavefront passes of ~1100 instructions per point on the busiest SIMD: the camera model with its correctly rounded arctangent),
// a linear pass 1.8 k, the median 4.6 k (one counted step of eighths of a binade + <= 64 keys ranked) to 6.5 k (binade steps),
// the 27 sums 2.0 k + their reduction 1.2-1.9 k, the step 2.0 k.
#ifndef PRR_NT
#define PRR_NT 512      // threads; 1024/PRR_NT points each.  256 x 4: 125.8 us per ten iterations at c3 (one wavefront per SIMD, 512 registers), 512 x 2: 102.4
#endif                  // (two wavefronts per SIMD at 255 registers overlap each other's latencies), 1024 x 1: 176.7 (eight-wavefront barriers, spills)
constexpr int PRR_THREADS = PRR_NT, PRR_PPT = 1024/PRR_NT;
#ifdef MCP_PRR_PROF
__device__ unsigned long long g_prr_prof[16*8 + 8];
#define PRR_STAMP(i) do { if (threadIdx.x == 0 && it < 16) g_prr_prof[it*8 + (i)] = clock64(); } while (0)
#define PRR_STAMP_K(i) do { if (threadIdx.x == 0) g_prr_prof[16*8 + (i)] = clock64(); } while (0)      // kernel entry, first iteration, after the last, exit
#else
#define PRR_STAMP(i) do {} while (0)
#define PRR_STAMP_K(i) do {} while (0)
#endif
constexpr int PRR_CAMS = 8;           // cameras whose models and CamFromBase poses are staged in LDS (a c5 rig has 8); more: the plain kernels
// dynamic LDS of k_pose_refine_regs: the Jacobians [12][1024], the world positions [3][1024], the arctangent's table, CamFromBase, camera models
constexpr size_t PRR_DYN_LDS = sizeof(double)*((12 + 3)*(size_t)PRR_THREADS*PRR_PPT + 130 + 12*PRR_CAMS) + sizeof(mcp_camera)*PRR_CAMS;
__global__ void __launch_bounds__(PRR_THREADS)
k_pose_refine_regs(int n, mcp_pose_point* __restrict__ pts, const mcp_camera* __restrict__ cams, const double* __restrict__ cfb_all,
                   double* __restrict__ bfw_io, int n_iter, const uint8_t* __restrict__ nonlinear, const double* __restrict__ override_sigma,
                   double* __restrict__ mu_out, double* __restrict__ w_out, int est, int ncam) {
#pragma clang fp contract(fast)                   // (the sums and the linear updates; the camera model keeps its own uncontracted arithmetic)
  constexpr int NT = PRR_THREADS, NW = NT/64, BPT = SEL_BINS/NT;
  PRR_STAMP_K(0);
  __shared__ unsigned int hist[SEL_BINS];
  __shared__ unsigned int wtot[NW], sres[3];
  __shared__ unsigned long long skey;
  __shared__ unsigned int vt[2][NW][SELV_NB + 1];
  __shared__ unsigned long long cand[SELV_MAXC], wsamp[NW];
  __shared__ unsigned int rk[SELV_MAXC];
  __shared__ double red[NW][28];
  __shared__ double pose[12], v6[6], tot[27];
  __shared__ int wcnt[PRR_PPT][NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  extern __shared__ double Jl[];                 // [12][NT*PPT]: a point's 2x6 Jacobian, one column of 12 per point (lane-consecutive: no bank conflicts)
  constexpr int JS = PRR_THREADS*PRR_PPT;
  // what the re-projecting iterations read, staged once:
  double* const wpl = Jl + 12*JS;                // [3][JS] world positions
  double* const satan = wpl + 3*JS;              // the arctangent's table (atan_cr.h): hi[65], lo[65]
  double* const cfbl = satan + 130;              // [ncam][12] CamFromBase
  mcp_camera* const caml = reinterpret_cast<mcp_camera*>(cfbl + 12*PRR_CAMS);
  __shared__ double sov[32];
  __shared__ unsigned char snl[32];
  // The FOUND points, compacted: a frame finds 60-80 % of its points, and a slot (t + k NT) no lane of a wavefront fills costs that
  // wavefront nothing in any pass -- 700 found points are 11 wavefront passes where 1000 slots are 16.  Order: slot-major, as given.
  int* const perm = reinterpret_cast<int*>(hist);        // (the histogram's 8 KB, before they are zeroed)
  bool fnd[PRR_PPT]; int src[PRR_PPT];
#pragma unroll
  for (int k = 0; k < PRR_PPT; ++k) { const int i = t + k*NT; fnd[k] = i < n && pts[i].found != 0; }
  for (int i = t; i < 65; i += NT) { satan[i] = mcp_atan::kAtanHi[i]; satan[65 + i] = mcp_atan::kAtanLo[i]; }
  for (int i = t; i < 12*ncam; i += NT) cfbl[i] = cfb_all[i];
  { const double* srcd = reinterpret_cast<const double*>(cams); double* dst = reinterpret_cast<double*>(caml);
    for (int i = t; i < (int)(sizeof(mcp_camera)/sizeof(double))*ncam; i += NT) dst[i] = srcd[i]; }
  if (t < 32 && t < n_iter) { sov[t] = override_sigma[t]; snl[t] = nonlinear[t]; }
  if (t < 12) pose[t] = bfw_io[t];
  if (t < 6) v6[t] = 0.0;
  unsigned long long fm[PRR_PPT];
#pragma unroll
  for (int k = 0; k < PRR_PPT; ++k) { fm[k] = __ballot(fnd[k]); if (lane == 0) wcnt[k][wave] = __popcll(fm[k]); }
  __syncthreads();
  int nf = 0;
#pragma unroll
  for (int k = 0; k < PRR_PPT; ++k) {
    int base = nf;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = wcnt[k][w]; if (w < wave) base += c; nf += c; }
    if (fnd[k]) perm[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(fm[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)fm[k], 0u))] = t + k*NT;
  }
  __syncthreads();
  int camk[PRR_PPT]; bool has[PRR_PPT];
  double fpos[PRR_PPT][2], sinv[PRR_PPT], img[PRR_PPT][2], ex[PRR_PPT][2];      // (the camera derivatives live only while a point's Jacobian is
                                                                                // built: read from / written to the record in those iterations)
#pragma unroll
  for (int k = 0; k < PRR_PPT; ++k) {
    const int col = t + k*NT;
    fnd[k] = col < nf; src[k] = fnd[k] ? perm[col] : 0; camk[k] = 0; sinv[k] = 0.0;
    has[k] = wave*64 + k*NT < nf;                // (wavefront-uniform: its first lane's slot is filled)
#pragma unroll
    for (int q = 0; q < 2; ++q) { fpos[k][q] = 0.0; img[k][q] = 0.0; ex[k][q] = 0.0; }
    if (fnd[k]) {
      const mcp_pose_point& p = pts[src[k]];
      camk[k] = p.cam; sinv[k] = p.sqrt_inv_noise;
      fpos[k][0] = p.found_pos[0]; fpos[k][1] = p.found_pos[1]; img[k][0] = p.image[0]; img[k][1] = p.image[1];
#pragma unroll
      for (int q = 0; q < 3; ++q) wpl[q*JS + col] = p.world_pos[q];
    }
  }
  __syncthreads();                               // (perm is read: the histogram may be zeroed)
  for (int b = t; b < SEL_BINS; b += NT) hist[b] = 0u;          // every digit pass leaves it zero again
  unsigned long long prev_key = 0ull, prev_dk = ~0ull; bool have_prev = false;      // the last median and how far it had moved
  __syncthreads();
  for (int it = 0; it < n_iter; ++it) {
    if (nf == 0) { if (t < 6) v6[t] = 0.0; __syncthreads(); continue; }           // no valid measurements: null update
    const bool nl = (it < 32 ? snl[it] : nonlinear[it]) != 0;
    if (it == 0) PRR_STAMP_K(1);
    PRR_STAMP(0);
#pragma unroll
    for (int k = 0; k < PRR_PPT; ++k) {
      if (!fnd[k]) continue;
      const int i = t + k*NT;
      if (nl) {
        const double* cfb = cfbl + 12*camk[k];
        const double wp[3] = { wpl[i], wpl[JS + i], wpl[2*JS + i] };
        double xb[3], xc[3];
        mat3_vec(pose, wp, xb); xb[0] += pose[9]; xb[1] += pose[10]; xb[2] += pose[11];
        mat3_vec(cfb, xb, xc); xc[0] += cfb[9]; xc[1] += cfb[10]; xc[2] += cfb[11];
        double cd[4];
        mcp_pose_point& rec = pts[src[k]];
        if (it != 0) {
          Projection pr; cam_project<true>(caml[camk[k]], xc, pr, satan, satan + 65);
          img[k][0] = pr.u; img[k][1] = pr.v;
#pragma unroll
          for (int q = 0; q < 4; ++q) { cd[q] = pr.D[q]; rec.cam_derivs[q] = pr.D[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) cd[q] = rec.cam_derivs[q];       // the search stage's derivatives
        }
        double dT[3], dP[3]; cam_sphere_deriv(xc, dT, dP);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          double mb[3], mc[3]; generator(m, xb, mb); mat3_vec(cfb, mb, mc);
          const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
          Jl[m*JS + i] = cd[0]*s0 + cd[1]*s1; Jl[(6 + m)*JS + i] = cd[2]*s0 + cd[3]*s1;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) { double a = 0.0; for (int q = 0; q < 6; ++q) a += Jl[(6*r + q)*JS + i]*v6[q]; img[k][r] += a; }
      }
      ex[k][0] = sinv[k]*(fpos[k][0] - img[k][0]); ex[k][1] = sinv[k]*(fpos[k][1] - img[k][1]);
    }
    PRR_STAMP(1);
    double s2 = it < 32 ? sov[it] : override_sigma[it];
    if (!(s2 > 0)) {
      // Tukey::FindSigmaSquared: exact [nf/2] order statistic of the squared errors, MSD radix select over the register-held keys.
      // Three barriers per digit: the histogram's owner threads read and clear their bins in one go (it is zero again for the
      // next digit), counts are scanned in 32 bits, prefix and rank travel in registers.
      unsigned long long key[PRR_PPT];
#pragma unroll
      for (int k = 0; k < PRR_PPT; ++k) key[k] = (unsigned long long)__double_as_longlong(fabs(ex[k][0]*ex[k][0] + ex[k][1]*ex[k][1]));
      unsigned long long prefix = 0ull; unsigned int kk = (unsigned int)(nf/2);
      // By votes first (regs_vote_select, ba_select.h), in brackets of key space around a guess: once the median has moved by less than
      // half a binade between two iterations, 16 eighths of a binade around the last one (usually <= 64 keys share the median's
      // eighth: ONE counted step, then they are ranked in LDS); else 16 binades -- 11 below the last median and 4 above (the errors
      // mostly shrink), in the first iteration 8 below and 7 above the first found point's error -- and 4 more bits per step.  One
      // barrier per step and no shared counters; a histogram pass is LDS atomics on a few hot counters, an owner scan and three
      // barriers.  A bracket that misses the rank: the next wider one, at last the histogram passes below from digit 0.
      bool have = false;
#ifndef PRR_NO_VOTE
      {
        unsigned long long r = 0ull;
        if (have_prev && prev_dk < (1ull << 51)) {
          const unsigned long long half = 8ull << 49;
          have = regs_vote_select<NT, PRR_PPT>(key, fnd, has, kk, prev_key > half ? prev_key - half : 0ull, 49, vt, cand, rk, r);
        }
        if (!have) {
          int e_lo;
          if (have_prev) e_lo = (int)(prev_key >> 52) - 11;
          else {
            // (slot 0 of wavefront 0 holds the first found point: the points are compacted)
            e_lo = (int)((unsigned int)__builtin_amdgcn_readfirstlane((int)(key[0] >> 32)) >> 20);
            if (lane == 0) wsamp[wave] = (unsigned long long)e_lo;
            __syncthreads();
            e_lo = (int)wsamp[0] - 8;
          }
          have = regs_vote_select<NT, PRR_PPT>(key, fnd, has, kk, (unsigned long long)(e_lo < 0 ? 0 : e_lo) << 52, 52, vt, cand, rk, r);
        }
        if (have) prefix = r;
      }
#endif
      const int pass0 = have ? SEL_PASSES : 0;
      for (int pass = pass0; pass < SEL_PASSES; ++pass) {
        const int sh = sel_shift(pass);
        const unsigned int dmask = (1u << sel_nbits(pass)) - 1u;
        const unsigned long long himask = (pass == 0) ? 0ull : (~0ull << sel_shift(pass - 1));
#pragma unroll
        for (int k = 0; k < PRR_PPT; ++k) if (fnd[k] && (key[k] & himask) == prefix) atomicAdd(&hist[(unsigned int)(key[k] >> sh) & dmask], 1u);
        __syncthreads();
        unsigned int hb[BPT], loc = 0;
#pragma unroll
        for (int b = 0; b < BPT; ++b) { hb[b] = hist[BPT*t + b]; hist[BPT*t + b] = 0u; loc += hb[b]; }
        unsigned int inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned int base = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) if (w < wave) base += wtot[w];
        const unsigned int incl = base + inc, excl = incl - loc;
        if (excl <= kk && kk < incl) {
          unsigned int acc = excl; int b = 0;
#pragma unroll
          for (int q = 0; q < BPT - 1; ++q) if (b == q && !(acc + hb[q] > kk)) { acc += hb[q]; b = q + 1; }
          unsigned int inb = hb[0];
#pragma unroll
          for (int q = 1; q < BPT; ++q) if (b == q) inb = hb[q];
          sres[0] = (unsigned int)(BPT*t + b); sres[1] = kk - acc; sres[2] = inb;
        }
        __syncthreads();
        prefix |= (unsigned long long)sres[0] << sh; kk = sres[1];
        if (sres[2] == 1u && sh > 0) {                 // one candidate left: it is the element
          const unsigned long long hm2 = ~0ull << sh;
#pragma unroll
          for (int k = 0; k < PRR_PPT; ++k) if (fnd[k] && (key[k] & hm2) == prefix) skey = key[k];
          __syncthreads();
          prefix = skey;
          break;
        }
      }
      const double med = __longlong_as_double((long long)prefix);
      prev_dk = have_prev ? (prefix > prev_key ? prefix - prev_key : prev_key - prefix) : ~0ull;
      prev_key = prefix; have_prev = true;
      s2 = mest_sigma_sq(est, (double)nf, med);
    }
    PRR_STAMP(2);
    // weighted normal equations: 21 + 6 partial sums per thread
    double a[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) a[q] = 0.0;
    const bool last = (it == n_iter - 1);
#pragma unroll
    for (int k = 0; k < PRR_PPT; ++k) {
      const int i = t + k*NT;
      if (!fnd[k]) continue;                       // (the weights of the points not found: zeroed by the host before the launch)
      const double w = mest_weight(est, ex[k][0]*ex[k][0] + ex[k][1]*ex[k][1], s2);
      if (last && w_out) w_out[src[k]] = w;
      if (w == 0.0) continue;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double Jr[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) Jr[q] = sinv[k]*Jl[(6*r + q)*JS + i];
        const double m = ex[k][r];
        int q = 0;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          a[21 + x] += w*m*Jr[x];
#pragma unroll
          for (int y = 0; y <= x; ++y) { a[q] += w*Jr[x]*Jr[y]; ++q; }
        }
      }
    }
    PRR_STAMP(6);
    {
      // the 27 sums over the wavefront as a reduce-scatter: at every butterfly step a lane hands half of its entries to its partner
      // and adds the partner's half of the entries it keeps -- 16 + 8 + 4 + 2 + 1 + 1 exchanges instead of 27 x 6, in a fixed tree
      double w32[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) w32[q] = q < 27 ? a[q] : 0.0;
      int idx; const double total = wave_reduce_scatter32(w32, lane, idx);
      if (!(lane & 1) && idx < 27) red[wave][idx] = total;
    }
    __syncthreads();
    PRR_STAMP(7);
    if (t < 27) { double sum = 0.0; for (int wv = 0; wv < NW; ++wv) sum += red[wv][t]; tot[t] = sum; }
    // (the totals go from lanes 0..26 to lane 0 of the SAME wavefront: its LDS operations complete in order, no workgroup barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PRR_STAMP(3);
    if (t == 0) {
      pose_step_from_sums(tot, pose, v6);
    }
    PRR_STAMP(4);
    __syncthreads();
    PRR_STAMP(5);
  }
  PRR_STAMP_K(2);
#pragma unroll
  for (int k = 0; k < PRR_PPT; ++k) {
    if (fnd[k]) {
      mcp_pose_point& p = pts[src[k]];
      p.image[0] = img[k][0]; p.image[1] = img[k][1];
    }
  }
  if (t < 12) bfw_io[t] = pose[t];
  if (t < 6) mu_out[t] = v6[t];
  PRR_STAMP_K(3);
}


// ---- camera-per-rank pose refinement (BASELINE config c5): every rank holds the found points of its own
// camera(s), the base pose is replicated.  One Gauss-Newton iteration of Tracker::TrackMap (src/Tracker.cc:1063-1075) =
//   k_pr_project  : PoseUpdateStep / PoseUpdateStepLinear + the covariance-scaled errors; the rank's squared errors go, compacted,
//                   into its own slot of a zero-filled (ranks x cap) table
//   all-reduce #1 : SUM of the table = gather of every rank's squared errors (+ the per-rank counts)
//   k_pr_accum    : Tukey sigma^2 from the exact [n/2] order statistic of ALL errors, then this rank's part of the WLS<6>
//                   accumulator: 21 entries of C^-1, 6 of the vector
//   all-reduce #2 : SUM of the 27 numbers
//   k_pr_solve    : prior 100, 6x6 Cholesky, mu, BaseFromWorld <- exp(mu) BaseFromWorld -- the same on every rank
__global__ void __launch_bounds__(256)
k_pr_project(int n, mcp_pose_point* __restrict__ pts, const mcp_camera* __restrict__ cams, const double* __restrict__ cfb_all,
             const double* __restrict__ pose, const double* __restrict__ v6, int it, int nl, double* __restrict__ J,
             double* __restrict__ ex, double* __restrict__ e2s, double* __restrict__ slot /* cap */, double* __restrict__ count_slot,
             unsigned int* __restrict__ counter) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  mcp_pose_point& p = pts[i];
  if (!p.found) return;
  double* Ji = J + 12*(size_t)i;
  if (nl) {
    const double* cfb = cfb_all + 12*(size_t)p.cam;
    double xb[3], xc[3];
    mat3_vec(pose, p.world_pos, xb); xb[0] += pose[9]; xb[1] += pose[10]; xb[2] += pose[11];
    mat3_vec(cfb, xb, xc); xc[0] += cfb[9]; xc[1] += cfb[10]; xc[2] += cfb[11];
    if (it != 0) {
      Projection pr; cam_project<true>(cams[p.cam], xc, pr);
      p.image[0] = pr.u; p.image[1] = pr.v; p.cam_derivs[0] = pr.D[0]; p.cam_derivs[1] = pr.D[1]; p.cam_derivs[2] = pr.D[2]; p.cam_derivs[3] = pr.D[3];
    }
    double dT[3], dP[3]; cam_sphere_deriv(xc, dT, dP);
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      double mb[3], mc[3]; generator(m, xb, mb); mat3_vec(cfb, mb, mc);
      const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
      Ji[m] = p.cam_derivs[0]*s0 + p.cam_derivs[1]*s1; Ji[6 + m] = p.cam_derivs[2]*s0 + p.cam_derivs[3]*s1;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 2; ++r) { double a = 0.0; for (int k = 0; k < 6; ++k) a += Ji[6*r + k]*v6[k]; p.image[r] += a; }
  }
  const double e0 = p.sqrt_inv_noise*(p.found_pos[0] - p.image[0]), e1 = p.sqrt_inv_noise*(p.found_pos[1] - p.image[1]);
  ex[2*(size_t)i] = e0; ex[2*(size_t)i + 1] = e1;
  const double e2 = e0*e0 + e1*e1;
  e2s[i] = e2;
  const unsigned int k = atomicAdd(counter, 1u);      // position inside the slot is irrelevant to an order statistic
  slot[k] = e2;
  atomicAdd(count_slot, 1.0);                          // (integer-valued: exact in any order)
}

__global__ void __launch_bounds__(1024)
k_pr_accum(int n, const mcp_pose_point* __restrict__ pts, const double* __restrict__ J, const double* __restrict__ ex, const double* __restrict__ e2s,
           const double* __restrict__ table /* world x cap */, const double* __restrict__ counts /* world */, int world, int cap,
           double override_sigma, int last, double* __restrict__ out27 /* [27] + [27] = total found */, double* __restrict__ w_out, int est) {
  __shared__ unsigned int hist[2048];
  __shared__ unsigned long long sel_sc[1024/64 + 3], sel_st[2];
  __shared__ double red[16][28];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double nf_d = 0.0;
  for (int r = 0; r < world; ++r) nf_d += counts[r];
  const unsigned long long nf = (unsigned long long)nf_d;
  if (nf == 0) { if (t < 28) out27[t] = 0.0; return; }
  double s2 = override_sigma;
  if (!(s2 > 0)) {
    const int m = world*cap;
    const unsigned long long sel_prefix = lds_radix_select_1024(m, nf/2, 0, 0ull, [&](int i, unsigned long long& key) {
      if ((double)(i % cap) >= counts[i/cap]) return false;
      key = (unsigned long long)__double_as_longlong(fabs(table[i]));
      return true; }, hist, sel_sc, sel_st);
    const double med = __longlong_as_double((long long)sel_prefix);
    s2 = mest_sigma_sq(est, (double)nf, med);
  }
  double a[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) a[k] = 0.0;
  for (int i = t; i < n; i += 1024) {
    const mcp_pose_point& p = pts[i];
    if (!p.found) { if (last && w_out) w_out[i] = 0.0; continue; }
    const double err2 = e2s[i];
    const double w = mest_weight(est, err2, s2);
    if (last && w_out) w_out[i] = w;
    if (w == 0.0) continue;
    const double* Ji = J + 12*(size_t)i;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double Jr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Jr[k] = p.sqrt_inv_noise*Ji[6*r + k];
      const double mm = ex[2*(size_t)i + r];
      int q = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        a[21 + x] += w*mm*Jr[x];
#pragma unroll
        for (int y = 0; y <= x; ++y) { a[q] += w*Jr[x]*Jr[y]; ++q; }
      }
    }
  }
  {
    double w32[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w32[k] = k < 27 ? a[k] : 0.0;
    int idx; const double total = wave_reduce_scatter32(w32, lane, idx);
    if (!(lane & 1) && idx < 27) red[wave][idx] = total;
  }
  __syncthreads();
  if (t < 27) { double sum = 0.0; for (int wv = 0; wv < 16; ++wv) sum += red[wv][t]; out27[t] = sum; }
  if (t == 27) out27[27] = 0.0;
}

// every rank: C^-1 (+ prior 100) mu = v, BaseFromWorld <- exp(mu) BaseFromWorld.  acc = the all-reduced 27 sums; nf_total = 0: null update.
__global__ void k_pr_solve(const double* __restrict__ acc, const double* __restrict__ counts, int world, double* __restrict__ pose, double* __restrict__ v6) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double nf = 0.0;
  for (int r = 0; r < world; ++r) nf += counts[r];
  if (nf == 0.0) { for (int k = 0; k < 6; ++k) v6[k] = 0.0; return; }
  pose_step_from_sums(acc, pose, v6);
}

}  // namespace mcp
