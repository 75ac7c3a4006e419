// img_api.hip -- C ABI of the KeyFrame / Tracker image path (include/mcp_img.h); host side only
// marshals buffers and launches the kernels of img_kernels.h.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <limits>
#include <map>
#include <memory>
#include <cmath>

#include "../../include/mcp_img.h"
#include "img_kernels.h"
#include "ba_select.h"

using namespace mcp;

extern void mcp_set_error(const char* s);     // ba_solver.hip
static int img_fail(const std::string& s) { mcp_set_error(s.c_str()); return -1; }
#define ICK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return img_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

namespace {
template <class T> struct Buf {
  T* p = nullptr; size_t n = 0;
  ~Buf() { if (p) (void)hipFree(p); }
  void swap(Buf& o) { std::swap(p, o.p); std::swap(n, o.n); }
  int alloc(size_t c) { if (c == 0) c = 1; if (c <= n) return 0; if (p) (void)hipFree(p); p = nullptr; n = 0;
    if (hipMalloc((void**)&p, c*sizeof(T)) != hipSuccess) { mcp_set_error("hipMalloc failed"); return -1; } n = c; return 0; }
};
// pinned host staging that outlives the call that fills it (the uploads of an enqueue-only helper are still in flight when it returns)
template <class T> struct PinBuf {
  T* p = nullptr; size_t n = 0;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  int alloc(size_t c) { if (c == 0) c = 1; if (c <= n) return 0; if (p) (void)hipHostFree(p); p = nullptr; n = 0;
    if (hipHostMalloc((void**)&p, c*sizeof(T)) != hipSuccess) { mcp_set_error("hipHostMalloc failed"); return -1; } n = c; return 0; }
};
struct Level {
  int w = 0, h = 0, cap = 0;
  Buf<uint8_t> img, mask, tmp_a, tmp_b;
  Buf<mcp_int2> corners, cand_pos;
  Buf<uint8_t> score8;                             // FAST score per pixel at the detection threshold (0 = no corner), scratch of MakeKeyFrame_Lite
  Buf<int> lut, rowcnt, blk_cnt, score_img;
  Buf<unsigned long long> scan;                    // k_row_tables: per 4-row workgroup, (epoch << 32 | kept corners)
  Buf<double> cand_score;
  Buf<LevelInfo> info;
  bool has_mask = false;
  std::vector<mcp_int2> h_cand; std::vector<double> h_cand_score;
  // Level::imagePrev / vCornersPrev (KeyFrame.h:147-148): the last NUM_PREV frames' level image, corners, row LUT and
  // counts stay resident; [0] is the oldest.  Buffers rotate by pointer swap, nothing is copied.
  static constexpr int NUM_PREV = 2;            // Level::snNumPrev, KeyFrame.cc:71
  int nprev = 0;
  Buf<uint8_t> pimg[NUM_PREV]; Buf<mcp_int2> pcorners[NUM_PREV]; Buf<int> plut[NUM_PREV]; Buf<LevelInfo> pinfo[NUM_PREV];
  int alloc_frame() {
    const size_t npx = (size_t)w*h;
    if (img.alloc(npx) || corners.alloc(cap) || lut.alloc(h) || info.alloc(1)) return -1;
    return 0;
  }
  int push_history() {                           // circular_buffer::push_back of the frame currently held
    if (nprev == NUM_PREV) { pimg[0].swap(pimg[1]); pcorners[0].swap(pcorners[1]); plut[0].swap(plut[1]); pinfo[0].swap(pinfo[1]); --nprev; }
    pimg[nprev].swap(img); pcorners[nprev].swap(corners); plut[nprev].swap(lut); pinfo[nprev].swap(info);
    ++nprev;
    return alloc_frame();                        // the dropped frame's buffers (or fresh ones) become the current frame
  }
};
}  // namespace

struct mcp_kf {
  int device = 0; hipStream_t st = nullptr;
  mcp_kf_params prm;
  Level lev[MCP_LEVELS];
  // scratch reused across calls (no hipMalloc on the per-frame path)
  Buf<DevTdIn> td_in; Buf<mcp_td_out> td_out;
  Buf<mcp_int2> mp_a, mp_b, mp_o; Buf<uint8_t> mp_f, mp_f2; Buf<int> mp_s;
  bool has_image = false;
  LevelInfo* h_info = nullptr;      // pinned, device-visible: the kernels leave the four levels' bookkeeping here (one wait per frame)
  Buf<int> work;                    // threshold histogram + detected-corner count per level; k_row_compact leaves it zero again
  bool work_dirty = true;
  unsigned int scan_epoch = 0;        // k_row_tables: this frame's tag
  Buf<SearchCam> stab; Buf<DevTdIn> bt_in; Buf<mcp_td_out> bt_out;      // batched search: camera table + points of all cameras
  Buf<PfTargetDev> pf_tab; Buf<PfItemDev> pf_items; Buf<int> pf_seq; Buf<mcp_pf_state> pf_state;      // mcp_patch_sequences
  PinBuf<SearchCam> h_stab; PinBuf<DevTdIn> h_bt_in;                                                   // host staging of the batched search ...
  PinBuf<mcp_td_out> h_bt_out;                                                                         // ... and, for mcp_track_frame, its results: the search kernel writes them here as well
  int view_first[MCP_MAX_FRAME_CAMS + 1] = {0}; int view_ncam = 0;                                     // mcp_track_frame_view: where camera c's results of the last frame start in h_bt_out
  PinBuf<PfTargetDev> h_pf_tab; PinBuf<PfItemDev> h_pf_items; PinBuf<int> h_pf_seq; PinBuf<mcp_pf_state> h_pf_state, h_pf_state_out;      // ... and of mcp_track_frame's finder sequences (states in / out)
  hipEvent_t ev = nullptr;
  // SmallBlurryImage of the frame currently held (KeyFrame::mpSBI): thumbnail, zero-mean blurred template, gradient image
  Buf<uint8_t> sbi_small; Buf<float> sbi_templ, sbi_jacs; bool has_sbi = false;
  Buf<uint8_t> sbi_last_small; Buf<float> sbi_last_templ, sbi_last_jacs; bool has_last_sbi = false;   // the SBI made before the current one (Tracker::mmpSBILastFrame)
  Buf<const float*> sbi_ptrs; Buf<double> sbi_out;
  ~mcp_kf() { if (st) (void)hipStreamDestroy(st); if (h_info) (void)hipHostFree(h_info); if (ev) (void)hipEventDestroy(ev); }
  DevKfView view() const {
    DevKfView v;
    for (int l = 0; l < MCP_LEVELS; ++l) { v.img[l] = lev[l].img.p; v.w[l] = lev[l].w; v.h[l] = lev[l].h; v.corners[l] = lev[l].corners.p; v.lut[l] = lev[l].lut.p; v.info[l] = lev[l].info.p; }
    return v;
  }
};

static bool gfx950(int dev) { hipDeviceProp_t p; return hipGetDeviceProperties(&p, dev) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0; }

extern "C" {

mcp_kf* mcp_kf_create(int w, int h, const mcp_kf_params* params) {
  if (w < 64 || h < 64) { mcp_set_error("mcp_kf_create: image too small"); return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { mcp_set_error("mcp_kf_create: no HIP device available (the HIP path has no CPU fallback)"); return nullptr; }
  mcp_kf_params p; p.adaptive_thresh = 1; p.glare_masking = 0; p.half_sample_pavgb = 0; p.device = -1;
  if (params) p = *params;
  int dev = p.device; if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= ndev || !gfx950(dev)) { mcp_set_error("mcp_kf_create: device is not a gfx950 (MI355X)"); return nullptr; }
  if (hipSetDevice(dev) != hipSuccess) { mcp_set_error("hipSetDevice failed"); return nullptr; }
  mcp_kf* k = new mcp_kf(); k->device = dev; k->prm = p;
  if (hipStreamCreateWithFlags(&k->st, hipStreamNonBlocking) != hipSuccess) { mcp_set_error("hipStreamCreate failed"); delete k; return nullptr; }
  if (hipEventCreateWithFlags(&k->ev, hipEventDisableTiming) != hipSuccess || hipHostMalloc((void**)&k->h_info, MCP_LEVELS*sizeof(LevelInfo)) != hipSuccess ||
      k->work.alloc(FRAME_WORK_INTS)) { mcp_set_error("mcp_kf_create: allocation failed"); delete k; return nullptr; }
  std::memset(k->h_info, 0, MCP_LEVELS*sizeof(LevelInfo));
  for (int l = 0; l < MCP_LEVELS; ++l) {
    Level& L = k->lev[l]; L.w = w >> l; L.h = h >> l; L.cap = std::max(1024, L.w*L.h/2);
    const size_t npx = (size_t)L.w*L.h; const int nblk = (L.cap + FAST_BLOCK - 1)/FAST_BLOCK + 2;
    if (L.img.alloc(npx) || L.mask.alloc(npx) || L.score8.alloc(npx) || L.rowcnt.alloc(L.h) || L.scan.alloc((L.h + 3)/4 + 1) || L.corners.alloc(L.cap) ||
        L.lut.alloc(L.h) || L.blk_cnt.alloc(nblk) || L.info.alloc(1) || L.cand_pos.alloc(L.cap) || L.cand_score.alloc(L.cap)) { delete k; return nullptr; }
    (void)hipMemset(L.info.p, 0, sizeof(LevelInfo));
    (void)hipMemset(L.scan.p, 0, ((size_t)(L.h + 3)/4 + 1)*sizeof(unsigned long long));      // (epoch 0 = never written; frames count from 1)
  }
  return k;
}
void mcp_kf_destroy(mcp_kf* k) { if (k) { (void)hipSetDevice(k->device); delete k; } }

// MakeKeyFrame_Lite of every camera of a frame in one submission (the loop of Tracker::TrackFrame, src/Tracker.cc:303-318): the
// uploads, three launches for all levels of all cameras (k_pyr_fast, k_row_count, k_row_compact) and one wait.
// (enqueue on kfs[0]->st without waiting; lite_batch_finish after the stream has been waited for)
// (ride: called after k_pyr_fast has been launched -- host work done here overlaps it -- to name up to two pinned-host -> device copies that
//  k_row_count's grid then carries in one more z-slice, FrameBatch::up_*)
struct FrameRide { int (*fn)(void* ctx, FrameBatch& B); void* ctx; };
static int lite_batch_enqueue(int ncam, mcp_kf* const* kfs, const uint8_t* const* imgs, const int* strides, int imgs_on_device,
                              const uint8_t* const* const* masks, const FrameRide* ride = nullptr) {
  if (ncam < 1 || ncam > MCP_MAX_FRAME_CAMS || !kfs || !imgs || !strides) return img_fail("mcp_kf_make_lite_batch: bad arguments");
  for (int c = 0; c < ncam; ++c) {
    if (!kfs[c] || !imgs[c] || kfs[c]->device != kfs[0]->device) return img_fail("mcp_kf_make_lite_batch: keyframes must live on one device");
    for (int d = 0; d < c; ++d) if (kfs[d] == kfs[c]) return img_fail("mcp_kf_make_lite_batch: a keyframe appears twice");
  }
  ICK(hipSetDevice(kfs[0]->device));
  hipStream_t st = kfs[0]->st;
  static const int fixed_t[4] = { 10, 15, 15, 10 };
  // launches are grouped by the settings a launch shares; a frame's cameras normally share all of them
  for (int c = 1; c < ncam; ++c)
    if (kfs[c]->prm.adaptive_thresh != kfs[0]->prm.adaptive_thresh || kfs[c]->prm.half_sample_pavgb != kfs[0]->prm.half_sample_pavgb)
      return img_fail("mcp_kf_make_lite_batch: the keyframes of a batch must share adaptive_thresh and half_sample_pavgb");
  FrameBatch B; std::memset(&B, 0, sizeof B);
  B.ncam = ncam; B.adaptive = kfs[0]->prm.adaptive_thresh; B.pavgb = kfs[0]->prm.half_sample_pavgb;
  for (int l = 0; l < MCP_LEVELS; ++l) B.detect_t[l] = B.adaptive ? MCP_MIN_FAST_THRESH : fixed_t[l];
  int maxtiles = 0, maxh = 0; bool glare = false;
  for (int c = 0; c < ncam; ++c) {
    mcp_kf* k = kfs[c];
    // the frame currently held moves into the history before it is overwritten, KeyFrame.cc:152-199
    if (k->has_image) for (int l = 0; l < MCP_LEVELS; ++l) if (k->lev[l].push_history()) return -1;
    k->has_image = true;
    if (k->work_dirty) { ICK(hipMemsetAsync(k->work.p, 0, FRAME_WORK_INTS*sizeof(int), st)); }
    k->work_dirty = true;                         // until k_row_compact of this frame has run to the end
    FrameCam& C = B.c[c];
    const Level& L0 = k->lev[0];
    C.w = L0.w; C.h = L0.h; C.work = k->work.p; C.host_info = k->h_info; C.epoch = ++k->scan_epoch;
    if (imgs_on_device) { C.src = imgs[c]; C.src_stride = strides[c]; }
    else { ICK(hipMemcpy2DAsync(L0.img.p, L0.w, imgs[c], strides[c], L0.w, L0.h, hipMemcpyHostToDevice, st)); C.src = L0.img.p; C.src_stride = L0.w; }
    for (int l = 0; l < MCP_LEVELS; ++l) {
      Level& L = k->lev[l];
      C.img[l] = L.img.p; C.score[l] = L.score8.p; C.corners[l] = L.corners.p; C.lut[l] = L.lut.p; C.rowcnt[l] = L.rowcnt.p; C.scan[l] = L.scan.p; C.info[l] = L.info.p; C.cap[l] = L.cap;
      const uint8_t* m = masks && masks[c] ? masks[c][l] : nullptr;
      if (m) ICK(hipMemcpyAsync(L.mask.p, m, (size_t)L.w*L.h, hipMemcpyHostToDevice, st));
      C.mask[l] = (m || k->prm.glare_masking) ? L.mask.p : nullptr;
      L.has_mask = C.mask[l] != nullptr;
    }
    glare = glare || k->prm.glare_masking;
    maxtiles = std::max(maxtiles, ((C.w + PYR_T - 1)/PYR_T)*((C.h + PYR_T - 1)/PYR_T)); maxh = std::max(maxh, C.h);
  }
  hipLaunchKernelGGL(k_pyr_fast, dim3(maxtiles, ncam), dim3(PYR_NT), 0, st, B);
  if (glare) for (int c = 0; c < ncam; ++c) {        // cv::dilate x5 of every level image, KeyFrame.cc:214-238 (needs the level images: after k_pyr_fast)
    mcp_kf* k = kfs[c];
    if (!k->prm.glare_masking) continue;
    for (int l = 0; l < MCP_LEVELS; ++l) {
      Level& L = k->lev[l];
      const size_t npx = (size_t)L.w*L.h;
      if (L.tmp_a.alloc(npx) || L.tmp_b.alloc(npx)) return -1;
      const uint8_t* internal = masks && masks[c] && masks[c][l] ? L.mask.p : nullptr;
      const uint8_t* src = L.img.p; uint8_t* a = L.tmp_a.p; uint8_t* b = L.tmp_b.p;
      for (int it = 0; it < 5; ++it) { hipLaunchKernelGGL(k_dilate5, dim3((L.w + 31)/32, (L.h + 7)/8), dim3(32, 8), 0, st, src, a, L.w, L.h); src = a; std::swap(a, b); }
      hipLaunchKernelGGL(k_glare_mask, dim3((unsigned)((npx + 255)/256)), dim3(256), 0, st, src, internal, L.mask.p, (int)npx);
    }
  }
  if (ride && ride->fn(ride->ctx, B)) return -1;
  const int up = (B.up_n8[0] > 0 || B.up_n8[1] > 0 || B.up_n8[2] > 0 || B.up_n8[3] > 0) ? 1 : 0;
  static const bool one_launch = [] { const char* e = getenv("MCP_IMG_ROW_TABLES"); return e ? atoi(e) != 0 : true; }();      // (0: k_row_count + k_row_compact, the round-5 pair)
  if (one_launch) hipLaunchKernelGGL(k_row_tables, dim3((maxh + 3)/4, MCP_LEVELS, ncam + up), dim3(256), 0, st, B);
  else {
    hipLaunchKernelGGL(k_row_count, dim3((maxh + 3)/4, MCP_LEVELS, ncam + up), dim3(256), 0, st, B);
    B.up_n8[0] = B.up_n8[1] = B.up_n8[2] = B.up_n8[3] = 0;
    hipLaunchKernelGGL(k_row_compact, dim3((maxh + 3)/4, MCP_LEVELS, ncam), dim3(256), 0, st, B);
  }
  ICK(hipGetLastError());
  return 0;
}
static int lite_batch_finish(int ncam, mcp_kf* const* kfs) {
  for (int c = 0; c < ncam; ++c) {
    kfs[c]->work_dirty = false;
    for (int l = 0; l < MCP_LEVELS; ++l) if (kfs[c]->h_info[l].overflow) return img_fail("mcp_kf_make_lite: corner capacity exceeded");
  }
  return 0;
}
int mcp_kf_make_lite_batch(int ncam, mcp_kf* const* kfs, const uint8_t* const* imgs, const int* strides, int imgs_on_device,
                           const uint8_t* const* const* masks) {
  if (lite_batch_enqueue(ncam, kfs, imgs, strides, imgs_on_device, masks)) return -1;
  ICK(hipStreamSynchronize(kfs[0]->st));
  return lite_batch_finish(ncam, kfs);
}
int mcp_kf_make_lite(mcp_kf* k, const uint8_t* img, int stride, const uint8_t* const* masks) {
  mcp_kf* kfs[1] = { k }; const uint8_t* imgs[1] = { img }; const int strides[1] = { stride }; const uint8_t* const* ms[1] = { masks };
  return mcp_kf_make_lite_batch(1, kfs, imgs, strides, 0, masks ? ms : nullptr);
}

static int get_info(mcp_kf* k, int level, LevelInfo* inf) {
  if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level");
  ICK(hipSetDevice(k->device));
  ICK(hipMemcpy(inf, k->lev[level].info.p, sizeof *inf, hipMemcpyDeviceToHost));
  return 0;
}
int mcp_kf_level_size(mcp_kf* k, int level, int* w, int* h) { if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level"); *w = k->lev[level].w; *h = k->lev[level].h; return 0; }
int mcp_kf_get_image(mcp_kf* k, int level, uint8_t* out) {
  if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level");
  ICK(hipSetDevice(k->device));
  ICK(hipMemcpy(out, k->lev[level].img.p, (size_t)k->lev[level].w*k->lev[level].h, hipMemcpyDeviceToHost)); return 0;
}
int mcp_kf_num_corners(mcp_kf* k, int level) { LevelInfo inf; if (get_info(k, level, &inf)) return -1; return inf.n_corners; }
int mcp_kf_get_corners(mcp_kf* k, int level, mcp_int2* out, int cap) {
  LevelInfo inf; if (get_info(k, level, &inf)) return -1;
  const int n = std::min(inf.n_corners, cap);
  if (n > 0) ICK(hipMemcpy(out, k->lev[level].corners.p, sizeof(mcp_int2)*(size_t)n, hipMemcpyDeviceToHost));
  return n;
}
int mcp_kf_get_row_lut(mcp_kf* k, int level, int* out) {
  if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level");
  ICK(hipSetDevice(k->device));
  ICK(hipMemcpy(out, k->lev[level].lut.p, sizeof(int)*(size_t)k->lev[level].h, hipMemcpyDeviceToHost)); return 0;
}
int mcp_kf_fast_thresh(mcp_kf* k, int level) { LevelInfo inf; if (get_info(k, level, &inf)) return -1; return inf.thresh; }
int mcp_kf_get_fast_frequency(mcp_kf* k, int level, double* out) {
  LevelInfo inf; if (get_info(k, level, &inf)) return -1;
  for (int t = 0; t <= MCP_MAX_FAST_THRESH; ++t) out[t] = (double)inf.hist[t];
  return 0;
}

int mcp_kf_make_rest(mcp_kf* k, int use_shi, int use_percent, double top_fraction, double thresh, int nonmax_score) {
  ICK(hipSetDevice(k->device));
  hipStream_t st = k->st;
  for (int l = 0; l < MCP_LEVELS; ++l) {
    Level& L = k->lev[l];
    const size_t npx = (size_t)L.w*L.h;
    if (L.score_img.alloc(npx)) return -1;
    ICK(hipMemsetAsync(L.score_img.p, 0, npx*sizeof(int), st));
    const int nb = (L.cap + FAST_BLOCK - 1)/FAST_BLOCK;
    hipLaunchKernelGGL(k_nonmax_scores, dim3(nb), dim3(FAST_BLOCK), 0, st, (const uint8_t*)L.img.p, L.w, (const mcp_int2*)L.corners.p, (const LevelInfo*)L.info.p, nonmax_score, L.score_img.p);
    hipLaunchKernelGGL((k_candidates<false>), dim3(nb), dim3(FAST_BLOCK), 0, st, (const uint8_t*)L.img.p, L.w, L.h, (const mcp_int2*)L.corners.p, L.info.p, (const int*)L.score_img.p, use_shi, L.blk_cnt.p, L.cand_pos.p, L.cand_score.p);
    hipLaunchKernelGGL((k_candidates<true>), dim3(nb), dim3(FAST_BLOCK), 0, st, (const uint8_t*)L.img.p, L.w, L.h, (const mcp_int2*)L.corners.p, L.info.p, (const int*)L.score_img.p, use_shi, L.blk_cnt.p, L.cand_pos.p, L.cand_score.p);
  }
  ICK(hipStreamSynchronize(st));
  // selection of the scored candidates (a few thousand pairs): sort / threshold on the host, KeyFrame.cc:422-452
  for (int l = 0; l < MCP_LEVELS; ++l) {
    Level& L = k->lev[l];
    LevelInfo inf; ICK(hipMemcpy(&inf, L.info.p, sizeof inf, hipMemcpyDeviceToHost));
    std::vector<mcp_int2> pos(inf.n_cand); std::vector<double> sc(inf.n_cand);
    if (inf.n_cand) { ICK(hipMemcpy(pos.data(), L.cand_pos.p, sizeof(mcp_int2)*pos.size(), hipMemcpyDeviceToHost)); ICK(hipMemcpy(sc.data(), L.cand_score.p, sizeof(double)*sc.size(), hipMemcpyDeviceToHost)); }
    std::vector<int> idx(inf.n_cand); for (int i = 0; i < inf.n_cand; ++i) idx[i] = i;
    L.h_cand.clear(); L.h_cand_score.clear();
    if (use_percent) {
      std::sort(idx.begin(), idx.end(), [&](int a, int b) {           // descending (score, ImageRef) as std::sort(rbegin, rend)
        if (sc[a] != sc[b]) return sc[a] > sc[b];
        if (pos[a].y != pos[b].y) return pos[a].y > pos[b].y;
        return pos[a].x > pos[b].x; });
      const int num = (int)(inf.n_cand*top_fraction);
      for (int i = 0; i < num && i < inf.n_cand; ++i) { L.h_cand.push_back(pos[idx[i]]); L.h_cand_score.push_back(sc[idx[i]]); }
    } else {
      for (int i = 0; i < inf.n_cand; ++i) if (sc[i] > thresh) { L.h_cand.push_back(pos[i]); L.h_cand_score.push_back(sc[i]); }
    }
  }
  // stability pruning, KeyFrame.cc:456-527: every candidate is followed back to the oldest stored frame and forward again
  // to the current one (MiniPatch, search radius 10 per stored frame); it survives if it lands within sqrt(2) pixels
  for (int l = 0; l < MCP_LEVELS; ++l) {
    Level& L = k->lev[l];
    const int n = (int)L.h_cand.size();
    if (L.nprev == 0 || n == 0) continue;
    if (k->mp_a.alloc(n) || k->mp_b.alloc(n) || k->mp_o.alloc(n) || k->mp_f.alloc(n) || k->mp_f2.alloc(n) || k->mp_s.alloc(n)) return -1;
    ICK(hipMemcpyAsync(k->mp_a.p, L.h_cand.data(), sizeof(mcp_int2)*(size_t)n, hipMemcpyHostToDevice, st));
    const int range = L.nprev*10;
    hipLaunchKernelGGL(k_minipatch, dim3(n), dim3(64), 0, st, (const uint8_t*)L.img.p, L.w, L.h, (const uint8_t*)L.pimg[0].p, L.w, L.h,
                       (const mcp_int2*)L.pcorners[0].p, (const LevelInfo*)L.pinfo[0].p, (const int*)L.plut[0].p, n, (const mcp_int2*)k->mp_a.p, (const mcp_int2*)k->mp_a.p, range,
                       k->mp_o.p, k->mp_f.p, k->mp_s.p);
    hipLaunchKernelGGL(k_minipatch, dim3(n), dim3(64), 0, st, (const uint8_t*)L.pimg[0].p, L.w, L.h, (const uint8_t*)L.img.p, L.w, L.h,
                       (const mcp_int2*)L.corners.p, (const LevelInfo*)L.info.p, (const int*)L.lut.p, n, (const mcp_int2*)k->mp_o.p, (const mcp_int2*)k->mp_o.p, range,
                       k->mp_b.p, k->mp_f2.p, k->mp_s.p);
    std::vector<mcp_int2> back(n); std::vector<uint8_t> f1(n), f2(n);
    ICK(hipMemcpyAsync(back.data(), k->mp_b.p, sizeof(mcp_int2)*(size_t)n, hipMemcpyDeviceToHost, st));
    ICK(hipMemcpyAsync(f1.data(), k->mp_f.p, (size_t)n, hipMemcpyDeviceToHost, st));
    ICK(hipMemcpyAsync(f2.data(), k->mp_f2.p, (size_t)n, hipMemcpyDeviceToHost, st));
    ICK(hipStreamSynchronize(st));
    int nk = 0;
    for (int i = 0; i < n; ++i) {
      if (!f1[i] || !f2[i]) continue;
      const int dx = back[i].x - L.h_cand[i].x, dy = back[i].y - L.h_cand[i].y;
      if (dx*dx + dy*dy > 2) continue;
      L.h_cand[nk] = L.h_cand[i]; L.h_cand_score[nk] = L.h_cand_score[i]; ++nk;
    }
    L.h_cand.resize(nk); L.h_cand_score.resize(nk);
  }
  return 0;
}
int mcp_kf_num_prev(mcp_kf* k) { return k->lev[0].nprev; }
int mcp_kf_num_candidates(mcp_kf* k, int level) { if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level"); return (int)k->lev[level].h_cand.size(); }
int mcp_kf_get_candidates(mcp_kf* k, int level, mcp_int2* pos, double* score, int cap) {
  if (level < 0 || level >= MCP_LEVELS) return img_fail("bad level");
  const Level& L = k->lev[level];
  const int n = std::min((int)L.h_cand.size(), cap);
  if (n) { std::memcpy(pos, L.h_cand.data(), sizeof(mcp_int2)*n); std::memcpy(score, L.h_cand_score.data(), sizeof(double)*n); }
  return n;
}

int mcp_minipatch_find(mcp_kf* src, mcp_kf* dst, int level, int n, const mcp_int2* src_pos, const mcp_int2* dst_pos, int range,
                       mcp_int2* out_pos, uint8_t* out_found, int* out_ssd) {
  if (level < 0 || level >= MCP_LEVELS || n < 0) return img_fail("mcp_minipatch_find: bad arguments");
  if (n == 0) return 0;
  ICK(hipSetDevice(dst->device));
  Buf<mcp_int2>& dsp = dst->mp_a; Buf<mcp_int2>& ddp = dst->mp_b; Buf<mcp_int2>& dop = dst->mp_o; Buf<uint8_t>& dfound = dst->mp_f; Buf<int>& dssd = dst->mp_s;
  if (dsp.alloc(n) || ddp.alloc(n) || dop.alloc(n) || dfound.alloc(n) || dssd.alloc(n)) return -1;
  ICK(hipMemcpy(dsp.p, src_pos, sizeof(mcp_int2)*(size_t)n, hipMemcpyHostToDevice));
  ICK(hipMemcpy(ddp.p, dst_pos, sizeof(mcp_int2)*(size_t)n, hipMemcpyHostToDevice));
  const Level& S = src->lev[level]; const Level& D = dst->lev[level];
  hipLaunchKernelGGL(k_minipatch, dim3(n), dim3(64), 0, dst->st, (const uint8_t*)S.img.p, S.w, S.h, (const uint8_t*)D.img.p, D.w, D.h,
                     (const mcp_int2*)D.corners.p, (const LevelInfo*)D.info.p, (const int*)D.lut.p, n, (const mcp_int2*)dsp.p, (const mcp_int2*)ddp.p, range, dop.p, dfound.p, dssd.p);
  ICK(hipStreamSynchronize(dst->st));
  ICK(hipMemcpy(out_pos, dop.p, sizeof(mcp_int2)*(size_t)n, hipMemcpyDeviceToHost));
  ICK(hipMemcpy(out_found, dfound.p, (size_t)n, hipMemcpyDeviceToHost));
  if (out_ssd) ICK(hipMemcpy(out_ssd, dssd.p, sizeof(int)*(size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

// a camera the device code may index: 0 (Newton mode) .. MCP_MAX_INV inverse-polynomial coefficients (as mcp_ba_create checks)
static bool cam_ok(const mcp_camera* c) { return c && c->n_inv >= 0 && c->n_inv <= MCP_MAX_INV; }

static bool est_ok(int e) { return e >= MCP_MEST_TUKEY && e <= MCP_MEST_HUBER; }
int mcp_track_pose_refine(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cfb, double bfw[12], int n_iter,
                          const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last) {
  return mcp_track_pose_refine_m(n, pts, ncam, cams, cfb, bfw, n_iter, nonlinear, override_sigma, mu_last, weights_last, MCP_MEST_TUKEY);
}
// device scratch of the pose iterations, reused across calls.  The small inputs travel in ONE block (bytes): [BaseFromWorld 12 d | mu 6 d |
// pad 6 d | override sigma n_iter d | CamFromBase 12 ncam d | camera models | nonlinear flags], the results [BaseFromWorld | mu] come back
// in one copy: 2 uploads + 2-3 downloads per call instead of 6 + 4.
struct RefineScratch { Buf<mcp_pose_point> dp, dp_keep; Buf<uint8_t> dblk; Buf<double> dJ, dex, de2, dw; Buf<PrmScratch> dprm; Buf<double> de2all; std::vector<uint8_t> hblk; bool last_multi = false;
                       // k_pose_refine_regs reads the block and leaves BaseFromWorld | mu and the weights in PINNED HOST memory: no upload of the
                       // block, no fill of the weights, no copy back (three copy-engine operations and their queue switches per frame)
                       PinBuf<uint8_t> pblk; PinBuf<double> pw; bool res_pinned = false; };
// one scratch per (thread, device): the buffers live on the device that was current when they were allocated, and a thread that serves
// keyframes on two devices must not hand one device's kernels the other's pointers
static RefineScratch& refine_scratch() {
  static thread_local std::map<int, std::unique_ptr<RefineScratch>> per_dev;
  int dev = 0; (void)hipGetDevice(&dev);
  std::unique_ptr<RefineScratch>& p = per_dev[dev];
  if (!p) p.reset(new RefineScratch());
  return *p;
}
// The iterations enqueued on `st`: the points come from host_pts (uploaded first) or are in the scratch's dp already (host_pts == nullptr:
// mcp_track_frame packs them on the device).  BaseFromWorld | mu are left at the head of the scratch's dblk, the weights in dw; *prm_err
// receives the multi-workgroup kernel's give-up flag once the stream has been waited for.
static int refine_enqueue(int n, const mcp_pose_point* host_pts, int ncam, const mcp_camera* cams, const double* cfb, const double bfw[12], int n_iter,
                          const uint8_t* nonlinear, const double* override_sigma, int est, hipStream_t st, unsigned int* prm_err) {
  RefineScratch& rs = refine_scratch();
  const size_t o_ov = 24*sizeof(double), o_cfb = o_ov + 8*(size_t)n_iter, o_cam = o_cfb + 96*(size_t)ncam;
  const size_t o_nl = o_cam + sizeof(mcp_camera)*(size_t)ncam, blk = ((o_nl + (size_t)n_iter + 15)/16)*16;
  static const int use_regs = [] { const char* e = getenv("MCP_TRACK_REFINE_REGS"); return e ? atoi(e) : 1; }();
  // (function attributes are per device: set on the device this call runs on, once per device and thread)
  static thread_local unsigned long long regs_set_mask = 0, regs_ok_mask = 0;
  int cur_dev = 0; (void)hipGetDevice(&cur_dev);
  const unsigned long long dbit = 1ull << (cur_dev & 63);
  if (!(regs_set_mask & dbit)) {
    regs_set_mask |= dbit;
    if (hipFuncSetAttribute((const void*)k_pose_refine_regs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PRR_DYN_LDS) == hipSuccess) regs_ok_mask |= dbit;
    else (void)hipGetLastError();
  }
  bool regs = use_regs && (regs_ok_mask & dbit) && n <= PRR_THREADS*PRR_PPT && ncam <= PRR_CAMS;          // the points fit the register-resident kernel, the rig its LDS
  // many points: the iterations over several workgroups (k_pose_refine_multi); MCP_TRACK_REFINE_MULTI = 0 never, 1 whenever the
  // points do not fit the register-resident kernel (default), 2 always
  const int use_multi = [] { const char* e = getenv("MCP_TRACK_REFINE_MULTI"); return e ? atoi(e) : 1; }();
  const bool multi = n_iter <= PRM_MAX_ITER && ((use_multi == 2 && n >= 64) || (use_multi == 1 && !regs && n > PRR_THREADS*PRR_PPT));
  if (multi) regs = false;
  rs.last_multi = multi;
  if (rs.dp.alloc(n) || rs.dblk.alloc(blk) || rs.dw.alloc(n)) return -1;
  auto alloc_plain = [&]() { return rs.dJ.alloc(12*(size_t)n) || rs.dex.alloc(2*(size_t)n) || rs.de2.alloc(n); };
  if (!regs && alloc_plain()) return -1;
  rs.hblk.assign(blk, 0);
  std::memcpy(rs.hblk.data(), bfw, 96);
  std::memcpy(rs.hblk.data() + o_ov, override_sigma, 8*(size_t)n_iter);
  std::memcpy(rs.hblk.data() + o_cfb, cfb, 96*(size_t)ncam);
  std::memcpy(rs.hblk.data() + o_cam, cams, sizeof(mcp_camera)*(size_t)ncam);
  std::memcpy(rs.hblk.data() + o_nl, nonlinear, (size_t)n_iter);
  if (host_pts) ICK(hipMemcpyAsync(rs.dp.p, host_pts, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyHostToDevice, st));
  rs.res_pinned = false;
  if (regs) {
    if (rs.pblk.alloc(blk) || rs.pw.alloc(n)) return -1;
    std::memcpy(rs.pblk.p, rs.hblk.data(), blk);
    std::memset(rs.pw.p, 0, 8*(size_t)n);                       // weights stay zero when a point was not found
    uint8_t* b = rs.pblk.p;
    double* p_bfw = reinterpret_cast<double*>(b);
    hipLaunchKernelGGL(k_pose_refine_regs, dim3(1), dim3(PRR_THREADS), PRR_DYN_LDS, st, n, rs.dp.p, reinterpret_cast<const mcp_camera*>(b + o_cam), reinterpret_cast<const double*>(b + o_cfb),
                       p_bfw, n_iter, (const uint8_t*)(b + o_nl), reinterpret_cast<const double*>(b + o_ov), p_bfw + 12, rs.pw.p, est, ncam);
    if (hipGetLastError() != hipSuccess) {       // the launch was refused (123 KB of dynamic LDS): the plain kernel does the same work from global memory
      regs = false; regs_ok_mask &= ~dbit;
      if (alloc_plain()) return -1;
    } else rs.res_pinned = true;
  }
  double* d_bfw = reinterpret_cast<double*>(rs.dblk.p); double* d_mu = d_bfw + 12;
  const double* d_ov = reinterpret_cast<const double*>(rs.dblk.p + o_ov); const double* d_cfb = reinterpret_cast<const double*>(rs.dblk.p + o_cfb);
  const mcp_camera* d_cam = reinterpret_cast<const mcp_camera*>(rs.dblk.p + o_cam); const uint8_t* d_nl = rs.dblk.p + o_nl;
  if (!regs) {
    ICK(hipMemcpyAsync(rs.dblk.p, rs.hblk.data(), blk, hipMemcpyHostToDevice, st));
    if (!multi) ICK(hipMemsetAsync(rs.dw.p, 0, 8*(size_t)n, st));             // weights stay zero when no point was found
  }
  *prm_err = 0;
  if (multi) {
    if (rs.dprm.alloc(1)) return -1;
    ICK(hipMemsetAsync(rs.dprm.p, 0, sizeof(PrmScratch), st));
    const int ppw = [] { const char* e = getenv("MCP_TRACK_REFINE_PPW"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();      // points per workgroup (measured at 8000 points: 128: 407 us, 256: 337, 512: 311, 1024: 322)
    const int nwg = std::max(1, std::min(PRM_MAX_WG, (n + ppw - 1)/ppw));
    // the median: per-digit global histograms (default), or -- MCP_TRACK_REFINE_GATHER=1, up to 16k points -- all squared errors through
    // every workgroup's LDS with one barrier.  Measured at 8000 points in 16 workgroups: 311 us vs 322 us per ten iterations; the
    // redundant 8000-key selection in every workgroup costs what the two extra barriers of the histogram form cost.
    const bool gather = n <= PRM_GATHER_MAX && [] { const char* e = getenv("MCP_TRACK_REFINE_GATHER"); return e ? atoi(e) != 0 : false; }();
    size_t dyn = 0;
    if (gather) {
      if (rs.de2all.alloc(2*(size_t)n)) return -1;
      dyn = (size_t)n*sizeof(double);
      static thread_local unsigned long long prm_attr_mask = 0;
      if (!(prm_attr_mask & dbit)) { ICK(hipFuncSetAttribute((const void*)k_pose_refine_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PRM_GATHER_MAX*sizeof(double)))); prm_attr_mask |= dbit; }
    }
    // the workgroups of this launch wait for each other; should one give up (not all of them resident next to the mapper's kernels), the
    // caller redoes the iterations with the single-workgroup kernel from this copy of the points (refine_redo_single)
    if (rs.dp_keep.alloc(n)) return -1;
    ICK(hipMemcpyAsync(rs.dp_keep.p, rs.dp.p, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyDeviceToDevice, st));
    // (the parameters stay in device memory -- this kernel reads the camera models inside its iterations --, the results go to pinned host memory)
    if (rs.pblk.alloc(blk) || rs.pw.alloc(n)) return -1;
    std::memset(rs.pw.p, 0, 8*(size_t)n);
    hipLaunchKernelGGL(k_pose_refine_multi, dim3(nwg), dim3(PRM_THREADS), dyn, st, n, rs.dp.p, d_cam, d_cfb, d_bfw, n_iter, d_nl, d_ov, rs.dJ.p, rs.dex.p, rs.de2.p, d_mu, rs.pw.p, est, rs.dprm.p,
                       gather ? rs.de2all.p : (double*)nullptr, reinterpret_cast<double*>(rs.pblk.p));
    ICK(hipGetLastError());
    rs.res_pinned = true;
    ICK(hipMemcpyAsync(prm_err, &rs.dprm.p->err, sizeof *prm_err, hipMemcpyDeviceToHost, st));
  } else if (!regs)
    hipLaunchKernelGGL(k_pose_refine, dim3(1), dim3(PR_THREADS), 0, st, n, rs.dp.p, d_cam, d_cfb, d_bfw, n_iter, d_nl, d_ov, rs.dJ.p, rs.dex.p, rs.de2.p, d_mu, rs.dw.p, est);
#ifdef MCP_PRR_PROF
  if (regs) {
    ICK(hipStreamSynchronize(st));
    unsigned long long pr[16*8 + 8]; (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prr_prof), sizeof pr);
    { unsigned long long sv[8]; (void)hipMemcpyFromSymbol(sv, HIP_SYMBOL(g_selv_prof), sizeof sv);
      fprintf(stderr, "[prr prof] last vote select: votes %llu  barrier %llu  scan %llu  further steps %llu  gather+barrier %llu  rank %llu  (steps %llu, %llu keys ranked)\n",
              sv[1] - sv[0], sv[2] - sv[1], sv[3] - sv[2], sv[4] - sv[3], sv[5] - sv[4], sv[6] - sv[5], sv[7] & 0xffffffffull, sv[7] >> 32); }
    fprintf(stderr, "[prr prof] kernel: entry to first iteration %llu  iterations %llu  write-back %llu  (cycles)\n", pr[129] - pr[128], pr[130] - pr[129], pr[131] - pr[130]);
    for (int it = 0; it < n_iter && it < 16; ++it) { const unsigned long long* q = pr + 8*it;
      fprintf(stderr, "[prr prof] it %d%s: points %llu  select %llu  accumulate %llu  reduce-scatter+barrier %llu  sum+barrier %llu  solve %llu  barrier %llu  (cycles)\n", it, nonlinear[it] ? " (re-projection)" : "",
              q[1] - q[0], q[2] - q[1], q[6] - q[2], q[7] - q[6], q[3] - q[7], q[4] - q[3], q[5] - q[4]); }
  }
#endif
  ICK(hipGetLastError());
  return 0;
}
// BaseFromWorld | mu (18 doubles) and the last weights to the caller: copies enqueued where the results are on the device; where the kernel
// left them in pinned host memory they are taken from there once the stream has been waited for
static int refine_results_enqueue(RefineScratch& rs, int n, double* back, double* weights_last, hipStream_t st) {
  if (rs.res_pinned) return 0;
  ICK(hipMemcpyAsync(back, rs.dblk.p, 18*sizeof(double), hipMemcpyDeviceToHost, st));
  if (weights_last) ICK(hipMemcpyAsync(weights_last, rs.dw.p, 8*(size_t)n, hipMemcpyDeviceToHost, st));
  return 0;
}
static void refine_results_finish(RefineScratch& rs, int n, double* back, double* weights_last) {
  if (!rs.res_pinned) return;
  std::memcpy(back, rs.pblk.p, 18*sizeof(double));
  if (weights_last) std::memcpy(weights_last, rs.pw.p, 8*(size_t)n);
}
// the multi-workgroup iterations gave up (prm_err): the same iterations again in ONE workgroup, from the kept copy of the points and the
// parameter block still in the scratch's host image; results where refine_enqueue leaves them.  The stream has been waited for.
static int refine_redo_single(int n, int n_iter, int ncam, int est, hipStream_t st) {
  RefineScratch& rs = refine_scratch();
  if (!rs.dp_keep.p || rs.hblk.empty()) return img_fail("pose iterations: nothing kept to redo them from");
  rs.res_pinned = false;
  const size_t o_ov = 24*sizeof(double), o_cfb = o_ov + 8*(size_t)n_iter, o_cam = o_cfb + 96*(size_t)ncam, o_nl = o_cam + sizeof(mcp_camera)*(size_t)ncam;
  ICK(hipMemcpyAsync(rs.dp.p, rs.dp_keep.p, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyDeviceToDevice, st));
  ICK(hipMemcpyAsync(rs.dblk.p, rs.hblk.data(), rs.hblk.size(), hipMemcpyHostToDevice, st));
  ICK(hipMemsetAsync(rs.dw.p, 0, 8*(size_t)n, st));
  double* d_bfw = reinterpret_cast<double*>(rs.dblk.p); double* d_mu = d_bfw + 12;
  hipLaunchKernelGGL(k_pose_refine, dim3(1), dim3(PR_THREADS), 0, st, n, rs.dp.p, reinterpret_cast<const mcp_camera*>(rs.dblk.p + o_cam), reinterpret_cast<const double*>(rs.dblk.p + o_cfb),
                     d_bfw, n_iter, (const uint8_t*)(rs.dblk.p + o_nl), reinterpret_cast<const double*>(rs.dblk.p + o_ov), rs.dJ.p, rs.dex.p, rs.de2.p, d_mu, rs.dw.p, est);
  ICK(hipGetLastError());
  return 0;
}
int mcp_track_pose_refine_m(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cfb, double bfw[12], int n_iter,
                            const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last, int est) {
  if (!mu_last) return img_fail("mcp_track_pose_refine: bad arguments");
  for (int k = 0; k < 6; ++k) mu_last[k] = 0;
  if (n < 0 || ncam <= 0 || n_iter < 0 || !cams || !cfb || !bfw || !est_ok(est) || (n > 0 && !pts) || (n_iter > 0 && (!nonlinear || !override_sigma)))
    return img_fail("mcp_track_pose_refine: bad arguments");
  for (int c = 0; c < ncam; ++c) if (!cam_ok(&cams[c])) return img_fail("mcp_track_pose_refine: bad camera");
  if (n == 0 || n_iter == 0) return 0;
  for (int i = 0; i < n; ++i) if (pts[i].cam < 0 || pts[i].cam >= ncam) return img_fail("mcp_track_pose_refine: camera index out of range");
  int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return img_fail("mcp_track_pose_refine: no HIP device");
  hipStream_t st = nullptr;
  unsigned int prm_err = 0;
  if (refine_enqueue(n, pts, ncam, cams, cfb, bfw, n_iter, nonlinear, override_sigma, est, st, &prm_err)) return -1;
  RefineScratch& rs = refine_scratch();
  double back[18];
  ICK(hipMemcpyAsync(pts, rs.dp.p, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyDeviceToHost, st));
  if (refine_results_enqueue(rs, n, back, weights_last, st)) return -1;
  ICK(hipStreamSynchronize(st));
  refine_results_finish(rs, n, back, weights_last);
  if (prm_err || (rs.last_multi && getenv("MCP_TRACK_TEST_PRM_GIVEUP"))) {
    if (refine_redo_single(n, n_iter, ncam, est, st)) return -1;
    ICK(hipMemcpyAsync(pts, rs.dp.p, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyDeviceToHost, st));
    if (refine_results_enqueue(rs, n, back, weights_last, st)) return -1;
    ICK(hipStreamSynchronize(st));
  }
  std::memcpy(bfw, back, 96); std::memcpy(mu_last, back + 12, 48);
  return 0;
}

int mcp_track_pose_refine_sharded(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cfb, double bfw[12], int n_iter,
                                  const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last,
                                  mcp_allreduce_fn allreduce, void* user, int rank, int world, int cap) {
  return mcp_track_pose_refine_sharded_m(n, pts, ncam, cams, cfb, bfw, n_iter, nonlinear, override_sigma, mu_last, weights_last, allreduce, user, rank, world, cap, MCP_MEST_TUKEY);
}
int mcp_track_pose_refine_sharded_m(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cfb, double bfw[12], int n_iter,
                                    const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last,
                                    mcp_allreduce_fn allreduce, void* user, int rank, int world, int cap, int est) {
  if (!mu_last) return img_fail("mcp_track_pose_refine_sharded: bad arguments");
  for (int k = 0; k < 6; ++k) mu_last[k] = 0;
  if (n < 0 || ncam <= 0 || n_iter < 0 || !cams || !cfb || !bfw || world < 1 || rank < 0 || rank >= world || cap < n || cap < 1 || (world > 1 && !allreduce) ||
      !est_ok(est) || (n > 0 && !pts) || (n_iter > 0 && (!nonlinear || !override_sigma)))
    return img_fail("mcp_track_pose_refine_sharded: bad arguments");
  for (int c = 0; c < ncam; ++c) if (!cam_ok(&cams[c])) return img_fail("mcp_track_pose_refine_sharded: bad camera");
  for (int i = 0; i < n; ++i) if (pts[i].cam < 0 || pts[i].cam >= ncam) return img_fail("mcp_track_pose_refine_sharded: camera index out of range");
  int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return img_fail("mcp_track_pose_refine_sharded: no HIP device");
  if (n_iter == 0) return 0;      // (a rank without points still joins every collective below)
  struct Scratch { Buf<mcp_pose_point> dp; Buf<mcp_camera> dc; Buf<double> dcfb, dpose, dv6, dJ, dex, de2, dtab, dacc, dw; Buf<unsigned int> dcnt; };
  static thread_local Scratch rs;
  const size_t tab = (size_t)world*cap;
  if (rs.dp.alloc(std::max(n, 1)) || rs.dc.alloc(ncam) || rs.dcfb.alloc(12*(size_t)ncam) || rs.dpose.alloc(12) || rs.dv6.alloc(6) || rs.dJ.alloc(12*(size_t)std::max(n, 1)) ||
      rs.dex.alloc(2*(size_t)std::max(n, 1)) || rs.de2.alloc(std::max(n, 1)) || rs.dtab.alloc(tab + world) || rs.dacc.alloc(28) || rs.dw.alloc(std::max(n, 1)) || rs.dcnt.alloc(1)) return -1;
  hipStream_t st = nullptr;
  if (n) ICK(hipMemcpyAsync(rs.dp.p, pts, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyHostToDevice, st));
  ICK(hipMemcpyAsync(rs.dc.p, cams, sizeof(mcp_camera)*(size_t)ncam, hipMemcpyHostToDevice, st));
  ICK(hipMemcpyAsync(rs.dcfb.p, cfb, 96*(size_t)ncam, hipMemcpyHostToDevice, st));
  ICK(hipMemcpyAsync(rs.dpose.p, bfw, 96, hipMemcpyHostToDevice, st));
  ICK(hipMemsetAsync(rs.dv6.p, 0, 48, st));
  ICK(hipMemsetAsync(rs.dw.p, 0, 8*(size_t)std::max(n, 1), st));
  for (int it = 0; it < n_iter; ++it) {
    ICK(hipMemsetAsync(rs.dtab.p, 0, (tab + world)*sizeof(double), st));
    ICK(hipMemsetAsync(rs.dcnt.p, 0, sizeof(unsigned int), st));
    if (n) hipLaunchKernelGGL(k_pr_project, dim3((n + 255)/256), dim3(256), 0, st, n, rs.dp.p, (const mcp_camera*)rs.dc.p, (const double*)rs.dcfb.p,
                              (const double*)rs.dpose.p, (const double*)rs.dv6.p, it, (int)(nonlinear[it] != 0), rs.dJ.p, rs.dex.p, rs.de2.p,
                              rs.dtab.p + (size_t)rank*cap, rs.dtab.p + tab + rank, rs.dcnt.p);
    if (world > 1) { ICK(hipStreamSynchronize(st)); if (allreduce(user, rs.dtab.p, tab + world, (void*)st) != 0) return img_fail("mcp_track_pose_refine_sharded: all-reduce hook failed"); }
    hipLaunchKernelGGL(k_pr_accum, dim3(1), dim3(1024), 0, st, n, (const mcp_pose_point*)rs.dp.p, (const double*)rs.dJ.p, (const double*)rs.dex.p, (const double*)rs.de2.p,
                       (const double*)rs.dtab.p, (const double*)(rs.dtab.p + tab), world, cap, override_sigma[it], (int)(it == n_iter - 1), rs.dacc.p, rs.dw.p, est);
    if (world > 1) { ICK(hipStreamSynchronize(st)); if (allreduce(user, rs.dacc.p, 27, (void*)st) != 0) return img_fail("mcp_track_pose_refine_sharded: all-reduce hook failed"); }
    hipLaunchKernelGGL(k_pr_solve, dim3(1), dim3(64), 0, st, (const double*)rs.dacc.p, (const double*)(rs.dtab.p + tab), world, rs.dpose.p, rs.dv6.p);
  }
  if (n) ICK(hipMemcpyAsync(pts, rs.dp.p, sizeof(mcp_pose_point)*(size_t)n, hipMemcpyDeviceToHost, st));
  ICK(hipMemcpyAsync(bfw, rs.dpose.p, 96, hipMemcpyDeviceToHost, st));
  ICK(hipMemcpyAsync(mu_last, rs.dv6.p, 48, hipMemcpyDeviceToHost, st));
  if (weights_last && n) ICK(hipMemcpyAsync(weights_last, rs.dw.p, 8*(size_t)n, hipMemcpyDeviceToHost, st));
  ICK(hipStreamSynchronize(st));
  return 0;
}

// ---- SmallBlurryImage / Relocaliser --------------------------------------------------------------------------------
// cv::resize's 8U INTER_LINEAR taps [3P-memory]: source coordinate (d+0.5)*scale-0.5 in float, clamped, 11-bit weights
static void sbi_resize_coeffs(int src, int dst, int* idx, short* w0, short* w1) {
  const double scale = (double)src/dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5)*scale - 0.5);
    int s0 = (int)floorf(f);
    f -= s0;
    if (s0 < 0) { f = 0; s0 = 0; }
    if (s0 >= src - 1) { f = 0; s0 = src - 1; }
    idx[d] = s0;
    w0[d] = (short)lrintf((1.f - f)*2048.f); w1[d] = (short)lrintf(f*2048.f);
  }
}
int mcp_kf_make_sbi(mcp_kf* k, double blur) {
  if (!k->has_image) return img_fail("mcp_kf_make_sbi: the handle holds no frame");
  if (!(blur > 0)) return img_fail("mcp_kf_make_sbi: blur must be positive");
  ICK(hipSetDevice(k->device));
  if (k->has_sbi) {     // the previous SBI becomes "last frame's" (pointer swap), Tracker.cc: mmpSBILastFrame / mmpSBIThisFrame
    k->sbi_last_small.swap(k->sbi_small); k->sbi_last_templ.swap(k->sbi_templ); k->sbi_last_jacs.swap(k->sbi_jacs);
    k->has_last_sbi = true;
  }
  if (k->sbi_small.alloc(SBI_N) || k->sbi_templ.alloc(SBI_N) || k->sbi_jacs.alloc(2*SBI_N)) return -1;
  SbiTables tb;
  const Level& L = k->lev[0];
  sbi_resize_coeffs(L.w, SBI_W, tb.xi, tb.xa, tb.xb);
  sbi_resize_coeffs(L.h, SBI_H, tb.yi, tb.ya, tb.yb);
  // CVD::convolveGaussian taps [3P-memory]: half size ceil(3 sigma), exp(-i^2/2 sigma^2), unit sum
  tb.ks = std::min(31, (int)std::ceil(3.0*blur));
  double sum = 1.0;
  for (int i = 1; i <= tb.ks; ++i) sum += 2.0*std::exp(-(double)i*i/(2.0*blur*blur));
  tb.k[0] = (float)(1.0/sum);
  for (int i = 1; i < 32; ++i) tb.k[i] = (i <= tb.ks) ? (float)(std::exp(-(double)i*i/(2.0*blur*blur))/sum) : 0.f;
  hipLaunchKernelGGL(k_sbi_make, dim3(1), dim3(256), 0, k->st, (const uint8_t*)L.img.p, L.w, L.h, tb, k->sbi_small.p, k->sbi_templ.p, k->sbi_jacs.p);
  ICK(hipStreamSynchronize(k->st));
  k->has_sbi = true;
  return 0;
}
int mcp_kf_get_sbi(mcp_kf* k, uint8_t* small_img, float* templ, float* jacs) {
  if (!k->has_sbi) return img_fail("mcp_kf_get_sbi: no SmallBlurryImage made");
  ICK(hipSetDevice(k->device));
  if (small_img) ICK(hipMemcpy(small_img, k->sbi_small.p, SBI_N, hipMemcpyDeviceToHost));
  if (templ) ICK(hipMemcpy(templ, k->sbi_templ.p, SBI_N*sizeof(float), hipMemcpyDeviceToHost));
  if (jacs) ICK(hipMemcpy(jacs, k->sbi_jacs.p, 2*SBI_N*sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}
int mcp_sbi_score(mcp_kf* cur, int n, mcp_kf* const* cands, double* scores, int* best) {
  if (!cur->has_sbi || n < 0) return img_fail("mcp_sbi_score: bad arguments");
  *best = -1;
  if (n == 0) return 0;
  ICK(hipSetDevice(cur->device));
  std::vector<const float*> ptrs(n);
  for (int i = 0; i < n; ++i) ptrs[i] = (cands[i] && cands[i]->has_sbi) ? cands[i]->sbi_templ.p : nullptr;   // "KF doesn't have small blurry image! Skipping"
  if (cur->sbi_ptrs.alloc(n) || cur->sbi_out.alloc(std::max(n, 8))) return -1;
  ICK(hipMemcpyAsync(cur->sbi_ptrs.p, ptrs.data(), sizeof(float*)*(size_t)n, hipMemcpyHostToDevice, cur->st));
  hipLaunchKernelGGL(k_sbi_score, dim3((n + 63)/64), dim3(64), 0, cur->st, (const float*)cur->sbi_templ.p, (const float* const*)cur->sbi_ptrs.p, n, cur->sbi_out.p);
  ICK(hipMemcpyAsync(scores, cur->sbi_out.p, sizeof(double)*(size_t)n, hipMemcpyDeviceToHost, cur->st));
  ICK(hipStreamSynchronize(cur->st));
  double b = std::numeric_limits<double>::max();
  for (int i = 0; i < n; ++i) if (ptrs[i] && scores[i] < b) { b = scores[i]; *best = i; }      // strict <: first smallest, Relocaliser.cc:113
  return 0;
}
int mcp_sbi_iterate(mcp_kf* cur, mcp_kf* target, int iterations, double se2[6], double* score) {
  if (!cur->has_sbi || !target->has_sbi || iterations < 0) return img_fail("mcp_sbi_iterate: both keyframes need a SmallBlurryImage with gradients");
  ICK(hipSetDevice(cur->device));
  if (cur->sbi_out.alloc(8)) return -1;
  hipLaunchKernelGGL(k_sbi_iterate, dim3(1), dim3(256), 0, cur->st, (const float*)cur->sbi_templ.p, (const float*)target->sbi_templ.p, (const float*)target->sbi_jacs.p, iterations, cur->sbi_out.p);
  double o[7];
  ICK(hipMemcpyAsync(o, cur->sbi_out.p, sizeof o, hipMemcpyDeviceToHost, cur->st));
  ICK(hipStreamSynchronize(cur->st));
  std::memcpy(se2, o, 6*sizeof(double)); *score = o[6];
  return 0;
}
// Tracker::CalcSBIRotation's alignment (Tracker.cc:1687-1720): this frame's SBI against the one made before it on this handle
int mcp_sbi_iterate_last(mcp_kf* k, int iterations, double se2[6], double* score) {
  if (!k->has_sbi || !k->has_last_sbi || iterations < 0) return img_fail("mcp_sbi_iterate_last: needs two consecutive SmallBlurryImages on the handle");
  ICK(hipSetDevice(k->device));
  if (k->sbi_out.alloc(8)) return -1;
  hipLaunchKernelGGL(k_sbi_iterate, dim3(1), dim3(256), 0, k->st, (const float*)k->sbi_templ.p, (const float*)k->sbi_last_templ.p, (const float*)k->sbi_last_jacs.p, iterations, k->sbi_out.p);
  double o[7];
  ICK(hipMemcpyAsync(o, k->sbi_out.p, sizeof o, hipMemcpyDeviceToHost, k->st));
  ICK(hipStreamSynchronize(k->st));
  std::memcpy(se2, o, 6*sizeof(double)); *score = o[6];
  return 0;
}
// SmallBlurryImage::SE3fromSE2 (:250-310): two points, three Gauss-Newton steps on SO3 -- control-plane arithmetic, run on the
// host with the same camera functions the kernels use (ba_device.h is __host__ __device__)
int mcp_sbi_se3_from_se2(const double se2[6], const mcp_camera* cs, const mcp_camera* ct, double R[9]) {
  if (!cs || !ct || cs->n_inv < 0 || cs->n_inv > MCP_MAX_INV || ct->n_inv < 0 || ct->n_inv > MCP_MAX_INV) return img_fail("mcp_sbi_se3_from_se2: bad camera");
  const double c[2] = { SBI_W/2, SBI_H/2 };
  const double off[2][2] = { { 5, 0 }, { -5, 0 } };
  double turned[2][2], orig[2][3];
  for (int i = 0; i < 2; ++i) {
    turned[i][0] = c[0] + se2[0]*off[i][0] + se2[1]*off[i][1] + se2[4];
    turned[i][1] = c[1] + se2[2]*off[i][0] + se2[3]*off[i][1] + se2[5];
    // TaylorCamera::UnProject, TaylorCamera.cc:319-347
    const double det = ct->affine[0]*ct->affine[3] - ct->affine[1]*ct->affine[2];
    const double ai[4] = { ct->affine[3]/det, -ct->affine[1]/det, -ct->affine[2]/det, ct->affine[0]/det };
    const double dx = c[0] + off[i][0] - ct->center[0], dy = c[1] + off[i][1] - ct->center[1];
    const double x = ai[0]*dx + ai[1]*dy, y = ai[2]*dx + ai[3]*dy;
    const double rho = std::sqrt(x*x + y*y);
    const double p[5] = { ct->params[0], 0.0, ct->params[1], ct->params[2], ct->params[3] };
    double z = p[4]; for (int q = 3; q >= 0; --q) z = z*rho + p[q];
    const double n = std::sqrt(x*x + y*y + z*z);
    orig[i][0] = x/n; orig[i][1] = y/n; orig[i][2] = z/n;
  }
  double so3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  for (int it = 0; it < 3; ++it) {
    double C[9] = { 10, 0, 0, 0, 10, 0, 0, 0, 10 }, v[3] = { 0, 0, 0 };
    for (int i = 0; i < 2; ++i) {
      double cam[3]; mat3_vec(so3, orig[i], cam);
      Projection P; cam_project<true>(*cs, cam, P);
      const double err[2] = { turned[i][0] - P.u, turned[i][1] - P.v };
      double dT[3], dP[3]; cam_sphere_deriv(cam, dT, dP);
      double J[2][3];
      for (int m = 0; m < 3; ++m) {
        double mot[3] = { 0, 0, 0 };
        mot[(m + 1)%3] = -cam[(m + 2)%3]; mot[(m + 2)%3] = cam[(m + 1)%3];
        const double sm[2] = { dT[0]*mot[0] + dT[1]*mot[1] + dT[2]*mot[2], dP[0]*mot[0] + dP[1]*mot[1] + dP[2]*mot[2] };
        J[0][m] = P.D[0]*sm[0] + P.D[1]*sm[1]; J[1][m] = P.D[2]*sm[0] + P.D[3]*sm[1];
      }
      for (int r = 0; r < 2; ++r) for (int a = 0; a < 3; ++a) { v[a] += J[r][a]*err[r]; for (int b = 0; b < 3; ++b) C[3*a + b] += J[r][a]*J[r][b]; }
    }
    const double c00 = C[4]*C[8] - C[5]*C[7], c01 = C[5]*C[6] - C[3]*C[8], c02 = C[3]*C[7] - C[4]*C[6];
    const double id = 1.0/(C[0]*c00 + C[1]*c01 + C[2]*c02);
    const double Ci[9] = { c00*id, (C[2]*C[7] - C[1]*C[8])*id, (C[1]*C[5] - C[2]*C[4])*id,
                           c01*id, (C[0]*C[8] - C[2]*C[6])*id, (C[2]*C[3] - C[0]*C[5])*id,
                           c02*id, (C[1]*C[6] - C[0]*C[7])*id, (C[0]*C[4] - C[1]*C[3])*id };
    double mu[3]; mat3_vec(Ci, v, mu);
    double E[9], Rn[9]; so3_exp(mu, E); mat3_mul(E, so3, Rn);
    std::memcpy(so3, Rn, sizeof Rn);
  }
  std::memcpy(R, so3, 9*sizeof(double));
  return 0;
}

int mcp_track_search(mcp_kf* target, const mcp_camera* cam, const double bfw[12], const double cfb[12], int n, const mcp_td_in* in,
                     int range, int subpix_its, int exhaustive, mcp_td_out* out) {
  if (n < 0 || !target || !cam_ok(cam) || !bfw || !cfb || (n > 0 && (!in || !out))) return img_fail("mcp_track_search: bad arguments");
  if (n == 0) return 0;
  ICK(hipSetDevice(target->device));
  std::vector<DevTdIn> h(n);
  for (int i = 0; i < n; ++i) {
    const mcp_td_in& p = in[i];
    if (!p.source_kf || p.source_level < 0 || p.source_level >= MCP_LEVELS) return img_fail("mcp_track_search: point without a resident source keyframe");
    std::memcpy(h[i].world_pos, p.world_pos, 24); std::memcpy(h[i].pixel_right_w, p.pixel_right_w, 24); std::memcpy(h[i].pixel_down_w, p.pixel_down_w, 24);
    const Level& S = p.source_kf->lev[p.source_level];
    h[i].src_img = S.img.p; h[i].src_w = S.w; h[i].src_h = S.h; h[i].center_x = p.center_x; h[i].center_y = p.center_y; h[i].fixed = p.fixed;
  }
  Buf<DevTdIn>& din = target->td_in; Buf<mcp_td_out>& dout = target->td_out;
  if (din.alloc(n) || dout.alloc(n)) return -1;
  ICK(hipMemcpyAsync(din.p, h.data(), sizeof(DevTdIn)*(size_t)n, hipMemcpyHostToDevice, target->st));
  Se3 B, C; std::memcpy(B.R, bfw, 72); std::memcpy(B.t, bfw + 9, 24); std::memcpy(C.R, cfb, 72); std::memcpy(C.t, cfb + 9, 24);
  hipLaunchKernelGGL(k_track_search, dim3(n), dim3(64), 0, target->st, target->view(), *cam, B, C, n, (const DevTdIn*)din.p, range, subpix_its, exhaustive, dout.p);
  ICK(hipMemcpyAsync(out, dout.p, sizeof(mcp_td_out)*(size_t)n, hipMemcpyDeviceToHost, target->st));
  ICK(hipStreamSynchronize(target->st));
  return 0;
}

// pack: arguments checked, the camera table and the points of all cameras written to the pinned staging (targets[0]->h_stab / h_bt_in), device
// buffers sized.  launch: the two uploads (unless something else has carried them: `uploaded`) and the kernel.
static int search_batch_pack(int ncam, mcp_kf* const* targets, const mcp_camera* cams, const double* cfb, const int* n, const mcp_td_in* const* in,
                             mcp_td_out* const* out, int* total_out, int* maxn_out, bool view = false /* mcp_track_frame's view mode: no caller arrays */) {
  *total_out = 0; *maxn_out = 0;
  if (ncam < 1 || ncam > MCP_MAX_FRAME_CAMS || !targets || !cams || !cfb || !n || !in || (!out && !view)) return img_fail("mcp_track_search_batch: bad arguments");
  int total = 0, maxn = 0;
  for (int c = 0; c < ncam; ++c) {
    if (!targets[c] || n[c] < 0 || !cam_ok(&cams[c]) || targets[c]->device != targets[0]->device || (n[c] > 0 && (!in[c] || (!view && !out[c])))) return img_fail("mcp_track_search_batch: bad arguments");
    total += n[c]; maxn = std::max(maxn, n[c]);
  }
  if (total == 0) return 0;
  mcp_kf* k0 = targets[0];
  ICK(hipSetDevice(k0->device));
  if (k0->h_stab.alloc(MCP_MAX_FRAME_CAMS) || k0->h_bt_in.alloc(total)) return -1;
  SearchCam* tab = k0->h_stab.p; DevTdIn* h = k0->h_bt_in.p;
  int first = 0;
  for (int c = 0; c < ncam; ++c) {
    SearchCam& S = tab[c];
    S.T = targets[c]->view(); S.cam = cams[c]; std::memcpy(S.cfb.R, cfb + 12*c, 72); std::memcpy(S.cfb.t, cfb + 12*c + 9, 24); S.n = n[c]; S.first = first;
    for (int i = 0; i < n[c]; ++i) {
      const mcp_td_in& p = in[c][i]; DevTdIn& d = h[first + i];
      if (!p.source_kf || p.source_level < 0 || p.source_level >= MCP_LEVELS) return img_fail("mcp_track_search_batch: point without a resident source keyframe");
      std::memcpy(d.world_pos, p.world_pos, 24); std::memcpy(d.pixel_right_w, p.pixel_right_w, 24); std::memcpy(d.pixel_down_w, p.pixel_down_w, 24);
      const Level& Sl = p.source_kf->lev[p.source_level];
      d.src_img = Sl.img.p; d.src_w = Sl.w; d.src_h = Sl.h; d.center_x = p.center_x; d.center_y = p.center_y; d.fixed = p.fixed;
    }
    first += n[c];
  }
  if (k0->stab.alloc(MCP_MAX_FRAME_CAMS) || k0->bt_in.alloc(total) || k0->bt_out.alloc(total)) return -1;
  *total_out = total; *maxn_out = maxn;
  return 0;
}
static int search_batch_launch(int ncam, mcp_kf* k0, int total, int maxn, const double bfw[12], int range, int subpix_its, int exhaustive, bool uploaded,
                               mcp_td_out* host_out, mcp_pose_point* pose_pts) {
  static_assert(sizeof(SearchCam) % 8 == 0 && sizeof(DevTdIn) % 8 == 0, "the frame's upload slice copies 8-byte words");
  hipStream_t st = k0->st;
  if (!uploaded) {
    ICK(hipMemcpyAsync(k0->stab.p, k0->h_stab.p, sizeof(SearchCam)*(size_t)ncam, hipMemcpyHostToDevice, st));
    ICK(hipMemcpyAsync(k0->bt_in.p, k0->h_bt_in.p, sizeof(DevTdIn)*(size_t)total, hipMemcpyHostToDevice, st));
  }
  Se3 Bw; std::memcpy(Bw.R, bfw, 72); std::memcpy(Bw.t, bfw + 9, 24);
  hipLaunchKernelGGL(k_track_search_batch, dim3(maxn, ncam), dim3(64), 0, st, (const SearchCam*)k0->stab.p, Bw, (const DevTdIn*)k0->bt_in.p, range, subpix_its, exhaustive, k0->bt_out.p,
                     host_out, pose_pts);
  ICK(hipGetLastError());
  return 0;
}
// SearchForPoints of every camera of a frame in one launch (the per-camera loops of Tracker::TrackMap, src/Tracker.cc:985-1030, 1299-1384)
// (the launch on targets[0]->st, results left in targets[0]->bt_out in camera-major order; *total_out = points of all cameras)
static int search_batch_enqueue(int ncam, mcp_kf* const* targets, const mcp_camera* cams, const double bfw[12], const double* cfb, const int* n,
                                const mcp_td_in* const* in, int range, int subpix_its, int exhaustive, mcp_td_out* const* out, int* total_out) {
  int total = 0, maxn = 0;
  *total_out = 0;
  if (!bfw) return img_fail("mcp_track_search_batch: bad arguments");
  if (search_batch_pack(ncam, targets, cams, cfb, n, in, out, &total, &maxn)) return -1;
  if (total == 0) return 0;
  if (search_batch_launch(ncam, targets[0], total, maxn, bfw, range, subpix_its, exhaustive, false, nullptr, nullptr)) return -1;
  *total_out = total;
  return 0;
}
static int search_batch_copy_out(int ncam, mcp_kf* k0, const int* n, mcp_td_out* const* out, int total) {
  // one copy when the caller's per-camera result arrays are the slices of one array (they are laid out like the device buffer)
  hipStream_t st = k0->st;
  bool contiguous = true;
  { mcp_td_out* expect = nullptr;
    for (int c = 0; c < ncam; ++c) { if (!n[c]) continue; if (expect && out[c] != expect) contiguous = false; expect = out[c] + n[c]; } }
  if (contiguous) {
    int c0 = 0; while (c0 < ncam && !n[c0]) ++c0;
    ICK(hipMemcpyAsync(out[c0], k0->bt_out.p, sizeof(mcp_td_out)*(size_t)total, hipMemcpyDeviceToHost, st));
  } else {
    int first = 0;
    for (int c = 0; c < ncam; ++c) {
      if (n[c]) ICK(hipMemcpyAsync(out[c], k0->bt_out.p + first, sizeof(mcp_td_out)*(size_t)n[c], hipMemcpyDeviceToHost, st));
      first += n[c];
    }
  }
  return 0;
}
int mcp_track_search_batch(int ncam, mcp_kf* const* targets, const mcp_camera* cams, const double bfw[12], const double* cfb, const int* n,
                           const mcp_td_in* const* in, int range, int subpix_its, int exhaustive, mcp_td_out* const* out) {
  int total = 0;
  if (search_batch_enqueue(ncam, targets, cams, bfw, cfb, n, in, range, subpix_its, exhaustive, out, &total)) return -1;
  if (total == 0) return 0;
  if (search_batch_copy_out(ncam, targets[0], n, out, total)) return -1;
  ICK(hipStreamSynchronize(targets[0]->st));
  return 0;
}

// One stage of Tracker::TrackMap for a whole frame in ONE submission (include/mcp_img.h): the launches of mcp_kf_make_lite_batch, the search
// (mcp_track_search_batch, or -- with finder states -- mcp_patch_sequences in MCP_PF_TRACK mode, one single-item sequence per point) and
// mcp_track_pose_refine_m back to back on the frame's stream, the TrackerData -> pose-point packing in between done on the device, one wait at
// the end.  Same kernels on the same data as the three calls: identical results.
static int track_sequences_pack(int ncam, mcp_kf* const* targets, const mcp_camera* cams, const double bfw[12], const double* cfb, const int* n,
                                const mcp_td_in* const* in, const int* const* point_key, mcp_pf_state* const* state, int* total_out) {
  *total_out = 0;
  int total = 0;
  for (int c = 0; c < ncam; ++c) {
    if (!targets[c] || n[c] < 0 || targets[c]->device != targets[0]->device || (n[c] > 0 && (!in[c] || !state[c] || !point_key[c]))) return img_fail("mcp_track_frame: bad arguments");
    total += n[c];
  }
  if (total == 0) return 0;
  mcp_kf* k0 = targets[0];
  if (k0->h_pf_tab.alloc(MCP_MAX_FRAME_CAMS) || k0->h_pf_items.alloc(total) || k0->h_pf_seq.alloc(total + 2) || k0->h_pf_state.alloc(total) || k0->h_pf_state_out.alloc(total)) return -1;
  PfTargetDev* tab = k0->h_pf_tab.p; PfItemDev* h = k0->h_pf_items.p; int* seq = k0->h_pf_seq.p; mcp_pf_state* hs = k0->h_pf_state.p;
  for (int c = 0; c < ncam; ++c) {
    PfTargetDev& D = tab[c];
    D.T = targets[c]->view(); D.mask0 = targets[c]->lev[0].has_mask ? targets[c]->lev[0].mask.p : nullptr; D.cam = cams[c];
    std::memcpy(D.bfw.R, bfw, 72); std::memcpy(D.bfw.t, bfw + 9, 24);
    std::memcpy(D.cfb.R, cfb + 12*c, 72); std::memcpy(D.cfb.t, cfb + 12*c + 9, 24);
  }
  int first = 0;
  for (int c = 0; c < ncam; ++c) {
    for (int i = 0; i < n[c]; ++i) {
      const mcp_td_in& p = in[c][i]; PfItemDev& d = h[first + i];
      if (!p.source_kf || p.source_level < 0 || p.source_level >= MCP_LEVELS) return img_fail("mcp_track_frame: point without a resident source keyframe");
      std::memcpy(d.p.world_pos, p.world_pos, 24); std::memcpy(d.p.pixel_right_w, p.pixel_right_w, 24); std::memcpy(d.p.pixel_down_w, p.pixel_down_w, 24);
      const Level& Sl = p.source_kf->lev[p.source_level];
      d.p.src_img = Sl.img.p; d.p.src_w = Sl.w; d.p.src_h = Sl.h; d.p.center_x = p.center_x; d.p.center_y = p.center_y; d.p.fixed = p.fixed;
      d.point_key = point_key[c][i]; d.target = c; d.start_x = 0.0; d.start_y = 0.0;
    }
    if (n[c]) std::memcpy(&hs[first], state[c], sizeof(mcp_pf_state)*(size_t)n[c]);
    first += n[c];
  }
  for (int i = 0; i <= total; ++i) seq[i] = i;
  seq[total + 1] = 0;                              // (padding: the ride copies 8-byte words)
  ICK(hipSetDevice(k0->device));
  if (k0->pf_tab.alloc(MCP_MAX_FRAME_CAMS) || k0->pf_items.alloc(total) || k0->pf_seq.alloc(total + 2) || k0->pf_state.alloc(total) || k0->bt_out.alloc(total)) return -1;
  *total_out = total;
  return 0;
}
static int track_sequences_launch(int ncam, mcp_kf* k0, int total, int range, int subpix_its, int exhaustive, bool uploaded, mcp_td_out* host_out, mcp_pf_state* host_state,
                                  mcp_pose_point* pose_pts) {
  static_assert(sizeof(PfTargetDev) % 8 == 0 && sizeof(PfItemDev) % 8 == 0 && sizeof(mcp_pf_state) % 8 == 0, "the frame's upload slice copies 8-byte words");
  hipStream_t st = k0->st;
  if (!uploaded) {
    ICK(hipMemcpyAsync(k0->pf_tab.p, k0->h_pf_tab.p, sizeof(PfTargetDev)*(size_t)ncam, hipMemcpyHostToDevice, st));
    ICK(hipMemcpyAsync(k0->pf_items.p, k0->h_pf_items.p, sizeof(PfItemDev)*(size_t)total, hipMemcpyHostToDevice, st));
    ICK(hipMemcpyAsync(k0->pf_seq.p, k0->h_pf_seq.p, sizeof(int)*(size_t)(total + 1), hipMemcpyHostToDevice, st));
    ICK(hipMemcpyAsync(k0->pf_state.p, k0->h_pf_state.p, sizeof(mcp_pf_state)*(size_t)total, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(k_patch_sequences, dim3(total), dim3(64), 0, st, (int)MCP_PF_TRACK, (const PfTargetDev*)k0->pf_tab.p, total, (const int*)k0->pf_seq.p,
                     (const PfItemDev*)k0->pf_items.p, k0->pf_state.p, range, subpix_its, exhaustive, k0->bt_out.p, host_out, host_state, pose_pts);
  ICK(hipGetLastError());
  return 0;
}
int mcp_track_frame(int ncam, mcp_kf* const* targets, const uint8_t* const* imgs, const int* strides, int imgs_on_device,
                    const uint8_t* const* const* masks, const mcp_camera* cams, double bfw[12], const double* cfb, const int* n,
                    const mcp_td_in* const* in, const int* const* point_key, mcp_pf_state* const* state, int range, int subpix_its, int exhaustive,
                    int n_iter, const uint8_t* nonlinear, const double* override_sigma, int est, mcp_td_out* const* out, mcp_pose_point* pts_out,
                    double mu_last[6], double* weights_last) {
  if (!mu_last) return img_fail("mcp_track_frame: bad arguments");
  for (int k = 0; k < 6; ++k) mu_last[k] = 0;
  const bool view = (out == nullptr);       // results stay in the library's pinned block: mcp_track_frame_view (include/mcp_img.h)
  if (ncam < 1 || ncam > MCP_MAX_FRAME_CAMS || !targets || !cams || !bfw || !cfb || !n || !in || n_iter < 0 || !est_ok(est) ||
      (n_iter > 0 && (!nonlinear || !override_sigma)) || (imgs && !strides) || (state && !point_key)) return img_fail("mcp_track_frame: bad arguments");
  for (int c = 0; c < ncam; ++c) if (!targets[c] || !cam_ok(&cams[c]) || n[c] < 0 || (n[c] > 0 && !view && !out[c])) return img_fail("mcp_track_frame: bad arguments");
  // every per-point argument is checked BEFORE the first enqueue: an input rejected later would leave launches and copies in flight
  for (int c = 0; c < ncam; ++c) {
    if (targets[c]->device != targets[0]->device || (n[c] > 0 && (!in[c] || (state && (!state[c] || !point_key[c]))))) return img_fail("mcp_track_frame: bad arguments");
    for (int i = 0; i < n[c]; ++i) if (!in[c][i].source_kf || in[c][i].source_level < 0 || in[c][i].source_level >= MCP_LEVELS) return img_fail("mcp_track_frame: point without a resident source keyframe");
  }
  mcp_kf* k0 = targets[0];
  hipStream_t st = k0->st;
  // ... and whatever fails after it waits for the stream before the stack variables the copies write to (back, prm_err) go away
  struct DrainOnError { hipStream_t st; bool armed; int ncam; mcp_kf* const* targets; bool lite; ~DrainOnError() { if (armed) { (void)hipStreamSynchronize(st); if (lite) (void)lite_batch_finish(ncam, targets); (void)hipGetLastError(); } } };
  DrainOnError drain{st, true, ncam, targets, imgs != nullptr};
  int total = 0, maxn = 0;
  bool rode = false;              // the search's inputs went to the device inside k_row_count's launch
  // the search's tables and points are packed on the host while the pyramids run, and ride to the device in the next launch
  struct Ctx { int ncam; mcp_kf* const* targets; const mcp_camera* cams; const double* bfw; const double* cfb; const int* n; const mcp_td_in* const* in; const int* const* point_key;
               mcp_pf_state* const* state; mcp_td_out* const* out; int* total; int* maxn; };
  Ctx ctx{ncam, targets, cams, bfw, cfb, n, in, point_key, state, out, &total, &maxn};
  auto words = [](size_t bytes) { return (int)((bytes + 7)/8); };
  if (imgs) {
    FrameRide ride{nullptr, &ctx};
    if (!state) ride.fn = [](void* c_, FrameBatch& B) -> int {
      Ctx& c = *static_cast<Ctx*>(c_);
      if (search_batch_pack(c.ncam, c.targets, c.cams, c.cfb, c.n, c.in, c.out, c.total, c.maxn, c.out == nullptr)) return -1;
      if (*c.total == 0) return 0;
      mcp_kf* k0 = c.targets[0];
      B.up_src[0] = reinterpret_cast<const unsigned long long*>(k0->h_stab.p); B.up_dst[0] = reinterpret_cast<unsigned long long*>(k0->stab.p); B.up_n8[0] = (int)(sizeof(SearchCam)*(size_t)c.ncam/8);
      B.up_src[1] = reinterpret_cast<const unsigned long long*>(k0->h_bt_in.p); B.up_dst[1] = reinterpret_cast<unsigned long long*>(k0->bt_in.p); B.up_n8[1] = (int)(sizeof(DevTdIn)*(size_t)*c.total/8);
      return 0; };
    else ride.fn = [](void* c_, FrameBatch& B) -> int {
      Ctx& c = *static_cast<Ctx*>(c_);
      if (track_sequences_pack(c.ncam, c.targets, c.cams, c.bfw, c.cfb, c.n, c.in, c.point_key, c.state, c.total)) return -1;
      if (*c.total == 0) return 0;
      mcp_kf* k0 = c.targets[0]; const size_t t = (size_t)*c.total;
      const void* src[4] = { k0->h_pf_tab.p, k0->h_pf_items.p, k0->h_pf_seq.p, k0->h_pf_state.p };
      void* dst[4] = { k0->pf_tab.p, k0->pf_items.p, k0->pf_seq.p, k0->pf_state.p };
      const size_t bytes[4] = { sizeof(PfTargetDev)*(size_t)c.ncam, sizeof(PfItemDev)*t, sizeof(int)*(t + 1), sizeof(mcp_pf_state)*t };
      for (int r = 0; r < 4; ++r) { B.up_src[r] = static_cast<const unsigned long long*>(src[r]); B.up_dst[r] = static_cast<unsigned long long*>(dst[r]); B.up_n8[r] = (int)((bytes[r] + 7)/8); }
      return 0; };
    if (lite_batch_enqueue(ncam, targets, imgs, strides, imgs_on_device, masks, &ride)) return -1;
    rode = true;
  }
  (void)words;
  ICK(hipSetDevice(k0->device));
  RefineScratch& rs = refine_scratch();
  bool out_pinned = false;        // the search kernel wrote the TrackerData results (and the finder states) to pinned host memory as well
  if (state) {
    if (!rode && track_sequences_pack(ncam, targets, cams, bfw, cfb, n, in, point_key, state, &total)) return -1;
    if (total > 0) {
      if (k0->h_bt_out.alloc(total) || rs.dp.alloc(total)) return -1;
      if (track_sequences_launch(ncam, k0, total, range, subpix_its, exhaustive, rode, k0->h_bt_out.p, k0->h_pf_state_out.p, rs.dp.p)) return -1;
      out_pinned = true;
    }
  } else {
    if (!rode && search_batch_pack(ncam, targets, cams, cfb, n, in, out, &total, &maxn, view)) return -1;
    if (total > 0) {
      // the search leaves its results in pinned host memory too and writes the pose iterations' records itself (no packing launch)
      if (k0->h_bt_out.alloc(total) || rs.dp.alloc(total)) return -1;
      if (search_batch_launch(ncam, k0, total, maxn, bfw, range, subpix_its, exhaustive, rode, k0->h_bt_out.p, rs.dp.p)) return -1;
      out_pinned = true;
    }
  }
  unsigned int prm_err = 0;
  double back[18];
  const bool iterate = total > 0 && n_iter > 0;
  if (total > 0) {
    if (iterate && refine_enqueue(total, nullptr, ncam, cams, cfb, bfw, n_iter, nonlinear, override_sigma, est, st, &prm_err)) return -1;
    if (!out_pinned && search_batch_copy_out(ncam, k0, n, out, total)) return -1;
    if (pts_out) ICK(hipMemcpyAsync(pts_out, rs.dp.p, sizeof(mcp_pose_point)*(size_t)total, hipMemcpyDeviceToHost, st));
    if (iterate) { if (refine_results_enqueue(rs, total, back, weights_last, st)) return -1; }
    else if (weights_last) std::memset(weights_last, 0, 8*(size_t)total);
  }
  ICK(hipStreamSynchronize(st));
  drain.armed = false;
  if (imgs && lite_batch_finish(ncam, targets)) return -1;
  { int first = 0; for (int c = 0; c < ncam; ++c) { k0->view_first[c] = first; first += n[c]; } k0->view_first[ncam] = first; k0->view_ncam = (out_pinned || total == 0) ? ncam : 0; }
  if (out_pinned && !view) { int first = 0; for (int c = 0; c < ncam; ++c) { if (n[c]) std::memcpy(out[c], k0->h_bt_out.p + first, sizeof(mcp_td_out)*(size_t)n[c]); first += n[c]; } }
  if (view && !out_pinned && total > 0) return img_fail("mcp_track_frame: results were not written to the pinned block (view mode needs a search of at least one point)");
  if (iterate) refine_results_finish(rs, total, back, weights_last);
  if (state && total > 0) { int first = 0; for (int c = 0; c < ncam; ++c) { if (n[c]) std::memcpy(state[c], k0->h_pf_state_out.p + first, sizeof(mcp_pf_state)*(size_t)n[c]); first += n[c]; } }
  if (iterate && (prm_err || (rs.last_multi && getenv("MCP_TRACK_TEST_PRM_GIVEUP")))) {
    // a workgroup of the multi-workgroup iterations gave up waiting for the others: the frame is not lost, one workgroup redoes them
    if (refine_redo_single(total, n_iter, ncam, est, st)) return -1;
    if (pts_out) ICK(hipMemcpyAsync(pts_out, rs.dp.p, sizeof(mcp_pose_point)*(size_t)total, hipMemcpyDeviceToHost, st));
    if (refine_results_enqueue(rs, total, back, weights_last, st)) return -1;
    ICK(hipStreamSynchronize(st));
  }
  if (iterate) { std::memcpy(bfw, back, 96); std::memcpy(mu_last, back + 12, 48); }
  return 0;
}

// the results of camera `cam` of the last mcp_track_frame on this first target, in the library's pinned block (include/mcp_img.h)
const mcp_td_out* mcp_track_frame_view(const mcp_kf* first_target, int cam, int* count) {
  if (count) *count = 0;
  if (!first_target || cam < 0 || cam >= first_target->view_ncam) { img_fail("mcp_track_frame_view: no frame's results for that camera"); return nullptr; }
  const int first = first_target->view_first[cam], m = first_target->view_first[cam + 1] - first;
  if (count) *count = m;
  return m > 0 ? first_target->h_bt_out.p + first : nullptr;
}

// PatchFinder with its members carried from call to call (include/mcp_img.h MCP_PF_*): sequences of items, one finder each
int mcp_patch_sequences(int mode, int n_targets, const mcp_pf_target* targets, int n_seq, const int* seq_start, const mcp_pf_item* items,
                        mcp_pf_state* state, int range, int subpix_its, int exhaustive, mcp_td_out* out) {
  if (mode < MCP_PF_TRACK || mode > MCP_PF_EPI_REFINE || n_targets < 1 || !targets || n_seq < 0 || !seq_start || !state) return img_fail("mcp_patch_sequences: bad arguments");
  if (n_seq == 0) return 0;
  const int total = seq_start[n_seq];
  if (seq_start[0] != 0 || total < 0 || (total > 0 && (!items || !out))) return img_fail("mcp_patch_sequences: bad sequence table");
  for (int q = 0; q < n_seq; ++q) if (seq_start[q + 1] < seq_start[q]) return img_fail("mcp_patch_sequences: bad sequence table");
  mcp_kf* k0 = targets[0].kf;
  if (!k0) return img_fail("mcp_patch_sequences: target without a keyframe");
  std::vector<PfTargetDev> tab(n_targets);
  for (int t = 0; t < n_targets; ++t) {
    const mcp_pf_target& G = targets[t];
    if (!G.kf || !cam_ok(G.cam) || G.kf->device != k0->device) return img_fail("mcp_patch_sequences: bad target");
    PfTargetDev& D = tab[t];
    D.T = G.kf->view(); D.mask0 = G.kf->lev[0].has_mask ? G.kf->lev[0].mask.p : nullptr; D.cam = *G.cam;
    std::memcpy(D.bfw.R, G.base_from_world, 72); std::memcpy(D.bfw.t, G.base_from_world + 9, 24);
    std::memcpy(D.cfb.R, G.cam_from_base, 72); std::memcpy(D.cfb.t, G.cam_from_base + 9, 24);
  }
  std::vector<PfItemDev> h(std::max(total, 1));
  for (int i = 0; i < total; ++i) {
    const mcp_pf_item& I = items[i]; const mcp_td_in& p = I.point; PfItemDev& d = h[i];
    if (I.target < 0 || I.target >= n_targets) return img_fail("mcp_patch_sequences: item with a bad target index");
    if (!p.source_kf || p.source_level < 0 || p.source_level >= MCP_LEVELS) return img_fail("mcp_patch_sequences: point without a resident source keyframe");
    std::memcpy(d.p.world_pos, p.world_pos, 24); std::memcpy(d.p.pixel_right_w, p.pixel_right_w, 24); std::memcpy(d.p.pixel_down_w, p.pixel_down_w, 24);
    const Level& Sl = p.source_kf->lev[p.source_level];
    d.p.src_img = Sl.img.p; d.p.src_w = Sl.w; d.p.src_h = Sl.h; d.p.center_x = p.center_x; d.p.center_y = p.center_y; d.p.fixed = p.fixed;
    d.point_key = I.point_key; d.target = I.target; d.start_x = I.start_pos[0]; d.start_y = I.start_pos[1];
  }
  ICK(hipSetDevice(k0->device));
  hipStream_t st = k0->st;
  if (k0->pf_tab.alloc(n_targets) || k0->pf_items.alloc(std::max(total, 1)) || k0->pf_seq.alloc(n_seq + 1) || k0->pf_state.alloc(n_seq) || k0->bt_out.alloc(std::max(total, 1))) return -1;
  ICK(hipMemcpyAsync(k0->pf_tab.p, tab.data(), sizeof(PfTargetDev)*(size_t)n_targets, hipMemcpyHostToDevice, st));
  if (total) ICK(hipMemcpyAsync(k0->pf_items.p, h.data(), sizeof(PfItemDev)*(size_t)total, hipMemcpyHostToDevice, st));
  ICK(hipMemcpyAsync(k0->pf_seq.p, seq_start, sizeof(int)*(size_t)(n_seq + 1), hipMemcpyHostToDevice, st));
  ICK(hipMemcpyAsync(k0->pf_state.p, state, sizeof(mcp_pf_state)*(size_t)n_seq, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_patch_sequences, dim3(n_seq), dim3(64), 0, st, mode, (const PfTargetDev*)k0->pf_tab.p, n_seq, (const int*)k0->pf_seq.p,
                     (const PfItemDev*)k0->pf_items.p, k0->pf_state.p, range, subpix_its, exhaustive, k0->bt_out.p);
  ICK(hipGetLastError());
  if (total) ICK(hipMemcpyAsync(out, k0->bt_out.p, sizeof(mcp_td_out)*(size_t)total, hipMemcpyDeviceToHost, st));
  ICK(hipMemcpyAsync(state, k0->pf_state.p, sizeof(mcp_pf_state)*(size_t)n_seq, hipMemcpyDeviceToHost, st));
  ICK(hipStreamSynchronize(st));
  return 0;
}

int mcp_track_pose_update(int n, const uint8_t* found, const double* fpos, const double* ipos, const double* sinv, const double* J,
                          double override_sigma, double mu[6], double* wout, double* sigma_out) {
  return mcp_track_pose_update_m(n, found, fpos, ipos, sinv, J, override_sigma, mu, wout, sigma_out, MCP_MEST_TUKEY);
}
int mcp_track_pose_update_m(int n, const uint8_t* found, const double* fpos, const double* ipos, const double* sinv, const double* J,
                          double override_sigma, double mu[6], double* wout, double* sigma_out, int est) {
  if (!est_ok(est)) return img_fail("mcp_track_pose_update: unknown M-estimator");
  for (int k = 0; k < 6; ++k) mu[k] = 0;
  if (sigma_out) *sigma_out = 0;
  if (n <= 0) return 0;
  int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return img_fail("mcp_track_pose_update: no HIP device");
  std::vector<int> slot(n, 0); int ne = 0;
  for (int i = 0; i < n; ++i) { slot[i] = ne; if (found[i]) ++ne; }
  if (wout) std::memset(wout, 0, sizeof(double)*(size_t)n);
  if (ne == 0) return 0;
  // scratch kept across calls (the tracker calls this ~20 times per frame)
  struct PoseScratch { Buf<uint8_t> dfound; Buf<double> dfp, dip, dsi, dJ, dex, de2, dsig, dmu, dw, dhist; Buf<int> dslot; Buf<SelState> dst; };
  static thread_local PoseScratch ps;
  Buf<uint8_t>& dfound = ps.dfound; Buf<double>& dfp = ps.dfp; Buf<double>& dip = ps.dip; Buf<double>& dsi = ps.dsi; Buf<double>& dJ = ps.dJ;
  Buf<double>& dex = ps.dex; Buf<double>& de2 = ps.de2; Buf<double>& dsig = ps.dsig; Buf<double>& dmu = ps.dmu; Buf<double>& dw = ps.dw;
  Buf<double>& dhist = ps.dhist; Buf<int>& dslot = ps.dslot; Buf<SelState>& dst = ps.dst;
  if (dfound.alloc(n) || dfp.alloc(2*(size_t)n) || dip.alloc(2*(size_t)n) || dsi.alloc(n) || dJ.alloc(12*(size_t)n) || dex.alloc(2*(size_t)n) ||
      de2.alloc(ne) || dsig.alloc(4) || dmu.alloc(8) || dw.alloc(n) || dhist.alloc((size_t)SEL_PASSES*SEL_BINS) || dslot.alloc(n) || dst.alloc(SEL_PASSES + 1)) return -1;
  ICK(hipMemcpy(dfound.p, found, (size_t)n, hipMemcpyHostToDevice));
  ICK(hipMemcpy(dfp.p, fpos, 16*(size_t)n, hipMemcpyHostToDevice)); ICK(hipMemcpy(dip.p, ipos, 16*(size_t)n, hipMemcpyHostToDevice));
  ICK(hipMemcpy(dsi.p, sinv, 8*(size_t)n, hipMemcpyHostToDevice)); ICK(hipMemcpy(dJ.p, J, 96*(size_t)n, hipMemcpyHostToDevice));
  ICK(hipMemcpy(dslot.p, slot.data(), sizeof(int)*(size_t)n, hipMemcpyHostToDevice));
  hipStream_t st = nullptr;
  hipLaunchKernelGGL(k_pose_errors, dim3((n + 255)/256), dim3(256), 0, st, n, (const uint8_t*)dfound.p, (const double*)dfp.p, (const double*)dip.p, (const double*)dsi.p, dex.p, de2.p, (const int*)dslot.p);
  if (!(override_sigma > 0)) {         // Tukey::FindSigmaSquared: exact median of the squared errors
    ICK(hipMemsetAsync(dhist.p, 0, (size_t)SEL_PASSES*SEL_BINS*sizeof(double), st));
    const int grid = std::max(1, std::min(256, (ne + SEL_BLOCK*4 - 1)/(SEL_BLOCK*4)));
    for (int p = 0; p < SEL_PASSES; ++p) hipLaunchKernelGGL(k_select_pass, dim3(grid), dim3(SEL_BLOCK), 0, st, p, ne, (const double*)de2.p, dhist.p, dst.p, (unsigned long long)(ne/2));
    hipLaunchKernelGGL(k_select_final, dim3(1), dim3(SEL_BLOCK), 0, st, (const double*)dhist.p, (const SelState*)dst.p, dsig.p + 1);
  }
  hipLaunchKernelGGL(k_tukey_sigma, dim3(1), dim3(64), 0, st, (const double*)(dsig.p + 1), (double)ne, override_sigma, dsig.p, est);
  hipLaunchKernelGGL(k_pose_solve, dim3(1), dim3(256), 0, st, n, (const uint8_t*)dfound.p, (const double*)dex.p, (const double*)dsi.p, (const double*)dJ.p, (const double*)dsig.p, dmu.p, dw.p, est);
  ICK(hipDeviceSynchronize());
  ICK(hipMemcpy(mu, dmu.p, 48, hipMemcpyDeviceToHost));
  if (wout) ICK(hipMemcpy(wout, dw.p, 8*(size_t)n, hipMemcpyDeviceToHost));
  if (sigma_out) ICK(hipMemcpy(sigma_out, dsig.p, 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
