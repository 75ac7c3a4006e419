/*
 * img_oracle.c -- CPU ORACLE for the KeyFrame / Tracker image path (test infrastructure only).
 * PARITY UNPINNED (see img_oracle.h).  Every function cites the reference lines it follows;
 * libCVD / OpenCV / TooN semantics are restated from their published algorithms [3P-memory].
 */
#include "img_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MIN_FAST_THRESH 5      /* include/mcptam/KeyFrame.h:88 */
#define MAX_FAST_THRESH 30     /* :89 */
#define NUM_PREV 2              /* Level::snNumPrev, src/KeyFrame.cc:71 */

typedef struct { int w, h; uint8_t* img; uint8_t* mask; int ncorners, ccorners; orc_int2* corners; int* lut;
                 int thresh; double freq[MAX_FAST_THRESH + 1];
                 int ncand; orc_int2* cand; double* cand_score;
                 /* Level::imagePrev / vCornersPrev: circular buffers of Level::snNumPrev = 2 (KeyFrame.cc:71, KeyFrame.h:124-125,147-148); [0] = oldest */
                 int nprev; uint8_t* pimg[NUM_PREV]; orc_int2* pcorners[NUM_PREV]; int pncorners[NUM_PREV]; } olevel;
#define SBI_N (ORC_SBI_W*ORC_SBI_H)
typedef struct { uint8_t small[SBI_N]; float templ[SBI_N]; float jacs[2*SBI_N]; } osbi;
struct orc_kf { olevel lev[ORC_LEVELS]; int adaptive, glare, pavgb, has_image; osbi* sbi; };

orc_kf* orc_kf_create(int w, int h, int adaptive, int glare, int pavgb) {
  orc_kf* k = (orc_kf*)calloc(1, sizeof *k);
  k->adaptive = adaptive; k->glare = glare; k->pavgb = pavgb;
  for (int l = 0; l < ORC_LEVELS; l++) {
    olevel* L = &k->lev[l];
    L->w = w >> l; L->h = h >> l;               /* size / 2 per level, KeyFrame.cc:189 */
    L->img = (uint8_t*)calloc((size_t)L->w*L->h + 1, 1);
    L->mask = (uint8_t*)malloc((size_t)L->w*L->h + 1);
    L->lut = (int*)calloc(L->h + 1, sizeof(int));
  }
  return k;
}
void orc_kf_destroy(orc_kf* k) {
  if (!k) return;
  for (int l = 0; l < ORC_LEVELS; l++) { olevel* L = &k->lev[l]; free(L->img); free(L->mask); free(L->corners); free(L->lut); free(L->cand); free(L->cand_score);
    for (int j = 0; j < NUM_PREV; j++) { free(L->pimg[j]); free(L->pcorners[j]); } }
  free(k->sbi);
  free(k);
}

/* CVD::halfSample [3P-memory]: generic template = truncating mean of the 2x2 block; the SSE2 byte
 * path = cascaded pavgb (round-half-up vertical average, then horizontal). */
static void half_sample(const uint8_t* in, int iw, int ih, uint8_t* out, int ow, int oh, int pavgb) {
  (void)ih;
  for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++) {
    const int a = in[(2*y)*iw + 2*x], b = in[(2*y)*iw + 2*x + 1], c = in[(2*y + 1)*iw + 2*x], d = in[(2*y + 1)*iw + 2*x + 1];
    if (pavgb) { const int v0 = (a + c + 1) >> 1, v1 = (b + d + 1) >> 1; out[y*ow + x] = (uint8_t)((v0 + v1 + 1) >> 1); }
    else out[y*ow + x] = (uint8_t)((a + b + c + d)/4);
  }
}

/* libCVD fast_pixel_ring [3P-memory]: Bresenham circle of radius 3 */
static const int RING_X[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int RING_Y[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

/* FAST-10 segment test: >= 10 contiguous ring pixels all > p+b or all < p-b */
int orc_fast10_is_corner(const uint8_t* p, int stride, int b) {
  const int cb = *p + b, c_b = *p - b;
  int bright = 0, dark = 0;
  for (int k = 0; k < 16; k++) {
    const int v = p[RING_Y[k]*stride + RING_X[k]];
    if (v > cb) bright |= 1 << k;
    if (v < c_b) dark |= 1 << k;
  }
  for (int pass = 0; pass < 2; pass++) {
    const int m = pass ? dark : bright;
    for (int s = 0; s < 16; s++) {
      int ok = 1;
      for (int j = 0; j < 10; j++) if (!(m & (1 << ((s + j) & 15)))) { ok = 0; break; }
      if (ok) return 1;
    }
  }
  return 0;
}
/* CVD::fast_corner_score_10 [3P-memory]: binary search for the largest threshold that still
 * passes the segment test */
int orc_fast10_score(const uint8_t* p, int stride, int bstart) {
  int bmin = bstart, bmax = 255, b = (bmax + bmin)/2;
  for (;;) {
    if (orc_fast10_is_corner(p, stride, b)) bmin = b; else bmax = b;
    if (bmin == bmax - 1 || bmin == bmax) return bmin;
    b = (bmin + bmax)/2;
  }
}
/* the classic FAST corner_score used by older fast_nonmax [3P-memory]: ring SAD above the barrier */
int orc_fast_ring_sad_score(const uint8_t* p, int stride, int barrier) {
  const int cb = *p + barrier, c_b = *p - barrier;
  int sp = 0, sn = 0;
  for (int k = 0; k < 16; k++) {
    const int v = p[RING_Y[k]*stride + RING_X[k]];
    if (v > cb) sp += v - cb; else if (v < c_b) sn += c_b - v;
  }
  return sp > sn ? sp : sn;
}
/* CVD::fast_corner_detect_10 [3P-memory]: raster scan of y in [3,h-3), x in [3,w-3) */
static void fast_detect(olevel* L, int b) {
  L->ncorners = 0;
  for (int y = 3; y < L->h - 3; y++) for (int x = 3; x < L->w - 3; x++)
    if (orc_fast10_is_corner(L->img + (size_t)y*L->w + x, L->w, b)) {
      if (L->ncorners >= L->ccorners) { L->ccorners = L->ccorners ? 2*L->ccorners : 4096; L->corners = (orc_int2*)realloc(L->corners, sizeof(orc_int2)*L->ccorners); }
      L->corners[L->ncorners].x = x; L->corners[L->ncorners].y = y; L->ncorners++;
    }
}

/* OpenCV dilate with the 5x5 MORPH_ELLIPSE element, constant border = -inf [3P-memory] (KeyFrame.cc:217) */
static void dilate5(const uint8_t* in, uint8_t* out, int w, int h) {
  static const int EL[5][5] = { {0,0,1,0,0}, {1,1,1,1,1}, {1,1,1,1,1}, {1,1,1,1,1}, {0,0,1,0,0} };
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    int m = 0;
    for (int dy = -2; dy <= 2; dy++) for (int dx = -2; dx <= 2; dx++) {
      if (!EL[dy + 2][dx + 2]) continue;
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      if (in[yy*w + xx] > m) m = in[yy*w + xx];
    }
    out[y*w + x] = (uint8_t)m;
  }
}

/* KeyFrame::MakeKeyFrame_Lite, KeyFrame.cc:145-360 */
int orc_kf_num_prev(orc_kf* k) { return k->lev[0].nprev; }
int orc_kf_make_lite(orc_kf* k, const uint8_t* img, int stride, const uint8_t* const* masks) {
  /* the image and corners currently held move into the history buffers before they are overwritten, KeyFrame.cc:152-199 */
  if (k->has_image) for (int l = 0; l < ORC_LEVELS; l++) {
    olevel* L = &k->lev[l];
    if (L->nprev == NUM_PREV) {              /* circular_buffer::push_back on a full buffer drops the oldest */
      free(L->pimg[0]); free(L->pcorners[0]);
      for (int j = 1; j < NUM_PREV; j++) { L->pimg[j-1] = L->pimg[j]; L->pcorners[j-1] = L->pcorners[j]; L->pncorners[j-1] = L->pncorners[j]; }
      L->nprev--;
    }
    const size_t npx = (size_t)L->w*L->h;
    L->pimg[L->nprev] = (uint8_t*)malloc(npx + 1); memcpy(L->pimg[L->nprev], L->img, npx);
    L->pcorners[L->nprev] = (orc_int2*)malloc(sizeof(orc_int2)*(L->ncorners + 1)); memcpy(L->pcorners[L->nprev], L->corners, sizeof(orc_int2)*L->ncorners);
    L->pncorners[L->nprev] = L->ncorners;
    L->nprev++;
  }
  k->has_image = 1;
  for (int l = 0; l < ORC_LEVELS; l++) {
    olevel* L = &k->lev[l];
    if (l == 0) { for (int y = 0; y < L->h; y++) memcpy(L->img + (size_t)y*L->w, img + (size_t)y*stride, L->w); }
    else half_sample(k->lev[l-1].img, k->lev[l-1].w, k->lev[l-1].h, L->img, L->w, L->h, k->pavgb);     /* :189-190 */
    const size_t npx = (size_t)L->w*L->h;
    /* mask, :214-243 */
    const uint8_t* internal = masks ? masks[l] : NULL;
    if (k->glare) {
      uint8_t* a = (uint8_t*)malloc(npx + 1); uint8_t* b = (uint8_t*)malloc(npx + 1);
      memcpy(a, L->img, npx);
      for (int it = 0; it < 5; it++) { dilate5(a, b, L->w, L->h); uint8_t* t = a; a = b; b = t; }      /* 5 iterations, :217 */
      for (size_t i = 0; i < npx; i++) { const uint8_t g = a[i] > 245 ? 0 : 255; L->mask[i] = internal ? (internal[i] & g) : g; }   /* :218, :227 */
      free(a); free(b);
    } else if (internal) memcpy(L->mask, internal, npx);
    else memset(L->mask, 255, npx);
    memset(L->freq, 0, sizeof L->freq); L->thresh = 0;
    if (k->adaptive) {
      fast_detect(L, MIN_FAST_THRESH);                                                          /* :259 */
      int* scores = (int*)malloc(sizeof(int)*(L->ncorners + 1));
      for (int j = 0; j < L->ncorners; j++) scores[j] = orc_fast10_score(L->img + (size_t)L->corners[j].y*L->w + L->corners[j].x, L->w, MIN_FAST_THRESH);   /* :262 */
      for (int j = 0; j < L->ncorners; j++)                                                       /* :264-275 */
        for (int t = MIN_FAST_THRESH; t <= MAX_FAST_THRESH; ++t) { if (scores[j] >= t) L->freq[t]++; if (scores[j] == t) break; }
      const double targetDeriv = -1*(L->w*L->h)/500.0;                                          /* :279 */
      L->thresh = MIN_FAST_THRESH;
      for (int t = MIN_FAST_THRESH; t <= MAX_FAST_THRESH; ++t) {                                 /* :283-300 */
        double deriv;
        if (t == MIN_FAST_THRESH) deriv = L->freq[t+1] - L->freq[t];
        else if (t == MAX_FAST_THRESH) deriv = L->freq[t] - L->freq[t-1];
        else deriv = (L->freq[t+1] - L->freq[t-1])/2.0;
        L->thresh = t;
        if (deriv > targetDeriv) break;
      }
      int n = 0;
      for (int j = 0; j < L->ncorners; j++) {                                                    /* :302-315 */
        if (L->mask[(size_t)L->corners[j].y*L->w + L->corners[j].x] < 255) continue;
        if (scores[j] < L->thresh) continue;
        L->corners[n++] = L->corners[j];
      }
      L->ncorners = n;
      free(scores);
    } else {                                                                                     /* :318-343 */
      static const int fixed_t[4] = { 10, 15, 15, 10 };
      fast_detect(L, fixed_t[l]); L->thresh = fixed_t[l];
    }
    unsigned v = 0;                                                                              /* :346-355 */
    for (int y = 0; y < L->h; y++) { while (v < (unsigned)L->ncorners && y > L->corners[v].y) v++; L->lut[y] = (int)v; }
  }
  return 0;
}
int orc_kf_level_size(orc_kf* k, int l, int* w, int* h) { *w = k->lev[l].w; *h = k->lev[l].h; return 0; }
const uint8_t* orc_kf_image(orc_kf* k, int l) { return k->lev[l].img; }
int orc_kf_num_corners(orc_kf* k, int l) { return k->lev[l].ncorners; }
const orc_int2* orc_kf_corners(orc_kf* k, int l) { return k->lev[l].corners; }
const int* orc_kf_row_lut(orc_kf* k, int l) { return k->lev[l].lut; }
int orc_kf_fast_thresh(orc_kf* k, int l) { return k->lev[l].thresh; }
const double* orc_kf_fast_frequency(orc_kf* k, int l) { return k->lev[l].freq; }

/* FindShiTomasiScoreAtPoint, ShiTomasi.cc:34-63 */
double orc_shi_tomasi(const uint8_t* img, int stride, int half, int cx, int cy) {
  double dXX = 0, dYY = 0, dXY = 0;
  for (int y = cy - half; y <= cy + half; y++) for (int x = cx - half; x <= cx + half; x++) {
    const double dx = img[y*stride + x + 1] - img[y*stride + x - 1];
    const double dy = img[(y + 1)*stride + x] - img[(y - 1)*stride + x];
    dXX += dx*dx; dYY += dy*dy; dXY += dx*dy;
  }
  const int nPixels = (2*half + 1)*(2*half + 1);
  dXX = dXX/(2.0*nPixels); dYY = dYY/(2.0*nPixels); dXY = dXY/(2.0*nPixels);
  return 0.5*(dXX + dYY - sqrt((dXX + dYY)*(dXX + dYY) - 4*(dXX*dYY - dXY*dXY)));
}

static int in_border(const olevel* L, int x, int y, int b) { return x >= b && y >= b && x < L->w - b && y < L->h - b; }

/* CVD::fast_nonmax -> nonmax_suppression [3P-memory]: keep a corner iff none of its 8 neighbours that are
 * corners has a strictly greater score; raster order preserved */
static int nonmax(const olevel* L, const int* score, orc_int2* out) {
  int n = 0;
  /* index image for neighbour lookup */
  int* idx = (int*)malloc(sizeof(int)*(size_t)L->w*L->h);
  for (size_t i = 0; i < (size_t)L->w*L->h; i++) idx[i] = -1;
  for (int i = 0; i < L->ncorners; i++) idx[(size_t)L->corners[i].y*L->w + L->corners[i].x] = i;
  for (int i = 0; i < L->ncorners; i++) {
    const int x = L->corners[i].x, y = L->corners[i].y; int keep = 1;
    for (int dy = -1; dy <= 1 && keep; dy++) for (int dx = -1; dx <= 1; dx++) {
      if (!dx && !dy) continue;
      const int xx = x + dx, yy = y + dy;
      if (xx < 0 || yy < 0 || xx >= L->w || yy >= L->h) continue;
      const int j = idx[(size_t)yy*L->w + xx];
      if (j >= 0 && score[j] > score[i]) { keep = 0; break; }
    }
    if (keep) out[n++] = L->corners[i];
  }
  free(idx);
  return n;
}
typedef struct { double s; orc_int2 p; } scored;
static int cmp_scored_desc(const void* a, const void* b) {       /* std::sort on reverse iterators of pair<double,ImageRef> */
  const scored* x = (const scored*)a; const scored* y = (const scored*)b;
  if (x->s != y->s) return (x->s < y->s) - (x->s > y->s);
  /* ImageRef operator< [3P-memory]: y first, then x */
  if (x->p.y != y->p.y) return (x->p.y < y->p.y) - (x->p.y > y->p.y);
  return (x->p.x < y->p.x) - (x->p.x > y->p.x);
}
/* KeyFrame::MakeKeyFrame_Rest, candidate part, KeyFrame.cc:363-450 */
static void prune_candidates(olevel* L);
int orc_kf_make_rest(orc_kf* k, int use_shi, int use_percent, double top_fraction, double thresh, int nonmax_score) {
  for (int l = 0; l < ORC_LEVELS; l++) {
    olevel* L = &k->lev[l];
    int* sc = (int*)malloc(sizeof(int)*(L->ncorners + 1));
    for (int i = 0; i < L->ncorners; i++) {
      const uint8_t* p = L->img + (size_t)L->corners[i].y*L->w + L->corners[i].x;
      sc[i] = nonmax_score ? orc_fast_ring_sad_score(p, L->w, L->thresh) : orc_fast10_score(p, L->w, L->thresh);
    }
    orc_int2* mx = (orc_int2*)malloc(sizeof(orc_int2)*(L->ncorners + 1));
    const int nm = nonmax(L, sc, mx);                                               /* :393 / :411 */
    scored* v = (scored*)malloc(sizeof(scored)*(nm + 1)); int nv = 0;
    for (int i = 0; i < nm; i++) {
      if (!in_border(L, mx[i].x, mx[i].y, 10)) continue;                             /* :402, :415 */
      const uint8_t* p = L->img + (size_t)mx[i].y*L->w + mx[i].x;
      v[nv].s = use_shi ? orc_shi_tomasi(L->img, L->w, 3, mx[i].x, mx[i].y) : (double)orc_fast10_score(p, L->w, L->thresh);   /* :396, :418 */
      v[nv].p = mx[i]; nv++;
    }
    free(L->cand); free(L->cand_score);
    L->cand = (orc_int2*)malloc(sizeof(orc_int2)*(nv + 1)); L->cand_score = (double*)malloc(sizeof(double)*(nv + 1)); L->ncand = 0;
    if (use_percent) {                                                               /* :424-438 */
      qsort(v, nv, sizeof(scored), cmp_scored_desc);
      const int num = (int)(nv*top_fraction);
      for (int i = 0; i < num && i < nv; i++) { L->cand[L->ncand] = v[i].p; L->cand_score[L->ncand++] = v[i].s; }
    } else {                                                                         /* :439-452 */
      for (int i = 0; i < nv; i++) if (v[i].s > thresh) { L->cand[L->ncand] = v[i].p; L->cand_score[L->ncand++] = v[i].s; }
    }
    free(sc); free(mx); free(v);
    prune_candidates(L);
  }
  return 0;
}
int orc_kf_num_candidates(orc_kf* k, int l) { return k->lev[l].ncand; }
int orc_kf_get_candidates(orc_kf* k, int l, orc_int2* pos, double* score, int cap) {
  const int n = k->lev[l].ncand < cap ? k->lev[l].ncand : cap;
  memcpy(pos, k->lev[l].cand, sizeof(orc_int2)*n); memcpy(score, k->lev[l].cand_score, sizeof(double)*n); return n;
}

/* MiniPatch::FindPatch without a row LUT (the form MakeKeyFrame_Rest uses, KeyFrame.cc:490,516): linear scan from the
 * first corner at or below the top of the box, MiniPatch.cc:77-83,93-113.  `patch` = 9x9 template.  Returns found. */
static int minipatch_scan(const uint8_t* patch, const uint8_t* dimg, int dw, int dh, const orc_int2* corners, int ncorners,
                          orc_int2* pos, int range) {
  const int H = 4, MAXSSD = 9999;
  int best = MAXSSD + 1; orc_int2 bp = *pos;
  const int tlx = pos->x - range, tly = pos->y - range, brx = pos->x + range, bry = pos->y + range;
  int c = 0;
  for (; c < ncorners; c++) if (corners[c].y >= tly) break;
  for (; c < ncorners; c++) {
    const orc_int2 p = corners[c];
    if (p.x < tlx || p.x > brx) continue;
    if (p.y > bry) break;
    int ssd;
    if (!(p.x >= H && p.y >= H && p.x < dw - H && p.y < dh - H)) ssd = MAXSSD + 1;
    else { ssd = 0; for (int r = 0; r < 9; r++) for (int q = 0; q < 9; q++) { const int df = dimg[(size_t)(p.y - H + r)*dw + p.x - H + q] - patch[9*r + q]; ssd += df*df; } }
    if (ssd < best) { bp = p; best = ssd; }
  }
  if (best < MAXSSD) { *pos = bp; return 1; }
  return 0;
}
static void minipatch_sample(const uint8_t* img, int w, orc_int2 p, uint8_t* patch) {
  for (int r = 0; r < 9; r++) memcpy(patch + 9*r, img + (size_t)(p.y - 4 + r)*w + p.x - 4, 9);
}
/* the stability pruning of MakeKeyFrame_Rest, KeyFrame.cc:456-527: follow each candidate back to the oldest stored frame
 * and forward again to the current one; keep it if it lands within sqrt(2) pixels of where it started */
static void prune_candidates(olevel* L) {
  if (L->nprev == 0) return;
  const int range = L->nprev*10;
  int nk = 0;
  for (int i = 0; i < L->ncand; i++) {
    const orc_int2 cur = L->cand[i];
    uint8_t patch[81];
    minipatch_sample(L->img, L->w, cur, patch);
    orc_int2 prev = cur;
    if (!minipatch_scan(patch, L->pimg[0], L->w, L->h, L->pcorners[0], L->pncorners[0], &prev, range)) continue;
    minipatch_sample(L->pimg[0], L->w, prev, patch);
    orc_int2 next = prev;
    if (!minipatch_scan(patch, L->img, L->w, L->h, L->corners, L->ncorners, &next, range)) continue;
    const int dx = next.x - cur.x, dy = next.y - cur.y;
    if (dx*dx + dy*dy > 2) continue;
    L->cand[nk] = L->cand[i]; L->cand_score[nk] = L->cand_score[i]; nk++;
  }
  L->ncand = nk;
}

/* MiniPatch::SampleFromImage + FindPatch + SSDAtPoint, MiniPatch.cc:34-122 (half size 4, max SSD 9999) */
int orc_minipatch_find(orc_kf* src, orc_kf* dst, int level, int n, const orc_int2* src_pos, const orc_int2* dst_pos,
                       int range, orc_int2* out_pos, uint8_t* out_found, int* out_ssd) {
  const olevel* S = &src->lev[level]; const olevel* D = &dst->lev[level];
  const int H = 4, MAXSSD = 9999;
  for (int i = 0; i < n; i++) {
    out_found[i] = 0; out_pos[i] = dst_pos[i]; if (out_ssd) out_ssd[i] = MAXSSD + 1;
    if (!in_border(S, src_pos[i].x, src_pos[i].y, H)) continue;                      /* assert in SampleFromImage */
    uint8_t patch[81];
    for (int r = 0; r < 9; r++) memcpy(patch + 9*r, S->img + (size_t)(src_pos[i].y - H + r)*S->w + src_pos[i].x - H, 9);
    int best = MAXSSD + 1; orc_int2 bp = dst_pos[i];
    const int tlx = dst_pos[i].x - range, tly = dst_pos[i].y - range, brx = dst_pos[i].x + range, bry = dst_pos[i].y + range;
    int top = tly; if (top < 0) top = 0; if (top >= D->h) top = D->h - 1;           /* :84-90 */
    for (int c = D->lut[top]; c < D->ncorners; c++) {                                /* :93-107 */
      const orc_int2 p = D->corners[c];
      if (p.x < tlx || p.x > brx) continue;
      if (p.y > bry) break;
      int ssd;
      if (!in_border(D, p.x, p.y, H)) ssd = MAXSSD + 1;
      else { ssd = 0; for (int r = 0; r < 9; r++) for (int q = 0; q < 9; q++) { const int df = D->img[(size_t)(p.y - H + r)*D->w + p.x - H + q] - patch[9*r + q]; ssd += df*df; } }
      if (ssd < best) { bp = p; best = ssd; }
    }
    if (out_ssd) out_ssd[i] = best;
    if (best < MAXSSD) { out_pos[i] = bp; out_found[i] = 1; }
  }
  return 0;
}

/* ------------------------------------------------------------------ tracker per-point path */
static void m3v(const double* A, const double* v, double* o) {
  const double a = A[0]*v[0] + A[1]*v[1] + A[2]*v[2], b = A[3]*v[0] + A[4]*v[1] + A[5]*v[2], c = A[6]*v[0] + A[7]*v[1] + A[8]*v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void gen_field(int i, const double* p, double* o) {
  o[0] = o[1] = o[2] = 0.0;
  if (i < 3) { o[i] = 1.0; return; }
  const int a = i - 3; o[(a+1)%3] = -p[(a+2)%3]; o[(a+2)%3] = p[(a+1)%3];
}
/* CVD::transform + sample [3P-memory]: incremental source position, bilinear sample in double,
 * truncating conversion to byte; returns the number of destination pixels that fell outside */
static int g_transform_round = 0;      /* oracle-only switch (sensitivity report): 1 = the byte conversion rounds half up instead of truncating */
void orc_img_set_variant(int key, int value) { if (key == 0) g_transform_round = value; }
static int cvd_transform8(const olevel* in, uint8_t* out /*8x8*/, const double M[4], double inx, double iny, double outx, double outy) {
  const int w = 8, h = 8, iw = in->w, ih = in->h;
  const double across[2] = { M[0], M[2] }, down[2] = { M[1], M[3] };
  double p0[2] = { inx - (M[0]*outx + M[1]*outy), iny - (M[2]*outx + M[3]*outy) };
  double min_x = p0[0], min_y = p0[1], max_x = min_x, max_y = min_y;
  if (across[0] < 0) min_x += w*across[0]; else max_x += w*across[0];
  if (down[0] < 0) min_x += h*down[0]; else max_x += h*down[0];
  if (across[1] < 0) min_y += w*across[1]; else max_y += w*across[1];
  if (down[1] < 0) min_y += h*down[1]; else max_y += h*down[1];
  const double cr[2] = { down[0] - w*across[0], down[1] - w*across[1] };
  const int inside = (min_x >= 0 && min_y >= 0 && max_x < iw - 1 && max_y < ih - 1);
  const double xb = iw - 1, yb = ih - 1;
  int count = 0;
  double p[2] = { p0[0], p0[1] };
  for (int i = 0; i < h; ++i, p[0] += cr[0], p[1] += cr[1])
    for (int j = 0; j < w; ++j, p[0] += across[0], p[1] += across[1]) {
      if (inside || (0 <= p[0] && 0 <= p[1] && p[0] < xb && p[1] < yb)) {
        const int lx = (int)p[0], ly = (int)p[1];
        const double x = p[0] - lx, y = p[1] - ly;
        const uint8_t* q = in->img + (size_t)ly*iw + lx;
        const double v = (1 - y)*((1 - x)*q[0] + x*q[1]) + y*((1 - x)*q[iw] + x*q[iw + 1]);
        out[i*8 + j] = g_transform_round ? (uint8_t)(v + 0.5) : (uint8_t)v;
      } else { out[i*8 + j] = 0; ++count; }
    }
  return count;
}
/* PatchFinder::ZMSSDAtPoint scalar branch, PatchFinder.cc:511-664 */
static int zmssd(const olevel* L, const uint8_t* T, int tsum, int tsumsq, int x, int y) {
  const int MAXSSD = 8*8*250;
  if (!in_border(L, x, y, 4)) return MAXSSD + 1;
  int isum = 0, isumsq = 0, cross = 0;
  for (int r = 0; r < 8; r++) { const uint8_t* ip = L->img + (size_t)(y - 4 + r)*L->w + x - 4; for (int c = 0; c < 8; c++) { const int n = ip[c]; isum += n; isumsq += n*n; cross += n*T[8*r + c]; } }
  const int SA = tsum, SB = isum, N = 64;
  return ((2*SA*SB - SA*SA - SB*SB)/N + isumsq + tsumsq - 2*cross);
}
static int inv3(const double* A, double* I) {
  const double c00 = A[4]*A[8] - A[5]*A[7], c01 = A[5]*A[6] - A[3]*A[8], c02 = A[3]*A[7] - A[4]*A[6];
  const double det = A[0]*c00 + A[1]*c01 + A[2]*c02;
  const double id = 1.0/det;
  I[0] = c00*id; I[1] = (A[2]*A[7] - A[1]*A[8])*id; I[2] = (A[1]*A[5] - A[2]*A[4])*id;
  I[3] = c01*id; I[4] = (A[0]*A[8] - A[2]*A[6])*id; I[5] = (A[2]*A[3] - A[0]*A[5])*id;
  I[6] = c02*id; I[7] = (A[1]*A[6] - A[0]*A[7])*id; I[8] = (A[0]*A[4] - A[1]*A[3])*id;
  return det != 0;
}

int orc_track_search(orc_kf* target, const orc_camera* cam, const double bfw[12], const double cfb[12], int n,
                     const orc_td_in* in, int range, int subpix_its, int exhaustive, orc_td_out* out) {
  /* CamFromWorld = CamFromBase * BaseFromWorld */
  double Rcw[9], tcw[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3*i+j] = cfb[3*i]*bfw[j] + cfb[3*i+1]*bfw[3+j] + cfb[3*i+2]*bfw[6+j];
  m3v(cfb, bfw + 9, tcw); tcw[0] += cfb[9]; tcw[1] += cfb[10]; tcw[2] += cfb[11];
  for (int i = 0; i < n; i++) {
    const orc_td_in* p = &in[i]; orc_td_out* o = &out[i];
    memset(o, 0, sizeof *o); o->search_level = -1; o->score = 8*8*250 + 1;
    /* TrackerData::Project, TrackerData.h:102-119 */
    double xc[3]; m3v(Rcw, p->world_pos, xc); xc[0] += tcw[0]; xc[1] += tcw[1]; xc[2] += tcw[2];
    const int invalid = orc_cam_project(cam, xc, o->image, o->cam_derivs);
    if (invalid) continue;
    if (o->image[0] < 0 || o->image[1] < 0 || o->image[0] > cam->image_size[0] || o->image[1] > cam->image_size[1]) continue;
    o->in_image = 1;
    /* CalcJacobian, TrackerData.h:155-175 */
    double dT[3], dP[3]; orc_cam_sphere_deriv(xc, dT, dP);
    { double xb[3]; m3v(bfw, p->world_pos, xb); xb[0] += bfw[9]; xb[1] += bfw[10]; xb[2] += bfw[11];
      for (int m = 0; m < 6; m++) {
        double mb[3], mc[3]; gen_field(m, xb, mb); m3v(cfb, mb, mc);
        const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
        o->jacobian[m] = o->cam_derivs[0]*s0 + o->cam_derivs[1]*s1; o->jacobian[6 + m] = o->cam_derivs[2]*s0 + o->cam_derivs[3]*s1;
      } }
    /* PatchFinder::CalcSearchLevelAndWarpMatrix, PatchFinder.cc:69-122 */
    double mr[3], md[3]; m3v(Rcw, p->pixel_right_w, mr); m3v(Rcw, p->pixel_down_w, md);
    const double sr0 = dT[0]*mr[0] + dT[1]*mr[1] + dT[2]*mr[2], sr1 = dP[0]*mr[0] + dP[1]*mr[1] + dP[2]*mr[2];
    const double sd0 = dT[0]*md[0] + dT[1]*md[1] + dT[2]*md[2], sd1 = dP[0]*md[0] + dP[1]*md[1] + dP[2]*md[2];
    double* WI = o->warp_inverse;
    WI[0] = o->cam_derivs[0]*sr0 + o->cam_derivs[1]*sr1; WI[2] = o->cam_derivs[2]*sr0 + o->cam_derivs[3]*sr1;   /* column 0 */
    WI[1] = o->cam_derivs[0]*sd0 + o->cam_derivs[1]*sd1; WI[3] = o->cam_derivs[2]*sd0 + o->cam_derivs[3]*sd1;   /* column 1 */
    double dDet = WI[0]*WI[3] - WI[1]*WI[2];
    int level = 0;
    while (dDet > 3 && level < ORC_LEVELS - 1) { level++; dDet *= 0.25; }
    if (dDet > 3 || dDet < 0.5 || !isfinite(dDet)) { o->template_bad = 1; continue; }
    o->search_level = level;
    /* MakeTemplateCoarseCont, :135-182 (cache neutralised: always refresh) */
    const int scale = 1 << level;
    double m2[4];
    { const double det = WI[0]*WI[3] - WI[1]*WI[2], id = 1.0/det;                         /* opts::M2Inverse, SmallMatrixOpts.h:67-79 */
      m2[0] = WI[3]*id*scale; m2[3] = WI[0]*id*scale; m2[2] = -WI[2]*id*scale; m2[1] = -WI[1]*id*scale; }
    const olevel* SL = &p->source_kf->lev[p->source_level];
    const int outside = cvd_transform8(SL, o->templ, m2, p->center_x, p->center_y, 4, 4);
    if (outside) { o->template_bad = 1; continue; }
    int tsum = 0, tsumsq = 0;                                                                 /* MakeTemplateSums :209-224 */
    for (int q = 0; q < 64; q++) { tsum += o->templ[q]; tsumsq += o->templ[q]*o->templ[q]; }
    /* FindPatchCoarse, :229-355 */
    const int bex = p->fixed || exhaustive;
    int its = subpix_its; if (bex) its = 10;                                                  /* Tracker.cc:1326-1331 */
    const olevel* L = &target->lev[level];
    int px = (int)o->image[0], py = (int)o->image[1];                                         /* CVD::ir truncation, Tracker.cc:1334 */
    px = px/scale; py = py/scale;                                                             /* ImageRef / int */
    const unsigned nr = ((unsigned)range + scale - 1)/scale;
    int top = py - (int)nr, bot1 = py + (int)nr + 1, left = px - (int)nr, right = px + (int)nr;
    o->searched = 1;
    const int MAXSSD = 8*8*250;
    int best = MAXSSD + 1, bx = 0, by = 0, early = 0;
    if (top < 0) top = 0;
    if (top >= L->h) early = 1;
    if (bot1 <= 0) early = 1;
    if (left < 0) left = 0;
    if (left >= L->w) early = 1;
    if (early) { o->found = 0; continue; }
    if (bex) {
      for (int y = top; y < bot1 && y < L->h; y++) for (int x = left; x <= right && x < L->w; x++) {
        if ((unsigned)((px - x)*(px - x) + (py - y)*(py - y)) > nr*nr) continue;
        const int s = zmssd(L, o->templ, tsum, tsumsq, x, y);
        if (s < best) { bx = x; by = y; best = s; }
      }
    } else {
      const int c0 = L->lut[top], c1 = (bot1 >= L->h) ? L->ncorners : L->lut[bot1];
      for (int c = c0; c < c1; c++) {
        const int x = L->corners[c].x, y = L->corners[c].y;
        if (x < left || x > right) continue;
        if ((unsigned)((px - x)*(px - x) + (py - y)*(py - y)) > nr*nr) continue;
        const int s = zmssd(L, o->templ, tsum, tsumsq, x, y);
        if (s < best) { bx = x; by = y; best = s; }
      }
    }
    o->score = best;
    if (!(best < MAXSSD)) { o->found = 0; continue; }
    o->found = 1; o->coarse_x = bx; o->coarse_y = by;
    double coarse[2] = { (bx + 0.5)*scale - 0.5, (by + 0.5)*scale - 0.5 };                   /* LevelZeroPos */
    o->sqrt_inv_noise = 1.0/scale;
    o->found_pos[0] = coarse[0]; o->found_pos[1] = coarse[1];
    if (its > 0) {
      o->did_subpix = 1;
      /* MakeSubPixTemplate, :362-390 */
      double jx[36], jy[36], H[9] = {0,0,0,0,0,0,0,0,0};
      for (int x = 1; x < 7; x++) for (int y = 1; y < 7; y++) {
        const double gx = 0.5*(o->templ[8*y + x + 1] - o->templ[8*y + x - 1]);
        const double gy = 0.5*(o->templ[8*(y + 1) + x] - o->templ[8*(y - 1) + x]);
        jx[(y - 1)*6 + x - 1] = gx; jy[(y - 1)*6 + x - 1] = gy;
        const double g[3] = { gx, gy, 1.0 };
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) H[3*a + b] += g[a]*g[b];
      }
      double Hinv[9]; inv3(H, Hinv);
      double sp[2] = { coarse[0], coarse[1] }, mean = 0.0;
      int converged = 0;
      for (int it = 0; it < its && !converged; it++) {                                         /* IterateSubPixToConvergence :392-410 */
        /* IterateSubPix :415-472 */
        const double cx = (sp[0] + 0.5)/scale - 0.5, cy = (sp[1] + 0.5)/scale - 0.5;           /* LevelNPos */
        const int rx = (int)round(cx), ry = (int)round(cy);
        if (!in_border(L, rx, ry, 5)) { converged = -1; break; }
        const double bxs = cx - 4, bys = cy - 4;
        const double dX = bxs - floor(bxs), dY = bys - floor(bys);
        const float fTL = (float)((1.0 - dX)*(1.0 - dY)), fTR = (float)(dX*(1.0 - dY)), fBL = (float)((1.0 - dX)*dY), fBR = (float)(dX*dY);
        double acc[3] = {0, 0, 0};
        for (int y = 1; y < 7; y++) {
          const uint8_t* q = L->img + (size_t)((int)bys + y)*L->w + (int)bxs + 1;
          for (int x = 1; x < 7; x++) {
            float fPixel = fTL*q[0] + fTR*q[1] + fBL*q[L->w] + fBR*q[L->w + 1];
            q++;
            const double d = fPixel - o->templ[8*y + x] + mean;
            acc[0] += d*jx[(y - 1)*6 + x - 1]; acc[1] += d*jy[(y - 1)*6 + x - 1]; acc[2] += d;
          }
        }
        double up[3]; m3v(Hinv, acc, up);
        sp[0] -= up[0]*scale; sp[1] -= up[1]*scale; mean -= up[2];
        const double u2 = up[0]*up[0] + up[1]*up[1];
        if (u2 < 0.03*0.03) converged = 1;
      }
      if (converged != 1) { o->found = 0; continue; }
      o->found_pos[0] = sp[0]; o->found_pos[1] = sp[1];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ PatchFinder as a stateful object
 * The reference's PatchFinder keeps members across calls (src/PatchFinder.cc:56-65): the template cache of MakeTemplateCoarseCont
 * (mpLastTemplateMapPoint, mm2LastWarpMatrix, mimTemplate, mbTemplateBad; :144-181), the sub-pixel Jacobians made by the last
 * MakeSubPixTemplate (mimJacs, mm3HInv; :362-390) and mdMeanDiff.  Its callers use that differently:
 *   ORC_PF_TRACK       Tracker::SearchForPoints (src/Tracker.cc:1299-1377): one finder per TrackerData, i.e. per (point, camera),
 *                      living over the frames; a point whose warp CalcSearchLevelAndWarpMatrix rejects never gets here (FindPVS)
 *   ORC_PF_REFIND      MapMakerServerBase::ReFind_Common (src/MapMakerServerBase.cc:921-1002): one static finder over all calls;
 *                      MakeTemplateCoarse = Calc + MakeTemplateCoarseCont WITHOUT looking at Calc's verdict (a refreshed template
 *                      overwrites mbTemplateBad); range 4; sub-pixel only above level 0, eight iterations, position kept whether
 *                      it converged or not (:981-987)
 *   ORC_PF_EPI_COARSE  MapMakerServerBase::AddPointEpipolar, first loop (:745-795): ONE finder and ONE MapPoint object over all depth
 *                      hypotheses -- the cache compares &point, so a template is kept while the warp moves less than 0.07, and a
 *                      rejected warp leaves mbTemplateBad = true for the next hypothesis that keeps its template
 *   ORC_PF_EPI_REFINE  the same function's second loop (:827-853): Calc + MakeTemplateCoarseCont (verdicts ignored), SetSubPixPos,
 *                      IterateSubPixToConvergence(10) with whatever mimJacs / mdMeanDiff the finder holds (MakeSubPixTemplate only
 *                      ran if the template was refreshed and good)
 * A sequence = the items one finder sees, in order; state[seq] is that finder's members, in and out. */
static void pf_make_subpix_template(orc_pf_state* S) {          /* MakeSubPixTemplate :362-390 (mv2SubPixPos is set by every caller afterwards) */
  memcpy(S->jac_templ, S->templ, 64); S->jacs_valid = 1; S->mean_diff = 0.0;
}
/* IterateSubPixToConvergence :392-410 + IterateSubPix :415-472 from sp[] with the finder's Jacobian template and mean difference;
 * returns 1 converged, 0 iterations used up, -1 left the image; sp / S->mean_diff hold the last state either way */
static int pf_iterate(const olevel* L, int scale, orc_pf_state* S, double sp[2], int max_its) {
  double jx[36], jy[36], H[9] = {0,0,0,0,0,0,0,0,0};
  const uint8_t* JT = S->jac_templ;
  for (int x = 1; x < 7; x++) for (int y = 1; y < 7; y++) {
    const double gx = 0.5*(JT[8*y + x + 1] - JT[8*y + x - 1]);
    const double gy = 0.5*(JT[8*(y + 1) + x] - JT[8*(y - 1) + x]);
    jx[(y - 1)*6 + x - 1] = gx; jy[(y - 1)*6 + x - 1] = gy;
    const double g[3] = { gx, gy, 1.0 };
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) H[3*a + b] += g[a]*g[b];
  }
  double Hinv[9]; inv3(H, Hinv);
  double mean = S->mean_diff;
  int converged = 0;
  for (int it = 0; it < max_its && !converged; it++) {
    const double cx = (sp[0] + 0.5)/scale - 0.5, cy = (sp[1] + 0.5)/scale - 0.5;           /* LevelNPos */
    const int rx = (int)round(cx), ry = (int)round(cy);
    if (!in_border(L, rx, ry, 5)) { converged = -1; break; }
    const double bxs = cx - 4, bys = cy - 4;
    const double dX = bxs - floor(bxs), dY = bys - floor(bys);
    const float fTL = (float)((1.0 - dX)*(1.0 - dY)), fTR = (float)(dX*(1.0 - dY)), fBL = (float)((1.0 - dX)*dY), fBR = (float)(dX*dY);
    double acc[3] = {0, 0, 0};
    for (int y = 1; y < 7; y++) {
      const uint8_t* q = L->img + (size_t)((int)bys + y)*L->w + (int)bxs + 1;
      for (int x = 1; x < 7; x++) {
        float fPixel = fTL*q[0] + fTR*q[1] + fBL*q[L->w] + fBR*q[L->w + 1];
        q++;
        const double d = fPixel - S->templ[8*y + x] + mean;
        acc[0] += d*jx[(y - 1)*6 + x - 1]; acc[1] += d*jy[(y - 1)*6 + x - 1]; acc[2] += d;
      }
    }
    double up[3]; m3v(Hinv, acc, up);
    sp[0] -= up[0]*scale; sp[1] -= up[1]*scale; mean -= up[2];
    const double u2 = up[0]*up[0] + up[1]*up[1];
    if (u2 < 0.03*0.03) converged = 1;
  }
  S->mean_diff = mean;
  return converged;
}
int orc_patch_sequences(int mode, int n_targets, const orc_pf_target* targets, int n_seq, const int* seq_start, const orc_pf_item* items,
                        orc_pf_state* state, int range, int subpix_its, int exhaustive, orc_td_out* out) {
  const int MAXSSD = 8*8*250;
  for (int sq = 0; sq < n_seq; sq++) {
    orc_pf_state* S = &state[sq];
    for (int ii = seq_start[sq]; ii < seq_start[sq + 1]; ii++) {
      const orc_pf_item* I = &items[ii]; const orc_td_in* p = &I->point; orc_td_out* o = &out[ii];
      if (I->target < 0 || I->target >= n_targets) return -1;
      const orc_pf_target* G = &targets[I->target];
      const orc_camera* cam = G->cam; const double* bfw = G->base_from_world; const double* cfb = G->cam_from_base;
      double Rcw[9], tcw[3];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3*i+j] = cfb[3*i]*bfw[j] + cfb[3*i+1]*bfw[3+j] + cfb[3*i+2]*bfw[6+j];
      m3v(cfb, bfw + 9, tcw); tcw[0] += cfb[9]; tcw[1] += cfb[10]; tcw[2] += cfb[11];
      memset(o, 0, sizeof *o); o->search_level = -1; o->score = MAXSSD + 1;
      double xc[3]; m3v(Rcw, p->world_pos, xc); xc[0] += tcw[0]; xc[1] += tcw[1]; xc[2] += tcw[2];
      const int invalid = orc_cam_project(cam, xc, o->image, o->cam_derivs);
      const olevel* L0 = &G->kf->lev[0];
      if (mode == ORC_PF_TRACK || mode == ORC_PF_REFIND) {                                    /* TrackerData.h:102-119; MapMakerServerBase.cc:941-953 */
        if (invalid) continue;
        if (o->image[0] < 0 || o->image[1] < 0 || o->image[0] > cam->image_size[0] || o->image[1] > cam->image_size[1]) continue;
      } else if (mode == ORC_PF_EPI_COARSE) {                                                 /* :757-765: Invalid, in_image(CVD::ir(v2Image)), mask == 0 */
        if (invalid) continue;
        const int ix = (int)o->image[0], iy = (int)o->image[1];
        if (!(ix >= 0 && iy >= 0 && ix < L0->w && iy < L0->h)) continue;
        if (L0->mask[(size_t)iy*L0->w + ix] == 0) continue;
      }
      o->in_image = !invalid;
      double dT[3], dP[3]; orc_cam_sphere_deriv(xc, dT, dP);
      { double xb[3]; m3v(bfw, p->world_pos, xb); xb[0] += bfw[9]; xb[1] += bfw[10]; xb[2] += bfw[11];
        for (int m = 0; m < 6; m++) {
          double mb[3], mc[3]; gen_field(m, xb, mb); m3v(cfb, mb, mc);
          const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
          o->jacobian[m] = o->cam_derivs[0]*s0 + o->cam_derivs[1]*s1; o->jacobian[6 + m] = o->cam_derivs[2]*s0 + o->cam_derivs[3]*s1;
        } }
      /* CalcSearchLevelAndWarpMatrix :69-122 */
      double mr[3], md[3]; m3v(Rcw, p->pixel_right_w, mr); m3v(Rcw, p->pixel_down_w, md);
      const double sr0 = dT[0]*mr[0] + dT[1]*mr[1] + dT[2]*mr[2], sr1 = dP[0]*mr[0] + dP[1]*mr[1] + dP[2]*mr[2];
      const double sd0 = dT[0]*md[0] + dT[1]*md[1] + dT[2]*md[2], sd1 = dP[0]*md[0] + dP[1]*md[1] + dP[2]*md[2];
      double* WI = o->warp_inverse;
      WI[0] = o->cam_derivs[0]*sr0 + o->cam_derivs[1]*sr1; WI[2] = o->cam_derivs[2]*sr0 + o->cam_derivs[3]*sr1;
      WI[1] = o->cam_derivs[0]*sd0 + o->cam_derivs[1]*sd1; WI[3] = o->cam_derivs[2]*sd0 + o->cam_derivs[3]*sd1;
      double dDet = WI[0]*WI[3] - WI[1]*WI[2];
      int level = 0;
      while (dDet > 3 && level < ORC_LEVELS - 1) { level++; dDet *= 0.25; }
      const int rejected = (dDet > 3 || dDet < 0.5 || !isfinite(dDet));
      if (rejected) {
        S->template_bad = 1;                                                                    /* :116-117: the member is set before returning -1 */
        if (mode == ORC_PF_TRACK || mode == ORC_PF_EPI_COARSE) { o->template_bad = 1; continue; }     /* FindPVS drops the point; :769-770 `continue` */
      }
      o->search_level = level;
      /* MakeTemplateCoarseCont :135-182 */
      const int scale = 1 << level;
      double m2[4];
      { const double det = WI[0]*WI[3] - WI[1]*WI[2], id = 1.0/det;                         /* opts::M2Inverse, SmallMatrixOpts.h:67-79 */
        m2[0] = WI[3]*id*scale; m2[3] = WI[0]*id*scale; m2[2] = -WI[2]*id*scale; m2[1] = -WI[1]*id*scale; }
      int refresh = !S->valid || S->point_key != I->point_key;
      for (int c = 0; !refresh && c < 2; c++) {                                               /* columns: m2.T()[c] = (m2[0][c], m2[1][c]) */
        const double d0 = m2[c] - S->last_warp[c], d1 = m2[2 + c] - S->last_warp[2 + c];
        if (d0*d0 + d1*d1 > 0.07*0.07) refresh = 1;
      }
      if (refresh) {
        const olevel* SL = &p->source_kf->lev[p->source_level];
        const int outside = cvd_transform8(SL, S->templ, m2, p->center_x, p->center_y, 4, 4);
        S->template_bad = outside ? 1 : 0;
        S->valid = 1; S->point_key = I->point_key; memcpy(S->last_warp, m2, sizeof m2);
        if (!S->template_bad) pf_make_subpix_template(S);
      }
      memcpy(o->templ, S->templ, 64); o->template_bad = S->template_bad;
      if (S->template_bad && mode != ORC_PF_EPI_REFINE) continue;                            /* Tracker.cc:1316; MapMakerServerBase.cc:774, 958 */
      int tsum = 0, tsumsq = 0;                                                                 /* MakeTemplateSums :209-224 */
      for (int q = 0; q < 64; q++) { tsum += S->templ[q]; tsumsq += S->templ[q]*S->templ[q]; }
      const olevel* L = &G->kf->lev[level];
      if (mode == ORC_PF_EPI_REFINE) {                                                        /* :845-851 */
        o->sqrt_inv_noise = 1.0/scale; o->did_subpix = 1;
        double sp[2] = { I->start_pos[0], I->start_pos[1] };
        const int conv = S->jacs_valid ? pf_iterate(L, scale, S, sp, 10) : 0;
        o->found = (conv == 1); o->found_pos[0] = sp[0]; o->found_pos[1] = sp[1];
        continue;
      }
      /* FindPatchCoarse :229-355 */
      const int bex = (mode == ORC_PF_TRACK) && (p->fixed || exhaustive);
      int its = subpix_its; if (bex) its = 10;                                                  /* Tracker.cc:1326-1331 */
      int px = (int)o->image[0], py = (int)o->image[1];                                         /* CVD::ir truncation */
      px = px/scale; py = py/scale;
      const unsigned nr = ((unsigned)range + scale - 1)/scale;
      int top = py - (int)nr, bot1 = py + (int)nr + 1, left = px - (int)nr, right = px + (int)nr;
      o->searched = 1;
      int best = MAXSSD + 1, bx = 0, by = 0, early = 0;
      if (top < 0) top = 0;
      if (top >= L->h) early = 1;
      if (bot1 <= 0) early = 1;
      if (left < 0) left = 0;
      if (left >= L->w) early = 1;
      if (early) { o->found = 0; continue; }
      if (bex) {
        for (int y = top; y < bot1 && y < L->h; y++) for (int x = left; x <= right && x < L->w; x++) {
          if ((unsigned)((px - x)*(px - x) + (py - y)*(py - y)) > nr*nr) continue;
          const int s = zmssd(L, S->templ, tsum, tsumsq, x, y);
          if (s < best) { bx = x; by = y; best = s; }
        }
      } else {
        const int c0 = L->lut[top], c1 = (bot1 >= L->h) ? L->ncorners : L->lut[bot1];
        for (int c = c0; c < c1; c++) {
          const int x = L->corners[c].x, y = L->corners[c].y;
          if (x < left || x > right) continue;
          if ((unsigned)((px - x)*(px - x) + (py - y)*(py - y)) > nr*nr) continue;
          const int s = zmssd(L, S->templ, tsum, tsumsq, x, y);
          if (s < best) { bx = x; by = y; best = s; }
        }
      }
      o->score = best;
      if (!(best < MAXSSD)) { o->found = 0; continue; }
      o->found = 1; o->coarse_x = bx; o->coarse_y = by;
      double coarse[2] = { (bx + 0.5)*scale - 0.5, (by + 0.5)*scale - 0.5 };                   /* LevelZeroPos */
      o->sqrt_inv_noise = 1.0/scale;
      o->found_pos[0] = coarse[0]; o->found_pos[1] = coarse[1];
      if (mode == ORC_PF_TRACK && its > 0) {                                                    /* Tracker.cc:1350-1366 */
        o->did_subpix = 1;
        pf_make_subpix_template(S);
        double sp[2] = { coarse[0], coarse[1] };
        if (pf_iterate(L, scale, S, sp, its) != 1) { o->found = 0; continue; }
        o->found_pos[0] = sp[0]; o->found_pos[1] = sp[1];
      } else if (mode == ORC_PF_REFIND && level > 0) {                                          /* MapMakerServerBase.cc:981-987 */
        o->did_subpix = 1;
        pf_make_subpix_template(S);
        double sp[2] = { coarse[0], coarse[1] };
        (void)pf_iterate(L, scale, S, sp, 8);
        o->found_pos[0] = sp[0]; o->found_pos[1] = sp[1];
      }
    }
  }
  return 0;
}

static int cmp_d(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }
/* The estimators Tracker::CalcPoseUpdate dispatches on (src/Tracker.cc:1388-1401, 1429-1468): include/mcptam/MEstimator.h
 * Tukey :84-124, Cauchy :131-157 (same sigma as Tukey, weight 1/(1 + e/s)), Huber :164-204 (1.345; weight 1 or sqrt(s/e)). */
static double mest_sigma_squared(int est, double* e2, int n) { return est == 2 ? orc_huber_sigma_squared(e2, n) : orc_tukey_sigma_squared(e2, n); }
static double mest_weight(int est, double e, double s2) {
  if (est == 1) return 1.0/(1.0 + e/s2);
  if (est == 2) return (e < s2) ? 1.0 : sqrt(s2/e);
  return orc_tukey_weight(e, s2);
}
/* Tracker::CalcPoseUpdate with TooN WLS<6> [3P-memory], Tracker.cc:1386-1512; est 0 Tukey (the default), 1 Cauchy, 2 Huber */
int orc_track_pose_update(int n, const uint8_t* found, const double* fpos, const double* ipos, const double* sin,
                          const double* J, double override_sigma, double mu[6], double* wout, double* sigma_out) {
  return orc_track_pose_update_m(n, found, fpos, ipos, sin, J, override_sigma, mu, wout, sigma_out, 0);
}
int orc_track_pose_update_m(int n, const uint8_t* found, const double* fpos, const double* ipos, const double* sin,
                            const double* J, double override_sigma, double mu[6], double* wout, double* sigma_out, int est) {
  double* e2 = (double*)malloc(sizeof(double)*(n + 1)); int ne = 0;
  double* ex = (double*)malloc(sizeof(double)*(2*(size_t)n + 2));
  for (int i = 0; i < n; i++) {
    if (wout) wout[i] = 0;
    if (!found[i]) continue;
    ex[2*i] = sin[i]*(fpos[2*i] - ipos[2*i]); ex[2*i+1] = sin[i]*(fpos[2*i+1] - ipos[2*i+1]);
    e2[ne++] = ex[2*i]*ex[2*i] + ex[2*i+1]*ex[2*i+1];
  }
  for (int k = 0; k < 6; k++) mu[k] = 0;
  if (ne == 0) { free(e2); free(ex); if (sigma_out) *sigma_out = 0; return 0; }
  double s2;
  if (override_sigma > 0) s2 = override_sigma; else s2 = mest_sigma_squared(est, e2, ne);
  if (sigma_out) *sigma_out = s2;
  double C[36], v[6];
  for (int a = 0; a < 36; a++) C[a] = 0;
  for (int a = 0; a < 6; a++) { C[7*a] = 100.0; v[a] = 0; }   /* add_prior(100) */
  for (int i = 0; i < n; i++) {
    if (!found[i]) continue;
    const double err2 = ex[2*i]*ex[2*i] + ex[2*i+1]*ex[2*i+1];
    const double w = mest_weight(est, err2, s2);
    if (wout) wout[i] = w;
    if (w == 0.0) continue;
    for (int r = 0; r < 2; r++) {
      double Jr[6]; for (int k = 0; k < 6; k++) Jr[k] = sin[i]*J[12*(size_t)i + 6*r + k];
      const double m = ex[2*i + r];
      for (int a = 0; a < 6; a++) { v[a] += w*m*Jr[a]; for (int b = 0; b < 6; b++) C[6*a + b] += w*Jr[a]*Jr[b]; }
    }
  }
  /* Cholesky solve of the 6x6 */
  double L[36]; memcpy(L, C, sizeof L);
  for (int i = 0; i < 6; i++) for (int j = 0; j <= i; j++) { double s = L[6*i + j]; for (int k = 0; k < j; k++) s -= L[6*i + k]*L[6*j + k]; L[6*i + j] = (i == j) ? sqrt(s) : s/L[6*j + j]; }
  for (int i = 0; i < 6; i++) { double s = v[i]; for (int k = 0; k < i; k++) s -= L[6*i + k]*mu[k]; mu[i] = s/L[6*i + i]; }
  for (int i = 5; i >= 0; i--) { double s = mu[i]; for (int k = i + 1; k < 6; k++) s -= L[6*k + i]*mu[k]; mu[i] = s/L[6*i + i]; }
  free(e2); free(ex);
  return 0;
}


/* ================================================================== SmallBlurryImage / Relocaliser */
#include "ba_oracle.h"
/* cv::resize(8U, INTER_LINEAR) [3P-memory, OpenCV imgproc resize.cpp]: source coordinate (dx+0.5)*scale-0.5 in float,
 * clamped at the borders, 11-bit fixed-point weights (INTER_RESIZE_COEF_BITS), horizontal pass in int, vertical pass
 * ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2. */
static void resize_coeffs(int src, int dst, int* idx, short* w0, short* w1) {
  const double scale = (double)src/dst;
  for (int d = 0; d < dst; d++) {
    float f = (float)((d + 0.5)*scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= src - 1) { f = 0; s = src - 1; }           /* the second tap is clamped to the last pixel with weight 0 */
    idx[d] = s;
    float a0 = (1.f - f)*2048.f, a1 = f*2048.f;
    w0[d] = (short)lrintf(a0); w1[d] = (short)lrintf(a1);
  }
}
static void resize_linear_u8(const uint8_t* in, int iw, int ih, uint8_t* out, int ow, int oh) {
  int* xi = (int*)malloc(sizeof(int)*ow); short* xa = (short*)malloc(2*ow); short* xb = (short*)malloc(2*ow);
  int* yi = (int*)malloc(sizeof(int)*oh); short* ya = (short*)malloc(2*oh); short* yb = (short*)malloc(2*oh);
  resize_coeffs(iw, ow, xi, xa, xb); resize_coeffs(ih, oh, yi, ya, yb);
  for (int y = 0; y < oh; y++) {
    const uint8_t* r0 = in + (size_t)yi[y]*iw; const uint8_t* r1 = in + (size_t)(yi[y] + 1 < ih ? yi[y] + 1 : ih - 1)*iw;
    for (int x = 0; x < ow; x++) {
      const int x0 = xi[x], x1 = x0 + 1 < iw ? x0 + 1 : iw - 1;
      const int h0 = r0[x0]*xa[x] + r0[x1]*xb[x], h1 = r1[x0]*xa[x] + r1[x1]*xb[x];
      out[y*ow + x] = (uint8_t)((((ya[y]*(h0 >> 4)) >> 16) + ((yb[y]*(h1 >> 4)) >> 16) + 2) >> 2);
    }
  }
  free(xi); free(xa); free(xb); free(yi); free(ya); free(yb);
}
/* CVD::convolveGaussian(Image<float>&, sigma) [3P-memory, cvd/convolution.h]: separable, kernel half-size
 * ceil(3 sigma), taps exp(-i^2/(2 sigma^2)) normalised to unit sum, float accumulation (centre tap first, then the
 * symmetric pairs outwards), samples outside the image contribute nothing; rows first, then columns, in place. */
static int gauss_kernel(double sigma, float* k /* >= 32 */) {
  int ks = (int)ceil(3.0*sigma); if (ks > 31) ks = 31;
  double sum = 1.0;
  for (int i = 1; i <= ks; i++) sum += 2.0*exp(-(double)i*i/(2.0*sigma*sigma));
  k[0] = (float)(1.0/sum);
  for (int i = 1; i <= ks; i++) k[i] = (float)(exp(-(double)i*i/(2.0*sigma*sigma))/sum);
  return ks;
}
static void convolve_gaussian(float* I, int w, int h, double sigma) {
  float k[32]; const int ks = gauss_kernel(sigma, k);
  float* tmp = (float*)malloc(sizeof(float)*w*h);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    float a = I[y*w + x]*k[0];
    for (int i = 1; i <= ks; i++) { float p = 0.f; if (x - i >= 0) p += I[y*w + x - i]; if (x + i < w) p += I[y*w + x + i]; a += p*k[i]; }
    tmp[y*w + x] = a;
  }
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    float a = tmp[y*w + x]*k[0];
    for (int i = 1; i <= ks; i++) { float p = 0.f; if (y - i >= 0) p += tmp[(y - i)*w + x]; if (y + i < h) p += tmp[(y + i)*w + x]; a += p*k[i]; }
    I[y*w + x] = a;
  }
  free(tmp);
}
/* SmallBlurryImage::MakeFromKF + MakeJacs, SmallBlurryImage.cc:67-118 */
int orc_kf_make_sbi(orc_kf* k, double blur) {
  if (!k->sbi) k->sbi = (osbi*)calloc(1, sizeof(osbi));
  osbi* s = k->sbi;
  const int W = ORC_SBI_W, H = ORC_SBI_H;
  resize_linear_u8(k->lev[0].img, k->lev[0].w, k->lev[0].h, s->small, W, H);                /* :76-79 */
  unsigned int sum = 0;
  for (int i = 0; i < SBI_N; i++) sum += s->small[i];                                         /* :81-85 */
  const float mean = ((float)sum)/SBI_N;                                                      /* :87 */
  for (int i = 0; i < SBI_N; i++) s->templ[i] = s->small[i] - mean;                            /* :89-92 */
  convolve_gaussian(s->templ, W, H, blur);                                                    /* :94 */
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {                                    /* MakeJacs :99-118, no 0.5 factor */
    float gx = 0.f, gy = 0.f;
    if (x >= 1 && y >= 1 && x < W - 1 && y < H - 1) { gx = s->templ[y*W + x + 1] - s->templ[y*W + x - 1]; gy = s->templ[(y + 1)*W + x] - s->templ[(y - 1)*W + x]; }
    s->jacs[2*(y*W + x)] = gx; s->jacs[2*(y*W + x) + 1] = gy;
  }
  return 0;
}
const uint8_t* orc_kf_sbi_small(orc_kf* k) { return k->sbi ? k->sbi->small : NULL; }
const float* orc_kf_sbi_template(orc_kf* k) { return k->sbi ? k->sbi->templ : NULL; }
const float* orc_kf_sbi_jacs(orc_kf* k) { return k->sbi ? k->sbi->jacs : NULL; }
/* SmallBlurryImage::ZMSSD, :122-134 */
double orc_sbi_zmssd(orc_kf* a, orc_kf* b) {
  double ssd = 0.0;
  for (int i = 0; i < SBI_N; i++) { const double d = a->sbi->templ[i] - b->sbi->templ[i]; ssd += d*d; }
  return ssd;
}
/* Relocaliser::ScoreKFs, Relocaliser.cc:93-121 (strict <: the first smallest wins) */
int orc_sbi_score(orc_kf* cur, int n, orc_kf* const* cands, double* scores) {
  double best = DBL_MAX; int bi = -1;
  for (int i = 0; i < n; i++) {
    if (!cands[i]->sbi) { scores[i] = DBL_MAX; continue; }
    scores[i] = orc_sbi_zmssd(cur, cands[i]);
    if (scores[i] < best) { best = scores[i]; bi = i; }
  }
  return bi;
}
/* CVD::transform on a float image with default value [3P-memory], as cvd_transform8 above */
static void cvd_transform_f(const float* in, int iw, int ih, float* out, const double M[4], double inx, double iny, float defval) {
  const int w = iw, h = ih;
  const double across[2] = { M[0], M[2] }, down[2] = { M[1], M[3] };
  double p0[2] = { inx, iny };                       /* outOrig = 0 */
  const double cr[2] = { down[0] - w*across[0], down[1] - w*across[1] };
  const double xb = iw - 1, yb = ih - 1;
  double p[2] = { p0[0], p0[1] };
  for (int i = 0; i < h; ++i, p[0] += cr[0], p[1] += cr[1])
    for (int j = 0; j < w; ++j, p[0] += across[0], p[1] += across[1]) {
      if (0 <= p[0] && 0 <= p[1] && p[0] < xb && p[1] < yb) {
        const int lx = (int)p[0], ly = (int)p[1];
        const double x = p[0] - lx, y = p[1] - ly;
        const float* q = in + (size_t)ly*iw + lx;
        out[i*w + j] = (float)((1 - y)*((1 - x)*q[0] + x*q[1]) + y*((1 - x)*q[iw] + x*q[iw + 1]));
      } else out[i*w + j] = defval;
    }
}
static void se2_mul(const double* A, const double* B, double* C) {     /* {R00,R01,R10,R11,tx,ty} */
  const double r[6] = { A[0]*B[0] + A[1]*B[2], A[0]*B[1] + A[1]*B[3], A[2]*B[0] + A[3]*B[2], A[2]*B[1] + A[3]*B[3],
                        A[0]*B[4] + A[1]*B[5] + A[4], A[2]*B[4] + A[3]*B[5] + A[5] };
  memcpy(C, r, sizeof r);
}
static void se2_inv(const double* A, double* C) {
  const double r[6] = { A[0], A[2], A[1], A[3], -(A[0]*A[4] + A[2]*A[5]), -(A[1]*A[4] + A[3]*A[5]) };
  memcpy(C, r, sizeof r);
}
/* 4x4 SPD solve (TooN Cholesky<4>::backsub, LDL^T) */
static void solve4(const double* A, const double* b, double* x) {
  double L[16] = {0}, D[4];
  for (int j = 0; j < 4; j++) {
    double d = A[5*j];
    for (int k = 0; k < j; k++) d -= L[4*j + k]*L[4*j + k]*D[k];
    D[j] = d;
    for (int i = j + 1; i < 4; i++) { double v = A[4*i + j]; for (int k = 0; k < j; k++) v -= L[4*i + k]*L[4*j + k]*D[k]; L[4*i + j] = v/d; }
  }
  double y[4];
  for (int i = 0; i < 4; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[4*i + k]*y[k]; y[i] = v; }
  for (int i = 0; i < 4; i++) y[i] /= D[i];
  for (int i = 3; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 4; k++) v -= L[4*k + i]*x[k]; x[i] = v; }
}
/* SmallBlurryImage::IteratePosRelToTarget (ESM tracking), :139-245 */
int orc_sbi_iterate(orc_kf* cur, orc_kf* target, int iterations, double se2[6], double* score) {
  const int W = ORC_SBI_W, H = ORC_SBI_H;
  const int cx = W/2, cy = H/2;
  double CtoC[6] = { 1, 0, 0, 1, 0, 0 };
  const double WfromC[6] = { 1, 0, 0, 1, (double)cx, (double)cy };
  double WfromCinv[6]; se2_inv(WfromC, WfromCinv);
  double mean_offset = 0.0, final_score = 0.0;
  float warped[SBI_N];
  const osbi* me = cur->sbi; const osbi* ot = target->sbi;
  for (int it = 0; it < iterations; it++) {
    final_score = 0.0;
    double acc[4] = { 0, 0, 0, 0 }, tri[10] = { 0 };
    double X[6], T1[6]; se2_mul(WfromC, CtoC, T1); se2_mul(T1, WfromCinv, X);
    cvd_transform_f(me->templ, W, H, warped, X, X[4], X[5], -9e20f);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
      if (!(x >= 1 && y >= 1 && x < W - 1 && y < H - 1)) continue;
      const float l = warped[y*W + x - 1], r = warped[y*W + x + 1], u = warped[(y - 1)*W + x], d = warped[(y + 1)*W + x], here = warped[y*W + x];
      if (l + r + u + d + here < -9999.9) continue;
      const double g0 = r - l, g1 = d - u;                                   /* float differences, :189-190 */
      const double s0 = 0.25*(g0 + (double)ot->jacs[2*(y*W + x)]), s1 = 0.25*(g1 + (double)ot->jacs[2*(y*W + x) + 1]);
      const double J[4] = { s0, s1, -(y - cy)*s0 + (x - cx)*s1, 1.0 };
      const double diff = (here - ot->templ[y*W + x]) + mean_offset;
      final_score += diff*diff;
      for (int a = 0; a < 4; a++) acc[a] += diff*J[a];
      tri[0] += J[0]*J[0]; tri[1] += J[1]*J[0]; tri[2] += J[1]*J[1]; tri[3] += J[2]*J[0]; tri[4] += J[2]*J[1];
      tri[5] += J[2]*J[2]; tri[6] += J[0]; tri[7] += J[1]; tri[8] += J[2]; tri[9] += 1.0;
    }
    double m4[16]; int v = 0;
    for (int j = 0; j < 4; j++) for (int i = 0; i <= j; i++) { m4[4*j + i] = m4[4*i + j] = tri[v++]; }
    double upd[4]; solve4(m4, acc, upd);
    const double th = -upd[2];
    const double U[6] = { cos(th), -sin(th), sin(th), cos(th), -upd[0], -upd[1] };
    se2_mul(CtoC, U, CtoC);
    mean_offset -= upd[3];
  }
  memcpy(se2, CtoC, sizeof CtoC);
  *score = final_score;
  return 0;
}
/* TaylorCamera::UnProject, TaylorCamera.cc:319-347 */
static void cam_unproject(const orc_camera* c, const double uv[2], double out[3]) {
  const double det = c->affine[0]*c->affine[3] - c->affine[1]*c->affine[2];
  const double ai[4] = { c->affine[3]/det, -c->affine[1]/det, -c->affine[2]/det, c->affine[0]/det };
  const double dx = uv[0] - c->center[0], dy = uv[1] - c->center[1];
  const double x = ai[0]*dx + ai[1]*dy, y = ai[2]*dx + ai[3]*dy;
  const double rho = sqrt(x*x + y*y);
  const double p[5] = { c->params[0], 0.0, c->params[1], c->params[2], c->params[3] };
  double z = p[4]; for (int i = 3; i >= 0; i--) z = z*rho + p[i];
  const double n = sqrt(x*x + y*y + z*z);
  out[0] = x/n; out[1] = y/n; out[2] = z/n;
}
/* SmallBlurryImage::SE3fromSE2, :250-310 (cameras already at SBI size) */
void orc_sbi_se3_from_se2(const double se2[6], const orc_camera* cs, const orc_camera* ct, double R[9]) {
  const double c[2] = { ORC_SBI_W/2, ORC_SBI_H/2 };
  double turned[2][2], orig[2][3];
  const double off[2][2] = { { 5, 0 }, { -5, 0 } };
  for (int i = 0; i < 2; i++) {
    turned[i][0] = c[0] + se2[0]*off[i][0] + se2[1]*off[i][1] + se2[4];
    turned[i][1] = c[1] + se2[2]*off[i][0] + se2[3]*off[i][1] + se2[5];
    const double px[2] = { c[0] + off[i][0], c[1] + off[i][1] };
    cam_unproject(ct, px, orig[i]);
  }
  double so3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  for (int it = 0; it < 3; it++) {
    double C[9] = { 10, 0, 0, 0, 10, 0, 0, 0, 10 }, v[3] = { 0, 0, 0 };        /* add_prior(10) */
    for (int i = 0; i < 2; i++) {
      double cam[3]; m3v(so3, orig[i], cam);
      double px[2], D[4]; orc_cam_project(cs, cam, px, D);
      const double err[2] = { turned[i][0] - px[0], turned[i][1] - px[1] };
      double dT[3], dP[3]; orc_cam_sphere_deriv(cam, dT, dP);
      double J[2][3];
      for (int m = 0; m < 3; m++) {
        double mot[3] = { 0, 0, 0 };                                           /* SO3::generator_field(m, cam) = e_m x cam */
        mot[(m + 1)%3] = -cam[(m + 2)%3]; mot[(m + 2)%3] = cam[(m + 1)%3];
        const double sm[2] = { dT[0]*mot[0] + dT[1]*mot[1] + dT[2]*mot[2], dP[0]*mot[0] + dP[1]*mot[1] + dP[2]*mot[2] };
        J[0][m] = D[0]*sm[0] + D[1]*sm[1]; J[1][m] = D[2]*sm[0] + D[3]*sm[1];
      }
      for (int r = 0; r < 2; r++) for (int a = 0; a < 3; a++) { v[a] += J[r][a]*err[r]; for (int b = 0; b < 3; b++) C[3*a + b] += J[r][a]*J[r][b]; }
    }
    double Ci[9]; inv3(C, Ci);
    double mu[3]; m3v(Ci, v, mu);
    double E[9]; orc_so3_exp(mu, E);
    double Rn[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rn[3*a + b] = E[3*a]*so3[b] + E[3*a + 1]*so3[3 + b] + E[3*a + 2]*so3[6 + b];
    memcpy(so3, Rn, sizeof Rn);
  }
  memcpy(R, so3, sizeof(double)*9);
}


/* ================================================================== Tracker::TrackMap pose iterations */
int orc_track_pose_refine(int n, orc_pose_point* pts, int ncam, const orc_camera* cams, const double* cfb_all,
                          double bfw[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                          double mu_last[6], double* weights_last) {
  return orc_track_pose_refine_m(n, pts, ncam, cams, cfb_all, bfw, n_iter, nonlinear, override_sigma, mu_last, weights_last, 0);
}
int orc_track_pose_refine_m(int n, orc_pose_point* pts, int ncam, const orc_camera* cams, const double* cfb_all,
                            double bfw[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                            double mu_last[6], double* weights_last, int est) {
  uint8_t* found = (uint8_t*)malloc(n + 1);
  double* fpos = (double*)malloc(sizeof(double)*(2*(size_t)n + 2)); double* ipos = (double*)malloc(sizeof(double)*(2*(size_t)n + 2));
  double* sinv = (double*)malloc(sizeof(double)*(n + 1)); double* J = (double*)calloc(12*(size_t)n + 12, sizeof(double));
  double v6[6] = { 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < n; i++) { found[i] = (uint8_t)(pts[i].found != 0); fpos[2*i] = pts[i].found_pos[0]; fpos[2*i+1] = pts[i].found_pos[1]; sinv[i] = pts[i].sqrt_inv_noise; }
  (void)ncam;
  for (int it = 0; it < n_iter; it++) {
    if (nonlinear[it]) {                                    /* PoseUpdateStep, :775-812 */
      for (int i = 0; i < n; i++) {
        orc_pose_point* p = &pts[i];
        if (!p->found) continue;
        const double* cfb = cfb_all + 12*(size_t)p->cam;
        double xb[3]; m3v(bfw, p->world_pos, xb); xb[0] += bfw[9]; xb[1] += bfw[10]; xb[2] += bfw[11];
        double xc[3]; m3v(cfb, xb, xc); xc[0] += cfb[9]; xc[1] += cfb[10]; xc[2] += cfb[11];
        if (it != 0) orc_cam_project(&cams[p->cam], xc, p->image, p->cam_derivs);          /* ProjectAndDerivs, :783-790 */
        double dT[3], dP[3]; orc_cam_sphere_deriv(xc, dT, dP);                                /* CalcJacobian, TrackerData.h:152-176 */
        for (int m = 0; m < 6; m++) {
          double mb[3], mc[3]; gen_field(m, xb, mb); m3v(cfb, mb, mc);
          const double s0 = dT[0]*mc[0] + dT[1]*mc[1] + dT[2]*mc[2], s1 = dP[0]*mc[0] + dP[1]*mc[1] + dP[2]*mc[2];
          J[12*(size_t)i + m] = p->cam_derivs[0]*s0 + p->cam_derivs[1]*s1; J[12*(size_t)i + 6 + m] = p->cam_derivs[2]*s0 + p->cam_derivs[3]*s1;
        }
      }
    } else {                                                /* PoseUpdateStepLinear, :815-838: mv2Image += J v6 */
      for (int i = 0; i < n; i++) {
        if (!pts[i].found) continue;
        for (int r = 0; r < 2; r++) { double a = 0; for (int k = 0; k < 6; k++) a += J[12*(size_t)i + 6*r + k]*v6[k]; pts[i].image[r] += a; }
      }
    }
    for (int i = 0; i < n; i++) { ipos[2*i] = pts[i].image[0]; ipos[2*i+1] = pts[i].image[1]; }
    double s2;
    orc_track_pose_update_m(n, found, fpos, ipos, sinv, J, override_sigma[it], v6, (it == n_iter - 1) ? weights_last : NULL, &s2, est);
    double E[9], et[3]; orc_se3_exp(v6, E, et);            /* mse3BaseFromWorld = exp(v6) * mse3BaseFromWorld */
    double nb[12];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) nb[3*a + b] = E[3*a]*bfw[b] + E[3*a+1]*bfw[3+b] + E[3*a+2]*bfw[6+b];
    m3v(E, bfw + 9, nb + 9); nb[9] += et[0]; nb[10] += et[1]; nb[11] += et[2];
    memcpy(bfw, nb, sizeof nb);
  }
  memcpy(mu_last, v6, sizeof v6);
  free(found); free(fpos); free(ipos); free(sinv); free(J);
  return 0;
}
