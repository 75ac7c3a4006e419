// The cut of the pose coupling graph that gives the one-launch factorisation (ba_chol2.h) several chains of block columns to walk
// beside each other (round 6).  Host code only: prepare() calls it with the graph it took from the measurements, the
// debug hook mcp_debug_pose_cut (include/mcp_ba.h) with a graph the caller hands over (tests/test_pose_cut.py runs without a GPU).
//
// The one-launch factorisation walks the block columns of a chain one after the other on one critical workgroup; chains that do not
// couple are walked beside each other.  The reduced system of a trajectory is mostly a band in add order -- a pose couples with the poses
// that see a point it sees -- closed to a ring when the trajectory returns, plus the odd group of poses that sees points from across a
// loop.  Cut the ring of free poses (add order, cyclic) into two to four arcs with a gap behind each; whatever still couples one arc with
// another goes to the separator too (greedily, the pose with the most such couplings first).  Ordered [arcs | gaps | the cover] -- two
// arcs as [A ascending | B DEscending], both ending at the gap between them -- the arcs are independent bands with a border on the
// separator's columns: ~(P/k + separator) dependent block columns instead of P.  The cut is searched: where the ring opens (a stride of
// rotations; an open band = no closing gap, rotation 0), how wide the gaps are, where the gaps sit so that every arc but the last ends
// on a tile boundary (16 poses = 3 tiles), scored by the block columns on the longest path.  The result does not depend on the number
// of threads: every choice between candidates is by a total order.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace mcp {

struct PoseCut {
  std::vector<int> order;      // position in the new order -> index in the order the graph was given in
  std::vector<int> segs;       // first tile of every chain of the plan, the separator's last (empty: one chain)
  bool relabelled = false, found = false, taken = false, looked_at_order = false;
  int k = 0, r = 0, g_first = 0, g_last = 0, steps = 0, t_all = 0, ncover = 0, sep = 0, arc_len[6] = {0, 0, 0, 0, 0, 0};
  double dist_add = 0, dist_cm = 0;      // mean distance of two coupled poses in the given order (round the ring) and breadth-first
};

typedef unsigned long long cut_u64;
// adj: [nf][W = (nf + 63)/64] bit rows of the coupling graph (symmetric, no diagonal; permuted in place if the poses are relabelled).
// par(fn): runs fn(tid) for tid in [0, T), possibly on T threads; lap(name): phase timer of the caller's trace (may do nothing).
inline void pose_cut(std::vector<cut_u64>& adj, int nf, int max_arcs_wanted, int T, const std::function<void(const std::function<void(int)>&)>& par,
                     const std::function<void(const char*)>& lap, PoseCut& out) {
  typedef cut_u64 u64;
  const int W = (nf + 63)/64;
  auto lo_of = [&](int tid, long n) { return (long)(n*tid/T); };
  std::vector<int> cur(nf); for (int i = 0; i < nf; ++i) cur[i] = i;      // position -> given index, through the relabelling
  {
    // An add order that is not a trajectory's (MCPTAM hands its key frames over in the order of a std::set of pointers): when the couplings
    // lie far apart in it, the poses are first relabelled breadth-first from a far end of the graph (Cuthill-McKee: neighbours by degree),
    // which lays any trajectory out as a band -- a ring as a band of twice its width -- and the cut below is searched in THAT order.
    bool relabelled = false;
    {
      auto deg_of = [&](int u) { int d = 0; for (int q = 0; q < W; ++q) d += __builtin_popcountll(adj[(size_t)u*W + q]); return d; };
      std::vector<int> dg(nf); long dist_add = 0, nedge = 0;
      for (int u = 0; u < nf; ++u) { dg[u] = deg_of(u);
        for (int q = 0; q < W; ++q) for (u64 b = adj[(size_t)u*W + q]; b; b &= b - 1) { const int v = 64*q + __builtin_ctzll(b); const int d = std::abs(u - v); dist_add += std::min(d, nf - d); ++nedge; } }
      if (nedge > 0 && dist_add > 12*nedge) {             // (mean distance round the ring above 12 poses: a trajectory in add order stays well below)
        std::vector<int> cm; cm.reserve(nf); std::vector<unsigned char> seen(nf, 0);
        auto bfs = [&](int start, std::vector<int>& out) {
          out.clear(); std::vector<unsigned char> mark(nf, 0); std::vector<int> nb;
          out.push_back(start); mark[start] = 1;
          for (size_t h = 0; h < out.size(); ++h) {
            const int u = out[h]; nb.clear();
            for (int q = 0; q < W; ++q) for (u64 b = adj[(size_t)u*W + q]; b; b &= b - 1) { const int v = 64*q + __builtin_ctzll(b); if (!mark[v] && !seen[v]) { mark[v] = 1; nb.push_back(v); } }
            std::sort(nb.begin(), nb.end(), [&](int x, int y) { return dg[x] != dg[y] ? dg[x] < dg[y] : x < y; });
            for (int v : nb) out.push_back(v);
          }
        };
        std::vector<int> comp;
        for (;;) {
          int s0 = -1; for (int u = 0; u < nf; ++u) if (!seen[u] && (s0 < 0 || dg[u] < dg[s0])) s0 = u;
          if (s0 < 0) break;
          bfs(s0, comp); bfs(comp.back(), comp); bfs(comp.back(), comp);       // (a far end of the component, twice refined)
          for (int u : comp) { seen[u] = 1; cm.push_back(u); }
        }
        long dist_cm = 0; std::vector<int> pos(nf); for (int i = 0; i < nf; ++i) pos[cm[i]] = i;
        for (int u = 0; u < nf; ++u) for (int q = 0; q < W; ++q) for (u64 b = adj[(size_t)u*W + q]; b; b &= b - 1) dist_cm += std::abs(pos[u] - pos[64*q + __builtin_ctzll(b)]);
        if (2*dist_cm < dist_add) {
          std::vector<u64> adj2((size_t)nf*W, 0); std::vector<int> fp2(nf);
          for (int i = 0; i < nf; ++i) { fp2[i] = cur[cm[i]];
            for (int q = 0; q < W; ++q) for (u64 b = adj[(size_t)cm[i]*W + q]; b; b &= b - 1) { const int v = pos[64*q + __builtin_ctzll(b)]; adj2[(size_t)i*W + (v >> 6)] |= 1ull << (v & 63); } }
          adj.swap(adj2); cur.swap(fp2);
          relabelled = true;
        }
        out.dist_add = (double)dist_add/nedge; out.dist_cm = (double)dist_cm/nedge; out.looked_at_order = true;
      }
    }
    auto tiles_of = [](int nposes) { return (6*nposes + CH_NB - 1)/CH_NB; };
    auto set_range = [&](u64* m, int r, int lo, int hi) { for (int q = lo; q < hi; ++q) { const int u = (q + r) % nf; m[u >> 6] |= 1ull << (u & 63); } };      // positions [lo, hi) of the ring opened at r
    auto touches = [&](int u, const u64* m) { const u64* a = &adj[(size_t)u*W]; int c = 0; for (int q = 0; q < W; ++q) c += __builtin_popcountll(a[q] & m[q]); return c; };
    lap("  chains: coupling graph");
    // A cut: the ring opened at r into k arcs (2 ... MAXA) of len[i] poses, gap g[i] behind arc i (the last gap closes the ring; 0 = an open band)
    constexpr int MAXA = 6;
    static_assert(MAXA + 1 <= CP_MAX_SEG, "a chain of the plan per arc + the separator's");
    struct Cut { int steps, sep, k, r, g[MAXA], len[MAXA];
                 bool operator<(const Cut& o) const { return steps != o.steps ? steps < o.steps : k != o.k ? k < o.k : sep != o.sep ? sep < o.sep : r != o.r ? r < o.r : g[0] != o.g[0] ? g[0] < o.g[0] : g[1] < o.g[1]; } };
    constexpr int KEEP = 4;
    Cut none; std::memset(&none, 0, sizeof none); none.steps = 1 << 30;
    auto arc_masks = [&](const Cut& c, u64* m /* [k][W] */) {
      std::fill(m, m + (size_t)c.k*W, 0);
      int p0 = 0;
      for (int i = 0; i < c.k; ++i) { set_range(m + (size_t)i*W, c.r, p0, p0 + c.len[i]); p0 += c.len[i] + c.g[i]; }
    };
    // stage 1: (rotation, gaps) with arcs of equal length; the cover estimated as half the poses that have a coupling into another arc
    // (two arcs: the smaller side); the best four go on
    static const int gaps[] = {0, 8, 16, 32};
    const int rstride = std::max(1, nf/40), nrot = (nf + rstride - 1)/rstride;
    const int max_arcs = std::max(2, std::min(max_arcs_wanted, MAXA));
    std::vector<Cut> best_t((size_t)T*KEEP, none);
    auto keep = [&](Cut* top, const Cut& c) { for (int k = 0; k < KEEP; ++k) if (c < top[k]) { for (int q = KEEP - 1; q > k; --q) top[q] = top[q - 1]; top[k] = c; break; } };
    par([&](int tid) {
      std::vector<u64> m((size_t)MAXA*W), oth(W);
      Cut* top = &best_t[(size_t)tid*KEEP];
      auto score = [&](Cut c) {
        arc_masks(c, m.data());
        int nx[MAXA] = {}, p0 = 0, lmax = 0, nxs = 0, gs = 0;
        for (int i = 0; i < c.k; ++i) {
          for (int q = 0; q < W; ++q) { oth[q] = 0; for (int j = 0; j < c.k; ++j) if (j != i) oth[q] |= m[(size_t)j*W + q]; }
          for (int q = p0; q < p0 + c.len[i]; ++q) nx[i] += touches((q + c.r) % nf, oth.data()) != 0;
          p0 += c.len[i] + c.g[i]; lmax = std::max(lmax, c.len[i]); nxs += nx[i]; gs += c.g[i];
        }
        c.sep = gs + (c.k == 2 ? std::min(nx[0], nx[1]) : nxs/2);
        c.steps = tiles_of(lmax) + tiles_of(c.sep);
        keep(top, c);
      };
      for (int ri = (int)lo_of(tid, nrot), re = (int)lo_of(tid + 1, nrot); ri < re; ++ri) {
        const int r = ri*rstride;
        for (int g1 : gaps) { if (g1 == 0) continue;
          for (int g2 : gaps) {
            if (g2 == 0 && r != 0) continue;
            const int rest = nf - g1 - g2, la = rest/2, lb = rest - la;
            Cut c = none; c.k = 2; c.r = r; c.g[0] = g1; c.g[1] = g2; c.len[0] = la; c.len[1] = lb;
            if (la >= 32) score(c);
          }
          for (int k = 3; k <= max_arcs && g1 <= 16; ++k) {
            const int rest = nf - k*g1, l = rest/k;
            if (l < 32) break;
            Cut c = none; c.k = k; c.r = r;
            for (int i = 0; i < k; ++i) { c.g[i] = g1; c.len[i] = i + 1 < k ? l : rest - (k - 1)*l; }
            score(c);
            if (r == 0) {                       // an open band: nothing behind the last arc
              const int rest2 = nf - (k - 1)*g1, l2 = rest2/k;
              c.g[k - 1] = 0; for (int i = 0; i < k; ++i) c.len[i] = i + 1 < k ? l2 : rest2 - (k - 1)*l2;
              score(c);
            }
          }
        }
      }
    });
    Cut top[KEEP] = {none, none, none, none};
    for (const Cut& c : best_t) if (c.steps < (1 << 30)) keep(top, c);
    lap("  chains: cuts");
    // stage 2, per kept cut: the gaps behind the arcs but the last slid one after the other over +- 20 poses (pass p slides gap p, the
    // gaps before it where their cover was smallest), the cover taken greedily (most couplings into other arcs first, degrees kept up to
    // date); every arc but the last a multiple of 16 poses; a separator under three tiles takes the poses at the last arc's end
    struct Fine { int steps = 1 << 30, cut = 0, s[MAXA] = {}, ncover = 0; std::vector<int> arc[MAXA], S;
                  bool better(const Fine& o) const { if (steps != o.steps) return steps < o.steps; if (cut != o.cut) return cut < o.cut; for (int i = 0; i < MAXA; ++i) if (s[i] != o.s[i]) return s[i] < o.s[i]; return false; } };
    const int t_all = tiles_of(nf);
    constexpr int SLIDE = 20;
    // shifts s[0 .. k-2]; need_aligned: how many leading arcs must end on a tile boundary; a partial pass (need_aligned < k - 1) returns with steps = 0
    auto refine_cut = [&](int ci, const int* sh, int need_aligned, Fine& out, std::vector<u64>& m, std::vector<int>& deg, std::vector<int>& arc_of, std::vector<int>& cover) -> bool {
      Cut c = top[ci];
      for (int i = 0; i + 1 < c.k; ++i) { c.len[i] += sh[i]; c.len[i + 1] -= sh[i]; }
      for (int i = 0; i < c.k; ++i) if (c.len[i] < 16) return false;
      arc_masks(c, m.data());
      u64* all = m.data() + (size_t)MAXA*W;            // union of the arcs
      for (int q = 0; q < W; ++q) { all[q] = 0; for (int i = 0; i < c.k; ++i) all[q] |= m[(size_t)i*W + q]; }
      std::fill(arc_of.begin(), arc_of.end(), -1);
      for (int i = 0; i < c.k; ++i) for (int q = 0; q < W; ++q) for (u64 b = m[(size_t)i*W + q]; b; b &= b - 1) arc_of[64*q + __builtin_ctzll(b)] = i;
      auto crossing = [&](int u) { const u64* a = &adj[(size_t)u*W]; const u64* mine = &m[(size_t)arc_of[u]*W]; int d = 0; for (int q = 0; q < W; ++q) d += __builtin_popcountll(a[q] & all[q] & ~mine[q]); return d; };
      for (int u = 0; u < nf; ++u) deg[u] = arc_of[u] >= 0 ? crossing(u) : 0;
      cover.clear();
      for (;;) {
        int bv = -1, bd = 0;
        for (int u = 0; u < nf; ++u) if (deg[u] > bd) { bd = deg[u]; bv = u; }
        if (bv < 0) break;
        if ((int)cover.size() > nf/4) return false;            // (a graph without a small separator: give this cut up)
        const u64* a = &adj[(size_t)bv*W]; const u64* mine = &m[(size_t)arc_of[bv]*W];
        for (int q = 0; q < W; ++q) for (u64 b = a[q] & all[q] & ~mine[q]; b; b &= b - 1) --deg[64*q + __builtin_ctzll(b)];
        m[(size_t)arc_of[bv]*W + (bv >> 6)] &= ~(1ull << (bv & 63)); all[bv >> 6] &= ~(1ull << (bv & 63));
        arc_of[bv] = -1; deg[bv] = 0; cover.push_back(bv);
      }
      Fine f; f.cut = ci; for (int i = 0; i + 1 < c.k; ++i) f.s[i] = sh[i]; f.ncover = (int)cover.size();
      int p0 = 0;
      for (int i = 0; i < c.k; ++i) {
        // two arcs: the second DEscending (both end at the gap between them; an open band's far end has nothing behind it)
        if (c.k == 2 && i == 1) { for (int q = p0 + c.len[i] - 1; q >= p0; --q) { const int u = (q + c.r) % nf; if (arc_of[u] == i) f.arc[i].push_back(u); } }
        else for (int q = p0; q < p0 + c.len[i]; ++q) { const int u = (q + c.r) % nf; if (arc_of[u] == i) f.arc[i].push_back(u); }
        p0 += c.len[i] + c.g[i];
      }
      for (int i = 0; i < need_aligned; ++i) if (f.arc[i].size() % 16 || f.arc[i].size() < 16) return false;
      if (need_aligned < c.k - 1) { out = std::move(f); out.steps = 0; return true; }
      std::vector<int>& last = f.arc[c.k - 1];
      int inarcs = 0; for (int i = 0; i < c.k; ++i) inarcs += (int)f.arc[i].size();
      while (last.size() > 16 && t_all - 6*inarcs/CH_NB < 3) { f.S.push_back(last.back()); last.pop_back(); --inarcs; }
      p0 = 0;
      for (int i = 0; i < c.k; ++i) { for (int q = p0 + c.len[i]; q < p0 + c.len[i] + c.g[i]; ++q) f.S.push_back((q + c.r) % nf); p0 += c.len[i] + c.g[i]; }
      std::sort(cover.begin(), cover.end()); for (int u : cover) f.S.push_back(u);
      int tb[MAXA + 1] = {}, cum = 0, tmax = 0;
      for (int i = 0; i < c.k; ++i) { cum += (int)f.arc[i].size(); tb[i + 1] = 6*cum/CH_NB; tmax = std::max(tmax, tb[i + 1] - tb[i]); if (tb[i + 1] - tb[i] < 3) return false; }
      if (t_all - tb[c.k] < 3) return false;
      f.steps = tmax + (t_all - tb[c.k]);
      out = std::move(f);
      return true;
    };
    std::vector<Fine> fine_t(T);
    int fixed[KEEP][MAXA] = {};                          // per kept cut: the slides settled by the passes so far
    bool alive[KEEP]; for (int ci = 0; ci < KEEP; ++ci) alive[ci] = top[ci].steps < (1 << 30) && 10*top[ci].steps <= 9*t_all;      // (nothing to gain by a cut: not refined)
    for (int pass = 0; pass + 1 < max_arcs; ++pass) {
      // pass p: cuts of k = p + 2 arcs finish here (their last free gap), cuts of more arcs settle gap p by the smallest cover
      std::vector<Fine> part_t((size_t)T*KEEP);
      bool any = false; for (int ci = 0; ci < KEEP; ++ci) any = any || (alive[ci] && top[ci].k >= pass + 2);
      if (!any) break;
      par([&](int tid) {
        std::vector<u64> m((size_t)(MAXA + 1)*W); std::vector<int> deg(nf), arc_of(nf), cover;
        for (int item = tid; item < KEEP*(2*SLIDE + 1); item += T) {
          const int ci = item/(2*SLIDE + 1), sl = item % (2*SLIDE + 1) - SLIDE;
          const Cut& c = top[ci];
          if (!alive[ci] || c.k < pass + 2) continue;
          int sh[MAXA]; for (int i = 0; i < MAXA; ++i) sh[i] = fixed[ci][i];
          sh[pass] = sl;
          Fine f;
          if (!refine_cut(ci, sh, pass + 1, f, m, deg, arc_of, cover)) continue;
          if (c.k == pass + 2) { if (f.better(fine_t[tid])) fine_t[tid] = std::move(f); }
          else { Fine& b = part_t[(size_t)tid*KEEP + ci];
            if (b.steps != 0 || f.ncover < b.ncover || (f.ncover == b.ncover && (std::abs(sl) < std::abs(b.s[pass]) || (std::abs(sl) == std::abs(b.s[pass]) && sl < b.s[pass])))) b = std::move(f); }
        }
      });
      for (int ci = 0; ci < KEEP; ++ci) {
        if (!alive[ci] || top[ci].k <= pass + 2) continue;
        const Fine* b = nullptr;
        for (int t = 0; t < T; ++t) { const Fine& f = part_t[(size_t)t*KEEP + ci];
          if (f.steps == 0 && (!b || f.ncover < b->ncover || (f.ncover == b->ncover && (std::abs(f.s[pass]) < std::abs(b->s[pass]) || (std::abs(f.s[pass]) == std::abs(b->s[pass]) && f.s[pass] < b->s[pass]))))) b = &f; }
        if (b) fixed[ci][pass] = b->s[pass]; else alive[ci] = false;
      }
    }
    Fine* fb = nullptr;
    for (auto& f : fine_t) if (f.steps < (1 << 30) && (!fb || f.better(*fb))) fb = &f;
    const Cut c1 = fb ? top[fb->cut] : none;
    const bool take = fb && 10*fb->steps <= 8*t_all;        // (worth it from a fifth fewer dependent block columns)
    out.relabelled = relabelled; out.found = fb != nullptr; out.taken = take; out.t_all = t_all;
    if (fb) { out.k = c1.k; out.r = c1.r; out.g_first = c1.g[0]; out.g_last = c1.g[c1.k - 1]; out.steps = fb->steps; out.ncover = fb->ncover; out.sep = (int)fb->S.size();
              for (int i = 0; i < c1.k; ++i) out.arc_len[i] = (int)fb->arc[i].size(); }
    out.segs.clear();
    if (take) {
      out.order.clear(); out.order.reserve(nf);
      out.segs.assign(1, 0);
      int cum = 0;
      for (int i = 0; i < c1.k; ++i) { for (int u : fb->arc[i]) out.order.push_back(cur[u]); cum += (int)fb->arc[i].size(); out.segs.push_back(6*cum/CH_NB); }
      for (int u : fb->S) out.order.push_back(cur[u]);
    } else out.order = cur;
  }
}

}  // namespace mcp
