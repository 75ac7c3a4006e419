// Host-logic check of mcp::CholPersist (mcptam_amd/csrc/ba_chol2.h): reads "n ntc" and an ntc x ntc 0/1 tile pattern from
// stdin, prints the plan of the one-launch factorisation (slots, helpers with their update lists, far tiles of the
// back-substitution).  Device buffers cannot be allocated without a GPU (build() then returns -1); the host vectors are complete.
#include "../../mcptam_amd/csrc/ba_chol.h"
#include <cstdio>
#include <vector>

int main() {
  int n = 0, ntc = 0, dense = 0;
  if (scanf("%d %d %d", &n, &ntc, &dense) != 3) return 2;
  std::vector<unsigned char> pat((size_t)ntc*ntc, 0);
  for (size_t i = 0; i < pat.size(); ++i) { int v = 0; if (scanf("%d", &v) != 1) return 2; pat[i] = (unsigned char)v; }
  mcp::CholPlan plan;                       // the assembly's tile list (what S holds) comes from the per-step plan
  (void)plan.build(n, dense ? std::vector<unsigned char>() : pat);
  (void)hipGetLastError();
  mcp::CholPersist& P = plan.persist;
  if (P.n != n) { (void)P.build(n, dense ? std::vector<unsigned char>() : pat, dense ? std::vector<int>() : plan.all_tiles); (void)hipGetLastError(); }
  printf("%d %d %d %d %d %d %d\n", P.n, P.ntc, P.nslots, P.nbslots, P.nhelpers, mcp::CP_W, mcp::CP_BACK_NEAR);
  for (int i = 0; i <= P.ntc; ++i) { for (int j = 0; j < P.ntc; ++j) printf("%d ", P.slot_of[(size_t)i*P.ntc + j]); printf("\n"); }
  for (int i = 0; i <= P.ntc; ++i) { for (int j = 0; j < P.ntc; ++j) printf("%d ", P.bslot_of[(size_t)i*P.ntc + j]); printf("\n"); }
  for (int i = 0; i <= P.ntc; ++i) printf("%d ", P.delta_of[i]);
  printf("\n");
  for (const mcp::CpHelper& h : P.helpers) {
    printf("%d %d %d %d %d %d %d %d %d %d:", h.ti, h.tj, h.slot, h.dslot, h.kind, h.in_s, h.nupd, h.pre, h.pre_flag, h.pre_diag);
    for (int u = 0; u < h.nupd; ++u) printf(" %d,%d", P.upd[h.upd0 + u].x, P.upd[h.upd0 + u].y);
    printf("\n");
  }
  for (int c = 0; c < P.ntc; ++c) {
    printf("far %d:", c);
    for (int f = P.far_start[c]; f < P.far_start[c + 1]; ++f) printf(" %d,%d", P.far_slot[f], P.far_row[f]);
    printf("\n");
  }
  return 0;
}
