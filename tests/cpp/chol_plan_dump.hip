// Host-logic check of mcp::CholPlan (mcptam_amd/csrc/ba_chol.h): reads "n ntc" and an ntc x ntc 0/1 tile pattern from stdin,
// prints the symbolic plan.  The device buffers cannot be allocated on a box without a GPU (build() then returns -1); the
// host-side vectors this program prints are complete before that point.
#include "../../mcptam_amd/csrc/ba_chol.h"
#include <cstdio>
#include <vector>

int main() {
  int n = 0, ntc = 0;
  if (scanf("%d %d", &n, &ntc) != 2) return 2;
  std::vector<unsigned char> pat((size_t)ntc*ntc, 0);
  for (size_t i = 0; i < pat.size(); ++i) { int v = 0; if (scanf("%d", &v) != 1) return 2; pat[i] = (unsigned char)v; }
  mcp::CholPlan plan;
  (void)plan.build(n, pat);
  (void)hipGetLastError();
  printf("%d %d %d\n", plan.n, plan.ntc, plan.ntr);
  for (int k = 0; k < plan.ntc; ++k) {
    printf("step %d:", k);
    for (int i = plan.step_start[k]; i < plan.step_start[k + 1]; ++i) printf(" %d,%d", plan.step_tiles[i] >> 16, plan.step_tiles[i] & 0xffff);
    printf("\n");
  }
  for (int r = 0; r < plan.ntc; ++r) {
    printf("row %d:", r);
    for (int i = plan.row_start[r]; i < plan.row_start[r + 1]; ++i) printf(" %d", plan.row_tiles[i]);
    printf("\n");
  }
  return 0;
}
