// Unit check of regs_vote_select (mcptam_amd/csrc/ba_select.h): the rank-k key of register-held keys by votes, against std::sort.
// Build: hipcc --offload-arch=gfx950 -O2 -I mcptam_amd/csrc tests/cpp/vote_select_check.hip -o vote_select_check ; prints "ok <cases>" or the failing cases.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "ba_select.h"
using namespace mcp;
constexpr int NT = 512, PPT = 2;
__global__ void __launch_bounds__(NT) k_sel(int n, const double* x, const unsigned char* live, unsigned int kk, unsigned long long lo, int s, unsigned long long* out) {
  __shared__ unsigned int vt[2][NT/64][SELV_NB + 1];
  __shared__ unsigned long long cand[SELV_MAXC];
  __shared__ unsigned int rk[SELV_MAXC];
  unsigned long long key[PPT]; bool ok[PPT], has[PPT];
  for (int q = 0; q < PPT; ++q) { const int i = threadIdx.x + q*NT; ok[q] = i < n && live[i]; key[q] = ok[q] ? (unsigned long long)__double_as_longlong(fabs(x[i])) : 0ull;
                                  has[q] = __ballot(ok[q]) != 0ull; }
  unsigned long long r = 0ull;
  for (int rep = 0; rep < 3; ++rep) {                  // (several calls in a row, as the ten iterations make them: the LDS tables are reused)
    const bool have = regs_vote_select<NT, PPT>(key, ok, has, kk, lo, s, vt, cand, rk, r);
    if (threadIdx.x == 0) { out[2*rep] = have ? 1ull : 0ull; out[2*rep + 1] = r; }
    __syncthreads();
  }
}
int main() {
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  double* dx; unsigned char* dl; unsigned long long* dout;
  (void)hipMalloc(&dx, 8*1024); (void)hipMalloc(&dl, 1024); (void)hipMalloc(&dout, 8*6);
  int cases = 0, bad = 0, missed = 0;
  for (int n : {1024, 1000, 513, 130, 64, 3, 1})
    for (int kind = 0; kind < 8; ++kind)
      for (int gaps = 0; gaps < 2; ++gaps) {
        std::vector<double> x(n); std::vector<unsigned char> live(n, 1);
        for (int i = 0; i < n; ++i) {
          switch (kind) {
            case 0: { const double v[3] = {0.25, 1.0, 4.0}; x[i] = v[g()%3]; } break;
            case 1: x[i] = std::exp2(60.0*U(g) - 30.0); break;
            case 2: x[i] = 0.5625; break;
            case 3: x[i] = U(g) < 0.4 ? 0.0 : 0.1 + 3.0*U(g); break;
            case 4: x[i] = std::exp(2.0*N(g)); break;
            case 5: x[i] = 1.0 + i*std::exp2(-50.0); break;
            case 6: x[i] = U(g) < 0.6 ? 0.0 : U(g); break;
            default: x[i] = std::exp2(std::floor(8.0*U(g)))*(1.0 + std::floor(4.0*U(g))/4.0); break;
          }
          if (gaps && U(g) < 0.2) live[i] = 0;
        }
        std::vector<unsigned long long> keys;
        for (int i = 0; i < n; ++i) if (live[i]) { unsigned long long b; const double a = std::fabs(x[i]); std::memcpy(&b, &a, 8); keys.push_back(b); }
        if (keys.empty()) continue;
        std::sort(keys.begin(), keys.end());
        (void)hipMemcpy(dx, x.data(), 8*n, hipMemcpyHostToDevice); (void)hipMemcpy(dl, live.data(), n, hipMemcpyHostToDevice);
        for (unsigned int kk : {(unsigned int)(keys.size()/2), 0u, (unsigned int)keys.size() - 1u}) {
          const unsigned long long want = keys[kk];
          const int e_true = (int)(want >> 52);
          // first brackets: 16 binades starting below, at and above the answer's (s = 52); eighths and 1/128ths of a binade around the
          // answer, around a key a quarter binade off, and beside it (s = 49, 45); single keys (s = 0) at and next to the answer
          struct W { unsigned long long lo; int s; };
          std::vector<W> ws;
          for (int off : {-11, -15, 0, -4, 3, -40}) { int e_lo = e_true + off; if (e_lo < 0) e_lo = 0; ws.push_back({(unsigned long long)e_lo << 52, 52}); }
          for (int s2 : {49, 45, 3, 0})
            for (long long rel : {-8LL, -15LL, 0LL, -16LL, 1LL, -3LL}) {
              const long long d = rel*(1LL << s2);
              unsigned long long lo = (d < 0 && (unsigned long long)(-d) > want) ? 0ull : want + (unsigned long long)d;
              ws.push_back({lo, s2});
            }
          for (const W& w : ws) {
            const bool inside = want >= w.lo && want - w.lo < ((unsigned long long)SELV_NB << w.s);
            hipLaunchKernelGGL(k_sel, dim3(1), dim3(NT), 0, 0, n, dx, dl, kk, w.lo, w.s, dout);
            unsigned long long out[6]; (void)hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
            ++cases;
            for (int rep = 0; rep < 3; ++rep) {
              const bool have = out[2*rep] != 0;
              if (have != inside) { ++missed; printf("window: n %d kind %d gaps %d kk %u lo %016llx s %d rep %d: have %d, inside %d (want %016llx)\n", n, kind, gaps, kk, w.lo, w.s, rep, (int)have, (int)inside, want); }
              else if (have && out[2*rep + 1] != want) { ++bad; printf("WRONG: n %d kind %d gaps %d kk %u lo %016llx s %d rep %d: got %016llx want %016llx\n", n, kind, gaps, kk, w.lo, w.s, rep, out[2*rep + 1], want); }
            }
          }
        }
      }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
  if (!bad && !missed) printf("ok %d\n", cases); else printf("FAILED: %d wrong, %d window mismatches of %d\n", bad, missed, cases);
  return (bad || missed) ? 1 : 0;
}
