// atan_cr (mcptam_amd/csrc/atan_cr.h, the device's correctly rounded arctangent) against round(atanq) from libquadmath,
// the route the oracle takes: every argument must give the same double.  Usage: atan_cr_check [count] -> prints
// "mismatches <n> of <count>", how often the fast (Ziv) path had to hand over to the double-double evaluation for the arguments a
// camera produces ("angle-uniform slow <n> of <m>"), and, for information, how often glibc's atan differs from the correctly rounded
// value.  Built twice by the test: as is, and with -DMCP_ATAN_NO_FAST (the double-double evaluation alone must pass as well).
#include <quadmath.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "../../mcptam_amd/csrc/atan_cr.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double uni() { return (double)(next() >> 11)*(1.0/9007199254740992.0); }

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 4000000;
  long bad = 0, glibc_bad = 0, cam_n = 0, cam_slow = 0;
  for (long i = 0; i < N; ++i) {
    double x;
    switch (i % 10) {
      case 0: x = uni()*2.0 - 1.0; break;                          // [-1, 1]
      case 1: x = (uni()*2.0 - 1.0)*64.0; break;                   // wide
      case 2: x = ldexp(uni() + 0.5, (int)(next() % 120) - 60); break;      // many binades
      case 3: x = (double)(next() % 65)/64.0 + (uni() - 0.5)*ldexp(1.0, -(int)(next() % 50)); break;   // around the table nodes
      case 4: x = 1.0 + (uni() - 0.5)*ldexp(1.0, -(int)(next() % 52)); break;                          // around 1
      case 5: { uint64_t b = next(); memcpy(&x, &b, 8); if (!(x == x) || std::isinf(x)) x = 0.5; break; }  // any bit pattern
      case 6: x = tan((uni() - 0.5)*3.0); break;                   // uniform in the angle (what a camera sees)
      case 7: x = -ldexp(uni(), -(int)(next() % 1000)); break;     // tiny / subnormal-ish
      case 8: x = 1.0/((double)(next() % 64 + 1)/64.0 + (uni() - 0.5)*ldexp(1.0, -(int)(next() % 50))); break;   // reciprocals of the table nodes
      default: x = ldexp(uni() + 0.5, (int)(next() % 300)); break; // large, past the float range of the fast path's node choice
    }
#if !defined(MCP_ATAN_NO_FAST)
    if (i % 10 == 6) { double f; ++cam_n; if (!mcp_atan::atan_fast(fabs(x), fabs(x) > 1.0, &f)) ++cam_slow; }
#endif
    const double want = (double)atanq((__float128)x);
    const double got = mcp_atan::atan_cr(x);
    if (memcmp(&want, &got, 8) != 0) { if (bad < 10) printf("x=%a want=%a got=%a\n", x, want, got); ++bad; }
    const double gl = atan(x);
    if (memcmp(&want, &gl, 8) != 0) ++glibc_bad;
  }
  printf("mismatches %ld of %ld (glibc atan differs from the correctly rounded value for %ld)\n", bad, N, glibc_bad);
  printf("angle-uniform slow %ld of %ld\n", cam_slow, cam_n);
  const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, 1e300, 5e-324, 0x1p-1022};
  for (double v : sp) { const double w = (double)atanq((__float128)v), g = mcp_atan::atan_cr(v); if (memcmp(&w, &g, 8)) { printf("special %a: want %a got %a\n", v, w, g); ++bad; } }
  return bad ? 1 : 0;
}
