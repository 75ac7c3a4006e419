// atan_cr.h -- correctly rounded double-precision arctangent in double-double arithmetic (host + gfx950 device).
//
// Why it exists: TaylorCamera::Project takes theta = atan(z / n) (src/TaylorCamera.cc:243), and in the tracker the
// projected position feeds CVD::transform's truncating byte conversion (src/PatchFinder.cc:164-165): in a flat image
// region the last ulp of atan decides between grey level g and g-1 of a template pixel.  The device math library and
// glibc differ in that last ulp (glibc 2.35's atan is itself not correctly rounded: 0.07 % of arguments), so neither can
// define the bytes.  The correctly rounded value is the one platform-independent definition: the device path computes it
// here (error < 2^-100 before the final rounding), the oracle computes it independently through libquadmath's atanq and
// rounds -- two routes to the same bits.
//
// What "correctly rounded" rests on: the fast path below carries a rounding test (Ziv), so whatever it returns is the correctly rounded
// value; an argument it cannot decide goes to the double-double evaluation (error < 2^-100), whose result is rounded without a further test --
// it would be wrong only for an argument within 2^-100 of a rounding boundary as well, i.e. with probability ~2^-46 per argument that reaches
// it (the known worst cases of atan in binary64 sit near 2^-110; none has been found by the 300 M-argument comparison with atanq, and
// the oracle's single cast of a 113-bit atanq has the same order of exposure).
//
// Method: |x| > 1 is inverted (atan x = pi/2 - atan 1/x, the reciprocal carried as a double-double); the argument u in
// [0, 1] is split at c = k/64 nearest to it, atan u = atan c + atan r with r = (u - c)/(1 + u c) in double-double
// (|r| <= 1/128); atan r = r - r^3/3 + r^5/5 - r^7/7 in double-double plus the r^9.. tail in double (it is below
// 2^-59 of the result); atan c comes from a 65-entry double-double table.  The sum is rounded once.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCP_ATAN_HD __host__ __device__ inline
#if defined(__HIP_DEVICE_COMPILE__)
#define MCP_ATAN_CONST static __device__ const      // device pass of hipcc: the table lives in device memory
#else
#define MCP_ATAN_CONST static const                 // host pass (cam_project is __host__ __device__)
#endif
#else                                               // plain C++ (tests/cpp/atan_cr_check.cpp, built with g++)
#include <math.h>
#define MCP_ATAN_HD static inline
#define MCP_ATAN_CONST static const
#endif

#if defined(__clang__)
#define MCP_NOCONTRACT _Pragma("clang fp contract(off)")     // first statement of a body: no fused multiply-add contraction in it
#else
#define MCP_NOCONTRACT                                        // g++ harness: built with -ffp-contract=off
#endif

namespace mcp_atan {

// atan(k/64), k = 0..64, as unevaluated sums hi + lo (generated with libquadmath atanq; scripts/gen_atan_table.c)
MCP_ATAN_CONST double kAtanHi[65] = {
  0x0p+0,  0x1.fff555bbb729bp-7,  0x1.ffd55bba97625p-6,  0x1.7fb818430da2ap-5,
  0x1.ff55bb72cfdeap-5,  0x1.3f59f0e7c559dp-4,  0x1.7ee182602f10fp-4,  0x1.be39ebe6f07c3p-4,
  0x1.fd5ba9aac2f6ep-4,  0x1.1e1fafb043727p-3,  0x1.3d6eee8c6626cp-3,  0x1.5c9811e3ec26ap-3,
  0x1.7b97b4bce5b02p-3,  0x1.9a6a8e96c8626p-3,  0x1.b90d7529260a2p-3,  0x1.d77d5df205736p-3,
  0x1.f5b75f92c80ddp-3,  0x1.09dc597d86362p-2,  0x1.18bf5a30bf178p-2,  0x1.278372057ef46p-2,
  0x1.362773707ebccp-2,  0x1.44aa436c2af0ap-2,  0x1.530ad9951cd4ap-2,  0x1.614840309cfe2p-2,
  0x1.6f61941e4def1p-2,  0x1.7d5604b63b3f7p-2,  0x1.8b24d394a1b25p-2,  0x1.98cd5454d6b18p-2,
  0x1.a64eec3cc23fdp-2,  0x1.b3a911da65c6cp-2,  0x1.c0db4c94ec9fp-2,  0x1.cde53432c1351p-2,
  0x1.dac670561bb4fp-2,  0x1.e77eb7f175a34p-2,  0x1.f40dd0b541418p-2,  0x1.0039c73c1a40cp-1,
  0x1.0657e94db30dp-1,  0x1.0c6145b5b43dap-1,  0x1.1255d9bfbd2a9p-1,  0x1.1835a88be7c13p-1,
  0x1.1e00babdefeb4p-1,  0x1.23b71e2cc9e6ap-1,  0x1.2958e59308e31p-1,  0x1.2ee628406cbcap-1,
  0x1.345f01cce37bbp-1,  0x1.39c391cd4171ap-1,  0x1.3f13fb89e96f4p-1,  0x1.445065b795b56p-1,
  0x1.4978fa3269ee1p-1,  0x1.4e8de5bb6ec04p-1,  0x1.538f57b89061fp-1,  0x1.587d81f732fbbp-1,
  0x1.5d58987169b18p-1,  0x1.6220d115d7b8ep-1,  0x1.66d663923e087p-1,  0x1.6b798920b3d99p-1,
  0x1.700a7c5784634p-1,  0x1.748978fba8e0fp-1,  0x1.78f6bbd5d315ep-1,  0x1.7d528289fa093p-1,
  0x1.819d0b7158a4dp-1,  0x1.85d69576cc2c5p-1,  0x1.89ff5ff57f1f8p-1,  0x1.8e17aa99cc05ep-1,
  0x1.921fb54442d18p-1,
};
MCP_ATAN_CONST double kAtanLo[65] = {
  0x0p+0,  -0x1.220c39d4dff5p-61,  -0x1.5ec431444912cp-60,  -0x1.86ef8f794f105p-63,
  -0x1.c934d86d23f1dp-60,  0x1.ac4ce285df847p-58,  -0x1.cfb654c0c3d98p-58,  0x1.f7b8f29a05987p-58,
  -0x1.cd37686760c17p-59,  -0x1.b485914dacf8cp-59,  0x1.61a3b0ce9281bp-57,  -0x1.054ab2c010f3dp-58,
  0x1.347b0b4f881cap-58,  0x1.cf601e7b4348ep-59,  0x1.17b10d2e0e5aap-61,  0x1.c648d1534597ep-57,
  0x1.8ab6e3cf7afbdp-57,  0x1.62e47390cb865p-56,  0x1.30ca4748b1bf8p-57,  -0x1.077cdd36dfc81p-56,
  -0x1.963a544b672d8p-57,  -0x1.5d5e43c55b3bap-56,  -0x1.2566480884082p-57,  -0x1.a725715711fp-56,
  -0x1.c63aae6f6e918p-56,  0x1.69c885c2b249ap-56,  0x1.b6d0ba3748fa8p-56,  0x1.9e6c988fd0a77p-56,
  -0x1.24dec1b50b7ffp-56,  0x1.ae187b1ca504p-56,  -0x1.cc1ce70934c34p-56,  -0x1.a2cfa4418f1adp-56,
  0x1.a2b7f222f65e2p-56,  0x1.0e53dc1bf3435p-56,  -0x1.a3992dc382a23p-57,  -0x1.b32c949c9d593p-55,
  -0x1.d5b495f6349e6p-56,  0x1.974fa13b5404fp-58,  -0x1.2bdaee1c0ee35p-58,  0x1.c621cec00c301p-55,
  -0x1.928df287a668fp-58,  0x1.c421c9f38224ep-57,  -0x1.09e73b0c6c087p-56,  0x1.c5d5e9ff0cf8dp-55,
  0x1.1021137c71102p-55,  -0x1.2304331d8bf46p-55,  0x1.ecf8b492644fp-56,  -0x1.f76d0163f79c8p-56,
  0x1.2419a87f2a458p-56,  0x1.4a33dbeb3796cp-55,  -0x1.1bb74abda520cp-55,  -0x1.5e5c9d8c5a95p-56,
  0x1.0028e4bc5e7cap-57,  -0x1.2b785350ee8c1p-57,  -0x1.6ea6febe8bbbap-56,  -0x1.a80386188c50ep-55,
  -0x1.8c34d25aadef6p-56,  0x1.7b2a6165884a2p-59,  0x1.406a08980374p-55,  0x1.560821e2f3aa9p-55,
  -0x1.bf76229d3b917p-56,  0x1.6b66e7fc8b8c4p-57,  -0x1.55b9a5e177a1bp-55,  -0x1.ec182ab042f61p-56,
  0x1.1a62633145c07p-55,
};
// pi/2
#define MCP_PIO2_HI 0x1.921fb54442d18p+0
#define MCP_PIO2_LO 0x1.1a62633145c07p-54
#define MCP_C3_HI 0x1.5555555555555p-2
#define MCP_C3_LO 0x1.5555555555555p-56
#define MCP_C5_HI 0x1.999999999999ap-3
#define MCP_C5_LO -0x1.999999999999ap-57
#define MCP_C7_HI 0x1.2492492492492p-3
#define MCP_C7_LO 0x1.2492492492492p-57

struct dd { double hi, lo; };

// error-free transformations (no contraction may fuse these: the image translation unit is built with -ffp-contract=off,
// the products use explicit fma)
MCP_ATAN_HD dd two_sum(double a, double b) { MCP_NOCONTRACT const double s = a + b, bb = s - a; dd r; r.hi = s; r.lo = (a - (s - bb)) + (b - bb); return r; }
MCP_ATAN_HD dd fast_two_sum(double a, double b) { MCP_NOCONTRACT const double s = a + b; dd r; r.hi = s; r.lo = b - (s - a); return r; }
MCP_ATAN_HD dd two_prod(double a, double b) { MCP_NOCONTRACT const double p = a*b; dd r; r.hi = p; r.lo = __builtin_fma(a, b, -p); return r; }
MCP_ATAN_HD dd dd_add(dd a, dd b) { MCP_NOCONTRACT
  dd s = two_sum(a.hi, b.hi); const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi; s = fast_two_sum(s.hi, s.lo); s.lo += t.lo; return fast_two_sum(s.hi, s.lo);
}
MCP_ATAN_HD dd dd_add_d(dd a, double b) { MCP_NOCONTRACT dd s = two_sum(a.hi, b); s.lo += a.lo; return fast_two_sum(s.hi, s.lo); }
MCP_ATAN_HD dd dd_neg(dd a) { MCP_NOCONTRACT dd r; r.hi = -a.hi; r.lo = -a.lo; return r; }
MCP_ATAN_HD dd dd_mul(dd a, dd b) { MCP_NOCONTRACT dd p = two_prod(a.hi, b.hi); p.lo += a.hi*b.lo + a.lo*b.hi; return fast_two_sum(p.hi, p.lo); }
MCP_ATAN_HD dd dd_mul_d(dd a, double b) { MCP_NOCONTRACT dd p = two_prod(a.hi, b); p.lo += a.lo*b; return fast_two_sum(p.hi, p.lo); }
MCP_ATAN_HD dd dd_div(dd a, dd b) { MCP_NOCONTRACT
  const double q1 = a.hi/b.hi;
  dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
  const double q2 = r.hi/b.hi;
  r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
  const double q3 = r.hi/b.hi;
  dd q = fast_two_sum(q1, q2);
  return dd_add_d(q, q3);
}

// Fast path (Ziv's strategy): the same range reduction with ONE division and the series carried as double + correction terms gives
// atan|x| = hi + lo with an error below 2^-80 of the result; if rounding hi + lo - e and hi + lo + e (e = 2^-75 of the result) agree,
// that double IS the correctly rounded value -- on any platform, whatever its division or fma do in the last place of the intermediate
// terms -- and the full double-double evaluation below is only needed for the ~2^-21 of the arguments that come closer to a rounding
// boundary.  Error budget (|r| <= 2^-7, V = result): the tail q (r^5/5 .. r^11/11 in double) 2^-50 |q| <= 2^-80 |r|; series truncation
// r^13/13 <= 2^-87 |r|; r, r^2, r^3, r^3/3 are double-doubles (errors ~2^-103 |r|); the table entries and pi/2 are exact to 2^-107;
// the final additions lose < 2^-104 V.  |r| <= V, and for |x| > 1 the result is >= pi/4 >= what it is subtracted from.
#define MCP_C9 0x1.c71c71c71c71cp-4
#define MCP_C11 0x1.745d1745d1746p-4
MCP_ATAN_HD bool atan_fast(double ax, bool inv, double* out, const double* tab_hi = kAtanHi, const double* tab_lo = kAtanLo) { MCP_NOCONTRACT
  // c = k/64 near u = ax (or 1/ax): any k within a hair of the nearest one keeps |r| <= 1/128 + 2^-22
  double rh, rl; int k;
  if (inv) {
    if (ax > 3.0e38) return false;                             // (beyond float range: the slow path's double-double reciprocal)
    const float uf = 1.0f/(float)ax;
    const double kf = __builtin_rint((double)uf*64.0);
    k = (int)kf;
    if (k == 0) {                                              // ax > 128: r = 1/ax
      rh = 1.0/ax;
      rl = __builtin_fma(-rh, ax, 1.0)*rh;
    } else {
      // r = (1/ax - c)/(1 + c/ax) = (1 - c ax)/(ax + c)
      const double c = kf*0.015625;
      const double p = c*ax, pl = __builtin_fma(c, ax, -p);    // c ax = p + pl exactly
      dd num = two_sum(1.0, -p); num.lo -= pl;                 // exact sum, then a tiny correction
      const dd den = two_sum(ax, c);
      const double id = 1.0/den.hi;
      rh = num.hi*id;
      rl = ((__builtin_fma(-rh, den.hi, num.hi) + num.lo) - rh*den.lo)*id;
    }
  } else {
    const double kf = __builtin_rint(ax*64.0);
    k = (int)kf;
    if (k == 0) { rh = ax; rl = 0.0; }
    else {
      const double c = kf*0.015625;
      const double n0 = ax - c;                                // exact (ax / c in [1/2, 2])
      const double p = ax*c, pl = __builtin_fma(ax, c, -p);
      dd den = two_sum(1.0, p); den.lo += pl;
      const double id = 1.0/den.hi;
      rh = n0*id;
      rl = (__builtin_fma(-rh, den.hi, n0) - rh*den.lo)*id;
    }
  }
  const double sh = rh*rh, sl = __builtin_fma(rh, rh, -sh) + 2.0*rh*rl;                      // r^2
  const double wh = rh*sh, wl = __builtin_fma(rh, sh, -wh) + (rh*sl + rl*sh);               // r^3
  const double th = wh*MCP_C3_HI, tl = __builtin_fma(wh, MCP_C3_HI, -th) + (wh*MCP_C3_LO + wl*MCP_C3_HI);      // r^3/3
  const double q = (wh*sh)*(MCP_C5_HI + sh*(-MCP_C7_HI + sh*(MCP_C9 - sh*MCP_C11)));       // r^5/5 - r^7/7 + r^9/9 - r^11/11
  const dd s1 = two_sum(tab_hi[k], rh);
  const dd s2 = two_sum(s1.hi, -th);
  const double low = s2.lo + (s1.lo + ((tab_lo[k] + rl) + (q - tl)));
  dd res = fast_two_sum(s2.hi, low);
  if (inv) {
    const dd p1 = two_sum(MCP_PIO2_HI, -res.hi);
    res = fast_two_sum(p1.hi, p1.lo + (MCP_PIO2_LO - res.lo));
  }
  const double e = res.hi*0x1p-75;
  const double a = res.hi + (res.lo - e), b = res.hi + (res.lo + e);
  *out = a;
  return a == b;
}

// (tab_hi / tab_lo: the 65-entry table; a kernel that keeps a copy in LDS passes it, so that the two look-ups of an arctangent are LDS
//  reads and do not queue behind the kernel's global stores in the vector-memory counter)
MCP_ATAN_HD double atan_cr(double x, const double* tab_hi = kAtanHi, const double* tab_lo = kAtanLo) { MCP_NOCONTRACT
  if (x != x) return x;
  const double ax = x < 0 ? -x : x;
  if (ax == 0.0) return x;
  const bool inv = ax > 1.0;
  dd u;
  if (ax > 1.8e16*1.8e16*1.8e16) {            // |x| > 2^162: atan = pi/2 to every bit (1/x < 2^-54 ulp); covers infinity
    const double r = MCP_PIO2_HI;
    return x < 0 ? -r : r;
  }
#if !defined(MCP_ATAN_NO_FAST)
  { double fast; if (atan_fast(ax, inv, &fast, tab_hi, tab_lo)) return x < 0 ? -fast : fast; }
#endif
  if (inv) { dd one; one.hi = 1.0; one.lo = 0.0; dd t; t.hi = ax; t.lo = 0.0; u = dd_div(one, t); }
  else { u.hi = ax; u.lo = 0.0; }
  const double kf = __builtin_rint(u.hi*64.0);
  const int k = (int)kf;
  const double c = kf*0.015625;
  dd r;
  if (k == 0) r = u;
  else {
    // u.hi - c is exact (u.hi / c in [1/2, 2]); 1 + u c in double-double
    dd num = two_sum(u.hi - c, u.lo);
    dd den = dd_add_d(dd_mul_d(u, c), 1.0);
    r = dd_div(num, den);
  }
  // atan r = r + r (-s/3 + s^2/5 - s^3/7 + tail),  s = r^2
  const dd s = dd_mul(r, r);
  const double sh = s.hi;
  const double tail = sh*sh*sh*sh*(1.0/9.0 + sh*(-1.0/11.0 + sh*(1.0/13.0 + sh*(-1.0/15.0 + sh*(1.0/17.0)))));
  dd c3; c3.hi = MCP_C3_HI; c3.lo = MCP_C3_LO;
  dd c5; c5.hi = MCP_C5_HI; c5.lo = MCP_C5_LO;
  dd c7; c7.hi = MCP_C7_HI; c7.lo = MCP_C7_LO;
  // Horner in double-double: h = s (-1/3 + s (1/5 - s/7))
  dd h = dd_add(c5, dd_neg(dd_mul(s, c7)));
  h = dd_add(dd_neg(c3), dd_mul(s, h));
  h = dd_mul(s, h);
  h = dd_add_d(h, tail);
  dd a = dd_add(r, dd_mul(r, h));
  dd tk; tk.hi = tab_hi[k]; tk.lo = tab_lo[k];
  a = dd_add(tk, a);
  if (inv) { dd p; p.hi = MCP_PIO2_HI; p.lo = MCP_PIO2_LO; a = dd_add(p, dd_neg(a)); }
  const double res = a.hi + a.lo;
  return x < 0 ? -res : res;
}

}  // namespace mcp_atan
