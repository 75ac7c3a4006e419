// Dependent-chain latencies of the fp64 operations the panel factorisation strings together (one wavefront, gfx950).
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/fp64_chain.hip -o build/fp64_chain ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
// the timer reads are tied to the chained value: the compiler may not move arithmetic across them
#define TIC(v) do { asm volatile("" : "+v"(v)); t0 = clock64(); asm volatile("" : "+v"(v)); } while (0)
#define TOC(v, i) do { asm volatile("" : "+v"(v)); t1 = clock64(); asm volatile("" : "+v"(v)); t[i] = t1 - t0; } while (0)
__device__ inline double rl(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out, long long* t, double seed) {
  __shared__ double lds[256];
  double x = seed + threadIdx.x*1e-3, y = 1.0000001;
  long long t0, t1;
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
  TOC(x, 0);
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = x*y;
  TOC(x, 1);
  x = fabs(x) + 2.0;
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
  TOC(x, 2);
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
  TOC(x, 3);
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = rl(x, 5) + 1.0;
  TOC(x, 4);
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) { lds[threadIdx.x] = x; x = lds[5] + 1.0; }
  TOC(x, 5);
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  TIC(a0);
#pragma unroll
  for (int i = 0; i < N/8; ++i) { a0 = __builtin_fma(a0, y, 1e-9); a1 = __builtin_fma(a1, y, 1e-9); a2 = __builtin_fma(a2, y, 1e-9); a3 = __builtin_fma(a3, y, 1e-9);
                                  a4 = __builtin_fma(a4, y, 1e-9); a5 = __builtin_fma(a5, y, 1e-9); a6 = __builtin_fma(a6, y, 1e-9); a7 = __builtin_fma(a7, y, 1e-9); }
  x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  TOC(x, 6);
  double p = fabs(x) + 3.0, b = p + 1.0, a = 0.5;
  TIC(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double tt = y0*(-p);
    const double e = __builtin_fma(tt, y0, 1.0);
    const double u = y0*e, qq = __builtin_fma(e, 0.375, 0.5);
    const double inv = __builtin_fma(u, qq, y0);
    const double l = a*inv;
    p = __builtin_fma(-l, l, b);
  }
  TOC(p, 7);
  TIC(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double r0 = __builtin_amdgcn_rcp(p);
    const double e = __builtin_fma(-p, r0, 1.0);
    const double r = __builtin_fma(r0, e, r0);
    const double l = a*r;
    p = __builtin_fma(-l, a, b);
  }
  TOC(p, 8);
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {x, x, x, x};
  double ax = x;
  TIC(ax);
  acc[0] = ax;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0);
  ax = acc[0] + acc[1] + acc[2] + acc[3];
  TOC(ax, 9);
  lds[threadIdx.x] = 0.0; __syncthreads();
  double di = (double)(threadIdx.x & 63);
  TIC(di);
#pragma unroll
  for (int i = 0; i < N; ++i) di = lds[(int)di] + (double)(threadIdx.x & 63);
  TOC(di, 10);
  // readlane feeding an fma as a scalar operand (the panel's fast path): x = fma(x, readlane(x, 7), 1e-9)
  TIC(x);
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, rl(y, 7), rl(x, 3)*1e-30);
  TOC(x, 11);
  asm volatile("" : "+v"(p)); long long w0 = wall_clock64(); t0 = clock64(); asm volatile("" : "+v"(p));
#pragma unroll
  for (int i = 0; i < 4*N; ++i) p = __builtin_fma(p, y, 1e-9);
  asm volatile("" : "+v"(p)); t1 = clock64(); long long w1 = wall_clock64(); t[12] = t1 - t0; t[13] = w1 - w0;
  out[threadIdx.x] = x + p + ax + di;
}
int main() {
  double* out; long long* t;
  (void)hipMalloc(&out, 64*8); (void)hipMalloc(&t, 16*8);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.25);
  (void)hipDeviceSynchronize();
  long long h[16]; (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[] = {"dependent v_fma_f64", "dependent v_mul_f64", "v_rsq_f64 + v_add_f64", "v_rcp_f64 + v_add_f64", "readlane pair + add", "LDS write + broadcast read + add",
                      "independent fma (8 chains) per fma", "Cholesky pivot chain (rsq + 3rd order)", "L D L^T pivot chain (rcp + Newton)", "mfma_f64_16x16x4 dependent", "ds_read_b64 dependent + add",
                      "fma with two readlane pairs"};
  for (int i = 0; i < 12; ++i) printf("%-42s %7.1f cycles per step\n", nm[i], (double)h[i]/(i == 9 ? 64 : N));
  printf("clock64 ticks per 10 ns wall tick: %.2f  (%lld / %lld)\n", (double)h[12]/(double)h[13], h[12], h[13]);
  return 0;
}
