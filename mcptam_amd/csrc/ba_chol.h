// ba_chol.h -- dense fp64 Cholesky of the reduced pose system on gfx950.
//
// Replaces g2o::LinearSolverCholmod (src/ChainBundle.cc:1156) for the (6P x 6P) system that
// remains after the points are eliminated.  S is row-major n x n with leading dimension n;
// only the lower triangle is read and written.  Blocked right-looking factorisation with
// 32-wide panels; a failed pivot (matrix not positive definite == CHOLMOD failure, which
// g2o turns into a rejected LM trial) raises *fail.
#pragma once
#include <hip/hip_runtime.h>

namespace mcp {

constexpr int CH_NB = 32;

// factor the diagonal block at k0 (nbe x nbe, nbe <= 32) with one wavefront
__global__ void __launch_bounds__(64)
k_potrf_diag(double* __restrict__ S, int n, int k0, int nbe, int* __restrict__ fail) {
  __shared__ double T[CH_NB][CH_NB + 1];
  const int t = threadIdx.x;
  for (int i = t; i < nbe*nbe; i += 64) { const int r = i / nbe, c = i % nbe; T[r][c] = (c <= r) ? S[(size_t)(k0 + r)*n + k0 + c] : 0.0; }
  __syncthreads();
  for (int j = 0; j < nbe; ++j) {
    if (t == j) {
      double d = T[j][j];
      if (!(d > 0.0)) { atomicOr(fail, 2); d = 1.0; }
      T[j][j] = sqrt(d);
    }
    __syncthreads();
    if (t > j && t < nbe) T[t][j] /= T[j][j];
    __syncthreads();
    if (t > j && t < nbe) { const double l = T[t][j]; for (int c = j + 1; c <= t; ++c) T[t][c] -= l*T[c][j]; }
    __syncthreads();
  }
  for (int i = t; i < nbe*nbe; i += 64) { const int r = i / nbe, c = i % nbe; if (c <= r) S[(size_t)(k0 + r)*n + k0 + c] = T[r][c]; }
}

// panel: rows below the diagonal block, X L_kk^T = A  (one thread per row)
__global__ void __launch_bounds__(64)
k_trsm_panel(double* __restrict__ S, int n, int k0) {
  __shared__ double L[CH_NB][CH_NB + 1];
  const int t = threadIdx.x;
  for (int i = t; i < CH_NB*CH_NB; i += 64) { const int r = i / CH_NB, c = i % CH_NB; L[r][c] = S[(size_t)(k0 + r)*n + k0 + c]; }
  __syncthreads();
  const int row = k0 + CH_NB + blockIdx.x*64 + t;
  if (row >= n) return;
  double a[CH_NB];
  double* p = S + (size_t)row*n + k0;
#pragma unroll
  for (int c = 0; c < CH_NB; ++c) a[c] = p[c];
#pragma unroll
  for (int c = 0; c < CH_NB; ++c) {
    double s = a[c];
#pragma unroll
    for (int j = 0; j < c; ++j) s -= a[j]*L[c][j];
    a[c] = s / L[c][c];
  }
#pragma unroll
  for (int c = 0; c < CH_NB; ++c) p[c] = a[c];
}

// trailing update: C(ti,tj) -= P_ti P_tj^T for 32x32 tiles ti >= tj beyond the panel
__global__ void __launch_bounds__(256)
k_syrk_tile(double* __restrict__ S, int n, int k0) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  __shared__ double Pi[CH_NB][CH_NB + 1];
  __shared__ double Pj[CH_NB][CH_NB + 1];
  const int base = k0 + CH_NB;
  const int r0 = base + ti*CH_NB, c0 = base + tj*CH_NB;
  for (int i = threadIdx.x; i < CH_NB*CH_NB; i += 256) {
    const int r = i / CH_NB, c = i % CH_NB;
    Pi[r][c] = (r0 + r < n) ? S[(size_t)(r0 + r)*n + k0 + c] : 0.0;
    Pj[r][c] = (c0 + r < n) ? S[(size_t)(c0 + r)*n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
  double acc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll 8
  for (int k = 0; k < CH_NB; ++k) {
    const double a0 = Pi[2*ty][k], a1 = Pi[2*ty+1][k], b0 = Pj[2*tx][k], b1 = Pj[2*tx+1][k];
    acc[0][0] += a0*b0; acc[0][1] += a0*b1; acc[1][0] += a1*b0; acc[1][1] += a1*b1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + 2*ty + i, c = c0 + 2*tx + j;
      if (r < n && c < n && c <= r) S[(size_t)r*n + c] -= acc[i][j];
    }
}

// solve L L^T x = b in place (b -> x); single workgroup, x staged in LDS.  n <= CH_TRSV_MAX.
constexpr int CH_TRSV_MAX = 6144;
__global__ void __launch_bounds__(1024)
k_chol_solve(const double* __restrict__ S, int n, double* __restrict__ b) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 1024) xs[i] = b[i];
  __syncthreads();
  // forward: L y = b
  for (int k0 = 0; k0 < n; k0 += CH_NB) {
    const int nbe = min(CH_NB, n - k0);
    if (t < 64) {        // wave 0 solves the diagonal block
      double y = (t < nbe) ? xs[k0 + t] : 0.0;
      for (int c = 0; c < nbe; ++c) {
        const double lcc = S[(size_t)(k0 + c)*n + k0 + c];
        const double yc = __shfl(y, c, 64) / lcc;
        if (t == c) y = yc;
        else if (t > c && t < nbe) y -= S[(size_t)(k0 + t)*n + k0 + c]*yc;
      }
      if (t < nbe) xs[k0 + t] = y;
    }
    __syncthreads();
    for (int r = k0 + nbe + t; r < n; r += 1024) {
      const double* Lr = S + (size_t)r*n + k0;
      double s = 0.0;
      for (int c = 0; c < nbe; ++c) s += Lr[c]*xs[k0 + c];
      xs[r] -= s;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  const int nblk = (n + CH_NB - 1)/CH_NB;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb*CH_NB, nbe = min(CH_NB, n - k0);
    if (t < 64) {
      double y = (t < nbe) ? xs[k0 + t] : 0.0;
      for (int c = nbe - 1; c >= 0; --c) {
        const double lcc = S[(size_t)(k0 + c)*n + k0 + c];
        const double xc = __shfl(y, c, 64) / lcc;
        if (t == c) y = xc;
        else if (t < c) y -= S[(size_t)(k0 + c)*n + k0 + t]*xc;
      }
      if (t < nbe) xs[k0 + t] = y;
    }
    __syncthreads();
    for (int c = t; c < k0; c += 1024) {
      double s = 0.0;
      for (int r = 0; r < nbe; ++r) s += S[(size_t)(k0 + r)*n + c]*xs[k0 + r];
      xs[c] -= s;
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 1024) b[i] = xs[i];
}

inline void chol_factor(hipStream_t st, double* S, int n, int* fail) {
  for (int k0 = 0; k0 < n; k0 += CH_NB) {
    const int nbe = (n - k0 < CH_NB) ? n - k0 : CH_NB;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(64), 0, st, S, n, k0, nbe, fail);
    const int rem = n - k0 - CH_NB;
    if (rem > 0) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((rem + 63)/64), dim3(64), 0, st, S, n, k0);
      const int T = (rem + CH_NB - 1)/CH_NB;
      hipLaunchKernelGGL(k_syrk_tile, dim3(T, T), dim3(256), 0, st, S, n, k0);
    }
  }
}
inline void chol_solve(hipStream_t st, const double* S, int n, double* b) {
  hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), (size_t)n*sizeof(double), st, S, n, b);
}

}  // namespace mcp
