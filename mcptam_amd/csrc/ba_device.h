// ba_device.h -- fp64 device math for the ChainBundle kernels (gfx950).
//
// Device-side equivalents of the pieces of the reference that run per measurement:
//   TooN SE3/SO3 exp + generator fields (used at src/ChainBundle.cc:84-85,265,512,564,606-617)
//   TaylorCamera::Project / GetProjectionDerivs / GetCamSphereDeriv
//     (src/TaylorCamera.cc:202-287, 353-383, 617-669)
//   the bearing + inverse-depth point frame of VertexRelPoint (src/ChainBundle.cc:253-272).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mcp_ba.h"
#include "atan_cr.h"

namespace mcp {

struct Se3 { double R[9]; double t[3]; };

__host__ __device__ inline void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3*i+j] = A[3*i]*B[j] + A[3*i+1]*B[3+j] + A[3*i+2]*B[6+j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__host__ __device__ inline void mat3t_mul(const double* A, const double* B, double* C) {   // A^T B
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3*i+j] = A[i]*B[j] + A[3+i]*B[3+j] + A[6+i]*B[6+j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__host__ __device__ inline void mat3_mul_t(const double* A, const double* B, double* C) {   // A B^T
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3*i+j] = A[3*i]*B[3*j] + A[3*i+1]*B[3*j+1] + A[3*i+2]*B[3*j+2];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__host__ __device__ inline void mat3_vec(const double* A, const double* v, double* o) {
  const double a = A[0]*v[0] + A[1]*v[1] + A[2]*v[2];
  const double b = A[3]*v[0] + A[4]*v[1] + A[5]*v[2];
  const double c = A[6]*v[0] + A[7]*v[1] + A[8]*v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
__host__ __device__ inline void mat3t_vec(const double* A, const double* v, double* o) {
  const double a = A[0]*v[0] + A[3]*v[1] + A[6]*v[2];
  const double b = A[1]*v[0] + A[4]*v[1] + A[7]*v[2];
  const double c = A[2]*v[0] + A[5]*v[1] + A[8]*v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
__host__ __device__ inline void se3_identity(Se3& T) {
#pragma unroll
  for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  T.t[0] = T.t[1] = T.t[2] = 0.0;
}
// (R1 R2, R1 t2 + t1)
__host__ __device__ inline void se3_compose(const Se3& A, const Se3& B, Se3& C) {
  Se3 T;
  mat3_mul(A.R, B.R, T.R);
  mat3_vec(A.R, B.t, T.t);
  T.t[0] += A.t[0]; T.t[1] += A.t[1]; T.t[2] += A.t[2];
  C = T;
}
__host__ __device__ inline void se3_apply(const Se3& A, const double* v, double* o) {
  double r[3];
  mat3_vec(A.R, v, r);
  o[0] = r[0] + A.t[0]; o[1] = r[1] + A.t[1]; o[2] = r[2] + A.t[2];
}
// x = A^-1 v = R^T (v - t)
__host__ __device__ inline void se3_apply_inv(const Se3& A, const double* v, double* o) {
  const double d[3] = { v[0] - A.t[0], v[1] - A.t[1], v[2] - A.t[2] };
  mat3t_vec(A.R, d, o);
}

// Rodrigues with TooN's coefficient convention: R = I + A [w]x + B [w]x^2, written out.
__host__ __device__ inline void rodrigues(const double* w, double A, double B, double* R) {
  const double wx2 = w[0]*w[0], wy2 = w[1]*w[1], wz2 = w[2]*w[2];
  R[0] = 1.0 - B*(wy2 + wz2);
  R[4] = 1.0 - B*(wx2 + wz2);
  R[8] = 1.0 - B*(wx2 + wy2);
  double a = A*w[2], b = B*(w[0]*w[1]);
  R[1] = b - a; R[3] = b + a;
  a = A*w[1]; b = B*(w[0]*w[2]);
  R[2] = b + a; R[6] = b - a;
  a = A*w[0]; b = B*(w[1]*w[2]);
  R[5] = b - a; R[7] = b + a;
}
// SO3 exponential, TooN thresholds (theta^2 < 1e-8, < 1e-6)
__host__ __device__ inline void so3_exp(const double* w, double* R) {
  const double th2 = w[0]*w[0] + w[1]*w[1] + w[2]*w[2];
  double A, B;
  if (th2 < 1e-8) { A = 1.0 - th2*(1.0/6.0); B = 0.5; }
  else if (th2 < 1e-6) { B = 0.5 - 0.25*(1.0/6.0)*th2; A = 1.0 - th2*(1.0/6.0)*(1.0 - (1.0/20.0)*th2); }
  else { const double th = sqrt(th2), inv = 1.0/th; A = sin(th)*inv; B = (1.0 - cos(th))*(inv*inv); }
  rodrigues(w, A, B, R);
}
// SE3 exponential, mu = (t, w)
__host__ __device__ inline void se3_exp(const double* mu, Se3& T) {
  const double* w = mu + 3;
  const double th2 = w[0]*w[0] + w[1]*w[1] + w[2]*w[2];
  const double cx = w[1]*mu[2] - w[2]*mu[1], cy = w[2]*mu[0] - w[0]*mu[2], cz = w[0]*mu[1] - w[1]*mu[0];
  double A, B;
  if (th2 < 1e-8) {
    A = 1.0 - th2*(1.0/6.0); B = 0.5;
    T.t[0] = mu[0] + 0.5*cx; T.t[1] = mu[1] + 0.5*cy; T.t[2] = mu[2] + 0.5*cz;
  } else {
    double C;
    if (th2 < 1e-6) { C = (1.0/6.0)*(1.0 - (1.0/20.0)*th2); A = 1.0 - th2*C; B = 0.5 - 0.25*(1.0/6.0)*th2; }
    else { const double th = sqrt(th2), inv = 1.0/th; A = sin(th)*inv; B = (1.0 - cos(th))*(inv*inv); C = (1.0 - A)*(inv*inv); }
    const double dx = w[1]*cz - w[2]*cy, dy = w[2]*cx - w[0]*cz, dz = w[0]*cy - w[1]*cx;
    T.t[0] = mu[0] + B*cx + C*dx; T.t[1] = mu[1] + B*cy + C*dy; T.t[2] = mu[2] + B*cz + C*dz;
  }
  rodrigues(w, A, B, T.R);
}

// Horner evaluation, x^0 coefficient first (TaylorCamera::PolyVal order of operations)
__host__ __device__ inline double poly_low_first(const double* c, int n, double x) {
  double v = 0.0;
  for (int i = n - 1; i > 0; --i) { v += c[i]; v *= x; }
  return v + c[0];
}

struct Projection { double u, v; double D[4]; int invalid; };

// Project + GetProjectionDerivs fused (the reference calls them back to back, ChainBundle.cc:390-392)
template <bool WITH_DERIVS>
__host__ __device__ inline void cam_project(const mcp_camera& cam, const double* xc, Projection& P,
                                            const double* atan_hi = mcp_atan::kAtanHi, const double* atan_lo = mcp_atan::kAtanLo /* the arctangent's table, or a copy of it in LDS */) {
  const double n = sqrt(xc[0]*xc[0] + xc[1]*xc[1]);
  double theta, rho, cphi, sphi;
  if (n == 0.0) { theta = 1.57079632679489661923; rho = 0.0; cphi = 0.0; sphi = 0.0; }
  else {
    theta = mcp_atan::atan_cr(xc[2]/n, atan_hi, atan_lo);      // correctly rounded: the one platform-independent value (atan_cr.h)
    const double ts = (theta - cam.theta_mean)/cam.theta_std;
    if (cam.n_inv > 0) rho = poly_low_first(cam.inv_coeffs, cam.n_inv, ts);
    else {
      // no usable inverse polynomial: linear inverse model, then FindRootWithNewton on a0 + (a1 - tan theta) rho + a2 rho^2 +
      // a3 rho^3 + a4 rho^4 (a1 = 0) until the step is below 0.01 (src/TaylorCamera.cc:258-270, 293-315; at most 50 steps,
      // where the reference asserts)
      const double a0 = cam.params[0], a2 = cam.params[1], a3 = cam.params[2], a4 = cam.params[3];
      const double tt = xc[2]/n;
      double prev = poly_low_first(cam.inv_coeffs, 2, ts);
      rho = prev;
      for (int it = 0; it < 50; ++it) {
        // PolyVal's order of operations (Horner, x^0 last)
        const double f = (((a4*prev + a3)*prev + a2)*prev + (0.0 - tt))*prev + a0;
        const double fp = ((4.0*a4*prev + 3.0*a3)*prev + 2.0*a2)*prev + (0.0 - tt);
        rho = prev - f/fp;
        const double err = fabs(rho - prev);
        prev = rho;
        if (!(err > 0.01)) break;
      }
    }
    cphi = xc[0]/n; sphi = xc[1]/n;
  }
  const double d0 = cphi*rho, d1 = sphi*rho;
  P.u = cam.affine[0]*d0 + cam.affine[1]*d1 + cam.center[0];
  P.v = cam.affine[2]*d0 + cam.affine[3]*d1 + cam.center[1];
  P.invalid = (theta < cam.min_theta) ||
              !(P.u >= 0 && P.u < cam.image_size[0] && P.v >= 0 && P.v < cam.image_size[1]);
  if (WITH_DERIVS) {
    const double a0 = cam.params[0], a2 = cam.params[1], a3 = cam.params[2], a4 = cam.params[3];
    // w = a0 + a2 r^2 + a3 r^3 + a4 r^4 ; denominator = -a0 + a2 r^2 + 2 a3 r^3 + 3 a4 r^4  (Horner, a1 = 0)
    const double w = (((a4*rho + a3)*rho + a2)*rho + 0.0)*rho + a0;
    const double den = (((3.0*a4*rho + 2.0*a3)*rho + a2)*rho + 0.0)*rho - a0;
    const double drdt = (rho*rho + w*w)/den;
    const double t0 = cphi*drdt, t1 = sphi*drdt, p0 = -sphi*rho, p1 = cphi*rho;
    P.D[0] = cam.affine[0]*t0 + cam.affine[1]*t1;
    P.D[2] = cam.affine[2]*t0 + cam.affine[3]*t1;
    P.D[1] = cam.affine[0]*p0 + cam.affine[1]*p1;
    P.D[3] = cam.affine[2]*p0 + cam.affine[3]*p1;
  }
}

// d(theta)/dx and d(phi)/dx of the camera-frame point (GetCamSphereDeriv incl. its zero guards)
__host__ __device__ inline void cam_sphere_deriv(const double* p, double* dT, double* dP) {
  const double x = p[0], y = p[1], z = p[2];
  const double n2 = x*x + y*y, n = sqrt(n2), z2 = z*z;
  if (n == 0.0) { dT[0] = dT[1] = dT[2] = 0.0; }
  else {
    const double nn = n*n, den = nn*n + n*z2;
    dT[0] = -z*x/den; dT[1] = -z*y/den; dT[2] = n/(nn + z2);
  }
  if (x == 0.0 && y == 0.0) { dP[0] = dP[1] = dP[2] = 0.0; }
  else { dP[0] = -y/n2; dP[1] = x/n2; dP[2] = 0.0; }
}

// generator field of SE3 at point p (homogeneous weight 1): k<3 -> e_k, k>=3 -> e_{k-3} x p
__host__ __device__ inline void generator(int k, const double* p, double* o) {
  switch (k) {
    case 0: o[0] = 1; o[1] = 0; o[2] = 0; break;
    case 1: o[0] = 0; o[1] = 1; o[2] = 0; break;
    case 2: o[0] = 0; o[1] = 0; o[2] = 1; break;
    case 3: o[0] = 0; o[1] = -p[2]; o[2] = p[1]; break;
    case 4: o[0] = p[2]; o[1] = 0; o[2] = -p[0]; break;
    default: o[0] = -p[1]; o[1] = p[0]; o[2] = 0; break;
  }
}

// rotation taking the point direction onto +z, and 1/|x| (no guard for x || z: the reference's
// live code has none either, ChainBundle.cc:258-262 vs the commented block :286-293)
__host__ __device__ inline void point_frame(const double* x, double* Rp, double* dir, double& rho) {
  const double len = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
  rho = 1.0/len;
  dir[0] = x[0]*rho; dir[1] = x[1]*rho; dir[2] = x[2]*rho;
  double ax[3] = { dir[1], -dir[0], 0.0 };
  const double nrm = sqrt(ax[0]*ax[0] + ax[1]*ax[1]);
  const double ang = asin(nrm);
  ax[0] = ax[0]/nrm*ang; ax[1] = ax[1]/nrm*ang;
  so3_exp(ax, Rp);
}
// VertexRelPoint::oplusImpl
__host__ __device__ inline void point_oplus(const double* x, const double* u, double* out) {
  double Rp[9], dir[3], rho;
  point_frame(x, Rp, dir, rho);
  const double w[3] = { u[0], u[1], 0.0 };
  double E[9], M[9], v[3];
  so3_exp(w, E);
  mat3t_mul(Rp, E, M);
  mat3_mul(M, Rp, M);
  mat3_vec(M, dir, v);
  const double s = 1.0/(rho + u[2]);
  double o0 = s*v[0], o1 = s*v[1], o2 = s*v[2];
  const double d = sqrt(o0*o0 + o1*o1 + o2*o2);
  if (d > 1e5) { const double f = 1e5/d; o0 *= f; o1 *= f; o2 *= f; }
  if (d < 1e-5) { const double f = 1e-5/d; o0 *= f; o1 *= f; o2 *= f; }
  out[0] = o0; out[1] = o1; out[2] = o2;
}

// Sum of v[0..31] over the 64 lanes of a wavefront, entry by entry, as a reduce-scatter butterfly: on return lanes 2i and 2i + 1 hold
// the total of entry idx = i (= lane >> 1).  At offset 32, 16, 8, 4, 2 a lane keeps the half of its entries its lane bit selects and
// receives the partner's contribution to them (16 + 8 + 4 + 2 + 1 exchanges), the last exchange (offset 1) completes the sum.
// The exchanges avoid the LDS crossbar (ds_bpermute, ~100 cycles each in a dependent chain) where the lane distance allows it:
// distance 32 and 16 -- 24 of the 32 exchanges -- are one v_permlane32_swap / v_permlane16_swap per register half, which hands BOTH
// partners what they are owed at once (the kept half of the lower lanes / even rows and the sent half of the upper lanes / odd rows
// travel in one register, the other two in the other: after the swap every lane adds its two registers); distance 8, 2 and 1 are DPP
// row rotations / quad permutations.  Same additions as with __shfl_xor (a + b vs b + a), so the same bits.
__device__ inline double mk_f64(unsigned int lo, unsigned int hi) { return __hiloint2double((int)hi, (int)lo); }
template <int OFF>
__device__ inline double wave_swap_add(double a /* entries j */, double b /* entries j + N/2 */) {
  static_assert(OFF == 32 || OFF == 16, "permlane swaps");
  const unsigned int alo = (unsigned int)__double2loint(a), ahi = (unsigned int)__double2hiint(a);
  const unsigned int blo = (unsigned int)__double2loint(b), bhi = (unsigned int)__double2hiint(b);
  if constexpr (OFF == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(alo, blo, false, false), q = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return mk_f64(r[0], q[0]) + mk_f64(r[1], q[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(alo, blo, false, false), q = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return mk_f64(r[0], q[0]) + mk_f64(r[1], q[1]);
  }
}
template <int CTRL>
__device__ inline double wave_dpp(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false));
}
template <int OFF>
__device__ inline double wave_xor(double v) {
  if constexpr (OFF == 8) return wave_dpp<0x128>(v);            // row_ror:8
  else if constexpr (OFF == 2) return wave_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
  else if constexpr (OFF == 1) return wave_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  else return __shfl_xor(v, OFF, 64);
}
template <int N, int OFF>
__device__ inline void wave_rs_step(const double* in, double* out, int lane) {
  if constexpr (OFF == 32 || OFF == 16) {
#pragma unroll
    for (int j = 0; j < N/2; ++j) out[j] = wave_swap_add<OFF>(in[j], in[j + N/2]);
  } else {
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int j = 0; j < N/2; ++j) {
      const double keep = up ? in[j + N/2] : in[j], send = up ? in[j] : in[j + N/2];
      out[j] = keep + wave_xor<OFF>(send);
    }
  }
}
__device__ inline double wave_reduce_scatter32(const double (&v)[32], int lane, int& idx) {
  double a16[16], a8[8], a4[4], a2[2], a1[1];
  wave_rs_step<32, 32>(v, a16, lane);
  wave_rs_step<16, 16>(a16, a8, lane);
  wave_rs_step<8, 8>(a8, a4, lane);
  wave_rs_step<4, 4>(a4, a2, lane);
  wave_rs_step<2, 2>(a2, a1, lane);
  idx = (lane >> 1) & 31;
  return a1[0] + wave_xor<1>(a1[0]);
}

}  // namespace mcp
