// C++ caller of the drop-in boundary: a 3-keyframe, single-camera bundle through mcptam_hip::ChainBundle.
// Built with plain g++ against include/ and libmcptam_hip.so; `--link-only` just proves that every symbol resolves
// (CPU container); without arguments it solves on the GPU and checks that the noise-free problem converges.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "mcptam_hip/ChainBundle.hpp"
#include "mcp_img.h"

// TaylorCamera state for w(rho) = 250 - 1.2e-3 rho^2: the inverse polynomial rho(theta) is fitted here the way
// TaylorCamera::FindInvPolyUsingRoots does (root of w(rho) - rho tan(theta), least squares on the centred angle)
static mcp_camera make_camera() {
  mcp_camera c; std::memset(&c, 0, sizeof c);
  const double a0 = 250, a2 = -1.2e-3;
  c.params[0] = a0; c.params[1] = a2; c.params[4] = 320; c.params[5] = 240; c.params[6] = 1;
  c.image_size[0] = 640; c.image_size[1] = 480; c.affine[0] = 1; c.affine[3] = 1; c.center[0] = 320; c.center[1] = 240;
  c.max_rho = 400; c.min_theta = std::atan((a0 + a2*400*400)/400);
  std::vector<double> th, rh;
  for (double t = c.min_theta; t < M_PI/2 - 1e-3; t += 0.01) {
    double lo = 0, hi = 400;                       // w(rho) - rho tan(t) is decreasing in rho
    for (int it = 0; it < 80; ++it) { const double m = 0.5*(lo + hi); if (a0 + a2*m*m - m*std::tan(t) > 0) lo = m; else hi = m; }
    th.push_back(t); rh.push_back(0.5*(lo + hi));
  }
  double mean = 0; for (double t : th) mean += t; mean /= th.size();
  double var = 0; for (double t : th) var += (t - mean)*(t - mean); const double sd = std::sqrt(var/th.size());
  const int D = 8;                                  // degree 7
  double A[D][D + 1]; std::memset(A, 0, sizeof A);
  for (size_t i = 0; i < th.size(); ++i) {
    double pw[2*D]; pw[0] = 1; const double s = (th[i] - mean)/sd;
    for (int k = 1; k < 2*D; ++k) pw[k] = pw[k - 1]*s;
    for (int r = 0; r < D; ++r) { for (int q = 0; q < D; ++q) A[r][q] += pw[r + q]; A[r][D] += pw[r]*rh[i]; }
  }
  for (int k = 0; k < D; ++k) {                    // Gauss-Jordan with partial pivoting
    int piv = k; for (int r = k + 1; r < D; ++r) if (std::fabs(A[r][k]) > std::fabs(A[piv][k])) piv = r;
    for (int q = 0; q <= D; ++q) std::swap(A[k][q], A[piv][q]);
    for (int r = 0; r < D; ++r) if (r != k) { const double f = A[r][k]/A[k][k]; for (int q = k; q <= D; ++q) A[r][q] -= f*A[k][q]; }
  }
  c.theta_mean = mean; c.theta_std = sd; c.n_inv = D;
  for (int k = 0; k < D; ++k) c.inv_coeffs[k] = A[k][D]/A[k][k];
  return c;
}
static void project(const mcp_camera& c, const double x[3], double uv[2]) {
  const double n = std::sqrt(x[0]*x[0] + x[1]*x[1]), th = std::atan(x[2]/n), sv = (th - c.theta_mean)/c.theta_std;
  double rho = 0; for (int k = c.n_inv - 1; k > 0; --k) { rho += c.inv_coeffs[k]; rho *= sv; } rho += c.inv_coeffs[0];
  uv[0] = rho*x[0]/n + c.center[0]; uv[1] = rho*x[1]/n + c.center[1];
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "--link-only") == 0) {
    std::printf("linked: %d gfx950 device(s) visible (%s)\n", mcp_device_count(), mcp_last_error());
    void* img_syms[] = { (void*)mcp_kf_create, (void*)mcp_kf_make_lite, (void*)mcp_track_search, (void*)mcp_track_pose_update, (void*)mcp_minipatch_find };
    return img_syms[0] ? 0 : 1;
  }
  using mcptam_hip::ChainBundle;
  std::vector<mcp_camera> cams(1, make_camera());
  ChainBundle bundle(cams, true, true, false);
  const double I[9] = {1,0,0, 0,1,0, 0,0,1};
  double t[3][3] = {{0,0,0}, {-0.3,0,0}, {-0.6,0.02,0}};
  int kf[3];
  for (int k = 0; k < 3; ++k) {
    double tt[3] = { t[k][0] + (k == 2 ? 0.01 : 0.0), t[k][1], t[k][2] };      // third pose starts 1 cm off
    kf[k] = bundle.AddPose(I, tt, k < 2);
  }
  std::mt19937 rng(7); std::uniform_real_distribution<double> U(-1.5, 1.5), Z(3.0, 8.0);
  int npt = 0;
  for (int i = 0; i < 200; ++i) {
    const double X[3] = { U(rng), U(rng), Z(rng) };                             // world == frame of kf[0]
    const double Xs[3] = { X[0]*1.03, X[1]*1.03, X[2]*1.03 };                  // depth perturbed by 3 %
    const int pid = bundle.AddPoint(Xs, std::vector<int>(1, kf[0]), false);
    for (int k = 0; k < 3; ++k) {
      const double xc[3] = { X[0] + t[k][0], X[1] + t[k][1], X[2] + t[k][2] };
      double uv[2]; project(cams[0], xc, uv);
      bundle.AddMeas(std::vector<int>(1, kf[k]), pid, uv, 1.0, 0);
    }
    ++npt;
  }
  bool abort_flag = false;
  const int n = bundle.Compute(&abort_flag, 40);
  double R[9], tt[3]; bundle.GetPose(kf[2], R, tt);
  const double err = std::fabs(tt[0] - t[2][0]) + std::fabs(tt[1] - t[2][1]) + std::fabs(tt[2] - t[2][2]);
  std::printf("iterations %d converged %d mean chi2 %.3g pose error %.3g outliers %zu\n", n, (int)bundle.Converged(),
              bundle.GetMeanChiSquared(), err, bundle.GetOutlierMeasurements().size());
  return (n > 0 && bundle.Converged() && err < 1e-7) ? 0 : 2;
}
