// ChainBundle.hpp -- C++ host-side mirror of MCPTAM's ChainBundle over the C ABI of mcp_ba.h.
//
// Same member names, argument meaning and return codes as the reference class
// (/root/reference/include/mcptam/ChainBundle.h:106-186); TooN types are replaced by plain arrays
// (SE3 = row-major R[9] + t[3]) so that the header has no dependency beyond the C ABI.  A MCPTAM tree
// uses the TooN-typed shim of INTEGRATION.md instead; this class is what a stand-alone C++ caller (and the
// C++ smoke test under tests/cpp) links against.
#pragma once
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../mcp_ba.h"

namespace mcptam_hip {

class ChainBundle {
 public:
  // statics of the reference (src/ChainBundle.cc:1132-1136)
  static inline int snMaxIterations = 100;
  static inline int snMaxTrialsAfterFailure = 100;
  static inline double sdUpdatePercentConvergenceLimit = 1e-10;
  static inline double sdUpdateRMSConvergenceLimit = 1e-10;
  static inline double sdMinMEstimatorSigma = 0.5;

  /// cameraModels: the fitted cameras, index = the camera "name" used by AddMeas
  ChainBundle(const std::vector<mcp_camera>& cameraModels, bool bUseRobust, bool bUseTukey, bool bVerbose) {
    mcp_ba_params p;
    p.max_iterations = snMaxIterations; p.max_trials_after_failure = snMaxTrialsAfterFailure;
    p.update_percent_limit = sdUpdatePercentConvergenceLimit; p.update_rms_limit = sdUpdateRMSConvergenceLimit;
    p.min_mestimator_sigma = sdMinMEstimatorSigma; p.disable_convergence = 0; p.device = -1; p.profile = 0;
    mpHandle = mcp_ba_create(cameraModels.data(), (int)cameraModels.size(), bUseRobust, bUseTukey, bVerbose, &p);
    if (!mpHandle) throw std::runtime_error(std::string("ChainBundle: ") + mcp_last_error());   // no CPU fallback
  }
  ~ChainBundle() { mcp_ba_destroy(mpHandle); }
  ChainBundle(const ChainBundle&) = delete;
  ChainBundle& operator=(const ChainBundle&) = delete;

  int AddPose(const double R[9], const double t[3], bool bFixed) { return mcp_ba_add_pose(mpHandle, R, t, bFixed); }
  int AddPoint(const double v3PointInCam[3], const std::vector<int>& vCams, bool bFixed) {
    const int id = mcp_ba_add_point(mpHandle, v3PointInCam, vCams.data(), (int)vCams.size(), bFixed);
    if (id < 0) throw std::invalid_argument(mcp_last_error());
    return id;
  }
  void AddMeas(const std::vector<int>& vCams, int nPointIdx, const double v2Pos[2], double dNoiseSigmaSquared, int nCameraIndex) {
    if (mcp_ba_add_meas(mpHandle, vCams.data(), (int)vCams.size(), nPointIdx, v2Pos, dNoiseSigmaSquared, nCameraIndex) != 0)
      throw std::invalid_argument(mcp_last_error());
  }
  /// returns the number of outer iterations run (>0), 0 = aborted before any step, -1 = failure
  int Compute(bool* pAbortSignal, int nNumIter = snMaxIterations, double dUserLambda = -1) {
    static_assert(sizeof(bool) == 1, "the abort flag is shared as one byte");
    const int rc = mcp_ba_compute(mpHandle, reinterpret_cast<volatile unsigned char*>(pAbortSignal), nNumIter, dUserLambda);
    if (rc == MCP_BA_ERR_RUNTIME) throw std::runtime_error(mcp_last_error());      // device failure, not a BA outcome
    return rc;
  }
  bool Converged() { return mcp_ba_converged(mpHandle) != 0; }
  int TotalIterations() { return mcp_ba_total_iterations(mpHandle); }
  void GetPoint(int n, double x[3]) { if (mcp_ba_get_point(mpHandle, n, x)) throw std::out_of_range(mcp_last_error()); }
  void GetPose(int n, double R[9], double t[3]) { if (mcp_ba_get_pose(mpHandle, n, R, t)) throw std::out_of_range(mcp_last_error()); }
  /// (point id, id of the first pose of the observer chain, camera index)
  std::vector<std::tuple<int, int, int> > GetOutlierMeasurements() {
    std::vector<int> raw(3*(size_t)mcp_ba_num_outliers(mpHandle) + 3);
    const int n = mcp_ba_get_outliers(mpHandle, raw.data(), (int)raw.size()/3);
    std::vector<std::tuple<int, int, int> > out;
    for (int i = 0; i < n; ++i) out.emplace_back(raw[3*i], raw[3*i + 1], raw[3*i + 2]);
    return out;
  }
  double GetSigmaSquared() { return mcp_ba_sigma_squared(mpHandle); }
  double GetMeanChiSquared() { return mcp_ba_mean_chi_squared(mpHandle); }
  double GetMaxCov() { return mcp_ba_max_cov(mpHandle); }
  double GetLambda() { return mcp_ba_lambda(mpHandle); }
  mcp_ba* handle() { return mpHandle; }

 private:
  mcp_ba* mpHandle = nullptr;
};

}  // namespace mcptam_hip
