// ChainBundle.cc -- MI355X back end of class ChainBundle: replaces /root/reference/src/ChainBundle.cc (all of it: the g2o vertex /
// edge / kernel / action classes :62-1131 and the ChainBundle members :1132-1492) by forwarding to libmcptam_hip.so through the C ABI
// of include/mcp_ba.h.  Compile with shim/ChainBundle.h in place of include/mcptam/ChainBundle.h; needs the one-line friend
// declaration of shim/README.md in TaylorCamera.h.  No g2o, CHOLMOD or Eigen is referenced.
#include <mcptam/ChainBundle.h>
#include <mcptam/TaylorCamera.h>
#include <mcp_ba.h>
#include "CameraExport.h"
#include <ros/ros.h>
#include <limits>
#include <stdexcept>

using namespace TooN;

// Static members (values of the reference, src/ChainBundle.cc:1132-1136; overridden from ROS parameters by LoadStaticParamsServer.h:63-66)
int ChainBundle::snMaxIterations = 100;
int ChainBundle::snMaxTrialsAfterFailure = 100;
double ChainBundle::sdUpdatePercentConvergenceLimit = 1e-10;
double ChainBundle::sdUpdateRMSConvergenceLimit = 1e-10;
double ChainBundle::sdMinMEstimatorSigma = 0.5;

static void ToArrays(const SE3<>& se3, double R[9], double t[3])
{
  const Matrix<3>& m3 = se3.get_rotation().get_matrix();
  for(int i = 0; i < 3; ++i)
  {
    for(int j = 0; j < 3; ++j)
      R[3*i + j] = m3(i, j);
    t[i] = se3.get_translation()[i];
  }
}

ChainBundle::ChainBundle(TaylorCameraMap& cameraModels, bool bUseRobust, bool bUseTukey, bool bVerbose)
: mpHandle(NULL)
, mmCameraModels(cameraModels)
, mbConverged(false)
, mbHitMaxIterations(false)
, mdLastMaxCov(std::numeric_limits<double>::max())
, mbUseRobust(bUseRobust)
, mbUseTukey(bUseTukey)
, mbVerbose(bVerbose)
, mnTotalIterations(0)
{
  // std::map iteration = name order, the order BundleAdjusterMulti meets the cameras in; any fixed order would do
  std::vector<mcp_camera> vCams;
  for(TaylorCameraMap::iterator it = mmCameraModels.begin(); it != mmCameraModels.end(); ++it)
  {
    mmCamIndex[it->first] = vCams.size();
    mvCamNames.push_back(it->first);
    vCams.push_back(mcptam_hip::CameraExport::Make(it->second));
  }

  mcp_ba_params params;
  params.max_iterations = ChainBundle::snMaxIterations;
  params.max_trials_after_failure = ChainBundle::snMaxTrialsAfterFailure;
  params.update_percent_limit = ChainBundle::sdUpdatePercentConvergenceLimit;
  params.update_rms_limit = ChainBundle::sdUpdateRMSConvergenceLimit;
  params.min_mestimator_sigma = ChainBundle::sdMinMEstimatorSigma;
  params.disable_convergence = 0;
  params.device = -1;   // the HIP device current on the calling thread
  params.profile = 0;

  mpHandle = mcp_ba_create(&vCams[0], vCams.size(), mbUseRobust, mbUseTukey, mbVerbose, &params);
  if(!mpHandle)
  {
    ROS_FATAL_STREAM("ChainBundle: cannot create the MI355X bundle: "<<mcp_last_error());
    ros::shutdown();
  }
}

ChainBundle::~ChainBundle()
{
  mcp_ba_destroy(mpHandle);
}

// ids start at 1 and are shared between poses and points, as mnCurrId did (src/ChainBundle.cc:1145,1201,1215)
int ChainBundle::AddPose(SE3<> se3PoseFromRef, bool bFixed)
{
  double R[9], t[3];
  ToArrays(se3PoseFromRef, R, t);
  return mcp_ba_add_pose(mpHandle, R, t, bFixed ? 1 : 0);
}

int ChainBundle::AddPoint(Vector<3> v3PointInCam, std::vector<int> vCams, bool bFixed)
{
  double x[3] = { v3PointInCam[0], v3PointInCam[1], v3PointInCam[2] };
  int nID = mcp_ba_add_point(mpHandle, x, &vCams[0], vCams.size(), bFixed ? 1 : 0);
  ROS_ASSERT_MSG(nID > 0, "ChainBundle::AddPoint: %s", mcp_last_error());
  return nID;
}

void ChainBundle::AddMeas(std::vector<int> vCams, int nPointIdx, Vector<2> v2Pos, double dNoiseSigmaSquared, std::string cameraName)
{
  ROS_ASSERT(mmCamIndex.count(cameraName));
  double uv[2] = { v2Pos[0], v2Pos[1] };
  // the information matrix I / sqrt(sigma^2) of :1244-1245 is formed inside the library
  int nRet = mcp_ba_add_meas(mpHandle, &vCams[0], vCams.size(), nPointIdx, uv, dNoiseSigmaSquared, mmCamIndex[cameraName]);
  ROS_ASSERT_MSG(nRet == 0, "ChainBundle::AddMeas: %s", mcp_last_error());
}

void ChainBundle::Initialize()
{
  mvOutlierMeasurementIdx.clear();
  mbConverged = false;
  mbHitMaxIterations = false;
}

int ChainBundle::Compute(bool* pAbortSignal, int nNumIter, double dUserLambda)
{
  Initialize();

  // bool is one byte on every ABI MCPTAM builds for; the tracker thread writes it, the solver raises it on convergence
  // (src/ChainBundle.cc:1027-1028,1105-1113) and the adapter clears it afterwards (src/BundleAdjusterMulti.cc:261)
  static_assert(sizeof(bool) == 1, "the abort flag is shared with the library as one byte");
  int nCounter = mcp_ba_compute(mpHandle, reinterpret_cast<volatile unsigned char*>(pAbortSignal), nNumIter, dUserLambda);
  if(nCounter == MCP_BA_ERR_RUNTIME)   // a device failure is not one of the reference's outcomes
  {
    ROS_FATAL_STREAM("ChainBundle: "<<mcp_last_error());
    ros::shutdown();
    return -1;
  }

  mnTotalIterations = mcp_ba_total_iterations(mpHandle);
  mbHitMaxIterations = (nCounter == nNumIter);
  mbConverged = mcp_ba_converged(mpHandle) != 0;
  mdLastMaxCov = mcp_ba_max_cov(mpHandle);

  // (point id, id of the first pose of the measuring chain, camera name), src/ChainBundle.cc:1385-1398
  int nOutliers = mcp_ba_num_outliers(mpHandle);
  std::vector<int> vRaw(3*nOutliers + 3);
  nOutliers = mcp_ba_get_outliers(mpHandle, &vRaw[0], nOutliers);
  for(int i = 0; i < nOutliers; ++i)
    mvOutlierMeasurementIdx.push_back(std::make_tuple(vRaw[3*i], vRaw[3*i + 1], mvCamNames[vRaw[3*i + 2]]));

  return nCounter;
}

Vector<3> ChainBundle::GetPoint(int n)
{
  double x[3];
  int nRet = mcp_ba_get_point(mpHandle, n, x);
  ROS_ASSERT_MSG(nRet == 0, "ChainBundle::GetPoint: %s", mcp_last_error());
  return makeVector(x[0], x[1], x[2]);
}

SE3<> ChainBundle::GetPose(int n)
{
  double R[9], t[3];
  int nRet = mcp_ba_get_pose(mpHandle, n, R, t);
  ROS_ASSERT_MSG(nRet == 0, "ChainBundle::GetPose: %s", mcp_last_error());
  Matrix<3> m3;
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 3; ++j)
      m3(i, j) = R[3*i + j];
  // the SO3 constructor re-orthonormalises (coerce), a no-op on a matrix that is a rotation to round-off
  return SE3<>(SO3<>(m3), makeVector(t[0], t[1], t[2]));
}

std::vector<std::tuple<int, int, std::string> > ChainBundle::GetOutlierMeasurements()
{
  return mvOutlierMeasurementIdx;
}

double ChainBundle::GetSigmaSquared()
{
  return mcp_ba_sigma_squared(mpHandle);
}

double ChainBundle::GetMeanChiSquared()
{
  return mcp_ba_mean_chi_squared(mpHandle);
}

double ChainBundle::GetLambda()
{
  return mcp_ba_lambda(mpHandle);
}
