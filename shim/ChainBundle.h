// ChainBundle.h -- drop-in replacement of include/mcptam/ChainBundle.h for the MI355X back end.
//
// Same class name, same public members with the same signatures, defaults and statics as the reference header
// (/root/reference/include/mcptam/ChainBundle.h:97-186): BundleAdjusterMulti / Single / Calib and the map makers compile against
// it unchanged.  What goes away are the g2o forward declarations and members (:21-36, :188-222): the optimiser, the robust-kernel
// data, the five action objects and the pose-chain helper map all live behind one `mcp_ba*` handle (include/mcp_ba.h).
#ifndef MCPTAM_CHAINBUNDLE_H
#define MCPTAM_CHAINBUNDLE_H

#include <mcptam/Types.h>
#include <TooN/TooN.h>
#include <TooN/se3.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>

struct mcp_ba;      // include/mcp_ba.h

class ChainBundle
{
public:
  ChainBundle(TaylorCameraMap& cameraModels, bool bUseRobust, bool bUseTukey, bool bVerbose);
  ~ChainBundle();

  int AddPose(TooN::SE3<> se3PoseFromRef, bool bFixed);
  int AddPoint(TooN::Vector<3> v3PointInCam, std::vector<int> vCams, bool bFixed);
  void AddMeas(std::vector<int> vCams, int nPointIdx, TooN::Vector<2> v2Pos, double dNoiseSigmaSquared, std::string cameraName);

  /// Kept for source compatibility (the reference's Compute calls it first); the handle prepares itself inside Compute.
  void Initialize();

  int Compute(bool* pAbortSignal, int nNumIter = ChainBundle::snMaxIterations, double dUserLambda = -1);

  inline bool Converged() { return mbConverged; }
  inline int TotalIterations() { return mnTotalIterations; }

  TooN::Vector<3> GetPoint(int n);
  TooN::SE3<> GetPose(int n);
  std::vector<std::tuple<int, int, std::string> > GetOutlierMeasurements();
  double GetSigmaSquared();
  double GetMeanChiSquared();
  double GetMaxCov() { return mdLastMaxCov; }
  double GetLambda();

  static int snMaxIterations;
  static int snMaxTrialsAfterFailure;
  static double sdUpdatePercentConvergenceLimit;
  static double sdUpdateRMSConvergenceLimit;
  static double sdMinMEstimatorSigma;

protected:
  mcp_ba* mpHandle;                                   ///< the device-side bundle (mcp_ba_create .. mcp_ba_destroy)
  std::map<std::string, int> mmCamIndex;              ///< camera name -> index into the camera array handed to the handle
  std::vector<std::string> mvCamNames;                ///< the inverse
  std::vector<std::tuple<int, int, std::string> > mvOutlierMeasurementIdx;
  TaylorCameraMap mmCameraModels;
  bool mbConverged;
  bool mbHitMaxIterations;
  double mdLastMaxCov;
  bool mbUseRobust;
  bool mbUseTukey;
  bool mbVerbose;
  int mnTotalIterations;
};

#endif
