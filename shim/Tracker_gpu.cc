// Tracker_gpu.cc -- MI355X bodies of Tracker::SearchForPoints and Tracker::CalcPoseUpdate.
//
// Replace /root/reference/src/Tracker.cc:1297-1377 (SearchForPoints) and :1379-1512 (CalcPoseUpdate): delete those two member
// functions there (or fence them with #ifndef MCPTAM_HIP) and add this file to the library's sources.  The tracker keeps all of its
// control flow -- FindPVS, the level buckets, the 1000-patch budget, the shuffles, the coarse / fine stages, the motion model -- and
// calls these two members exactly where it did (src/Tracker.cc:841-906, 1027-1075).  The per-point inner loops (TrackerData::Project /
// CalcJacobian, PatchFinder::MakeTemplateCoarseCont / FindPatchCoarse / IterateSubPixToConvergence, the WLS accumulation) run as one
// batched device call each.  Needs KeyFrame::mpDev (shim/KeyFrame_gpu.cc) and shim/CameraExport.h.
#include <mcptam/Tracker.h>
#include <mcptam/TrackerData.h>
#include <mcptam/MapPoint.h>
#include <mcptam/KeyFrame.h>
#include <mcptam/LevelHelpers.h>
#include <mcp_img.h>
#include "CameraExport.h"
#include <TooN/SVD.h>
#include <ros/ros.h>

using namespace TooN;


namespace
{
void ToArray12(const SE3<>& se3, double a[12])
{
  const Matrix<3>& m3 = se3.get_rotation().get_matrix();
  for(int i = 0; i < 3; ++i)
  {
    for(int j = 0; j < 3; ++j)
      a[3*i + j] = m3(i, j);
    a[9 + i] = se3.get_translation()[i];
  }
}
}  // namespace

// Find points in the image: one device call for the whole vector (include/mcp_img.h, mcp_patch_sequences in MCP_PF_TRACK mode).
// PatchFinder's members that live from frame to frame -- the template cache of MakeTemplateCoarseCont (src/PatchFinder.cc:144-181)
// and the sub-pixel state -- move from TrackerData::mFinder into an mcp_pf_state per TrackerData: add
//     mcp_pf_state mFinderState;      // zero-initialised in the constructor (a PatchFinder that has seen nothing)
// to class TrackerData (include/mcptam/TrackerData.h:75, next to mFinder, which keeps CalcSearchLevelAndWarpMatrix for FindPVS).
// One sequence of one item per tracked point; the map point's address is its key, as the reference compares &point.
int Tracker::SearchForPoints(TrackerDataPtrVector& vTD, std::string cameraName, int nRange, int nSubPixIts, bool bExhaustive)
{
  if(vTD.empty())
    return 0;

  KeyFrame& kf = *mpCurrentMKF->mmpKeyFrames[cameraName];
  ROS_ASSERT(kf.mpDev);

  std::vector<mcp_pf_item> vIn(vTD.size());
  std::vector<mcp_td_out> vOut(vTD.size());
  std::vector<mcp_pf_state> vState(vTD.size());
  std::vector<int> vSeqStart(vTD.size() + 1);
  for(unsigned i = 0; i < vTD.size(); ++i)
  {
    MapPoint& point = vTD[i]->mPoint;
    vSeqStart[i] = (int)i;
    vState[i] = vTD[i]->mFinderState;
    vIn[i].point_key = (int)(reinterpret_cast<uintptr_t>(&point) >> 4);   // identity of the MapPoint object (the finder of a TrackerData only ever sees this one)
    vIn[i].target = 0;
    vIn[i].start_pos[0] = vIn[i].start_pos[1] = 0.0;
    mcp_td_in& in = vIn[i].point;
    for(int k = 0; k < 3; ++k)
    {
      in.world_pos[k] = point.mv3WorldPos[k];
      in.pixel_right_w[k] = point.mv3PixelRight_W[k];
      in.pixel_down_w[k] = point.mv3PixelDown_W[k];
    }
    ROS_ASSERT(point.mpPatchSourceKF && point.mpPatchSourceKF->mpDev);   // source pyramids stay resident on the device
    in.source_kf = point.mpPatchSourceKF->mpDev;
    in.source_level = point.mnSourceLevel;
    in.center_x = point.mirCenter.x;
    in.center_y = point.mirCenter.y;
    in.fixed = point.mbFixed ? 1 : 0;
  }

  vSeqStart[vTD.size()] = (int)vTD.size();
  mcp_camera cam = mcptam_hip::CameraExport::Make(mmCameraModels[cameraName]);
  mcp_pf_target target;
  target.kf = kf.mpDev;
  target.cam = &cam;
  ToArray12(mpCurrentMKF->mse3BaseFromWorld, target.base_from_world);
  ToArray12(kf.mse3CamFromBase, target.cam_from_base);

  if(mcp_patch_sequences(MCP_PF_TRACK, 1, &target, (int)vTD.size(), &vSeqStart[0], &vIn[0], &vState[0], nRange, nSubPixIts, bExhaustive ? 1 : 0, &vOut[0]) != 0)
  {
    ROS_FATAL_STREAM("Tracker::SearchForPoints: "<<mcp_last_error());
    ros::shutdown();
    return 0;
  }

  int nFound = 0;
  for(unsigned i = 0; i < vTD.size(); ++i)
  {
    TrackerData& td = *vTD[i];
    const mcp_td_out& out = vOut[i];
    td.mFinderState = vState[i];      // the finder's members after this frame

    // the device re-derives the projection it searches around: identical to what FindPVS left in the TrackerData
    td.mv2Image = makeVector(out.image[0], out.image[1]);
    td.mm2CamDerivs(0, 0) = out.cam_derivs[0]; td.mm2CamDerivs(0, 1) = out.cam_derivs[1];
    td.mm2CamDerivs(1, 0) = out.cam_derivs[2]; td.mm2CamDerivs(1, 1) = out.cam_derivs[3];
    for(int r = 0; r < 2; ++r)
      for(int c = 0; c < 6; ++c)
        td.mm26Jacobian(r, c) = out.jacobian[6*r + c];
    td.mnSearchLevel = out.search_level;

    if(out.template_bad)   // PatchFinder::TemplateBad(): warp rejected or source footprint outside the source level
    {
      td.mbInImage = td.mbFound = false;
      continue;
    }
    mmMeasAttemptedLevels[cameraName][out.search_level]++;

    td.mbSearched = out.searched != 0;
    td.mbFound = out.found != 0;
    td.mbDidSubPix = out.did_subpix != 0;
    if(!td.mbFound)
      continue;     // not found in the coarse stage, or the sub-pixel iterations did not converge (counters as the reference nets them)

    td.mdSqrtInvNoise = out.sqrt_inv_noise;   // 1 / LevelScale(search level)
    td.mv2Found = makeVector(out.found_pos[0], out.found_pos[1]);
    nFound++;
    mmMeasFoundLevels[cameraName][out.search_level]++;
  }
  return nFound;
}

// Pose update from the found measurements: M-estimator weights + WLS<6> with prior 100 on the device (mcp_track_pose_update_m),
// with the estimator Tracker::sMEstimatorName selects (src/Tracker.cc:1388-1401): Tukey, Cauchy or Huber.
Vector<6> Tracker::CalcPoseUpdate(std::vector<TrackerDataPtrVector>& vIterationSets, double dOverrideSigma, bool bMarkOutliers)
{
  int nEstimator = MCP_MEST_TUKEY;
  if(Tracker::sMEstimatorName == "Tukey")
    nEstimator = MCP_MEST_TUKEY;
  else if(Tracker::sMEstimatorName == "Cauchy")
    nEstimator = MCP_MEST_CAUCHY;
  else if(Tracker::sMEstimatorName == "Huber")
    nEstimator = MCP_MEST_HUBER;
  else
  {
    ROS_FATAL_STREAM("Tracker: Invalid Tracker MEstimator selected: "<<Tracker::sMEstimatorName<<", choices are [Tukey, Cauchy, Huber]");
    ros::shutdown();
    return makeVector(0, 0, 0, 0, 0, 0);
  }

  std::vector<TrackerData*> vpTD;
  for(unsigned i = 0; i < mvCurrCamNames.size(); ++i)
    for(unsigned j = 0; j < vIterationSets[i].size(); ++j)
      vpTD.push_back(vIterationSets[i][j].get());

  const int n = (int)vpTD.size();
  std::vector<uint8_t> vFound(n > 0 ? n : 1);
  std::vector<double> vFoundPos(2*n + 2), vImagePos(2*n + 2), vSqrtInvNoise(n + 1), vJac(12*n + 12), vWeights(n + 1);
  int nUsed = 0;
  for(int i = 0; i < n; ++i)
  {
    TrackerData& td = *vpTD[i];
    vFound[i] = td.mbFound ? 1 : 0;
    if(!td.mbFound)
      continue;
    ++nUsed;
    td.mv2Error_CovScaled = td.mdSqrtInvNoise * (td.mv2Found - td.mv2Image);
    vFoundPos[2*i] = td.mv2Found[0]; vFoundPos[2*i + 1] = td.mv2Found[1];
    vImagePos[2*i] = td.mv2Image[0]; vImagePos[2*i + 1] = td.mv2Image[1];
    vSqrtInvNoise[i] = td.mdSqrtInvNoise;
    for(int r = 0; r < 2; ++r)
      for(int c = 0; c < 6; ++c)
        vJac[12*i + 6*r + c] = td.mm26Jacobian(r, c);
  }
  if(nUsed == 0)
    return makeVector(0, 0, 0, 0, 0, 0);

  double adMu[6], dSigmaSquared = 0;
  if(mcp_track_pose_update_m(n, &vFound[0], &vFoundPos[0], &vImagePos[0], &vSqrtInvNoise[0], &vJac[0], dOverrideSigma, adMu, &vWeights[0], &dSigmaSquared, nEstimator) != 0)
  {
    ROS_FATAL_STREAM("Tracker::CalcPoseUpdate: "<<mcp_last_error());
    ros::shutdown();
    return makeVector(0, 0, 0, 0, 0, 0);
  }

  // inlier / outlier bookkeeping of the marking iteration (src/Tracker.cc:1448-1487) and the covariance (:1500-1502)
  mnNumInliers = 0;
  Matrix<6> m6CInv = 100.0 * Identity;      // wls.add_prior(100)
  for(int i = 0; i < n; ++i)
  {
    TrackerData& td = *vpTD[i];
    if(!td.mbFound)
    {
      if(td.mbSearched && bMarkOutliers && !IsLost())
        td.mPoint.mnMEstimatorOutlierCount++;
      continue;
    }
    const double dWeight = vWeights[i];
    if(dWeight == 0.0)
    {
      if(bMarkOutliers)
        td.mPoint.mnMEstimatorOutlierCount++;
      continue;
    }
    if(bMarkOutliers)
    {
      td.mPoint.mnMEstimatorInlierCount++;
      mnNumInliers++;
      // C^-1 += w J^T J with J = sqrt_inv_noise * Jacobian row (WLS::add_mJ twice)
      for(int r = 0; r < 2; ++r)
      {
        const Vector<6> v6 = td.mdSqrtInvNoise * td.mm26Jacobian[r];
        m6CInv += dWeight * v6.as_col() * v6.as_row();
      }
    }
  }
  if(bMarkOutliers)
    mm6PoseCovariance = TooN::SVD<6>(m6CInv).get_pinv();

  return makeVector(adMu[0], adMu[1], adMu[2], adMu[3], adMu[4], adMu[5]);
}


// ---- one TrackMap stage in one submission (optional) -----------------------------------------------------------------------------
// Tracker::TrackMap spends a stage as: SearchForPoints per camera, then ten PoseUpdateStep / PoseUpdateStepLinear iterations over all
// cameras (src/Tracker.cc:1013-1035 coarse, 1043-1075 fine).  mcp_track_frame runs that whole stage -- and, when the frame has just
// arrived, the MakeKeyFrame_Lite of every camera before it (src/Tracker.cc:303-318) -- as one device submission with one wait: the
// searches of all cameras in one launch with the persistent finders, the pose-iteration records built on the device, the ten iterations
// in one kernel.  A maintainer who wants it declares
//     Vector<6> TrackStageOnDevice(std::vector<TrackerDataPtrVector>& vIterationSets, int nRange, int nSubPixIts, bool bFineStage, bool bMakeLite);
// in Tracker.h and calls it where the stage's SearchForPoints loop + iteration loop stood, after having collected every camera's
// points to search into vIterationSets[i] (TestForCoarse / SetupFineTracking minus their SearchForPoints calls).  Results are those
// of the per-camera calls above followed by mcp_track_pose_refine_m (tests/test_img_gpu.py::
// test_track_frame_in_one_submission_equals_the_three_calls).  The outlier marking of the last fine iteration stays on the host:
// weights_last is what CalcPoseUpdate's marking branch reads.
Vector<6> Tracker::TrackStageOnDevice(std::vector<TrackerDataPtrVector>& vIterationSets, int nRange, int nSubPixIts, bool bFineStage, bool bMakeLite)
{
  const int nCams = (int)mvCurrCamNames.size();
  std::vector<mcp_kf*> vKF(nCams);
  std::vector<mcp_camera> vCams(nCams);
  std::vector<double> vCfB(12 * nCams);
  std::vector<int> vN(nCams);
  std::vector<std::vector<mcp_td_in> > vvIn(nCams);
  std::vector<std::vector<int> > vvKey(nCams);
  std::vector<std::vector<mcp_pf_state> > vvState(nCams);
  std::vector<std::vector<mcp_td_out> > vvOut(nCams);
  std::vector<const mcp_td_in*> vpIn(nCams);
  std::vector<const int*> vpKey(nCams);
  std::vector<mcp_pf_state*> vpState(nCams);
  std::vector<mcp_td_out*> vpOut(nCams);
  std::vector<const uint8_t*> vpImg(nCams);
  std::vector<int> vStride(nCams);
  int nTotal = 0;
  for(int c = 0; c < nCams; ++c)
  {
    KeyFrame& kf = *mpCurrentMKF->mmpKeyFrames[mvCurrCamNames[c]];
    ROS_ASSERT(kf.mpDev);
    vKF[c] = kf.mpDev;
    vCams[c] = mcptam_hip::CameraExport::Make(mmCameraModels[mvCurrCamNames[c]]);
    ToArray12(kf.mse3CamFromBase, &vCfB[12*c]);
    TrackerDataPtrVector& vTD = vIterationSets[c];
    vN[c] = (int)vTD.size();
    vvIn[c].resize(vTD.size()); vvKey[c].resize(vTD.size()); vvState[c].resize(vTD.size()); vvOut[c].resize(vTD.size());
    for(unsigned i = 0; i < vTD.size(); ++i)
    {
      MapPoint& point = vTD[i]->mPoint;
      mcp_td_in& in = vvIn[c][i];
      for(int k = 0; k < 3; ++k)
      {
        in.world_pos[k] = point.mv3WorldPos[k];
        in.pixel_right_w[k] = point.mv3PixelRight_W[k];
        in.pixel_down_w[k] = point.mv3PixelDown_W[k];
      }
      in.source_kf = point.mpPatchSourceKF->mpDev;
      in.source_level = point.mnSourceLevel;
      in.center_x = point.mirCenter.x;
      in.center_y = point.mirCenter.y;
      in.fixed = point.mbFixed ? 1 : 0;
      vvKey[c][i] = (int)(reinterpret_cast<uintptr_t>(&point) >> 4);
      vvState[c][i] = vTD[i]->mFinderState;
    }
    vpIn[c] = vvIn[c].empty() ? NULL : &vvIn[c][0];
    vpKey[c] = vvKey[c].empty() ? NULL : &vvKey[c][0];
    vpState[c] = vvState[c].empty() ? NULL : &vvState[c][0];
    vpOut[c] = vvOut[c].empty() ? NULL : &vvOut[c][0];
    vpImg[c] = kf.maLevels[0].image.data();                 // only read when bMakeLite (the frame the tracker was handed)
    vStride[c] = kf.maLevels[0].image.row_stride();
    nTotal += vN[c];
  }

  // coarse stage: ten full re-projections, sigma override 1.0; fine stage: re-projection at 0, 4, 9, override 16.0 (src/Tracker.cc:1027-1030,
  // 1063-1072); PoseUpdateStep / PoseUpdateStepLinear drop the override up to the fifth iteration (:800-802)
  uint8_t abNonlinear[10];
  double adOverride[10];
  for(int it = 0; it < 10; ++it)
  {
    abNonlinear[it] = (!bFineStage || it == 0 || it == 4 || it == 9) ? 1 : 0;
    adOverride[it] = it <= 5 ? 0.0 : (bFineStage ? 16.0 : 1.0);
  }
  int nEstimator = MCP_MEST_TUKEY;
  if(Tracker::sMEstimatorName == "Cauchy") nEstimator = MCP_MEST_CAUCHY;
  else if(Tracker::sMEstimatorName == "Huber") nEstimator = MCP_MEST_HUBER;

  double adBfW[12], adMu[6];
  ToArray12(mpCurrentMKF->mse3BaseFromWorld, adBfW);
  std::vector<double> vWeights(std::max(nTotal, 1));
  if(mcp_track_frame(nCams, &vKF[0], bMakeLite ? &vpImg[0] : NULL, &vStride[0], 0, NULL, &vCams[0], adBfW, &vCfB[0], &vN[0], &vpIn[0], &vpKey[0], &vpState[0],
                     nRange, nSubPixIts, 0, 10, abNonlinear, adOverride, nEstimator, &vpOut[0], NULL, adMu, &vWeights[0]) != 0)
  {
    ROS_FATAL_STREAM("Tracker::TrackStageOnDevice: "<<mcp_last_error());
    ros::shutdown();
    return Zeros;
  }

  // BaseFromWorld as the ten iterations left it, TrackerData as SearchForPoints leaves it
  Matrix<3> m3R;
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 3; ++j)
      m3R(i, j) = adBfW[3*i + j];
  mpCurrentMKF->mse3BaseFromWorld = SE3<>(SO3<>(m3R), makeVector(adBfW[9], adBfW[10], adBfW[11]));
  int w = 0;
  for(int c = 0; c < nCams; ++c)
  {
    TrackerDataPtrVector& vTD = vIterationSets[c];
    for(unsigned i = 0; i < vTD.size(); ++i, ++w)
    {
      TrackerData& td = *vTD[i];
      const mcp_td_out& out = vvOut[c][i];
      td.mFinderState = vvState[c][i];
      td.mnSearchLevel = out.search_level;
      td.mbSearched = out.searched != 0;
      td.mbFound = out.found != 0 && !out.template_bad;
      td.mbDidSubPix = out.did_subpix != 0;
      if(out.template_bad) { td.mbInImage = false; continue; }
      if(!td.mbFound) continue;
      td.mdSqrtInvNoise = out.sqrt_inv_noise;
      td.mv2Found = makeVector(out.found_pos[0], out.found_pos[1]);
      if(bFineStage && !IsLost())        // the marking iteration's bookkeeping (src/Tracker.cc:1448-1487)
      {
        if(vWeights[w] == 0.0) td.mPoint.mnMEstimatorOutlierCount++;
        else td.mPoint.mnMEstimatorInlierCount++;
      }
    }
  }
  return makeVector(adMu[0], adMu[1], adMu[2], adMu[3], adMu[4], adMu[5]);
}
