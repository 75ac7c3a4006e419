// MapMakerServerBase_gpu.cc -- MI355X bodies of the map maker's PatchFinder callers.
//
// Replace, in /root/reference/src/MapMakerServerBase.cc:
//   * the two PatchFinder loops of AddPointEpipolar (:745-795 hypothesis search, :827-853 sub-pixel refinement) by the two helper
//     calls below (the arc construction :604-743 above them and the triangulation / MapPoint creation :855-918 below stay), and
//   * ReFind_Common (:921-1002), ReFindInSingleKeyFrame (:1005-1019), ReFindNewlyMade (:1024-1059), ReFindFromFailureQueue (:1062-1080)
//     by the bodies below.
// The reference runs ONE PatchFinder through each of these loops and PatchFinder is stateful (template cache of
// MakeTemplateCoarseCont, src/PatchFinder.cc:144-181; Jacobians / mean difference of the sub-pixel iteration), so the loops become
// SEQUENCES handed to mcp_patch_sequences (include/mcp_img.h): one wavefront walks one finder's items in order, its members live in
// an mcp_pf_state.  The map-side bookkeeping (measurement maps, never-retry sets, failure queue) is the reference's, unchanged.
// Needs KeyFrame::mpDev (shim/KeyFrame_gpu.cc) and shim/CameraExport.h.
#include <mcptam/MapMakerServerBase.h>
#include <mcptam/MapPoint.h>
#include <mcptam/KeyFrame.h>
#include <mcptam/LevelHelpers.h>
#include <mcp_img.h>
#include "CameraExport.h"
#include <ros/ros.h>
#include <algorithm>
#include <cstring>

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

void Identity12(double a[12])
{
  for(int i = 0; i < 12; ++i)
    a[i] = 0.0;
  a[0] = a[4] = a[8] = 1.0;
}

void FillPoint(const MapPoint& point, mcp_td_in& in)
{
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

int KeyOf(const MapPoint* pPoint)   // identity of a MapPoint object (the reference compares addresses, PatchFinder.cc:148)
{
  return (int)(reinterpret_cast<uintptr_t>(pPoint) >> 4);
}
}  // namespace

// ---- AddPointEpipolar, first loop (:745-795): every hypothesised position along the epipolar arc through ONE finder, in order.
// vMapPointPositions: (world position, position in the target camera) per step, as built at :706-723; `point` is the probe MapPoint
// of :726-738.  Fills vScoresIndicesBestMatches / nBest / nBestZMSSD / v2BestMatch exactly as the loop did; `state` is the finder
// (zero-initialised by the caller = `PatchFinder finder;` at :740) and has to be passed on to EpipolarRefine.
bool MapMakerServerBase::EpipolarSearch(KeyFrame& kfTarget, TaylorCamera& cameraTarget, MapPoint& point,
                                        const std::vector<std::pair<Vector<3>, Vector<3> > >& vMapPointPositions, mcp_pf_state& state,
                                        std::vector<mcp_pf_item>& vItems,
                                        std::vector<std::tuple<int, int, Vector<2> > >& vScoresIndicesBestMatches, int& nBest, int& nBestZMSSD,
                                        Vector<2>& v2BestMatch)
{
  const int n = (int)vMapPointPositions.size();
  vItems.resize(n);
  for(int i = 0; i < n; ++i)
  {
    point.mv3WorldPos = vMapPointPositions[i].first;
    point.RefreshPixelVectors();                    // src/MapPoint.cc:62-87, per hypothesis as at :749-750
    FillPoint(point, vItems[i].point);
    vItems[i].point_key = KeyOf(&point);            // one MapPoint object for all hypotheses: the template cache sees "the same point"
    vItems[i].target = 0;
    vItems[i].start_pos[0] = vItems[i].start_pos[1] = 0.0;
  }
  mcp_camera cam = mcptam_hip::CameraExport::Make(cameraTarget);
  mcp_pf_target target;
  target.kf = kfTarget.mpDev;                       // its level-0 mask (if any) was uploaded by MakeKeyFrame_Lite: the check of :763-765
  target.cam = &cam;
  ToArray12(kfTarget.mse3CamFromWorld, target.base_from_world);
  Identity12(target.cam_from_base);
  std::vector<mcp_td_out> vOut(n > 0 ? n : 1);
  const int anSeq[2] = { 0, n };
  if(mcp_patch_sequences(MCP_PF_EPI_COARSE, 1, &target, 1, anSeq, &vItems[0], &state, 3, 0, 0, &vOut[0]) != 0)   // range 3, :779
  {
    ROS_FATAL_STREAM("MapMakerServerBase::AddPointEpipolar: "<<mcp_last_error());
    ros::shutdown();
    return false;
  }
  for(int i = 0; i < n; ++i)
  {
    if(!vOut[i].found)
      continue;
    const Vector<2> v2Match = makeVector(vOut[i].found_pos[0], vOut[i].found_pos[1]);     // finder.GetCoarsePosAsVector()
    vScoresIndicesBestMatches.push_back(std::make_tuple(vOut[i].score, i, v2Match));
    if(vOut[i].score < nBestZMSSD)
    {
      nBestZMSSD = vOut[i].score;
      nBest = i;
      v2BestMatch = v2Match;
    }
  }
  return nBest != -1;
}

// ---- AddPointEpipolar, second loop (:827-853): the one to three surviving matches, best first, on the SAME finder; the first whose
// sub-pixel iteration converges wins.
bool MapMakerServerBase::EpipolarRefine(KeyFrame& kfTarget, TaylorCamera& cameraTarget, const std::vector<mcp_pf_item>& vItems, mcp_pf_state& state,
                                        const std::vector<std::tuple<int, int, Vector<2> > >& vScoresIndicesBestMatches, Vector<2>& v2SubPixPos)
{
  const int n = (int)vScoresIndicesBestMatches.size();
  std::vector<mcp_pf_item> vRefine(n);
  for(int i = 0; i < n; ++i)
  {
    vRefine[i] = vItems[std::get<1>(vScoresIndicesBestMatches[i])];     // same point, same pixel vectors as in the first loop (:835-836)
    vRefine[i].start_pos[0] = std::get<2>(vScoresIndicesBestMatches[i])[0];   // finder.SetSubPixPos(v2CurrBestMatch), :843
    vRefine[i].start_pos[1] = std::get<2>(vScoresIndicesBestMatches[i])[1];
  }
  mcp_camera cam = mcptam_hip::CameraExport::Make(cameraTarget);
  mcp_pf_target target;
  target.kf = kfTarget.mpDev;
  target.cam = &cam;
  ToArray12(kfTarget.mse3CamFromWorld, target.base_from_world);
  Identity12(target.cam_from_base);
  std::vector<mcp_td_out> vOut(n > 0 ? n : 1);
  const int anSeq[2] = { 0, n };
  if(mcp_patch_sequences(MCP_PF_EPI_REFINE, 1, &target, 1, anSeq, &vRefine[0], &state, 3, 10, 0, &vOut[0]) != 0)
  {
    ROS_FATAL_STREAM("MapMakerServerBase::AddPointEpipolar: "<<mcp_last_error());
    ros::shutdown();
    return false;
  }
  // The reference stops at the first candidate that converges; the later ones only change the finder, which is discarded
  // (`PatchFinder finder` is a local of AddPointEpipolar), so evaluating all of them is not observable.
  for(int i = 0; i < n; ++i)
  {
    if(vOut[i].found)
    {
      v2SubPixPos = makeVector(vOut[i].found_pos[0], vOut[i].found_pos[1]);
      return true;
    }
  }
  return false;
}

// ---- ReFind_Common (:921-1002) for one keyframe and MANY points, or one point and MANY keyframes: the checks that need no image
// (:925-937) first, then one sequence per finder.  The reference's finder is `static` (:939): its cache can only hit when consecutive
// calls carry the same MapPoint, which is what ReFindNewlyMade does (a new point walked over all keyframes), so there every point is
// one sequence over its keyframes; ReFindInSingleKeyFrame and ReFindFromFailureQueue change the point from call to call, i.e. every
// pair is a sequence of one.  mFinderStateReFind (a new member, zero-initialised) carries the static finder across calls.
int MapMakerServerBase::ReFindBatch(std::vector<std::pair<KeyFrame*, MapPoint*> >& vPairs, bool bOneFinderPerPoint)
{
  // the early-outs of :925-937
  std::vector<std::pair<KeyFrame*, MapPoint*> > vWork;
  static gvar3<int> gvnCrossCamera("CrossCamera", 1, HIDDEN|SILENT);
  for(unsigned i = 0; i < vPairs.size(); ++i)
  {
    KeyFrame& kf = *vPairs[i].first;
    MapPoint& point = *vPairs[i].second;
    if(point.mMMData.spMeasurementKFs.count(&kf) || point.mMMData.spNeverRetryKFs.count(&kf))
      continue;
    if(point.mbBad || kf.mpParent->mbBad)
      continue;
    if(!*gvnCrossCamera && kf.mCamName != point.mpPatchSourceKF->mCamName)
      continue;
    vWork.push_back(vPairs[i]);
  }
  if(vWork.empty())
    return 0;

  // targets: the distinct keyframes
  std::vector<KeyFrame*> vKFs;
  std::vector<mcp_camera> vCams;
  std::vector<mcp_pf_target> vTargets;
  std::map<KeyFrame*, int> mTargetIdx;
  for(unsigned i = 0; i < vWork.size(); ++i)
  {
    KeyFrame* pKF = vWork[i].first;
    if(mTargetIdx.count(pKF))
      continue;
    mTargetIdx[pKF] = (int)vKFs.size();
    vKFs.push_back(pKF);
  }
  vCams.resize(vKFs.size());
  vTargets.resize(vKFs.size());
  for(unsigned t = 0; t < vKFs.size(); ++t)
  {
    ROS_ASSERT(vKFs[t]->mpDev);
    vCams[t] = mcptam_hip::CameraExport::Make(mmCameraModels[vKFs[t]->mCamName]);
    vTargets[t].kf = vKFs[t]->mpDev;
    vTargets[t].cam = &vCams[t];
    ToArray12(vKFs[t]->mse3CamFromWorld, vTargets[t].base_from_world);
    Identity12(vTargets[t].cam_from_base);
  }

  // sequences
  std::vector<mcp_pf_item> vItems(vWork.size());
  std::vector<int> vSeqStart;
  std::vector<mcp_pf_state> vStates;
  for(unsigned i = 0; i < vWork.size(); ++i)
  {
    const bool bNewSeq = (i == 0) || !bOneFinderPerPoint || vWork[i].second != vWork[i - 1].second;
    if(bNewSeq)
    {
      vSeqStart.push_back((int)i);
      mcp_pf_state fresh;
      std::memset(&fresh, 0, sizeof fresh);
      vStates.push_back(i == 0 ? mFinderStateReFind : fresh);       // the static finder enters the first sequence ...
    }
    FillPoint(*vWork[i].second, vItems[i].point);
    vItems[i].point_key = KeyOf(vWork[i].second);
    vItems[i].target = mTargetIdx[vWork[i].first];
    vItems[i].start_pos[0] = vItems[i].start_pos[1] = 0.0;
  }
  vSeqStart.push_back((int)vWork.size());
  std::vector<mcp_td_out> vOut(vWork.size());
  if(mcp_patch_sequences(MCP_PF_REFIND, (int)vTargets.size(), &vTargets[0], (int)vStates.size(), &vSeqStart[0], &vItems[0], &vStates[0], 4, 8, 0, &vOut[0]) != 0)
  {
    ROS_FATAL_STREAM("MapMakerServerBase::ReFind: "<<mcp_last_error());
    ros::shutdown();
    return 0;
  }
  mFinderStateReFind = vStates.back();                                  // ... and leaves with the last one

  // the rest of ReFind_Common per pair (:941-1001)
  int nFound = 0;
  for(unsigned i = 0; i < vWork.size(); ++i)
  {
    KeyFrame& kf = *vWork[i].first;
    MapPoint& point = *vWork[i].second;
    const mcp_td_out& out = vOut[i];
    // camera.Invalid() / outside the image (:945-955), TemplateBad (:960-964), not found (:967-971)
    if(!out.in_image || out.template_bad || !out.found)
    {
      point.mMMData.spNeverRetryKFs.insert(&kf);
      continue;
    }
    Measurement* pMeas = new Measurement;
    pMeas->nLevel = out.search_level;
    pMeas->eSource = Measurement::SRC_REFIND;
    pMeas->v2RootPos = makeVector(out.found_pos[0], out.found_pos[1]);    // sub-pixel position above level 0 (kept converged or not), coarse at level 0
    pMeas->bSubPix = out.did_subpix != 0;
    if(kf.mmpMeasurements.count(&point))
      ROS_BREAK();
    kf.AddMeasurement(&point, pMeas);
    nFound++;
  }
  return nFound;
}

bool MapMakerServerBase::ReFind_Common(KeyFrame& kf, MapPoint& point)
{
  std::vector<std::pair<KeyFrame*, MapPoint*> > vPairs(1, std::make_pair(&kf, &point));
  return ReFindBatch(vPairs, false) == 1;
}

// A general data-association update for a single keyframe (:1005-1019): all map points against one keyframe in one call
int MapMakerServerBase::ReFindInSingleKeyFrame(KeyFrame& kf)
{
  std::vector<std::pair<KeyFrame*, MapPoint*> > vPairs;
  for(MapPointPtrList::iterator it = mMap.mlpPoints.begin(); it != mMap.mlpPoints.end(); ++it)
    vPairs.push_back(std::make_pair(&kf, *it));
  return ReFindBatch(vPairs, false);
}

// New map points against every keyframe (:1024-1059): one sequence per point, so that keyframes with similar warps share its
// template as they do through the reference's static finder.  The queue is drained in chunks so that the IncomingQueueSize()
// check of the reference keeps its meaning (it is evaluated between chunks instead of between single keyframes).
void MapMakerServerBase::ReFindNewlyMade()
{
  while(!mlpNewQueue.empty() && IncomingQueueSize() == 0)
  {
    std::vector<std::pair<KeyFrame*, MapPoint*> > vPairs;
    for(int nTaken = 0; nTaken < 64 && !mlpNewQueue.empty(); ++nTaken)
    {
      MapPoint* pPointNew = mlpNewQueue.front();
      mlpNewQueue.pop_front();
      if(pPointNew->mbBad)
        continue;
      for(MultiKeyFramePtrList::iterator it = mMap.mlpMultiKeyFrames.begin(); it != mMap.mlpMultiKeyFrames.end(); ++it)
      {
        MultiKeyFrame& mkf = *(*it);
        if(mkf.mbBad)
          continue;
        for(KeyFramePtrMap::iterator jiter = mkf.mmpKeyFrames.begin(); jiter != mkf.mmpKeyFrames.end(); ++jiter)
          vPairs.push_back(std::make_pair(jiter->second, pPointNew));
      }
    }
    ReFindBatch(vPairs, true);
  }
}

// Dud measurements get a second chance (:1062-1080)
void MapMakerServerBase::ReFindFromFailureQueue()
{
  if(mlFailureQueue.size() == 0)
    return;
  mlFailureQueue.sort();
  while(!mlFailureQueue.empty() && IncomingQueueSize() == 0)
  {
    std::vector<std::pair<KeyFrame*, MapPoint*> > vPairs;
    for(int nTaken = 0; nTaken < 256 && !mlFailureQueue.empty(); ++nTaken)
    {
      vPairs.push_back(mlFailureQueue.front());
      mlFailureQueue.pop_front();
    }
    ReFindBatch(vPairs, false);
  }
}
