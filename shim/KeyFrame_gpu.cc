// KeyFrame_gpu.cc -- MI355X bodies of KeyFrame::MakeKeyFrame_Lite and KeyFrame::MakeKeyFrame_Rest.
//
// Replaces /root/reference/src/KeyFrame.cc:144-360 and :362-536 (delete those two member functions there, or fence them with
// #ifndef MCPTAM_HIP, and add this file to the library's sources).  Everything else of KeyFrame.cc -- MultiKeyFrame, the scene
// depth statistics, ownership, serialisation -- is untouched.  Header delta (include/mcptam/KeyFrame.h, class KeyFrame):
//     struct mcp_kf;                       // before the class
//     mcp_kf* mpDev;                       // new member, NULL in the constructors, mcp_kf_destroy(mpDev) in ~KeyFrame and in
//                                          // whatever resets the levels (KeyFrame::RemoveImageData / EraseBackingData)
// The pyramid, FAST-10 corners + scores, adaptive threshold, masks, row LUTs, candidate scoring, non-maximum suppression and the
// back/forward MiniPatch stability test run on the device; the Level containers the rest of MCPTAM reads (image, vCorners,
// vCornerRowLUT, nFastThresh, vFastFrequency, vCandidates, lastMask) are filled from it.  The device handle keeps the previous two
// frames itself (Level::imagePrev / vCornersPrev, snNumPrev = 2), so the host circular buffers only carry the images for drawing.
#include <mcptam/KeyFrame.h>
#include <mcp_img.h>
#include <cvd/vision.h>
#include <ros/ros.h>
#include <cstring>

using namespace TooN;

namespace
{
// Level::lastMask is only drawn by the tracker (src/Tracker.cc:353-354): rebuild it on the host when someone will look at it.
// internal mask AND (optionally) "no pixel brighter than 245 within five 5x5 elliptical dilations" (src/KeyFrame.cc:214-241).
void FillLastMask(Level& lev, bool bGlareMasking)
{
  const CVD::ImageRef irSize = lev.image.size();
  lev.lastMask.resize(irSize);
  if(lev.mask.totalsize() > 0)
    CVD::copy(lev.mask, lev.lastMask);
  else
    lev.lastMask.fill(255);

  if(!bGlareMasking)
    return;

  // five passes of the 5x5 ellipse = one pass of their Minkowski sum; done separably would not be the same shape, so walk the
  // element.  Rows of cv::getStructuringElement(MORPH_ELLIPSE, 5x5): half-widths 0, 2, 2, 2, 0 about the centre column.
  static const int anHalf[5] = { 0, 2, 2, 2, 0 };
  CVD::Image<CVD::byte> imA(irSize), imB(irSize);
  CVD::copy(lev.image, imA);
  for(int nPass = 0; nPass < 5; ++nPass)
  {
    for(int y = 0; y < irSize.y; ++y)
      for(int x = 0; x < irSize.x; ++x)
      {
        CVD::byte best = 0;   // the border value of cv::dilate is -infinity: outside pixels never win
        for(int dy = -2; dy <= 2; ++dy)
        {
          const int yy = y + dy;
          if(yy < 0 || yy >= irSize.y)
            continue;
          for(int dx = -anHalf[dy + 2]; dx <= anHalf[dy + 2]; ++dx)
          {
            const int xx = x + dx;
            if(xx < 0 || xx >= irSize.x)
              continue;
            if(imA[yy][xx] > best)
              best = imA[yy][xx];
          }
        }
        imB[y][x] = best;
      }
    std::swap(imA, imB);
  }
  for(int y = 0; y < irSize.y; ++y)
    for(int x = 0; x < irSize.x; ++x)
      if(imA[y][x] > 245)
        lev.lastMask[y][x] = 0;
}
}  // namespace

std::tuple<double, double, double> KeyFrame::MakeKeyFrame_Lite(CVD::Image<CVD::byte>& im, bool bDeepCopy, bool bGlareMasking)
{
  ros::WallTime startTime = ros::WallTime::now();

  // host-side history for the GUI: the previous pyramid moves into imagePrev / vCornersPrev exactly when the handle pushes its own
  const bool bPushBack = maLevels[0].image.totalsize() > 0;
  for(int i = 0; i < LEVELS; ++i)
  {
    Level& lev = maLevels[i];
    if(bPushBack)
    {
      lev.imagePrev.push_back(lev.image);
      lev.vCornersPrev.push_back(std::vector<CVD::ImageRef>());
      lev.vCornersPrev.back().swap(lev.vCorners);
    }
    lev.vCorners.clear();
    lev.vCandidates.clear();
    lev.vScoresAndMaxCorners.clear();
    lev.vFastFrequency = TooN::Zeros;
    lev.nFastThresh = 0;
  }

  if(bDeepCopy)
  {
    maLevels[0].image.resize(im.size());
    CVD::copy(im, maLevels[0].image);
  }
  else
    maLevels[0].image = im;   // reference-counted, no pixel copy

  if(!mpDev)
  {
    mcp_kf_params params;
    params.adaptive_thresh = KeyFrame::sbAdaptiveThresh ? 1 : 0;
    params.glare_masking = bGlareMasking ? 1 : 0;
    params.half_sample_pavgb = 0;   // libCVD's generic halfSample (truncating mean); 1 = its SSE2 byte path
    params.device = -1;
    mpDev = mcp_kf_create(im.size().x, im.size().y, &params);
    if(!mpDev)
    {
      ROS_FATAL_STREAM("KeyFrame: cannot create the MI355X keyframe: "<<mcp_last_error());
      ros::shutdown();
      return std::make_tuple(0.0, 0.0, 0.0);
    }
  }

  // internal masks per level (NULL = none); they must be tightly packed, which CVD::Image is when it owns its pixels
  const uint8_t* apMasks[LEVELS];
  bool bAnyMask = false;
  for(int i = 0; i < LEVELS; ++i)
  {
    Level& lev = maLevels[i];
    const bool bHave = lev.mask.totalsize() > 0;
    ROS_ASSERT(!bHave || lev.mask.row_stride() == lev.mask.size().x);
    apMasks[i] = bHave ? lev.mask.data() : NULL;
    bAnyMask |= bHave;
  }

  if(mcp_kf_make_lite(mpDev, maLevels[0].image.data(), maLevels[0].image.row_stride(), bAnyMask ? apMasks : NULL) != 0)
  {
    ROS_FATAL_STREAM("KeyFrame::MakeKeyFrame_Lite: "<<mcp_last_error());
    ros::shutdown();
    return std::make_tuple(0.0, 0.0, 0.0);
  }
  const double dDeviceTime = (ros::WallTime::now() - startTime).toSec();
  startTime = ros::WallTime::now();

  // read back what host code consumes
  static_assert(sizeof(CVD::ImageRef) == sizeof(mcp_int2), "CVD::ImageRef is two ints");
  for(int i = 0; i < LEVELS; ++i)
  {
    Level& lev = maLevels[i];
    int w = 0, h = 0;
    mcp_kf_level_size(mpDev, i, &w, &h);
    if(i != 0)
    {
      lev.image.resize(CVD::ImageRef(w, h));
      mcp_kf_get_image(mpDev, i, lev.image.data());
    }
    const int nCorners = mcp_kf_num_corners(mpDev, i);
    lev.vCorners.resize(nCorners);
    if(nCorners > 0)
      mcp_kf_get_corners(mpDev, i, reinterpret_cast<mcp_int2*>(&lev.vCorners[0]), nCorners);
    lev.vCornerRowLUT.resize(h);
    if(h > 0)
      mcp_kf_get_row_lut(mpDev, i, &lev.vCornerRowLUT[0]);
    lev.nFastThresh = mcp_kf_fast_thresh(mpDev, i);
    double adFreq[MAX_FAST_THRESH + 1];
    mcp_kf_get_fast_frequency(mpDev, i, adFreq);
    for(int t = 0; t <= MAX_FAST_THRESH; ++t)
      lev.vFastFrequency[t] = (int)adFreq[t];
    if(bGlareMasking || lev.mask.totalsize() > 0)
      FillLastMask(lev, bGlareMasking);
    else
      lev.lastMask = CVD::Image<CVD::byte>();
  }
  const double dReadbackTime = (ros::WallTime::now() - startTime).toSec();

  // (down-sample, mask, feature) seconds as the Tracker's timing message reports them: the device does all three in one go
  return std::make_tuple(0.0, dReadbackTime, dDeviceTime);
}

void KeyFrame::MakeKeyFrame_Rest()
{
  ROS_ASSERT(mpDev);
  const bool bShi = (KeyFrame::ssCandidateType == "shi");
  const bool bPercent = (KeyFrame::ssCandidateCriterion == "percent");
  // nonmax_score 0: CVD::fast_nonmax scoring with the FAST-10 score (see include/mcp_img.h on libCVD vintages)
  if(mcp_kf_make_rest(mpDev, bShi ? 1 : 0, bPercent ? 1 : 0, KeyFrame::sdCandidateTopFraction, KeyFrame::sdCandidateThresh, 0) != 0)
  {
    ROS_FATAL_STREAM("KeyFrame::MakeKeyFrame_Rest: "<<mcp_last_error());
    ros::shutdown();
    return;
  }
  for(int l = 0; l < LEVELS; ++l)
  {
    Level& lev = maLevels[l];
    const int n = mcp_kf_num_candidates(mpDev, l);
    std::vector<mcp_int2> vPos(n > 0 ? n : 1);
    std::vector<double> vScore(n > 0 ? n : 1);
    if(n > 0)
      mcp_kf_get_candidates(mpDev, l, &vPos[0], &vScore[0], n);
    lev.vCandidates.clear();
    lev.vScoresAndMaxCorners.clear();
    for(int i = 0; i < n; ++i)
    {
      Candidate c;
      c.irLevelPos = CVD::ImageRef(vPos[i].x, vPos[i].y);
      c.dSTScore = vScore[i];
      lev.vCandidates.push_back(c);
      // (only the survivors are known on the host; nothing outside this function reads vScoresAndMaxCorners)
      lev.vScoresAndMaxCorners.push_back(std::make_pair(vScore[i], c.irLevelPos));
    }
  }
  // the reference ends with MakeSBI() (src/KeyFrame.cc:536): the host SmallBlurryImage stays for code that reads mpSBI directly; the
  // device twin (thumbnail, blurred template, gradients) serves Relocaliser::ScoreKFs / CalcSBIRotation through mcp_sbi_*
  MakeSBI();
  mcp_kf_make_sbi(mpDev, 2.5);
}
