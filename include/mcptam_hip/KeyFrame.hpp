// KeyFrame.hpp -- C++ host-side mirror of MCPTAM's KeyFrame / Level, SmallBlurryImage, Relocaliser scoring and the per-point
// part of Tracker over the C ABI of mcp_img.h.
//
// Member names and argument meaning follow the reference classes (/root/reference/include/mcptam/KeyFrame.h:93-260,
// SmallBlurryImage.h, Relocaliser.h, MiniPatch.h, Tracker.h); CVD / TooN types are replaced by plain arrays (images: packed
// bytes, SE3 = row-major R[9] + t[3] as 12 doubles, SE2 = { R00, R01, R10, R11, tx, ty }) so that the header depends on the
// C ABI only.  A MCPTAM tree uses the CVD/TooN-typed shims of INTEGRATION.md; this is what a stand-alone C++ caller and the
// C++ link test under tests/cpp use.  There is no CPU fallback: constructors throw when no gfx950 device is usable.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../mcp_img.h"

namespace mcptam_hip {

/// What MakeKeyFrame_Lite / MakeKeyFrame_Rest leave in one pyramid level (KeyFrame.h:93-150), read back from the device.
struct Level {
  int w = 0, h = 0;
  std::vector<uint8_t> image;                 // Level::image, packed
  std::vector<mcp_int2> vCorners;             // raster order
  std::vector<int> vCornerRowLUT;
  int nFastThresh = 0;
  std::vector<double> vFastFrequency;         // index = threshold, MCP_MAX_FAST_THRESH + 1 entries
  std::vector<mcp_int2> vCandidates;          // Candidate::irLevelPos
  std::vector<double> vCandidateScores;       // Candidate::dSTScore
};

class KeyFrame {
 public:
  // statics / GVars of the reference (src/KeyFrame.cc:56-63, src/System.cc:121)
  static inline bool sbAdaptiveThresh = true;
  static inline double sdCandidateThresh = 70;
  static inline double sdCandidateTopFraction = 0.8;
  static inline std::string ssCandidateType = "fast";              // "fast" | "shi"
  static inline std::string ssCandidateCriterion = "percent";      // "percent" | "thresh"

  KeyFrame(int w, int h, bool bGlareMasking = false, bool bHalfSamplePavgb = false, int device = -1) {
    mcp_kf_params p;
    p.adaptive_thresh = sbAdaptiveThresh; p.glare_masking = bGlareMasking; p.half_sample_pavgb = bHalfSamplePavgb; p.device = device;
    mpDev = mcp_kf_create(w, h, &p);
    if (!mpDev) throw std::runtime_error(std::string("KeyFrame: ") + mcp_last_error());
  }
  ~KeyFrame() { mcp_kf_destroy(mpDev); }
  KeyFrame(const KeyFrame&) = delete;
  KeyFrame& operator=(const KeyFrame&) = delete;

  /// KeyFrame::MakeKeyFrame_Lite (src/KeyFrame.cc:145-360).  masks: nullptr or MCP_LEVELS pointers (each nullptr or packed).
  void MakeKeyFrame_Lite(const uint8_t* im, int stride, const uint8_t* const* masks = nullptr) {
    if (mcp_kf_make_lite(mpDev, im, stride, masks) != 0) throw std::runtime_error(mcp_last_error());
  }
  /// KeyFrame::MakeKeyFrame_Rest, candidate part incl. the stability pruning against the stored history (:363-527)
  void MakeKeyFrame_Rest(int nNonmaxScore = 0) {
    if (mcp_kf_make_rest(mpDev, ssCandidateType == "shi", ssCandidateCriterion == "percent", sdCandidateTopFraction,
                         sdCandidateThresh, nNonmaxScore) != 0) throw std::runtime_error(mcp_last_error());
  }
  /// frames in Level::imagePrev / vCornersPrev (0..2)
  int NumPrev() { return mcp_kf_num_prev(mpDev); }

  Level GetLevel(int l) {
    Level L;
    if (mcp_kf_level_size(mpDev, l, &L.w, &L.h) != 0) throw std::out_of_range(mcp_last_error());
    L.image.resize((size_t)L.w*L.h);
    check(mcp_kf_get_image(mpDev, l, L.image.data()));
    L.vCorners.resize((size_t)std::max(0, mcp_kf_num_corners(mpDev, l)));
    if (!L.vCorners.empty()) L.vCorners.resize((size_t)mcp_kf_get_corners(mpDev, l, L.vCorners.data(), (int)L.vCorners.size()));
    L.vCornerRowLUT.resize(L.h);
    check(mcp_kf_get_row_lut(mpDev, l, L.vCornerRowLUT.data()));
    L.nFastThresh = mcp_kf_fast_thresh(mpDev, l);
    L.vFastFrequency.resize(MCP_MAX_FAST_THRESH + 1);
    check(mcp_kf_get_fast_frequency(mpDev, l, L.vFastFrequency.data()));
    const int nc = std::max(0, mcp_kf_num_candidates(mpDev, l));
    L.vCandidates.resize(nc); L.vCandidateScores.resize(nc);
    if (nc) {
      const int got = mcp_kf_get_candidates(mpDev, l, L.vCandidates.data(), L.vCandidateScores.data(), nc);
      L.vCandidates.resize(got); L.vCandidateScores.resize(got);
    }
    return L;
  }

  // ---- SmallBlurryImage (KeyFrame::MakeSBI, src/KeyFrame.cc:539-545; src/SmallBlurryImage.cc)
  void MakeSBI(double dBlur = 2.5) { check(mcp_kf_make_sbi(mpDev, dBlur)); }
  /// SmallBlurryImage::IteratePosRelToTarget: (SE2, final sum of squares)
  std::pair<std::array<double, 6>, double> IteratePosRelToTarget(KeyFrame& other, int nIterations = 10) {
    std::array<double, 6> se2{}; double score = 0;
    check(mcp_sbi_iterate(mpDev, other.mpDev, nIterations, se2.data(), &score));
    return { se2, score };
  }
  /// Tracker::CalcSBIRotation's per-camera step: this frame's SBI against the one it replaced
  std::pair<std::array<double, 6>, double> IteratePosRelToLast(int nIterations = 6) {
    std::array<double, 6> se2{}; double score = 0;
    check(mcp_sbi_iterate_last(mpDev, nIterations, se2.data(), &score));
    return { se2, score };
  }
  /// SmallBlurryImage::SE3fromSE2 (cameras already at the 40x30 size): rotation, row-major
  static std::array<double, 9> SE3fromSE2(const std::array<double, 6>& se2, const mcp_camera& camSrc, const mcp_camera& camTarget) {
    std::array<double, 9> R{};
    if (mcp_sbi_se3_from_se2(se2.data(), &camSrc, &camTarget, R.data()) != 0) throw std::runtime_error(mcp_last_error());
    return R;
  }
  /// Relocaliser::ScoreKFs: (index of the best candidate or -1, all scores)
  std::pair<int, std::vector<double> > ScoreKFs(const std::vector<KeyFrame*>& vCandidates) {
    std::vector<mcp_kf*> h; h.reserve(vCandidates.size());
    for (KeyFrame* k : vCandidates) h.push_back(k ? k->mpDev : nullptr);
    std::vector<double> scores(vCandidates.size() + 1); int best = -1;
    check(mcp_sbi_score(mpDev, (int)vCandidates.size(), h.data(), scores.data(), &best));
    scores.resize(vCandidates.size());
    return { best, scores };
  }

  // ---- MiniPatch::SampleFromImage + FindPatch, batched (src/MiniPatch.cc:34-122): patches of *this at vSrc searched in `target`
  struct PatchMatch { mcp_int2 pos; bool found; int ssd; };
  std::vector<PatchMatch> FindPatches(KeyFrame& target, int nLevel, const std::vector<mcp_int2>& vSrc, const std::vector<mcp_int2>& vStart, int nRange) {
    if (vSrc.size() != vStart.size()) throw std::invalid_argument("FindPatches: one start position per patch");
    const int n = (int)vSrc.size();
    std::vector<mcp_int2> pos(n + 1); std::vector<uint8_t> found(n + 1); std::vector<int> ssd(n + 1);
    check(mcp_minipatch_find(mpDev, target.mpDev, nLevel, n, vSrc.data(), vStart.data(), nRange, pos.data(), found.data(), ssd.data()));
    std::vector<PatchMatch> out(n);
    for (int i = 0; i < n; ++i) out[i] = { pos[i], found[i] != 0, ssd[i] };
    return out;
  }

  // ---- the per-point part of Tracker::SearchForPoints (src/Tracker.cc:1299-1377) against this (the current) frame
  std::vector<mcp_td_out> SearchForPoints(const mcp_camera& cam, const double base_from_world[12], const double cam_from_base[12],
                                          const std::vector<mcp_td_in>& vTD, int nRange, int nSubPixIts, bool bExhaustive = false) {
    std::vector<mcp_td_out> out(vTD.size() + 1);
    check(mcp_track_search(mpDev, &cam, base_from_world, cam_from_base, (int)vTD.size(), vTD.data(), nRange, nSubPixIts, bExhaustive, out.data()));
    out.resize(vTD.size());
    return out;
  }

  mcp_kf* handle() { return mpDev; }

  // ---- the cameras of a frame in one submission (the per-camera loops of Tracker::TrackFrame, src/Tracker.cc:303-318, and of
  // Tracker::TrackMap, :985-1030): results equal the per-camera calls bit for bit
  /// MakeKeyFrame_Lite on every KeyFrame of `kfs` (<= MCP_MAX_FRAME_CAMS, one device); on_device: `ims` are device pointers
  static void MakeKeyFrameLiteBatch(const std::vector<KeyFrame*>& kfs, const std::vector<const uint8_t*>& ims, const std::vector<int>& strides,
                                    bool on_device = false, const std::vector<const uint8_t* const*>* masks = nullptr) {
    std::vector<mcp_kf*> h; for (KeyFrame* k : kfs) h.push_back(k->mpDev);
    check(mcp_kf_make_lite_batch((int)h.size(), h.data(), ims.data(), strides.data(), on_device ? 1 : 0, masks ? masks->data() : nullptr));
  }
  /// SearchForPoints of every camera against its KeyFrame in one launch; cams_from_base: 12 doubles per camera
  static std::vector<std::vector<mcp_td_out>> SearchForPointsBatch(const std::vector<KeyFrame*>& kfs, const std::vector<mcp_camera>& cams,
                                                                    const double base_from_world[12], const std::vector<double>& cams_from_base,
                                                                    const std::vector<std::vector<mcp_td_in>>& vTD, int nRange, int nSubPixIts,
                                                                    bool bExhaustive = false) {
    const int nc = (int)kfs.size();
    std::vector<mcp_kf*> h; std::vector<int> n; std::vector<const mcp_td_in*> in; std::vector<mcp_td_out*> op;
    std::vector<std::vector<mcp_td_out>> out(nc);
    for (int c = 0; c < nc; ++c) { h.push_back(kfs[c]->mpDev); n.push_back((int)vTD[c].size()); in.push_back(vTD[c].data()); out[c].resize(vTD[c].size() + 1); op.push_back(out[c].data()); }
    check(mcp_track_search_batch(nc, h.data(), cams.data(), base_from_world, cams_from_base.data(), n.data(), in.data(), nRange, nSubPixIts, bExhaustive, op.data()));
    for (int c = 0; c < nc; ++c) out[c].resize(vTD[c].size());
    return out;
  }

 private:
  static void check(int rc) { if (rc < 0) throw std::runtime_error(mcp_last_error()); }
  mcp_kf* mpDev = nullptr;
};

/// Tracker::CalcPoseUpdate (src/Tracker.cc:1386-1512): mu, Tukey sigma^2; weights (0 = outlier) if asked for
inline std::pair<std::array<double, 6>, double> CalcPoseUpdate(const std::vector<uint8_t>& vFound, const std::vector<double>& vFoundPos /*2n*/,
                                                               const std::vector<double>& vImagePos /*2n*/, const std::vector<double>& vSqrtInvNoise /*n*/,
                                                               const std::vector<double>& vJacobian /*12n*/, double dOverrideSigma = -1.0,
                                                               std::vector<double>* pvWeights = nullptr) {
  const int n = (int)vFound.size();
  if ((int)vFoundPos.size() != 2*n || (int)vImagePos.size() != 2*n || (int)vSqrtInvNoise.size() != n || (int)vJacobian.size() != 12*n)
    throw std::invalid_argument("CalcPoseUpdate: array sizes");
  std::array<double, 6> mu{}; double sigma = 0;
  if (pvWeights) pvWeights->assign(n, 0.0);
  if (mcp_track_pose_update(n, vFound.data(), vFoundPos.data(), vImagePos.data(), vSqrtInvNoise.data(), vJacobian.data(), dOverrideSigma,
                            mu.data(), pvWeights ? pvWeights->data() : nullptr, &sigma) < 0) throw std::runtime_error(mcp_last_error());
  return { mu, sigma };
}

/// The ten pose iterations of Tracker::TrackMap in one launch (src/Tracker.cc:775-838, 1038-1075); base_from_world is updated
inline std::array<double, 6> TrackMapPoseIterations(std::vector<mcp_pose_point>& vPoints, const std::vector<mcp_camera>& vCams,
                                                    const std::vector<double>& vCamFromBase /*12 per camera*/, double base_from_world[12],
                                                    const std::vector<uint8_t>& vNonlinear, const std::vector<double>& vOverrideSigma,
                                                    std::vector<double>* pvWeightsLast = nullptr) {
  if (vNonlinear.size() != vOverrideSigma.size() || vCamFromBase.size() != 12*vCams.size()) throw std::invalid_argument("TrackMapPoseIterations: array sizes");
  std::array<double, 6> mu{};
  if (pvWeightsLast) pvWeightsLast->assign(vPoints.size(), 0.0);
  if (mcp_track_pose_refine((int)vPoints.size(), vPoints.data(), (int)vCams.size(), vCams.data(), vCamFromBase.data(), base_from_world,
                            (int)vNonlinear.size(), vNonlinear.data(), vOverrideSigma.data(), mu.data(),
                            pvWeightsLast ? pvWeightsLast->data() : nullptr) < 0) throw std::runtime_error(mcp_last_error());
  return mu;
}

}  // namespace mcptam_hip
