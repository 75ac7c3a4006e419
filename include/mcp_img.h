/*
 * mcp_img.h -- C ABI of the MI355X (gfx950) KeyFrame / Tracker image path.
 *
 * Second drop-in boundary of the MCPTAM back end (SURVEY.md 8(b) "Image/track seam"): the
 * inner loops of KeyFrame::MakeKeyFrame_Lite / MakeKeyFrame_Rest
 * (/root/reference/src/KeyFrame.cc:145-360, 363-450), ShiTomasi.cc, MiniPatch.cc, PatchFinder.cc
 * and the per-point part of Tracker::SearchForPoints / CalcPoseUpdate
 * (src/Tracker.cc:1299-1377, 1386-1512; include/mcptam/TrackerData.h:102-185).
 * Control flow (which points to search, shuffles, budgets) stays in the reference's Tracker /
 * MapMaker; they hand BATCHES to these entry points.
 *
 * A keyframe handle owns the 4-level pyramid, the FAST corner lists and the row look-up
 * tables ON THE DEVICE: source keyframes of map points stay resident so that template warps
 * (PatchFinder::MakeTemplateCoarseCont) gather from HBM, not over PCIe.
 * Integer results (pyramids, corner lists and order, LUTs, thresholds, ZMSSD / SSD scores,
 * coarse positions) are bit-exact against the CPU oracle; floating-point results follow the
 * reference's float/double mix.  No CPU fallback: every entry point fails (-1 / NULL, see
 * mcp_last_error()) when no gfx950 device is usable.
 */
#ifndef MCP_IMG_H
#define MCP_IMG_H

#include <stdint.h>
#include "mcp_ba.h"      /* mcp_camera, mcp_last_error */

#ifdef __cplusplus
extern "C" {
#endif

#define MCP_LEVELS 4            /* LEVELS, include/mcptam/KeyFrame.h:85 */
#define MCP_MIN_FAST_THRESH 5   /* KeyFrame.h:88 */
#define MCP_MAX_FAST_THRESH 30  /* KeyFrame.h:89 */

typedef struct mcp_kf mcp_kf;

typedef struct mcp_int2 { int x, y; } mcp_int2;

/* options of MakeKeyFrame_Lite that are statics / GVars in the reference */
typedef struct mcp_kf_params {
  int adaptive_thresh;        /* KeyFrame::sbAdaptiveThresh (default 1)                        */
  int glare_masking;          /* GVar GlareMasking (default 0, src/System.cc:121)              */
  int half_sample_pavgb;      /* 0: truncating 2x2 mean (libCVD generic halfSample);
                                 1: cascaded round-half-up averages (libCVD SSE2 byte path)    */
  int device;                 /* HIP device ordinal, -1 = current                              */
} mcp_kf_params;

/* KeyFrame + its Levels (include/mcptam/KeyFrame.h:93-150).  w,h = level-0 size. */
mcp_kf* mcp_kf_create(int w, int h, const mcp_kf_params* params);
void    mcp_kf_destroy(mcp_kf*);

/* KeyFrame::MakeKeyFrame_Lite(CVD::Image<byte>& im, bool, bool bGlareMasking)  KeyFrame.cc:145-360
 * img: level-0 image, row stride in bytes.  masks: NULL, or MCP_LEVELS pointers (each NULL or a
 * tightly packed mask of that level's size; a corner is kept only where mask == 255, :305). */
int mcp_kf_make_lite(mcp_kf*, const uint8_t* img, int stride, const uint8_t* const* masks);

/* The MakeKeyFrame_Lite loop over the cameras of a frame (Tracker::TrackFrame, src/Tracker.cc:303-318) as ONE submission:
 * uploads + three kernel launches for all levels of all cameras + one wait.  kfs: ncam distinct handles on one device that
 * share adaptive_thresh / half_sample_pavgb; imgs_on_device != 0: imgs[] are device pointers (a capture ring that already
 * lives in HBM), nothing crosses PCIe.  masks: NULL, or per camera NULL / MCP_LEVELS pointers as in mcp_kf_make_lite.
 * Results are those of ncam mcp_kf_make_lite calls, bit for bit. */
#define MCP_MAX_FRAME_CAMS 8
int mcp_kf_make_lite_batch(int ncam, mcp_kf* const* kfs, const uint8_t* const* imgs, const int* strides, int imgs_on_device,
                           const uint8_t* const* const* masks);

/* read-back of what MakeKeyFrame_Lite leaves in Level (image, vCorners, vCornerRowLUT, nFastThresh,
 * vFastFrequency) */
int mcp_kf_level_size(mcp_kf*, int level, int* w, int* h);
int mcp_kf_get_image(mcp_kf*, int level, uint8_t* out /* w*h, packed */);
int mcp_kf_num_corners(mcp_kf*, int level);
int mcp_kf_get_corners(mcp_kf*, int level, mcp_int2* out, int cap);
int mcp_kf_get_row_lut(mcp_kf*, int level, int* out /* h */);
int mcp_kf_fast_thresh(mcp_kf*, int level);
int mcp_kf_get_fast_frequency(mcp_kf*, int level, double* out /* MCP_MAX_FAST_THRESH+1 */);

/* frames held in the Level::imagePrev / vCornersPrev history (0..2, KeyFrame.h:147-148): every mcp_kf_make_lite on a handle
 * that already holds a frame pushes that frame (device-resident) before overwriting it, KeyFrame.cc:152-199 */
int mcp_kf_num_prev(mcp_kf*);

/* KeyFrame::MakeKeyFrame_Rest, candidate part                         KeyFrame.cc:363-450, 456-527
 * use_shi: ssCandidateType ("shi" = 1 / "fast" = 0); use_percent: ssCandidateCriterion;
 * top_fraction = sdCandidateTopFraction (0.8); thresh = sdCandidateThresh (70).
 * nonmax_score: score used by CVD::fast_nonmax -- 0: FAST-10 binary-search score,
 * 1: the classic ring SAD corner_score (libCVD vintage dependent).
 * When the handle holds history, the candidates are pruned by the back/forward MiniPatch stability test (:456-527). */
int mcp_kf_make_rest(mcp_kf*, int use_shi, int use_percent, double top_fraction, double thresh, int nonmax_score);
int mcp_kf_num_candidates(mcp_kf*, int level);
int mcp_kf_get_candidates(mcp_kf*, int level, mcp_int2* pos, double* score, int cap);

/* MiniPatch::SampleFromImage + FindPatch, batched                      MiniPatch.cc:34-122
 * For each i: 9x9 patch of `src` level `level` at src_pos[i], searched among the FAST corners of
 * `dst` level `level` inside +-range of dst_pos[i] (row LUT used).  out_pos / out_found per i. */
int mcp_minipatch_find(mcp_kf* src, mcp_kf* dst, int level, int n, const mcp_int2* src_pos,
                       const mcp_int2* dst_pos, int range, mcp_int2* out_pos, uint8_t* out_found, int* out_ssd);

/* The Gauss-Newton pose iterations of Tracker::TrackMap in one call (src/Tracker.cc:775-838, 1038-1075).  Per iteration i:
 * nonlinear[i] != 0 -> PoseUpdateStep (found points are re-projected unless i == 0, CalcJacobian), else PoseUpdateStepLinear
 * (LinearUpdate with the previous update); then CalcPoseUpdate with override_sigma[i] (<= 0: Tukey sigma^2 from the
 * median -- the caller applies the "no override up to iteration 5" rule) and BaseFromWorld <- exp(mu) BaseFromWorld.
 * Points of all cameras go in one array; image / cam_derivs are updated in place, weights_last (may be NULL) receives the
 * Tukey weights of the last iteration (0 = outlier, as bMarkOutliers counts them). */
typedef struct mcp_pose_point {
  double world_pos[3];
  double found_pos[2];
  double sqrt_inv_noise;
  double image[2];
  double cam_derivs[4];
  int    cam;
  int    found;
} mcp_pose_point;
int mcp_track_pose_refine(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cam_from_base /* ncam x 12 */,
                          double base_from_world[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                          double mu_last[6], double* weights_last);

/* The same pose iterations with the cameras spread over ranks (one camera per GPU, BASELINE config c5 / SURVEY.md 8(e)): every rank
 * passes the points of ITS camera(s) and the same BaseFromWorld; per iteration the ranks exchange the squared errors (exact global
 * Tukey median) and the 6x6 + 6 WLS accumulator through `allreduce` (SUM of doubles in place on a device buffer, the hook type of
 * mcp_ba.h; e.g. RCCL over xGMI), so that every rank applies the identical update.  cap >= the largest n of any rank.  world = 1
 * with allreduce = NULL runs the same kernels on one device. */
int mcp_track_pose_refine_sharded(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cam_from_base /* ncam x 12 */,
                                  double base_from_world[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                                  double mu_last[6], double* weights_last, mcp_allreduce_fn allreduce, void* user, int rank, int world, int cap);

/* ---- SmallBlurryImage / Relocaliser -------------------------------- src/SmallBlurryImage.cc:67-330, src/Relocaliser.cc:61-121
 * The 40x30 thumbnail of the frame the handle holds, its zero-mean Gaussian-blurred float template and gradient image
 * (MakeFromKF + MakeJacs) live on the device with the keyframe (KeyFrame::mpSBI).  blur = 2.5 in the reference. */
#define MCP_SBI_W 40
#define MCP_SBI_H 30
int mcp_kf_make_sbi(mcp_kf*, double blur);
int mcp_kf_get_sbi(mcp_kf*, uint8_t* small_img /*1200 or NULL*/, float* templ /*1200 or NULL*/, float* jacs /*2400 (gx,gy) or NULL*/);
/* Relocaliser::ScoreKFs: ZMSSD of cur against n candidate keyframes (NULL / SBI-less entries are skipped with a score of
 * DBL_MAX); *best = index of the first smallest score or -1. */
int mcp_sbi_score(mcp_kf* cur, int n, mcp_kf* const* cands, double* scores, int* best);
/* SmallBlurryImage::IteratePosRelToTarget (ESM): se2 = { R00, R01, R10, R11, tx, ty }, *score = final sum of squares */
int mcp_sbi_iterate(mcp_kf* cur, mcp_kf* target, int iterations, double se2[6], double* score);
/* Tracker::CalcSBIRotation's per-camera step (src/Tracker.cc:1687-1720): every mcp_kf_make_sbi keeps the SBI it replaces as
 * "last frame's"; this aligns the current one to it. */
int mcp_sbi_iterate_last(mcp_kf*, int iterations, double se2[6], double* score);
/* SmallBlurryImage::SE3fromSE2; the cameras are the 40x30 instances (TaylorCamera::SetImageSize(sirSize)) */
int mcp_sbi_se3_from_se2(const double se2[6], const mcp_camera* cam_src, const mcp_camera* cam_target, double R[9]);

/* one tracked map point as seen by Tracker::SearchForPoints */
typedef struct mcp_td_in {
  double world_pos[3];          /* MapPoint::mv3WorldPos                                   */
  double pixel_right_w[3];      /* MapPoint::mv3PixelRight_W                               */
  double pixel_down_w[3];       /* MapPoint::mv3PixelDown_W                                */
  const mcp_kf* source_kf;      /* MapPoint::mpPatchSourceKF (resident pyramid)            */
  int source_level;             /* MapPoint::mnSourceLevel                                 */
  int center_x, center_y;       /* MapPoint::mirCenter                                     */
  int fixed;                    /* MapPoint::mbFixed (exhaustive search + 10 sub-pix its)  */
} mcp_td_in;

typedef struct mcp_td_out {
  double image[2];              /* TrackerData::mv2Image (projection)                      */
  double cam_derivs[4];         /* mm2CamDerivs, row-major                                 */
  double jacobian[12];          /* mm26Jacobian 2x6 row-major (w.r.t. the BASE pose)       */
  double found_pos[2];          /* mv2Found (sub-pixel or coarse, level-0 coordinates)     */
  double sqrt_inv_noise;        /* mdSqrtInvNoise = 1 / LevelScale                         */
  double warp_inverse[4];       /* PatchFinder::mm2WarpInverse                             */
  int in_image;                 /* mbInImage                                               */
  int search_level;             /* PatchFinder::mnSearchLevel, -1 = rejected warp          */
  int template_bad;             /* PatchFinder::TemplateBad()                              */
  int searched, found, did_subpix;
  int coarse_x, coarse_y;       /* best corner at search level (irBest)                    */
  int score;                    /* nBestSSD                                                */
  uint8_t templ[64];            /* mimTemplate (8x8, row-major)                            */
} mcp_td_out;

/* TrackerData::Project + GetDerivsUnsafe + CalcJacobian, PatchFinder::CalcSearchLevelAndWarpMatrix,
 * MakeTemplateCoarseCont (with a PatchFinder that has seen nothing: the template is always made; mcp_patch_sequences carries
 * the finder's template cache from call to call), FindPatchCoarse,
 * MakeSubPixTemplate + IterateSubPixToConvergence, for n points against keyframe `target`.
 * base_from_world / cam_from_base: (R row-major 9, t 3).  range, subpix_its, exhaustive as
 * Tracker::SearchForPoints(vTD, cam, nRange, nSubPixIts, bExhaustive). */
int mcp_track_search(mcp_kf* target, const mcp_camera* cam, const double base_from_world[12],
                     const double cam_from_base[12], int n, const mcp_td_in* in, int range,
                     int subpix_its, int exhaustive, mcp_td_out* out);
/* SearchForPoints for every camera of a frame in one launch (the per-camera loops of Tracker::TrackMap, src/Tracker.cc:985-1030):
 * targets[c], cams[c], cam_from_base[12*c..], n[c] points in[c] -> out[c]; results equal ncam mcp_track_search calls. */
int mcp_track_search_batch(int ncam, mcp_kf* const* targets, const mcp_camera* cams, const double base_from_world[12],
                           const double* cam_from_base /* ncam x 12 */, const int* n, const mcp_td_in* const* in,
                           int range, int subpix_its, int exhaustive, mcp_td_out* const* out);

/* ---- PatchFinder with its members carried from call to call ------------------------------------------------------------
 * The reference's PatchFinder is stateful (src/PatchFinder.cc:56-65): MakeTemplateCoarseCont keeps the template while it works on
 * the same MapPoint and neither column of the warp matrix has moved by more than 0.07 (:144-181; mbTemplateBad and the sums stay
 * too), MakeSubPixTemplate's Jacobians stay until it runs again (:362-390) and mdMeanDiff is only reset there.  What a search
 * returns therefore depends on what the finder saw before.  mcp_pf_state holds those members; the caller owns one per PatchFinder
 * object of the reference, zero-initialised (valid = 0) for a new one, and passes it in and out.
 * A SEQUENCE is what one finder sees, in order (one wavefront walks it); sequences run in parallel.  Modes = the finder's callers:
 *  MCP_PF_TRACK       Tracker::SearchForPoints (src/Tracker.cc:1299-1377).  One finder per TrackerData = per (point, camera), kept
 *                     over the frames: one sequence of one item per tracked point.  range / subpix_its / exhaustive as there.
 *                     (mcp_track_search* = this with finders that have seen nothing.)
 *  MCP_PF_REFIND      MapMakerServerBase::ReFind_Common (src/MapMakerServerBase.cc:921-1002): ONE static finder over all calls --
 *                     ReFindNewlyMade walks every keyframe with the same point, so templates are shared between keyframes with
 *                     similar warps (one sequence per new point, one item per keyframe).  MakeTemplateCoarse ignores the verdict
 *                     of CalcSearchLevelAndWarpMatrix; range 4 (pass it); sub-pixel iteration (8) only when the level is > 0, and
 *                     its position is kept whether or not it converged (found stays 1, did_subpix = 1).
 *  MCP_PF_EPI_COARSE  MapMakerServerBase::AddPointEpipolar, first loop (:745-795): one finder and ONE MapPoint object for all depth
 *                     hypotheses of a candidate (same point_key): one sequence per candidate, one item per hypothesis.  Hypotheses
 *                     that project outside the level-0 image / onto a zero of the target's level-0 mask are skipped; range 3.
 *  MCP_PF_EPI_REFINE  the second loop (:827-853) on the SAME finder (pass the state the coarse sequence returned): Calc +
 *                     MakeTemplateCoarseCont, SetSubPixPos(start_pos), IterateSubPixToConvergence(10); found = converged,
 *                     found_pos = the sub-pixel position.
 * mcp_td_out per item as in mcp_track_search (jacobian w.r.t. base_from_world; template_bad = the finder's flag after the item;
 * searched = FindPatchCoarse ran).  Targets must live on one device. */
#define MCP_PF_TRACK 0
#define MCP_PF_REFIND 1
#define MCP_PF_EPI_COARSE 2
#define MCP_PF_EPI_REFINE 3
typedef struct mcp_pf_state {
  int     valid;          /* mpLastTemplateMapPoint != NULL                                                       */
  int     point_key;      /* the map point the template was made for (caller's id; the reference compares &point) */
  double  last_warp[4];   /* mm2LastWarpMatrix, row-major                                                         */
  int     template_bad;   /* mbTemplateBad                                                                        */
  int     jacs_valid;     /* MakeSubPixTemplate has run (mimJacs / mm3HInv hold something)                        */
  double  mean_diff;      /* mdMeanDiff                                                                           */
  uint8_t templ[64];      /* mimTemplate                                                                          */
  uint8_t jac_templ[64];  /* the template mimJacs / mm3HInv were made from: differs from templ after a refresh that
                             hit pixels outside the source image (:165-180 skips MakeSubPixTemplate then)         */
} mcp_pf_state;
typedef struct mcp_pf_target {
  mcp_kf* kf;                   /* keyframe searched in (its level-0 mask, if it has one, serves MCP_PF_EPI_COARSE) */
  const mcp_camera* cam;
  double base_from_world[12];   /* (R row-major 9, t 3); map-maker callers pass the keyframe's CamFromWorld here ...   */
  double cam_from_base[12];     /* ... and the identity here                                                        */
} mcp_pf_target;
typedef struct mcp_pf_item {
  mcp_td_in point;
  int point_key;                /* identity of the MapPoint object                                                  */
  int target;                   /* index into targets[]                                                             */
  double start_pos[2];          /* MCP_PF_EPI_REFINE: SetSubPixPos (level-0 coordinates)                            */
} mcp_pf_item;
int mcp_patch_sequences(int mode, int n_targets, const mcp_pf_target* targets, int n_seq, const int* seq_start /* n_seq + 1 */,
                        const mcp_pf_item* items, mcp_pf_state* state /* n_seq, in/out */, int range, int subpix_its, int exhaustive,
                        mcp_td_out* out /* one per item */);

/* Tracker::CalcPoseUpdate (Tukey M-estimator, WLS<6> with prior 100)   Tracker.cc:1386-1512
 * found[i] != 0 rows contribute.  override_sigma <= 0: Tukey sigma^2 from the median.
 * mu[6] out; weights_out[n] (may be NULL) = Tukey weight per row (0 = outlier). */
int mcp_track_pose_update(int n, const uint8_t* found, const double* found_pos /*n*2*/,
                          const double* image_pos /*n*2*/, const double* sqrt_inv_noise /*n*/,
                          const double* jacobian /*n*12*/, double override_sigma, double mu[6],
                          double* weights_out, double* sigma_sq_out);
/* The same three entries with the M-estimator Tracker::CalcPoseUpdate dispatches on (Tracker::sMEstimatorName, src/Tracker.cc:1388-1401):
 * Tukey (the default; what the entries without _m use), Cauchy, Huber -- weights and sigma^2 of include/mcptam/MEstimator.h:84-204.
 * weights == 0 (the reference's outlier test, :1470) only ever happens with Tukey, as in the reference. */
#define MCP_MEST_TUKEY 0
#define MCP_MEST_CAUCHY 1
#define MCP_MEST_HUBER 2
int mcp_track_pose_update_m(int n, const uint8_t* found, const double* found_pos, const double* image_pos, const double* sqrt_inv_noise,
                            const double* jacobian, double override_sigma, double mu[6], double* weights_out, double* sigma_sq_out, int estimator);
int mcp_track_pose_refine_m(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cam_from_base, double base_from_world[12],
                            int n_iter, const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last, int estimator);
int mcp_track_pose_refine_sharded_m(int n, mcp_pose_point* pts, int ncam, const mcp_camera* cams, const double* cam_from_base, double base_from_world[12],
                                    int n_iter, const uint8_t* nonlinear, const double* override_sigma, double mu_last[6], double* weights_last,
                                    mcp_allreduce_fn allreduce, void* user, int rank, int world, int cap, int estimator);

/* ---- One stage of Tracker::TrackMap for a whole frame in ONE submission ------------------------------------------------------
 * What Tracker::TrackFrame / TrackMap do per frame and stage (src/Tracker.cc:303-318 MakeKeyFrame_Lite of every camera, :985-1030 +
 * :1299-1384 SearchForPoints per camera, :1040-1075 the ten CalcPoseUpdate iterations) as one call with one wait:
 *   imgs != NULL : mcp_kf_make_lite_batch(ncam, targets, imgs, strides, imgs_on_device, masks) first (NULL: the pyramids are current,
 *                  e.g. the fine stage after the coarse one);
 *   the search   : state == NULL: mcp_track_search_batch (finders that have seen nothing);  state != NULL: the reference's persistent
 *                  finders -- state[c][i] / point_key[c][i] belong to point i of camera c, as one single-item MCP_PF_TRACK sequence
 *                  each of mcp_patch_sequences (states updated in place);
 *   the records of the pose iterations are built from the search results on the device (world position from in[c][i], camera = c);
 *   mcp_track_pose_refine_m(total, ..., n_iter, nonlinear, override_sigma, ..., estimator) over the points of all cameras.
 * Same kernels on the same data as the separate calls: out[c] (n[c] results), pts_out (total records as the iterations left them,
 * may be NULL), base_from_world (in: the prior, out: refined), mu_last, weights_last (total, camera-major; may be NULL) are bit-identical
 * to theirs.  n_iter = 0: search only.  It saves the two host round trips between the three calls, and with a fresh frame (imgs != NULL) every
 * copy-engine operation: the small inputs ride to the device inside the pyramids' second launch, the results are written to pinned host memory
 * by the kernels that produce them and copied into the caller's arrays after the one wait (five kernels back to back on the device). */
int mcp_track_frame(int ncam, mcp_kf* const* targets, const uint8_t* const* imgs, const int* strides, int imgs_on_device,
                    const uint8_t* const* const* masks, const mcp_camera* cams, double base_from_world[12], const double* cam_from_base /* ncam x 12 */,
                    const int* n, const mcp_td_in* const* in, const int* const* point_key, mcp_pf_state* const* state,
                    int range, int subpix_its, int exhaustive, int n_iter, const uint8_t* nonlinear, const double* override_sigma, int estimator,
                    mcp_td_out* const* out, mcp_pose_point* pts_out, double mu_last[6], double* weights_last);
/* ZERO-COPY RESULTS (round 6).  The search kernel writes the TrackerData results into a pinned block of the library; with caller arrays
 * (`out` != NULL) mcp_track_frame copies them out after its one wait -- 300 bytes per point, ~25 us of the host's time per 640x480 x 4
 * frame.  A native caller that only walks the results once (Tracker::TrackMap updating its TrackerData, src/Tracker.cc:1040-1075) passes
 * out = NULL and reads them in place:
 *     const mcp_td_out* r = mcp_track_frame_view(targets[0], c, &count);      // count == n[c]
 * valid until the next mcp_track_frame / mcp_track_search_batch call whose first target is targets[0]; NULL (count 0) for a camera
 * without points, NULL + mcp_last_error() for a camera index the last frame did not have. */
const mcp_td_out* mcp_track_frame_view(const mcp_kf* first_target, int cam, int* count);

#ifdef __cplusplus
}
#endif
#endif
