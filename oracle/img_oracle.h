/*
 * img_oracle.h -- CPU ORACLE for the KeyFrame / Tracker image path.
 *
 * TEST INFRASTRUCTURE ONLY (see ba_oracle.h): never linked into or called by the product
 * library.  Plain-C restatement of
 *   /root/reference/src/KeyFrame.cc:145-450, ShiTomasi.cc:34-63, MiniPatch.cc:34-122,
 *   PatchFinder.cc:69-472,511-664, include/mcptam/TrackerData.h:102-185,
 *   Tracker.cc:1386-1512, include/mcptam/LevelHelpers.h:55-98
 * and of the libCVD / OpenCV / TooN primitives they call (halfSample, fast_corner_detect_10,
 * fast_corner_score_10, fast_nonmax, transform, dilate, WLS<6>), restated from their published
 * algorithms [3P-memory] as listed in SURVEY.md Appendix A.6/A.7.
 *
 * PARITY UNPINNED: the reference has no tests or fixtures for this path and libCVD/OpenCV/
 * TooN are absent, so rounding conventions of halfSample / transform / fast_nonmax are
 * restated from memory; each is a switch or is documented where it is restated.
 */
#ifndef MCPTAM_IMG_ORACLE_H
#define MCPTAM_IMG_ORACLE_H
#include <stdint.h>
#include "ba_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LEVELS 4

typedef struct orc_kf orc_kf;
typedef struct orc_int2 { int x, y; } orc_int2;

orc_kf* orc_kf_create(int w, int h, int adaptive, int glare, int pavgb);
void    orc_kf_destroy(orc_kf*);
int     orc_kf_make_lite(orc_kf*, const uint8_t* img, int stride, const uint8_t* const* masks);
int     orc_kf_num_prev(orc_kf*);      /* frames held in Level::imagePrev (0..2) */
int     orc_kf_level_size(orc_kf*, int level, int* w, int* h);
const uint8_t* orc_kf_image(orc_kf*, int level);
int     orc_kf_num_corners(orc_kf*, int level);
const orc_int2* orc_kf_corners(orc_kf*, int level);
const int* orc_kf_row_lut(orc_kf*, int level);
int     orc_kf_fast_thresh(orc_kf*, int level);
const double* orc_kf_fast_frequency(orc_kf*, int level);
int     orc_kf_make_rest(orc_kf*, int use_shi, int use_percent, double top_fraction, double thresh, int nonmax_score);
int     orc_kf_num_candidates(orc_kf*, int level);
int     orc_kf_get_candidates(orc_kf*, int level, orc_int2* pos, double* score, int cap);

/* oracle-only [3P-memory] switch (scripts/oracle_sensitivity.py): key 0 = CVD::transform's float -> byte conversion, 0 truncating
 * (default), 1 rounding half up */
void orc_img_set_variant(int key, int value);

/* primitives exposed for unit checks */
int    orc_fast10_is_corner(const uint8_t* p, int stride, int b);
int    orc_fast10_score(const uint8_t* p, int stride, int bstart);
int    orc_fast_ring_sad_score(const uint8_t* p, int stride, int barrier);
double orc_shi_tomasi(const uint8_t* img, int stride, int half, int x, int y);

int orc_minipatch_find(orc_kf* src, orc_kf* dst, int level, int n, const orc_int2* src_pos,
                       const orc_int2* dst_pos, int range, orc_int2* out_pos, uint8_t* out_found, int* out_ssd);

typedef struct orc_td_in {
  double world_pos[3], pixel_right_w[3], pixel_down_w[3];
  const orc_kf* source_kf;
  int source_level, center_x, center_y, fixed;
} orc_td_in;
typedef struct orc_td_out {
  double image[2], cam_derivs[4], jacobian[12], found_pos[2], sqrt_inv_noise, warp_inverse[4];
  int in_image, search_level, template_bad, searched, found, did_subpix, coarse_x, coarse_y, score;
  uint8_t templ[64];
} orc_td_out;

int orc_track_search(orc_kf* target, const orc_camera* cam, const double base_from_world[12],
                     const double cam_from_base[12], int n, const orc_td_in* in, int range,
                     int subpix_its, int exhaustive, orc_td_out* out);
/* PatchFinder with its members carried from call to call: the template cache and the sub-pixel state (see img_oracle.c) */
typedef struct orc_pf_state {
  int valid, point_key; double last_warp[4]; int template_bad, jacs_valid; double mean_diff; uint8_t templ[64], jac_templ[64];
} orc_pf_state;
typedef struct orc_pf_target { orc_kf* kf; const struct orc_camera* cam; double base_from_world[12], cam_from_base[12]; } orc_pf_target;
typedef struct orc_pf_item { orc_td_in point; int point_key, target; double start_pos[2]; } orc_pf_item;
enum { ORC_PF_TRACK = 0, ORC_PF_REFIND = 1, ORC_PF_EPI_COARSE = 2, ORC_PF_EPI_REFINE = 3 };
int orc_patch_sequences(int mode, int n_targets, const orc_pf_target* targets, int n_seq, const int* seq_start, const orc_pf_item* items,
                        orc_pf_state* state, int range, int subpix_its, int exhaustive, orc_td_out* out);
int orc_track_pose_update(int n, const uint8_t* found, const double* found_pos, const double* image_pos,
                          const double* sqrt_inv_noise, const double* jacobian, double override_sigma,
                          double mu[6], double* weights_out, double* sigma_sq_out);

int orc_track_pose_update_m(int n, const uint8_t* found, const double* found_pos, const double* image_pos,
                            const double* sqrt_inv_noise, const double* jacobian, double override_sigma,
                            double mu[6], double* weights_out, double* sigma_sq_out, int estimator /* 0 Tukey, 1 Cauchy, 2 Huber */);

/* The ten Gauss-Newton pose iterations of Tracker::TrackMap (src/Tracker.cc:775-838, 1038-1075): per iteration either
 * PoseUpdateStep (re-project the found points unless it is iteration 0, CalcJacobian, CalcPoseUpdate, BaseFromWorld <-
 * exp(mu) BaseFromWorld) or PoseUpdateStepLinear (LinearUpdate with the previous mu instead of re-projection). */
typedef struct orc_pose_point {
  double world_pos[3];      /* MapPoint::mv3WorldPos */
  double found_pos[2];      /* TrackerData::mv2Found */
  double sqrt_inv_noise;    /* mdSqrtInvNoise */
  double image[2];          /* mv2Image: in = projection from the search stage, out = after the last iteration */
  double cam_derivs[4];     /* mm2CamDerivs, same */
  int cam;                  /* index into cams / cam_from_base */
  int found;                /* mbFound */
} orc_pose_point;
int orc_track_pose_refine(int n, orc_pose_point* pts, int ncam, const struct orc_camera* cams, const double* cam_from_base /* ncam x 12 */,
                          double base_from_world[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                          double mu_last[6], double* weights_last);

int orc_track_pose_refine_m(int n, orc_pose_point* pts, int ncam, const struct orc_camera* cams, const double* cam_from_base,
                            double base_from_world[12], int n_iter, const uint8_t* nonlinear, const double* override_sigma,
                            double mu_last[6], double* weights_last, int estimator);

/* ---- SmallBlurryImage / Relocaliser (src/SmallBlurryImage.cc:67-330, src/Relocaliser.cc:61-121) ----------------
 * 40x30 thumbnail of level 0 (cv::resize INTER_LINEAR [3P-memory]), zero-mean float template blurred with
 * CVD::convolveGaussian [3P-memory], gradient image, ZMSSD, ESM SE2 alignment, SE2 -> camera rotation. */
#define ORC_SBI_W 40
#define ORC_SBI_H 30
int orc_kf_make_sbi(orc_kf*, double blur);             /* MakeFromKF + MakeJacs */
const uint8_t* orc_kf_sbi_small(orc_kf*);              /* mimSmall, 1200 bytes */
const float*   orc_kf_sbi_template(orc_kf*);           /* mimTemplate, 1200 floats */
const float*   orc_kf_sbi_jacs(orc_kf*);               /* mimImageJacs, 1200 x (gx, gy) */
double orc_sbi_zmssd(orc_kf* a, orc_kf* b);
/* Relocaliser::ScoreKFs: best (first smallest) ZMSSD among n candidates; scores[n]; returns best index or -1 */
int orc_sbi_score(orc_kf* cur, int n, orc_kf* const* cands, double* scores);
/* IteratePosRelToTarget: se2 = { R00, R01, R10, R11, tx, ty } */
int orc_sbi_iterate(orc_kf* cur, orc_kf* target, int iterations, double se2[6], double* score);
/* SE3fromSE2 (cameras already at SBI size): rotation R (row-major) */
struct orc_camera;
void orc_sbi_se3_from_se2(const double se2[6], const struct orc_camera* cam_src, const struct orc_camera* cam_target, double R[9]);

#ifdef __cplusplus
}
#endif
#endif
