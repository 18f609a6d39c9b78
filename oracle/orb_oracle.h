/*
 * orb_oracle.h -- CPU ORACLE for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's CPU algorithm (saber/ORB_SLAM2_Annotate,
 * src/ORBextractor.cc, src/ORBmatcher.cc, src/Frame.cc:512-686) used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER.
 * Nothing under orb_slam2_annotate_amd/ may include, link or call this.
 *
 * PARITY UNPINNED: the reference ships no tests, no golden vectors and cannot be
 * built here (OpenCV/Eigen absent, version unpinned; see DESIGN.md "Oracle").
 * The pixel primitives that live in OpenCV (resize, FAST, GaussianBlur,
 * fastAtan2, cvRound) are restated from their published algorithm as the
 * "canonical spec" frozen in SURVEY.md section 8(c); cos/sin is this project's
 * own correctly-rounded-double routine (documented deviation from glibc cosf).
 * What IS pinned: every constant derivable from the reference source
 * (pattern table, umax, quotas, scale tables, thresholds) -- see
 * tests/test_oracle_known_answers.py.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* Layout-compatible with cv::KeyPoint (28 bytes). */
typedef struct orc_keypoint {
  float x, y;     /* pt */
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} orc_keypoint;

/* include/ORBextractor.h:52-114 -- constructor-derived constant tables. */
typedef struct orc_extractor {
  int nfeatures;
  double scaleFactor; /* (double)(float)1.2 -- the member is double, include/ORBextractor.h:100 */
  int nlevels, iniThFAST, minThFAST;
  float mvScaleFactor[ORC_MAX_LEVELS];
  float mvInvScaleFactor[ORC_MAX_LEVELS];
  float mvLevelSigma2[ORC_MAX_LEVELS];
  float mvInvLevelSigma2[ORC_MAX_LEVELS];
  int mnFeaturesPerLevel[ORC_MAX_LEVELS];
  int umax[16];
  /* per-stage wall-clock accumulators (seconds), filled by orc_extract */
  double t_pyramid, t_fast, t_octree, t_orient, t_blur, t_desc;
  int blur_spec; /* 0 (default) / 1 / 2: GaussianBlur arithmetic variant, see orc_gaussian_blur7_spec */
} orc_extractor;

/* ---- arithmetic primitives (canonical spec) ---- */
int orc_cvround(double v);                          /* round half to even */
float orc_fast_atan2(float y, float x);             /* degrees in [0,360) */
void orc_sincos(float rad, float *c, float *s);     /* shared-spec sincos */
const signed char *orc_pattern(void);               /* 1024 int8 */

/* ---- extractor ---- */
void orc_extractor_init(orc_extractor *e, int nfeatures, float scaleFactor, int nlevels,
                        int iniThFAST, int minThFAST);        /* src/ORBextractor.cc:415-486 */
void orc_level_size(const orc_extractor *e, int W, int H, int level, int *w, int *h);
/* cv::resize(INTER_LINEAR) 8UC1, src/ORBextractor.cc:1219 */
void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                       int dh, int dstride);
/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), src/ORBextractor.cc:1175 */
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
void orc_gaussian_blur7_spec(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int spec);
/* cv::FAST(sub-image, thr, nonmax=true) -> keypoints relative to the sub-image.
 * Returns count; writes up to cap (x,y,score) triples. */
int orc_fast_nms(const uint8_t *img, int w, int h, int stride, int threshold, int *xs, int *ys,
                 int *scores, int cap);
int orc_fast_nms_bruteforce(const uint8_t *img, int w, int h, int stride, int threshold, int *xs,
                            int *ys, int *scores, int cap);
/* FAST corner score S-1 for one pixel (>= threshold iff corner at threshold). */
int orc_fast_score_pixel(const uint8_t *p, int stride, int threshold);
/* Grid stage only (src/ORBextractor.cc:815-896): candidates in emission order,
 * coordinates relative to (minBorderX,minBorderY). Returns count. */
int orc_grid_candidates(const orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                        float *xs, float *ys, float *resp, int cap);
/* DistributeOctTree (src/ORBextractor.cc:566-808). In/out arrays of (x,y,response);
 * returns number selected; out_idx receives indices into the input in list order. */
int orc_distribute_octtree(const float *xs, const float *ys, const float *resp, int n, int minX,
                           int maxX, int minY, int maxY, int N, int *out_idx, int cap);
float orc_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax);
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg,
                    uint8_t desc[32]);

/* Whole ORBextractor::operator() (src/ORBextractor.cc:1119-1197).
 * pyr_out (optional): receives nlevels contiguous images, level l of w_l*h_l bytes
 * (stride w_l), back to back (= mvImagePyramid contents).
 * Returns 0 on success, -1 if capacity too small. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride, orc_keypoint *kps,
                uint8_t *desc, int capacity, int *n_out, uint8_t *pyr_out);

/* ---- matcher (src/ORBmatcher.cc) ---- */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b); /* :1828-1844 */
/* number of orc_descriptor_distance calls of the calling thread since the last reset (measurement aid) */
void orc_distance_calls_reset(void);
int64_t orc_distance_calls(void);
void orc_stereo_counters(int64_t *out); /* [bucket entries scanned, SAD refinements] of this thread's last stereo call */
void orc_three_maxima(const int *histo_sizes, int L, int *ind1, int *ind2, int *ind3); /* :1777-1821 */

/* A DBoW2::FeatureVector flattened: node_ids ascending, CSR offsets into indices. */
typedef struct orc_featvec {
  int n_nodes;
  const uint32_t *node_ids;
  const int32_t *offsets; /* n_nodes+1 */
  const uint32_t *indices;
} orc_featvec;

/* SearchByBoW(KeyFrame*,Frame&,...) :185-325.  has_mp1[i]!=0 <=> KF feature i has a
 * good MapPoint.  match_f[j] = KF feature index matched to frame feature j, or -1. */
int orc_search_by_bow(const uint8_t *desc1, const uint8_t *has_mp1, const float *ang1, int n1,
                      const orc_featvec *fv1, const uint8_t *desc2, const float *ang2, int n2,
                      const orc_featvec *fv2, float nnratio, int check_ori, int32_t *match_f);
/* SearchByBoW(KeyFrame*,KeyFrame*,...) :610-743.  match12[i] = KF2 index or -1. */
int orc_search_by_bow_kf(const uint8_t *desc1, const uint8_t *has_mp1, const float *ang1, int n1,
                         const orc_featvec *fv1, const uint8_t *desc2, const uint8_t *has_mp2,
                         const float *ang2, int n2, const orc_featvec *fv2, float nnratio,
                         int check_ori, int32_t *match12);
/* SearchForTriangulation :754-928.  kp = (x,y,angle,octave) SoA; F12 row-major 3x3;
 * stereo flags = mvuRight>=0.  match12[i] = KF2 index or -1 (pairs ascending in i). */
int orc_search_for_triangulation(const uint8_t *desc1, const uint8_t *has_mp1, const float *x1,
                                 const float *y1, const float *ang1, const uint8_t *stereo1, int n1,
                                 const orc_featvec *fv1, const uint8_t *desc2,
                                 const uint8_t *has_mp2, const float *x2, const float *y2,
                                 const float *ang2, const int32_t *oct2, const uint8_t *stereo2,
                                 int n2, const orc_featvec *fv2, const float *F12, float ex,
                                 float ey, const float *scale_factors2, const float *level_sigma2_2,
                                 int only_stereo, int check_ori, int32_t *match12);

/* Frame::ComputeStereoMatches, src/Frame.cc:512-686.  pyrL/pyrR: level images
 * packed as orc_extract's pyr_out.  Outputs uRight[N], depth[N] (-1 = none). */
int orc_compute_stereo_matches(const orc_extractor *e, int W, int H, const orc_keypoint *kpL,
                               const uint8_t *descL, int N, const orc_keypoint *kpR,
                               const uint8_t *descR, int Nr, const uint8_t *pyrL,
                               const uint8_t *pyrR, float mbf, float mb, float *uRight,
                               float *depth);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:269-333): desc = the n observation
 * descriptors of one map point; returns the index of the descriptor with the least median
 * distance to the others (median = sorted row [(size_t)(0.5*(n-1))], first minimum wins). */
int orc_distinctive_descriptor(const uint8_t *desc, int n);
/* cv::cvtColor(..., CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) on 8U (src/Tracking.cc:
 * 176-262): (R*4899 + G*9617 + B*1868 + 2^13) >> 14 (OpenCV fixed-point coefficients, yuv_shift 14). */
void orc_cvt_gray(const uint8_t *src, int w, int h, int sstride, int channels, int rgb_order,
                  uint8_t *dst, int dstride);

/* ---- Frame grid + projection searches ---- */
#define ORC_GRID_COLS 64 /* include/Frame.h:44-45 */
#define ORC_GRID_ROWS 48
typedef struct orc_frame {
  int N;
  const float *x, *y;      /* mvKeysUn[i].pt */
  const int32_t *octave;   /* mvKeysUn[i].octave */
  const float *angle;      /* mvKeysUn[i].angle */
  const float *uRight;     /* mvuRight (may be NULL: all -1) */
  const uint8_t *desc;     /* mDescriptors */
  float mnMinX, mnMaxX, mnMinY, mnMaxY; /* image bounds (src/Frame.cc:697-728) */
  /* built by orc_frame_build_grid: */
  int32_t *cell_off;       /* [64*48+1], cell id = ix*48 + iy (mGrid[ix][iy]) */
  int32_t *cell_idx;
} orc_frame;
void orc_frame_build_grid(orc_frame *f);  /* AssignFeaturesToGrid + PosInGrid, src/Frame.cc:246-267,417-427 */
void orc_frame_free_grid(orc_frame *f);
/* GetFeaturesInArea, src/Frame.cc:358-415: returns count, writes up to cap indices in scan order */
int orc_features_in_area(const orc_frame *f, float x, float y, float r, int minLevel, int maxLevel, int32_t *out, int cap);

/* SearchByProjection(Frame &F, const vector<MapPoint*>&, th), src/ORBmatcher.cc:51-138.  Per map point:
 * in_view (mbTrackInView && !isBad), level (mnTrackScaleLevel), view_cos, proj_x/y/xr, desc.
 * blocked[idx] != 0 <=> F.mvpMapPoints[idx] exists with Observations()>0 before the call;
 * mp_obs_positive[iMP] = pMP->Observations()>0 (NULL: all positive, the local-map case).
 * match[idx] = map point index or -1.  Returns nmatches. */
int orc_search_by_projection_mappoints(orc_frame *F, const float *scale_factors, const uint8_t *blocked,
                                       int nMP, const uint8_t *in_view, const int32_t *level, const float *view_cos,
                                       const float *proj_x, const float *proj_y, const float *proj_xr,
                                       const uint8_t *mp_desc, const uint8_t *mp_obs_positive, float th, float nnratio,
                                       int32_t *match);

/* SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), src/ORBmatcher.cc:1484-1633,
 * after the caller's projection: per last-frame point valid (has map point, not outlier, invzc >= 0, (u,v)
 * inside the image), u, v, invzc, last octave/angle, map point descriptor, obs_positive (Observations()>0).
 * mode: 0 normal, 1 forward, 2 backward.  match_cur[i2] = last index or -1.  Returns nmatches. */
int orc_search_by_projection_lastframe(orc_frame *Cur, const float *scale_factors, float mbf, int nLast,
                                       const uint8_t *valid, const float *u, const float *v, const float *invzc,
                                       const int32_t *last_octave, const float *last_angle, const uint8_t *mp_desc,
                                       const uint8_t *obs_positive, const uint8_t *blocked_at_entry, int mode, float th,
                                       int check_ori, int32_t *match_cur);

/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist),
 * src/ORBmatcher.cc:1641-1775 (relocalisation), after the caller's projection: valid = map point exists, not bad,
 * not already found, in image, inside its distance range; level = PredictScale; kf_angle = pKF->mvKeysUn[i].angle;
 * blocked[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL before the call. */
int orc_search_by_projection_reloc(orc_frame *Cur, const float *scale_factors, int n, const uint8_t *valid,
                                   const float *u, const float *v, const int32_t *level, const float *kf_angle,
                                   const uint8_t *mp_desc, const uint8_t *blocked, float th, int orb_dist,
                                   int check_ori, int32_t *match_cur);

/* SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:335-449 (loop
 * closing), after the caller's projection.  matched[idx] != 0 <=> vpMatched[idx] != NULL before the call.
 * match[idx] = index of the point newly matched to keypoint idx, or -1. */
int orc_search_by_projection_sim3(orc_frame *KF, const float *scale_factors, int n, const uint8_t *valid,
                                  const float *u, const float *v, const int32_t *level, const uint8_t *mp_desc,
                                  const uint8_t *matched, float th, int32_t *match);

/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:469-603.
 * prev_x / prev_y are vbPrevMatched (updated in place as the reference does).  match12[i1] = i2 or -1. */
int orc_search_for_initialization(const orc_frame *F1, orc_frame *F2, float *prev_x, float *prev_y, int window,
                                  float nnratio, int check_ori, int32_t *match12);

/* The search both Fuse overloads run per map point (src/ORBmatcher.cc:940-1110 with the chi-square gate,
 * :1112-1249 without): best keypoint of the window with octave in [level-1, level]; best_idx[i] = keypoint with
 * bestDist <= TH_LOW or -1.  ur: projected right coordinate (read only when chi2 != 0 and KF->uRight). */
void orc_fuse_search(orc_frame *KF, const float *scale_factors, const float *inv_level_sigma2, int n,
                     const uint8_t *valid, const float *u, const float *v, const float *ur, const int32_t *level,
                     const uint8_t *mp_desc, float th, int chi2, int32_t *best_idx);

/* SearchBySim3, src/ORBmatcher.cc:1251-1482, after the caller's two projections: valid1[i1] = KF1 keypoint i1 has
 * a good, not-yet-matched map point that projects into KF2 at (u1,v1) with predicted level level1 (desc1 = its
 * descriptor); same for KF2 -> KF1.  match12[i1] = i2 of the mutually consistent pair or -1; returns nFound. */
int orc_search_by_sim3(orc_frame *KF1, orc_frame *KF2, const float *sf1, const float *sf2, const uint8_t *valid1,
                       const float *u1, const float *v1, const int32_t *level1, const uint8_t *desc1,
                       const uint8_t *valid2, const float *u2, const float *v2, const int32_t *level2,
                       const uint8_t *desc2, float th, int32_t *match12);

/* ---- cv::remap(src, dst, map1 CV_32FC1, map2 CV_32FC1, INTER_LINEAR) with the default BORDER_CONSTANT(0),
 * 8UC1: the EuRoC rectification at Examples/Stereo/stereo_euroc.cc:136-137.  OpenCV is a third-party dependency
 * absent from the reference tree (CMakeLists.txt asks for 3.0, falls back to 2.4.3); this restates its
 * published algorithm (imgproc/src/imgwarp.cpp, RemapInvoker + remapBilinear<FixedPtCast<int,uchar,15>>):
 * maps are rounded to 1/32 px (cvRound(v*32)), coordinates saturate to int16, weights are the exact products
 * (32-fy)(32-fx)*32 of 2^15, result (sum + 2^14) >> 15; a tap outside the source reads 0.  Parity unpinned. */
void orc_remap_linear(const uint8_t *src, int sw, int sh, int sstride, const float *mapx, const float *mapy,
                      int map_stride, int dw, int dh, uint8_t *dst, int dstride);

/* ---- cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) as called by Frame::UndistortKeyPoints /
 * ComputeImageBounds (src/Frame.cc:443-475, 481-510).  OpenCV (third party, absent; 3.0 / 2.4.3 per
 * CMakeLists.txt) -- restated from its published cvUndistortPoints (imgproc/src/undistort.cpp): float inputs
 * widened to double, 5 fixed-point iterations of the radial-tangential model, re-projection with P = K,
 * results narrowed to float.  K = fx, fy, cx, cy (floats, as mK is CV_32F); dist = k1,k2,p1,p2[,k3[,k4,k5,k6]].
 * Parity unpinned. */
void orc_undistort_points(const float *xy, int n, const float *K4, const float *dist, int n_dist, float *out_xy);
/* Frame::ComputeImageBounds (src/Frame.cc:481-510) -> mnMinX, mnMaxX, mnMinY, mnMaxY */
int orc_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P, int w,
                                   int h, float *map_x, float *map_y);
void orc_image_bounds(int cols, int rows, const float *K4, const float *dist, int n_dist, float *bounds4);
/* Frame::ComputeStereoFromRGBD (src/Frame.cc:689-713): d = imDepth.at<float>(v,u) with the float
 * coordinates of the DISTORTED keypoint truncated to int; mvuRight = kpU.x - mbf/d for d > 0. */
void orc_stereo_from_rgbd(const float *kx, const float *ky, const float *kux, int n, const float *depth_img, int w,
                          int h, int stride_floats, float mbf, float *uRight, float *depth);

/* ---- DBoW2 vocabulary (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) ---- */
typedef struct orc_vocab {
  int k, L, scoring, weighting;
  int n_nodes, n_words;
  int32_t *parent;     /* [n_nodes] */
  int32_t *child_off;  /* [n_nodes+1] into child_idx, children in file order */
  int32_t *child_idx;
  uint8_t *desc;       /* [n_nodes*32] */
  double *weight;      /* [n_nodes] */
  int32_t *word_id;    /* [n_nodes], -1 when the node is not a word */
} orc_vocab;

/* loadFromTextFile (:1338-1424).  Deviation: an empty (trailing) line is ignored; the reference
 * turns it into a child of the root with an UNINITIALISED descriptor (UB). */
orc_vocab *orc_vocab_load_text(const char *path);
orc_vocab *orc_vocab_from_arrays(int k, int L, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                                 const uint8_t *desc, const double *weight);
void orc_vocab_free(orc_vocab *v);
/* transform(feature, word_id, weight, nid, levelsup) (:1218-1259) for n features.
 * Returns the number of features with weight > 0 (the ones transform(features,...) :1127-1194
 * adds to the Bow/Feature vectors). */
int orc_vocab_transform(const orc_vocab *v, const uint8_t *desc, int n, int levelsup,
                        uint32_t *word_id, double *weight, uint32_t *node_id);

#ifdef __cplusplus
}
#endif
#endif
