/*
 * orbfe.h -- C-ABI of the MI355X-native ORB front-end (liborbfe.so).
 *
 * Drop-in boundary for the ONE hot path of saber/ORB_SLAM2_Annotate:
 * ORBextractor::operator() + Hamming matchers + Frame::ComputeStereoMatches.
 * Plain pointers and sizes only; no C++/torch/OpenCV types; no exceptions cross
 * this boundary.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the thin C++
 * classes (include/ORBextractor.h, include/ORBmatcher.h of this repo) that keep
 * the reference's class API on top of these calls.
 *
 * Threading contract (reference: src/Frame.cc:78-81, src/LocalMapping.cc:261,
 * src/LoopClosing.cc:294): one extractor handle per thread, handles are fully
 * independent (own HIP stream + workspace); matcher entry points are re-entrant.
 *
 * There is NO CPU fallback: every call needs a gfx950 device and returns
 * ORBFE_ERR_HIP when the HIP runtime reports none.
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_MAX_LEVELS 16

enum {
  ORBFE_OK = 0,
  ORBFE_ERR_INVALID = -1,  /* bad argument (null pointer, non-positive size, ...) */
  ORBFE_ERR_CAPACITY = -2, /* caller-provided output buffer too small */
  ORBFE_ERR_HIP = -3,      /* HIP runtime / device failure (see orbfe_last_error) */
  ORBFE_ERR_NOMEM = -4
};

/* Layout-identical to cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response,
 * octave, class_id -- what ORBextractor::operator() fills (src/ORBextractor.cc:905-916,1187-1195). */
typedef struct orbfe_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} orbfe_keypoint;

typedef struct orbfe_extractor orbfe_extractor;

/* Thread-local text of the last failure (never NULL). */
const char *orbfe_last_error(void);
/* Number of visible HIP devices (0 if none / runtime error). */
int orbfe_device_count(void);

/* ------------------------------------------------------------------------- */
/* Extractor                                                                  */
/* ------------------------------------------------------------------------- */

/* Replaces ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels,
 * int iniThFAST, int minThFAST)  (include/ORBextractor.h:52-53, src/ORBextractor.cc:415-486).
 * `device` = HIP device ordinal. */
int orbfe_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                           int minThFAST, int device, orbfe_extractor **out);
void orbfe_extractor_destroy(orbfe_extractor *e);

/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors /
 * GetScaleSigmaSquares / GetInverseScaleSigmaSquares (include/ORBextractor.h:66-87).
 * Each `out` receives nlevels floats. */
int orbfe_extractor_get_levels(const orbfe_extractor *e);
float orbfe_extractor_get_scale_factor(const orbfe_extractor *e);
int orbfe_extractor_get_scale_factors(const orbfe_extractor *e, float *out);
int orbfe_extractor_get_inverse_scale_factors(const orbfe_extractor *e, float *out);
int orbfe_extractor_get_scale_sigma_squares(const orbfe_extractor *e, float *out);
int orbfe_extractor_get_inverse_scale_sigma_squares(const orbfe_extractor *e, float *out);
/* mnFeaturesPerLevel (src/ORBextractor.cc:448-458) and umax (:471-485), for tests. */
int orbfe_extractor_get_features_per_level(const orbfe_extractor *e, int32_t *out);
int orbfe_extractor_get_umax(const orbfe_extractor *e, int32_t *out16);
/* Upper bound on keypoints one frame can return (quota + octree overshoot, Appendix A3). */
int orbfe_extractor_max_keypoints(const orbfe_extractor *e);
/* Exact output bound for a width x height image: sum over levels of max(quota + 3, 4 * nIni), where
 * nIni = round(width' / height') roots start the octree (very wide, flat images can exceed the bound of
 * orbfe_extractor_max_keypoints, which assumes nIni <= 16). */
int orbfe_extractor_max_keypoints_for(const orbfe_extractor *e, int width, int height);

/* Replaces ORBextractor::operator()(InputArray image, InputArray mask (ignored),
 * vector<KeyPoint>&, OutputArray descriptors)  (src/ORBextractor.cc:1119-1197).
 * `image`: host, 8-bit single channel, row-major, `stride` bytes per row.
 * Writes up to `capacity` keypoints and capacity*32 descriptor bytes (row i <-> keypoint i);
 * *n_out = count.  Empty image (NULL or w/h<=0) -> ORBFE_OK with *n_out = 0
 * (the reference returns silently, :1122-1123). */
int orbfe_extract(orbfe_extractor *e, const uint8_t *image, int width, int height, int stride,
                  orbfe_keypoint *keypoints, uint8_t *descriptors, int capacity, int *n_out);

/* Batched form of the same call for n_frames equally-sized frames (frame f starts at
 * images + f*frame_stride).  Per-frame outputs are packed at fixed slots of
 * `capacity` entries: keypoints[f*capacity + i], descriptors[(f*capacity + i)*32];
 * n_out[f] = count of frame f.  This is how a stereo Frame (2 images, src/Frame.cc:78-81)
 * or a whole sequence shard is pushed through the GPU in one pass. */
int orbfe_extract_batch(orbfe_extractor *e, const uint8_t *images, int n_frames, int width,
                        int height, int stride, size_t frame_stride, orbfe_keypoint *keypoints,
                        uint8_t *descriptors, int capacity, int *n_out);

/* Pinned (page-locked) host memory for the buffers of orbfe_extract_batch_pipelined. */
int orbfe_host_alloc(void **p, size_t bytes);
void orbfe_host_free(void *p);

/* operator() for a host batch, end to end and pipelined: the batch is cut into chunks of chunk_frames (0 = 256); the
 * H2D copy of chunk k+1, the kernels of chunk k and the D2H copy of chunk k-1 overlap on separate HIP streams (two
 * input slabs / output blocks in HBM, ordered by events, no host wait inside the loop).  Output layout as
 * orbfe_extract_batch (frame f at keypoints + f*capacity, descriptors + f*capacity*32; rows past n_out[f] are
 * unspecified).  Buffers from orbfe_host_alloc move at PCIe speed; other memory is page-locked for the call. */
int orbfe_extract_batch_pipelined(orbfe_extractor *e, const uint8_t *images, int n_frames, int width, int height,
                                  int stride, size_t frame_stride, orbfe_keypoint *keypoints,
                                  uint8_t *descriptors, int capacity, int *n_out, int chunk_frames);

/* Same, but `d_images`, `d_keypoints`, `d_descriptors`, `d_n_out` are DEVICE pointers
 * (HBM-resident in, HBM-resident out; nothing crosses PCIe except per-batch control words).
 * The call returns after the work is complete on the handle's stream.
 * Frames are read IN PLACE at any stride and alignment; the kernels stage whole rows with 16-byte requests bounded by
 * the row PITCH, so every frame -- the last one included -- must be readable for stride * height bytes (a buffer that
 * ends at the last pixel of a frame with stride > width is not enough): frame_stride >= stride * height is required. */
int orbfe_extract_batch_device(orbfe_extractor *e, const uint8_t *d_images, int n_frames,
                               int width, int height, int stride, size_t frame_stride,
                               orbfe_keypoint *d_keypoints, uint8_t *d_descriptors, int capacity,
                               int32_t *d_n_out);

/* Asynchronous form: enqueues the whole batch on the handle's stream and returns; results are
 * valid after orbfe_extractor_synchronize (calls on one handle execute in order, so a caller may
 * keep several batches in flight as long as each uses its own output buffers). */
int orbfe_extract_batch_device_async(orbfe_extractor *e, const uint8_t *d_images, int n_frames,
                                     int width, int height, int stride, size_t frame_stride,
                                     orbfe_keypoint *d_keypoints, uint8_t *d_descriptors,
                                     int capacity, int32_t *d_n_out);
int orbfe_extractor_synchronize(orbfe_extractor *e);

/* Replaces reads of the public member `mvImagePyramid[level]` (include/ORBextractor.h:86;
 * read by Frame::ComputeStereoMatches, src/Frame.cc:519,609,621,626): copies level `level`
 * of frame `frame` of the LAST extract call into `dst` (host, dst_stride bytes per row).
 * level_size gives the dimensions for any input size.
 * `frame` always counts in the caller's batch.  The chunked host paths (orbfe_extract_batch_pipelined, and
 * orbfe_extract_batch when it routes a large batch there) keep the intermediate data of their LAST chunk only:
 * asking for a frame before it fails with ORBFE_ERR_INVALID ("pyramid of frame F not retained ...") -- this
 * holds for every frame-indexed accessor below and for orbfe_compute_stereo_matches. */
int orbfe_extractor_level_size(const orbfe_extractor *e, int width, int height, int level,
                               int *w, int *h);
int orbfe_extractor_get_pyramid_level(orbfe_extractor *e, int frame, int level, uint8_t *dst,
                                      int dst_stride);
/* Device view of the same (valid until the next extract call on this handle). */
int orbfe_extractor_pyramid_level_device(orbfe_extractor *e, int frame, int level,
                                         const uint8_t **d_ptr, int *pitch, int *w, int *h);

/* Stage diagnostics used by tests: raw grid-stage candidates of one level of frame 0 of
 * the last call, in emission order (src/ORBextractor.cc:846-896), coordinates relative to
 * (minBorderX,minBorderY).  Returns count or a negative status. */
int orbfe_extractor_debug_candidates(orbfe_extractor *e, int frame, int level, float *xs,
                                     float *ys, float *resp, int cap);
/* Blurred level (the `workingMat` of src/ORBextractor.cc:1169-1175) of the last call. */
int orbfe_extractor_debug_blurred_level(orbfe_extractor *e, int frame, int level, uint8_t *dst,
                                        int dst_stride);

/* Host-logic debug entry points; they run WITHOUT a GPU (CPU tests compare them with the oracle).
 * debug_octree_host: the library's host DistributeOctTree on flat arrays (candidates relative to
 * minBorder in emission order -> selected keypoints in list order, level coordinates); returns the count.
 * debug_geometry: per level 9 ints (w, h, nCols, nRows, wCell, hCell, nCells, quota, nIni) and the FAST
 * grid cells as 5 int16 (level, x0, y0, w, h: the detection rectangle); returns the cell count.
 * debug_resize_tables: the cv::resize fixed-point tables (xofs[dw], alpha[2*dw], yofs[dh], beta[2*dh]). */
int orbfe_debug_octree_host(const uint16_t *xs, const uint16_t *ys, const uint8_t *resp, int n, int minX, int maxX,
                            int minY, int maxY, int N, uint16_t *out_x, uint16_t *out_y, uint8_t *out_resp,
                            int cap);
int orbfe_debug_geometry(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width,
                         int height, int32_t *levels9, float *tables4 /* scale, invScale, sigma2, invSigma2 per level; may be NULL */,
                         int16_t *cells5, int cell_cap);
int orbfe_debug_resize_tables(int sw, int sh, int dw, int dh, int32_t *xofs, int16_t *alpha, int32_t *yofs,
                              int16_t *beta);
/* Tile ownership of the fused blur + resize kernel for one level pair: tile_gx[(sw+63)/64 + 1] / tile_dy[(sh+63)/64 + 1]
 * (first 4-column group / output row owned by each 64 x 64 source tile, then the totals), group_start[(dw+3)/4] (first
 * source column of each group's 8-byte tap window), row_upper[dh] (upper source row of each output row, clamped).
 * *n_tiles_x / *n_tiles_y are 0 when the fused kernel is not used for these sizes. */
int orbfe_debug_resize_tiles(int sw, int sh, int dw, int dh, int32_t *tile_gx, int *n_tiles_x, int32_t *tile_dy,
                             int *n_tiles_y, int32_t *group_start, int32_t *row_upper);

/* Debug cross-check: when enabled, DistributeOctTree runs in the library's host implementation
 * (D2H/H2D round trip) instead of the device kernel.  Off by default; results are identical. */
int orbfe_extractor_debug_host_octree(orbfe_extractor *e, int enable);

/* Order of the two FAST thresholds inside the grid-stage kernel (src/ORBextractor.cc:874-882): 1 = iniThFAST first and
 * minThFAST only for the cells that found nothing (the reference's own order; fastest on textured images), 2 = one
 * attempt at the lower threshold that classifies for both (fastest when most cells fall back: sparse, low-contrast
 * images), 0 = auto (default; $ORBFE_FAST_MODE=auto|high|low): per call, from the fallback rate the last finished
 * launch measured.  The results are identical in every mode. */
int orbfe_extractor_set_fast_mode(orbfe_extractor *e, int mode);

/* Orientation + descriptor stage: 0 (default) = per-keypoint gathers (k_orient_desc), 1 = tile form (k_orient_desc_tiles:
 * a workgroup stages a 128 x 128 tile of the level and of the blurred level in LDS once for all keypoints inside it),
 * -1 = $ORBFE_DESC_TILES or the default.  Identical results; see DESIGN.md 4 for when which one is faster. */
int orbfe_extractor_set_desc_tiles(orbfe_extractor *e, int enable);
/* How the sub-batches of a device-batch call (orbfe_extractor_set_streams) are scheduled: 0 = one independent HIP
 * stream per sub-batch; 1 ($ORBFE_LANES) = three lanes shared by all sub-batches -- pyramid | FAST + blur |
 * gather + octree + orientation/descriptors -- ordered by events into a software pipeline, so that exactly one
 * VALU-bound kernel runs beside the latency-bound ones at any time.  Results are identical. */
int orbfe_extractor_set_schedule(orbfe_extractor *e, int lanes);

/* Which cv::GaussianBlur(7x7, sigma 2) arithmetic the extractor reproduces.  The reference does not pin its OpenCV
 * (CMakeLists.txt:33-39 accepts 2.4.3 and 3.x, README.md:74 names 2.4.11 / 3.2) and the 8-bit path changed:
 *   ORBFE_BLUR_CV4        (0, default; $ORBFE_BLUR_SPEC) OpenCV >= 3.4.1 / 4.x: taps 18 34 48 56 48 34 18 (/256),
 *                          (x + 2^15) >> 16;
 *   ORBFE_BLUR_CV2_SCALAR (1) OpenCV 2.4.x / 3.0-3.3 without SIMD: taps 18 34 49 55 49 34 18 (each rounded on its own,
 *                          sum 257), saturate((x + 2^15) >> 16);
 *   ORBFE_BLUR_CV2_SSE2   (2) the same versions on x86 with SSE2: round-half-to-even on the first width & ~3 columns
 *                          (float column pass + cvtps2dq), the scalar form on the last width & 3.
 * All three are restated from OpenCV's published sources and unverifiable in this image; the CPU
 * oracle implements the same three and the GPU equals it for each. */
enum { ORBFE_BLUR_CV4 = 0, ORBFE_BLUR_CV2_SCALAR = 1, ORBFE_BLUR_CV2_SSE2 = 2 };
int orbfe_extractor_set_blur_spec(orbfe_extractor *e, int spec);

/* GaussianBlur fused into the FAST kernel (1) or run as its own launch (0, the default; $ORBFE_FUSED).  Results
 * are identical; the fused form measured no faster and is kept as a parity-tested alternative. */
int orbfe_extractor_set_fused(orbfe_extractor *e, int enable);

/* ComputePyramid (src/ORBextractor.cc:1203-1234) and the per-level GaussianBlur (:1169-1175) as ONE kernel per level
 * -- the tile staged for the blur of level l also yields the part of level l+1 whose taps start in it -- (1, the
 * default; $ORBFE_PYRBLUR) or as separate resize and blur launches (0).  Identical results; the fused form reads every
 * level once instead of twice (KITTI +4 % stereo frames/s).  Ignored with the lane schedule and with
 * orbfe_extractor_set_fused(1). */
int orbfe_extractor_set_pyramid_blur(orbfe_extractor *e, int enable);

/* Calls of at most 8 frames (the live camera) build the pyramid with n-1 dependent resize launches of a few microseconds
 * each.  1: ONE launch instead -- a workgroup owns a tile of one level and recomputes, in LDS, the rectangles of the levels
 * below it from level 0 (k_pyramid_chain: same fixed-point step from the same inputs, identical pixels).  0 (default;
 * $ORBFE_PYR_CHAIN): the launches -- the one-launch form measured no faster (32 vs 29 us per KITTI frame). */
int orbfe_extractor_set_pyramid_chain(orbfe_extractor *e, int enable);

/* Order of the two separable passes of the 7x7 GaussianBlur kernel (src/ORBextractor.cc:1169-1175), process-wide.  Both
 * passes are exact integer sums, so the blurred bytes do not depend on it.  1 (default; $ORBFE_BLUR_HFIRST): horizontal
 * pass on the staged bytes with v_dot4_u32_u8, vertical pass on row pairs with v_dot2_u32_u16 (~160 VALU instructions per
 * wave for the two passes); 0: the rounds 1-3 form, vertical packed-16 pass first (216 per wave).  Returns the order now in force; a
 * negative argument only queries.  Takes effect at the next launch. */
int orbfe_set_blur_pass_order(int order);

/* A call's frames are split into n consecutive sub-batches that run concurrently on n HIP
 * streams with private workspace slices (1..32, default 1 or $ORBFE_STREAMS); results do not
 * depend on n.  Stage timing covers the kernels of sub-batch 0 (frames_out reports how many
 * frames those launches processed). */
int orbfe_extractor_set_streams(orbfe_extractor *e, int n);

/* Per-kernel timing, measured with HIP events on the handle's own stream (events are recorded
 * inside the calls and read back at the next synchronisation, so timing does not stall the
 * pipeline).  stage_mask: bit i enables ORBFE_STAGE_i, -1 = all, 0 = off; setting it resets the
 * accumulators.  get returns, per stage, total milliseconds and launches since the last reset. */
enum {
  ORBFE_STAGE_H2D = 0,
  ORBFE_STAGE_PYRAMID,
  ORBFE_STAGE_FAST,
  ORBFE_STAGE_OCTREE,
  ORBFE_STAGE_BLUR,
  ORBFE_STAGE_ORIENT_DESC,
  ORBFE_STAGE_D2H,
  ORBFE_STAGE_MATCH, /* batched matcher enqueued on the handle's stream (stereo / consecutive-frame BoW) */
  ORBFE_STAGE_COUNT
};
int orbfe_extractor_profile(orbfe_extractor *e, int stage_mask);
int orbfe_extractor_profile_get(orbfe_extractor *e, double *ms_out /*[ORBFE_STAGE_COUNT]*/,
                                int64_t *launches_out /*[ORBFE_STAGE_COUNT]*/,
                                double *frames_out /*[ORBFE_STAGE_COUNT], may be NULL*/);
const char *orbfe_stage_name(int stage);

/* Standalone primitives on host buffers (tests of the individual kernels). */
int orbfe_resize_linear(int device, const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                        int dw, int dh, int dstride);
int orbfe_gaussian_blur7_spec(int device, int spec, const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                             int dstride);
int orbfe_gaussian_blur7(int device, const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                         int dstride);

/* ------------------------------------------------------------------------- */
/* Matcher                                                                    */
/* ------------------------------------------------------------------------- */

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>, ascending node id;
 * Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45) flattened to CSR. */
typedef struct orbfe_featvec {
  int32_t n_nodes;
  const uint32_t *node_ids; /* ascending */
  const int32_t *offsets;   /* n_nodes + 1 */
  const uint32_t *indices;  /* feature indices, grouped by node in insertion order */
} orbfe_featvec;

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1828-1844) for n pairs:
 * out[i] = popcount(a[i] ^ b[i]) over 256 bits. Host buffers. */
int orbfe_descriptor_distance(int device, const uint8_t *a, const uint8_t *b, int n, int32_t *out);

/* Dense n1 x n2 Hamming matrix (row-major int16 would do; int32 for simplicity). Host buffers. */
int orbfe_hamming_matrix(int device, const uint8_t *desc1, int n1, const uint8_t *desc2, int n2,
                         int32_t *out);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>&)  (src/ORBmatcher.cc:185-325).
 * has_mp1[i] != 0 <=> KF feature i has a non-bad MapPoint.  match_f[j] (size n2) = index of the KF
 * feature whose MapPoint the wrapper assigns to frame feature j, or -1.  Returns nmatches >= 0,
 * or a negative status. */
int orbfe_search_by_bow(int device, const uint8_t *desc1, const uint8_t *has_mp1,
                        const float *angle1, int n1, const orbfe_featvec *fv1,
                        const uint8_t *desc2, const float *angle2, int n2,
                        const orbfe_featvec *fv2, float nnratio, int check_orientation,
                        int32_t *match_f);

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)  (src/ORBmatcher.cc:610-743).
 * match12[i] (size n1) = KF2 feature index or -1. */
int orbfe_search_by_bow_kf(int device, const uint8_t *desc1, const uint8_t *has_mp1,
                           const float *angle1, int n1, const orbfe_featvec *fv1,
                           const uint8_t *desc2, const uint8_t *has_mp2, const float *angle2,
                           int n2, const orbfe_featvec *fv2, float nnratio, int check_orientation,
                           int32_t *match12);

/* ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12,
 * vector<pair<size_t,size_t>>&, bool bOnlyStereo)  (src/ORBmatcher.cc:754-928).
 * Keypoints are mvKeysUn as SoA; stereo flags are (mvuRight[i] >= 0); F12 is row-major 3x3;
 * (ex,ey) the epipole computed by the caller as at :766-769; scale_factors2 / level_sigma2_2 are
 * KF2's mvScaleFactors / mvLevelSigma2.  match12[i] = KF2 index or -1; the wrapper emits the
 * pairs in ascending i exactly as :920-925. */
int orbfe_search_for_triangulation(int device, const uint8_t *desc1, const uint8_t *has_mp1,
                                   const float *x1, const float *y1, const float *angle1,
                                   const uint8_t *stereo1, int n1, const orbfe_featvec *fv1,
                                   const uint8_t *desc2, const uint8_t *has_mp2, const float *x2,
                                   const float *y2, const float *angle2, const int32_t *octave2,
                                   const uint8_t *stereo2, int n2, const orbfe_featvec *fv2,
                                   const float *F12, float ex, float ey,
                                   const float *scale_factors2, const float *level_sigma2_2,
                                   int n_levels2, int only_stereo, int check_orientation,
                                   int32_t *match12);

/* Frame::ComputeStereoMatches()  (src/Frame.cc:512-686).  `left`/`right` are the two extractor
 * handles whose LAST extract call produced the keypoints (their pyramids are read on the device,
 * replacing mpORBextractorLeft/Right->mvImagePyramid); frameL/frameR select the frame of that
 * batch.  kp/desc are host arrays exactly as returned by orbfe_extract.  Outputs mvuRight / mvDepth
 * (N floats each, -1 = no stereo).  mbf, mb as in src/Frame.cc:114,542-544. */
int orbfe_compute_stereo_matches(orbfe_extractor *left, int frameL, orbfe_extractor *right,
                                 int frameR, const orbfe_keypoint *kpL, const uint8_t *descL,
                                 int N, const orbfe_keypoint *kpR, const uint8_t *descR, int Nr,
                                 float mbf, float mb, float *uRight, float *depth);

/* The same for a device-resident batch: the handle's LAST orbfe_extract_batch_device(_async) call
 * processed frames L0,R0,L1,R1,... (pair p = frames 2p, 2p+1, both images through ONE handle);
 * d_keypoints / d_descriptors / d_n are that call's outputs (same capacity).  Writes mvuRight and
 * mvDepth of pair p at [p*capacity + i] (i < capacity; -1 beyond the left frame's keypoints) and
 * the number of surviving stereo matches at d_n_stereo[p].  Enqueued on the handle's stream after
 * the extraction; wait with orbfe_extractor_synchronize. */
int orbfe_stereo_match_batch_device(orbfe_extractor *e, int n_pairs, const orbfe_keypoint *d_keypoints,
                                    const uint8_t *d_descriptors, const int32_t *d_n, int capacity,
                                    float mbf, float mb, float *d_uRight, float *d_depth,
                                    int32_t *d_n_stereo);

/* The front end of the stereo Frame constructor in ONE call (src/Frame.cc:78-96: ExtractORB(0, imLeft) and
 * ExtractORB(1, imRight) on two threads, join, ComputeStereoMatches): both eyes go through handle e as a two-frame
 * batch, the stereo matcher runs on the records while they are still in HBM, and keypoints / descriptors of both eyes,
 * mvuRight and mvDepth (n_left floats each, -1 = no stereo) come back in one download.  Same results as two
 * orbfe_extract calls + orbfe_compute_stereo_matches.  The handle then holds the pair as frames 0 (left) and 1 (right):
 * orbfe_frame_from_extractor(e, 0, ...) builds the resident Frame from it. */
int orbfe_extract_stereo_frame(orbfe_extractor *e, const uint8_t *left, const uint8_t *right, int width, int height,
                               int stride, orbfe_keypoint *kp_left, uint8_t *desc_left, int *n_left,
                               orbfe_keypoint *kp_right, uint8_t *desc_right, int *n_right, int capacity, float mbf,
                               float mb, float *uRight, float *depth);

/* ------------------------------------------------------------------------- */
/* DBoW2 vocabulary (the step right before SearchByBoW)       */
/* ------------------------------------------------------------------------- */
typedef struct orbfe_vocabulary orbfe_vocabulary;

/* TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424):
 * first line "k L scoring weighting", then one node per line "parent is_leaf d0..d31 weight".
 * Deviation: empty lines are ignored (the reference turns the file's trailing empty line into a
 * child of the root with an UNINITIALISED descriptor). */
int orbfe_vocabulary_load_text(const char *path, int device, orbfe_vocabulary **out);
/* The same tree from arrays -- what the node lines of the text file hold (:1374-1417): entry i is node i+1 (the
 * root is node 0), parent[i] the NodeId of its parent, is_leaf[i] != 0 makes it the next word (word ids count leaves
 * in array order, :1402-1408), descriptors[32*i ..] and weight[i] as in the line.  k, L, scoring, weighting = the
 * header line.  For callers that hold or convert a vocabulary in memory (ORBvoc.txt is 145 MB of text). */
int orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                            const uint8_t *is_leaf, const uint8_t *descriptors, const double *weight,
                            int device, orbfe_vocabulary **out);
void orbfe_vocabulary_destroy(orbfe_vocabulary *v);
int orbfe_vocabulary_info(const orbfe_vocabulary *v, int *k, int *L, int *n_nodes, int *n_words);

/* transform(feature, word_id, weight, nid, levelsup) (:1218-1259) for n host descriptors: the
 * k-ary descent with FORB::distance.  Features with weight > 0 are the ones transform(features,
 * BowVector&, FeatureVector&, levelsup) (:1127-1194; called with levelsup = 4 at src/Frame.cc:438)
 * adds as BowVector.addWeight(word_id, weight) / FeatureVector.addFeature(node_id, i).  Returns
 * their number, or a negative status. */
int orbfe_vocabulary_transform(orbfe_vocabulary *v, const uint8_t *descriptors, int n, int levelsup,
                               uint32_t *word_id, double *weight, uint32_t *node_id);

/* Frame::ComputeBoW for every frame of a device-resident extractor batch: FeatureVector f as CSR at
 * d_fv_nodes[f*capacity ..] (ascending, d_fv_count[f] of them), d_fv_offsets[f*(capacity+1) ..],
 * d_fv_indices[f*capacity ..]; optional per-feature word ids / weights (both or neither). */
int orbfe_vocabulary_featvec_batch_device(orbfe_vocabulary *v, const uint8_t *d_descriptors,
                                          const int32_t *d_n, int n_frames, int capacity, int levelsup,
                                          uint32_t *d_fv_nodes, int32_t *d_fv_offsets,
                                          uint32_t *d_fv_indices, int32_t *d_fv_count, uint32_t *d_word,
                                          double *d_weight);

/* ComputeBoW + SearchByBoW(KF = frame t-1, F = frame t) for t = 1..n_frames-1 of a device-resident
 * batch (every KF feature counts as having a MapPoint): d_match[(t-1)*capacity + j] = index of the
 * frame t-1 feature matched to feature j of frame t, or -1; d_nmatches[t-1] = nmatches. */
int orbfe_bow_match_consecutive_batch_device(orbfe_vocabulary *v, int n_frames,
                                             const orbfe_keypoint *d_keypoints,
                                             const uint8_t *d_descriptors, const int32_t *d_n,
                                             int capacity, int levelsup, float nnratio,
                                             int check_orientation, int32_t *d_match,
                                             int32_t *d_nmatches);
/* The same, enqueued on extractor e's stream behind every sub-batch of its last
 * orbfe_extract_batch_device_async call (ordered on the device, no host wait); returns at once,
 * orbfe_extractor_synchronize(e) waits.  The next extract call on e may be enqueued right away: its
 * sub-batch streams wait for this matcher before they overwrite the keypoints it reads. */
int orbfe_bow_match_consecutive_batch_device_async(orbfe_vocabulary *v, orbfe_extractor *e, int n_frames,
                                                   const orbfe_keypoint *d_keypoints,
                                                   const uint8_t *d_descriptors, const int32_t *d_n,
                                                   int capacity, int levelsup, float nnratio,
                                                   int check_orientation, int32_t *d_match,
                                                   int32_t *d_nmatches);

/* The same over the LEFT frames of an (L0, R0, L1, R1, ...) stereo batch of extractor e -- a stereo Frame's
 * ComputeBoW / SearchByBoW use its left keypoints (mvKeys, mDescriptors; src/Frame.cc:61-117): pair t-1 against
 * pair t for t = 1..n_pairs-1; d_match[(t-1)*capacity + j], d_nmatches[t-1] as above.  d_keypoints / d_descriptors /
 * d_n are the extractor batch's arrays (2*n_pairs frames). */
int orbfe_bow_match_consecutive_stereo_batch_device_async(orbfe_vocabulary *v, orbfe_extractor *e, int n_pairs,
                                                          const orbfe_keypoint *d_keypoints,
                                                          const uint8_t *d_descriptors, const int32_t *d_n,
                                                          int capacity, int levelsup, float nnratio,
                                                          int check_orientation, int32_t *d_match,
                                                          int32_t *d_nmatches);

/* ------------------------------------------------------------------------- */
/* Next to the path                                */
/* ------------------------------------------------------------------------- */

/* cv::cvtColor(im, im, CV_RGB2GRAY | CV_BGR2GRAY | CV_RGBA2GRAY | CV_BGRA2GRAY) of
 * Tracking::GrabImage{Stereo,RGBD,Monocular} (src/Tracking.cc:176-262), 8-bit, channels 3 or 4,
 * rgb_order != 0 for RGB(A) input: (R*4899 + G*9617 + B*1868 + 2^13) >> 14.  Host buffers. */
int orbfe_cvt_gray(int device, const uint8_t *src, int width, int height, int stride, int channels,
                   int rgb_order, uint8_t *dst, int dst_stride);
/* Same on device-resident frames (so colour input never round-trips through the host). */
int orbfe_cvt_gray_batch_device(int device, const uint8_t *d_src, int n_frames, int width, int height,
                                int stride, size_t frame_stride, int channels, int rgb_order,
                                uint8_t *d_dst, int dst_stride, size_t dst_frame_stride);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:269-333) for n_points map points at once:
 * map point m owns descriptors [offsets[m], offsets[m+1]) (its observations, 32 bytes each);
 * best_index[m] = index inside that range of the descriptor with the least median Hamming distance
 * to the others (median = sorted row [(size_t)(0.5*(n-1))]; first minimum wins), -1 if empty. */
int orbfe_distinctive_descriptors(int device, const uint8_t *descriptors, const int32_t *offsets,
                                  int n_points, int32_t *best_index);

/* ------------------------------------------------------------------------- */
/* Tracking-thread projection searches                */
/* ------------------------------------------------------------------------- */

/* The fields of an ORB_SLAM2::Frame the searches read (include/Frame.h:120-190): the undistorted
 * keypoints mvKeysUn split into arrays, mvuRight, mDescriptors and the undistorted image bounds
 * mnMinX..mnMaxY (src/Frame.cc:697-728).  At most 16384 keypoints. */
typedef struct orbfe_frame_view {
  int32_t n;
  const float *x, *y;       /* mvKeysUn[i].pt */
  const int32_t *octave;    /* mvKeysUn[i].octave */
  const float *angle;       /* mvKeysUn[i].angle (only read with check_orientation) */
  const float *u_right;     /* mvuRight, NULL for monocular frames */
  const uint8_t *desc;      /* mDescriptors, n x 32 */
  float min_x, max_x, min_y, max_y;
  const struct orbfe_frame *resident; /* NULL, or the handle orbfe_frame_upload made of this frame: the search then
                                         uploads nothing of the frame (use orbfe_frame_get_view) */
} orbfe_frame_view;

/* Device-resident Frame / KeyFrame operands.  LocalMapping matches one key frame against 10-20 neighbours
 * (src/LocalMapping.cc:256-315, 517-573), relocalisation one frame against several candidates
 * (src/Tracking.cc:1478-1498): orbfe_frame_upload moves what a frame contributes to any search -- keypoint arrays,
 * descriptors, the 64 x 48 grid of Frame::AssignFeaturesToGrid (built once), the FeatureVector's index list (fv may be
 * NULL for frames that only take part in projection searches) -- to the device once.  The view needs x, y, octave,
 * desc and the bounds; angle / u_right as the searches it will take part in need them.  The handle copies what it
 * needs (the caller's arrays may go away), is immutable, and may be used from any thread concurrently. */
typedef struct orbfe_frame orbfe_frame;
int orbfe_frame_upload(int device, const orbfe_frame_view *view, const orbfe_featvec *fv, orbfe_frame **out);
void orbfe_frame_release(orbfe_frame *f);
/* The handle's own view (host copies inside the handle, `resident` set): pass it wherever an orbfe_frame_view is
 * taken.  Valid until orbfe_frame_release. */
const orbfe_frame_view *orbfe_frame_get_view(const orbfe_frame *f);

/* Frame::Frame (src/Frame.cc:61-117) is ExtractORB -> UndistortKeyPoints -> ComputeStereoMatches -> AssignFeaturesToGrid:
 * when the Frame is built the extractor's keypoint records and descriptors are still in HBM.  These two build the
 * resident operands FROM THEM -- records split into the x / y / angle / octave arrays, descriptors copied device to device,
 * grid built on the device -- so of a frame's 60 bytes per keypoint only mvuRight (and, with ORBFE_FRAME_XY_FROM_VIEW, the
 * caller's undistorted positions) cross PCIe.  `view` holds the host arrays the argument checks and the host-pointer forms read
 * (mvKeysUn / mvuRight / mDescriptors as the caller has them after its own constructor steps); view->n records are taken.
 *   orbfe_frame_from_extractor: frame `frame` of the handle's last orbfe_extract / small orbfe_extract_batch call
 *       (ORBFE_ERR_INVALID after a device-batch or pipelined call: those outputs are the caller's);
 *   orbfe_frame_from_device:    any device arrays in the extractor's output layout (orbfe_extract_batch_device outputs at
 *       d_keypoints + frame * capacity, d_descriptors + frame * capacity * 32), complete when the call is made.
 * Like orbfe_frame_upload neither waits for the device: searches are ordered behind the build by the frame's own event.
 * Released slabs are pooled (no hipMalloc / hipFree -- which waits for the whole device -- per key frame). */
#define ORBFE_FRAME_XY_FROM_VIEW 1 /* x / y come from `view` (undistorted by the caller: cameras with distortion) */
int orbfe_frame_from_extractor(orbfe_extractor *e, int frame, const orbfe_frame_view *view, const orbfe_featvec *fv,
                               int flags, orbfe_frame **out);
int orbfe_frame_from_device(int device, const orbfe_keypoint *d_keypoints, const uint8_t *d_descriptors,
                            const orbfe_frame_view *view, const orbfe_featvec *fv, int flags, orbfe_frame **out);
/* Frame::ComputeBoW runs after the constructor (src/Tracking.cc:836-843, src/Frame.cc:433-440): attach the FeatureVector
 * to a frame made resident without one.  Call it before other threads use the handle. */
int orbfe_frame_set_featvec(orbfe_frame *f, const orbfe_featvec *fv);

/* Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (src/Frame.cc:246-267, 358-427) for
 * n_queries windows at once: count[q] features lie in the window of query q; the first
 * min(count[q], capacity) of them -- in the reference's order (grid column, grid row, feature
 * index) -- are written to indices[q*capacity ..].  ORBFE_ERR_CAPACITY (counts still exact) when a
 * window holds more than `capacity` features. */
int orbfe_features_in_area(int device, const orbfe_frame_view *frame, int n_queries, const float *x,
                           const float *y, const float *r, const int32_t *min_level,
                           const int32_t *max_level, int capacity, int32_t *count, int32_t *indices);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
 * (src/ORBmatcher.cc:51-138).  Map point i is described by what Frame::isInFrustum left on it:
 * in_view (mbTrackInView && !isBad()), level (mnTrackScaleLevel), view_cos (mTrackViewCos), proj_x /
 * proj_y / proj_xr (mTrackProjX/Y/XR), mp_desc (GetDescriptor()), mp_obs_positive (Observations()>0,
 * NULL = all).  blocked[idx] != 0 <=> F.mvpMapPoints[idx] already holds a point with observations
 * (NULL = none).  match[idx] = map point assigned to frame feature idx or -1; *n_matches as the
 * reference counts them. */
int orbfe_search_by_projection(int device, const orbfe_frame_view *F, const float *scale_factors,
                               int n_levels, const uint8_t *blocked, int n_mp, const uint8_t *in_view,
                               const int32_t *level, const float *view_cos, const float *proj_x,
                               const float *proj_y, const float *proj_xr, const uint8_t *mp_desc,
                               const uint8_t *mp_obs_positive, float th, float nnratio,
                               int32_t *match, int32_t *n_matches);

/* The FeatureVector searches on resident frames (uploaded WITH their FeatureVector and angles): per call only the
 * shared-node list, the MapPoint masks and the result travel.  Semantics and outputs of orbfe_search_by_bow /
 * orbfe_search_by_bow_kf. */
int orbfe_search_by_bow_resident(const orbfe_frame *kf, const uint8_t *has_mp_kf, const orbfe_frame *f, float nnratio,
                                 int check_orientation, int32_t *match_f);
int orbfe_search_by_bow_kf_resident(const orbfe_frame *kf1, const uint8_t *has_mp1, const orbfe_frame *kf2,
                                    const uint8_t *has_mp2, float nnratio, int check_orientation, int32_t *match12);
/* SearchByBoW against n_keyframes candidates in ONE call (one upload of the K shared-node lists and masks, K + 1
 * launches on one stream, one download):
 *   orbfe_search_by_bow_multi:    Tracking::Relocalization, src/Tracking.cc:1478-1498 -- SearchByBoW(pKF_k, mCurrentFrame,
 *       vvpMapPointMatches[k]) for every candidate key frame; match_f [k * f->n + i2] = feature of key frame k matched to
 *       feature i2 of the frame, or -1; n_matches [k];
 *   orbfe_search_by_bow_kf_multi: LoopClosing::ComputeSim3, src/LoopClosing.cc:294-321 -- SearchByBoW(mpCurrentKF, pKF_k,
 *       vvpMapPointMatches[k]); match12 [k * kf1->n + i1] = feature of candidate k matched to feature i1 of kf1, or -1.
 * Every output equals the corresponding single-pair call. */
int orbfe_search_by_bow_multi(int n_keyframes, const orbfe_frame *const *kf, const uint8_t *const *has_mp_kf,
                              const orbfe_frame *f, float nnratio, int check_orientation, int32_t *match_f,
                              int32_t *n_matches);
int orbfe_search_by_bow_kf_multi(const orbfe_frame *kf1, const uint8_t *has_mp1, int n_keyframes,
                                 const orbfe_frame *const *kf2, const uint8_t *const *has_mp2, float nnratio,
                                 int check_orientation, int32_t *match12, int32_t *n_matches);
/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:754-928) of ONE key frame against n_neighbours key frames in
 * one call -- the loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:283-315): F12[9*k ..], ex[k], ey[k]
 * the fundamental matrix and epipole of neighbour k, has_mp2[k] its MapPoint mask; the stereo flags come from the
 * frames' u_right.  match12[k*n1 + i] = feature of neighbour k matched to feature i of kf1, or -1; n_matches[k].
 * One upload, 2 launches per neighbour on one stream, one download. */
int orbfe_search_for_triangulation_multi(const orbfe_frame *kf1, const uint8_t *has_mp1, int n_neighbours,
                                         const orbfe_frame *const *kf2, const uint8_t *const *has_mp2,
                                         const float *F12, const float *ex, const float *ey,
                                         const float *scale_factors2, const float *level_sigma2_2, int n_levels2,
                                         int only_stereo, int check_orientation, int32_t *match12,
                                         int32_t *n_matches);
/* The per-point search of ORBmatcher::Fuse (orbfe_fuse_search) for the SAME n map points against n_keyframes key
 * frames in one call -- LocalMapping::SearchInNeighbors (src/LocalMapping.cc:542-549).  valid / u / v / ur / level /
 * best_idx are [k*n + i]; mp_desc [n*32] is shared; resident views upload nothing of the key frames. */
int orbfe_fuse_search_multi(int device, int n_keyframes, const orbfe_frame_view *const *KF,
                            const float *scale_factors, const float *inv_level_sigma2, int n_levels, int n,
                            const uint8_t *valid, const float *u, const float *v, const float *ur,
                            const int32_t *level, const uint8_t *mp_desc, float th, int chi2_gate,
                            int32_t *best_idx);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th,
 * const bool bMono) (src/ORBmatcher.cc:1484-1633) after the caller's pose arithmetic: last-frame
 * point i is valid when it has a non-outlier map point with invzc >= 0 projecting to (u, v) inside
 * the image; last_octave / last_angle are LastFrame.mvKeysUn[i]; mp_desc its map point's
 * descriptor; obs_positive[i] = Observations()>0 (NULL = all).  mode 0: levels
 * [octave-1, octave+1]; 1 (bForward): >= octave; 2 (bBackward): <= octave.  mbf / invzc are read
 * only for frames with u_right.  blocked[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] holds a point with
 * Observations() > 0 at entry (:1572-1574; NULL = none, which is the state Tracking::TrackWithMotionModel calls it in).
 * match_cur[i2] = last-frame index or -1. */
int orbfe_search_by_projection_last_frame(int device, const orbfe_frame_view *Cur,
                                          const float *scale_factors, int n_levels, float mbf,
                                          int n_last, const uint8_t *valid, const float *u,
                                          const float *v, const float *invzc,
                                          const int32_t *last_octave, const float *last_angle,
                                          const uint8_t *mp_desc, const uint8_t *obs_positive,
                                          const uint8_t *blocked, int mode, float th, int check_orientation,
                                          int32_t *match_cur, int32_t *n_matches);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound,
 * const float th, const int ORBdist) (src/ORBmatcher.cc:1641-1775, relocalisation) after the caller's
 * projection: point i is valid when pKF's map point i exists, is good, is not in sAlreadyFound,
 * projects to (u, v) inside the image and lies inside its distance range; level = PredictScale;
 * kf_angle = pKF->mvKeysUn[i].angle.  blocked[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL
 * (NULL = none). */
int orbfe_search_by_projection_keyframe(int device, const orbfe_frame_view *Cur,
                                        const float *scale_factors, int n_levels,
                                        const uint8_t *blocked, int n, const uint8_t *valid,
                                        const float *u, const float *v, const int32_t *level,
                                        const float *kf_angle, const uint8_t *mp_desc, float th,
                                        int orb_dist, int check_orientation, int32_t *match_cur,
                                        int32_t *n_matches);

/* The same search of ONE current frame against the projected map points of n_candidates key frames in one call
 * (Tracking::Relocalization, src/Tracking.cc:1577,1595: per candidate whose PnP pose survived, th = 10 then 3, ORBdist = 100
 * then 64).  Candidate k brings n[k] points and its own arrays valid[k] / u[k] / v[k] / level[k] / kf_angle[k] / mp_desc[k] /
 * blocked[k] (blocked, or blocked[k], may be NULL), window th[k] and bound orb_dist[k].  One upload of all query lists, one
 * launch group, one download; the frame itself goes up once (or not at all when resident).
 * match_cur [k * Cur->n + i2], n_matches [k]; every output equals the single call. */
int orbfe_search_by_projection_keyframe_multi(int device, const orbfe_frame_view *Cur, const float *scale_factors,
                                              int n_levels, int n_candidates, const uint8_t *const *blocked,
                                              const int32_t *n, const uint8_t *const *valid, const float *const *u,
                                              const float *const *v, const int32_t *const *level,
                                              const float *const *kf_angle, const uint8_t *const *mp_desc,
                                              const float *th, const int32_t *orb_dist, int check_orientation,
                                              int32_t *match_cur, int32_t *n_matches);

/* ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints,
 * vector<MapPoint*> &vpMatched, int th) (src/ORBmatcher.cc:335-449, loop closing) after the caller's
 * Sim3 projection.  matched[idx] != 0 <=> vpMatched[idx] != NULL before the call (NULL = none);
 * match[idx] = point newly matched to keypoint idx or -1. */
int orbfe_search_by_projection_sim3(int device, const orbfe_frame_view *KF, const float *scale_factors,
                                    int n_levels, const uint8_t *matched, int n, const uint8_t *valid,
                                    const float *u, const float *v, const int32_t *level,
                                    const uint8_t *mp_desc, float th, int32_t *match,
                                    int32_t *n_matches);

/* ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched,
 * vector<int> &vnMatches12, int windowSize) (src/ORBmatcher.cc:469-603).  prev_x / prev_y are
 * vbPrevMatched, updated in place like the reference; match12[i1] = i2 or -1. */
int orbfe_search_for_initialization(int device, const orbfe_frame_view *F1, const orbfe_frame_view *F2,
                                    float *prev_x, float *prev_y, int window_size, float nnratio,
                                    int check_orientation, int32_t *match12, int32_t *n_matches);

/* Diagnostics: the projection searches above replay the reference's order-dependent claim loops on the device as a
 * fixed-point iteration (csrc/k_window.hip: k_window_claim); this is the largest number of rounds a claim job of the calling
 * thread's LAST such call took (1 + the longest chain of points that depend on each other; 2-3 on real frames). */
int orbfe_debug_last_claim_rounds(void);

/* The search ORBmatcher::Fuse runs per map point (src/ORBmatcher.cc:940-1110 with chi2_gate != 0,
 * the Sim3 overload :1112-1249 with chi2_gate == 0): best_idx[i] = keypoint of pKF with octave in
 * [level-1, level], passing the reprojection gate (5.99 mono / 7.8 stereo on e2 *
 * inv_level_sigma2[octave]) and the least descriptor distance <= TH_LOW, or -1.  What happens to a hit
 * (Replace / AddObservation) is map bookkeeping and stays with the caller. */
int orbfe_fuse_search(int device, const orbfe_frame_view *KF, const float *scale_factors,
                      const float *inv_level_sigma2, int n_levels, int n, const uint8_t *valid,
                      const float *u, const float *v, const float *ur, const int32_t *level,
                      const uint8_t *mp_desc, float th, int chi2_gate, int32_t *best_idx);

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1251-1482) after the caller's two projections:
 * valid1[i1] <=> keypoint i1 of KF1 has a good map point, not matched yet, that projects into KF2 at
 * (u1, v1) inside the image and its distance range, level1 = PredictScale, desc1 = its descriptor;
 * the "2" arrays are the KF2 -> KF1 direction.  match12[i1] = i2 for mutually consistent pairs. */
int orbfe_search_by_sim3(int device, const orbfe_frame_view *KF1, const orbfe_frame_view *KF2,
                         const float *scale_factors1, const float *scale_factors2, int n_levels,
                         const uint8_t *valid1, const float *u1, const float *v1, const int32_t *level1,
                         const uint8_t *desc1, const uint8_t *valid2, const float *u2, const float *v2,
                         const int32_t *level2, const uint8_t *desc2, float th, int32_t *match12,
                         int32_t *n_found);

/* cv::initUndistortRectifyMap(K, D, R, P.rowRange(0,3).colRange(0,3), cv::Size(cols, rows), CV_32F, M1, M2) as
 * Examples/Stereo/stereo_euroc.cc:97-98 calls it once at start-up: the two CV_32F maps orbfe_rectifier_create takes, from the
 * calibration of the settings file (Examples/Stereo/EuRoC.yaml: LEFT.K / LEFT.D / LEFT.R / LEFT.P).  K, R, P: row-major 3 x 3
 * doubles (R NULL = identity, P NULL = K; pass the first three columns of a 3 x 4 projection); D: n_dist = 0, 4, 5 or 8
 * coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6]]); map_x / map_y: width * height floats each, written on the host.  Evaluated on the
 * device, one thread per map row (a row is the reference's sequential running-sum chain in double precision). */
int orbfe_init_undistort_rectify_map(int device, const double *K, const double *D, int n_dist, const double *R,
                                     const double *P, int width, int height, float *map_x, float *map_y);

/* cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) with CV_32F maps and the default constant-0
 * border, 8-bit single channel: the EuRoC rectification of Examples/Stereo/stereo_euroc.cc:136-137
 * (the maps come from cv::initUndistortRectifyMap at :97-98, once, and are handed over here once).
 * Destination size = map size (width x height, map_stride floats per row). */
typedef struct orbfe_rectifier orbfe_rectifier;
int orbfe_rectifier_create(int device, const float *map_x, const float *map_y, int width, int height,
                       int map_stride, orbfe_rectifier **out);
void orbfe_rectifier_destroy(orbfe_rectifier *r);
/* Host buffers, one frame. */
int orbfe_remap(orbfe_rectifier *r, const uint8_t *src, int src_width, int src_height, int src_stride,
                uint8_t *dst, int dst_stride);
/* Device-resident frames: rectified images are written where orbfe_extract_batch_device reads them. */
int orbfe_remap_batch_device(orbfe_rectifier *r, const uint8_t *d_src, int n_frames, int src_width,
                             int src_height, int src_stride, size_t src_frame_stride, uint8_t *d_dst,
                             int dst_stride, size_t dst_frame_stride);

/* The stereo front end of Examples/Stereo/stereo_euroc.cc for a device-resident batch of RAW pairs: cv::remap of the
 * left and of the right image (:136-137) and the two ExtractORB calls of the stereo Frame constructor
 * (src/Frame.cc:78-81), enqueued sub-batch by sub-batch on extractor e's streams (returns at once, like
 * orbfe_extract_batch_device_async; no host wait between rectification and extraction).  Raw frame p of either eye at
 * d_raw_*[p*src_frame_stride ..]; rectified frames are written tightly packed and interleaved to d_rectified
 * (2*n_pairs frames of width x height of the rectifiers: L0, R0, L1, R1, ...), which is also where the extractor's
 * level 0 -- and orbfe_stereo_match_batch_device's SAD windows -- are read from, so it must stay valid until the
 * matchers of this batch are done.  Outputs as orbfe_extract_batch_device for 2*n_pairs frames. */
int orbfe_extract_stereo_rectified_batch_device_async(
    orbfe_extractor *e, orbfe_rectifier *rect_left, orbfe_rectifier *rect_right, const uint8_t *d_raw_left,
    const uint8_t *d_raw_right, int n_pairs, int src_width, int src_height, int src_stride, size_t src_frame_stride,
    uint8_t *d_rectified, orbfe_keypoint *d_keypoints, uint8_t *d_descriptors, int capacity, int32_t *d_n_out);

/* cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) as Frame::UndistortKeyPoints and
 * Frame::ComputeImageBounds call it (src/Frame.cc:443-475, 481-510).  K4 = fx, fy, cx, cy (the
 * CV_32F entries of mK); dist = k1, k2, p1, p2[, k3[, k4, k5, k6]] (n_dist 0, 4, 5 or 8); xy /
 * out_xy are n (x, y) float pairs. */
int orbfe_undistort_points(int device, const float *xy, int n, const float *K4, const float *dist,
                           int n_dist, float *out_xy);
/* mvKeysUn of a device-resident extractor batch (the layout orbfe_extract_batch_device writes):
 * records copied, pt undistorted. */
int orbfe_undistort_keypoints_batch_device(int device, const orbfe_keypoint *d_keypoints,
                                           const int32_t *d_n, int n_frames, int capacity,
                                           const float *K4, const float *dist, int n_dist,
                                           orbfe_keypoint *d_keypoints_un);
/* Frame::ComputeImageBounds (src/Frame.cc:481-510): bounds4 = mnMinX, mnMaxX, mnMinY, mnMaxY. */
int orbfe_compute_image_bounds(int device, int cols, int rows, const float *K4, const float *dist,
                               int n_dist, float *bounds4);
/* Frame::ComputeStereoFromRGBD (src/Frame.cc:689-713): kx / ky = mvKeys[i].pt, kux = mvKeysUn[i].pt.x,
 * depth_image = imDepth (CV_32F, after the mDepthMapFactor scaling of src/Tracking.cc:226-227). */
int orbfe_stereo_from_rgbd(int device, const float *kx, const float *ky, const float *kux, int n,
                           const float *depth_image, int width, int height, int stride_floats,
                           float mbf, float *u_right, float *depth);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
