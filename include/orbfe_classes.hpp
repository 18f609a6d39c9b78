// orbfe_classes.hpp -- header-only C++ host layer over the C-ABI (orbfe.h), OpenCV-free.
//
// Mirrors the reference's operator/plugin interface for the hot path with the same names,
// argument meaning and error behaviour:
//   ORB_SLAM2::ORBextractor  include/ORBextractor.h:46-114   (reference paths)
//   ORB_SLAM2::ORBmatcher    include/ORBmatcher.h:55-103     (all 11 methods + DescriptorDistance)
//   Frame::ComputeStereoMatches  src/Frame.cc:512-686
// cv::KeyPoint / cv::Mat / KeyFrame* / MapPoint* are replaced by layout-compatible PODs and flat arrays so
// that tests and tools build without OpenCV.  include/ORBextractor.h (extractor) and include/ORBmatcher.h +
// src/ORBmatcher_orbfe.cc (matcher) of this repo wrap these classes once more with the reference's exact
// signatures for a build that has OpenCV and the reference's own headers; those two are not compiled here.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "orbfe.h"

namespace orbfe_cpp {

struct KeyPoint {  // == cv::KeyPoint, 28 bytes
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == sizeof(orbfe_keypoint), "layout");

struct Image {  // minimal stand-in for a CV_8UC1 cv::Mat
  int cols = 0, rows = 0;
  std::vector<uint8_t> data;
  const uint8_t* ptr(int y) const { return data.data() + (size_t)y * cols; }
};

inline void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + orbfe_last_error());
}

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0) {
    check(orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_), "ORBextractor");
  }
  ~ORBextractor() { orbfe_extractor_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // operator()(InputArray image, InputArray mask /*ignored*/, vector<KeyPoint>&, OutputArray descriptors)
  // Empty image: returns silently leaving the outputs untouched (src/ORBextractor.cc:1122-1123);
  // zero keypoints: descriptors released (:1145-1146).
  void operator()(const uint8_t* image, int width, int height, int stride, std::vector<KeyPoint>& keypoints,
                  std::vector<uint8_t>& descriptors) {
    if (!image || width <= 0 || height <= 0) return;
    int cap = orbfe_extractor_max_keypoints_for(h_, width, height);  // exact bound for this image size
    if (cap < orbfe_extractor_max_keypoints(h_)) cap = orbfe_extractor_max_keypoints(h_);
    keypoints.resize(cap);
    descriptors.resize((size_t)cap * 32);
    int n = 0;
    check(orbfe_extract(h_, image, width, height, stride, reinterpret_cast<orbfe_keypoint*>(keypoints.data()),
                        descriptors.data(), cap, &n), "ORBextractor::operator()");
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    w_ = width;
    h0_ = height;
    pyramidValid_ = false;
  }

  // The front end of the stereo Frame constructor in one call (src/Frame.cc:78-96: ExtractORB(0, imLeft) and
  // ExtractORB(1, imRight) on two threads, join, ComputeStereoMatches): both eyes through THIS handle, the stereo matcher on
  // the records still in HBM, one download.  mvuRight / mvDepth get keysLeft.size() entries (-1 = no stereo).
  void extractStereoFrame(const uint8_t* left, const uint8_t* right, int width, int height, int stride, float mbf, float mb,
                          std::vector<KeyPoint>& keysLeft, std::vector<uint8_t>& descLeft, std::vector<KeyPoint>& keysRight,
                          std::vector<uint8_t>& descRight, std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    if (!left || !right || width <= 0 || height <= 0) return;
    int cap = orbfe_extractor_max_keypoints_for(h_, width, height);
    if (cap < orbfe_extractor_max_keypoints(h_)) cap = orbfe_extractor_max_keypoints(h_);
    keysLeft.resize(cap); keysRight.resize(cap);
    descLeft.resize((size_t)cap * 32); descRight.resize((size_t)cap * 32);
    mvuRight.assign(cap, -1.0f); mvDepth.assign(cap, -1.0f);
    int nl = 0, nr = 0;
    check(orbfe_extract_stereo_frame(h_, left, right, width, height, stride, reinterpret_cast<orbfe_keypoint*>(keysLeft.data()),
                                     descLeft.data(), &nl, reinterpret_cast<orbfe_keypoint*>(keysRight.data()), descRight.data(),
                                     &nr, cap, mbf, mb, mvuRight.data(), mvDepth.data()), "ORBextractor::extractStereoFrame");
    keysLeft.resize(nl); keysRight.resize(nr);
    descLeft.resize((size_t)nl * 32); descRight.resize((size_t)nr * 32);
    mvuRight.resize(nl); mvDepth.resize(nl);
    w_ = width;
    h0_ = height;
    pyramidValid_ = false;
  }

  int GetLevels() { return orbfe_extractor_get_levels(h_); }
  float GetScaleFactor() { return orbfe_extractor_get_scale_factor(h_); }
  std::vector<float> GetScaleFactors() { return vec(orbfe_extractor_get_scale_factors); }
  std::vector<float> GetInverseScaleFactors() { return vec(orbfe_extractor_get_inverse_scale_factors); }
  std::vector<float> GetScaleSigmaSquares() { return vec(orbfe_extractor_get_scale_sigma_squares); }
  std::vector<float> GetInverseScaleSigmaSquares() { return vec(orbfe_extractor_get_inverse_scale_sigma_squares); }

  // The reference exposes `std::vector<cv::Mat> mvImagePyramid` (include/ORBextractor.h:86) that
  // Frame::ComputeStereoMatches reads.  Here the pyramid stays in HBM; this accessor copies it to
  // the host on first use after an extraction (the device stereo matcher never needs it).
  const std::vector<Image>& mvImagePyramid() {
    if (!pyramidValid_) {
      const int nl = GetLevels();
      pyr_.assign(nl, Image());
      for (int l = 0; l < nl && w_ > 0; l++) {
        Image& im = pyr_[l];
        check(orbfe_extractor_level_size(h_, w_, h0_, l, &im.cols, &im.rows), "level_size");
        im.data.resize((size_t)im.cols * im.rows);
        check(orbfe_extractor_get_pyramid_level(h_, 0, l, im.data.data(), im.cols), "get_pyramid_level");
      }
      pyramidValid_ = true;
    }
    return pyr_;
  }

  orbfe_extractor* handle() { return h_; }

 private:
  template <typename F>
  std::vector<float> vec(F f) {
    std::vector<float> v(GetLevels());
    check(f(h_, v.data()), "getter");
    return v;
  }
  orbfe_extractor* h_ = nullptr;
  int w_ = 0, h0_ = 0;
  bool pyramidValid_ = false;
  std::vector<Image> pyr_;
};

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) -> CSR view for the C-ABI
struct FeatureVectorCSR {
  std::vector<uint32_t> node_ids, indices;
  std::vector<int32_t> offsets;
  orbfe_featvec c{};
  template <typename Map>
  explicit FeatureVectorCSR(const Map& fv) {
    offsets.push_back(0);
    for (const auto& kv : fv) {
      node_ids.push_back((uint32_t)kv.first);
      for (unsigned i : kv.second) indices.push_back(i);
      offsets.push_back((int32_t)indices.size());
    }
    c.n_nodes = (int32_t)node_ids.size();
    c.node_ids = node_ids.data();
    c.offsets = offsets.data();
    c.indices = indices.data();
  }
};

// The arrays of a Frame / KeyFrame the projection searches read (include/Frame.h:120-190), split out of
// std::vector<cv::KeyPoint> mvKeysUn once per frame; owns the storage behind an orbfe_frame_view.
class FrameArrays {
 public:
  // bounds = mnMinX, mnMaxX, mnMinY, mnMaxY (src/Frame.cc:697-728); mvuRight may be empty (monocular)
  FrameArrays(const std::vector<KeyPoint>& mvKeysUn, const std::vector<uint8_t>& mDescriptors, float mnMinX,
              float mnMaxX, float mnMinY, float mnMaxY, const std::vector<float>& mvuRight = std::vector<float>())
      : desc(mDescriptors), uRight(mvuRight) {
    const size_t n = mvKeysUn.size();
    x.resize(n); y.resize(n); angle.resize(n); octave.resize(n);
    for (size_t i = 0; i < n; i++) {
      x[i] = mvKeysUn[i].x; y[i] = mvKeysUn[i].y; angle[i] = mvKeysUn[i].angle; octave[i] = mvKeysUn[i].octave;
    }
    c.n = (int32_t)n;
    c.x = x.data(); c.y = y.data(); c.octave = octave.data(); c.angle = angle.data();
    c.u_right = uRight.empty() ? nullptr : uRight.data();
    c.desc = desc.data();
    c.min_x = mnMinX; c.max_x = mnMaxX; c.min_y = mnMinY; c.max_y = mnMaxY;
    c.resident = nullptr;
  }
  ~FrameArrays() { if (resident_) orbfe_frame_release(resident_); }
  FrameArrays(const FrameArrays&) = delete;
  FrameArrays& operator=(const FrameArrays&) = delete;
  int N() const { return c.n; }

  // Move this frame's operands to the device ONCE (keypoint arrays, descriptors, the 64 x 48 grid, and -- for the
  // FeatureVector searches -- mFeatVec's index list): every later search on this object uploads nothing of the frame.
  // Call it where the reference fills the object for good: at the end of the Frame constructor / in
  // KeyFrame::ComputeBoW (src/KeyFrame.cc:64-73).  A KeyFrame is matched against 10-20 neighbours per insertion.
  void makeResident(const FeatureVectorCSR* mFeatVec = nullptr, int device = 0) {
    if (resident_) { orbfe_frame_release(resident_); resident_ = nullptr; c.resident = nullptr; }
    check(orbfe_frame_upload(device, &c, mFeatVec ? &mFeatVec->c : nullptr, &resident_), "FrameArrays::makeResident");
    c.resident = resident_;
  }
  // The same from the extractor's OWN device records -- Frame::Frame is ExtractORB -> UndistortKeyPoints ->
  // ComputeStereoMatches -> AssignFeaturesToGrid (src/Frame.cc:61-117): the keypoints and descriptors operator() just
  // returned are still in HBM, so only mvuRight (and, for cameras with distortion, the undistorted positions:
  // undistorted = true) travel.  `e` = the extractor whose last operator() produced this frame's keypoints.
  void makeResidentFromExtractor(const orbfe_extractor* e, bool undistorted = false, const FeatureVectorCSR* mFeatVec = nullptr) {
    if (resident_) { orbfe_frame_release(resident_); resident_ = nullptr; c.resident = nullptr; }
    check(orbfe_frame_from_extractor(const_cast<orbfe_extractor*>(e), 0, &c, mFeatVec ? &mFeatVec->c : nullptr,
                                     undistorted ? ORBFE_FRAME_XY_FROM_VIEW : 0, &resident_), "FrameArrays::makeResidentFromExtractor");
    c.resident = resident_;
  }
  // Frame::ComputeBoW (src/Frame.cc:433-440) runs after the constructor: attach mFeatVec to the resident frame
  void setFeatVec(const FeatureVectorCSR& mFeatVec) { check(orbfe_frame_set_featvec(resident_, &mFeatVec.c), "FrameArrays::setFeatVec"); }
  const orbfe_frame* resident() const { return resident_; }

  // vector<size_t> Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) const
  std::vector<size_t> GetFeaturesInArea(float qx, float qy, float r, int minLevel = -1, int maxLevel = -1,
                                        int device = 0) const {
    int32_t count = 0;
    std::vector<int32_t> idx(64);
    for (;;) {
      const int rc = orbfe_features_in_area(device, &c, 1, &qx, &qy, &r, &minLevel, &maxLevel, (int)idx.size(), &count,
                                            idx.data());
      if (rc == ORBFE_ERR_CAPACITY) { idx.resize(count); continue; }
      check(rc, "GetFeaturesInArea");
      break;
    }
    return std::vector<size_t>(idx.begin(), idx.begin() + count);
  }

  std::vector<float> x, y, angle;
  std::vector<int32_t> octave;
  std::vector<uint8_t> desc;
  std::vector<float> uRight;
  orbfe_frame_view c;

 private:
  orbfe_frame* resident_ = nullptr;
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50;
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

  ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0)
      : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}

  // static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b, int device = 0) {
    int32_t d = 0;
    check(orbfe_descriptor_distance(device, a, b, 1, &d), "DescriptorDistance");
    return d;
  }

  // int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches):
  // matchF[j] = KF feature whose MapPoint goes to frame feature j (or -1).
  int SearchByBoW(const uint8_t* descKF, const uint8_t* hasMapPointKF, const float* angleKF, int nKF,
                  const FeatureVectorCSR& fvKF, const uint8_t* descF, const float* angleF, int nF,
                  const FeatureVectorCSR& fvF, std::vector<int32_t>& matchF) {
    matchF.assign(nF, -1);
    int rc = orbfe_search_by_bow(device_, descKF, hasMapPointKF, angleKF, nKF, &fvKF.c, descF, angleF, nF, &fvF.c,
                                 mfNNratio, mbCheckOrientation, matchF.data());
    check(rc, "SearchByBoW");
    return rc;
  }
  // int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)
  int SearchByBoW(const uint8_t* desc1, const uint8_t* hasMp1, const float* angle1, int n1,
                  const FeatureVectorCSR& fv1, const uint8_t* desc2, const uint8_t* hasMp2, const float* angle2,
                  int n2, const FeatureVectorCSR& fv2, std::vector<int32_t>& match12) {
    match12.assign(n1, -1);
    int rc = orbfe_search_by_bow_kf(device_, desc1, hasMp1, angle1, n1, &fv1.c, desc2, hasMp2, angle2, n2, &fv2.c,
                                    mfNNratio, mbCheckOrientation, match12.data());
    check(rc, "SearchByBoW(KF,KF)");
    return rc;
  }

  // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th = 3):
  // the per-point fields are what Frame::isInFrustum stored on each MapPoint; match[idx] = point or -1.
  int SearchByProjection(const FrameArrays& F, const std::vector<float>& mvScaleFactors,
                         const std::vector<uint8_t>& mbTrackInView, const std::vector<int32_t>& mnTrackScaleLevel,
                         const std::vector<float>& mTrackViewCos, const std::vector<float>& mTrackProjX,
                         const std::vector<float>& mTrackProjY, const std::vector<float>& mTrackProjXR,
                         const std::vector<uint8_t>& mpDescriptors, const std::vector<uint8_t>& frameHasObservedPoint,
                         std::vector<int32_t>& match, float th = 3.0f) {
    match.assign(F.N(), -1);
    int32_t n = 0;
    check(orbfe_search_by_projection(device_, &F.c, mvScaleFactors.data(), (int)mvScaleFactors.size(),
                                     frameHasObservedPoint.empty() ? nullptr : frameHasObservedPoint.data(),
                                     (int)mbTrackInView.size(), mbTrackInView.data(), mnTrackScaleLevel.data(),
                                     mTrackViewCos.data(), mTrackProjX.data(), mTrackProjY.data(),
                                     mTrackProjXR.empty() ? nullptr : mTrackProjXR.data(), mpDescriptors.data(), nullptr,
                                     th, mfNNratio, match.data(), &n),
          "SearchByProjection(Frame,MapPoints)");
    return n;
  }

  // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
  // after the projection loop prologue (src/ORBmatcher.cc:1525-1541); mode 0/1/2 = normal/bForward/bBackward.
  int SearchByProjection(const FrameArrays& CurrentFrame, const std::vector<float>& mvScaleFactors, float mbf,
                         const std::vector<uint8_t>& valid, const std::vector<float>& u, const std::vector<float>& v,
                         const std::vector<float>& invzc, const std::vector<int32_t>& lastOctave,
                         const std::vector<float>& lastAngle, const std::vector<uint8_t>& mpDescriptors, int mode,
                         float th, std::vector<int32_t>& matchCur,
                         const std::vector<uint8_t>& obsPositive = std::vector<uint8_t>(),
                         const std::vector<uint8_t>& curBlocked = std::vector<uint8_t>()) {
    matchCur.assign(CurrentFrame.N(), -1);
    int32_t n = 0;
    check(orbfe_search_by_projection_last_frame(device_, &CurrentFrame.c, mvScaleFactors.data(),
                                                (int)mvScaleFactors.size(), mbf, (int)valid.size(), valid.data(),
                                                u.data(), v.data(), invzc.empty() ? nullptr : invzc.data(),
                                                lastOctave.data(), lastAngle.data(), mpDescriptors.data(),
                                                obsPositive.empty() ? nullptr : obsPositive.data(),
                                                curBlocked.empty() ? nullptr : curBlocked.data(), mode,
                                                th, mbCheckOrientation, matchCur.data(), &n),
          "SearchByProjection(Frame,Frame)");
    return n;
  }

  // int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
  //                             vector<int>& vnMatches12, int windowSize = 10)
  int SearchForInitialization(const FrameArrays& F1, const FrameArrays& F2, std::vector<float>& prevX,
                              std::vector<float>& prevY, std::vector<int32_t>& vnMatches12, int windowSize = 10) {
    vnMatches12.assign(F1.N(), -1);
    int32_t n = 0;
    check(orbfe_search_for_initialization(device_, &F1.c, &F2.c, prevX.data(), prevY.data(), windowSize, mfNNratio,
                                          mbCheckOrientation, vnMatches12.data(), &n),
          "SearchForInitialization");
    return n;
  }

  // int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
  //                            vector<pair<size_t,size_t>>& vMatchedPairs, const bool bOnlyStereo)
  // (src/ORBmatcher.cc:754-928).  KF1 / KF2 carry mvKeysUn + mvuRight; hasMp = "GetMapPoint(idx) != NULL";
  // F12 row-major 3x3; (ex, ey) = the epipole of :766-769; mvScaleFactors2 / mvLevelSigma2_2 of pKF2.
  int SearchForTriangulation(const FrameArrays& KF1, const std::vector<uint8_t>& hasMp1, const FeatureVectorCSR& fv1,
                             const FrameArrays& KF2, const std::vector<uint8_t>& hasMp2, const FeatureVectorCSR& fv2,
                             const float F12[9], float ex, float ey, const std::vector<float>& mvScaleFactors2,
                             const std::vector<float>& mvLevelSigma2_2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                             bool bOnlyStereo) {
    const int n1 = KF1.N(), n2 = KF2.N();
    std::vector<uint8_t> st1(n1, 0), st2(n2, 0);  // bStereo1 = mvuRight[idx] >= 0 (:806, :830)
    for (int i = 0; i < n1 && !KF1.uRight.empty(); i++) st1[i] = KF1.uRight[i] >= 0;
    for (int i = 0; i < n2 && !KF2.uRight.empty(); i++) st2[i] = KF2.uRight[i] >= 0;
    std::vector<int32_t> match12(n1 > 0 ? n1 : 1, -1);
    const int rc = orbfe_search_for_triangulation(
        device_, KF1.desc.data(), hasMp1.data(), KF1.x.data(), KF1.y.data(), KF1.angle.data(), st1.data(), n1, &fv1.c,
        KF2.desc.data(), hasMp2.data(), KF2.x.data(), KF2.y.data(), KF2.angle.data(), KF2.octave.data(), st2.data(), n2,
        &fv2.c, F12, ex, ey, mvScaleFactors2.data(), mvLevelSigma2_2.data(), (int)mvScaleFactors2.size(), bOnlyStereo,
        mbCheckOrientation, match12.data());
    check(rc, "SearchForTriangulation");
    vMatchedPairs.clear();
    vMatchedPairs.reserve(rc);
    for (int i = 0; i < n1; i++)  // :920-925: ascending idx1
      if (match12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match12[i]));
    return rc;
  }

  // The loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:283-315) in ONE call: SearchForTriangulation of
  // pKF1 against every neighbour.  All frames resident (makeResident with their mFeatVec).  F12s[k] / epipoles[k] as the
  // single-pair form takes them; vMatchedPairs[k] = the pairs of neighbour k (ascending idx1).
  void SearchForTriangulationMulti(const FrameArrays& KF1, const std::vector<uint8_t>& hasMp1,
                                   const std::vector<const FrameArrays*>& neighbours,
                                   const std::vector<const std::vector<uint8_t>*>& hasMp2, const std::vector<float>& F12s /* 9 per neighbour */,
                                   const std::vector<float>& ex, const std::vector<float>& ey,
                                   const std::vector<float>& mvScaleFactors2, const std::vector<float>& mvLevelSigma2_2,
                                   std::vector<std::vector<std::pair<size_t, size_t> > >& vMatchedPairs, bool bOnlyStereo) {
    const int K = (int)neighbours.size(), n1 = KF1.N();
    std::vector<const orbfe_frame*> fr(K > 0 ? K : 1, nullptr);
    std::vector<const uint8_t*> mk(K > 0 ? K : 1, nullptr);
    for (int k = 0; k < K; k++) { fr[k] = neighbours[k]->resident(); mk[k] = hasMp2[k]->data(); }
    std::vector<int32_t> match((size_t)(K > 0 ? K : 1) * (n1 > 0 ? n1 : 1), -1), cnt(K > 0 ? K : 1, 0);
    check(orbfe_search_for_triangulation_multi(KF1.resident(), hasMp1.data(), K, fr.data(), mk.data(), F12s.data(), ex.data(),
                                               ey.data(), mvScaleFactors2.data(), mvLevelSigma2_2.data(),
                                               (int)mvScaleFactors2.size(), bOnlyStereo, mbCheckOrientation, match.data(), cnt.data()),
          "SearchForTriangulationMulti");
    vMatchedPairs.assign(K, std::vector<std::pair<size_t, size_t> >());
    for (int k = 0; k < K; k++)
      for (int i = 0; i < n1; i++)
        if (match[(size_t)k * n1 + i] >= 0) vMatchedPairs[k].push_back(std::make_pair((size_t)i, (size_t)match[(size_t)k * n1 + i]));
  }

  // The loop of Tracking::Relocalization (src/Tracking.cc:1478-1498) in ONE call: SearchByBoW(pKF_k, mCurrentFrame,
  // vvpMapPointMatches[k]) for every candidate key frame.  All frames resident with their mFeatVec.  vnMatches[k][i2] =
  // feature of key frame k whose MapPoint goes to feature i2 of the frame, or -1; returns the counts.
  std::vector<int> SearchByBoWMulti(const std::vector<const FrameArrays*>& vpCandidateKFs,
                                    const std::vector<const std::vector<uint8_t>*>& hasMp, const FrameArrays& F,
                                    std::vector<std::vector<int32_t> >& vnMatches) {
    const int K = (int)vpCandidateKFs.size(), n = F.N();
    std::vector<const orbfe_frame*> fr(K > 0 ? K : 1, nullptr);
    std::vector<const uint8_t*> mk(K > 0 ? K : 1, nullptr);
    for (int k = 0; k < K; k++) { fr[k] = vpCandidateKFs[k]->resident(); mk[k] = hasMp[k]->data(); }
    std::vector<int32_t> match((size_t)(K > 0 ? K : 1) * (n > 0 ? n : 1), -1), cnt(K > 0 ? K : 1, 0);
    check(orbfe_search_by_bow_multi(K, fr.data(), mk.data(), F.resident(), mfNNratio, mbCheckOrientation, match.data(), cnt.data()),
          "SearchByBoWMulti");
    vnMatches.assign(K, std::vector<int32_t>());
    for (int k = 0; k < K; k++) vnMatches[k].assign(match.begin() + (size_t)k * n, match.begin() + (size_t)(k + 1) * n);
    return std::vector<int>(cnt.begin(), cnt.begin() + K);
  }
  // The loop of LoopClosing::ComputeSim3 (src/LoopClosing.cc:294-321): SearchByBoW(mpCurrentKF, pKF_k, vvpMapPointMatches[k]).
  // vnMatches12[k][i1] = feature of candidate k matched to feature i1 of the current key frame, or -1.
  std::vector<int> SearchByBoWMulti(const FrameArrays& KF1, const std::vector<uint8_t>& hasMp1,
                                    const std::vector<const FrameArrays*>& vpCandidateKFs,
                                    const std::vector<const std::vector<uint8_t>*>& hasMp2,
                                    std::vector<std::vector<int32_t> >& vnMatches12) {
    const int K = (int)vpCandidateKFs.size(), n = KF1.N();
    std::vector<const orbfe_frame*> fr(K > 0 ? K : 1, nullptr);
    std::vector<const uint8_t*> mk(K > 0 ? K : 1, nullptr);
    for (int k = 0; k < K; k++) { fr[k] = vpCandidateKFs[k]->resident(); mk[k] = hasMp2[k]->data(); }
    std::vector<int32_t> match((size_t)(K > 0 ? K : 1) * (n > 0 ? n : 1), -1), cnt(K > 0 ? K : 1, 0);
    check(orbfe_search_by_bow_kf_multi(KF1.resident(), hasMp1.data(), K, fr.data(), mk.data(), mfNNratio, mbCheckOrientation,
                                       match.data(), cnt.data()),
          "SearchByBoWMulti(KF,KF)");
    vnMatches12.assign(K, std::vector<int32_t>());
    for (int k = 0; k < K; k++) vnMatches12[k].assign(match.begin() + (size_t)k * n, match.begin() + (size_t)(k + 1) * n);
    return std::vector<int>(cnt.begin(), cnt.begin() + K);
  }

  // SearchByProjection(mCurrentFrame, vpCandidateKFs[k], sFound, th_k, ORBdist_k) (src/Tracking.cc:1577,1595) for several
  // candidates in ONE call; one ProjectedKeyFrame per candidate = the caller's projection prologue (:1657-1700).
  struct ProjectedKeyFrame {
    std::vector<uint8_t> valid, mpDescriptors, curHasMapPoint /* may be empty */;
    std::vector<float> u, v, kfAngle;
    std::vector<int32_t> level;
    float th; int ORBdist;
  };
  std::vector<int> SearchByProjectionMulti(const FrameArrays& CurrentFrame, const std::vector<float>& mvScaleFactors,
                                           const std::vector<ProjectedKeyFrame>& cands, std::vector<std::vector<int32_t> >& matchCur) {
    const int K = (int)cands.size(), n = CurrentFrame.N(), K1 = K > 0 ? K : 1;
    std::vector<const uint8_t*> blk(K1, nullptr), va(K1, nullptr), md(K1, nullptr);
    std::vector<const float*> u(K1, nullptr), v(K1, nullptr), ka(K1, nullptr);
    std::vector<const int32_t*> lv(K1, nullptr);
    std::vector<int32_t> nk(K1, 0), od(K1, 0), match((size_t)K1 * (n > 0 ? n : 1), -1), cnt(K1, 0);
    std::vector<float> th(K1, 0.f);
    for (int k = 0; k < K; k++) {
      const ProjectedKeyFrame& c = cands[k];
      blk[k] = c.curHasMapPoint.empty() ? nullptr : c.curHasMapPoint.data();
      va[k] = c.valid.data(); md[k] = c.mpDescriptors.data(); u[k] = c.u.data(); v[k] = c.v.data(); ka[k] = c.kfAngle.data();
      lv[k] = c.level.data(); nk[k] = (int32_t)c.valid.size(); od[k] = c.ORBdist; th[k] = c.th;
    }
    check(orbfe_search_by_projection_keyframe_multi(device_, &CurrentFrame.c, mvScaleFactors.data(), (int)mvScaleFactors.size(), K,
                                                    blk.data(), nk.data(), va.data(), u.data(), v.data(), lv.data(), ka.data(), md.data(),
                                                    th.data(), od.data(), mbCheckOrientation, match.data(), cnt.data()),
          "SearchByProjectionMulti");
    matchCur.assign(K, std::vector<int32_t>());
    for (int k = 0; k < K; k++) matchCur[k].assign(match.begin() + (size_t)k * n, match.begin() + (size_t)(k + 1) * n);
    return std::vector<int>(cnt.begin(), cnt.begin() + K);
  }

  // The per-point search of Fuse for the SAME map points against K key frames in one call (LocalMapping::SearchInNeighbors,
  // src/LocalMapping.cc:542-549): per-key-frame arrays are [k * n + i] (the caller's projection prologue per key frame,
  // src/ORBmatcher.cc:960-1020); bestIdx[k * n + i] = keypoint of key frame k chosen for point i, or -1.
  void FuseSearchMulti(const std::vector<const FrameArrays*>& KFs, const std::vector<float>& mvScaleFactors,
                       const std::vector<float>& mvInvLevelSigma2, int nPoints, const std::vector<uint8_t>& valid,
                       const std::vector<float>& u, const std::vector<float>& v, const std::vector<float>& ur,
                       const std::vector<int32_t>& level, const std::vector<uint8_t>& mpDescriptors, float th,
                       std::vector<int32_t>& bestIdx) {
    const int K = (int)KFs.size();
    std::vector<const orbfe_frame_view*> views(K > 0 ? K : 1, nullptr);
    for (int k = 0; k < K; k++) views[k] = &KFs[k]->c;
    bestIdx.assign((size_t)K * nPoints, -1);
    check(orbfe_fuse_search_multi(device_, K, views.data(), mvScaleFactors.data(), mvInvLevelSigma2.data(),
                                  (int)mvScaleFactors.size(), nPoints, valid.data(), u.data(), v.data(),
                                  ur.empty() ? nullptr : ur.data(), level.data(), mpDescriptors.data(), th, 1, bestIdx.data()),
          "FuseSearchMulti");
  }

  // int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th,
  //                        const int ORBdist) (src/ORBmatcher.cc:1641-1775, relocalisation) after the caller's
  // projection prologue (:1657-1700): valid[i] = pKF's map point i exists, is good, not in sAlreadyFound, projects
  // inside the image and its distance range; level = PredictScale; kfAngle = pKF->mvKeysUn[i].angle;
  // curHasMapPoint[i2] = CurrentFrame.mvpMapPoints[i2] != NULL (may be empty).  matchCur[i2] = index i or -1.
  int SearchByProjection(const FrameArrays& CurrentFrame, const std::vector<float>& mvScaleFactors,
                         const std::vector<uint8_t>& curHasMapPoint, const std::vector<uint8_t>& valid,
                         const std::vector<float>& u, const std::vector<float>& v, const std::vector<int32_t>& level,
                         const std::vector<float>& kfAngle, const std::vector<uint8_t>& mpDescriptors, float th,
                         int ORBdist, std::vector<int32_t>& matchCur) {
    matchCur.assign(CurrentFrame.N(), -1);
    int32_t n = 0;
    check(orbfe_search_by_projection_keyframe(device_, &CurrentFrame.c, mvScaleFactors.data(), (int)mvScaleFactors.size(),
                                              curHasMapPoint.empty() ? nullptr : curHasMapPoint.data(), (int)valid.size(),
                                              valid.data(), u.data(), v.data(), level.data(), kfAngle.data(),
                                              mpDescriptors.data(), th, ORBdist, mbCheckOrientation, matchCur.data(), &n),
          "SearchByProjection(Frame,KeyFrame)");
    return n;
  }

  // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
  //                        vector<MapPoint*>& vpMatched, int th) (src/ORBmatcher.cc:335-449, loop closing) after the
  // caller's Sim3 projection (:355-400): valid[i] = point i is good, not already in vpMatched, in front of the
  // camera, inside the image, its distance range and viewing cone.  alreadyMatched[idx] = vpMatched[idx] != NULL
  // (may be empty).  match[idx] = point newly matched to keypoint idx, or -1.
  int SearchByProjection(const FrameArrays& KF, const std::vector<float>& mvScaleFactors,
                         const std::vector<uint8_t>& alreadyMatched, const std::vector<uint8_t>& valid,
                         const std::vector<float>& u, const std::vector<float>& v, const std::vector<int32_t>& level,
                         const std::vector<uint8_t>& mpDescriptors, int th, std::vector<int32_t>& match) {
    match.assign(KF.N(), -1);
    int32_t n = 0;
    check(orbfe_search_by_projection_sim3(device_, &KF.c, mvScaleFactors.data(), (int)mvScaleFactors.size(),
                                          alreadyMatched.empty() ? nullptr : alreadyMatched.data(), (int)valid.size(),
                                          valid.data(), u.data(), v.data(), level.data(), mpDescriptors.data(), (float)th,
                                          match.data(), &n),
          "SearchByProjection(KeyFrame,Scw)");
    return n;
  }

  // The search inside int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th = 3.0)
  // (src/ORBmatcher.cc:940-1110): per map point, after the caller's projection (:960-1020, valid / u / v / ur /
  // PredictScale level), bestIdx[i] = the keypoint of pKF to fuse with, or -1.  Replace / AddObservation on a hit
  // (:1088-1106) is map bookkeeping and stays with the caller.  ur is read for stereo keypoints only.
  void Fuse(const FrameArrays& KF, const std::vector<float>& mvScaleFactors, const std::vector<float>& mvInvLevelSigma2,
            const std::vector<uint8_t>& valid, const std::vector<float>& u, const std::vector<float>& v,
            const std::vector<float>& ur, const std::vector<int32_t>& level, const std::vector<uint8_t>& mpDescriptors,
            std::vector<int32_t>& bestIdx, float th = 3.0f) {
    bestIdx.assign(valid.size(), -1);
    check(orbfe_fuse_search(device_, &KF.c, mvScaleFactors.data(), mvInvLevelSigma2.data(), (int)mvScaleFactors.size(),
                            (int)valid.size(), valid.data(), u.data(), v.data(), ur.empty() ? nullptr : ur.data(),
                            level.data(), mpDescriptors.data(), th, 1, bestIdx.data()),
          "Fuse");
  }
  // The search inside int Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th,
  //                            vector<MapPoint*>& vpReplacePoint) (src/ORBmatcher.cc:1112-1249): no chi2 gate.
  void Fuse(const FrameArrays& KF, const std::vector<float>& mvScaleFactors, const std::vector<uint8_t>& valid,
            const std::vector<float>& u, const std::vector<float>& v, const std::vector<int32_t>& level,
            const std::vector<uint8_t>& mpDescriptors, float th, std::vector<int32_t>& bestIdx) {
    bestIdx.assign(valid.size(), -1);
    check(orbfe_fuse_search(device_, &KF.c, mvScaleFactors.data(), nullptr, (int)mvScaleFactors.size(),
                            (int)valid.size(), valid.data(), u.data(), v.data(), nullptr, level.data(),
                            mpDescriptors.data(), th, 0, bestIdx.data()),
          "Fuse(Scw)");
  }

  // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const float& s12,
  //                  const cv::Mat& R12, const cv::Mat& t12, const float th) (src/ORBmatcher.cc:1251-1482) after the
  // caller's two projection loops: valid1[i1] = map point of KF1 keypoint i1 is good, not matched yet, projects
  // into KF2 (u1, v1) inside the image and its distance range, level1 = PredictScale(pKF2), desc1 = its
  // descriptor; the "2" arrays are the KF2 -> KF1 direction.  match12[i1] = i2 for mutually consistent pairs.
  int SearchBySim3(const FrameArrays& KF1, const FrameArrays& KF2, const std::vector<float>& mvScaleFactors1,
                   const std::vector<float>& mvScaleFactors2, const std::vector<uint8_t>& valid1,
                   const std::vector<float>& u1, const std::vector<float>& v1, const std::vector<int32_t>& level1,
                   const std::vector<uint8_t>& desc1, const std::vector<uint8_t>& valid2, const std::vector<float>& u2,
                   const std::vector<float>& v2, const std::vector<int32_t>& level2, const std::vector<uint8_t>& desc2,
                   float th, std::vector<int32_t>& match12) {
    match12.assign(KF1.N(), -1);
    int32_t n = 0;
    check(orbfe_search_by_sim3(device_, &KF1.c, &KF2.c, mvScaleFactors1.data(), mvScaleFactors2.data(),
                               (int)mvScaleFactors1.size(), valid1.data(), u1.data(), v1.data(), level1.data(),
                               desc1.data(), valid2.data(), u2.data(), v2.data(), level2.data(), desc2.data(), th,
                               match12.data(), &n),
          "SearchBySim3");
    return n;
  }

  float mfNNratio;
  bool mbCheckOrientation;

 private:
  int device_;
};

// void Frame::ComputeStereoMatches(): fills mvuRight / mvDepth
inline int ComputeStereoMatches(ORBextractor& left, ORBextractor& right, const std::vector<KeyPoint>& kL,
                                const std::vector<uint8_t>& dL, const std::vector<KeyPoint>& kR,
                                const std::vector<uint8_t>& dR, float mbf, float mb, std::vector<float>& mvuRight,
                                std::vector<float>& mvDepth) {
  mvuRight.assign(kL.size(), -1.0f);
  mvDepth.assign(kL.size(), -1.0f);
  int rc = orbfe_compute_stereo_matches(left.handle(), 0, right.handle(), 0,
                                        reinterpret_cast<const orbfe_keypoint*>(kL.data()), dL.data(), (int)kL.size(),
                                        reinterpret_cast<const orbfe_keypoint*>(kR.data()), dR.data(), (int)kR.size(),
                                        mbf, mb, mvuRight.data(), mvDepth.data());
  check(rc, "ComputeStereoMatches");
  return rc;
}

}  // namespace orbfe_cpp
