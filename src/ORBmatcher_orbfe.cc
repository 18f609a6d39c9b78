// ORBmatcher_orbfe.cc -- drop-in replacement of the reference's src/ORBmatcher.cc: ORB_SLAM2::ORBmatcher with
// the exact signatures of the reference header (include/ORBmatcher.h:55-97 there, include/ORBmatcher.h here),
// every search running on liborbfe.so (C-ABI include/orbfe.h, through the OpenCV-free orbfe_cpp::ORBmatcher).
//
// Shape of every method: flatten -> call -> un-flatten.
//   * flatten: what the search reads from KeyFrame / Frame / MapPoint objects becomes flat arrays.  The loop
//     prologue of each projection search (pose arithmetic on cv::Mat, isBad(), PredictScale, distance and viewing
//     cone gates) is the caller-side part of the reference algorithm and is kept with the same cv::Mat
//     expressions, because its float results must be the ones the reference computes (file:line cited per method);
//   * call: window gather, 256-bit Hamming distances, best / second-best, rotation histogram -- on the GPU;
//   * un-flatten: index results back into MapPoint* bookkeeping (mvpMapPoints, vpMatched, Replace / AddObservation).
//
// A build against OpenCV and the reference's MapPoint.h / KeyFrame.h / Frame.h is not possible in this repository
// (neither exists in the image).  What IS done: tests/test_gpu_dropin.py compiles this file (g++ -std=c++11) against
// functional test doubles of exactly the cv::Mat / Frame / KeyFrame / MapPoint members it touches
// (tests/cpp/doubles/, authored here) and RUNS all 12 methods on the GPU, comparing the un-flattened MapPoint*
// results with the oracle behind an independent numpy-float32 prologue; tests/test_dropin_headers.py additionally
// parses it against declaration stubs and compares the 12 public signatures with the reference header.
#include "ORBmatcher.h"

#include <stdint.h>

#include <cmath>
#include <cstring>
#include <utility>

#include "orbfe_classes.hpp"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;  // src/ORBmatcher.cc:37-39
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

namespace {

static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_cpp::KeyPoint), "cv::KeyPoint must be the 28-byte record of the C-ABI");

std::vector<orbfe_cpp::KeyPoint> flat_keys(const std::vector<cv::KeyPoint>& v) {
  std::vector<orbfe_cpp::KeyPoint> out(v.size());
  if (!v.empty()) std::memcpy(static_cast<void*>(out.data()), static_cast<const void*>(v.data()), v.size() * sizeof(orbfe_cpp::KeyPoint));
  return out;
}
std::vector<uint8_t> flat_rows(const cv::Mat& d) {  // N x 32 CV_8U, any step
  std::vector<uint8_t> out((size_t)d.rows * 32);
  for (int i = 0; i < d.rows; i++) std::memcpy(&out[(size_t)i * 32], d.ptr<uchar>(i), 32);
  return out;
}
void put_row(std::vector<uint8_t>& dst, size_t i, const cv::Mat& d) { std::memcpy(&dst[i * 32], d.ptr<uchar>(0), 32); }

// Frame: static undistorted bounds (include/Frame.h:201-204); KeyFrame: its own copies (include/KeyFrame.h:191-197)
struct FrameFlat {
  orbfe_cpp::FrameArrays a;
  explicit FrameFlat(const Frame& F)
      : a(flat_keys(F.mvKeysUn), flat_rows(F.mDescriptors), Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, F.mvuRight) {}
  explicit FrameFlat(const KeyFrame* pKF)
      : a(flat_keys(pKF->mvKeysUn), flat_rows(pKF->mDescriptors), (float)pKF->mnMinX, (float)pKF->mnMaxX, (float)pKF->mnMinY,
          (float)pKF->mnMaxY, pKF->mvuRight) {}
};

// One projected map point of the Sim3 / Fuse style prologues (src/ORBmatcher.cc:355-400, 960-1020, 1130-1180):
// camera-frame depth test, pinhole projection, image test, distance range, optional viewing cone, PredictScale.
struct Projection {
  std::vector<uint8_t> valid, desc;
  std::vector<float> u, v, ur;
  std::vector<int32_t> level;
  explicit Projection(size_t n) : valid(n, 0), desc(n * 32, 0), u(n, 0.f), v(n, 0.f), ur(n, -1.f), level(n, 0) {}
};

}  // namespace

// A single pair: answered on the host (exact integer; the reference's bit trick, :1828-1844, is a popcount).
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
  const uint32_t* pa = a.ptr<uint32_t>();
  const uint32_t* pb = b.ptr<uint32_t>();
  int dist = 0;
  for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
  return dist;
}

// ---- (1) Tracking::SearchLocalPoints, src/ORBmatcher.cc:51-138 ----
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
  const size_t n = vpMapPoints.size();
  std::vector<uint8_t> inView(n, 0), obsPos(n, 0), desc(n * 32, 0), blocked(F.N, 0);
  std::vector<int32_t> level(n, 0);
  std::vector<float> viewCos(n, 0.f), px(n, 0.f), py(n, 0.f), pxr(n, 0.f);
  for (size_t i = 0; i < n; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP->mbTrackInView || pMP->isBad()) continue;  // :57-62
    inView[i] = 1;
    level[i] = pMP->mnTrackScaleLevel;
    viewCos[i] = pMP->mTrackViewCos;
    px[i] = pMP->mTrackProjX;
    py[i] = pMP->mTrackProjY;
    pxr[i] = pMP->mTrackProjXR;
    obsPos[i] = pMP->Observations() > 0;  // a point assigned to a keypoint blocks it for later points only then (:92-94)
    put_row(desc, i, pMP->GetDescriptor());
  }
  for (int idx = 0; idx < F.N; idx++) blocked[idx] = F.mvpMapPoints[idx] && F.mvpMapPoints[idx]->Observations() > 0;
  FrameFlat f(F);
  std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
  int32_t nmatches = 0;
  orbfe_cpp::check(orbfe_search_by_projection(0, &f.a.c, F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), blocked.data(),
                                              (int)n, inView.data(), level.data(), viewCos.data(), px.data(), py.data(),
                                              F.mvuRight.empty() ? NULL : pxr.data(), desc.data(), obsPos.data(), th, mfNNratio,
                                              match.data(), &nmatches),
                   "SearchByProjection(Frame,MapPoints)");
  for (int idx = 0; idx < F.N; idx++)
    if (match[idx] >= 0) F.mvpMapPoints[idx] = vpMapPoints[match[idx]];  // :131
  return nmatches;
}

// ---- (2) Tracking::TrackWithMotionModel, src/ORBmatcher.cc:1484-1633 ----
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
  const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);  // :1494-1512, unchanged pose arithmetic
  const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
  const cv::Mat twc = -Rcw.t() * tcw;
  const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
  const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
  const cv::Mat tlc = Rlw * twc + tlw;
  const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
  const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;
  const int N = LastFrame.N;
  std::vector<uint8_t> valid(N, 0), desc((size_t)N * 32, 0), obs(N, 0);
  std::vector<float> u(N, 0.f), v(N, 0.f), invzc(N, 0.f), ang(N, 0.f);
  std::vector<int32_t> oct(N, 0);
  for (int i = 0; i < N; i++) {  // :1514-1541 up to the window query
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const cv::Mat x3Dw = pMP->GetWorldPos();
    const cv::Mat x3Dc = Rcw * x3Dw + tcw;
    const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
    const float iz = 1.0 / x3Dc.at<float>(2);
    if (iz < 0) continue;
    const float uu = CurrentFrame.fx * xc * iz + CurrentFrame.cx;
    const float vv = CurrentFrame.fy * yc * iz + CurrentFrame.cy;
    if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX) continue;
    if (vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
    valid[i] = 1;
    u[i] = uu;
    v[i] = vv;
    invzc[i] = iz;
    oct[i] = LastFrame.mvKeys[i].octave;   // :1543
    ang[i] = LastFrame.mvKeysUn[i].angle;  // :1601
    obs[i] = pMP->Observations() > 0;      // :1572-1574 for points already assigned to a keypoint of CurrentFrame
    put_row(desc, i, pMP->GetDescriptor());
  }
  FrameFlat cur(CurrentFrame);
  std::vector<int32_t> matchCur(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
  std::vector<uint8_t> blocked(CurrentFrame.N > 0 ? CurrentFrame.N : 1, 0);  // :1572-1574, the state at entry
  for (int i2 = 0; i2 < CurrentFrame.N; i2++)
    blocked[i2] = CurrentFrame.mvpMapPoints[i2] && CurrentFrame.mvpMapPoints[i2]->Observations() > 0;
  int32_t nmatches = 0;
  orbfe_cpp::check(orbfe_search_by_projection_last_frame(0, &cur.a.c, CurrentFrame.mvScaleFactors.data(),
                                                         (int)CurrentFrame.mvScaleFactors.size(), CurrentFrame.mbf, N,
                                                         valid.data(), u.data(), v.data(), invzc.data(), oct.data(),
                                                         ang.data(), desc.data(), obs.data(), blocked.data(),
                                                         bForward ? 1 : (bBackward ? 2 : 0),
                                                         th, mbCheckOrientation, matchCur.data(), &nmatches),
                   "SearchByProjection(Frame,Frame)");
  for (int i2 = 0; i2 < CurrentFrame.N; i2++)
    if (matchCur[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[matchCur[i2]];  // :1597
  return nmatches;
}

// ---- (3) Tracking::Relocalization, src/ORBmatcher.cc:1641-1775 ----
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist) {
  const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);  // :1645-1647
  const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
  const cv::Mat Ow = -Rcw.t() * tcw;
  const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
  const size_t n = vpMPs.size();
  std::vector<uint8_t> valid(n, 0), desc(n * 32, 0), blocked(CurrentFrame.N, 0);
  std::vector<float> u(n, 0.f), v(n, 0.f), kfAngle(n, 0.f);
  std::vector<int32_t> level(n, 0);
  for (size_t i = 0; i < n; i++) {  // :1657-1693
    MapPoint* pMP = vpMPs[i];
    if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
    const cv::Mat x3Dw = pMP->GetWorldPos();
    const cv::Mat x3Dc = Rcw * x3Dw + tcw;
    const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
    const float iz = 1.0 / x3Dc.at<float>(2);
    const float uu = CurrentFrame.fx * xc * iz + CurrentFrame.cx;
    const float vv = CurrentFrame.fy * yc * iz + CurrentFrame.cy;
    if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX) continue;
    if (vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
    const cv::Mat PO = x3Dw - Ow;
    const float dist3D = cv::norm(PO);
    if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
    valid[i] = 1;
    u[i] = uu;
    v[i] = vv;
    level[i] = pMP->PredictScale(dist3D, &CurrentFrame);
    kfAngle[i] = pKF->mvKeysUn[i].angle;  // :1732
    put_row(desc, i, pMP->GetDescriptor());
  }
  for (int i2 = 0; i2 < CurrentFrame.N; i2++) blocked[i2] = CurrentFrame.mvpMapPoints[i2] != NULL;  // :1712
  FrameFlat cur(CurrentFrame);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> matchCur;
  const int nmatches = m.SearchByProjection(cur.a, CurrentFrame.mvScaleFactors, blocked, valid, u, v, level, kfAngle, desc, th,
                                            ORBdist, matchCur);
  for (int i2 = 0; i2 < CurrentFrame.N; i2++)
    if (matchCur[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[matchCur[i2]];  // :1728
  return nmatches;
}

namespace {
// Scw -> (Rcw, tcw = t/s, Ow) of src/ORBmatcher.cc:342-347 / :1120-1125
void split_sim3(const cv::Mat& Scw, cv::Mat& Rcw, cv::Mat& tcw, cv::Mat& Ow) {
  const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
  const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
  Rcw = sRcw / scw;
  tcw = Scw.rowRange(0, 3).col(3) / scw;
  Ow = -Rcw.t() * tcw;
}
// the per-point gates shared by SearchByProjection(KF,Scw), Fuse and Fuse(Scw): :366-400, :975-1008, :1139-1172
bool project_into_keyframe(MapPoint* pMP, KeyFrame* pKF, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow, float bf,
                           float* u, float* v, float* ur, int* level) {
  const cv::Mat p3Dw = pMP->GetWorldPos();
  const cv::Mat p3Dc = Rcw * p3Dw + tcw;
  if (p3Dc.at<float>(2) < 0.0f) return false;
  const float invz = 1.0 / p3Dc.at<float>(2);
  const float x = p3Dc.at<float>(0) * invz, y = p3Dc.at<float>(1) * invz;
  *u = pKF->fx * x + pKF->cx;
  *v = pKF->fy * y + pKF->cy;
  if (!pKF->IsInImage(*u, *v)) return false;
  *ur = *u - bf * invz;
  const cv::Mat PO = p3Dw - Ow;
  const float dist3D = cv::norm(PO);
  if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) return false;
  const cv::Mat Pn = pMP->GetNormal();
  if (PO.dot(Pn) < 0.5 * dist3D) return false;  // viewing angle below 60 degrees
  *level = pMP->PredictScale(dist3D, pKF);
  return true;
}
}  // namespace

// ---- (4) LoopClosing::ComputeSim3, src/ORBmatcher.cc:335-449 ----
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                                   std::vector<MapPoint*>& vpMatched, int th) {
  cv::Mat Rcw, tcw, Ow;
  split_sim3(Scw, Rcw, tcw, Ow);
  std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());  // :349-350
  spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
  const size_t n = vpPoints.size();
  Projection p(n);
  for (size_t i = 0; i < n; i++) {
    MapPoint* pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    int lev = 0;
    if (!project_into_keyframe(pMP, pKF, Rcw, tcw, Ow, 0.0f, &p.u[i], &p.v[i], &p.ur[i], &lev)) continue;
    p.valid[i] = 1;
    p.level[i] = lev;
    put_row(p.desc, i, pMP->GetDescriptor());
  }
  std::vector<uint8_t> already(vpMatched.size());
  for (size_t idx = 0; idx < vpMatched.size(); idx++) already[idx] = vpMatched[idx] != NULL;  // :417
  FrameFlat kf(pKF);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> match;
  const int nmatches = m.SearchByProjection(kf.a, pKF->mvScaleFactors, already, p.valid, p.u, p.v, p.level, p.desc, th, match);
  for (size_t idx = 0; idx < match.size() && idx < vpMatched.size(); idx++)
    if (match[idx] >= 0) vpMatched[idx] = vpPoints[match[idx]];  // :440
  return nmatches;
}

// ---- (5) Tracking::TrackReferenceKeyFrame / Relocalization, src/ORBmatcher.cc:185-325 ----
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
  const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
  std::vector<uint8_t> hasMp(vpMapPointsKF.size());
  std::vector<float> angKF(pKF->N), angF(F.N);
  for (size_t i = 0; i < hasMp.size(); i++) hasMp[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :222-228
  for (int i = 0; i < pKF->N; i++) angKF[i] = pKF->mvKeysUn[i].angle;  // :272
  for (int i = 0; i < F.N; i++) angF[i] = F.mvKeys[i].angle;
  const orbfe_cpp::FeatureVectorCSR fvKF(pKF->mFeatVec), fvF(F.mFeatVec);  // DBoW2 std::map -> CSR, same order
  const std::vector<uint8_t> dKF = flat_rows(pKF->mDescriptors), dF = flat_rows(F.mDescriptors);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> matchF;
  const int n = m.SearchByBoW(dKF.data(), hasMp.data(), angKF.data(), pKF->N, fvKF, dF.data(), angF.data(), F.N, fvF, matchF);
  vpMapPointMatches.assign(F.N, static_cast<MapPoint*>(NULL));  // :189
  for (int j = 0; j < F.N; j++)
    if (matchF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[matchF[j]];  // :267
  return n;
}

// ---- (6) LoopClosing::ComputeSim3, src/ORBmatcher.cc:610-743 ----
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
  const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  std::vector<uint8_t> has1(vpMapPoints1.size()), has2(vpMapPoints2.size());
  for (size_t i = 0; i < has1.size(); i++) has1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();  // :648-652
  for (size_t i = 0; i < has2.size(); i++) has2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();  // :667-671
  std::vector<float> ang1(pKF1->N), ang2(pKF2->N);
  for (int i = 0; i < pKF1->N; i++) ang1[i] = pKF1->mvKeysUn[i].angle;
  for (int i = 0; i < pKF2->N; i++) ang2[i] = pKF2->mvKeysUn[i].angle;
  const orbfe_cpp::FeatureVectorCSR fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
  const std::vector<uint8_t> d1 = flat_rows(pKF1->mDescriptors), d2 = flat_rows(pKF2->mDescriptors);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> match12;
  const int n = m.SearchByBoW(d1.data(), has1.data(), ang1.data(), pKF1->N, fv1, d2.data(), has2.data(), ang2.data(), pKF2->N,
                              fv2, match12);
  vpMatches12.assign(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));  // :619
  for (size_t i = 0; i < match12.size() && i < vpMatches12.size(); i++)
    if (match12[i] >= 0) vpMatches12[i] = vpMapPoints2[match12[i]];  // :690
  return n;
}

// ---- (7) Tracking::MonocularInitialization, src/ORBmatcher.cc:469-603 ----
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                        std::vector<int>& vnMatches12, int windowSize) {
  FrameFlat f1(F1), f2(F2);
  std::vector<float> px(vbPrevMatched.size()), py(vbPrevMatched.size());
  for (size_t i = 0; i < vbPrevMatched.size(); i++) { px[i] = vbPrevMatched[i].x; py[i] = vbPrevMatched[i].y; }
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> m12;
  const int n = m.SearchForInitialization(f1.a, f2.a, px, py, m12, windowSize);
  vnMatches12.assign(m12.begin(), m12.end());
  for (size_t i = 0; i < vbPrevMatched.size(); i++) vbPrevMatched[i] = cv::Point2f(px[i], py[i]);  // :596-599, done by the library
  return n;
}

// ---- (8) LocalMapping::CreateNewMapPoints, src/ORBmatcher.cc:754-928 ----
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                                       std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo) {
  // epipole of camera 1 in image 2, :761-769 (unchanged pose arithmetic)
  const cv::Mat Cw = pKF1->GetCameraCenter();
  const cv::Mat R2w = pKF2->GetRotation();
  const cv::Mat t2w = pKF2->GetTranslation();
  const cv::Mat C2 = R2w * Cw + t2w;
  const float invz = 1.0f / C2.at<float>(2);
  const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
  const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
  std::vector<uint8_t> has1(pKF1->N), has2(pKF2->N);
  for (int i = 0; i < pKF1->N; i++) has1[i] = pKF1->GetMapPoint(i) != NULL;  // :800-803: only keypoints WITHOUT a point
  for (int i = 0; i < pKF2->N; i++) has2[i] = pKF2->GetMapPoint(i) != NULL;  // :822-826
  float F[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
  FrameFlat k1(pKF1), k2(pKF2);
  const orbfe_cpp::FeatureVectorCSR fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  return m.SearchForTriangulation(k1.a, has1, fv1, k2.a, has2, fv2, F, ex, ey, pKF2->mvScaleFactors, pKF2->mvLevelSigma2,
                                  vMatchedPairs, bOnlyStereo);
}

// ---- (9) LoopClosing::ComputeSim3, src/ORBmatcher.cc:1251-1482 ----
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12,
                             const cv::Mat& R12, const cv::Mat& t12, const float th) {
  const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();  // :1260-1270
  const cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
  const cv::Mat sR12 = s12 * R12;
  const cv::Mat sR21 = (1.0 / s12) * R12.t();
  const cv::Mat t21 = -sR21 * t12;
  const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
  std::vector<bool> done1(N1, false), done2(N2, false);  // :1278-1291
  for (int i = 0; i < N1; i++) {
    MapPoint* pMP = vpMatches12[i];
    if (!pMP) continue;
    done1[i] = true;
    const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
    if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
  }
  // one direction of :1297-1340 / :1369-1410: points of `from` projected into `to` through (sR, t) after (Rw, tw)
  struct Dir {
    static void run(const std::vector<MapPoint*>& pts, const std::vector<bool>& done, const cv::Mat& Rw, const cv::Mat& tw,
                    const cv::Mat& sR, const cv::Mat& t, KeyFrame* to, float fx, float fy, float cx, float cy, Projection& p) {
      for (size_t i = 0; i < pts.size(); i++) {
        MapPoint* pMP = pts[i];
        if (!pMP || done[i] || pMP->isBad()) continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        const cv::Mat pa = Rw * p3Dw + tw;
        const cv::Mat pb = sR * pa + t;
        if (pb.at<float>(2) < 0.0) continue;
        const float invz = 1.0 / pb.at<float>(2);
        const float x = pb.at<float>(0) * invz, y = pb.at<float>(1) * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!to->IsInImage(u, v)) continue;
        const float dist3D = cv::norm(pb);
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
        p.valid[i] = 1;
        p.u[i] = u;
        p.v[i] = v;
        p.level[i] = pMP->PredictScale(dist3D, to);
        put_row(p.desc, i, pMP->GetDescriptor());
      }
    }
  };
  Projection p1(N1), p2(N2);
  Dir::run(vpMapPoints1, done1, R1w, t1w, sR21, t21, pKF2, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, p1);  // :1253-1256: KF1's intrinsics
  Dir::run(vpMapPoints2, done2, R2w, t2w, sR12, t12, pKF1, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, p2);
  FrameFlat k1(pKF1), k2(pKF2);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> match12;
  const int nFound = m.SearchBySim3(k1.a, k2.a, pKF1->mvScaleFactors, pKF2->mvScaleFactors, p1.valid, p1.u, p1.v, p1.level, p1.desc,
                                    p2.valid, p2.u, p2.v, p2.level, p2.desc, th, match12);
  for (int i1 = 0; i1 < N1; i1++)
    if (match12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match12[i1]];  // :1470-1474
  return nFound;
}

// ---- (10) LocalMapping::SearchInNeighbors, src/ORBmatcher.cc:940-1110 ----
// The search of every map point reads only geometry and descriptors, so all of them go to the GPU in one call with
// the state at entry; the reference's sequential side effects are replayed afterwards in the original order, and a
// point that an EARLIER iteration made bad or attached to pKF (Replace / AddObservation) is skipped exactly where
// the reference's loop head would skip it (:955-959).  (One residual difference: Replace() recomputes the surviving
// point's distinctive descriptor; if that same MapPoint* occurs AGAIN later in vpMapPoints, the reference searches
// it with the new descriptor, this replay with the one at entry.  The callers pass each point once.)
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th) {
  const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
  const size_t n = vpMapPoints.size();
  Projection p(n);
  for (size_t i = 0; i < n; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
    int lev = 0;
    if (!project_into_keyframe(pMP, pKF, Rcw, tcw, Ow, pKF->mbf, &p.u[i], &p.v[i], &p.ur[i], &lev)) continue;
    p.valid[i] = 1;
    p.level[i] = lev;
    put_row(p.desc, i, pMP->GetDescriptor());
  }
  FrameFlat kf(pKF);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> bestIdx;
  m.Fuse(kf.a, pKF->mvScaleFactors, pKF->mvInvLevelSigma2, p.valid, p.u, p.v, p.ur, p.level, p.desc, bestIdx, th);
  int nFused = 0;
  for (size_t i = 0; i < n; i++) {  // :1086-1106
    if (bestIdx[i] < 0) continue;
    MapPoint* pMP = vpMapPoints[i];
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // became so during this loop
    MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, bestIdx[i]);
      pKF->AddMapPoint(pMP, bestIdx[i]);
    }
    nFused++;
  }
  return nFused;
}

// ---- (11) LoopClosing::SearchAndFuse, src/ORBmatcher.cc:1112-1249 ----
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th,
                     std::vector<MapPoint*>& vpReplacePoint) {
  cv::Mat Rcw, tcw, Ow;
  split_sim3(Scw, Rcw, tcw, Ow);
  const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();  // :1127
  const size_t n = vpPoints.size();
  Projection p(n);
  for (size_t i = 0; i < n; i++) {
    MapPoint* pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    int lev = 0;
    if (!project_into_keyframe(pMP, pKF, Rcw, tcw, Ow, 0.0f, &p.u[i], &p.v[i], &p.ur[i], &lev)) continue;
    p.valid[i] = 1;
    p.level[i] = lev;
    put_row(p.desc, i, pMP->GetDescriptor());
  }
  FrameFlat kf(pKF);
  orbfe_cpp::ORBmatcher m(mfNNratio, mbCheckOrientation);
  std::vector<int32_t> bestIdx;
  m.Fuse(kf.a, pKF->mvScaleFactors, p.valid, p.u, p.v, p.level, p.desc, th, bestIdx);
  int nFused = 0;
  for (size_t i = 0; i < n; i++) {  // :1228-1243
    if (bestIdx[i] < 0) continue;
    MapPoint* pMP = vpPoints[i];
    MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, bestIdx[i]);
      pKF->AddMapPoint(pMP, bestIdx[i]);
    }
    nFused++;
  }
  return nFused;
}

}  // namespace ORB_SLAM2
