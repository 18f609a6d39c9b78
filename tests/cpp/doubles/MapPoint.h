// Functional test double of ORB_SLAM2::MapPoint: the members src/ORBmatcher_orbfe.cc touches; see README.md.
#pragma once
#include <cmath>
#include <map>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class Frame;
class MapPoint {
 public:
  int id = -1;                       // test bookkeeping: index of the world point
  cv::Mat pos, normal, desc;         // 3x1 CV_32F, 3x1 CV_32F, 1x32 CV_8U
  int nObs = 0;
  bool bad = false;
  float minDist = 0.f, maxDist = 1e30f;   // already scaled by 0.8 / 1.2 like Get*DistanceInvariance
  std::map<KeyFrame*, size_t> obs;
  MapPoint* replacedBy = nullptr;    // Replace() recorded here (the test checks who replaced whom)
  cv::Mat GetWorldPos() { return pos.clone(); }
  cv::Mat GetNormal() { return normal.clone(); }
  int Observations() { return nObs; }
  void AddObservation(KeyFrame* pKF, size_t idx) { if (!obs.count(pKF)) { obs[pKF] = idx; nObs++; } }
  int GetIndexInKeyFrame(KeyFrame* pKF) { return obs.count(pKF) ? (int)obs[pKF] : -1; }
  bool IsInKeyFrame(KeyFrame* pKF) { return obs.count(pKF) != 0; }
  bool isBad() { return bad; }
  void Replace(MapPoint* pMP) { if (pMP == this) return; bad = true; replacedBy = pMP; }
  cv::Mat GetDescriptor() { return desc.clone(); }
  float GetMinDistanceInvariance() { return minDist; }
  float GetMaxDistanceInvariance() { return maxDist; }
  float maxDistRaw = 1.f;            // mfMaxDistance
  template <class T> int predict(float currentDist, T* p) {
    const float ratio = maxDistRaw / currentDist;
    int nScale = (int)std::ceil(std::log((double)ratio) / (double)p->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= p->mnScaleLevels) nScale = p->mnScaleLevels - 1;
    return nScale;
  }
  int PredictScale(const float& currentDist, KeyFrame* pKF);
  int PredictScale(const float& currentDist, Frame* pF);
  float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
  bool mbTrackInView = false;
  int mnTrackScaleLevel = 0;
  float mTrackViewCos = 0;
};
}  // namespace ORB_SLAM2
