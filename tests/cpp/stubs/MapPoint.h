// Declarations only; see tests/cpp/stubs/README.md.  (reference include/MapPoint.h:44-105)
#pragma once
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class Frame;
class MapPoint {
 public:
  cv::Mat GetWorldPos();
  cv::Mat GetNormal();
  int Observations();
  void AddObservation(KeyFrame* pKF, size_t idx);
  int GetIndexInKeyFrame(KeyFrame* pKF);
  bool IsInKeyFrame(KeyFrame* pKF);
  bool isBad();
  void Replace(MapPoint* pMP);
  cv::Mat GetDescriptor();
  float GetMinDistanceInvariance();
  float GetMaxDistanceInvariance();
  int PredictScale(const float& currentDist, KeyFrame* pKF);
  int PredictScale(const float& currentDist, Frame* pF);
  float mTrackProjX, mTrackProjY, mTrackProjXR;
  bool mbTrackInView;
  int mnTrackScaleLevel;
  float mTrackViewCos;
};
}  // namespace ORB_SLAM2
