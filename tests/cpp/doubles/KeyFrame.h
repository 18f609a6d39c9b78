// Functional test double of ORB_SLAM2::KeyFrame; see README.md.
#pragma once
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class KeyFrame {
 public:
  cv::Mat Tcw;  // 4x4 CV_32F
  cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
  cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
  cv::Mat GetCameraCenter() { return -(GetRotation().t()) * GetTranslation(); }  // Ow = -Rwc * tcw
  void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
  std::set<MapPoint*> GetMapPoints() {
    std::set<MapPoint*> s;
    for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
    return s;
  }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
  bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
  float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  std::vector<float> mvuRight, mvDepth;
  cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;
  int mnScaleLevels = 8;
  float mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  std::vector<MapPoint*> mvpMapPoints;
};
inline int MapPoint::PredictScale(const float& d, KeyFrame* pKF) { return predict(d, pKF); }
}  // namespace ORB_SLAM2
