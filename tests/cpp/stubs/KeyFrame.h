// Declarations only; see tests/cpp/stubs/README.md.  (reference include/KeyFrame.h:86-215)
#pragma once
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class KeyFrame {
 public:
  cv::Mat GetRotation();
  cv::Mat GetTranslation();
  cv::Mat GetCameraCenter();
  void AddMapPoint(MapPoint* pMP, const size_t& idx);
  std::set<MapPoint*> GetMapPoints();
  std::vector<MapPoint*> GetMapPointMatches();
  MapPoint* GetMapPoint(const size_t& idx);
  bool IsInImage(const float& x, const float& y) const;
  const float fx, fy, cx, cy, invfx, invfy, mbf, mb, mThDepth;
  const int N;
  const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  const std::vector<float> mvuRight, mvDepth;
  const cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;
  const int mnScaleLevels;
  const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  const int mnMinX, mnMinY, mnMaxX, mnMaxY;
};
}  // namespace ORB_SLAM2
