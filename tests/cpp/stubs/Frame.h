// Declarations only; see tests/cpp/stubs/README.md.  (reference include/Frame.h:120-204)
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class Frame {
 public:
  float fx, fy, cx, cy, mbf, mb;
  int N;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<float> mvuRight, mvDepth;
  DBoW2::FeatureVector mFeatVec;
  cv::Mat mDescriptors, mDescriptorsRight;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mTcw;
  int mnScaleLevels;
  std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
}  // namespace ORB_SLAM2
