// Functional test double of ORB_SLAM2::Frame; see README.md.
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class Frame {
 public:
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<float> mvuRight, mvDepth;
  DBoW2::FeatureVector mFeatVec;
  cv::Mat mDescriptors, mDescriptorsRight;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mTcw;
  int mnScaleLevels = 8;
  float mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
inline int MapPoint::PredictScale(const float& d, Frame* pF) { return predict(d, pF); }
}  // namespace ORB_SLAM2
