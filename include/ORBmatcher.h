// ORBmatcher.h -- drop-in replacement of the reference header of the same name (reference:
// include/ORBmatcher.h:55-118) for builds that have OpenCV and the reference's own SLAM headers
// (MapPoint.h, KeyFrame.h, Frame.h): class ORB_SLAM2::ORBmatcher with the reference's exact public
// signatures, implemented in src/ORBmatcher_orbfe.cc of THIS repository on the C-ABI of liborbfe.so.
// Tracking / LocalMapping / LoopClosing keep calling it unchanged (src/Tracking.cc:840,1478,
// src/LocalMapping.cc:261,542, src/LoopClosing.cc:294,691).
//
// Integration: replace the reference's include/ORBmatcher.h by this file, replace src/ORBmatcher.cc by
// src/ORBmatcher_orbfe.cc in CMakeLists.txt:54-73, add -lorbfe.  No OpenCV / Eigen exists in this image, so the
// pair is exercised against functional test doubles instead (tests/test_gpu_dropin.py: all 12 methods run on the GPU
// and match the oracle); the OpenCV-free twin it forwards to -- orbfe_cpp::ORBmatcher, include/orbfe_classes.hpp --
// is compiled and parity-tested through tests/cpp/test_classes.cpp.
#ifndef ORBFE_DROPIN_ORBMATCHER_H
#define ORBFE_DROPIN_ORBMATCHER_H

#include <set>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

namespace ORB_SLAM2 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true);

  // Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1828-1844).  A single pair is answered on the
  // host (one 256-bit popcount is not worth a PCIe round trip; the result is an exact integer either way); the
  // searches below run their distances on the GPU.
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

  // Tracking: local map (src/ORBmatcher.cc:51-138)
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
  // Tracking: motion model (src/ORBmatcher.cc:1484-1633)
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
  // Tracking: relocalisation (src/ORBmatcher.cc:1641-1775)
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                         const int ORBdist);
  // LoopClosing: Sim3 projection (src/ORBmatcher.cc:335-449)
  int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                         std::vector<MapPoint*>& vpMatched, int th);

  // Brute force inside vocabulary nodes (src/ORBmatcher.cc:185-325, 610-743)
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

  // Monocular initialisation (src/ORBmatcher.cc:469-603)
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10);

  // LocalMapping: new map points (src/ORBmatcher.cc:754-928)
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                             std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);

  // LoopClosing: Sim3 [s12*R12|t12] guided matches (src/ORBmatcher.cc:1251-1482)
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12,
                   const cv::Mat& R12, const cv::Mat& t12, const float th);

  // Duplicate map points (src/ORBmatcher.cc:940-1110, 1112-1249)
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
  int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th,
           std::vector<MapPoint*>& vpReplacePoint);

 public:
  static const int TH_LOW;
  static const int TH_HIGH;
  static const int HISTO_LENGTH;

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2

#endif
