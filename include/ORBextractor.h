// ORBextractor.h -- drop-in replacement of the reference header of the same name
// (reference: include/ORBextractor.h) for builds that have OpenCV: class
// ORB_SLAM2::ORBextractor with the reference's exact public signatures, implemented on the
// C-ABI of liborbfe.so.  Frame/Tracking keep calling it unchanged (src/Frame.cc:272-278).
// No OpenCV exists in this image: tests/test_gpu_dropin.py compiles and RUNS this class against a functional
// cv::Mat test double (tests/cpp/doubles/); the OpenCV-free twin it forwards to (orbfe_classes.hpp) is tested directly.
#ifndef ORBFE_DROPIN_ORBEXTRACTOR_H
#define ORBFE_DROPIN_ORBEXTRACTOR_H

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <vector>

#include "orbfe_classes.hpp"

namespace ORB_SLAM2 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST), nlevels_(nlevels) {
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() {}

  // Mask is ignored, as in the reference.
  void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                  cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_cpp::KeyPoint), "cv::KeyPoint layout");
    std::vector<orbfe_cpp::KeyPoint> kps;
    std::vector<uint8_t> desc;
    impl_(image.data, image.cols, image.rows, (int)image.step, kps, desc);
    _keypoints.resize(kps.size());
    if (!kps.empty()) std::memcpy((void*)_keypoints.data(), kps.data(), kps.size() * sizeof(orbfe_cpp::KeyPoint));
    if (kps.empty()) {
      _descriptors.release();
    } else {
      _descriptors.create((int)kps.size(), 32, CV_8U);
      std::memcpy(_descriptors.getMat().data, desc.data(), desc.size());
    }
#ifndef ORBFE_LAZY_PYRAMID
    // public member read by Frame::ComputeStereoMatches (src/Frame.cc:519,609,621,626)
    const std::vector<orbfe_cpp::Image>& pyr = impl_.mvImagePyramid();
    for (int l = 0; l < nlevels_; l++)
      mvImagePyramid[l] = cv::Mat(pyr[l].rows, pyr[l].cols, CV_8UC1, (void*)pyr[l].data.data()).clone();
#endif
  }

  int inline GetLevels() { return impl_.GetLevels(); }
  float inline GetScaleFactor() { return impl_.GetScaleFactor(); }
  std::vector<float> inline GetScaleFactors() { return impl_.GetScaleFactors(); }
  std::vector<float> inline GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
  std::vector<float> inline GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }

  std::vector<cv::Mat> mvImagePyramid;

  orbfe_cpp::ORBextractor& impl() { return impl_; }

 protected:
  orbfe_cpp::ORBextractor impl_;
  int nlevels_;
};

}  // namespace ORB_SLAM2

#endif
