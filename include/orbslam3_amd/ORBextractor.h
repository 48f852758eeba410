// ORBextractor.h -- drop-in C++ adaptor with the reference's class name and signatures
// (reference include/ORBextractor.h:41-119) over the orbx_* C ABI.  Frame::ExtractORB
// (reference src/Frame.cc:555-562) and the Frame constructors that read the scale tables
// (src/Frame.cc:190-196) compile against this header unchanged when OpenCV is present; without OpenCV
// the cv:: types are the stand-ins of cv_compat.h.  Header-only; link liborbx.so.
#ifndef ORBSLAM3_AMD_ORBEXTRACTOR_H_
#define ORBSLAM3_AMD_ORBEXTRACTOR_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "../orbx.h"
#include "cv_compat.h"

namespace ORB_SLAM3 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  // reference include/ORBextractor.h:48-50 ; device_id selects the GPU (one extractor = one stream + workspace)
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id = 0)
      : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST) {
    if (orbx_create(&h_, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, device_id) != 0)
      throw std::runtime_error(std::string("orbx_create: ") + orbx_last_error());
    mvScaleFactor.resize(nlevels), mvInvScaleFactor.resize(nlevels), mvLevelSigma2.resize(nlevels), mvInvLevelSigma2.resize(nlevels);
    orbx_get_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
    mnFeaturesPerLevel.resize(nlevels);
    orbx_get_features_per_level(h_, mnFeaturesPerLevel.data());
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { orbx_destroy(h_); }
  ORBextractor(const ORBextractor&)            = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image; the mask is ignored (reference :52-61).
  // Returns monoIndex as the reference does, -1 for an empty image.
  int operator()(const orbx_cv::Mat& image, const orbx_cv::Mat& /*mask*/, std::vector<orbx_cv::KeyPoint>& keypoints,
                 orbx_cv::Mat& descriptors, std::vector<int>& vLappingArea) {
    if (image.empty()) return -1;
    const int cap = orbx_max_keypoints(h_, image.rows, image.cols);
    if (cap < 0) throw std::runtime_error(std::string("orbx_max_keypoints: ") + orbx_last_error());
    kp_.resize(cap);
    desc_.resize((size_t)cap * 32);
    if (mbDownloadPyramid != downloading_) orbx_set_pyramid_download(h_, mbDownloadPyramid ? 1 : 0), downloading_ = mbDownloadPyramid;
    int       n  = 0;
    const int rc = orbx_extract(h_, image.ptr(0), image.rows, image.cols, (size_t)image.step, vLappingArea.at(0), vLappingArea.at(1), kp_.data(),
                                desc_.data(), cap, &n);
    if (rc <= -2) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error());
    static_assert(sizeof(orbx_cv::KeyPoint) == sizeof(orbx_keypoint_t), "cv::KeyPoint must be the 28-byte record");
    keypoints.resize(n);
    if (n) std::memcpy(reinterpret_cast<void*>(keypoints.data()), kp_.data(), (size_t)n * sizeof(orbx_keypoint_t));
    if (n == 0) {
      descriptors.release();
    } else {
      descriptors.create(n, 32, orbx_cv::CV_8U);
      for (int i = 0; i < n; i++) std::memcpy(descriptors.ptr(i), desc_.data() + (size_t)i * 32, 32);
    }
    if (mbDownloadPyramid) {  // Frame::ComputeStereoMatches reads mvImagePyramid (src/Frame.cc:950, 1039-1058)
      // levels >= 1 already lie in pinned host memory (orbx_set_pyramid_download: they came home behind the records); level 0 is the
      // device's level 0 as well, so a fused input resize (orbx_set_input_resize) shows up here as it does in the reference
      for (int l = 0; l < nlevels; l++) {
        int r = 0, c = 0;
        if (orbx_pyramid_level(h_, 0, l, nullptr, 0, &r, &c) != 0) throw std::runtime_error(std::string("orbx_pyramid_level: ") + orbx_last_error());
        if (l == 0 && r == image.rows && c == image.cols) {  // no resize: level 0 IS the input image
          mvImagePyramid[0] = image.clone();
          continue;
        }
        mvImagePyramid[l].create(r, c, orbx_cv::CV_8U);
        if (orbx_pyramid_level(h_, 0, l, mvImagePyramid[l].ptr(0), (size_t)mvImagePyramid[l].step, &r, &c) != 0)
          throw std::runtime_error(std::string("orbx_pyramid_level: ") + orbx_last_error());
      }
    } else {
      for (auto& m : mvImagePyramid) m.release();
    }
    return rc;
  }

  int                inline GetLevels() { return nlevels; }
  float              inline GetScaleFactor() { return (float)scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<orbx_cv::Mat> mvImagePyramid;            // host copies of the device pyramid of the last call (reference include/ORBextractor.h:87)
  // ON by default: an UNCHANGED Frame::ComputeStereoMatches reads mvImagePyramid (src/Frame.cc:950, 1039-1058) and must find the levels
  // there.  The levels ride home inside the call (orbx_set_pyramid_download), so the price is one more copy in the queue, not a round
  // trip per level.  A host that does not read the pyramid (monocular tracking; stereo through orbx_stereo_matches, which works on the
  // device-resident levels) opts out with `extractor->mbDownloadPyramid = false;` -- the one line INTEGRATION.md section 1 lists as optional.
  bool                      mbDownloadPyramid = true;
  orbx_t*                   handle() { return h_; }

 protected:
  int                nfeatures;
  double             scaleFactor;
  int                nlevels;
  int                iniThFAST;
  int                minThFAST;
  std::vector<int>   mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

 private:
  orbx_t*                      h_ = nullptr;
  bool                         downloading_ = false;  // what orbx_set_pyramid_download was last told
  std::vector<orbx_keypoint_t> kp_;
  std::vector<uint8_t>         desc_;
};

}  // namespace ORB_SLAM3
#endif
