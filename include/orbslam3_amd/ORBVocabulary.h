// ORBVocabulary.h -- C++ adaptor for the vocabulary transform (reference include/ORBVocabulary.h:29:
// typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary) over orbv_*.
// Only what the SLAM front-end calls per frame is here: loadFromTextFile (src/System.cc) and
// transform(features, BowVector&, FeatureVector&, levelsup) (src/Frame.cc:879-884, src/KeyFrame.cc ComputeBoW).
// The descriptor argument is the N x 32 CV_8U matrix the extractor filled -- the per-row cv::Mat vector the reference
// builds with Converter::toDescriptorVector is not needed.
#ifndef ORBSLAM3_AMD_ORBVOCABULARY_H_
#define ORBSLAM3_AMD_ORBVOCABULARY_H_

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../orbx.h"
#include "cv_compat.h"

namespace ORB_SLAM3_AMD {

class ORBVocabulary {
 public:
  typedef std::map<unsigned int, double>                    BowVector;      // DBoW2::BowVector  (WordId -> WordValue)
  typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;  // DBoW2::FeatureVector (NodeId -> feature indices)

  explicit ORBVocabulary(int device_id = 0) : device_(device_id) {}
  ~ORBVocabulary() { orbv_destroy(h_); }
  ORBVocabulary(const ORBVocabulary&)            = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;

  // TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1337-1420)
  bool loadFromTextFile(const std::string& filename) {
    orbv_destroy(h_);
    h_ = nullptr;
    return orbv_load_text(&h_, filename.c_str(), device_) == 0;
  }
  bool empty() const {
    int n = 0;
    return !h_ || orbv_info(h_, nullptr, nullptr, &n, nullptr) != 0 || n == 0;
  }
  unsigned int size() const {  // number of words
    int w = 0;
    if (h_) orbv_info(h_, nullptr, nullptr, nullptr, &w);
    return (unsigned)w;
  }

  // transform(features, v, fv, levelsup) (:1127-1194)
  void transform(const orbx_cv::Mat& descriptors, BowVector& v, FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    const int n = descriptors.rows;
    if (!h_ || n <= 0) return;
    std::vector<uint32_t> wid(n), nid(n), fidx(n);
    std::vector<double>   wval(n);
    std::vector<int>      noff(n + 1);
    int                   nw = 0, nn = 0;
    if (orbv_transform(h_, descriptors.ptr(0), n, levelsup, wid.data(), wval.data(), &nw, nid.data(), noff.data(), fidx.data(), &nn) != 0)
      throw std::runtime_error(std::string("orbv_transform: ") + orbv_last_error());
    for (int i = 0; i < nw; i++) v.emplace_hint(v.end(), wid[i], wval[i]);
    for (int i = 0; i < nn; i++) fv.emplace_hint(fv.end(), nid[i], std::vector<unsigned int>(fidx.begin() + noff[i], fidx.begin() + noff[i + 1]));
  }

  orbv_t* handle() const { return h_; }

 private:
  orbv_t* h_ = nullptr;
  int     device_;
};

}  // namespace ORB_SLAM3_AMD
#endif
