// ORBmatcher.h -- C++ adaptor for the matcher inner loops (reference include/ORBmatcher.h:35-148) over orbm_*.
// The reference's Search* methods take Frame / KeyFrame objects; the array-level cores they all share are
// exposed here with the reference's constants and semantics.  INTEGRATION.md shows how
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) forwards to SearchByBoWArrays().
#ifndef ORBSLAM3_AMD_ORBMATCHER_H_
#define ORBSLAM3_AMD_ORBMATCHER_H_

#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../orbx.h"
#include "cv_compat.h"

// Own namespace: the reference keeps its ORB_SLAM3::ORBmatcher (Frame/KeyFrame-level methods) and forwards to this one.
namespace ORB_SLAM3_AMD {

class ORBmatcher {
 public:
  static constexpr int TH_LOW       = 50;   // src/ORBmatcher.cc:31
  static constexpr int TH_HIGH      = 100;  // src/ORBmatcher.cc:30
  static constexpr int HISTO_LENGTH = 30;   // src/ORBmatcher.cc:32

  ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device_id = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    if (orbm_create(&h_, device_id) != 0) throw std::runtime_error(std::string("orbm_create: ") + orbx_last_error());
  }
  ~ORBmatcher() { orbm_destroy(h_); }
  ORBmatcher(const ORBmatcher&)            = delete;
  ORBmatcher& operator=(const ORBmatcher&) = delete;

  // Computes the Hamming distance between two ORB descriptors (reference :40)
  static int DescriptorDistance(const orbx_cv::Mat& a, const orbx_cv::Mat& b) { return orbm_hamming(a.ptr(0), b.ptr(0)); }

  // DBoW2::FeatureVector is std::map<NodeId, std::vector<unsigned int>>
  typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;

  // Core of SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) (src/ORBmatcher.cc:237-421):
  // kfHasMapPoint[i] = (vpMapPointsKF[i] && !isBad()); matchKFofF[j] = KeyFrame feature whose MapPoint frame
  // feature j receives, -1 = NULL.  strictLess selects the KeyFrame-KeyFrame overload's `< TH_LOW` (:864).
  // fNleft = F.Nleft: -1 for a one-camera frame; otherwise descF / angleF hold the left keys [0, Nleft) followed by the right keys
  // (as F.mDescriptors does) and the two-camera branch (:305-333, 362-389) runs -- a KeyFrame feature may then hand its MapPoint to
  // a left AND a right frame feature, so matchFofKF is not filled in that mode (fHasMapPoint / strictLess do not apply either).
  int SearchByBoWArrays(const orbx_cv::Mat& descKF, const std::vector<float>& angleKF, const FeatureVector& fvKF,
                        const std::vector<unsigned char>& kfHasMapPoint, const orbx_cv::Mat& descF, const std::vector<float>& angleF,
                        const FeatureVector& fvF, std::vector<int>& matchKFofF, const std::vector<unsigned char>* fHasMapPoint = nullptr,
                        bool strictLess = false, std::vector<int>* matchFofKF = nullptr, int fNleft = -1) {
    std::vector<int> qoff(1, 0), qidx, toff(1, 0), tidx;
    FeatureVector::const_iterator a = fvKF.begin(), b = fvF.begin();
    while (a != fvKF.end() && b != fvF.end()) {  // merge walk of :259-399
      if (a->first == b->first) {
        qidx.insert(qidx.end(), a->second.begin(), a->second.end());
        tidx.insert(tidx.end(), b->second.begin(), b->second.end());
        qoff.push_back((int)qidx.size()), toff.push_back((int)tidx.size());
        ++a, ++b;
      } else if (a->first < b->first) {
        a = fvKF.lower_bound(b->first);
      } else {
        b = fvF.lower_bound(a->first);
      }
    }
    matchKFofF.assign(descF.rows, -1);
    if (matchFofKF) matchFofKF->assign(descKF.rows, -1);
    const int ng = (int)qoff.size() - 1;
    if (ng == 0) return 0;
    std::vector<uint8_t> q = pack(descKF), t = pack(descF);
    if (fNleft != -1) {
      int n2 = orbm_match_groups_two_cameras(h_, q.data(), descKF.rows, t.data(), descF.rows, fNleft, ng, qoff.data(), qidx.data(), toff.data(),
                                             tidx.data(), kfHasMapPoint.empty() ? nullptr : kfHasMapPoint.data(), angleKF.data(), angleF.data(), TH_LOW,
                                             mfNNratio, mbCheckOrientation ? 1 : 0, matchKFofF.data());
      if (n2 <= -2) throw std::runtime_error(std::string("orbm_match_groups_two_cameras: ") + orbx_last_error());
      return n2;
    }
    int n = orbm_match_groups(h_, q.data(), descKF.rows, t.data(), descF.rows, ng, qoff.data(), qidx.data(), toff.data(), tidx.data(),
                              kfHasMapPoint.empty() ? nullptr : kfHasMapPoint.data(), fHasMapPoint ? fHasMapPoint->data() : nullptr,
                              angleKF.data(), angleF.data(), TH_LOW, strictLess ? 1 : 0, mfNNratio, mbCheckOrientation ? 1 : 0,
                              matchFofKF ? matchFofKF->data() : nullptr, matchKFofF.data());
    if (n <= -2) throw std::runtime_error(std::string("orbm_match_groups: ") + orbx_last_error());
    return n;
  }

  // SearchForInitialization (src/ORBmatcher.cc:666-782); bounds = Frame::mnMinX, mnMaxX, mnMinY, mnMaxY.
  int SearchForInitializationArrays(const std::vector<orbx_cv::KeyPoint>& keysUn1, const orbx_cv::Mat& desc1,
                                    const std::vector<orbx_cv::KeyPoint>& keysUn2, const orbx_cv::Mat& desc2,
                                    std::vector<orbx_cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, const float bounds[4],
                                    int windowSize = 10) {
    vnMatches12.assign(keysUn1.size(), -1);
    std::vector<uint8_t> d1 = pack(desc1), d2 = pack(desc2);
    int n = orbm_match_window(h_, d1.data(), reinterpret_cast<const orbx_keypoint_t*>(keysUn1.data()), (int)keysUn1.size(), d2.data(),
                              reinterpret_cast<const orbx_keypoint_t*>(keysUn2.data()), (int)keysUn2.size(),
                              reinterpret_cast<float*>(vbPrevMatched.data()), bounds[0], bounds[1], bounds[2], bounds[3], windowSize, TH_LOW,
                              mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data());
    if (n <= -2) throw std::runtime_error(std::string("orbm_match_window: ") + orbx_last_error());
    return n;
  }

  // Array core of SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1684-1908,
  // Nleft == -1).  The caller keeps the loop head (:1706-1729): lastValid[i] = LastFrame.mvpMapPoints[i] && !mvbOutlier[i] &&
  // invzc >= 0 && uv inside the image bounds; lastUV / lastInvZ = projection and 1/z; lastDesc row i = pMP->GetDescriptor();
  // lastKeys = LastFrame.mvKeysUn (octave, angle).  curOccupied[i2] = CurrentFrame.mvpMapPoints[i2] with Observations() > 0.
  // curOfLast[i] = the CurrentFrame feature that receives LastFrame.mvpMapPoints[i] (-1 = none).  direction: 0 neither,
  // 1 bForward, 2 bBackward (:1701-1702).  INTEGRATION.md shows the forwarding body.
  int SearchByProjectionArrays(const orbx_cv::Mat& lastDesc, const std::vector<orbx_cv::Point2f>& lastUV, const std::vector<float>& lastInvZ,
                               const std::vector<orbx_cv::KeyPoint>& lastKeys, const std::vector<unsigned char>& lastValid,
                               const orbx_cv::Mat& curDesc, const std::vector<orbx_cv::KeyPoint>& curKeysUn, const std::vector<float>* curURight,
                               const std::vector<unsigned char>* curOccupied, const std::vector<float>& scaleFactors, float th, int direction,
                               float mbf, const float bounds[4], std::vector<int>& curOfLast) {
    const int n1 = (int)lastKeys.size(), n2 = (int)curKeysUn.size();
    curOfLast.assign(n1, -1);
    std::vector<uint8_t> d1 = pack(lastDesc), d2 = pack(curDesc);
    std::vector<int>     oct(n1);
    std::vector<float>   ang(n1);
    for (int i = 0; i < n1; i++) oct[i] = lastKeys[i].octave, ang[i] = lastKeys[i].angle;
    int n = orbm_search_by_projection(h_, d1.data(), reinterpret_cast<const float*>(lastUV.data()), lastInvZ.empty() ? nullptr : lastInvZ.data(),
                                      oct.data(), ang.data(), lastValid.empty() ? nullptr : lastValid.data(), n1, d2.data(),
                                      reinterpret_cast<const orbx_keypoint_t*>(curKeysUn.data()), curURight ? curURight->data() : nullptr,
                                      curOccupied ? curOccupied->data() : nullptr, n2, scaleFactors.data(), (int)scaleFactors.size(), th, direction,
                                      mbf, bounds[0], bounds[1], bounds[2], bounds[3], TH_HIGH, mbCheckOrientation ? 1 : 0, curOfLast.data(), nullptr);
    if (n <= -2) throw std::runtime_error(std::string("orbm_search_by_projection: ") + orbx_last_error());
    return n;
  }

  // Array core of SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
  // (src/ORBmatcher.cc:38-227, Nleft == -1).  Per MapPoint: mpValid = mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth >
  // thFarPoints); mpUV = (mTrackProjX, mTrackProjY); mpProjXR = mTrackProjXR (read only with curURight); mpLevel = mnTrackScaleLevel;
  // mpRadius = RadiusByViewingCos(mTrackViewCos) (* th when th != 1).  curOfMp[i] = the feature that receives MapPoint i.
  int SearchByProjectionMapPointsArrays(const orbx_cv::Mat& mpDesc, const std::vector<orbx_cv::Point2f>& mpUV, const std::vector<float>& mpProjXR,
                                        const std::vector<int>& mpLevel, const std::vector<float>& mpRadius, const std::vector<unsigned char>& mpValid,
                                        const orbx_cv::Mat& curDesc, const std::vector<orbx_cv::KeyPoint>& curKeysUn, const std::vector<float>* curURight,
                                        const std::vector<unsigned char>* curOccupied, const std::vector<float>& scaleFactors, const float bounds[4],
                                        std::vector<int>& curOfMp) {
    const int n1 = (int)mpLevel.size(), n2 = (int)curKeysUn.size();
    curOfMp.assign(n1, -1);
    std::vector<uint8_t> d1 = pack(mpDesc), d2 = pack(curDesc);
    int n = orbm_search_by_projection_map(h_, d1.data(), reinterpret_cast<const float*>(mpUV.data()), mpProjXR.empty() ? nullptr : mpProjXR.data(),
                                          mpLevel.data(), mpRadius.data(), mpValid.empty() ? nullptr : mpValid.data(), n1, d2.data(),
                                          reinterpret_cast<const orbx_keypoint_t*>(curKeysUn.data()), curURight ? curURight->data() : nullptr,
                                          curOccupied ? curOccupied->data() : nullptr, n2, scaleFactors.data(), (int)scaleFactors.size(), mfNNratio,
                                          bounds[0], bounds[1], bounds[2], bounds[3], TH_HIGH, curOfMp.data(), nullptr);
    if (n <= -2) throw std::runtime_error(std::string("orbm_search_by_projection_map: ") + orbx_last_error());
    return n;
  }

  // The same search for a two-camera frame (F.Nleft != -1, src/ORBmatcher.cc:139-146, 152-226): one set of per-point arrays per camera
  // (the ...R members of MapPoint for the right one; radius without the th factor there, :158); curDesc / curKeys = left keys [0, nleft)
  // then right keys (mvKeys, mvKeysRight); leftToRight / rightToLeft = F.mvLeftToRightMatch / mvRightToLeftMatch.  mpOfCur[j] = map point
  // that F.mvpMapPoints[j] received (-1 = untouched): one point may land on up to four features.
  struct CameraView {
    std::vector<orbx_cv::Point2f> uv;
    std::vector<int>              level;
    std::vector<float>            radius;
    std::vector<unsigned char>    valid;
  };
  int SearchByProjectionMapPointsTwoCamerasArrays(const orbx_cv::Mat& mpDesc, const CameraView& left, const CameraView& right, const orbx_cv::Mat& curDesc,
                                                  const std::vector<orbx_cv::KeyPoint>& curKeys, int nleft, const std::vector<unsigned char>* curOccupied,
                                                  const std::vector<int>* leftToRight, const std::vector<int>* rightToLeft,
                                                  const std::vector<float>& scaleFactors, const float bounds[4], std::vector<int>& mpOfCur) {
    const int n1 = (int)left.level.size(), n = (int)curKeys.size();
    mpOfCur.assign(n, -1);
    std::vector<uint8_t> d1 = pack(mpDesc), d2 = pack(curDesc);
    int nm = orbm_search_by_projection_map_two_cameras(
        h_, d1.data(), n1, reinterpret_cast<const float*>(left.uv.data()), left.level.data(), left.radius.data(), left.valid.data(),
        reinterpret_cast<const float*>(right.uv.data()), right.level.data(), right.radius.data(), right.valid.data(), d2.data(),
        reinterpret_cast<const orbx_keypoint_t*>(curKeys.data()), n, nleft, curOccupied ? curOccupied->data() : nullptr,
        leftToRight ? leftToRight->data() : nullptr, rightToLeft ? rightToLeft->data() : nullptr, scaleFactors.data(), (int)scaleFactors.size(), mfNNratio,
        bounds[0], bounds[1], bounds[2], bounds[3], TH_HIGH, mpOfCur.data());
    if (nm <= -2) throw std::runtime_error(std::string("orbm_search_by_projection_map_two_cameras: ") + orbx_last_error());
    return nm;
  }

  // Array core of SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, th, ORBdist)
  // (src/ORBmatcher.cc:1910-2036), the relocalisation search: the Frame-to-Frame core on predicted levels.  mpLevel[i] =
  // PredictScale(dist3D, &CurrentFrame), kfAngle[i] = pKF->mvKeysUn[i].angle, mpValid[i] = MapPoint present, not bad, not in
  // sAlreadyFound, inside the bounds and the distance range; curOccupied[i2] = (CurrentFrame.mvpMapPoints[i2] != NULL).
  int SearchByProjectionKeyFrameArrays(const orbx_cv::Mat& mpDesc, const std::vector<orbx_cv::Point2f>& mpUV, const std::vector<int>& mpLevel,
                                       const std::vector<float>& kfAngle, const std::vector<unsigned char>& mpValid, const orbx_cv::Mat& curDesc,
                                       const std::vector<orbx_cv::KeyPoint>& curKeysUn, const std::vector<unsigned char>* curOccupied,
                                       const std::vector<float>& scaleFactors, float th, int ORBdist, const float bounds[4], std::vector<int>& curOfMp) {
    const int n1 = (int)mpLevel.size(), n2 = (int)curKeysUn.size();
    curOfMp.assign(n1, -1);
    std::vector<uint8_t> d1 = pack(mpDesc), d2 = pack(curDesc);
    int n = orbm_search_by_projection(h_, d1.data(), reinterpret_cast<const float*>(mpUV.data()), nullptr, mpLevel.data(), kfAngle.data(),
                                      mpValid.empty() ? nullptr : mpValid.data(), n1, d2.data(), reinterpret_cast<const orbx_keypoint_t*>(curKeysUn.data()),
                                      nullptr, curOccupied ? curOccupied->data() : nullptr, n2, scaleFactors.data(), (int)scaleFactors.size(), th, 0, 0.f,
                                      bounds[0], bounds[1], bounds[2], bounds[3], ORBdist, mbCheckOrientation ? 1 : 0, curOfMp.data(), nullptr);
    if (n <= -2) throw std::runtime_error(std::string("orbm_search_by_projection: ") + orbx_last_error());
    return n;
  }

  // Array core of the remaining windowed searches (see orbm_search_in_windows in orbx.h):
  //   SearchByProjection(KeyFrame*, Sim3f&, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher.cc:423-538, :540-664):
  //     claim = true, kfOccupied[idx] = (vpMatched[idx] != NULL), thDist = (int)std::floor(TH_LOW * ratioHamming)
  //   Fuse(KeyFrame*, vpMapPoints, th, bRight) (:1149-1328): claim = false, invLevelSigma2 = &pKF->mvInvLevelSigma2, thDist = TH_LOW
  //     (stereo keyframes: mpUR[i] = uv(0) - bf * invz, kfURight = &pKF->mvuRight)
  //   Fuse(KeyFrame*, Sim3f&, vpPoints, th, vpReplacePoint) (:1330-1452): claim = false, thDist = TH_LOW
  //   each direction of SearchBySim3 (:1499-1658): claim = false, thDist = TH_HIGH
  // kfOfMp[i] = the keyframe feature chosen for point i (-1 = none); returns the number of matched points.
  int SearchInWindowsArrays(const orbx_cv::Mat& mpDesc, const std::vector<orbx_cv::Point2f>& mpUV, const std::vector<float>* mpUR,
                            const std::vector<int>& mpLevel, const std::vector<unsigned char>& mpValid, const orbx_cv::Mat& kfDesc,
                            const std::vector<orbx_cv::KeyPoint>& kfKeysUn, const std::vector<float>* kfURight,
                            const std::vector<unsigned char>* kfOccupied, const std::vector<float>& scaleFactors,
                            const std::vector<float>* invLevelSigma2, float th, bool claim, int thDist, const float bounds[4], std::vector<int>& kfOfMp) {
    const int n1 = (int)mpLevel.size(), n2 = (int)kfKeysUn.size();
    kfOfMp.assign(n1, -1);
    std::vector<uint8_t> d1 = pack(mpDesc), d2 = pack(kfDesc);
    int n = orbm_search_in_windows(h_, d1.data(), reinterpret_cast<const float*>(mpUV.data()), mpUR ? mpUR->data() : nullptr, mpLevel.data(),
                                   mpValid.empty() ? nullptr : mpValid.data(), n1, d2.data(), reinterpret_cast<const orbx_keypoint_t*>(kfKeysUn.data()),
                                   kfURight ? kfURight->data() : nullptr, kfOccupied ? kfOccupied->data() : nullptr, n2, scaleFactors.data(),
                                   invLevelSigma2 ? invLevelSigma2->data() : nullptr, (int)scaleFactors.size(), th, claim ? 1 : 0, bounds[0], bounds[1],
                                   bounds[2], bounds[3], thDist, kfOfMp.data());
    if (n <= -2) throw std::runtime_error(std::string("orbm_search_in_windows: ") + orbx_last_error());
    return n;
  }

  // Array core of SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:916-1147) for keyframes of one
  // pinhole camera.  The common nodes of the two FeatureVectors are flattened by the caller exactly as for SearchByBoW (qOff / qIdx / tOff /
  // tIdx); hasMp1 / hasMp2[i] = pKF->GetMapPoint(i) != NULL; stereo1 / stereo2[i] = mvuRight[i] >= 0 (NULL = monocular); F12 = K1^-T [t12]x R12
  // K2^-1 (row major) and ep as the reference computes them.  vMatchedPairs comes back in ascending first index.
  int SearchForTriangulationArrays(const orbx_cv::Mat& desc1, const std::vector<orbx_cv::KeyPoint>& keysUn1, const std::vector<unsigned char>& hasMp1,
                                   const std::vector<unsigned char>* stereo1, const orbx_cv::Mat& desc2, const std::vector<orbx_cv::KeyPoint>& keysUn2,
                                   const std::vector<unsigned char>& hasMp2, const std::vector<unsigned char>* stereo2, const std::vector<int>& qOff,
                                   const std::vector<int>& qIdx, const std::vector<int>& tOff, const std::vector<int>& tIdx, const float F12[9],
                                   const float ep[2], const std::vector<float>& scaleFactors, const std::vector<float>& levelSigma2, bool bOnlyStereo,
                                   bool bCoarse, std::vector<std::pair<std::size_t, std::size_t> >& vMatchedPairs) {
    const int n1 = (int)keysUn1.size(), n2 = (int)keysUn2.size();
    std::vector<uint8_t> d1 = pack(desc1), d2 = pack(desc2);
    std::vector<int>     m12(n1 + 1, -1);
    int n = orbm_search_for_triangulation(h_, d1.data(), reinterpret_cast<const orbx_keypoint_t*>(keysUn1.data()), hasMp1.empty() ? nullptr : hasMp1.data(),
                                          stereo1 ? stereo1->data() : nullptr, n1, d2.data(), reinterpret_cast<const orbx_keypoint_t*>(keysUn2.data()),
                                          hasMp2.empty() ? nullptr : hasMp2.data(), stereo2 ? stereo2->data() : nullptr, n2, (int)qOff.size() - 1, qOff.data(),
                                          qIdx.data(), tOff.data(), tIdx.data(), F12, ep, scaleFactors.data(), levelSigma2.data(), (int)scaleFactors.size(),
                                          bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, TH_LOW, mbCheckOrientation ? 1 : 0, m12.data());
    if (n <= -2) throw std::runtime_error(std::string("orbm_search_for_triangulation: ") + orbx_last_error());
    vMatchedPairs.clear();
    vMatchedPairs.reserve(n);
    for (int i = 0; i < n1; i++)
      if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((std::size_t)i, (std::size_t)m12[i]));
    return n;
  }

  orbm_t* handle() { return h_; }

  // The reference constructs its ORBmatcher on the stack at every call site (ORBmatcher matcher(0.9, true), src/Tracking.cc and
  // others).  A device handle per such object would create a stream and buffers each time: the forwarding bodies use ONE handle
  // per host thread instead and set the call site's parameters on it.
  void SetParameters(float nnratio, bool checkOri) { mfNNratio = nnratio, mbCheckOrientation = checkOri; }
  static ORBmatcher& ThreadInstance(float nnratio, bool checkOri, int device_id = 0) {
    static thread_local ORBmatcher inst(nnratio, checkOri, device_id);
    inst.SetParameters(nnratio, checkOri);
    return inst;
  }

 protected:
  float mfNNratio;
  bool  mbCheckOrientation;

 private:
  static std::vector<uint8_t> pack(const orbx_cv::Mat& m) {  // rows may be strided: make them contiguous
    std::vector<uint8_t> v((size_t)m.rows * 32 + 32);
    for (int r = 0; r < m.rows; r++) std::memcpy(v.data() + (size_t)r * 32, m.ptr(r), 32);
    return v;
  }
  orbm_t* h_ = nullptr;
};

}  // namespace ORB_SLAM3_AMD
#endif
