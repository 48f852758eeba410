// reference_matcher_bodies.cpp -- the reference-side forwarding bodies of INTEGRATION.md section 3, COMPILED and RUN.
//
// ORB_SLAM3::ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference src/ORBmatcher.cc:237-421), ::SearchForInitialization
// (:666-782) and ::SearchByProjection(Frame&, const Frame&, th, bMono) (:1684-1908) keep their signatures; their bodies become
// flatten -> ORB_SLAM3_AMD::ORBmatcher -> map back.  The real Frame / KeyFrame / MapPoint need Eigen, Sophus, DBoW2 and OpenCV, none
// of which exists in this image, so the bodies are compiled here against MOCK classes that carry exactly the members those bodies
// touch, under the reference's member names (include/Frame.h:183-366, include/KeyFrame.h, include/MapPoint.h).  The mock pose /
// camera arithmetic is plain float code; tests/test_cpp_adaptor.py feeds the dumped projections to the oracle, so nothing here
// needs to agree with Eigen's evaluation order.
//
//   reference_matcher_bodies rows cols frameA.raw frameB.raw out.txt vocab.txt
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include <orbslam3_amd/ORBVocabulary.h>
#include <orbslam3_amd/ORBextractor.h>
#include <orbslam3_amd/ORBmatcher.h>

namespace cv = orbx_cv;

// ---- stand-ins for the third-party value types the three bodies mention -------------------------------------------------------
namespace Eigen {
struct Vector3f {
  float v[3];
  float operator()(int i) const { return v[i]; }
};
struct Vector2f {
  float v[2];
  float operator()(int i) const { return v[i]; }
};
}  // namespace Eigen
namespace Sophus {
struct SE3f {  // rotation (row major) + translation
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  Eigen::Vector3f operator*(const Eigen::Vector3f& p) const {
    Eigen::Vector3f q;
    for (int i = 0; i < 3; i++) q.v[i] = R[3 * i] * p.v[0] + R[3 * i + 1] * p.v[1] + R[3 * i + 2] * p.v[2] + t[i];
    return q;
  }
  SE3f inverse() const {
    SE3f o;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) o.R[3 * i + j] = R[3 * j + i];
    for (int i = 0; i < 3; i++) o.t[i] = -(o.R[3 * i] * t[0] + o.R[3 * i + 1] * t[1] + o.R[3 * i + 2] * t[2]);
    return o;
  }
  Eigen::Vector3f translation() const { return Eigen::Vector3f{{t[0], t[1], t[2]}}; }
};
}  // namespace Sophus
namespace DBoW2 {
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;
typedef std::map<unsigned int, double>                    BowVector;
}  // namespace DBoW2

namespace ORB_SLAM3 {

class GeometricCamera {  // include/CameraModels/GeometricCamera.h: only project() is used by these bodies (Pinhole.cpp:59-66)
 public:
  float           fx = 460.f, fy = 460.f, cx = 320.f, cy = 240.f;
  Eigen::Vector2f project(const Eigen::Vector3f& p) const { return Eigen::Vector2f{{fx * p(0) / p(2) + cx, fy * p(1) / p(2) + cy}}; }
};

class MapPoint {  // include/MapPoint.h
 public:
  bool            mbBad = false;
  int             nObs  = 1;
  cv::Mat         mDescriptor;
  Eigen::Vector3f mWorldPos{{0, 0, 1}};
  bool            isBad() const { return mbBad; }
  int             Observations() const { return nObs; }
  cv::Mat         GetDescriptor() const { return mDescriptor; }
  Eigen::Vector3f GetWorldPos() const { return mWorldPos; }
};

class KeyFrame {  // include/KeyFrame.h
 public:
  std::vector<cv::KeyPoint> mvKeysUn;
  cv::Mat                   mDescriptors;
  DBoW2::FeatureVector      mFeatVec;
  std::vector<MapPoint*>    mvpMapPoints;
  std::vector<MapPoint*>    GetMapPointMatches() const { return mvpMapPoints; }
};

class Frame {  // include/Frame.h:183-366
 public:
  int                       N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
  std::vector<float>        mvuRight;
  cv::Mat                   mDescriptors;
  DBoW2::BowVector          mBowVec;
  DBoW2::FeatureVector      mFeatVec;
  std::vector<MapPoint*>    mvpMapPoints;
  std::vector<bool>         mvbOutlier;
  std::vector<float>        mvScaleFactors;
  float                     mbf = 0.f, mb = 0.f;
  int                       Nleft = -1;
  GeometricCamera*          mpCamera = nullptr;
  Sophus::SE3f              mTcw;
  Sophus::SE3f              GetPose() const { return mTcw; }
  static float              mnMinX, mnMaxX, mnMinY, mnMaxY;
};
float Frame::mnMinX = 0.f, Frame::mnMaxX = 0.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 0.f;

class ORBmatcher {  // include/ORBmatcher.h:35-148: same constructor, same method signatures
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

  // last loop-head outputs, kept for the test dump only
  std::vector<cv::Point2f>   dbgUV;
  std::vector<unsigned char> dbgValid;

 protected:
  float mfNNratio;
  bool  mbCheckOrientation;
};

// ===== the bodies a maintainer puts into src/ORBmatcher.cc =========================================

int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
  const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
  std::vector<unsigned char>   has(vpMapPointsKF.size());
  for (size_t i = 0; i < has.size(); i++) has[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();  // :271-279
  std::vector<float> aKF(pKF->mvKeysUn.size()), aF(F.mvKeys.size());
  for (size_t i = 0; i < aKF.size(); i++) aKF[i] = pKF->mvKeysUn[i].angle;  // :338-346
  for (size_t i = 0; i < aF.size(); i++) aF[i] = F.mvKeys[i].angle;
  std::vector<int>           kfOfF;
  ORB_SLAM3_AMD::ORBmatcher& gpu = ORB_SLAM3_AMD::ORBmatcher::ThreadInstance(mfNNratio, mbCheckOrientation);
  const int n = gpu.SearchByBoWArrays(pKF->mDescriptors, aKF, pKF->mFeatVec, has, F.mDescriptors, aF, F.mFeatVec, kfOfF);
  vpMapPointMatches.assign(F.N, static_cast<MapPoint*>(NULL));
  for (int j = 0; j < F.N; j++)
    if (kfOfF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[kfOfF[j]];
  return n;
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
  const float                bounds[4] = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY};
  ORB_SLAM3_AMD::ORBmatcher& gpu       = ORB_SLAM3_AMD::ORBmatcher::ThreadInstance(mfNNratio, mbCheckOrientation);
  return gpu.SearchForInitializationArrays(F1.mvKeysUn, F1.mDescriptors, F2.mvKeysUn, F2.mDescriptors, vbPrevMatched, vnMatches12, bounds, windowSize);
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
  const Sophus::SE3f    Tcw = CurrentFrame.GetPose();  // :1697-1702, unchanged
  const Eigen::Vector3f twc = Tcw.inverse().translation();
  const Sophus::SE3f    Tlw = LastFrame.GetPose();
  const Eigen::Vector3f tlc = Tlw * twc;
  const bool            bForward  = tlc(2) > CurrentFrame.mb && !bMono;
  const bool            bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
  const int             N1 = LastFrame.N;
  std::vector<cv::Point2f>   uv(N1);
  std::vector<float>         invz(N1);
  std::vector<unsigned char> valid(N1, 0);
  cv::Mat                    dMP(N1, 32, cv::CV_8U);
  for (int i = 0; i < N1; i++) {  // :1704-1729, unchanged geometry
    for (int b = 0; b < 32; b++) dMP.ptr(i)[b] = 0;
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const Eigen::Vector3f x3Dc  = Tcw * pMP->GetWorldPos();
    const float           invzc = 1.0 / x3Dc(2);
    if (invzc < 0) continue;
    const Eigen::Vector2f p = CurrentFrame.mpCamera->project(x3Dc);
    if (p(0) < CurrentFrame.mnMinX || p(0) > CurrentFrame.mnMaxX || p(1) < CurrentFrame.mnMinY || p(1) > CurrentFrame.mnMaxY) continue;
    uv[i] = cv::Point2f(p(0), p(1)), invz[i] = invzc, valid[i] = 1;
    const cv::Mat d = pMP->GetDescriptor();
    for (int b = 0; b < 32; b++) dMP.ptr(i)[b] = d.ptr(0)[b];
  }
  std::vector<unsigned char> occ(CurrentFrame.N);
  for (int j = 0; j < CurrentFrame.N; j++) occ[j] = CurrentFrame.mvpMapPoints[j] && CurrentFrame.mvpMapPoints[j]->Observations() > 0;
  const float                bounds[4] = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY};
  std::vector<int>           curOfLast;
  ORB_SLAM3_AMD::ORBmatcher& gpu = ORB_SLAM3_AMD::ORBmatcher::ThreadInstance(mfNNratio, mbCheckOrientation);
  const int n = gpu.SearchByProjectionArrays(dMP, uv, invz, LastFrame.mvKeysUn, valid, CurrentFrame.mDescriptors, CurrentFrame.mvKeysUn,
                                             bMono ? nullptr : &CurrentFrame.mvuRight, &occ, CurrentFrame.mvScaleFactors, th,
                                             bForward ? 1 : (bBackward ? 2 : 0), CurrentFrame.mbf, bounds, curOfLast);
  for (int i = 0; i < N1; i++)
    if (curOfLast[i] >= 0) CurrentFrame.mvpMapPoints[curOfLast[i]] = LastFrame.mvpMapPoints[i];
  dbgUV = uv, dbgValid = valid;
  return n;
}

}  // namespace ORB_SLAM3

// ===== driver: two frames through the extractor adaptor, mock Frames / KeyFrame around them, the three bodies, a dump ===========
static bool readRaw(const char* path, int rows, int cols, cv::Mat& m) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  m.create(rows, cols, cv::CV_8U);
  const size_t n = fread(m.ptr(0), 1, (size_t)rows * cols, f);
  fclose(f);
  return n == (size_t)rows * cols;
}

int main(int argc, char** argv) {
  using namespace ORB_SLAM3;
  if (argc < 7) return 2;
  const int rows = atoi(argv[1]), cols = atoi(argv[2]);
  cv::Mat   imA, imB;
  if (!readRaw(argv[3], rows, cols, imA) || !readRaw(argv[4], rows, cols, imB)) return 3;
  ORBextractor ext(1000, 1.2f, 8, 20, 7);
  ORB_SLAM3_AMD::ORBVocabulary voc;
  if (!voc.loadFromTextFile(argv[6])) return 4;
  GeometricCamera cam;
  Frame::mnMinX = 0.f, Frame::mnMaxX = (float)cols, Frame::mnMinY = 0.f, Frame::mnMaxY = (float)rows;
  auto makeFrame = [&](const cv::Mat& im, Frame& F) {
    std::vector<int> lap = {0, 1000};  // src/Frame.cc:451
    ext(im, cv::Mat(), F.mvKeys, F.mDescriptors, lap);
    F.N = (int)F.mvKeys.size(), F.mvKeysUn = F.mvKeys;  // no distortion: mvKeysUn = mvKeys (src/Frame.cc:887-890)
    F.mvpMapPoints.assign(F.N, static_cast<MapPoint*>(NULL)), F.mvbOutlier.assign(F.N, false), F.mvuRight.assign(F.N, -1.f);
    F.mvScaleFactors = ext.GetScaleFactors(), F.mpCamera = &cam;
    voc.transform(F.mDescriptors, F.mBowVec, F.mFeatVec, 1);
  };
  Frame FA, FB;
  makeFrame(imA, FA), makeFrame(imB, FB);
  // every third feature of frame A owns a MapPoint on the ray through its pixel at depth 2..4, some are bad / outliers
  std::vector<MapPoint> points(FA.N);
  for (int i = 0; i < FA.N; i++) {
    MapPoint& p = points[i];
    p.mDescriptor = FA.mDescriptors.row(i);
    const float z = 2.0f + (float)(i % 5) * 0.5f;
    p.mWorldPos   = Eigen::Vector3f{{(FA.mvKeysUn[i].pt.x - cam.cx) / cam.fx * z, (FA.mvKeysUn[i].pt.y - cam.cy) / cam.fy * z, z}};
    p.mbBad       = (i % 11) == 0;
    if (i % 3 != 2) FA.mvpMapPoints[i] = &p;
    FA.mvbOutlier[i] = (i % 17) == 0;
  }
  KeyFrame KF;  // frame A promoted to a keyframe
  KF.mvKeysUn = FA.mvKeysUn, KF.mDescriptors = FA.mDescriptors, KF.mFeatVec = FA.mFeatVec, KF.mvpMapPoints = FA.mvpMapPoints;
  FILE* o = fopen(argv[5], "w");
  if (!o) return 5;
  // (1) SearchByBoW(KeyFrame*, Frame&): ratio 0.7 as Tracking::TrackReferenceKeyFrame
  {
    ORBmatcher             matcher(0.7f, true);
    std::vector<MapPoint*> vpMapPointMatches;
    const int              n = matcher.SearchByBoW(&KF, FB, vpMapPointMatches);
    fprintf(o, "%d %d %d\n", FA.N, FB.N, n);
    for (int j = 0; j < FB.N; j++) fprintf(o, "%d\n", vpMapPointMatches[j] ? (int)(vpMapPointMatches[j] - points.data()) : -1);
  }
  // (2) SearchForInitialization: ratio 0.9, window 100 as Tracking::MonocularInitialization
  {
    ORBmatcher               matcher(0.9f, true);
    std::vector<cv::Point2f> vbPrevMatched(FA.mvKeysUn.size());
    for (size_t i = 0; i < vbPrevMatched.size(); i++) vbPrevMatched[i] = FA.mvKeysUn[i].pt;
    std::vector<int> vnMatches12;
    const int        n = matcher.SearchForInitialization(FA, FB, vbPrevMatched, vnMatches12, 100);
    fprintf(o, "%d\n", n);
    for (size_t i = 0; i < vnMatches12.size(); i++) fprintf(o, "%d %a %a\n", vnMatches12[i], vbPrevMatched[i].x, vbPrevMatched[i].y);
  }
  // (3) SearchByProjection(CurrentFrame = B, LastFrame = A): camera moved 5 cm along x and 2 cm along z, monocular, th 15
  {
    FB.mTcw.t[0] = -0.05f, FB.mTcw.t[2] = -0.02f;
    ORBmatcher matcher(0.9f, true);
    const int  n = matcher.SearchByProjection(FB, FA, 15.0f, true);
    fprintf(o, "%d\n", n);
    for (int i = 0; i < FA.N; i++) fprintf(o, "%d %a %a\n", (int)matcher.dbgValid[i], matcher.dbgUV[i].x, matcher.dbgUV[i].y);
    for (int j = 0; j < FB.N; j++) fprintf(o, "%d\n", FB.mvpMapPoints[j] ? (int)(FB.mvpMapPoints[j] - points.data()) : -1);
  }
  fclose(o);
  return 0;
}
