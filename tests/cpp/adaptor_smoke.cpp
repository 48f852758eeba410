// Drives the C++ adaptors exactly as Frame::ExtractORB (reference src/Frame.cc:555-562) and
// Tracking::TrackReferenceKeyFrame (src/Tracking.cc:2603-2606) drive the reference classes, on a raw PGM-less
// buffer passed on stdin-free: argv = <rows> <cols> <raw u8 file A> <raw u8 file B> <out file>.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbslam3_amd/ORBextractor.h"
#include "orbslam3_amd/ORBmatcher.h"
#include "orbslam3_amd/ORBVocabulary.h"

static std::vector<unsigned char> slurp(const char* p, size_t n) {
  std::vector<unsigned char> v(n);
  FILE*                      f = fopen(p, "rb");
  if (!f || fread(v.data(), 1, n, f) != n) {
    fprintf(stderr, "cannot read %s\n", p);
    exit(2);
  }
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int rows = atoi(argv[1]), cols = atoi(argv[2]);
  std::vector<unsigned char> A = slurp(argv[3], (size_t)rows * cols), B = slurp(argv[4], (size_t)rows * cols);
  ORB_SLAM3::ORBextractor*   ext = new ORB_SLAM3::ORBextractor(1000, 1.2f, 8, 20, 7);  // src/Tracking.cc:576
  std::vector<int>           vLapping = {0, 1000};                                      // mono convention, src/Frame.cc:451
  orbx_cv::Mat               imA(rows, cols, orbx_cv::CV_8U, A.data()), imB(rows, cols, orbx_cv::CV_8U, B.data()), dA, dB;
  std::vector<orbx_cv::KeyPoint> kA, kB;
  int monoA = (*ext)(imA, orbx_cv::Mat(), kA, dA, vLapping);
  int monoB = (*ext)(imB, orbx_cv::Mat(), kB, dB, vLapping);
  std::vector<float> sf = ext->GetScaleFactors();
  // brute-force SearchByBoW: one vocabulary node holding every feature
  ORB_SLAM3_AMD::ORBmatcher                matcher(0.7f, true);
  ORB_SLAM3_AMD::ORBmatcher::FeatureVector fA, fB;
  for (size_t i = 0; i < kA.size(); i++) fA[0].push_back((unsigned)i);
  for (size_t i = 0; i < kB.size(); i++) fB[0].push_back((unsigned)i);
  std::vector<float> aA(kA.size()), aB(kB.size());
  for (size_t i = 0; i < kA.size(); i++) aA[i] = kA[i].angle;
  for (size_t i = 0; i < kB.size(); i++) aB[i] = kB[i].angle;
  std::vector<int>           match;
  std::vector<unsigned char> has(kA.size(), 1);
  int                        nm = matcher.SearchByBoWArrays(dA, aA, fA, has, dB, aB, fB, match);
  FILE*                      o  = fopen(argv[5], "wb");
  fprintf(o, "%d %d %zu %zu %d %d %d %.9g %d\n", monoA, monoB, kA.size(), kB.size(), nm, ext->GetLevels(), ext->mvImagePyramid[7].cols, sf[7],
          orbx_cv::Mat(dA.row(0)).cols);
  for (size_t i = 0; i < kA.size(); i++) {
    fprintf(o, "%.9g %.9g %.9g %.9g %d", kA[i].pt.x, kA[i].pt.y, kA[i].angle, kA[i].response, kA[i].octave);
    for (int b = 0; b < 32; b++) fprintf(o, " %d", dA.ptr((int)i)[b]);
    fprintf(o, "\n");
  }
  for (size_t j = 0; j < match.size(); j++) fprintf(o, "%d\n", match[j]);
  if (argc > 6) {  // Frame::ComputeBoW on both frames (src/Frame.cc:879-884), then SearchByBoW over the FeatureVector groups
    ORB_SLAM3_AMD::ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[6])) return 3;
    ORB_SLAM3_AMD::ORBVocabulary::BowVector     bA, bB;
    ORB_SLAM3_AMD::ORBVocabulary::FeatureVector gA, gB;
    voc.transform(dA, bA, gA, 1);
    voc.transform(dB, bB, gB, 1);
    std::vector<int> match2;
    const int        nm2 = matcher.SearchByBoWArrays(dA, aA, gA, has, dB, aB, gB, match2);
    fprintf(o, "%u %zu %zu %zu %d\n", voc.size(), bA.size(), gA.size(), gB.size(), nm2);
    for (auto& kv : bA) fprintf(o, "%u %a\n", kv.first, kv.second);
    for (auto& kv : gA) {
      fprintf(o, "%u", kv.first);
      for (unsigned f : kv.second) fprintf(o, " %u", f);
      fprintf(o, "\n");
    }
    for (size_t j = 0; j < match2.size(); j++) fprintf(o, "%d\n", match2[j]);
    // the projection family on arrays: frame A's keypoints "projected" into frame B by a fixed shift
    const int                     n1 = (int)kA.size();
    std::vector<orbx_cv::Point2f> uv(n1);
    std::vector<int>              lvl(n1);
    std::vector<float>            invz, radius(n1, 4.0f * 3.0f), s2(sf.size()), sg(sf.size());
    std::vector<unsigned char>    valid;
    for (int i = 0; i < n1; i++) uv[i] = orbx_cv::Point2f(kA[i].pt.x + 3.0f, kA[i].pt.y + 1.0f), lvl[i] = kA[i].octave;
    for (size_t l = 0; l < sf.size(); l++) sg[l] = sf[l] * sf[l], s2[l] = 1.0f / sg[l];
    const float      bounds[4] = {0.f, (float)cols, 0.f, (float)rows};
    std::vector<int> r;
    auto             dump = [&](int n) {
      fprintf(o, "%d", n);
      for (int v : r) fprintf(o, " %d", v);
      fprintf(o, "\n");
    };
    dump(matcher.SearchByProjectionArrays(dA, uv, invz, kA, valid, dB, kB, nullptr, nullptr, sf, 15.0f, 0, 0.f, bounds, r));
    dump(matcher.SearchByProjectionMapPointsArrays(dA, uv, invz, lvl, radius, valid, dB, kB, nullptr, nullptr, sf, bounds, r));
    dump(matcher.SearchByProjectionKeyFrameArrays(dA, uv, lvl, aA, valid, dB, kB, nullptr, sf, 10.0f, 64, bounds, r));
    dump(matcher.SearchInWindowsArrays(dA, uv, nullptr, lvl, valid, dB, kB, nullptr, nullptr, sf, nullptr, 4.0f, true, 50, bounds, r));
    dump(matcher.SearchInWindowsArrays(dA, uv, nullptr, lvl, valid, dB, kB, nullptr, nullptr, sf, &s2, 3.0f, false, 50, bounds, r));
    std::vector<int> qOff(1, 0), qIdx, tOff(1, 0), tIdx;  // common nodes of the two FeatureVectors (merge walk of SearchByBoW)
    for (auto& kv : gA) {
      auto it = gB.find(kv.first);
      if (it == gB.end()) continue;
      for (unsigned f : kv.second) qIdx.push_back((int)f);
      for (unsigned f : it->second) tIdx.push_back((int)f);
      qOff.push_back((int)qIdx.size()), tOff.push_back((int)tIdx.size());
    }
    // F12 for a pure translation along x with fx = fy = 460, cx = 320, cy = 240: K^-T [t]x K^-1
    const float F12[9] = {0.f, 0.f, 0.f, 0.f, 0.f, -1.0f / 460.0f, 0.f, 1.0f / 460.0f, 0.f};
    const float ep[2]  = {1e6f, 240.f};
    std::vector<std::pair<std::size_t, std::size_t> > pairs;
    std::vector<unsigned char>                        none;
    const int nt = matcher.SearchForTriangulationArrays(dA, kA, none, nullptr, dB, kB, none, nullptr, qOff, qIdx, tOff, tIdx, F12, ep, sf, sg, false, false, pairs);
    fprintf(o, "%d", nt);
    for (auto& pr : pairs) fprintf(o, " %zu %zu", pr.first, pr.second);
    fprintf(o, "\n");
  }
  fclose(o);
  delete ext;
  return 0;
}
