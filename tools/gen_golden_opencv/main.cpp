// gen_golden_opencv -- runs the UNMODIFIED reference ORBextractor (linked from ORB_SLAM3_DIR/src/ORBextractor.cc) on exported
// frames and writes .npy files that tools/gen_golden_opencv/pack.py turns into tests/golden/opencv_<name>.npz.
//
//   gen_golden_opencv frames.npy outdir nfeatures scaleFactor nlevels iniThFAST minThFAST
//
// frames.npy: uint8 [n][rows][cols], C order (tools/gen_golden_opencv/export_frames.py).  Per frame f it writes
//   kps_lap1000_f<f>.npy / desc_lap1000_f<f>.npy / mono_lap1000_f<f>.npy   operator() with vLappingArea {0, 1000} (src/Frame.cc:451)
//   kps_lap0_f<f>.npy    / desc_lap0_f<f>.npy    / mono_lap0_f<f>.npy      ... with {0, 0} (src/Frame.cc:202-203)
//   pyr_f<f>_l<l>.npy    mvImagePyramid[l] (include/ORBextractor.h:87)
//   blur_f<f>_l<l>.npy   cv::GaussianBlur(level.clone(), 7x7, 2, 2, BORDER_REFLECT_101), the call of src/ORBextractor.cc:1103-1104
// plus meta.txt (OpenCV version, whether this binary was compiled with FMA contraction available, the parameters).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/core/version.hpp>
#include <opencv2/imgproc.hpp>

#include "ORBextractor.h"

static void write_npy(const std::string& path, const std::string& descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
  std::string sh = "(";
  for (size_t i = 0; i < shape.size(); i++) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  sh += ")";
  std::string hdr = "{'descr': " + descr + ", 'fortran_order': False, 'shape': " + sh + ", }";
  while ((10 + hdr.size() + 1) % 64) hdr += ' ';
  hdr += '\n';
  std::ofstream f(path, std::ios::binary);
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  f.write((const char*)magic, 8);
  const uint16_t hl = (uint16_t)hdr.size();
  f.write((const char*)&hl, 2);
  f.write(hdr.data(), hdr.size());
  f.write((const char*)data, bytes);
}

static bool read_frames(const char* path, std::vector<uint8_t>& data, int& n, int& rows, int& cols) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  char magic[10];
  f.read(magic, 10);
  const int hl = (unsigned char)magic[8] | ((unsigned char)magic[9] << 8);
  std::string hdr(hl, ' ');
  f.read(&hdr[0], hl);
  if (hdr.find("'|u1'") == std::string::npos && hdr.find("'u1'") == std::string::npos) return false;
  const size_t p = hdr.find("'shape': (");
  if (p == std::string::npos || sscanf(hdr.c_str() + p + 10, "%d, %d, %d", &n, &rows, &cols) != 3) return false;
  data.resize((size_t)n * rows * cols);
  f.read((char*)data.data(), data.size());
  return (bool)f;
}

int main(int argc, char** argv) {
  if (argc < 8) {
    std::cerr << "usage: gen_golden_opencv frames.npy outdir nfeatures scaleFactor nlevels iniThFAST minThFAST\n";
    return 2;
  }
  std::vector<uint8_t> frames;
  int                  n = 0, rows = 0, cols = 0;
  if (!read_frames(argv[1], frames, n, rows, cols)) {
    std::cerr << "cannot read " << argv[1] << " (expected a uint8 [n][rows][cols] .npy)\n";
    return 2;
  }
  const std::string out = argv[2];
  const int         nfeatures = atoi(argv[3]), nlevels = atoi(argv[5]), iniTh = atoi(argv[6]), minTh = atoi(argv[7]);
  const float       scaleFactor = (float)atof(argv[4]);
  static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint is expected to be the 7-field 28-byte record");
  const std::string kpDescr = "[('x', '<f4'), ('y', '<f4'), ('size', '<f4'), ('angle', '<f4'), ('response', '<f4'), ('octave', '<i4'), ('class_id', '<i4')]";
  ORB_SLAM3::ORBextractor ex(nfeatures, scaleFactor, nlevels, iniTh, minTh);
  for (int f = 0; f < n; f++) {
    cv::Mat im(rows, cols, CV_8UC1, frames.data() + (size_t)f * rows * cols);
    for (int lap = 0; lap < 2; lap++) {
      std::vector<int>          vLap = {0, lap == 0 ? 1000 : 0};
      std::vector<cv::KeyPoint> kps;
      cv::Mat                   desc;
      const int32_t             mono = ex(im, cv::Mat(), kps, desc, vLap);
      const std::string         tag  = std::string(lap == 0 ? "lap1000" : "lap0") + "_f" + std::to_string(f);
      cv::Mat                   dc   = desc.isContinuous() ? desc : desc.clone();
      write_npy(out + "/kps_" + tag + ".npy", kpDescr, {kps.size()}, kps.data(), kps.size() * sizeof(cv::KeyPoint));
      write_npy(out + "/desc_" + tag + ".npy", "'|u1'", {(size_t)dc.rows, 32}, dc.data, (size_t)dc.rows * 32);
      write_npy(out + "/mono_" + tag + ".npy", "'<i4'", {}, &mono, 4);
    }
    for (int l = 0; l < nlevels; l++) {
      cv::Mat lvl = ex.mvImagePyramid[l].clone();
      write_npy(out + "/pyr_f" + std::to_string(f) + "_l" + std::to_string(l) + ".npy", "'|u1'", {(size_t)lvl.rows, (size_t)lvl.cols}, lvl.data,
                (size_t)lvl.rows * lvl.cols);
      cv::Mat bl = lvl.clone();
      cv::GaussianBlur(bl, bl, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      write_npy(out + "/blur_f" + std::to_string(f) + "_l" + std::to_string(l) + ".npy", "'|u1'", {(size_t)bl.rows, (size_t)bl.cols}, bl.data,
                (size_t)bl.rows * bl.cols);
    }
  }
  std::ofstream meta(out + "/meta.txt");
  meta << "opencv_version " << CV_VERSION << "\n";
#if defined(__FMA__) || defined(__ARM_FEATURE_FMA)
  meta << "built_with_fma 1\n";  // the tap rotation of computeOrbDescriptor is then contracted (orbx_set_tap_contraction mode 1)
#else
  meta << "built_with_fma 0\n";
#endif
  meta << "params " << nfeatures << " " << scaleFactor << " " << nlevels << " " << iniTh << " " << minTh << "\n";
  meta << "frames " << n << " " << rows << " " << cols << "\n";
  std::cout << "wrote " << out << " (OpenCV " << CV_VERSION << ")\n";
  return 0;
}
