// static_lifetime.cpp -- handles that outlive main().
//
// The reference never deletes its extractors (src/Tracking.cc:576-585: plain `new`, no matching delete), and a host that wraps them in
// objects with static storage duration destroys them from exit-time destructors, in an order relative to the HIP runtime's own teardown
// that the host does not control.  Three holders cover the orders that occur:
//   early   constructed (empty) during static initialisation, filled in main(): its destructor is registered BEFORE the library's exit
//           handler and therefore runs AFTER it -- the destroy calls must then free host state only;
//   late    a function-local static created in main() after the first handle exists: destroyed BEFORE the exit handler, on a live runtime;
//   atexit  a handler registered first thing in main() that destroys a raw handle.
// Every path must end the process with status 0 and no fault.  argv[1] = "leak" additionally leaves one handle of each kind
// undestroyed, as the reference does.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include <orbslam3_amd/ORBVocabulary.h>
#include <orbslam3_amd/ORBextractor.h>
#include <orbslam3_amd/ORBmatcher.h>

struct Holder {
  std::unique_ptr<ORB_SLAM3::ORBextractor>   ext;
  std::unique_ptr<ORB_SLAM3_AMD::ORBmatcher> matcher;
  void*                                      pinned = nullptr;
  void*                                      stream = nullptr;
  ~Holder() {
    matcher.reset();
    ext.reset();
    orbx_host_free(pinned);
    orbx_stream_destroy(stream);
  }
};

static Holder  g_early;           // constructed before main
static orbx_t* g_raw = nullptr;   // destroyed from an atexit handler
static orbm_t* g_rawMatcher = nullptr;

static void at_exit_destroy() {
  orbx_destroy(g_raw);
  orbm_destroy(g_rawMatcher);
}

static int run_once(ORB_SLAM3::ORBextractor& e, int rows, int cols) {
  std::vector<unsigned char> px((size_t)rows * cols);
  unsigned                   s = 12345u;
  for (auto& p : px) s = s * 1664525u + 1013904223u, p = (unsigned char)(s >> 24);
  orbx_cv::Mat                   im(rows, cols, orbx_cv::CV_8U, px.data()), desc;
  std::vector<orbx_cv::KeyPoint> kps;
  std::vector<int>               lap = {0, 0};
  const int                      rc = e(im, orbx_cv::Mat(), kps, desc, lap);
  return rc < 0 ? rc : (int)kps.size();
}

int main(int argc, char** argv) {
  atexit(at_exit_destroy);  // registered before the library arms its own handler: runs after it
  const bool leak = argc > 1 && !strcmp(argv[1], "leak");
  g_early.ext.reset(new ORB_SLAM3::ORBextractor(500, 1.2f, 8, 20, 7));
  g_early.matcher.reset(new ORB_SLAM3_AMD::ORBmatcher(0.7f, true));
  g_early.pinned = orbx_host_alloc(1 << 20);
  g_early.stream = orbx_stream_create(0);
  static Holder late;  // function-local static: destroyed first at exit
  late.ext.reset(new ORB_SLAM3::ORBextractor(500, 1.2f, 8, 20, 7));
  late.pinned = orbx_host_alloc(4096);
  if (orbx_create(&g_raw, 300, 1.2f, 8, 20, 7, 0) != 0 || orbm_create(&g_rawMatcher, 0) != 0) {
    fprintf(stderr, "create: %s\n", orbx_last_error());
    return 3;
  }
  const int n1 = run_once(*g_early.ext, 240, 320), n2 = run_once(*late.ext, 240, 320);
  if (n1 <= 0 || n1 != n2) {
    fprintf(stderr, "extraction: %d %d\n", n1, n2);
    return 4;
  }
  if (leak) {
    new ORB_SLAM3::ORBextractor(500, 1.2f, 8, 20, 7);  // never deleted, as in the reference
    orbx_t* h = nullptr;
    orbx_create(&h, 100, 1.2f, 4, 20, 7, 0);
  }
  printf("ok %d\n", n1);
  return 0;  // destructors and handlers take it from here
}
