// orbx_extractor.hip -- host orchestration + C ABI (include/orbx.h) of the ORB front-end on gfx950.
//
// One handle owns a HIP stream and a device workspace sized for (image size, batch).  Host-side tables
// (scale factors, feature quotas, per-level sizes, FAST cell grids, resize index/weight tables) are
// computed with the reference's exact float/double sequences (src/ORBextractor.cc:403-465, 1140-1150,
// 755-775; OpenCV resize.cpp) so that every kernel is integer-only where the reference is.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbx.h"
#include "orbx_kernels.h"

namespace {

thread_local std::string g_err;

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                        \
      return -3;                                                                                        \
    }                                                                                                   \
  } while (0)

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorF(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline int alignUp(int v, int a) { return (v + a - 1) / a * a; }

const signed char kPatternHost[1024] = {
#include "orb_pattern.inc"
};

// A recorded hipGraph of orbx_extract holds device addresses: a reallocation of one of the handle's buffers retires it (GraphKey::epoch).
// The counter is the HANDLE's (EpochScope names it for the duration of an entry point); allocations made outside any scope -- none today
// -- fall back to a process-wide counter that is part of every key, so a forgotten scope costs replays, never correctness.
static std::atomic<unsigned long long> g_allocEpochFallback{0};
thread_local unsigned long long*       t_allocEpoch = nullptr;
struct EpochScope {
  unsigned long long* prev;
  explicit EpochScope(unsigned long long* e) : prev(t_allocEpoch) { t_allocEpoch = e; }
  ~EpochScope() { t_allocEpoch = prev; }
};

// Process teardown.  A host may destroy its handles from static destructors or interpreter finalizers that run after the HIP runtime has
// begun to shut down, or never destroy them at all (the reference never deletes its extractors, src/Tracking.cc:576-585).  Both ends are
// covered by one exit handler, armed behind the library's first HIP call so that it runs BEFORE the runtime's own exit handlers:
//   1. it releases the device side (streams, graphs, events, memory) of every handle that is still alive -- handles leaked into the
//      runtime's teardown were measured to end the process with SIGSEGV inside __cxa_finalize when a profiler is attached;
//   2. from then on the destroy / free entry points of this library release host state only and make no device call.
static std::atomic<bool> g_runtimeDown{false};
static thread_local bool t_releasingAtExit = false;  // set while the exit handler itself releases the live handles (the runtime is still up)
struct LiveHandles {
  std::mutex                             m;
  std::vector<std::pair<void*, void (*)(void*)>> v;  // object, "release your device side" (must leave the object destroyable on the host side)
};
static LiveHandles& live_handles() {
  static LiveHandles* L = new LiveHandles();  // never destroyed: exit handlers of other libraries may still unregister
  return *L;
}
static void mark_runtime_down() {
  if (g_runtimeDown.exchange(true)) return;
  std::vector<std::pair<void*, void (*)(void*)>> v;
  {
    std::lock_guard<std::mutex> lock(live_handles().m);
    v.swap(live_handles().v);
  }
  t_releasingAtExit = true;
  for (auto it = v.rbegin(); it != v.rend(); ++it) it->second(it->first);
  t_releasingAtExit = false;
}
struct DownSentinel {  // the library's own static destruction marks it too (dlclose, or an exit path that skipped atexit handlers)
  ~DownSentinel() { g_runtimeDown.store(true); }
} g_downSentinel;
static void arm_exit_handler() {
  static std::once_flag armed;
  std::call_once(armed, [] { atexit(mark_runtime_down); });
}

template <class T>
struct DevBuf {
  T*     p = nullptr;
  size_t n = 0;
  int    ensure(size_t count) {
    if (count <= n) return 0;
    if (t_allocEpoch) ++*t_allocEpoch; else g_allocEpochFallback++;
    if (p) hipFree(p);
    p = nullptr;
    n = 0;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
      g_err = std::string("hipMalloc: ") + hipGetErrorString(e);
      return -3;
    }
    n = count;
    return 0;
  }
  void release() {
    if (p && (!g_runtimeDown.load() || t_releasingAtExit)) hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// cv::resize INTER_LINEAR index + Q11 weight tables for one (source, destination) size pair (OpenCV resize.cpp)
void buildResizeTables(int sw, int sh, int dw, int dh, std::vector<int2>& xt, std::vector<int2>& yt) {
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  auto         q11 = [](float v) { return (int)(short)std::min(std::max(cvRoundF(v * 2048), -32768), 32767); };
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int   sx = cvFloorF(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= sw - 1) fx = 0, sx = sw - 1;
    xt.push_back(make_int2(sx, (q11(1.f - fx) & 0xffff) | (q11(fx) << 16)));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int   sy = cvFloorF(fy);
    fy -= sy;
    yt.push_back(make_int2(sy, (q11(1.f - fy) & 0xffff) | (q11(fy) << 16)));
  }
}

}  // namespace

struct orbx_extractor {
  // ---- reference parameters and tables (src/ORBextractor.cc:403-465)
  int                nfeatures, nlevels, iniTh, minTh;
  float              scaleFactor;
  std::vector<float> sf, invSf, sigma2, invSigma2;
  std::vector<int>   quota;
  int                device = 0;
  hipStream_t        stream = nullptr;
  uint8_t*           hOut = nullptr;  // pinned landing buffer of orbx_extract: counts + keypoints + descriptors in one batch of copies
  size_t             hOutCap = 0;
  uint8_t*           hIn = nullptr;   // pinned take-off buffer of orbx_extract: the caller's (pageable) image is copied here first
  size_t             hInCap = 0;
  uint8_t*           hPyr = nullptr;  // pinned host copy of levels >= 1 of orbx_extract's frame (orbx_set_pyramid_download)
  size_t             hPyrCap = 0;
  bool               pyrToHost = false, hPyrValid = false;
  bool               profiling = false;
  int                tapMode = 1;  // orbx_set_tap_contraction
  // ---- streaming entry points (orbx_extract_async / orbx_wait): a ring of in-flight chunks.  Chunk k's upload runs on the
  // copy-in stream, its kernels on the handle's stream, its download on the copy-out stream, so three consecutive chunks
  // overlap: H2D(k + 1) || kernels(k) || D2H(k - 1).  The extractor's own workspace (pyramid, candidates, ...) is shared: kernels of
  // consecutive chunks are ordered on one stream anyway.
  struct AsyncSlot {
    DevBuf<uint8_t>     dIn, dDesc;
    DevBuf<orbx::KpRec> dKps;
    DevBuf<int>         dCounts;
    hipEvent_t          evIn = nullptr, evCompute = nullptr, evDone = nullptr;
    uint8_t *           hIn = nullptr, *hOut = nullptr;  // pinned staging, only used when the caller's memory is pageable
    size_t              hInCap = 0, hOutCap = 0;
    int                 ticket = -1;  // ticket in flight (-1 = free)
    int                 nframes = 0, cap = 0;
    // deferred host copies out of hOut (pageable destinations)
    orbx_keypoint_t*    uKps = nullptr;
    uint8_t*            uDesc = nullptr;
    int*                uCounts = nullptr;
    bool                stagedOut = false, pendingOut = false;
  };
  static constexpr int kMaxAsyncDepth = 8;
  AsyncSlot          slots[kMaxAsyncDepth];
  int                asyncDepth = 3;
  int                nextTicket = 0;
  hipStream_t        sIn = nullptr, sOut = nullptr;
  static constexpr int kProfSlots = 32;  // ring of event sets: one per profiled batch
  hipEvent_t         ev[kProfSlots][ORBX_NUM_STAGES] = {};
  bool               evValid = false;
  int                profCount = 0;  // batches recorded since profiling was enabled
  // ---- geometry / workspace
  orbx::FrameGeom geom{};
  bool            haveGeom = false;
  int             capB     = 0;
  int             lastB    = 0;
  orbx::Bufs      bufs{};
  DevBuf<uint8_t> dPyr, dBlur, dNCh, dStage, dRaw, dDescOut, dOutBlock;
  DevBuf<int>     dStereoSad, dStereoRowStart, dStereoRowIdx, sSad, sKept, sRow, sRidx;
  DevBuf<orbx::KpRec> sKl, sKr;
  DevBuf<uint8_t> sDl, sDr;
  DevBuf<float>   sU, sDepth;
  DevBuf<int2>    dXtab, dYtab;
    DevBuf<long long> dDbgT;
  DevBuf<int>       dFault;  // Bufs::fault
  DevBuf<int>     dCellSlot, dCellCount, dCellOffs, dCandCount, dLvlCount, dCounts;
  DevBuf<int4>    dOutIdx;
  DevBuf<uint16_t> dNodeOf, dD, dDIndex, dPushOff, dMultiOff, dKeptRank, dVA, dVB;
  int              octLdsNodeCap = 0, octLdsPtsCap = 0, octLdsBytes = 0;  // dynamic LDS plan of k_octree (makeGeometry)
  struct GraphKey {
    int            rows, cols;
    size_t         step;
    int            lap0, lap1, need;
    const uint8_t *    src, *dst;
    int                tapMode, inW, inH;
    const uint8_t*     pyrDst;  // pinned pyramid copy (null: not downloaded)
    unsigned long long epoch, epochFallback;
    bool operator==(const GraphKey& o) const {
      return rows == o.rows && cols == o.cols && step == o.step && lap0 == o.lap0 && lap1 == o.lap1 && need == o.need && src == o.src && dst == o.dst &&
             tapMode == o.tapMode && inW == o.inW && inH == o.inH && pyrDst == o.pyrDst && epoch == o.epoch && epochFallback == o.epochFallback;
    }
  };
  unsigned long long allocEpoch = 0;  // bumped by (re)allocations of THIS handle's device buffers (EpochScope)
  bool           graphReplay = true, graphSeen = false;  // orbx_extract as a replayed hipGraph (orbx_set_graph_replay)
  const uint8_t* graphImg0 = nullptr;  // what run() left in bufs.img0 / img0Pitch / img0FrameStride when the graph was recorded: a replay restores it
  int            graphImg0Pitch = 0;
  long long      graphImg0Stride = 0;
  GraphKey       graphKey{};
  hipGraphExec_t graphExec = nullptr;
  int              octLatencyFrames = 8;  // up to this many frames per call the quadtree runs a workgroup, not a wave, per (frame, level)
  DevBuf<uint32_t>           dSlots, dCand, dLvlKp;
  DevBuf<orbx::OtNode>       dNodesA, dNodesB;
  DevBuf<orbx::Int4>         dChildCnt;
  DevBuf<orbx::Idx4>         dChildIdx;
  DevBuf<orbx::PatternTab>   dPattern;
  DevBuf<orbx::KpRec>        dKpOut;
  int                        stagePitch = 0;
  // fused pyramid launches (k_pyramid_bands): one plan for batches, one for calls with a handful of frames
  struct PyrPlanHost {
    orbx::PyrPlan          plan{};
    DevBuf<orbx::PyrBand>  dBands;
    int                    ldsBytes = 0;
    bool                   valid = false;
  };
  bool                       pyrLdsRaised[2] = {false, false};
  bool                       pyrBandsOnly = false;  // a level's source footprint does not fit k_resize_level's LDS stage (scale factors above ~1.3): every call uses k_pyramid_bands
  PyrPlanHost                pyrLatency[2];  // calls with a handful of frames: levels [1, mid) and [mid, nlevels) as two launches
  std::vector<int2>          hYtab;  // host copy of the row tables (band ranges are derived from them)
  int buildPyrPlan(int firstLevel, int endLevel, int bandRows, int ldsLimit, PyrPlanHost& out);
  DevBuf<orbx::FastTile>     dTiles[2];      // k_fast_tiles: tile table + XCD-aware launch order; [0] batches, [1] a handful of frames
  DevBuf<int>                dTileOrder[2];
  int                        tileGrid[2] = {0, 0};
  orbx::FastLds2             fastLds2[2] = {};

  // optional cv::resize of the input before extraction (System::TrackMonocular etc., src/System.cc:472-478)
  int          preW = 0, preH = 0, preSrcW = 0, preSrcH = 0;
  DevBuf<int2> dPreX, dPreY;
  DevBuf<uint8_t> dPre;
  bool resizing(int rows, int cols) const { return preW > 0 && (rows != preH || cols != preW); }
  void effective(int rows, int cols, int* er, int* ec) const {
    *er = resizing(rows, cols) ? preH : rows;
    *ec = resizing(rows, cols) ? preW : cols;
  }
  int runResized(const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frameStride, int lap0, int lap1, orbx::KpRec* d_kps,
                 uint8_t* d_desc, int* d_counts, int cap, hipStream_t s);
  int makeGeometry(int rows, int cols);
  int ensureWorkspace(int rows, int cols, int B);
  int run(const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frameStride, int lap0, int lap1, orbx::KpRec* d_kps,
          uint8_t* d_desc, int* d_counts, int cap, hipStream_t s);
};

// Row ranges of k_pyramid_bands for levels [firstLevel, endLevel) with bands of `bandRows` rows of level endLevel - 1; out.valid = false
// when the LDS budget does not allow it.
int orbx_extractor::buildPyrPlan(int firstLevel, int endLevel, int bandRows, int ldsLimit, PyrPlanHost& out) {
  using namespace orbx;
  out.valid = false;
  const FrameGeom& g = geom;
  const int        last = endLevel - 1;
  if (firstLevel < 1 || firstLevel > last || endLevel > g.nlevels || bandRows < 1) return 0;
  const int            nb = (g.lv[last].h + bandRows - 1) / bandRows;
  std::vector<PyrBand> bands(nb);
  std::vector<int>     cLo(nb), cHi(nb), nLo(nb), nHi(nb);  // computed rows of the level above / needed rows of this one (inclusive)
  std::vector<int>     yRows(nb, 0);
  for (int k = 0; k < nb; k++) {
    cLo[k] = k * bandRows, cHi[k] = std::min((k + 1) * bandRows, g.lv[last].h) - 1;
    PyrBand& B = bands[k];
    memset(&B, 0, sizeof(B));
    B.lo[last] = B.ownLo[last] = (short)cLo[k], B.n[last] = B.ownN[last] = (short)(cHi[k] - cLo[k] + 1);
    yRows[k] = B.n[last];
  }
  int half = 0;
  for (int l = last - 1; l >= firstLevel - 1; l--) {
    const LevelGeom& L  = g.lv[l];
    const int2*      yt = hYtab.data() + g.lv[l + 1].ytabOff;
    for (int k = 0; k < nb; k++) {
      nLo[k] = std::min(std::max(yt[cLo[k]].x, 0), L.h - 1);
      nHi[k] = std::min(std::max(yt[cHi[k]].x + 1, 0), L.h - 1);
    }
    for (int k = 0; k < nb; k++) {
      PyrBand& B = bands[k];
      int      lo = nLo[k], hi = nHi[k];
      if (l >= firstLevel) {  // rows of this level nobody above needs still have to be produced: the bands partition [0, h)
        const int oLo = k == 0 ? 0 : nLo[k], oHi = k == nb - 1 ? L.h - 1 : nLo[k + 1] - 1;
        B.ownLo[l] = (short)oLo, B.ownN[l] = (short)std::max(oHi - oLo + 1, 0);
        if (oHi >= oLo) lo = std::min(lo, oLo), hi = std::max(hi, oHi);
        yRows[k] += hi - lo + 1;
      }
      B.lo[l] = (short)lo, B.n[l] = (short)(hi - lo + 1);
      cLo[k] = lo, cHi[k] = hi;
      half   = std::max(half, (hi - lo + 2) * ((L.w + 3) & ~3));  // one spare row: the row below the last one is addressed with weight 0
    }
  }
  half = alignUp(half + 16, 16);
  auto magic = [](unsigned d) { return d <= 1 ? 0u : 0xffffffffu / d + 1u; };
  PyrPlan& P = out.plan;
  memset(&P, 0, sizeof(P));
  P.firstLevel = firstLevel, P.endLevel = endLevel, P.nbands = nb, P.ldsHalf = half;
  int xtEntries = 0;
  for (int l = 0; l < g.nlevels; l++) {
    P.quadMagic[l] = magic((unsigned)((g.lv[l].w + 3) >> 2));
    if (l >= firstLevel && l < endLevel) P.xtOff[l] = xtEntries, xtEntries += ((g.lv[l].w + 3) >> 2) * 4;
  }
  P.xtLds = 2 * half;
  P.ytLds = P.xtLds + xtEntries * 8;
  out.ldsBytes = P.ytLds + *std::max_element(yRows.begin(), yRows.end()) * 8;
  if (out.ldsBytes > ldsLimit) return 0;
  if (out.dBands.ensure(bands.size())) return -3;
  HIP_TRY(hipMemcpy(out.dBands.p, bands.data(), bands.size() * sizeof(PyrBand), hipMemcpyHostToDevice));
  out.valid = true;
  return 0;
}

// Level sizes, FAST grids, quadtree roots, buffer offsets and resize tables for one image size.
int orbx_extractor::makeGeometry(int rows, int cols) {
  using namespace orbx;
  if (haveGeom && geom.rows == rows && geom.cols == cols) return 0;
  if (rows <= 0 || cols <= 0) return -1;
  if (cols > 4000 || rows > 4000) {
    g_err = "image larger than 4000 px is not supported (12-bit candidate coordinates)";
    return -2;
  }
  FrameGeom g{};
  g.nlevels = nlevels;
  g.rows = rows, g.cols = cols;
  std::vector<int2> xt, yt;
  std::vector<int>  cellSlot;
  int               pyrOff = 0, blurOff = 0, cellBase = 0, slotBase = 0, nodeBase = 0, kpBase = 0, tileBase = 0;
  for (int l = 0; l < nlevels; l++) {
    LevelGeom& L = g.lv[l];
    L.w = cvRoundF((float)cols * invSf[l]);  // src/ORBextractor.cc:1142-1143
    L.h = cvRoundF((float)rows * invSf[l]);
    if (L.w < 1 || L.h < 1 || (l == 0 && (L.w < 8 || L.h < 8))) {  // cv::resize asserts on an empty destination; levels of 1 .. 7 px are produced (k_resize_tiny)
      g_err = "image too small for the pyramid depth: a level would be empty (or the image itself is smaller than 8 x 8)";
      return -2;
    }
    L.pitch = alignUp(L.w, 64);
    L.pyrOff = pyrOff;
    if (l > 0) pyrOff += L.pitch * alignUp(L.h, 4);
    L.blurPitch = alignUp(L.w, 64);
    L.blurOff   = blurOff;
    blurOff += L.blurPitch * alignUp(L.h, 4);
    // FAST grid (src/ORBextractor.cc:761-775)
    L.maxBX            = L.w - kEdgeThreshold + 3;
    L.maxBY            = L.h - kEdgeThreshold + 3;
    const float width  = (float)(L.maxBX - kMinBorder);
    const float height = (float)(L.maxBY - kMinBorder);
    const float W      = 35;
    L.nCols            = (int)(width / W);
    L.nRows            = (int)(height / W);
    // A level with less than one 35-px cell of usable area: the reference's cell loops do not execute (its wCell = ceil(width / 0) is
    // never used), so the level contributes no keypoint.  It is still part of the pyramid chain and is blurred like any other.
    const bool noCells = L.nCols <= 0 || L.nRows <= 0;
    if (noCells) L.nCols = L.nRows = 0;
    L.wCell            = noCells ? 0 : (int)std::ceil(width / L.nCols);
    L.hCell            = noCells ? 0 : (int)std::ceil(height / L.nRows);
    if (L.wCell + 6 > kFastTileMax || L.hCell + 6 > kFastTileMax) {
      g_err = "FAST cell larger than the kernel tile";
      return -2;
    }
    L.cellBase = cellBase;
    L.slotBase = slotBase;
    int slots = 0;
    for (int i = 0; i < L.nRows; i++)
      for (int j = 0; j < L.nCols; j++) {
        const int iniY = kMinBorder + i * L.hCell, iniX = kMinBorder + j * L.wCell;
        int       capc = 0;
        if (!(iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6)) {
          const int maxY = std::min(iniY + L.hCell + 6, L.maxBY), maxX = std::min(iniX + L.wCell + 6, L.maxBX);
          const int iw = maxX - iniX - 6, ih = maxY - iniY - 6;
          // strict 3x3 NMS: no two keypoints are 8-neighbours -> at most one per 2x2 block
          if (iw > 0 && ih > 0) capc = ((iw + 1) / 2) * ((ih + 1) / 2);
        }
        cellSlot.push_back(slots);
        slots += capc;
      }
    const int nc = L.nCols * L.nRows;
    L.slotCount  = alignUp(std::max(slots, nc) + 16, 16);
    cellBase += nc;
    slotBase += L.slotCount;
    // quadtree (src/ORBextractor.cc:566-570)
    L.N    = quota[l];
    L.nIni = noCells ? 1 : (int)std::round(width / height);
    if (L.nIni <= 0) {
      g_err = "portrait image with height > 2 x width: DistributeOctTree would divide by zero (nIni == 0)";
      return -2;
    }
    L.hX       = noCells ? 1.0f : width / L.nIni;
    L.nodeCap  = alignUp(std::max(L.N + 3, 4 * L.nIni) + 16, 16);
    if (L.nodeCap > kOtMaxNodes) {
      g_err = "more than 16383 quadtree nodes on one level is not supported";
      return -2;
    }
    L.nodeBase = nodeBase;
    nodeBase += L.nodeCap;
    L.kpCap  = std::max(L.N + 3, 4 * L.nIni);
    L.kpBase = kpBase;
    kpBase += L.kpCap;
    L.scale     = sf[l];
    L.patchSize = (int)(kPatchSize * sf[l]);  // src/ORBextractor.cc:865
    // blur tiles
    L.blurTilesX   = (L.w + 3) / 4;                          // 4-column groups
    L.blurTilesY   = level_is_tiny(L) ? 0 : (L.h + kBlurRows - 1) / kBlurRows;  // row strips (a tiny level is not blurred: no keypoint can lie on it)
    {
      // interior groups (12-byte window inside the row) go to k_blur_levels, the 2-3 edge groups of every strip to
      // k_blur_edges; an unaligned level 0 sends ALL its groups to the edge kernel, so that grid is sized for it
      const int nInt = std::max((L.w - 8) >> 2, 0);
      L.blurTileBase = tileBase;  // in workgroups of 256 items
      tileBase += (nInt * L.blurTilesY + 255) / 256;
    }
    // cv::resize INTER_LINEAR tables (OpenCV resize.cpp)
    L.xtabOff = (int)xt.size();
    L.ytabOff = (int)yt.size();
    if (l > 0) buildResizeTables(g.lv[l - 1].w, g.lv[l - 1].h, L.w, L.h, xt, yt);
  }
  g.totalCells     = cellBase;
  g.totalSlots     = slotBase;
  g.totalNodes     = nodeBase;
  g.totalKp        = kpBase;
  if (kpBase > orbx::kLayoutMaxRounds * 256) {
    g_err = "more than 32768 keypoints per frame";
    return -2;
  }
  g.totalBlurTiles = tileBase;
  g.pyrBytes       = alignUp(pyrOff, 256);
  g.blurBytes      = alignUp(blurOff, 256);
  // ---- k_fast_tiles: tiles = runs of horizontally adjacent valid cells of one cell row (orbx_fast_tiles.h).  Two tilings of the same cells:
  // [0] for batches (8 KB of pixels per tile: the per-workgroup costs are paid once per ~4 cells), [1] for calls with a handful of frames
  // (4 KB: twice as many, shorter workgroups -- a single frame's 150 large tiles leave 100 CUs idle and the launch lasts as long as one
  // tile's whole chain of phases)
  for (int set = 0; set < 2; set++) {
    auto magic = [](unsigned d) { return d <= 1 ? 0u : 0xffffffffu / d + 1u; };
    int                    kTileBudget = set == 0 ? 8192 : 4096;  // LDS bytes of one pixel tile (latency form measured at 2.5 / 3 / 4 / 6 / 8 KB: 10.6 / 10.8 / 10.4 / 12.3 / 27.7 us per VGA frame)
#ifdef ORBX_TUNING
    if (getenv("ORBX_FAST_TILE_BYTES") && set == 0) kTileBudget = atoi(getenv("ORBX_FAST_TILE_BYTES"));
    if (getenv("ORBX_FAST_TILE_BYTES_LATENCY") && set == 1) kTileBudget = atoi(getenv("ORBX_FAST_TILE_BYTES_LATENCY"));
#endif
    std::vector<FastTile>  tiles;
    struct RowRef {
      int first, count, px;
    };
    std::vector<RowRef> rowsOfTiles;
    int maxTileBytes = 0, maxCells = 1, maxW = 1;
    for (int l = 0; l < nlevels; l++) {
      const LevelGeom& L = g.lv[l];
      int              lastTile = -1;  // tile that zeroes the counts of the skipped cells after it
      for (int i = 0; i < L.nRows; i++) {
        const int iniY = kMinBorder + i * L.hCell;
        int       nv   = 0;  // valid cells of this row are a prefix 0 .. nv - 1
        int       ih   = 0;
        if (iniY < L.maxBY - 3) {
          ih = std::min(iniY + L.hCell + 6, L.maxBY) - iniY - 6;
          if (ih > 0)
            for (int j = 0; j < L.nCols; j++) {
              const int iniX = kMinBorder + j * L.wCell;
              if (iniX >= L.maxBX - 6 || std::min(iniX + L.wCell + 6, L.maxBX) - iniX - 6 <= 0) break;
              nv++;
            }
        }
        const int rowCell0 = L.cellBase + i * L.nCols;
        if (nv == 0) {  // the whole row is skipped by the reference: its counts are zeroed by the previous tile
          if (lastTile < 0) {
            g_err = "FAST grid: first cell row of a level is empty";
            return -2;
          }
          tiles[lastTile].nzero += L.nCols;
          continue;
        }
        const int th = ih + 6;
        // cells per tile: LDS pitch x rows within the budget, pitch <= 256 bytes (64 staging lanes), <= 16 pre-test passes
        int G = 1;
        for (int c = 2; c <= nv; c++) {
          const int pitch = 4 * ((3 + c * L.wCell + 6 + 3) / 4 + 2);
          const int ng    = (c * L.wCell + 3 + 3) / 4 + 1;
          if (pitch > 256 || pitch * th > kTileBudget || (ng * ih + kFastTileThreads - 1) / kFastTileThreads > kFastMaxPass) break;
          G = c;
        }
        const int nt = (nv + G - 1) / G;
        RowRef    rr{(int)tiles.size(), nt, 0};
        for (int t = 0, j0 = 0; t < nt; t++) {
          const int nc = nv / nt + (t < nv % nt ? 1 : 0);
          FastTile  T{};
          const int iniX0 = kMinBorder + j0 * L.wCell;             // first cell's sub-image origin
          const int X0 = iniX0 + 3, xa = (X0 - 3) & ~3, ox = (X0 - 3) - xa;
          const int jl = j0 + nc - 1, iniXl = kMinBorder + jl * L.wCell;
          const int iwLast = std::min(iniXl + L.wCell + 6, L.maxBX) - iniXl - 6;
          T.level = l, T.srcPitch = L.pitch;
          T.srcOff = l == 0 ? (xa | (iniY << 16)) : (L.pyrOff + iniY * L.pitch + xa);
          T.IW = (nc - 1) * L.wCell + iwLast, T.ih = ih, T.th = th, T.ox3 = ox + 3;
          T.tw4 = (ox + T.IW + 6 + 3) / 4;
          T.ldsPitch = 4 * (T.tw4 + 1);
          if (((T.ldsPitch / 4) & 1) == 0) T.ldsPitch += 4;  // odd dword pitch: consecutive rows start on different banks
          T.wCell = L.wCell, T.iwLast = iwLast, T.ncells = nc;
          T.cell0 = rowCell0 + j0, T.slotBase = L.slotBase;
          T.xoff = 3 + j0 * L.wCell, T.yoff = 3 + i * L.hCell;
          T.g0 = T.ox3 >> 2;
          T.NG = ((T.ox3 + T.IW - 1) >> 2) - T.g0 + 1;
          T.nitems = T.NG * ih;
          T.W = (L.wCell * ih + 31) / 32;
          T.mNG = magic(T.NG), T.mWc = magic(L.wCell), T.mIw = magic(L.wCell), T.mIwLast = magic(iwLast);
          T.nzero = (t == nt - 1) ? L.nCols - nv : 0;
          if ((T.nitems + kFastTileThreads - 1) / kFastTileThreads > kFastMaxPass || T.tw4 > 64 || th > 127 || T.ldsPitch > 511) {
            g_err = "FAST tile exceeds the kernel's limits (cell larger than expected)";
            return -2;
          }
          maxTileBytes = std::max(maxTileBytes, T.ldsPitch * th), maxCells = std::max(maxCells, nc), maxW = std::max(maxW, T.W);
          rr.px += T.IW * ih;
          lastTile = (int)tiles.size();
          tiles.push_back(T);
          j0 += nc;
        }
        rowsOfTiles.push_back(rr);
      }
    }
    // launch order: whole cell rows onto the least loaded XCD (by pixels), largest first; block b runs on XCD b % 8
    std::stable_sort(rowsOfTiles.begin(), rowsOfTiles.end(), [](const RowRef& a, const RowRef& b) { return a.px > b.px; });
    std::vector<std::vector<int>> perXcd(8);
    long long                     load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const RowRef& r : rowsOfTiles) {
      int best = 0;
      for (int x = 1; x < 8; x++)
        if (load[x] < load[best]) best = x;
      load[best] += r.px;
      for (int t = 0; t < r.count; t++) perXcd[best].push_back(r.first + t);
    }
    size_t per = 0;
    for (auto& v : perXcd) per = std::max(per, v.size());
    std::vector<int> order(per * 8, -1);
    for (int x = 0; x < 8; x++)
      for (size_t k = 0; k < perXcd[x].size(); k++) order[k * 8 + x] = perXcd[x][k];
    tileGrid[set] = (int)order.size();
    if (dTiles[set].ensure(tiles.size() + 1) || dTileOrder[set].ensure(order.size() + 1)) return -3;
    HIP_TRY(hipMemcpy(dTiles[set].p, tiles.data(), tiles.size() * sizeof(FastTile), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dTileOrder[set].p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    // LDS plan
    const int region = alignUp(maxTileBytes + 32, 16);
    FastLds2  fl{};
    // {tile, 7-bit copy} + candidate list + corner list + masks + counters; the score map reuses the copy after the pre-test (orbx_fast_tiles.h)
    fl.t7Off    = alignUp(16 + region, 16) + 16;  // the copy is addressed like the tile: reads reach 4 bytes before its origin
    fl.scoreOff = fl.t7Off, fl.scoreBytes = region;
    fl.candOff  = fl.t7Off + region;
    // candidate list entries; lanes whose survivors do not fit score them themselves (ORBX_FAST_CAND_CAP: tests shrink the list to reach that path)
    fl.candCap  = getenv("ORBX_FAST_CAND_CAP") ? std::max(atoi(getenv("ORBX_FAST_CAND_CAP")), 2) : 2048;
    fl.listCOff = alignUp(fl.candOff + 2 * fl.candCap, 16);
    fl.mask1Off = alignUp(fl.listCOff + 2 * fl.candCap, 16);
    fl.mask2Off = fl.mask1Off + 4 * maxCells * maxW;
    fl.cntOff   = fl.mask2Off + 4 * maxCells * maxW;
    fl.total    = alignUp(fl.cntOff + 4 * (maxCells + 3), 16);
    if (fl.total > 160 * 1024) {
      g_err = "FAST tile LDS plan exceeds 160 KB";
      return -2;
    }
    fastLds2[set] = fl;
  }
  // upload tables
  if (dXtab.ensure(xt.size() + 1) || dYtab.ensure(yt.size() + 1) || dCellSlot.ensure(cellSlot.size() + 1)) return -3;
  HIP_TRY(hipMemcpy(dXtab.p, xt.data(), xt.size() * sizeof(int2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dYtab.p, yt.data(), yt.size() * sizeof(int2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dCellSlot.p, cellSlot.data(), cellSlot.size() * sizeof(int), hipMemcpyHostToDevice));
  // k_octree LDS plan: all node-level state of a level in LDS (kOctNodeBytes per node) plus as many 16-bit point labels
  // as fit.  Up to ~20 KB per wave keeps 8 waves (= 8 problems) resident per CU; larger quotas trade residency for
  // LDS, and beyond 64 KB the state stays in global memory.
  {
    int maxCap = 0;
    for (int l = 0; l < nlevels; l++) maxCap = std::max(maxCap, g.lv[l].nodeCap);
    const int fixed = kOctFixedInts * 4, nodeBytes = maxCap * kOctNodeBytes;
    if (fixed + nodeBytes + 2 * 512 <= 20224) {
      octLdsNodeCap = maxCap, octLdsBytes = 20224, octLdsPtsCap = (octLdsBytes - fixed - nodeBytes) / 2;
    } else if (fixed + nodeBytes + 2 * 4096 <= 40960) {
      // larger quotas (2000 features: 30 KB of node state): 4 problems per CU, and 4096 point labels still ride along
      octLdsNodeCap = maxCap, octLdsPtsCap = 4096, octLdsBytes = fixed + nodeBytes + 2 * octLdsPtsCap;
    } else if (fixed + nodeBytes <= 65536) {
      octLdsNodeCap = maxCap, octLdsBytes = fixed + nodeBytes, octLdsPtsCap = 0;
    } else {
      octLdsNodeCap = 0, octLdsBytes = fixed, octLdsPtsCap = 0;
    }
#ifdef ORBX_TUNING
    if (getenv("ORBX_OCT_NO_LDS")) octLdsNodeCap = 0, octLdsBytes = fixed, octLdsPtsCap = 0;  // experiment: the quadtree state in global memory
#endif
  }
  geom     = g;
  haveGeom = true;
  capB     = 0;  // workspace must be re-laid out
  // k_resize_level stages the source footprint of a 128 x 16 destination tile in a fixed LDS block (176 x 24 source pixels): fine up to a
  // scale factor of ~1.3 (every configuration the reference ships uses 1.2); above that the band kernel, which sizes its LDS from the
  // tables, produces every level for batches as well
  pyrBandsOnly = false;
  for (int l = 1; l < g.nlevels; l++) {
    if (level_is_tiny(g.lv[l])) continue;  // k_resize_tiny
    const double sx = (double)g.lv[l - 1].w / g.lv[l].w, sy = (double)g.lv[l - 1].h / g.lv[l].h;
    if (sx * kRsTileW + 6 > kRsLdsW || sy * (kRsTileH - 1) + 3 > kRsLdsH - 1) pyrBandsOnly = true;
  }
  // fused pyramid launches of the latency form: the chain in two halves (shorter chains recompute fewer shared rows)
  hYtab = yt;
  pyrLatency[0].valid = pyrLatency[1].valid = false;
  if (g.nlevels > 1) {
    int chainEnd = g.nlevels;  // the band kernel takes the levels before the first tiny one
    for (int l = g.nlevels - 1; l >= 1; l--)
      if (level_is_tiny(g.lv[l])) chainEnd = l;
    const int mid = std::min((chainEnd + 1) / 2, chainEnd - 1);
    const int ends[3] = {1, chainEnd > 2 ? std::max(mid, 1) : chainEnd, chainEnd};
    int       rc2 = 0;
    for (int part = 0; part < 2; part++) {
      if (ends[part] >= ends[part + 1]) continue;
      for (int rowsPerBand : {8, 4, 2, 1})
        if (!pyrLatency[part].valid && (rc2 = buildPyrPlan(ends[part], ends[part + 1], rowsPerBand, 150 * 1024, pyrLatency[part]))) return rc2;
    }
  }
  return 0;
}

int orbx_extractor::ensureWorkspace(int rows, int cols, int B) {
  int rc = makeGeometry(rows, cols);
  if (rc) return rc;
  if (B <= capB) return 0;
  const orbx::FrameGeom& g  = geom;
  const size_t           nb = (size_t)B;
  const size_t           np = nb * g.totalSlots, nn = nb * g.totalNodes;
  if (dPyr.ensure(nb * g.pyrBytes + 256) || dBlur.ensure(nb * g.blurBytes + 256) || dSlots.ensure(np) || dCellCount.ensure(nb * g.totalCells) ||
      dCand.ensure(np) || dNodeOf.ensure(np) || dCellOffs.ensure(nb * g.totalCells) ||
      dNodesA.ensure(nn) || dNodesB.ensure(nn) || dChildCnt.ensure(nn) || dChildIdx.ensure(nn) || dD.ensure(nn) ||
      dDIndex.ensure(nn) || dNCh.ensure(nn) || dPushOff.ensure(nn) || dMultiOff.ensure(nn) || dKeptRank.ensure(nn) || dVA.ensure(nn) ||
      dVB.ensure(nn) || dCandCount.ensure(nb * g.nlevels) || dLvlKp.ensure(nb * g.totalKp) ||
      dLvlCount.ensure(nb * g.nlevels) || dOutIdx.ensure(2 * nb * g.totalKp))
    return -3;
  orbx::Bufs& b = bufs;
  b.pyr = dPyr.p, b.blur = dBlur.p, b.xtab = dXtab.p, b.ytab = dYtab.p, b.cellSlot = dCellSlot.p, b.slots = dSlots.p;
  b.cellCount = dCellCount.p, b.cand = dCand.p, b.nodeOf = dNodeOf.p, b.cellOffs = dCellOffs.p;
  b.nodesA = dNodesA.p, b.nodesB = dNodesB.p, b.childCnt = dChildCnt.p, b.childIdx = dChildIdx.p, b.D = dD.p;
  b.dIndex = dDIndex.p, b.nCh = dNCh.p, b.pushOff = dPushOff.p, b.multiOff = dMultiOff.p, b.keptRank = dKeptRank.p, b.vA = dVA.p;
  b.vB = dVB.p, b.candCount = dCandCount.p, b.lvlKp = dLvlKp.p, b.lvlCount = dLvlCount.p, b.outIdx = dOutIdx.p;
  capB = B;
  return 0;
}

// Extraction of frames that System::Track* would first cv::resize to (preW, preH): the resize runs on the device with the
// pyramid's own kernel body, then the normal pipeline takes the resized frames as its level 0.
int orbx_extractor::runResized(const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frameStride, int lap0, int lap1,
                               orbx::KpRec* d_kps, uint8_t* d_desc, int* d_counts, int cap, hipStream_t s) {
  using namespace orbx;
  if (!resizing(rows, cols)) {
    // Frames whose rows are not 4-byte aligned (odd widths with a tight step: KITTI's 1241) would put level 0 on the byte-load
    // paths of FAST / blur / resize.  One device-side pitched copy into an aligned buffer is cheaper (60 MB per 128 frames).
    const bool al4 = (step % 4 == 0) && (frameStride % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_imgs) & 3) == 0);
    // ORBX_NO_REPITCH=1 keeps the direct route (byte-load kernel variants) reachable for the parity tests
    if (al4 || getenv("ORBX_NO_REPITCH")) return run(d_imgs, B, rows, cols, step, frameStride, lap0, lap1, d_kps, d_desc, d_counts, cap, s);
    const int    pitch = alignUp(cols, 64);
    const size_t fs    = (size_t)pitch * rows;
    if (dPre.ensure(fs * B + 256)) return -3;
    // an own kernel, not hipMemcpy2DAsync: the runtime's rectangle copy on several CU-masked streams at once was the one operation the
    // stereo workload (the only one with unaligned rows) had that the others had not, when it met intermittent GPU memory faults with
    // three chunks in flight
    hipLaunchKernelGGL(k_repitch, dim3((cols + 1023) / 1024, rows, B), dim3(256), 0, s, d_imgs, (long long)step, (long long)frameStride, dPre.p, pitch, (long long)fs, cols, rows);
    return run(dPre.p, B, rows, cols, pitch, fs, lap0, lap1, d_kps, d_desc, d_counts, cap, s);
  }
  // footprint of a 128 x 16 destination tile must fit the LDS stage of resize_tile (176 x 24 source pixels)
  if ((double)cols / preW * kRsTileW + 6 > kRsLdsW || (double)rows / preH * (kRsTileH - 1) + 3 > kRsLdsH - 1) {
    g_err = "input resize: shrink factor above 1.32 (x) / 1.4 (y) is not supported";
    return -2;
  }
  if (preSrcW != cols || preSrcH != rows) {
    std::vector<int2> xt, yt;
    buildResizeTables(cols, rows, preW, preH, xt, yt);
    if (dPreX.ensure(xt.size() + 1) || dPreY.ensure(yt.size() + 1)) return -3;
    HIP_TRY(hipMemcpy(dPreX.p, xt.data(), xt.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dPreY.p, yt.data(), yt.size() * sizeof(int2), hipMemcpyHostToDevice));
    preSrcW = cols, preSrcH = rows;
  }
  const int    pitch = alignUp(preW, 64);
  const size_t fs    = (size_t)pitch * preH;
  if (dPre.ensure(fs * B + 256)) return -3;
  dim3 grid((preW + kRsTileW - 1) / kRsTileW, (preH + kRsTileH - 1) / kRsTileH, B);
  hipLaunchKernelGGL(k_resize_input, grid, dim3(256), 0, s, d_imgs, (int)step, (long long)frameStride, cols, rows, dPre.p, pitch, (long long)fs, preW,
                     preH, dPreX.p, dPreY.p);
  return run(dPre.p, B, preH, preW, pitch, fs, lap0, lap1, d_kps, d_desc, d_counts, cap, s);
}

int orbx_extractor::run(const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frameStride, int lap0, int lap1,
                        orbx::KpRec* d_kps, uint8_t* d_desc, int* d_counts, int cap, hipStream_t s) {
  using namespace orbx;
  int rc = ensureWorkspace(rows, cols, B);
  if (rc) return rc;
  const FrameGeom& g = geom;
  Bufs             b = bufs;
  b.dbgT  = nullptr;
  b.fault = dFault.p;
#ifdef ORBX_TUNING
  if (getenv("ORBX_DBG_OCT")) {  // quadtree phase stamps (orbx_debug_octree_times)
    if (dDbgT.ensure((size_t)B * g.nlevels * 8)) return -3;
    HIP_TRY(hipMemsetAsync(dDbgT.p, 0, (size_t)B * g.nlevels * 8 * sizeof(long long), s));
    b.dbgT = dDbgT.p;
  }
#endif
  b.img0            = d_imgs;
  b.img0FrameStride = (long long)frameStride;
  b.img0Pitch       = (int)step;
  bufs              = b;  // remembered for the pyramid / debug accessors
  lastB             = B;
  hPyrValid         = false;  // the device pyramid is about to change; orbx_extract re-validates its host copy
  const bool prof   = profiling;
  if (prof && !evValid) {
    for (auto& slot : ev)
      for (auto& e : slot) HIP_TRY(hipEventCreate(&e));
    evValid = true;
  }
  hipEvent_t* ev = this->ev[profCount % kProfSlots];
  // level 0 is the caller's buffer: aligned-dword paths need its base, pitch and frame stride 4-byte aligned
  const bool al4 = (step % 4 == 0) && (frameStride % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_imgs) & 3) == 0);
  const int  l0a = al4 ? 1 : 0;
  // blur: interior groups -> k_blur_levels, edge columns -> k_blur_edges, an unaligned level 0 -> k_blur_level0_bytes
  auto launchBlur = [&](hipStream_t st) {
    const FrameGeom& gg = geom;
    int              edgeBlocks = 0;
    for (int l = al4 ? 0 : 1; l < gg.nlevels; l++) edgeBlocks += blur_edge_blocks(gg.lv[l]);
    if (edgeBlocks == 0) {
      hipLaunchKernelGGL(k_blur_levels, dim3(8 * ((gg.totalBlurTiles + 7) / 8), B), dim3(256), 0, st, bufs, gg, l0a);
    } else {
      const int edgePad = 8 * ((edgeBlocks + 7) / 8);
      hipLaunchKernelGGL(k_blur_all, dim3(edgePad + 8 * ((gg.totalBlurTiles + 7) / 8), B), dim3(256), 0, st, bufs, gg, l0a, al4 ? 0 : 1, edgeBlocks, edgePad);
    }
    if (!al4) {
      const int items = ((gg.lv[0].w + 3) / 4) * ((gg.lv[0].h + kBlurEdgeRows - 1) / kBlurEdgeRows);
      hipLaunchKernelGGL(k_blur_level0_bytes, dim3((items + 255) / 256, B), dim3(256), 0, st, bufs, gg);
    }
  };
  if (prof) HIP_TRY(hipEventRecord(ev[0], s));
  {
    // batches: one launch per level.  A handful of frames: the chain in two launches of k_pyramid_bands (levels they do not cover -- a
    // plan that did not fit the LDS -- fall back to their own launches).
    int covered[2][2] = {{0, 0}, {0, 0}};
    if (B <= octLatencyFrames || pyrBandsOnly)
      for (int part = 0; part < 2; part++)
        if (pyrLatency[part].valid) covered[part][0] = pyrLatency[part].plan.firstLevel, covered[part][1] = pyrLatency[part].plan.endLevel;
    for (int l = 1; l < g.nlevels;) {
      int part = -1;
      for (int q = 0; q < 2; q++)
        if (covered[q][0] == l && covered[q][1] > l) part = q;
      if (level_is_tiny(g.lv[l])) {
        hipLaunchKernelGGL(k_resize_tiny, dim3(B), dim3(256), 0, s, b, g, l);
        l++;
        continue;
      }
      if (part < 0) {
        if (pyrBandsOnly) {
          g_err = "scale factor above ~1.3 on an image this wide: the pyramid band kernel's LDS plan does not fit";
          return -2;
        }
        const int tx = (g.lv[l].w + kRsTileW - 1) / kRsTileW, ty = (g.lv[l].h + kRsTileH - 1) / kRsTileH;
        hipLaunchKernelGGL(k_resize_level, dim3(8 * tx, ty, (B + 7) / 8), dim3(256), 0, s, b, g, l, B);
        l++;
        continue;
      }
      const PyrPlanHost& pp = pyrLatency[part];
      bool* const        raised = pyrLdsRaised;  // more than 64 KB of dynamic LDS has to be asked for, once per kernel (and device: per handle)
      const void*        fn = al4 ? (const void*)k_pyramid_bands<true> : (const void*)k_pyramid_bands<false>;
      if (pp.ldsBytes > 64 * 1024 && !raised[al4 ? 1 : 0]) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised[al4 ? 1 : 0] = true;
      }
      if (al4)
        hipLaunchKernelGGL(k_pyramid_bands<true>, dim3(pp.plan.nbands, B), dim3(kPyrBandThreads), pp.ldsBytes, s, b, g, pp.dBands.p, pp.plan, B);
      else
        hipLaunchKernelGGL(k_pyramid_bands<false>, dim3(pp.plan.nbands, B), dim3(kPyrBandThreads), pp.ldsBytes, s, b, g, pp.dBands.p, pp.plan, B);
      l = pp.plan.endLevel;
    }
  }
  if (prof) HIP_TRY(hipEventRecord(ev[1], s));
  {
    const int t1 = std::min(std::max(iniTh, 0), 255), t2 = std::min(std::max(minTh, 0), 255);
    int       dbg = 0;
#ifdef ORBX_TUNING
    static const int dbgEnv = getenv("ORBX_DBG_SKIP_FAST") ? atoi(getenv("ORBX_DBG_SKIP_FAST")) : 0;  // phase ablation (wrong results by design)
    dbg = dbgEnv;
#endif
    const int ts = B <= octLatencyFrames ? 1 : 0;  // tiling: short tiles for a handful of frames
    if (tileGrid[ts] == 0) {
      // no level has a FAST cell: nothing to detect (the quadtree launch zeroes the counts)
    } else if (al4)
      hipLaunchKernelGGL((k_fast_tiles<true>), dim3(tileGrid[ts], B), dim3(kFastTileThreads), fastLds2[ts].total, s, b, g, dTiles[ts].p, dTileOrder[ts].p, t1, t2, fastLds2[ts], dbg);
    else
      hipLaunchKernelGGL((k_fast_tiles<false>), dim3(tileGrid[ts], B), dim3(kFastTileThreads), fastLds2[ts].total, s, b, g, dTiles[ts].p, dTileOrder[ts].p, t1, t2, fastLds2[ts], dbg);
  }
  if (prof) HIP_TRY(hipEventRecord(ev[2], s));
  // level-major grid: the large levels start first and every level spreads over all XCDs
  if (B <= octLatencyFrames)
    hipLaunchKernelGGL((k_octree<DevCtx, kOctThreadsLatency>), dim3(B, g.nlevels), dim3(kOctThreadsLatency), octLdsBytes, s, b, g, octLdsNodeCap, octLdsPtsCap);
  else
    hipLaunchKernelGGL((k_octree<WaveCtx, kOctThreads>), dim3(B, g.nlevels), dim3(kOctThreads), octLdsBytes, s, b, g, octLdsNodeCap, octLdsPtsCap);
  if (prof) HIP_TRY(hipEventRecord(ev[3], s));
  hipLaunchKernelGGL(k_layout, dim3(B), dim3(256), 0, s, b, g, lap0, lap1, d_counts, cap);
  if (prof) HIP_TRY(hipEventRecord(ev[4], s));
  launchBlur(s);
  if (prof) HIP_TRY(hipEventRecord(ev[5], s));
  {
    const int nbPerFrame = (g.totalKp + kOdKpPerBlock - 1) / kOdKpPerBlock;
    hipLaunchKernelGGL(k_orient_describe, dim3(8 * ((B + 7) / 8) * nbPerFrame), dim3(256), 0, s, b, g, dPattern.p, d_counts, d_kps, d_desc, cap, B,
                       nbPerFrame, tapMode);
  }
  if (prof) {
    HIP_TRY(hipEventRecord(ev[6], s));
    profCount++;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static bool is_pinned(const void* p);

extern "C" {

const char* orbx_last_error(void) { return g_err.c_str(); }

// Not part of include/orbx.h: the matcher and vocabulary translation units report through the same per-thread slot.
void orbx_internal_set_error(const char* s) { g_err = s ? s : ""; }

static hipStream_t create_queue_stream(int device_id);
void  orbx_internal_register(void* obj, void (*release_device)(void*));
void  orbx_internal_unregister(void* obj);
static void release_device_side(orbx_extractor* h);

int orbx_create(orbx_t** out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device_id) {
  if (!out) return -2;
  *out = nullptr;
  if (nfeatures < 1 || nlevels < 1 || nlevels > orbx::kMaxLevels || !(scale_factor > 1.0f)) {
    g_err = "orbx_create: bad parameters";
    return -2;
  }
  HIP_TRY(hipSetDevice(device_id));
  orbx_extractor* h = new orbx_extractor();
  EpochScope      es(&h->allocEpoch);
  arm_exit_handler();  // after the first HIP call above: runs before the runtime's own exit handlers
  h->nfeatures = nfeatures, h->nlevels = nlevels, h->iniTh = ini_th_fast, h->minTh = min_th_fast, h->scaleFactor = scale_factor;
  h->device = device_id;
  h->graphReplay = !(getenv("ORBX_GRAPH") != nullptr && atoi(getenv("ORBX_GRAPH")) == 0);  // on by default; ORBX_GRAPH=0 or orbx_set_graph_replay(h, 0) turns it off
  // src/ORBextractor.cc:409-440, exact float sequence
  h->sf.resize(nlevels), h->sigma2.resize(nlevels), h->invSf.resize(nlevels), h->invSigma2.resize(nlevels), h->quota.resize(nlevels);
  h->sf[0] = 1.0f, h->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    h->sf[i]     = h->sf[i - 1] * scale_factor;
    h->sigma2[i] = h->sf[i] * h->sf[i];
  }
  for (int i = 0; i < nlevels; i++) {
    h->invSf[i]     = 1.0f / h->sf[i];
    h->invSigma2[i] = 1.0f / h->sigma2[i];
  }
  float factor                   = 1.0f / scale_factor;
  float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int   sumFeatures              = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    h->quota[level] = cvRoundF(nDesiredFeaturesPerScale);
    sumFeatures += h->quota[level];
    nDesiredFeaturesPerScale *= factor;
  }
  h->quota[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
  // the handle's own stream owns a hardware queue (see orbx_stream_create): two handles driven side by side then really overlap
  h->stream    = create_queue_stream(device_id);
  hipError_t e = h->stream ? hipSuccess : hipErrorUnknown;
  if (e != hipSuccess) {
    g_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
    delete h;
    return -3;
  }
  if (h->dPattern.ensure(1) || h->dFault.ensure(4) || hipMemset(h->dFault.p, 0, 16) != hipSuccess) {
    delete h;
    return -3;
  }
  {
    static orbx::PatternTab tab;  // rBRIEF pattern + IC_Angle disc masks
    memcpy(tab.v, kPatternHost, 1024);
    static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // src/ORBextractor.cc:448-464
    for (int idx = 0; idx < orbx::kMomSlots; idx++) {
      const int v = (idx >> 3) - orbx::kHalfPatch, d = umax[v < 0 ? -v : v];
      uint32_t  m = 0;
      for (int t = 0; t < 4; t++) {
        const int u = -orbx::kHalfPatch + (idx & 7) * 4 + t;
        if (u >= -d && u <= d) m |= 0xffu << (8 * t);
      }
      tab.momMask[idx] = m;
    }
    e = hipMemcpy(h->dPattern.p, &tab, sizeof(tab), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    g_err = std::string("hipMemcpy(pattern): ") + hipGetErrorString(e);
    delete h;
    return -3;
  }
  orbx_internal_register(h, [](void* p) { release_device_side(static_cast<orbx_extractor*>(p)); });
  *out = h;
  return 0;
}

// Everything of a handle that lives on the device or in the runtime; leaves the host object valid (all pointers nulled).
static void release_device_side(orbx_extractor* h) {
  hipSetDevice(h->device);
  if (h->stream) {
    hipStreamSynchronize(h->stream);
    hipStreamDestroy(h->stream);
  }
  if (h->graphExec) hipGraphExecDestroy(h->graphExec);
  if (h->hOut) hipHostFree(h->hOut);
  if (h->hIn) hipHostFree(h->hIn);
  if (h->hPyr) hipHostFree(h->hPyr);
  for (hipStream_t st : {h->sIn, h->sOut})
    if (st) {
      hipStreamSynchronize(st);
      hipStreamDestroy(st);
    }
  for (auto& sl : h->slots) {
    sl.dIn.release(), sl.dDesc.release(), sl.dKps.release(), sl.dCounts.release();
    for (hipEvent_t e : {sl.evIn, sl.evCompute, sl.evDone})
      if (e) hipEventDestroy(e);
    if (sl.hIn) hipHostFree(sl.hIn);
    if (sl.hOut) hipHostFree(sl.hOut);
  }
  if (h->evValid)
    for (auto& slot : h->ev)
      for (auto& e : slot) hipEventDestroy(e);
  h->dPyr.release(), h->dBlur.release(), h->dStereoSad.release(), h->dStereoRowStart.release(), h->dStereoRowIdx.release(), h->sSad.release(), h->sKept.release(), h->sRow.release(), h->sRidx.release(), h->sKl.release(), h->sKr.release(), h->sDl.release(), h->sDr.release(), h->sU.release(), h->sDepth.release(), h->dRaw.release(), h->dPre.release(), h->dPreX.release(), h->dPreY.release(), h->dStage.release(), h->dDescOut.release(), h->dOutBlock.release(), h->dXtab.release(), h->dYtab.release();
  h->pyrLatency[0].dBands.release(), h->pyrLatency[1].dBands.release(), h->dTiles[0].release(), h->dTiles[1].release(), h->dTileOrder[0].release(), h->dTileOrder[1].release(), h->dCellSlot.release(), h->dCellCount.release(), h->dNodeOf.release(), h->dCellOffs.release();
  h->dD.release(), h->dDIndex.release(), h->dNCh.release(), h->dPushOff.release(), h->dMultiOff.release();
  h->dKeptRank.release(), h->dVA.release(), h->dVB.release(), h->dCandCount.release(), h->dLvlCount.release(), h->dOutIdx.release();
  h->dCounts.release(), h->dSlots.release(), h->dCand.release(), h->dLvlKp.release(), h->dNodesA.release(), h->dNodesB.release();
  h->dChildCnt.release(), h->dChildIdx.release(), h->dPattern.release(), h->dKpOut.release(), h->dFault.release();
  h->stream = nullptr, h->graphExec = nullptr, h->hOut = h->hIn = h->hPyr = nullptr, h->hOutCap = h->hInCap = h->hPyrCap = 0, h->sIn = h->sOut = nullptr;
  for (auto& sl : h->slots) sl.evIn = sl.evCompute = sl.evDone = nullptr, sl.hIn = sl.hOut = nullptr, sl.hInCap = sl.hOutCap = 0;
  h->evValid = false, h->haveGeom = false, h->capB = 0;
}

void orbx_destroy(orbx_t* h) {
  if (!h) return;
  orbx_internal_unregister(h);
  if (!g_runtimeDown.load()) release_device_side(h);  // else: process teardown, the exit handler has already released it (or the runtime is gone)
  delete h;
}

int orbx_get_levels(const orbx_t* h) { return h->nlevels; }

void orbx_get_scale_tables(const orbx_t* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
  for (int i = 0; i < h->nlevels; i++) {
    if (scale) scale[i] = h->sf[i];
    if (inv_scale) inv_scale[i] = h->invSf[i];
    if (sigma2) sigma2[i] = h->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->invSigma2[i];
  }
}

void orbx_get_features_per_level(const orbx_t* h, int* out) {
  for (int i = 0; i < h->nlevels; i++) out[i] = h->quota[i];
}

int orbx_set_input_resize(orbx_t* h, int new_cols, int new_rows) {
  if (!h || new_cols < 0 || new_rows < 0 || ((new_cols == 0) != (new_rows == 0))) return -2;
  h->preW = new_cols, h->preH = new_rows, h->preSrcW = h->preSrcH = 0;
  return 0;
}

// A HIP stream that owns its hardware queue.  HIP multiplexes ordinary streams onto a handful of hardware queues (least-referenced
// first), and two streams that share one execute in submission order -- chunks that are meant to overlap (the latency-bound quadtree and
// claim resolution of one chunk beside the VALU-bound FAST / blur / top-k of another) then do not.  A stream created with a CU mask
// always gets a queue of its own; the mask enables every CU.
static hipStream_t create_queue_stream(int device_id) {
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  const uint32_t allCus[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  hipStream_t    st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, 8, allCus) != hipSuccess) {
    (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      g_err = "orbx_stream_create: hipStreamCreate failed";
      return nullptr;
    }
  }
  return st;
}

void* orbx_stream_create(int device_id) {
  hipStream_t st = create_queue_stream(device_id);
  if (st)
    orbx_internal_register((void*)st, [](void* p) {
      hipStreamSynchronize((hipStream_t)p);
      hipStreamDestroy((hipStream_t)p);
    });
  return (void*)st;
}

void orbx_stream_destroy(void* stream) {
  if (!stream) return;
  orbx_internal_unregister(stream);
  if (!g_runtimeDown.load()) {
    hipStreamSynchronize((hipStream_t)stream);
    hipStreamDestroy((hipStream_t)stream);
  }
}

int orbx_set_tap_contraction(orbx_t* h, int mode) {
  if (!h || mode < 0 || mode > 2) return -2;
  h->tapMode = mode;
  return 0;
}

int orbx_set_pyramid_download(orbx_t* h, int enable) {
  if (!h) return -2;
  h->pyrToHost = enable != 0;
  if (!h->pyrToHost) h->hPyrValid = false;
  return 0;
}

int orbx_set_graph_replay(orbx_t* h, int enable) {
  if (!h) return -2;
  h->graphReplay = enable != 0;
  if (!h->graphReplay && h->graphExec) hipGraphExecDestroy(h->graphExec), h->graphExec = nullptr;
  h->graphSeen = false;
  return 0;
}

int orbx_max_keypoints(orbx_t* h, int rows, int cols) {
  if (!h) return -2;
  EpochScope es(&h->allocEpoch);
  hipSetDevice(h->device);
  h->effective(rows, cols, &rows, &cols);
  int rc = h->makeGeometry(rows, cols);
  if (rc) return rc < 0 ? rc : -2;
  return h->geom.totalKp;
}

int orbx_extract_batch_device(orbx_t* h, const uint8_t* d_imgs, int nframes, int rows, int cols, size_t step_bytes,
                              size_t frame_stride_bytes, int lap0, int lap1, orbx_keypoint_t* d_kps, uint8_t* d_desc, int* d_counts,
                              int cap, void* stream) {
  if (!h || !d_kps || !d_desc || !d_counts || nframes < 0) {
    g_err = "orbx_extract_batch_device: null argument";
    return -2;
  }
  if (!d_imgs || rows <= 0 || cols <= 0 || nframes == 0) return -1;
  if (step_bytes < (size_t)cols) {
    g_err = "step_bytes < cols";
    return -2;
  }
  EpochScope es(&h->allocEpoch);
  HIP_TRY(hipSetDevice(h->device));
  int er, ec;
  h->effective(rows, cols, &er, &ec);
  int rc = h->makeGeometry(er, ec);
  if (rc) return rc;
  if (cap < h->geom.totalKp) {
    g_err = "output capacity below orbx_max_keypoints()";
    return -2;
  }
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  return h->runResized(d_imgs, nframes, rows, cols, step_bytes, frame_stride_bytes, lap0, lap1, (orbx::KpRec*)d_kps, d_desc, d_counts, cap, s);
}

int orbx_extract(orbx_t* h, const uint8_t* img, int rows, int cols, size_t step_bytes, int lap0, int lap1, orbx_keypoint_t* kps,
                 uint8_t* desc, int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!h) return -2;
  if (!img || rows <= 0 || cols <= 0) return -1;  // _image.empty() (src/ORBextractor.cc:1060-1062)
  if (!kps || !desc || step_bytes < (size_t)cols) {
    g_err = "orbx_extract: bad arguments";
    return -2;
  }
  EpochScope es(&h->allocEpoch);
  HIP_TRY(hipSetDevice(h->device));
  int er, ec;
  h->effective(rows, cols, &er, &ec);
  int rc = h->makeGeometry(er, ec);
  if (rc) return rc;
  const int need = h->geom.totalKp;
  if (cap < need) {
    g_err = "orbx_extract: cap below orbx_max_keypoints()";
    return -2;
  }
  const int pitch = alignUp(cols, 64);
  // the records of the call live in ONE device block laid out like the pinned host block they are copied to: {counts, pad to 16 B,
  // keypoints[need], descriptors[need][32]} -- one download instead of three
  const size_t kpBytes = (size_t)need * sizeof(orbx_keypoint_t), dBytes = (size_t)need * 32, outBytes = 16 + kpBytes + dBytes;
  if (h->dStage.ensure((size_t)pitch * rows) || h->dOutBlock.ensure(outBytes)) return -3;
  int* const             dCountsP = reinterpret_cast<int*>(h->dOutBlock.p);
  orbx_keypoint_t* const dKpP     = reinterpret_cast<orbx_keypoint_t*>(h->dOutBlock.p + 16);
  uint8_t* const         dDescP   = h->dOutBlock.p + 16 + kpBytes;
  // Host -> device as ONE linear copy (a pitched copy from pageable memory degenerates to a row-by-row transfer: 3 ms for a
  // 1241-px-wide image), then, if the caller's step is not the 64-byte aligned pitch the kernels like, a device-side repitch.
  // The caller's buffer is pageable: the runtime would stage it through its own pinned chunks with a synchronisation per chunk.
  // One host memcpy into the handle's pinned buffer and one DMA is shorter.
  const size_t rawBytes = (int)step_bytes == pitch ? (size_t)pitch * rows : step_bytes * (size_t)(rows - 1) + cols;
  const uint8_t*    hsrc     = img;
  if (!is_pinned(img)) {  // an image that already lives in pinned memory (orbx_host_alloc) is uploaded from where it is
    if (h->hInCap < rawBytes) {
      if (h->hIn) hipHostFree(h->hIn);
      h->hIn = nullptr, h->hInCap = 0;
      HIP_TRY(hipHostMalloc((void**)&h->hIn, rawBytes, hipHostMallocDefault));
      h->hInCap = rawBytes;
    }
    memcpy(h->hIn, img, rawBytes);
    hsrc = h->hIn;
  }
  if ((int)step_bytes != pitch && h->dRaw.ensure(rawBytes)) return -3;
  // Device -> host: the count is not known before the copies are queued, so the whole fixed-capacity record block (<= 75 KB at 1000
  // features, 3 us of PCIe) lands in pinned memory behind the kernels and ONE synchronisation ends the call; the n live records
  // are then copied out on the host.  (Counts first, then two dependent blocking copies cost two more round trips: 0.26 -> 0.2x ms.)
  if (h->hOutCap < outBytes) {
    if (h->hOut) hipHostFree(h->hOut);
    h->hOut = nullptr, h->hOutCap = 0;
    HIP_TRY(hipHostMalloc((void**)&h->hOut, outBytes, hipHostMallocDefault));
    h->hOutCap = outBytes;
  }
  const size_t pyrBytes = h->pyrToHost ? (size_t)h->geom.pyrBytes : 0;  // levels >= 1 ride home behind the records: no extra round trip
  if (pyrBytes && h->hPyrCap < pyrBytes) {
    if (h->hPyr) hipHostFree(h->hPyr);
    h->hPyr = nullptr, h->hPyrCap = 0;
    HIP_TRY(hipHostMalloc((void**)&h->hPyr, pyrBytes, hipHostMallocDefault));
    h->hPyrCap = pyrBytes;
  }
  auto enqueue = [&]() -> int {  // upload, the kernel chain, the three downloads: everything between the host copies in and out
    if ((int)step_bytes == pitch) {
      HIP_TRY(hipMemcpyAsync(h->dStage.p, hsrc, rawBytes, hipMemcpyHostToDevice, h->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(h->dRaw.p, hsrc, rawBytes, hipMemcpyHostToDevice, h->stream));
      hipLaunchKernelGGL(orbx::k_repitch, dim3((cols + 1023) / 1024, rows, 1), dim3(256), 0, h->stream, h->dRaw.p, (long long)step_bytes, 0ll, h->dStage.p, pitch, 0ll, cols, rows);
    }
    const int rc2 = h->runResized(h->dStage.p, 1, rows, cols, pitch, (size_t)pitch * rows, lap0, lap1, reinterpret_cast<orbx::KpRec*>(dKpP), dDescP, dCountsP, need, h->stream);
    if (rc2) return rc2;
    HIP_TRY(hipMemcpyAsync(h->hOut, h->dOutBlock.p, outBytes, hipMemcpyDeviceToHost, h->stream));
    if (pyrBytes && h->nlevels > 1) HIP_TRY(hipMemcpyAsync(h->hPyr, h->dPyr.p, pyrBytes, hipMemcpyDeviceToHost, h->stream));
    return 0;
  };
  // Replay form (default; orbx_set_graph_replay(h, 0) / ORBX_GRAPH=0 turn it off): the enqueue sequence of a call is a pure function of this key, so the second
  // call with the same key records it into a hipGraph (stream capture) and later calls replay it with one launch.
  bool done = false;
  if (h->graphReplay && !h->profiling && (rc = h->ensureWorkspace(er, ec, 1))) return rc;  // allocations happen before the key is taken
  const orbx_extractor::GraphKey key{rows, cols, step_bytes, lap0, lap1, need, hsrc, h->hOut, h->tapMode, h->preW, h->preH, pyrBytes ? h->hPyr : nullptr, h->allocEpoch, g_allocEpochFallback.load()};
  if (h->graphReplay && !h->profiling) {
    if (h->graphExec && key == h->graphKey) {
      HIP_TRY(hipGraphLaunch(h->graphExec, h->stream));
      // host-side state an eager enqueue leaves behind (run()): level 0 of the pyramid accessors and of orbx_stereo_matches is this call's
      // staged image again, whatever a batch / async call on the same handle pointed it at in between
      h->bufs.img0 = h->graphImg0, h->bufs.img0Pitch = h->graphImg0Pitch, h->bufs.img0FrameStride = h->graphImg0Stride, h->lastB = 1;
      done = true;
    } else if (h->graphSeen && key == h->graphKey) {  // seen once eagerly (every allocation has happened): record
      if (h->graphExec) hipGraphExecDestroy(h->graphExec), h->graphExec = nullptr;
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int  rc3 = enqueue();
        hipError_t e   = hipStreamEndCapture(h->stream, &graph);
        if (rc3 == 0 && e == hipSuccess && graph && hipGraphInstantiate(&h->graphExec, graph, nullptr, nullptr, 0) == hipSuccess) {
          HIP_TRY(hipGraphLaunch(h->graphExec, h->stream));
          h->graphImg0 = h->bufs.img0, h->graphImg0Pitch = h->bufs.img0Pitch, h->graphImg0Stride = h->bufs.img0FrameStride;  // set by run() during the capture
          done = true;
        } else {
          (void)hipGetLastError();
          h->graphExec = nullptr, h->graphReplay = false;  // this runtime cannot capture the sequence: stay eager
        }
        if (graph) hipGraphDestroy(graph);
      } else {
        (void)hipGetLastError();
        h->graphReplay = false;
      }
    } else {
      if (h->graphExec) hipGraphExecDestroy(h->graphExec), h->graphExec = nullptr;
      h->graphKey = key, h->graphSeen = true;
    }
  }
  if (!done && (rc = enqueue())) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->hPyrValid      = pyrBytes != 0;
  const int* counts = reinterpret_cast<const int*>(h->hOut);
  const int  n      = std::min(counts[0], need);
  if (n > 0) {
    memcpy(kps, h->hOut + 16, (size_t)n * sizeof(orbx_keypoint_t));
    memcpy(desc, h->hOut + 16 + kpBytes, (size_t)n * 32);
  }
  if (n_out) *n_out = n;
  return counts[1];
}

// ---- streaming form: host frames in, host records out, several chunks in flight --------------------------------------------
void* orbx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void orbx_host_free(void* p) {
  if (p && !g_runtimeDown.load()) hipHostFree(p);
}

// Not part of include/orbx.h: the other translation units ask before they touch the device in their destroy paths, and enrol their
// handles for the exit handler.
int orbx_internal_runtime_down(void) { return g_runtimeDown.load() ? 1 : 0; }

void orbx_internal_register(void* obj, void (*release_device)(void*)) {
  arm_exit_handler();
  std::lock_guard<std::mutex> lock(live_handles().m);
  live_handles().v.emplace_back(obj, release_device);
}

void orbx_internal_unregister(void* obj) {
  std::lock_guard<std::mutex> lock(live_handles().m);
  auto& v = live_handles().v;
  for (size_t i = 0; i < v.size(); i++)
    if (v[i].first == obj) {
      v.erase(v.begin() + i);
      break;
    }
}

static bool is_pinned(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int orbx_set_async_depth(orbx_t* h, int depth) {
  if (!h || depth < 1 || depth > orbx_extractor::kMaxAsyncDepth) return -2;
  for (auto& sl : h->slots)
    if (sl.ticket >= 0) {
      g_err = "orbx_set_async_depth: chunks are in flight";
      return -2;
    }
  h->asyncDepth = depth;
  return 0;
}

static int ensure_pinned(uint8_t** p, size_t* capBytes, size_t need) {
  if (*capBytes >= need) return 0;
  if (*p) hipHostFree(*p);
  *p = nullptr, *capBytes = 0;
  HIP_TRY(hipHostMalloc((void**)p, need, hipHostMallocDefault));
  *capBytes = need;
  return 0;
}

// Device -> pinned host on `s`: the low-occupancy copy kernel when both ends are 16-byte aligned, else the runtime's copy.
static int copy_out(hipStream_t s, void* host_dst, const void* dev_src, size_t bytes) {
  if (bytes == 0) return 0;
  if (((reinterpret_cast<uintptr_t>(host_dst) | reinterpret_cast<uintptr_t>(dev_src)) & 15) == 0 && (bytes & 3) == 0) {
    const long long n16 = (long long)(bytes >> 4);
    hipLaunchKernelGGL(orbx::k_copy_out, dim3(orbx::kCopyOutBlocks), dim3(256), 0, s, (const orbx::u32x4*)dev_src, (orbx::u32x4*)host_dst, n16, (int)((bytes & 15) >> 2));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
  return 0;
}

// Enqueue the copy-out of a slot's records behind everything that is on the kernel stream NOW.
static int flush_copy_out(orbx_t* h, orbx_extractor::AsyncSlot& sl) {
  if (!sl.pendingOut) return 0;
  sl.pendingOut = false;
  const size_t nk = (size_t)sl.nframes * sl.cap;
  const size_t kpBytes = nk * sizeof(orbx_keypoint_t), dBytes = nk * 32, cBytes = (size_t)sl.nframes * 2 * sizeof(int);
  HIP_TRY(hipEventRecord(sl.evCompute, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->sOut, sl.evCompute, 0));
  int rc;
  if (sl.stagedOut) {
    rc = ensure_pinned(&sl.hOut, &sl.hOutCap, kpBytes + dBytes + cBytes);
    if (rc) return rc;
    if ((rc = copy_out(h->sOut, sl.hOut, sl.dKps.p, kpBytes)) || (rc = copy_out(h->sOut, sl.hOut + kpBytes, sl.dDesc.p, dBytes)) ||
        (rc = copy_out(h->sOut, sl.hOut + kpBytes + dBytes, sl.dCounts.p, cBytes)))
      return rc;
  } else {
    if ((rc = copy_out(h->sOut, sl.uKps, sl.dKps.p, kpBytes)) || (rc = copy_out(h->sOut, sl.uDesc, sl.dDesc.p, dBytes)) ||
        (rc = copy_out(h->sOut, sl.uCounts, sl.dCounts.p, cBytes)))
      return rc;
  }
  HIP_TRY(hipEventRecord(sl.evDone, h->sOut));
  return 0;
}

int orbx_extract_async(orbx_t* h, const uint8_t* imgs, int nframes, int rows, int cols, size_t step_bytes, size_t frame_stride_bytes, int lap0,
                       int lap1, orbx_keypoint_t* kps, uint8_t* desc, int* counts, int cap, int* ticket) {
  if (ticket) *ticket = -1;
  if (!h || !ticket || nframes < 0) return -2;
  if (!imgs || rows <= 0 || cols <= 0 || nframes == 0) return -1;
  if (!kps || !desc || !counts || step_bytes < (size_t)cols || (nframes > 1 && frame_stride_bytes < step_bytes * (size_t)(rows - 1) + cols)) {
    g_err = "orbx_extract_async: bad arguments";
    return -2;
  }
  EpochScope es(&h->allocEpoch);
  HIP_TRY(hipSetDevice(h->device));
  int er, ec;
  h->effective(rows, cols, &er, &ec);
  int rc = h->makeGeometry(er, ec);
  if (rc) return rc;
  if (cap < h->geom.totalKp) {
    g_err = "orbx_extract_async: cap below orbx_max_keypoints()";
    return -2;
  }
  const int                  t   = h->nextTicket;
  orbx_extractor::AsyncSlot* slp = nullptr;  // any free slot: tickets may be waited for in any order
  for (int i = 0; i < h->asyncDepth && !slp; i++)
    if (h->slots[(t + i) % h->asyncDepth].ticket < 0) slp = &h->slots[(t + i) % h->asyncDepth];
  if (!slp) {
    g_err = "orbx_extract_async: all slots in flight (orbx_wait a ticket first, or raise orbx_set_async_depth)";
    return -2;
  }
  orbx_extractor::AsyncSlot& sl = *slp;
  if (!h->sIn) {
    HIP_TRY(hipStreamCreateWithFlags(&h->sIn, hipStreamNonBlocking));
    // The copy-out kernels must not share a hardware queue with the extraction kernels: HIP multiplexes its streams onto a few
    // hardware queues (least-referenced first), and two streams on one queue execute in submission order -- measured, the copy-out
    // then ran strictly between orientation and matching instead of beside them.  A stream created with a CU mask always owns a
    // queue; the mask itself enables every CU.
    {
      const uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (hipExtStreamCreateWithCUMask(&h->sOut, 8, mask) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipStreamCreateWithFlags(&h->sOut, hipStreamNonBlocking));
      }
    }
  }
  if (!sl.evIn) {
    HIP_TRY(hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.evCompute, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
  }
  for (auto& other : h->slots)  // records of earlier tickets that nobody has asked for yet go out now
    if (other.ticket >= 0 && (rc = flush_copy_out(h, other))) return rc;
  const size_t inBytes = frame_stride_bytes * (size_t)(nframes - 1) + step_bytes * (size_t)(rows - 1) + cols;
  const size_t nk = (size_t)nframes * cap;
  if (sl.dIn.ensure(inBytes + 256) || sl.dKps.ensure(nk) || sl.dDesc.ensure(nk * 32) || sl.dCounts.ensure((size_t)nframes * 2)) return -3;
  // upload: straight from the caller's memory when it is pinned (orbx_host_alloc / hipHostRegister), else through pinned staging
  const uint8_t* hsrc = imgs;
  if (!is_pinned(imgs)) {
    rc = ensure_pinned(&sl.hIn, &sl.hInCap, inBytes);
    if (rc) return rc;
    memcpy(sl.hIn, imgs, inBytes);
    hsrc = sl.hIn;
  }
  HIP_TRY(hipMemcpyAsync(sl.dIn.p, hsrc, inBytes, hipMemcpyHostToDevice, h->sIn));
  HIP_TRY(hipEventRecord(sl.evIn, h->sIn));
  HIP_TRY(hipStreamWaitEvent(h->stream, sl.evIn, 0));
  rc = h->runResized(sl.dIn.p, nframes, rows, cols, step_bytes, frame_stride_bytes, lap0, lap1, sl.dKps.p, sl.dDesc.p, sl.dCounts.p, cap, h->stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(sl.evCompute, h->stream));
  // The download of the fixed-capacity records is enqueued LATER (flush_copy_out): at the first orbx_async_download / orbx_wait for
  // this ticket or at the next submission.  Chained work (a matcher on the ticket's stream) is then already in the queue, the
  // copy-out waits for it too and runs beside the NEXT chunk's pyramid / FAST launches -- thousands of small workgroups that share
  // the machine gracefully -- instead of beside the chained kernels (measured: the brute-force match took twice as long with the
  // copy-out next to it).
  sl.stagedOut = !(is_pinned(kps) && is_pinned(desc) && is_pinned(counts));
  sl.pendingOut = true;
  sl.ticket = t, sl.nframes = nframes, sl.cap = cap, sl.uKps = kps, sl.uDesc = desc, sl.uCounts = counts;
  h->nextTicket = t + 1;
  *ticket = t;
  return 0;
}

static orbx_extractor::AsyncSlot* find_slot(orbx_t* h, int ticket) {
  if (!h || ticket < 0) return nullptr;
  for (auto& sl : h->slots)
    if (sl.ticket == ticket) return &sl;
  return nullptr;
}

int orbx_async_device_buffers(orbx_t* h, int ticket, const orbx_keypoint_t** d_kps, const uint8_t** d_desc, const int** d_counts, void** stream) {
  orbx_extractor::AsyncSlot* sl = find_slot(h, ticket);
  if (!sl) {
    g_err = "orbx_async_device_buffers: unknown ticket";
    return -2;
  }
  if (d_kps) *d_kps = (const orbx_keypoint_t*)sl->dKps.p;
  if (d_desc) *d_desc = sl->dDesc.p;
  if (d_counts) *d_counts = sl->dCounts.p;
  if (stream) *stream = (void*)h->stream;
  return 0;
}

int orbx_async_download(orbx_t* h, int ticket, void* host_dst, const void* device_src, size_t bytes) {
  orbx_extractor::AsyncSlot* sl = find_slot(h, ticket);
  if (!sl || !host_dst || !device_src) {
    g_err = "orbx_async_download: unknown ticket or null pointer";
    return -2;
  }
  if (bytes == 0) return 0;
  if (!is_pinned(host_dst)) {
    g_err = "orbx_async_download: the destination must be pinned host memory (orbx_host_alloc)";
    return -2;
  }
  HIP_TRY(hipSetDevice(h->device));
  // behind everything enqueued so far on the ticket's kernel stream, on the copy-out stream; orbx_wait(ticket) then covers it too
  int rc = 0;
  if (sl->pendingOut) {
    rc = flush_copy_out(h, *sl);  // records its own event behind the chained work
    if (rc) return rc;
  } else {
    HIP_TRY(hipEventRecord(sl->evCompute, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->sOut, sl->evCompute, 0));
  }
  rc = copy_out(h->sOut, host_dst, device_src, bytes);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(sl->evDone, h->sOut));
  return 0;
}

int orbx_wait(orbx_t* h, int ticket) {
  orbx_extractor::AsyncSlot* sl = find_slot(h, ticket);
  if (!sl) {
    g_err = "orbx_wait: unknown ticket";
    return -2;
  }
  // whatever happens below, the ticket is over: a slot that stayed reserved after an error would be lost for good
  struct Release {
    orbx_extractor::AsyncSlot* sl;
    ~Release() { sl->ticket = -1, sl->pendingOut = false; }
  } release{sl};
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_copy_out(h, *sl);
  if (rc) return rc;
  HIP_TRY(hipEventSynchronize(sl->evDone));
  if (sl->stagedOut) {
    const size_t nk = (size_t)sl->nframes * sl->cap, kpBytes = nk * sizeof(orbx_keypoint_t), dBytes = nk * 32;
    memcpy(sl->uKps, sl->hOut, kpBytes);
    memcpy(sl->uDesc, sl->hOut + kpBytes, dBytes);
    memcpy(sl->uCounts, sl->hOut + kpBytes + dBytes, (size_t)sl->nframes * 2 * sizeof(int));
  }
  return 0;
}

static int copy_level(orbx_t* h, const uint8_t* base, int pitch, int w, int hgt, uint8_t* dst, size_t dst_step, int* rows, int* cols) {
  if (rows) *rows = hgt;
  if (cols) *cols = w;
  if (!dst) return 0;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy2D(dst, dst_step, base, pitch, w, hgt, hipMemcpyDeviceToHost));
  return 0;
}

int orbx_pyramid_level(orbx_t* h, int frame, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols) {
  if (!h || !h->haveGeom || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastB) return -2;
  HIP_TRY(hipSetDevice(h->device));
  const orbx::LevelGeom& L = h->geom.lv[level];
  if (level == 0)
    return copy_level(h, h->bufs.img0 + (long long)frame * h->bufs.img0FrameStride, h->bufs.img0Pitch, L.w, L.h, dst, dst_step, rows, cols);
  if (h->hPyrValid && frame == 0) {  // orbx_set_pyramid_download: the level already lies in host memory, no device call
    if (rows) *rows = L.h;
    if (cols) *cols = L.w;
    if (dst)
      for (int r = 0; r < L.h; r++) memcpy(dst + (size_t)r * dst_step, h->hPyr + L.pyrOff + (size_t)r * L.pitch, (size_t)L.w);
    return 0;
  }
  return copy_level(h, h->bufs.pyr + (long long)frame * h->geom.pyrBytes + L.pyrOff, L.pitch, L.w, L.h, dst, dst_step, rows, cols);
}

int orbx_debug_blurred_level(orbx_t* h, int frame, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols) {
  if (!h || !h->haveGeom || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastB) return -2;
  HIP_TRY(hipSetDevice(h->device));
  const orbx::LevelGeom& L = h->geom.lv[level];
  if (orbx::level_is_tiny(L)) {
    g_err = "orbx_debug_blurred_level: levels below 8 x 8 px are not blurred (no keypoint can lie on them)";
    return -2;
  }
  return copy_level(h, h->bufs.blur + (long long)frame * h->geom.blurBytes + L.blurOff, L.blurPitch, L.w, L.h, dst, dst_step, rows, cols);
}

static int copy_counted(orbx_t* h, const int* d_count, const uint32_t* d_src, uint32_t* out, int cap) {
  int n = 0;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(&n, d_count, sizeof(int), hipMemcpyDeviceToHost));
  if (out && n > 0) HIP_TRY(hipMemcpy(out, d_src, (size_t)std::min(n, cap) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return n;
}

int orbx_debug_level_candidates(orbx_t* h, int frame, int level, uint32_t* out, int cap) {
  if (!h || !h->haveGeom || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastB) return -2;
  HIP_TRY(hipSetDevice(h->device));
  const orbx::FrameGeom& g = h->geom;
  return copy_counted(h, h->bufs.candCount + (long long)frame * g.nlevels + level,
                      h->bufs.cand + (long long)frame * g.totalSlots + g.lv[level].slotBase, out, cap);
}

int orbx_debug_level_keypoints(orbx_t* h, int frame, int level, uint32_t* out, int cap) {
  if (!h || !h->haveGeom || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastB) return -2;
  HIP_TRY(hipSetDevice(h->device));
  const orbx::FrameGeom& g = h->geom;
  return copy_counted(h, h->bufs.lvlCount + (long long)frame * g.nlevels + level,
                      h->bufs.lvlKp + (long long)frame * g.totalKp + g.lv[level].kpBase, out, cap);
}

int orbx_debug_octree_device(orbx_t* h, const uint32_t* cand, int n, int W, int H, int N, uint32_t* out, int* out_count) {
  return orbx_debug_octree_sorts(h, cand, n, W, H, N, out, out_count, nullptr, 0);
}

// The same, and the records of every std::sort re-enactment of the run as the kernel saw them: sorts[0] = their number, then per sort
// 1 + 2 * 256 words (V, the V records before, the V records after); max_sorts bounds what is kept.  Diagnosis hook (tools/dbg_octree67.py).
int orbx_debug_octree_sorts(orbx_t* h, const uint32_t* cand, int n, int W, int H, int N, uint32_t* out, int* out_count, unsigned long long* sorts,
                            int max_sorts) {
  using namespace orbx;
  if (!h || n < 0 || W <= 0 || H <= 0 || N < 0 || max_sorts < 0) return -2;
  unsigned long long scratchEpoch = 0;  // the temporaries below are not the handle's: they retire nobody's graph
  EpochScope         es(&scratchEpoch);
  HIP_TRY(hipSetDevice(h->device));
  const int nIni = (int)std::round((float)W / (float)H);
  if (nIni <= 0) return -2;
  const int    maxNodes = alignUp(std::max(N + 3, 4 * nIni) + 16, 16);
  const size_t np = (size_t)n + 16, nn = (size_t)maxNodes;
  DevBuf<uint32_t>           dc, dout;
  DevBuf<uint16_t>           nA, D, dI, pO, mO, kR, vA, vB;
  DevBuf<int>                cnt;
  DevBuf<uint8_t>            nCh;
  DevBuf<OtNode>             nodesA, nodesB;
  DevBuf<Int4>               sc;
  DevBuf<Idx4>               ec;
  DevBuf<unsigned long long> sb, bst, dsort;
  const size_t               sortWords = sorts ? 1 + (size_t)max_sorts * 513 : 0;
  if (sorts && (dsort.ensure(sortWords) || hipMemset(dsort.p, 0, sortWords * 8) != hipSuccess)) return -3;
  if (dc.ensure(np) || dout.ensure(nn) || nA.ensure(np) || bst.ensure(nn) ||
      nodesA.ensure(nn) || nodesB.ensure(nn) || sc.ensure(nn) || ec.ensure(nn) || D.ensure(nn) || dI.ensure(nn) ||
      nCh.ensure(nn) || pO.ensure(nn) || mO.ensure(nn) || kR.ensure(nn) || vA.ensure(nn) || vB.ensure(nn) || sb.ensure(nn) || cnt.ensure(1))
    return -3;
  if (n > 0) HIP_TRY(hipMemcpy(dc.p, cand, (size_t)n * 4, hipMemcpyHostToDevice));
  OtArgs a;
  memset(&a, 0, sizeof(a));
  a.dbgT = nullptr;
  a.cand = dc.p, a.n = n, a.W = W, a.H = H, a.N = N, a.nIni = nIni, a.hX = (float)W / nIni, a.maxNodes = maxNodes;
  a.nodeOf = nA.p, a.best = bst.p, a.nodesA = nodesA.p, a.nodesB = nodesB.p;
  a.childCnt = sc.p, a.childIdx = ec.p, a.D = D.p, a.dIndex = dI.p, a.nCh = nCh.p, a.pushOff = pO.p, a.multiOff = mO.p, a.keptRank = kR.p;
  a.vA = vA.p, a.vB = vB.p, a.sortbuf = sb.p, a.outKp = dout.p, a.outCount = cnt.p;
  a.dbgSort = sorts ? dsort.p : nullptr, a.dbgSortCap = max_sorts;
  hipLaunchKernelGGL(k_octree_only, dim3(1), dim3(kOctThreads), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  int c = 0;
  HIP_TRY(hipMemcpy(&c, cnt.p, sizeof(int), hipMemcpyDeviceToHost));
  if (c > 0) HIP_TRY(hipMemcpy(out, dout.p, (size_t)c * 4, hipMemcpyDeviceToHost));
  if (out_count) *out_count = c;
  if (sorts) HIP_TRY(hipMemcpy(sorts, dsort.p, (1 + (size_t)max_sorts * 513) * 8, hipMemcpyDeviceToHost));
  dsort.release();
  dc.release(), dout.release(), nA.release(), bst.release(), D.release(), dI.release();
  nCh.release(), pO.release(), mO.release(), kR.release(), vA.release(), vB.release(), cnt.release(), nodesA.release();
  nodesB.release(), sc.release(), ec.release(), sb.release();
  return 0;
}

// Parity hook: the kernel's std::sort re-enactment (ot_wave_sort, one wave) on n host records, in place; order = (record >> 16), the
// low 16 bits are payload; depth_limit < 0 = the library's own (small values reach its heap-sort path on any input).  What
// orbx_debug_std_sort (the host build of the library algorithm) is compared with.
int orbx_debug_wave_sort(orbx_t* h, unsigned long long* records, int n, int depth_limit) {
  using namespace orbx;
  if (!h || !records || n < 0) return -2;
  if (n == 0) return 0;
  unsigned long long scratchEpoch = 0;
  EpochScope         es(&scratchEpoch);
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<unsigned long long> d;
  if (d.ensure((size_t)n)) return -3;
  HIP_TRY(hipMemcpy(d.p, records, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_wave_sort_only, dim3(1), dim3(64), 0, h->stream, d.p, n, depth_limit);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(records, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  d.release();
  return 0;
}

static orbx::UndistortParams undistortParams(const float K[4], const float* dist, int ndist) {
  orbx::UndistortParams p;
  p.fx = K[0], p.fy = K[1], p.cx = K[2], p.cy = K[3];  // cvConvert(float -> double)
  for (int i = 0; i < 12; i++) p.k[i] = i < ndist ? (double)dist[i] : 0.0;
  p.pfx = p.fx, p.pfy = p.fy, p.pcx = p.cx, p.pcy = p.cy;  // P = mK
  return p;
}

// Frame::ComputeImageBounds (src/Frame.cc:915-943): four corners through the same arithmetic, on the host.
int orbx_image_bounds(int cols, int rows, const float K[4], const float* dist, int ndist, float out[4]) {
  if (!K || !out || ndist < 0 || ndist > 12 || (ndist > 0 && !dist)) return -2;
  if (ndist >= 1 && dist[0] != 0.0) {
    const orbx::UndistortParams p = undistortParams(K, dist, ndist);
    const float                 src[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
    float                       m[4][2];
    for (int i = 0; i < 4; i++) orbx::undistort_point(p, src[i][0], src[i][1], &m[i][0], &m[i][1]);
    out[0] = std::min(m[0][0], m[2][0]);
    out[1] = std::max(m[1][0], m[3][0]);
    out[2] = std::min(m[0][1], m[1][1]);
    out[3] = std::max(m[2][1], m[3][1]);
  } else {
    out[0] = 0.0f, out[1] = (float)cols, out[2] = 0.0f, out[3] = (float)rows;
  }
  return 0;
}

int orbx_undistort_keypoints_device(orbx_t* h, const orbx_keypoint_t* d_kps, const int* d_counts, int nframes, int cap, const float K[4],
                                    const float* dist, int ndist, orbx_keypoint_t* d_out, void* stream) {
  if (!h || !d_kps || !d_out || !K || nframes < 0 || cap < 0 || ndist < 0 || ndist > 12 || (ndist > 0 && !dist)) return -2;
  if (nframes == 0 || cap == 0) return 0;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (ndist < 1 || dist[0] == 0.0) {  // mvKeysUn = mvKeys (src/Frame.cc:887-890)
    if (d_out != d_kps) HIP_TRY(hipMemcpyAsync(d_out, d_kps, (size_t)nframes * cap * sizeof(orbx_keypoint_t), hipMemcpyDeviceToDevice, s));
    return 0;
  }
  hipLaunchKernelGGL(orbx::k_undistort, dim3((cap + 255) / 256, nframes), dim3(256), 0, s, (const orbx::KpRec*)d_kps, d_counts, cap,
                     undistortParams(K, dist, ndist), (orbx::KpRec*)d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int orbx_undistort_keypoints(orbx_t* h, const orbx_keypoint_t* kps, int n, const float K[4], const float* dist, int ndist, orbx_keypoint_t* out) {
  if (!h || n < 0 || (n > 0 && (!kps || !out)) || !K) return -2;
  if (n == 0) return 0;
  EpochScope es(&h->allocEpoch);
  HIP_TRY(hipSetDevice(h->device));
  if (h->dKpOut.ensure((size_t)2 * n)) return -3;
  HIP_TRY(hipMemcpyAsync(h->dKpOut.p, kps, (size_t)n * sizeof(orbx_keypoint_t), hipMemcpyHostToDevice, h->stream));
  int rc = orbx_undistort_keypoints_device(h, (const orbx_keypoint_t*)h->dKpOut.p, nullptr, 1, n, K, dist, ndist, (orbx_keypoint_t*)(h->dKpOut.p + n), h->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->dKpOut.p + n, (size_t)n * sizeof(orbx_keypoint_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// Frame::ComputeStereoMatches (reference src/Frame.cc:944-1113).  Both handles must have extracted an image of the same
// size with their single-frame orbx_extract (the pyramids of that call are still resident on the device).
int orbx_stereo_matches(orbx_t* left, orbx_t* right, const orbx_keypoint_t* kl, const uint8_t* dl, int nl, const orbx_keypoint_t* kr,
                        const uint8_t* dr, int nr, float mb, float mbf, float* u_right, float* depth) {
  using namespace orbx;
  if (!left || !right || nl < 0 || nr < 0 || (nl > 0 && (!kl || !dl || !u_right || !depth)) || (nr > 0 && (!kr || !dr)) || !(mb > 0.f)) {
    g_err = "orbx_stereo_matches: bad arguments";
    return -2;
  }
  if (!left->haveGeom || !right->haveGeom || left->lastB < 1 || right->lastB < 1 || left->geom.rows != right->geom.rows ||
      left->geom.cols != right->geom.cols || left->nlevels != right->nlevels || left->device != right->device) {
    g_err = "orbx_stereo_matches: both extractors must hold the pyramid of a same-sized image on the same device";
    return -2;
  }
  if (nr > 65535) {
    g_err = "orbx_stereo_matches: more than 65535 right keypoints";
    return -2;
  }
  if (nl == 0) return 0;
  EpochScope es(&left->allocEpoch);
  HIP_TRY(hipSetDevice(left->device));
  HIP_TRY(hipStreamSynchronize(right->stream));
  hipStream_t      s = left->stream;
  // staging of the host-pointer form lives in the left handle (grow-only): ten hipMalloc / hipFree pairs per call were most of its latency
  DevBuf<KpRec>&   dkl = left->sKl, &dkr = left->sKr;
  DevBuf<uint8_t>& ddl = left->sDl, &ddr = left->sDr;
  DevBuf<float>&   du = left->sU, &dd = left->sDepth;
  DevBuf<int>&     dsad = left->sSad, &dkept = left->sKept, &drow = left->sRow, &dridx = left->sRidx;
  const bool       byRow = left->geom.rows <= kStereoMaxRows && nr > 0;
  if (dkl.ensure(nl) || dkr.ensure(nr + 1) || ddl.ensure((size_t)nl * 32) || ddr.ensure((size_t)nr * 32 + 32) || du.ensure(nl) || dd.ensure(nl) ||
      dsad.ensure(nl) || dkept.ensure(1) || (byRow && (drow.ensure(left->geom.rows + 2) || dridx.ensure(nr + 1))))
    return -3;
  HIP_TRY(hipMemcpyAsync(dkl.p, kl, (size_t)nl * sizeof(KpRec), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ddl.p, dl, (size_t)nl * 32, hipMemcpyHostToDevice, s));
  if (nr) {
    HIP_TRY(hipMemcpyAsync(dkr.p, kr, (size_t)nr * sizeof(KpRec), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ddr.p, dr, (size_t)nr * 32, hipMemcpyHostToDevice, s));
  }
  StereoArgs a;
  memset(&a, 0, sizeof(a));
  a.kl = dkl.p, a.kr = dkr.p, a.dl = ddl.p, a.dr = ddr.p, a.nl = nl, a.nr = nr, a.nrows = left->geom.rows, a.mb = mb, a.mbf = mbf;
  a.pyrL0 = left->bufs.img0, a.pyrR0 = right->bufs.img0, a.pitch0 = left->bufs.img0Pitch;
  if (right->bufs.img0Pitch != left->bufs.img0Pitch) {
    g_err = "orbx_stereo_matches: level-0 pitches differ";
    return -2;
  }
  a.pyrL = left->bufs.pyr, a.pyrR = right->bufs.pyr;
  a.uRight = du.p, a.depth = dd.p, a.sad = dsad.p;
  if (byRow) {
    a.rowStart = drow.p, a.rowIdx = dridx.p, a.rowBand = (int)std::ceil(2.0f * left->sf[left->nlevels - 1]) + 1;
    hipLaunchKernelGGL(k_stereo_rows, dim3(1), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(k_stereo_match, dim3((nl + 3) / 4), dim3(256), 0, s, a, left->geom);
  hipLaunchKernelGGL(k_stereo_cut, dim3(1), dim3(256), 0, s, a, dkept.p);
  HIP_TRY(hipGetLastError());
  int kept = 0;
  HIP_TRY(hipMemcpyAsync(u_right, du.p, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(depth, dd.p, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&kept, dkept.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return kept;
}

// Frame::ComputeStereoMatches for every frame of the two handles' last extraction batches, device-resident (see include/orbx.h).
int orbx_stereo_matches_batch_device(orbx_t* left, orbx_t* right, const orbx_keypoint_t* d_kl, const uint8_t* d_dl, const int* d_counts_l,
                                     const orbx_keypoint_t* d_kr, const uint8_t* d_dr, const int* d_counts_r, int nframes, int cap, float mb,
                                     float mbf, float* d_u_right, float* d_depth, int* d_kept, void* stream) {
  using namespace orbx;
  if (!left || !right || !d_kl || !d_dl || !d_counts_l || !d_kr || !d_dr || !d_counts_r || nframes < 0 || cap < 1 || !d_u_right || !d_depth ||
      !d_kept || !(mb > 0.f)) {
    g_err = "orbx_stereo_matches_batch_device: bad arguments";
    return -2;
  }
  if (!left->haveGeom || !right->haveGeom || left->lastB < nframes || right->lastB < nframes || left->geom.rows != right->geom.rows ||
      left->geom.cols != right->geom.cols || left->nlevels != right->nlevels || left->device != right->device ||
      left->bufs.img0Pitch != right->bufs.img0Pitch) {
    g_err = "orbx_stereo_matches_batch_device: both extractors must hold the pyramids of a same-sized batch on the same device";
    return -2;
  }
  if (cap > 65535) {
    g_err = "orbx_stereo_matches_batch_device: more than 65535 keypoints per frame";
    return -2;
  }
  if (nframes == 0) return 0;
  EpochScope es(&left->allocEpoch);
  HIP_TRY(hipSetDevice(left->device));
  const bool byRow = left->geom.rows <= kStereoMaxRows;
  const size_t hadIdx = left->dStereoRowIdx.n;
  if (left->dStereoSad.ensure((size_t)nframes * cap) ||
      (byRow && (left->dStereoRowStart.ensure((size_t)nframes * (left->geom.rows + 1) + 1) || left->dStereoRowIdx.ensure((size_t)nframes * cap))))
    return -3;
  if (left->dStereoRowIdx.n != hadIdx) HIP_TRY(hipMemsetAsync(left->dStereoRowIdx.p, 0, left->dStereoRowIdx.n * sizeof(int), stream ? (hipStream_t)stream : left->stream));  // fresh memory holds no indices
  hipStream_t s = stream ? (hipStream_t)stream : left->stream;
  StereoArgs  a;
  memset(&a, 0, sizeof(a));
  a.kl = (const KpRec*)d_kl, a.kr = (const KpRec*)d_kr, a.dl = d_dl, a.dr = d_dr, a.nrows = left->geom.rows, a.mb = mb, a.mbf = mbf;
  a.pyrL0 = left->bufs.img0, a.pyrR0 = right->bufs.img0, a.pitch0 = left->bufs.img0Pitch;
  a.pyrL = left->bufs.pyr, a.pyrR = right->bufs.pyr, a.uRight = d_u_right, a.depth = d_depth, a.sad = left->dStereoSad.p;
  a.cap = cap, a.countsL = d_counts_l, a.countsR = d_counts_r, a.img0StrideL = left->bufs.img0FrameStride;
  a.fault = left->dFault.p;
  a.img0StrideR = right->bufs.img0FrameStride, a.pyrBytes = left->geom.pyrBytes;
  if (byRow) {
    a.rowStart = left->dStereoRowStart.p, a.rowIdx = left->dStereoRowIdx.p, a.rowBand = (int)std::ceil(2.0f * left->sf[left->nlevels - 1]) + 1;
    hipLaunchKernelGGL(k_stereo_rows, dim3(nframes), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(k_stereo_match, dim3((cap + 3) / 4, nframes), dim3(256), 0, s, a, left->geom);
  hipLaunchKernelGGL(k_stereo_cut, dim3(nframes), dim3(256), 0, s, a, d_kept);
  HIP_TRY(hipGetLastError());
  return 0;
}

// The first inconsistent count a kernel of this handle met (checked_count in orbx_kernels.h): out = {code, frame, value, where}, code
// 1 = FAST cell count above the cell's slot capacity, 2 = level keypoint count above its capacity, 3 / 4 = left / right keypoint count of
// a stereo batch outside [0, cap].  Returns 1 if one is on record (and clears it), 0 if none, < 0 on error.  A healthy run never records one.
int orbx_debug_fault(orbx_t* h, int out[4]) {
  if (!h || !out) return -2;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, h->dFault.p, 16, hipMemcpyDeviceToHost));
  if (out[0] == 0) return 0;
  HIP_TRY(hipMemset(h->dFault.p, 0, 16));
  return 1;
}

// Diagnosis hook: (address, bytes) of the handle's device buffers, in the order of the name list in tools/dbg_buffer_map.py -- what a
// "Memory access fault ... on address X" message is matched against.
int orbx_debug_buffers(orbx_t* h, unsigned long long* out, int cap) {
  if (!h || !out) return -2;
  int n = 0;
  auto put = [&](const void* p, size_t bytes) {
    if (2 * n + 1 < cap) out[2 * n] = (unsigned long long)(uintptr_t)p, out[2 * n + 1] = (unsigned long long)bytes;
    n++;
  };
#define ORBX_PUT(buf) put(h->buf.p, h->buf.n * sizeof(*h->buf.p))
  ORBX_PUT(dPyr), ORBX_PUT(dBlur), ORBX_PUT(dPre), ORBX_PUT(dStage), ORBX_PUT(dSlots), ORBX_PUT(dCand), ORBX_PUT(dNodeOf), ORBX_PUT(dCellCount), ORBX_PUT(dCellOffs);
  ORBX_PUT(dCandCount), ORBX_PUT(dLvlKp), ORBX_PUT(dLvlCount), ORBX_PUT(dOutIdx), ORBX_PUT(dNodesA), ORBX_PUT(dNodesB), ORBX_PUT(dChildCnt), ORBX_PUT(dChildIdx);
  ORBX_PUT(dD), ORBX_PUT(dDIndex), ORBX_PUT(dNCh), ORBX_PUT(dPushOff), ORBX_PUT(dMultiOff), ORBX_PUT(dKeptRank), ORBX_PUT(dVA), ORBX_PUT(dVB);
  ORBX_PUT(dStereoSad), ORBX_PUT(dStereoRowStart), ORBX_PUT(dStereoRowIdx), ORBX_PUT(dXtab), ORBX_PUT(dYtab), ORBX_PUT(dCellSlot), ORBX_PUT(dPattern);
  ORBX_PUT(dTiles[0]), ORBX_PUT(dTiles[1]), ORBX_PUT(dTileOrder[0]), ORBX_PUT(dTileOrder[1]);
#undef ORBX_PUT
  return n;
}

// PMC calibration hook (see k_calib_read): reads `nbytes` of a scratch buffer with `width` bytes per lane per load.
int orbx_debug_calib_read(orbx_t* h, long long nbytes, int width) {
  if (!h || nbytes <= 0 || (width != 1 && width != 4 && width != 16)) return -2;
  unsigned long long scratchEpoch = 0;
  EpochScope         es(&scratchEpoch);
  HIP_TRY(hipSetDevice(h->device));
  DevBuf<uint8_t>  buf;
  DevBuf<unsigned> sink;
  if (buf.ensure((size_t)nbytes) || sink.ensure(1)) return -3;
  HIP_TRY(hipMemsetAsync(buf.p, 1, (size_t)nbytes, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int grid = 2048;
  if (width == 1) hipLaunchKernelGGL(orbx::k_calib_read<1>, dim3(grid), dim3(256), 0, h->stream, buf.p, nbytes, sink.p);
  if (width == 4) hipLaunchKernelGGL(orbx::k_calib_read<4>, dim3(grid), dim3(256), 0, h->stream, buf.p, nbytes, sink.p);
  if (width == 16) hipLaunchKernelGGL(orbx::k_calib_read<16>, dim3(grid), dim3(256), 0, h->stream, buf.p, nbytes, sink.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  buf.release(), sink.release();
  return 0;
}

// Tuning hook (ORBX_DBG_OCT=1): per (frame, level) 8 values of the last batch (see OtArgs::dbgT).
int orbx_debug_octree_times(orbx_t* h, long long* out, int cap) {
  if (!h || !h->dDbgT.p) return -2;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const int n = std::min(cap, h->lastB * h->nlevels * 8);
  HIP_TRY(hipMemcpy(out, h->dDbgT.p, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return n;
}

void orbx_set_profiling(orbx_t* h, int enable) {
  if (!h) return;
  h->profiling = enable != 0;
  h->profCount = 0;
}

// Mean over the batches recorded since orbx_set_profiling(h, 1) (at most the last 32); returns how many.
int orbx_get_stage_ms(orbx_t* h, float* ms) {
  if (!h || !ms || !h->evValid || h->profCount == 0) return -2;
  HIP_TRY(hipSetDevice(h->device));
  const int nrec = std::min(h->profCount, orbx_extractor::kProfSlots);
  for (int i = 0; i < ORBX_NUM_STAGES; i++) ms[i] = 0.f;
  for (int r = 0; r < nrec; r++) {
    hipEvent_t* ev = h->ev[r];
    HIP_TRY(hipEventSynchronize(ev[ORBX_NUM_STAGES - 1]));
    float t;
    for (int i = 0; i < ORBX_NUM_STAGES - 1; i++) {
      HIP_TRY(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      ms[i] += t;
    }
    HIP_TRY(hipEventElapsedTime(&t, ev[0], ev[ORBX_NUM_STAGES - 1]));
    ms[ORBX_NUM_STAGES - 1] += t;
  }
  for (int i = 0; i < ORBX_NUM_STAGES; i++) ms[i] /= nrec;
  return nrec;
}

}  // extern "C"
