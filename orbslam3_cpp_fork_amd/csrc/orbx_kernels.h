// orbx_kernels.h -- gfx950 kernels of the ORB front-end.  Included once by orbx_extractor.hip.
//
// Stage map (reference function -> kernel):
//   ComputePyramid            src/ORBextractor.cc:1140-1173 -> k_resize_level   (one launch per level >= 1)
//   cv::FAST per cell         :800-839                      -> k_fast_tiles     (one workgroup per run of adjacent cells, orbx_fast_tiles.h)
//   DistributeOctTree         :556-753                      -> k_octree         (one workgroup per frame x level)
//   operator() output order   :1115-1134                    -> k_layout         (one workgroup per frame)
//   cv::GaussianBlur 7x7      :1103-1104                    -> k_blur_levels    (64x16 tiles, all levels)
//   IC_Angle + rBRIEF + emit  :66-139, :1120-1132           -> k_orient_describe (one wave per keypoint)
// All arithmetic is integer except fastAtan2, the tap rotation and pt *= scale, which follow the
// reference's float sequence (compile with -ffp-contract=off).
#ifndef ORBX_KERNELS_H_
#define ORBX_KERNELS_H_

#include <hip/hip_runtime.h>

#include <type_traits>

#include "octree_core.h"
#include "orb_sincosf_port.h"
#include "orbx_geom.h"
#include "undistort_core.h"

namespace orbx {

struct KpRec {  // == orbx_keypoint_t
  float x, y, size, angle, response;
  int   octave, class_id;
};

struct Bufs {
  // level 0 of every frame is the caller's image itself
  const uint8_t* img0;
  long long      img0FrameStride;
  int            img0Pitch;
  long long*     dbgT;  // optional quadtree timing stamps [frame][level][8] (ORBX_DBG_OCT=1)
  int*           fault; // [4] first inconsistency a kernel met in a count it consumes: {code, frame, value, where}; 0 = none (orbx_debug_fault)
  uint8_t*       pyr;   // [B][pyrBytes]   levels >= 1
  uint8_t*       blur;  // [B][blurBytes]  all levels
  const int2*    xtab;  // resize tables: .x = source index, .y = a0 | a1 << 16 (Q11)
  const int2*    ytab;
  const int*     cellSlot;   // [totalCells] slot base of each cell inside the frame's slot array
  uint32_t*      slots;      // [B][totalSlots] per-cell candidate slots written by FAST
  int*           cellCount;  // [B][totalCells]
  // quadtree workspace, per frame: point arrays [totalSlots], node arrays [totalNodes]
  uint32_t*           cand;
  // (used when a level's state does not fit the LDS share of its wave -- see k_octree)
  uint16_t*           nodeOf;
  int*                cellOffs;  // [B][totalCells] feed-order offset of every cell
  OtNode *            nodesA, *nodesB;
  Int4*               childCnt;
  Idx4*               childIdx;
  OtIdx *             D, *dIndex, *pushOff, *multiOff, *keptRank, *vA, *vB;
  uint8_t*            nCh;
  int*                candCount;  // [B][L]
  uint32_t*           lvlKp;      // [B][totalKp]
  int*                lvlCount;   // [B][L]
  // [B][totalKp][2] per keypoint in natural order, written by k_layout so that k_orient_describe needs no level tables:
  //   [0] = {packed (x, y, response), level | output row << 4, raw pitch, blurred pitch}
  //   [1] = {byte offset of the 31 x 31 patch origin in the raw level (from the frame's level-0 image / pyramid block),
  //          byte offset of the aligned 37 x 44 window origin in the frame's blurred block, scale (float bits), patch size}
  int4*               outIdx;
};

// A count that drives a loop over device memory is clamped to the capacity of what it indexes; the first one that had to be is put on
// record instead of being walked (a garbage count would otherwise send a kernel across the end of its buffers).
__device__ __forceinline__ int checked_count(int* fault, int code, int f, int where, int v, int cap) {
  if (v < 0 || v > cap) {
    if (fault && atomicCAS(&fault[0], 0, code) == 0) fault[1] = f, fault[2] = v, fault[3] = where;
    return v < 0 ? 0 : cap;
  }
  return v;
}
enum { kFaultCellCount = 1, kFaultLevelCount = 2, kFaultStereoLeft = 3, kFaultStereoRight = 4 };

__device__ __forceinline__ const uint8_t* level_raw(const Bufs& b, const FrameGeom& g, int level, int f, int* pitch) {
  if (level == 0) {
    *pitch = b.img0Pitch;
    return b.img0 + (long long)f * b.img0FrameStride;
  }
  *pitch = g.lv[level].pitch;
  return b.pyr + (long long)f * g.pyrBytes + g.lv[level].pyrOff;
}

// idx / d with one multiply-high, m = floor(2^32 / d) + 1: exact whenever idx * d < 2^32 (every use here: idx < 65536, d < 65536).
__device__ __forceinline__ int div_magic(unsigned idx, unsigned m) { return m ? (int)__umulhi(idx, m) : (int)idx; }  // m == 0 encodes d == 1

// ------------------------------------------------------------------------------------------------
// ComputePyramid: cv::resize(INTER_LINEAR) 8UC1 fixed point (OpenCV resize.cpp).
// Tables (source index + Q11 weights) are built on the host with the exact float/double recipe, so
// the kernel is integer-only.  Thread = 4 adjacent destination pixels, one 32-bit store.
// ------------------------------------------------------------------------------------------------
// Workgroup = 128 x 16 destination tile.  The source footprint (<= 160 x 24 px at scale 1.2) is staged in LDS
// with aligned 32-bit loads, then every thread produces 2 rows x 4 px from LDS bytes.
constexpr int kRsTileW = 128, kRsTileH = 16, kRsLdsW = 176, kRsLdsH = 25;  // one spare LDS row: see the vertical tap below

// One 128 x 16 destination tile of cv::resize(INTER_LINEAR).  S / sp / sw / sh: source image of this frame; D / dp / dw / dh:
// destination; xtab / ytab: the host-built index + Q11 weight tables of this (source, destination) size pair.
// CLAMP_ROWS: the second source row may clamp onto the first with a non-zero weight (cv::resize keeps fy at the top and bottom
// rows when it enlarges); the pyramid only shrinks (scale > 1), where row sy + 1 always exists, and skips the test.
template <bool CLAMP_ROWS>
__device__ __forceinline__ void resize_tile(const uint8_t* __restrict__ S, int sp, int sw, int sh, uint8_t* __restrict__ D, int dp, int dw, int dh,
                                            const int2* __restrict__ xtab, const int2* __restrict__ ytab, int tileX, int tileY) {
  __shared__ __align__(16) uint8_t src[kRsLdsH][kRsLdsW];
  const int        tid = threadIdx.x;
  const int        dx0 = tileX * kRsTileW, dy0 = tileY * kRsTileH;
  const int        dxl = min(dx0 + kRsTileW, dw) - 1, dyl = min(dy0 + kRsTileH, dh) - 1;
  // This thread's own table entries (4 destination columns, 2 destination rows) are requested first, together with the footprint
  // lookups below: they do not depend on the staged tile, and behind the barrier they were one more exposed round trip per workgroup.
  const int  dx  = dx0 + (tid & 31) * 4;
  const int  dyA = dy0 + (tid >> 5) * 2;
  int2       xts[4];
#pragma unroll
  for (int k = 0; k < 4; k++) xts[k] = xtab[min(dx + k, dw - 1)];
  const int2 yts[2] = {ytab[min(dyA, dh - 1)], ytab[min(dyA + 1, dh - 1)]};
  // source footprint of the tile (tables are monotone)
  const int sxa = xtab[dx0].x & ~3;
  const int sxb = min(xtab[dxl].x + 1, sw - 1);
  const int sya = min(max(ytab[dy0].x, 0), sh - 1);
  const int syb = min(max(ytab[dyl].x + 1, 0), sh - 1);
  const int nw4 = (sxb - sxa + 4) >> 2, nrow = syb - sya + 1;  // <= 44 dwords x 24 rows
  const bool al4 = ((sp & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 3) == 0);
  {  // staging: thread = (dword column tid % 64; rows tid / 64, +4, ...): no index division.  The loads are unconditional
     // (clamped to the footprint) so that all six are in flight together; only the LDS stores are predicated.
    const bool colIn = (tid & 63) < nw4;
    const int  c     = min(tid & 63, nw4 - 1) * 4;
    uint32_t   v[6];  // nrow <= 24 = 6 rounds of 4 rows
    const uint8_t* S0 = S + (long long)sya * sp + sxa;  // wave-uniform base; lanes add 32-bit offsets (no 64-bit multiply-adds per load)
    if (al4) {
#pragma unroll
      for (int it = 0; it < 6; it++) {
        const int r = min((tid >> 6) + 4 * it, nrow - 1);
        v[it]       = *reinterpret_cast<const uint32_t*>(S0 + (unsigned)(r * sp + c));  // may run past sw inside the aligned row
      }
    } else {
#pragma unroll
      for (int it = 0; it < 6; it++) {
        const int      r = min((tid >> 6) + 4 * it, nrow - 1);
        const uint8_t* p = S0 + (unsigned)(r * sp + c);
        v[it]            = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v[it] |= (uint32_t)p[min(k, sw - 1 - sxa - c)] << (8 * k);
      }
    }
#pragma unroll
    for (int it = 0; it < 6; it++) {
      const int r = (tid >> 6) + 4 * it;
      if (colIn && r < nrow) *reinterpret_cast<uint32_t*>(&src[r][c]) = v[it];
    }
  }
  __syncthreads();
  if (dx >= dw) return;
  // The four taps of a pixel sit at LDS offsets {0, 1, row, row + 1} from ONE address.  Where cv::resize clamps the second
  // column / row to the last source column / row its weight is 0 (fx = 0 / fy = 0 there), so the unclamped neighbour --
  // a stale byte of the LDS row, or the spare LDS row -- is read and multiplied by 0.
  int sx[4], a0[4], a1[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int2 xt = xts[k];
    sx[k] = xt.x - sxa;
    a0[k] = (short)(xt.y & 0xffff);
    a1[k] = xt.y >> 16;
  }
#pragma unroll
  for (int rr = 0; rr < 2; rr++) {
    const int dy = dyA + rr;
    if (dy >= dh) break;
    const int2 yt = yts[rr];
    const int  r0 = min(max(yt.x, 0), sh - 1) - sya;
    const int  r1 = CLAMP_ROWS ? min(max(yt.x + 1, 0), sh - 1) - sya : r0 + 1;
    const int  b0 = (short)(yt.y & 0xffff), b1 = yt.y >> 16;
    uint32_t   out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // Byte reads on purpose: left alone the compiler fuses columns c and c + 1 into one ds_read_u16 at an odd address
      // half of the time, and unaligned LDS reads run at a third of the speed here (measured: the level chain went from
      // 0.24 to 0.65 ms).  The row below is the same address + one LDS row (immediate offset).
      const int c = sx[k];
      int       c1 = c + 1;
      asm volatile("" : "+v"(c1));
      const int h0 = src[r0][c] * a0[k] + src[r0][c1] * a1[k];
      const int h1 = src[r1][c] * a0[k] + src[r1][c1] * a1[k];
      const int      v  = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xff) << (8 * k);
    }
    // the destination pitch is a multiple of 64: lanes past dw write into the row's padding
    *reinterpret_cast<uint32_t*>(D + (unsigned)(dy * dp + dx)) = out;
  }
}

// Frame -> XCD affinity: workgroups are dealt round-robin over the 8 XCDs in linearised block order, and gridDim.x is a multiple of
// 8, so blockIdx.x % 8 names the XCD.  grid = (8 x tiles across, tiles down, frames / 8): every tile of frame (x % 8) + 8 z runs on one
// XCD and each source line of level l - 1 is pulled into ONE L2 -- with a plain (x, y, frame) grid neighbouring tiles, which share
// source lines, went to different XCDs and the level was fetched almost twice (626 MB measured against 321 MB algorithmic per 256
// VGA frames).  No index division anywhere.
__global__ __launch_bounds__(256) void k_resize_level(Bufs b, FrameGeom g, int level, int nframes) {
  const int f = (blockIdx.x & 7) + 8 * blockIdx.z;
  if (f >= nframes) return;
  const LevelGeom& L  = g.lv[level];
  const LevelGeom& Ls = g.lv[level - 1];
  int              sp;
  const uint8_t*   S = level_raw(b, g, level - 1, f, &sp);
  resize_tile<false>(S, sp, Ls.w, Ls.h, b.pyr + (long long)f * g.pyrBytes + L.pyrOff, L.pitch, L.w, L.h, b.xtab + L.xtabOff, b.ytab + L.ytabOff,
                     blockIdx.x >> 3, blockIdx.y);
}

// A level narrower or lower than 8 px (deep pyramids of small or very elongated images: 1077 x 137 at scale 1.5 has a 42 x 5 level 8).
// It is far too small for a FAST cell, so nothing downstream reads it except the next level of the chain and the public pyramid
// accessor; the tiled kernels (4-px groups, single reflection at the borders) do not take it.  One thread per pixel, the same tables
// and the same arithmetic as resize_tile; such a level is not blurred (no keypoint can lie on it).
__host__ __device__ inline bool level_is_tiny(const LevelGeom& L) { return L.w < 8 || L.h < 8; }

__global__ __launch_bounds__(256) void k_resize_tiny(Bufs b, FrameGeom g, int level) {
  const int        f  = blockIdx.x;
  const LevelGeom& L  = g.lv[level];
  const LevelGeom& Ls = g.lv[level - 1];
  int              sp;
  const uint8_t*   S = level_raw(b, g, level - 1, f, &sp);
  uint8_t*         D = b.pyr + (long long)f * g.pyrBytes + L.pyrOff;
  const int2 *     xtab = b.xtab + L.xtabOff, *ytab = b.ytab + L.ytabOff;
  for (int i = threadIdx.x; i < L.w * L.h; i += 256) {
    const int  dy = i / L.w, dx = i - dy * L.w;
    const int2 xt = xtab[dx], yt = ytab[dy];
    const int  c = xt.x, c1 = min(c + 1, Ls.w - 1), a0 = (short)(xt.y & 0xffff), a1 = xt.y >> 16;
    const int  r0 = min(max(yt.x, 0), Ls.h - 1), r1 = min(max(yt.x + 1, 0), Ls.h - 1), b0 = (short)(yt.y & 0xffff), b1 = yt.y >> 16;
    const int  h0 = S[(long long)r0 * sp + c] * a0 + S[(long long)r0 * sp + c1] * a1;
    const int  h1 = S[(long long)r1 * sp + c] * a0 + S[(long long)r1 * sp + c1] * a1;
    D[dy * L.pitch + dx] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
  }
}

// ------------------------------------------------------------------------------------------------
// ComputePyramid, several levels in ONE launch -- the latency form of the chain (src/ORBextractor.cc:1140-1173: each level is resized from
// the previous one, so the chain itself cannot go, but for a handful of frames it need not be a chain of seven launches, each of which
// pays a dispatch gap for a few microseconds of work).  A workgroup owns a BAND of rows of level endLevel - 1 and walks levels
// firstLevel .. endLevel - 1 inside LDS: it stages the rows of level firstLevel - 1 its band descends from, together with the x tables of
// the levels and its own slices of the y tables (ONE exposure to global-memory latency), produces its rows of every level from the previous
// level's rows in LDS, and writes the rows it OWNS of every level to the pyramid block.  Rows a neighbouring band needs as well (1-2 per
// level boundary, accumulating downwards) are computed by both and written by one, so bands never wait for each other.  Row ranges are
// host-built (PyrBand) from the same tables the per-level kernel reads; the arithmetic is resize_tile's.
// Batches keep one launch per level: there every launch fills the chip, the step is bound by instruction issue, and the rows computed
// twice would cost more than the six dispatch gaps saved (measured: 243.6 -> 243.3 k frames/s with levels 4-7 fused).
// ------------------------------------------------------------------------------------------------
struct PyrBand {  // rows [lo, lo + n) of level l are computed, [ownLo, ownLo + ownN) written; entry firstLevel - 1 = the staged source rows
  short lo[kMaxLevels], n[kMaxLevels], ownLo[kMaxLevels], ownN[kMaxLevels];
};

struct PyrPlan {  // by value to the kernel
  int      firstLevel, endLevel, nbands;
  int      ldsHalf;               // bytes of each of the two LDS row buffers
  int      xtLds, ytLds;          // byte offsets of the staged x / y tables behind them
  int      xtOff[kMaxLevels];     // first entry of level l's x table in the staged copy (4 * ceil(w / 4) entries, the last one repeated)
  unsigned quadMagic[kMaxLevels]; // div_magic divider of ceil(w_l / 4)
};

constexpr int kPyrBandThreads = 1024;

template <bool AL4>
__global__ __launch_bounds__(kPyrBandThreads) void k_pyramid_bands(Bufs b, FrameGeom g, const PyrBand* __restrict__ bands, PyrPlan plan, int nframes) {
  extern __shared__ __align__(16) uint8_t pyrLds[];
  constexpr int NT = kPyrBandThreads;
  const int     f  = blockIdx.y;
  const PyrBand& bd  = bands[blockIdx.x];
  const int      tid = threadIdx.x;
  uint8_t*       buf[2] = {pyrLds, pyrLds + plan.ldsHalf};
  int2*          xtL = reinterpret_cast<int2*>(pyrLds + plan.xtLds);
  int2*          ytL = reinterpret_cast<int2*>(pyrLds + plan.ytLds);
  // ---- stage: source rows (level firstLevel - 1) with aligned dwords, the x tables, this band's y-table slices
  {
    const int        sl = plan.firstLevel - 1;
    const LevelGeom& Ls = g.lv[sl];
    int              sp;
    const uint8_t*   S   = level_raw(b, g, sl, f, &sp) + (long long)bd.lo[sl] * sp;
    const int        P   = (Ls.w + 3) & ~3, nw4 = P >> 2, items = bd.n[sl] * nw4;
    const unsigned   mq  = plan.quadMagic[sl];
    for (int it = tid; it < items; it += NT) {
      const int r = div_magic(it, mq), c = (it - r * nw4) * 4;
      uint32_t  v;
      if (AL4 || sl > 0) {
        v = *reinterpret_cast<const uint32_t*>(S + (unsigned)(r * sp + c));  // may run past w inside the aligned row
      } else {
        const uint8_t* p = S + (unsigned)(r * sp);
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)p[min(c + k, Ls.w - 1)] << (8 * k);
      }
      *reinterpret_cast<uint32_t*>(buf[0] + r * P + c) = v;
    }
    int yo = 0;
    for (int l = plan.firstLevel; l < plan.endLevel; l++) {
      const LevelGeom& L  = g.lv[l];
      const int        n4 = ((L.w + 3) >> 2) * 4;
      for (int i = tid; i < n4; i += NT) xtL[plan.xtOff[l] + i] = b.xtab[L.xtabOff + min(i, L.w - 1)];
      for (int i = tid; i < bd.n[l]; i += NT) ytL[yo + i] = b.ytab[L.ytabOff + bd.lo[l] + i];
      yo += bd.n[l];
    }
  }
  __syncthreads();
  uint8_t* pyrF = b.pyr + (long long)f * g.pyrBytes;
  int      yo   = 0;
  for (int l = plan.firstLevel; l < plan.endLevel; l++) {
    const LevelGeom& L    = g.lv[l];
    const uint8_t*   S    = buf[(l - plan.firstLevel) & 1];
    uint8_t*         D    = buf[(l - plan.firstLevel + 1) & 1];
    const int        SP   = (g.lv[l - 1].w + 3) & ~3, DP = (L.w + 3) & ~3, nx = DP >> 2;
    const int        sLo  = bd.lo[l - 1], dLo = bd.lo[l], items = bd.n[l] * nx;
    const int        oLo  = bd.ownLo[l], oHi = oLo + bd.ownN[l];
    const int2*      xt4  = xtL + plan.xtOff[l];
    const int2*      ytB  = ytL + yo;
    const unsigned   mq   = plan.quadMagic[l];
    const bool       last = l == plan.endLevel - 1;
    for (int it = tid; it < items; it += NT) {
      const int  r = div_magic(it, mq), dx = (it - r * nx) * 4, dy = dLo + r;
      const int2 yt = ytB[r];
      const int  r0 = yt.x - sLo;  // the pyramid only shrinks: source row yt.x + 1 exists (see resize_tile) and lies in the staged band
      const int  b0 = (short)(yt.y & 0xffff), b1 = yt.y >> 16;
      const uint8_t* s0 = S + r0 * SP;
      uint32_t       out = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int2 xt = xt4[dx + k];
        const int  a0 = (short)(xt.y & 0xffff), a1 = xt.y >> 16;
        int        c = xt.x, c1 = c + 1;
        asm volatile("" : "+v"(c1));  // byte reads on purpose: a fused ds_read_u16 at an odd address runs at a sixteenth of the speed
        const int h0 = s0[c] * a0 + s0[c1] * a1;
        const int h1 = s0[SP + c] * a0 + s0[SP + c1] * a1;
        const int v  = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 0xff) << (8 * k);
      }
      if (!last) *reinterpret_cast<uint32_t*>(D + r * DP + dx) = out;
      // the level's pitch is a multiple of 64: lanes past w write into the row's padding
      if (dy >= oLo && dy < oHi) *reinterpret_cast<uint32_t*>(pyrF + L.pyrOff + (unsigned)(dy * L.pitch + dx)) = out;
    }
    yo += bd.n[l];
    __syncthreads();
  }
}

// Rows that are not 4-byte aligned (odd widths with a tight step: KITTI's 1241) are copied once into a buffer with a 64-byte pitch, so
// that every later pass reads aligned dwords.  One thread = 4 destination bytes of one row (an unaligned 32-bit load, an aligned store);
// the columns past the image width are written too (they read the next bytes of the source, never past the frame's end + 3: the host
// sizes its buffers with that slack, and the values are dead).
__global__ __launch_bounds__(256) void k_repitch(const uint8_t* __restrict__ src, long long srcStep, long long srcFrameStride, uint8_t* __restrict__ dst, int dstPitch,
                                                 long long dstFrameStride, int cols, int rows) {
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
  if (x4 >= cols) return;
  const uint8_t* p = src + (long long)f * srcFrameStride + (long long)y * srcStep + x4;
  uint32_t       v;
  if (x4 + 4 <= cols) {
    __builtin_memcpy(&v, p, 4);
  } else {  // the row's last, partial dword: byte loads, nothing is read past the row
    v = 0;
    for (int k = 0; x4 + k < cols; k++) v |= (uint32_t)p[k] << (8 * k);
  }
  *reinterpret_cast<uint32_t*>(dst + (long long)f * dstFrameStride + (long long)y * dstPitch + x4) = v;
}

// cv::resize(im, im, cv::Size(newWidth, newHeight)) of System::TrackMonocular / TrackStereo / TrackRGBD (src/System.cc:472-478 and
// siblings) before the extractor sees the image: same arithmetic, any size pair.
__global__ __launch_bounds__(256) void k_resize_input(const uint8_t* __restrict__ src, int sp, long long srcFrameStride, int sw, int sh,
                                                      uint8_t* __restrict__ dst, int dp, long long dstFrameStride, int dw, int dh,
                                                      const int2* __restrict__ xtab, const int2* __restrict__ ytab) {
  resize_tile<true>(src + (long long)blockIdx.z * srcFrameStride, sp, sw, sh, dst + (long long)blockIdx.z * dstFrameStride, dp, dw, dh, xtab, ytab,
                    blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// cv::FAST(cell, kps, iniThFAST, true) with the minThFAST retry, per cell (src/ORBextractor.cc:777-849).
//
// The corner score of OpenCV (cornerScore<16>) is max(A, B) - 1 with
//   A = max over the 16 arcs of 9 contiguous circle pixels of min(v - p), B likewise for (p - v),
// and a pixel is a corner at threshold t iff max(A, B) > t, i.e. score >= t.  So one score map serves
// both thresholds: pass(t) keeps s_t = (score >= t ? score : 0) and applies the strict 3x3 NMS on
// s_t, with neighbours outside the cell's 3-px-inset interior counting as 0 (they are never scored
// by cv::FAST on the sub-image).  Output order is row-major, as cv::FAST emits.
// ------------------------------------------------------------------------------------------------
// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so
// giving XCD x the contiguous range [x * per, (x + 1) * per) of work items keeps spatial neighbours -- which share
// cache lines and halos -- behind ONE L2 instead of eight.  The grid is padded to 8 * per; returns -1 for padding.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_item(int block, int nitems) {
  const int per  = (nitems + 7) >> 3;
  const int item = (block & 7) * per + (block >> 3);
  return ((block >> 3) < per && item < nitems) ? item : -1;
}

// ---- packed 16-bit lanes: two pixels per VGPR (v_pk_min_i16 / v_pk_max_i16 / v_pk_sub_i16) ----------------------
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16     pk_from(uint32_t u) { return __builtin_bit_cast(pk16, u); }
__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16     pk_min(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk16     pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk16     pk_pair(int lo, int hi) { return pk_from((uint32_t)lo | ((uint32_t)hi << 16)); }  // 0 <= lo, hi < 65536

// Corner scores of TWO pixels at once (low half: pixel A at cA, high half: pixel B at cB); same arithmetic as
// fast_score, every min/max done on both 16-bit halves by one packed instruction.
__device__ __forceinline__ void fast_score2(const uint8_t* __restrict__ cA, const uint8_t* __restrict__ cB, int ts, int* sA, int* sB) {
  const int off[16] = {3 * ts,      3 * ts + 1,  2 * ts + 2,  ts + 3, 3, -ts + 3, -2 * ts + 2, -3 * ts + 1,
                       -3 * ts,     -3 * ts - 1, -2 * ts - 2, -ts - 3, -3, ts - 3,  2 * ts - 2,  3 * ts - 1};
  const pk16 v = pk_pair(cA[0], cB[0]);
  pk16       d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = v - pk_pair(cA[off[k]], cB[off[k]]);
  pk16 mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    mn2[i] = pk_min(d[i], d[(i + 1) & 15]);
    mx2[i] = pk_max(d[i], d[(i + 1) & 15]);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    mn4[i] = pk_min(mn2[i], mn2[(i + 2) & 15]);
    mx4[i] = pk_max(mx2[i], mx2[(i + 2) & 15]);
  }
  pk16 A = pk_pair(0xfc18, 0xfc18), Bm = pk_pair(1000, 1000);  // -1000 / +1000
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const pk16 m9 = pk_min(pk_min(mn4[i], mn4[(i + 4) & 15]), d[(i + 8) & 15]);
    const pk16 M9 = pk_max(pk_max(mx4[i], mx4[(i + 4) & 15]), d[(i + 8) & 15]);
    A             = pk_max(A, m9);
    Bm            = pk_min(Bm, M9);
  }
  const pk16 sc = pk_max(A, -Bm) - pk16{1, 1};
  *sA           = sc.x;
  *sB           = sc.y;
}

}  // namespace orbx
#include "orbx_fast_tiles.h"
namespace orbx {

// ------------------------------------------------------------------------------------------------
// DistributeOctTree: one workgroup per (frame, level).  First gathers the per-cell candidate slots into
// the feed order of the reference (cells row-major, row-major inside a cell), then runs octree_core.h.
// ------------------------------------------------------------------------------------------------
constexpr int kOctThreads   = 64;   // throughput form: one wave per (frame, level), barriers degenerate to wave syncs
constexpr int kOctThreadsLatency = 256;  // latency form (a handful of frames): a workgroup per (frame, level) shortens the passes over the points
constexpr int kOctNodeBytes = 64;   // LDS bytes per node over all node-level arrays (see ot_carve_lds)
constexpr int kOctFixedInts = 80 + 192;  // per-wave totals of the multi-wave context + the sort work list
__device__ __forceinline__ void octree_args(const Bufs& b, const FrameGeom& g, int f, int level, OtArgs* a) {
  const LevelGeom& L  = g.lv[level];
  const long long  po = (long long)f * g.totalSlots + L.slotBase;
  const long long  no = (long long)f * g.totalNodes + L.nodeBase;
  a->cand    = b.cand + po;
  a->W       = L.maxBX - kMinBorder;
  a->H       = L.maxBY - kMinBorder;
  a->N       = L.N;
  a->nIni    = L.nIni;
  a->hX      = L.hX;
  a->maxNodes = L.nodeCap;
  a->nodeOf = b.nodeOf + po;
  a->nodesA = b.nodesA + no, a->nodesB = b.nodesB + no, a->childCnt = b.childCnt + no, a->childIdx = b.childIdx + no;
  a->best    = reinterpret_cast<unsigned long long*>(a->childCnt);
  a->sortbuf = reinterpret_cast<unsigned long long*>(a->childIdx);
  a->D = b.D + no, a->dIndex = b.dIndex + no, a->nCh = b.nCh + no, a->pushOff = b.pushOff + no, a->multiOff = b.multiOff + no;
  a->keptRank = b.keptRank + no, a->vA = b.vA + no, a->vB = b.vB + no;
  a->outKp    = b.lvlKp + (long long)f * g.totalKp + L.kpBase;
  a->outCount = b.lvlCount + (long long)f * g.nlevels + level;
  a->dbgT     = b.dbgT ? b.dbgT + ((long long)f * g.nlevels + level) * 8 : nullptr;
}

// Carve the node-level arrays (and, if they fit, the point labels) out of `lds`.  Layout per node: 2 x OtNode (24),
// childCnt/best (16), childIdx/sortbuf (8), seven 16-bit index arrays (14), nCh (1, padded to 2) = 64 bytes.
__device__ __forceinline__ void ot_carve_lds(OtArgs* a, unsigned char* lds, int nodeCap, bool withLabels) {
  unsigned char* p = lds;
  a->childCnt = reinterpret_cast<Int4*>(p), a->best = reinterpret_cast<unsigned long long*>(p), p += 16 * nodeCap;
  a->childIdx = reinterpret_cast<Idx4*>(p), a->sortbuf = reinterpret_cast<unsigned long long*>(p), p += 8 * nodeCap;
  a->nodesA = reinterpret_cast<OtNode*>(p), p += 12 * nodeCap;
  a->nodesB = reinterpret_cast<OtNode*>(p), p += 12 * nodeCap;
  OtIdx* q = reinterpret_cast<OtIdx*>(p);
  a->D = q, q += nodeCap;
  a->dIndex = q, q += nodeCap;
  a->keptRank = q, q += nodeCap;
  a->pushOff = q, q += nodeCap;
  a->multiOff = q, q += nodeCap;
  a->vA = q, q += nodeCap;
  a->vB = q, q += nodeCap;
  a->nCh = reinterpret_cast<uint8_t*>(q), q += nodeCap;
  if (withLabels) a->nodeOf = q;
}

// ldsNodeCap: nodes the dynamic LDS block was sized for (0 = everything stays in global memory); ldsPtsCap: point labels.
// CTX = WaveCtx (64 threads) or DevCtx (NT threads): the same algorithm, its passes spread over one or NT / 64 waves.
template <class CTX, int NT>
__global__ __launch_bounds__(NT) void k_octree(Bufs b, FrameGeom g, int ldsNodeCap, int ldsPtsCap) {
  extern __shared__ __align__(16) unsigned char ldsDyn[];
  int*             scratch = reinterpret_cast<int*>(ldsDyn);
  const int        level = blockIdx.y, f = blockIdx.x;
  const LevelGeom& L = g.lv[level];
  CTX              c;
  c.tid = threadIdx.x, c.nt = NT, c.lds = scratch;
  OtArgs a;
  octree_args(b, g, f, level, &a);
  // feed order: exclusive scan of the cell counts, then copy the slots
  const int  nc     = L.nCols * L.nRows;
  if (nc == 0) {  // a level without a FAST cell (makeGeometry): no candidates, no keypoints
    if (c.tid == 0) b.candCount[(long long)f * g.nlevels + level] = 0, *a.outCount = 0;
    return;
  }
  const int* counts = b.cellCount + (long long)f * g.totalCells + L.cellBase;
  int*       offs   = b.cellOffs + (long long)f * g.totalCells + L.cellBase;
  int        lo, hi;
  ot_chunk(c, nc, &lo, &hi);
  const int* cslot0 = b.cellSlot + L.cellBase;
  auto cellCount = [&](int i) { return checked_count(b.fault, kFaultCellCount, f, level * 65536 + i, counts[i], (i + 1 < nc ? cslot0[i + 1] : L.slotCount) - cslot0[i]); };
  Int4 loc = {{0, 0, 0, 0}};
  for (int i = lo; i < hi; i++) loc.v[0] += cellCount(i);
  Int4 tot;
  Int4 ex  = c.scan4(loc, &tot);
  int  run = ex.v[0];
  for (int i = lo; i < hi; i++) {
    offs[i] = run;
    run += cellCount(i);
  }
  c.sync();
  const int                    n     = tot.v[0];
  const uint32_t* __restrict__ slots = b.slots + (long long)f * g.totalSlots + L.slotBase;
  const int*                   cslot = b.cellSlot + L.cellBase;
  uint32_t* __restrict__       cand  = b.cand + (long long)f * g.totalSlots + L.slotBase;
  for (int cidx = c.tid; cidx < nc; cidx += c.nt) {
    const int cnt = cellCount(cidx), o = offs[cidx], sb = cslot[cidx];
#pragma unroll 4
    for (int k = 0; k < cnt; k++) cand[o + k] = slots[sb + k];
  }
  if (c.tid == 0) b.candCount[(long long)f * g.nlevels + level] = n;
  c.sync();
  a.n = n;
  // Three inlined copies of the core, one per placement of its state: with the pointers' provenance fixed in each copy the
  // compiler addresses LDS with ds_* instructions (a runtime choice between LDS and global pointers leaves it with flat_*
  // accesses, which pay the flat-address path on every node-array access of this latency-bound kernel).
  if (L.nodeCap <= ldsNodeCap) {
    if (n <= ldsPtsCap) {
      ot_carve_lds(&a, ldsDyn + kOctFixedInts * 4, L.nodeCap, true);
      ot_distribute(c, a);
    } else {
      ot_carve_lds(&a, ldsDyn + kOctFixedInts * 4, L.nodeCap, false);
      ot_distribute(c, a);
    }
  } else {
    ot_distribute(c, a);
  }
}

// stand-alone quadtree on caller-supplied candidates (parity hook)
__global__ __launch_bounds__(kOctThreads) void k_octree_only(OtArgs a) {
  __shared__ int scratch[kOctFixedInts];
  WaveCtx        c;
  c.tid = threadIdx.x, c.nt = 64, c.lds = scratch;
  ot_distribute(c, a);
}

// the std::sort re-enactment of one wave on caller-supplied records (parity hook)
__global__ __launch_bounds__(64) void k_wave_sort_only(unsigned long long* buf, int n, int depth_limit) {
  __shared__ int stk[192];
  ot_wave_sort(buf, n, stk, depth_limit);
}

// ------------------------------------------------------------------------------------------------
// Output layout of operator() (src/ORBextractor.cc:1090-1134): natural order is level-major / list
// order; keypoints whose scaled x lies in [lap0, lap1] fill the output from the back, the others from
// the front.  One workgroup per frame computes every keypoint's output row.
// ------------------------------------------------------------------------------------------------
// Keypoint i (natural order) is handled by thread i % 256 in round i / 256: every load of a round is independent of every other, and
// the rank of a "stereo" keypoint (the number of them before it) is a ballot inside the wave plus a prefix over (round, wave) counts
// held in LDS -- two short rounds over the keypoints instead of two dependent walks of a per-thread chunk.
constexpr int kLayoutMaxRounds = 128;  // x 256 keypoints per frame (host-checked against totalKp)

__global__ __launch_bounds__(256) void k_layout(Bufs b, FrameGeom g, int lap0, int lap1, int* counts, int cap) {
  __shared__ int sLbase[kMaxLevels + 1];
  __shared__ int sCnt[kLayoutMaxRounds * 4 + 1];  // stereo keypoints per (round, wave), then their exclusive prefix
  const int      f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int*     lc = b.lvlCount + (long long)f * g.nlevels;
  if (wave == 0) {  // level bases: one lane per level loads its count (one memory round trip, not one per level), prefix by shuffles
    const int c = lane < g.nlevels ? checked_count(b.fault, kFaultLevelCount, f, lane, lc[lane], g.lv[lane].kpCap) : 0;
    int       inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane < g.nlevels) sLbase[lane] = inc - c;
    if (lane == g.nlevels - 1) sLbase[g.nlevels] = inc;
  }
  __syncthreads();
  const int       n      = sLbase[g.nlevels];
  const int       rounds = (n + 255) >> 8;
  const uint32_t* kp     = b.lvlKp + (long long)f * g.totalKp;
  int4*           oix    = b.outIdx + (long long)f * g.totalKp * 2;
  uint32_t        rOut = 0;
  int             lOut = 0;
  auto pred = [&](int i) -> bool {
    int l = 0;
    while (l + 1 < g.nlevels && i >= sLbase[l + 1]) l++;
    const uint32_t r  = kp[g.lv[l].kpBase + (i - sLbase[l])];
    float          xs = (float)(ot_cx(r) + kMinBorder);
    if (l != 0) xs = xs * g.lv[l].scale;
    rOut = r, lOut = l;
    return xs >= (float)lap0 && xs <= (float)lap1;
  };
  // The first kKeep rounds (2048 keypoints: every frame of the usual sizes) keep their record, level and flag in registers: their loads
  // are all in flight at once, and the second pass does not load again.
  constexpr int kKeep = 8;
  uint32_t      rk[kKeep];
  int           lk[kKeep];
  unsigned      pk = 0;
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    const int i = min(j * 256 + tid, max(n - 1, 0));
    int       l = 0;
    while (l + 1 < g.nlevels && i >= sLbase[l + 1]) l++;
    lk[j] = l;
    rk[j] = n > 0 ? kp[g.lv[l].kpBase + (i - sLbase[l])] : 0u;
  }
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    if (j < rounds) {
      float xs = (float)(ot_cx(rk[j]) + kMinBorder);
      if (lk[j] != 0) xs = xs * g.lv[lk[j]].scale;
      const bool               p  = j * 256 + tid < n && xs >= (float)lap0 && xs <= (float)lap1;
      const unsigned long long bm = __ballot(p);
      pk |= (p ? 1u : 0u) << j;
      if (lane == 0) sCnt[j * 4 + wave] = __popcll(bm);
    }
  }
  for (int j = kKeep; j < rounds; j++) {
    const int                i  = j * 256 + tid;
    const unsigned long long bm = __ballot(i < n && pred(i));
    if (lane == 0) sCnt[j * 4 + wave] = __popcll(bm);
  }
  __syncthreads();
  if (wave == 0) {  // exclusive prefix over the rounds * 4 counts (<= 512 of them: eight per lane)
    constexpr int kPer = kLayoutMaxRounds * 4 / 64;
    const int     m = rounds * 4;
    int           v[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      v[k] = lane * kPer + k < m ? sCnt[lane * kPer + k] : 0;
      sum += v[k];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    int run = inc - sum;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      if (lane * kPer + k < m) sCnt[lane * kPer + k] = run;
      run += v[k];
    }
    if (lane == 63) sCnt[kLayoutMaxRounds * 4] = inc;  // total
  }
  __syncthreads();
  auto emit = [&](int j, int i, bool p) {
    const unsigned long long bm = __ballot(p);
    if (i < n) {
      const int stereo = sCnt[j * 4 + wave] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
      const int row    = p ? (n - 1 - stereo) : (i - stereo);
      // everything k_orient_describe needs about keypoint i
      const LevelGeom& L  = g.lv[lOut];
      const int        px = ot_cx(rOut) + kMinBorder, py = ot_cy(rOut) + kMinBorder;
      const int        rawPitch = lOut == 0 ? b.img0Pitch : L.pitch;
      const int        rawOrg   = (lOut == 0 ? 0 : L.pyrOff) + (py - kHalfPatch) * rawPitch + (px - kHalfPatch);
      const int        blurOrg  = L.blurOff + (py - 18) * L.blurPitch + ((px - 18) & ~3);
      oix[2 * i]     = make_int4((int)rOut, lOut | (row << 4), rawPitch, L.blurPitch);
      oix[2 * i + 1] = make_int4(rawOrg, blurOrg, __float_as_int(L.scale), L.patchSize);
    }
  };
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    if (j < rounds) {
      rOut = rk[j], lOut = lk[j];
      emit(j, j * 256 + tid, (pk >> j) & 1u);
    }
  }
  for (int j = kKeep; j < rounds; j++) {
    const int i = j * 256 + tid;
    emit(j, i, i < n && pred(i));
  }
  if (tid == 0) {
    counts[2 * f]     = n;
    counts[2 * f + 1] = n - sCnt[kLayoutMaxRounds * 4];  // monoIndex
  }
}

// ------------------------------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) fixed point: Q8 taps
// {18,34,48,56,48,34,18}; horizontal pass exact in u16, vertical (sum + 32768) >> 16.
// Tile = 64 x 16 outputs, staged with a 3-px reflected halo in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// Register sliding-window form: a thread owns 4 adjacent columns of a kBlurRows-row strip.  It streams the
// strip's kBlurRows + 6 input rows once (three aligned 32-bit loads per row: columns x0-4 .. x0+7), forms the
// horizontal Q8 sums of its 4 columns, keeps the last 7 of them in registers and emits one output row
// (4 bytes, one store) per input row.  No LDS, no second pass; neighbouring lanes' overlapping loads hit L1.
// Work items are dealt level by level, 256 per workgroup (LevelGeom::blurTileBase counts workgroups).
constexpr int kBlurRows = 16;  // rows per strip (32-row strips save the re-filtered halo rows but leave too few waves to hide the row loads: 0.19 -> 0.22 ms)
#ifndef ORBX_BLUR_PREFETCH
#define ORBX_BLUR_PREFETCH 3
#endif
constexpr int kBlurPrefetch = ORBX_BLUR_PREFETCH;  // rows of loads in flight ahead of the arithmetic
constexpr int kBlurEdgeRows = 4;  // strip height of the edge items: few items, so short strips keep enough of them in flight

// Three dwords covering columns x0-4 .. x0+7 of one row (reflect-101 at the image edges).
__device__ __forceinline__ int reflect101_once(int p, int n) {  // valid for -n < p < 2n - 1
  p = p < 0 ? -p : p;
  return p >= n ? 2 * n - 2 - p : p;
}

template <bool FAST>
__device__ __forceinline__ void blur_load12(const uint8_t* __restrict__ row, int x0, int w, uint32_t& a, uint32_t& m, uint32_t& c) {
  if (FAST) {
    a = *reinterpret_cast<const uint32_t*>(row + x0 - 4);
    m = *reinterpret_cast<const uint32_t*>(row + x0);
    c = *reinterpret_cast<const uint32_t*>(row + x0 + 4);
  } else {
    a = m = c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a |= (uint32_t)row[reflect101_once(x0 - 4 + i, w)] << (8 * i);
      m |= (uint32_t)row[reflect101_once(x0 + i, w)] << (8 * i);
      c |= (uint32_t)row[reflect101_once(x0 + 4 + i, w)] << (8 * i);
    }
  }
}

// One 4-column x kBlurRows strip.  The 7-row window of horizontal sums is a ring indexed with compile-time
// constants (the row loop is fully unrolled), so nothing is shifted.
template <bool FAST, int ROWS>
__device__ __forceinline__ void blur_strip(const uint8_t* __restrict__ src, int pitch, uint8_t* __restrict__ dst, int dpitch, int w, int h, int x0,
                                           int y0) {
  // Q8 taps {18,34,48,56,48,34,18} as two byte vectors for v_dot4_u32_u8: window bytes p[i..i+3] and p[i+4..i+6]
  const uint32_t T0 = 18u | (34u << 8) | (48u << 16) | (56u << 24), T1 = 48u | (34u << 8) | (18u << 16);
  // Vertical pass with v_dot2_u32_u16: the horizontal sums (<= 255 * 256, 16 bits) of rows r-1 and r are kept PACKED in
  // one register, pr[r] = h[r-1] | h[r] << 16, so the 7-tap column sum of output row r-3 is three dot2 + one mad:
  //   32768 + (18,34).pr[r-5] + (48,56).pr[r-3] + (48,34).pr[r-1] + 18 h[r]      (< 2^24: byte 2 is the rounded result)
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 W01 = {18, 34}, W23 = {48, 56}, W45 = {48, 34};
  uint32_t  pr[6][4];  // ring of packed row pairs, indexed with compile-time constants (the row loop is fully unrolled)
  uint32_t  hp[4] = {0, 0, 0, 0};
  // Row loads run kBlurPrefetch rows ahead of the arithmetic (a ring of register slots, compile-time indices): the store of an output
  // row sits in its own `oy < h` block, and a load placed behind that block cannot be moved above it by the compiler -- every row then
  // paid one full memory round trip (load, s_waitcnt vmcnt(0), arithmetic, store).
  constexpr int NS = kBlurPrefetch + 1;
  uint32_t      la[NS], lm[NS], lc[NS];
  auto          issue = [&](int r, int slot) {
    int sy = y0 + r - 3;  // reflect-101 rows; beyond the image bottom the strip's extra rows are never used
    sy     = sy < 0 ? -sy : sy;
    sy     = sy >= h ? max(2 * h - 2 - sy, 0) : sy;
    blur_load12<FAST>(src + (long long)sy * pitch, x0, w, la[slot], lm[slot], lc[slot]);
  };
#pragma unroll
  for (int r = 0; r < kBlurPrefetch; r++) issue(r, r % NS);
#pragma unroll
  for (int r = 0; r < ROWS + 6; r++) {
    if (r + kBlurPrefetch < ROWS + 6) issue(r + kBlurPrefetch, (r + kBlurPrefetch) % NS);
    const uint32_t a = la[r % NS], m = lm[r % NS], c = lc[r % NS];
    // horizontal pass: output column x0+i sees bytes (x0+i-3 .. x0+i+3); alignbyte slides the 12-byte window
    uint32_t hn[4];
    hn[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, m, 1), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, a, 1), T0, 0u, false), false);
    hn[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, m, 2), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, a, 2), T0, 0u, false), false);
    hn[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, m, 3), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, a, 3), T0, 0u, false), false);
    hn[3] = __builtin_amdgcn_udot4(c, T1, __builtin_amdgcn_udot4(m, T0, 0u, false), false);
    if (r >= 6) {
      const int oy = y0 + r - 6;
      if (oy < h) {
        const uint32_t *p5 = pr[(r - 5) % 6], *p3 = pr[(r - 3) % 6], *p1 = pr[(r - 1) % 6];
        uint32_t        sv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, p5[i]), W01, 32768u, false);
          acc          = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, p3[i]), W23, acc, false);
          acc          = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, p1[i]), W45, acc, false);
          sv[i]        = acc + 18u * hn[i];
        }
        // byte 2 of the four sums -> one dword
        const uint32_t out = __builtin_amdgcn_perm(sv[1], sv[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sv[3], sv[2], 0x06020c0cu);
        *reinterpret_cast<uint32_t*>(dst + (long long)oy * dpitch + x0) = out;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      pr[r % 6][i] = hp[i] | (hn[i] << 16);
      hp[i]        = hn[i];
    }
  }
}

// Work items of a level = (row strip, 4-column group).  Groups whose 12-byte window lies inside the row take the
// aligned-dword path (k_blur_levels: 56 registers, 8 waves per SIMD); the 2-3 edge groups of every strip need reflected
// byte loads and live in a kernel of their own (k_blur_edges) so that their register appetite does not set the occupancy of
// the other 98 %.  A level whose source is not 4-byte aligned (only the caller's level 0 can be) goes to the edge kernel whole.
__device__ __forceinline__ void blur_levels_block(const Bufs& b, const FrameGeom& g, int l0Aligned, int block, int f) {
  const int tid = threadIdx.x;
  int       t = xcd_item(block, g.totalBlurTiles), level = 0;
  if (t < 0) return;
  while (level + 1 < g.nlevels && t >= g.lv[level + 1].blurTileBase) level++;
  const LevelGeom& L = g.lv[level];
  t -= L.blurTileBase;
  int            pitch;
  const uint8_t* src = level_raw(b, g, level, f, &pitch);
  uint8_t*       dst = b.blur + (long long)f * g.blurBytes + L.blurOff;
  if (level == 0 && !l0Aligned) return;  // every group of an unaligned level 0 is an edge item
  // interior groups 1 .. gLast: x0 >= 4 and x0 + 8 <= w
  const int nInt = max((L.w - 8) >> 2, 0), strips = L.blurTilesY;
  const int item = t * 256 + tid;
  if (item >= nInt * strips) return;
  const int strip = item / nInt, gi = 1 + item - strip * nInt;
  blur_strip<true, kBlurRows>(src, pitch, dst, L.blurPitch, L.w, L.h, gi * 4, strip * kBlurRows);
}

__global__ __launch_bounds__(256, 8) void k_blur_levels(Bufs b, FrameGeom g, int l0Aligned) { blur_levels_block(b, g, l0Aligned, blockIdx.x, blockIdx.y); }

// Edge columns of the aligned levels: the first 4 columns and the last 4-8 columns (everything from 4 * (gLast + 1) on) of
// every row.  A lane owns one ROW: its loads stay inside the first / last cache line of that row, so every edge line is
// fetched once (column strips re-fetch each of them for every strip that overlaps it, scattered 128-byte reads that cost
// as much as the whole interior pass).  Horizontal sums go through LDS, then lanes switch to one OUTPUT row each.
constexpr int kBlurEdgeOutRows = 250;  // output rows per workgroup (256 input rows with the 3 + 3 halo)
constexpr int kBlurEdgeCols    = 12;   // 4 left + up to 8 right

__host__ __device__ inline int blur_edge_blocks(const LevelGeom& L) { return level_is_tiny(L) ? 0 : (L.h + kBlurEdgeOutRows - 1) / kBlurEdgeOutRows; }

__device__ __forceinline__ void blur_edges_block(const Bufs& b, const FrameGeom& g, int firstLevel, int block, int f) {
  __shared__ uint32_t hs[256][kBlurEdgeCols + 1];
  const int tid = threadIdx.x;
  int       t = block, level = firstLevel;
  for (;; level++) {
    const int nb = blur_edge_blocks(g.lv[level]);
    if (t < nb || level + 1 >= g.nlevels) break;
    t -= nb;
  }
  const LevelGeom& L = g.lv[level];
  int              pitch;
  const uint8_t*   src = level_raw(b, g, level, f, &pitch);
  uint8_t*         dst = b.blur + (long long)f * g.blurBytes + L.blurOff;
  const int        w = L.w, h = L.h, y0 = t * kBlurEdgeOutRows;
  const int        xr0 = 4 * (((w - 8) >> 2) + 1), nRight = min(4 * ((w + 3) >> 2) - xr0, kBlurEdgeCols - 4);
  {  // horizontal sums of input row y0 + tid - 3 (reflect-101): five aligned dword loads, bytes picked at compile time
    int sy = y0 + tid - 3;
    sy     = sy < 0 ? -sy : sy;
    sy     = sy >= h ? max(2 * h - 2 - sy, 0) : sy;
    const uint8_t* __restrict__ row = src + (long long)sy * pitch;
    const int      d  = w - xr0;  // real columns right of xr0: 4 .. 7
    const uint32_t l0 = *reinterpret_cast<const uint32_t*>(row), l1 = *reinterpret_cast<const uint32_t*>(row + 4);
    const uint32_t r0 = *reinterpret_cast<const uint32_t*>(row + xr0 - 4), r1 = *reinterpret_cast<const uint32_t*>(row + xr0);
    const uint32_t r2 = d > 4 ? *reinterpret_cast<const uint32_t*>(row + xr0 + 4) : 0u;  // never read past a row that ends at xr0 + 4
    int Lb[8], Rb[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Lb[i] = (l0 >> (8 * i)) & 0xff, Lb[4 + i] = (l1 >> (8 * i)) & 0xff;
      Rb[i] = (r0 >> (8 * i)) & 0xff, Rb[4 + i] = (r1 >> (8 * i)) & 0xff, Rb[8 + i] = (r2 >> (8 * i)) & 0xff;
    }
    const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
#pragma unroll
    for (int c = 0; c < 4; c++) {  // left: column -k mirrors to k
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) acc += (uint32_t)(taps[k] * Lb[c + k - 3 < 0 ? 3 - c - k : c + k - 3]);
      hs[tid][c] = acc;
    }
    // right: Rb[i] = column xr0 - 4 + i; column xr0 + t with t >= d mirrors to xr0 + 2 (d - 1) - t
    auto right = [&](auto dTag) {
      constexpr int D = decltype(dTag)::value;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        uint32_t acc = 0;
        if (j < D) {
#pragma unroll
          for (int k = 0; k < 7; k++) {
            const int t = j + k - 3;
            acc += (uint32_t)(taps[k] * Rb[(t < D ? t : 2 * (D - 1) - t) + 4]);
          }
        }
        hs[tid][4 + j] = acc;
      }
    };
    if (d == 4)
      right(std::integral_constant<int, 4>{});
    else if (d == 5)
      right(std::integral_constant<int, 5>{});
    else if (d == 6)
      right(std::integral_constant<int, 6>{});
    else
      right(std::integral_constant<int, 7>{});
  }
  __syncthreads();
  const int oy = y0 + tid;
  if (tid >= kBlurEdgeOutRows || oy >= h) return;
  uint32_t o[kBlurEdgeCols];
#pragma unroll
  for (int j = 0; j < kBlurEdgeCols; j++) {
    const uint32_t sacc = 18u * (hs[tid][j] + hs[tid + 6][j]) + 34u * (hs[tid + 1][j] + hs[tid + 5][j]) + 48u * (hs[tid + 2][j] + hs[tid + 4][j]) + 56u * hs[tid + 3][j];
    o[j]                = (sacc + 32768u) >> 16;
  }
  uint8_t* drow = dst + (long long)oy * L.blurPitch;
  *reinterpret_cast<uint32_t*>(drow) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
  *reinterpret_cast<uint32_t*>(drow + xr0) = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
  if (nRight > 4) *reinterpret_cast<uint32_t*>(drow + xr0 + 4) = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
}

__global__ __launch_bounds__(256) void k_blur_edges(Bufs b, FrameGeom g, int firstLevel) { blur_edges_block(b, g, firstLevel, blockIdx.x, blockIdx.y); }

// Both in one launch: the edge blocks (latency-bound: every lane walks one row of the first / last cache line) are dealt first and run
// beside the interior blocks (VALU-bound) instead of after them.  edgePad = the edge block count rounded up to a multiple of 8, so that
// the interior blocks keep their block -> XCD mapping.
__global__ __launch_bounds__(256, 8) void k_blur_all(Bufs b, FrameGeom g, int l0Aligned, int firstEdgeLevel, int edgeBlocks, int edgePad) {
  if ((int)blockIdx.x < edgePad) {
    if ((int)blockIdx.x < edgeBlocks) blur_edges_block(b, g, firstEdgeLevel, blockIdx.x, blockIdx.y);
  } else {
    blur_levels_block(b, g, l0Aligned, blockIdx.x - edgePad, blockIdx.y);
  }
}

// Unaligned level 0 (the caller's buffer is not 4-byte aligned / pitched): every group takes the byte-load strip.
__global__ __launch_bounds__(256) void k_blur_level0_bytes(Bufs b, FrameGeom g) {
  const int        f = blockIdx.y, tid = threadIdx.x;
  const LevelGeom& L = g.lv[0];
  int              pitch;
  const uint8_t*   src = level_raw(b, g, 0, f, &pitch);
  uint8_t*         dst = b.blur + (long long)f * g.blurBytes + L.blurOff;
  const int        groups = (L.w + 3) >> 2, strips = (L.h + kBlurEdgeRows - 1) / kBlurEdgeRows;
  const int        item = blockIdx.x * 256 + tid;
  if (item >= groups * strips) return;
  const int strip = item / groups, gi = item - strip * groups;
  blur_strip<false, kBlurEdgeRows>(src, pitch, dst, L.blurPitch, L.w, L.h, gi * 4, strip * kBlurEdgeRows);
}

// ------------------------------------------------------------------------------------------------
// IC_Angle (:66-91) + computeOrbDescriptor (:94-139) + output record (:1115-1134): one wave per keypoint.
// ------------------------------------------------------------------------------------------------
// IC_Angle disc (src/ORBextractor.cc:66-91, umax :448-464): the 31 rows are read as 31 x 8 dwords (columns u = -15 .. 16 of
// row v); momMask[slot] has 0xff in the bytes with |u| <= umax[|v|].  With I' = I & mask:
//   m10 = sum(u I') = dot4(I', {u + 15}) - 15 dot4(I', {1}),   m01 = sum_v v * dot4(I', {1}).
constexpr int kMomSlots = 31 * 8;
struct PatternTab {
  signed char v[1024];
  uint32_t    momMask[kMomSlots];
};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2 scalar path
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float       a, c, c2;
  if (ax >= ay) {
    c  = ay / (ax + eps);
    c2 = c * c;
    a  = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c  = ax / (ay + eps);
    c2 = c * c;
    a  = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // src/ORBextractor.cc:448-464

constexpr int kOdKpPerBlock = 8;  // two keypoints per wave (one per 32-lane half), four waves

__global__ __launch_bounds__(256, 6) void k_orient_describe(Bufs b, FrameGeom g, const PatternTab* __restrict__ pat, const int* counts,
                                                          KpRec* kps, uint8_t* desc, int cap, int nframes, int nbPerFrame, int tapMode) {
  // 1-D grid with frame -> XCD affinity: all workgroups of frame f run on XCD f % 8 (blocks b, b + 8, ... share an
  // L2), so the raw + blurred levels of a frame (1.9 MB at VGA) are pulled into ONE 4-MiB L2 instead of all eight.
  // A keypoint is a short chain of dependent steps (info -> pixels -> angle -> taps), so the kernel is bound by latency x
  // resident waves, not by arithmetic: each wave therefore carries TWO keypoints, one per 32-lane half -- the same
  // instruction stream, twice the work in flight.
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
  const int q = blockIdx.x >> 3, f = (blockIdx.x & 7) + 8 * (q / nbPerFrame);
  if (f >= nframes) return;
  // the 256 pattern pairs (4 bytes each) go to LDS once per workgroup: the taps are then a short LDS read right where they are
  // needed instead of 8 registers held across the trigonometry
  __shared__ float4 sPat[256];  // (x0, y0, x1, y1) of pair i as floats: one 16-byte LDS read per pair instead of four byte-to-float conversions
  {
    const char4 c4    = __builtin_bit_cast(char4, reinterpret_cast<const uint32_t*>(pat->v)[threadIdx.x]);
    sPat[threadIdx.x] = make_float4((float)c4.x, (float)c4.y, (float)c4.z, (float)c4.w);
  }
  __syncthreads();
  const int i = (q % nbPerFrame) * kOdKpPerBlock + (threadIdx.x >> 6) * 2 + half;
  // (packed candidate, level, output row) were laid down by k_layout: one load, requested together with the frame's count
  const int  n    = counts[2 * f];
  int4       inf0 = make_int4(0, 0, 0, 0), inf1 = inf0;
  if (i < g.totalKp) {
    inf0 = b.outIdx[((long long)f * g.totalKp + i) * 2];
    inf1 = b.outIdx[((long long)f * g.totalKp + i) * 2 + 1];
  }
  if (i - half >= n) return;   // both halves idle (wave-uniform)
  const bool valid = i < n;
  if (!valid) {  // idle half: mirror the wave's first keypoint (in-bounds addresses, results discarded)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      (&inf0.x)[c] = __shfl((&inf0.x)[c], 0, 64);
      (&inf1.x)[c] = __shfl((&inf1.x)[c], 0, 64);
    }
  }
  const uint32_t r     = (uint32_t)inf0.x;
  const int      level = inf0.y & 15, row = inf0.y >> 4;
  const int      px = ot_cx(r) + kMinBorder, py = ot_cy(r) + kMinBorder;
  const int      pitch = inf0.z, blurPitch = inf0.w;
  // The blurred 37 x 37 window around the keypoint only depends on (px, py): request it now, next to the moment loads
  // below, so that the wave pays one memory round trip for both.
  const int      ax  = (px - 18) & ~3;  // aligned first column; the blurred level base and pitch are 64-byte aligned
  const uint8_t* blr = b.blur + (long long)f * g.blurBytes;  // uniform base + 32-bit lane offsets
  uint32_t       win[13];
  // window dword idx = hl + 32 k sits at (row idx / 11, dword idx % 11) of the blurred level and -- the LDS copy being 11 dwords x 37
  // rows back to back -- at byte 4 idx of the keypoint's LDS slice.  The division once, then +32 = +2 rows +10 dwords with a carry.
  {
    int rr = hl / 11, cc = hl - rr * 11;
    uint32_t off = (uint32_t)(inf1.y + rr * blurPitch + cc * 4);
    const uint32_t step = (uint32_t)(2 * blurPitch + 40), carry = (uint32_t)(blurPitch - 44);
#pragma unroll
    for (int k = 0; k < 13; k++) {  // unconditional loads (the last round's lanes past dword 406 re-read it): all 13 in flight together
      const bool in = hl + 32 * k < 37 * 11;
      win[k]        = *reinterpret_cast<const uint32_t*>(blr + (in ? off : (uint32_t)(inf1.y + 36 * blurPitch + 40)));
      off += step, cc += 10;
      if (cc >= 11) cc -= 11, off += carry;
    }
  }

  // ---- orientation on the raw level
  int m10 = 0, m01 = 0;
  {
    // Half-lane l owns the slots l, l + 32, ... of the 31 x 8 dword grid; the pixel dword is an unaligned 32-bit load
    // straight at (px - 15 + 4 j, py - 15 + row) and the disc mask / coordinates sit in the per-slot weight vectors.
    const uint8_t* rp = (level == 0 ? b.img0 + (long long)f * b.img0FrameStride : b.pyr + (long long)f * g.pyrBytes) + inf1.x;
    uint32_t       su = 0, s1 = 0;
    int            sv = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int      idx = min(hl + 32 * k, kMomSlots - 1);  // 248 slots = 7.75 rounds: the clamped duplicates are masked off
      const uint32_t msk = hl + 32 * k < kMomSlots ? pat->momMask[idx] : 0u;
      const uint8_t* p   = rp + (idx >> 3) * pitch + (idx & 7) * 4;
      uint32_t       dw;
      __builtin_memcpy(&dw, p, 4);
      dw &= msk;
      const uint32_t t1 = __builtin_amdgcn_udot4(dw, 0x01010101u, 0u, false);
      su                = __builtin_amdgcn_udot4(dw, 0x03020100u + (uint32_t)(idx & 7) * 0x04040404u, su, false);  // bytes u + 15
      s1 += t1;
      sv += ((idx >> 3) - kHalfPatch) * (int)t1;
    }
    m10 = (int)su - kHalfPatch * (int)s1;
    m01 = sv;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {  // within the 32-lane half
      m10 += __shfl_xor(m10, off, 64);
      m01 += __shfl_xor(m01, off, 64);
    }
  }
  // The window goes to this keypoint's LDS slice as soon as it arrives (its 13 registers are free for the trigonometry
  // below); the taps later are LDS byte reads.
  __shared__ __align__(16) uint8_t patch[kOdKpPerBlock][37 * 44];
  uint8_t*       pw  = patch[(threadIdx.x >> 6) * 2 + half];
#pragma unroll
  for (int k = 0; k < 13; k++) {
    if (hl + 32 * k < 37 * 11) *reinterpret_cast<uint32_t*>(pw + (hl + 32 * k) * 4) = win[k];
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // ---- steered BRIEF on the blurred level
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float       sn, cs;
  orbport::sincosf_port<false>(angle * factorPI, &sn, &cs);
  const float    a = cs, bb = sn;
  // The 512 taps land anywhere in a 37 x 37 window; gathering them straight from global memory costs one cache-line
  // lookup per lane per tap.  The window (requested above with coalesced aligned 32-bit loads, 11 dwords x 37 rows) goes
  // into this keypoint's LDS slice, then the taps are LDS byte reads.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // single wave: its own LDS writes (issued before the moments) are complete
  const uint8_t* bl = pw + 18 * 44 + (px - ax);  // window centre
  const int      bp = 44;
  uint32_t       mine = 0;  // half-lane w < 8 ends up with descriptor word w of its keypoint
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const float4 p4 = sPat[w * 32 + hl];
    const float  x0 = p4.x, y0 = p4.y, x1 = p4.z, y1 = p4.w;
    // tap rotation as the reference's compiler evaluates it (include/orbx.h: orbx_set_tap_contraction; wave-uniform switch)
    int r0, c0, r1, c1;
    if (tapMode == 1) {  // g++ -O3 -march=native on an FMA host: fma(x, b, y * a), fma(x, a, -(y * b))
      r0 = __float2int_rn(__fmaf_rn(x0, bb, y0 * a)), c0 = __float2int_rn(__fmaf_rn(x0, a, -(y0 * bb)));
      r1 = __float2int_rn(__fmaf_rn(x1, bb, y1 * a)), c1 = __float2int_rn(__fmaf_rn(x1, a, -(y1 * bb)));
    } else if (tapMode == 2) {  // the other contraction: fma(y, a, x * b), fma(-y, b, x * a)
      r0 = __float2int_rn(__fmaf_rn(y0, a, x0 * bb)), c0 = __float2int_rn(__fmaf_rn(-y0, bb, x0 * a));
      r1 = __float2int_rn(__fmaf_rn(y1, a, x1 * bb)), c1 = __float2int_rn(__fmaf_rn(-y1, bb, x1 * a));
    } else {  // no contraction
      r0 = __float2int_rn(x0 * bb + y0 * a), c0 = __float2int_rn(x0 * a - y0 * bb);
      r1 = __float2int_rn(x1 * bb + y1 * a), c1 = __float2int_rn(x1 * a - y1 * bb);
    }
    const int   t0 = bl[r0 * bp + c0], t1 = bl[r1 * bp + c1];
    const unsigned long long bal = __ballot(t0 < t1);
    const uint32_t           wv  = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
    mine                         = hl == w ? wv : mine;
  }
  // ---- output row
  if (valid && row < cap) {
    if (hl < 8) *reinterpret_cast<uint32_t*>(desc + ((long long)f * cap + row) * 32 + hl * 4) = mine;
    if (hl == 0) {
      KpRec k;
      k.x = (float)px;
      k.y = (float)py;
      if (level != 0) {
        k.x = k.x * __int_as_float(inf1.z);
        k.y = k.y * __int_as_float(inf1.z);
      }
      k.size     = (float)inf1.w;
      k.angle    = angle;
      k.response = (float)ot_cr(r);
      k.octave   = level;
      k.class_id = -1;
      kps[(long long)f * cap + row] = k;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (reference src/Frame.cc:944-1113) on the two device-resident pyramids.
// One wave per left keypoint: (1) the row-band candidate scan -- the reference's vRowIndices[vL] holds right keypoint
// iR iff floor(yR - r) <= (int)vL <= ceil(yR + r), r = 2 * scale[octave R], lists in increasing iR -- as a strided scan
// over all right keypoints with the same octave and u-range filters, best Hamming = wave-min of (dist << 16 | iR)
// (strict '<' in increasing iR = smallest iR among equal distances); (2) 11 x 11 SAD over the 11 shifts on the
// keypoint's pyramid level (integer sums, exact), parabola fit and depth in the reference's float sequence by lane 0.
// k_stereo_cut then applies the median-distance cut (:1101-1112): the median is the (n/2)-th smallest SAD distance.
// ------------------------------------------------------------------------------------------------
struct StereoArgs {
  const KpRec *  kl, *kr;
  const uint8_t *dl, *dr;
  int            nl, nr, nrows;
  float          mb, mbf;
  const uint8_t *pyrL0, *pyrR0;  // level-0 images (row pitch pitch0)
  int            pitch0;
  const uint8_t *pyrL, *pyrR;    // levels >= 1 (frame 0 of each handle)
  float *        uRight, *depth;
  int*           sad;            // best SAD distance per left keypoint, -1 = not a match
  // batch form (cap > 0): blockIdx.y = frame f of the two handles' last extraction batch; every array is [frames][cap],
  // counts are the extraction batches' [frames][2] arrays
  int            cap;
  const int *    countsL, *countsR;
  long long      img0StrideL, img0StrideR, pyrBytes;
  // row index of the right keypoints (k_stereo_rows): rowIdx[rowStart[y] .. rowStart[y + 1]) = the right keypoints with (int)pt.y == y;
  // a left keypoint on row v only has to look at rows v - rowBand .. v + rowBand (rowBand >= 2 * the largest scale factor + 1 covers
  // every iR whose band [floor(y - r), ceil(y + r)] holds v).  NULL = scan all right keypoints.
  int *rowStart, *rowIdx;
  int  rowBand;
  int* fault;  // Bufs::fault of the left handle (checked_count)
};

constexpr int kStereoMaxRows = 4096;  // image rows the row index is built for (taller images scan all right keypoints)

__device__ __forceinline__ StereoArgs stereo_view(const StereoArgs& a, int f) {
  if (a.cap <= 0) return a;
  StereoArgs      v  = a;
  const long long fc = (long long)f * a.cap;
  v.kl = a.kl + fc, v.kr = a.kr + fc, v.dl = a.dl + fc * 32, v.dr = a.dr + fc * 32;
  v.nl = checked_count(a.fault, kFaultStereoLeft, f, 0, a.countsL[2 * f], a.cap);
  v.nr = checked_count(a.fault, kFaultStereoRight, f, 0, a.countsR[2 * f], a.cap);
  v.pyrL0 = a.pyrL0 + f * a.img0StrideL, v.pyrR0 = a.pyrR0 + f * a.img0StrideR;
  v.pyrL = a.pyrL + f * a.pyrBytes, v.pyrR = a.pyrR + f * a.pyrBytes;
  v.uRight = a.uRight + fc, v.depth = a.depth + fc, v.sad = a.sad + fc;
  if (a.rowStart) v.rowStart = a.rowStart + (long long)f * (a.nrows + 1), v.rowIdx = a.rowIdx + fc;
  return v;
}

// vRowIndices of Frame::ComputeStereoMatches (src/Frame.cc:955-973) as a counting sort of the right keypoints by image row: one
// workgroup per frame.  The per-row LISTS of the reference (every keypoint entered in all rows of its band) are not built; the
// matcher walks the rows of the band instead and applies the reference's own band test to every keypoint it finds there.
__global__ __launch_bounds__(256) void k_stereo_rows(StereoArgs a0) {
  __shared__ int hist[kStereoMaxRows + 1];
  __shared__ int part[256];
  const StereoArgs a   = stereo_view(a0, blockIdx.x);
  const int        tid = threadIdx.x, nrows = a.nrows;
  for (int i = tid; i <= nrows; i += 256) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < a.nr; i += 256) atomicAdd(&hist[min(max((int)a.kr[i].y, 0), nrows - 1)], 1);
  __syncthreads();
  // exclusive scan over the rows: per-thread chunks, then the chunk totals
  const int per = (nrows + 255) / 256, lo = min(tid * per, nrows), hi = min(lo + per, nrows);
  int       sum = 0;
  for (int i = lo; i < hi; i++) sum += hist[i];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; i++) {
      const int t = part[i];
      part[i]     = run;
      run += t;
    }
  }
  __syncthreads();
  int run = part[tid];
  for (int i = lo; i < hi; i++) {
    const int t   = hist[i];
    hist[i]       = run;  // becomes the fill cursor of the row
    a.rowStart[i] = run;
    run += t;
  }
  if (tid == 255) a.rowStart[nrows] = a.nr;
  __syncthreads();
  // (the slot is clamped: should the keypoints change between the two passes -- they must not -- a row's cursor cannot leave the list)
  for (int i = tid; i < a.nr; i += 256) a.rowIdx[min(atomicAdd(&hist[min(max((int)a.kr[i].y, 0), nrows - 1)], 1), a.nr - 1)] = i;
}

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a0, FrameGeom g) {
  const StereoArgs a  = stereo_view(a0, blockIdx.y);
  const int        iL = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (iL >= a.nl) return;
  const KpRec kpL    = a.kl[iL];
  const int   levelL = min(max(kpL.octave, 0), g.nlevels - 1);  // (records are the extractor's own; the clamps here and below only fence off memory)
  const float vL = kpL.y, uL = kpL.x;
  const float minD = 0.f, maxD = a.mbf / a.mb;
  const float minU = uL - maxD, maxU = uL - minD;
  int         sadOut = -1;
  float       uOut = -1.0f, dOut = -1.0f;
  const int   rowL = (int)vL;  // vRowIndices[vL]
  uint32_t    best = 0xffffffffu;
  if (!(maxU < 0)) {
    const uint4 a0 = *reinterpret_cast<const uint4*>(a.dl + (size_t)iL * 32), a1 = *reinterpret_cast<const uint4*>(a.dl + (size_t)iL * 32 + 16);
    const bool byRow = a.rowStart != nullptr;
    int        pLo = 0, pHi = a.nr;
    if (byRow) {
      const bool in = rowL >= 0 && rowL < a.nrows;
      pLo           = in ? a.rowStart[max(rowL - a.rowBand, 0)] : 0;
      pHi           = in ? a.rowStart[min(rowL + a.rowBand + 1, a.nrows)] : 0;
    }
    for (int p = pLo + lane; p < pHi; p += 64) {
      const int   iR   = byRow ? min(max(a.rowIdx[p], 0), a.nr - 1) : p;  // an index read from memory never leaves the keypoint list
      const KpRec kpR  = a.kr[iR];
      const float r    = 2.0f * g.lv[min(max(kpR.octave, 0), g.nlevels - 1)].scale;
      const int   maxr = (int)ceilf(kpR.y + r), minr = (int)floorf(kpR.y - r);
      if (rowL < minr || rowL > maxr || rowL < 0 || rowL >= a.nrows) continue;
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      if (kpR.x >= minU && kpR.x <= maxU) {
        const uint4 b0 = *reinterpret_cast<const uint4*>(a.dr + (size_t)iR * 32), b1 = *reinterpret_cast<const uint4*>(a.dr + (size_t)iR * 32 + 16);
        const int   d  = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                      __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        best = min(best, ((uint32_t)d << 16) | (uint32_t)iR);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
  }
  const int bestDist = best == 0xffffffffu ? 100 : (int)(best >> 16);
  if (bestDist < 100 && bestDist < (100 + 50) / 2) {  // < TH_HIGH (strict update from TH_HIGH) and < thOrbDist
    const int   bestIdxR    = (int)(best & 0xffffu);
    const float uR0         = a.kr[bestIdxR].x;
    const float scaleFactor = 1.0f / g.lv[levelL].scale;  // mvInvScaleFactor (src/ORBextractor.cc:416)
    const float scaleduL = roundf(kpL.x * scaleFactor), scaledvL = roundf(kpL.y * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
    const int   w = 5, Ls = 5;
    const float iniu = scaleduR0 + Ls - w, endu = scaleduR0 + Ls + w + 1;
    if (!(iniu < 0 || endu >= g.lv[levelL].w)) {
      const uint8_t *imL, *imR;
      int            pitch;
      if (levelL == 0) {
        imL = a.pyrL0, imR = a.pyrR0, pitch = a.pitch0;
      } else {
        imL = a.pyrL + g.lv[levelL].pyrOff, imR = a.pyrR + g.lv[levelL].pyrOff, pitch = g.lv[levelL].pitch;
      }
      const int y0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w), xr0 = (int)(scaleduR0 - w);
      // keypoints lie >= 19 px inside their level, so the 11 x 11 window and its +-5 shifts do too; a record that does not say so is skipped
      const bool inside = y0 >= 0 && y0 + 11 <= g.lv[levelL].h && xl0 >= 0 && xl0 + 11 <= g.lv[levelL].w && xr0 - 5 >= 0 && xr0 + 16 <= g.lv[levelL].w;
      int        acc[11];
#pragma unroll
      for (int k = 0; k < 11; k++) acc[k] = 0;
      for (int p = lane; inside && p < 121; p += 64) {
        const int yy = p / 11, xx = p - yy * 11;
        const int vl = imL[(long long)(y0 + yy) * pitch + xl0 + xx];
        const uint8_t* rr = imR + (long long)(y0 + yy) * pitch + xr0 + xx;
#pragma unroll
        for (int k = 0; k < 11; k++) acc[k] += abs(vl - (int)rr[k - 5]);
      }
#pragma unroll
      for (int k = 0; k < 11; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
      int   bestS = 0x7fffffff, bestinc = 0;
      float vDists[11];
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float dist = (float)acc[k];
        if (dist < (float)bestS) {
          bestS   = (int)dist;
          bestinc = k - 5;
        }
        vDists[k] = dist;
      }
      if (inside && !(bestinc == -Ls || bestinc == Ls)) {
        float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
        for (int k = 1; k < 10; k++)
          if (k - 5 == bestinc) dist1 = vDists[k - 1], dist2 = vDists[k], dist3 = vDists[k + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR    = g.lv[levelL].scale * ((float)scaleduR0 + (float)bestinc + deltaR);
          float disparity = (uL - bestuR);
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
              disparity = 0.01;
              bestuR    = uL - 0.01;
            }
            dOut   = a.mbf / disparity;
            uOut   = bestuR;
            sadOut = bestS;
          }
        }
      }
    }
  }
  if (lane == 0) {
    a.uRight[iL] = uOut;
    a.depth[iL]  = dOut;
    a.sad[iL]    = sadOut;
  }
}

__global__ __launch_bounds__(256) void k_stereo_cut(StereoArgs a0, int* kept0) {
  const StereoArgs a    = stereo_view(a0, blockIdx.x);
  int*             kept = kept0 + blockIdx.x;
  __shared__ int   sMedian, sCount;
  const int        tid = threadIdx.x;
  if (tid == 0) sMedian = -1, sCount = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < a.nl; i += 256) cnt += a.sad[i] >= 0;
  atomicAdd(&sCount, cnt);
  __syncthreads();
  const int n = sCount;
  if (n == 0) {
    if (tid == 0) *kept = 0;
    return;
  }
  // std::sort of (dist, iL) pairs, element n/2: only its DISTANCE is used, i.e. the (k + 1)-th smallest distance = the
  // smallest d with #(dist <= d) >= k + 1.  Found by bisection on d; each step counts with all threads (each keeps its
  // distances in registers, loaded once).
  const int k = n / 2;
  constexpr int kPer = 32;  // 256 x 32 = 8192 left keypoints held in registers; beyond that the tail is re-read
  int           mine[kPer];
#pragma unroll
  for (int r = 0; r < kPer; r++) {
    const int i = tid + 256 * r;
    mine[r]     = i < a.nl ? a.sad[i] : -1;
  }
  int lo = 0, hi = 11 * 11 * 255;  // SAD of an 11 x 11 u8 window
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int       c   = 0;
#pragma unroll
    for (int r = 0; r < kPer; r++) c += (mine[r] >= 0 && mine[r] <= mid) ? 1 : 0;
    for (int i = tid + 256 * kPer; i < a.nl; i += 256) {
      const int d = a.sad[i];
      c += (d >= 0 && d <= mid) ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) sCount = 0;
    __syncthreads();
    if (c) atomicAdd(&sCount, c);
    __syncthreads();
    if (sCount >= k + 1)
      hi = mid;
    else
      lo = mid + 1;
  }
  if (tid == 0) sMedian = lo;
  __syncthreads();
  const float median = (float)sMedian;
  const float thDist = 1.5f * 1.4f * median;
  if (tid == 0) sCount = 0;
  __syncthreads();
  int keep = 0;
  for (int i = tid; i < a.nl; i += 256) {
    const int di = a.sad[i];
    if (di < 0) continue;
    if ((float)di < thDist) {
      keep++;
    } else {
      a.uRight[i] = -1.0f;
      a.depth[i]  = -1.0f;
    }
  }
  atomicAdd(&sCount, keep);
  __syncthreads();
  if (tid == 0) *kept = sCount;
}

// ------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints (src/Frame.cc:886-913): one thread per keypoint, cv::undistortPoints in doubles (undistort_core.h);
// every other KeyPoint field is copied.  counts == NULL: all `cap` records of every frame.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_undistort(const KpRec* __restrict__ kps, const int* __restrict__ counts, int cap, UndistortParams p,
                                                   KpRec* __restrict__ out) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = counts ? counts[2 * f] : cap;
  if (i >= n) return;
  KpRec k = kps[(long long)f * cap + i];
  undistort_point(p, k.x, k.y, &k.x, &k.y);
  out[(long long)f * cap + i] = k;
}

// ------------------------------------------------------------------------------------------------
// Copy-out of result records to pinned host memory (orbx_extract_async / orbx_async_download).  The runtime's own device -> host
// copy is a full-grid shader blit (no SDMA for this direction here), and ANY kernel that stores to host memory faster than PCIe
// drains it fills the memory-side write queues with host-bound lines: kernels running beside it then stall on their own HBM
// writes (measured: a 66 us pyramid launch took 300 us next to an unthrottled copy).  So this copy keeps only about one
// few KB in flight -- kCopyOutBlocks x 4 waves x 4 KB, every wave drains its stores (vmcnt(0)) before it issues
// the next 4 KB -- and runs on a stream of its own beside the next chunk's kernels.  n16 = 16-byte words; tail4 dwords follow.
// ------------------------------------------------------------------------------------------------
constexpr int kCopyOutBlocks = 2;  // measured at VGA: 1 -> 152 k, 2 -> 160 k, 3 -> 154 k, 4 -> 147 k, 8 -> 140 k frames/s PCIe-inclusive (no copy-out at all: 162 k)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_out(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n16, int tail4) {
  const long long stride = (long long)gridDim.x * 256;
  long long       i      = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // throttle: at most 4 KB per wave on its way to the host
  }
  for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail4)
    reinterpret_cast<uint32_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint32_t*>(src + n16)[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// PMC calibration: stream `n` bytes with W bytes per lane per load (W = 1, 4, 16) and fold them into one word,
// so that rocprofv3's FETCH_SIZE can be compared with a known byte count for each access width this library
// uses (MI355X_MICROARCH.md, HBM: the counter is only calibrated for 16 B/lane streams).
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_calib_read(const uint8_t* __restrict__ src, long long n, unsigned* sink) {
  const long long stride = (long long)gridDim.x * 256 * W;
  unsigned        acc    = 0;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * W; i + W <= n; i += stride) {
    if (W == 1) acc += src[i];
    if (W == 4) acc += *reinterpret_cast<const unsigned*>(src + i);
    if (W == 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + i);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345677u) *sink = acc;  // keeps the loads alive
}

}  // namespace orbx
#endif
