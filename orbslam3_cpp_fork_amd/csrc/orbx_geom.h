// orbx_geom.h -- per-image-size geometry tables shared by host orchestration and kernels.
#ifndef ORBX_GEOM_H_
#define ORBX_GEOM_H_

#include <stdint.h>

namespace orbx {

constexpr int kMaxLevels      = 16;
constexpr int kEdgeThreshold  = 19;  // EDGE_THRESHOLD      (src/ORBextractor.cc:64)
constexpr int kHalfPatch      = 15;  // HALF_PATCH_SIZE     (src/ORBextractor.cc:63)
constexpr int kPatchSize      = 31;  // PATCH_SIZE          (src/ORBextractor.cc:62)
constexpr int kMinBorder      = kEdgeThreshold - 3;  // minBorderX/Y (src/ORBextractor.cc:761-762)
constexpr int kFastTileMax    = 80;  // FAST cell + 6 px overlap is at most 77 px wide/high (see make_geometry)
constexpr int kBlurTileW      = 64;
constexpr int kBlurTileH      = 16;

struct LevelGeom {
  int   w, h;               // level size (cvRound(cols * inv scale), src/ORBextractor.cc:1143)
  int   pitch;              // row pitch of this level in the pyramid / blurred buffers (levels >= 1; level 0 raw = input)
  int   pyrOff;             // byte offset of the level inside one frame's pyramid block (level 0: unused)
  int   blurOff;            // byte offset inside one frame's blurred block
  int   blurPitch;
  int   nCols, nRows, wCell, hCell;  // FAST cell grid (src/ORBextractor.cc:769-775)
  int   maxBX, maxBY;       // maxBorderX/Y
  int   cellBase;           // index of the level's first cell in per-frame cell arrays
  int   slotBase;           // index of the level's first candidate slot in per-frame point arrays
  int   slotCount;          // candidate capacity of the level (sum of cell capacities)
  int   nodeBase, nodeCap;  // per-frame node arrays
  int   kpBase, kpCap;      // per-frame per-level keypoint staging
  int   N;                  // mnFeaturesPerLevel[level]
  int   nIni;               // round(width / height) root nodes
  float hX;
  float scale;              // mvScaleFactor[level]
  int   patchSize;          // (int)(PATCH_SIZE * scale)
  int   xtabOff, ytabOff;   // offsets into the resize tables (entries), valid for level >= 1
  int   blurTileBase, blurTilesX, blurTilesY;  // interior work items: first workgroup of the level, 4-column groups, row strips
};

struct FrameGeom {
  int       nlevels;
  int       rows, cols;
  int       totalCells, totalSlots, totalNodes, totalKp, totalBlurTiles;
  int       pyrBytes, blurBytes;  // per frame
  LevelGeom lv[kMaxLevels];
};

}  // namespace orbx
#endif
