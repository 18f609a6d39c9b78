// orb_params.h -- host-side constant tables of the extractor and the per-resolution geometry.
// Restates ORBextractor::ORBextractor (src/ORBextractor.cc:415-486), the level sizes of
// ComputePyramid (:1207-1208) and the FAST grid of ComputeKeyPointsOctTree (:823-869).
#pragma once
#include <stdint.h>
#include <vector>

#include "orb_spec.h"

namespace orbfe {

constexpr int kMaxLevels = 16;

struct ExtractorTables {
  int nfeatures = 0;
  double scaleFactor = 0;  // the reference member is double (include/ORBextractor.h:100)
  float scaleFactorArg = 0;
  int nlevels = 0, iniThFAST = 0, minThFAST = 0;
  float scale[kMaxLevels] = {}, invScale[kMaxLevels] = {};
  float sigma2[kMaxLevels] = {}, invSigma2[kMaxLevels] = {};
  int quota[kMaxLevels] = {};  // mnFeaturesPerLevel
  int umax[16] = {};
  void init(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);
};

// One FAST grid cell = one cv::FAST call of the reference (:874); (x0,y0,w,h) is the cell's
// DETECTION rectangle in level coordinates (sub-image minus its 3-px rim).
struct CellDesc {
  int16_t level;
  int16_t x0, y0, w, h;
  int16_t flags;     // kCell* bits; 0 for a FAST cell
  int32_t slotBase;  // first candidate slot of this cell inside the per-frame slot array
};
// Blur-only cells of the fused FAST+blur kernel: the frame of the level that no detection rectangle covers.
constexpr int kCellBlurOnly = 1;    // no FAST phases
constexpr int kCellRowReflect = 2;  // tile rows leave [0, h): BORDER_REFLECT_101 on the row index
constexpr int kCellColReflect = 4;  // tile columns leave [0, w): byte-wise staging with reflected columns

struct LevelGeom {
  int w, h, pitch;
  uint32_t off;  // byte offset of the level inside a per-frame pyramid buffer
  int nCols, nRows, wCell, hCell;
  int cellStart, nCells;
  int slotStart, slotCount;  // candidate slots (worst case) of this level
  int quota;                 // mnFeaturesPerLevel[level]
  int nIni;                  // root nodes of DistributeOctTree (:570)
  int kpStart, kpCap;        // per-level keypoint slots inside a per-frame keypoint array
};

struct ResizeTables {  // cv::resize INTER_LINEAR fixed-point coefficients (level l-1 -> l)
  std::vector<int32_t> xofs, yofs;
  std::vector<int16_t> alpha, beta;  // 2 per output column / row
  // the same coefficients packed for k_resize_flat (empty when its preconditions do not hold):
  // colrec: 12 dwords per group of 4 output columns = v_perm selectors x4 | (a0,a1) u16 pairs x4 | window start, 0,0,0
  // rowrec: 4 dwords per output row = clamped source rows r0, r1 | b0 << 16 | b1 << 16
  std::vector<uint32_t> colrec, rowrec;
  // for the fused blur + resize kernel (64 x 64 tiles of the SOURCE level): tileGx[tx] = first column group whose window
  // starts at or right of column 64*tx, tileDy[ty] = first output row whose upper source row is >= 64*ty; both end with
  // the totals, so tile (tx, ty) owns groups [tileGx[tx], tileGx[tx+1]) x rows [tileDy[ty], tileDy[ty+1])
  std::vector<int32_t> tileGx, tileDy;
};

struct FrameGeom {
  int W = 0, H = 0, nlevels = 0;
  LevelGeom lv[kMaxLevels] = {};
  std::vector<CellDesc> cells;  // FAST cells of all levels (level-major), then the blur-only frame cells
  int nFastCells = 0;           // length of the FAST prefix of `cells`
  bool fusedBlur = false;       // the FAST rectangles + frame cells tile every level exactly: fused kernel usable
  ResizeTables rz[kMaxLevels];
  uint32_t pyrBytes = 0;   // bytes of levels 1..n-1 (+ level 0 when owned) per frame
  int totalSlots = 0;      // candidate slots per frame
  int totalKpCap = 0;      // keypoint slots per frame
  int maxCellW = 0, maxCellH = 0;  // largest FAST detection rectangle (sizes the FAST kernel's LDS)
  void build(const ExtractorTables& t, int W, int H);
};

void build_resize_tables(int sw, int sh, int dw, int dh, ResizeTables* out);

}  // namespace orbfe
