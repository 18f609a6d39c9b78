#include "orb_params.h"

#include <cmath>
#include <cstring>

namespace orbfe {

static int cv_round_d(double v) { return (int)__builtin_nearbyint(v); }

// src/ORBextractor.cc:415-486
void ExtractorTables::init(int nf, float sf, int nl, int ini, int mn) {
  if (nl > kMaxLevels) nl = kMaxLevels;
  if (nl < 1) nl = 1;
  nfeatures = nf;
  scaleFactorArg = sf;
  scaleFactor = (double)sf;
  nlevels = nl;
  iniThFAST = ini;
  minThFAST = mn;
  scale[0] = 1.0f;
  sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    scale[i] = (float)((double)scale[i - 1] * scaleFactor);  // float * double member
    sigma2[i] = scale[i] * scale[i];
  }
  for (int i = 0; i < nl; i++) {
    invScale[i] = 1.0f / scale[i];
    invSigma2[i] = 1.0f / sigma2[i];
  }
  float factor = (float)(1.0 / scaleFactor);
  float want = (float)nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    quota[l] = cv_round_d((double)want);
    sum += quota[l];
    want *= factor;
  }
  quota[nl - 1] = nf - sum > 0 ? nf - sum : 0;
  // circular patch row half-widths, :471-485
  int v, v0;
  const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

static int16_t sat_short(float v) {
  int i = cv_round(v);
  return (int16_t)(i > 32767 ? 32767 : (i < -32768 ? -32768 : i));
}

// cv::resize INTER_LINEAR 8U coefficient tables (OpenCV imgproc, fixed point 1<<11).
void build_resize_tables(int sw, int sh, int dw, int dh, ResizeTables* t) {
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
  t->xofs.resize(dw);
  t->alpha.resize(2 * (size_t)dw);
  t->yofs.resize(dh);
  t->beta.resize(2 * (size_t)dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    t->xofs[dx] = sx;
    t->alpha[2 * dx] = sat_short((1.f - fx) * 2048);
    t->alpha[2 * dx + 1] = sat_short(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    t->yofs[dy] = sy;  // rows are clamped at use; the weights are not
    t->beta[2 * dy] = sat_short((1.f - fy) * 2048);
    t->beta[2 * dy + 1] = sat_short(fy * 2048);
  }
  t->colrec.clear();
  t->rowrec.clear();
  if (sw >= 8 && (long long)sw <= 2LL * dw) {  // the 4 columns' taps span <= 8 source bytes
    const int ngx = (dw + 3) / 4;
    t->colrec.resize(12 * (size_t)ngx);
    for (int g = 0; g < ngx; g++) {
      int sx[4];
      uint32_t* rec = &t->colrec[12 * (size_t)g];
      for (int k = 0; k < 4; k++) {
        const int dx = 4 * g + k < dw ? 4 * g + k : dw - 1;
        sx[k] = t->xofs[dx];
        rec[4 + k] = (uint32_t)(uint16_t)t->alpha[2 * dx] | ((uint32_t)(uint16_t)t->alpha[2 * dx + 1] << 16);
      }
      int sxb = sx[0];
      if (sxb > sw - 8) sxb = sw - 8;  // keep the 8-byte window inside the row
      if (sxb < 0) sxb = 0;
      for (int k = 0; k < 4; k++) {
        const uint32_t o = (uint32_t)(sx[k] - sxb);
        const uint32_t o1 = o + 1 < 8 ? o + 1 : o;  // tap sx+1 beyond the window only when its weight is 0
        rec[k] = o | 0x0c00u | (o1 << 16) | 0x0c000000u;
      }
      rec[8] = (uint32_t)sxb;
      rec[9] = rec[10] = rec[11] = 0;
    }
    t->rowrec.resize(4 * (size_t)dh);
    for (int dy = 0; dy < dh; dy++) {
      const int sy = t->yofs[dy];
      const int r0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
      const int r1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
      uint32_t* rec = &t->rowrec[4 * (size_t)dy];
      rec[0] = (uint32_t)r0;
      rec[1] = (uint32_t)r1;
      rec[2] = (uint32_t)(uint16_t)t->beta[2 * dy] << 16;
      rec[3] = (uint32_t)(uint16_t)t->beta[2 * dy + 1] << 16;
    }
    // ownership of the output by 64 x 64 source tiles (window starts and upper source rows are non-decreasing)
    const int tilesX = (sw + 63) / 64, tilesY = (sh + 63) / 64;
    t->tileGx.assign((size_t)tilesX + 1, ngx);
    t->tileDy.assign((size_t)tilesY + 1, dh);
    for (int tx = tilesX - 1, g = ngx - 1; tx >= 0; tx--) {
      while (g >= 0 && (int)t->colrec[12 * (size_t)g + 8] >= 64 * tx) g--;
      t->tileGx[tx] = g + 1;
    }
    for (int ty = tilesY - 1, dy = dh - 1; ty >= 0; ty--) {
      while (dy >= 0 && (int)t->rowrec[4 * (size_t)dy] >= 64 * ty) dy--;
      t->tileDy[ty] = dy + 1;
    }
    bool fits = true;  // the kernel keeps a tile's column records in 16 LDS slots and its row records in 80
    for (int tx = 0; tx < tilesX; tx++) fits = fits && t->tileGx[tx + 1] - t->tileGx[tx] <= 16;
    for (int ty = 0; ty < tilesY; ty++) fits = fits && t->tileDy[ty + 1] - t->tileDy[ty] <= 80;
    if (!fits) { t->tileGx.clear(); t->tileDy.clear(); }
  } else {
    t->tileGx.clear();
    t->tileDy.clear();
  }
}

void FrameGeom::build(const ExtractorTables& t, int W_, int H_) {
  W = W_;
  H = H_;
  nlevels = t.nlevels;
  cells.clear();
  maxCellW = maxCellH = 0;
  uint32_t off = 0;
  int slots = 0, kps = 0;
  for (int l = 0; l < nlevels; l++) {
    LevelGeom& g = lv[l];
    std::memset(&g, 0, sizeof(g));
    const float s = t.invScale[l];
    g.w = cv_round((float)W * s);  // :1208
    g.h = cv_round((float)H * s);
    g.pitch = (g.w + 63) & ~63;
    g.off = off;
    off += (uint32_t)g.pitch * (uint32_t)(g.h > 0 ? g.h : 0);
    off = (off + 255u) & ~255u;
    g.quota = t.quota[l];
    g.cellStart = (int)cells.size();
    g.slotStart = slots;
    // grid, :823-842 (float arithmetic as in the reference)
    const int minBX = kMinBorder, minBY = kMinBorder;
    const int maxBX = g.w - kEdgeThreshold + 3, maxBY = g.h - kEdgeThreshold + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const float Wc = 30;
    g.nCols = (int)(width / Wc);
    g.nRows = (int)(height / Wc);
    if (g.nCols > 0 && g.nRows > 0) {
      g.wCell = (int)std::ceil(width / (float)g.nCols);
      g.hCell = (int)std::ceil(height / (float)g.nRows);
      for (int i = 0; i < g.nRows; i++) {
        const float iniY = (float)(minBY + i * g.hCell);
        float maxY = iniY + g.hCell + 6;
        if (iniY >= maxBY - 3) continue;  // :854
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < g.nCols; j++) {
          const float iniX = (float)(minBX + j * g.wCell);
          float maxX = iniX + g.wCell + 6;
          if (iniX >= maxBX - 6) continue;  // :866
          if (maxX > maxBX) maxX = (float)maxBX;
          // cv::FAST detects on rows/cols [3, size-4] of the sub-image
          const int x0 = (int)iniX + 3, x1 = (int)maxX - 4;
          const int y0 = (int)iniY + 3, y1 = (int)maxY - 4;
          if (x1 < x0 || y1 < y0) continue;
          CellDesc c;
          c.level = (int16_t)l;
          c.x0 = (int16_t)x0;
          c.y0 = (int16_t)y0;
          c.w = (int16_t)(x1 - x0 + 1);
          c.h = (int16_t)(y1 - y0 + 1);
          c.flags = 0;
          c.slotBase = slots;
          // strict 3x3 NMS keeps no two 8-adjacent pixels -> at most ceil(w/2)*ceil(h/2)
          slots += ((c.w + 1) / 2) * ((c.h + 1) / 2);
          cells.push_back(c);
          if (c.w > maxCellW) maxCellW = c.w;
          if (c.h > maxCellH) maxCellH = c.h;
        }
      }
    } else {
      g.nCols = g.nRows = 0;
    }
    g.nCells = (int)cells.size() - g.cellStart;
    g.slotCount = slots - g.slotStart;
    // DistributeOctTree roots, :570
    const int bw = maxBX - minBX, bh = maxBY - minBY;
    g.nIni = (bw > 0 && bh > 0) ? (int)std::round((float)bw / (float)bh) : 0;
    if (g.nIni < 0) g.nIni = 0;
    // per-level output bound: phase 2 stops at >= quota after adding at most 3 per split;
    // the first split pass can create 4*nIni leaves regardless of the quota (Appendix A3).
    int cap = g.quota + 3;
    if (cap < 4 * g.nIni) cap = 4 * g.nIni;
    g.kpStart = kps;
    g.kpCap = cap;
    kps += cap;
    if (l > 0) build_resize_tables(lv[l - 1].w, lv[l - 1].h, g.w, g.h, &rz[l]);
  }
  pyrBytes = off;
  totalSlots = slots;
  totalKpCap = kps;
  // ---- blur-only frame cells (fused FAST+blur kernel): per level, the level rectangle minus the bounding box
  //      of its detection rectangles, cut into strips of cells no larger than 32 x 32 ----
  nFastCells = (int)cells.size();
  fusedBlur = true;
  std::vector<CellDesc> frame;
  for (int l = 0; l < nlevels; l++) {
    const LevelGeom& g = lv[l];
    if (g.w <= 0 || g.h <= 0) continue;
    int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;  // bounding box of the FAST rectangles; empty when the level has none
    long long area = 0;
    for (int c = g.cellStart; c < g.cellStart + g.nCells; c++) {
      const CellDesc& cd = cells[c];
      if (area == 0) { bx0 = cd.x0; bx1 = cd.x0 + cd.w; by0 = cd.y0; by1 = cd.y0 + cd.h; }
      if (cd.x0 < bx0) bx0 = cd.x0;
      if (cd.x0 + cd.w > bx1) bx1 = cd.x0 + cd.w;
      if (cd.y0 < by0) by0 = cd.y0;
      if (cd.y0 + cd.h > by1) by1 = cd.y0 + cd.h;
      area += (long long)cd.w * cd.h;
    }
    // the reference's grid tiles its bounding box exactly (sub-images overlap by 6 px, detection rims by 0);
    // if a geometry ever does not, the unfused kernels are used instead
    if (area != (long long)(bx1 - bx0) * (by1 - by0)) fusedBlur = false;
    auto add_rect = [&](int rx0, int ry0, int rx1, int ry1) {
      for (int y = ry0; y < ry1; y += 32)
        for (int x = rx0; x < rx1; x += 32) {
          CellDesc c;
          c.level = (int16_t)l;
          c.x0 = (int16_t)x;
          c.y0 = (int16_t)y;
          c.w = (int16_t)(rx1 - x < 32 ? rx1 - x : 32);
          c.h = (int16_t)(ry1 - y < 32 ? ry1 - y : 32);
          c.flags = (int16_t)kCellBlurOnly;
          if (c.y0 - 3 < 0 || c.y0 + c.h + 2 >= g.h) c.flags |= kCellRowReflect;
          // the staging loads whole dwords x0-4 .. x0+4*ceil(w/4)+3: every column the blur reads must be in the level
          if (c.x0 - 3 < 0 || c.x0 + c.w + 2 >= g.w) c.flags |= kCellColReflect;
          c.slotBase = 0;
          frame.push_back(c);
          if (c.w > maxCellW) maxCellW = c.w;
          if (c.h > maxCellH) maxCellH = c.h;
        }
    };
    if (area == 0) {
      add_rect(0, 0, g.w, g.h);
    } else {
      add_rect(0, 0, g.w, by0);       // top strip
      add_rect(0, by1, g.w, g.h);     // bottom strip
      add_rect(0, by0, bx0, by1);     // left strip
      add_rect(bx1, by0, g.w, by1);   // right strip
    }
  }
  cells.insert(cells.end(), frame.begin(), frame.end());
}

}  // namespace orbfe
