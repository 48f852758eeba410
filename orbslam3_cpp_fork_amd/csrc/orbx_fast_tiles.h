// orbx_fast_tiles.h -- cv::FAST(cell, kps, iniThFAST, true) with the minThFAST retry for every cell of a frame
// (reference src/ORBextractor.cc:777-849), second-generation kernel.  Included by orbx_kernels.h.
//
// Work unit = a TILE: up to ~6 horizontally adjacent cells of one cell row (host-built table, FastTile), one 256-thread workgroup.
// Per-cell semantics are kept exactly (strict 3x3 NMS inside a cell's 3-px-inset interior only, the iniThFAST / minThFAST choice per
// cell, row-major emission per cell) while the per-workgroup costs -- descriptor decode, staging, four barriers -- are paid once per
// ~6 cells, and the sparse phases (a few hundred candidates per tile instead of ~80 per cell) run with full waves.
//
//   S1  stage the tile (interior + 3 px) into LDS with aligned dwords, twice: the pixels themselves (for the exact score) and the pixels
//       halved to 7 bits (byte = pixel >> 1); zero the per-cell bit masks
//   S2  dense pre-test, 4 horizontally adjacent pixels per lane, FOUR PIXELS PER 32-BIT OPERATION on the 7-bit copies: bit 7 of every
//       byte is free, so "tap <= centre - c" is the top bit of (0x80 | centre - c) - tap and "tap >= centre + c" the top bit of
//       tap + (0x80 - centre - c), four pixels at a time and with nothing but v_sub / v_add / v_and / v_or / v_xor / v_lshrrev on VGPR
//       operands -- the instructions gfx950 issues at twice the rate of every VOP3 / VOP3P / SDWA encoding and of any VOP2 with an SGPR
//       or literal operand (csrc/orbx_calib.hip).  With c = (t + 1) >> 1 the halved test is implied by the exact
//       one ("darker than centre - t" / "brighter than centre + t" on an opposite pair of every one of the four compass axes -- a 9-arc
//       holds one of every two opposite circle points), i.e. it passes a superset: an effective threshold of t - 1 or t - 2.  The
//       taps that do not start on a dword are cut out of two aligned ones with v_alignbyte / v_alignbit (six per item).  Survivor bits stay
//       in two per-lane registers over all passes and are compacted into an LDS list once per tile
//   S3  exact corner score (OpenCV cornerScore<16> = max(A, B) - 1) of the listed candidates -> u8 score map
//   S4  strict 3x3 NMS under both thresholds on the score map; winners set a bit (row-major pixel index inside their cell) in the
//       cell's iniTh / minTh mask
//   S5  one wave per cell: pick the mask (iniTh if it has any bit, else minTh, :810-839), prefix-sum the popcounts, emit in bit order
//       = cv::FAST's row-major order.
#ifndef ORBX_FAST_TILES_H_
#define ORBX_FAST_TILES_H_

namespace orbx {

struct FastTile {  // host-built (orbx_extractor::makeGeometry), read with scalar loads; 96 bytes
  int      level, srcOff, srcPitch, tw4;  // srcOff: levels >= 1: byte offset of the tile origin in the frame's pyramid block; level 0: xa | y0 << 16
  int      th, ldsPitch, ox3, IW;         // staged rows, LDS row pitch, tile column of the first interior pixel, interior width of all cells
  int      ih, wCell, iwLast, ncells;     // interior height; cell stride; interior width of the tile's last cell (the others: wCell)
  int      cell0, slotBase, xoff, yoff;   // first cell (per-frame cell arrays), the level's slot base, packed output = (ic + xoff, iy + yoff)
  int      NG, g0, W, nitems;             // 4-px groups covering the interior, first group, bit-mask words per cell, NG * ih
  unsigned mNG, mWc, mIw, mIwLast;        // div_magic dividers: / NG, / wCell, / wCell (cell pixel index -> row), / iwLast
  int      nzero, pad0, pad1, pad2;       // cells right after this tile that the reference skips (their counts are written as 0)
};

struct FastLds2 {  // dynamic LDS plan (bytes); the pixel tile sits at offset 16
  int scoreOff, scoreBytes, candOff, candCap, mask1Off, mask2Off, cntOff, total;
  int t7Off;            // the 7-bit copy of the tile (same pitch and origin as the tile); the score map takes its place after the pre-test
  int listCOff;         // scored corners (score > 0) for the NMS pass, candCap entries
};

constexpr int kFastTileThreads = 256;
constexpr int kFastMaxPass     = 16;  // pre-test passes per tile (64 survivor bits per lane)

template <bool AL4>
__global__ __launch_bounds__(kFastTileThreads) void k_fast_tiles(Bufs b, FrameGeom g, const FastTile* __restrict__ tiles, const int* __restrict__ order,
                                                                  int iniTh, int minTh, FastLds2 fl, int dbgSkip) {  // dbgSkip: phase ablation, only read in -DORBX_TUNING builds
  extern __shared__ __align__(16) uint8_t fastLds2[];
  const int tix = order[blockIdx.x];
  if (tix < 0) return;
  const FastTile T   = tiles[tix];
  const int      f   = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int      wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row addresses and per-cell loops are computed on the scalar unit
  uint8_t*       tile  = fastLds2 + 16;
  uint8_t*       score = fastLds2 + fl.scoreOff;
  unsigned short* cand = reinterpret_cast<unsigned short*>(fastLds2 + fl.candOff);
  uint32_t*      mask1 = reinterpret_cast<uint32_t*>(fastLds2 + fl.mask1Off);
  uint32_t*      mask2 = reinterpret_cast<uint32_t*>(fastLds2 + fl.mask2Off);
  int*           cnt   = reinterpret_cast<int*>(fastLds2 + fl.cntOff);  // [0] candidates, [1 + c] iniTh winners of cell c, [1 + ncells] scored corners
  const int      P     = T.ldsPitch;
  const int      lowTh = max(min(iniTh, minTh), 1);

  uint8_t*        t7    = fastLds2 + fl.t7Off;
  unsigned short* listC = reinterpret_cast<unsigned short*>(fastLds2 + fl.listCOff);
  // Operands of the dense phases live in VGPRs on purpose: a VOP2 with an SGPR or literal source issues at half the rate (orbx_calib.hip)
  uint32_t kH = 0x80808080u, kL7 = 0x7f7f7f7fu, kC = (uint32_t)((lowTh + 1) >> 1) * 0x01010101u;
  asm volatile("" : "+v"(kH), "+v"(kL7), "+v"(kC));

  // ---- S1: stage.  Wave w takes rows w, w + 4, ...; lane = dword column.
  {
    const uint8_t* src;
    int            sp;
    if (T.level == 0) {
      sp  = b.img0Pitch;
      src = b.img0 + (long long)f * b.img0FrameStride + (long long)(T.srcOff >> 16) * sp + (T.srcOff & 0xffff);
    } else {
      sp  = T.srcPitch;
      src = b.pyr + (long long)f * g.pyrBytes + T.srcOff;
    }
    const int  col = lane, tw = (T.ox3 - 3) + T.IW + 6;  // bytes per staged row that belong to the image row
    const bool cin = col < T.tw4;
    const int  cb  = min(col, T.tw4 - 1) * 4;             // byte column (clamped: every lane loads, only staged columns are stored)
    // kStageRows loads in flight per lane (a tile is ~45 rows: one trip of the loop, one exposure to the memory latency)
    constexpr int kStageRows = 12;
    auto load4 = [&](const uint8_t* p, int c0) -> uint32_t {
      if (AL4) return *reinterpret_cast<const uint32_t*>(p + (unsigned)c0);
      p += c0;
      uint32_t u = p[0];
      if (c0 + 1 < tw) u |= (uint32_t)p[1] << 8;
      if (c0 + 2 < tw) u |= (uint32_t)p[2] << 16;
      if (c0 + 3 < tw) u |= (uint32_t)p[3] << 24;
      return u;
    };
    for (int r0 = wave; r0 < T.th; r0 += 4 * kStageRows) {
      uint32_t v[kStageRows];
#pragma unroll
      for (int k = 0; k < kStageRows; k++) {
        const int      r = min(r0 + 4 * k, T.th - 1);
        const uint8_t* p = src + (long long)r * sp;  // uniform row base (scalar unit) + 32-bit lane offset
        v[k] = load4(p, cb);
      }
#pragma unroll
      for (int k = 0; k < kStageRows; k++) {
        const int r = r0 + 4 * k;
        if (cin && r < T.th) {
          const int o = r * P + col * 4;
          *reinterpret_cast<uint32_t*>(tile + o) = v[k];
          *reinterpret_cast<uint32_t*>(t7 + o)   = (v[k] >> 1) & kL7;
        }
      }
    }
    // The dwords of every LDS row behind the staged ones are read as taps of pixels outside the interior (first / last group): they must be
    // 7-bit clean like every other tap, or their borrows / carries would run into the neighbouring, interior, byte.
    {
      const int npad = (P >> 2) - T.tw4;  // 1 or 2 dwords per row
      for (int i = tid; i < T.th * npad; i += kFastTileThreads) {
        const int r = npad == 1 ? i : (i >> 1), k = npad == 1 ? 0 : (i & 1);
        *reinterpret_cast<uint32_t*>(t7 + r * P + (T.tw4 + k) * 4) = 0;
      }
    }
    const int nmask = T.ncells * T.W;
    for (int i = tid; i < nmask; i += kFastTileThreads) mask1[i] = 0, mask2[i] = 0;
    if (tid <= T.ncells + 1) cnt[tid] = 0;
  }
  __syncthreads();

  // ---- S2: pre-test, item = (interior row, 4-px group); see the header comment for the arithmetic
  // Survivor flags: bit 7 of byte k of F = pixel k of the item.  A word takes 8 passes: every pass shifts it right by one and ors the new
  // flags in, so after the n passes of a word the flag of its pass j sits at bit 8 - n + j of byte k.
  uint32_t  sv0 = 0, sv1 = 0;
  const int npass = (T.nitems + kFastTileThreads - 1) / kFastTileThreads;
  const int n0 = min(npass, 8), n1 = npass - n0;
  int       P2 = 2 * P, P3 = 3 * P;
  asm volatile("" : "+s"(P2), "+s"(P3));  // opaque: keeps (row + 3) * P -+ 3 P from being re-derived with a multiply per row
#ifdef ORBX_TUNING
  if (dbgSkip & 1) return;
#endif
  for (int it = 0; it < npass; it++) {
    const int item = tid + kFastTileThreads * it;
    uint32_t  F    = 0;
    if (item < T.nitems) {
      const int      row = div_magic(item, T.mNG), gg = item - row * T.NG + T.g0;
      const uint8_t* A   = t7 + ((row + 3) * P + gg * 4);
      const uint32_t V = *reinterpret_cast<const uint32_t*>(A), Wm = *reinterpret_cast<const uint32_t*>(A - 4), Wp = *reinterpret_cast<const uint32_t*>(A + 4);
      const uint32_t tN = *reinterpret_cast<const uint32_t*>(A - P3), tS = *reinterpret_cast<const uint32_t*>(A + P3);
      const uint32_t u0 = *reinterpret_cast<const uint32_t*>(A - P2 - 4), u1 = *reinterpret_cast<const uint32_t*>(A - P2), u2 = *reinterpret_cast<const uint32_t*>(A - P2 + 4);
      const uint32_t d0 = *reinterpret_cast<const uint32_t*>(A + P2 - 4), d1 = *reinterpret_cast<const uint32_t*>(A + P2), d2 = *reinterpret_cast<const uint32_t*>(A + P2 + 4);
      const uint32_t tNW = __builtin_amdgcn_alignbit(u1, u0, 16), tNE = __builtin_amdgcn_alignbit(u2, u1, 16);  // pixels x - 2 .. x + 1, x + 2 .. x + 5
      const uint32_t tSW = __builtin_amdgcn_alignbit(d1, d0, 16), tSE = __builtin_amdgcn_alignbit(d2, d1, 16);
      const uint32_t tW = __builtin_amdgcn_alignbyte(V, Wm, 1);   // pixels x - 3 .. x
      const uint32_t tE = __builtin_amdgcn_alignbyte(Wp, V, 3);   // pixels x + 3 .. x + 6
      // per pixel: D = 0x80 | (q - c), or 0x7f when q < c (then no tap can be "darker"): D - tap keeps bit 7 iff tap <= q - c, never borrows
      const uint32_t X  = (V | kH) - kC;
      const uint32_t nX = (X & kH) ^ kH;                    // 0x80 in the bytes with q < c
      const uint32_t D  = X | (nX - (nX >> 7));
      // the same on the complemented centre gives K = 0x80 - min(q + c, 0x80): tap + K reaches bit 7 iff tap >= q + c, never carries
      const uint32_t Y  = ((V ^ kL7) | kH) - kC;
      const uint32_t nY = (Y & kH) ^ kH;
      const uint32_t K  = (Y | (nY - (nY >> 7))) - kL7;
      const uint32_t dk = ((D - tN) | (D - tS)) & ((D - tE) | (D - tW)) & ((D - tNE) | (D - tSW)) & ((D - tNW) | (D - tSE));
      const uint32_t br = ((tN + K) | (tS + K)) & ((tE + K) | (tW + K)) & ((tNE + K) | (tSW + K)) & ((tNW + K) | (tSE + K));
      F = (dk | br) & kH;
    }
    if (it < 8)
      sv0 = (sv0 >> 1) | F;
    else
      sv1 = (sv1 >> 1) | F;
  }
#ifdef ORBX_TUNING
  if (dbgSkip & 8) {  // the dense loop without the compaction
    asm volatile("" ::"v"(sv0), "v"(sv1));
    return;
  }
#endif
  // A survivor is listed as pass << 10 | thread << 2 | pixel of the item (what the appending lane knows without arithmetic); whoever picks the
  // entry up decodes it to (interior row, tile column): columns outside the interior (first / last group) decode to col < 0.
  auto decode = [&](unsigned e, int* row, int* col) {
    const int item = (int)((e >> 2) & 255u) + kFastTileThreads * (int)(e >> 10);
    const int r = div_magic(item, T.mNG), c = (item - r * T.NG + T.g0) * 4 + (int)(e & 3u);
    *row = r;
    *col = (c >= T.ox3 && c < T.ox3 + T.IW) ? c : -1;
  };
  // survivor bit bi of word w -> list entry (the flag of pass j of a word with n passes sits at bit 8 - n + j of its pixel's byte)
  const int eb0 = ((n0 - 8) << 10) | (tid << 2), eb1 = ((n1 - 8 + 8) << 10) | (tid << 2);
  auto      entry = [&](int w, int bi) -> unsigned { return (unsigned)((w ? eb1 : eb0) + ((bi & 7) << 10) + (bi >> 3)); };
  // compaction: every lane appends its survivors (any order) to the candidate list.
  // A lane whose entries do not fit keeps them (ovf) and scores them itself later: correct for any survivor count.
  uint32_t ovf0 = 0, ovf1 = 0;
  {
    const int n = __popc(sv0) + __popc(sv1);
    // one LDS atomic per wave (64 lanes adding to one address are served one after the other): exclusive scan of the lanes' counts first
    int inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    const int tot = __shfl(inc, 63, 64);
    int       wbase = 0;
    if (tot) {
      if (lane == 0) wbase = atomicAdd(&cnt[0], tot);
      wbase = __shfl(wbase, 0, 64);
    }
    if (n) {
      int base = wbase + inc - n;
      if (base + n <= fl.candCap) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
          uint32_t m = w ? sv1 : sv0;
          while (m) {
            const int bi = __builtin_ctz(m);
            m &= m - 1;
            cand[base++] = (unsigned short)entry(w, bi);
          }
        }
      } else {
        for (int k = base; k < min(base + n, fl.candCap); k++) cand[k] = 0xffffu;  // holes
        ovf0 = sv0, ovf1 = sv1;
      }
    }
  }
  __syncthreads();
  // The 7-bit copy is dead from here on: the score map takes its place (same shape as the tile, 16-byte aligned, a multiple of 16 bytes
  // long, cleared with 16-byte stores) -- LDS per workgroup decides how many tiles a CU holds, and with them how much of one tile's
  // staging latency the others' arithmetic hides.
  for (int i = tid * 16, n = min(fl.scoreBytes, (T.th * P + 15) & ~15); i < n; i += kFastTileThreads * 16) *reinterpret_cast<uint4*>(score + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- S3: exact score of the candidates
#ifdef ORBX_TUNING
  if (dbgSkip & 2) return;
#endif
  const int ncand = min(cnt[0], fl.candCap);
  auto      score_one = [&](int row, int col) {
    const int o = (row + 3) * P + col;
    int       s0, s1;
    fast_score2(tile + o, tile + o, P, &s0, &s1);
    score[o] = (uint8_t)(s0 >= lowTh ? s0 : 0);  // scores below both thresholds are never corners
  };
  // two listed candidates per lane: the two-pixel form of the score (both pixels' margins in the halves of one register, a min chain
  // and a max chain) costs ~96 vector instructions per pixel against ~134 for the one-pixel, both-polarities form
  for (int j0 = 0; 2 * j0 < ncand; j0 += kFastTileThreads) {  // uniform trip count: the corner list is appended with wave ballots
    const int  j  = j0 + tid;
    const unsigned l0 = 2 * j < ncand ? cand[2 * j] : 0xffffu, l1 = 2 * j + 1 < ncand ? cand[2 * j + 1] : 0xffffu;
    int            r0 = 0, x0 = -1, r1 = 0, x1 = -1;
    if (l0 != 0xffffu) decode(l0, &r0, &x0);
    if (l1 != 0xffffu) decode(l1, &r1, &x1);
    const bool v0 = x0 >= 0, v1 = x1 >= 0;
    bool       c0 = false, c1 = false;
    unsigned   e0 = (unsigned)((r0 << 9) | x0), e1 = (unsigned)((r1 << 9) | x1);  // (row, column) form, as the corner list keeps it
    if (v0 | v1) {
      e0 = v0 ? e0 : e1, e1 = v1 ? e1 : e0;
      const int o0 = ((int)(e0 >> 9) + 3) * P + (int)(e0 & 511u), o1 = ((int)(e1 >> 9) + 3) * P + (int)(e1 & 511u);
      int       s0, s1;
      fast_score2(tile + o0, tile + o1, P, &s0, &s1);
      s0 = (v0 && s0 >= lowTh) ? s0 : 0, s1 = (v1 && s1 >= lowTh) ? s1 : 0;  // scores below both thresholds are never corners
      if (v0) score[o0] = (uint8_t)s0;
      if (v1) score[o1] = (uint8_t)s1;
      c0 = s0 > 0, c1 = s1 > 0;
    }
    // the corners (~40 % of the candidates) go to their own list: the NMS pass then runs with full waves
    const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1);
    if (b0 | b1) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&cnt[1 + T.ncells], __popcll(b0) + __popcll(b1));
      base = __shfl(base, 0, 64);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (c0) listC[base + __popcll(b0 & below)] = (unsigned short)e0;
      if (c1) listC[base + __popcll(b0) + __popcll(b1 & below)] = (unsigned short)e1;
    }
  }
  if (__ballot((ovf0 | ovf1) != 0)) {
#pragma unroll
    for (int w = 0; w < 2; w++) {
      uint32_t m = w ? ovf1 : ovf0;
      while (m) {
        const int bi = __builtin_ctz(m);
        m &= m - 1;
        int row, col;
        decode(entry(w, bi), &row, &col);
        if (col >= 0) score_one(row, col);
      }
    }
  }
  __syncthreads();

#ifdef ORBX_TUNING
  if (dbgSkip & 4) return;
#endif
  // ---- S4: strict 3x3 NMS under both thresholds inside the candidate's own cell
  auto nms_one = [&](int row, int col) {
    const int o = (row + 3) * P + col;
    const int s = score[o];
    if (s == 0) return;
    const int ic = col - T.ox3;                 // interior column inside the tile
    const int c  = div_magic(ic, T.mWc);        // cell
    const int cx = ic - c * T.wCell;            // column inside the cell's interior
    const int iw = c == T.ncells - 1 ? T.iwLast : T.wCell;
    int       t[8];
    t[0] = score[o - P - 1], t[1] = score[o - P], t[2] = score[o - P + 1];
    t[3] = score[o - 1], t[4] = score[o + 1];
    t[5] = score[o + P - 1], t[6] = score[o + P], t[7] = score[o + P + 1];
    // neighbours in the next cell are not part of this cell's sub-image: they count as 0 (rows above / below the interior
    // are never scored, so they already are)
    if (cx == 0) t[0] = t[3] = t[5] = 0;
    if (cx == iw - 1) t[2] = t[4] = t[7] = 0;
    // the largest neighbour under a threshold th is max{t : t >= th} = (M >= th ? M : 0) with M the largest neighbour overall: one
    // max over the eight instead of one per threshold (minThFAST may exceed iniThFAST: the reference does not check, nor does this)
    const int M  = max(max(max(t[0], t[1]), max(t[2], t[3])), max(max(t[4], t[5]), max(t[6], t[7])));
    const int m1 = M >= iniTh ? M : 0, m2 = M >= minTh ? M : 0;
    const bool f1 = s >= iniTh && s > m1, f2 = s >= minTh && s > m2;
    if (f1 | f2) {
      const int      idx = row * iw + cx;
      const uint32_t bit = 1u << (idx & 31);
      const int      w   = c * T.W + (idx >> 5);
      if (f1) {
        atomicOr(&mask1[w], bit);
        atomicAdd(&cnt[1 + c], 1);
      }
      if (f2) atomicOr(&mask2[w], bit);
    }
  };
  const int ncorn = cnt[1 + T.ncells];  // <= ncand <= candCap
  for (int k = tid; k < ncorn; k += kFastTileThreads) {
    const unsigned e = listC[k];
    nms_one((int)(e >> 9), (int)(e & 511u));
  }
  if (__ballot((ovf0 | ovf1) != 0)) {
#pragma unroll
    for (int w = 0; w < 2; w++) {
      uint32_t m = w ? ovf1 : ovf0;
      while (m) {
        const int bi = __builtin_ctz(m);
        m &= m - 1;
        int row, col;
        decode(entry(w, bi), &row, &col);
        if (col >= 0) nms_one(row, col);
      }
    }
  }
  __syncthreads();

  // ---- S5: one wave per cell, emission in bit order = row-major inside the cell
  uint32_t* slotsF = b.slots + (long long)f * g.totalSlots + T.slotBase;
  int*      countF = b.cellCount + (long long)f * g.totalCells;
  for (int c = wave; c < T.ncells; c += kFastTileThreads / 64) {
    const uint32_t* mk   = (cnt[1 + c] > 0 ? mask1 : mask2) + c * T.W;  // vKeysCell.empty() at iniThFAST -> minThFAST (:810-839)
    const int       iw   = c == T.ncells - 1 ? T.iwLast : T.wCell;
    const unsigned  mIw  = c == T.ncells - 1 ? T.mIwLast : T.mIw;
    uint32_t*       out  = slotsF + b.cellSlot[T.cell0 + c];
    int             run  = 0;  // keypoints emitted by earlier 64-word chunks
    for (int w0 = 0; w0 < T.W; w0 += 64) {
      uint32_t word = (w0 + lane < T.W) ? mk[w0 + lane] : 0u;
      const int n   = __popc(word);
      int       inc = n;  // inclusive prefix sum over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      int rank = run + inc - n;
      while (word) {
        const int bi = __builtin_ctz(word);
        word &= word - 1;
        const int idx = (w0 + lane) * 32 + bi;
        const int y = div_magic(idx, mIw), x = idx - y * iw;
        const int ic = c * T.wCell + x;
        out[rank++]  = (uint32_t)(ic + T.xoff) | ((uint32_t)(y + T.yoff) << 12) | ((uint32_t)score[(y + 3) * P + T.ox3 + ic] << 24);
      }
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) countF[T.cell0 + c] = run;
  }
  if (tid < T.nzero) countF[T.cell0 + T.ncells + tid] = 0;
}

}  // namespace orbx
#endif
