// orbx_fast_tiles.h -- cv::FAST(cell, kps, iniThFAST, true) with the minThFAST retry for every cell of a frame
// (reference src/ORBextractor.cc:777-849), second-generation kernel.  Included by orbx_kernels.h.
//
// Work unit = a TILE: up to ~6 horizontally adjacent cells of one cell row (host-built table, FastTile), one 256-thread workgroup.
// Per-cell semantics are kept exactly (strict 3x3 NMS inside a cell's 3-px-inset interior only, the iniThFAST / minThFAST choice per
// cell, row-major emission per cell) while the per-workgroup costs -- descriptor decode, staging, four barriers -- are paid once per
// ~6 cells, and the sparse phases (a few hundred candidates per tile instead of ~80 per cell) run with full waves.
//
//   S1  stage the tile (interior + 3 px) into LDS with aligned dwords, zero the score map and the per-cell bit masks
//   S2  dense pre-test, 4 horizontally adjacent pixels per lane from 11 LDS DWORD reads (rows y-3, y+3: 1; y-2, y+2, y: 3 each);
//       the taps are cut out of the dwords with v_perm_b32 straight into packed 16-bit pairs, the opposite-pair test itself is the
//       one of the first-generation kernel (a 9-arc holds one of every two opposite circle points).  Survivor bits stay in a
//       per-lane register mask across all passes and are compacted into an LDS list once per tile
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
};

constexpr int kFastTileThreads = 256;
constexpr int kFastMaxPass     = 16;  // pre-test passes per tile (64 survivor bits per lane)

// Two adjacent bytes I, I + 1 of the 12-byte window {w0, w1, w2} as a packed pair of 16-bit lanes (zero-extended).
template <int I>
__device__ __forceinline__ pk16 pair_at(uint32_t w0, uint32_t w1, uint32_t w2) {
  static_assert(I >= 0 && I <= 10, "window index");
  constexpr int      d = I >> 2, l = I & 3;
  const uint32_t     lo = d == 0 ? w0 : (d == 1 ? w1 : w2), hi = d == 0 ? w1 : w2;  // hi only read when l == 3
  constexpr uint32_t sel = 0x0c000c00u | (uint32_t)l | ((uint32_t)(l + 1) << 16);    // byte l -> lane 0, byte l + 1 -> lane 1 (4 = first byte of hi)
  return pk_from(__builtin_amdgcn_perm(hi, lo, sel));
}

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
  int*           cnt   = reinterpret_cast<int*>(fastLds2 + fl.cntOff);  // [0] candidates, [1 + c] iniTh winners of cell c
  const int      P     = T.ldsPitch;
  const int      lowTh = max(min(iniTh, minTh), 1);

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
    for (int r0 = wave; r0 < T.th; r0 += 4 * kStageRows) {
      uint32_t v[kStageRows];
#pragma unroll
      for (int k = 0; k < kStageRows; k++) {
        const int      r = min(r0 + 4 * k, T.th - 1);
        const uint8_t* p = src + (long long)r * sp;  // uniform row base (scalar unit) + 32-bit lane offset
        if (AL4) {
          v[k] = *reinterpret_cast<const uint32_t*>(p + (unsigned)cb);
        } else {
          p += cb;
          uint32_t u = p[0];
          if (cb + 1 < tw) u |= (uint32_t)p[1] << 8;
          if (cb + 2 < tw) u |= (uint32_t)p[2] << 16;
          if (cb + 3 < tw) u |= (uint32_t)p[3] << 24;
          v[k] = u;
        }
      }
#pragma unroll
      for (int k = 0; k < kStageRows; k++) {
        const int r = r0 + 4 * k;
        if (cin && r < T.th) *reinterpret_cast<uint32_t*>(tile + r * P + col * 4) = v[k];
      }
    }
    // the score map (same shape as the tile, 16-byte aligned, a multiple of 16 bytes long) is cleared with 16-byte stores
    for (int i = tid * 16, n = min(fl.scoreBytes, (T.th * P + 15) & ~15); i < n; i += kFastTileThreads * 16) *reinterpret_cast<uint4*>(score + i) = make_uint4(0, 0, 0, 0);
    const int nmask = T.ncells * T.W;
    for (int i = tid; i < nmask; i += kFastTileThreads) mask1[i] = 0, mask2[i] = 0;
    if (tid <= T.ncells) cnt[tid] = 0;
  }
  __syncthreads();

  // ---- S2: pre-test, item = (interior row, 4-px group)
  unsigned long long surv = 0;  // bit 4 * pass + k: pixel k of this lane's item of that pass survives
  const int          npass = (T.nitems + kFastTileThreads - 1) / kFastTileThreads;
  const pk16         tpk   = pk_pair(lowTh, lowTh);
  int                P2 = 2 * P, P3 = 3 * P;
  asm volatile("" : "+s"(P2), "+s"(P3));  // opaque: keeps (row + 3) * P -+ 3 P from being re-derived with a multiply per row
  const int      glast  = T.g0 + T.NG - 1;
  const uint32_t mFirst = 0xfu & ~((1u << (T.ox3 & 3)) - 1u);                      // pixels of the first group that are interior
  const uint32_t mLast  = (1u << (((T.ox3 + T.IW - 1) & 3) + 1)) - 1u;             // ... of the last group
#ifdef ORBX_TUNING
  if (dbgSkip & 1) return;
#endif
  for (int it = 0; it < npass; it++) {
    const int item = tid + kFastTileThreads * it;
    if (item < T.nitems) {
      const int      row = div_magic(item, T.mNG), gg = item - row * T.NG + T.g0;
      const uint8_t* A   = tile + ((row + 3) * P + gg * 4);
      const uint32_t top = *reinterpret_cast<const uint32_t*>(A - P3), bot = *reinterpret_cast<const uint32_t*>(A + P3);
      const uint32_t m0 = *reinterpret_cast<const uint32_t*>(A - 4), m1 = *reinterpret_cast<const uint32_t*>(A), m2 = *reinterpret_cast<const uint32_t*>(A + 4);
      uint32_t       neg[2];
      const uint32_t u0 = *reinterpret_cast<const uint32_t*>(A - P2 - 4), u1 = *reinterpret_cast<const uint32_t*>(A - P2),
                     u2 = *reinterpret_cast<const uint32_t*>(A - P2 + 4);
      const uint32_t d0 = *reinterpret_cast<const uint32_t*>(A + P2 - 4), d1 = *reinterpret_cast<const uint32_t*>(A + P2),
                     d2 = *reinterpret_cast<const uint32_t*>(A + P2 + 4);
#pragma unroll
      for (int h = 0; h < 2; h++) {  // h = 0: pixels 0, 1 (window bytes 4, 5); h = 1: pixels 2, 3 (bytes 6, 7)
        pk16 v, p[8];
        if (h == 0) {
          v    = pair_at<4>(m0, m1, m2);
          p[0] = pair_at<4>(0, bot, 0), p[4] = pair_at<4>(0, top, 0);              // (0, +3), (0, -3)
          p[1] = pair_at<6>(d0, d1, d2), p[5] = pair_at<2>(u0, u1, u2);            // (+2, +2), (-2, -2)
          p[2] = pair_at<7>(m0, m1, m2), p[6] = pair_at<1>(m0, m1, m2);            // (+3, 0), (-3, 0)
          p[3] = pair_at<6>(u0, u1, u2), p[7] = pair_at<2>(d0, d1, d2);            // (+2, -2), (-2, +2)
        } else {
          v    = pair_at<6>(m0, m1, m2);
          p[0] = pair_at<6>(0, bot, 0), p[4] = pair_at<6>(0, top, 0);
          p[1] = pair_at<8>(d0, d1, d2), p[5] = pair_at<4>(u0, u1, u2);
          p[2] = pair_at<9>(m0, m1, m2), p[6] = pair_at<3>(m0, m1, m2);
          p[3] = pair_at<8>(u0, u1, u2), p[7] = pair_at<4>(d0, d1, d2);
        }
        pk16 dmin = pk_min(p[0], p[4]), bmax = pk_max(p[0], p[4]);
#pragma unroll
        for (int i = 1; i < 4; i++) {
          dmin = pk_max(dmin, pk_min(p[i], p[i + 4]));  // largest  "darker one of an opposite pair"
          bmax = pk_min(bmax, pk_max(p[i], p[i + 4]));  // smallest "brighter one of an opposite pair"
        }
        // passes iff dmin < v - t or bmax > v + t  <=>  max(v - dmin, bmax - v) > t  <=>  t - max(...) < 0
        neg[h] = pk_bits(tpk - pk_max(v - dmin, bmax - v));
      }
      // sign bits (15 / 31 of neg[0], neg[1]) -> 4-bit nibble; the first / last group's pixels outside the interior are masked off
      const uint32_t sb  = __builtin_amdgcn_perm(neg[1], neg[0], 0x07050301u);           // bytes: px0, px1, px2, px3 (bit 7 each)
      uint32_t       nib = (((sb >> 7) & 0x01010101u) * 0x01020408u) >> 24;               // bit k = pixel k (byte k -> bit 24 + k)
      nib &= (gg == T.g0 ? mFirst : 0xfu) & (gg == glast ? mLast : 0xfu);
      surv |= (unsigned long long)nib << (4 * it);
    }
  }
  // compaction: every lane appends its survivors (any order) to the candidate list; entry = row << 9 | tile column.
  // A lane whose entries do not fit keeps them (ovf) and scores them itself later: correct for any survivor count.
  unsigned long long ovf = 0;
  {
    const int n = __popcll(surv);
    if (n) {
      int base = atomicAdd(&cnt[0], n);
      if (base + n <= fl.candCap) {
        unsigned long long m = surv;
        while (m) {
          const int bi = __builtin_ctzll(m);
          m &= m - 1;
          const int item = tid + kFastTileThreads * (bi >> 2);
          const int row = div_magic(item, T.mNG), col = (item - row * T.NG + T.g0) * 4 + (bi & 3);
          cand[base++] = (unsigned short)((row << 9) | col);
        }
      } else {
        for (int k = base; k < min(base + n, fl.candCap); k++) cand[k] = 0xffffu;  // holes
        ovf = surv;
      }
    }
  }
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
  for (int j = tid; 2 * j < ncand; j += kFastTileThreads) {
    unsigned   e0 = cand[2 * j], e1 = 2 * j + 1 < ncand ? cand[2 * j + 1] : 0xffffu;
    const bool v0 = e0 != 0xffffu, v1 = e1 != 0xffffu;
    if (v0 | v1) {
      e0 = v0 ? e0 : e1, e1 = v1 ? e1 : e0;
      const int o0 = ((int)(e0 >> 9) + 3) * P + (int)(e0 & 511u), o1 = ((int)(e1 >> 9) + 3) * P + (int)(e1 & 511u);
      int       s0, s1;
      fast_score2(tile + o0, tile + o1, P, &s0, &s1);
      if (v0) score[o0] = (uint8_t)(s0 >= lowTh ? s0 : 0);  // scores below both thresholds are never corners
      if (v1) score[o1] = (uint8_t)(s1 >= lowTh ? s1 : 0);
    }
  }
  if (__ballot(ovf != 0)) {
    unsigned long long m = ovf;
    while (m) {
      const int bi = __builtin_ctzll(m);
      m &= m - 1;
      const int item = tid + kFastTileThreads * (bi >> 2);
      const int row = div_magic(item, T.mNG), col = (item - row * T.NG + T.g0) * 4 + (bi & 3);
      score_one(row, col);
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
  for (int k = tid; k < ncand; k += kFastTileThreads) {
    const unsigned e = cand[k];
    if (e != 0xffffu) nms_one((int)(e >> 9), (int)(e & 511u));
  }
  if (__ballot(ovf != 0)) {
    unsigned long long m = ovf;
    while (m) {
      const int bi = __builtin_ctzll(m);
      m &= m - 1;
      const int item = tid + kFastTileThreads * (bi >> 2);
      const int row = div_magic(item, T.mNG);
      nms_one(row, (item - row * T.NG + T.g0) * 4 + (bi & 3));
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
