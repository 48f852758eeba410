// orbx_bf_mfma.h -- the brute-force form of the SearchByBoW top-K pass on the matrix cores.  Included by orbx_matcher.hip.
//
// ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:2075-2089) is a popcount of a xor, and for ONE pair that is all it can be.
// The brute-force pass of a frame against its predecessor, though, is ALL pairs: a 960 x 960 matrix of distances per frame pair, and with
// the descriptor bits written as +-1 bytes (bit set -> +1, clear -> -1)
//      dot(q, t) = #equal bits - #different bits = 256 - 2 * Hamming(q, t)          (exact, in int32)
// that matrix is an int8 GEMM with K = 256 -- v_mfma_i32_32x32x32_i8, 8 per 32 x 32 tile.  The popcount kernel it replaces (one query per
// lane, train rows as LDS broadcasts: 8 xor + 8 v_bcnt_u32_b32 per pair per lane) was the second largest consumer of vector-instruction
// issue in a step (98 of 490 M wave-instructions, 0.19 ms per 256 frame pairs at 60 % of its issue ceiling) while the matrix pipe stood
// idle; here a pair costs 1/128 of an MFMA and one compare, and the matrix pipe works beside the other chunks' vector kernels.
// Results are the same keys as before (distance << 22 | train row, the sorted 8 smallest below the pruning bound per query, and the count
// of candidates below it), so k_group_resolve -- greedy claims, ratio test, rotation histogram (:286-418) -- is untouched.
//
//   workgroup = 8 waves x 32 queries.  B operand (K x 32 queries): the wave's queries as +-1 bytes, 8 k-steps x 16 bytes per lane, in
//   registers for the whole pass.  A operand (32 train rows x K): train rows are expanded 64 at a time into an LDS tile of +-1 bytes
//   (a 256-entry byte -> 8 x +-1 table, also in LDS) and read with ds_read_b128; the next tile's descriptor bytes are already in
//   registers while the current one is multiplied.  C (32 x 32 int32): the column is the lane's query, the 16 registers are train rows:
//   `dot > 256 - 2 * bound` is the only per-pair vector work; the rare hit is parked in LDS and merged into the lane's sorted list of 8
//   as in the popcount kernel.  The two lane halves of a query column (rows 4h .. 4h + 3 of every group of 8) merge their lists at the end.
#ifndef ORBX_BF_MFMA_H_
#define ORBX_BF_MFMA_H_

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

constexpr int kMfRows  = 64;   // train rows per staged step (two 32-row A tiles)
constexpr int kMfPitch = 272;  // bytes per expanded row: 256 + 16, so that the 16 rows one ds_read_b128 group touches start on 16 different banks

constexpr int kMfThreads = 512;  // 8 waves x 32 queries share one expanded train tile

template <bool TVALID>
__global__ __launch_bounds__(kMfThreads) void k_bf_mfma(GroupArgs a0) {
  __shared__ __align__(16) uint8_t tileT[2][kMfRows * kMfPitch];
  __shared__ __align__(8) uint2    lut[256];
  __shared__ uint8_t               tval[2][TVALID ? kMfRows : 1];
  __shared__ uint32_t              hitq[kBfQueue][kMfThreads];  // parked hits, slot-major (conflict-free)
  __shared__ short                 dots[16][kMfThreads];        // the 16 dot products of a lane's half tile, for the walk over its hit mask
  const GroupArgs a   = pair_view(a0, blockIdx.y);
  const int       tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int       nq  = a.d_nq ? *a.d_nq : a.nq;
  const int       nt  = a.d_nt ? *a.d_nt : a.nt;
  if ((int)blockIdx.x * (kMfThreads / 2) >= nq) return;  // whole workgroup
  {  // byte -> eight +-1 bytes (bit i of the byte -> byte i: the k order is the same for queries and train rows, which is all a dot product needs)
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lo |= (((tid >> i) & 1) ? 0x01u : 0xffu) << (8 * i);
      hi |= (((tid >> (i + 4)) & 1) ? 0x01u : 0xffu) << (8 * i);
    }
    if (tid < 256) lut[tid] = make_uint2(lo, hi);
  }
  // ---- staging of 64 train rows: thread = (row, 4 of its 32 bytes); the bytes of the NEXT tile wait in a register
  const int srow = tid >> 3, spart = tid & 7;
  auto fetch = [&](int base) -> uint32_t {
    const int row = base + srow;
    return row < nt ? *reinterpret_cast<const uint32_t*>(a.t + (size_t)row * 32 + spart * 4) : 0u;
  };
  auto expand = [&](int buf, int base, uint32_t w) {
    uint8_t* dst = &tileT[buf][srow * kMfPitch + spart * 32];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint2 e0 = lut[(w >> (16 * i)) & 0xffu], e1 = lut[(w >> (16 * i + 8)) & 0xffu];
      *reinterpret_cast<uint4*>(dst + 16 * i) = make_uint4(e0.x, e0.y, e1.x, e1.y);
    }
    if (TVALID && spart == 0) tval[buf][srow] = (base + srow < nt) ? a.t_valid[base + srow] : 0;
  };
  uint32_t nextBytes = fetch(0);
  __syncthreads();  // the table
  // ---- this wave's 32 queries as the B operand
  const int  q      = blockIdx.x * (kMfThreads / 2) + wave * 32 + r;
  const bool active = q < nq && !(a.q_valid && !a.q_valid[q]);
  mf_v4i     bq[8];
  {
    const uint4* qd = reinterpret_cast<const uint4*>(a.q + (size_t)min(q, nq - 1) * 32);
    const uint4  d0 = qd[0], d1 = qd[1];
    const uint32_t w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int s = 0; s < 8; s++) {  // k-step s: bits 32 s + 16 hh .. + 15 = bytes 2 hh, 2 hh + 1 of dword s
      const uint32_t two = (w[s] >> (16 * hh)) & 0xffffu;
      const uint2    e0 = lut[two & 0xffu], e1 = lut[two >> 8];
      bq[s] = mf_v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
    }
  }
  expand(0, 0, nextBytes);
  uint32_t k[kTopK];
#pragma unroll
  for (int j = 0; j < kTopK; j++) k[j] = kEmpty;
  int                  cnt = 0;
  unsigned char* const hitqBytes = reinterpret_cast<unsigned char*>(&hitq[0][0]);
  int                  dlane     = active ? a.dstar : 0;  // 0: no distance is below it
  bool                 tightened = false;
  uint32_t             qslot     = tid * 4;               // next free parking slot as a byte offset (slot * 4 kMfThreads + tid * 4)
  constexpr uint32_t   kSlotBytes = 4u * kMfThreads;
  const uint32_t       qlimit    = (uint32_t)(kBfQueue - 1) * kSlotBytes + (uint32_t)tid * 4u;  // at or above the last slot: drain before the next hit
  auto flush = [&]() {
    const int qn = (int)(qslot / kSlotBytes);
    for (int i = 0; __ballot(i < qn) != 0ull; i++) {
      uint32_t key = i < qn ? hitq[i][tid] : kEmpty;  // kEmpty falls through the network without touching the list
#pragma unroll
      for (int j = 0; j < kTopK; j++) {
        const uint32_t lo = min(key, k[j]);
        key               = max(key, k[j]);
        k[j]              = lo;
      }
    }
    cnt += qn;
    qslot = tid * 4;
    // Once the list is full, a later hit only enters it with a distance below the list's last one (a lane's rows arrive in ascending
    // order, so an equal distance would sort behind it): the lane's bound shrinks to that distance.  What a lane drops this way cannot be
    // among the 8 smallest of the query either -- those are among the 8 smallest of each half.  Hits skipped are not counted, so a lane that
    // has tightened its bound reports "more candidates than listed" (all the resolve pass asks of nvalid).
    if (k[kTopK - 1] != kEmpty) {
      const int dLast = (int)(k[kTopK - 1] >> kPosBits);
      if (dLast < dlane) dlane = dLast, tightened = true;
    }
    // The other half of the query's rows has its own list: once THAT is full, its last distance bounds this half as well (eight keys
    // at or below it exist; a key of this half with a larger distance sorts behind all of them; equal distances are kept: + 1).
    const uint32_t oLast = (uint32_t)__shfl_xor((int)k[kTopK - 1], 32, 64);
    if (oLast != kEmpty) {
      const int dOther = (int)(oLast >> kPosBits) + 1;
      if (dOther < dlane) dlane = dOther, tightened = true;
    }
  };
  __syncthreads();  // tile 0
  int buf = 0;
  for (int base = 0; base < nt; base += kMfRows) {
    const bool more = base + kMfRows < nt;
    if (more) nextBytes = fetch(base + kMfRows);
    const bool partial = base + kMfRows > nt;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (base + 32 * half >= nt) break;
      mf_v16i        acc = {0};
      const uint8_t* A   = &tileT[buf][(32 * half + r) * kMfPitch + 16 * hh];
#pragma unroll
#if defined(ORBX_MF_VARIANT) && (ORBX_MF_VARIANT & 4)
      for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[(s + 1) & 7], bq[s], acc, 0, 0, 0);  // no LDS operand reads
#else
      for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const mf_v4i*>(A + 32 * s), bq[s], acc, 0, 0, 0);
#endif
      // C: column = this lane's query, register g = train row (g & 3) + 8 (g >> 2) + 4 hh of the tile
      const int thr  = 256 - 2 * dlane;  // Hamming < bound  <=>  dot > 256 - 2 bound
      const int row0 = base + 32 * half + 4 * hh;
#if defined(ORBX_MF_VARIANT) && (ORBX_MF_VARIANT & 1)
      asm volatile("" ::"v"(acc));
      continue;
#endif
      // ~3 % of all pairs of a real stream lie below the pruning bound (30 per query): nearly every one of the 16 registers has a hit in
      // SOME lane, while a lane has one in every other half tile.  Testing and parking register by register would run sixteen divergent
      // blocks for two active lanes each; instead every lane collects a 16-bit hit mask (one compare and one shift-in per register), the
      // dots go to a per-lane LDS scratch as 16-bit values, and the lanes then walk their OWN hits together: the loop runs for the busiest
      // lane's count (2-3), each trip parking one hit per lane.
      uint32_t hm = 0;
#pragma unroll
      for (int g = 15; g >= 0; g--) {
        const int dot = acc[g];
        dots[g][tid] = (short)dot;
        bool hit = dot > thr;
        if (TVALID) hit = hit && tval[buf][32 * half + 4 * hh + (g & 3) + 8 * (g >> 2)];
        hm = (hm << 1) | (hit ? 1u : 0u);
      }
      if (partial) {  // rows past the end of the train set (last tile only)
#pragma unroll
        for (int g = 0; g < 16; g++)
          if (row0 + (g & 3) + 8 * (g >> 2) >= nt) hm &= ~(1u << g);
      }
      while (__ballot(hm != 0u) != 0ull) {
        if (hm != 0u) {
          const int g = __builtin_ctz(hm);
          hm &= hm - 1u;
          const int dot = dots[g][tid], row = row0 + (g & 3) + 8 * (g >> 2);
#if !(defined(ORBX_MF_VARIANT) && (ORBX_MF_VARIANT & 8))
          *reinterpret_cast<uint32_t*>(hitqBytes + qslot) = ((uint32_t)((256 - dot) >> 1) << kPosBits) | (uint32_t)row;
          qslot += kSlotBytes;
#endif
        }
#if defined(ORBX_MF_VARIANT) && (ORBX_MF_VARIANT & 16)
        if (qslot > qlimit) qslot = tid * 4;  // never merge: the queue just wraps
#else
        if (__ballot(qslot > qlimit) != 0ull) flush();
#endif
      }
    }
#if defined(ORBX_MF_VARIANT) && (ORBX_MF_VARIANT & 2)
    asm volatile("" ::"v"(nextBytes));
#else
    if (more) expand(buf ^ 1, base + kMfRows, nextBytes);
#endif
    __syncthreads();
    buf ^= 1;
  }
  flush();
  // ---- the two lane halves of a query column hold the two halves of its rows: merge the lists (lane r takes lane r + 32's)
  {
    uint32_t o[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; j++) o[j] = (uint32_t)__shfl_xor((int)k[j], 32, 64);
    const int  ocnt = __shfl_xor(cnt, 32, 64);
    const bool otig = __shfl_xor((int)tightened, 32, 64) != 0;
#pragma unroll
    for (int i = 0; i < kTopK; i++) {
      uint32_t key = o[i];
#pragma unroll
      for (int j = 0; j < kTopK; j++) {
        const uint32_t lo = min(key, k[j]);
        key               = max(key, k[j]);
        k[j]              = lo;
      }
    }
    cnt += ocnt;
    tightened = tightened || otig;
  }
  if (hh == 0 && q < nq) {
    uint4* o = reinterpret_cast<uint4*>(a.topk + (size_t)q * kTopK);
    o[0]     = make_uint4(k[0], k[1], k[2], k[3]);
    o[1]     = make_uint4(k[4], k[5], k[6], k[7]);
    a.nvalid[q] = tightened ? max(cnt, kTopK + 1) : cnt;
  }
}

#endif
