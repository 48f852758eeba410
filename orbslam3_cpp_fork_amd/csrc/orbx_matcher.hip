// orbx_matcher.hip -- Hamming matcher kernels + C ABI (include/orbx.h, orbm_*) on gfx950.
//
// Every ORBmatcher search bottoms out in: for a query descriptor scan a candidate list, keep best and
// second-best Hamming distance, threshold + ratio test, then a rotation-histogram vote
// (reference src/ORBmatcher.cc:286-360).  Two things make that loop order-dependent in the reference:
//   * SearchByBoW skips train features already claimed by an earlier query (:290-292 / :837),
//   * SearchForInitialization skips candidates whose recorded distance is not worse (:713) and
//     re-assigns train features (:728-735).
// The GPU form keeps the semantics exactly while doing the distance work in parallel:
//   1. k_*_topk: one wave per query computes all distances of its candidate list (64 lanes x
//      popcount of 256-bit XOR) and keeps the K smallest (distance, scan position) keys -- the order the
//      reference's strict `<` updates induce.  Claims only ever *remove* candidates, so the first two
//      surviving entries of that sorted list are the reference's best / second-best ...
//   2. k_*_resolve: one wave walks the queries in reference order with the claim state in LDS, consults
//      the K-lists, and falls back to a wave-parallel rescan only when a truncated list runs dry.
// No MFMA: this is xor + popcount + integer compare (VALU), inputs are tens of KB.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbx.h"

namespace {

constexpr int      kTopK       = 8;      // merged list depth handed to the resolve pass
constexpr int      kLaneK      = 4;      // per-lane depth kept while scanning
constexpr uint32_t kEmpty      = 0xffffffffu;
constexpr int      kPosBits    = 22;
constexpr uint32_t kPosMask    = (1u << kPosBits) - 1u;
constexpr int      kHistoLen   = 30;     // ORBmatcher::HISTO_LENGTH (src/ORBmatcher.cc:32)
constexpr int      kMaxTrain   = 65536;  // claim bitmap in LDS
constexpr int      kMaxTrainW  = 16384;  // vMatchedDistance (u16) in LDS
constexpr int      kGridCols = 64, kGridRows = 48;  // FRAME_GRID_COLS / ROWS (include/Frame.h:36-37)

#define HIP_TRYM(expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      set_err(std::string(#expr) + ": " + hipGetErrorString(e_));                                       \
      return -3;                                                                                        \
    }                                                                                                   \
  } while (0)

thread_local std::string g_merr;
extern "C" int           orbx_internal_runtime_down(void);
extern "C" void          orbx_internal_register(void* obj, void (*release_device)(void*));
extern "C" void          orbx_internal_unregister(void* obj);
extern "C" void          orbx_internal_set_error(const char* s);  // orbx_extractor.hip: the slot orbx_last_error() reads
void                     set_err(const std::string& s) {
  g_merr = s;
  orbx_internal_set_error(s.c_str());  // one message slot per thread for the whole library (include/orbx.h)
}

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint8_t* __restrict__ p) {
  const uint4 b0 = *reinterpret_cast<const uint4*>(p);
  const uint4 b1 = *reinterpret_cast<const uint4*>(p + 16);
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ void topk_insert(uint32_t (&k)[kLaneK], uint32_t key) {
#pragma unroll
  for (int j = 0; j < kLaneK; j++) {
    if (key < k[j]) {
      const uint32_t t = k[j];
      k[j]             = key;
      key              = t;
    }
  }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Scan positions [lo, hi) with 64 lanes; cand(pos) returns the train row or -1 (filtered);
// dist_ok(row, dist) is a second, distance-dependent filter.  Every lane keeps its kLaneK smallest keys; the wave
// then pops the global minimum up to kTopK times.  The popped sequence is the exact sorted prefix of ALL
// candidates as long as no lane that saw more than kLaneK candidates has run out of kept keys -- the merge stops
// right there, so `out` is always an exact prefix (unused slots = kEmpty) and *nvalid tells whether it is complete.
template <class CandFn, class DistFn>
__device__ __forceinline__ void wave_topk(const uint8_t* __restrict__ qd, const uint8_t* __restrict__ tdesc, int lo, int hi, CandFn cand,
                                          DistFn dist_ok, uint32_t (&out)[kTopK], int* nvalid) {
  const int   lane = threadIdx.x & 63;
  const uint4 a0 = *reinterpret_cast<const uint4*>(qd), a1 = *reinterpret_cast<const uint4*>(qd + 16);
  uint32_t    k[kLaneK];
#pragma unroll
  for (int j = 0; j < kLaneK; j++) k[j] = kEmpty;
  int cnt = 0;
  for (int pos = lo + lane; pos < hi; pos += 64) {
    const int row = cand(pos);
    if (row < 0) continue;
    const int d = hamming256(a0, a1, tdesc + (size_t)row * 32);
    if (!dist_ok(row, d)) continue;
    cnt++;
    topk_insert(k, ((uint32_t)d << kPosBits) | (uint32_t)pos);
  }
  bool stop = false;
#pragma unroll
  for (int j = 0; j < kTopK; j++) {
    uint32_t m = kEmpty;
    if (!stop) m = wave_min_u32(k[0]);
    out[j]        = m;
    bool exhausted = false;
    if (m != kEmpty && k[0] == m) {  // positions are unique: exactly one lane pops
#pragma unroll
      for (int t = 0; t + 1 < kLaneK; t++) k[t] = k[t + 1];
      k[kLaneK - 1] = kEmpty;
      exhausted     = k[0] == kEmpty && cnt > kLaneK;  // this lane dropped candidates: what follows is unknown
    }
    stop = stop || m == kEmpty || __ballot(exhausted) != 0ull;
  }
  *nvalid = wave_sum_i32(cnt);
}

__device__ __forceinline__ int rot_bin(float a1, float a2) {  // src/ORBmatcher.cc:348-357
  const float factor = 1.0f / kHistoLen;
  float       rot    = a1 - a2;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == kHistoLen) bin = 0;
  return bin;
}

__device__ __forceinline__ void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3) {  // :2038-2071
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = histo[i];
    if (s > max1) {
      max3 = max2, max2 = max1, max1 = s;
      ind3 = ind2, ind2 = ind1, ind1 = i;
    } else if (s > max2) {
      max3 = max2, max2 = s;
      ind3 = ind2, ind2 = i;
    } else if (s > max3) {
      max3 = s, ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1, ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

// ------------------------------------------------------------------------------------------------
// SearchByBoW over flattened groups
// ------------------------------------------------------------------------------------------------
struct GroupArgs {
  const uint8_t *q, *t;
  const int *    d_nq, *d_nt;  // optional device-side counts (brute-force stream form); else nq / nt
  int            nq, nt, ngroups, nqe;  // nqe = entries of the query lists (== nq for brute force)
  const int *    q_off, *q_idx, *t_off, *t_idx;  // NULL q_idx/t_idx = identity (ngroups == 1)
  const uint8_t *q_valid, *t_valid;
  const float *  q_angle, *t_angle;
  int            q_astride, t_astride;  // in floats
  int            th, strict_lt, check_ori;
  float          ratio;
  int            dstar;      // pruning bound: candidates at distance >= dstar cannot change any decision (see prune_bound)
  uint32_t*      topk;       // [nqe][kTopK]
  int*           nvalid;     // [nqe] candidates seen by the top-k pass
  int*           binOf;      // [nq]
  int *          match_t_of_q, *match_q_of_t, *nmatches;
  // batch form: blockIdx.y = pair p; pairs[2p] / pairs[2p+1] = query / train frame of a
  // [frames][cap] x 32 B descriptor block (orbx_extract_batch_device layout); NULL = single pair
  const int* pairs;
  const int* counts;  // [frames][2]: n, monoIndex
  int        cap;
  const int* t_hi;    // optional end of every group's candidate range (else t_off[g + 1])
  // batch form fed by orbv_transform_batch_device (FeatureVectors of the frames, [frames][cap (+1)]): the query order and the
  // groups of pair p come from frame fq's FeatureVector, the candidates of group g from frame ft's range grpLo/grpHi[p][g]
  const uint32_t* fvFeat;
  const int *     fvOff, *fvNNodes, *grpLo, *grpHi;
  int*       dbg;  // optional tuning counters: [0] speculative rounds, [1] exact rescans, [2] 64-query chunks
};

// Per-pair view of the batch form.
__device__ __forceinline__ GroupArgs pair_view(const GroupArgs& a, int p) {
  if (!a.pairs) return a;
  GroupArgs v = a;
  const int fq = a.pairs[2 * p], ft = a.pairs[2 * p + 1];
  v.q       = a.q + (size_t)fq * a.cap * 32;
  v.t       = a.t + (size_t)ft * a.cap * 32;
  v.d_nq    = a.counts + 2 * fq;
  v.d_nt    = a.counts + 2 * ft;
  v.q_angle = a.q_angle ? a.q_angle + (size_t)fq * a.cap * a.q_astride : nullptr;
  v.t_angle = a.t_angle ? a.t_angle + (size_t)ft * a.cap * a.t_astride : nullptr;
  v.topk    = a.topk + (size_t)p * a.cap * kTopK;
  v.nvalid  = a.nvalid + (size_t)p * a.cap;
  v.binOf   = a.binOf + (size_t)p * a.cap;
  v.match_t_of_q = a.match_t_of_q + (size_t)p * a.cap;
  v.match_q_of_t = a.match_q_of_t + (size_t)p * a.cap;
  v.nmatches     = a.nmatches + p;
  if (a.fvFeat) {  // SearchByBoW proper: merge walk of the two FeatureVectors (src/ORBmatcher.cc:256-268)
    v.q_idx   = reinterpret_cast<const int*>(a.fvFeat + (size_t)fq * a.cap);
    v.q_off   = a.fvOff + (size_t)fq * (a.cap + 1);
    v.ngroups = a.fvNNodes[fq];
    v.nqe     = v.q_off[v.ngroups];
    v.t_idx   = reinterpret_cast<const int*>(a.fvFeat + (size_t)ft * a.cap);
    v.t_off   = a.grpLo + (size_t)p * a.cap;
    v.t_hi    = a.grpHi + (size_t)p * a.cap;
  }
  return v;
}

__device__ __forceinline__ int group_hi(const GroupArgs& a, int g) { return a.t_hi ? a.t_hi[g] : a.t_off[g + 1]; }

__device__ __forceinline__ int group_of(const GroupArgs& a, int e) {
  int lo = 0, hi = a.ngroups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.q_off[mid] <= e)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_group_topk(GroupArgs a0) {
  const GroupArgs a = pair_view(a0, blockIdx.y);
  const int       e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nq  = a.d_nq ? *a.d_nq : a.nq;
  const int nt  = a.d_nt ? *a.d_nt : a.nt;
  const int nqe = a.q_idx ? a.nqe : nq;
  if (e >= nqe) return;
  int g = 0, lo = 0, hi = nt;
  if (a.q_idx) {
    g  = group_of(a, e);
    lo = a.t_off[g], hi = group_hi(a, g);
  }
  const int rq = a.q_idx ? a.q_idx[e] : e;
  uint32_t  out[kTopK];
  int       nv = 0;
  if (a.q_valid && !a.q_valid[rq]) {
#pragma unroll
    for (int j = 0; j < kTopK; j++) out[j] = kEmpty;
  } else {
    const int*     t_idx   = a.t_idx;
    const uint8_t* t_valid = a.t_valid;
    wave_topk(
        a.q + (size_t)rq * 32, a.t, lo, hi,
        [t_idx, t_valid](int pos) {
          const int row = t_idx ? t_idx[pos] : pos;
          return (t_valid && !t_valid[row]) ? -1 : row;
        },
        [dstar = a.dstar](int, int d) { return d < dstar; }, out, &nv);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < kTopK; j++) a.topk[(size_t)e * kTopK + j] = out[j];
    a.nvalid[e] = nv;  // candidates below the pruning bound
  }
}

// Brute-force form of the top-K pass (one group, identity index lists -- the per-frame stream case): orbx_bf_mfma.h (the distance matrix
// of a frame pair as an int8 GEMM on the matrix cores).  k_group_topk (one wave per query, lanes stride the candidates, popcounts) stays
// for the grouped DBoW2 form, whose candidate lists are short and differ per query.
constexpr int kBfQueue = 8;   // hits a lane may park between two merges

#include "orbx_bf_mfma.h"

// Ordered claim resolution.  The reference walks the queries one by one and every accepted match removes its
// train row from all later queries' candidate sets (src/ORBmatcher.cc:290-292, 337).  A query's outcome can
// therefore only be changed by an EARLIER query that claims a row of its K-list.  The kernel takes 64 queries
// at a time and runs speculative rounds:
//   1. every unresolved lane evaluates its own query against the committed claim bitmap;
//   2. accepting lanes post their lane id on their train row (atomicMin into a small hashed table), so that
//      every row knows the EARLIEST lane proposing it in this round;
//   3. a lane is "touched" if an earlier lane proposes a row its decision rests on: its own best row, or -- for a
//      lane that currently fails the ratio test -- its second-best row (losing that row can only RAISE the
//      second distance, which never turns an accept into a reject, so accepting lanes ignore it);
//   4. the untouched prefix (in query order) is committed -- the first unresolved lane always is, its
//      evaluation being exact -- and the rest re-evaluate in the next round.
// Hash collisions only add false "touched" flags (an extra round), never a wrong commit.
// The rare exact rescan (truncated K-list that cannot be decided by the dK bound) is done by the whole wave
// for the first unresolved lane only.
constexpr int kResolveThreads = 256;
constexpr int kSuper          = 512;   // queries staged in LDS per super-chunk
constexpr int kLaneTab        = 4096;  // proposal table entries (hash = row mod 4096: exact below 4096 train rows)

__global__ __launch_bounds__(kResolveThreads) void k_group_resolve(GroupArgs a0) {
  const GroupArgs     a = pair_view(a0, blockIdx.x);
  __shared__ uint32_t claimed[kMaxTrain / 32];
  __shared__ uint32_t minLane[kLaneTab];      // earliest proposing lane per (hashed) train row in the current round
  __shared__ uint32_t sList[kSuper][kTopK];  // dist << 16 | train row  (0xffffffff = empty)
  __shared__ int      sBestT[kSuper];        // committed train row per staged query (-1 = none)
  __shared__ uint8_t  sTrunc[kSuper];
  __shared__ int      hist[kHistoLen];
  __shared__ int      sNm;
  const int           tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int           nq  = a.d_nq ? *a.d_nq : a.nq;
  const int           nt  = a.d_nt ? *a.d_nt : a.nt;
  const int           nqe = a.q_idx ? a.nqe : nq;
  for (int i = tid; i < (nt + 31) / 32; i += kResolveThreads) claimed[i] = 0;
  for (int i = tid; i < kLaneTab; i += kResolveThreads) minLane[i] = 0xffu;
  for (int i = tid; i < kHistoLen; i += kResolveThreads) hist[i] = 0;
  for (int i = tid; i < nq; i += kResolveThreads) a.match_t_of_q[i] = -1;
  for (int i = tid; i < nt; i += kResolveThreads) a.match_q_of_t[i] = -1;
  if (tid == 0) sNm = 0;
  __syncthreads();
  for (int s0 = 0; s0 < nqe; s0 += kSuper) {
    const int ns = min(kSuper, nqe - s0);
    // ---- stage: all waves fetch the K-lists of this super-chunk (one round of global latency for 1024 queries)
    for (int i = tid; i < ns; i += kResolveThreads) {
      const int   e   = s0 + i;
      const uint4 kk0 = *reinterpret_cast<const uint4*>(a.topk + (size_t)e * kTopK);
      const uint4 kk1 = *reinterpret_cast<const uint4*>(a.topk + (size_t)e * kTopK + 4);
      const uint32_t ks[kTopK] = {kk0.x, kk0.y, kk0.z, kk0.w, kk1.x, kk1.y, kk1.z, kk1.w};
      int            len       = 0;
#pragma unroll
      for (int j = 0; j < kTopK; j++) {
        uint32_t v = kEmpty;
        if (ks[j] != kEmpty) {
          const int p   = (int)(ks[j] & kPosMask);
          const int row = a.t_idx ? a.t_idx[p] : p;
          v             = ((ks[j] >> kPosBits) << 16) | (uint32_t)row;
          len           = j + 1;
        }
        sList[i][j] = v;
      }
      sTrunc[i] = a.nvalid[e] > len;  // candidates exist beyond the (exact) list prefix
      sBestT[i] = -1;
    }
    __syncthreads();
    // ---- resolve: wave 0 walks the staged queries 64 at a time in speculative rounds (see above), LDS only
    if (wave == 0) {
      int nmatches = 0;
      for (int c0 = 0; c0 < ns; c0 += 64) {
        const int i = c0 + lane;
        uint32_t  ks[kTopK];
#pragma unroll
        for (int j = 0; j < kTopK; j++) ks[j] = i < ns ? sList[i][j] : kEmpty;
        const bool truncated = i < ns ? sTrunc[i] != 0 : false;
        bool       pos       = false;
        if (ks[0] != kEmpty) {
          const int d = (int)(ks[0] >> 16);
          pos         = a.strict_lt ? d < a.th : d <= a.th;
        }
        int dK = 0;  // distance of the last list entry: lower bound for every candidate outside the list
#pragma unroll
        for (int j = 0; j < kTopK; j++)
          if (ks[j] != kEmpty) dK = (int)(ks[j] >> 16);
        int                myBestT   = -1;
        unsigned long long remaining = __ballot(pos);
        if (a.dbg && lane == 0) atomicAdd(&a.dbg[2], 1);
        while (remaining) {
          if (a.dbg && lane == 0) atomicAdd(&a.dbg[0], 1);
          const bool active = (remaining >> lane) & 1ull;
          int        best = 256, second = 256, bestT = -1, secondT = -1, found = 0;
          bool       rescan = false, accept = false, passBest = false;
          if (active) {
            bool free_[kTopK];  // independent LDS reads first, the ordered pick afterwards
#pragma unroll
            for (int j = 0; j < kTopK; j++) {
              const int row = (int)(ks[j] & 0xffffu);
              free_[j]      = ks[j] != kEmpty && !(claimed[row >> 5] & (1u << (row & 31)));
            }
#pragma unroll
            for (int j = 0; j < kTopK; j++) {
              if (found < 2 && free_[j]) {
                const int row = (int)(ks[j] & 0xffffu);
                const int d   = (int)(ks[j] >> 16);
                if (found == 0)
                  best = d, bestT = row;
                else
                  second = d, secondT = row;
                found++;
              }
            }
            // a truncated K-list that ran dry still bounds what lies beyond it: every other candidate is >= dK away
            passBest = a.strict_lt ? best < a.th : best <= a.th;
            if (found < 2 && truncated) {
              if (found == 1) {
                if (passBest && !((float)best < a.ratio * (float)dK)) rescan = true;
                if (passBest && !rescan) second = dK;  // any value >= dK gives the same decision
              } else {
                rescan = a.strict_lt ? dK < a.th : dK <= a.th;
              }
            }
            if (!rescan) accept = passBest && (float)best < a.ratio * (float)second;
          }
          const int first = __ffsll((long long)remaining) - 1;
          if (__shfl((int)rescan, first, 64)) {
            // exact rescan of the first unresolved query's group with the claim filter (whole wave)
            if (a.dbg && lane == 0) atomicAdd(&a.dbg[1], 1);
            const int ee = s0 + c0 + first;
            const int rq = a.q_idx ? a.q_idx[ee] : ee;
            int       lo = 0, hi = nt;
            if (a.q_idx) {
              const int g = group_of(a, ee);
              lo = a.t_off[g], hi = group_hi(a, g);
            }
            const int*      t_idx   = a.t_idx;
            const uint8_t*  t_valid = a.t_valid;
            const uint32_t* cl      = claimed;
            uint32_t        out[kTopK];
            int             nv;
            wave_topk(
                a.q + (size_t)rq * 32, a.t, lo, hi,
                [t_idx, t_valid, cl](int p) {
                  const int row = t_idx ? t_idx[p] : p;
                  if (t_valid && !t_valid[row]) return -1;
                  return (cl[row >> 5] & (1u << (row & 31))) ? -1 : row;
                },
                [](int, int) { return true; }, out, &nv);
            int b1 = 256, b2 = 256, bT = -1;
            if (out[0] != kEmpty) {
              const int p = (int)(out[0] & kPosMask);
              b1          = (int)(out[0] >> kPosBits);
              bT          = a.t_idx ? a.t_idx[p] : p;
            }
            if (out[1] != kEmpty) b2 = (int)(out[1] >> kPosBits);
            const bool pass = a.strict_lt ? b1 < a.th : b1 <= a.th;
            if (pass && (float)b1 < a.ratio * (float)b2) {
              if (lane == first) {
                claimed[bT >> 5] |= 1u << (bT & 31);
                myBestT = bT;
              }
              nmatches++;
            }
            remaining &= ~(1ull << first);
            continue;
          }
          // tentative proposals: earliest proposing lane per row
          if (active && accept) atomicMin(&minLane[bestT & (kLaneTab - 1)], (uint32_t)lane);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          bool touched = false;
          if (active) {
            if (rescan) {
              touched = true;
            } else if (passBest && found > 0) {
              touched = minLane[bestT & (kLaneTab - 1)] < (uint32_t)lane;
              if (!accept && secondT >= 0) touched = touched || minLane[secondT & (kLaneTab - 1)] < (uint32_t)lane;
            }
          }
          const unsigned long long tmask = __ballot(touched) & ~(1ull << first);
          const int                limit = tmask ? __ffsll((long long)tmask) - 1 : 64;
          const unsigned long long fin   = limit >= 64 ? remaining : (remaining & ((1ull << limit) - 1ull));
          const bool               mine  = (fin >> lane) & 1ull;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (active && accept) minLane[bestT & (kLaneTab - 1)] = 0xffu;  // retract (all reads above are done: same wave, in order)
          if (mine && accept) {
            atomicOr(&claimed[bestT >> 5], 1u << (bestT & 31));
            myBestT = bestT;
          }
          nmatches += __popcll(__ballot(mine && accept));
          remaining &= ~fin;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (i < ns) sBestT[i] = myBestT;
      }
      if (lane == 0) sNm += nmatches;
    }
    __syncthreads();
    // ---- publish: all waves write the committed matches of the super-chunk and vote in the rotation histogram
    for (int i = tid; i < ns; i += kResolveThreads) {
      const int bT = sBestT[i];
      if (bT >= 0) {
        const int e  = s0 + i;
        const int rq = a.q_idx ? a.q_idx[e] : e;
        a.match_t_of_q[rq] = bT;
        a.match_q_of_t[bT] = rq;
        if (a.check_ori) {
          const int bin = rot_bin(a.q_angle[(size_t)rq * a.q_astride], a.t_angle[(size_t)bT * a.t_astride]);
          atomicAdd(&hist[bin], 1);
          a.binOf[rq] = bin;
        }
      }
    }
    __syncthreads();
  }
  if (a.check_ori) {
    __shared__ int sRemoved;
    if (tid == 0) sRemoved = 0;
    __syncthreads();
    int i1, i2, i3;
    three_maxima(hist, kHistoLen, i1, i2, i3);
    int removed = 0;
    for (int q = tid; q < nq; q += kResolveThreads) {
      const int m = a.match_t_of_q[q];
      if (m >= 0) {
        const int bin = a.binOf[q];
        if (bin != i1 && bin != i2 && bin != i3) {
          removed++;
          a.match_q_of_t[m] = -1;
          a.match_t_of_q[q] = -1;
        }
      }
    }
    if (removed) atomicAdd(&sRemoved, removed);
    __syncthreads();
    if (tid == 0) sNm -= sRemoved;
  }
  __syncthreads();
  if (tid == 0) *a.nmatches = sNm;
}

// ------------------------------------------------------------------------------------------------
// SearchForInitialization (src/ORBmatcher.cc:666-782)
// ------------------------------------------------------------------------------------------------
struct WindowArgs {
  const uint8_t *d1, *d2;
  const orbx_keypoint_t *kp1, *kp2;
  int                    n1, n2, ng;  // ng = level-0, in-grid F2 keypoints in grid scan order
  const int*             g_idx;       // [ng] F2 keypoint index
  const uint8_t *        g_cx, *g_cy; // [ng] grid cell
  float*                 prev;        // [n1][2]
  float                  minX, minY, gwInv, ghInv, r;
  int                    th, check_ori;
  float                  ratio;
  int                    dstar;
  uint32_t*              topk;
  int*                   nvalid;
  int *                  m12, *m21, *binOf, *nmatches;
};

struct WinRange {
  int  x0, x1, y0, y1;
  bool empty;
};

// Frame::GetFeaturesInArea cell range (src/Frame.cc:802-830)
__device__ __forceinline__ WinRange window_cells(const WindowArgs& a, float x, float y) {
  WinRange w;
  w.x0    = max(0, (int)floorf((x - a.minX - a.r) * a.gwInv));
  w.x1    = min(kGridCols - 1, (int)ceilf((x - a.minX + a.r) * a.gwInv));
  w.y0    = max(0, (int)floorf((y - a.minY - a.r) * a.ghInv));
  w.y1    = min(kGridRows - 1, (int)ceilf((y - a.minY + a.r) * a.ghInv));
  w.empty = (w.x0 >= kGridCols) || (w.x1 < 0) || (w.y0 >= kGridRows) || (w.y1 < 0);
  return w;
}

__global__ __launch_bounds__(256) void k_window_topk(WindowArgs a) {
  const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i1 >= a.n1) return;
  uint32_t out[kTopK];
  int      nv = 0;
#pragma unroll
  for (int j = 0; j < kTopK; j++) out[j] = kEmpty;
  if (a.kp1[i1].octave <= 0) {  // level1 > 0 -> continue (:688-690)
    const float    x = a.prev[2 * i1], y = a.prev[2 * i1 + 1];
    const WinRange w = window_cells(a, x, y);
    if (!w.empty) {
      const WindowArgs& ar = a;
      wave_topk(
          a.d1 + (size_t)i1 * 32, a.d2, 0, a.ng,
          [&ar, w, x, y](int pos) {
            const int cx = ar.g_cx[pos], cy = ar.g_cy[pos];
            if (cx < w.x0 || cx > w.x1 || cy < w.y0 || cy > w.y1) return -1;
            const int   i2 = ar.g_idx[pos];
            const float dx = ar.kp2[i2].x - x, dy = ar.kp2[i2].y - y;
            return (fabsf(dx) < ar.r && fabsf(dy) < ar.r) ? i2 : -1;
          },
          [dstar = ar.dstar](int, int d) { return d < dstar; }, out, &nv);
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < kTopK; j++) a.topk[(size_t)i1 * kTopK + j] = out[j];
    a.nvalid[i1] = nv;
  }
}

__global__ __launch_bounds__(64) void k_window_resolve(WindowArgs a) {
  __shared__ unsigned short matchedDist[kMaxTrainW];  // vMatchedDistance; 0xffff = INT_MAX
  __shared__ int            hist[kHistoLen];
  const int                 lane = threadIdx.x;
  for (int i = lane; i < a.n2; i += 64) {
    matchedDist[i] = 0xffff;
    a.m21[i]       = -1;
  }
  for (int i = lane; i < a.n1; i += 64) a.m12[i] = -1;
  for (int i = lane; i < kHistoLen; i += 64) hist[i] = 0;
  __syncthreads();
  int nmatches = 0;
  for (int e0 = 0; e0 < a.n1; e0 += 64) {
    const int e   = e0 + lane;
    bool      pos = false;
    if (e < a.n1) {
      const uint32_t k0 = a.topk[(size_t)e * kTopK];
      pos               = k0 != kEmpty && (int)(k0 >> kPosBits) <= a.th;
    }
    unsigned long long mask = __ballot(pos);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int i1   = e0 + b;
      int       best = 0x7fffffff, second = 0x7fffffff, best2 = -1, found = 0;
      for (int j = 0; j < kTopK && found < 2; j++) {
        const uint32_t key = a.topk[(size_t)i1 * kTopK + j];
        if (key == kEmpty) break;
        const int p = (int)(key & kPosMask), d = (int)(key >> kPosBits);
        const int i2 = a.g_idx[p];
        if ((int)matchedDist[i2] <= d) continue;  // vMatchedDistance[i2] <= dist (:713)
        if (found == 0) {
          best = d, best2 = i2;
        } else {
          second = d;
        }
        found++;
      }
      bool rescan = false;
      int listLen = 0, dK = 0;
      for (int j = 0; j < kTopK; j++) {
        const uint32_t key = a.topk[(size_t)i1 * kTopK + j];
        if (key == kEmpty) break;
        listLen = j + 1;
        dK      = (int)(key >> kPosBits);
      }
      if (found < 2 && a.nvalid[i1] > listLen) {  // same dK bound as in k_group_resolve
        if (found == 1) {
          if (best <= a.th && !((float)best < (float)dK * a.ratio)) rescan = true;
          if (best <= a.th && !rescan) second = dK;
        } else {
          rescan = dK <= a.th;
        }
      }
      if (rescan) {
        const float               x = a.prev[2 * i1], y = a.prev[2 * i1 + 1];
        const WinRange            w  = window_cells(a, x, y);
        const WindowArgs&         ar = a;
        const unsigned short*     md = matchedDist;
        uint32_t                  out[kTopK];
        int                       nv;
        wave_topk(
            a.d1 + (size_t)i1 * 32, a.d2, 0, a.ng,
            [&ar, w, x, y](int pos2) {
              const int cx = ar.g_cx[pos2], cy = ar.g_cy[pos2];
              if (cx < w.x0 || cx > w.x1 || cy < w.y0 || cy > w.y1) return -1;
              const int   i2 = ar.g_idx[pos2];
              const float dx = ar.kp2[i2].x - x, dy = ar.kp2[i2].y - y;
              return (fabsf(dx) < ar.r && fabsf(dy) < ar.r) ? i2 : -1;
            },
            [md](int row, int d) { return !((int)md[row] <= d); }, out, &nv);
        best = 0x7fffffff, second = 0x7fffffff, best2 = -1;
        if (out[0] != kEmpty) {
          best  = (int)(out[0] >> kPosBits);
          best2 = a.g_idx[out[0] & kPosMask];
        }
        if (out[1] != kEmpty) second = (int)(out[1] >> kPosBits);
      }
      // bestDist <= TH_LOW && bestDist < (float)bestDist2 * mfNNratio (:723-724)
      if (best <= a.th && (float)best < (float)second * a.ratio) {
        if (lane == 0) {
          const int prevQ = a.m21[best2];
          if (prevQ >= 0) a.m12[prevQ] = -1;
          a.m12[i1]          = best2;
          a.m21[best2]       = i1;
          matchedDist[best2] = (unsigned short)best;
          if (a.check_ori) {
            const int bin = rot_bin(a.kp1[i1].angle, a.kp2[best2].angle);
            hist[bin]++;
            a.binOf[i1] = bin;
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  if (a.check_ori) {
    int i1m, i2m, i3m;
    three_maxima(hist, kHistoLen, i1m, i2m, i3m);
    for (int q = lane; q < a.n1; q += 64) {
      if (a.m12[q] >= 0) {
        const int bin = a.binOf[q];
        if (bin != i1m && bin != i2m && bin != i3m) a.m12[q] = -1;
      }
    }
  }
  __syncthreads();
  // nmatches == number of surviving pairs (every increment / decrement of the reference tracks one); prev update (:775-779)
  for (int q0 = 0; q0 < a.n1; q0 += 64) {
    const int q  = q0 + lane;
    bool      on = false;
    if (q < a.n1 && a.m12[q] >= 0) {
      on                = true;
      a.prev[2 * q]     = a.kp2[a.m12[q]].x;
      a.prev[2 * q + 1] = a.kp2[a.m12[q]].y;
    }
    nmatches += __popcll(__ballot(on));
  }
  if (lane == 0) *a.nmatches = nmatches;
}

// ------------------------------------------------------------------------------------------------
// N3: SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1684-1908, Nleft == -1)
// Query = a last-frame MapPoint already projected by the caller (uv, 1/z, octave, descriptor); candidates come from
// Frame::GetFeaturesInArea(u, v, th * scale[octave], level range) over the current frame's 64 x 48 grid; the best
// descriptor distance wins if <= TH_HIGH (no ratio test), a matched feature is closed to later points, rotation histogram.
// Top-k lists per query in parallel (k_proj_topk), then one wave walks the queries in order with the claim bitmap in LDS.
// ------------------------------------------------------------------------------------------------
struct ProjArgs {
  const uint8_t *        d1, *d2;
  const float *          uv, *invz, *angle1;
  const int*             oct1;
  const uint8_t*         valid1;
  const orbx_keypoint_t* kp2;
  const float*           uRight;    // NULL = monocular
  const uint8_t*         occupied;  // NULL = none
  int                    n1, n2, ng;
  const int*             g_idx;
  const uint8_t *        g_cx, *g_cy;
  const int*             colStart;  // [kGridCols + 1]: first list position of grid column c (the list is sorted by cell = cx * rows + cy); NULL = scan the whole list
  const float*           sf;
  float                  th, mbf, minX, minY, gwInv, ghInv;
  int                    dir, thDist, check_ori;
  uint32_t*              topk;
  int*                   nvalid;
  int *                  m12, *m21, *binOf, *nmatches;
  // batch form (pairs != NULL): blockIdx.y = pair p = {last frame, current frame} of one extraction batch ([frames][cap] buffers);
  // per-pair arrays are [npairs][cap]; the grid arrays come from k_assign_grid ([frames][cap], ngOf[frames])
  const int*             pairs;
  const int*             counts;
  const int*             ngOf;
  const orbx_keypoint_t* kp1;  // last frame's keypoints: octave and angle are read from here when set
  int                    cap, hasLastDesc;
  // mode 1 = SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) (src/ORBmatcher.cc:38-227): radius1[i] = RadiusByViewingCos * th,
  // levels [predicted - 1, predicted], `invz` carries mTrackProjXR, best / second-best with the same-level ratio rule
  int          mode;
  const float* radius1;
  float        ratio;
  int          dstar;  // mode 1: a second-best at distance >= dstar can never veto a best <= thDist
  // mode 2 = the point-by-point search of Fuse / SearchBySim3 (src/ORBmatcher.cc:1149-1452, :1499-1658): levels [predicted - 1, predicted],
  // no claims, `invz` carries uv(0) - bf / z; invSigma2 != NULL adds Fuse's reprojection gate (:1255-1283)
  const float* invSigma2;
};

__device__ __forceinline__ ProjArgs proj_view(const ProjArgs& a, int p) {
  if (!a.pairs) return a;
  ProjArgs     v  = a;
  const int    fl = a.pairs[2 * p], fc = a.pairs[2 * p + 1];
  const size_t pc = (size_t)p * a.cap, lc = (size_t)fl * a.cap, cc = (size_t)fc * a.cap;
  v.d1     = a.hasLastDesc ? a.d1 + pc * 32 : a.d2 + lc * 32;
  v.d2     = a.d2 + cc * 32;
  v.uv     = a.uv + pc * 2;
  v.invz   = a.invz ? a.invz + pc : nullptr;
  v.valid1 = a.valid1 ? a.valid1 + pc : nullptr;
  v.kp1    = a.kp2 + lc;
  v.kp2    = a.kp2 + cc;
  v.uRight = a.uRight ? a.uRight + cc : nullptr;
  v.occupied = a.occupied ? a.occupied + pc : nullptr;
  v.n1 = a.counts[2 * fl], v.n2 = a.counts[2 * fc], v.ng = a.ngOf[fc];
  v.g_idx = a.g_idx + cc, v.g_cx = a.g_cx + cc, v.g_cy = a.g_cy + cc;
  v.colStart = a.colStart ? a.colStart + (size_t)fc * (kGridCols + 1) : nullptr;
  v.topk = a.topk + pc * kTopK, v.nvalid = a.nvalid + pc, v.m12 = a.m12 + pc, v.m21 = a.m21 + pc, v.binOf = a.binOf + pc;
  v.nmatches = a.nmatches + p;
  return v;
}

struct ProjQuery {
  float u, v, r, invz;
  int   minL, maxL, x0, x1, y0, y1;
  bool  empty;
};

__device__ __forceinline__ ProjQuery proj_query(const ProjArgs& a, int i) {
  ProjQuery q;
  q.u = a.uv[2 * i], q.v = a.uv[2 * i + 1];
  q.invz      = a.invz ? a.invz[i] : 0.f;
  const int o = a.kp1 ? a.kp1[i].octave : a.oct1[i];
  q.r         = a.th * a.sf[o];  // :1736
  if (a.mode == 1) {
    q.r    = a.radius1[i] * a.sf[o];  // r * F.mvScaleFactors[nPredictedLevel] (:70-76)
    q.minL = o - 1, q.maxL = o;
  } else if (a.mode == 2 || a.dir == 3)
    q.minL = o - 1, q.maxL = o;  // kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel -> continue (:497-499, :1251-1253)
  else if (a.dir == 1)
    q.minL = o, q.maxL = -1;  // bForward: GetFeaturesInArea(u, v, radius, nLastOctave)
  else if (a.dir == 2)
    q.minL = 0, q.maxL = o;   // bBackward
  else
    q.minL = o - 1, q.maxL = o + 1;
  q.x0    = max(0, (int)floorf((q.u - a.minX - q.r) * a.gwInv));  // src/Frame.cc:802-830
  q.x1    = min(kGridCols - 1, (int)ceilf((q.u - a.minX + q.r) * a.gwInv));
  q.y0    = max(0, (int)floorf((q.v - a.minY - q.r) * a.ghInv));
  q.y1    = min(kGridRows - 1, (int)ceilf((q.v - a.minY + q.r) * a.ghInv));
  q.empty = (q.x0 >= kGridCols) || (q.x1 < 0) || (q.y0 >= kGridRows) || (q.y1 < 0);
  return q;
}

// list positions that can hold a candidate of query q: the grid columns x0 .. x1 of its window are one contiguous run of the list
__device__ __forceinline__ void proj_range(const ProjArgs& a, const ProjQuery& q, int* lo, int* hi) {
  *lo = a.colStart ? a.colStart[q.x0] : 0;
  *hi = a.colStart ? a.colStart[q.x1 + 1] : a.ng;
}

// candidate at grid-scan position pos for query q: the feature index, or -1 (outside the window / level range / stereo
// band, or holding a MapPoint with observations before the call)
__device__ __forceinline__ int proj_candidate(const ProjArgs& a, const ProjQuery& q, int pos) {
  const int cx = a.g_cx[pos], cy = a.g_cy[pos];
  if (cx < q.x0 || cx > q.x1 || cy < q.y0 || cy > q.y1) return -1;
  const int i2 = a.g_idx[pos];
  if (a.occupied && a.occupied[i2]) return -1;  // :1764-1768
  const orbx_keypoint_t kp = a.kp2[i2];
  if ((q.minL > 0) || (q.maxL >= 0)) {  // bCheckLevels (src/Frame.cc:832-848)
    if (kp.octave < q.minL) return -1;
    if (q.maxL >= 0 && kp.octave > q.maxL) return -1;
  }
  const float dx = kp.x - q.u, dy = kp.y - q.v;
  if (!(fabsf(dx) < q.r && fabsf(dy) < q.r)) return -1;
  if (a.mode == 2) {
    if (a.invSigma2) {  // Fuse (:1255-1283): chi-square gate on the reprojection error at the keypoint's level
      const float ex = q.u - kp.x, ey = q.v - kp.y;
      if (a.uRight && a.uRight[i2] >= 0) {
        const float er = q.invz - a.uRight[i2];
        const float e2 = ex * ex + ey * ey + er * er;
        if ((double)(e2 * a.invSigma2[kp.octave]) > 7.8) return -1;
      } else {
        const float e2 = ex * ex + ey * ey;
        if ((double)(e2 * a.invSigma2[kp.octave]) > 5.99) return -1;
      }
    }
    return i2;
  }
  if (a.uRight && a.uRight[i2] > 0) {  // :1770-1776 / :101-106
    const float ur = a.mode == 1 ? q.invz : q.u - a.mbf * q.invz;  // mode 1: pMP->mTrackProjXR
    const float er = fabsf(ur - a.uRight[i2]);
    if (er > q.r) return -1;
  }
  return i2;
}

__global__ __launch_bounds__(256) void k_proj_topk(ProjArgs a0) {
  const ProjArgs a  = proj_view(a0, blockIdx.y);
  const int      i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i1 >= a.n1) return;
  uint32_t out[kTopK];
  int      nv = 0;
#pragma unroll
  for (int j = 0; j < kTopK; j++) out[j] = kEmpty;
  if (!a.valid1 || a.valid1[i1]) {
    const ProjQuery q = proj_query(a, i1);
    if (!q.empty) {
      const ProjArgs& ar = a;
      int             pLo, pHi;
      proj_range(a, q, &pLo, &pHi);
      wave_topk(
          a.d1 + (size_t)i1 * 32, a.d2, pLo, pHi, [&ar, &q](int pos) { return proj_candidate(ar, q, pos); },
          [thd = ar.mode == 1 ? ar.dstar - 1 : ar.thDist](int, int d) { return d <= thd; }, out, &nv);  // mode 0: only "bestDist <= TH_HIGH" can match (:1788)
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < kTopK; j++) a.topk[(size_t)i1 * kTopK + j] = out[j];
    a.nvalid[i1] = nv;
  }
}

// mode 2: every point keeps the head of its list (first minimum in scan order) when that is within thDist -- the points do not
// interact (Fuse :1289-1308, SearchBySim3 :1571-1573)
__global__ __launch_bounds__(256) void k_proj_take_best(ProjArgs a) {
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < a.n1; i += 256) {
    const uint32_t key = a.topk[(size_t)i * kTopK];
    const bool     ok  = key != kEmpty && (int)(key >> kPosBits) <= a.thDist;
    a.m12[i]           = ok ? a.g_idx[key & kPosMask] : -1;
    mine += ok;
  }
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) *a.nmatches = total;
}

// Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:524-553, 867-877) for every frame of a batch: the features that fall in
// the 64 x 48 grid, listed in the scan order of GetFeaturesInArea (cell (ix, iy) ascending, feature index ascending inside).
// One workgroup per frame; keys cell << 16 | index bitonic-sorted in LDS (NP = power of two >= cap, <= 8192).
__global__ __launch_bounds__(256) void k_assign_grid(const orbx_keypoint_t* __restrict__ kps, const int* __restrict__ counts, int cap, int NP,
                                                     float minX, float minY, float gwInv, float ghInv, int* __restrict__ g_idx,
                                                     uint8_t* __restrict__ g_cx, uint8_t* __restrict__ g_cy, int* __restrict__ ngOf,
                                                     int* __restrict__ colStart) {
  extern __shared__ uint32_t gridKeys[];
  __shared__ int             sCount;
  __shared__ int             sCol[kGridCols + 1];
  const int                  f = blockIdx.x, tid = threadIdx.x;
  const int                  n = min(counts ? counts[2 * f] : cap, NP);
  if (tid == 0) sCount = 0;
  __syncthreads();
  int mine = 0;
  for (int p = tid; p < NP; p += 256) {
    uint32_t key = 0xffffffffu;
    if (p < n) {
      const orbx_keypoint_t kp = kps[(size_t)f * cap + p];
      const int posX = (int)roundf((kp.x - minX) * gwInv), posY = (int)roundf((kp.y - minY) * ghInv);
      if (!(posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows)) key = ((uint32_t)(posX * kGridRows + posY) << 16) | (uint32_t)p, mine++;
    }
    gridKeys[p] = key;
  }
  if (mine) atomicAdd(&sCount, mine);
  for (int k = 2; k <= NP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < (NP >> 1); t += 256) {
        const int      lo = 2 * t - (t & (j - 1)), hi = lo + j;
        const bool     up = (lo & k) == 0;
        const uint32_t x = gridKeys[lo], y = gridKeys[hi];
        if ((x > y) == up) gridKeys[lo] = y, gridKeys[hi] = x;
      }
    }
  __syncthreads();
  const int ng = sCount;
  for (int p = tid; p < ng; p += 256) {
    const uint32_t key  = gridKeys[p];
    const int      cell = (int)(key >> 16);
    g_idx[(size_t)f * cap + p] = (int)(key & 0xffffu);
    g_cx[(size_t)f * cap + p]  = (uint8_t)(cell / kGridRows);
    g_cy[(size_t)f * cap + p]  = (uint8_t)(cell % kGridRows);
  }
  if (tid == 0) ngOf[f] = ng;
  if (colStart) {  // first list position of every grid column (the list is sorted by cell): a window search only walks its columns' run
    for (int c = tid; c <= kGridCols; c += 256) sCol[c] = ng;
    __syncthreads();
    for (int p = tid; p < ng; p += 256) atomicMin(&sCol[(int)(gridKeys[p] >> 16) / kGridRows], p);
    __syncthreads();
    if (tid == 0)
      for (int c = kGridCols - 1; c >= 0; c--) sCol[c] = min(sCol[c], sCol[c + 1]);
    __syncthreads();
    for (int c = tid; c <= kGridCols; c += 256) colStart[(size_t)f * (kGridCols + 1) + c] = sCol[c];
  }
}

// The reference walks the points one by one; a point's choice only depends on which features EARLIER points took.  The wave
// takes 64 consecutive points (lane = point, its 8-deep list in registers) and runs rounds: every undecided lane proposes its
// first feature not yet taken; the proposals of the lanes before the first CONFLICT (a lane proposing what an earlier lane of
// this round proposes) are exactly what the sequential walk would do -- a committed proposal can only remove features that the
// later lanes' lists had already skipped or that they were not proposing -- so that prefix commits at once and the rest
// retries.  A lane whose list ran dry although more candidates exist stops the prefix and is rescanned exactly.
constexpr int kProjTab = 1024;  // proposal table (earliest proposing lane per hashed feature)

__global__ __launch_bounds__(64) void k_proj_resolve(ProjArgs a0) {
  const ProjArgs      a = proj_view(a0, blockIdx.x);
  __shared__ uint32_t claimed[kMaxTrainW / 32];
  __shared__ uint32_t minLane[kProjTab];
  __shared__ int      hist[kHistoLen];
  const int           lane = threadIdx.x;
  for (int i = lane; i < kMaxTrainW / 32; i += 64) claimed[i] = 0;
  for (int i = lane; i < kProjTab; i += 64) minLane[i] = 0xffu;
  for (int i = lane; i < a.n2; i += 64) a.m21[i] = -1;
  for (int i = lane; i < a.n1; i += 64) a.m12[i] = -1;
  for (int i = lane; i < kHistoLen; i += 64) hist[i] = 0;
  __syncthreads();
  int nmatches = 0;
  for (int c0 = 0; c0 < a.n1; c0 += 64) {
    const int i1 = c0 + lane;
    uint32_t  row[kTopK];  // feature index per list entry (kEmpty = none); the list is sorted by (distance, scan position)
    uint32_t  dl[kTopK];   // mode 1: distance << 8 | level of that entry
    int       len = 0, nv = 0, dK = 0;
#pragma unroll
    for (int j = 0; j < kTopK; j++) {
      row[j] = kEmpty, dl[j] = 0;
      if (i1 < a.n1) {
        const uint32_t key = a.topk[(size_t)i1 * kTopK + j];
        if (key != kEmpty) {
          row[j] = (uint32_t)a.g_idx[key & kPosMask], len = j + 1, dK = (int)(key >> kPosBits);
          if (a.mode == 1) dl[j] = ((key >> kPosBits) << 8) | (uint32_t)(a.kp2[row[j]].octave & 0xff);
        }
      }
    }
    if (i1 < a.n1) nv = a.nvalid[i1];
    const bool         truncated = nv > len;
    int                mine      = -1;
    unsigned long long pending   = __ballot(len > 0);
    while (pending) {
      const bool act = (pending >> lane) & 1ull;
      int        prop = -1, dep1 = -1, dep2 = -1;  // proposal; the best / second-best free features the decision depends on
      bool       dry  = false;          // the list cannot decide this point
      if (act) {
        int j1 = -1, j2 = -1;
#pragma unroll
        for (int j = kTopK - 1; j >= 0; j--)
          if (row[j] != kEmpty && !(claimed[row[j] >> 5] & (1u << (row[j] & 31)))) j2 = j1, j1 = j;
        uint32_t r1 = kEmpty, r2 = kEmpty, e1 = 0, e2 = 0;
#pragma unroll
        for (int j = 0; j < kTopK; j++) {
          if (j == j1) r1 = row[j], e1 = dl[j];
          if (j == j2) r2 = row[j], e2 = dl[j];
        }
        dep1 = j1 >= 0 ? (int)r1 : -1;
        if (a.mode != 1) {
          prop = dep1;
          dry  = prop < 0;
        } else if (j1 < 0) {
          dry = true;  // nothing free in the list: decided "no match" unless the list is a truncated prefix
        } else {
          const int d1 = (int)(e1 >> 8), l1 = (int)(e1 & 0xff);
          if (d1 <= a.thDist) {  // bestDist <= TH_HIGH (:137)
            bool accept;
            if (j2 >= 0) {
              const int d2 = (int)(e2 >> 8), l2 = (int)(e2 & 0xff);
              accept       = !(l1 == l2 && (float)d1 > a.ratio * (float)d2);  // :138-142
              dep2         = (int)r2;
            } else if (!truncated) {
              accept = true;  // no second candidate: bestLevel2 == -1
            } else {
              // whatever comes after the list is at least dK away: it cannot veto if even dK does not
              accept = !((float)d1 > a.ratio * (float)dK);
              dry    = !accept;  // undecidable from the list -> exact rescan
            }
            if (accept) prop = (int)r1;
          }
        }
      }
      if (act && prop >= 0) atomicMin(&minLane[prop & (kProjTab - 1)], (uint32_t)lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // a lane is blocked by an earlier lane of this round PROPOSING (hashing to the slot of) a feature its decision depends on:
      // its best free candidate -- also when the lane itself vetoes it -- or, in mode 1, its runner-up; false sharing only delays
      const bool blocked = act && ((dep1 >= 0 && minLane[dep1 & (kProjTab - 1)] < (uint32_t)lane) ||
                                   (dep2 >= 0 && minLane[dep2 & (kProjTab - 1)] < (uint32_t)lane));
      const bool rescan  = act && dry && truncated;
      const unsigned long long stopMask = __ballot(blocked || rescan);
      const int                f        = stopMask ? __ffsll((long long)stopMask) - 1 : 64;
      const bool               commit   = act && lane < f;
      if (commit && prop >= 0) {
        atomicOr(&claimed[prop >> 5], 1u << (prop & 31));  // CurrentFrame.mvpMapPoints[bestIdx2] = pMP (:1789)
        mine = prop;
      }
      if (act && prop >= 0) minLane[prop & (kProjTab - 1)] = 0xffu;  // every writer of a slot resets it
      pending &= ~((f >= 64 ? ~0ull : ((1ull << f) - 1ull)) & pending);  // the prefix is decided (matched or dry for good)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (f < 64 && ((__ballot(rescan) >> f) & 1ull)) {
        // lane f is the first undecided point and its list ran dry: exact rescan of its window with the claim filter
        const int       iq = c0 + f;
        const ProjQuery q  = proj_query(a, iq);
        const ProjArgs& ar = a;
        const uint32_t* cl = claimed;
        uint32_t        out[kTopK];
        int             nv2, pLo, pHi;
        proj_range(a, q, &pLo, &pHi);
        wave_topk(
            a.d1 + (size_t)iq * 32, a.d2, pLo, pHi,
            [&ar, &q, cl](int pos) {
              const int i2 = proj_candidate(ar, q, pos);
              return (i2 >= 0 && (cl[i2 >> 5] & (1u << (i2 & 31)))) ? -1 : i2;
            },
            [thd = ar.mode == 1 ? ar.dstar - 1 : ar.thDist](int, int d) { return d <= thd; }, out, &nv2);
        bool take = out[0] != kEmpty;
        if (take && a.mode == 1) {
          const int d1 = (int)(out[0] >> kPosBits), l1 = a.kp2[a.g_idx[out[0] & kPosMask]].octave;
          take         = d1 <= a.thDist;
          if (take && out[1] != kEmpty) {
            const int d2 = (int)(out[1] >> kPosBits), l2 = a.kp2[a.g_idx[out[1] & kPosMask]].octave;
            take         = !(l1 == l2 && (float)d1 > a.ratio * (float)d2);
          }
        }
        if (take) {
          const int b = a.g_idx[out[0] & kPosMask];
          if (lane == f) {
            claimed[b >> 5] |= 1u << (b & 31);
            mine = b;
          }
        }
        pending &= ~(1ull << f);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    if (mine >= 0) {
      a.m12[i1]   = mine;
      a.m21[mine] = i1;
      if (a.check_ori) {  // :1792-1809
        const int bin = rot_bin(a.kp1 ? a.kp1[i1].angle : a.angle1[i1], a.kp2[mine].angle);
        atomicAdd(&hist[bin], 1);
        a.binOf[i1] = bin;
      }
    }
    nmatches += __popcll(__ballot(mine >= 0));
  }
  __syncthreads();
  if (a.check_ori) {  // :1889-1905
    int i1m, i2m, i3m;
    three_maxima(hist, kHistoLen, i1m, i2m, i3m);
    int removed = 0;
    for (int q0 = 0; q0 < a.n1; q0 += 64) {
      const int q   = q0 + lane;
      bool      off = false;
      if (q < a.n1 && a.m12[q] >= 0) {
        const int bin = a.binOf[q];
        if (bin != i1m && bin != i2m && bin != i3m) {
          a.m21[a.m12[q]] = -1;
          a.m12[q]        = -1;
          off             = true;
        }
      }
      removed += __popcll(__ballot(off));
    }
    nmatches -= removed;
  }
  if (lane == 0) *a.nmatches = nmatches;
}

// Exact pruning bound of the top-K pass.  A match needs best <= th (or < th) and best < ratio * second.  Let D* be the
// smallest integer with (float)th < ratio * (float)D*.  A candidate at distance >= D* can never be the best of an
// accepted match (D* > th), and as a runner-up it passes the ratio test for EVERY admissible best -- exactly as a
// missing runner-up (256 / INT_MAX) does.  So the K-lists only need the candidates below D*; the rest are dropped
// after one compare, before any sorting work.  (Random 256-bit descriptors sit around 128 +- 8: almost all pruned.)
int prune_bound(int th, float ratio) {
  int d = th + 1;
  while (d < 257 && !((float)th < ratio * (float)d)) d++;
  return d;
}

template <class T>
struct DBuf {
  T*     p = nullptr;
  size_t n = 0;
  int    ensure(size_t count) {
    if (count <= n && p) return 0;
    if (p) hipFree(p);
    p = nullptr, n = 0;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
      set_err(std::string("hipMalloc: ") + hipGetErrorString(e));
      return -3;
    }
    n = count;
    return 0;
  }
  void release() {
    if (p) hipFree(p);
    p = nullptr, n = 0;
  }
};

}  // namespace

// Host-pointer entry points are latency-bound: a dozen small hipMemcpyAsync calls cost more than the kernels.  The inputs of a
// call are packed into one pinned staging block and go up with ONE copy; results come back the same way.
struct Arena {
  uint8_t *h = nullptr, *d = nullptr;
  size_t   cap = 0, off = 0;
  int      reserve(size_t bytes) {
    off = 0, overflow = false;
    if (bytes <= cap) return 0;
    release();
    if (hipHostMalloc((void**)&h, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&d, bytes) != hipSuccess) {
      set_err("staging arena allocation failed");
      release();
      return -3;
    }
    cap = bytes;
    return 0;
  }
  void release() {
    if (h) hipHostFree(h);
    if (d) hipFree(d);
    h = d = nullptr, cap = off = 0;
  }
  bool overflow = false;  // a put() did not fit: checked by the caller after the last put (ok())
  template <class T>
  T* put(const T* src, size_t n) {  // returns the DEVICE address; src == NULL just reserves
    off   = (off + 15) & ~(size_t)15;
    T* dp = reinterpret_cast<T*>(d + off);
    if (off + n * sizeof(T) > cap) {  // never write past the reservation: the caller's byte budget was wrong
      overflow = true;
      return reinterpret_cast<T*>(d);
    }
    if (src && n) memcpy(h + off, src, n * sizeof(T));
    off += n * sizeof(T);
    return dp;
  }
  bool ok() {
    if (!overflow) return true;
    set_err("internal: staging arena budget exceeded");
    return false;
  }
  template <class T>
  const T* host_of(const T* dev_ptr) const { return reinterpret_cast<const T*>(h + (reinterpret_cast<const uint8_t*>(dev_ptr) - d)); }
};

struct orbm_matcher {
  int         device = 0;
  hipStream_t stream = nullptr;
  Arena       in, out;
  DBuf<uint8_t>         dq, dt, dqv, dtv, gcx, gcy;
  DBuf<int>             qoff, qidx, toff, tidx, nvalid, binOf, mq, mt, nm, gidx, dbg, colStart;
  int*                  colStartOut = nullptr;  // set while the internal grid build runs (orbm_search_by_projection_batch_device)
  bool                  dbgOn = false;
  DBuf<float>           qa, ta, prev;
  DBuf<uint32_t>        topk;
  DBuf<orbx_keypoint_t> kp1, kp2;
};

extern "C" {

const char* orbm_last_error(void) { return g_merr.c_str(); }

static void orbm_release_device_side(orbm_matcher* h);

int orbm_create(orbm_t** out, int device_id) {
  if (!out) return -2;
  *out = nullptr;
  HIP_TRYM(hipSetDevice(device_id));
  orbm_matcher* h = new orbm_matcher();
  h->device       = device_id;
  hipError_t e    = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    delete h;
    return -3;
  }
  orbx_internal_register(h, [](void* p) { orbm_release_device_side(static_cast<orbm_matcher*>(p)); });
  *out = h;
  return 0;
}

static void orbm_release_device_side(orbm_matcher* h) {
  hipSetDevice(h->device);
  if (h->stream) {
    hipStreamSynchronize(h->stream);
    hipStreamDestroy(h->stream);
  }
  h->dq.release(), h->dt.release(), h->dqv.release(), h->dtv.release(), h->gcx.release(), h->gcy.release(), h->qoff.release();
  h->qidx.release(), h->toff.release(), h->tidx.release(), h->nvalid.release(), h->binOf.release(), h->mq.release(), h->mt.release();
  h->nm.release(), h->gidx.release(), h->qa.release(), h->ta.release(), h->prev.release(), h->topk.release(), h->kp1.release();
  h->kp2.release(), h->dbg.release();
  h->in.release(), h->out.release();
  h->stream = nullptr;
}

void orbm_destroy(orbm_t* h) {
  if (!h) return;
  orbx_internal_unregister(h);
  if (!orbx_internal_runtime_down()) orbm_release_device_side(h);  // else: process teardown, the exit handler (orbx_extractor.hip) has released it
  delete h;
}

// Tuning hook: enable == 1 zeroes and arms the counters {speculative rounds, exact rescans, 64-query chunks} of
// orbm_match_pairs_device; enable == 0 synchronises and reads them into out[3].
int orbm_debug_counters(orbm_t* h, int enable, int* out) {
  if (!h) return -2;
  HIP_TRYM(hipSetDevice(h->device));
  if (enable) {
    if (h->dbg.ensure(4)) return -3;
    HIP_TRYM(hipMemset(h->dbg.p, 0, 16));
    h->dbgOn = true;
    return 0;
  }
  HIP_TRYM(hipDeviceSynchronize());
  if (out && h->dbg.p) HIP_TRYM(hipMemcpy(out, h->dbg.p, 12, hipMemcpyDeviceToHost));
  h->dbgOn = false;
  return 0;
}

int orbm_hamming(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 4; i++) {
    unsigned long long x, y;
    memcpy(&x, a + 8 * i, 8);
    memcpy(&y, b + 8 * i, 8);
    dist += __builtin_popcountll(x ^ y);
  }
  return dist;
}

void orbm_three_maxima(const int* counts, int len, int* ind1, int* ind2, int* ind3) {
  int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
  for (int i = 0; i < len; i++) {
    const int s = counts[i];
    if (s > max1) {
      max3 = max2, max2 = max1, max1 = s;
      i3 = i2, i2 = i1, i1 = i;
    } else if (s > max2) {
      max3 = max2, max2 = s;
      i3 = i2, i2 = i;
    } else if (s > max3) {
      max3 = s, i3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    i2 = -1, i3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    i3 = -1;
  }
  *ind1 = i1, *ind2 = i2, *ind3 = i3;
}


// For pair p and every node g of the query frame's FeatureVector: the range of the train frame's FeatureVector payload that
// belongs to the same node (empty if the train frame has no feature under it) -- the merge walk of SearchByBoW as a lookup.
__global__ __launch_bounds__(256) void k_bow_pair_groups(const int* __restrict__ pairs, int cap, const uint32_t* __restrict__ fvNodes,
                                                         const int* __restrict__ fvOff, const int* __restrict__ fvNNodes, int* __restrict__ grpLo,
                                                         int* __restrict__ grpHi) {
  const int p = blockIdx.y, g = blockIdx.x * 256 + threadIdx.x;
  const int fq = pairs[2 * p], ft = pairs[2 * p + 1];
  if (g >= fvNNodes[fq]) return;
  const uint32_t  node = fvNodes[(size_t)fq * cap + g];
  const uint32_t* tn   = fvNodes + (size_t)ft * cap;
  int             lo = 0, hi = fvNNodes[ft];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tn[mid] < node)
      lo = mid + 1;
    else
      hi = mid;
  }
  int a = 0, b = 0;
  if (lo < fvNNodes[ft] && tn[lo] == node) a = fvOff[(size_t)ft * (cap + 1) + lo], b = fvOff[(size_t)ft * (cap + 1) + lo + 1];
  grpLo[(size_t)p * cap + g] = a;
  grpHi[(size_t)p * cap + g] = b;
}

static int launch_groups(orbm_matcher* h, GroupArgs& a, int nq_cap, hipStream_t s, int npairs = 1) {
  const int nqe = a.q_idx ? a.nqe : nq_cap;
  if (nqe > 0) {
    if (a.q_idx || a.fvFeat)
      hipLaunchKernelGGL(k_group_topk, dim3((nqe + 3) / 4, npairs), dim3(256), 0, s, a);
    else if (a.t_valid)
      hipLaunchKernelGGL(k_bf_mfma<true>, dim3((nqe + kMfThreads / 2 - 1) / (kMfThreads / 2), npairs), dim3(kMfThreads), 0, s, a);
    else
      hipLaunchKernelGGL(k_bf_mfma<false>, dim3((nqe + kMfThreads / 2 - 1) / (kMfThreads / 2), npairs), dim3(kMfThreads), 0, s, a);
  }
  hipLaunchKernelGGL(k_group_resolve, dim3(npairs), dim3(kResolveThreads), 0, s, a);
  HIP_TRYM(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace {

// ------------------------------------------------------------------------------------------------
// SearchByBoW(KeyFrame*, Frame&) for a two-camera frame (F.Nleft != -1, src/ORBmatcher.cc:305-333 and 362-389).  The train
// features are the frame's left keys [0, nleft) followed by its right keys; a query keeps a best / second-best pair PER CAMERA
// from one scan of its group, accepts the left match under the usual tests and the right one whenever its distance is <= th
// ("|| true", :363) -- inside the left camera's `bestDist1 <= TH_LOW` block, as the reference nests it -- so one query can claim
// two train features.  Not a headline path (no BASELINE configuration has a second fisheye camera): one wave walks the queries in
// the reference's order, its lanes share a query's candidates; the group's candidate descriptors and a slice of 64 query
// descriptors are staged in LDS so that the walk itself touches no global memory, and the claims (vpMapPointMatches) live in LDS.
// The sequential scan's tie rule (strict `<`: the first of equal distances wins) is kept by reducing on (distance, position).
// ------------------------------------------------------------------------------------------------
constexpr int kTwoCamMaxTrain = 8192;  // claims + bins in LDS
constexpr int kTwoCamStage    = 256;   // candidates of a group held in LDS (larger groups are read from global memory)

struct TwoCamArgs {
  const uint8_t *q, *t;
  int            nq, nt, nleft, ngroups;
  const int *    q_off, *q_idx, *t_off, *t_idx;
  const uint8_t* q_valid;
  const float *  q_angle, *t_angle;
  int            th, check_ori;
  float          ratio;
  int *          match_q_of_t, *nmatches;
};

__global__ __launch_bounds__(64) void k_group_two_cameras(TwoCamArgs a) {
  __shared__ int      sMt[kTwoCamMaxTrain];
  __shared__ uint8_t  sBin[kTwoCamMaxTrain];
  __shared__ uint32_t sT[kTwoCamStage][9];  // 8 words + 1 pad: lanes read different rows, same word
  __shared__ int      sTi[kTwoCamStage];
  __shared__ uint32_t sQ[64][9];
  __shared__ int      sQi[64];
  __shared__ int      hist[kHistoLen];
  const int           lane = threadIdx.x;
  for (int i = lane; i < a.nt; i += 64) sMt[i] = -1, sBin[i] = 0xff;
  if (lane < kHistoLen) hist[lane] = 0;
  __syncthreads();
  int            nmatches = 0;
  const uint32_t kNone    = (256u << 20) | 0xfffffu;
  for (int g = 0; g < a.ngroups; g++) {
    const int q0 = a.q_off ? a.q_off[g] : 0, q1 = a.q_off ? a.q_off[g + 1] : a.nq;
    const int t0 = a.t_off ? a.t_off[g] : 0, t1 = a.t_off ? a.t_off[g + 1] : a.nt;
    const int nT = t1 - t0;
    if (q1 <= q0 || nT <= 0) continue;
    const bool staged = nT <= kTwoCamStage;
    if (staged) {
      for (int p = lane; p < nT; p += 64) {
        const int       rt = a.t_idx ? a.t_idx[t0 + p] : t0 + p;
        const uint32_t* d  = reinterpret_cast<const uint32_t*>(a.t + (size_t)rt * 32);
        sTi[p]             = rt;
#pragma unroll
        for (int w = 0; w < 8; w++) sT[p][w] = d[w];
      }
    }
    for (int qb = q0; qb < q1; qb += 64) {
      __syncthreads();  // the previous slice (and the staged candidates) are no longer / now readable
      if (qb + lane < q1) {
        const int       rq = a.q_idx ? a.q_idx[qb + lane] : qb + lane;
        const uint32_t* d  = reinterpret_cast<const uint32_t*>(a.q + (size_t)rq * 32);
        sQi[lane]          = (a.q_valid && !a.q_valid[rq]) ? -1 : rq;  // !pMP || pMP->isBad()
#pragma unroll
        for (int w = 0; w < 8; w++) sQ[lane][w] = d[w];
      }
      __syncthreads();
      const int nq = min(64, q1 - qb);
      for (int j = 0; j < nq; j++) {
        const int rq = sQi[j];  // wave-uniform
        if (rq < 0) continue;
        uint32_t qw[8];
#pragma unroll
        for (int w = 0; w < 8; w++) qw[w] = sQ[j][w];
        uint32_t l1 = kNone, l2 = kNone, r1 = kNone, r2 = kNone;  // this lane's two smallest (distance << 20 | position) per camera
        for (int p = lane; p < nT; p += 64) {
          int rt, d = 0;
          if (staged) {
            rt = sTi[p];
            if (sMt[rt] >= 0) continue;  // vpMapPointMatches[realIdxF]
#pragma unroll
            for (int w = 0; w < 8; w++) d += __popc(qw[w] ^ sT[p][w]);
          } else {
            rt = a.t_idx ? a.t_idx[t0 + p] : t0 + p;
            if (sMt[rt] >= 0) continue;
            const uint32_t* td = reinterpret_cast<const uint32_t*>(a.t + (size_t)rt * 32);
#pragma unroll
            for (int w = 0; w < 8; w++) d += __popc(qw[w] ^ td[w]);
          }
          const uint32_t key = ((uint32_t)d << 20) | (uint32_t)p;
          if (rt < a.nleft) {
            l2 = min(l2, max(l1, key)), l1 = min(l1, key);
          } else {
            r2 = min(r2, max(r1, key)), r1 = min(r1, key);
          }
        }
        // the wave's smallest key per camera, and the smallest of the rest (its distance is bestDist2)
        const uint32_t L1 = wave_min_u32(l1), L2 = wave_min_u32(l1 == L1 ? l2 : l1);
        const uint32_t R1 = wave_min_u32(r1);
        const int      bestDist1 = (int)(L1 >> 20), bestDist2 = (int)(L2 >> 20), bestDist1R = (int)(R1 >> 20);
        if (bestDist1 <= a.th) {
          if ((float)bestDist1 < a.ratio * (float)bestDist2) {
            const int pL = (int)(L1 & 0xfffffu);
            const int rt = staged ? sTi[pL] : (a.t_idx ? a.t_idx[t0 + pL] : t0 + pL);
            if (lane == 0) {
              sMt[rt] = rq;
              if (a.check_ori) {
                const int bin = rot_bin(a.q_angle[rq], a.t_angle[rt]);
                sBin[rt]      = (uint8_t)bin;
                hist[bin]++;
              }
            }
            nmatches++;
          }
          if (bestDist1R <= a.th) {
            const int pR = (int)(R1 & 0xfffffu);
            const int rt = staged ? sTi[pR] : (a.t_idx ? a.t_idx[t0 + pR] : t0 + pR);
            if (lane == 0) {
              sMt[rt] = rq;
              if (a.check_ori) {
                const int bin = rot_bin(a.q_angle[rq], a.t_angle[rt]);
                sBin[rt]      = (uint8_t)bin;
                hist[bin]++;
              }
            }
            nmatches++;
          }
          __syncthreads();  // the claims are visible to every lane before the next query reads them
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  int removed = 0;
  if (a.check_ori) {
    int i1, i2, i3;
    three_maxima(hist, kHistoLen, i1, i2, i3);
    for (int i = lane; i < a.nt; i += 64) {
      const int b = sBin[i];
      if (sMt[i] >= 0 && b != i1 && b != i2 && b != i3) sMt[i] = -1, removed++;
    }
    removed = wave_sum_i32(removed);
  }
  __syncthreads();
  for (int i = lane; i < a.nt; i += 64) a.match_q_of_t[i] = sMt[i];
  if (lane == 0) *a.nmatches = nmatches - removed;
}

}  // namespace

extern "C" {

int orbm_match_groups_two_cameras(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int t_nleft, int ngroups, const int* q_off,
                                  const int* q_idx, const int* t_off, const int* t_idx, const uint8_t* q_valid, const float* q_angle,
                                  const float* t_angle, int th, float ratio, int check_ori, int* match_q_of_t) {
  if (!h || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || ngroups < 1 || t_nleft < 0 || t_nleft > nt) {
    set_err("orbm_match_groups_two_cameras: bad arguments");
    return -2;
  }
  const bool grouped = q_idx || t_idx || q_off || t_off;
  if (grouped && !(q_idx && t_idx && q_off && t_off)) {
    set_err("orbm_match_groups_two_cameras: q_off, q_idx, t_off, t_idx must be given together");
    return -2;
  }
  if (!grouped && ngroups != 1) {
    set_err("orbm_match_groups_two_cameras: ngroups > 1 needs the group arrays");
    return -2;
  }
  if (check_ori && (!q_angle || !t_angle)) {
    set_err("orbm_match_groups_two_cameras: check_ori needs angles");
    return -2;
  }
  if (nt > kTwoCamMaxTrain) {
    set_err("orbm_match_groups_two_cameras: more than 8192 train features");
    return -2;
  }
  const int nqe = grouped ? q_off[ngroups] : nq, nte = grouped ? t_off[ngroups] : nt;
  if (grouped) {
    for (int g = 0; g < ngroups; g++)
      if (q_off[g] > q_off[g + 1] || t_off[g] > t_off[g + 1] || q_off[g] < 0 || t_off[g] < 0) {
        set_err("orbm_match_groups_two_cameras: offsets must be non-decreasing");
        return -2;
      }
    for (int i = 0; i < nqe; i++)
      if (q_idx[i] < 0 || q_idx[i] >= nq) {
        set_err("orbm_match_groups_two_cameras: q_idx out of range");
        return -2;
      }
    for (int i = 0; i < nte; i++)
      if (t_idx[i] < 0 || t_idx[i] >= nt) {
        set_err("orbm_match_groups_two_cameras: t_idx out of range");
        return -2;
      }
  }
  HIP_TRYM(hipSetDevice(h->device));
  hipStream_t  s       = h->stream;
  const size_t inBytes = (size_t)nq * (32 + 4 + 1) + (size_t)nt * (32 + 4) + (size_t)(2 * ngroups + 2 + nqe + nte) * 4 + 16 * 16;
  if (h->in.reserve(inBytes) || h->out.reserve((size_t)(nt + 4) * 4 + 64)) return -3;
  TwoCamArgs a;
  memset(&a, 0, sizeof(a));
  Arena& in = h->in;
  a.q = in.put(q, (size_t)nq * 32), a.t = in.put(t, (size_t)nt * 32);
  a.q_angle = q_angle ? in.put(q_angle, nq) : nullptr, a.t_angle = t_angle ? in.put(t_angle, nt) : nullptr;
  a.q_valid = q_valid ? in.put(q_valid, nq) : nullptr;
  if (grouped) {
    a.q_off = in.put(q_off, ngroups + 1), a.t_off = in.put(t_off, ngroups + 1);
    a.q_idx = in.put(q_idx, nqe), a.t_idx = in.put(t_idx, nte);
  }
  if (!in.ok()) return -3;
  if (in.off) HIP_TRYM(hipMemcpyAsync(in.d, in.h, in.off, hipMemcpyHostToDevice, s));
  Arena& out = h->out;
  a.nmatches = out.put((const int*)nullptr, 4), a.match_q_of_t = out.put((const int*)nullptr, nt);
  if (!out.ok()) return -3;
  a.nq = nq, a.nt = nt, a.nleft = t_nleft, a.ngroups = ngroups, a.th = th, a.check_ori = check_ori, a.ratio = ratio;
  hipLaunchKernelGGL(k_group_two_cameras, dim3(1), dim3(64), 0, s, a);
  HIP_TRYM(hipGetLastError());
  HIP_TRYM(hipMemcpyAsync(out.h, out.d, out.off, hipMemcpyDeviceToHost, s));
  HIP_TRYM(hipStreamSynchronize(s));
  if (match_q_of_t && nt) memcpy(match_q_of_t, out.host_of(a.match_q_of_t), (size_t)nt * 4);
  return *out.host_of(a.nmatches);
}

int orbm_match_groups(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int ngroups, const int* q_off, const int* q_idx,
                      const int* t_off, const int* t_idx, const uint8_t* q_valid, const uint8_t* t_valid, const float* q_angle,
                      const float* t_angle, int th, int strict_lt, float ratio, int check_ori, int* match_t_of_q, int* match_q_of_t) {
  if (!h || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || ngroups < 1) {
    set_err("orbm_match_groups: bad arguments");
    return -2;
  }
  const bool grouped = q_idx || t_idx || q_off || t_off;
  if (grouped && !(q_idx && t_idx && q_off && t_off)) {
    set_err("orbm_match_groups: q_off, q_idx, t_off, t_idx must be given together");
    return -2;
  }
  if (!grouped && ngroups != 1) {
    set_err("orbm_match_groups: ngroups > 1 needs the group arrays");
    return -2;
  }
  if (check_ori && (!q_angle || !t_angle)) {
    set_err("orbm_match_groups: check_ori needs angles");
    return -2;
  }
  if (nt > kMaxTrain || nt >= (1 << kPosBits)) {
    set_err("orbm_match_groups: more than 65536 train features");
    return -2;
  }
  HIP_TRYM(hipSetDevice(h->device));
  const int nqe = grouped ? q_off[ngroups] : nq;
  const int nte = grouped ? t_off[ngroups] : nt;
  if (grouped) {
    for (int g = 0; g < ngroups; g++)
      if (q_off[g] > q_off[g + 1] || t_off[g] > t_off[g + 1] || q_off[g] < 0 || t_off[g] < 0) {
        set_err("orbm_match_groups: offsets must be non-decreasing");
        return -2;
      }
    for (int i = 0; i < nqe; i++)
      if (q_idx[i] < 0 || q_idx[i] >= nq) {
        set_err("orbm_match_groups: q_idx out of range");
        return -2;
      }
    for (int i = 0; i < nte; i++)
      if (t_idx[i] < 0 || t_idx[i] >= nt) {
        set_err("orbm_match_groups: t_idx out of range");
        return -2;
      }
  }
  hipStream_t  s       = h->stream;
  // one packed upload, one packed download (see Arena)
  const size_t inBytes = (size_t)nq * (32 + 4 + 1) + (size_t)nt * (32 + 4 + 1) + (size_t)(2 * ngroups + 2 + nqe + nte) * 4 + 16 * 16;
  if (h->in.reserve(inBytes) || h->out.reserve((size_t)(nq + nt + 4) * 4 + 64) || h->topk.ensure((size_t)(nqe + 1) * kTopK) ||
      h->nvalid.ensure(nqe + 1) || h->binOf.ensure(nq + 1))
    return -3;
  GroupArgs a;
  memset(&a, 0, sizeof(a));
  Arena& in = h->in;
  a.q = in.put(q, (size_t)nq * 32), a.t = in.put(t, (size_t)nt * 32);
  a.q_angle = q_angle ? in.put(q_angle, nq) : nullptr, a.t_angle = t_angle ? in.put(t_angle, nt) : nullptr;
  a.q_valid = q_valid ? in.put(q_valid, nq) : nullptr, a.t_valid = t_valid ? in.put(t_valid, nt) : nullptr;
  if (grouped) {
    a.q_off = in.put(q_off, ngroups + 1), a.t_off = in.put(t_off, ngroups + 1);
    a.q_idx = in.put(q_idx, nqe), a.t_idx = in.put(t_idx, nte);
  }
  if (!in.ok() || !h->out.ok()) return -3;
  if (in.off) HIP_TRYM(hipMemcpyAsync(in.d, in.h, in.off, hipMemcpyHostToDevice, s));
  Arena& out = h->out;
  a.nmatches = out.put((const int*)nullptr, 4), a.match_t_of_q = out.put((const int*)nullptr, nq), a.match_q_of_t = out.put((const int*)nullptr, nt);
  if (!out.ok()) return -3;
  a.nq = nq, a.nt = nt, a.ngroups = ngroups, a.nqe = nqe, a.q_astride = 1, a.t_astride = 1;
  a.th = th, a.strict_lt = strict_lt, a.check_ori = check_ori, a.ratio = ratio, a.dstar = prune_bound(th, ratio);
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p;
  int rc = launch_groups(h, a, nq, s);
  if (rc) return rc;
  HIP_TRYM(hipMemcpyAsync(out.h, out.d, out.off, hipMemcpyDeviceToHost, s));
  HIP_TRYM(hipStreamSynchronize(s));
  if (match_t_of_q && nq) memcpy(match_t_of_q, out.host_of(a.match_t_of_q), (size_t)nq * 4);
  if (match_q_of_t && nt) memcpy(match_q_of_t, out.host_of(a.match_q_of_t), (size_t)nt * 4);
  return *out.host_of(a.nmatches);
}

int orbm_match_bruteforce_device(orbm_t* h, const uint8_t* d_q, const orbx_keypoint_t* d_q_kp, const int* d_nq, int cap_q,
                                 const uint8_t* d_t, const orbx_keypoint_t* d_t_kp, const int* d_nt, int cap_t, int th, int strict_lt,
                                 float ratio, int check_ori, int* d_match_t_of_q, int* d_nmatches, void* stream) {
  if (!h || !d_q || !d_t || !d_nq || !d_nt || !d_match_t_of_q || !d_nmatches || cap_q <= 0 || cap_t <= 0 || cap_t > kMaxTrain) {
    set_err("orbm_match_bruteforce_device: bad arguments");
    return -2;
  }
  if (check_ori && (!d_q_kp || !d_t_kp)) {
    set_err("orbm_match_bruteforce_device: check_ori needs the keypoint arrays");
    return -2;
  }
  HIP_TRYM(hipSetDevice(h->device));
  if (h->topk.ensure((size_t)(cap_q + 1) * kTopK) || h->nvalid.ensure(cap_q + 1) || h->binOf.ensure(cap_q + 1) || h->mt.ensure(cap_t + 1))
    return -3;
  GroupArgs a;
  memset(&a, 0, sizeof(a));
  a.q = d_q, a.t = d_t, a.d_nq = d_nq, a.d_nt = d_nt, a.nq = cap_q, a.nt = cap_t, a.ngroups = 1, a.nqe = cap_q;
  a.q_angle = d_q_kp ? &d_q_kp->angle : nullptr, a.t_angle = d_t_kp ? &d_t_kp->angle : nullptr;
  a.q_astride = a.t_astride = sizeof(orbx_keypoint_t) / 4;
  a.th = th, a.strict_lt = strict_lt, a.check_ori = check_ori, a.ratio = ratio, a.dstar = prune_bound(th, ratio);
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p;
  a.match_t_of_q = d_match_t_of_q, a.match_q_of_t = h->mt.p, a.nmatches = d_nmatches;
  return launch_groups(h, a, cap_q, stream ? (hipStream_t)stream : h->stream);
}

int orbm_match_pairs_device(orbm_t* h, const uint8_t* d_desc, const orbx_keypoint_t* d_kps, const int* d_counts, int cap,
                            const int* d_pairs, int npairs, int th, int strict_lt, float ratio, int check_ori, int* d_match_t_of_q,
                            int* d_nmatches, void* stream) {
  if (!h || !d_desc || !d_counts || !d_pairs || !d_match_t_of_q || !d_nmatches || cap <= 0 || cap > kMaxTrain || npairs < 0) {
    set_err("orbm_match_pairs_device: bad arguments");
    return -2;
  }
  if (check_ori && !d_kps) {
    set_err("orbm_match_pairs_device: check_ori needs the keypoint array");
    return -2;
  }
  if (npairs == 0) return 0;
  HIP_TRYM(hipSetDevice(h->device));
  const size_t np = (size_t)npairs * cap;
  if (h->topk.ensure(np * kTopK) || h->nvalid.ensure(np) || h->binOf.ensure(np) || h->mt.ensure(np)) return -3;
  GroupArgs a;
  memset(&a, 0, sizeof(a));
  a.q = d_desc, a.t = d_desc, a.nq = cap, a.nt = cap, a.ngroups = 1, a.nqe = cap;
  a.q_angle = d_kps ? &d_kps->angle : nullptr, a.t_angle = a.q_angle;
  a.q_astride = a.t_astride = sizeof(orbx_keypoint_t) / 4;
  a.th = th, a.strict_lt = strict_lt, a.check_ori = check_ori, a.ratio = ratio, a.dstar = prune_bound(th, ratio);
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p;
  a.match_t_of_q = d_match_t_of_q, a.match_q_of_t = h->mt.p, a.nmatches = d_nmatches;
  a.pairs = d_pairs, a.counts = d_counts, a.cap = cap;
  if (h->dbgOn && !h->dbg.ensure(4)) a.dbg = h->dbg.p;
  return launch_groups(h, a, cap, stream ? (hipStream_t)stream : h->stream, npairs);
}

// SearchByBoW(KeyFrame*, Frame&, ...) proper for the frame pairs of an extraction batch: groups from the FeatureVectors that
// orbv_transform_batch_device produced (see include/orbx.h).  Enqueue only.
int orbm_match_pairs_bow_device(orbm_t* h, const uint8_t* d_desc, const orbx_keypoint_t* d_kps, const int* d_counts, int cap, const int* d_pairs,
                                int npairs, const uint32_t* d_fv_nodes, const int* d_fv_off, const uint32_t* d_fv_feat, const int* d_n_nodes,
                                int th, int strict_lt, float ratio, int check_ori, int* d_match_t_of_q, int* d_nmatches, void* stream) {
  if (!h || !d_desc || !d_counts || !d_pairs || !d_fv_nodes || !d_fv_off || !d_fv_feat || !d_n_nodes || !d_match_t_of_q || !d_nmatches ||
      cap <= 0 || npairs < 0 || cap > kMaxTrain) {
    set_err("orbm_match_pairs_bow_device: bad arguments");
    return -2;
  }
  if (check_ori && !d_kps) {
    set_err("orbm_match_pairs_bow_device: check_ori needs the keypoint array");
    return -2;
  }
  if (npairs == 0) return 0;
  HIP_TRYM(hipSetDevice(h->device));
  hipStream_t  s  = stream ? (hipStream_t)stream : h->stream;
  const size_t np = (size_t)npairs * cap;
  if (h->topk.ensure(np * kTopK) || h->nvalid.ensure(np) || h->binOf.ensure(np) || h->mt.ensure(np) || h->toff.ensure(np) || h->tidx.ensure(np))
    return -3;
  hipLaunchKernelGGL(k_bow_pair_groups, dim3((cap + 255) / 256, npairs), dim3(256), 0, s, d_pairs, cap, d_fv_nodes, d_fv_off, d_n_nodes, h->toff.p,
                     h->tidx.p);
  GroupArgs a;
  memset(&a, 0, sizeof(a));
  a.q = d_desc, a.t = d_desc, a.nq = cap, a.nt = cap, a.ngroups = 1, a.nqe = cap;
  a.q_angle = d_kps ? &d_kps->angle : nullptr, a.t_angle = a.q_angle;
  a.q_astride = a.t_astride = sizeof(orbx_keypoint_t) / 4;
  a.th = th, a.strict_lt = strict_lt, a.check_ori = check_ori, a.ratio = ratio, a.dstar = prune_bound(th, ratio);
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p;
  a.match_t_of_q = d_match_t_of_q, a.match_q_of_t = h->mt.p, a.nmatches = d_nmatches;
  a.pairs = d_pairs, a.counts = d_counts, a.cap = cap;
  a.fvFeat = d_fv_feat, a.fvOff = d_fv_off, a.fvNNodes = d_n_nodes, a.grpLo = h->toff.p, a.grpHi = h->tidx.p;
  return launch_groups(h, a, cap, s, npairs);
}

int orbm_match_window(orbm_t* h, const uint8_t* d1, const orbx_keypoint_t* kp1, int n1, const uint8_t* d2, const orbx_keypoint_t* kp2,
                      int n2, float* prev_matched, float min_x, float max_x, float min_y, float max_y, int window, int th, float ratio,
                      int check_ori, int* matches12) {
  if (!h || n1 < 0 || n2 < 0 || !matches12 || (n1 > 0 && (!d1 || !kp1 || !prev_matched)) || (n2 > 0 && (!d2 || !kp2)) ||
      !(max_x > min_x) || !(max_y > min_y)) {
    set_err("orbm_match_window: bad arguments");
    return -2;
  }
  if (n2 > kMaxTrainW) {
    set_err("orbm_match_window: more than 16384 train features");
    return -2;
  }
  HIP_TRYM(hipSetDevice(h->device));
  // Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:524-553, 867-877): candidate generator, host side
  // (SURVEY.md section 8(a) A14).  Scan order of GetFeaturesInArea = cells (ix, iy) ascending, insertion order inside.
  const float      gwInv = static_cast<float>(kGridCols) / (max_x - min_x);
  const float      ghInv = static_cast<float>(kGridRows) / (max_y - min_y);
  std::vector<int> cellCount(kGridCols * kGridRows + 1, 0), cellOf(n2, -1);
  for (int i = 0; i < n2; i++) {
    const int posX = (int)std::round((kp2[i].x - min_x) * gwInv);
    const int posY = (int)std::round((kp2[i].y - min_y) * ghInv);
    if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) continue;
    if (kp2[i].octave != 0) continue;  // only level-0 queries exist (:688) and they ask for level 0 only (:692-693)
    cellOf[i] = posX * kGridRows + posY;
    cellCount[cellOf[i] + 1]++;
  }
  for (int c = 0; c < kGridCols * kGridRows; c++) cellCount[c + 1] += cellCount[c];
  const int            ng = cellCount[kGridCols * kGridRows];
  std::vector<int>     gidx(ng + 1);
  std::vector<uint8_t> gcx(ng + 1), gcy(ng + 1);
  {
    std::vector<int> fill(cellCount.begin(), cellCount.end() - 1);
    for (int i = 0; i < n2; i++)
      if (cellOf[i] >= 0) {
        const int p = fill[cellOf[i]]++;
        gidx[p]     = i;
        gcx[p]      = (uint8_t)(cellOf[i] / kGridRows);
        gcy[p]      = (uint8_t)(cellOf[i] % kGridRows);
      }
  }
  hipStream_t s = h->stream;
  if (h->dq.ensure((size_t)n1 * 32 + 32) || h->dt.ensure((size_t)n2 * 32 + 32) || h->kp1.ensure(n1 + 1) || h->kp2.ensure(n2 + 1) ||
      h->gidx.ensure(ng + 1) || h->gcx.ensure(ng + 1) || h->gcy.ensure(ng + 1) || h->prev.ensure((size_t)2 * n1 + 2) ||
      h->topk.ensure((size_t)(n1 + 1) * kTopK) || h->nvalid.ensure(n1 + 1) || h->binOf.ensure(n1 + 1) || h->mq.ensure(n1 + 1) ||
      h->mt.ensure(n2 + 1) || h->nm.ensure(1))
    return -3;
  auto up = [&](void* d, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRYM(up(h->dq.p, d1, (size_t)n1 * 32));
  HIP_TRYM(up(h->dt.p, d2, (size_t)n2 * 32));
  HIP_TRYM(up(h->kp1.p, kp1, (size_t)n1 * sizeof(orbx_keypoint_t)));
  HIP_TRYM(up(h->kp2.p, kp2, (size_t)n2 * sizeof(orbx_keypoint_t)));
  HIP_TRYM(up(h->gidx.p, gidx.data(), (size_t)ng * 4));
  HIP_TRYM(up(h->gcx.p, gcx.data(), ng));
  HIP_TRYM(up(h->gcy.p, gcy.data(), ng));
  HIP_TRYM(up(h->prev.p, prev_matched, (size_t)n1 * 8));
  WindowArgs a;
  memset(&a, 0, sizeof(a));
  a.d1 = h->dq.p, a.d2 = h->dt.p, a.kp1 = h->kp1.p, a.kp2 = h->kp2.p, a.n1 = n1, a.n2 = n2, a.ng = ng;
  a.g_idx = h->gidx.p, a.g_cx = h->gcx.p, a.g_cy = h->gcy.p, a.prev = h->prev.p;
  a.minX = min_x, a.minY = min_y, a.gwInv = gwInv, a.ghInv = ghInv, a.r = (float)window;
  a.th = th, a.check_ori = check_ori, a.ratio = ratio, a.dstar = prune_bound(th, ratio);
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.m12 = h->mq.p, a.m21 = h->mt.p, a.binOf = h->binOf.p, a.nmatches = h->nm.p;
  if (n1 > 0) hipLaunchKernelGGL(k_window_topk, dim3((n1 + 3) / 4), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_window_resolve, dim3(1), dim3(64), 0, s, a);
  HIP_TRYM(hipGetLastError());
  int nm = 0;
  HIP_TRYM(hipMemcpyAsync(&nm, h->nm.p, 4, hipMemcpyDeviceToHost, s));
  if (n1) {
    HIP_TRYM(hipMemcpyAsync(matches12, h->mq.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRYM(hipMemcpyAsync(prev_matched, h->prev.p, (size_t)n1 * 8, hipMemcpyDeviceToHost, s));
  }
  HIP_TRYM(hipStreamSynchronize(s));
  return nm;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) at array level (see include/orbx.h)
static int proj_single(orbm_t* h, int mode, const float* inv_sigma2, const float* radius1, float ratio, const uint8_t* last_desc, const float* last_uv, const float* last_invz,
                       const int* last_octave, const float* last_angle, const uint8_t* last_valid, int n1, const uint8_t* cur_desc,
                       const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                       const float* scale_factors, int nlevels, float th, int dir, float mbf, float min_x, float max_x, float min_y,
                       float max_y, int th_dist, int check_ori, int* cur_of_last, int* last_of_cur);

int orbm_search_by_projection(orbm_t* h, const uint8_t* last_desc, const float* last_uv, const float* last_invz, const int* last_octave,
                              const float* last_angle, const uint8_t* last_valid, int n1, const uint8_t* cur_desc,
                              const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                              const float* scale_factors, int nlevels, float th, int dir, float mbf, float min_x, float max_x,
                              float min_y, float max_y, int th_dist, int check_ori, int* cur_of_last, int* last_of_cur) {
  if (n1 > 0 && !last_angle) {
    set_err("orbm_search_by_projection: bad arguments");
    return -2;
  }
  return proj_single(h, 0, nullptr, nullptr, 0.f, last_desc, last_uv, last_invz, last_octave, last_angle, last_valid, n1, cur_desc, cur_kp, cur_u_right,
                     cur_occupied, n2, scale_factors, nlevels, th, dir, mbf, min_x, max_x, min_y, max_y, th_dist, check_ori, cur_of_last, last_of_cur);
}

// SearchByProjection(Frame& F, const std::vector<MapPoint*>&, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:38-227) on arrays
int orbm_search_by_projection_map(orbm_t* h, const uint8_t* mp_desc, const float* mp_uv, const float* mp_proj_xr, const int* mp_level,
                                  const float* mp_radius, const uint8_t* mp_valid, int n1, const uint8_t* cur_desc,
                                  const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                                  const float* scale_factors, int nlevels, float ratio, float min_x, float max_x, float min_y, float max_y,
                                  int th_dist, int* cur_of_mp, int* mp_of_cur) {
  if (n1 > 0 && !mp_radius) {
    set_err("orbm_search_by_projection_map: bad arguments");
    return -2;
  }
  return proj_single(h, 1, nullptr, mp_radius, ratio, mp_desc, mp_uv, mp_proj_xr, mp_level, nullptr, mp_valid, n1, cur_desc, cur_kp, cur_u_right, cur_occupied,
                     n2, scale_factors, nlevels, 1.0f, 0, 0.f, min_x, max_x, min_y, max_y, th_dist, 0, cur_of_mp, mp_of_cur);
}

static int proj_single(orbm_t* h, int mode, const float* inv_sigma2, const float* radius1, float ratio, const uint8_t* last_desc, const float* last_uv, const float* last_invz,
                       const int* last_octave, const float* last_angle, const uint8_t* last_valid, int n1, const uint8_t* cur_desc,
                       const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                       const float* scale_factors, int nlevels, float th, int dir, float mbf, float min_x, float max_x, float min_y,
                       float max_y, int th_dist, int check_ori, int* cur_of_last, int* last_of_cur) {
  if (!h || n1 < 0 || n2 < 0 || !cur_of_last || (n1 > 0 && (!last_desc || !last_uv || !last_octave)) ||
      (n2 > 0 && (!cur_desc || !cur_kp)) || !scale_factors || nlevels < 1 || nlevels > 64 || dir < 0 || dir > 3 || !(max_x > min_x) ||
      !(max_y > min_y) || (cur_u_right && n1 > 0 && !last_invz)) {
    set_err("orbm_search_by_projection: bad arguments");
    return -2;
  }
  if (n2 > kMaxTrainW) {
    set_err("orbm_search_by_projection: more than 16384 current-frame features");
    return -2;
  }
  for (int i = 0; i < n1; i++)
    if ((!last_valid || last_valid[i]) && (last_octave[i] < 0 || last_octave[i] >= nlevels)) {
      set_err("orbm_search_by_projection: octave outside the scale table");
      return -2;
    }
  HIP_TRYM(hipSetDevice(h->device));
  // Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:524-553, 867-877), host side as in orbm_match_window: the scan order
  // of GetFeaturesInArea is cells (ix, iy) ascending, insertion (= index) order inside, so one global order serves every window.
  const float      gwInv = static_cast<float>(kGridCols) / (max_x - min_x);
  const float      ghInv = static_cast<float>(kGridRows) / (max_y - min_y);
  std::vector<int> cellCount(kGridCols * kGridRows + 1, 0), cellOf(n2, -1);
  for (int i = 0; i < n2; i++) {
    const int posX = (int)std::round((cur_kp[i].x - min_x) * gwInv);
    const int posY = (int)std::round((cur_kp[i].y - min_y) * ghInv);
    if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) continue;
    cellOf[i] = posX * kGridRows + posY;
    cellCount[cellOf[i] + 1]++;
  }
  for (int c = 0; c < kGridCols * kGridRows; c++) cellCount[c + 1] += cellCount[c];
  const int            ng = cellCount[kGridCols * kGridRows];
  std::vector<int>     gidx(ng + 1);
  std::vector<uint8_t> gcx(ng + 1), gcy(ng + 1);
  {
    std::vector<int> fill(cellCount.begin(), cellCount.end() - 1);
    for (int i = 0; i < n2; i++)
      if (cellOf[i] >= 0) {
        const int p = fill[cellOf[i]]++;
        gidx[p]     = i;
        gcx[p]      = (uint8_t)(cellOf[i] / kGridRows);
        gcy[p]      = (uint8_t)(cellOf[i] % kGridRows);
      }
  }
  hipStream_t  s  = h->stream;
  const size_t inBytes = (size_t)n1 * (32 + 8 + 4 + 4 + 4 + 4 + 1) + (size_t)n2 * (32 + sizeof(orbx_keypoint_t) + 4 + 1) + (size_t)ng * 6 + (size_t)2 * nlevels * 4 + (kGridCols + 1) * 4 + 16 * 25;
  const size_t outBytes = (size_t)(n1 + n2 + 4) * 4 + 64;
  if (h->in.reserve(inBytes) || h->out.reserve(outBytes) || h->topk.ensure((size_t)(n1 + 1) * kTopK) || h->nvalid.ensure(n1 + 1) ||
      h->binOf.ensure(n1 + 1))
    return -3;
  ProjArgs a;
  memset(&a, 0, sizeof(a));
  Arena& in = h->in;
  a.d1 = in.put(last_desc, (size_t)n1 * 32), a.d2 = in.put(cur_desc, (size_t)n2 * 32), a.kp2 = in.put(cur_kp, n2);
  a.g_idx = in.put(gidx.data(), ng), a.g_cx = in.put(gcx.data(), ng), a.g_cy = in.put(gcy.data(), ng);
  {
    int cs[kGridCols + 1];
    for (int c = 0; c <= kGridCols; c++) cs[c] = cellCount[c * kGridRows];
    a.colStart = in.put(cs, kGridCols + 1);
  }
  a.uv = in.put(last_uv, (size_t)2 * n1), a.angle1 = last_angle ? in.put(last_angle, n1) : nullptr, a.invz = last_invz ? in.put(last_invz, n1) : nullptr;
  a.radius1 = radius1 ? in.put(radius1, n1) : nullptr, a.mode = mode, a.ratio = ratio;
  a.invSigma2 = inv_sigma2 ? in.put(inv_sigma2, nlevels) : nullptr;
  if (mode == 1) {  // a runner-up at distance D vetoes a best d1 <= th only if d1 > ratio * D: none can from the first D with !(th > ratio * D)
    int D = th_dist + 1;
    while (D < 257 && (float)th_dist > ratio * (float)D) D++;
    a.dstar = D;
  }
  a.sf = in.put(scale_factors, nlevels), a.uRight = cur_u_right ? in.put(cur_u_right, n2) : nullptr, a.oct1 = in.put(last_octave, n1);
  a.valid1 = last_valid ? in.put(last_valid, n1) : nullptr, a.occupied = cur_occupied ? in.put(cur_occupied, n2) : nullptr;
  if (!in.ok() || !h->out.ok()) return -3;
  HIP_TRYM(hipMemcpyAsync(in.d, in.h, in.off, hipMemcpyHostToDevice, s));
  Arena& out = h->out;
  a.nmatches = out.put((const int*)nullptr, 4), a.m12 = out.put((const int*)nullptr, n1), a.m21 = out.put((const int*)nullptr, n2);
  if (!out.ok()) return -3;
  a.n1 = n1, a.n2 = n2, a.ng = ng, a.th = th, a.mbf = mbf, a.minX = min_x, a.minY = min_y, a.gwInv = gwInv, a.ghInv = ghInv, a.dir = dir;
  a.thDist = th_dist, a.check_ori = check_ori, a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p;
  if (n1 > 0) hipLaunchKernelGGL(k_proj_topk, dim3((n1 + 3) / 4), dim3(256), 0, s, a);
  if (mode == 2)
    hipLaunchKernelGGL(k_proj_take_best, dim3(1), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(64), 0, s, a);
  HIP_TRYM(hipGetLastError());
  HIP_TRYM(hipMemcpyAsync(out.h, out.d, out.off, hipMemcpyDeviceToHost, s));
  HIP_TRYM(hipStreamSynchronize(s));
  const int nm = *out.host_of(a.nmatches);
  if (n1) memcpy(cur_of_last, out.host_of(a.m12), (size_t)n1 * 4);
  if (n2 && last_of_cur && mode != 2) memcpy(last_of_cur, out.host_of(a.m21), (size_t)n2 * 4);
  return nm;
}

// ---------------------------------------------------------------------------------------------------------------
// SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) for a two-camera frame (F.Nleft != -1, src/ORBmatcher.cc:139-146 and
// 152-226).  Every map point searches the left grid and then the right grid; an accepted left match also hands the point to the
// stereo partner mvLeftToRightMatch[best] + Nleft, an accepted right match to mvRightToLeftMatch[best]; a ratio veto in the left
// block (`continue`, :138-140) skips the point's right search.  The two window searches per point are independent of the claims, so
// their sorted candidate lists come from the parallel top-K pass of the one-camera form (k_proj_topk, mode 1, once per camera);
// one wave then walks the points in order with the occupancy of all N features in LDS, reading 64 points' lists at a time.
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct MapTwoCamArgs {
  ProjArgs   L, R;  // per-camera view (mode 1): R.d2 / R.kp2 / R.occupied start at the first right feature
  int        nleft, n;
  const int *l2r, *r2l;
  int *      mp_of_cur, *nmatches;
};

constexpr uint32_t kTcEmpty = 0xffffffffu;  // list entry: distance << 20 | level << 14 | feature (camera-relative)

// Best and second-best free candidate of one camera for point i, exactly: from the staged list when it decides, else by an exact
// rescan of the window with the occupancy filter.  Returns e1 / e2 (kTcEmpty = none).
__device__ __forceinline__ void tc_pick(const ProjArgs& a, int i, const uint32_t* list, int nv, const uint8_t* occ, int occBase, uint32_t& e1, uint32_t& e2) {
  const int      lane = threadIdx.x & 63;
  const uint32_t e    = lane < kTopK ? list[lane] : kTcEmpty;
  const bool     free_ = e != kTcEmpty && !occ[occBase + (int)(e & 0x3fffu)];
  const unsigned long long fm = __ballot(free_), em = __ballot(e != kTcEmpty);
  const int      len = __popcll(em);
  const bool     truncated = nv > len;
  const int      j1 = fm ? __ffsll((long long)fm) - 1 : -1;
  const unsigned long long fm2 = j1 >= 0 ? fm & (fm - 1) : 0ull;
  const int      j2 = fm2 ? __ffsll((long long)fm2) - 1 : -1;
  e1 = j1 >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)e, j1) : kTcEmpty;
  e2 = j2 >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)e, j2) : kTcEmpty;
  bool need = false;
  if (truncated) {
    if (j1 < 0)
      need = true;  // everything listed is taken, more candidates exist
    else if (j2 < 0) {
      // one free candidate; the runner-up is somewhere beyond the list, at distance >= dK: it matters only if it could veto
      const int d1 = (int)(e1 >> 20), dK = (int)((uint32_t)__builtin_amdgcn_readlane((int)e, len - 1) >> 20);
      need         = d1 <= a.thDist && (float)d1 > a.ratio * (float)dK;
    }
  }
  if (!need) return;
  const ProjQuery q  = proj_query(a, i);
  const ProjArgs& ar = a;
  uint32_t        out[kTopK];
  int             nv2, pLo, pHi;
  proj_range(a, q, &pLo, &pHi);
  wave_topk(
      a.d1 + (size_t)i * 32, a.d2, pLo, pHi,
      [&ar, &q, occ, occBase](int pos) {
        const int i2 = proj_candidate(ar, q, pos);
        return (i2 >= 0 && occ[occBase + i2]) ? -1 : i2;
      },
      [thd = ar.dstar - 1](int, int d) { return d <= thd; }, out, &nv2);
  auto conv = [&](uint32_t key) -> uint32_t {
    if (key == kEmpty) return kTcEmpty;
    const int f = a.g_idx[key & kPosMask];
    return ((key >> kPosBits) << 20) | ((uint32_t)(a.kp2[f].octave & 63) << 14) | (uint32_t)f;
  };
  e1 = conv(out[0]), e2 = conv(out[1]);
}

__global__ __launch_bounds__(64) void k_map_two_cam_resolve(MapTwoCamArgs a) {
  __shared__ uint8_t  occ[kMaxTrainW];
  __shared__ uint32_t sList[64][2][kTopK];
  __shared__ int      sNv[64][2];
  const int           lane = threadIdx.x, nright = a.n - a.nleft;
  for (int i = lane; i < a.nleft; i += 64) occ[i] = a.L.occupied ? a.L.occupied[i] : 0;
  for (int i = lane; i < nright; i += 64) occ[a.nleft + i] = a.R.occupied ? a.R.occupied[i] : 0;
  for (int i = lane; i < a.n; i += 64) a.mp_of_cur[i] = -1;
  __syncthreads();
  int nmatches = 0;
  for (int c0 = 0; c0 < a.L.n1; c0 += 64) {
    {  // this lane's point: both lists, converted to (distance, level, feature); -1 in sNv = the camera does not see the point
      const int i = c0 + lane;
#pragma unroll
      for (int cam = 0; cam < 2; cam++) {
        const ProjArgs& v  = cam ? a.R : a.L;
        const bool      on = i < v.n1 && v.valid1 && v.valid1[i];
        sNv[lane][cam]     = on ? v.nvalid[i] : -1;
#pragma unroll
        for (int j = 0; j < kTopK; j++) {
          uint32_t e = kTcEmpty;
          if (on) {
            const uint32_t key = v.topk[(size_t)i * kTopK + j];
            if (key != kEmpty) {
              const int f = v.g_idx[key & kPosMask];
              e           = ((key >> kPosBits) << 20) | ((uint32_t)(v.kp2[f].octave & 63) << 14) | (uint32_t)f;
            }
          }
          sList[lane][cam][j] = e;
        }
      }
    }
    __syncthreads();
    const int np = min(64, a.L.n1 - c0);
    for (int j = 0; j < np; j++) {
      const int i = c0 + j;
      const int nvL = sNv[j][0], nvR = sNv[j][1];
      if (nvL < 0 && nvR < 0) continue;  // :50-62
      bool skipRight = false;
      if (nvL >= 0) {
        uint32_t e1, e2;
        tc_pick(a.L, i, sList[j][0], nvL, occ, 0, e1, e2);
        if (e1 != kTcEmpty) {
          const int d1 = (int)(e1 >> 20), l1 = (int)((e1 >> 14) & 63), f1 = (int)(e1 & 0x3fffu);
          if (d1 <= a.L.thDist) {  // :137
            const bool veto = e2 != kTcEmpty && l1 == (int)((e2 >> 14) & 63) && (float)d1 > a.L.ratio * (float)(int)(e2 >> 20);
            if (veto) {
              skipRight = true;  // `continue` over the map points (:138-140)
            } else {
              const int partner = a.l2r ? a.l2r[f1] : -1;
              if (lane == 0) {
                occ[f1] = 1, a.mp_of_cur[f1] = i;
                if (partner != -1) occ[a.nleft + partner] = 1, a.mp_of_cur[a.nleft + partner] = i;  // :145-149
              }
              nmatches += partner != -1 ? 2 : 1;
              __syncthreads();
            }
          }
        }
      }
      if (skipRight || nvR < 0) continue;
      uint32_t e1, e2;
      tc_pick(a.R, i, sList[j][1], nvR, occ, a.nleft, e1, e2);
      if (e1 == kTcEmpty) continue;
      const int d1 = (int)(e1 >> 20), l1 = (int)((e1 >> 14) & 63), f1 = (int)(e1 & 0x3fffu);
      if (d1 > a.R.thDist) continue;
      if (e2 != kTcEmpty && l1 == (int)((e2 >> 14) & 63) && (float)d1 > a.R.ratio * (float)(int)(e2 >> 20)) continue;  // :209-211
      const int partner = a.r2l ? a.r2l[f1] : -1;
      if (lane == 0) {
        if (partner != -1) occ[partner] = 1, a.mp_of_cur[partner] = i;  // :214-218
        occ[a.nleft + f1] = 1, a.mp_of_cur[a.nleft + f1] = i;
      }
      nmatches += partner != -1 ? 2 : 1;
      __syncthreads();
    }
    __syncthreads();
  }
  if (lane == 0) *a.nmatches = nmatches;
}

}  // namespace

extern "C" {

int orbm_search_by_projection_map_two_cameras(orbm_t* h, const uint8_t* mp_desc, int n1, const float* uv_l, const int* level_l, const float* radius_l,
                                              const uint8_t* valid_l, const float* uv_r, const int* level_r, const float* radius_r,
                                              const uint8_t* valid_r, const uint8_t* cur_desc, const orbx_keypoint_t* cur_kp, int n, int nleft,
                                              const uint8_t* cur_occupied, const int* left_to_right, const int* right_to_left,
                                              const float* scale_factors, int nlevels, float ratio, float min_x, float max_x, float min_y, float max_y,
                                              int th_dist, int* mp_of_cur) {
  if (!h || n1 < 0 || n < 0 || nleft < 0 || nleft > n || !mp_of_cur || (n1 > 0 && (!mp_desc || !uv_l || !level_l || !radius_l || !valid_l || !uv_r || !level_r ||
      !radius_r || !valid_r)) || (n > 0 && (!cur_desc || !cur_kp)) || !scale_factors || nlevels < 1 || nlevels > 64 || !(max_x > min_x) || !(max_y > min_y)) {
    set_err("orbm_search_by_projection_map_two_cameras: bad arguments");
    return -2;
  }
  if (n > kMaxTrainW) {
    set_err("orbm_search_by_projection_map_two_cameras: more than 16384 frame features");
    return -2;
  }
  const int nright = n - nleft;
  for (int i = 0; i < n1; i++)
    if ((valid_l[i] && (level_l[i] < 0 || level_l[i] >= nlevels)) || (valid_r[i] && (level_r[i] < 0 || level_r[i] >= nlevels))) {
      set_err("orbm_search_by_projection_map_two_cameras: predicted level outside the scale table");
      return -2;
    }
  for (int i = 0; i < nleft; i++)
    if (left_to_right && (left_to_right[i] < -1 || left_to_right[i] >= nright)) {
      set_err("orbm_search_by_projection_map_two_cameras: left_to_right out of range");
      return -2;
    }
  for (int i = 0; i < nright; i++)
    if (right_to_left && (right_to_left[i] < -1 || right_to_left[i] >= nleft)) {
      set_err("orbm_search_by_projection_map_two_cameras: right_to_left out of range");
      return -2;
    }
  HIP_TRYM(hipSetDevice(h->device));
  // Frame::AssignFeaturesToGrid for the two grids (src/Frame.cc:524-553): one scan-ordered list per camera
  const float gwInv = static_cast<float>(kGridCols) / (max_x - min_x), ghInv = static_cast<float>(kGridRows) / (max_y - min_y);
  struct CamGrid {
    std::vector<int>     gidx;
    std::vector<uint8_t> gcx, gcy;
    int                  ng = 0;
    int                  colStart[kGridCols + 1];
  } grids[2];
  for (int cam = 0; cam < 2; cam++) {
    const orbx_keypoint_t* kp = cur_kp + (cam ? nleft : 0);
    const int              m  = cam ? nright : nleft;
    std::vector<int>       cellCount(kGridCols * kGridRows + 1, 0), cellOf(m, -1);
    for (int i = 0; i < m; i++) {
      const int posX = (int)std::round((kp[i].x - min_x) * gwInv), posY = (int)std::round((kp[i].y - min_y) * ghInv);
      if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) continue;
      cellOf[i] = posX * kGridRows + posY;
      cellCount[cellOf[i] + 1]++;
    }
    for (int c = 0; c < kGridCols * kGridRows; c++) cellCount[c + 1] += cellCount[c];
    CamGrid& G = grids[cam];
    G.ng       = cellCount[kGridCols * kGridRows];
    for (int c = 0; c <= kGridCols; c++) G.colStart[c] = cellCount[c * kGridRows];
    G.gidx.resize(G.ng + 1), G.gcx.resize(G.ng + 1), G.gcy.resize(G.ng + 1);
    std::vector<int> fill(cellCount.begin(), cellCount.end() - 1);
    for (int i = 0; i < m; i++)
      if (cellOf[i] >= 0) {
        const int p2 = fill[cellOf[i]]++;
        G.gidx[p2]   = i;
        G.gcx[p2]    = (uint8_t)(cellOf[i] / kGridRows);
        G.gcy[p2]    = (uint8_t)(cellOf[i] % kGridRows);
      }
  }
  hipStream_t  s       = h->stream;
  const size_t inBytes = (size_t)n1 * (32 + 2 * (8 + 4 + 4 + 1)) + (size_t)n * (32 + sizeof(orbx_keypoint_t) + 1 + 4 + 6) + (size_t)nlevels * 4 + 2 * (kGridCols + 1) * 4 + 16 * 34;
  if (h->in.reserve(inBytes) || h->out.reserve((size_t)(n + 4) * 4 + 64) || h->topk.ensure((size_t)(2 * n1 + 2) * kTopK) || h->nvalid.ensure(2 * n1 + 2)) return -3;
  Arena&        in = h->in;
  MapTwoCamArgs a;
  memset(&a, 0, sizeof(a));
  const uint8_t*         dDesc = in.put(cur_desc, (size_t)n * 32);
  const orbx_keypoint_t* dKp   = in.put(cur_kp, n);
  const uint8_t*         dOcc  = cur_occupied ? in.put(cur_occupied, n) : nullptr;
  const uint8_t*         dMp   = in.put(mp_desc, (size_t)n1 * 32);
  const float*           dSf   = in.put(scale_factors, nlevels);
  int D = th_dist + 1;  // a runner-up at distance D vetoes a best d1 <= th only if d1 > ratio * D (see proj_single)
  while (D < 257 && (float)th_dist > ratio * (float)D) D++;
  for (int cam = 0; cam < 2; cam++) {
    ProjArgs& v = cam ? a.R : a.L;
    CamGrid&  G = grids[cam];
    v.d1 = dMp, v.d2 = dDesc + (size_t)(cam ? nleft : 0) * 32, v.kp2 = dKp + (cam ? nleft : 0);
    v.occupied = dOcc ? dOcc + (cam ? nleft : 0) : nullptr;
    v.g_idx = in.put(G.gidx.data(), G.ng), v.g_cx = in.put(G.gcx.data(), G.ng), v.g_cy = in.put(G.gcy.data(), G.ng);
    v.colStart = in.put(G.colStart, kGridCols + 1);
    v.uv = in.put(cam ? uv_r : uv_l, (size_t)2 * n1), v.radius1 = in.put(cam ? radius_r : radius_l, n1), v.oct1 = in.put(cam ? level_r : level_l, n1);
    v.valid1 = in.put(cam ? valid_r : valid_l, n1);
    v.mode = 1, v.ratio = ratio, v.dstar = D, v.sf = dSf;
    v.n1 = n1, v.n2 = cam ? nright : nleft, v.ng = G.ng, v.th = 1.0f, v.minX = min_x, v.minY = min_y, v.gwInv = gwInv, v.ghInv = ghInv;
    v.thDist = th_dist, v.topk = h->topk.p + (size_t)cam * n1 * kTopK, v.nvalid = h->nvalid.p + (size_t)cam * n1;
  }
  a.l2r = left_to_right ? in.put(left_to_right, nleft) : nullptr, a.r2l = right_to_left ? in.put(right_to_left, nright) : nullptr;
  a.nleft = nleft, a.n = n;
  if (!in.ok()) return -3;
  HIP_TRYM(hipMemcpyAsync(in.d, in.h, in.off, hipMemcpyHostToDevice, s));
  Arena& out = h->out;
  a.nmatches = out.put((const int*)nullptr, 4), a.mp_of_cur = out.put((const int*)nullptr, n);
  if (!out.ok()) return -3;
  if (n1 > 0) {
    hipLaunchKernelGGL(k_proj_topk, dim3((n1 + 3) / 4), dim3(256), 0, s, a.L);
    hipLaunchKernelGGL(k_proj_topk, dim3((n1 + 3) / 4), dim3(256), 0, s, a.R);
  }
  hipLaunchKernelGGL(k_map_two_cam_resolve, dim3(1), dim3(64), 0, s, a);
  HIP_TRYM(hipGetLastError());
  HIP_TRYM(hipMemcpyAsync(out.h, out.d, out.off, hipMemcpyDeviceToHost, s));
  HIP_TRYM(hipStreamSynchronize(s));
  if (n) memcpy(mp_of_cur, out.host_of(a.mp_of_cur), (size_t)n * 4);
  return *out.host_of(a.nmatches);
}

// the windowed search of SearchByProjection(KeyFrame*, Sim3f&, ...) / Fuse / SearchBySim3 on arrays (include/orbx.h)
int orbm_search_in_windows(orbm_t* h, const uint8_t* q_desc, const float* q_uv, const float* q_ur, const int* q_level, const uint8_t* q_valid, int n1,
                           const uint8_t* t_desc, const orbx_keypoint_t* t_kp, const float* t_u_right, const uint8_t* t_occupied, int n2,
                           const float* scale_factors, const float* inv_level_sigma2, int nlevels, float th, int claim, float min_x, float max_x,
                           float min_y, float max_y, int th_dist, int* t_of_q) {
  if ((claim != 0 && claim != 1) || (claim && inv_level_sigma2) || (inv_level_sigma2 && t_u_right && n1 > 0 && !q_ur)) {
    set_err("orbm_search_in_windows: bad arguments");
    return -2;
  }
  if (inv_level_sigma2 && t_kp)  // the gate indexes mvInvLevelSigma2 by the keypoint's level
    for (int i = 0; i < n2; i++)
      if (t_kp[i].octave < 0 || t_kp[i].octave >= nlevels) {
        set_err("orbm_search_in_windows: keypoint octave outside the scale table");
        return -2;
      }
  if (claim)  // mode 0 with the [level - 1, level] window: occupied features are skipped, a match closes its feature
    return proj_single(h, 0, nullptr, nullptr, 0.f, q_desc, q_uv, nullptr, q_level, nullptr, q_valid, n1, t_desc, t_kp, nullptr, t_occupied, n2,
                       scale_factors, nlevels, th, 3, 0.f, min_x, max_x, min_y, max_y, th_dist, 0, t_of_q, nullptr);
  return proj_single(h, 2, inv_level_sigma2, nullptr, 0.f, q_desc, q_uv, q_ur, q_level, nullptr, q_valid, n1, t_desc, t_kp,
                     inv_level_sigma2 ? t_u_right : nullptr, nullptr, n2, scale_factors, nlevels, th, 0, 0.f, min_x, max_x, min_y, max_y, th_dist, 0,
                     t_of_q, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:916-1147), keyframes of one pinhole camera.  The reference never sets
// vbMatched2, so every feature of KF1 searches its node's features of KF2 on its own: one wave per list entry, candidates over
// the lanes, winner = smallest distance among the candidates that pass the gates, the LAST of equals in list order (a candidate
// at `dist == bestDist` replaces the holder, :1019).
// ---------------------------------------------------------------------------------------------------------------
struct TriArgs {
  const uint8_t *        d1, *d2, *mp1, *mp2, *st1, *st2;
  const orbx_keypoint_t *kp1, *kp2;
  const int *            q_idx, *q_grp, *t_off, *t_idx;
  const float *          sf, *sigma2;
  float                  F[9], ep[2];
  int                    n1, nqe, onlyStereo, coarse, th, check_ori;
  int *                  m12, *binOf, *nmatches;
};

__global__ __launch_bounds__(256) void k_tri_search(TriArgs a) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= a.nqe) return;
  const int idx1 = a.q_idx[e];
  if (a.mp1 && a.mp1[idx1]) return;  // :980-984
  const bool bStereo1 = a.st1 && a.st1[idx1];
  if (a.onlyStereo && !bStereo1) return;
  const orbx_keypoint_t k1 = a.kp1[idx1];
  const uint4           q0 = *reinterpret_cast<const uint4*>(a.d1 + (size_t)idx1 * 32);
  const uint4           q1 = *reinterpret_cast<const uint4*>(a.d1 + (size_t)idx1 * 32 + 16);
  // epipolar line of kp1 in the second image (src/CameraModels/Pinhole.cpp:158-161)
  const float la = k1.x * a.F[0] + k1.y * a.F[3] + a.F[6];
  const float lb = k1.x * a.F[1] + k1.y * a.F[4] + a.F[7];
  const float lc = k1.x * a.F[2] + k1.y * a.F[5] + a.F[8];
  const float den = la * la + lb * lb;
  const int   g = a.q_grp[e], t0 = a.t_off[g], t1 = a.t_off[g + 1];
  uint32_t    best = 0xffffffffu;  // distance << 20 | (0xfffff - position in the node's list): the minimum is the reference's winner
  for (int p = t0 + lane; p < t1; p += 64) {
    const int idx2 = a.t_idx[p];
    if (a.mp2 && a.mp2[idx2]) continue;  // :1007-1011
    const bool bStereo2 = a.st2 && a.st2[idx2];
    if (a.onlyStereo && !bStereo2) continue;
    const int dist = hamming256(q0, q1, a.d2 + (size_t)idx2 * 32);
    if (dist > a.th) continue;  // :1019
    const orbx_keypoint_t k2 = a.kp2[idx2];
    if (!bStereo1 && !bStereo2) {  // :1031-1037
      const float distex = a.ep[0] - k2.x, distey = a.ep[1] - k2.y;
      if (distex * distex + distey * distey < 100 * a.sf[k2.octave]) continue;
    }
    if (!a.coarse) {  // Pinhole::epipolarConstrain (:163-173)
      const float num = la * k2.x + lb * k2.y + lc;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)a.sigma2[k2.octave])) continue;
    }
    const uint32_t key = ((uint32_t)dist << 20) | (uint32_t)(0xfffff - (p - t0));
    best               = min(best, key);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
  if (lane == 0 && best != 0xffffffffu) {
    const int idx2 = a.t_idx[t0 + (0xfffff - (int)(best & 0xfffff))];
    a.m12[idx1]    = idx2;
    if (a.check_ori) a.binOf[idx1] = rot_bin(k1.angle, a.kp2[idx2].angle);  // :1085-1098
  }
}

// rotation histogram over the matches, three maxima, removal of the other bins (:1112-1129), and the count
__global__ __launch_bounds__(256) void k_tri_finish(TriArgs a) {
  __shared__ int hist[kHistoLen];
  __shared__ int total;
  const int      tid = threadIdx.x;
  if (tid < kHistoLen) hist[tid] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < a.n1; i += 256)
    if (a.m12[i] >= 0) {
      mine++;
      if (a.check_ori) atomicAdd(&hist[a.binOf[i]], 1);
    }
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (a.check_ori) {
    int i1, i2, i3;
    three_maxima(hist, kHistoLen, i1, i2, i3);
    int removed = 0;
    for (int i = tid; i < a.n1; i += 256)
      if (a.m12[i] >= 0) {
        const int b = a.binOf[i];
        if (b != i1 && b != i2 && b != i3) a.m12[i] = -1, removed++;
      }
    if (removed) atomicSub(&total, removed);
    __syncthreads();
  }
  if (tid == 0) *a.nmatches = total;
}

int orbm_search_for_triangulation(orbm_t* h, const uint8_t* d1, const orbx_keypoint_t* kp1, const uint8_t* mp1, const uint8_t* stereo1, int n1,
                                  const uint8_t* d2, const orbx_keypoint_t* kp2, const uint8_t* mp2, const uint8_t* stereo2, int n2, int ngroups,
                                  const int* q_off, const int* q_idx, const int* t_off, const int* t_idx, const float F12[9], const float ep[2],
                                  const float* scale_factors, const float* level_sigma2, int nlevels, int only_stereo, int coarse, int th_low,
                                  int check_ori, int* matches12) {
  if (!h || n1 < 0 || n2 < 0 || ngroups < 0 || !matches12 || (n1 > 0 && (!d1 || !kp1)) || (n2 > 0 && (!d2 || !kp2)) ||
      (ngroups > 0 && (!q_off || !q_idx || !t_off || !t_idx)) || !ep || !scale_factors || nlevels < 1 || nlevels > 64 || (!coarse && (!F12 || !level_sigma2))) {
    set_err("orbm_search_for_triangulation: bad arguments");
    return -2;
  }
  const int nqe = ngroups ? q_off[ngroups] : 0, nte = ngroups ? t_off[ngroups] : 0;
  for (int g = 0; g < ngroups; g++)
    if (q_off[g] < 0 || t_off[g] < 0 || q_off[g] > q_off[g + 1] || t_off[g] > t_off[g + 1] || t_off[g + 1] - t_off[g] > 0xfffff) {
      set_err("orbm_search_for_triangulation: offsets must be non-decreasing (at most 2^20 - 1 features under one node)");
      return -2;
    }
  std::vector<int> qgrp(nqe + 1);
  for (int g = 0; g < ngroups; g++)
    for (int e = q_off[g]; e < q_off[g + 1]; e++) {
      if (q_idx[e] < 0 || q_idx[e] >= n1) {
        set_err("orbm_search_for_triangulation: q_idx out of range");
        return -2;
      }
      qgrp[e] = g;
    }
  for (int e = 0; e < nte; e++)
    if (t_idx[e] < 0 || t_idx[e] >= n2 || kp2[t_idx[e]].octave < 0 || kp2[t_idx[e]].octave >= nlevels) {
      set_err("orbm_search_for_triangulation: t_idx or a keypoint octave out of range");
      return -2;
    }
  HIP_TRYM(hipSetDevice(h->device));
  hipStream_t  s       = h->stream;
  const size_t inBytes = (size_t)(n1 + n2) * (32 + sizeof(orbx_keypoint_t) + 2) + (size_t)(2 * nqe + nte + ngroups + 2) * 4 + 128 * 4 + 20 * 16;
  if (h->in.reserve(inBytes) || h->out.reserve((size_t)(n1 + 4) * 4 + 64) || h->binOf.ensure(n1 + 1)) return -3;
  TriArgs a;
  memset(&a, 0, sizeof(a));
  Arena& in = h->in;
  a.d1 = in.put(d1, (size_t)n1 * 32), a.d2 = in.put(d2, (size_t)n2 * 32), a.kp1 = in.put(kp1, n1), a.kp2 = in.put(kp2, n2);
  a.mp1 = mp1 ? in.put(mp1, n1) : nullptr, a.mp2 = mp2 ? in.put(mp2, n2) : nullptr;
  a.st1 = stereo1 ? in.put(stereo1, n1) : nullptr, a.st2 = stereo2 ? in.put(stereo2, n2) : nullptr;
  a.q_idx = in.put(q_idx, nqe), a.q_grp = in.put(qgrp.data(), nqe), a.t_off = in.put(t_off, ngroups + 1), a.t_idx = in.put(t_idx, nte);
  a.sf = in.put(scale_factors, nlevels), a.sigma2 = level_sigma2 ? in.put(level_sigma2, nlevels) : nullptr;
  for (int i = 0; i < 9; i++) a.F[i] = F12 ? F12[i] : 0.f;
  a.ep[0] = ep[0], a.ep[1] = ep[1];
  if (!in.ok() || !h->out.ok()) return -3;
  HIP_TRYM(hipMemcpyAsync(in.d, in.h, in.off, hipMemcpyHostToDevice, s));
  Arena& out = h->out;
  a.nmatches = out.put((const int*)nullptr, 4), a.m12 = out.put((const int*)nullptr, n1);
  if (!out.ok()) return -3;
  a.n1 = n1, a.nqe = nqe, a.onlyStereo = only_stereo, a.coarse = coarse, a.th = th_low, a.check_ori = check_ori, a.binOf = h->binOf.p;
  if (n1) HIP_TRYM(hipMemsetAsync(a.m12, 0xff, (size_t)n1 * 4, s));
  if (nqe > 0) hipLaunchKernelGGL(k_tri_search, dim3((nqe + 3) / 4), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_tri_finish, dim3(1), dim3(256), 0, s, a);
  HIP_TRYM(hipGetLastError());
  HIP_TRYM(hipMemcpyAsync(out.h, out.d, out.off, hipMemcpyDeviceToHost, s));
  HIP_TRYM(hipStreamSynchronize(s));
  if (n1) memcpy(matches12, out.host_of(a.m12), (size_t)n1 * 4);
  return *out.host_of(a.nmatches);
}

// Frame::AssignFeaturesToGrid for the frames of an extraction batch, device-resident (see k_assign_grid).
int orbm_assign_features_to_grid_device(orbm_t* h, const orbx_keypoint_t* d_kps_un, const int* d_counts, int nframes, int cap, float min_x,
                                        float max_x, float min_y, float max_y, int* d_g_idx, uint8_t* d_g_cx, uint8_t* d_g_cy, int* d_ng,
                                        void* stream) {
  if (!h || !d_kps_un || !d_g_idx || !d_g_cx || !d_g_cy || !d_ng || nframes < 0 || cap < 1 || !(max_x > min_x) || !(max_y > min_y)) return -2;
  if (cap > 8192) {
    set_err("orbm_assign_features_to_grid_device: more than 8192 features per frame");
    return -2;
  }
  if (nframes == 0) return 0;
  HIP_TRYM(hipSetDevice(h->device));
  int NP = 256;
  while (NP < cap) NP <<= 1;
  const float gwInv = static_cast<float>(kGridCols) / (max_x - min_x), ghInv = static_cast<float>(kGridRows) / (max_y - min_y);
  hipLaunchKernelGGL(k_assign_grid, dim3(nframes), dim3(256), (size_t)NP * 4, stream ? (hipStream_t)stream : h->stream, d_kps_un, d_counts, cap, NP,
                     min_x, min_y, gwInv, ghInv, d_g_idx, d_g_cx, d_g_cy, d_ng, h->colStartOut);
  HIP_TRYM(hipGetLastError());
  return 0;
}

// SearchByProjection(CurrentFrame, LastFrame, th, bMono) for many (last, current) frame pairs of one extraction batch; everything
// stays on the device (see include/orbx.h).  Enqueue only.
int orbm_search_by_projection_batch_device(orbm_t* h, int npairs, const int* d_pairs, const uint8_t* d_desc, const orbx_keypoint_t* d_kps_un,
                                           const int* d_counts, int nframes, int cap, const uint8_t* d_last_desc, const float* d_last_uv,
                                           const float* d_last_invz, const uint8_t* d_last_valid, const float* d_u_right,
                                           const uint8_t* d_occupied, const float* scale_factors, int nlevels, float th, int dir, float mbf,
                                           float min_x, float max_x, float min_y, float max_y, int th_dist, int check_ori, int* d_cur_of_last,
                                           int* d_last_of_cur, int* d_nmatches, void* stream) {
  if (!h || npairs < 0 || !d_pairs || !d_desc || !d_kps_un || !d_counts || nframes < 1 || cap < 1 || !d_last_uv || !scale_factors ||
      nlevels < 1 || nlevels > 64 || dir < 0 || dir > 2 || !(max_x > min_x) || !(max_y > min_y) || !d_cur_of_last || !d_last_of_cur ||
      !d_nmatches || (d_u_right && !d_last_invz)) {
    set_err("orbm_search_by_projection_batch_device: bad arguments");
    return -2;
  }
  if (cap > 8192) {
    set_err("orbm_search_by_projection_batch_device: more than 8192 features per frame");
    return -2;
  }
  if (npairs == 0) return 0;
  HIP_TRYM(hipSetDevice(h->device));
  hipStream_t  s  = stream ? (hipStream_t)stream : h->stream;
  const size_t fc = (size_t)nframes * cap, pc = (size_t)npairs * cap;
  if (h->gidx.ensure(fc) || h->gcx.ensure(fc) || h->gcy.ensure(fc) || h->qoff.ensure(nframes + 1) || h->ta.ensure(64) ||
      h->topk.ensure(pc * kTopK) || h->nvalid.ensure(pc) || h->binOf.ensure(pc) || h->colStart.ensure((size_t)nframes * (kGridCols + 1)))
    return -3;
  h->colStartOut = h->colStart.p;  // the internal grid build also writes the per-column starts
  int rc = orbm_assign_features_to_grid_device(h, d_kps_un, d_counts, nframes, cap, min_x, max_x, min_y, max_y, h->gidx.p, h->gcx.p, h->gcy.p,
                                               h->qoff.p, s);
  h->colStartOut = nullptr;
  if (rc) return rc;
  HIP_TRYM(hipMemcpyAsync(h->ta.p, scale_factors, (size_t)nlevels * 4, hipMemcpyHostToDevice, s));
  ProjArgs a;
  memset(&a, 0, sizeof(a));
  a.pairs = d_pairs, a.counts = d_counts, a.ngOf = h->qoff.p, a.cap = cap, a.hasLastDesc = d_last_desc ? 1 : 0;
  a.d1 = d_last_desc, a.d2 = d_desc, a.kp2 = d_kps_un, a.uv = d_last_uv, a.invz = d_last_invz, a.valid1 = d_last_valid;
  a.uRight = d_u_right, a.occupied = d_occupied, a.g_idx = h->gidx.p, a.g_cx = h->gcx.p, a.g_cy = h->gcy.p, a.sf = h->ta.p;
  a.colStart = h->colStart.p;
  a.th = th, a.mbf = mbf, a.minX = min_x, a.minY = min_y, a.gwInv = static_cast<float>(kGridCols) / (max_x - min_x);
  a.ghInv = static_cast<float>(kGridRows) / (max_y - min_y), a.dir = dir, a.thDist = th_dist, a.check_ori = check_ori;
  a.topk = h->topk.p, a.nvalid = h->nvalid.p, a.binOf = h->binOf.p, a.m12 = d_cur_of_last, a.m21 = d_last_of_cur, a.nmatches = d_nmatches;
  hipLaunchKernelGGL(k_proj_topk, dim3((cap + 3) / 4, npairs), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_proj_resolve, dim3(npairs), dim3(64), 0, s, a);
  HIP_TRYM(hipGetLastError());
  return 0;
}

}  // extern "C"
