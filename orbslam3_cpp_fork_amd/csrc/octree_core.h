// octree_core.h -- DistributeOctTree (reference src/ORBextractor.cc:556-753) in array form.
//
// The reference keeps a std::list of nodes, each owning a std::vector of keypoints, splits nodes
// 4-ways with push_front, and in its last phase splits nodes in std::sort order until the list holds
// N nodes.  Which keypoints survive -- and the order they come out in -- depends on the list order and
// on how libstdc++'s (unstable) introsort arranges ties.  This file reproduces that behaviour exactly
// with data structures a GPU workgroup can run:
//
//   * the list is an array stored in list order (index 0 = front); one step rebuilds it as
//        [children of the nodes processed in this step, in reverse push order] ++ [surviving nodes, in order]
//     which is what erase + push_front produce;
//   * a node is a rectangle plus a point COUNT; every candidate carries the index of the node that owns it.
//     DivideNode's 4-way split of vKeys (:513-526) becomes: count the points of each child (one atomic add per
//     point), create the children, relabel the points.  The points themselves never move: the only things the
//     reference's per-node vectors are used for are their sizes and, at the very end, "the point with the largest
//     response, the first one in the vector on ties" -- and the vector order inside a node is always the feed order,
//     so that pick is max(response), then min(feed index), an order-free reduction;
//   * std::sort (:685) is re-enacted operation for operation (libstdc++ bits/stl_algo.h: introsort with
//     median-of-3 pivot, threshold 16, final insertion sort; heap-sort fallback included).
//
// One source, two builds: the HIP kernel (one workgroup per frame x level; CTX = LDS scans + barriers)
// and a single-threaded host model (CTX = trivial) that the CPU test-suite checks against the oracle.
#ifndef ORBX_OCTREE_CORE_H_
#define ORBX_OCTREE_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OT_HD __host__ __device__
#else
#define OT_HD
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define OT_CLOCK() ((long long)__builtin_readcyclecounter())
#else
#define OT_CLOCK() 0LL
#endif

namespace orbx {

// Node-level state is kept compact (64 bytes per node over all arrays) so that a whole problem fits in the LDS share of
// one wave (see k_octree): coordinates and list indices are 16-bit (levels are < 4096 px, lists < 16384 nodes).
typedef uint16_t OtIdx;
constexpr int   kOtMaxNodes = 16383;   // node index + 2 quadrant bits in a 16-bit point label
constexpr OtIdx kOtNone     = 0xffff;  // "not in D" (dIndex) / "no such child" (childIdx)

struct OtNode {
  uint16_t x0, x1, y0, y1;  // UL.x, UR.x, UL.y, BR.y (all four corners follow: the nodes are rectangles)
  int      cnt;             // vKeys.size()
};

struct Int4 {
  int v[4];
};

struct Idx4 {
  OtIdx v[4];
};

// Candidate record: x | y << 12 | response << 24, coordinates relative to (minBorderX, minBorderY)
// exactly as the reference feeds them (src/ORBextractor.cc:842-848).
OT_HD inline int ot_cx(uint32_t c) { return (int)(c & 0xfffu); }
OT_HD inline int ot_cy(uint32_t c) { return (int)((c >> 12) & 0xfffu); }
OT_HD inline int ot_cr(uint32_t c) { return (int)(c >> 24); }

struct OtArgs {
  // inputs
  const uint32_t* cand;  // n candidates in feed order
  int             n;
  int             W, H;  // maxX - minX, maxY - minY
  int             N;     // wanted features for this level
  int             nIni;  // round(W / H), >= 1 (checked by the host)
  float           hX;    // (float)W / nIni
  int             maxNodes;
  // workspace, sizes: points -> n, nodes -> maxNodes (<= kOtMaxNodes).  Every array may live in LDS or in global memory.
  uint16_t*           nodeOf;  // per candidate: owner node (index into the current list) | child quadrant << 14
  OtNode *            nodesA, *nodesB;
  Int4*               childCnt;  // per entry of D: points per child.           `best` may alias it (used after the last step)
  Idx4*               childIdx;  // per entry of D: list index of each child.   `sortbuf` may alias it (consumed before step C)
  unsigned long long* best;      // per node: (response << 32 | ~feed index) of its best point
  OtIdx *             D, *dIndex, *pushOff, *multiOff, *keptRank, *vA, *vB;
  uint8_t*            nCh;       // per entry of D: children | multi-point children << 4
  unsigned long long* sortbuf;   // >= maxNodes entries
  // outputs
  uint32_t* outKp;  // kept candidates (same packing), list order
  int*      outCount;
  long long* dbgT;  // optional diagnostic stamps (device tuning builds): [0] sort cycles, [1] total cycles, [2] division steps, [3] sorts,
                    // [4] cycles in the passes over the points, [5] per-node scans, [6] node rewrite, [7] root set-up
  unsigned long long* dbgSort;  // optional (orbx_debug_octree_sorts): [0] = sorts on record; per sort 1 + 2 * 256 words: V, records before, records after
  int                 dbgSortCap;  // sorts the buffer holds (later ones are not recorded)
};

// ------------------------------------------------------------------------------------------------
// libstdc++ std::sort re-enactment on 64-bit records; order = (rec >> 16), low 16 bits = payload.
// ------------------------------------------------------------------------------------------------
OT_HD inline bool ot_less(unsigned long long a, unsigned long long b) { return (a >> 16) < (b >> 16); }

OT_HD inline void ot_swap(unsigned long long* b, int i, int j) {
  unsigned long long t = b[i];
  b[i]                 = b[j];
  b[j]                 = t;
}

OT_HD inline void ot_unguarded_linear_insert(unsigned long long* b, int last) {
  unsigned long long val  = b[last];
  int                next = last - 1;
  while (ot_less(val, b[next])) {
    b[last] = b[next];
    last    = next;
    --next;
  }
  b[last] = val;
}

OT_HD inline void ot_insertion_sort(unsigned long long* b, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (ot_less(b[i], b[first])) {
      unsigned long long val = b[i];
      for (int k = i; k > first; --k) b[k] = b[k - 1];  // move_backward(first, i, i + 1)
      b[first] = val;
    } else {
      ot_unguarded_linear_insert(b, i);
    }
  }
}

OT_HD inline void ot_push_heap(unsigned long long* f, int holeIndex, int topIndex, unsigned long long value) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && ot_less(f[parent], value)) {
    f[holeIndex] = f[parent];
    holeIndex    = parent;
    parent       = (holeIndex - 1) / 2;
  }
  f[holeIndex] = value;
}

OT_HD inline void ot_adjust_heap(unsigned long long* f, int holeIndex, int len, unsigned long long value) {
  const int topIndex    = holeIndex;
  int       secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (ot_less(f[secondChild], f[secondChild - 1])) secondChild--;
    f[holeIndex] = f[secondChild];
    holeIndex    = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild  = 2 * (secondChild + 1);
    f[holeIndex] = f[secondChild - 1];
    holeIndex    = secondChild - 1;
  }
  ot_push_heap(f, holeIndex, topIndex, value);
}

OT_HD inline void ot_heap_sort(unsigned long long* b, int first, int last) {  // __partial_sort(first, last, last)
  unsigned long long* f   = b + first;
  int                 len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      unsigned long long value = f[parent];
      ot_adjust_heap(f, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  while (len > 1) {
    --len;
    unsigned long long value = f[len];
    f[len]                   = f[0];
    ot_adjust_heap(f, 0, len, value);
  }
}

OT_HD inline int ot_partition_pivot(unsigned long long* b, int first, int last) {
  int mid = first + (last - first) / 2;
  int a = first + 1, bb = mid, c = last - 1;
  // __move_median_to_first(first, a, b, c)
  if (ot_less(b[a], b[bb])) {
    if (ot_less(b[bb], b[c]))
      ot_swap(b, first, bb);
    else if (ot_less(b[a], b[c]))
      ot_swap(b, first, c);
    else
      ot_swap(b, first, a);
  } else if (ot_less(b[a], b[c]))
    ot_swap(b, first, a);
  else if (ot_less(b[bb], b[c]))
    ot_swap(b, first, c);
  else
    ot_swap(b, first, bb);
  // __unguarded_partition(first + 1, last, first)
  int lo = first + 1, hi = last;
  while (true) {
    while (ot_less(b[lo], b[first])) ++lo;
    --hi;
    while (ot_less(b[first], b[hi])) --hi;
    if (!(lo < hi)) return lo;
    ot_swap(b, lo, hi);
    ++lo;
  }
}

// depth_limit < 0 -> the library's own 2*floor(log2(n)); tests pass small values to reach the heap path.
// `stk` = 3 x 64 ints of work-list storage (the device passes LDS: private arrays would live in scratch memory).
OT_HD inline void ot_std_sort(unsigned long long* b, int n, int depth_limit, int* stk) {
  if (n <= 0) return;
  int lg = 0;
  while ((1 << (lg + 1)) <= n) lg++;
  int depth0 = depth_limit < 0 ? 2 * lg : depth_limit;
  // __introsort_loop with an explicit stack (the two sub-ranges are disjoint, order of visiting is free)
  int *sf = stk, *sl = stk + 64, *sd = stk + 128, sp = 0;
  sf[0] = 0, sl[0] = n, sd[0] = depth0, sp = 1;
  while (sp > 0) {
    --sp;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) {
        ot_heap_sort(b, first, last);
        break;
      }
      --depth;
      int cut = ot_partition_pivot(b, first, last);
      if (sp < 64) {
        sf[sp] = cut, sl[sp] = last, sd[sp] = depth;
        sp++;
      }
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    ot_insertion_sort(b, 0, 16);
    for (int i = 16; i < n; ++i) ot_unguarded_linear_insert(b, i);
  } else {
    ot_insertion_sort(b, 0, n);
  }
}

OT_HD inline void ot_std_sort(unsigned long long* b, int n, int depth_limit = -1) {
  int stk[192];
  ot_std_sort(b, n, depth_limit, stk);
}

// ------------------------------------------------------------------------------------------------
// Execution contexts.  A context offers: tid, nt, sync(), scan4() (exclusive scan of one Int4 per
// thread across the workgroup, returning the grand total too), bcast() and min_reduce().
// ------------------------------------------------------------------------------------------------
struct HostCtx {
  int  tid = 0, nt = 1;
  void sync() {}
  Int4 scan4(const Int4& v, Int4* total) {
    *total = v;
    Int4 z = {{0, 0, 0, 0}};
    return z;
  }
  int  min_reduce(int v) { return v; }
  void sort(unsigned long long* buf, int n) { ot_std_sort(buf, n); }
  void atomic_add(int* p, int v) { *p += v; }
  void atomic_max(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
  int  atomic_load(const int* p) const { return *p; }
  unsigned long long atomic_load(const unsigned long long* p) const { return *p; }
};

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------
// Wave-parallel re-enactment of libstdc++ std::sort for n <= 256 records (one wave, record j lives in lane j % 64,
// register slot j / 64).  The array after every partition, and after the final insertion phase, is the one ot_std_sort
// leaves -- equal keys included, which is what makes the quadtree's node order the reference's -- but neither phase is
// walked serially: a partition's swaps are all found from two ballots per register slot (WaveSort::partition) and the
// final insertion sort is a windowed stable rank (ot_wave_sort_r).  A single lane walking the library algorithm through
// LDS needs ~2000 dependent LDS round trips (~100 us) for the ~100 nodes of a level; this needs ~60 wave instructions per
// partition.  (The heap-sort path of the library -- its depth limit -- is walked by one lane on the range it applies to.)
// ------------------------------------------------------------------------------------------------
template <int R>
struct WaveSort {
  unsigned long long   v[R];  // this lane's records
  int                  lane;

  __device__ __forceinline__ unsigned long long get(int j) const {  // j wave-uniform
    const int          s = j >> 6;
    unsigned long long x = v[0];
#pragma unroll
    for (int t = 1; t < R; t++) x = s == t ? v[t] : x;
    const int l = j & 63;
    const unsigned     lo = __builtin_amdgcn_readlane((int)(unsigned)x, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
  }
  __device__ __forceinline__ void set(int j, unsigned long long x) {
    const int s = j >> 6;
    if (lane == (j & 63)) {
#pragma unroll
      for (int t = 0; t < R; t++)
        if (s == t) v[t] = x;
    }
  }
  __device__ __forceinline__ void swap(int i, int j) {
    const unsigned long long a = get(i), b = get(j);
    set(i, b);
    set(j, a);
  }
  // __unguarded_partition(first + 1, last, first) with all of its swaps done at once.  On the array as it stands before the loop let
  // G = the positions of (first, last) whose record is not below the pivot, in ascending order (L_0 < L_1 < ...), and S = those whose
  // record is not above it, in descending order (H_0 > H_1 > ...).  The library loop's k-th iteration stops its two scans at L_k and
  // H_k -- neither scan can reach a position an earlier swap wrote before the loop ends -- and swaps them while L_k < H_k; with K such
  // iterations it returns min(L_K, H_(K-1)).  For a position p:  a(p) = |G below p|,  c(p) = |S above p|;  p = L_k is swapped iff
  // c(p) > a(p) (then k = a(p)), p = H_k iff a(p) > c(p) (k = c(p)) -- all from two ballots per register slot.  The exchange goes
  // through `scratch` (>= last - first records; the sort buffer itself, unused while the records live in registers).
  __device__ __forceinline__ int partition(int first, int last, unsigned long long pivot, unsigned long long* scratch) {
    const unsigned long long pk = pivot >> 16;
    const int                s0 = (first + 1) >> 6, s1 = (last - 1) >> 6;
    unsigned long long       mg[R], mh[R];
#pragma unroll
    for (int s = 0; s < R; s++) {
      mg[s] = 0, mh[s] = 0;
      if (s >= s0 && s <= s1) {
        const int                idx = s * 64 + lane;
        const bool               in  = idx > first && idx < last;
        const unsigned long long k   = v[s] >> 16;
        mg[s] = __ballot(in && !(k < pk));
        mh[s] = __ballot(in && !(pk < k));
      }
    }
    int a[R], c[R], acc = 0;
#pragma unroll
    for (int s = 0; s < R; s++) {
      a[s] = acc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mg[s] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg[s], 0u));
      acc += __popcll(mg[s]);
    }
    acc = 0;
#pragma unroll
    for (int s = R - 1; s >= 0; s--) {
      const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mh[s] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mh[s], 0u));
      const int tot   = __popcll(mh[s]);
      c[s] = acc + tot - below - (int)((mh[s] >> lane) & 1ull);
      acc += tot;
    }
    bool left[R], right[R];
    int  K = 0;
#pragma unroll
    for (int s = 0; s < R; s++) {
      left[s]  = ((mg[s] >> lane) & 1ull) && c[s] > a[s];
      right[s] = ((mh[s] >> lane) & 1ull) && a[s] > c[s];
      K += __popcll(__ballot(left[s]));
    }
    if (K > 0) {
#pragma unroll
      for (int s = 0; s < R; s++) {
        if (left[s]) scratch[2 * a[s]] = v[s];
        if (right[s]) scratch[2 * c[s] + 1] = v[s];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < R; s++) {
        if (left[s]) v[s] = scratch[2 * a[s] + 1];
        if (right[s]) v[s] = scratch[2 * c[s]];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    int cut = last;
#pragma unroll
    for (int s = 0; s < R; s++) {
      const unsigned long long m = __ballot((((mg[s] >> lane) & 1ull) && a[s] == K) || (((mh[s] >> lane) & 1ull) && c[s] == K - 1));
      if (m && cut == last) cut = s * 64 + (__ffsll((long long)m) - 1);
    }
    return cut;
  }

  // stk: 3 x 64 ints in LDS (wave-uniform work list); depth_limit < 0 -> the library's own 2 * floor(log2(n)) (tests pass small values)
  __device__ __forceinline__ void run(int n, int* stk, unsigned long long* scratch, int depth_limit) {
    int lg = 0;
    while ((1 << (lg + 1)) <= n) lg++;
    int *sf = stk, *sl = stk + 64, *sd = stk + 128, sp = 0;
    sf[0] = 0, sl[0] = n, sd[0] = depth_limit < 0 ? 2 * lg : depth_limit, sp = 1;
    while (sp > 0) {
      --sp;
      int first = sf[sp], last = sl[sp], depth = sd[sp];
      while (last - first > 16) {
        if (depth == 0) {
          // the library's __partial_sort(first, last, last) = heap sort of this range, walked by one lane on the array as it stands
          // (rare: 2 * log2(n) partitions without getting a range down to 16 records -- long runs of descending keys do it).  The
          // records go through `scratch`, which holds nothing else between two partitions; a heap-sorted range is longer than 16
          // records but already in its final order, which the windowed rank of ot_wave_sort_r leaves as it is.
#pragma unroll
          for (int s = 0; s < R; s++)
            if (s * 64 + lane < n) scratch[s * 64 + lane] = v[s];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) ot_heap_sort(scratch, first, last);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int s = 0; s < R; s++)
            if (s * 64 + lane < n) v[s] = scratch[s * 64 + lane];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          break;
        }
        --depth;
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const int                mid = first + (last - first) / 2;
        const int                ia = first + 1, ib = mid, ic = last - 1;
        const unsigned long long A = get(ia) >> 16, B = get(ib) >> 16, C = get(ic) >> 16;
        if (A < B) {
          if (B < C)
            swap(first, ib);
          else if (A < C)
            swap(first, ic);
          else
            swap(first, ia);
        } else if (A < C)
          swap(first, ia);
        else if (B < C)
          swap(first, ic);
        else
          swap(first, ib);
        // __unguarded_partition(first + 1, last, first)
        const int cut = partition(first, last, get(first), scratch);
        sf[sp] = cut, sl[sp] = last, sd[sp] = depth;
        sp++;
        last = cut;
      }
    }
    // __final_insertion_sort is done by the caller (ot_wave_sort_r): see there
  }
};

template <int R>
__device__ __forceinline__ void ot_wave_sort_r(unsigned long long* buf, int n, int* stk, int depth_limit) {
  const int   lane = threadIdx.x & 63;
  WaveSort<R> w;
  w.lane = lane;
#pragma unroll
  for (int s = 0; s < R; s++) w.v[s] = (s * 64 + lane) < n ? buf[s * 64 + lane] : ~0ull;
  w.run(n, stk, buf, depth_limit);
  // __final_insertion_sort (guarded insertion on the first 16 records, unguarded linear inserts after) without replaying its n
  // serial insertions.  An insertion sort never moves a record past an equal one, so its result is THE stable sort of the array the
  // introsort loop left behind; and that array is block-sorted -- every range the loop stopped on holds at most 16 records, all
  // records of an earlier range are <= all records of a later one -- so a record ends inside its own range, at most 15 places from
  // where it stands.  Hence: final position of record i = max(0, i - 15) + the number of records j of [i - 15, i + 15] with
  // (key_j, j) < (key_i, i); every lane counts for its own records in parallel, then the records are scattered.  (A range the loop
  // heap-sorted instead may be longer, but it is in order already: the count then returns every record of it to where it stands.)
#pragma unroll
  for (int s = 0; s < R; s++)
    if (s * 64 + lane < n) buf[s * 64 + lane] = w.v[s];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int pos[R];
#pragma unroll
  for (int s = 0; s < R; s++) {
    const int                i  = s * 64 + lane;
    const unsigned long long ki = w.v[s] >> 16;
    const int                lo = i - 15 > 0 ? i - 15 : 0, hi = i + 15 < n - 1 ? i + 15 : n - 1;
    int                      c  = 0;
    if (i < n) {
      for (int j = lo; j <= hi; j++) {
        const unsigned long long kj = buf[j] >> 16;
        c += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
      }
    }
    pos[s] = lo + c;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int s = 0; s < R; s++)
    if (s * 64 + lane < n) buf[pos[s]] = w.v[s];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ inline void ot_wave_sort(unsigned long long* buf, int n, int* stk, int depth_limit = -1) {  // called by ONE full wave; buf in LDS or global
  if (n <= 64)
    ot_wave_sort_r<1>(buf, n, stk, depth_limit);
  else if (n <= 128)
    ot_wave_sort_r<2>(buf, n, stk, depth_limit);
  else if (n <= 256)
    ot_wave_sort_r<4>(buf, n, stk, depth_limit);
  else if ((threadIdx.x & 63) == 0)
    ot_std_sort(buf, n, depth_limit, stk);  // large inputs: the serial re-enactment
}

struct DevCtx {
  int   tid, nt;
  int*  lds;  // >= 80 ints of LDS scratch (per-wave totals) followed by 192 ints of sort work list
  __device__ void sync() { __syncthreads(); }
  __device__ Int4 scan4(const Int4& v, Int4* total) {
    const int lane = tid & 63, wave = tid >> 6, nw = (nt + 63) >> 6;
    Int4      inc  = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int t = __shfl_up(inc.v[k], off, 64);
        if (lane >= off) inc.v[k] += t;
      }
    }
    __syncthreads();  // protect lds from a previous use
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 4; k++) lds[wave * 4 + k] = inc.v[k];
    }
    __syncthreads();
    Int4 base = {{0, 0, 0, 0}}, tot = {{0, 0, 0, 0}};
    for (int w = 0; w < nw; w++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int t = lds[w * 4 + k];
        if (w < wave) base.v[k] += t;
        tot.v[k] += t;
      }
    }
    *total = tot;
    Int4 ex;
#pragma unroll
    for (int k = 0; k < 4; k++) ex.v[k] = base.v[k] + inc.v[k] - v.v[k];
    return ex;
  }
  __device__ void sort(unsigned long long* buf, int n) {
    if (tid < 64) ot_wave_sort(buf, n, lds + 80);
  }
  __device__ int min_reduce(int v) {
    const int lane = tid & 63, wave = tid >> 6, nw = (nt + 63) >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    int r = lds[0];
    for (int w = 1; w < nw; w++) r = min(r, lds[w]);
    return r;
  }
  __device__ void atomic_add(int* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ void atomic_max(unsigned long long* p, unsigned long long v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ int  atomic_load(const int* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ unsigned long long atomic_load(const unsigned long long* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
#endif

#if defined(__HIPCC__)
// One wave = one problem: the scan and the reductions are pure cross-lane shuffles (no LDS, no barrier), and sync() only
// has to make this wave's own global-memory writes visible to its other lanes before the next pass reads them.
struct WaveCtx {
  int  tid;  // lane
  int  nt;   // 64
  int* lds;  // 80 unused ints (the multi-wave context keeps its per-wave totals there) followed by 192 ints of sort work list
  __device__ void sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
  __device__ Int4 scan4(const Int4& v, Int4* total) {
    Int4 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int t = __shfl_up(inc.v[k], off, 64);
        if (tid >= off) inc.v[k] += t;
      }
    }
    Int4 ex;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      total->v[k] = __builtin_amdgcn_readlane(inc.v[k], 63);
      ex.v[k]     = inc.v[k] - v.v[k];
    }
    return ex;
  }
  __device__ int min_reduce(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
  }
  __device__ void sort(unsigned long long* buf, int n) { ot_wave_sort(buf, n, lds + 80); }
  // counters live in LDS or in global memory (generic pointers); reads go through the same agent-scope path as the
  // atomics so that a line cached by an earlier plain load is never returned
  __device__ void atomic_add(int* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ void atomic_max(unsigned long long* p, unsigned long long v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ int  atomic_load(const int* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ unsigned long long atomic_load(const unsigned long long* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
#endif

// chunk of [0, n) owned by this thread (contiguous, so that per-thread running counts are prefix counts)
template <class CTX>
OT_HD inline void ot_chunk(const CTX& c, int n, int* lo, int* hi) {
  int per = (n + c.nt - 1) / c.nt;
  int l   = c.tid * per;
  int h   = l + per;
  if (l > n) l = n;
  if (h > n) h = n;
  *lo = l;
  *hi = h;
}

// Exclusive scan of an int array of length n in place is not needed; what the algorithm needs is
// "rank of flagged elements": out[i] = number of flagged j < i, plus the total.
template <class CTX, class F, class OUT>
OT_HD inline int ot_rank_flags(CTX& c, int n, F flag, OUT* out) {
  int lo, hi;
  ot_chunk(c, n, &lo, &hi);
  Int4 loc = {{0, 0, 0, 0}};
  for (int i = lo; i < hi; i++) loc.v[0] += flag(i) ? 1 : 0;
  Int4 tot;
  Int4 ex  = c.scan4(loc, &tot);
  int  run = ex.v[0];
  for (int i = lo; i < hi; i++) {
    out[i] = (OUT)run;
    run += flag(i) ? 1 : 0;
  }
  c.sync();
  return tot.v[0];
}

// ------------------------------------------------------------------------------------------------
// One "division step": the nodes D[0..nD) of the current list (processing order) are candidates for
// DivideNode; mode_ordered applies the `size >= N -> break` rule of :722-724.  Returns the new list size.
// ------------------------------------------------------------------------------------------------
struct OtState {
  OtNode* nodes;
  OtNode* nodesNew;
  OtIdx * v, *vNew;
  int     nNodes;
  int     nV;  // multi-point children created by the last step (vSizeAndPointerToNode)
};

// child quadrant of DivideNode (:480-526): n1,n2,n3,n4 = 0,1,2,3; the split lines are UL + ceil(half extent)
OT_HD inline int ot_mid(int lo, int hi) { return lo + ((hi - lo + 1) >> 1); }

template <class CTX>
OT_HD inline void ot_divide_step(CTX& c, const OtArgs& a, OtState& s, int nD, bool mode_ordered) {
  if (a.dbgT && c.tid == 0) a.dbgT[2] += 1;
  const int n = a.n;
  // -- A: count the points of every child of every flagged node (counters indexed by position in D)
  long long tp = a.dbgT ? OT_CLOCK() : 0;
  for (int j = c.tid; j < nD; j += c.nt) a.childCnt[j] = Int4{{0, 0, 0, 0}};
  c.sync();
  for (int i = c.tid; i < n; i += c.nt) {
    const int nd = a.nodeOf[i] & 0x3fff;
    const int di = a.dIndex[nd];
    if (di != kOtNone) {
      const OtNode   nn = s.nodes[nd];
      const uint32_t cd = a.cand[i];
      const int      q  = (ot_cx(cd) < ot_mid(nn.x0, nn.x1) ? 0 : 1) + (ot_cy(cd) < ot_mid(nn.y0, nn.y1) ? 0 : 2);
      a.nodeOf[i]       = (uint16_t)(nd | (q << 14));
      c.atomic_add(&a.childCnt[di].v[q], 1);
    }
  }
  c.sync();
  if (a.dbgT && c.tid == 0) a.dbgT[4] += OT_CLOCK() - tp, tp = OT_CLOCK();
  // -- B: per processed node: children / multi-point children; exclusive offsets in processing order
  int dlo, dhi;
  ot_chunk(c, nD, &dlo, &dhi);
  Int4 l2 = {{0, 0, 0, 0}};
  for (int j = dlo; j < dhi; j++) {
    int ch = 0, mu = 0;
    for (int q = 0; q < 4; q++) {
      const int cq = c.atomic_load(&a.childCnt[j].v[q]);
      ch += cq > 0;
      mu += cq > 1;
    }
    a.nCh[j] = (uint8_t)(ch | (mu << 4));
    l2.v[0] += ch;
    l2.v[1] += mu;
  }
  Int4 t2;
  Int4 r2 = c.scan4(l2, &t2);
  for (int j = dlo; j < dhi; j++) {
    a.pushOff[j]  = (OtIdx)r2.v[0];
    a.multiOff[j] = (OtIdx)r2.v[1];
    r2.v[0] += a.nCh[j] & 0xf;
    r2.v[1] += a.nCh[j] >> 4;
  }
  c.sync();
  // -- break rule of the ordered phase: stop after the first processed node that brings the list to >= N
  int K = nD;
  if (mode_ordered) {
    int best = nD;
    for (int j = dlo; j < dhi; j++) {
      // list size after processing entries 0..j: + children - 1 each
      int sizeAfter = s.nNodes + (a.pushOff[j] + (a.nCh[j] & 0xf)) - (j + 1);
      if (sizeAfter >= a.N) {
        best = j + 1;
        break;
      }
    }
    K = c.min_reduce(best);
  }
  const int Ptot = (K < nD) ? a.pushOff[K] : t2.v[0];
  const int Mtot = (K < nD) ? a.multiOff[K] : t2.v[1];
  // -- rank of the surviving nodes
  const OtIdx* dIndex = a.dIndex;  // kOtNone compares above every K
  ot_rank_flags(c, s.nNodes, [dIndex, K](int i) { return !(dIndex[i] < K); }, a.keptRank);
  if (a.dbgT && c.tid == 0) a.dbgT[5] += OT_CLOCK() - tp, tp = OT_CLOCK();
  // -- C: write the new node array (children in reverse push order, then the survivors)
  for (int j = c.tid; j < K; j += c.nt) {
    const int    d  = a.D[j];
    const OtNode nn = s.nodes[d];
    const int    mx = ot_mid(nn.x0, nn.x1), my = ot_mid(nn.y0, nn.y1);
    int          k = 0, m = 0;
    Idx4         childIdx;
    for (int q = 0; q < 4; q++) {
      const int cq  = c.atomic_load(&a.childCnt[j].v[q]);
      childIdx.v[q] = kOtNone;
      if (cq > 0) {
        const int idx = Ptot - 1 - (a.pushOff[j] + k);
        OtNode    ch;
        ch.x0 = (uint16_t)((q & 1) ? mx : nn.x0);
        ch.x1 = (uint16_t)((q & 1) ? nn.x1 : mx);
        ch.y0 = (uint16_t)((q & 2) ? my : nn.y0);
        ch.y1 = (uint16_t)((q & 2) ? nn.y1 : my);
        ch.cnt          = cq;
        s.nodesNew[idx] = ch;
        childIdx.v[q]   = (OtIdx)idx;
        if (cq > 1) {
          s.vNew[a.multiOff[j] + m] = (OtIdx)idx;
          m++;
        }
        k++;
      }
    }
    a.childIdx[j] = childIdx;
  }
  for (int i = c.tid; i < s.nNodes; i += c.nt) {
    const int di = a.dIndex[i];
    if (!(di < K)) s.nodesNew[Ptot + a.keptRank[i]] = s.nodes[i];
  }
  c.sync();
  if (a.dbgT && c.tid == 0) a.dbgT[6] += OT_CLOCK() - tp, tp = OT_CLOCK();
  // -- C': relabel the points
  for (int i = c.tid; i < n; i += c.nt) {
    const int lb = a.nodeOf[i], nd = lb & 0x3fff;
    const int di = a.dIndex[nd];
    a.nodeOf[i]  = (uint16_t)((di < K) ? a.childIdx[di].v[lb >> 14] : Ptot + a.keptRank[nd]);
  }
  c.sync();
  if (a.dbgT && c.tid == 0) a.dbgT[4] += OT_CLOCK() - tp;
  const int newSize = Ptot + (s.nNodes - K);
  // swap ping-pong buffers; clear the flags of the new list
  {
    OtNode* tn = s.nodes;
    s.nodes    = s.nodesNew;
    s.nodesNew = tn;
    OtIdx* t   = s.v;
    s.v        = s.vNew;
    s.vNew     = t;
  }
  s.nNodes = newSize;
  s.nV     = Mtot;
  for (int i = c.tid; i < newSize; i += c.nt) a.dIndex[i] = kOtNone;
  c.sync();
}

// ------------------------------------------------------------------------------------------------
// DistributeOctTree
// ------------------------------------------------------------------------------------------------
template <class CTX>
OT_HD inline void ot_distribute(CTX& c, const OtArgs& a) {
  const long long tStart = OT_CLOCK();
  const int n = a.n;
  if (n <= 0) {
    if (c.tid == 0) *a.outCount = 0;
    c.sync();
    return;
  }
  OtState s;
  s.nodes = a.nodesA, s.nodesNew = a.nodesB;
  s.v = a.vA, s.vNew = a.vB;
  // ---- roots (:566-600): vpIniNodes[kp.pt.x / hX]; empty roots erased.  nIni <= maxNodes / 4 (host-checked).
  {
    Int4* rootCnt = a.childCnt;  // one counter per root in .v[0]
    for (int r = c.tid; r < a.nIni; r += c.nt) rootCnt[r] = Int4{{0, 0, 0, 0}};
    c.sync();
    for (int i = c.tid; i < n; i += c.nt) {
      const int r = (int)((float)ot_cx(a.cand[i]) / a.hX);
      a.nodeOf[i] = (uint16_t)r;
      c.atomic_add(&rootCnt[r].v[0], 1);
    }
    c.sync();
    const int nn = ot_rank_flags(c, a.nIni, [&c, rootCnt](int r) { return c.atomic_load(&rootCnt[r].v[0]) > 0; }, a.keptRank);
    for (int r = c.tid; r < a.nIni; r += c.nt) {
      const int cnt = c.atomic_load(&rootCnt[r].v[0]);
      if (cnt > 0) {
        OtNode nd;
        nd.x0 = (uint16_t)(int)(a.hX * (float)r);
        nd.x1 = (uint16_t)(int)(a.hX * (float)(r + 1));
        nd.y0 = 0;
        nd.y1 = (uint16_t)a.H;
        nd.cnt = cnt;
        s.nodes[a.keptRank[r]] = nd;
      }
    }
    for (int i = c.tid; i < n; i += c.nt) a.nodeOf[i] = a.keptRank[a.nodeOf[i]];
    s.nNodes = nn;
    for (int i = c.tid; i < nn; i += c.nt) a.dIndex[i] = kOtNone;
    c.sync();
  }
  if (a.dbgT && c.tid == 0) a.dbgT[7] += OT_CLOCK() - tStart;
  // ---- main loop (:608-733)
  bool bFinish = false;
  while (!bFinish) {
    const int prevSize = s.nNodes;
    // sweep: every node that holds more than one point, in list order
    const OtNode* nodes = s.nodes;
    int nD = ot_rank_flags(c, s.nNodes, [nodes](int i) { return nodes[i].cnt > 1; }, a.keptRank);
    for (int i = c.tid; i < s.nNodes; i += c.nt) {
      if (s.nodes[i].cnt > 1) {
        a.D[a.keptRank[i]] = (OtIdx)i;
        a.dIndex[i]        = a.keptRank[i];
      }
    }
    c.sync();
    ot_divide_step(c, a, s, nD, false);
    const int nToExpand = s.nV;
    if (s.nNodes >= a.N || s.nNodes == prevSize) {
      bFinish = true;
    } else if (s.nNodes + nToExpand * 3 > a.N) {
      while (!bFinish) {
        const int prev2 = s.nNodes;
        const int V     = s.nV;
        // std::sort(vPrevSizeAndPointerToNode, compareNodes): key (count, UL.x), payload = position in vPrev
        for (int j = c.tid; j < V; j += c.nt) {
          const OtNode nd = s.nodes[s.v[j]];
          a.sortbuf[j]    = ((unsigned long long)(unsigned)nd.cnt << 32) | ((unsigned long long)(unsigned)(nd.x0 & 0xffff) << 16) |
                         (unsigned long long)(unsigned)j;
        }
        c.sync();
        const bool recordSort = a.dbgSort && V <= 256 && (int)a.dbgSort[0] < a.dbgSortCap;
        if (recordSort) {
          unsigned long long* d = a.dbgSort + 1 + a.dbgSort[0] * 513;
          if (c.tid == 0) d[0] = (unsigned long long)V;
          for (int j = c.tid; j < V; j += c.nt) d[1 + j] = a.sortbuf[j];
          c.sync();
        }
        const long long t0 = OT_CLOCK();
        c.sort(a.sortbuf, V);
        c.sync();
        if (recordSort) {
          unsigned long long* d = a.dbgSort + 1 + a.dbgSort[0] * 513;
          for (int j = c.tid; j < V; j += c.nt) d[257 + j] = a.sortbuf[j];
          c.sync();
          if (c.tid == 0) a.dbgSort[0] += 1;
          c.sync();
        }
        if (a.dbgT && c.tid == 0) a.dbgT[0] += OT_CLOCK() - t0, a.dbgT[3] += 1;
        // processed from the back of the sorted vector (:686)
        for (int k = c.tid; k < V; k += c.nt) {
          const int node = s.v[(int)(a.sortbuf[V - 1 - k] & 0xffffu)];
          a.D[k]         = (OtIdx)node;
          a.dIndex[node] = (OtIdx)k;
        }
        c.sync();
        ot_divide_step(c, a, s, V, true);
        if (s.nNodes >= a.N || s.nNodes == prev2) bFinish = true;
      }
    }
  }
  // ---- best point per node (:735-750): largest response, the earliest in feed order on ties
  for (int i = c.tid; i < s.nNodes; i += c.nt) a.best[i] = 0ull;
  c.sync();
  for (int i = c.tid; i < n; i += c.nt)
    c.atomic_max(&a.best[a.nodeOf[i] & 0x3fff], ((unsigned long long)(unsigned)ot_cr(a.cand[i]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i));
  c.sync();
  for (int i = c.tid; i < s.nNodes; i += c.nt) {
    const unsigned idx = 0xffffffffu - (unsigned)(c.atomic_load(&a.best[i]) & 0xffffffffull);
    a.outKp[i]         = a.cand[idx < (unsigned)n ? idx : 0u];  // every node holds a point; the bound only fences off memory
  }
  if (c.tid == 0) *a.outCount = s.nNodes;
  if (a.dbgT && c.tid == 0) a.dbgT[1] += OT_CLOCK() - tStart;
  c.sync();
}

}  // namespace orbx
#endif
