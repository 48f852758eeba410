// orbx_vocab.hip -- ORBVocabulary::transform on gfx950.
//
// Reference: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> as ORB-SLAM3 uses it (Frame::ComputeBoW, src/Frame.cc:879-884:
// transform(features, BowVector&, FeatureVector&, levelsup = 4)), in the configuration of ORBvoc.txt (k = 10, L = 6, scoring
// L1_NORM, weighting TF_IDF):
//   * per feature, a greedy descent of the k-ary tree by Hamming distance, first child wins ties
//     (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) -> word id, word weight (idf), node id at level L - levelsup;
//   * BowVector: word id -> sum of the weights of the features that fell into it, added in feature order
//     (BowVector::addWeight, BowVector.cpp:34-46), then divided by the L1 norm summed in ascending word order (:62-84);
//   * FeatureVector: node id -> feature indices in increasing order (FeatureVector::addFeature, FeatureVector.cpp:34-49);
//   * stopped words (weight <= 0) contribute to neither (:1161-1165).
// Everything here is integer work except the double-precision weight sums, which repeat the reference's additions in the
// reference's order (all features of a word add the SAME weight, so the sum only depends on the count).
//
// Kernels: k_bow_descend (32 lanes per feature: lane c scores child c, xor-shuffle argmin) and k_bow_build (one workgroup
// per frame: ranks by counting, no sort, no atomics on doubles).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/orbx.h"

namespace {

thread_local std::string g_verr;
extern "C" int           orbx_internal_runtime_down(void);
extern "C" void          orbx_internal_register(void* obj, void (*release_device)(void*));
extern "C" void          orbx_internal_unregister(void* obj);
extern "C" void          orbx_internal_set_error(const char* s);  // orbx_extractor.hip: the slot orbx_last_error() reads
void                     set_verr(const std::string& s) {
  g_verr = s;
  orbx_internal_set_error(s.c_str());
}

#define HIP_TRYV(expr)                                                       \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      set_verr(std::string(#expr) + ": " + hipGetErrorString(e_));           \
      return -3;                                                             \
    }                                                                        \
  } while (0)

template <class T>
struct VBuf {
  T*     p = nullptr;
  size_t n = 0;
  int    ensure(size_t count) {
    if (count <= n && p) return 0;
    if (p) hipFree(p);
    p = nullptr, n = 0;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
      set_verr(std::string("hipMalloc: ") + hipGetErrorString(e));
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

struct VocabDev {
  const uint4*    desc;      // [nodes][2]
  const int*      childOff;  // [nodes]
  const uint8_t*  childCnt;  // [nodes]   0 = leaf (Node::isLeaf(), TemplatedVocabulary.h:328)
  const int*      childIds;  // flat, children of a node contiguous in insertion order
  const uint32_t* wordId;    // [nodes]
  const double*   weight;    // [nodes]
  int             L;
};

constexpr uint32_t kStopped = 0xffffffffu;  // featWord value of a feature whose word is stopped (weight <= 0)

// One feature per 32-lane half wave.  Lane c of the half scores child c of the current node.
__global__ __launch_bounds__(256) void k_bow_descend(VocabDev v, const uint8_t* __restrict__ desc, const int* __restrict__ counts, int cap,
                                                     int levelsup, uint32_t* __restrict__ featWord, uint32_t* __restrict__ featNode,
                                                     double* __restrict__ featW) {
  const int f = blockIdx.y, c = threadIdx.x & 31;
  const int e = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int n = counts ? counts[2 * f] : cap;
  const bool live = e < n;
  uint4      a0 = make_uint4(0, 0, 0, 0), a1 = a0;
  if (live) {
    const uint4* q = reinterpret_cast<const uint4*>(desc + ((size_t)f * cap + e) * 32);
    a0 = q[0], a1 = q[1];
  }
  const int nidLevel = v.L - levelsup;
  int       node = 0, level = 0;
  uint32_t  nid = 0;  // "if (nid_level <= 0) *nid = 0" (:1229)
  bool      walking = live && v.childCnt[0] > 0;
  while (__ballot(walking)) {  // the two halves of a wave may stop at different depths
    if (walking) {
      ++level;
      const int off = v.childOff[node], cnt = v.childCnt[node];
      uint32_t  key = 0xffffffffu;
      int       id  = 0;
      if (c < cnt) {
        id = v.childIds[off + c];
        const uint4 b0 = v.desc[2 * (size_t)id], b1 = v.desc[2 * (size_t)id + 1];
        const int   d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                      __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        key = ((uint32_t)d << 8) | (uint32_t)c;  // strict "d < best_d" (:1244): the first child wins ties
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, 64));
      node = __shfl(id, (threadIdx.x & 32) | (int)(key & 0xffu), 64);
      if (level == nidLevel) nid = (uint32_t)node;
      walking = v.childCnt[node] > 0;
    }
  }
  if (live && c == 0) {
    const double w = v.weight[node];
    const size_t o = (size_t)f * cap + e;
    featWord[o] = w > 0 ? v.wordId[node] : kStopped;  // "if (w > 0) // not stopped" (:1161)
    featNode[o] = nid;
    featW[o]    = w;
  }
}

// Fallback for frames with more than 4096 features (the sorted form below needs them in LDS): ranks by counting, O(n^2).
// One workgroup per frame:
//   word heads  = first feature of every distinct word  -> BowVector entries, position = number of smaller distinct words
//   valid feats = features of non-stopped words         -> FeatureVector payload, position = rank of (node, feature index)
//   node heads  = first feature of every distinct node  -> FeatureVector keys, offset = payload position of that feature
__global__ __launch_bounds__(256) void k_bow_build_counting(const int* __restrict__ counts, int cap, const uint32_t* __restrict__ featWord,
                                                   const uint32_t* __restrict__ featNode, const double* __restrict__ featW,
                                                   uint8_t* __restrict__ headFlags, double* __restrict__ parkVal, int* __restrict__ parkOff,
                                                   uint32_t* __restrict__ bowWords,
                                                   double* __restrict__ bowVals, int* __restrict__ nWords, uint32_t* __restrict__ fvNodes,
                                                   int* __restrict__ fvOff, uint32_t* __restrict__ fvFeat, int* __restrict__ nNodes) {
  __shared__ int    sWords, sNodes, sValid;
  __shared__ double sNorm;
  const int         f = blockIdx.x, tid = threadIdx.x;
  const int         n = counts ? counts[2 * f] : cap;
  const uint32_t*   W = featWord + (size_t)f * cap;
  const uint32_t*   N = featNode + (size_t)f * cap;
  uint8_t*          H = headFlags + (size_t)f * cap;
  if (tid == 0) sWords = 0, sNodes = 0, sValid = 0;
  __syncthreads();
  // pass 1: per feature -- multiplicity and head flag of its word, payload rank and head flag of its node
  for (int i = tid; i < n; i += 256) {
    const uint32_t wi = W[i], ni = N[i];
    uint8_t        flags = 0;
    if (wi != kStopped) {
      int  cntW = 0, rankF = 0;
      bool headW = true, headN = true;
      for (int j = 0; j < n; j++) {
        const uint32_t wj = W[j];
        if (wj == kStopped) continue;
        const uint32_t nj = N[j];
        cntW += wj == wi;
        headW = headW && !(wj == wi && j < i);
        headN = headN && !(nj == ni && j < i);
        rankF += (nj < ni) || (nj == ni && j < i);
      }
      flags = (headW ? 1 : 0) | (headN ? 2 : 0);
      fvFeat[(size_t)f * cap + rankF] = (uint32_t)i;
      if (headW) {
        // BowVector::addWeight in feature order: the first feature inserts w, each later one adds w (:34-46)
        const double w = featW[(size_t)f * cap + i];
        double       s = w;
        for (int t = 1; t < cntW; t++) s += w;
        parkVal[(size_t)f * cap + i] = s;  // until the word's rank is known (pass 2)
        atomicAdd(&sWords, 1);
      }
      if (headN) atomicAdd(&sNodes, 1);
      atomicAdd(&sValid, 1);
      flags |= 4;
      // remember the payload rank of a node head: it is the node's offset
      if (headN) parkOff[(size_t)f * cap + i] = rankF;
    }
    H[i] = flags;
  }
  __threadfence_block();
  __syncthreads();
  // pass 2: rank of every head among the heads = its position in the (ascending) output
  for (int i = tid; i < n; i += 256) {
    const uint8_t h = H[i];
    if (!(h & 3)) continue;
    const uint32_t wi = W[i], ni = N[i];
    int            rw = 0, rn = 0;
    for (int j = 0; j < n; j++) {
      const uint8_t hj = H[j];
      rw += (hj & 1) && W[j] < wi;
      rn += (hj & 2) && N[j] < ni;
    }
    if (h & 1) {
      bowWords[(size_t)f * cap + rw] = wi;
      bowVals[(size_t)f * cap + rw]  = parkVal[(size_t)f * cap + i];
    }
    if (h & 2) {
      fvNodes[(size_t)f * cap + rn]     = ni;
      fvOff[(size_t)f * (cap + 1) + rn] = parkOff[(size_t)f * cap + i];
    }
  }
  __threadfence_block();
  __syncthreads();
  // L1 normalisation: the norm is summed in ascending word order by ONE thread (BowVector::normalize, :62-84)
  const int nw = sWords, nn = sNodes;
  if (tid == 0) {
    double norm = 0.0;
    for (int k = 0; k < nw; k++) norm += fabs(bowVals[(size_t)f * cap + k]);
    sNorm                                = norm;
    nWords[f]                            = nw;
    nNodes[f]                            = nn;
    fvOff[(size_t)f * (cap + 1) + nn]    = sValid;
  }
  __syncthreads();
  const double norm = sNorm;
  if (norm > 0.0)
    for (int k = tid; k < nw; k += 256) bowVals[(size_t)f * cap + k] /= norm;
}


// ---- sorted form: one workgroup per frame, everything in LDS -------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan(int v, int* lds /* >= 8 ints */, int* total) {  // 256 threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int       inc  = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int t = lds[w];
    base += w < wave ? t : 0;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// Ascending bitonic sort of NP (power of two) 64-bit keys in LDS by 256 threads.
__device__ __forceinline__ void block_bitonic_sort(unsigned long long* key, int NP) {
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (NP >> 1); t += 256) {
        const int  lo = 2 * t - (t & (j - 1)), hi = lo + j;  // pair (lo, lo + j) of step j
        const bool up = (lo & k) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) key[lo] = b, key[hi] = a;
      }
    }
  }
  __syncthreads();
}

// Sort (id << 32 | feature index) of the frame's non-stopped features; equal ids end up adjacent and in feature order, which
// is the order of the std::map<id, ...> walks of the reference.  `select` picks the id array (words, then nodes).
// headPos (LDS, NP + 1 ints) receives the sorted position of the first feature of every distinct id; returns their number.
__device__ __forceinline__ int sort_and_heads(unsigned long long* key, int* headPos, int* scanLds, int NP, int n, const uint32_t* W,
                                              const uint32_t* ids, int* nValid) {
  for (int p = threadIdx.x; p < NP; p += 256) key[p] = (p < n && W[p] != kStopped) ? (((unsigned long long)ids[p] << 32) | (unsigned)p) : ~0ull;
  block_bitonic_sort(key, NP);
  // heads, in chunks of contiguous positions per thread so that the scan yields ranks in position order
  const int per = NP >> 8;  // positions per thread (NP >= 256)
  const int p0  = threadIdx.x * per;
  int       loc = 0, val = 0;
  for (int p = p0; p < p0 + per; p++) {
    const bool live = key[p] != ~0ull;
    val += live;
    loc += live && (p == 0 || (uint32_t)(key[p] >> 32) != (uint32_t)(key[p - 1] >> 32));
  }
  int       nHeads, nv;
  int       rank = block_excl_scan(loc, scanLds, &nHeads);
  (void)block_excl_scan(val, scanLds, &nv);
  for (int p = p0; p < p0 + per; p++) {
    const bool live = key[p] != ~0ull;
    if (live && (p == 0 || (uint32_t)(key[p] >> 32) != (uint32_t)(key[p - 1] >> 32))) headPos[rank++] = p;
  }
  if (threadIdx.x == 0) headPos[nHeads] = nv;
  __syncthreads();
  *nValid = nv;
  return nHeads;
}

__global__ __launch_bounds__(256) void k_bow_build(const int* __restrict__ counts, int cap, int NP, const uint32_t* __restrict__ featWord,
                                                   const uint32_t* __restrict__ featNode, const double* __restrict__ featW,
                                                   uint32_t* __restrict__ bowWords, double* __restrict__ bowVals, int* __restrict__ nWords,
                                                   uint32_t* __restrict__ fvNodes, int* __restrict__ fvOff, uint32_t* __restrict__ fvFeat,
                                                   int* __restrict__ nNodes) {
  extern __shared__ __align__(16) unsigned char bowLds[];
  unsigned long long* key     = reinterpret_cast<unsigned long long*>(bowLds);
  int*                headPos = reinterpret_cast<int*>(bowLds + (size_t)NP * 8);
  __shared__ int      scanLds[8];
  __shared__ double   sNorm;
  const int           f = blockIdx.x, tid = threadIdx.x;
  const int           n = min(counts ? counts[2 * f] : cap, NP);
  const uint32_t*     W = featWord + (size_t)f * cap;
  const uint32_t*     N = featNode + (size_t)f * cap;
  // ---- BowVector: distinct words ascending; value = the word's weight added once per feature, in feature order
  int       nv;
  const int nw = sort_and_heads(key, headPos, scanLds, NP, n, W, W, &nv);
  for (int r = tid; r < nw; r += 256) {
    const int                p   = headPos[r], cnt = headPos[r + 1] - p;
    const unsigned long long kk  = key[p];
    const double             w   = featW[(size_t)f * cap + (uint32_t)kk];
    double                   sum = w;  // BowVector::addWeight: the first feature inserts w, each later one adds w (:34-46)
    for (int t = 1; t < cnt; t++) sum += w;
    bowWords[(size_t)f * cap + r] = (uint32_t)(kk >> 32);
    bowVals[(size_t)f * cap + r]  = sum;
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {  // BowVector::normalize(L1): summed in ascending word order by one thread (:62-84)
    double norm = 0.0;
    for (int r = 0; r < nw; r++) norm += fabs(bowVals[(size_t)f * cap + r]);
    sNorm     = norm;
    nWords[f] = nw;
  }
  __syncthreads();
  const double norm = sNorm;
  if (norm > 0.0)
    for (int r = tid; r < nw; r += 256) bowVals[(size_t)f * cap + r] /= norm;
  // ---- FeatureVector: features sorted by (node, index) are the payload; the first of every node gives key and offset
  const int nn = sort_and_heads(key, headPos, scanLds, NP, n, W, N, &nv);
  for (int p = tid; p < nv; p += 256) fvFeat[(size_t)f * cap + p] = (uint32_t)key[p];
  for (int r = tid; r <= nn; r += 256) {
    fvOff[(size_t)f * (cap + 1) + r] = headPos[r];
    if (r < nn) fvNodes[(size_t)f * cap + r] = (uint32_t)(key[headPos[r]] >> 32);
  }
  if (tid == 0) nNodes[f] = nn;
}

}  // namespace

struct orbv_vocab {
  int                device = 0;
  hipStream_t        stream = nullptr;
  int                k = 0, L = 0, nNodes = 0, nWords = 0;
  VBuf<uint4>        dDesc;
  VBuf<int>          dChildOff, dChildIds;
  VBuf<uint8_t>      dChildCnt;
  VBuf<uint32_t>     dWordId;
  VBuf<double>       dWeight;
  // workspace of the transform
  VBuf<uint32_t>     wFeatWord, wFeatNode, wBowWords, wFvNodes, wFvFeat;
  VBuf<double>       wFeatW, wBowVals, wParkVal;
  VBuf<uint8_t>      wHead, wDesc;
  VBuf<int>          wFvOff, wNW, wNN, wParkOff;
  VocabDev           dev{};
};

static int vocab_from_arrays(orbv_vocab* h, int k, int L, int n, const int* parent, const uint8_t* is_leaf, const uint8_t* desc,
                             const double* weight) {
  if (k < 1 || k > 32 || L < 1 || n < 0) {
    set_verr("vocabulary: k must be 1..32 (one child per lane of a half wave), L >= 1");
    return -2;
  }
  const int                     nodes = n + 1;
  std::vector<std::vector<int>> children(nodes);
  for (int i = 0; i < n; i++) {
    const int pid = parent[i];
    if (pid < 0 || pid > i) {  // node i + 1 must come after its parent (text format order)
      set_verr("vocabulary: a node is listed before its parent");
      return -2;
    }
    children[pid].push_back(i + 1);  // m_nodes[pid].children.push_back(nid) (TemplatedVocabulary.h:1388)
  }
  std::vector<int>      off(nodes), ids;
  std::vector<uint8_t>  cnt(nodes);
  std::vector<uint32_t> wid(nodes, 0);
  std::vector<double>   wt(nodes, 0.0);
  std::vector<uint8_t>  dsc((size_t)nodes * 32, 0);
  ids.reserve(n);
  int words = 0;
  for (int p = 0; p < nodes; p++) {
    if ((int)children[p].size() > 32) {
      set_verr("vocabulary: a node has more than 32 children");
      return -2;
    }
    off[p] = (int)ids.size();
    cnt[p] = (uint8_t)children[p].size();
    for (int c : children[p]) ids.push_back(c);
  }
  for (int i = 0; i < n; i++) {
    memcpy(&dsc[(size_t)(i + 1) * 32], desc + (size_t)i * 32, 32);
    wt[i + 1] = weight[i];
    if (is_leaf[i]) wid[i + 1] = (uint32_t)words++;  // words number the leaves in file order (:1405-1411)
  }
  h->k = k, h->L = L, h->nNodes = n, h->nWords = words;
  if (h->dDesc.ensure((size_t)nodes * 2) || h->dChildOff.ensure(nodes) || h->dChildCnt.ensure(nodes) || h->dChildIds.ensure(ids.size() + 1) ||
      h->dWordId.ensure(nodes) || h->dWeight.ensure(nodes))
    return -3;
  HIP_TRYV(hipMemcpy(h->dDesc.p, dsc.data(), dsc.size(), hipMemcpyHostToDevice));
  HIP_TRYV(hipMemcpy(h->dChildOff.p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRYV(hipMemcpy(h->dChildCnt.p, cnt.data(), cnt.size(), hipMemcpyHostToDevice));
  if (!ids.empty()) HIP_TRYV(hipMemcpy(h->dChildIds.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRYV(hipMemcpy(h->dWordId.p, wid.data(), wid.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRYV(hipMemcpy(h->dWeight.p, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice));
  h->dev.desc = h->dDesc.p, h->dev.childOff = h->dChildOff.p, h->dev.childCnt = h->dChildCnt.p, h->dev.childIds = h->dChildIds.p;
  h->dev.wordId = h->dWordId.p, h->dev.weight = h->dWeight.p, h->dev.L = L;
  return 0;
}

static void orbv_release_device_side(orbv_vocab* h);
static int vocab_new(orbv_vocab** out, int device_id) {
  *out = nullptr;
  HIP_TRYV(hipSetDevice(device_id));
  orbv_vocab* h = new orbv_vocab();
  h->device     = device_id;
  hipError_t e  = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_verr(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    delete h;
    return -3;
  }
  orbx_internal_register(h, [](void* p) { orbv_release_device_side(static_cast<orbv_vocab*>(p)); });
  *out = h;
  return 0;
}

extern "C" {

const char* orbv_last_error(void) { return g_verr.c_str(); }

static void orbv_release_device_side(orbv_vocab* h) {
  hipSetDevice(h->device);
  if (h->stream) {
    hipStreamSynchronize(h->stream);
    hipStreamDestroy(h->stream);
  }
  h->dDesc.release(), h->dChildOff.release(), h->dChildIds.release(), h->dChildCnt.release(), h->dWordId.release(), h->dWeight.release();
  h->wFeatWord.release(), h->wFeatNode.release(), h->wBowWords.release(), h->wFvNodes.release(), h->wFvFeat.release(), h->wFeatW.release();
  h->wBowVals.release(), h->wParkVal.release(), h->wParkOff.release(), h->wHead.release(), h->wDesc.release(), h->wFvOff.release(), h->wNW.release(), h->wNN.release();
  h->stream = nullptr;
}

void orbv_destroy(orbv_t* h) {
  if (!h) return;
  orbx_internal_unregister(h);
  if (!orbx_internal_runtime_down()) orbv_release_device_side(h);  // else: process teardown, the exit handler (orbx_extractor.hip) has released it
  delete h;
}

int orbv_create(orbv_t** out, int k, int L, int n_nodes, const int* parent, const uint8_t* is_leaf, const uint8_t* desc32,
                const double* weight, int device_id) {
  if (!out || (n_nodes > 0 && (!parent || !is_leaf || !desc32 || !weight))) {
    set_verr("orbv_create: null argument");
    return -2;
  }
  orbv_vocab* h  = nullptr;
  int         rc = vocab_new(&h, device_id);
  if (rc) return rc;
  rc = vocab_from_arrays(h, k, L, n_nodes, parent, is_leaf, desc32, weight);
  if (rc) {
    orbv_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

// TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1337-1420).  Two deliberate differences: blank lines are
// skipped (the reference's `while (!f.eof())` loop turns the empty read after the last newline into one more child of the
// root with an UNINITIALISED descriptor -- undefined behaviour that cannot be reproduced), and only the configuration
// ORB-SLAM3 ships (scoring 0 = L1_NORM, weighting 0 = TF_IDF) is accepted.
int orbv_load_text(orbv_t** out, const char* path, int device_id) {
  if (!out || !path) return -2;
  std::ifstream f(path);
  if (!f.is_open()) {
    set_verr(std::string("cannot open ") + path);
    return -2;
  }
  std::string s;
  std::getline(f, s);
  std::stringstream ss(s);
  int               k = -1, L = -1, n1 = -1, n2 = -1;
  ss >> k >> L >> n1 >> n2;
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {  // :1358-1362
    set_verr("vocabulary text file: bad header");
    return -2;
  }
  if (n1 != 0 || n2 != 0) {
    set_verr("vocabulary text file: only scoring L1_NORM (0) with weighting TF_IDF (0) is supported");
    return -2;
  }
  std::vector<int>     parent;
  std::vector<uint8_t> leaf, desc;
  std::vector<double>  weight;
  while (std::getline(f, s)) {
    if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::stringstream sn(s);
    int               pid = 0, isLeaf = 0;
    sn >> pid >> isLeaf;
    uint8_t d[32];
    for (int i = 0; i < 32; i++) {  // FORB::fromString (FORB.cpp:120-135)
      int b = 0;
      sn >> b;
      d[i] = (uint8_t)b;
    }
    double w = 0;
    sn >> w;
    if (sn.fail()) {
      set_verr("vocabulary text file: malformed node line");
      return -2;
    }
    parent.push_back(pid), leaf.push_back(isLeaf > 0 ? 1 : 0), weight.push_back(w);
    desc.insert(desc.end(), d, d + 32);
  }
  return orbv_create(out, k, L, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), device_id);
}

int orbv_info(const orbv_t* h, int* k, int* L, int* n_nodes, int* n_words) {
  if (!h) return -2;
  if (k) *k = h->k;
  if (L) *L = h->L;
  if (n_nodes) *n_nodes = h->nNodes;
  if (n_words) *n_words = h->nWords;
  return 0;
}

static int ensure_ws(orbv_vocab* h, size_t nframes, size_t cap) {
  const size_t t = nframes * cap;
  if (h->wFeatWord.ensure(t) || h->wFeatNode.ensure(t) || h->wFeatW.ensure(t) || h->wHead.ensure(t) || h->wParkVal.ensure(t) || h->wParkOff.ensure(t))
    return -3;
  return 0;
}

int orbv_transform_batch_device(orbv_t* h, const uint8_t* d_desc, const int* d_counts, int nframes, int cap, int levelsup,
                                uint32_t* d_bow_words, double* d_bow_vals, int* d_n_words, uint32_t* d_fv_nodes, int* d_fv_off,
                                uint32_t* d_fv_feat, int* d_n_nodes, void* stream) {
  if (!h || !d_desc || nframes < 0 || cap < 1 || !d_bow_words || !d_bow_vals || !d_n_words || !d_fv_nodes || !d_fv_off || !d_fv_feat ||
      !d_n_nodes)
    return -2;
  if (nframes == 0) return 0;
  HIP_TRYV(hipSetDevice(h->device));
  int rc = ensure_ws(h, (size_t)nframes, (size_t)cap);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (h->nNodes == 0) {  // empty(): transform returns empty vectors (:1138-1141)
    HIP_TRYV(hipMemsetAsync(d_n_words, 0, sizeof(int) * nframes, s));
    HIP_TRYV(hipMemsetAsync(d_n_nodes, 0, sizeof(int) * nframes, s));
    return 0;
  }
  hipLaunchKernelGGL(k_bow_descend, dim3((cap + 7) / 8, nframes), dim3(256), 0, s, h->dev, d_desc, d_counts, cap, levelsup, h->wFeatWord.p,
                     h->wFeatNode.p, h->wFeatW.p);
  int NP = 256;
  while (NP < cap) NP <<= 1;
  if (NP <= 4096) {  // sorted form: keys + head positions in LDS (12 bytes per slot)
    hipLaunchKernelGGL(k_bow_build, dim3(nframes), dim3(256), (size_t)NP * 12 + 16, s, d_counts, cap, NP, h->wFeatWord.p, h->wFeatNode.p,
                       h->wFeatW.p, d_bow_words, d_bow_vals, d_n_words, d_fv_nodes, d_fv_off, d_fv_feat, d_n_nodes);
  } else {
    hipLaunchKernelGGL(k_bow_build_counting, dim3(nframes), dim3(256), 0, s, d_counts, cap, h->wFeatWord.p, h->wFeatNode.p, h->wFeatW.p,
                       h->wHead.p, h->wParkVal.p, h->wParkOff.p, d_bow_words, d_bow_vals, d_n_words, d_fv_nodes, d_fv_off, d_fv_feat, d_n_nodes);
  }
  HIP_TRYV(hipGetLastError());
  return 0;
}

int orbv_transform(orbv_t* h, const uint8_t* desc32, int n, int levelsup, uint32_t* word_ids, double* word_vals, int* n_words,
                   uint32_t* node_ids, int* node_off, uint32_t* feat_idx, int* n_nodes) {
  if (!h || n < 0 || (n > 0 && !desc32) || !word_ids || !word_vals || !n_words || !node_ids || !node_off || !feat_idx || !n_nodes) return -2;
  *n_words = 0, *n_nodes = 0;
  node_off[0] = 0;
  if (n == 0) return 0;
  HIP_TRYV(hipSetDevice(h->device));
  const size_t cap = (size_t)n;
  if (h->wDesc.ensure(cap * 32) || h->wBowWords.ensure(cap) || h->wBowVals.ensure(cap) || h->wFvNodes.ensure(cap) || h->wFvOff.ensure(cap + 1) ||
      h->wFvFeat.ensure(cap) || h->wNW.ensure(1) || h->wNN.ensure(1))
    return -3;
  HIP_TRYV(hipMemcpyAsync(h->wDesc.p, desc32, cap * 32, hipMemcpyHostToDevice, h->stream));
  int rc = orbv_transform_batch_device(h, h->wDesc.p, nullptr, 1, n, levelsup, h->wBowWords.p, h->wBowVals.p, h->wNW.p, h->wFvNodes.p,
                                       h->wFvOff.p, h->wFvFeat.p, h->wNN.p, h->stream);
  if (rc) return rc;
  int nw = 0, nn = 0;
  HIP_TRYV(hipMemcpyAsync(&nw, h->wNW.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRYV(hipMemcpyAsync(&nn, h->wNN.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRYV(hipStreamSynchronize(h->stream));
  if (nw > 0) {
    HIP_TRYV(hipMemcpy(word_ids, h->wBowWords.p, sizeof(uint32_t) * nw, hipMemcpyDeviceToHost));
    HIP_TRYV(hipMemcpy(word_vals, h->wBowVals.p, sizeof(double) * nw, hipMemcpyDeviceToHost));
  }
  HIP_TRYV(hipMemcpy(node_off, h->wFvOff.p, sizeof(int) * (nn + 1), hipMemcpyDeviceToHost));
  if (nn > 0) HIP_TRYV(hipMemcpy(node_ids, h->wFvNodes.p, sizeof(uint32_t) * nn, hipMemcpyDeviceToHost));
  if (node_off[nn] > 0) HIP_TRYV(hipMemcpy(feat_idx, h->wFvFeat.p, sizeof(uint32_t) * node_off[nn], hipMemcpyDeviceToHost));
  *n_words = nw, *n_nodes = nn;
  return 0;
}

}  // extern "C"
