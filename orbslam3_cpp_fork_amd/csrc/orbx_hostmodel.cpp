// orbx_hostmodel.cpp -- host builds of the shared host/device algorithm cores, exported for CPU tests.
// These entry points run the *same source* the kernels compile (octree_core.h, orb_sincosf_port.h) on
// the CPU, so that the logic can be checked against the oracle without a GPU.  They are debug hooks,
// not part of the product path: orbx_extract* never calls them.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "octree_core.h"
#include "orb_sincosf_port.h"

extern "C" {

// DistributeOctTree host model.  cand: packed candidates (x | y<<12 | resp<<24) in feed order.
// Returns 0, or -2 if nIni == 0.  out must hold max(N + 3, 4 * nIni) + 8 entries.
int orbx_debug_octree_host(const uint32_t* cand, int n, int W, int H, int N, uint32_t* out, int* out_count) {
  using namespace orbx;
  const int nIni = (int)__builtin_roundf((float)W / (float)H);
  if (nIni <= 0) return -2;
  OtArgs a;
  memset(&a, 0, sizeof(a));
  const int maxNodes = (N + 3 > 4 * nIni ? N + 3 : 4 * nIni) + 8;
  if (maxNodes > kOtMaxNodes) return -2;
  std::vector<uint16_t>           nA(n + 1);
  std::vector<OtNode>             nodesA(maxNodes), nodesB(maxNodes);
  std::vector<Int4>               sc(maxNodes);
  std::vector<Idx4>               ec(maxNodes);
  std::vector<OtIdx>              D(maxNodes), dI(maxNodes), pO(maxNodes), mO(maxNodes), kR(maxNodes), vA(maxNodes), vB(maxNodes);
  std::vector<uint8_t>            nCh(maxNodes);
  std::vector<unsigned long long> sb(maxNodes), bst(maxNodes);
  a.cand = cand, a.n = n, a.W = W, a.H = H, a.N = N, a.nIni = nIni, a.hX = (float)W / nIni, a.maxNodes = maxNodes;
  a.nodeOf = nA.data(), a.best = bst.data();
  a.nodesA = nodesA.data(), a.nodesB = nodesB.data(), a.childCnt = sc.data(), a.childIdx = ec.data();
  a.D = D.data(), a.dIndex = dI.data(), a.nCh = nCh.data(), a.pushOff = pO.data(), a.multiOff = mO.data();
  a.keptRank = kR.data(), a.vA = vA.data(), a.vB = vB.data(), a.sortbuf = sb.data();
  a.outKp = out, a.outCount = out_count;
  HostCtx c;
  ot_distribute(c, a);
  return 0;
}

// libstdc++ std::sort re-enactment on (key << 16 | payload) records.
void orbx_debug_std_sort(unsigned long long* buf, int n, int depth_limit) { orbx::ot_std_sort(buf, n, depth_limit); }

// glibc sinf/cosf port (plain-double evaluation order) for x = deg * (float)(pi/180).
void orbx_debug_sincos_deg(const float* deg, int n, float* cos_out, float* sin_out) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  for (int i = 0; i < n; i++) orbport::sincosf_port<false>(deg[i] * factorPI, &sin_out[i], &cos_out[i]);
}
}
