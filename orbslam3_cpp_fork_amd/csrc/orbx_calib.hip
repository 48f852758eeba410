// orbx_calib.hip -- measured VALU issue ceilings of gfx950 for the instruction classes the ORB kernels are made of.
//
// The HBM roofline is the wrong ceiling for kernels that execute 20-60 integer lane-operations per byte; the right one is the rate
// at which a SIMD issues those instructions.  This file measures it instead of assuming it: one kernel per instruction class issues
// kCalibChains independent dependency chains of that instruction (inline asm, so nothing is folded away), kCalibUnroll x iters times,
// with 1 .. 8 waves resident per SIMD (grid = 256 CUs x k workgroups of 256 threads; a uniform grid is spread evenly).  Reported:
// wave-instructions per nanosecond over the whole chip (HIP events around the launch) and shader cycles per instruction per wave
// (s_memtime inside the kernel, median over waves), from which cycles per instruction per SIMD = cycles per wave / waves per SIMD.
// Debug / tuning hook only (orbx_debug_valu_calib); tools/valu_calibrate.py drives it and writes profiles/r02_valu_calibration.json.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/orbx.h"

extern "C" void orbx_internal_set_error(const char* s);

namespace {

constexpr int kCalibChains = 8;
constexpr int kCalibUnroll = 16;

enum CalibOp { OP_ADD_U32 = 0, OP_PK_MIN_I16, OP_PK_SUB_I16, OP_PERM_B32, OP_BCNT, OP_DOT4_U8, OP_MUL_LO_U32, OP_ALIGNBYTE, OP_MIN3_I32, OP_XOR, OP_MAD_U24, OP_COUNT };
const char* kOpNames[OP_COUNT] = {"v_add_u32", "v_pk_min_i16", "v_pk_sub_i16", "v_perm_b32", "v_bcnt_u32_b32", "v_dot4_u32_u8", "v_mul_lo_u32",
                                  "v_alignbyte_b32", "v_min3_i32", "v_xor_b32", "v_mad_u32_u24"};

template <int OP>
__device__ __forceinline__ void one(uint32_t& a, uint32_t b) {
  if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == OP_PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == OP_PK_SUB_I16) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == OP_PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));
  if (OP == OP_DOT4_U8) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a) : "v"(b));
  if (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == OP_ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a) : "v"(b));
  if (OP == OP_MIN3_I32) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
}

template <int OP>
__global__ __launch_bounds__(256) void k_valu_calib(int iters, uint32_t* sink, unsigned long long* cycles) {
  uint32_t r[kCalibChains];
#pragma unroll
  for (int c = 0; c < kCalibChains; c++) r[c] = threadIdx.x * 2654435761u + c;
  const uint32_t b = threadIdx.x | 0x01010101u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < kCalibUnroll; u++)
#pragma unroll
      for (int c = 0; c < kCalibChains; c++) one<OP>(r[c], b);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t                 x  = 0;
#pragma unroll
  for (int c = 0; c < kCalibChains; c++) x ^= r[c];
  if (x == 0x12345u) *sink = x;  // keeps the chains alive
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void launch(int blocks, int iters, uint32_t* sink, unsigned long long* cyc, hipStream_t s) {
  hipLaunchKernelGGL(k_valu_calib<OP>, dim3(blocks), dim3(256), 0, s, iters, sink, cyc);
}

}  // namespace

extern "C" {

// op: index into the list below; waves_per_simd 1 .. 8; out = {nanoseconds of the launch (HIP events, best of 5), median shader cycles
// per wave, wave-instructions per wave}.  Returns 0, -2 (bad arguments), -3 (HIP error).
int orbx_debug_valu_calib(int op, int waves_per_simd, int iters, double out[3]) {
  if (op < 0 || op >= OP_COUNT || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !out) return -2;
  int           ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
  const int           blocks = ncu * waves_per_simd;  // 4 waves per workgroup = one per SIMD
  uint32_t*           sink   = nullptr;
  unsigned long long* cyc    = nullptr;
  hipEvent_t          e0, e1;
  if (hipMalloc((void**)&sink, 4) != hipSuccess || hipMalloc((void**)&cyc, (size_t)blocks * 4 * 8) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess) {
    orbx_internal_set_error("orbx_debug_valu_calib: allocation failed");
    return -3;
  }
  void (*fn[OP_COUNT])(int, int, uint32_t*, unsigned long long*, hipStream_t) = {
      launch<0>, launch<1>, launch<2>, launch<3>, launch<4>, launch<5>, launch<6>, launch<7>, launch<8>, launch<9>, launch<10>};
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0, nullptr);
    fn[op](blocks, iters, sink, cyc, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  out[0] = (double)best * 1e6;
  out[1] = (double)h[h.size() / 2];
  out[2] = (double)iters * kCalibUnroll * kCalibChains;
  hipFree(sink), hipFree(cyc), hipEventDestroy(e0), hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char* orbx_debug_valu_calib_name(int op) { return op >= 0 && op < OP_COUNT ? kOpNames[op] : nullptr; }

}  // extern "C"
