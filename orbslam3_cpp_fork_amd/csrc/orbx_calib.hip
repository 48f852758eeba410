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

enum CalibOp { OP_COUNT = 70 };
const char* kOpNames[OP_COUNT] = {"v_add_u32", "v_pk_min_i16", "v_pk_sub_i16", "v_perm_b32", "v_bcnt_u32_b32", "v_dot4_u32_u8", "v_mul_lo_u32", "v_alignbyte_b32", "v_min3_i32", "v_xor_b32", "v_mad_u32_u24", "v_min_u16_sdwa", "v_lshlrev_b32(const)", "v_and_b32(sgpr)", "v_pk_min_u16", "v_pk_sub_u16_clamp", "v_and_or_b32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32(vgpr)", "v_lshrrev_b32(const)", "v_min_u32", "v_max_i32", "v_max_u16", "v_sub_u16", "v_sub_u32", "v_add_u32(sgpr)", "v_add_u32(const)", "v_xor_b32(sgpr)", "v_cndmask_b32", "v_mul_u32_u24", "v_bfe_u32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_sad_u8", "v_max_u16_sdwa(word)", "v_mov_b32_sdwa", "v_pk_add_u16", "v_pk_lshlrev_b16", "v_add_f32", "v_fma_f32", "v_cmp_lt_u32+cndmask", "v_min_u16", "v_add_u16", "v_max_i16", "v_lshlrev_b16", "v_lshrrev_b16", "v_ashrrev_i32", "v_not_b32", "v_subrev_u32", "v_max_u32", "v_min_i32", "v_mul_f32", "v_fmac_f32", "v_mul_hi_u32", "v_alignbit_b32", "v_bfi_b32", "v_or3_b32", "v_bitop3_b32", "v_mbcnt_lo_u32_b32", "v_cmp_lt_u32(vcc)", "v_cvt_f32_u32", "v_add_u32(literal)", "v_and_b32(literal)", "v_ffbl_b32", "v_bcnt(vop3 const)", "v_mov_b32_dpp", "v_readfirstlane+mov"};

template <int OP>
__device__ __forceinline__ void one(uint32_t& a, uint32_t b) {
  if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 1) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 2) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));
  if (OP == 5) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a) : "v"(b));
  if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 7) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a) : "v"(b));
  if (OP == 8) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 11) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "+v"(a) : "v"(b));
  if (OP == 12) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 13) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "s"(0x7fff7fffu));
  if (OP == 14) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 15) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a) : "v"(b));
  if (OP == 16) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
  if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 19) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 20) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));
  if (OP == 21) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 22) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 23) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 24) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 25) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 27) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(0x7fff7fffu));
  if (OP == 28) asm volatile("v_add_u32 %0, 3, %0" : "+v"(a) : "v"(b));
  if (OP == 29) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "s"(0x7fff7fffu));
  if (OP == 30) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
  if (OP == 31) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 32) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a) : "v"(b));
  if (OP == 33) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 34) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
  if (OP == 35) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
  if (OP == 36) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 37) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a) : "v"(b));
  if (OP == 38) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(a) : "v"(b));
  if (OP == 39) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 40) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 41) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 42) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 43) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
  if (OP == 44) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 45) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 46) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 47) asm volatile("v_lshlrev_b16 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 48) asm volatile("v_lshrrev_b16 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 49) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a) : "v"(b));
  if (OP == 50) asm volatile("v_not_b32 %0, %1" : "+v"(a) : "v"(b));
  if (OP == 51) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 52) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 53) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 54) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 55) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 56) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 57) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a) : "v"(b));
  if (OP == 58) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 59) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (OP == 60) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x80" : "+v"(a) : "v"(b));
  if (OP == 61) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));
  if (OP == 62) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc");
  if (OP == 63) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a) : "v"(b));
  if (OP == 64) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a) : "v"(b));
  if (OP == 65) asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(a) : "v"(b));
  if (OP == 66) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a) : "v"(b));
  if (OP == 67) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(a) : "v"(b));
  if (OP == 68) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
  if (OP == 69) asm volatile("v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20" : "+v"(a) : "v"(b) : "s20");
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

// LDS read issue rate: every lane reads kLdsReads dwords (or 8 / 16 bytes) per iteration at `lane * width + off + k * pitch`; with off != 0 the
// reads are not naturally aligned.  The kernel also checks what the hardware returned against the bytes it wrote (ok = 0 when an
// unaligned read does not return bytes [addr, addr + width)).
template <int WIDTH>
__global__ __launch_bounds__(256) void k_lds_calib(int iters, int off, uint32_t* sink, unsigned long long* cycles, int* bad) {
  __shared__ __align__(16) uint8_t lds[32768];
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (uint8_t)(i * 7 + (i >> 8));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;  // LDS byte address of the array (keeps the init stores alive)
  const uint32_t base = lds0 + (uint32_t)(wave * 8192 + lane * WIDTH + off);
  uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  // correctness of one read
  {
    uint32_t r[4] = {0, 0, 0, 0};
    if (WIDTH == 4) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r[0]) : "v"(base) : "memory");
    if (WIDTH == 8) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<unsigned long long*>(r)) : "v"(base) : "memory");
    for (int k = 0; k < WIDTH; k++) {
      const int a = (int)(base - lds0) + k;
      if (((r[k >> 2] >> (8 * (k & 3))) & 0xff) != (uint8_t)(a * 7 + (a >> 8))) atomicAdd(bad, 1);
    }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (WIDTH == 4) {
        uint32_t a, b, c, d;
        asm volatile("ds_read_b32 %0, %4 offset:0\n ds_read_b32 %1, %4 offset:512\n ds_read_b32 %2, %4 offset:1024\n ds_read_b32 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(base + u * 1024)
                     : "memory");
        acc0 ^= a, acc1 ^= b, acc2 ^= c, acc3 ^= d;
      } else {
        unsigned long long a, b, c, d;
        asm volatile("ds_read_b64 %0, %4 offset:0\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(base + u * 1024)
                     : "memory");
        acc0 ^= (uint32_t)a ^ (uint32_t)(a >> 32), acc1 ^= (uint32_t)b, acc2 ^= (uint32_t)c, acc3 ^= (uint32_t)(d >> 32);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((acc0 ^ acc1 ^ acc2 ^ acc3) == 0x12345u) *sink = acc0;
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
  void (*fn[OP_COUNT])(int, int, uint32_t*, unsigned long long*, hipStream_t) = {launch<0>, launch<1>, launch<2>, launch<3>, launch<4>, launch<5>, launch<6>, launch<7>, launch<8>, launch<9>, launch<10>, launch<11>, launch<12>, launch<13>, launch<14>, launch<15>, launch<16>, launch<17>, launch<18>, launch<19>, launch<20>, launch<21>, launch<22>, launch<23>, launch<24>, launch<25>, launch<26>, launch<27>, launch<28>, launch<29>, launch<30>, launch<31>, launch<32>, launch<33>, launch<34>, launch<35>, launch<36>, launch<37>, launch<38>, launch<39>, launch<40>, launch<41>, launch<42>, launch<43>, launch<44>, launch<45>, launch<46>, launch<47>, launch<48>, launch<49>, launch<50>, launch<51>, launch<52>, launch<53>, launch<54>, launch<55>, launch<56>, launch<57>, launch<58>, launch<59>, launch<60>, launch<61>, launch<62>, launch<63>, launch<64>, launch<65>, launch<66>, launch<67>, launch<68>, launch<69>};
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

// width 4 / 8 bytes per lane, byte offset `off` added to every lane's address (0 = aligned).  out = {ns of the launch, median shader cycles
// per wave, LDS read instructions per wave, lanes whose unaligned read returned other bytes than [addr, addr + width)}.
int orbx_debug_lds_calib(int width, int off, int waves_per_simd, int iters, double out[4]) {
  if ((width != 4 && width != 8) || off < 0 || off > 64 || waves_per_simd < 1 || waves_per_simd > 4 || iters < 1 || !out) return -2;
  int             ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
  const int           blocks = ncu * waves_per_simd;
  uint32_t*           sink   = nullptr;
  unsigned long long* cyc    = nullptr;
  int*                bad    = nullptr;
  hipEvent_t          e0, e1;
  if (hipMalloc((void**)&sink, 4) != hipSuccess || hipMalloc((void**)&bad, 4) != hipSuccess || hipMalloc((void**)&cyc, (size_t)blocks * 4 * 8) != hipSuccess ||
      hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    orbx_internal_set_error("orbx_debug_lds_calib: allocation failed");
    return -3;
  }
  hipMemset(bad, 0, 4);
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0, nullptr);
    if (width == 4)
      hipLaunchKernelGGL(k_lds_calib<4>, dim3(blocks), dim3(256), 0, nullptr, iters, off, sink, cyc, bad);
    else
      hipLaunchKernelGGL(k_lds_calib<8>, dim3(blocks), dim3(256), 0, nullptr, iters, off, sink, cyc, bad);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0) best = std::min(best, ms);
  }
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  int hbad = 0;
  hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  out[0] = (double)best * 1e6;
  out[1] = (double)h[h.size() / 2];
  out[2] = (double)iters * 16;
  out[3] = (double)hbad;
  hipFree(sink), hipFree(cyc), hipFree(bad), hipEventDestroy(e0), hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char* orbx_debug_valu_calib_name(int op) { return op >= 0 && op < OP_COUNT ? kOpNames[op] : nullptr; }

}  // extern "C"
