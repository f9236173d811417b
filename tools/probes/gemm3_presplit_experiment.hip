// NOT BUILT INTO THE LIBRARY -- kept as the record of a measured alternative ("Why not 2.7x").
// The fp32 GEMMs of the split arithmetic with operands split ONCE by their producer and kept as three bf16 planes in HBM,
// fed to the matrix cores by global -> LDS DMA (256x128 tiles on 8 waves).  Result on the 13x13 1024->1024 position GEMMs:
// 0.50 ms, the same as conv_gemm_kernel<..., SPLIT> (0.49 ms), which splits in the staging registers and needs none of the
// producer-side changes; the board runs at its power limit in both.
// fp32 GEMMs on the bf16 matrix cores at fp32 accuracy ("split" operands).
//
// An fp32 value is carried as three bfloat16 planes  x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 3 x 8 significant bits = the 24 of fp32, both residuals are exact.  A product a*b is accumulated
// in fp32 (v_mfma_f32_32x32x16_bf16) from the six cross terms down to 2^-16 relative,
//     a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1;
// the three dropped terms (a2b3, a3b2, a3b3) are <= 2^-24 |ab| each, below the rounding of one fp32 product.  Measured
// against a double-precision sum (tools/probes/split_gemm_probe.hip, profiles/r03_split_probe.txt): relative L2 error
// 0.98e-6 at K = 4608 against 1.20e-6 for the native fp32 MFMA (v_mfma_f32_32x32x2_f32), identical to all nine terms.
// Six bf16 MFMAs cost 6/16 of the fp32 MFMAs they replace: 2.48 PFLOP/s against 138 TFLOP/s measured issue rate.
//
// The split is done ONCE, by whoever produces the operand (the Winograd transforms, the weight transform): the planes
// live in HBM as  [row][K/32][3][32] bf16  -- per row, per 32-wide k-chunk, the three planes back to back (192 bytes), so a
// k-chunk of all three planes is one contiguous run for the global -> LDS DMA.  6 bytes per element instead of 4.
//
// Kernels here:
//   split3_pack_kernel   fp32 [rows][K] -> planes (tests, and operands that have no fused producer)
//   gemm3_kernel         y[b] = A[b] * B[b]^T, both operands K-major planes, fp32 result      (Winograd position GEMMs)
//   wgrad3_kernel        dU[b] = sum_t Wt[b][t][:]^T V[b][t][:], operands as they lie ([t][channels] planes), the
//                        fragments come out of LDS through the transposing read             (Winograd weight gradient)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 32;              // k elements per chunk
constexpr int kChunk3 = 3 * kChunk;     // bf16 elements of one row-chunk (three planes): 192 bytes = 12 pieces of 16 bytes

__device__ __attribute__((aligned(16))) u16 g_zero16[8];     // zero source of the DMA (rows past the end of a reduction)

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// LDS fragment reads as inline assembly: a ds_read that FOLLOWS a global_load_lds in program order makes the compiler wait
// for every outstanding DMA first (it cannot prove the two do not alias), which would serialise the prefetch of the next
// stage with the reads of this one.  The price: the waits for these reads are ours (lgkm_fence below).
__device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

__device__ __forceinline__ void split3(const f32x4& v, uint2& h, uint2& m, uint2& l) {
  u16 hs[4], ms[4], ls[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 b1 = (__bf16)v[i];
    const float r1 = v[i] - (float)b1;
    const __bf16 b2 = (__bf16)r1;
    const float r2 = r1 - (float)b2;
    const __bf16 b3 = (__bf16)r2;
    hs[i] = __builtin_bit_cast(u16, b1);
    ms[i] = __builtin_bit_cast(u16, b2);
    ls[i] = __builtin_bit_cast(u16, b3);
  }
  h = make_uint2((unsigned)hs[0] | ((unsigned)hs[1] << 16), (unsigned)hs[2] | ((unsigned)hs[3] << 16));
  m = make_uint2((unsigned)ms[0] | ((unsigned)ms[1] << 16), (unsigned)ms[2] | ((unsigned)ms[3] << 16));
  l = make_uint2((unsigned)ls[0] | ((unsigned)ls[1] << 16), (unsigned)ls[2] | ((unsigned)ls[3] << 16));
}

// one thread: 4 consecutive k of one row
__global__ __launch_bounds__(256) void split3_pack_kernel(const float* __restrict__ x, long long x_ld, u16* __restrict__ out,
                                                          long long rows, int K) {
  const int k4 = K >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * k4) return;
  const long long row = idx / k4;
  const int k = (int)(idx - row * k4) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * x_ld + k);
  uint2 h, m, l;
  split3(v, h, m, l);
  u16* dst = out + row * (3LL * K) + (k >> 5) * kChunk3 + (k & 31);
  *reinterpret_cast<uint2*>(dst) = h;
  *reinterpret_cast<uint2*>(dst + kChunk) = m;
  *reinterpret_cast<uint2*>(dst + 2 * kChunk) = l;
}

struct Gemm3Args {
  const u16* a;        // planes [M][K/32][3][32]
  const u16* b;        // planes [N rounded up to 128][K/32][3][32]
  float* y;            // [M][y_ld]
  long long a_bs, b_bs, y_bs;      // element strides between batches
  long long y_ld;
  int M, N, K;
  int m_tiles, n_tiles;
  int b_rows;          // rows that exist in b (>= n_tiles * BN)
  int dbg;             // timing experiments (FSD_G3_DEBUG): 1 = no DMA after the prologue, 2 = no fragment reads, 4 = no barrier
};

// LDS image of one stage: the (BM + BN) row-chunks back to back, 12 sixteen-byte slots per row, written linearly by the
// DMA.  A 16-lane service group of ds_read_b128 (MI355X_MICROARCH.md: lanes {0-3, 12-15, 20-27}, ... -- every value of
// row & 15 once) reads ONE piece index of 16 rows; with 192-byte rows the 256-byte bank row repeats every 4 rows, so the
// slot of a piece is rotated by (row >> 2) & 3 inside its row:  slot = (piece + ((row >> 2) & 3)) % 12  -- the 16 rows of
// a group then cover all 16 sixteen-byte bank slots.  The rotation is applied on the FETCH side (which global piece a
// lane asks for), the contiguous 192-byte run per row is unchanged.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm3_kernel(Gemm3Args p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
  constexpr int A_INSTR = BM * 12 / 64, B_INSTR = BN * 12 / 64;       // 1-KiB DMA instructions per stage
  constexpr int A_PER = (A_INSTR + NW - 1) / NW, B_PER = (B_INSTR + NW - 1) / NW;      // (the last round may be partial)
  constexpr int STAGE = (BM + BN) * kChunk3;                          // bf16 elements per stage
  extern __shared__ __attribute__((aligned(16))) u16 smem_g[];

  // XCD-aware order over the FLAT (batch, tile) space: workgroups are dealt round-robin to the 8 XCDs in dispatch order,
  // so XCD c takes the c-th eighth of the flat space and walks it batch by batch -- the tiles of one batch share one L2:
  // with six bytes per operand element and a third of the native kernel's time per tile, re-fetching a batch's operands
  // into several L2s (the per-batch remap of conv_gemm_kernel) makes the kernel wait for HBM / MALL.
  const int tiles = gridDim.x;
  const int Lf = xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int batch = Lf / tiles, L = Lf - batch * tiles;
  const u16* a = p.a + (long long)batch * p.a_bs;
  const u16* b = p.b + (long long)batch * p.b_bs;
  const int mt = L / p.n_tiles, nt = L - mt * p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const unsigned row_ld = (unsigned)p.K * 3u;

  unsigned a_off[A_PER], b_off[B_PER];        // element offsets of this lane's pieces at k-chunk 0 (launcher: < 2^32)
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int s = (i * NW + wave) * 64 + lane;
    const int row = s / 12, j = s - row * 12;
    int piece = j - ((row >> 2) & 3);
    piece += piece < 0 ? 12 : 0;
    int g = m0 + row;
    g = g < p.M ? g : p.M - 1;                // rows past M: any valid row (their results are not stored)
    a_off[i] = (unsigned)g * row_ld + (unsigned)piece * 8u;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int s = (i * NW + wave) * 64 + lane;
    const int row = s / 12, j = s - row * 12;
    int piece = j - ((row >> 2) & 3);
    piece += piece < 0 ? 12 : 0;
    int g = n0 + row;
    g = g < p.b_rows ? g : p.b_rows - 1;
    b_off[i] = (unsigned)g * row_ld + (unsigned)piece * 8u;
  }
  // piece i of this wave's share of a stage (A pieces first).  Issuing one costs the wave 60-180 cycles
  // (MI355X_MICROARCH.md), so the pieces of the NEXT stage are dealt out between the MFMA groups of this one instead of
  // being issued in a burst behind the barrier, where both waves of a SIMD would leave the matrix pipe idle.
  constexpr int PIECES = A_PER + B_PER;
  auto piece = [&](int i, int kc, u16* st) {
    if (p.dbg & 1) return;
    const unsigned ko = (unsigned)kc * kChunk3;
    if (i < A_PER) {
      if (A_INSTR % NW == 0 || i * NW + wave < A_INSTR) dma16(a + (a_off[i] + ko), st + (i * NW + wave) * 512);
    } else if (i < PIECES) {
      const int ib = i - A_PER;
      if (B_INSTR % NW == 0 || ib * NW + wave < B_INSTR) dma16(b + (b_off[ib] + ko), st + BM * kChunk3 + (ib * NW + wave) * 512);
    }
  };
  auto stage = [&](int kc, u16* st) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece(i, kc, st);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: lane = tile row (lane & 31) and k half (lane >> 5) of a 16-wide MFMA step; piece = plane * 4 +
  // step * 2 + half, rotated like the fetch side.  (row >> 2) & 3 == (lane >> 2) & 3: tile rows start at multiples of 32.
  const int rot = (lane >> 2) & 3, half = lane >> 5;
  unsigned po[3][2];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int s = 0; s < 2; ++s) po[q][s] = (unsigned)(((q * 4 + s * 2 + half + rot) % 12) * 16);
  const unsigned smem_base = lds_addr(smem_g);
  const unsigned sa0 = (unsigned)((wm * TM * 32 + (lane & 31)) * kChunk3 * 2);
  const unsigned sb0 = (unsigned)((BM + wn * TN * 32 + (lane & 31)) * kChunk3 * 2);

  bf16x8 af[2][3][TM], bf[2][3][TN];
  auto frags = [&](unsigned st_addr, int s, int buf) {
    if (p.dbg & 2) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[buf][q][i] = lds_read128(st_addr + sa0 + (unsigned)(i * 32 * kChunk3 * 2) + po[q][s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[buf][q][j] = lds_read128(st_addr + sb0 + (unsigned)(j * 32 * kChunk3 * 2) + po[q][s]);
    }
  };
  // wait for every outstanding LDS read; the fragment registers are operands so that nothing that uses them moves above
  auto lgkm_fence = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[buf][q][i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[buf][q][j]));
    }
  };
  // the six terms of one k-step, smallest first; each term sweeps the TM x TN independent accumulators.  After term t the
  // wave issues the DMA pieces [pc0 + t * PPT, pc0 + (t + 1) * PPT) of the next stage (more == false: nothing left to fetch).
  constexpr int PPT = (PIECES + 5) / 6;              // pieces per term slot (dealt out over the six terms of the second k-step)
  auto mfmas = [&](int buf, int pc0, bool more, int kc_next, u16* st_next) {
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][TA[t]][i], bf[buf][TB[t]][j], acc[i][j], 0, 0, 0);
      if (more) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) piece(pc0 + t * PPT + q, kc_next, st_next);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Schedule of one stage (two 16-wide k-steps, fragment registers R0 / R1, ONE barrier):
  //   MFMAs(step 0)  with the step-1 fragment reads in flight
  //   barrier        every wave has read this stage into registers -> its LDS buffer is free; the next stage (DMA issued
  //                  one stage ago) has landed -> published
  //   MFMAs(step 1)  with the next stage's step-0 fragment reads in flight and the DMA pieces of the stage after next
  //                  dealt out between the MFMA groups, into the buffer just freed
  // so a DMA piece has a full stage of matrix work (~3000 cycles) to land and no MFMA group waits for an LDS read issued
  // behind a barrier.
  const int nk = p.K / kChunk;
  stage(0, smem_g);
  if (nk > 1) stage(1, smem_g + STAGE);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  frags(smem_base, 0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    const int cur = kc & 1;
    const unsigned st_addr = smem_base + (unsigned)(cur * STAGE * 2), nx_addr = smem_base + (unsigned)((cur ^ 1) * STAGE * 2);
    lgkm_fence(0);
    frags(st_addr, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0, 0, false, 0, nullptr);
    lgkm_fence(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage kc + 1 have landed
    if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();
    if (kc + 1 < nk) frags(nx_addr, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1, 0, kc + 2 < nk, kc + 2, smem_g + cur * STAGE);
  }
  __syncthreads();
  ConvArgs e;
  e.y = p.y + (long long)batch * p.y_bs;
  e.y_ld = p.y_ld;
  e.bias = nullptr;
  e.bn_partial = nullptr;
  e.M = p.M;
  e.Cout = p.N;
  e.slope = 1.f;
  e.wide = 1;
  e.part_base = 0;
  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN, false>(e, acc, reinterpret_cast<float*>(smem_g), m0, n0, mt, tid, lane, wm, wn);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int BM, int BN, int WM, int WN>
int launch_gemm3(Gemm3Args a, int batches, hipStream_t stream) {
  a.m_tiles = (a.M + BM - 1) / BM;
  a.n_tiles = (a.N + BN - 1) / BN;
  size_t lds = 2 * (size_t)(BM + BN) * kChunk3 * sizeof(u16);
  const size_t tile_bytes = (size_t)BM * BN * sizeof(float);
  if (lds < tile_bytes) lds = tile_bytes;
  auto k = gemm3_kernel<BM, BN, WM, WN>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  // work figure: the algorithmic fp32 FLOPs of the product (six bf16 MFMA terms each)
  fsd_prof::Scope prof(fsd_prof::kGemmSplit, 2.0 * a.m_tiles * BM * (double)(a.n_tiles * BN) * a.K * batches, stream);
  hipLaunchKernelGGL(k, dim3(a.m_tiles * a.n_tiles, batches), dim3(WM * WN * 64), lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

int fsd_conv::gemm3_batched(const unsigned short* a3, long long a_bs, const unsigned short* b3, long long b_bs, float* y,
                            long long y_ld, long long y_bs, long long rows, int k, int cout, int batches, hipStream_t stream) {
  if (k % kChunk != 0 || rows < 1 || cout < 1 || (cout & 3) || (y_ld & 3) || (y_bs & 3)) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(a3) & 15) || (reinterpret_cast<uintptr_t>(b3) & 15))
    return FSD_ERR_ARG;
  const int b_rows = round_up(cout, 128);
  if ((rows + 1) * 3LL * k >= 0xffffffffLL || (long long)(b_rows + 1) * 3LL * k >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;
  Gemm3Args a;
  a.a = a3; a.b = b3; a.y = y;
  a.a_bs = a_bs; a.b_bs = b_bs; a.y_bs = y_bs; a.y_ld = y_ld;
  a.M = (int)rows; a.N = cout; a.K = k;
  a.b_rows = b_rows;
  static const char* dbg_env = getenv("FSD_G3_DEBUG");
  a.dbg = dbg_env ? atoi(dbg_env) : 0;
  static const char* env = getenv("FSD_GEMM3_TILE");
  const char pick = env ? env[0] : (cout <= 64 ? 'b' : 'a');
  if (pick == 'b') return launch_gemm3<256, 64, 4, 2>(a, batches, stream);
  if (pick == 'c') return launch_gemm3<128, 128, 2, 2>(a, batches, stream);
  return launch_gemm3<256, 128, 4, 2>(a, batches, stream);
}

extern "C" size_t fsd_split3_elems(long long rows, int k) { return (size_t)rows * 3 * (size_t)k; }

extern "C" int fsd_split3_pack(const float* x, long long x_ld, void* planes, long long rows, int k, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !planes || rows < 1 || k < 32 || k % kChunk != 0 || (x_ld & 3) || x_ld < k) return FSD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return FSD_ERR_ARG;
  const long long n = rows * (k / 4);
  hipLaunchKernelGGL(split3_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, x_ld,
                     static_cast<u16*>(planes), rows, k);
  return (int)hipGetLastError();
}

extern "C" int fsd_gemm3(const void* a_planes, const void* b_planes, float* y, long long y_ld, long long rows, int k,
                         int cout, int batches, hipStream_t stream) {
  (void)hipGetLastError();
  if (!a_planes || !b_planes || !y || batches < 1) return FSD_ERR_ARG;
  return fsd_conv::gemm3_batched(static_cast<const u16*>(a_planes), rows * 3LL * k, static_cast<const u16*>(b_planes),
                                 (long long)round_up(cout, 128) * 3LL * k, y, y_ld, rows * y_ld, rows, k, cout, batches, stream);
}
